"""Mirror of the ``spconv`` package surface used by GAPartNet (``import spconv.pytorch as spconv``)."""
from . import pytorch  # noqa: F401
