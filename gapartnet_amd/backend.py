"""Which raw-operator module the host-side wrappers call.

The default — and the only one the product ships — is ``gapartnet_amd.hip_ops`` (libgpn_hip.so on a MI355X).
It is never replaced automatically: a missing HIP extension or a CPU tensor raises.  ``use(...)`` exists so that
``tests/`` and ``bench.py``'s ``cpu_baseline`` leg can run the same host glue over the CPU oracle
(``oracle.torch_ops``); nothing inside this package calls it.
"""
import contextlib

from . import hip_ops as _hip

_current = _hip


def raw():
    return _current


def use(module):
    """Install another raw-op module (tests / cpu_baseline only). Returns the previous one."""
    global _current
    prev, _current = _current, module
    return prev


@contextlib.contextmanager
def using(module):
    prev = use(module)
    try:
        yield
    finally:
        use(prev)
