"""Minimal stand-ins for the ``lightning.pytorch`` pieces the reference touches (LightningModule,
LightningDataModule), used when Lightning is not installed (it is absent from the MI355X image).  If
``lightning.pytorch`` is importable the real classes are used instead, so ``GAPartNet`` drops into
``gapartnet/train.py`` (LightningCLI) unchanged; ``gapartnet_amd.trainer.Trainer`` drives either.
"""
import inspect
from typing import Any, Dict, Optional

import torch
import torch.nn as nn

try:  # pragma: no cover - depends on the environment
    import lightning.pytorch as _lp
    HAVE_LIGHTNING = True
except Exception:  # ModuleNotFoundError and friends
    _lp = None
    HAVE_LIGHTNING = False


class _AttrDict(dict):
    def __getattr__(self, k):
        if k.startswith("__"):
            raise AttributeError(k)
        return self.get(k)

    def __setattr__(self, k, v):
        self[k] = v


def _caller_init_args(depth: int = 2) -> Dict[str, Any]:
    frame = inspect.currentframe()
    for _ in range(depth):
        frame = frame.f_back
    args, _, _, values = inspect.getargvalues(frame)
    return {a: values[a] for a in args if a != "self"}


class _LiteLightningModule(nn.Module):
    """the hooks/attributes GAPartNet uses: save_hyperparameters, hparams, log, current_epoch, device, trainer."""

    def __init__(self):
        super().__init__()
        self._hparams = _AttrDict()
        self.trainer = None
        self._current_epoch = 0
        self._log_sink = None  # set by the Trainer: callable(name, value, batch_size, sync_dist)

    def save_hyperparameters(self, *_, **__):
        self._hparams = _AttrDict(_caller_init_args(depth=2))

    @property
    def hparams(self):
        return self._hparams

    @property
    def current_epoch(self) -> int:
        return self._current_epoch

    @property
    def device(self) -> torch.device:
        for p in self.parameters():
            return p.device
        return torch.device("cpu")

    def log(self, name: str, value, batch_size: Optional[int] = None, on_epoch: bool = True, prog_bar: bool = False,
            logger: bool = True, sync_dist: bool = False, **_):
        if self._log_sink is not None:
            self._log_sink(name, value, batch_size, sync_dist)

    # hooks the Trainer calls if present
    def configure_optimizers(self):
        raise NotImplementedError


class _LiteLightningDataModule:
    def __init__(self):
        self._hparams = _AttrDict()

    def save_hyperparameters(self, *_, **__):
        self._hparams = _AttrDict(_caller_init_args(depth=2))

    @property
    def hparams(self):
        return self._hparams

    def setup(self, stage: Optional[str] = None):
        pass


LightningModule = _lp.LightningModule if HAVE_LIGHTNING else _LiteLightningModule
LightningDataModule = _lp.LightningDataModule if HAVE_LIGHTNING else _LiteLightningDataModule
