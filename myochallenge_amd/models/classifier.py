"""Task classifier of the winning ensemble (/root/reference/src/models/classifier.py:160-174).

A 234 -> 200 -> 100 -> 1 ReLU MLP on the first N_OBS_PER_TRIAL = 13 observations' slice [29:47]
(object-2 position / velocity, both targets and their errors: DIMS_PER_OBS = 18), standardised with a
scikit-learn ``StandardScaler``; ``round(sigmoid(logit)) == 0`` means the HOLD task
(src/eval_mixture_of_ensembles.py:186-188).  Layer names match the reference so ``classifier.pt`` loads
unchanged; the scaler pickle is read without scikit-learn.
"""
from __future__ import annotations

import pickle

import numpy as np
import torch

N_OBS_PER_TRIAL = 13
DIMS_PER_OBS = 18
OBS_SLICE = (29, 47)          # src/eval_mixture_of_ensembles.py:181


class TaskClassifier(torch.nn.Module):
    def __init__(self, n_obs_per_trial: int = N_OBS_PER_TRIAL):
        super().__init__()
        self.layer_1 = torch.nn.Linear(n_obs_per_trial * DIMS_PER_OBS, 200)
        self.layer_2 = torch.nn.Linear(200, 100)
        self.layer_out = torch.nn.Linear(100, 1)
        self.activation = torch.nn.ReLU()

    def forward(self, x):
        x = self.activation(self.layer_1(x))
        x = self.activation(self.layer_2(x))
        return self.layer_out(x)

    @torch.no_grad()
    def predict_task(self, x) -> torch.Tensor:
        """0 = HOLD, 1 = rotating task (update_task, src/eval_mixture_of_ensembles.py:186-188)."""
        return torch.round(torch.sigmoid(self(x))).reshape(-1).to(torch.long)


class _Stub:
    def __setstate__(self, s):
        self.__dict__.update(s if isinstance(s, dict) else {"_state": s})


class _ScalerUnpickler(pickle.Unpickler):
    def find_class(self, module, name):
        if module.split(".")[0] in ("numpy", "builtins", "collections", "copyreg", "_codecs"):
            return super().find_class(module, name)
        return type(name, (_Stub,), {"__module__": module})


def load_scaler(path: str):
    """(mean, scale) float64 arrays of a pickled sklearn StandardScaler: transform(x) = (x - mean) / scale."""
    with open(path, "rb") as fh:
        sc = _ScalerUnpickler(fh).load()
    mean = np.asarray(sc.mean_, np.float64) if getattr(sc, "with_mean", True) else np.zeros_like(sc.scale_)
    scale = np.asarray(sc.scale_, np.float64) if getattr(sc, "with_std", True) else np.ones_like(mean)
    return mean, scale
