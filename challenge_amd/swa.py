"""Stochastic weight averaging callback (reference swa.py:13-44) for the torch model:
from `start_epoch` on, every `swa_freq` epochs the running average of the weights is
updated; `finalize()` loads the average into the model (and returns the state dict the
reference saves as *_SWA.h5)."""
from __future__ import annotations

import copy

import torch


class NO_SWA_ERROR(Exception):
    """Raised by finalize() when training stopped before any SWA snapshot was taken
    (reference swa.py:5-10)."""


class SWA:
    def __init__(self, start_epoch: int, swa_freq: int = 1):
        self.start_epoch = max(int(start_epoch) - 1, 0)
        self.swa_freq = max(int(swa_freq), 1)
        self.n_models = 0
        self.swa_state = None

    @torch.no_grad()
    def on_epoch_end(self, epoch: int, model: torch.nn.Module) -> None:
        if epoch < self.start_epoch or (epoch - self.start_epoch) % self.swa_freq:
            return
        state = {k: v.detach().clone() for k, v in model.state_dict().items()}
        if self.swa_state is None:
            self.swa_state = state
        else:
            for k, v in state.items():
                if torch.is_floating_point(v):
                    self.swa_state[k].mul_(self.n_models / (self.n_models + 1.0)).add_(v / (self.n_models + 1.0))
                else:
                    self.swa_state[k] = v
        self.n_models += 1

    def finalize(self, model: torch.nn.Module):
        if self.swa_state is None:
            raise NO_SWA_ERROR("training ended before the first SWA epoch")
        model.load_state_dict(copy.deepcopy(self.swa_state))
        return self.swa_state
