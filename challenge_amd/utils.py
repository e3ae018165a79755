"""Host-side helpers mirroring the on-path part of the reference's utils.py:
EPSILON / label_downsample_model (utils.py:6-7), list_to_generator (:77-85),
load_data (:88-94), safe_div (:114-116), sigmoid_focal_crossentropy (:291-347),
unitwise_norm / compute_norm (:350-366).  Tensors are torch tensors on any device
(these are cheap glue ops, not the feature hot path)."""
from __future__ import annotations

import pickle

import numpy as np
import torch

EPSILON = 1e-8
label_downsample_model = (3, 6, 7, 8, 9)


def list_to_generator(dataset):
    """utils.py:77-85: a tuple of lists is zipped, anything else is iterated."""
    def _gen():
        if isinstance(dataset, tuple):
            for z in zip(*dataset):
                yield z
        else:
            for data in dataset:
                yield data
    return _gen


def load_data(path: str):
    if path.endswith(".pickle"):
        with open(path, "rb") as f:
            return pickle.load(f)
    elif path.endswith(".npy"):
        return np.load(path)
    raise ValueError("invalid file format")


def safe_div(x: torch.Tensor, y: torch.Tensor, eps: float = EPSILON) -> torch.Tensor:
    """x / max(y, eps) (utils.py:114-116)."""
    return x / torch.clamp(y, min=eps)


def sigmoid_focal_crossentropy(y_true, y_pred, alpha=0.25, gamma=2.0, from_logits: bool = False):
    """utils.py:291-347: focal loss, summed over the class axis, mean over the rest
    of each sample's axes (returns one value per batch item like the reference)."""
    if gamma and gamma < 0:
        raise ValueError("Value of gamma should be greater than or equal to zero.")
    y_true = y_true.to(y_pred.dtype)
    if from_logits:
        ce = torch.nn.functional.binary_cross_entropy_with_logits(y_pred, y_true, reduction="none")
        pred_prob = torch.sigmoid(y_pred)
    else:
        p = torch.clamp(y_pred, 1e-7, 1 - 1e-7)  # Keras backend epsilon clipping
        ce = -(y_true * torch.log(p) + (1 - y_true) * torch.log(1 - p))
        pred_prob = y_pred
    p_t = y_true * pred_prob + (1 - y_true) * (1 - pred_prob)
    alpha_factor = y_true * alpha + (1 - y_true) * (1 - alpha) if alpha else 1.0
    modulating = torch.pow(1.0 - p_t, gamma) if gamma else 1.0
    return torch.mean(torch.sum(alpha_factor * modulating * ce, dim=-1), dim=-1)


def compute_norm(x: torch.Tensor, axis, keepdims: bool) -> torch.Tensor:
    if axis is None:
        return torch.sum(x ** 2) ** 0.5
    return torch.sum(x ** 2, dim=axis, keepdim=keepdims) ** 0.5


def unitwise_norm(x: torch.Tensor) -> torch.Tensor:
    """utils.py:350-362 for PyTorch parameter layouts.  The reference reduces over every
    axis except the output unit: Keras Dense kernels are [in, out] (axis 0) and conv
    kernels HWIO (axes 0,1,2); torch Linear weights are [out, in] and conv weights OIHW,
    so the reduction runs over all axes but 0."""
    if x.dim() <= 1:
        return compute_norm(x, None, False)
    if x.dim() in (2, 3, 4):
        return compute_norm(x, tuple(range(1, x.dim())), True)
    raise ValueError(f"Got a parameter with shape not in [1, 2, 3, 4]! {tuple(x.shape)}")
