"""Gradient-based channel attribution on the device.

``compute_channel_importance`` of 07_explainability.py:203-285 runs, for every window i of a batch,
``outputs[i, pred_class[i]].backward(retain_graph=True)`` and reads ``X_batch.grad[i]`` (07:248-258): B
backward passes per batch.  The windows of a batch are independent (no BatchNorm; LayerNorm per row; softmax per
window), so the gradient of ``sum_i outputs[i, pred_class[i]]`` w.r.t. ``X_batch`` carries exactly those B
per-window gradients in its rows: ONE vector-Jacobian launch per batch gives the same numbers.  The
``|grad|.mean(time)`` + sum-over-windows reduction is a HIP kernel (lob_abs_colsum_f32); nothing returns to the
host until the final (C,) vector.
"""
from __future__ import annotations

import numpy as np
import torch

from . import ops


def input_gradients(lstm_model, X_batch, target_class=None):
    """Per-window gradient of the chosen logit w.r.t. the window: ``(grad (B,T,C), pred_class (B,))``.
    ``target_class=None`` uses the predicted class of each window (07:245-246)."""
    from .autograd import input_grad_only
    X_batch = X_batch.detach().clone().requires_grad_(True)
    with input_grad_only():          # only d logit / d x is wanted: no weight-gradient GEMMs in the backward
        outputs = lstm_model(X_batch)
    if isinstance(outputs, tuple):
        outputs = outputs[0]
    pred = outputs.argmax(dim=1) if target_class is None else target_class
    seed = torch.zeros_like(outputs)
    seed.scatter_(1, pred.reshape(-1, 1), 1.0)       # d/dx sum_i outputs[i, pred_i]
    (grad,) = torch.autograd.grad(outputs, X_batch, grad_outputs=seed)
    return grad, pred


def compute_channel_importance(lstm_model, X_test, n_samples=100, batch_size=32, channel_names=None,
                               train_mode=True, device=None):
    """07_explainability.py:203-285.  Returns a pandas DataFrame with columns ``Channel`` / ``Importance`` sorted by
    importance (descending), importances normalised to sum 1.

    ``train_mode=True`` mirrors the reference, which switches the model to ``train()`` for this computation
    (07:219, a cuDNN requirement there), so dropout noise is part of its attributions; pass False for
    deterministic attributions.  ``batch_size`` may be far larger than the reference's 32 here (one backward per
    batch, not one per window)."""
    import pandas as pd
    dev = device or next(lstm_model.parameters()).device
    was_training = lstm_model.training
    lstm_model.train(bool(train_mode))
    n_channels = X_test.shape[2]
    if channel_names is None or len(channel_names) != n_channels:
        channel_names = [f"Ch{i + 1}" for i in range(n_channels)]
    n_samples = min(n_samples, len(X_test))
    indices = np.random.choice(len(X_test), n_samples, replace=False)
    X_subset = np.asarray(X_test)[indices]
    T = X_subset.shape[1]
    importance = torch.zeros(n_channels, device=dev, dtype=torch.float32)
    for s in range(0, n_samples, batch_size):
        xb = torch.as_tensor(np.asarray(X_subset[s:s + batch_size], dtype=np.float32)).to(dev)
        grad, _ = input_gradients(lstm_model, xb)
        # X_batch.grad[i].abs().mean(dim=0), summed over the windows (07:257-258)
        ops.abs_colsum(grad.reshape(-1, n_channels), importance, scale=1.0 / T)
    imp = importance.double().cpu().numpy() / n_samples
    imp = imp / imp.sum()
    lstm_model.train(was_training)
    df = pd.DataFrame({"Channel": list(channel_names), "Importance": imp})
    return df.sort_values("Importance", ascending=False)
