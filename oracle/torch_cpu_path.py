"""ORACLE -- TEST INFRASTRUCTURE ONLY.  Never imported by the product package.

The reference's CPU path, semantically: the same torch layer stack the
reference builds (04_lstm_model.py:163-204), so that on CPU it lands in the
same ``aten::lstm`` -> oneDNN ``mkldnn_rnn_layer`` kernels the reference uses
(BASELINE.md §3).  Two uses:

* gradient oracle: torch autograd through this stack (fp32 or fp64) gives the
  reference gradients for every parameter and for the input;
* ``bench.py``'s ``cpu_baseline`` leg times it on the GPU box's host cores
  (``kind: "port"``), because the reference's .py files cannot travel.

Pinned against the reference itself by ``tests/golden/*.npz`` (logits,
attention, intermediates and gradients captured from the imported reference;
``tests/test_oracle_vs_golden.py``).
"""
from __future__ import annotations

import time

import numpy as np
import torch
import torch.nn as nn


class _AdditiveAttention(nn.Module):
    """tanh-MLP attention pooling over time (04_lstm_model.py:112-128)."""

    def __init__(self, width):
        super().__init__()
        self.attention = nn.Sequential(nn.Linear(width, width // 2), nn.Tanh(),
                                       nn.Linear(width // 2, 1))

    def forward(self, v):
        w = torch.softmax(self.attention(v), dim=1)
        return (w * v).sum(dim=1), w.squeeze(-1)


class TorchCpuModel(nn.Module):
    """Layer stack with the reference's state_dict keys (04_lstm_model.py:163-222)."""

    def __init__(self, input_size, hidden_size=128, num_layers=3, num_classes=2,
                 dropout=0.4, bidirectional=True, use_attention=True, use_layer_norm=True):
        super().__init__()
        D = 2 if bidirectional else 1
        # use_attention / use_layer_norm: the ablation variants of 09_sensitivity_analysis.py:176-242
        # (mean pooling over time instead of attention; nn.Identity instead of both LayerNorms)
        self.input_proj = nn.Sequential(nn.Linear(input_size, hidden_size),
                                        nn.LayerNorm(hidden_size) if use_layer_norm else nn.Identity(),
                                        nn.GELU(), nn.Dropout(dropout / 2))
        self.lstm = nn.LSTM(hidden_size, hidden_size, num_layers, batch_first=True,
                            dropout=dropout if num_layers > 1 else 0,
                            bidirectional=bidirectional)
        self.layer_norm = nn.LayerNorm(hidden_size * D) if use_layer_norm else nn.Identity()
        self.attention = _AdditiveAttention(hidden_size * D) if use_attention else None
        self.classifier = nn.Sequential(
            nn.Linear(hidden_size * D, hidden_size), nn.GELU(), nn.Dropout(dropout),
            nn.Linear(hidden_size, hidden_size // 2), nn.GELU(), nn.Dropout(dropout),
            nn.Linear(hidden_size // 2, num_classes))

    def forward(self, x, return_attention=False, return_all=False):
        a = self.input_proj(x)
        y, _ = self.lstm(a)
        v = self.layer_norm(y)
        if self.attention is not None:
            ctx, w = self.attention(v)
        else:
            ctx, w = v.mean(dim=1), torch.full(v.shape[:2], 1.0 / v.shape[1], dtype=v.dtype)
        logits = self.classifier(ctx)
        if return_all:
            return {"input_proj": a, "lstm": y, "layer_norm": v, "context": ctx,
                    "attn": w, "logits": logits}
        return (logits, w) if return_attention else logits


def build(sd_numpy, input_size, hidden_size, num_layers=3, num_classes=2,
          bidirectional=True, dropout=0.4, dtype=torch.float32, use_attention=True, use_layer_norm=True):
    m = TorchCpuModel(input_size, hidden_size, num_layers, num_classes, dropout, bidirectional,
                      use_attention, use_layer_norm)
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd_numpy.items()},
                      strict=True)
    return m.to(dtype).eval()


def loss_and_grads(model, x, y, class_weight=None):
    """Weighted CE (04_lstm_model.py:430-435) + autograd; eval-mode (dropout off).

    Returns (loss, {param_name: grad}, grad_x) as numpy arrays.
    """
    model.eval()
    model.zero_grad(set_to_none=True)
    x = x.detach().clone().requires_grad_(True)
    logits = model(x)
    w = None if class_weight is None else torch.as_tensor(class_weight, dtype=logits.dtype)
    loss = nn.functional.cross_entropy(logits, y, weight=w)
    loss.backward()
    grads = {k: p.grad.detach().numpy().copy() for k, p in model.named_parameters()}
    return float(loss.detach()), grads, x.grad.detach().numpy().copy()


def time_forward(model, x, iters=5, warmup=2):
    """windows/s of eval-mode forward, best of ``iters`` (BASELINE.md §3 protocol)."""
    model.eval()
    ts = []
    with torch.no_grad():
        for i in range(warmup + iters):
            t0 = time.perf_counter()
            model(x, return_attention=True)
            dt = time.perf_counter() - t0
            if i >= warmup:
                ts.append(dt)
    return x.shape[0] / min(ts), x.shape[0] / float(np.median(ts))


def time_train_step(model, x, y, iters=3, warmup=1):
    """windows/s of train-mode fwd+bwd (dropout on, weighted CE), best / median."""
    model.train()
    w = torch.tensor([1.0, 1.0], dtype=x.dtype)
    ts = []
    for i in range(warmup + iters):
        model.zero_grad(set_to_none=True)
        t0 = time.perf_counter()
        loss = nn.functional.cross_entropy(model(x), y, weight=w)
        loss.backward()
        dt = time.perf_counter() - t0
        if i >= warmup:
            ts.append(dt)
    return x.shape[0] / min(ts), x.shape[0] / float(np.median(ts))
