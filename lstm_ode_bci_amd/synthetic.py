"""Platform-stable synthetic weights and EEG windows.

Everything is drawn from ``numpy.random.default_rng`` so the same seed gives the
same bytes in the survey container, in the tests and on the GPU box (torch's
own initialisers are not bit-stable across builds).

The tensor names and shapes are the ``state_dict`` contract of the reference's
``EnhancedLSTMModel`` (/root/reference/04_lstm_model.py:163-204): 40 tensors
for the 3-layer bidirectional configuration.
"""
from __future__ import annotations

from collections import OrderedDict

import numpy as np

INPUT_SEED = 20260104
WEIGHT_SEED = 42


def state_dict_layout(input_size, hidden_size, num_layers=3, num_classes=2,
                      bidirectional=True):
    """(name, shape, kind) triples in ``state_dict`` key order.

    kind: 'lstm' -> U(-1/sqrt(H), 1/sqrt(H)); 'linear_w'/'linear_b' ->
    U(-1/sqrt(fan_in), 1/sqrt(fan_in)); 'ones'/'zeros' -> LayerNorm affine.
    """
    H = hidden_size
    D = 2 if bidirectional else 1
    out = []
    out.append(("input_proj.0.weight", (H, input_size), ("linear", input_size)))
    out.append(("input_proj.0.bias", (H,), ("linear", input_size)))
    out.append(("input_proj.1.weight", (H,), ("ones", 0)))
    out.append(("input_proj.1.bias", (H,), ("zeros", 0)))
    for layer in range(num_layers):
        in_l = H if layer == 0 else H * D
        for sfx in ([""] if D == 1 else ["", "_reverse"]):
            out.append((f"lstm.weight_ih_l{layer}{sfx}", (4 * H, in_l), ("lstm", H)))
            out.append((f"lstm.weight_hh_l{layer}{sfx}", (4 * H, H), ("lstm", H)))
            out.append((f"lstm.bias_ih_l{layer}{sfx}", (4 * H,), ("lstm", H)))
            out.append((f"lstm.bias_hh_l{layer}{sfx}", (4 * H,), ("lstm", H)))
    W = H * D
    out.append(("layer_norm.weight", (W,), ("ones", 0)))
    out.append(("layer_norm.bias", (W,), ("zeros", 0)))
    out.append(("attention.attention.0.weight", (W // 2, W), ("linear", W)))
    out.append(("attention.attention.0.bias", (W // 2,), ("linear", W)))
    out.append(("attention.attention.2.weight", (1, W // 2), ("linear", W // 2)))
    out.append(("attention.attention.2.bias", (1,), ("linear", W // 2)))
    out.append(("classifier.0.weight", (H, W), ("linear", W)))
    out.append(("classifier.0.bias", (H,), ("linear", W)))
    out.append(("classifier.3.weight", (H // 2, H), ("linear", H)))
    out.append(("classifier.3.bias", (H // 2,), ("linear", H)))
    out.append(("classifier.6.weight", (num_classes, H // 2), ("linear", H // 2)))
    out.append(("classifier.6.bias", (num_classes,), ("linear", H // 2)))
    return out


def make_state_dict(input_size, hidden_size, num_layers=3, num_classes=2,
                    bidirectional=True, seed=WEIGHT_SEED, lstm_scale=1.0,
                    affine_jitter=0.0):
    """Seeded numpy float32 weights in ``state_dict`` key order.

    ``lstm_scale=3`` is the gate-saturation stress variant of SURVEY.md §8d.
    ``affine_jitter`` perturbs the LayerNorm affine parameters away from
    (1, 0) so that tests exercise them.
    """
    rng = np.random.default_rng(seed)
    sd = OrderedDict()
    for name, shape, (kind, fan) in state_dict_layout(
            input_size, hidden_size, num_layers, num_classes, bidirectional):
        if kind == "ones":
            a = np.ones(shape, np.float32)
            if affine_jitter:
                a = a + affine_jitter * rng.standard_normal(shape).astype(np.float32)
        elif kind == "zeros":
            a = np.zeros(shape, np.float32)
            if affine_jitter:
                a = a + affine_jitter * rng.standard_normal(shape).astype(np.float32)
        else:
            bound = 1.0 / np.sqrt(float(fan))
            a = rng.uniform(-bound, bound, size=shape).astype(np.float32)
            if kind == "lstm":
                a = (a * np.float32(lstm_scale)).astype(np.float32)
        sd[name] = a
    return sd


def make_windows(batch, seq_len=256, channels=61, seed=INPUT_SEED):
    """z-scored-looking EEG windows ``(B, T, C)`` float32 and labels ``(B,)``."""
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((batch, seq_len, channels), dtype=np.float32)
    y = rng.integers(0, 2, batch).astype(np.int64)
    return x, y


def make_probs(batch, seed=7):
    """``[P(open), P(closed)]`` rows for ODE-only workloads."""
    rng = np.random.default_rng(seed)
    p_closed = rng.uniform(0.0, 1.0, batch).astype(np.float32)
    return np.stack([np.float32(1.0) - p_closed, p_closed], axis=1)


DEFAULT_RATES = {"k_ap": 0.1, "k_af": 0.02, "k_pa": 0.15,
                 "k_pf": 0.08, "k_fa": 0.05, "k_fp": 0.1}
# README-fitted rates (reference README.md:230-233); k_pa, k_fp unpublished -> defaults.
FITTED_RATES = {"k_ap": 0.020, "k_af": 0.095, "k_pa": 0.15,
                "k_pf": 0.626, "k_fa": 0.139, "k_fp": 0.1}
RATE_KEYS = ("k_ap", "k_af", "k_pa", "k_pf", "k_fa", "k_fp")
