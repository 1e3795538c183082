"""Drop-in ``EnhancedLSTMModel`` / ``Attention`` (reference: 04_lstm_model.py:112-128,
153-222; the copies in 06:66-143, 07:102-158, 08:74-126, 10:66-114 are the same classes).

Same constructor signature, same attribute and sub-module names, same ``state_dict`` keys
and shapes (strict ``load_state_dict`` of a reference checkpoint works), same
``forward(x, return_attention=False)`` contract.  The sub-modules only HOLD the parameters;
``forward`` never calls them -- every FLOP runs in the hand-written HIP kernels of
``liblob.so`` through :mod:`lstm_ode_bci_amd.ops`.
"""
from __future__ import annotations

import math

import torch
import torch.nn as nn

from . import ops
from .autograd import lob_forward


class _LSTMParams(nn.Module):
    """Parameter container with torch.nn.LSTM's names, shapes, order and initialisation
    (weight_ih_l{k}[_reverse] (4H,in), weight_hh_l{k}[_reverse] (4H,H), bias_ih/bias_hh (4H),
    gate rows stacked [i|f|g|o]); reference call site 04_lstm_model.py:181-188."""

    def __init__(self, input_size, hidden_size, num_layers, bidirectional, dropout):
        super().__init__()
        self.input_size, self.hidden_size = input_size, hidden_size
        self.num_layers, self.bidirectional, self.dropout = num_layers, bidirectional, dropout
        D = 2 if bidirectional else 1
        bound = 1.0 / math.sqrt(hidden_size)
        for layer in range(num_layers):
            in_l = input_size if layer == 0 else hidden_size * D
            for sfx in ([""] if D == 1 else ["", "_reverse"]):
                for name, shape in ((f"weight_ih_l{layer}{sfx}", (4 * hidden_size, in_l)),
                                    (f"weight_hh_l{layer}{sfx}", (4 * hidden_size, hidden_size)),
                                    (f"bias_ih_l{layer}{sfx}", (4 * hidden_size,)),
                                    (f"bias_hh_l{layer}{sfx}", (4 * hidden_size,))):
                    p = nn.Parameter(torch.empty(shape).uniform_(-bound, bound))
                    self.register_parameter(name, p)

    def layer_params(self, layer):
        """[(w_ih, w_hh, b_ih, b_hh)] per direction."""
        out = []
        for sfx in ([""] if not self.bidirectional else ["", "_reverse"]):
            out.append(tuple(getattr(self, f"{n}_l{layer}{sfx}")
                             for n in ("weight_ih", "weight_hh", "bias_ih", "bias_hh")))
        return out

    def extra_repr(self):
        return (f"{self.input_size}, {self.hidden_size}, num_layers={self.num_layers}, batch_first=True, "
                f"dropout={self.dropout}, bidirectional={self.bidirectional}")


class Attention(nn.Module):
    """Additive (tanh-MLP) attention pooling over time (04_lstm_model.py:112-128).

    ``forward(lstm_output (B,T,W)) -> (context (B,W), weights (B,T))``.
    """

    def __init__(self, hidden_size):
        super().__init__()
        self.attention = nn.Sequential(nn.Linear(hidden_size, hidden_size // 2), nn.Tanh(),
                                       nn.Linear(hidden_size // 2, 1))

    def forward(self, lstm_output):
        from .autograd import attention_forward
        return attention_forward(lstm_output, self.attention[0].weight, self.attention[0].bias,
                                 self.attention[2].weight, self.attention[2].bias)


class EnhancedLSTMModel(nn.Module):
    """BiLSTM(L x H) + LayerNorm + additive attention pooling + MLP head on MI355X.

    Constructor and ``forward`` mirror 04_lstm_model.py:163-222.  ``num_heads`` is accepted
    and ignored, as in the reference (04:164).
    """

    def __init__(self, input_size=14, hidden_size=128, num_layers=3, num_classes=2, dropout=0.4,
                 bidirectional=True, num_heads=4):
        super().__init__()
        self.hidden_size = hidden_size
        self.num_layers = num_layers
        self.bidirectional = bidirectional
        self.num_directions = 2 if bidirectional else 1
        self.input_proj = nn.Sequential(nn.Linear(input_size, hidden_size), nn.LayerNorm(hidden_size),
                                        nn.GELU(), nn.Dropout(dropout / 2))
        self.lstm = _LSTMParams(hidden_size, hidden_size, num_layers, bidirectional,
                                dropout if num_layers > 1 else 0)
        width = hidden_size * self.num_directions
        self.layer_norm = nn.LayerNorm(width)
        self.attention = Attention(width)
        self.classifier = nn.Sequential(nn.Linear(width, hidden_size), nn.GELU(), nn.Dropout(dropout),
                                        nn.Linear(hidden_size, hidden_size // 2), nn.GELU(), nn.Dropout(dropout),
                                        nn.Linear(hidden_size // 2, num_classes))
        self._seed_counter = 0

    def _dropout_seed(self):
        # one fresh 64-bit stream id per training forward, derived from torch's generator so
        # torch.manual_seed() makes runs repeatable
        return int(torch.randint(0, 2 ** 62, (1,)).item())

    def forward(self, x, return_attention=False):
        if x.dim() != 3:
            raise ValueError(f"expected (batch, seq_len, channels), got {tuple(x.shape)}")
        p_in = self.input_proj[3].p if self.training else 0.0
        p_lstm = self.lstm.dropout if self.training else 0.0
        p_cls = self.classifier[2].p if self.training else 0.0
        seed = self._dropout_seed() if (self.training and max(p_in, p_lstm, p_cls) > 0) else 0
        logits, attn = lob_forward(self, x, (p_in, p_lstm, p_cls), seed)
        if return_attention:
            return logits, attn
        return logits


class AblationLSTMModel(EnhancedLSTMModel):
    """The configurable variant used by the ablation study (09_sensitivity_analysis.py:176-242): same pipeline,
    ``use_attention=False`` pools by the mean over time, ``use_layer_norm=False`` replaces both LayerNorms by
    nn.Identity.  ``forward(x)`` returns the logits only (09:226-242); state_dict keys match the reference's."""

    def __init__(self, input_size=61, hidden_size=256, num_layers=3, num_classes=2, dropout=0.4,
                 bidirectional=True, use_attention=True, use_layer_norm=True):
        super().__init__(input_size=input_size, hidden_size=hidden_size, num_layers=num_layers,
                         num_classes=num_classes, dropout=dropout, bidirectional=bidirectional)
        self.use_attention = use_attention
        if not use_layer_norm:
            self.input_proj[1] = nn.Identity()
            self.layer_norm = nn.Identity()
        if not use_attention:
            self.attention = None

    def forward(self, x):
        return super().forward(x, return_attention=False)
