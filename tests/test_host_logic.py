"""CPU: host-side mirror of the reference interface (names, signatures, state_dict contract,
error behaviour, sharding arithmetic)."""
import inspect
import os

import numpy as np
import pytest
import torch

from lstm_ode_bci_amd import (Attention, CognitiveStateODE, EnhancedLSTMModel, LSTMODEIntegration, _lib,
                              sharding)
from lstm_ode_bci_amd import synthetic as syn

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


@pytest.mark.parametrize("C,H,L,bi", [(61, 128, 3, True), (61, 256, 3, True), (5, 8, 1, False), (14, 32, 2, True)])
def test_state_dict_contract(C, H, L, bi):
    m = EnhancedLSTMModel(input_size=C, hidden_size=H, num_layers=L, num_classes=2, dropout=0.4, bidirectional=bi)
    layout = syn.state_dict_layout(C, H, L, 2, bi)
    sd = m.state_dict()
    assert list(sd) == [n for n, _, _ in layout]
    for name, shape, _ in layout:
        assert tuple(sd[name].shape) == shape, name
    m.load_state_dict({k: torch.from_numpy(v) for k, v in syn.make_state_dict(C, H, L, 2, bi).items()}, strict=True)
    assert (m.hidden_size, m.num_layers, m.bidirectional, m.num_directions) == (H, L, bi, 2 if bi else 1)
    if (C, H, L) == (61, 128, 3):
        assert len(sd) == 40 and sum(v.numel() for v in sd.values()) == 1137731


def test_loads_reference_weights_from_golden():
    d = np.load(os.path.join(GOLDEN, "g1_tiny_L3_bi1.npz"))
    sd = {k[2:]: torch.from_numpy(d[k]) for k in d.files if k.startswith("w:")}
    m = EnhancedLSTMModel(5, 8, 3, 2, 0.4, True)
    m.load_state_dict(sd, strict=True)


def test_signatures_mirror_the_reference():
    sig = inspect.signature(EnhancedLSTMModel.__init__)
    assert list(sig.parameters)[1:] == ["input_size", "hidden_size", "num_layers", "num_classes", "dropout",
                                        "bidirectional", "num_heads"]
    assert [p.default for p in list(sig.parameters.values())[1:]] == [14, 128, 3, 2, 0.4, True, 4]
    assert list(inspect.signature(EnhancedLSTMModel.forward).parameters) == ["self", "x", "return_attention"]
    assert list(inspect.signature(LSTMODEIntegration.__init__).parameters) == ["self", "lstm_model", "ode_model",
                                                                               "coupling_strength"]
    pb = inspect.signature(LSTMODEIntegration.predict_batch)
    assert list(pb.parameters) == ["self", "X_batch", "forecast_steps", "batch_size", "show_progress"]
    assert [p.default for p in list(pb.parameters.values())[2:]] == [20, 512, True]
    pt = inspect.signature(LSTMODEIntegration.predict_trajectory)
    assert list(pt.parameters) == ["self", "X", "initial_state", "forecast_steps"]
    assert list(inspect.signature(CognitiveStateODE.solve).parameters)[:4] == ["self", "initial_state", "t_span",
                                                                              "n_points"]
    assert isinstance(Attention(16).attention, torch.nn.Sequential)


def test_ode_host_helpers_against_reference_goldens():
    d = np.load(os.path.join(GOLDEN, "g3_ode.npz"))
    ode = CognitiveStateODE(dict(syn.FITTED_RATES))
    assert np.allclose(ode.ode_system([0.2, -0.1, 0.9], 0.0), d["ode_system"], atol=1e-15)
    assert np.array_equal(ode.get_transition_matrix(), d["q_matrix"])
    assert CognitiveStateODE().params == syn.DEFAULT_RATES
    # params stays a plain mutable dict attribute that callers re-assign (06:296, 386)
    ode.params = {"k_ap": 1.0, "k_af": 1.0, "k_pa": 1.0, "k_pf": 1.0, "k_fa": 1.0, "k_fp": 1.0}
    assert ode.ode_system([1, 0, 0], 0) == [-2.0, 1.0, 1.0]
    for ai, alpha in enumerate(d["alphas"]):
        integ = LSTMODEIntegration(torch.nn.Linear(1, 1), CognitiveStateODE(dict(syn.FITTED_RATES)), float(alpha))
        probs = d[f"probs_fitted_a{ai}_s10"]
        for i in range(len(probs)):
            m = integ.modulate_ode_rates(probs[i, 1], probs[i, 0])
            assert np.array_equal(np.array([m[k] for k in syn.RATE_KEYS]), d[f"rates_fitted_a{ai}_s10"][i])


def test_no_cpu_fallback():
    m = EnhancedLSTMModel(5, 8, 1, 2, 0.4, False)
    with pytest.raises(_lib.LobError):
        m(torch.zeros(2, 12, 5))
    with pytest.raises(ValueError):
        m(torch.zeros(12, 5))
    with pytest.raises(_lib.LobError):
        Attention(16)(torch.zeros(2, 3, 16))


def test_product_never_imports_the_oracle():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    pkg = os.path.join(root, "lstm_ode_bci_amd")
    for fn in os.listdir(pkg):
        if fn.endswith(".py"):
            src = open(os.path.join(pkg, fn)).read()
            assert "import oracle" not in src and "from oracle" not in src, fn


def test_shard_bounds_cover_exactly():
    for n in (0, 1, 7, 64, 65536, 1001):
        for world in (1, 2, 3, 8):
            spans = [sharding.shard_bounds(n, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            for (a, b), (c, e) in zip(spans, spans[1:]):
                assert b == c and b >= a
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
