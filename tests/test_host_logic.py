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
    # the reference's four arguments first, in order, with its defaults (06:308); the extra keywords have defaults
    assert list(pb.parameters)[:5] == ["self", "X_batch", "forecast_steps", "batch_size", "show_progress"]
    assert [p.default for p in list(pb.parameters.values())[2:5]] == [20, 512, True]
    assert [p.default for p in list(pb.parameters.values())[5:]] == [None, False]       # use_amp, respect_batch_size
    pt = inspect.signature(LSTMODEIntegration.predict_trajectory)
    assert list(pt.parameters)[:4] == ["self", "X", "initial_state", "forecast_steps"]      # + use_amp=None
    assert [p.default for p in list(pt.parameters.values())[4:]] == [None]
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


def test_artifacts_round_trip_reference_formats(tmp_path):
    """A checkpoint written in the reference's format (04:921-933) by a plain torch module with
    the reference's layer stack loads strictly into the drop-in model; ode_model.pkl round-trips."""
    import pickle
    from lstm_ode_bci_amd import artifacts
    from oracle import torch_cpu_path as TP
    sd = syn.make_state_dict(61, 32, 2, 2, True)
    ref_like = TP.build(sd, 61, 32, 2, 2, True)
    cfg = {"input_size": 61, "hidden_size": 32, "num_layers": 2, "num_classes": 2, "dropout": 0.4,
           "bidirectional": True, "num_heads": 4}
    torch.save({"model_state_dict": ref_like.state_dict(), "model_config": cfg, "history": {"train_loss": [1.0]}},
               tmp_path / artifacts.LSTM_FILE)
    with open(tmp_path / artifacts.ODE_FILE, "wb") as f:
        pickle.dump({"params": dict(syn.FITTED_RATES), "model_class": "CognitiveStateODE"}, f)
    lstm, ode = artifacts.load_models(str(tmp_path), device="cpu")
    assert isinstance(lstm, EnhancedLSTMModel) and not lstm.training and ode.params == syn.FITTED_RATES
    for k, v in ref_like.state_dict().items():
        assert torch.equal(lstm.state_dict()[k], v)
    # and back: what we save, the reference-format reader (plain torch + pickle) can load strictly
    artifacts.save_lstm_checkpoint(lstm, tmp_path / "out.pt", cfg, history={"val_f1": [0.5]})
    ck = torch.load(tmp_path / "out.pt", weights_only=False)
    assert set(ck) == {"model_state_dict", "model_config", "history"} and ck["model_config"] == cfg
    TP.TorchCpuModel(61, 32, 2, 2, 0.4, True).load_state_dict(ck["model_state_dict"], strict=True)
    artifacts.save_ode_model(ode, tmp_path / "ode2.pkl")
    assert pickle.load(open(tmp_path / "ode2.pkl", "rb")) == {"params": dict(syn.FITTED_RATES),
                                                               "model_class": "CognitiveStateODE"}
    assert artifacts.model_config_of(lstm, 61) == cfg


def test_consumer_signatures():
    from lstm_ode_bci_amd import consumers
    assert list(inspect.signature(consumers.get_three_state_probabilities).parameters) == \
        ["lstm_model", "ode_model", "X", "batch_size"]
    assert list(inspect.signature(consumers.multistep_forecast).parameters) == ["probs", "ode_params", "horizons"]
    assert list(inspect.signature(consumers.rolling_forecast_evaluation).parameters) == \
        ["probs", "ode_params", "window_size", "horizon"]
    assert list(inspect.signature(consumers.predict_trajectory).parameters)[:4] == \
        ["initial_state", "params", "n_steps", "dt"]
    grid = np.load(os.path.join(GOLDEN, "g5_consumers.npz"))["fc_state_grid"]
    got = np.array([consumers.prob_to_ode_state(np.float32(p)) for p in np.linspace(0, 1, 21)])
    assert np.array_equal(got, grid)


# ---- training / attribution host logic -----------------------------------
def test_training_host_pieces_match_oracle_restatement():
    from lstm_ode_bci_amd import training as TR
    from oracle import train_harness as TH
    y = np.array([0] * 13 + [1] * 5)
    assert np.array_equal(TR.class_weights_from_labels(y), TH.class_weights(y))
    assert abs(float(TR.class_weights_from_labels(y).sum()) - 2.0) < 1e-6
    for epochs, warm in ((100, 5), (4, 2), (10, 1)):
        for e in range(epochs + 1):
            assert TR.warmup_cosine(e, warm, epochs) == TH.lr_factor(e, warm, epochs)
    assert TR.warmup_cosine(0, 5, 100) == 0.2 and TR.warmup_cosine(4, 5, 100) == 1.0
    assert abs(TR.warmup_cosine(100, 5, 100)) < 1e-12
    rng = np.random.default_rng(0)
    for _ in range(20):
        t, p = rng.integers(0, 2, 17), rng.integers(0, 2, 17)
        assert TR.binary_f1(t, p) == TH.binary_f1(t, p)
    assert TR.binary_f1([0, 0], [0, 0]) == 0.0
    assert [c["name"] for c in TR.ABLATION_CONFIGS] == ["Full Model", "No Attention", "Unidirectional", "1 Layer",
                                                         "2 Layers", "Minimal"]
    assert list(inspect.signature(TR.train_model).parameters)[:10] == [
        "model", "train_loader", "val_loader", "y_train", "epochs", "learning_rate", "patience", "weight_decay",
        "warmup_epochs", "gradient_accumulation_steps"]


def test_device_window_loader_semantics_on_cpu_tensors():
    from lstm_ode_bci_amd.training import DeviceWindowLoader
    X = np.arange(10 * 2 * 3, dtype=np.float64).reshape(10, 2, 3)       # float64 as in processed_sequences.npz
    y = np.array([0, 0, 0, 0, 0, 0, 0, 0, 1, 1])
    seq = DeviceWindowLoader(X, y, 4, "sequential", "cpu")
    bs = list(seq)
    assert len(seq) == 3 and [len(b[1]) for b in bs] == [4, 4, 2] and bs[0][0].dtype == torch.float32
    assert torch.equal(torch.cat([b[0] for b in bs]), torch.from_numpy(X).float())
    torch.manual_seed(0)
    sh = DeviceWindowLoader(X, y, 4, "shuffle", "cpu")
    ys = torch.cat([b[0][:, 0, 0] for b in sh])
    assert sorted(ys.tolist()) == sorted(X[:, 0, 0].tolist())           # a permutation
    wl = DeviceWindowLoader(np.repeat(X, 100, 0), np.repeat(y, 100), 250, "weighted", "cpu")
    frac = torch.cat([b[1] for b in wl]).float().mean().item()
    assert 0.4 < frac < 0.6                                              # class-balanced draws (04:358-367)


def test_training_and_attribution_refuse_cpu():
    from lstm_ode_bci_amd import AblationLSTMModel, _lib
    from lstm_ode_bci_amd.attribution import input_gradients
    from lstm_ode_bci_amd.training import FusedAdamW, WeightedCrossEntropy
    m = AblationLSTMModel(5, 8, 1, 2, 0.4, False, use_attention=False, use_layer_norm=False)
    assert "attention.attention.0.weight" not in m.state_dict() and "layer_norm.weight" not in m.state_dict()
    with pytest.raises(_lib.LobError):
        FusedAdamW(m.parameters())
    with pytest.raises(_lib.LobError):
        WeightedCrossEntropy()(torch.zeros(2, 2), torch.zeros(2, dtype=torch.long))
    with pytest.raises(_lib.LobError):
        input_gradients(m, torch.zeros(2, 12, 5))


def test_load_processed_sequences_formats(tmp_path):
    from lstm_ode_bci_amd.artifacts import load_processed_sequences
    rng = np.random.default_rng(1)
    X, y = rng.standard_normal((40, 4, 3)), rng.integers(0, 2, 40)
    np.savez_compressed(tmp_path / "a.npz", X_train=X, y_train=y, X_val=X[:5], y_val=y[:5], X_test=X[:7], y_test=y[:7])
    out = load_processed_sequences(tmp_path / "a.npz")
    assert [len(a) for a in out] == [40, 40, 5, 5, 7, 7]
    np.savez_compressed(tmp_path / "b.npz", X_train=X, y_train=y, X_val=np.array([]), y_val=np.array([]),
                        X_test=X[:7], y_test=y[:7])                     # 02_preprocessing.py:403-404
    np.random.seed(0)
    Xt, yt, Xv, yv, _, _ = load_processed_sequences(tmp_path / "b.npz")
    assert len(Xv) == 6 and len(Xt) == 34 and len(yv) == 6
    assert sorted(np.concatenate([Xt, Xv])[:, 0, 0].tolist()) == sorted(X[:, 0, 0].tolist())


def test_batch_size_is_a_lower_bound_by_default_and_an_upper_bound_on_request():
    """06:308 `batch_size`: the device pass is at least `min_device_chunk` windows unless the caller caps it."""
    integ = LSTMODEIntegration.__new__(LSTMODEIntegration)
    assert integ._chunk(512, False) == 4096 and integ._chunk(8192, False) == 8192
    assert integ._chunk(512, True) == 512
    integ.max_device_chunk = 1000
    assert integ._chunk(512, False) == 1000 and integ._chunk(512, True) == 512 and integ._chunk(2048, True) == 1000
    with pytest.raises(ValueError):
        integ._chunk(0, False)


def test_solve_method_argument_follows_the_reference_branching():
    """05:154-163: `if method == 'odeint': ... else: solve_ivp(...)` -- every other string takes the second branch (no
    ValueError in the reference), and both branches are served by the same RK4 kernel here."""
    import inspect
    src = inspect.getsource(CognitiveStateODE.solve)
    assert "raise ValueError" not in src and 'method="odeint"' in src


class _FakeDev:
    def __init__(self, index):
        self.index, self.type = index, "cuda"

    def __eq__(self, o):
        return self.index == o.index

    def __repr__(self):
        return f"cuda:{self.index}"


class _FakeTensor:
    """Enough of a device tensor for the operand checks (no GPU in the CPU suite)."""
    is_cuda, dtype = True, torch.float32

    def __init__(self, index):
        self.device = _FakeDev(index)

    def is_contiguous(self):
        return True


def test_device_mismatch_raises_instead_of_launching(monkeypatch):
    """Every lob_* launch goes to the current device: operands of another device, or of two devices, must raise
    LobError before any pointer reaches a kernel (a model moved to cuda:1 while cuda:0 is current)."""
    from lstm_ode_bci_amd import ops
    monkeypatch.setattr(torch.cuda, "current_device", lambda: 0)
    ops._chk(_FakeTensor(0), "a")                                  # same device: fine
    with pytest.raises(_lib.LobError, match="current device"):
        ops._chk(_FakeTensor(1), "a")
    with pytest.raises(_lib.LobError, match="different devices"):
        ops.same_device([_FakeTensor(0), None, _FakeTensor(1)], "forward")
    assert ops.same_device([_FakeTensor(1), None, _FakeTensor(1)], "forward").index == 1
    with pytest.raises(_lib.LobError, match="no CPU fallback"):
        ops.same_device([torch.zeros(2)], "forward")


def test_library_identity_and_variant_table():
    """lob_build_id() of the loaded library equals the hash of the sources next to it (a stale prebuilt .so is refused
    at load time); the test-only variant table round-trips and rejects unknown indices."""
    from lstm_ode_bci_amd import build as B
    lib = _lib.lib()
    assert lib.lob_build_id().decode() == B.source_id() == B.built_id()
    assert lib.lob_debug_set_variant(999, 1) == -1 and lib.lob_debug_get_variant(-1) == -1
    before = _lib.get_variant("GATE_WS")
    with _lib.variant(GATE_WS=0, REC_BWD_DMA=0):
        assert _lib.get_variant("GATE_WS") == 0 and _lib.get_variant("REC_BWD_DMA") == 0
    assert _lib.get_variant("GATE_WS") == before and _lib.get_variant("REC_BWD_DMA") == 1


def test_variant_table_ignores_stray_environment_variables(tmp_path):
    """a stray LOB_* variable on a user's box must not re-route product kernels.  The table is seeded from
    the environment only under LOB_DEBUG_VARIANTS=1 (A/B runs)."""
    import subprocess
    import sys
    code = ("import sys; sys.path.insert(0, %r); from lstm_ode_bci_amd import _lib; "
            "print(_lib.get_variant('GATE_WS'), _lib.get_variant('DMA_KT'))" % os.path.dirname(os.path.dirname(__file__)))
    env = dict(os.environ, LOB_GATE_WS="0", LOB_DMA_KT="32")
    env.pop("LOB_DEBUG_VARIANTS", None)
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, check=True).stdout.split()
    assert out == ["1", "64"]                       # defaults, whatever the environment says
    env["LOB_DEBUG_VARIANTS"] = "1"
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, check=True).stdout.split()
    assert out == ["0", "32"]


def test_integration_precision_is_resolved_per_call_and_fp32_default_says_so_once():
    """06_lstm_ode_integration.py:340: the reference's GPU runs enter autocast.  Here use_amp=None keeps the fp32 parity
    path but names the switch once; an explicit choice (argument or attribute) is silent; nothing is stored on the
    object between calls."""
    import warnings
    integ = LSTMODEIntegration.__new__(LSTMODEIntegration)
    LSTMODEIntegration._warned_fp32_default = False
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        assert integ._resolve_amp(None) is False
        assert integ._resolve_amp(None) is False
    assert len(w) == 1 and "use_amp=True" in str(w[0].message)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        assert integ._resolve_amp(True) is True and integ._resolve_amp(False) is False
        integ.use_amp = True
        assert integ._resolve_amp(None) is True and integ._resolve_amp(False) is False
    assert not w and not hasattr(integ, "_amp_now")
    sig = inspect.signature(LSTMODEIntegration.predict_trajectory)
    assert list(sig.parameters)[:4] == ["self", "X", "initial_state", "forecast_steps"]


def test_model_with_attached_optimizer_state_stays_picklable():
    """the gradient-sink registration lives outside the nn.Module (a weakref attribute on it broke
    torch.save(model) / pickle / spawn)."""
    import pickle
    from lstm_ode_bci_amd import training
    m = EnhancedLSTMModel(5, 8, 1, 2, 0.0, False)

    class _Opt:                                      # stands in for a FusedAdamW (which needs GPU parameters)
        pass
    opt = _Opt()
    import weakref
    training._GRAD_SINKS[m] = weakref.ref(opt)
    assert training.grad_sink_of(m) is opt
    assert len(pickle.dumps(m)) > 0 and not hasattr(m, "_lob_grad_sink")
    del opt
    assert training.grad_sink_of(m) is None


def test_chunk_schedule_of_host_array_calls():
    """predict_batch on a host array: the device chunks ramp up (fill of the upload pipeline = a small chunk's), full chunks
    follow, the remainder ramps down (drain = a small chunk's); every schedule covers exactly n windows in chunks the
    staging buffers can hold; one-chunk calls, ramp 0 and respect_batch_size-style small chunks keep equal chunks."""
    import random
    from lstm_ode_bci_amd.integration import _chunk_schedule as cs
    assert cs(12288, 4096, 1024) == [1024, 2048, 4096, 4096, 1024]
    assert cs(4096, 4096, 1024) == [4096] and cs(100, 4096, 1024) == [100] and cs(0, 4096, 1024) == []
    assert cs(4097, 4096, 1024) == [1024, 2048, 1025]                      # no sliver pass at the end
    assert cs(12288, 4096, 0) == [4096, 4096, 4096] and cs(3000, 512, 1024) == [512] * 5 + [440]
    rnd = random.Random(7)
    for _ in range(5000):
        n, c, r = rnd.randint(1, 70000), rnd.choice([512, 1000, 4096, 8192]), rnd.choice([0, 256, 1024, 5000])
        s = cs(n, c, r)
        assert sum(s) == n and all(0 < x <= c for x in s), (n, c, r, s)


def test_result_pool_recycles_memory_only_after_the_last_view_is_gone():
    """integration._ResultPool (round 4): the trajectories `predict_batch` returns are views of pooled storage; the storage may
    go back to the pool only when the array AND every slice / reshape taken from it are gone, and two live results never
    share memory."""
    import gc
    import numpy as np
    from lstm_ode_bci_amd.integration import _ResultPool
    pool = _ResultPool()
    small = pool.empty((10, 3), np.float64)
    assert small.flags.owndata                      # below MIN_BYTES: a plain array
    shape = (4096, 300, 3)
    a = pool.empty(shape, np.float64)
    assert a.shape == shape and a.dtype == np.float64 and a.flags.writeable and a.flags.aligned and a.flags.c_contiguous
    a[:] = 1.0
    addr = a.ctypes.data
    b = pool.empty(shape, np.float64)               # a is alive: other memory
    assert b.ctypes.data != addr
    b[:] = 2.0
    assert float(a[5, 7, 1]) == 1.0
    view = a[100:200, ::2]                          # what a caller may keep
    flat = a.reshape(-1)
    del a
    gc.collect()
    c = pool.empty(shape, np.float64)               # views alive: a's memory must NOT come back
    assert c.ctypes.data != addr
    c[:] = 3.0
    assert float(view[0, 0, 0]) == 1.0 and float(flat[-1]) == 1.0
    del view
    gc.collect()
    assert pool._held == 0
    del flat
    gc.collect()
    assert pool._held == int(np.prod(shape)) * 8    # the last view is gone: parked
    d = pool.empty(shape, np.float64)
    assert d.ctypes.data == addr and pool._held == 0
    assert float(b[0, 0, 0]) == 2.0 and float(c[0, 0, 0]) == 3.0
    # never more than MAX_HELD parked
    pool.MAX_HELD = 0
    del d
    gc.collect()
    assert pool._held == 0 and not any(pool._free.values())
