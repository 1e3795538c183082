"""Pin the oracle (oracle/restatement.py, oracle/torch_cpu_path.py) to the
fixtures captured from the reference itself (tests/golden/make_goldens.py).
CPU only."""
import os

import numpy as np
import pytest
import torch

from lstm_ode_bci_amd import synthetic as syn
from oracle import restatement as R
from oracle import torch_cpu_path as TP

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def _weights(d):
    return {k[2:]: d[k] for k in d.files if k.startswith("w:")}


@pytest.mark.parametrize("L", [1, 3])
@pytest.mark.parametrize("bi", [0, 1])
def test_tiny_forward_every_intermediate(L, bi):
    d = np.load(os.path.join(GOLDEN, f"g1_tiny_L{L}_bi{bi}.npz"))
    sd = _weights(d)
    # the synthetic generator reproduces the stored weights bit for bit
    sd2 = syn.make_state_dict(5, 8, L, 2, bool(bi), seed=100 + L + 10 * bi, affine_jitter=0.1)
    assert list(sd2) == list(sd)
    for k in sd:
        assert np.array_equal(sd[k], sd2[k]), k
    r = R.model_forward(sd, d["x"], L, bool(bi), dtype=np.float64)
    for name in ("input_proj", "lstm", "layer_norm", "context", "attn", "logits"):
        assert np.abs(r[name] - d[name]).max() < 2e-6, name
    r32 = R.model_forward(sd, d["x"], L, bool(bi), dtype=np.float32)
    assert np.abs(r32["logits"] - d["logits"]).max() < 5e-6


@pytest.mark.parametrize("L", [1, 3])
@pytest.mark.parametrize("bi", [0, 1])
def test_tiny_gradients_torch_twin(L, bi):
    d = np.load(os.path.join(GOLDEN, f"g1_tiny_L{L}_bi{bi}.npz"))
    m = TP.build(_weights(d), 5, 8, L, 2, bool(bi), dtype=torch.float64)
    loss, gp, gx = TP.loss_and_grads(m, torch.from_numpy(d["x"]).double(), torch.from_numpy(d["y"]))
    assert abs(loss - float(d["loss"])) < 1e-6
    assert np.abs(gx - d["grad_x"]).max() < 1e-6
    for k, g in gp.items():
        assert np.abs(g - d["g:" + k]).max() < 2e-6, k


@pytest.mark.parametrize("H", [128, 256])
def test_full_size_logits_attention(H):
    d = np.load(os.path.join(GOLDEN, f"g2_full_H{H}.npz"))
    sd = syn.make_state_dict(61, H, 3, 2, True)
    x, y = syn.make_windows(8)
    m = TP.build(sd, 61, H)
    with torch.no_grad():
        r = m(torch.from_numpy(x), return_all=True)
    assert np.abs(r["logits"].numpy() - d["logits"]).max() < 1e-6
    assert np.abs(r["attn"].numpy() - d["attn"]).max() < 1e-7
    assert np.abs(r["context"].numpy() - d["context"]).max() < 1e-6
    assert np.abs(r["lstm"].numpy()[0, ::16] - d["lstm_slice"]).max() < 1e-6
    if H == 128:      # the numpy restatement at full size (fp64, 2 windows: seconds)
        rr = R.model_forward(sd, x[:2], 3, True, dtype=np.float64)
        assert np.abs(rr["logits"] - d["logits"][:2]).max() < 1e-6
        assert np.abs(rr["attn"] - d["attn"][:2]).max() < 1e-7
        sd3 = syn.make_state_dict(61, H, 3, 2, True, lstm_scale=3.0)
        r3 = R.model_forward(sd3, x[:2], 3, True, dtype=np.float64)
        assert np.abs(r3["logits"] - d["logits_stress"][:2]).max() < 2e-6
    loss, gp, gx = TP.loss_and_grads(m, torch.from_numpy(x), torch.from_numpy(y))
    assert abs(loss - float(d["loss"])) < 1e-6
    names = [str(n) for n in d["grad_names"]]
    l2 = np.array([np.sqrt((gp[k].astype(np.float64) ** 2).sum()) for k in names])
    assert np.allclose(l2, d["grad_l2"], rtol=1e-4, atol=1e-9)
    assert np.abs(gx[:, ::32] - d["grad_x_slice"]).max() < 1e-7
    assert np.abs(gp["classifier.6.weight"] - d["grad_cls6_w"]).max() < 1e-6


def test_ode_grid_against_reference():
    d = np.load(os.path.join(GOLDEN, "g3_ode.npz"))
    alphas = d["alphas"]
    n = 0
    for pname, rates in (("default", syn.DEFAULT_RATES), ("fitted", syn.FITTED_RATES)):
        for ai, alpha in enumerate(alphas):
            for steps in (10, 20, 300):
                key = f"{pname}_a{ai}_s{steps}"
                if "traj_" + key not in d.files:
                    continue
                probs = d["probs_" + key]
                traj, pred = R.predict_from_probs(probs, rates, float(alpha), steps, R.solve_odeint)
                assert np.array_equal(traj, d["traj_" + key]), key     # same LSODA, same inputs
                assert np.array_equal(pred, d["pred_" + key]), key
                # closed form and the kernel's algorithm (RK4, 16 sub-steps) agree with LSODA
                te, _ = R.predict_from_probs(probs, rates, float(alpha), steps, R.solve_expm)
                assert np.abs(te - d["traj_" + key]).max() < 1e-6
                if steps != 300 or ai == 4:
                    tr, pr = R.predict_from_probs(
                        probs, rates, float(alpha), steps,
                        lambda y0, ts, npts, p: R.solve_rk4(y0, ts, npts, p, substeps=16))
                    assert np.abs(tr - d["traj_" + key]).max() < 1e-6
                    assert np.array_equal(pr, d["pred_" + key])
                if "rates_" + key in d.files:
                    for i in range(len(probs)):
                        m = R.modulate_rates(rates, float(alpha), probs[i, 1], probs[i, 0])
                        assert np.array_equal(np.array([m[k] for k in syn.RATE_KEYS]), d["rates_" + key][i])
                n += 1
    assert n == 24
    t, s = R.solve_odeint([0.5, 0.3, 0.2], (0, 7.5), 33, syn.FITTED_RATES)
    assert np.array_equal(t, d["solve_t"]) and np.array_equal(s, d["solve_sol"])
    assert np.allclose(R.ode_rhs([0.2, -0.1, 0.9], 0.0, syn.FITTED_RATES), d["ode_system"], atol=1e-15)
    assert np.allclose(R.q_matrix(syn.FITTED_RATES), d["q_matrix"], atol=0)


def test_coupled_predict_batch():
    d = np.load(os.path.join(GOLDEN, "g4_coupled.npz"))
    sd = syn.make_state_dict(61, 128, 3, 2, True)
    sd["classifier.6.weight"] = sd["classifier.6.weight"] * d["cls6_scale"]
    sd["classifier.6.bias"] = d["cls6_bias"]
    x, _ = syn.make_windows(32, seed=11)
    m = TP.build(sd, 61, 128)
    with torch.no_grad():
        probs = torch.softmax(m(torch.from_numpy(x)), dim=1).numpy()
    for pname, rates, alpha in (("default", syn.DEFAULT_RATES, 0.5), ("fitted", syn.FITTED_RATES, 0.5),
                                ("fitted_a1", syn.FITTED_RATES, 1.0)):
        assert np.abs(probs - d["probs_" + pname]).max() < 2e-5      # logit scale x400
        traj, pred = R.predict_from_probs(d["probs_" + pname], rates, alpha, 20)
        assert np.array_equal(traj, d["traj_" + pname])
        assert np.array_equal(pred, d["pred_" + pname])
    assert d["pred_fitted_a1"].sum() > 0 and (d["pred_fitted_a1"] == 0).sum() > 0


def test_consumers_08_and_10():
    d = np.load(os.path.join(GOLDEN, "g5_consumers.npz"))
    for pname, rates in (("default", syn.DEFAULT_RATES), ("fitted", syn.FITTED_RATES)):
        final, pred = R.three_state_from_probs(d[f"three_lstm_probs_{pname}"], rates)
        assert np.array_equal(final, d[f"three_state_{pname}"])
        assert np.array_equal(pred, d[f"three_pred_{pname}"])
        res = R.multistep_forecast(d["fc_probs"], rates)
        for h in (5, 10, 20):
            assert np.array_equal(res[h]["predictions"], d[f"fc_pred_{pname}_h{h}"])
            assert np.array_equal(res[h]["actuals"], d[f"fc_act_{pname}_h{h}"])
        assert np.array_equal(R.forecast_raw(R.prob_to_ode_state(d["fc_probs"][5, 1]), rates, 20), d[f"fc_traj_{pname}"])
        roll = np.array(R.rolling_forecast(d["roll_probs"], rates, window_size=20, horizon=10), np.float64)
        assert np.array_equal(roll, d[f"roll_{pname}"])
    grid = np.array([R.prob_to_ode_state(np.float32(p)) for p in np.linspace(0, 1, 21)])
    assert np.array_equal(grid, d["fc_state_grid"])
    assert set(np.unique(d["three_pred_fitted"])) >= {0, 1}


# ---- the steps either side of fwd+bwd ---------------------------------------
def _batches(x, y, bs):
    return [(torch.from_numpy(x[i:i + bs]), torch.from_numpy(y[i:i + bs])) for i in range(0, len(x), bs)]


def test_training_harness_restatement_matches_reference_train_model():
    from oracle import train_harness as TH
    d = np.load(os.path.join(GOLDEN, "g6_training.npz"))
    kw = dict(zip([str(k) for k in d["kw_names"]], d["kw_vals"]))
    sd0 = {k[3:]: d[k] for k in d.files if k.startswith("w0:")}
    m = TP.build(sd0, 5, 8, 3, 2, True, dropout=0.0)
    hist = TH.train_model(m, _batches(d["x_train"], d["y_train"], 4), _batches(d["x_val"], d["y_val"], 5),
                          d["y_train"], epochs=int(kw["epochs"]), learning_rate=kw["learning_rate"],
                          patience=int(kw["patience"]), weight_decay=kw["weight_decay"],
                          warmup_epochs=int(kw["warmup_epochs"]),
                          gradient_accumulation_steps=int(kw["gradient_accumulation_steps"]))
    for k in ("train_loss", "val_loss", "train_acc", "val_acc", "val_f1", "learning_rates"):
        assert np.allclose(hist[k], d["hist:" + k], rtol=0, atol=2e-6), k
    sd1 = m.state_dict()
    moved = 0.0
    for k, v in sd1.items():
        ref = d["w1:" + k]
        if k == "attention.attention.2.bias":
            # b2 cancels in the softmax: its gradient is analytically 0, numerically ~1e-9 rounding noise that
            # Adam turns into an O(lr * g / (|g| + eps)) step.  Not a property of the algorithm; no output sees it.
            assert np.abs(v.numpy() - ref).max() < 5e-3
            continue
        assert np.abs(v.numpy() - ref).max() < 2e-5, k
        moved = max(moved, np.abs(ref - sd0[k]).max())
    assert moved > 1e-2          # the fixture really trains (8 optimizer steps at lr up to 5e-3)


def test_channel_importance_restatement_matches_reference():
    from oracle import train_harness as TH
    d = np.load(os.path.join(GOLDEN, "g7_channel_importance.npz"))
    m = TP.build(_weights(d), 5, 8, 3, 2, True, dropout=0.0)
    imp = TH.channel_importance(m, d["x"], batch_size=3)
    assert np.abs(imp - d["importance"]).max() < 1e-6
    assert abs(imp.sum() - 1.0) < 1e-12


@pytest.mark.parametrize("name", ["full", "noattn", "noln", "minimal", "bare"])
def test_ablation_variants_torch_twin(name):
    d = np.load(os.path.join(GOLDEN, "g8_ablation.npz"))
    L, bi, att, ln = (int(v) for v in d[name + ":cfg"])
    pre = name + ":w:"
    sd = {k[len(pre):]: d[k] for k in d.files if k.startswith(pre)}
    m = TP.build(sd, 5, 8, L, 2, bool(bi), use_attention=bool(att), use_layer_norm=bool(ln))
    loss, gp, gx = TP.loss_and_grads(m, torch.from_numpy(d["x"]), torch.from_numpy(d["y"]))
    with torch.no_grad():
        logits = m(torch.from_numpy(d["x"])).numpy()
    assert np.abs(logits - d[name + ":logits"]).max() < 1e-6
    assert abs(loss - float(d[name + ":loss"])) < 1e-6
    assert np.abs(gx - d[name + ":grad_x"]).max() < 1e-6
    for k, g in gp.items():
        assert np.abs(g - d[f"{name}:g:{k}"]).max() < 2e-6, k
