#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE's own classes.

Runs only in the build container (it reads /root/reference, which does not
exist on the GPU box).  The reference files are exec-loaded unmodified with two
accommodations that do not touch arithmetic: a stub ``seaborn``
module (imported at top level, used only for plots) and ``__file__`` redirected
to a scratch directory (the scripts mkdir ``<parent>/outputs`` at import).

Only arrays are written: inputs and the reference's outputs.  No reference
source, bytecode or pickled class leaves /root/reference.

    python tests/golden/make_goldens.py
"""
import os
import sys
import tempfile
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from lstm_ode_bci_amd import synthetic as syn  # noqa: E402

REF = "/root/reference"
SCRATCH = tempfile.mkdtemp(prefix="lob_ref_")


def load_ref(fname):
    sys.modules.setdefault("seaborn", types.ModuleType("seaborn"))
    import matplotlib
    matplotlib.use("Agg")
    path = os.path.join(REF, fname)
    g = {"__name__": "ref_" + fname[:2], "__file__": os.path.join(SCRATCH, "ref", fname)}
    with open(path) as f:
        exec(compile(f.read(), path, "exec"), g)
    return g


def ref_model(g, sd, C, H, L, bi):
    m = g["EnhancedLSTMModel"](input_size=C, hidden_size=H, num_layers=L, num_classes=2,
                               dropout=0.4, bidirectional=bi)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    m.eval()
    return m


def run_with_hooks(m, x):
    caps = {}

    def hook(name):
        def fn(_mod, _inp, out):
            caps[name] = out[0] if (isinstance(out, tuple) and name == "lstm") else out
        return fn
    hs = [m.input_proj.register_forward_hook(hook("input_proj")),
          m.lstm.register_forward_hook(hook("lstm")),
          m.layer_norm.register_forward_hook(hook("layer_norm")),
          m.attention.register_forward_hook(hook("attention"))]
    logits, attn = m(x, return_attention=True)
    for h in hs:
        h.remove()
    return logits, attn, caps


def grads_of(m, x, y):
    m.zero_grad(set_to_none=True)
    xg = x.clone().requires_grad_(True)
    loss = torch.nn.functional.cross_entropy(m(xg), y)
    loss.backward()
    return float(loss), {k: p.grad.numpy().copy() for k, p in m.named_parameters()}, xg.grad.numpy().copy()


def g1_tiny(g4):
    C, H, T, B = 5, 8, 12, 3
    for L in (1, 3):
        for bi in (False, True):
            sd = syn.make_state_dict(C, H, L, 2, bi, seed=100 + L + 10 * bi, affine_jitter=0.1)
            x, y = syn.make_windows(B, T, C, seed=5)
            m = ref_model(g4, sd, C, H, L, bi)
            xt, yt = torch.from_numpy(x), torch.from_numpy(y)
            with torch.no_grad():
                logits, attn, caps = run_with_hooks(m, xt)
            loss, gp, gx = grads_of(m, xt, yt)
            out = {"x": x, "y": y, "logits": logits.numpy(), "attn": attn.numpy(),
                   "input_proj": caps["input_proj"].numpy(), "lstm": caps["lstm"].numpy(),
                   "layer_norm": caps["layer_norm"].numpy(),
                   "context": caps["attention"][0].numpy(), "loss": np.float64(loss),
                   "grad_x": gx}
            out.update({"w:" + k: v for k, v in sd.items()})
            out.update({"g:" + k: v for k, v in gp.items()})
            np.savez_compressed(os.path.join(HERE, f"g1_tiny_L{L}_bi{int(bi)}.npz"), **out)
            print("g1", L, bi, logits.numpy().ravel()[:2])


def g2_full(g4):
    C, T, B = 61, 256, 8
    for H in (128, 256):
        sd = syn.make_state_dict(C, H, 3, 2, True)
        x, y = syn.make_windows(B, T, C)
        m = ref_model(g4, sd, C, H, 3, True)
        xt, yt = torch.from_numpy(x), torch.from_numpy(y)
        with torch.no_grad():
            logits, attn, caps = run_with_hooks(m, xt)
        loss, gp, gx = grads_of(m, xt, yt)
        names = list(gp.keys())
        out = {"logits": logits.numpy(), "attn": attn.numpy(),
               "context": caps["attention"][0].numpy(),
               "lstm_slice": caps["lstm"].numpy()[0, ::16, :],
               "input_proj_slice": caps["input_proj"].numpy()[0, ::16, :],
               "loss": np.float64(loss),
               "grad_names": np.array(names),
               "grad_l2": np.array([np.sqrt((gp[k].astype(np.float64) ** 2).sum()) for k in names]),
               "grad_sum": np.array([gp[k].astype(np.float64).sum() for k in names]),
               "grad_x_slice": gx[:, ::32, :],
               "grad_cls6_w": gp["classifier.6.weight"],
               "grad_whh_l2_slice": gp["lstm.weight_hh_l2"][::64, ::16],
               "grad_wih_l0r_slice": gp["lstm.weight_ih_l0_reverse"][::64, ::16],
               "grad_proj_w_slice": gp["input_proj.0.weight"][::16, :]}
        # stress variant: LSTM weights x3 pushes the gates toward saturation
        sd3 = syn.make_state_dict(C, H, 3, 2, True, lstm_scale=3.0)
        m3 = ref_model(g4, sd3, C, H, 3, True)
        with torch.no_grad():
            l3, a3 = m3(xt, return_attention=True)
        out["logits_stress"] = l3.numpy()
        out["attn_stress"] = a3.numpy()
        np.savez_compressed(os.path.join(HERE, f"g2_full_H{H}.npz"), **out)
        print("g2", H, logits.numpy()[0], loss)


class _FakeModel(torch.nn.Module):
    """Emits logits = log([1-p, p]) for p = X[:,0,0] so that the reference's own
    predict_batch sees chosen probabilities."""

    def forward(self, X, return_attention=False):
        p = X[:, 0, 0].double().clamp(1e-30, 1.0)
        q = (1.0 - X[:, 0, 0].double()).clamp(1e-30, 1.0)
        logits = torch.stack([q.log(), p.log()], dim=1).float()
        return logits, torch.zeros(X.shape[0], X.shape[1])


def g3_ode(g5, g6):
    pcs = np.round(np.arange(0, 1.0001, 0.1), 3).astype(np.float32)
    extra = np.array([0.59, 0.6, 0.61, 0.39, 0.4, 0.41, 0.999], np.float32)
    pcs = np.concatenate([pcs, extra])
    alphas = [0.0, 0.25, 0.5, 0.75, 1.0]
    out = {"p_closed_in": pcs, "alphas": np.array(alphas)}
    Integ, ODE = g6["LSTMODEIntegration"], g6["CognitiveStateODE"]
    X = pcs.reshape(-1, 1, 1)
    for pname, rates in (("default", syn.DEFAULT_RATES), ("fitted", syn.FITTED_RATES)):
        for ai, alpha in enumerate(alphas):
            for steps in (10, 20, 300):
                if steps == 300 and alpha not in (0.5, 1.0):
                    continue
                integ = Integ(_FakeModel(), ODE(dict(rates)), coupling_strength=alpha)
                traj, probs, pred = integ.predict_batch(X, forecast_steps=steps, show_progress=False)
                key = f"{pname}_a{ai}_s{steps}"
                out["traj_" + key] = traj
                out["probs_" + key] = probs
                out["pred_" + key] = pred
                if steps == 10:
                    mods = [integ.modulate_ode_rates(probs[i, 1], probs[i, 0]) for i in range(len(pcs))]
                    out["rates_" + key] = np.array([[m[k] for k in syn.RATE_KEYS] for m in mods])
    # the 05 variant of solve must equal the 06 variant (and exercises n_points != t1)
    o5 = g5["CognitiveStateODE"](dict(syn.FITTED_RATES))
    o6 = ODE(dict(syn.FITTED_RATES))
    t5, s5 = o5.solve([0.5, 0.3, 0.2], (0, 7.5), 33)
    t6, s6 = o6.solve([0.5, 0.3, 0.2], (0, 7.5), 33)
    assert np.array_equal(s5, s6)
    out["solve_t"] = t5
    out["solve_sol"] = s5
    out["ode_system"] = np.array(o6.ode_system([0.2, -0.1, 0.9], 0.0))
    out["q_matrix"] = o5.get_transition_matrix()
    np.savez_compressed(os.path.join(HERE, "g3_ode.npz"), **out)
    print("g3 done")


def g4_coupled(g6):
    C, H, T, N = 61, 128, 256, 32
    sd = syn.make_state_dict(C, H, 3, 2, True)
    x, _ = syn.make_windows(N, T, C, seed=11)
    # widen the logit spread so the three initial-state branches all occur
    sd["classifier.6.weight"] = sd["classifier.6.weight"] * np.float32(400.0)
    m = ref_model(g6, sd, C, H, 3, True)
    with torch.no_grad():   # centre the logit gap so P(closed) straddles 0.4 / 0.6
        lg = m(torch.from_numpy(x)).numpy()
    sd["classifier.6.bias"] = sd["classifier.6.bias"].copy()
    sd["classifier.6.bias"][1] += np.float32(np.median(lg[:, 0] - lg[:, 1]))
    m = ref_model(g6, sd, C, H, 3, True)
    out_bias = sd["classifier.6.bias"].copy()
    out = {}
    for pname, rates, alpha in (("default", syn.DEFAULT_RATES, 0.5), ("fitted", syn.FITTED_RATES, 0.5),
                                ("fitted_a1", syn.FITTED_RATES, 1.0)):
        integ = g6["LSTMODEIntegration"](m, g6["CognitiveStateODE"](dict(rates)), coupling_strength=alpha)
        traj, probs, pred = integ.predict_batch(x, forecast_steps=20, batch_size=16, show_progress=False)
        t1, p1, a1 = integ.predict_trajectory(x[:1], forecast_steps=20)
        assert np.allclose(t1, traj[0], atol=1e-6)   # B=1 vs B=16 oneDNN paths differ in the last ulp
        out[f"traj0_single_{pname}"] = t1
        out[f"traj_{pname}"] = traj
        out[f"probs_{pname}"] = probs
        out[f"pred_{pname}"] = pred
        out[f"attn0_{pname}"] = a1
    out["cls6_scale"] = np.float32(400.0)
    out["cls6_bias"] = out_bias
    np.savez_compressed(os.path.join(HERE, "g4_coupled.npz"), **out)
    print("g4 probs", out["probs_default"][:4].ravel(), "pred", out["pred_fitted"][:8])


def g5_consumers(g6, g8, g10):
    """08_forecasting.py / 10_three_state_probabilities.py consumers on the G4 model and windows."""
    C, H, T, N = 61, 128, 256, 32
    d4 = np.load(os.path.join(HERE, "g4_coupled.npz"))
    sd = syn.make_state_dict(C, H, 3, 2, True)
    sd["classifier.6.weight"] = sd["classifier.6.weight"] * d4["cls6_scale"]
    sd["classifier.6.bias"] = d4["cls6_bias"]
    x, _ = syn.make_windows(N, T, C, seed=11)
    out = {}
    m10 = ref_model(g10, sd, C, H, 3, True)
    for pname, rates in (("default", syn.DEFAULT_RATES), ("fitted", syn.FITTED_RATES)):
        ode = g10["CognitiveStateODE"](dict(rates))
        lp, three, pred = g10["get_three_state_probabilities"](m10, ode, x, batch_size=16)
        out[f"three_lstm_probs_{pname}"] = lp
        out[f"three_state_{pname}"] = three
        out[f"three_pred_{pname}"] = pred
    # 08: multistep_forecast on a probability series (the reference prints, so silence it)
    import contextlib, io
    probs = syn.make_probs(64, seed=3)
    out["fc_probs"] = probs
    for pname, rates in (("default", syn.DEFAULT_RATES), ("fitted", syn.FITTED_RATES)):
        with contextlib.redirect_stdout(io.StringIO()):
            res = g8["multistep_forecast"](probs, dict(rates), horizons=[5, 10, 20])
        for h in (5, 10, 20):
            out[f"fc_pred_{pname}_h{h}"] = res[h]["predictions"]
            out[f"fc_act_{pname}_h{h}"] = res[h]["actuals"]
        out[f"fc_traj_{pname}"] = g8["predict_trajectory"](g8["prob_to_ode_state"](probs[5, 1]), dict(rates), 20)
    out["fc_state_grid"] = np.array([g8["prob_to_ode_state"](np.float32(p)) for p in np.linspace(0, 1, 21)])
    # 08: rolling_forecast_evaluation (window accuracy / MAE) on a longer probability series
    probs_r = syn.make_probs(150, seed=9)
    out["roll_probs"] = probs_r
    for pname, rates in (("default", syn.DEFAULT_RATES), ("fitted", syn.FITTED_RATES)):
        with contextlib.redirect_stdout(io.StringIO()):
            df = g8["rolling_forecast_evaluation"](probs_r, dict(rates), window_size=20, horizon=10)
        out[f"roll_{pname}"] = df[["window", "accuracy", "mae"]].to_numpy(np.float64)
    np.savez_compressed(os.path.join(HERE, "g5_consumers.npz"), **out)
    print("g5 three_pred", out["three_pred_fitted"][:12], "fc", out["fc_pred_fitted_h20"][:3])


def _silence():
    import contextlib, io
    return contextlib.redirect_stdout(io.StringIO())


def g6_training(g4):
    """04_lstm_model.py:406-596 train_model on a tiny model (dropout 0 -> deterministic): weighted CE, gradient
    accumulation, clip 1.0, AdamW, warm-up + cosine.  Pins the harness arithmetic, incl. the last-epoch weights."""
    from torch.utils.data import DataLoader, TensorDataset
    C, H, T, L = 5, 8, 12, 3
    sd = syn.make_state_dict(C, H, L, 2, True, seed=321, affine_jitter=0.1)
    xtr, ytr = syn.make_windows(24, T, C, seed=61)
    ytr[:5] = 1; ytr[5:9] = 0                      # both classes present, unbalanced
    xva, yva = syn.make_windows(10, T, C, seed=62)
    yva[:3] = 1; yva[3:6] = 0
    m = g4["EnhancedLSTMModel"](input_size=C, hidden_size=H, num_layers=L, num_classes=2, dropout=0.0,
                                bidirectional=True)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    tl = DataLoader(TensorDataset(torch.from_numpy(xtr), torch.from_numpy(ytr)), batch_size=4, shuffle=False)
    vl = DataLoader(TensorDataset(torch.from_numpy(xva), torch.from_numpy(yva)), batch_size=5, shuffle=False)
    kw = dict(epochs=4, learning_rate=5e-3, patience=50, weight_decay=1e-2, warmup_epochs=2,
              gradient_accumulation_steps=2)
    with _silence():
        m, hist = g4["train_model"](m, tl, vl, ytr, **kw)
    out = {"x_train": xtr, "y_train": ytr, "x_val": xva, "y_val": yva,
           "kw_names": np.array(list(kw)), "kw_vals": np.array([float(v) for v in kw.values()])}
    out.update({"w0:" + k: v for k, v in sd.items()})
    out.update({"w1:" + k: v.detach().numpy().copy() for k, v in m.state_dict().items()})
    out.update({"hist:" + k: np.array(v, dtype=np.float64) for k, v in hist.items()})
    np.savez_compressed(os.path.join(HERE, "g6_training.npz"), **out)
    print("g6 train_loss", hist["train_loss"], "lr", hist["learning_rates"])


def g7_channel_importance(g7):
    """07_explainability.py:203-285 compute_channel_importance (per-sample input gradients of the predicted
    logit, |.| averaged over time, summed over samples, normalised)."""
    C, H, T, L, N = 5, 8, 12, 3, 7
    sd = syn.make_state_dict(C, H, L, 2, True, seed=77, affine_jitter=0.1)
    x, _ = syn.make_windows(N, T, C, seed=71)
    m = g7["EnhancedLSTMModel"](input_size=C, hidden_size=H, num_layers=L, num_classes=2, dropout=0.0,
                                bidirectional=True)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    m.eval()
    np.random.seed(0)
    with _silence():
        df = g7["compute_channel_importance"](m, x, n_samples=N, batch_size=3)
    df = df.sort_index()                     # back to channel order
    out = {"x": x, "importance": df["Importance"].to_numpy(np.float64), "channels": np.array(list(df["Channel"]))}
    out.update({"w:" + k: v for k, v in sd.items()})
    np.savez_compressed(os.path.join(HERE, "g7_channel_importance.npz"), **out)
    print("g7", out["importance"])


def g8_ablation(g9):
    """09_sensitivity_analysis.py:176-242 AblationLSTMModel variants (mean pooling, no LayerNorm, uni-directional,
    fewer layers): logits and gradients in eval mode."""
    C, H, T, B = 5, 8, 12, 4
    x, y = syn.make_windows(B, T, C, seed=81)
    xt, yt = torch.from_numpy(x), torch.from_numpy(y)
    out = {"x": x, "y": y}
    variants = [("full", 3, True, True, True), ("noattn", 3, True, False, True), ("noln", 2, True, True, False),
                ("minimal", 1, False, False, True), ("bare", 1, False, False, False)]
    for name, L, bi, att, ln in variants:
        sd = syn.make_state_dict(C, H, L, 2, bi, seed=800 + len(name), affine_jitter=0.1)
        m = g9["AblationLSTMModel"](input_size=C, hidden_size=H, num_layers=L, num_classes=2, dropout=0.4,
                                    bidirectional=bi, use_attention=att, use_layer_norm=ln)
        keys = set(m.state_dict().keys())
        m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items() if k in keys}, strict=True)
        m.eval()
        m.zero_grad(set_to_none=True)
        xg = xt.clone().requires_grad_(True)
        logits = m(xg)
        loss = torch.nn.functional.cross_entropy(logits, yt)
        loss.backward()
        out[f"{name}:cfg"] = np.array([L, int(bi), int(att), int(ln)])
        out[f"{name}:logits"] = logits.detach().numpy()
        out[f"{name}:loss"] = np.float64(loss.item())
        out[f"{name}:grad_x"] = xg.grad.numpy().copy()
        for k, v in sd.items():
            if k in keys:
                out[f"{name}:w:{k}"] = v
        for k, p_ in m.named_parameters():
            out[f"{name}:g:{k}"] = p_.grad.numpy().copy()
        print("g8", name, logits.detach().numpy()[0])
    np.savez_compressed(os.path.join(HERE, "g8_ablation.npz"), **out)


if __name__ == "__main__":
    torch.manual_seed(0)
    only = set(sys.argv[1:])
    g4 = load_ref("04_lstm_model.py")
    if only:          # e.g. `make_goldens.py g6 g7 g8`: regenerate just those fixtures
        if "g6" in only:
            g6_training(g4)
        if "g7" in only:
            g7_channel_importance(load_ref("07_explainability.py"))
        if "g8" in only:
            g8_ablation(load_ref("09_sensitivity_analysis.py"))
        if "g5" in only:
            g5_consumers(load_ref("06_lstm_ode_integration.py"), load_ref("08_forecasting.py"),
                         load_ref("10_three_state_probabilities.py"))
        sys.exit(0)
    g5 = load_ref("05_ode_model.py")
    g6 = load_ref("06_lstm_ode_integration.py")
    g1_tiny(g4)
    g2_full(g4)
    g3_ode(g5, g6)
    g4_coupled(g6)
    g5_consumers(g6, load_ref("08_forecasting.py"), load_ref("10_three_state_probabilities.py"))
    g6_training(g4)
    g7_channel_importance(load_ref("07_explainability.py"))
    g8_ablation(load_ref("09_sensitivity_analysis.py"))
