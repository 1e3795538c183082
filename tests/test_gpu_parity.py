"""GPU parity tests (run with ``-m gpu`` on an MI355X): the HIP path, called through the
C-ABI of liblob.so, against the oracle and the committed golden fixtures.

Tolerances (fp32 path; BASELINE.json configs[1]):
  logits / attention / context     <= 1e-5 abs vs the reference goldens and the oracle
  ODE trajectories (RK4, 16 sub)   <= 1e-6 abs vs the reference's LSODA output (budget 5e-6)
"""
import os

import numpy as np
import pytest
import torch

from lstm_ode_bci_amd import synthetic as syn

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
TOL = 1e-5


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    from lstm_ode_bci_amd import _lib
    assert _lib.lib().lob_version() >= 200
    return torch.device("cuda:0")


def _model(sd, C, H, L, bi, dev):
    from lstm_ode_bci_amd import EnhancedLSTMModel
    m = EnhancedLSTMModel(input_size=C, hidden_size=H, num_layers=L, num_classes=2, dropout=0.4,
                          bidirectional=bi)
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
    return m.to(dev).eval()


# ------------------------------------------------------------------------------------------
# kernels one by one
# ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("M,N,K,act", [(256, 128, 64, 0), (300, 130, 61, 2), (1000, 512, 256, 1),
                                       (5, 2, 64, 0), (129, 257, 33, 0), (4096, 1024, 128, 0)])
def test_gemm_nt(dev, M, N, K, act):
    from lstm_ode_bci_amd import ops
    rng = np.random.default_rng(M + N + K)
    a = torch.from_numpy(rng.standard_normal((M, K), dtype=np.float32)).to(dev)
    w = torch.from_numpy(rng.standard_normal((N, K), dtype=np.float32)).to(dev)
    b = torch.from_numpy(rng.standard_normal((N,), dtype=np.float32)).to(dev)
    out = ops.gemm_nt(a, w, b, act=act).cpu().double()
    ref = a.cpu().double() @ w.cpu().double().T + b.cpu().double()
    if act == 1:
        ref = torch.tanh(ref)
    elif act == 2:
        ref = torch.nn.functional.gelu(ref)
    assert (out - ref).abs().max().item() < 2e-5 * max(1.0, K ** 0.5)


def test_gemm_identity_asymmetric(dev):
    """A = I with an asymmetric W catches a transposed C write (cdna guide §3)."""
    from lstm_ode_bci_amd import ops
    n = 128
    a = torch.eye(n, device=dev)
    w = torch.arange(n * n, device=dev, dtype=torch.float32).reshape(n, n) / 7.0
    out = ops.gemm_nt(a, w)
    assert torch.equal(out, w.T.contiguous())


def test_layernorm_remap_and_gelu(dev):
    from lstm_ode_bci_amd import ops
    from oracle import restatement as R
    rng = np.random.default_rng(3)
    B, T, Wd = 5, 7, 128
    x = rng.standard_normal((B * T, Wd), dtype=np.float32) * 3 + 1
    g = rng.standard_normal(Wd).astype(np.float32)
    b = rng.standard_normal(Wd).astype(np.float32)
    Bp = 32
    out = ops.layernorm_act(torch.from_numpy(x).to(dev), torch.from_numpy(g).to(dev), torch.from_numpy(b).to(dev),
                            act=ops.ACT_GELU, remap=(T, B, Bp)).cpu().numpy().reshape(T, Bp, Wd)
    ref = R.gelu_erf(R.layer_norm(x.astype(np.float64), g, b)).reshape(B, T, Wd).transpose(1, 0, 2)
    assert np.abs(out[:, :B] - ref).max() < 2e-6
    assert np.all(out[:, B:] == 0)
    for Wd2 in (8, 16, 256, 512):
        x2 = rng.standard_normal((33, Wd2), dtype=np.float32)
        g2 = np.ones(Wd2, np.float32)
        b2 = np.zeros(Wd2, np.float32)
        o2 = ops.layernorm_act(torch.from_numpy(x2).to(dev), torch.from_numpy(g2).to(dev),
                               torch.from_numpy(b2).to(dev)).cpu().numpy()
        assert np.abs(o2 - R.layer_norm(x2.astype(np.float64), g2, b2)).max() < 2e-6


@pytest.mark.parametrize("H,D,B", [(128, 2, 40), (128, 1, 32), (8, 2, 3), (32, 2, 5), (64, 1, 9), (256, 2, 33), (96, 2, 7)])
def test_lstm_layer_vs_oracle(dev, H, D, B):
    """gate GEMM + persistent recurrent kernel (fast H=128 path and generic path) vs the numpy
    restatement of one layer."""
    from lstm_ode_bci_amd import ops
    from oracle import restatement as R
    rng = np.random.default_rng(H + D + B)
    T, K = 20, 48
    Bp = ops.ceil32(B)
    bound = 1 / np.sqrt(H)
    x = rng.standard_normal((B, T, K), dtype=np.float32)
    ws = [[rng.uniform(-bound, bound, s).astype(np.float32) for s in ((4 * H, K), (4 * H, H), (4 * H,), (4 * H,))]
          for _ in range(D)]
    xt = np.zeros((T, Bp, K), np.float32)
    xt[:, :B] = x.transpose(1, 0, 2)
    wih = torch.from_numpy(np.concatenate([w[0] for w in ws], 0)).to(dev)
    whh = torch.from_numpy(np.stack([w[1] for w in ws], 0)).to(dev)
    bias = torch.from_numpy(np.concatenate([w[2] + w[3] for w in ws], 0)).to(dev)
    for save in (False, True):
        P = ops.gate_gemm_x(torch.from_numpy(xt.reshape(T * Bp, K)).to(dev), wih, bias, T, Bp, H, D, ops.uses_frag(H))
        Y, Cs, _, _ = ops.lstm_rec_fwd(P, whh, T, Bp, H, D, save)
        y = Y.cpu().numpy().reshape(T, Bp, D * H)[:, :B].transpose(1, 0, 2)
        ref = np.concatenate([R.lstm_direction(x.astype(np.float64), *[w.astype(np.float64) for w in ws[d]], d == 1)
                              for d in range(D)], -1)
        assert np.abs(y - ref).max() < 5e-6, (H, D, save)


def test_attention_pool_vs_oracle(dev):
    from lstm_ode_bci_amd import Attention
    from oracle import restatement as R
    rng = np.random.default_rng(9)
    B, T, Wd = 6, 50, 64
    att = Attention(Wd).to(dev)
    v = rng.standard_normal((B, T, Wd), dtype=np.float32)
    ctx, w = att(torch.from_numpy(v).to(dev))
    assert ctx.requires_grad and w.requires_grad          # an ordinary trainable nn.Module, as 04_lstm_model.py:112-128
    ctx, w = ctx.detach(), w.detach()
    sd = {k: p.detach().cpu().numpy().astype(np.float64) for k, p in att.state_dict().items()}
    u = np.tanh(v @ sd["attention.0.weight"].T + sd["attention.0.bias"])
    s = u @ sd["attention.2.weight"].T + sd["attention.2.bias"]
    a = R.softmax(s, axis=1)
    assert np.abs(w.cpu().numpy() - a[..., 0]).max() < 1e-6
    assert np.abs(ctx.cpu().numpy() - (a * v).sum(1)).max() < 2e-6
    assert np.abs(w.cpu().numpy().sum(1) - 1).max() < 1e-6


# ------------------------------------------------------------------------------------------
# whole forward against the reference goldens
# ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("L", [1, 3])
@pytest.mark.parametrize("bi", [0, 1])
def test_forward_tiny_goldens(dev, L, bi):
    d = np.load(os.path.join(GOLDEN, f"g1_tiny_L{L}_bi{bi}.npz"))
    sd = {k[2:]: d[k] for k in d.files if k.startswith("w:")}
    m = _model(sd, 5, 8, L, bool(bi), dev)
    with torch.no_grad():
        logits, attn = m(torch.from_numpy(d["x"]).to(dev), return_attention=True)
    assert np.abs(logits.cpu().numpy() - d["logits"]).max() < TOL
    assert np.abs(attn.cpu().numpy() - d["attn"]).max() < TOL


@pytest.mark.parametrize("H", [128, 256])
def test_forward_full_size_goldens(dev, H):
    d = np.load(os.path.join(GOLDEN, f"g2_full_H{H}.npz"))
    x, _ = syn.make_windows(8)
    m = _model(syn.make_state_dict(61, H, 3, 2, True), 61, H, 3, True, dev)
    with torch.no_grad():
        logits, attn = m(torch.from_numpy(x).to(dev), return_attention=True)
    assert logits.shape == (8, 2) and attn.shape == (8, 256) and logits.dtype == torch.float32
    assert np.abs(logits.cpu().numpy() - d["logits"]).max() < TOL
    assert np.abs(attn.cpu().numpy() - d["attn"]).max() < TOL
    m3 = _model(syn.make_state_dict(61, H, 3, 2, True, lstm_scale=3.0), 61, H, 3, True, dev)
    with torch.no_grad():
        l3, a3 = m3(torch.from_numpy(x).to(dev), return_attention=True)
    assert np.abs(l3.cpu().numpy() - d["logits_stress"]).max() < 5 * TOL    # saturated gates
    assert np.abs(a3.cpu().numpy() - d["attn_stress"]).max() < TOL


def test_forward_config1_b1024_vs_oracle(dev):
    """BASELINE.json configs[1]: fwd-only, H=128, B=1024 (ragged: 1000), fp32, <= 1e-5 vs the CPU path."""
    from oracle import torch_cpu_path as TP
    B = 1000
    sd = syn.make_state_dict(61, 128, 3, 2, True)
    x, _ = syn.make_windows(B)
    m = _model(sd, 61, 128, 3, True, dev)
    with torch.no_grad():
        logits, attn = m(torch.from_numpy(x).to(dev), return_attention=True)
    ref = TP.build(sd, 61, 128)
    idx = np.r_[0:48, B - 48:B]                      # oracle on the first/last windows: seconds
    with torch.no_grad():
        rl, ra = ref(torch.from_numpy(x[idx]), return_attention=True)
    assert np.abs(logits.cpu().numpy()[idx] - rl.numpy()).max() < TOL
    assert np.abs(attn.cpu().numpy()[idx] - ra.numpy()).max() < TOL
    # batch-composition independence: every window is independent of its neighbours
    with torch.no_grad():
        l2 = m(torch.from_numpy(x[idx]).to(dev))
    assert np.abs(l2.cpu().numpy() - logits.cpu().numpy()[idx]).max() < 1e-6


# ------------------------------------------------------------------------------------------
# ODE + coupling
# ------------------------------------------------------------------------------------------
def test_ode_grid_vs_reference(dev):
    from lstm_ode_bci_amd import ops
    d = np.load(os.path.join(GOLDEN, "g3_ode.npz"))
    worst = 0.0
    for pname, rates in (("default", syn.DEFAULT_RATES), ("fitted", syn.FITTED_RATES)):
        for ai, alpha in enumerate(d["alphas"]):
            for steps in (10, 20, 300):
                key = f"{pname}_a{ai}_s{steps}"
                if "traj_" + key not in d.files:
                    continue
                probs = torch.from_numpy(d["probs_" + key]).to(dev)
                traj, final, pred = ops.ode_rk4([rates[k] for k in syn.RATE_KEYS], steps, 0.0, float(steps), 16,
                                                probs=probs, alpha=float(alpha), want_final=True)
                err = np.abs(traj.cpu().numpy() - d["traj_" + key]).max()
                worst = max(worst, err)
                assert err < 1e-6, (key, err)
                assert np.array_equal(pred.cpu().numpy(), d["pred_" + key]), key
                assert np.abs(final.cpu().numpy() - d["traj_" + key][:, -1]).max() < 1e-6
    print("worst ODE |err| vs LSODA:", worst)


def test_ode_class_solve(dev):
    from lstm_ode_bci_amd import CognitiveStateODE
    d = np.load(os.path.join(GOLDEN, "g3_ode.npz"))
    ode = CognitiveStateODE(dict(syn.FITTED_RATES))
    t, sol = ode.solve([0.5, 0.3, 0.2], (0, 7.5), 33)
    assert sol.dtype == np.float64 and sol.shape == (33, 3)
    assert np.array_equal(t, d["solve_t"])
    assert np.abs(sol - d["solve_sol"]).max() < 1e-6
    assert np.allclose(ode.ode_system([0.2, -0.1, 0.9], 0.0), d["ode_system"], atol=1e-15)
    assert np.array_equal(ode.get_transition_matrix(), d["q_matrix"])
    # conservation, steady state = null vector of Q^T
    assert np.abs(sol.sum(1) - 1).max() < 1e-12
    ss = CognitiveStateODE().get_steady_state()
    q = CognitiveStateODE().get_transition_matrix()
    assert np.abs(q.T @ np.array([ss["Active"], ss["Passive"], ss["Fatigued"]])).max() < 1e-8
    # the reference's RK45 branch (05:157-163): same kernel, inside that solver's own tolerance (rtol 1e-3) of scipy's output
    from oracle import restatement as R45
    for y0_, span, n in (([0.5, 0.3, 0.2], (0, 7.5), 33), ([0.9, 0.05, 0.05], (0, 30), 100), ([1, 1, 2], (0, 2), 5)):
        t5, s5 = ode.solve(y0_, span, n, method="solve_ivp")
        tr, sr = R45.solve_ivp_rk45(y0_, span, n, syn.FITTED_RATES)
        assert np.array_equal(t5, tr) and np.abs(s5 - sr).max() < 2e-3
        assert np.array_equal(s5, ode.solve(y0_, span, n)[1])
        assert np.array_equal(s5, ode.solve(y0_, span, n, method="RK45")[1])      # any non-'odeint' string: the else branch (05:157)
    # n_points = 1 and ragged batch sizes
    t1, s1 = ode.solve([1, 1, 2], (0, 5), 1)
    assert np.allclose(s1, [[0.25, 0.25, 0.5]])
    y0 = np.random.default_rng(0).uniform(0.1, 1, (131, 3))
    _, sb = ode.solve_batch(y0, (0, 20), 20)
    from oracle import restatement as R
    for i in (0, 64, 130):
        _, r = R.solve_odeint(y0[i], (0, 20), 20, syn.FITTED_RATES)
        assert np.abs(sb[i] - r).max() < 1e-6


def test_coupled_predict_batch_goldens(dev):
    from lstm_ode_bci_amd import CognitiveStateODE, LSTMODEIntegration
    d = np.load(os.path.join(GOLDEN, "g4_coupled.npz"))
    sd = syn.make_state_dict(61, 128, 3, 2, True)
    sd["classifier.6.weight"] = sd["classifier.6.weight"] * d["cls6_scale"]
    sd["classifier.6.bias"] = d["cls6_bias"]
    x, _ = syn.make_windows(32, seed=11)
    m = _model(sd, 61, 128, 3, True, dev)
    for pname, rates, alpha in (("default", syn.DEFAULT_RATES, 0.5), ("fitted", syn.FITTED_RATES, 0.5),
                                ("fitted_a1", syn.FITTED_RATES, 1.0)):
        ode = CognitiveStateODE(dict(rates))
        integ = LSTMODEIntegration(m, ode, coupling_strength=alpha)
        traj, probs, pred = integ.predict_batch(x, forecast_steps=20, batch_size=16, show_progress=False)
        assert traj.shape == (32, 20, 3) and traj.dtype == np.float64
        assert probs.shape == (32, 2) and probs.dtype == np.float32
        assert pred.shape == (32,) and pred.dtype == np.int64
        # logits are scaled x400 in this fixture, so 1e-5 logit error -> ~1e-3 in probability
        assert np.abs(probs - d["probs_" + pname]).max() < 2e-3
        stable = np.abs(d["probs_" + pname] - 0.6).min(1) > 5e-3      # away from the branch thresholds
        stable &= np.abs(d["probs_" + pname] - 0.4).min(1) > 5e-3
        assert np.abs(traj[stable] - d["traj_" + pname][stable]).max() < 2e-3
        far = np.abs(d["traj_" + pname][:, -1, 2] - 0.5) > 5e-3
        assert np.array_equal(pred[far & stable], d["pred_" + pname][far & stable])
        assert ode.params == dict(rates)
        t1, p1, a1 = integ.predict_trajectory(x[:1], forecast_steps=20)
        assert np.abs(t1 - traj[0]).max() < 1e-6 and p1.shape == (1, 2) and a1.shape == (1, 256)
        pr, at = integ.get_lstm_probabilities(x[:4])
        assert np.abs(pr - probs[:4]).max() < 1e-6 and at.shape == (4, 256)
        # exactness of stage 2 on the reference's own probabilities
        from lstm_ode_bci_amd import ops
        tr2, _, pd2 = ops.ode_rk4([rates[k] for k in syn.RATE_KEYS], 20, 0.0, 20.0, 16,
                                  probs=torch.from_numpy(d["probs_" + pname]).to(dev), alpha=alpha)
        assert np.abs(tr2.cpu().numpy() - d["traj_" + pname]).max() < 1e-6
        assert np.array_equal(pd2.cpu().numpy(), d["pred_" + pname])


# ------------------------------------------------------------------------------------------
# backward: gradients of every parameter and of the input vs the reference (goldens + oracle)
# ------------------------------------------------------------------------------------------
def _grads(m, x, y, dev):
    m.zero_grad(set_to_none=True)
    xg = torch.from_numpy(x).to(dev).requires_grad_(True)
    loss = torch.nn.functional.cross_entropy(m(xg), torch.from_numpy(y).to(dev))
    loss.backward()
    return float(loss), {k: p.grad.detach().cpu().numpy() for k, p in m.named_parameters()}, xg.grad.cpu().numpy()


def _close(a, b, rtol=2e-4, atol=2e-6):
    return np.abs(a - b).max() <= atol + rtol * np.abs(b).max()


@pytest.mark.parametrize("M,N,Kc", [(128, 128, 4096), (1024, 256, 10000), (61, 128, 999), (2, 64, 37), (512, 128, 70000)])
def test_gemm_tn_and_colsum(dev, M, N, Kc):
    from lstm_ode_bci_amd import ops
    rng = np.random.default_rng(M + N)
    a = torch.from_numpy(rng.standard_normal((Kc, M), dtype=np.float32)).to(dev)
    b = torch.from_numpy(rng.standard_normal((Kc, N), dtype=np.float32)).to(dev)
    out = torch.zeros((M, N), device=dev)
    ops.gemm_tn(a, b, out)
    ref = a.cpu().double().T @ b.cpu().double()
    assert (out.cpu().double() - ref).abs().max().item() < 3e-5 * Kc ** 0.5
    cs = ops.colsum(a).cpu().double()
    assert (cs - a.cpu().double().sum(0)).abs().max().item() < 3e-5 * Kc ** 0.5


@pytest.mark.parametrize("L", [1, 3])
@pytest.mark.parametrize("bi", [0, 1])
def test_backward_tiny_goldens(dev, L, bi):
    d = np.load(os.path.join(GOLDEN, f"g1_tiny_L{L}_bi{bi}.npz"))
    sd = {k[2:]: d[k] for k in d.files if k.startswith("w:")}
    m = _model(sd, 5, 8, L, bool(bi), dev)
    loss, gp, gx = _grads(m, d["x"], d["y"], dev)
    assert abs(loss - float(d["loss"])) < 1e-5
    assert _close(gx, d["grad_x"]), np.abs(gx - d["grad_x"]).max()
    for k, g in gp.items():
        assert _close(g, d["g:" + k]), (k, np.abs(g - d["g:" + k]).max())


def test_backward_full_size_vs_reference(dev):
    from oracle import torch_cpu_path as TP
    d = np.load(os.path.join(GOLDEN, "g2_full_H128.npz"))
    sd = syn.make_state_dict(61, 128, 3, 2, True)
    x, y = syn.make_windows(8)
    m = _model(sd, 61, 128, 3, True, dev)
    loss, gp, gx = _grads(m, x, y, dev)
    assert abs(loss - float(d["loss"])) < 1e-5
    names = [str(n) for n in d["grad_names"]]
    l2 = np.array([np.sqrt((gp[k].astype(np.float64) ** 2).sum()) for k in names])
    assert np.allclose(l2, d["grad_l2"], rtol=5e-4, atol=1e-8), np.abs(l2 / np.maximum(d["grad_l2"], 1e-30) - 1).max()
    assert _close(gx[:, ::32], d["grad_x_slice"])
    assert _close(gp["classifier.6.weight"], d["grad_cls6_w"])
    assert _close(gp["lstm.weight_hh_l2"][::64, ::16], d["grad_whh_l2_slice"])
    assert _close(gp["lstm.weight_ih_l0_reverse"][::64, ::16], d["grad_wih_l0r_slice"])
    assert _close(gp["input_proj.0.weight"][::16], d["grad_proj_w_slice"])
    # every element of every gradient against the oracle's autograd (ragged batch: 37 windows)
    x2, y2 = syn.make_windows(37, seed=3)
    ref = TP.build(sd, 61, 128)
    rl, rgp, rgx = TP.loss_and_grads(ref, torch.from_numpy(x2), torch.from_numpy(y2))
    l2_, gp2, gx2 = _grads(m, x2, y2, dev)
    assert abs(l2_ - rl) < 1e-5
    assert _close(gx2, rgx)
    for k in rgp:
        assert _close(gp2[k], rgp[k]), (k, np.abs(gp2[k] - rgp[k]).max(), np.abs(rgp[k]).max())


def test_backward_retained_graph_and_input_grads(dev):
    """07_explainability.py:242-257: one forward, B backward calls on a retained graph, reading X.grad."""
    sd = syn.make_state_dict(61, 128, 3, 2, True)
    x, _ = syn.make_windows(4, 64, 61, seed=2)
    m = _model(sd, 61, 128, 3, True, dev)
    xg = torch.from_numpy(x).to(dev).requires_grad_(True)
    out = m(xg)
    grads = []
    for i in range(4):
        m.zero_grad()
        if xg.grad is not None:
            xg.grad.zero_()
        out[i, 1].backward(retain_graph=True)
        grads.append(xg.grad.clone())
    for i in range(4):          # window i's logit depends on window i only
        for j in range(4):
            if i != j:
                assert grads[i][j].abs().max().item() == 0.0
        assert grads[i][i].abs().max().item() > 0
    m.zero_grad(); xg.grad.zero_()
    out[0, 1].backward(retain_graph=True)
    assert torch.equal(xg.grad, grads[0])            # saved activations were not modified


def test_train_mode_dropout(dev):
    sd = syn.make_state_dict(61, 128, 3, 2, True)
    x, y = syn.make_windows(16, 32, 61, seed=4)
    m = _model(sd, 61, 128, 3, True, dev).train()
    torch.manual_seed(123)
    l1, g1, gx1 = _grads(m, x, y, dev)
    torch.manual_seed(123)
    l2, g2, gx2 = _grads(m, x, y, dev)
    assert l1 == l2 and np.array_equal(gx1, gx2)      # same torch seed -> same masks
    l3, _, _ = _grads(m, x, y, dev)
    assert l3 != l1                                  # fresh masks on the next call
    m.eval()
    l4, _, _ = _grads(m, x, y, dev)
    assert l4 != l1
    # the masks keep ~ (1-p) of the elements, scaled by 1/(1-p)
    from lstm_ode_bci_amd import ops
    ones = torch.ones(1 << 20, device=dev)
    dr = ops.dropout(ones, 0.4, 777)
    keep = (dr > 0).float().mean().item()
    assert abs(keep - 0.6) < 5e-3 and abs(dr.max().item() - 1 / 0.6) < 1e-6
    # finite-difference check of the train-mode backward under a FIXED seed (dropout included)
    from lstm_ode_bci_amd.autograd import _LobModelFn, _collect
    cfg = (3, 2, 128, (0.2, 0.4, 0.4), 99, False)
    xs = torch.from_numpy(x[:4]).to(dev)
    ps = _collect(m)
    w = ps[-2]                                        # classifier.6.weight
    yy = torch.from_numpy(y[:4]).to(dev)

    def f():
        return torch.nn.functional.cross_entropy(_LobModelFn.apply(xs, cfg, None, *ps)[0], yy)
    m.zero_grad()
    f().backward()
    ga = w.grad[0, 3].item()
    with torch.no_grad():
        w[0, 3] += 1e-2
        lp = f().item()
        w[0, 3] -= 2e-2
        lm = f().item()
        w[0, 3] += 1e-2
    assert abs((lp - lm) / 2e-2 - ga) < 2e-3 * max(1.0, abs(ga))


# ------------------------------------------------------------------------------------------
# mixed precision (BASELINE.json configs[2]: bf16 gate-GEMMs + fp32 recurrence).
# Tolerances: logits <= 5e-3 abs, gradients <= 2e-2 of max|ref| per tensor.
# ------------------------------------------------------------------------------------------
def _bf16_round(a):
    return torch.from_numpy(a).to(torch.bfloat16).double()


@pytest.mark.parametrize("M,N,K", [(256, 128, 64), (1000, 1024, 256), (4096, 256, 1024), (130, 130, 72)])
def test_gemm_nt_bf16(dev, M, N, K):
    from lstm_ode_bci_amd import ops
    rng = np.random.default_rng(M + K)
    a = rng.standard_normal((M, K), dtype=np.float32)
    w = rng.standard_normal((N, K), dtype=np.float32)
    b = rng.standard_normal((N,), dtype=np.float32)
    ref = _bf16_round(a) @ _bf16_round(w).T + torch.from_numpy(b).double()
    out = ops.gemm_nt(torch.from_numpy(a).to(dev), torch.from_numpy(w).to(dev), torch.from_numpy(b).to(dev), mixed=True)
    assert (out.cpu().double() - ref).abs().max().item() < 1e-5 * K ** 0.5 * 4
    out2 = ops.gemm_nt(torch.from_numpy(a).to(dev).to(torch.bfloat16), torch.from_numpy(w).to(dev),
                       torch.from_numpy(b).to(dev))
    assert (out2.cpu().double() - ref).abs().max().item() < 1e-5 * K ** 0.5 * 4
    acc = out.clone()
    ops.gemm_nt(torch.from_numpy(a).to(dev), torch.from_numpy(w).to(dev), None, out=acc, accumulate=True, mixed=True)
    assert (acc.cpu().double() - (2 * ref - torch.from_numpy(b).double())).abs().max().item() < 1e-4 * K ** 0.5


@pytest.mark.parametrize("M,N,Kc", [(128, 128, 4096), (1024, 256, 20000), (512, 128, 70001), (136, 72, 999)])
def test_gemm_tn_bf16(dev, M, N, Kc):
    from lstm_ode_bci_amd import ops
    rng = np.random.default_rng(M + N + 1)
    a = rng.standard_normal((Kc, M), dtype=np.float32)
    b = rng.standard_normal((Kc, N), dtype=np.float32)
    ref = _bf16_round(a).T @ _bf16_round(b)
    tol = 4e-5 * Kc ** 0.5
    for a16 in (False, True):
        for b16 in (False, True):
            ta = torch.from_numpy(a).to(dev)
            tb = torch.from_numpy(b).to(dev)
            if a16:
                ta = ta.to(torch.bfloat16)
            if b16:
                tb = tb.to(torch.bfloat16)
            out = torch.zeros((M, N), device=dev)
            ops.gemm_tn(ta, tb, out, mixed=True)
            assert (out.cpu().double() - ref).abs().max().item() < tol, (a16, b16)
    # column-sliced operands with a row offset, as the dW_hh call uses them
    if M >= 256:
        ta = torch.from_numpy(a).to(dev).to(torch.bfloat16)
        tb = torch.from_numpy(b).to(dev)
        out = torch.zeros((128, 64), device=dev)
        ops.gemm_tn(ta[32:, 128:256], tb[:Kc - 32, 64:128], out)
        ref2 = _bf16_round(a)[32:, 128:256].T @ _bf16_round(b)[:Kc - 32, 64:128]
        assert (out.cpu().double() - ref2).abs().max().item() < tol
    cs = ops.colsum(torch.from_numpy(a).to(dev).to(torch.bfloat16)).cpu().double()
    assert (cs - _bf16_round(a).sum(0)).abs().max().item() < tol


@pytest.mark.parametrize("pg_bf16", [False, True])
def test_mixed_precision_forward_backward(dev, monkeypatch, pg_bf16):
    from oracle import torch_cpu_path as TP
    from lstm_ode_bci_amd import ops
    monkeypatch.setattr(ops, "PG_BF16", pg_bf16)
    sd = syn.make_state_dict(61, 128, 3, 2, True)
    x, y = syn.make_windows(40, seed=8)
    m = _model(sd, 61, 128, 3, True, dev)
    ref = TP.build(sd, 61, 128)
    rl, rgp, rgx = TP.loss_and_grads(ref, torch.from_numpy(x), torch.from_numpy(y))
    with torch.no_grad():
        rlog = ref(torch.from_numpy(x)).numpy()
    with torch.autocast("cuda", dtype=torch.bfloat16):
        with torch.no_grad():
            logits, attn = m(torch.from_numpy(x).to(dev), return_attention=True)
        assert logits.dtype == torch.float32
        assert np.abs(logits.cpu().numpy() - rlog).max() < 5e-3
        m.zero_grad(set_to_none=True)
        xg = torch.from_numpy(x).to(dev).requires_grad_(True)
        loss = torch.nn.functional.cross_entropy(m(xg).float(), torch.from_numpy(y).to(dev))
    loss.backward()
    assert abs(float(loss) - rl) < 5e-3
    worst = 0.0
    for k, p in m.named_parameters():
        if np.abs(rgp[k]).max() < 1e-7:
            continue                                  # attention.2.bias: exactly 0 (cancels in the softmax)
        err = np.abs(p.grad.cpu().numpy() - rgp[k]).max() / np.abs(rgp[k]).max()
        worst = max(worst, err)
        assert err < 2e-2, (k, err)
    assert np.abs(xg.grad.cpu().numpy() - rgx).max() / np.abs(rgx).max() < 2e-2
    print("mixed (P/G bf16=%s): logits err %.2e, worst rel grad err %.2e, grad_x rel err %.2e" %
          (pg_bf16, np.abs(logits.cpu().numpy() - rlog).max(), worst,
           np.abs(xg.grad.cpu().numpy() - rgx).max() / np.abs(rgx).max()))
    # the attribute switch selects the same path without autocast
    m.gate_gemm_dtype = "bf16"
    with torch.no_grad():
        l2 = m(torch.from_numpy(x).to(dev))
    assert torch.equal(l2, logits)


# ------------------------------------------------------------------------------------------
# §8f consumers: 10_three_state_probabilities.py and 08_forecasting.py as batched device calls
# ------------------------------------------------------------------------------------------
def test_consumers_vs_reference(dev):
    from lstm_ode_bci_amd import CognitiveStateODE, consumers
    d = np.load(os.path.join(GOLDEN, "g5_consumers.npz"))
    d4 = np.load(os.path.join(GOLDEN, "g4_coupled.npz"))
    sd = syn.make_state_dict(61, 128, 3, 2, True)
    sd["classifier.6.weight"] = sd["classifier.6.weight"] * d4["cls6_scale"]
    sd["classifier.6.bias"] = d4["cls6_bias"]
    x, _ = syn.make_windows(32, seed=11)
    m = _model(sd, 61, 128, 3, True, dev)
    for pname, rates in (("default", syn.DEFAULT_RATES), ("fitted", syn.FITTED_RATES)):
        ode = CognitiveStateODE(dict(rates))
        lp, three, pred = consumers.get_three_state_probabilities(m, ode, x, batch_size=16)
        assert lp.shape == (32, 2) and three.shape == (32, 3) and three.dtype == np.float64
        assert np.abs(lp - d[f"three_lstm_probs_{pname}"]).max() < 2e-3          # logits scaled x400
        stable = (np.abs(d[f"three_lstm_probs_{pname}"] - 0.6).min(1) > 5e-3) & \
                 (np.abs(d[f"three_lstm_probs_{pname}"] - 0.4).min(1) > 5e-3)
        assert np.abs(three[stable] - d[f"three_state_{pname}"][stable]).max() < 2e-3
        far = (np.abs(d[f"three_state_{pname}"] - 0.5).min(1) > 5e-3) & stable
        assert np.array_equal(pred[far], d[f"three_pred_{pname}"][far])
        assert ode.params == dict(rates)
        # 08: multistep forecast from a probability series -- exact inputs, so tight tolerance
        res = consumers.multistep_forecast(d["fc_probs"], dict(rates), horizons=[5, 10, 20])
        for h in (5, 10, 20):
            assert np.abs(res[h]["predictions"] - d[f"fc_pred_{pname}_h{h}"]).max() < 1e-6
            assert np.array_equal(res[h]["actuals"], d[f"fc_act_{pname}_h{h}"])
        df = consumers.rolling_forecast_evaluation(d["roll_probs"], dict(rates), window_size=20, horizon=10)
        gold = d[f"roll_{pname}"]
        assert list(df.columns) == ["window", "accuracy", "mae"] and len(df) == len(gold)
        assert np.array_equal(df["window"].to_numpy(), gold[:, 0])
        assert np.abs(df["mae"].to_numpy() - gold[:, 2]).max() < 1e-6
        assert np.abs(df["accuracy"].to_numpy() - gold[:, 1]).max() <= 1 / 20 + 1e-12   # one 0.5-boundary flip at most
        assert len(consumers.rolling_forecast_evaluation(d["roll_probs"][:25], dict(rates), 20, 10)) == 0
        tr = consumers.predict_trajectory(consumers.prob_to_ode_state(d["fc_probs"][5, 1]), dict(rates), 20)
        assert tr.shape == (21, 3) and np.abs(tr - d[f"fc_traj_{pname}"]).max() < 1e-6
    pr = consumers.get_lstm_probabilities(m, x, batch_size=8)
    assert np.abs(pr - d["three_lstm_probs_default"]).max() < 2e-3


@pytest.mark.parametrize("H,L,bi", [(256, 3, True), (64, 2, True), (32, 1, False)])
def test_streaming_kernels_forward_backward_vs_oracle(dev, H, L, bi):
    """H = 256 (the reference's real checkpoints) / 64 / 32: MFMA kernels with W_hh streamed from L2."""
    from oracle import torch_cpu_path as TP
    from lstm_ode_bci_amd import ops
    assert ops.uses_frag(H)
    sd = syn.make_state_dict(61, H, L, 2, bi, seed=77)
    x, y = syn.make_windows(37, 48, 61, seed=6)
    m = _model(sd, 61, H, L, bi, dev)
    ref = TP.build(sd, 61, H, L, 2, bi)
    rl, rgp, rgx = TP.loss_and_grads(ref, torch.from_numpy(x), torch.from_numpy(y))
    loss, gp, gx = _grads(m, x, y, dev)
    assert abs(loss - rl) < 1e-5
    assert _close(gx, rgx)
    for k in rgp:
        assert _close(gp[k], rgp[k]), (k, np.abs(gp[k] - rgp[k]).max(), np.abs(rgp[k]).max())
    with torch.autocast("cuda", dtype=torch.bfloat16):      # mixed mode: bf16 GEMMs around the fp32 stream kernels
        m.zero_grad(set_to_none=True)
        xg = torch.from_numpy(x).to(dev).requires_grad_(True)
        l2 = torch.nn.functional.cross_entropy(m(xg).float(), torch.from_numpy(y).to(dev))
    l2.backward()
    assert abs(float(l2) - rl) < 5e-3
    for k, p in m.named_parameters():
        if np.abs(rgp[k]).max() > 1e-7:
            assert np.abs(p.grad.cpu().numpy() - rgp[k]).max() / np.abs(rgp[k]).max() < 2e-2, k


def test_fused_dropout_equals_separate_kernels(dev, monkeypatch):
    """Mixed mode fuses nn.LSTM's inter-layer dropout into the recurrent kernel (bf16 copy) and its
    backward into the dX GEMM epilogue; the masks are the same counter-based hash, so the result must
    equal the un-fused pipeline (dropout kernel -> GEMM) to rounding."""
    from lstm_ode_bci_amd import ops
    sd = syn.make_state_dict(61, 128, 3, 2, True)
    x, y = syn.make_windows(40, 64, 61, seed=12)
    m = _model(sd, 61, 128, 3, True, dev).train()

    def run():
        torch.manual_seed(5)
        m.zero_grad(set_to_none=True)
        xg = torch.from_numpy(x).to(dev).requires_grad_(True)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            loss = torch.nn.functional.cross_entropy(m(xg).float(), torch.from_numpy(y).to(dev))
        loss.backward()
        return float(loss), {k: p.grad.clone() for k, p in m.named_parameters()}, xg.grad.clone()

    lf, gf, xf = run()
    monkeypatch.setattr(ops, "can_fuse_dropout", lambda H, mixed: False)
    lu, gu, xu = run()
    # the two pipelines run different GEMM kernels (LDS-DMA vs register-staged: other accumulation order),
    # so equality is to fp32 rounding, not bitwise; a mask mismatch would be an O(1) difference
    # (the fused path also rounds h to bf16 BEFORE the 1/(1-p) scaling: one extra bf16 rounding)
    assert abs(lf - lu) < 1e-4
    assert (xf - xu).abs().max().item() <= 1e-2 * xu.abs().max().item()
    for k in gf:          # (5e-2: bf16 rounding noise of the two pipelines is uncorrelated; masks that differ give O(1))
        assert (gf[k] - gu[k]).abs().max().item() <= 5e-2 * max(1e-6, gu[k].abs().max().item()) + 1e-9, k


@pytest.mark.parametrize("M,N,K", [(4096, 1024, 256), (8192, 256, 1024), (1000, 256, 128), (128, 128, 512)])
def test_gemm_nt_lds_dma(dev, M, N, K):
    """bf16 x bf16 NT GEMM through the LDS-DMA kernel (row-major epilogue, ragged M, fused dropout)."""
    from lstm_ode_bci_amd import ops
    rng = np.random.default_rng(M + N + K)
    a = torch.from_numpy(rng.standard_normal((M, K), dtype=np.float32)).to(dev).to(torch.bfloat16)
    w = torch.from_numpy(rng.standard_normal((N, K), dtype=np.float32)).to(dev).to(torch.bfloat16)
    ref = a.cpu().double() @ w.cpu().double().T
    out = ops.gemm_nt(a, w)
    assert (out.cpu().double() - ref).abs().max().item() < 1e-5 * K ** 0.5 * 4
    # twice in a row on the same buffers (persistent ring must start clean), and with the dropout epilogue
    out2 = ops.gemm_nt(a, w, drop_p=0.4, seed=99)
    mask = ops.dropout(torch.ones((M, N), device=dev), 0.4, 99)
    assert torch.equal(out2, out * mask)


def test_gate_gemm_lds_dma_fragment_layout(dev):
    """bf16 x bf16 gate GEMM (fragment epilogue, fp32 and bf16 P) == the register-staged kernel."""
    from lstm_ode_bci_amd import ops
    rng = np.random.default_rng(5)
    T, Bp, H, D, K = 6, 128, 128, 2, 256
    x = torch.from_numpy(rng.standard_normal((T * Bp, K), dtype=np.float32)).to(dev).to(torch.bfloat16)
    wih = torch.from_numpy(rng.uniform(-0.1, 0.1, (D * 4 * H, K)).astype(np.float32)).to(dev)
    bias = torch.from_numpy(rng.standard_normal(D * 4 * H).astype(np.float32)).to(dev)
    for pg in (False, True):
        ops.PG_BF16 = pg
        try:
            p_reg = ops.gate_gemm_x(x, wih, bias, T, Bp, H, D, True, mixed=True)
            p_dma = ops.gate_gemm_x(x, wih.to(torch.bfloat16), bias, T, Bp, H, D, True, mixed=True)
        finally:
            ops.PG_BF16 = True
        assert p_reg.dtype == p_dma.dtype == (torch.bfloat16 if pg else torch.float32)
        # the fp32 weights are rounded to bf16 inside the register-staged kernel: same products
        assert (p_reg.float() - p_dma.float()).abs().max().item() < (2e-2 if pg else 1e-5)


# ------------------------------------------------------------------------------------------
# BASELINE.json full sizes: size-independent properties (the oracle cannot run these in seconds)
# ------------------------------------------------------------------------------------------
def test_full_size_b4096_properties(dev):
    """B = 4096 (configs[2] size), fp32 path: window independence, attention normalisation, oracle on a
    sample, gradient additivity over the batch (sum-reduced loss), and the mixed path within tolerance."""
    from oracle import torch_cpu_path as TP
    B = 4096
    sd = syn.make_state_dict(61, 128, 3, 2, True)
    x, y = syn.make_windows(B)
    m = _model(sd, 61, 128, 3, True, dev)
    xt, yt = torch.from_numpy(x).to(dev), torch.from_numpy(y).to(dev)
    with torch.no_grad():
        logits, attn = m(xt, return_attention=True)
        sub = torch.cat([xt[:40], xt[-24:]])
        l_sub = m(sub)
    assert torch.isfinite(logits).all() and logits.shape == (B, 2)
    assert (attn.sum(1) - 1).abs().max().item() < 2e-6 and attn.min().item() >= 0
    assert (l_sub - torch.cat([logits[:40], logits[-24:]])).abs().max().item() < 1e-6      # independence
    idx = np.r_[0:8, 2044:2052, B - 8:B]
    ref = TP.build(sd, 61, 128)
    with torch.no_grad():
        rl, ra = ref(torch.from_numpy(x[idx]), return_attention=True)
    assert np.abs(logits.cpu().numpy()[idx] - rl.numpy()).max() < TOL
    assert np.abs(attn.cpu().numpy()[idx] - ra.numpy()).max() < TOL

    def grads(xs, ys):
        m.zero_grad(set_to_none=True)
        torch.nn.functional.cross_entropy(m(xs), ys, reduction="sum").backward()
        return {k: p.grad.clone() for k, p in m.named_parameters()}
    g_all = grads(xt, yt)
    g_a, g_b = grads(xt[:1504], yt[:1504]), grads(xt[1504:], yt[1504:])        # ragged split
    for k in g_all:
        ref_ = g_a[k] + g_b[k]
        assert (g_all[k] - ref_).abs().max().item() <= 3e-4 * ref_.abs().max().item() + 1e-7, k
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        lm = m(xt)
    assert (lm - logits).abs().max().item() < 5e-3


def test_full_size_coupled_b65536_properties(dev):
    """Global batch of configs[4] through the ODE stage on one device: conservation, bounds, decision rule,
    and agreement with the closed form on a sample."""
    from lstm_ode_bci_amd import ops
    from oracle import restatement as R
    probs_np = syn.make_probs(65536)
    probs = torch.from_numpy(probs_np).to(dev)
    rates = [syn.FITTED_RATES[k] for k in syn.RATE_KEYS]
    traj, final, pred = ops.ode_rk4(rates, 300, 0.0, 300.0, 16, probs=probs, alpha=1.0, want_final=True)
    assert traj.shape == (65536, 300, 3)
    assert (traj.sum(2) - 1).abs().max().item() < 1e-12 and traj.min().item() >= 0 and traj.max().item() <= 1
    assert torch.equal(final, traj[:, -1]) and torch.equal(pred, (traj[:, -1, 2] > 0.5).long())
    for i in (0, 777, 65535):
        mp = R.modulate_rates(syn.FITTED_RATES, 1.0, probs_np[i, 1], probs_np[i, 0])
        _, te = R.solve_expm(R.initial_state_rule(probs_np[i, 1], probs_np[i, 0]), (0, 300), 300, mp)
        assert np.abs(traj[i].cpu().numpy() - te).max() < 1e-6


@pytest.mark.parametrize("H", [128])
def test_forward_full_size_goldens_exact_fp32_mfma_kernels(dev, H):
    """The exact-fp32 MFMA kernels (LOB_VAR_F32_SPLIT = 0; the default fp32 path runs the two-way fp16 split) against
    the same reference goldens, <= 1e-5."""
    from lstm_ode_bci_amd import _lib
    d = np.load(os.path.join(GOLDEN, f"g2_full_H{H}.npz"))
    sd = syn.make_state_dict(61, H, 3, 2, True)
    x, _ = syn.make_windows(8)
    m = _model(sd, 61, H, 3, True, dev)
    with _lib.variant(F32_SPLIT=0), torch.no_grad():
        logits, attn = m(torch.from_numpy(x).to(dev), return_attention=True)
    with torch.no_grad():
        l2, a2 = m(torch.from_numpy(x).to(dev), return_attention=True)
    assert np.abs(logits.cpu().numpy() - d["logits"]).max() < TOL and np.abs(attn.cpu().numpy() - d["attn"]).max() < TOL
    print(f"H={H}: |logits split - exact| {float((l2 - logits).abs().max()):.2e}, split vs golden "
          f"{np.abs(l2.cpu().numpy() - d['logits']).max():.2e}, exact vs golden {np.abs(logits.cpu().numpy() - d['logits']).max():.2e}")
    assert np.abs(l2.cpu().numpy() - d["logits"]).max() < TOL


@pytest.mark.parametrize("H", [128, 256])
def test_serving_shapes_one_to_five_windows_vs_oracle(dev, H):
    """The serving call of 06_lstm_ode_integration.py:340-360 with a handful of windows (B = 1 .. 5: one 16-row tile
    per direction, the few-window recurrent kernel below 4): fp32 logits / attention <= 1e-5 vs the CPU path, the mixed
    path (autocast, 06:349) within its 5e-3, and every window independent of how many others share the call."""
    from oracle import torch_cpu_path as TP
    sd = syn.make_state_dict(61, H, 3, 2, True)
    x, _ = syn.make_windows(5)
    m = _model(sd, 61, H, 3, True, dev)
    ref = TP.build(sd, 61, H)
    with torch.no_grad():
        rl, ra = ref(torch.from_numpy(x), return_attention=True)
    xt = torch.from_numpy(x).to(dev)
    for Bn in (1, 2, 3, 4, 5):
        with torch.no_grad():
            lg, at = m(xt[:Bn], return_attention=True)
            with torch.autocast("cuda", dtype=torch.bfloat16):
                lm, am = m(xt[:Bn], return_attention=True)
        assert lg.shape == (Bn, 2) and at.shape == (Bn, 256)
        assert np.abs(lg.cpu().numpy() - rl.numpy()[:Bn]).max() < TOL, Bn
        assert np.abs(at.cpu().numpy() - ra.numpy()[:Bn]).max() < TOL, Bn
        assert np.abs(lm.float().cpu().numpy() - rl.numpy()[:Bn]).max() < 5e-3, Bn
        assert np.abs(am.float().cpu().numpy() - ra.numpy()[:Bn]).max() < 5e-3, Bn
        assert abs(float(am.sum(1).mean()) - 1.0) < 1e-3
