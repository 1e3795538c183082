"""GPU parity tests of the steps either side of fwd+bwd: weighted CE, clip + AdamW on
flat buffers, the training harness, the ablation variants, batched gradient attribution.  Checked against the
fixtures captured from the reference's own train_model / AblationLSTMModel / compute_channel_importance
(tests/golden/g6-g8) and against the oracle on larger shapes."""
import os

import numpy as np
import pytest
import torch

from lstm_ode_bci_amd import synthetic as syn

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    from lstm_ode_bci_amd import _lib
    assert _lib.lib().lob_version() >= 200
    return torch.device("cuda:0")


def _load(m, sd, dev):
    keys = set(m.state_dict().keys())
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items() if k in keys}, strict=True)
    return m.to(dev)


# ------------------------------------------------------------------------------------------
# kernels
# ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("B,C,weighted", [(7, 2, True), (512, 2, True), (4096, 2, False), (33, 5, True), (1, 2, True)])
def test_weighted_ce_vs_torch(dev, B, C, weighted):
    from lstm_ode_bci_amd import ops
    rng = np.random.default_rng(B + C)
    z = torch.from_numpy(rng.standard_normal((B, C)).astype(np.float32) * 3).to(dev)
    y = torch.from_numpy(rng.integers(0, C, B)).to(dev)
    w = torch.from_numpy(rng.uniform(0.2, 2.0, C).astype(np.float32)).to(dev) if weighted else None
    loss, dl, correct = ops.weighted_ce(z, y, w, scale=0.25)
    zr = z.double().cpu().requires_grad_(True)
    ref = torch.nn.functional.cross_entropy(zr, y.cpu(), weight=None if w is None else w.double().cpu())
    ref.backward()
    assert abs(loss.item() - ref.item()) < 2e-6 * max(1.0, abs(ref.item()))
    assert (dl.cpu().double() - 0.25 * zr.grad).abs().max().item() < 1e-7
    assert int(correct.item()) == int((z.argmax(1) == y).sum().item())


@pytest.mark.parametrize("n", [1, 3, 64, 1000, 1137731, 4_000_001])
def test_sumsq_clip_adamw_flat_vs_torch(dev, n):
    from lstm_ode_bci_amd import ops
    rng = np.random.default_rng(n)
    p0 = torch.from_numpy(rng.standard_normal(n).astype(np.float32)).to(dev)
    gs = [torch.from_numpy((rng.standard_normal(n) * s).astype(np.float32)).to(dev) for s in (3.0, 1e-4, 0.5)]
    # reference: torch.optim.AdamW + clip_grad_norm_ on a fp64 copy
    pr = torch.nn.Parameter(p0.double().cpu())
    opt = torch.optim.AdamW([pr], lr=3e-3, weight_decay=0.05, betas=(0.9, 0.999), eps=1e-8)
    p = p0.clone()
    m = torch.zeros_like(p)
    v = torch.zeros_like(p)
    for step, g in enumerate(gs, 1):
        nsq = ops.sumsq(g)
        assert abs(nsq.item() - float((g.double() ** 2).sum())) < 1e-4 * float((g.double() ** 2).sum()) + 1e-30
        ops.adamw_(p, g, m, v, step, 3e-3, (0.9, 0.999), 1e-8, 0.05, normsq=nsq, max_norm=1.0)
        pr.grad = g.double().cpu()
        torch.nn.utils.clip_grad_norm_([pr], 1.0)
        opt.step()
        assert (p.double().cpu() - pr.detach()).abs().max().item() < 5e-6, step
    # in-place clip
    g = gs[0].clone()
    ops.clip_scale_(g, ops.sumsq(g), 1.0)
    assert abs(g.double().norm().item() - min(1.0, gs[0].double().norm().item())) < 1e-4


def test_abs_colsum(dev):
    from lstm_ode_bci_amd import ops
    rng = np.random.default_rng(2)
    for rows, C in ((12 * 7, 5), (256 * 33, 61), (1000, 130)):
        g = torch.from_numpy(rng.standard_normal((rows, C)).astype(np.float32)).to(dev)
        out = torch.ones(C, device=dev)
        ops.abs_colsum(g, out, scale=0.5)
        ref = 1.0 + 0.5 * g.double().abs().sum(0)
        assert (out.double() - ref).abs().max().item() < 1e-4 * ref.max().item()


# ------------------------------------------------------------------------------------------
# ablation variants (09_sensitivity_analysis.py:176-242)
# ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", ["full", "noattn", "noln", "minimal", "bare"])
def test_ablation_variants_vs_reference(dev, name):
    from lstm_ode_bci_amd import AblationLSTMModel
    d = np.load(os.path.join(GOLDEN, "g8_ablation.npz"))
    L, bi, att, ln = (int(v) for v in d[name + ":cfg"])
    pre = name + ":w:"
    sd = {k[len(pre):]: d[k] for k in d.files if k.startswith(pre)}
    m = AblationLSTMModel(5, 8, L, 2, 0.4, bool(bi), bool(att), bool(ln))
    assert set(m.state_dict().keys()) == set(sd)          # same keys as the reference class
    m = _load(m, sd, dev).eval()
    x = torch.from_numpy(d["x"]).to(dev).requires_grad_(True)
    logits = m(x)
    assert np.abs(logits.detach().cpu().numpy() - d[name + ":logits"]).max() < 1e-5
    loss = torch.nn.functional.cross_entropy(logits, torch.from_numpy(d["y"]).to(dev))
    loss.backward()
    assert abs(loss.item() - float(d[name + ":loss"])) < 1e-5
    assert np.abs(x.grad.cpu().numpy() - d[name + ":grad_x"]).max() < 1e-5
    for k, p in m.named_parameters():
        ref = d[f"{name}:g:{k}"]
        assert np.abs(p.grad.cpu().numpy() - ref).max() < 2e-5 * max(1.0, np.abs(ref).max()), k


@pytest.mark.parametrize("att,ln,mixed", [(False, True, False), (True, False, False), (False, False, False),
                                          (False, False, True), (False, True, True)])
def test_ablation_variants_h128_vs_oracle(dev, att, ln, mixed):
    """Vectorised-width kernels (H = 128, W = 256) with the identity-LayerNorm / mean-pool paths."""
    from lstm_ode_bci_amd import AblationLSTMModel
    from oracle import torch_cpu_path as TP
    C, H, L, T, B = 61, 128, 2, 48, 40
    sd = syn.make_state_dict(C, H, L, 2, True, seed=9)
    x, y = syn.make_windows(B, T, C, seed=4)
    sd_f = {k: v for k, v in sd.items()
            if (ln or not (k.startswith("input_proj.1") or k.startswith("layer_norm")))
            and (att or not k.startswith("attention"))}
    ref = TP.build(sd_f, C, H, L, 2, True, use_attention=att, use_layer_norm=ln)
    loss_r, gp_r, gx_r = TP.loss_and_grads(ref, torch.from_numpy(x), torch.from_numpy(y))
    with torch.no_grad():
        lr_ = ref(torch.from_numpy(x)).numpy()
    m = _load(AblationLSTMModel(C, H, L, 2, 0.4, True, att, ln), sd_f, dev).eval()
    if mixed:
        m.gate_gemm_dtype = "bf16"
    xg = torch.from_numpy(x).to(dev).requires_grad_(True)
    logits = m(xg)
    tol_l, tol_g = (5e-3, 2e-2) if mixed else (1e-5, 2e-4)
    assert np.abs(logits.detach().cpu().numpy() - lr_).max() < tol_l
    torch.nn.functional.cross_entropy(logits, torch.from_numpy(y).to(dev)).backward()
    assert np.abs(xg.grad.cpu().numpy() - gx_r).max() < tol_g * max(np.abs(gx_r).max(), 1e-6)
    for k, p in m.named_parameters():
        r = gp_r[k]
        if np.abs(r).max() < 1e-7:
            continue
        assert np.abs(p.grad.cpu().numpy() - r).max() < tol_g * np.abs(r).max(), k


# ------------------------------------------------------------------------------------------
# FusedAdamW + the harness (04_lstm_model.py:406-596)
# ------------------------------------------------------------------------------------------
def test_fused_adamw_flat_views_and_torch_equivalence(dev):
    from lstm_ode_bci_amd import EnhancedLSTMModel
    from lstm_ode_bci_amd.training import FusedAdamW, WeightedCrossEntropy
    C, H, T, B = 5, 8, 12, 6
    sd = syn.make_state_dict(C, H, 2, 2, True, seed=3, affine_jitter=0.1)
    x, y = syn.make_windows(B, T, C, seed=8)
    xt, yt = torch.from_numpy(x).to(dev), torch.from_numpy(y).to(dev)
    ma = _load(EnhancedLSTMModel(C, H, 2, 2, 0.0, True), sd, dev).train()
    mb = _load(EnhancedLSTMModel(C, H, 2, 2, 0.0, True), sd, dev).train()
    oa = FusedAdamW(ma.parameters(), lr=2e-3, weight_decay=0.02)
    ob = torch.optim.AdamW(mb.parameters(), lr=2e-3, weight_decay=0.02)
    crit = WeightedCrossEntropy(torch.tensor([0.7, 1.3])).to(dev)
    for p in ma.parameters():         # parameters and gradients are views of the flat buffers
        assert oa.flat_param.data_ptr() <= p.data_ptr() < oa.flat_param.data_ptr() + 4 * oa.flat_param.numel()
        assert p.data_ptr() % 256 == 0
    for it in range(4):
        oa.zero_grad()
        ob.zero_grad()
        la = crit(ma(xt), yt)
        la.backward()
        lb = torch.nn.functional.cross_entropy(mb(xt), yt, weight=crit.weight)
        lb.backward()
        assert abs(la.item() - lb.item()) < 1e-5
        if it == 2:                   # the non-fused clip (in place) must agree with torch's as well
            na = oa.clip_grad_norm_(1e-3)
            nb = torch.nn.utils.clip_grad_norm_(mb.parameters(), 1e-3)
            assert abs(na.item() - nb.item()) < 1e-5 * nb.item()
            oa.step()
        else:
            torch.nn.utils.clip_grad_norm_(mb.parameters(), 0.05)
            oa.step(clip_grad_norm=0.05)
        ob.step()
        for (k, pa), pb in zip(ma.named_parameters(), mb.parameters()):
            if k == "attention.attention.2.bias":      # exact-zero gradient here, rounding noise in torch's
                continue
            assert (pa - pb).abs().max().item() < 2e-5, (it, k)
    # model.zero_grad() detaches the views; the next step() must still see the gradients
    ma.zero_grad(set_to_none=True)
    mb.zero_grad(set_to_none=True)
    crit(ma(xt), yt).backward()
    torch.nn.functional.cross_entropy(mb(xt), yt, weight=crit.weight).backward()
    oa.step()
    ob.step()
    for (k, pa), pb in zip(ma.named_parameters(), mb.parameters()):
        if k != "attention.attention.2.bias":
            assert (pa - pb).abs().max().item() < 3e-5, k


def test_train_model_matches_reference_train_model(dev):
    from lstm_ode_bci_amd import EnhancedLSTMModel
    from lstm_ode_bci_amd.training import DeviceWindowLoader, train_model
    d = np.load(os.path.join(GOLDEN, "g6_training.npz"))
    kw = dict(zip([str(k) for k in d["kw_names"]], d["kw_vals"]))
    sd0 = {k[3:]: d[k] for k in d.files if k.startswith("w0:")}
    m = _load(EnhancedLSTMModel(5, 8, 3, 2, 0.0, True), sd0, dev)
    tl = DeviceWindowLoader(d["x_train"], d["y_train"], 4, "sequential", dev)
    vl = DeviceWindowLoader(d["x_val"], d["y_val"], 5, "sequential", dev)
    m, hist = train_model(m, tl, vl, d["y_train"], epochs=int(kw["epochs"]), learning_rate=kw["learning_rate"],
                          patience=int(kw["patience"]), weight_decay=kw["weight_decay"],
                          warmup_epochs=int(kw["warmup_epochs"]),
                          gradient_accumulation_steps=int(kw["gradient_accumulation_steps"]), use_amp=False,
                          verbose=False)
    for k in ("train_loss", "val_loss", "learning_rates"):
        assert np.allclose(hist[k], d["hist:" + k], rtol=0, atol=2e-5), (k, hist[k], d["hist:" + k])
    for k in ("train_acc", "val_acc", "val_f1"):
        assert np.allclose(hist[k], d["hist:" + k], rtol=0, atol=1e-12), k
    moved = 0.0
    for k, v in m.state_dict().items():
        ref = d["w1:" + k]
        if k == "attention.attention.2.bias":
            continue          # analytically zero gradient; the reference's update is Adam-amplified rounding noise
        assert np.abs(v.cpu().numpy() - ref).max() < 2e-4, k
        moved = max(moved, np.abs(ref - sd0[k]).max())
    assert moved > 1e-2


def test_train_model_mixed_h128_learns(dev):
    """The bf16-autocast harness on the real layer sizes: a separable synthetic task must be learnt."""
    from lstm_ode_bci_amd import EnhancedLSTMModel
    from lstm_ode_bci_amd.training import create_dataloaders, train_model
    torch.manual_seed(0)
    rng = np.random.default_rng(0)
    n, T, C = 384, 64, 61
    y = rng.integers(0, 2, n)
    X = rng.standard_normal((n, T, C)).astype(np.float64)          # float64 on disk, as processed_sequences.npz
    X[:, :, :8] += (2.0 * y[:, None, None] - 1.0) * 0.8
    m = EnhancedLSTMModel(C, 128, 3, 2, 0.4, True).to(dev)
    tl, vl, _ = create_dataloaders(X[:256], y[:256], X[256:], y[256:], X[256:], y[256:], batch_size=64,
                                   val_batch_size=128, device=dev)
    m, hist = train_model(m, tl, vl, y[:256], epochs=6, learning_rate=2e-3, warmup_epochs=1,
                          gradient_accumulation_steps=2, verbose=False)
    assert hist["val_acc"][-1] > 0.9 and hist["train_loss"][-1] < 0.6 * hist["train_loss"][0], hist


# ------------------------------------------------------------------------------------------
# gradient attribution (07_explainability.py:203-285)
# ------------------------------------------------------------------------------------------
def test_channel_importance_matches_reference(dev):
    from lstm_ode_bci_amd import EnhancedLSTMModel
    from lstm_ode_bci_amd.attribution import compute_channel_importance, input_gradients
    d = np.load(os.path.join(GOLDEN, "g7_channel_importance.npz"))
    sd = {k[2:]: d[k] for k in d.files if k.startswith("w:")}
    m = _load(EnhancedLSTMModel(5, 8, 3, 2, 0.0, True), sd, dev).eval()
    np.random.seed(0)
    df = compute_channel_importance(m, d["x"], n_samples=len(d["x"]), batch_size=3)
    assert not m.training                                    # mode restored (07:281-282)
    assert list(df["Importance"]) == sorted(df["Importance"], reverse=True)
    imp = df.sort_index()["Importance"].to_numpy()
    assert np.abs(imp - d["importance"]).max() < 2e-6
    assert list(df.sort_index()["Channel"]) == [str(c) for c in d["channels"]]
    # one vector-Jacobian launch == B per-window backward passes with retain_graph (07:248-258)
    xb = torch.from_numpy(d["x"][:4]).to(dev)
    g_all, pred = input_gradients(m, xb)
    xr = xb.clone().requires_grad_(True)
    out = m(xr)
    for i in range(4):
        if xr.grad is not None:
            xr.grad.zero_()
        out[i, pred[i]].backward(retain_graph=True)
        assert (xr.grad[i] - g_all[i]).abs().max().item() < 1e-7
        assert xr.grad[[j for j in range(4) if j != i]].abs().max().item() == 0.0


# ------------------------------------------------------------------------------------------
# 16-row bf16 recurrent kernels (two workgroups per CU, LDS-DMA operand ring in BPTT): edge shapes
# ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("T,B,bi,L", [(1, 16, False, 1), (2, 33, True, 2), (3, 40, False, 2), (5, 7, True, 3)])
def test_mixed_recurrent_kernels_short_sequences(dev, T, B, bi, L):
    """The operand ring of the BPTT kernel runs two steps ahead and its tail re-fetches the last step: sequences
    shorter than the ring, odd lengths, one direction, ragged batches."""
    from lstm_ode_bci_amd import EnhancedLSTMModel
    from oracle import torch_cpu_path as TP
    C, H = 61, 128
    sd = syn.make_state_dict(C, H, L, 2, bi, seed=T + 10 * B)
    x, y = syn.make_windows(B, T, C, seed=T)
    ref = TP.build(sd, C, H, L, 2, bi)
    loss_r, gp_r, gx_r = TP.loss_and_grads(ref, torch.from_numpy(x), torch.from_numpy(y))
    with torch.no_grad():
        lr_ = ref(torch.from_numpy(x)).numpy()
    m = _load(EnhancedLSTMModel(C, H, L, 2, 0.4, bi), sd, dev).eval()
    m.gate_gemm_dtype = "bf16"
    xg = torch.from_numpy(x).to(dev).requires_grad_(True)
    logits = m(xg)
    assert np.abs(logits.detach().cpu().numpy() - lr_).max() < 5e-3
    torch.nn.functional.cross_entropy(logits, torch.from_numpy(y).to(dev)).backward()
    assert np.abs(xg.grad.cpu().numpy() - gx_r).max() < 2e-2 * max(np.abs(gx_r).max(), 1e-6)
    for k, p in m.named_parameters():
        r = gp_r[k]
        if np.abs(r).max() < 1e-7:
            continue
        # with T <= 5 the attention weights are nearly uniform and the score-MLP gradients are differences of
        # almost equal terms (max |grad| ~1e-5): the bf16 rounding of v shows up relatively larger there
        tol = 6e-2 if k.startswith("attention") else 2e-2
        assert np.abs(p.grad.cpu().numpy() - r).max() < tol * np.abs(r).max(), k


def test_bias_gradients_are_distinct_tensors(dev):
    """b_ih and b_hh share their gradient VALUE; if they shared the tensor, torch's in-place clip_grad_norm_
    (the reference's loop, 04:501) would scale it twice."""
    from lstm_ode_bci_amd import EnhancedLSTMModel
    sd = syn.make_state_dict(5, 8, 2, 2, True, seed=3)
    x, y = syn.make_windows(4, 6, 5, seed=1)
    m = _load(EnhancedLSTMModel(5, 8, 2, 2, 0.0, True), sd, dev).train()
    for _ in range(3):
        m.zero_grad(set_to_none=True)
        torch.nn.functional.cross_entropy(m(torch.from_numpy(x).to(dev)), torch.from_numpy(y).to(dev)).backward()
        ptrs = [p.grad.data_ptr() for p in m.parameters()]
        assert len(set(ptrs)) == len(ptrs)
        before = {k: p.grad.clone() for k, p in m.named_parameters()}
        torch.nn.utils.clip_grad_norm_(m.parameters(), 1e-3)
        ratio = {k: float((p.grad.norm() / before[k].norm()).item()) for k, p in m.named_parameters()
                 if float(before[k].norm()) > 0}
        assert max(ratio.values()) - min(ratio.values()) < 1e-5 * max(ratio.values()), ratio


def test_input_grad_only_skips_parameter_gradients_and_matches(dev):
    from lstm_ode_bci_amd import EnhancedLSTMModel
    from lstm_ode_bci_amd.attribution import input_gradients
    from lstm_ode_bci_amd.autograd import input_grad_only
    sd = syn.make_state_dict(61, 128, 3, 2, True)
    x, _ = syn.make_windows(40, 24, 61, seed=2)
    m = _load(EnhancedLSTMModel(61, 128, 3, 2, 0.4, True), sd, dev).eval()
    xb = torch.from_numpy(x).to(dev)
    g_fast, pred = input_gradients(m, xb)
    assert all(p.grad is None for p in m.parameters())
    xr = xb.clone().requires_grad_(True)
    out = m(xr)
    out.gather(1, pred.reshape(-1, 1)).sum().backward()        # the full backward (weight gradients included)
    assert all(p.grad is not None for p in m.parameters())
    assert (xr.grad - g_fast).abs().max().item() <= 1e-6 * xr.grad.abs().max().item()
    with input_grad_only():
        pass
    m.zero_grad(set_to_none=True)
    m(xb.clone().requires_grad_(True)).sum().backward()         # the flag is restored after the context
    assert m.classifier[6].weight.grad is not None


# ------------------------------------------------------------------------------------------
# H = 256 (the reference's real checkpoint size) on the mixed path: W_hh streamed from L2, bf16 MFMA
# ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("T,B,bi,L", [(24, 40, True, 2), (3, 33, False, 1), (1, 7, True, 3)])
def test_mixed_h256_forward_backward_vs_oracle(dev, T, B, bi, L):
    from lstm_ode_bci_amd import EnhancedLSTMModel, ops
    from oracle import torch_cpu_path as TP
    C, H = 61, 256
    assert ops.bf16_rec(H) and ops.can_fuse_dropout(H, True)
    sd = syn.make_state_dict(C, H, L, 2, bi, seed=T + B)
    x, y = syn.make_windows(B, T, C, seed=T)
    ref = TP.build(sd, C, H, L, 2, bi)
    loss_r, gp_r, gx_r = TP.loss_and_grads(ref, torch.from_numpy(x), torch.from_numpy(y))
    with torch.no_grad():
        lr_ = ref(torch.from_numpy(x)).numpy()
    m = _load(EnhancedLSTMModel(C, H, L, 2, 0.4, bi), sd, dev).eval()
    xg = torch.from_numpy(x).to(dev).requires_grad_(True)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        logits = m(xg)
    assert np.abs(logits.detach().cpu().numpy() - lr_).max() < 5e-3
    torch.nn.functional.cross_entropy(logits, torch.from_numpy(y).to(dev)).backward()
    assert np.abs(xg.grad.cpu().numpy() - gx_r).max() < 2e-2 * max(np.abs(gx_r).max(), 1e-6)
    for k, p in m.named_parameters():
        r = gp_r[k]
        if np.abs(r).max() < 1e-7:
            continue
        tol = 6e-2 if (k.startswith("attention") and T <= 5) else 2e-2
        assert np.abs(p.grad.cpu().numpy() - r).max() < tol * np.abs(r).max(), k


def test_mixed_h256_train_mode_is_reproducible_and_finite(dev):
    """Dropout fused into the H = 256 recurrent kernel / dX epilogue: same torch seed -> same loss and gradients."""
    from lstm_ode_bci_amd import EnhancedLSTMModel
    sd = syn.make_state_dict(61, 256, 3, 2, True)
    x, y = syn.make_windows(33, 16, 61, seed=5)
    m = _load(EnhancedLSTMModel(61, 256, 3, 2, 0.4, True), sd, dev).train()
    outs = []
    for _ in range(2):
        torch.manual_seed(11)
        m.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            loss = torch.nn.functional.cross_entropy(m(torch.from_numpy(x).to(dev)), torch.from_numpy(y).to(dev))
        loss.backward()
        outs.append((loss.item(), m.lstm.weight_hh_l1.grad.clone(), m.input_proj[0].weight.grad.clone()))
    assert outs[0][0] == outs[1][0] and np.isfinite(outs[0][0])
    assert (outs[0][1] - outs[1][1]).abs().max().item() <= 1e-6 * outs[0][1].abs().max().item()
    assert (outs[0][2] - outs[1][2]).abs().max().item() <= 1e-5 * outs[0][2].abs().max().item()


@pytest.mark.parametrize("T,Bp,nx,D", [(5, 32, 256, 2), (3, 64, 128, 2), (40, 96, 256, 2), (2, 32, 128, 1), (7, 4096, 256, 2),
                                         (2, 32, 256, 1), (3, 8192, 128, 2), (256, 32, 256, 2)])
def test_fused_layer_weight_gradients(dev, T, Bp, nx, D):
    """dW_ih and dW_hh of a layer from one pass over dP == the exact products of the same bf16 operands (fp64 here),
    including the time shift of h_prev (t-1 forward, t+1 reverse) and the step without a predecessor."""
    from lstm_ode_bci_amd import ops
    H = 128
    rng = np.random.default_rng(T * Bp + nx)
    bf = torch.bfloat16
    dP = torch.from_numpy(rng.standard_normal((T * Bp, D * 4 * H), dtype=np.float32)).to(dev).to(bf)
    Xw = torch.from_numpy(rng.standard_normal((T * Bp, nx + 64), dtype=np.float32)).to(dev).to(bf)
    X = Xw[:, :nx]                                          # a column slice of a wider tensor: ld != nx
    Y = torch.from_numpy(rng.standard_normal((T * Bp, D * H), dtype=np.float32)).to(dev).to(bf)
    assert ops.can_fuse_dw(dP, X, Y, T, Bp, H, D)
    dwih, dwhh = ops.lstm_dw(dP, X, Y, T, Bp, H, D)
    p64, x64, y64 = dP.double(), X.double(), Y.double()
    ref_ih = p64.T @ x64
    tol = 2e-5 * (T * Bp) ** 0.5 * 4
    assert (dwih.double() - ref_ih).abs().max().item() < tol
    for d in range(D):
        a = p64[:, d * 4 * H:(d + 1) * 4 * H]
        y = y64[:, d * H:(d + 1) * H]
        ref = a[Bp:].T @ y[:(T - 1) * Bp] if d == 0 else a[:(T - 1) * Bp].T @ y[Bp:]
        assert (dwhh[d].double() - ref).abs().max().item() < tol, d
    # shapes the kernel refuses are reported, not computed wrongly
    assert not ops.can_fuse_dw(dP, X, Y, T, Bp, 64, D)
    assert not ops.can_fuse_dw(dP.float(), X, Y, T, Bp, H, D)


def test_fused_weight_gradients_same_step_as_separate_gemms(dev, monkeypatch):
    """The whole mixed backward with the fused dW kernel == with the separate TN GEMMs (fp32 accumulation order differs)."""
    from lstm_ode_bci_amd import ops, EnhancedLSTMModel
    torch.manual_seed(0)
    m = EnhancedLSTMModel(input_size=61, hidden_size=128, num_layers=3, dropout=0.0).to(dev)
    x = torch.randn(48, 24, 61, device=dev)
    y = torch.randint(0, 2, (48,), device=dev)

    def grads():
        m.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            loss = torch.nn.functional.cross_entropy(m(x).float(), y)
        loss.backward()
        return {k: p.grad.clone() for k, p in m.named_parameters()}
    from lstm_ode_bci_amd import _lib
    with _lib.variant(FUSED_DW=1):
        g1 = grads()
    with _lib.variant(FUSED_DW=0):
        g0 = grads()
    for k in g0:
        sc = g0[k].abs().max().item() + 1e-12
        assert (g1[k] - g0[k]).abs().max().item() <= 1e-4 * sc + 1e-7, k


@pytest.mark.parametrize("mixed", [False, True])
@pytest.mark.parametrize("kw", [dict(), dict(bidirectional=False, num_layers=2), dict(use_attention=False), dict(use_layer_norm=False)])
def test_gradient_sink_equals_autograd_accumulation(dev, mixed, kw):
    """FusedAdamW(model=...) as the model's gradient sink: the backward accumulates every parameter gradient straight
    into flat_grad (autograd gets None).  Same gradients as the plain autograd path (fp32 atomics: summation order
    only), over two accumulated micro-batches, after model.zero_grad(set_to_none=True), for the ablation variants and a
    unidirectional model; and one fused step equals the un-sunk optimizer's."""
    import copy
    from lstm_ode_bci_amd import AblationLSTMModel
    from lstm_ode_bci_amd.training import FusedAdamW, WeightedCrossEntropy
    torch.manual_seed(5)
    args = dict(input_size=61, hidden_size=128, num_layers=3, num_classes=2, dropout=0.0, bidirectional=True,
                use_attention=True, use_layer_norm=True)
    args.update(kw)
    m1 = AblationLSTMModel(**args).to(dev).train()
    m2 = copy.deepcopy(m1)
    xs = [torch.randn(40, 24, 61, device=dev) for _ in range(2)]
    ys = [torch.randint(0, 2, (40,), device=dev) for _ in range(2)]
    crit = WeightedCrossEntropy(torch.tensor([0.8, 1.2])).to(dev)
    o1 = FusedAdamW(m1.parameters(), lr=3e-4, weight_decay=1e-4, model=m1)          # sink
    o2 = FusedAdamW(m2.parameters(), lr=3e-4, weight_decay=1e-4)                    # plain autograd accumulation
    from lstm_ode_bci_amd.training import grad_sink_of
    assert grad_sink_of(m1) is o1 and grad_sink_of(m2) is None and not hasattr(m1, "_lob_grad_sink")

    def run(m, o, drop_grads=False):
        o.zero_grad()
        if drop_grads:
            m.zero_grad(set_to_none=True)
        for x, y in zip(xs, ys):
            with torch.autocast("cuda", dtype=torch.bfloat16, enabled=mixed):
                loss = crit(m(x), y) / 2
            loss.backward()
        return {k: p.grad.clone() for k, p in m.named_parameters()}
    for drop in (False, True):
        g1, g2 = run(m1, o1, drop), run(m2, o2, drop)
        for k in g2:
            sc = g2[k].abs().max().item() + 1e-12
            assert (g1[k] - g2[k]).abs().max().item() <= 2e-4 * sc + 1e-7, (k, drop)
    # every .grad is still a view of the flat buffer, and the fused step agrees
    for p, off in zip(o1.param_groups[0]["params"], o1._offsets):
        assert p.grad.data_ptr() == o1.flat_grad.data_ptr() + 4 * off
    o1.step(clip_grad_norm=1.0)
    o2.step(clip_grad_norm=1.0)
    for (k, p1), (_, p2) in zip(m1.named_parameters(), m2.named_parameters()):
        assert (p1 - p2).abs().max().item() <= 2e-6, k
    # detaching restores autograd's own accumulation
    o1.detach_model()
    g1 = run(m1, o1)
    assert grad_sink_of(m1) is None and all(torch.isfinite(v).all() for v in g1.values())


def test_weight_gradients_on_the_side_stream_equal_the_single_stream_step(dev):
    """ops.OVERLAP_DW (round 4): at small batches the fused dW GEMM of LSTM layer l runs on a second stream next to the BPTT of
    layer l - 1.  The flat gradient buffer (sink; dropout on; two accumulated micro-batches, three times over) with and
    without it: equal up to the order of the split-k atomics, i.e. the side stream's gradients are complete when the backward
    returns and no buffer was recycled under the GEMM; then one clipped AdamW step stays finite."""
    from lstm_ode_bci_amd import EnhancedLSTMModel, ops
    from lstm_ode_bci_amd.training import FusedAdamW, WeightedCrossEntropy
    torch.manual_seed(11)
    m = EnhancedLSTMModel(61, 128, 3, 2, 0.4, True).to(dev).train()
    xs = [torch.randn(96, 40, 61, device=dev) for _ in range(2)]
    ys = [torch.randint(0, 2, (96,), device=dev) for _ in range(2)]
    crit = WeightedCrossEntropy(torch.tensor([0.8, 1.2])).to(dev)
    o = FusedAdamW(m.parameters(), lr=3e-3, weight_decay=1e-4, model=m)
    assert ops.OVERLAP_DW and ops.rec_underfilled(128, 96, 2)

    def grads(overlap):
        ops.OVERLAP_DW = overlap
        try:
            torch.manual_seed(3)
            o.zero_grad()
            for x, y in zip(xs, ys):
                with torch.autocast("cuda", dtype=torch.bfloat16):
                    loss = crit(m(x), y) / 2
                loss.backward()
            return o.flat_grad.clone()           # same stream as the backward's join: complete here
        finally:
            ops.OVERLAP_DW = True
    for _ in range(3):
        ga, gb = grads(True), grads(False)
        mx = gb.abs().max().item()
        assert mx > 0 and torch.isfinite(ga).all()
        assert (ga - gb).abs().max().item() <= 1e-5 * mx, ((ga - gb).abs().max().item(), mx)
    grads(True)
    o.step(clip_grad_norm=1.0)
    assert all(torch.isfinite(p).all() for p in m.parameters())
