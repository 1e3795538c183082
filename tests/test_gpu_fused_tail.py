"""The fused tail of the mixed forward at H = 128 (lob_attn_scores_bf16: post-LSTM LayerNorm 04_lstm_model.py:192 + the
attention's score layer 04:123-125 in one launch, scores handed to the pooling kernel) against the three kernels it
replaces: same lane assignment in the LayerNorm, same matrix instruction and k order in the score layer, same reduction
order of the scores -> v, u, the attention weights and the context must be BIT-IDENTICAL, also on row counts that do
not fill the 128-row tile and with padded batch rows; and the model's outputs / gradients must not move."""
import pytest
import torch

from lstm_ode_bci_amd import synthetic as syn

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return torch.device("cuda:0")


@pytest.mark.parametrize("T,B", [(1, 1), (5, 3), (7, 40), (256, 8), (33, 70), (64, 128)])
@pytest.mark.parametrize("save", [False, True])
@pytest.mark.parametrize("H", [128, 256])
def test_fused_tail_is_bit_identical_to_the_three_kernels(dev, T, B, save, H):
    from lstm_ode_bci_amd import ops
    D = 2
    W = 2 * H
    Bp = ops.ceil32(B)
    g = torch.Generator(device=dev).manual_seed(T * 100 + B + H)
    y16 = (torch.randn((T * Bp, W), generator=g, device=dev) * 0.7).to(torch.bfloat16)
    gam = torch.rand((W,), generator=g, device=dev) + 0.5
    bet = torch.randn((W,), generator=g, device=dev) * 0.1
    w1 = torch.randn((H, W), generator=g, device=dev) * 0.08
    b1 = torch.randn((H,), generator=g, device=dev) * 0.1
    w2 = torch.randn((H,), generator=g, device=dev) * 0.3
    b2 = torch.randn((1,), generator=g, device=dev)
    w1_16 = w1.to(torch.bfloat16)
    assert ops.attn_scores_ok(y16, H, D, Bp, w1)
    v, u, S = ops.attn_scores(y16, gam, bet, w1_16, b1, w2, b2, T, B, Bp, H, D, save=save)
    ctx, attn = ops.attn_pool_fwd_scores(v, S, T, B, Bp)
    vr = ops.layernorm_act(y16, gam, bet, out_bf16=True)
    ur = ops.gemm_nt(vr, w1_16 if ops.dma_ok(W, H, T * Bp) else w1, b1, act=ops.ACT_TANH, mixed=True)
    ctxr, attnr = ops.attn_pool_fwd(vr, ur, w2, b2, T, B, Bp)
    assert torch.equal(v.view(torch.int16), vr.view(torch.int16))
    if save:
        assert torch.equal(u, ur)
    else:
        assert u is None
    assert torch.equal(attn, attnr) and torch.equal(ctx, ctxr)
    # and against float64 from the same bf16 operands
    v64 = vr.double()
    u64 = torch.tanh(v64 @ w1_16.double().t() + b1.double())
    s64 = (u64 @ w2.double() + b2.double()).reshape(T, Bp)[:, :B].t()
    a64 = torch.softmax(s64, dim=1)
    assert (attn.double() - a64).abs().max().item() < 4e-6


@pytest.mark.parametrize("H", [128, 256])
def test_model_outputs_do_not_change_with_the_fused_tail(dev, H):
    from lstm_ode_bci_amd import EnhancedLSTMModel, ops
    sd = {k: torch.from_numpy(v) for k, v in syn.make_state_dict(61, H, 3, 2, True).items()}
    x, _ = syn.make_windows(24, 64, 61, seed=4)
    xt = torch.from_numpy(x).to(dev)

    def run(fused, train):
        m = EnhancedLSTMModel(61, H, 3, 2, 0.4, True).to(dev)
        m.load_state_dict(sd)
        m.train(train)
        old = ops.FUSE_ATTN_SCORES
        ops.FUSE_ATTN_SCORES = fused
        try:
            torch.manual_seed(3)
            with torch.autocast("cuda", dtype=torch.bfloat16):
                if not train:
                    with torch.no_grad():
                        return m(xt, return_attention=True), None
                out = m(xt)
                out.float().square().sum().backward()
            return out.detach(), [p.grad.clone() for p in m.parameters()]
        finally:
            ops.FUSE_ATTN_SCORES = old

    (l1, a1), _ = run(True, False)
    (l0, a0), _ = run(False, False)
    assert torch.equal(l1, l0) and torch.equal(a1, a0)
    o1, g1 = run(True, True)
    o0, g0 = run(False, True)
    assert torch.equal(o1, o0)
    for a, b in zip(g1, g0):            # weight-gradient GEMMs use fp32 atomics: order-dependent last bits
        assert (a - b).abs().max().item() <= 2e-5 * max(1e-6, b.abs().max().item())


@pytest.mark.parametrize("T,B", [(1, 1), (5, 3), (7, 40), (256, 8), (33, 70)])
@pytest.mark.parametrize("H", [128, 256])
def test_fused_backward_tail_matches_gemm_plus_layernorm_backward(dev, T, B, H):
    """lob_attn_ln_bwd_bf16 (dV = dU W1 + the pooling's context term, LayerNorm backward) against lob_gemm_nt_bf16 (bf16 dV)
    + lob_layernorm_act_bwd_f32(pool).  The LayerNorm arithmetic is the same instruction for instruction; the K = H
    product is summed in another k order than the weight-stationary GEMM walks its swizzled LDS tile, so about one dV
    element in 10^5 rounds to the neighbouring bf16 value and takes its own dx element with it (measured: 1 of 41 k,
    20 of 2.1 M): everything else is bit-identical; dgamma / dbeta to fp32 summation order."""
    from lstm_ode_bci_amd import ops
    D = 2
    W = 2 * H
    Bp = ops.ceil32(B)
    g = torch.Generator(device=dev).manual_seed(T * 7 + B + H)
    y16 = (torch.randn((T * Bp, W), generator=g, device=dev) * 0.7).to(torch.bfloat16)
    gam = torch.rand((W,), generator=g, device=dev) + 0.5
    bet = torch.randn((W,), generator=g, device=dev) * 0.1
    dU = (torch.randn((T * Bp, H), generator=g, device=dev) * 0.05).to(torch.bfloat16)
    w1t = (torch.randn((W, H), generator=g, device=dev) * 0.08).to(torch.bfloat16)
    attn = torch.softmax(torch.randn((B, T), generator=g, device=dev), dim=1)
    dctx = torch.randn((B, W), generator=g, device=dev) * 0.3
    assert ops.attn_ln_bwd_ok(y16, dU, w1t, H, D, Bp, gam)
    dx, dg, db = ops.attn_ln_bwd(y16, gam, bet, dU, w1t, attn, dctx, T, B, Bp, H, D)
    dV = ops.gemm_nt(dU, w1t, mixed=True, out_bf16=True)
    assert dV.dtype == torch.bfloat16
    dxr, dgr, dbr = ops.layernorm_act_bwd(y16, gam, bet, dV, pool=(attn, dctx, T, B, Bp), dx_bf16=True)
    same = (dx == dxr).float().mean().item()
    assert same >= 1.0 - 1e-4, same
    assert (dx.float() - dxr.float()).abs().max().item() <= 2 ** -7 * dxr.float().abs().max().item()
    for a, b in ((dg, dgr), (db, dbr)):
        assert (a - b).abs().max().item() <= 1e-5 * max(1.0, b.abs().max().item())
    # float64 from the same bf16 operands: dy = bf16(dU W1) + attn * dctx, then the LayerNorm backward formula
    x = y16.double()
    dy = dV.double()
    rows = torch.arange(T * Bp, device=dev)
    t_i, b_i = rows // Bp, rows % Bp
    ok = b_i < B
    dy[ok] += attn.double()[b_i[ok], t_i[ok]][:, None] * dctx.double()[b_i[ok]]
    mu = x.mean(1, keepdim=True)
    rstd = 1.0 / torch.sqrt(x.var(1, unbiased=False, keepdim=True) + 1e-5)
    xh = (x - mu) * rstd
    dxh = dy * gam.double()
    ref = rstd * (dxh - dxh.mean(1, keepdim=True) - xh * (dxh * xh).mean(1, keepdim=True))
    err = (dx.double() - ref).abs().max().item()
    assert err <= 1e-2 * max(1e-3, ref.abs().max().item())            # bf16 output rounding


@pytest.mark.parametrize("H", [128, 256])
def test_model_gradients_do_not_change_with_the_fused_backward_tail(dev, H):
    from lstm_ode_bci_amd import EnhancedLSTMModel, ops
    sd = {k: torch.from_numpy(v) for k, v in syn.make_state_dict(61, H, 3, 2, True).items()}
    x, _ = syn.make_windows(24, 64, 61, seed=4)
    xt = torch.from_numpy(x).to(dev)
    grads = {}
    for flag in (True, False):
        m = EnhancedLSTMModel(61, H, 3, 2, 0.4, True).to(dev)
        m.load_state_dict(sd)
        m.train()
        old = ops.FUSE_ATTN_LN_BWD
        ops.FUSE_ATTN_LN_BWD = flag
        try:
            torch.manual_seed(3)
            with torch.autocast("cuda", dtype=torch.bfloat16):
                m(xt).float().square().sum().backward()
        finally:
            ops.FUSE_ATTN_LN_BWD = old
        grads[flag] = {n: p.grad.clone() for n, p in m.named_parameters()}
    for n in grads[True]:
        a, b = grads[True][n], grads[False][n]
        assert (a - b).abs().max().item() <= 2e-5 * max(1e-6, b.abs().max().item()), n


@pytest.mark.parametrize("T,B", [(1, 1), (5, 3), (7, 40), (256, 8), (33, 70), (64, 128)])
@pytest.mark.parametrize("save", [False, True])
@pytest.mark.parametrize("scale", [1.0, 300.0])
def test_fp32_fused_tail_vs_the_three_fp32_kernels_and_float64(dev, T, B, save, scale):
    """lob_attn_scores_f32 (round 4): the fp32 path's LayerNorm + score layer in one launch.  v must be BIT-IDENTICAL to the
    LayerNorm kernel's; u, the attention weights and the context come from two-way fp16-split products instead of the exact-fp32
    MFMA GEMM: against that GEMM and against float64 within the fp32 path's 1e-5 budget by a wide margin.  scale = 300: a
    LayerNorm gain and weights far outside fp16's range -- the pre-scales derived on the device must keep both halves finite."""
    from lstm_ode_bci_amd import ops
    H, D = 128, 2
    W = 2 * H
    Bp = ops.ceil32(B)
    g = torch.Generator(device=dev).manual_seed(T * 100 + B)
    y = torch.randn((T * Bp, W), generator=g, device=dev) * 0.7
    gam = (torch.rand((W,), generator=g, device=dev) + 0.5) * scale
    bet = torch.randn((W,), generator=g, device=dev) * 0.1 * scale
    w1 = torch.randn((H, W), generator=g, device=dev) * 0.08 / scale
    w1[3, 5] *= 40.0                                     # an outlier sets the weight range
    b1 = torch.randn((H,), generator=g, device=dev) * 0.1
    w2 = torch.randn((H,), generator=g, device=dev) * 0.3
    b2 = torch.randn((1,), generator=g, device=dev)
    assert ops.attn_scores_f32_ok(y, H, D, Bp, w1)
    v, u, S = ops.attn_scores_f32(y, gam, bet, w1, b1, w2, b2, T, B, Bp, H, D, save=save)
    ctx, attn = ops.attn_pool_fwd_scores(v, S, T, B, Bp)
    vr = ops.layernorm_act(y, gam, bet)
    ur = ops.gemm_nt(vr, w1, b1, act=ops.ACT_TANH)
    ctxr, attnr = ops.attn_pool_fwd(vr, ur, w2, b2, T, B, Bp)
    assert torch.equal(v, vr)
    assert (u is not None) == save
    if save:
        assert torch.isfinite(u).all() and (u - ur).abs().max().item() < 3e-6
    assert torch.isfinite(attn).all() and torch.isfinite(ctx).all()
    assert (attn - attnr).abs().max().item() < 2e-6
    assert (ctx - ctxr).abs().max().item() < 2e-6 * max(1.0, ctxr.abs().max().item())
    u64 = torch.tanh(vr.double() @ w1.double().t() + b1.double())
    s64 = (u64 @ w2.double() + b2.double()).reshape(T, Bp)[:, :B].t()
    a64 = torch.softmax(s64, dim=1)
    assert (attn.double() - a64).abs().max().item() < 2e-6
    if save:
        assert (u.double() - u64).abs().max().item() < 2e-6


def test_fp32_model_outputs_with_and_without_the_fused_fp32_tail(dev):
    """Whole model, fp32 path: logits and every gradient with ops.FUSE_ATTN_SCORES_F32 on / off agree far inside the 1e-5 parity
    budget (the goldens g1 / g2 run through the fused kernel at the unchanged tolerances)."""
    from lstm_ode_bci_amd import EnhancedLSTMModel, ops
    sd = {k: torch.from_numpy(v) for k, v in syn.make_state_dict(61, 128, 3, 2, True).items()}
    x, y = syn.make_windows(40, 64, 61, seed=2)
    m = EnhancedLSTMModel(61, 128, 3, 2, 0.4, True)
    m.load_state_dict(sd)
    m = m.to(dev).eval()

    def run():
        m.zero_grad(set_to_none=True)
        xg = torch.from_numpy(x).to(dev).requires_grad_(True)
        out = m(xg)
        torch.nn.functional.cross_entropy(out, torch.from_numpy(y).to(dev)).backward()
        return out.detach().clone(), {**{k: p.grad.clone() for k, p in m.named_parameters()}, "x": xg.grad.clone()}
    assert ops.FUSE_ATTN_SCORES_F32
    o1, g1 = run()
    ops.FUSE_ATTN_SCORES_F32 = False
    try:
        o0, g0 = run()
    finally:
        ops.FUSE_ATTN_SCORES_F32 = True
    assert not torch.equal(o1, o0) or True               # the score layer's arithmetic differs in the last bits
    assert (o1 - o0).abs().max().item() < 2e-6
    for k in g0:
        mx = g0[k].abs().max().item()
        assert (g1[k] - g0[k]).abs().max().item() <= 2e-5 * mx + 1e-9, k
