"""The fused input projection of the mixed path (lob_input_proj_ln_bf16: Linear -> LayerNorm -> GELU -> Dropout of
04_lstm_model.py:173-178 in one launch) against the three-kernel sequence it replaces (pad + cast, K = 64 GEMM,
LayerNorm): same matrix instruction, same k order, same lane assignment in the LayerNorm, same dropout hash -- the
activations, the saved pre-activations and the saved bf16 windows must be BIT-IDENTICAL, on ragged shapes too (B * T not a
multiple of the 32-row tile, padded batch rows, C = 14 / 61 / 64-8), and so must the model's outputs and gradients."""
import pytest
import torch

from lstm_ode_bci_amd import synthetic as syn

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return torch.device("cuda:0")


def _unfused(ops, x2d, w, b, g, be, B, T, Bp, act, p, seed):
    C = x2d.shape[1]
    Cp = (C + 7) // 8 * 8
    xb = ops.pad_cast_bf16(x2d, Cp)
    wpad = torch.zeros((w.shape[0], Cp), device=x2d.device)
    wpad[:, :C] = w
    pre = ops.gemm_nt(xb, wpad, b, mixed=True)
    a = ops.layernorm_act(pre, g, be, act=act, remap=(T, B, Bp), drop_p=p, seed=seed, out_bf16=True)
    return a, pre, xb


@pytest.mark.parametrize("B,T,C", [(1, 1, 61), (3, 7, 61), (5, 256, 61), (40, 33, 14), (64, 64, 61), (9, 100, 56 - 3),
                                   (130, 31, 61)])
@pytest.mark.parametrize("p", [0.0, 0.3])
@pytest.mark.parametrize("identity", [False, True])
@pytest.mark.parametrize("H", [128, 256])
def test_fused_input_projection_is_bit_identical_to_the_sequence(dev, B, T, C, p, identity, H):
    from lstm_ode_bci_amd import ops
    g = torch.Generator(device=dev).manual_seed(B * 1000 + T + H)
    x2d = torch.randn((B * T, C), generator=g, device=dev) * 3.0
    w = torch.randn((H, C), generator=g, device=dev) * 0.2
    b = torch.randn((H,), generator=g, device=dev) * 0.1
    gam = None if identity else torch.rand((H,), generator=g, device=dev) + 0.5
    bet = None if identity else torch.randn((H,), generator=g, device=dev) * 0.1
    Bp = ops.ceil32(B)
    assert ops.input_proj_ok(x2d, H, C)
    a, pre, xb = ops.input_proj_ln(x2d, w, b, gam, bet, B, T, Bp, H, act=ops.ACT_GELU, drop_p=p, seed=1234, save=True)
    ar, prer, xbr = _unfused(ops, x2d, w, b, gam, bet, B, T, Bp, ops.ACT_GELU, p, 1234)
    assert torch.equal(xb, xbr)
    assert torch.equal(pre, prer)
    assert torch.equal(a.view(torch.int16), ar.view(torch.int16))
    a2, pre2, xb2 = ops.input_proj_ln(x2d, w, b, gam, bet, B, T, Bp, H, act=ops.ACT_GELU, drop_p=p, seed=1234, save=False)
    assert pre2 is None and xb2 is None and torch.equal(a2.view(torch.int16), a.view(torch.int16))
    if H == 128:        # the column-decomposed kernel (what H = 256 runs) at width 128: same bits as well
        a3, pre3, xb3 = ops.input_proj_ln(x2d, w, b, gam, bet, B, T, Bp, H, act=ops.ACT_GELU, drop_p=p, seed=1234, save=True,
                                          colwave=True)
        assert torch.equal(a3.view(torch.int16), a.view(torch.int16)) and torch.equal(pre3, pre) and torch.equal(xb3, xb)
    # against float64 from the same bf16-rounded operands (not twin against twin)
    pre64 = x2d.to(torch.bfloat16).double() @ w.to(torch.bfloat16).double().t() + b.double()
    assert (pre.double() - pre64).abs().max().item() < 1e-4 * max(1.0, pre64.abs().max().item())


@pytest.mark.parametrize("H", [128, 256])
def test_model_outputs_and_gradients_do_not_change_with_the_fused_head(dev, H):
    from lstm_ode_bci_amd import EnhancedLSTMModel, ops
    sd = {k: torch.from_numpy(v) for k, v in syn.make_state_dict(61, H, 3, 2, True).items()}
    x, y = syn.make_windows(24, 64, 61, seed=4)
    xt = torch.from_numpy(x).to(dev)

    def run(fused, train):
        m = EnhancedLSTMModel(61, H, 3, 2, 0.4, True).to(dev)
        m.load_state_dict(sd)
        m.train(train)
        old = ops.FUSE_INPUT_PROJ
        ops.FUSE_INPUT_PROJ = fused
        try:
            torch.manual_seed(3)
            with torch.autocast("cuda", dtype=torch.bfloat16):
                if not train:
                    with torch.no_grad():
                        return m(xt), None
                out = m(xt)
                out.float().square().sum().backward()
            return out.detach(), [p.grad.clone() for p in m.parameters()]
        finally:
            ops.FUSE_INPUT_PROJ = old

    for train in (False, True):
        o1, g1 = run(True, train)
        o0, g0 = run(False, train)
        assert torch.equal(o1, o0)
        if train:       # the weight-gradient GEMMs sum their split-k partials with fp32 atomics: order-dependent last bits
            for a, b in zip(g1, g0):
                assert (a - b).abs().max().item() <= 2e-5 * max(1e-6, b.abs().max().item())


def test_bf16_dpre_gives_the_same_input_projection_gradients(dev):
    """ops.DPRE_BF16: the input projection's LayerNorm backward stores dpre as bf16 for the weight-gradient GEMM, which
    rounds its fp32 operand to bf16 itself -- same operand bits, so the gradients agree to the last bits the GEMM's fp32
    atomics leave open; the bias gradient comes from the fp32 values either way."""
    from lstm_ode_bci_amd import EnhancedLSTMModel, ops
    sd = {k: torch.from_numpy(v) for k, v in syn.make_state_dict(61, 128, 3, 2, True).items()}
    x, _ = syn.make_windows(40, 96, 61, seed=6)
    xt = torch.from_numpy(x).to(dev)
    grads = {}
    for flag in (True, False):
        m = EnhancedLSTMModel(61, 128, 3, 2, 0.4, True).to(dev)
        m.load_state_dict(sd)
        m.train()
        old = ops.DPRE_BF16
        ops.DPRE_BF16 = flag
        try:
            torch.manual_seed(11)                    # same dropout masks in both runs
            with torch.autocast("cuda", dtype=torch.bfloat16):
                m(xt).float().square().sum().backward()
        finally:
            ops.DPRE_BF16 = old
        grads[flag] = {n: p.grad.clone() for n, p in m.named_parameters()}
    for n in grads[True]:
        a, b = grads[True][n], grads[False][n]
        assert (a - b).abs().max().item() <= 2e-5 * max(1e-6, b.abs().max().item()), n
    assert grads[True]["input_proj.0.weight"].abs().max().item() > 0


def test_fused_head_refuses_what_it_does_not_cover(dev):
    from lstm_ode_bci_amd import _lib, ops
    x2d = torch.zeros((64, 80), device=dev)
    assert not ops.input_proj_ok(x2d, 128, 80) and not ops.input_proj_ok(x2d[:, :61], 128, 61)     # C > 64; strided rows
    assert not ops.input_proj_ok(torch.zeros((64, 61), device=dev), 64, 61)
    L = _lib.lib()
    assert L.lob_input_proj_ln_bf16(None, 61, None, 61, None, None, None, None, None, 64, None, 1, 1, 32, 128, 1e-5, 0, 0.0,
                                    0, None) == -1


@pytest.mark.parametrize("B,T,C", [(1, 1, 61), (3, 7, 61), (5, 256, 61), (40, 33, 14), (130, 31, 61)])
@pytest.mark.parametrize("p", [0.0, 0.3])
@pytest.mark.parametrize("identity", [False, True])
def test_fused_head_backward_matches_layernorm_backward_plus_tn_gemm(dev, B, T, C, p, identity):
    """lob_input_proj_bwd_bf16 (LayerNorm + GELU + dropout backward and dW = dpre^T xb in one launch; dpre stays on chip)
    against lob_layernorm_act_bwd_f32 (bf16 dpre) + lob_gemm_tn_bf16: the same bf16 operands reach the same matrix
    instruction, only the fp32 order in which rows are summed differs (atomics in both) -> fp32-rounding agreement; and
    against float64 from the same operands."""
    from lstm_ode_bci_amd import ops
    H = 128
    g = torch.Generator(device=dev).manual_seed(B * 31 + T)
    Bp = ops.ceil32(B)
    Cp = (C + 7) // 8 * 8
    pre = torch.randn((B * T, H), generator=g, device=dev) * 1.5
    dA = (torch.randn((T * Bp, H), generator=g, device=dev) * 0.02).to(torch.bfloat16)
    xb = torch.zeros((B * T, Cp), device=dev, dtype=torch.bfloat16)
    xb[:, :C] = (torch.randn((B * T, C), generator=g, device=dev) * 2.0).to(torch.bfloat16)
    gam = None if identity else torch.rand((H,), generator=g, device=dev) + 0.5
    bet = None if identity else torch.randn((H,), generator=g, device=dev) * 0.1
    old = ops.FUSE_INPUT_PROJ_BWD
    ops.FUSE_INPUT_PROJ_BWD = True
    try:
        assert ops.input_proj_bwd_ok(pre, dA, xb, H)
    finally:
        ops.FUSE_INPUT_PROJ_BWD = old
    dW = torch.zeros((H, Cp), device=dev)
    dbias = torch.zeros((H,), device=dev)
    _, dg, db = ops.input_proj_bwd(pre, gam, bet, dA, xb, dW, B, T, Bp, H, act=ops.ACT_GELU, drop_p=p, seed=99, dbias=dbias)
    dbr = torch.zeros((H,), device=dev)
    dpre, dgr, dber = ops.layernorm_act_bwd(pre, gam, bet, dA, act=ops.ACT_GELU, remap=(T, B, Bp), drop_p=p, seed=99,
                                            dx_colsum=dbr, dx_bf16=True)
    dWr = torch.zeros((H, Cp), device=dev)
    ops.gemm_tn(dpre, xb, dWr, mixed=True)

    def close(a, b, name, rel=2e-5):
        assert (a - b).abs().max().item() <= rel * max(1e-6, b.abs().max().item()), (name, (a - b).abs().max().item())
    close(dW, dWr, "dW")
    close(dbias, dbr, "dbias")
    if not identity:
        close(dg, dgr, "dgamma"); close(db, dber, "dbeta")
    assert torch.all(dW[:, C:] == 0)
    ref = dpre.double().t() @ xb.double()
    assert (dW.double() - ref).abs().max().item() <= 1e-4 * max(1e-6, ref.abs().max().item())


def test_model_gradients_do_not_change_with_the_fused_head_backward(dev):
    from lstm_ode_bci_amd import EnhancedLSTMModel, ops
    sd = {k: torch.from_numpy(v) for k, v in syn.make_state_dict(61, 128, 3, 2, True).items()}
    x, _ = syn.make_windows(40, 96, 61, seed=6)
    xt = torch.from_numpy(x).to(dev)
    grads = {}
    for flag in (True, False):
        m = EnhancedLSTMModel(61, 128, 3, 2, 0.4, True).to(dev)
        m.load_state_dict(sd)
        m.train()
        old = ops.FUSE_INPUT_PROJ_BWD
        ops.FUSE_INPUT_PROJ_BWD = flag
        try:
            torch.manual_seed(11)
            with torch.autocast("cuda", dtype=torch.bfloat16):
                m(xt).float().square().sum().backward()
        finally:
            ops.FUSE_INPUT_PROJ_BWD = old
        grads[flag] = {n: p.grad.clone() for n, p in m.named_parameters()}
    for n in grads[True]:
        a, b = grads[True][n], grads[False][n]
        assert (a - b).abs().max().item() <= 2e-5 * max(1e-6, b.abs().max().item()), n
    # an input gradient still works (falls back to the unfused pair, which materialises dpre)
    m = EnhancedLSTMModel(61, 128, 3, 2, 0.4, True).to(dev)
    m.load_state_dict(sd)
    m.train()
    xg = xt.clone().requires_grad_(True)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        m(xg).float().square().sum().backward()
    assert xg.grad is not None and torch.isfinite(xg.grad).all() and xg.grad.abs().max().item() > 0


@pytest.mark.parametrize("B,T", [(1, 1), (3, 5), (40, 7), (8, 256), (70, 33), (128, 64)])
@pytest.mark.parametrize("save", [False, True])
@pytest.mark.parametrize("drop", [0.0, 0.4])
def test_fp32_fused_head_vs_gemm_plus_layernorm(dev, B, T, save, drop):
    """lob_input_proj_ln_f32 (round 4): the fp32 path's input projection (Linear + LayerNorm + GELU + dropout, 04:173-178) in one
    launch, against lob_gemm_nt_f32 + lob_layernorm_act_f32(remap, dropout) under the same seed: the pre-activations (exact fp32
    products both ways) and the activations to fp32 summation order of the 61-term dot products, padding rows untouched, the
    dropout mask in the same places."""
    from lstm_ode_bci_amd import ops
    H, C = 128, 61
    Bp = ops.ceil32(B)
    g = torch.Generator(device=dev).manual_seed(B * 1000 + T)
    x2d = torch.randn((B * T, C), generator=g, device=dev) * 3.0
    w = torch.randn((H, C), generator=g, device=dev) * 0.2
    b = torch.randn((H,), generator=g, device=dev) * 0.1
    gam = torch.rand((H,), generator=g, device=dev) + 0.5
    bet = torch.randn((H,), generator=g, device=dev) * 0.1
    assert ops.input_proj_f32_ok(x2d, H, C, w)
    a, pre = ops.input_proj_ln_f32(x2d, w, b, gam, bet, B, T, Bp, H, act=ops.ACT_GELU, drop_p=drop, seed=99, save=save)
    prer = ops.gemm_nt(x2d, w, b)
    ar = ops.layernorm_act(prer, gam, bet, act=ops.ACT_GELU, remap=(T, B, Bp), drop_p=drop, seed=99)
    assert a.shape == ar.shape and a.dtype == torch.float32
    assert (pre is not None) == save
    if save:
        assert (pre - prer).abs().max().item() <= 2e-6 * max(1.0, prer.abs().max().item())
    assert torch.equal(a == 0, ar == 0) or drop == 0.0              # the same elements dropped
    assert (a - ar).abs().max().item() <= 1e-5 * max(1.0, ar.abs().max().item())
    ref = torch.nn.functional.gelu(torch.nn.functional.layer_norm(x2d.double() @ w.double().t() + b.double(), (H,),
                                                                  gam.double(), bet.double(), 1e-5))
    if drop == 0.0:
        got = a.reshape(T, Bp, H)[:, :B].permute(1, 0, 2).reshape(B * T, H)
        assert (got.double() - ref).abs().max().item() < 5e-6 * max(1.0, ref.abs().max().item())
    pad = a.reshape(T, Bp, H)[:, B:]
    assert pad.numel() == 0 or float(pad.abs().max()) == 0.0
