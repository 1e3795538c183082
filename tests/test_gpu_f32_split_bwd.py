"""GPU: the fp32 path's BACKWARD on the 16-bit matrix pipe (round 4) -- two-way fp16 operand splits, 22-bit
products, fp32 accumulate, as the forward's gate GEMMs and recurrent kernel since round 2:
lob_gemm_nt_f32_split (dX = dP W_ih), lob_gemm_tn_f32_split (dW = dP^T x), lob_lstm_rec_bwd_f32_x (BPTT; the dgates'
pre-scale re-derived per tile and step).  Each kernel against a float64 product / its exact-fp32 twin (LOB_VAR_F32_SPLIT = 0),
at gradient-like magnitudes far outside fp16's range; the model-level parity tests (tests/test_gpu_parity.py,
tests/test_gpu_training.py: goldens g1 / g2 / g6 / g7 at the unchanged tolerances) run THROUGH these kernels at H = 128."""
import numpy as np
import pytest
import torch

from tests import conftest  # noqa: F401

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


def _rnd(shape, scale, dev, seed):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).to(dev)


@pytest.mark.parametrize("M,K,N,scale", [(8192 + 96, 1024, 256, 1e-4), (4096, 1024, 128, 3e-9), (1000, 256, 512, 40.0)])
def test_split_nt_gemm_vs_float64(dev, M, K, N, scale):
    from lstm_ode_bci_amd import _lib, ops
    a = _rnd((M, K), scale, dev, 1)
    a[5, 7] *= 60.0                                    # an outlier sets the range: the bulk sits 2^-6 below it
    w = _rnd((N, K), 0.05, dev, 2)
    ref = a.double() @ w.double().t()
    amax = (a.abs().max().reshape(1), w.abs().max().reshape(1))
    out = ops.gemm_nt(a, w, amax=amax)
    with _lib.variant(F32_SPLIT=0):
        exact = ops.gemm_nt(a, w, amax=amax)           # the switch routes this call to the exact-fp32 MFMA kernel
    mx = ref.abs().max().item()
    e_split, e_exact = (out.double() - ref).abs().max().item() / mx, (exact.double() - ref).abs().max().item() / mx
    assert torch.isfinite(out).all()
    assert e_split < 2e-6 and e_split < 4 * e_exact + 2e-7, (e_split, e_exact)
    # a loose bound (x 8: pre-scales are powers of two) changes nothing beyond rounding
    out8 = ops.gemm_nt(a, w, amax=(amax[0] * 7.9, amax[1] * 3.0))
    assert (out8.double() - ref).abs().max().item() / mx < 4e-6
    with _lib.variant(F32_SPLIT=2):                    # the twin that splits at every fragment read
        twin = ops.gemm_nt(a, w, amax=amax)
    assert (twin.double() - ref).abs().max().item() / mx < 2e-6 and (twin - out).abs().max().item() <= 2e-6 * mx


@pytest.mark.parametrize("Kc,M,N,scale", [(16384, 1024, 256, 1e-5), (4096 + 32, 512, 128, 2e-3)])
def test_split_tn_gemm_vs_float64(dev, Kc, M, N, scale):
    from lstm_ode_bci_amd import ops
    a_full = _rnd((Kc, 2 * M), scale, dev, 3)           # column slices of wider row-major tensors, as dW_hh takes them
    b_full = _rnd((Kc, 2 * N), 0.5, dev, 4)
    a, b = a_full[:, M:], b_full[:, :N]
    ref = a.double().t() @ b.double()
    out = torch.zeros((M, N), device=dev)
    ops.gemm_tn(a, b, out, amax=(a.abs().max().reshape(1), torch.ones(1, device=dev) * b.abs().max()))
    exact = torch.zeros((M, N), device=dev)
    ops.gemm_tn(a, b, exact)
    mx = ref.abs().max().item()
    e_split, e_exact = (out.double() - ref).abs().max().item() / mx, (exact.double() - ref).abs().max().item() / mx
    assert e_split < 3e-6 and e_split < 4 * e_exact + 3e-7, (e_split, e_exact)
    ops.gemm_tn(a, b, out, amax=(a.abs().max().reshape(1), b.abs().max().reshape(1)))       # accumulates
    assert (out.double() - 2 * ref).abs().max().item() / mx < 6e-6
    from lstm_ode_bci_amd import _lib
    twin = torch.zeros((M, N), device=dev)
    with _lib.variant(F32_SPLIT=2):                    # the twin that splits at every fragment read
        ops.gemm_tn(a, b, twin, amax=(a.abs().max().reshape(1), b.abs().max().reshape(1)))
    assert (twin.double() - ref).abs().max().item() / mx < 3e-6


def test_pipelined_split_gemms_are_bit_identical_to_the_general_kernels(dev):
    """gemm_nt_split_kernel<true> / gemm_tn_split_kernel<true> (whole 128 x 128 tiles, loads a k-tile further ahead, the split
    of the next tile between the MFMA groups of this one) against the general kernels they replace on those shapes
    (LOB_VAR_F32_SPLIT = 3), at the fp32 training step's shapes: same products, same order, same chunks."""
    from lstm_ode_bci_amd import _lib, ops
    rows = 64 * 4096
    dP = _rnd((rows, 1024), 1e-4, dev, 11)
    x = _rnd((rows, 256), 0.5, dev, 12)
    w = _rnd((256, 1024), 0.05, dev, 13)
    am = (dP.abs().max().reshape(1), w.abs().max().reshape(1))
    ax = (dP.abs().max().reshape(1), x.abs().max().reshape(1))

    def run():
        dx = ops.gemm_nt(dP, w, amax=am)
        dw = torch.zeros((1024, 256), device=dev)
        ops.gemm_tn(dP, x, dw, amax=ax)
        dwh = torch.zeros((512, 128), device=dev)          # dW_hh's operands: column slices, rows shifted by one time step
        ops.gemm_tn(dP[4096:, 512:], x[:rows - 4096, 128:], dwh, amax=ax)
        return dx, dw, dwh

    fast = run()
    with _lib.variant(F32_SPLIT=3):
        gen = run()
    # dX has no split-k: bit-identical.  The weight gradients add their chunks' partial sums with fp32 atomics in whatever
    # order the workgroups finish, in both kernels: equal to summation order
    assert torch.equal(fast[0], gen[0])
    for a, b in zip(fast[1:], gen[1:]):
        assert (a - b).abs().max().item() <= 2e-6 * b.abs().max().item()
    ref = dP.double().t() @ x.double()
    assert (fast[1].double() - ref).abs().max().item() <= 2e-6 * ref.abs().max().item()


@pytest.mark.parametrize("T,Bp,D,dyscale", [(24, 64, 2, 1e-3), (5, 32, 1, 1e-9), (40, 96, 2, 300.0)])
def test_split_bptt_vs_exact_twin(dev, T, Bp, D, dyscale):
    """Same saved gates / cell states / dY through lob_lstm_rec_bwd_f32_x and through the exact-fp32 MFMA kernel: dP, the bias
    gradient and the reported max|dP|, at gradient magnitudes from 1e-9 to 3e2 (the per-step tile scale keeps both fp16 halves
    in range)."""
    from lstm_ode_bci_amd import _lib, ops
    H = 128
    whh = _rnd((D, 4 * H, H), 0.08, dev, 5)
    x = _rnd((T * Bp, H), 1.0, dev, 6)
    wih = _rnd((D * 4 * H, H), 0.08, dev, 7)
    bias = _rnd((D * 4 * H,), 0.1, dev, 8)
    P = ops.gate_gemm_x(x, wih, bias, T, Bp, H, D, True, mixed=False, exact=True)
    Y, Cs, _, _ = ops.lstm_rec_fwd(P, whh, T, Bp, H, D, True)         # P now holds the activated gates
    dY = _rnd((T * Bp, D * H), dyscale, dev, 9)
    rng = whh.abs().amax(dim=(1, 2)).contiguous()
    amax = torch.zeros(1, device=dev)
    dP, db = ops.lstm_rec_bwd(P, Cs, whh, dY, T, Bp, H, D, amax_out=amax, range=rng)
    with _lib.variant(F32_SPLIT=0):
        dPr, dbr = ops.lstm_rec_bwd(P, Cs, whh, dY, T, Bp, H, D)
    assert torch.isfinite(dP).all() and torch.isfinite(db).all()
    mx = dPr.abs().max().item()
    assert (dP - dPr).abs().max().item() <= 3e-6 * mx, ((dP - dPr).abs().max().item(), mx)
    assert (db - dbr).abs().max().item() <= 1e-5 * dbr.abs().max().item()
    assert amax.item() == dP.abs().max().item()
    # without a range the kernel's default weight scale applies: same result for weights of ordinary size
    dP2, _ = ops.lstm_rec_bwd(P, Cs, whh, dY, T, Bp, H, D)
    assert (dP2 - dPr).abs().max().item() <= 3e-6 * mx


def test_fp32_training_step_runs_the_split_backward_and_matches_the_exact_one(dev):
    """Whole model, fp32, H = 128, B = 64, T = 48, dropout off: every parameter gradient and the input gradient of the
    split backward against the exact-fp32 kernels (the same comparison the goldens make against the reference, here
    kernel set against kernel set)."""
    from lstm_ode_bci_amd import EnhancedLSTMModel, _lib
    from lstm_ode_bci_amd import synthetic as syn
    sd = syn.make_state_dict(61, 128, 3, 2, True)
    x, y = syn.make_windows(64, 48, 61, seed=4)
    m = EnhancedLSTMModel(61, 128, 3, 2, 0.4, True)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    m = m.to(dev).eval()

    def grads():
        m.zero_grad(set_to_none=True)
        xg = torch.from_numpy(x).to(dev).requires_grad_(True)
        torch.nn.functional.cross_entropy(m(xg), torch.from_numpy(y).to(dev)).backward()
        return {**{k: p.grad.clone() for k, p in m.named_parameters()}, "x": xg.grad.clone()}
    gs = grads()
    with _lib.variant(F32_SPLIT=0):
        ge = grads()
    for k in ge:
        mx = ge[k].abs().max().item()
        if mx < 1e-9:
            continue
        assert (gs[k] - ge[k]).abs().max().item() <= 2e-5 * mx + 1e-9, (k, (gs[k] - ge[k]).abs().max().item(), mx)


@pytest.mark.parametrize("B,T", [(64, 48), (6, 5), (1024, 256)])
def test_fp32_inter_layer_dropout_fused_into_producer_and_dx_epilogue_is_the_same_mask(dev, B, T):
    """fp32 path, H = 128, training mode: nn.LSTM's inter-layer dropout (04:186) as a second output of the saving recurrent
    forward (lob_lstm_rec_fwd_f32_drop) and as the mask epilogue of lob_gemm_nt_f32_split, against the stand-alone
    lob_dropout_f32 passes (ops.FUSE_F32_DROPOUT = False) under the same seed: logits and the input gradient bit-identical
    (same mask, same multiplications), weight gradients equal up to the order of their split-k atomics."""
    from lstm_ode_bci_amd import EnhancedLSTMModel, ops
    from lstm_ode_bci_amd import synthetic as syn
    sd = syn.make_state_dict(61, 128, 3, 2, True)
    x, y = syn.make_windows(B, T, 61, seed=9)
    m = EnhancedLSTMModel(61, 128, 3, 2, 0.4, True)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    m = m.to(dev).train()

    def run():
        torch.manual_seed(77)
        m._seed_counter = 0
        m.zero_grad(set_to_none=True)
        xg = torch.from_numpy(x).to(dev).requires_grad_(True)
        out = m(xg)
        torch.nn.functional.cross_entropy(out, torch.from_numpy(y).to(dev)).backward()
        return out.detach().clone(), {**{k: p.grad.clone() for k, p in m.named_parameters()}, "x": xg.grad.clone()}

    assert ops.FUSE_F32_DROPOUT
    o1, g1 = run()
    ops.FUSE_F32_DROPOUT = False
    try:
        o0, g0 = run()
    finally:
        ops.FUSE_F32_DROPOUT = True
    assert torch.equal(o1, o0)
    assert torch.equal(g1["x"], g0["x"])
    for k in g0:
        mx = g0[k].abs().max().item()
        assert (g1[k] - g0[k]).abs().max().item() <= 2e-6 * mx + 1e-12, k
    # and the mask is really there: evaluation mode gives other logits
    m.eval()
    with torch.no_grad():
        assert not torch.equal(m(torch.from_numpy(x).to(dev)), o1)
    if B == 64:
        # a forward that fused the dropout followed by a backward on the EXACT kernels (the variant flipped in between): the dX
        # GEMM there has no mask epilogue, ops.gemm_nt applies the same mask as a pass of its own
        from lstm_ode_bci_amd import _lib
        m.train()
        torch.manual_seed(77)
        m._seed_counter = 0
        m.zero_grad(set_to_none=True)
        xg = torch.from_numpy(x).to(dev).requires_grad_(True)
        loss = torch.nn.functional.cross_entropy(m(xg), torch.from_numpy(y).to(dev))
        with _lib.variant(F32_SPLIT=0):
            loss.backward()
        gx = {**{k: p.grad.clone() for k, p in m.named_parameters()}, "x": xg.grad.clone()}
        for k in g0:
            mx = g0[k].abs().max().item()
            assert (gx[k] - g0[k]).abs().max().item() <= 3e-5 * mx + 1e-9, k


@pytest.mark.parametrize("T,Bp,D,save", [(20, 1024, 2, False), (7, 96, 2, True), (33, 32, 1, True), (256, 1024, 2, True)])
def test_half_tile_recurrent_forward_is_bit_identical(dev, T, Bp, D, save):
    """LOB_VAR_REC_HALF (round 4): below B = 2048 two workgroups share each 16-row tile of the fp32
    recurrent forward (rows j < 2 / j >= 2 of the MFMA's D layout).  Same arithmetic per row: outputs, saved gates and cell
    states equal the full-tile kernel's to the last bit."""
    from lstm_ode_bci_amd import _lib, ops
    H = 128
    whh = _rnd((D, 4 * H, H), 0.08, dev, 15)
    x = _rnd((T * Bp, H), 1.0, dev, 16)
    wih = _rnd((D * 4 * H, H), 0.08, dev, 17)
    bias = _rnd((D * 4 * H,), 0.1, dev, 18)
    P0 = ops.gate_gemm_x(x, wih, bias, T, Bp, H, D, True, mixed=False, exact=True)
    out = {}
    for v in (1, 2, 4, 0):           # the default split for this tile count, two / four workgroups per tile forced, full tiles
        P = P0.clone()
        with _lib.variant(REC_HALF=v):
            Y, Cs, _, _ = ops.lstm_rec_fwd(P, whh, T, Bp, H, D, save)
        out[v] = (Y, Cs, P)
    for v in (1, 2, 4):
        assert torch.equal(out[v][0], out[0][0])
        if save:
            assert torch.equal(out[v][1], out[0][1]) and torch.equal(out[v][2], out[0][2])
    assert torch.isfinite(out[1][0]).all() and out[1][0].abs().max().item() > 0.01


@pytest.mark.parametrize("T,B", [(20, 512), (9, 96), (33, 17), (256, 256)])
def test_half_tile_mixed_inference_forward_is_bit_identical(dev, T, B):
    """The mixed inference forward (the reference's GPU inference arithmetic, 06:349) on half tiles below B = 1024: whole
    model under autocast + no_grad, LOB_VAR_REC_HALF = 1 against 0: logits and attention weights equal to the last bit (the
    fp32 path's recurrent forward runs half tiles in the same call range: covered by the second comparison)."""
    from lstm_ode_bci_amd import EnhancedLSTMModel, _lib
    from lstm_ode_bci_amd import synthetic as syn
    sd = syn.make_state_dict(61, 128, 3, 2, True)
    x, _ = syn.make_windows(B, T, 61, seed=6)
    m = EnhancedLSTMModel(61, 128, 3, 2, 0.4, True)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    m = m.to(dev).eval()
    xd = torch.from_numpy(x).to(dev)
    res = {}
    for v in (1, 2, 4, 0):
        with _lib.variant(REC_HALF=v), torch.no_grad():
            with torch.autocast("cuda", dtype=torch.bfloat16):
                lm, am = m(xd, return_attention=True)
            lf, af = m(xd, return_attention=True)
        res[v] = (lm.clone(), am.clone(), lf.clone(), af.clone())
    for v in (1, 2, 4):
        for a, b in zip(res[v], res[0]):
            assert torch.equal(a, b)
    assert torch.isfinite(res[1][0]).all()


@pytest.mark.parametrize("T,Bp,D,drop", [(12, 64, 2, 0.4), (5, 32, 1, 0.0), (40, 512, 2, 0.4)])
def test_h256_part_tiles_forward_and_bptt_are_bit_identical(dev, T, Bp, D, drop):
    """H = 256 (the reference's checkpoint size) on few tiles -- its own training batch of 512 windows is 16 tiles per direction:
    four workgroups per 32-row tile (LOB_VAR_REC_HALF) in the saving forward (with the fused dropout copy) and in BPTT against
    full tiles: saved gates, cell states, outputs and dP equal to the last bit; the bias gradient (atomics over four times the
    workgroups) to fp32 rounding."""
    from lstm_ode_bci_amd import _lib, ops
    H = 256
    whh = _rnd((D, 4 * H, H), 0.05, dev, 25)
    x = _rnd((T * Bp, 2 * H), 1.0, dev, 26).to(torch.bfloat16)
    wih = _rnd((D * 4 * H, 2 * H), 0.05, dev, 27).to(torch.bfloat16)
    bias = _rnd((D * 4 * H,), 0.1, dev, 28)
    P0 = ops.gate_gemm_x(x, wih, bias, T, Bp, H, D, True, mixed=True)
    dY = _rnd((T * Bp, D * H), 1e-3, dev, 29).to(torch.bfloat16)
    res = {}
    for v in (1, 0):
        with _lib.variant(REC_HALF=v):
            P = P0.clone()
            Y, Cs, Y16, Yd = ops.lstm_rec_fwd(P, whh, T, Bp, H, D, True, mixed=True, drop_p=drop, seed=11, want_f32=False,
                                              want_bf16=True)
            dP, db = ops.lstm_rec_bwd(P, Cs, whh, dY, T, Bp, H, D, dp_bf16=True)
            Yi, _, Y16i, _ = ops.lstm_rec_fwd(P0.clone(), whh, T, Bp, H, D, False, mixed=True, want_f32=True, want_bf16=True)
        res[v] = (P, Cs, Y16, Yd, dP, db, Yi, Y16i)
    for i, (a, b) in enumerate(zip(res[1], res[0])):
        if a is None:
            assert b is None
        elif i == 5:
            assert (a - b).abs().max().item() <= 1e-5 * max(b.abs().max().item(), 1e-12)
        else:
            assert torch.equal(a, b), i
    assert torch.isfinite(res[1][4].float()).all() and res[1][4].float().abs().max().item() > 0
