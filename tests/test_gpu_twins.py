"""Full-size parity of the kernels the bench times (run with ``-m gpu`` on an MI355X).

The hand-synchronised kernels (LDS-DMA rings, hand-counted ``s_waitcnt vmcnt(N)``) are only correct if the counts are
right, and a stale-ring read shows up under load, at the step's REAL shapes -- not at B = 40.  So each of them is run
at rows = T * Bp = 1,048,576 (B = 4096, T = 256) against its slower twin, selected per call through the test-only
variant table of the C-ABI (``lob_debug_set_variant``, include/lob.h):

* same arithmetic in the same order -> the outputs must be BIT-IDENTICAL (BPTT dP, gate GEMM, NT GEMM);
* split-k partial sums added with fp32 atomics (weight gradients) -> equal to fp32 rounding.

Plus the mixed fwd+bwd of the whole model at B = 4096 (reference training step, 04_lstm_model.py:482-512 under
autocast 04:487): gradient additivity over a ragged batch split, and mixed gradients against the fp32 path's on the
same weights, per tensor, worst tensor printed.
"""
import numpy as np
import pytest
import torch

from lstm_ode_bci_amd import synthetic as syn

pytestmark = pytest.mark.gpu

T, B = 256, 4096


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    from lstm_ode_bci_amd import _lib
    assert _lib.lib().lob_version() >= 200
    return torch.device("cuda:0")


def _rand(shape, dev, seed, scale=1.0, dtype=torch.float32):
    g = torch.Generator(device=dev).manual_seed(seed)
    return (torch.randn(shape, generator=g, device=dev) * scale).to(dtype)


@pytest.fixture(scope="module")
def layer_state(dev):
    """One H = 128 layer at the step's real shapes, through the product kernels: P -> (G, c, Y16) -> dP."""
    from lstm_ode_bci_amd import ops
    H, D, K = 128, 2, 256
    Bp = ops.ceil32(B)
    rows = T * Bp
    x = _rand((rows, K), dev, 1, dtype=torch.bfloat16)
    wih = (_rand((D * 4 * H, K), dev, 2, 0.06)).to(torch.bfloat16)
    bias = _rand((D * 4 * H,), dev, 3, 0.1)
    whh = _rand((D, 4 * H, H), dev, 4, 0.06)
    P = ops.gate_gemm_x(x, wih, bias, T, Bp, H, D, True, mixed=True)
    G = P.clone()
    assert ops.C_BF16
    Y, Cs, Y16, _ = ops.lstm_rec_fwd(G, whh, T, Bp, H, D, True, mixed=True, want_f32=False, want_bf16=True)
    assert Cs.dtype == torch.bfloat16                 # default: bf16 saved cell states
    ops.C_BF16 = False
    try:
        G32 = P.clone()
        _, Cs32, Y16b, _ = ops.lstm_rec_fwd(G32, whh, T, Bp, H, D, True, mixed=True, want_f32=False, want_bf16=True)
    finally:
        ops.C_BF16 = True
    # the storage type of c changes nothing the forward computes
    assert Cs32.dtype == torch.float32 and torch.equal(G32, G) and torch.equal(Y16b, Y16)
    dY = _rand((rows, D * H), dev, 5, 1e-3)
    return dict(H=H, D=D, K=K, Bp=Bp, rows=rows, x=x, wih=wih, bias=bias, whh=whh, P=P, G=G, Cs=Cs, Cs32=Cs32, Y16=Y16, dY=dY)


def test_gate_gemm_weight_stationary_bit_identical_to_tiled(dev, layer_state):
    """gate_gemm_ws_kernel (weights in registers, activations through a 4-slot LDS-DMA ring, vmcnt(32)) against
    gemm_nt_dma_kernel<1,256,256>: same MFMA, same k order, bias added last -> bit-identical P, K = 256 and 128."""
    from lstm_ode_bci_amd import _lib, ops
    s = layer_state
    for K in (256, 128):
        x = s["x"][:, :K].contiguous()
        w = s["wih"][:, :K].contiguous()
        with _lib.variant(GATE_WS=1):
            p_ws = ops.gate_gemm_x(x, w, s["bias"], T, s["Bp"], s["H"], s["D"], True, mixed=True)
            p_ws2 = ops.gate_gemm_x(x, w, s["bias"], T, s["Bp"], s["H"], s["D"], True, mixed=True)
        with _lib.variant(GATE_WS=0):
            p_tl = ops.gate_gemm_x(x, w, s["bias"], T, s["Bp"], s["H"], s["D"], True, mixed=True)
        assert p_ws.dtype == torch.bfloat16 and torch.equal(p_ws, p_ws2)
        nbad = int((p_ws.view(torch.int16) != p_tl.view(torch.int16)).sum())
        assert nbad == 0, f"K={K}: {nbad} of {p_ws.numel()} elements differ"


@pytest.mark.parametrize("K", [512, 256])
def test_gate_gemm_weight_stationary_h256_full_size_and_ragged(dev, K):
    """H = 256 (the reference's checkpoint size, 04_lstm_model.py:877): the weight-stationary gate GEMM at K = 512 (32
    columns per wave, 32-row tiles, vmcnt(14)) and K = 256, at B = 4096 and at ragged row counts, against the tiled /
    register-staged kernels -- same MFMA in the same k order -> bit-identical fragment-order P."""
    from lstm_ode_bci_amd import _lib, ops
    H = 256
    for Tn, Bn, D in ((T, B, 2), (3, 32, 2), (5, 96, 1), (1, 32, 2), (7, 160, 2)):
        x = _rand((Tn * Bn, K), dev, 31 + Tn, dtype=torch.bfloat16)
        w = _rand((D * 4 * H, K), dev, 32 + Bn, 0.04, dtype=torch.bfloat16)
        bias = _rand((D * 4 * H,), dev, 33, 0.1)
        with _lib.variant(GATE_WS=1):
            p_ws = ops.gate_gemm_x(x, w, bias, Tn, Bn, H, D, True, mixed=True)
        with _lib.variant(GATE_WS=0):
            p_tl = (ops.gate_gemm_x(x, w, bias, Tn, Bn, H, D, True, mixed=True) if (Tn * Bn) % 256 == 0 else
                    ops.gate_gemm_x(x, w.float(), bias, Tn, Bn, H, D, True, mixed=True))    # register-staged kernel
        assert p_ws.dtype == torch.bfloat16
        nbad = int((p_ws.view(torch.int16) != p_tl.view(torch.int16)).sum())
        assert nbad == 0, f"K={K} T={Tn} B={Bn} D={D}: {nbad} of {p_ws.numel()} elements differ"


@pytest.mark.parametrize("Tn,Bn,K", [(3, 32, 256), (5, 96, 128), (7, 160, 256), (1, 32, 128), (12, 32, 256)])
def test_gate_gemm_weight_stationary_ragged(dev, Tn, Bn, K):
    """Tile counts below the ring depth, M % 64 == 32 tails, unidirectional: against a float64 product of the bf16
    operands, un-permuted from the fragment order by the recurrent kernel's own layout rule (include/lob.h)."""
    from lstm_ode_bci_amd import _lib, ops
    H = 128
    for D in (1, 2):
        x = _rand((Tn * Bn, K), dev, 11 + Tn, dtype=torch.bfloat16)
        w = _rand((D * 4 * H, K), dev, 12 + Bn, 0.06, dtype=torch.bfloat16)
        bias = _rand((D * 4 * H,), dev, 13, 0.1)
        with _lib.variant(GATE_WS=1):
            p_ws = ops.gate_gemm_x(x, w, bias, Tn, Bn, H, D, True, mixed=True)
        with _lib.variant(GATE_WS=0):
            p_tl = (ops.gate_gemm_x(x, w, bias, Tn, Bn, H, D, True, mixed=True) if (Tn * Bn) % 256 == 0 else
                    ops.gate_gemm_x(x, w.float(), bias, Tn, Bn, H, D, True, mixed=True))    # register-staged kernel
        assert torch.equal(p_ws, p_tl), (Tn, Bn, K, D)


@pytest.mark.parametrize("c_key", ["Cs", "Cs32"])
@pytest.mark.parametrize("dy_dtype", [torch.float32, torch.bfloat16])
def test_bptt_dma_ring_bit_identical_to_register_prefetch(dev, layer_state, dy_dtype, c_key):
    """lstm_rec_bwd_h128_bf16_s16_dma_kernel (wave-private LDS-DMA ring, counted vmcnt: 14/18 with fp32 cell states,
    13/17 with bf16 ones) against the register-prefetch kernel: same arithmetic in the same order -> dP bit-identical
    over all 1M rows; the bias gradient (fp32 atomics over 512 workgroups) to rounding.  All four combinations of the
    storage types of the incoming gradient (fp32 | bf16: hand-issued global_load_ushort) and of the saved cell state."""
    from lstm_ode_bci_amd import _lib, ops
    s = layer_state
    dY = s["dY"].to(dy_dtype)
    Cs = s[c_key]
    with _lib.variant(REC_BWD_DMA=1):
        dP1, db1 = ops.lstm_rec_bwd(s["G"], Cs, s["whh"], dY, T, s["Bp"], s["H"], s["D"], dp_bf16=True)
    with _lib.variant(REC_BWD_DMA=0):
        dP0, db0 = ops.lstm_rec_bwd(s["G"], Cs, s["whh"], dY, T, s["Bp"], s["H"], s["D"], dp_bf16=True)
    if dy_dtype == torch.bfloat16:          # a bf16 dY == the same values handed over as fp32
        with _lib.variant(REC_BWD_DMA=1):
            dPf, _ = ops.lstm_rec_bwd(s["G"], Cs, s["whh"], dY.float(), T, s["Bp"], s["H"], s["D"], dp_bf16=True)
        assert torch.equal(dP1.view(torch.int16), dPf.view(torch.int16))
    if c_key == "Cs":                       # bf16 cell states: close to the fp32-c result (one rounding of c per use)
        with _lib.variant(REC_BWD_DMA=1):
            dPc, _ = ops.lstm_rec_bwd(s["G"], s["Cs32"], s["whh"], dY, T, s["Bp"], s["H"], s["D"], dp_bf16=True)
        rel = (dP1.float() - dPc.float()).abs().max().item() / dPc.float().abs().max().item()
        assert 0 < rel < 2e-2, rel
    nbad = int((dP1.view(torch.int16) != dP0.view(torch.int16)).sum())
    assert nbad == 0, f"{nbad} of {dP1.numel()} dP elements differ"
    assert torch.isfinite(dP1.float()).all() and dP1.float().abs().max().item() > 0
    assert (db1 - db0).abs().max().item() <= 1e-4 * db0.abs().max().item() + 1e-9


def test_nt_dma_gemm_matches_register_staged_twin(dev, layer_state):
    """dX = dP W_ih through gemm_nt_dma_kernel<0,256,256,...,ADEEP> (LDS-DMA rings, VM_STEADY / VM_EPI) against the
    register-staged gemm_nt_bf16_kernel on the same bf16 values (M = 1,048,576, K = 1024, N = 256): same MFMA and
    k order -> identical to fp32 rounding (both accumulate 64 k-steps in the same sequence: in practice bit-equal)."""
    from lstm_ode_bci_amd import ops
    s = layer_state
    dP = _rand((s["rows"], s["D"] * 4 * s["H"]), dev, 21, 1e-2, dtype=torch.bfloat16)
    wt = s["wih"].t().contiguous()                        # (256, 1024) bf16
    out_dma = ops.gemm_nt(dP, wt, mixed=True)
    out_reg = ops.gemm_nt(dP, wt.float(), mixed=True)     # fp32 weights -> register-staged kernel (rounds to the same bf16)
    diff = (out_dma - out_reg).abs().max().item()
    assert diff <= 1e-6 * out_reg.abs().max().item(), diff
    # and with the fused dropout-backward epilogue
    o2 = ops.gemm_nt(dP, wt, mixed=True, drop_p=0.4, seed=77)
    o3 = ops.gemm_nt(dP, wt.float(), mixed=True, drop_p=0.4, seed=77)
    assert (o2 - o3).abs().max().item() <= 1e-6 * o3.abs().max().item()


def test_fused_dw_matches_separate_tn_gemms(dev, layer_state):
    """lstm_dw_h128_kernel<256> (4-slot LDS-DMA ring, inline-asm transposing reads, hand-counted lgkmcnt) against the
    register-staged TN GEMM twins at rows = 1,048,576: split-k fp32 atomics -> equal to fp32 rounding of a
    1M-term sum (2e-4 of the largest entry)."""
    from lstm_ode_bci_amd import _lib, ops
    s = layer_state
    H, D, Bp = s["H"], s["D"], s["Bp"]
    dP = _rand((s["rows"], D * 4 * H), dev, 31, 1e-2, dtype=torch.bfloat16)
    X, Y = s["x"], s["Y16"]
    assert ops.can_fuse_dw(dP, X, Y, T, Bp, H, D)
    dwih, dwhh = ops.lstm_dw(dP, X, Y, T, Bp, H, D)
    with _lib.variant(NT_DMA=0):                        # register-staged TN kernel
        ref_ih = torch.zeros_like(dwih)
        ops.gemm_tn(dP, X, ref_ih, mixed=True)
        ref_hh = torch.zeros_like(dwhh)
        ops.gemm_tn(dP[Bp:, :4 * H], Y[:(T - 1) * Bp, :H], ref_hh[0], mixed=True)
        ops.gemm_tn(dP[:(T - 1) * Bp, 4 * H:], Y[Bp:, H:], ref_hh[1], mixed=True)
    for got, ref, nm in ((dwih, ref_ih, "dW_ih"), (dwhh, ref_hh, "dW_hh")):
        err = (got - ref).abs().max().item()
        assert err <= 2e-4 * ref.abs().max().item(), (nm, err, ref.abs().max().item())
    with _lib.variant(NT_DMA=1):                        # and the LDS-DMA TN kernel against the same reference
        dma_ih = torch.zeros_like(dwih)
        ops.gemm_tn(dP, X, dma_ih, mixed=True)
    assert (dma_ih - ref_ih).abs().max().item() <= 2e-4 * ref_ih.abs().max().item()


def test_h256_full_size_kernels_vs_twins(dev):
    """H = 256 (the reference's real checkpoint size, 04_lstm_model.py:877) at B = 4096: the streamed-W_hh bf16
    recurrent kernels on a full-size layer.  Forward: h of the first steps against a torch restatement on a sample of
    windows; BPTT: run-to-run bit identity of dP, and dP of the first BPTT steps of each direction against an fp64
    restatement on the same sample."""
    from lstm_ode_bci_amd import _lib, ops
    H, D, K = 256, 2, 512
    Bp = ops.ceil32(B)
    rows = T * Bp
    x = _rand((rows, K), dev, 41, dtype=torch.bfloat16)
    wih = _rand((D * 4 * H, K), dev, 42, 0.04, dtype=torch.bfloat16)
    bias = _rand((D * 4 * H,), dev, 43, 0.1)
    whh = _rand((D, 4 * H, H), dev, 44, 0.04)
    P = ops.gate_gemm_x(x, wih, bias, T, Bp, H, D, True, mixed=True)
    assert P.dtype == torch.bfloat16
    G = P.clone()
    Y, Cs, Y16, _ = ops.lstm_rec_fwd(G, whh, T, Bp, H, D, True, mixed=True, want_f32=True, want_bf16=True)
    # forward direction, first 3 steps, windows 0..7 and the last 8: fp64 restatement on the same bf16 operands
    idx = torch.cat([torch.arange(0, 8), torch.arange(B - 8, B)]).to(dev)
    w_ih0 = wih[:4 * H].double()
    w_hh0 = whh[0].to(torch.bfloat16).double()
    h = torch.zeros((16, H), device=dev, dtype=torch.float64)
    c = torch.zeros_like(h)
    for t in range(3):
        xt = x[t * Bp + idx].double()
        z = (xt @ w_ih0.T + bias[:4 * H].double()).to(torch.bfloat16).double() + h.to(torch.bfloat16).double() @ w_hh0.T
        i, f, g, o = z[:, :H].sigmoid(), z[:, H:2 * H].sigmoid(), z[:, 2 * H:3 * H].tanh(), z[:, 3 * H:].sigmoid()
        c = f * c + i * g
        h = o * c.tanh()
        got = Y[t * Bp + idx, :H].double()
        assert (got - h).abs().max().item() < 2e-3, t
    dY = _rand((rows, D * H), dev, 45, 1e-3)
    dP1, db1 = ops.lstm_rec_bwd(G, Cs, whh, dY, T, Bp, H, D, dp_bf16=True)
    dP0, db0 = ops.lstm_rec_bwd(G, Cs, whh, dY, T, Bp, H, D, dp_bf16=True)
    assert torch.isfinite(dP1.float()).all() and dP1.float().abs().max().item() > 0
    # the BPTT kernel has no atomics on dP: two runs agree bit for bit (a stale weight fragment or a race would not)
    assert int((dP1.view(torch.int16) != dP0.view(torch.int16)).sum()) == 0
    assert (db1 - db0).abs().max().item() <= 1e-4 * db0.abs().max().item() + 1e-9
    # BPTT against an fp64 restatement (Appendix A.2 of SURVEY.md differentiated) on the kernel's own saved gates / cell
    # states, for the first three BPTT steps of each direction on the first and the last 8 windows
    NBT = Bp // 32
    Gr = (G.view(D, T, NBT, 8, 4, 2, 2, 32, 2, 4)             # d t bt w g pq hi c ehi elo (include/lob.h, H = 256)
          .permute(1, 2, 5, 8, 6, 9, 0, 4, 3, 7).reshape(T, Bp, D, 4, H))   # row = 16 pq + 8 ehi + 4 hi + elo, unit = 32 w + c
    assert Cs.dtype == torch.bfloat16          # default: bf16 saved cell states, in the element order of one saved gate
    Cr = (Cs.view(D, T, NBT, 8, 2, 2, 32, 2, 4)               # d t bt w pq hi c ehi elo
          .permute(1, 2, 4, 7, 5, 8, 0, 3, 6).reshape(T, Bp, D, H))
    dPr = dP1.view(T, Bp, D, 4, H)
    for d in range(D):
        w64 = whh[d].to(torch.bfloat16).double()                           # (4H, H)
        order = range(T - 1, T - 4, -1) if d == 0 else range(0, 3)
        dhrec = torch.zeros((16, H), device=dev, dtype=torch.float64)
        dcar = torch.zeros_like(dhrec)
        for t in order:
            tp = t - 1 if d == 0 else t + 1                                # the step before t in this direction's time
            gi, gf, gg, go = (Gr[t, idx, d, k].double() for k in range(4))
            ct, cp = Cr[t, idx, d].double(), Cr[tp, idx, d].double()
            dh = dY[t * Bp + idx, d * H:(d + 1) * H].double() + dhrec
            tc = ct.tanh()
            dc = dcar + dh * go * (1 - tc * tc)
            dcar = dc * gf
            dg4 = torch.stack([dc * gg * gi * (1 - gi), dc * cp * gf * (1 - gf), dc * gi * (1 - gg * gg),
                               dh * tc * go * (1 - go)], 1)                # (16, 4, H)
            got = dPr[t, idx, d].double()
            scale = dg4.abs().max().item()
            assert (got - dg4).abs().max().item() <= 1e-2 * scale, (d, t)
            dhrec = dg4.to(torch.bfloat16).double().reshape(16, 4 * H) @ w64
    # kernel twins (LOB_VAR_H256_LDSW): part of each wave's weight fragments resident in LDS against every fragment
    # streamed -- the same MFMAs in the same order: forward outputs and dP must be bit-identical
    with _lib.variant(H256_LDSW=0):
        G0 = P.clone()
        Y0, Cs0, Y160, _ = ops.lstm_rec_fwd(G0, whh, T, Bp, H, D, True, mixed=True, want_f32=True, want_bf16=True)
        dPs, dbs = ops.lstm_rec_bwd(G, Cs, whh, dY, T, Bp, H, D, dp_bf16=True)
        Yi0, _, Y16i0, _ = ops.lstm_rec_fwd(P, whh, T, Bp, H, D, False, mixed=True, want_f32=True, want_bf16=True)
    assert _lib.get_variant("H256_LDSW") == 1
    Yi1, _, Y16i1, _ = ops.lstm_rec_fwd(P, whh, T, Bp, H, D, False, mixed=True, want_f32=True, want_bf16=True)
    assert torch.equal(G0, G) and torch.equal(Y0, Y) and torch.equal(Cs0, Cs) and torch.equal(Y160, Y16)
    assert torch.equal(Yi0, Yi1) and torch.equal(Y16i0, Y16i1) and torch.equal(Yi1, Y)
    assert torch.equal(dPs, dP1) and (dbs - db1).abs().max().item() <= 1e-4 * db1.abs().max().item() + 1e-9
    # storage types: a bf16 dY carries the same values as its widened copy -> bit-identical dP; fp32 cell states give
    # the same forward and a dP that differs by the rounding of c only
    dY16 = dY.to(torch.bfloat16)
    dPa, _ = ops.lstm_rec_bwd(G, Cs, whh, dY16, T, Bp, H, D, dp_bf16=True)
    dPb, _ = ops.lstm_rec_bwd(G, Cs, whh, dY16.float(), T, Bp, H, D, dp_bf16=True)
    assert torch.equal(dPa, dPb)
    ops.C_BF16 = False
    try:
        G32 = P.clone()
        Y_b, Cs32, Y16_b, _ = ops.lstm_rec_fwd(G32, whh, T, Bp, H, D, True, mixed=True, want_f32=True, want_bf16=True)
    finally:
        ops.C_BF16 = True
    assert Cs32.dtype == torch.float32 and torch.equal(G32, G) and torch.equal(Y16_b, Y16) and torch.equal(Y_b, Y)
    dPc, _ = ops.lstm_rec_bwd(G, Cs32, whh, dY, T, Bp, H, D, dp_bf16=True)
    ref = dPc.float()
    assert (dP1.float() - ref).abs().max().item() <= 2e-2 * ref.abs().max().item()


def test_full_size_mixed_fwd_bwd_properties(dev):
    """B = 4096, T = 256, H = 128 under autocast (the bench's default step): (i) gradient additivity over the ragged
    1504 / 2592 batch split with a sum-reduced loss -- every hand-counted kernel runs at three different grid sizes and
    must produce the same sums; (ii) mixed gradients against the fp32 path's on the same weights, <= 2e-2 of the
    tensor's largest fp32 entry, worst tensor printed."""
    from lstm_ode_bci_amd import EnhancedLSTMModel
    sd = syn.make_state_dict(61, 128, 3, 2, True)
    x, y = syn.make_windows(B)
    m = EnhancedLSTMModel(input_size=61, hidden_size=128, num_layers=3, num_classes=2, dropout=0.4, bidirectional=True)
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
    m = m.to(dev).eval()
    xt, yt = torch.from_numpy(x).to(dev), torch.from_numpy(y).to(dev)

    def grads(xs, ys, mixed):
        m.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=mixed):
            loss = torch.nn.functional.cross_entropy(m(xs).float(), ys, reduction="sum")
        loss.backward()
        return {k: p.grad.clone() for k, p in m.named_parameters()}
    g_all = grads(xt, yt, True)
    g_a, g_b = grads(xt[:1504], yt[:1504], True), grads(xt[1504:], yt[1504:], True)
    for k in g_all:
        ref_ = g_a[k] + g_b[k]
        # the three runs round the same bf16 values (windows are independent); only fp32 summation order differs
        assert (g_all[k] - ref_).abs().max().item() <= 3e-4 * ref_.abs().max().item() + 1e-7, k
    g32 = grads(xt, yt, False)
    worst, wk = 0.0, None
    for k in g32:
        sc = g32[k].abs().max().item()
        if sc < 1e-7:
            continue
        e = (g_all[k] - g32[k]).abs().max().item() / sc
        if e > worst:
            worst, wk = e, k
    print(f"mixed vs fp32 gradients at B={B}: worst tensor {wk} rel err {worst:.3e}")
    assert worst <= 2e-2, (wk, worst)


def _unfrag_f32(P, T_, Bp, D):
    """fp32 fragment-order P (include/lob.h: [d][t][bt][w 4][gate 4][q 4][lane 64][4]) -> row-major (T*Bp, D*512):
    lane = 32 lh + c; row in the 32-row block = 8 q + 4 lh + e, column = d*512 + gate*128 + w*32 + c."""
    v = P.view(D, T_, Bp // 32, 4, 4, 4, 2, 32, 4)           # d t bt w g q lh c e
    return v.permute(1, 2, 5, 6, 8, 0, 4, 3, 7).reshape(T_ * Bp, D * 512)


def test_fp32_split_gate_gemm_and_recurrence_vs_exact_fp32_mfma(dev):
    """The fp32 path's default kernels carry every operand as two fp16 halves (22 bits) on the 16-bit matrix pipe
    (gate_gemm_ws_split.hip, lstm_rec_f32_split.hip); LOB_VAR_F32_SPLIT = 0 selects the exact-fp32 MFMA kernels.  At
    B = 4096, T = 256 (configs[1]/[3] arithmetic): the split gate GEMM against a float64 product is as close as the
    exact-fp32 MFMA kernel is (errors of both are fp32 accumulation noise), and a whole layer (256 recurrent steps)
    stays within 5e-6 of the exact kernels' h."""
    from lstm_ode_bci_amd import _lib, ops
    H, D, K = 128, 2, 256
    Bp = ops.ceil32(B)
    rows = T * Bp
    x = _rand((rows, K), dev, 51)                      # layer-1 input scale (|h| <= 1 x dropout scale)
    wih = _rand((D * 4 * H, K), dev, 52, 0.06)
    bias = _rand((D * 4 * H,), dev, 53, 0.1)
    whh = _rand((D, 4 * H, H), dev, 54, 0.06)
    with _lib.variant(F32_SPLIT=1):
        p_s = ops.gate_gemm_x(x, wih, bias, T, Bp, H, D, True)
        p_s2 = ops.gate_gemm_x(x, wih, bias, T, Bp, H, D, True)
    with _lib.variant(F32_SPLIT=0):
        p_e = ops.gate_gemm_x(x, wih, bias, T, Bp, H, D, True)
    assert p_s.dtype == torch.float32 and torch.equal(p_s, p_s2)
    d_se = (p_s - p_e).abs().max().item()
    # float64 truth on a sample of rows (first, middle, last time steps; ragged offsets inside the 32-row blocks)
    idx = torch.cat([torch.arange(0, 96), torch.arange(rows // 2 + 5, rows // 2 + 69), torch.arange(rows - 64, rows)]).to(dev)
    ref = x[idx].double() @ wih.double().T + bias.double()
    e_s = (_unfrag_f32(p_s, T, Bp, D)[idx].double() - ref).abs().max().item()
    e_e = (_unfrag_f32(p_e, T, Bp, D)[idx].double() - ref).abs().max().item()
    print(f"gate GEMM K={K}: |split - exact| {d_se:.2e}; vs float64: split {e_s:.2e}, exact fp32 MFMA {e_e:.2e}")
    # (measured: split 1.0e-6, exact 2.9e-6 -- the split kernel's two accumulators lose fewer low bits than one fp32 chain)
    assert e_e < 6e-6 and e_s < max(2 * e_e, 2e-6) and d_se < 1.2e-5
    for Kx in (128,):                                  # layer-0 shape
        xs, ws = x[:, :Kx].contiguous(), wih[:, :Kx].contiguous()
        with _lib.variant(F32_SPLIT=1):
            a = ops.gate_gemm_x(xs, ws, bias, T, Bp, H, D, True)
        with _lib.variant(F32_SPLIT=0):
            b = ops.gate_gemm_x(xs, ws, bias, T, Bp, H, D, True)
        assert (a - b).abs().max().item() < 8e-6
    # one whole layer: 256 dependent steps
    with _lib.variant(F32_SPLIT=1):
        y_s, _, _, _ = ops.lstm_rec_fwd(p_e.clone(), whh, T, Bp, H, D, False)
        pk = p_e.clone()
        y_sv, c_sv, _, _ = ops.lstm_rec_fwd(pk, whh, T, Bp, H, D, True)
    with _lib.variant(F32_SPLIT=0):
        y_e, _, _, _ = ops.lstm_rec_fwd(p_e.clone(), whh, T, Bp, H, D, False)
        pk2 = p_e.clone()
        y_ev, c_ev, _, _ = ops.lstm_rec_fwd(pk2, whh, T, Bp, H, D, True)
    dy = (y_s - y_e).abs().max().item()
    print(f"recurrent layer, 256 steps: |h_split - h_exact| max {dy:.2e}")
    assert dy < 5e-6 and torch.equal(y_s, y_sv)                    # save mode computes the same h
    assert (pk - pk2).abs().max().item() < 5e-6 and (c_sv - c_ev).abs().max().item() < 2e-5      # saved gates, c


def test_bf16_gradient_carries_round_the_same_fp32_values(dev, layer_state):
    """ops.DY_BF16_CARRY: dX = dP W_ih written as bf16 by the NT GEMM's epilogue and the LayerNorm backward's bf16 dy in /
    dx out are the fp32 kernels' values rounded once (RNE) -- bit-exact against rounding the fp32 results on the host
    side; at the step's real shapes."""
    from lstm_ode_bci_amd import ops
    s = layer_state
    rows = s["rows"]
    dP = _rand((rows, s["D"] * 4 * s["H"]), dev, 61, 1e-2, dtype=torch.bfloat16)
    wt = s["wih"].t().contiguous()
    for kw in (dict(), dict(drop_p=0.4, seed=5)):
        o32 = ops.gemm_nt(dP, wt, mixed=True, **kw)
        o16 = ops.gemm_nt(dP, wt, mixed=True, out_bf16=True, **kw)
        assert o16.dtype == torch.bfloat16 and torch.equal(o16, o32.to(torch.bfloat16))
    # LayerNorm backward, width 256 (post-LSTM): dy fp32|bf16 -> dx bf16, with the fused attention-context term
    W = 256
    x = _rand((rows, W), dev, 62)
    gam, bet = _rand((W,), dev, 63) + 1.0, _rand((W,), dev, 64)
    dy = _rand((rows, W), dev, 65, 1e-2)
    attn = torch.softmax(_rand((B, T), dev, 66), 1).contiguous()
    dctx = _rand((B, W), dev, 67, 1e-2)
    pool = (attn, dctx, T, B, s["Bp"])
    ref, rg, rb = ops.layernorm_act_bwd(x, gam, bet, dy, pool=pool)
    got, gg, gb = ops.layernorm_act_bwd(x, gam, bet, dy, pool=pool, dx_bf16=True)
    assert got.dtype == torch.bfloat16 and torch.equal(got, ref.to(torch.bfloat16))
    assert (gg - rg).abs().max().item() <= 1e-4 * rg.abs().max().item()          # atomics: order only
    dy16 = dy.to(torch.bfloat16)
    ref2, _, _ = ops.layernorm_act_bwd(x, gam, bet, dy16.float(), pool=pool)
    got2, _, _ = ops.layernorm_act_bwd(x, gam, bet, dy16, pool=pool, dx_bf16=True)
    assert torch.equal(got2, ref2.to(torch.bfloat16))
    # width 128 with the (b,t) -> (t,b) relayout, GELU and dropout (input projection): dy bf16 -> dx fp32
    Bs, Ts, W1 = 96, 40, 128
    x1 = _rand((Bs * Ts, W1), dev, 71)
    g1, b1 = _rand((W1,), dev, 72) + 1.0, _rand((W1,), dev, 73)
    dy1 = _rand((Ts * Bs, W1), dev, 74, 1e-2).to(torch.bfloat16)
    kw = dict(act=ops.ACT_GELU, remap=(Ts, Bs, Bs), drop_p=0.2, seed=9)
    r32, _, _ = ops.layernorm_act_bwd(x1, g1, b1, dy1.float(), **kw)
    r16, _, _ = ops.layernorm_act_bwd(x1, g1, b1, dy1, **kw)
    assert r16.dtype == torch.float32 and torch.equal(r16, r32)


def test_mixed_step_with_and_without_bf16_carries(dev):
    """The whole mixed backward with the gradient carries and the saved cell states in bf16 (default) against fp32
    storage of both: the two differ by one extra rounding per layer boundary / per use of c -- well inside the mixed
    tolerance; worst tensor printed."""
    from lstm_ode_bci_amd import EnhancedLSTMModel, ops
    sd = syn.make_state_dict(61, 128, 3, 2, True)
    x, y = syn.make_windows(256)
    m = EnhancedLSTMModel(input_size=61, hidden_size=128, num_layers=3, num_classes=2, dropout=0.4, bidirectional=True)
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
    m = m.to(dev).train()
    xt, yt = torch.from_numpy(x).to(dev), torch.from_numpy(y).to(dev)

    def grads(mixed=True):
        m.zero_grad(set_to_none=True)
        torch.manual_seed(3)                           # same dropout masks
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=mixed):
            torch.nn.functional.cross_entropy(m(xt).float(), yt).backward()
        return {k: p.grad.clone() for k, p in m.named_parameters()}
    assert ops.DY_BF16_CARRY
    g16 = grads()
    ops.DY_BF16_CARRY = False
    ops.C_BF16 = False
    try:
        g32 = grads()
    finally:
        ops.DY_BF16_CARRY = True
        ops.C_BF16 = True
    gref = grads(mixed=False)                          # fp32 path, same masks
    worst = (0.0, None, 0.0)
    for k in g32:
        sc = gref[k].abs().max().item()
        if sc < 1e-7:
            continue
        e16 = (g16[k] - gref[k]).abs().max().item() / sc
        e32 = (g32[k] - gref[k]).abs().max().item() / sc
        if e16 > worst[0]:
            worst = (e16, k, e32)
        assert e16 <= 2e-2, (k, e16)
    print(f"bf16 carries + bf16 c: worst tensor {worst[1]} rel err {worst[0]:.3e} (fp32 storage of both: {worst[2]:.3e})")
    assert any(not torch.equal(g16[k], g32[k]) for k in g16), "the bf16-carry switch changed nothing"


@pytest.mark.parametrize("M,N,K", [(T * 4096, 256, 1024), (T * 4096, 128, 1024), (16, 256, 1024), (48, 128, 512),
                                   (4096 + 16, 256, 512), (2064, 128, 1024)])
def test_dx_ksplit_kernel_vs_tiled_twin(dev, M, N, K):
    """dx_ksplit_kernel (weights stationary, contraction split over the eight waves, hand-issued A loads three tiles
    ahead with a counted vmcnt, cross-wave reduction through LDS) against the tiled LDS-DMA NT GEMM / the register-staged
    kernel on the same bf16 operands: equal to fp32 summation order (the k-split adds eight partial sums); fp32 and bf16
    outputs, with and without the fused dropout-backward mask; full size and tile counts below the look-ahead depth."""
    from lstm_ode_bci_amd import _lib, ops
    a = _rand((M, K), dev, 81, 1e-2, dtype=torch.bfloat16)
    wt = _rand((N, K), dev, 82, 0.06, dtype=torch.bfloat16)
    for kw in (dict(), dict(drop_p=0.4, seed=11)):
        with _lib.variant(DX_KSPLIT=1):
            o_ks = ops.gemm_nt(a, wt, mixed=True, **kw)
            o_ks2 = ops.gemm_nt(a, wt, mixed=True, **kw)
            o_ks16 = ops.gemm_nt(a, wt, mixed=True, out_bf16=True, **kw)
        with _lib.variant(DX_KSPLIT=0):
            o_tl = ops.gemm_nt(a, wt if M % 256 == 0 else wt.float(), mixed=True, **kw)
        assert torch.equal(o_ks, o_ks2)                               # deterministic (no atomics)
        sc = o_tl.abs().max().item()
        assert (o_ks - o_tl).abs().max().item() <= 2e-6 * sc, (kw, (o_ks - o_tl).abs().max().item(), sc)
        assert torch.equal(o_ks16, o_ks.to(torch.bfloat16))
    if M <= 4096 + 16:                                                # float64 truth on the small cases
        ref = a.double() @ wt.double().T
        with _lib.variant(DX_KSPLIT=1):
            o = ops.gemm_nt(a, wt, mixed=True)
        assert (o.double() - ref).abs().max().item() <= 2e-6 * ref.abs().max().item()


@pytest.mark.parametrize("M,N,K", [(T * 4096, 256, 128), (T * 4096, 512, 256), (96, 256, 128), (32, 256, 256), (4096 + 32, 256, 128)])
def test_row_major_weight_stationary_gemm_vs_tiled_twin(dev, M, N, K):
    """The row-major epilogue of the weight-stationary kernel (operands swapped so that a lane holds 4 consecutive
    columns: dV = dPreU W1 of the attention pooling's backward, K = 128, N = 256) against the tiled LDS-DMA / register-
    staged kernels: same MFMA products in the same k order -> bit-identical; bf16 and fp32 C; M % 64 == 32 tails."""
    from lstm_ode_bci_amd import _lib, ops
    a = _rand((M, K), dev, 91, 1e-2, dtype=torch.bfloat16)
    w = _rand((N, K), dev, 92, 0.06, dtype=torch.bfloat16)
    with _lib.variant(GATE_WS=1):
        o_ws = ops.gemm_nt(a, w, mixed=True)
        o_ws16 = ops.gemm_nt(a, w, mixed=True, out_bf16=True)
    with _lib.variant(GATE_WS=0):
        o_tl = ops.gemm_nt(a, w if M % 256 == 0 else w.float(), mixed=True)
    assert torch.equal(o_ws, o_tl), (o_ws - o_tl).abs().max().item()
    assert torch.equal(o_ws16, o_tl.to(torch.bfloat16))


@pytest.mark.parametrize("nwin", [1, 2, 3])
@pytest.mark.parametrize("outs", ["f32", "bf16", "both"])
def test_few_window_inference_kernel_vs_full_tile_twin(dev, nwin, outs):
    """The single-window serving call (06_lstm_ode_integration.py:340-360 with one (256, 61) window): the FEW variant of
    the mixed recurrent forward skips the cell update of the MFMA output registers that only hold padding rows.  The
    rows that carry windows must equal the full-tile kernel's bit for bit, the skipped
    padding rows must come out as zeros."""
    from lstm_ode_bci_amd import _lib, ops
    H, D, Bp = 128, 2, 32
    P = _rand((D * T * Bp * 4 * H,), dev, 120 + nwin, 0.8).to(torch.bfloat16)
    whh = _rand((D, 4 * H, H), dev, 121, 0.08)
    kw = dict(want_f32=outs in ("f32", "both"), want_bf16=outs in ("bf16", "both"))
    # (REC_HALF = 0: since round 4 two or three windows -- and every batch below 1024 -- run on part tiles by default; FEW stays
    # the product kernel of the one-window call and selectable for 2 / 3 windows, the full-tile kernel is the twin of both)
    with _lib.variant(REC_FEW=1, REC_HALF=0):
        Yf, _, Y16f, _ = ops.lstm_rec_fwd(P, whh, T, Bp, H, D, False, mixed=True, nvalid=nwin, **kw)
    with _lib.variant(REC_FEW=0, REC_HALF=0):
        Yt, _, Y16t, _ = ops.lstm_rec_fwd(P, whh, T, Bp, H, D, False, mixed=True, nvalid=nwin, **kw)
    with _lib.variant(REC_FEW=1, REC_HALF=1):       # the product's choice: FEW for one window, four workgroups per tile otherwise
        Yp, _, Y16p, _ = ops.lstm_rec_fwd(P, whh, T, Bp, H, D, False, mixed=True, nvalid=nwin, **kw)
    for prod, twin in ((Yp, Yt), (Y16p, Y16t)):
        if prod is not None:
            assert torch.equal(prod.reshape(T, Bp, D * H)[:, :nwin], twin.reshape(T, Bp, D * H)[:, :nwin])
    # same arithmetic in the same order (the cell update's contraction is pinned with an explicit fma in every forward
    # kernel: left to hipcc, one instantiation fused fg*c, the other ig*gg, and a last fp32 bit now and then flipped
    # a bf16 rounding of the fed-back h -- 4e-4 after 256 steps): bit-identical
    for few, twin in ((Yf, Yt), (Y16f, Y16t)):
        if few is None:
            assert twin is None
            continue
        few, twin = few.float().reshape(T, Bp, D * H), twin.float().reshape(T, Bp, D * H)
        assert torch.isfinite(few).all()
        assert torch.equal(few[:, :nwin], twin[:, :nwin])
        # padding rows: tile row 4 rq + j lives in output register j, so rows with (row % 4) >= nwin are skipped and
        # leave as zeros (rows 4, 8, 12 share register 0 with window 0 and are computed as before); so does the
        # second tile, which holds no window at all
        r = torch.arange(Bp, device=dev)
        skipped = (r >= 16) | ((r % 4) >= nwin)
        assert float(few[:, skipped].abs().max()) == 0.0
        assert float(twin[:, skipped].abs().max()) > 0.0        # (the full-tile kernel computes them)
        assert torch.equal(few[:, ~skipped], twin[:, ~skipped])


@pytest.mark.parametrize("B1", [1, 3, 40])
def test_no_grad_forward_does_not_save_and_matches_grad_mode(dev, B1):
    """torch.no_grad() inference (04_lstm_model.py:557, 06_lstm_ode_integration.py:347) must not run the saving
    forward kernels: same logits as the grad-mode forward, no autograd node, and (mixed, B < 4) the few-window
    kernel is what runs."""
    from lstm_ode_bci_amd import EnhancedLSTMModel
    sd = syn.make_state_dict(61, 128, 3, 2, True)
    m = EnhancedLSTMModel(61, 128, 3, 2, 0.4, True)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    m = m.to(dev).eval()
    x = _rand((B1, T, 61), dev, 7)
    for mixed in (False, True):
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=mixed):
            lg, ag = m(x, return_attention=True)
            with torch.no_grad():
                ln, an = m(x, return_attention=True)
        assert lg.requires_grad and not ln.requires_grad and ln.grad_fn is None
        tol = 3e-3 if mixed else 1e-5
        assert (lg - ln).abs().max().item() <= tol and (ag - an).abs().max().item() <= tol


def test_layernorm_bf16_input_rows_equal_the_widened_rows(dev):
    """LOB_X_BF16 (the last LSTM layer hands its output to the post-LSTM LayerNorm, 04_lstm_model.py:212, as bf16 only):
    reading bf16 rows must give exactly what the fp32-input kernels give on the same values widened -- forward (bf16 and
    fp32 out) and backward (bf16 dy / dx, fused pooling term), at rows = 1,048,576 and at a ragged row count."""
    from lstm_ode_bci_amd import _lib, ops
    W = 256
    for rows, Tn, Bn in ((T * B, T, B), (3 * 96, 3, 96)):
        x16 = _rand((rows, W), dev, 71, 0.5, dtype=torch.bfloat16)
        gm, bt = 1 + _rand((W,), dev, 72, 0.1), _rand((W,), dev, 73, 0.1)
        for ob in (True, False):
            a = ops.layernorm_act(x16, gm, bt, out_bf16=ob)
            b = ops.layernorm_act(x16.float(), gm, bt, out_bf16=ob)
            assert a.dtype == b.dtype
            assert (a.float() - b.float()).abs().max().item() <= (3e-2 if ob else 1e-5)
        dy = _rand((rows, W), dev, 74, 1e-3, dtype=torch.bfloat16)
        attn = torch.softmax(_rand((Bn, Tn), dev, 75), 1)
        dctx = _rand((Bn, W), dev, 76, 1e-3)
        pool = (attn, dctx, Tn, Bn, Bn)
        dxa, dga, dba = ops.layernorm_act_bwd(x16, gm, bt, dy, pool=pool, dx_bf16=True)
        dxb, dgb, dbb = ops.layernorm_act_bwd(x16.float(), gm, bt, dy, pool=pool, dx_bf16=True)
        # (bf16 rows are read on 32 lanes per row, fp32 rows on 64: the row sums associate differently -> one bf16 ulp)
        assert dxa.dtype == torch.bfloat16
        assert (dxa.float() - dxb.float()).abs().max().item() <= 8e-3 * dxb.float().abs().max().item()
        with _lib.variant(LN_LPR=64):          # the 64-lane kernels read both storage types the same way
            dxc, _, _ = ops.layernorm_act_bwd(x16, gm, bt, dy, pool=pool, dx_bf16=True)
            a64 = ops.layernorm_act(x16, gm, bt, out_bf16=True)
            b64 = ops.layernorm_act(x16.float(), gm, bt, out_bf16=True)
        assert torch.equal(dxc, dxb) and torch.equal(a64, b64)
        # dgamma / dbeta: fp32 atomics, order not fixed
        assert (dga - dgb).abs().max().item() <= 1e-4 * dgb.abs().max().item()
        assert (dba - dbb).abs().max().item() <= 1e-4 * dbb.abs().max().item()
