"""The two-workgroup recurrent forward at H = 256 (csrc/lstm_rec_h256_pair.hip; the reference's real checkpoint size,
04_lstm_model.py:877; `nn.LSTM` call 04:181-188, 211) against its single-workgroup twin (LOB_VAR_H256_PAIR = 0).

Two CUs share 64 batch rows of a direction, each keeps half of W_hh resident and they exchange their halves of h every
step through a workspace.  Same arithmetic per element; the matrix instruction differs (16x16x32 against 32x32x16: another
summation tree inside one k-block), so outputs agree to fp32 rounding of the pre-activations -- which a bf16 output
may round to the neighbouring value in rare elements.  Every output combination of the entry point is exercised,
the exchange protocol's error word must stay zero, and shapes the pair kernel does not take (Bp % 64 != 0) must fall
back to the twin bit-identically."""
import pytest
import torch

pytestmark = pytest.mark.gpu

H, D = 256, 2


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    from lstm_ode_bci_amd import _lib
    assert _lib.lib().lob_version() >= 203
    return torch.device("cuda:0")


def _rand(shape, dev, seed, scale=1.0, dtype=torch.float32):
    g = torch.Generator(device=dev).manual_seed(seed)
    return (torch.randn(shape, generator=g, device=dev) * scale).to(dtype)


def _inputs(dev, T, Bp, seed=1):
    from lstm_ode_bci_amd import ops
    rows = T * Bp
    x = _rand((rows, 2 * H), dev, seed, 1.0, torch.bfloat16)
    w = _rand((D * 4 * H, 2 * H), dev, seed + 1, 0.04, torch.bfloat16)
    bias = _rand((D * 4 * H,), dev, seed + 2, 0.1)
    whh = _rand((D, 4 * H, H), dev, seed + 3, 0.05)
    P = ops.gate_gemm_x(x, w, bias, T, Bp, H, D, True, mixed=True)
    return P, whh


def _close(a, b, name, atol, frac_exact=0.0):
    a, b = a.float(), b.float()
    err = (a - b).abs().max().item()
    assert err <= atol, (name, err, atol)
    if frac_exact:
        assert (a == b).float().mean().item() >= frac_exact, (name, (a == b).float().mean().item())


def _run(P, whh, T, Bp, pair, **kw):
    from lstm_ode_bci_amd import _lib, ops
    G = P.clone()
    with _lib.variant(H256_PAIR=1 if pair else 0):
        out = ops.lstm_rec_fwd(G, whh, T, Bp, H, D, kw.pop("save", False), mixed=True, **kw)
        ops.pair_check(sync=True)
    return out, G


@pytest.mark.parametrize("T,Bp", [(1, 64), (2, 64), (7, 64), (12, 192), (5, 64 * 9)])
@pytest.mark.parametrize("mode", ["save_bf16", "save_drop", "infer_f32", "infer_bf16", "save_f32y"])
def test_pair_forward_matches_single_workgroup_twin(dev, T, Bp, mode):
    from lstm_ode_bci_amd import ops
    P, whh = _inputs(dev, T, Bp)
    kw = {"save_bf16": dict(save=True, want_f32=False, want_bf16=True),
          "save_drop": dict(save=True, want_f32=False, want_bf16=True, drop_p=0.4, seed=77),
          "infer_f32": dict(save=False, want_f32=True, want_bf16=False),
          "infer_bf16": dict(save=False, want_f32=False, want_bf16=True),
          "save_f32y": dict(save=True, want_f32=True, want_bf16=True)}[mode]
    (Y, Cs, Y16, Yd), G = _run(P, whh, T, Bp, True, **dict(kw))
    (Yr, Csr, Y16r, Ydr), Gr = _run(P, whh, T, Bp, False, **dict(kw))
    assert int(ops._lib.lib().lob_rec_pair_ws_bytes(H, Bp, D)) > 0
    # h is in (-1, 1): one bf16 ulp is at most 2^-8.  The fp32 outputs differ by summation order -- and, from the second
    # step on, by the rare h whose bf16 rounding (the MFMA operand) flips: 2^-9 x |W_hh| per flipped element
    if Y is not None:
        _close(Y, Yr, "Y", 5e-4)
    if Y16 is not None:
        _close(Y16, Y16r, "Y16", 2 ** -8, frac_exact=0.995)
    if Yd is not None:
        _close(Yd, Ydr, "Yd", 2 ** -7, frac_exact=0.995)
        assert torch.equal(Yd == 0, Ydr == 0) or ((Yd == 0) != (Ydr == 0)).float().mean().item() < 1e-4   # same mask
    if kw.get("save"):
        _close(G, Gr, "saved gates", 2 ** -8, frac_exact=0.995)
        _close(Cs, Csr, "cell states", 2e-2 if Cs.dtype == torch.bfloat16 else 1e-3,
               frac_exact=0.99 if Cs.dtype == torch.bfloat16 else 0.0)
    else:
        assert torch.equal(G, P)                    # inference leaves P alone


def test_pair_forward_outputs_are_bounded_and_close_to_the_twin_at_fp32(dev):
    """fp32 outputs (inference, no bf16 rounding of the result): h stays inside (-1, 1) and within the MFMA-shape rounding
    of the single-workgroup kernel."""
    T, Bp = 9, 128
    P, whh = _inputs(dev, T, Bp, seed=11)
    (Y, _, _, _), _ = _run(P, whh, T, Bp, True, save=False, want_f32=True, want_bf16=False)
    (Yr, _, _, _), _ = _run(P, whh, T, Bp, False, save=False, want_f32=True, want_bf16=False)
    assert (Y - Yr).abs().max().item() < 5e-4
    assert torch.isfinite(Y).all() and Y.abs().max().item() <= 1.0


def test_pair_forward_full_size_and_error_word(dev):
    """B = 4096, T = 256 (256 workgroups: every CU holds one side of a pair): twin comparison at the step's real size,
    repeated launches give identical bits, the protocol's error word stays zero."""
    from lstm_ode_bci_amd import ops
    T, Bp = 256, 4096
    P, whh = _inputs(dev, T, Bp, seed=5)
    kw = dict(save=True, want_f32=False, want_bf16=True, drop_p=0.4, seed=3)
    (_, Cs, Y16, Yd), G = _run(P, whh, T, Bp, True, **dict(kw))
    (_, Cs2, Y162, Yd2), G2 = _run(P, whh, T, Bp, True, **dict(kw))
    assert torch.equal(Y16, Y162) and torch.equal(Yd, Yd2) and torch.equal(G, G2) and torch.equal(Cs, Cs2)
    (_, Csr, Y16r, Ydr), Gr = _run(P, whh, T, Bp, False, **dict(kw))
    # 256 steps of a recurrence: rounding differences of single elements feed back; the bulk stays identical
    _close(Y16, Y16r, "Y16", 2 ** -5, frac_exact=0.98)
    _close(G, Gr, "saved gates", 2 ** -5, frac_exact=0.98)
    assert (Y16.float() - Y16r.float()).abs().mean().item() < 2e-5


def test_shapes_outside_the_pair_kernel_fall_back_bit_identically(dev):
    from lstm_ode_bci_amd import ops
    T, Bp = 5, 96                                   # Bp % 64 != 0
    assert int(ops._lib.lib().lob_rec_pair_ws_bytes(H, Bp, D)) == 0
    P, whh = _inputs(dev, T, Bp)
    (Y, _, _, _), _ = _run(P, whh, T, Bp, True, save=False, want_f32=True, want_bf16=False)
    (Yr, _, _, _), _ = _run(P, whh, T, Bp, False, save=False, want_f32=True, want_bf16=False)
    assert torch.equal(Y, Yr)
