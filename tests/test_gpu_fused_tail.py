"""The fused tail of the mixed forward at H = 128 (lob_attn_scores_bf16: post-LSTM LayerNorm 04_lstm_model.py:192 + the
attention's score layer 04:123-125 in one launch, scores handed to the pooling kernel) against the three kernels it
replaces: same lane assignment in the LayerNorm, same matrix instruction and k order in the score layer, same reduction
order of the scores -> v, u, the attention weights and the context must be BIT-IDENTICAL, also on row counts that do
not fill the 128-row tile and with padded batch rows; and the model's outputs / gradients must not move."""
import numpy as np
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
def test_fused_tail_is_bit_identical_to_the_three_kernels(dev, T, B, save):
    from lstm_ode_bci_amd import ops
    H, D = 128, 2
    Bp = ops.ceil32(B)
    g = torch.Generator(device=dev).manual_seed(T * 100 + B)
    y16 = (torch.randn((T * Bp, 256), generator=g, device=dev) * 0.7).to(torch.bfloat16)
    gam = torch.rand((256,), generator=g, device=dev) + 0.5
    bet = torch.randn((256,), generator=g, device=dev) * 0.1
    w1 = torch.randn((128, 256), generator=g, device=dev) * 0.08
    b1 = torch.randn((128,), generator=g, device=dev) * 0.1
    w2 = torch.randn((128,), generator=g, device=dev) * 0.3
    b2 = torch.randn((1,), generator=g, device=dev)
    w1_16 = w1.to(torch.bfloat16)
    assert ops.attn_scores_ok(y16, H, D, Bp, w1)
    v, u, S = ops.attn_scores(y16, gam, bet, w1_16, b1, w2, b2, T, B, Bp, H, D, save=save)
    ctx, attn = ops.attn_pool_fwd_scores(v, S, T, B, Bp)
    vr = ops.layernorm_act(y16, gam, bet, out_bf16=True)
    ur = ops.gemm_nt(vr, w1_16 if ops.dma_ok(256, 128, T * Bp) else w1, b1, act=ops.ACT_TANH, mixed=True)
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
    assert (attn.double() - a64).abs().max().item() < 2e-6


def test_model_outputs_do_not_change_with_the_fused_tail(dev):
    from lstm_ode_bci_amd import EnhancedLSTMModel, ops
    sd = {k: torch.from_numpy(v) for k, v in syn.make_state_dict(61, 128, 3, 2, True).items()}
    x, _ = syn.make_windows(24, 64, 61, seed=4)
    xt = torch.from_numpy(x).to(dev)

    def run(fused, train):
        m = EnhancedLSTMModel(61, 128, 3, 2, 0.4, True).to(dev)
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
