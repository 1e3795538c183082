"""The reference's OWN loop bodies, run on the drop-in model (``-m gpu``).

* 04_lstm_model.py:486-507 -- ``torch.cuda.amp.autocast()`` (fp16 default) + ``GradScaler``: ``scale(loss).backward()``,
  ``unscale_``, ``clip_grad_norm_(1.0)``, ``scaler.step(optim.AdamW)``, ``scaler.update()``, with gradient
  accumulation -- against the oracle's fp32 step at the mixed tolerance;
* 07_explainability.py:219-258 -- the model in ``train()`` (dropout ON), one forward, B ``backward(retain_graph=True)``
  calls: every pass must regenerate the SAME dropout masks;
* 06_lstm_ode_integration.py:340-351 -- ``predict_batch`` entering autocast (``use_amp``), against the reference's g4
  fixtures at the mixed tolerance; ``batch_size`` as an upper bound on request.
"""
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
    return torch.device("cuda:0")


def _model(sd, C, H, dev, dropout=0.4):
    from lstm_ode_bci_amd import EnhancedLSTMModel
    m = EnhancedLSTMModel(input_size=C, hidden_size=H, num_layers=3, num_classes=2, dropout=dropout, bidirectional=True)
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
    return m.to(dev)


def test_reference_amp_training_loop_body(dev):
    """04:482-507 verbatim on the build's model: fp16 autocast + GradScaler + clip + torch.optim.AdamW, accumulation
    over 4 micro-batches.  Checked against the oracle (same layer stack on the CPU, fp32, no AMP): the unscaled,
    clipped gradients per tensor (mixed tolerance: <= 3e-2 of the tensor's largest entry), the scaler must not have
    skipped the step (no inf / nan from the 65536-fold loss scale in the bf16 streams), and the weights must move."""
    from oracle import torch_cpu_path as TP
    C, H, T, B, ACC = 61, 128, 48, 24, 4
    sd = syn.make_state_dict(C, H, 3, 2, True)
    m = _model(sd, C, H, dev, dropout=0.0).train()
    ref = TP.build(sd, C, H, dropout=0.0).train()
    xs, ys = zip(*[syn.make_windows(B, T, C, seed=100 + i) for i in range(ACC)])
    cw = torch.tensor([0.8, 1.2])
    criterion = torch.nn.CrossEntropyLoss(weight=cw.to(dev))
    optimizer = torch.optim.AdamW(m.parameters(), lr=3e-4, weight_decay=1e-4)
    scaler = torch.cuda.amp.GradScaler()
    scale0 = scaler.get_scale()
    before = {k: p.detach().clone() for k, p in m.named_parameters()}
    optimizer.zero_grad()
    for b in range(ACC):
        X_batch, y_batch = torch.from_numpy(xs[b]).to(dev), torch.from_numpy(ys[b]).to(dev)
        with torch.cuda.amp.autocast():
            outputs = m(X_batch)
            loss = criterion(outputs, y_batch) / ACC
        scaler.scale(loss).backward()
    scaler.unscale_(optimizer)
    total = torch.nn.utils.clip_grad_norm_(m.parameters(), max_norm=1.0)
    got = {k: p.grad.detach().cpu().clone() for k, p in m.named_parameters()}
    scaler.step(optimizer)
    scaler.update()
    optimizer.zero_grad()
    assert torch.isfinite(total) and scaler.get_scale() == scale0, "GradScaler skipped the step (inf/nan gradients)"
    # oracle: the same four micro-batches in fp32, then clip
    ref.zero_grad(set_to_none=True)
    rc = torch.nn.CrossEntropyLoss(weight=cw)
    for b in range(ACC):
        (rc(ref(torch.from_numpy(xs[b])), torch.from_numpy(ys[b])) / ACC).backward()
    rtotal = torch.nn.utils.clip_grad_norm_(ref.parameters(), max_norm=1.0)
    assert abs(float(total) - float(rtotal)) <= 3e-2 * float(rtotal)
    worst, wk = 0.0, None
    for k, p in ref.named_parameters():
        sc = p.grad.abs().max().item()
        if sc < 1e-7:
            continue
        e = (got[k] - p.grad).abs().max().item() / sc
        if e > worst:
            worst, wk = e, k
    print(f"reference AMP loop body: worst tensor {wk} rel grad err {worst:.3e}, |g| {float(total):.4f} vs {float(rtotal):.4f}")
    assert worst <= 3e-2, (wk, worst)
    moved = max((p.detach() - before[k]).abs().max().item() for k, p in m.named_parameters())
    assert 1e-5 < moved < 1e-2          # one AdamW step at lr 3e-4


def test_retained_graph_passes_in_train_mode_reuse_the_dropout_masks(dev):
    """07:219-258: lstm_model.train(), one forward, then B backward passes on the retained graph.  The masks are a
    hash of (seed, element): every pass regenerates them from the seed the forward drew, so (i) repeating a pass gives
    bit-identical gradients, (ii) rows of other windows stay exactly zero, (iii) the B per-window passes agree with ONE
    vector-Jacobian pass over the same graph (what attribution.compute_channel_importance launches)."""
    C, H, T, B = 61, 128, 40, 6
    sd = syn.make_state_dict(C, H, 3, 2, True)
    m = _model(sd, C, H, dev).train()                # dropout 0.2 / 0.4 / 0.4 active
    x, _ = syn.make_windows(B, T, C, seed=5)
    X_batch = torch.from_numpy(x).to(dev)
    X_batch.requires_grad = True
    torch.manual_seed(7)
    outputs = m(X_batch)
    pred_class = outputs.argmax(dim=1)
    grads = []
    for i in range(B):
        m.zero_grad()
        if X_batch.grad is not None:
            X_batch.grad.zero_()
        outputs[i, pred_class[i]].backward(retain_graph=True)
        grads.append(X_batch.grad.clone())
    for i in range(B):
        for j in range(B):
            if j != i:
                assert grads[i][j].abs().max().item() == 0.0
        assert grads[i][i].abs().max().item() > 0
    X_batch.grad.zero_()
    outputs[2, pred_class[2]].backward(retain_graph=True)
    assert torch.equal(X_batch.grad, grads[2])                      # same masks on a repeated pass
    X_batch.grad.zero_()
    outputs.gather(1, pred_class[:, None]).sum().backward(retain_graph=True)
    per_window = torch.stack([grads[i][i] for i in range(B)])
    assert (X_batch.grad - per_window).abs().max().item() <= 1e-6 * per_window.abs().max().item() + 1e-12
    # a second forward draws fresh masks: the eval-mode gradient differs from the train-mode one
    m.eval()
    Xe = torch.from_numpy(x).to(dev).requires_grad_(True)
    oe = m(Xe)
    oe[0, pred_class[0]].backward()
    assert (Xe.grad[0] - grads[0][0]).abs().max().item() > 0


def test_predict_batch_use_amp_and_batch_size_cap(dev):
    """06:340-351: the reference's predict_batch enters autocast on a GPU.  use_amp=True runs the mixed path: against
    the reference's g4 fixture (fp32 CPU) at the mixed tolerance (logits are scaled x400 in this fixture: a 1e-4
    logit error is 4e-2 in the logit and up to ~1e-2 in probability), and against the fp32 path.  respect_batch_size
    makes batch_size an upper bound: same numbers, more device passes."""
    from lstm_ode_bci_amd import CognitiveStateODE, LSTMODEIntegration
    d = np.load(os.path.join(GOLDEN, "g4_coupled.npz"))
    sd = syn.make_state_dict(61, 128, 3, 2, True)
    sd["classifier.6.weight"] = sd["classifier.6.weight"] * d["cls6_scale"]
    sd["classifier.6.bias"] = d["cls6_bias"]
    x, _ = syn.make_windows(32, seed=11)
    m = _model(sd, 61, 128, dev).eval()
    integ = LSTMODEIntegration(m, CognitiveStateODE(dict(syn.DEFAULT_RATES)), coupling_strength=0.5)
    t32, p32, y32 = integ.predict_batch(x, forecast_steps=20, batch_size=16, show_progress=False)
    tam, pam, yam = integ.predict_batch(x, forecast_steps=20, batch_size=16, show_progress=False, use_amp=True)
    assert pam.dtype == np.float32 and tam.dtype == np.float64 and yam.dtype == np.int64
    assert not np.array_equal(pam, p32), "use_amp=True must run the mixed path"
    assert np.abs(pam - d["probs_default"]).max() < 5e-2 and np.abs(pam - p32).max() < 5e-2
    stable = (np.abs(d["probs_default"] - 0.6).min(1) > 6e-2) & (np.abs(d["probs_default"] - 0.4).min(1) > 6e-2)
    assert stable.sum() >= 8
    assert np.abs(tam[stable] - d["traj_default"][stable]).max() < 5e-2
    far = np.abs(d["traj_default"][:, -1, 2] - 0.5) > 5e-2
    assert np.array_equal(yam[far & stable], d["pred_default"][far & stable])
    # the class attribute is the object-level switch (the reference's use_amp = torch.cuda.is_available())
    integ.use_amp = True
    _, pam2, _ = integ.predict_batch(x, forecast_steps=20, batch_size=16, show_progress=False)
    integ.use_amp = False
    assert np.array_equal(pam2, pam)
    # batch_size as a true upper bound: 32 windows in passes of 5 -> 7 forward calls, same results
    calls = []
    orig = m.forward

    def counting(xx, return_attention=False):
        calls.append(xx.shape[0])
        return orig(xx, return_attention)
    m.forward = counting
    try:
        tc, pc, yc = integ.predict_batch(x, forecast_steps=20, batch_size=5, show_progress=False, respect_batch_size=True)
        assert calls == [5] * 6 + [2]
        calls.clear()
        integ.predict_batch(x, forecast_steps=20, batch_size=5, show_progress=False)
        assert calls == [32]                                  # default: a lower bound
        calls.clear()
        integ.max_device_chunk = 8
        integ.predict_batch(x, forecast_steps=20, batch_size=512, show_progress=False)
        assert calls == [8] * 4
    finally:
        m.forward = orig
        integ.max_device_chunk = None
    assert np.abs(pc - p32).max() < 1e-6 and np.array_equal(yc, y32) and np.abs(tc - t32).max() < 1e-6


def test_predict_batch_host_pipeline_matches_device_resident(dev):
    """numpy in -> numpy out through the pinned double-buffered upload (several chunks, ragged last one, float64
    input as the .npz holds it, 04:346) == the device-resident path."""
    from lstm_ode_bci_amd import CognitiveStateODE, LSTMODEIntegration
    sd = syn.make_state_dict(61, 128, 3, 2, True)
    x, _ = syn.make_windows(70, 64, 61, seed=3)
    m = _model(sd, 61, 128, dev).eval()
    integ = LSTMODEIntegration(m, CognitiveStateODE(), 0.5)
    td, pd_, yd = integ.predict_batch_device(torch.from_numpy(x).to(dev), forecast_steps=10, batch_size=70)
    for xin in (x, x.astype(np.float64), torch.from_numpy(x)):
        th, ph, yh = integ.predict_batch(xin, forecast_steps=10, batch_size=16, show_progress=False, respect_batch_size=True)
        assert np.abs(ph - pd_.cpu().numpy()).max() < 1e-6
        assert np.abs(th - td.cpu().numpy()).max() < 1e-6 and np.array_equal(yh, yd.cpu().numpy())
