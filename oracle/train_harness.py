"""ORACLE -- TEST INFRASTRUCTURE ONLY.  Never imported by the product package.

CPU restatement of the steps either side of fwd+bwd:

* the training harness of 04_lstm_model.py:406-596 -- inverse-frequency class weights and weighted
  cross-entropy (04:430-435), gradient accumulation (04:489-493), global-norm clipping at 1.0 (04:501),
  AdamW (04:438; the algorithm of torch.optim.AdamW written out), linear warm-up + cosine schedule
  stepped once per epoch (04:441-450, 549), the history dict (04:551-558);
* the gradient attribution of 07_explainability.py:203-285 (per-sample input gradient of the
  predicted-class logit, |.| averaged over time, summed over samples, normalised to sum 1).

The forward/backward of the model itself is oracle.torch_cpu_path.TorchCpuModel (the reference's layer
stack).  Pinned against the reference's own train_model / compute_channel_importance by
tests/golden/g6_training.npz and g7_channel_importance.npz (tests/test_oracle_vs_golden.py).
"""
from __future__ import annotations

import math

import numpy as np
import torch


def class_weights(y_train):
    """04:430-432: 1/count per class, normalised to sum 2."""
    counts = np.bincount(np.asarray(y_train))
    w = np.array([1.0 / c for c in counts], dtype=np.float32)
    return w / w.sum() * np.float32(2.0)


def weighted_ce(logits, y, w):
    """nn.CrossEntropyLoss(weight=w), reduction 'mean': sum_i w[y_i] nll_i / sum_i w[y_i]."""
    logp = torch.log_softmax(logits, dim=1)
    wi = w[y]
    return -(wi * logp[torch.arange(len(y)), y]).sum() / wi.sum()


def lr_factor(epoch, warmup_epochs, epochs):
    """04:441-448."""
    if epoch < warmup_epochs:
        return (epoch + 1) / warmup_epochs
    progress = (epoch - warmup_epochs) / (epochs - warmup_epochs)
    return 0.5 * (1 + np.cos(np.pi * progress))


def clip_coef(grads, max_norm=1.0):
    """torch.nn.utils.clip_grad_norm_: min(1, max_norm / (||g||_2 + 1e-6)) over all tensors."""
    total = math.sqrt(sum(float((g.double() ** 2).sum()) for g in grads))
    return min(1.0, max_norm / (total + 1e-6)), total


def adamw_update(p, g, m, v, step, lr, weight_decay, beta1=0.9, beta2=0.999, eps=1e-8):
    """One torch.optim.AdamW step on one tensor, in place (decoupled weight decay first)."""
    p.mul_(1.0 - lr * weight_decay)
    m.mul_(beta1).add_(g, alpha=1.0 - beta1)
    v.mul_(beta2).addcmul_(g, g, value=1.0 - beta2)
    bc1 = 1.0 - beta1 ** step
    bc2 = 1.0 - beta2 ** step
    denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
    p.addcdiv_(m, denom, value=-lr / bc1)


def binary_f1(true, pred):
    """sklearn f1_score(true, pred, zero_division=0) for labels {0,1}, positive class 1."""
    true, pred = np.asarray(true), np.asarray(pred)
    tp = int(((pred == 1) & (true == 1)).sum())
    fp = int(((pred == 1) & (true == 0)).sum())
    fn = int(((pred == 0) & (true == 1)).sum())
    return 0.0 if 2 * tp + fp + fn == 0 else 2 * tp / (2 * tp + fp + fn)


def train_model(model, train_batches, val_batches, y_train, epochs=100, learning_rate=3e-4, patience=15,
                weight_decay=1e-4, warmup_epochs=5, gradient_accumulation_steps=4):
    """04:406-596 on lists of (x, y) torch batches.  Returns the history dict; `model` holds the weights of
    the LAST epoch run (the reference's best-state snapshot is a shallow copy)."""
    w = torch.from_numpy(class_weights(y_train))
    params = [p for p in model.parameters()]
    m = [torch.zeros_like(p) for p in params]
    v = [torch.zeros_like(p) for p in params]
    step = 0
    hist = {k: [] for k in ("train_loss", "val_loss", "train_acc", "val_acc", "val_f1", "learning_rates")}
    best_f1, stale = 0.0, 0
    acc = gradient_accumulation_steps
    for epoch in range(epochs):
        lr = learning_rate * lr_factor(epoch, warmup_epochs, epochs)
        model.train()
        model.zero_grad(set_to_none=True)
        tl, tc, tt = 0.0, 0, 0
        for bi, (x, y) in enumerate(train_batches):
            out = model(x)
            loss = weighted_ce(out, y, w) / acc
            loss.backward()
            if (bi + 1) % acc == 0:
                grads = [p.grad for p in params]
                coef, _ = clip_coef(grads, 1.0)
                step += 1
                with torch.no_grad():
                    for p, g, mm, vv in zip(params, grads, m, v):
                        adamw_update(p, g * coef, mm, vv, step, lr, weight_decay)
                model.zero_grad(set_to_none=True)
            tl += float(loss.detach()) * acc * len(x)
            tc += int((out.argmax(1) == y).sum())
            tt += len(y)
        model.eval()
        vl, vc, vt, vp, vy = 0.0, 0, 0, [], []
        with torch.no_grad():
            for x, y in val_batches:
                out = model(x)
                vl += float(weighted_ce(out, y, w)) * len(x)
                pred = out.argmax(1)
                vc += int((pred == y).sum())
                vt += len(y)
                vp.extend(pred.tolist())
                vy.extend(y.tolist())
        f1 = binary_f1(vy, vp)
        hist["train_loss"].append(tl / tt)
        hist["val_loss"].append(vl / vt)
        hist["train_acc"].append(tc / tt)
        hist["val_acc"].append(vc / vt)
        hist["val_f1"].append(f1)
        hist["learning_rates"].append(learning_rate * lr_factor(epoch + 1, warmup_epochs, epochs))
        if f1 > best_f1:
            best_f1, stale = f1, 0
        else:
            stale += 1
        if stale >= patience:
            break
    return hist


def channel_importance(model, X, batch_size=32):
    """07:232-268 over all of X in the given order (the reference's random subset is a permutation when
    n_samples == len(X); the sum is order-independent).  The reference switches the model to train() for
    this (07:219); with dropout 0 that changes nothing."""
    C = X.shape[2]
    imp = np.zeros(C)
    for s in range(0, len(X), batch_size):
        xb = torch.from_numpy(np.asarray(X[s:s + batch_size], dtype=np.float32)).requires_grad_(True)
        out = model(xb)
        pred = out.argmax(1)
        for i in range(len(xb)):
            model.zero_grad(set_to_none=True)
            if xb.grad is not None:
                xb.grad.zero_()
            out[i, pred[i]].backward(retain_graph=True)
            imp += xb.grad[i].abs().mean(dim=0).numpy()
    imp /= len(X)
    return imp / imp.sum()
