"""The steps either side of fwd+bwd in the reference's training loops, on the device.

* :class:`WeightedCrossEntropy` -- ``nn.CrossEntropyLoss(weight=class_weights)`` (04_lstm_model.py:430-435):
  one launch gives the loss, its gradient and the correct-prediction count.
* :class:`FusedAdamW` -- ``optim.AdamW`` (04:438) over ONE flat parameter / gradient / moment buffer: the
  parameters and their ``.grad`` are re-pointed to views of the flat buffers, so ``clip_grad_norm_`` + ``step`` of
  the reference (04:501-502) are two launches (sum of squares, fused clip + AdamW) with no host round trip, and a
  data-parallel run all-reduces ``flat_grad`` as one 4.5 MB message.
* :func:`warmup_cosine` (04:441-448), :class:`DeviceWindowLoader` (04:336-402 with the windows resident in HBM),
  :func:`train_model` (04:406-596), :func:`quick_train_evaluate` / :func:`run_architecture_ablation`
  (09_sensitivity_analysis.py:265-378).

Mixed precision: the reference trains under ``autocast()`` + ``GradScaler`` (fp16, 04:487-503).  Here the policy is
bf16 MFMA inputs with fp32 accumulation and state: bf16 has the fp32 exponent range, so there is
no loss scaling and no skipped steps.
"""
from __future__ import annotations

import math
import time
import weakref as _weakref

import numpy as np
import torch
import torch.nn as nn

from . import ops
from .model import AblationLSTMModel
from .sharding import all_reduce_flat_grad_, dp_assert_equal, dp_broadcast_, dp_sum_


# ---------------------------------------------------------------------------------------------
# loss
# ---------------------------------------------------------------------------------------------
class _WeightedCEFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, target, weight):
        loss, dl, correct = ops.weighted_ce(logits.detach().float().contiguous(), target.contiguous(), weight,
                                            want_grad=ctx.needs_input_grad[0])
        ctx.dl = dl
        ctx.mark_non_differentiable(correct)
        return loss.reshape(()), correct

    @staticmethod
    def backward(ctx, gloss, _gc):
        return ctx.dl * gloss, None, None


class WeightedCrossEntropy(nn.Module):
    """``criterion(outputs, y)`` of 04_lstm_model.py:435/489 (reduction 'mean' with class weights).  The count of
    correct predictions of the last call (04:508-509) is left on the device in ``last_correct``."""

    def __init__(self, weight=None):
        super().__init__()
        self.register_buffer("weight", None if weight is None else torch.as_tensor(weight, dtype=torch.float32))
        self.last_correct = None

    def forward(self, logits, target):
        if not logits.is_cuda:
            raise ops._lib.LobError("WeightedCrossEntropy: logits must be on the GPU (no CPU fallback)")
        ops.same_device([logits, target, self.weight], "WeightedCrossEntropy (logits, target, class weights)")
        with ops.on_device(logits.device), torch.autocast(device_type="cuda", enabled=False):
            loss, self.last_correct = _WeightedCEFn.apply(logits, target, self.weight)
        return loss


def class_weights_from_labels(y_train, counts=None):
    """04:430-432: inverse class frequency, normalised to sum 2.  ``counts``: class counts obtained elsewhere (the
    data-parallel run sums them over the ranks' shards)."""
    if counts is None:
        counts = np.bincount(np.asarray(y_train))
    w = np.array([1.0 / c for c in counts], dtype=np.float32)
    return w / w.sum() * np.float32(2.0)


def warmup_cosine(current_epoch, warmup_epochs, epochs):
    """``lr_lambda`` of 04:441-448."""
    if current_epoch < warmup_epochs:
        return (current_epoch + 1) / warmup_epochs
    progress = (current_epoch - warmup_epochs) / (epochs - warmup_epochs)
    return 0.5 * (1 + np.cos(np.pi * progress))


# ---------------------------------------------------------------------------------------------
# optimizer
# ---------------------------------------------------------------------------------------------
#: model -> weakref of the FusedAdamW that is its gradient sink.  Kept OUTSIDE the nn.Module (an attribute holding a
#: weakref made the model unpicklable: torch.save(model), pickle, spawn-ed workers)
_GRAD_SINKS = _weakref.WeakKeyDictionary()


def grad_sink_of(model):
    """The live FusedAdamW attached to `model` as its gradient sink, or None."""
    ref = _GRAD_SINKS.get(model)
    return ref() if ref is not None else None


class _GradSink:
    """Accumulation targets of one backward inside FusedAdamW's flat gradient buffer: `offs[i]` = offset of the i-th
    parameter of autograd._collect(model) (None entries: absent sub-modules of the ablation variants)."""

    def __init__(self, flat_grad, offs):
        self.flat_grad, self.offs = flat_grad, offs

    def view(self, i, shape):
        n = 1
        for s in shape:
            n *= int(s)
        return self.flat_grad[self.offs[i]:self.offs[i] + n].view(tuple(shape))

    def span(self, i, step, count, shape):
        """Parameters i, i + step, ..., (i + (count - 1) step) as ONE view: they must lie back to back in the flat buffer
        (FusedAdamW._kernel_order lays the directions of an LSTM tensor out that way; sink_for checks it)."""
        n = 1
        for s in shape:
            n *= int(s)
        per = n // max(int(count), 1)
        for k in range(1, int(count)):
            if self.offs[i + step * k] != self.offs[i] + k * per:
                raise ops._lib.LobError("gradient sink: parameters of one span are not contiguous in the flat buffer")
        return self.view(i, shape)


class FusedAdamW(torch.optim.Optimizer):
    """torch.optim.AdamW semantics (04:438) on flat buffers; a torch Optimizer, so LambdaLR (04:450) drives its
    ``param_groups[0]['lr']``.  Construct it AFTER ``model.to(device)``: it re-points ``p.data`` / ``p.grad``.

    ``model=`` (optional; an EnhancedLSTMModel / AblationLSTMModel whose parameters these are) makes the optimizer the
    model's GRADIENT SINK: the flat buffers are laid out in the order the backward kernels produce gradients (the two
    directions' W_ih / W_hh / biases of a layer side by side), the model's backward accumulates every parameter gradient
    straight into ``flat_grad`` and autograd receives None for the parameters -- no per-parameter accumulation launches.
    ``p.grad`` stays a view of ``flat_grad``, so everything that reads gradients afterwards (clipping, the all-reduce,
    hooks on ``.grad``) sees them; only ``torch.autograd.grad(loss, parameters)`` returns None while a sink is attached
    (``detach_model()`` restores the plain autograd path)."""

    _ALIGN = 64        # floats: every tensor starts on a 256-B boundary of the flat buffers

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, model=None):
        params = list(params)
        if model is not None:
            params = self._kernel_order(model, params)
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        if len(self.param_groups) != 1:
            raise ValueError("FusedAdamW: one parameter group (the reference uses model.parameters())")
        ps = self.param_groups[0]["params"]
        if not ps or not all(p.is_cuda and p.dtype == torch.float32 for p in ps):
            raise ops._lib.LobError("FusedAdamW: fp32 parameters on the GPU (no CPU fallback)")
        dev = ops.same_device(ps, "FusedAdamW (parameters)")
        self._offsets, total = [], 0
        for p in ps:
            self._offsets.append(total)
            total += (p.numel() + self._ALIGN - 1) // self._ALIGN * self._ALIGN
        self.flat_param = torch.zeros(total, device=dev)
        self.flat_grad = torch.zeros(total, device=dev)
        self.exp_avg = torch.zeros(total, device=dev)
        self.exp_avg_sq = torch.zeros(total, device=dev)
        self._normsq = torch.zeros(1, device=dev)
        self._scratch = torch.zeros(512, device=dev)       # partial sums of the gradient norm (fixed summation order)
        self.step_count = 0
        with torch.no_grad():
            for p, off in zip(ps, self._offsets):
                view = self.flat_param[off:off + p.numel()].view_as(p)
                view.copy_(p.data)
                p.data = view
        self._attach_grads()
        self._off_of = {id(p): off for p, off in zip(ps, self._offsets)}
        self._model_ref = None
        if model is not None:
            self._model_ref = _weakref.ref(model)
            _GRAD_SINKS[model] = _weakref.ref(self)

    @staticmethod
    def _kernel_order(model, params):
        """The model's parameters with the two directions of every LSTM tensor side by side (what lob_lstm_dw_bf16 / the
        BPTT bias gradient write as ONE buffer)."""
        ids = {id(p) for p in params}
        order, seen = [], set()

        def put(p):
            if p is not None and id(p) in ids and id(p) not in seen:
                seen.add(id(p))
                order.append(p)
        lstm = getattr(model, "lstm", None)
        for name, p in model.named_parameters():
            if lstm is not None and name.startswith("lstm."):
                continue
            if name.startswith("layer_norm.") and lstm is not None:      # first parameter after the LSTM block
                for layer in range(lstm.num_layers):
                    dirs = lstm.layer_params(layer)
                    for j in range(4):
                        for d in dirs:
                            put(d[j])
            put(p)
        if lstm is not None:                       # ablation variants without a post-LSTM LayerNorm
            for layer in range(lstm.num_layers):
                dirs = lstm.layer_params(layer)
                for j in range(4):
                    for d in dirs:
                        put(d[j])
        for p in params:
            put(p)
        return order

    def detach_model(self):
        m = self._model_ref() if self._model_ref is not None else None
        if m is not None and grad_sink_of(m) is self:
            del _GRAD_SINKS[m]
        self._model_ref = None

    def sink_for(self, params, D=None):
        """A _GradSink for this list of parameters (autograd._collect order), or None when the backward cannot
        accumulate in place: a parameter that is not ours / does not require grad, a .grad that is not our view, or LSTM
        tensors of the two directions that are not back to back."""
        offs = []
        for p in params:
            if p is None:
                offs.append(None)
                continue
            off = self._off_of.get(id(p))
            if off is None or not p.requires_grad:
                return None
            if p.grad is None or p.grad.data_ptr() != self.flat_grad.data_ptr() + 4 * off:
                # gradients were dropped (model.zero_grad(set_to_none=True)) or replaced: a fresh zero slice
                view = self._grad_view(p, off)
                if p.grad is None:
                    view.zero_()
                else:
                    view.copy_(p.grad)
                p.grad = view
            offs.append(off)
        n = len(params)
        # autograd._collect order: 4 projection tensors, 4 * D per LSTM layer, then the tail (2 LayerNorm + 4 attention +
        # 6 classifier positions, None where a sub-module is absent).  A model whose list does not have that shape is not
        # one whose backward can write into the flat buffer: say so instead of silently taking another path
        from .autograd import COLLECT_HEAD, COLLECT_TAIL
        nl = n - COLLECT_HEAD - COLLECT_TAIL
        if D is None or nl <= 0 or nl % (4 * D):
            raise ops._lib.LobError(f"FusedAdamW.sink_for: {n} parameter positions do not fit the model layout "
                                    f"({COLLECT_HEAD} + 4*D*L + {COLLECT_TAIL}, D = {D})")
        for base in range(COLLECT_HEAD, COLLECT_HEAD + nl, 4 * D):
            for j in range(4):
                for d in range(1, D):
                    a, b = offs[base + j], offs[base + j + 4 * d]
                    if a is None or b is None or b != a + d * params[base + j].numel():
                        return None
        return _GradSink(self.flat_grad, offs)

    def _grad_view(self, p, off):
        return self.flat_grad[off:off + p.numel()].view_as(p)

    def _attach_grads(self):
        """Make every p.grad the view of flat_grad.  A foreign gradient tensor (autograd assigns one when p.grad
        was None, e.g. after model.zero_grad()) is copied in; a missing gradient counts as zero."""
        for p, off in zip(self.param_groups[0]["params"], self._offsets):
            view = self._grad_view(p, off)
            if p.grad is None:
                view.zero_()
            elif p.grad.data_ptr() != view.data_ptr():
                view.copy_(p.grad)
            else:
                continue
            p.grad = view

    def zero_grad(self, set_to_none=False):
        """One memset; the gradients stay views of ``flat_grad`` (set_to_none is accepted and ignored)."""
        self.flat_grad.zero_()
        for p, off in zip(self.param_groups[0]["params"], self._offsets):
            if p.grad is None or p.grad.data_ptr() != self.flat_grad.data_ptr() + 4 * off:
                p.grad = self._grad_view(p, off)

    def grad_norm_sq(self):
        self._attach_grads()
        self._normsq.zero_()
        return ops.sumsq(self.flat_grad, self._normsq, self._scratch)

    def clip_grad_norm_(self, max_norm=1.0):
        """torch.nn.utils.clip_grad_norm_(model.parameters(), max_norm) in place (04:501); returns the total norm
        as a device tensor."""
        with ops.on_device(self.flat_grad.device):
            nsq = self.grad_norm_sq()
            total = nsq.sqrt()
            ops.clip_scale_(self.flat_grad, nsq, max_norm)
        return total.reshape(())

    @torch.no_grad()
    def step(self, closure=None, clip_grad_norm=None, grad_scale=1.0):
        """One AdamW step.  ``clip_grad_norm=m`` folds ``clip_grad_norm_(params, m)`` into the same launch (the
        gradients themselves are left unclipped); ``grad_scale`` multiplies the gradient first (e.g. 1/world)."""
        if closure is not None:
            raise ValueError("FusedAdamW.step: closures are not supported")
        g = self.param_groups[0]
        with ops.on_device(self.flat_param.device):
            nsq = self.grad_norm_sq() if clip_grad_norm is not None else None
            if nsq is None:
                self._attach_grads()
            self.step_count += 1
            ops.adamw_(self.flat_param, self.flat_grad, self.exp_avg, self.exp_avg_sq, self.step_count, g["lr"],
                       g["betas"], g["eps"], g["weight_decay"], normsq=nsq,
                       max_norm=clip_grad_norm if clip_grad_norm is not None else 1.0, grad_scale=grad_scale)


# ---------------------------------------------------------------------------------------------
# data: windows resident in HBM
# ---------------------------------------------------------------------------------------------
class DeviceWindowLoader:
    """Batches of ``(X (b,T,C) f32, y (b,) int64)`` gathered on the device from tensors resident in HBM.

    ``sampling='weighted'``: class-balanced sampling with replacement, ``len(y)`` draws per epoch
    (WeightedRandomSampler, 04:358-367); ``'shuffle'``: a fresh permutation per epoch (09:287);
    ``'sequential'``: in order (validation / test loaders, 04:383-396)."""

    def __init__(self, X, y, batch_size, sampling="sequential", device="cuda"):
        # torch.FloatTensor(X) / torch.LongTensor(y) of 04:346-351: the cast happens on the host, so half the bytes
        # of a float64 .npz cross PCIe
        self.X = torch.as_tensor(np.asarray(X, dtype=np.float32)).to(device) if not torch.is_tensor(X) else X.to(device).float()
        self.y = torch.as_tensor(np.asarray(y, dtype=np.int64)).to(device) if not torch.is_tensor(y) else y.to(device).long()
        self.batch_size, self.sampling = int(batch_size), sampling
        if sampling == "weighted":
            counts = torch.bincount(self.y).float()
            self._w = (1.0 / counts)[self.y]
        elif sampling not in ("shuffle", "sequential"):
            raise ValueError(sampling)

    def __len__(self):
        return (len(self.y) + self.batch_size - 1) // self.batch_size

    def __iter__(self):
        n = len(self.y)
        if self.sampling == "weighted":
            order = torch.multinomial(self._w, n, replacement=True)
        elif self.sampling == "shuffle":
            order = torch.randperm(n, device=self.y.device)
        else:
            order = None
        for s in range(0, n, self.batch_size):
            if order is None:
                yield self.X[s:s + self.batch_size], self.y[s:s + self.batch_size]
            else:
                idx = order[s:s + self.batch_size]
                yield self.X.index_select(0, idx), self.y.index_select(0, idx)


def create_dataloaders(X_train, y_train, X_val, y_val, X_test, y_test, batch_size=512, val_batch_size=1024,
                       device="cuda"):
    """04:336-402 with device-resident data: (train_loader, val_loader, test_loader)."""
    return (DeviceWindowLoader(X_train, y_train, batch_size, "weighted", device),
            DeviceWindowLoader(X_val, y_val, val_batch_size, "sequential", device),
            DeviceWindowLoader(X_test, y_test, val_batch_size, "sequential", device))


# ---------------------------------------------------------------------------------------------
# harness
# ---------------------------------------------------------------------------------------------
def binary_f1(true, pred):
    """sklearn.metrics.f1_score(true, pred, zero_division=0) for labels {0, 1} (04:547)."""
    true, pred = np.asarray(true), np.asarray(pred)
    tp = int(((pred == 1) & (true == 1)).sum())
    fp = int(((pred == 1) & (true == 0)).sum())
    fn = int(((pred == 0) & (true == 1)).sum())
    return f1_from_counts(tp, fp, fn)


def f1_from_counts(tp, fp, fn):
    return 0.0 if 2 * tp + fp + fn == 0 else 2 * tp / (2 * tp + fp + fn)


def _to_dev(t, dev):
    return t.to(dev, non_blocking=True) if t.device != dev else t


def train_model(model, train_loader, val_loader, y_train, epochs=100, learning_rate=3e-4, patience=15,
                weight_decay=1e-4, warmup_epochs=5, gradient_accumulation_steps=4, use_amp=True, verbose=True,
                data_parallel=False, process_group=None):
    """``train_model`` of 04_lstm_model.py:406-596: same arguments, same ``(model, history)`` result, same
    arithmetic -- weighted CE / accumulation steps, clip at 1.0, AdamW, warm-up + cosine stepped per epoch,
    early stopping on validation F1.  ``use_amp`` selects the mixed path (bf16 autocast, no GradScaler).

    As in the reference the "best model" snapshot is a shallow ``state_dict().copy()`` (04:576), i.e. the model
    that is returned carries the weights of the last epoch that ran.

    ``data_parallel=True`` (opt-in; the reference is single-process): one process per GPU, every rank passes ITS shard
    of the data in ``train_loader`` / ``val_loader``.  All ranks then run the same number of optimizer steps (checked
    up front: unequal batch counts raise on every rank), the flat gradient is all-reduced once per optimizer step, and
    the epoch metrics -- including the validation F1 that drives early stopping -- are summed over the ranks, so every
    rank takes the same stop decision and holds the same weights and history.  Without the flag no collective is
    issued, whether or not a process group exists."""
    dev = next(model.parameters()).device
    start_time = time.time()
    dp = bool(data_parallel)
    if dp:
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized()):
            raise RuntimeError("train_model(data_parallel=True) needs an initialised torch.distributed process group")
        dp_assert_equal((len(train_loader), len(val_loader), int(gradient_accumulation_steps), int(epochs)),
                        process_group, "batches per epoch (train, val), accumulation steps, epochs")

    counts = None
    if dp:
        # the CE weights of 04:430-432 come from the class counts of the WHOLE training set: sum the shards' counts
        n_cls = int(model.classifier[-1].out_features)
        counts = torch.as_tensor(np.bincount(np.asarray(y_train), minlength=n_cls)[:n_cls].astype(np.float64), device=dev)
        counts = dp_sum_(counts, process_group).cpu().numpy()
    criterion = WeightedCrossEntropy(class_weights_from_labels(y_train, counts)).to(dev)
    optimizer = FusedAdamW(model.parameters(), lr=learning_rate, weight_decay=weight_decay, model=model)
    if dp:
        # every rank starts from rank 0's weights (ranks built with different seeds / checkpoints would otherwise
        # all-reduce gradients onto different weights and drift apart without any error); the moments start at zero
        dp_broadcast_(optimizer.flat_param, 0, process_group)
    scheduler = torch.optim.lr_scheduler.LambdaLR(optimizer, lambda e: warmup_cosine(e, warmup_epochs, epochs))
    acc_steps = gradient_accumulation_steps
    best_val_f1, best_state, stale = 0, None, 0
    history = {k: [] for k in ("train_loss", "val_loss", "train_acc", "val_acc", "val_f1", "learning_rates")}

    def amp():
        return torch.autocast(device_type="cuda", dtype=torch.bfloat16, enabled=bool(use_amp))

    for epoch in range(epochs):
        epoch_start = time.time()
        model.train()
        loss_sum = torch.zeros((), device=dev)
        correct = torch.zeros((), device=dev, dtype=torch.int64)
        total = 0
        optimizer.zero_grad()
        for batch_idx, (xb, yb) in enumerate(train_loader):
            xb, yb = _to_dev(xb, dev), _to_dev(yb, dev)
            with amp():
                outputs = model(xb)
                loss = criterion(outputs, yb) / acc_steps
            loss.backward()
            if (batch_idx + 1) % acc_steps == 0:
                # data-parallel: one all-reduce of the flat gradient; the mean over ranks is taken inside the
                # optimizer launch (every rank's loss is the mean over ITS batch, as DistributedDataParallel does)
                gscale = 1.0
                if dp:
                    _, gscale = all_reduce_flat_grad_(optimizer.flat_grad, process_group)
                optimizer.step(clip_grad_norm=1.0, grad_scale=gscale)     # clip_grad_norm_(…, 1.0) + optimizer.step(), 04:501-502
                optimizer.zero_grad()
            loss_sum += loss.detach() * (acc_steps * xb.size(0))
            correct += criterion.last_correct[0]
            total += yb.size(0)
        tm = torch.stack([loss_sum.double(), correct.double(), torch.tensor(float(total), device=dev, dtype=torch.float64)])
        if dp:
            dp_sum_(tm, process_group)
        tm = tm.tolist()
        train_loss = tm[0] / tm[2]
        train_acc = int(tm[1]) / int(tm[2])

        model.eval()
        vloss = torch.zeros((), device=dev)
        vcorrect = torch.zeros((), device=dev, dtype=torch.int64)
        vtotal, vpreds, vtrue = 0, [], []
        with torch.no_grad():
            for xb, yb in val_loader:
                xb, yb = _to_dev(xb, dev), _to_dev(yb, dev)
                with amp():
                    outputs = model(xb)
                    loss = criterion(outputs, yb)
                vloss += loss * xb.size(0)
                vcorrect += criterion.last_correct[0]
                vtotal += yb.size(0)
                vpreds.append(outputs.argmax(1))
                vtrue.append(yb)
        vt, vp = torch.cat(vtrue), torch.cat(vpreds)
        vm = torch.stack([vloss.double(), vcorrect.double(), torch.tensor(float(vtotal), device=dev, dtype=torch.float64),
                          ((vp == 1) & (vt == 1)).sum().double(), ((vp == 1) & (vt == 0)).sum().double(),
                          ((vp == 0) & (vt == 1)).sum().double()])
        if dp:                      # the stop decision below must be the same on every rank
            dp_sum_(vm, process_group)
        vm = vm.tolist()
        val_loss = vm[0] / vm[2]
        val_acc = int(vm[1]) / int(vm[2])
        val_f1 = f1_from_counts(int(vm[3]), int(vm[4]), int(vm[5]))

        scheduler.step()
        current_lr = optimizer.param_groups[0]["lr"]
        for k, v in (("train_loss", train_loss), ("val_loss", val_loss), ("train_acc", train_acc),
                     ("val_acc", val_acc), ("val_f1", val_f1), ("learning_rates", current_lr)):
            history[k].append(v)
        if verbose and ((epoch + 1) % 5 == 0 or epoch == 0 or epoch == warmup_epochs - 1):
            print(f"Epoch [{epoch + 1:3d}/{epochs}] | Loss: {train_loss:.4f}/{val_loss:.4f} | "
                  f"Acc: {train_acc:.4f}/{val_acc:.4f} | F1: {val_f1:.4f} | LR: {current_lr:.2e} | "
                  f"Time: {time.time() - epoch_start:.1f}s", flush=True)
        if val_f1 > best_val_f1:
            best_val_f1, stale = val_f1, 0
            best_state = model.state_dict().copy()
        else:
            stale += 1
        if stale >= patience:
            if verbose:
                print(f"\nEarly stopping at epoch {epoch + 1} (no improvement for {patience} epochs)", flush=True)
            break
    if best_state is not None:
        model.load_state_dict(best_state)
    if verbose:
        print(f"\nTraining completed in {(time.time() - start_time) / 60:.1f} minutes")
        print(f"Best validation F1: {best_val_f1:.4f}", flush=True)
    return model, history


def _mcc(true, pred):
    true, pred = np.asarray(true), np.asarray(pred)
    tp = float(((pred == 1) & (true == 1)).sum()); tn = float(((pred == 0) & (true == 0)).sum())
    fp = float(((pred == 1) & (true == 0)).sum()); fn = float(((pred == 0) & (true == 1)).sum())
    den = math.sqrt((tp + fp) * (tp + fn) * (tn + fp) * (tn + fn))
    return 0.0 if den == 0 else (tp * tn - fp * fn) / den


def quick_train_evaluate(model, X_train, y_train, X_val, y_val, X_test, y_test, epochs=10, batch_size=512,
                         lr=0.001, device="cuda"):
    """09_sensitivity_analysis.py:265-327: a short un-weighted-CE / AdamW(lr) run on (at most) 20,000 random
    training windows, then test metrics ``({'accuracy','f1','mcc'}, predictions)``."""
    model = model.to(device)
    n_train = min(len(X_train), 20000)
    indices = np.random.choice(len(X_train), n_train, replace=False)
    train_loader = DeviceWindowLoader(np.asarray(X_train)[indices], np.asarray(y_train)[indices], batch_size,
                                      "shuffle", device)
    test_loader = DeviceWindowLoader(X_test, y_test, batch_size * 2, "sequential", device)
    criterion = WeightedCrossEntropy().to(device)
    optimizer = FusedAdamW(model.parameters(), lr=lr, model=model)
    model.train()
    for _ in range(epochs):
        for xb, yb in train_loader:
            optimizer.zero_grad()
            loss = criterion(model(xb), yb)
            loss.backward()
            optimizer.step()
    model.eval()
    preds = []
    with torch.no_grad():
        for xb, _ in test_loader:
            preds.append(model(xb).argmax(dim=1))
    preds = torch.cat(preds).cpu().numpy()
    labels = np.asarray(y_test)
    metrics = {"accuracy": float((preds == labels).mean()), "f1": binary_f1(labels, preds), "mcc": _mcc(labels, preds)}
    return metrics, preds


ABLATION_CONFIGS = [        # 09:340-347
    {"name": "Full Model", "bidirectional": True, "use_attention": True, "num_layers": 3},
    {"name": "No Attention", "bidirectional": True, "use_attention": False, "num_layers": 3},
    {"name": "Unidirectional", "bidirectional": False, "use_attention": True, "num_layers": 3},
    {"name": "1 Layer", "bidirectional": True, "use_attention": True, "num_layers": 1},
    {"name": "2 Layers", "bidirectional": True, "use_attention": True, "num_layers": 2},
    {"name": "Minimal", "bidirectional": False, "use_attention": False, "num_layers": 1},
]


def run_architecture_ablation(X_train, y_train, X_val, y_val, X_test, y_test, hidden_size=256, epochs=10,
                              batch_size=512, device="cuda"):
    """09:330-378: ``(results, predictions)`` keyed by configuration name."""
    input_size = np.asarray(X_train).shape[2]
    results, predictions = {}, {}
    for config in ABLATION_CONFIGS:
        model = AblationLSTMModel(input_size=input_size, hidden_size=hidden_size, num_layers=config["num_layers"],
                                  num_classes=2, dropout=0.4, bidirectional=config["bidirectional"],
                                  use_attention=config["use_attention"])
        metrics, preds = quick_train_evaluate(model, X_train, y_train, X_val, y_val, X_test, y_test, epochs=epochs,
                                              batch_size=batch_size, device=device)
        results[config["name"]] = {"config": {k: v for k, v in config.items() if k != "name"}, "metrics": metrics}
        predictions[config["name"]] = preds
    return results, predictions
