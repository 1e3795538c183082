"""Drop-in ``LSTMODEIntegration`` (reference: 06_lstm_ode_integration.py:183-406).

Same constructor and method signatures and return types.  The reference runs the LSTM in
chunks of ``batch_size`` on the device and then a per-sample Python loop of ``odeint`` solves
on the host (06:372-401); here both stages run on the GPU: logits -> softmax -> rate
modulation -> initial-state rule -> RK4 -> clip/renormalise -> decision rule are one batched
kernel launch over all windows.
"""
from __future__ import annotations

import ctypes
import os
import threading
import weakref

import numpy as np
import torch

from . import ops
from .synthetic import RATE_KEYS


_COPY_THREADS = max(4, min(16, (os.cpu_count() or 8)))
_pool = None
#: tools/api_probe.py sets this to a list; predict_batch then appends (label, time.perf_counter()) at its phase boundaries
_probe = None


def _stamp(label):
    if _probe is not None:
        import time
        _probe.append((label, time.perf_counter()))


class _ResultPool:
    """Pages of returned trajectory arrays that the caller has let go of, kept for the next call.

    `predict_batch` returns fresh host arrays like the reference (06:396-406); at 300 points the trajectories are 7.2 KB per
    window, i.e. 88 MB per 3 x 4096 windows.  As plain `np.empty` that is 22 k page faults on first touch in the call and an
    munmap of as many pages when the caller drops the result: 0.6 ms per chunk + ~4 ms per call, a tenth of a call that the GPU
    spends idle (tools/api_probe.py).  Here the memory of a result returns to a small pool when -- and only when -- the last
    array referring to it is gone: the returned array is a view (`np.frombuffer`) of a ctypes owner object laid over the pooled
    storage; every slice / reshape the caller takes has that same base chain, so the owner dies only with the LAST view, and
    its finalizer puts the storage back.  Arrays from different calls never share memory while any of them is alive."""
    MIN_BYTES = 4 << 20          # smaller results: plain np.empty
    MAX_HELD = 1 << 30           # bytes parked in the pool at most

    def __init__(self):
        self._free = {}
        self._held = 0
        self._lock = threading.Lock()

    def empty(self, shape, dtype):
        dtype = np.dtype(dtype)
        nbytes = int(np.prod(shape)) * dtype.itemsize
        if nbytes < self.MIN_BYTES:
            return np.empty(shape, dtype)
        with self._lock:
            lst = self._free.get(nbytes)
            store = lst.pop() if lst else None
            if store is not None:
                self._held -= nbytes
        if store is None:
            store = np.empty(nbytes, np.uint8)
        owner = (ctypes.c_char * nbytes).from_buffer(store)
        weakref.finalize(owner, self._give_back, store)
        return np.frombuffer(owner, dtype=dtype).reshape(shape)

    def _give_back(self, store):
        with self._lock:
            if self._held + store.nbytes <= self.MAX_HELD:
                self._free.setdefault(store.nbytes, []).append(store)
                self._held += store.nbytes


_RESULTS = _ResultPool()


def _copy_pool():
    global _pool
    if _pool is None:
        from concurrent.futures import ThreadPoolExecutor
        _pool = ThreadPoolExecutor(max_workers=_COPY_THREADS, thread_name_prefix="lob-h2d")
    return _pool


def _chunk_schedule(n, chunk, ramp):
    """Sizes of the device chunks of an n-window host array: [ramp, 2 ramp, 4 ramp, ...] up to `chunk`, full chunks, and
    the remainder split largest-first down to `ramp` (ramp <= 0 or a single chunk: equal chunks, as before round 4)."""
    if n <= 0:
        return []
    if ramp <= 0 or n <= chunk or ramp >= chunk:
        return [min(chunk, n - i) for i in range(0, n, chunk)]
    head, sz, left = [], int(ramp), n
    while sz < chunk and left > 2 * sz:          # keep at least one full-size pass' worth for the body
        head.append(sz)
        left -= sz
        sz *= 2
    body = [chunk] * (left // chunk)
    left -= chunk * len(body)
    if body and left and left < ramp:            # a sliver: give it to the last full chunk's successor as is
        tail = [left]
    else:
        tail, sz = [], chunk // 2
        while left > 0:
            while sz > left and sz > ramp:
                sz //= 2
            take = min(left, max(sz, 1)) if left >= ramp else left
            tail.append(take)
            left -= take
    out = head + body + tail
    if len(out) > 1 and out[-1] < ramp and out[-2] + out[-1] <= chunk:      # no sliver pass at the end
        out[-2:] = [out[-2] + out[-1]]
    return out


class LSTMODEIntegration:
    #: windows per device pass.  By default ``predict_batch``'s ``batch_size`` (the reference's 512 "fits 24 GB"
    #: value, 06:308) is only a LOWER bound: an MI355X pass of fewer than 4096 windows leaves CUs idle, and results
    #: do not depend on the chunking (every window is independent).  ``respect_batch_size=True`` (per call) or
    #: ``max_device_chunk = n`` (per object) make the caller's number a true upper bound -- a memory cap.
    min_device_chunk = 4096
    max_device_chunk = None
    #: the reference enters ``autocast()`` whenever CUDA is available (06:340, 349): ITS GPU runs are mixed precision,
    #: its CPU runs fp32.  ``use_amp=True`` (per call, or this attribute) selects the mixed path (bf16 MFMA inputs, fp32
    #: accumulate / state; 2.3x the window rate), ``False`` the fp32 path that matches the reference's CPU results to
    #: <= 1e-5.  Left at ``None`` the calls run fp32 and say so ONCE (``warnings.warn``), naming the switch -- a
    #: maintainer who only swaps the import should learn that the reference's own GPU setting is the other one.
    use_amp = None
    #: first device chunk of a multi-chunk host-array call (``predict_batch`` / ``predict_batch_device`` with numpy input):
    #: the chunks grow ramp, 2 ramp, ... up to the device chunk and shrink again at the end, so that the pipeline's fill
    #: (first upload) and drain (last download) are those of a small chunk.  0 = equal chunks.
    ramp_chunk = 1024
    #: windows per staging piece of an upload (page-locked copy of piece j+1 overlaps the H2D of piece j)
    stage_piece = 1024
    _warned_fp32_default = False

    def __init__(self, lstm_model, ode_model, coupling_strength=0.5):
        self.lstm_model = lstm_model
        self.ode_model = ode_model
        self.coupling_strength = coupling_strength
        self.base_params = ode_model.params.copy()

    # ---------------------------------------------------------------------------------
    def _device(self):
        return next(self.lstm_model.parameters()).device

    def _resolve_amp(self, use_amp=None):
        """Precision of one call: the call's argument, else the object's / class's ``use_amp``, else fp32 with a one-time
        note (see the class attribute).  Resolved per call -- no mutable per-object state, so the single-window and the
        batch entry points agree and concurrent callers cannot see each other's choice."""
        if use_amp is not None:
            return bool(use_amp)
        if self.use_amp is not None:
            return bool(self.use_amp)
        if not LSTMODEIntegration._warned_fp32_default:
            LSTMODEIntegration._warned_fp32_default = True
            import warnings
            warnings.warn("LSTMODEIntegration runs the fp32 parity path (matches the reference's CPU results to 1e-5). "
                          "The reference's own GPU runs use autocast (06_lstm_ode_integration.py:340): pass use_amp=True "
                          "or set integration.use_amp = True for the mixed path (about 2.3x the window rate), or "
                          "use_amp=False to keep fp32 without this note.", stacklevel=3)
        return False

    def _probs_device(self, X, amp=False):
        """(probs, attention) device tensors for a batch-first (B,T,C) array/tensor; amp: run the model under autocast."""
        if isinstance(X, np.ndarray):
            X = torch.from_numpy(np.ascontiguousarray(X, dtype=np.float32))
        dev = self._device()
        X = X.to(dev, dtype=torch.float32)
        with ops.on_device(dev), torch.autocast("cuda", dtype=torch.bfloat16, enabled=bool(amp)):
            logits, attention = self.lstm_model(X, return_attention=True)
            return ops.softmax_rows(logits.float().contiguous()), attention

    def release_staging(self):
        """Free the page-locked host buffers, device double buffers and side streams that ``predict_batch`` keeps
        between calls (2 x 256 MB pinned + 2 x 256 MB of HBM at 4096 x 256 x 61 windows, plus the result staging)."""
        for a in ("_h2d_key", "_h2d_stream", "_h2d_stage", "_h2d_dbuf", "_d2h_key", "_d2h_stream", "_d2h_stage"):
            if hasattr(self, a):
                delattr(self, a)

    def _chunk(self, batch_size, respect_batch_size):
        chunk = int(batch_size)
        if chunk <= 0:
            raise ValueError(f"batch_size must be positive, got {batch_size}")
        if not respect_batch_size:
            chunk = max(chunk, self.min_device_chunk)
        if self.max_device_chunk is not None:
            chunk = min(chunk, int(self.max_device_chunk))
        return chunk

    def get_lstm_probabilities(self, X, use_amp=None):
        """(probs (B,2) [P(open), P(closed)], attention (B,T)) as numpy (06:216-234)."""
        self.lstm_model.eval()
        amp = self._resolve_amp(use_amp)
        with torch.no_grad(), ops.on_device(self._device()):
            probs, attention = self._probs_device(X, amp)
        return probs.cpu().numpy(), attention.cpu().numpy()

    def modulate_ode_rates(self, p_closed, p_open):
        """Host-side scalar version of the rate modulation (06:236-264)."""
        alpha = self.coupling_strength
        params = self.base_params.copy()
        params["k_af"] = params["k_af"] * (1 + alpha * p_closed)
        params["k_pf"] = params["k_pf"] * (1 + alpha * p_closed)
        params["k_fa"] = params["k_fa"] * (1 + alpha * p_open)
        params["k_pa"] = params["k_pa"] * (1 + alpha * p_open)
        for k in params:
            params[k] = max(0.001, params[k])
        return params

    def _base_rates(self):
        return [float(self.base_params[k]) for k in RATE_KEYS]

    def _substeps(self):
        return getattr(self.ode_model, "rk4_substeps", 16)

    # ---------------------------------------------------------------------------------
    def predict_trajectory(self, X, initial_state=None, forecast_steps=10, use_amp=None):
        """(trajectory (steps,3), probs (1,2), attention (1,T)) for one window (06:266-306)."""
        self.lstm_model.eval()
        amp = self._resolve_amp(use_amp)
        with torch.no_grad(), ops.on_device(self._device()):
            probs, attention = self._probs_device(X, amp)
            if initial_state is None:
                traj, _, _ = ops.ode_rk4(self._base_rates(), forecast_steps, 0.0, float(forecast_steps),
                                         self._substeps(), probs=probs[:1].contiguous(),
                                         alpha=self.coupling_strength, want_pred=False)
            else:
                pr = probs[0].cpu().numpy()
                mod = self.modulate_ode_rates(pr[1], pr[0])
                y0 = torch.as_tensor(np.asarray(initial_state, np.float64).reshape(1, 3), device=probs.device)
                traj, _, _ = ops.ode_rk4([float(mod[k]) for k in RATE_KEYS], forecast_steps, 0.0,
                                         float(forecast_steps), self._substeps(), y0=y0, want_pred=False)
        self.ode_model.params = self.base_params.copy()
        return traj[0].cpu().numpy(), probs.cpu().numpy(), attention.cpu().numpy()

    def predict_batch_device(self, X_batch, forecast_steps=20, batch_size=512, want_traj=True, use_amp=None,
                             respect_batch_size=False):
        """Device-resident result tensors: (traj (N,steps,3) f64 | None, probs (N,2) f32, pred (N,) i64).
        ``X_batch``: numpy array (host; uploaded chunk by chunk through pinned staging buffers, the copy of chunk
        i+1 overlapping the LSTM pass of chunk i on a side stream) or a device tensor."""
        n = len(X_batch)
        chunk = self._chunk(batch_size, respect_batch_size)
        self.lstm_model.eval()
        dev = self._device()
        amp = self._resolve_amp(use_amp)
        with torch.no_grad(), ops.on_device(dev):
            probs_all = []
            for Xc in self._device_chunks(X_batch, n, chunk, dev):
                probs, _ = self._probs_device(Xc, amp)
                probs_all.append(probs)
            probs = torch.cat(probs_all, 0) if len(probs_all) > 1 else probs_all[0]
            traj, _, pred = ops.ode_rk4(self._base_rates(), forecast_steps, 0.0, float(forecast_steps),
                                        self._substeps(), probs=probs, alpha=self.coupling_strength,
                                        want_traj=want_traj, want_pred=True)
        return traj, probs, pred

    def _device_chunks(self, X_batch, n, chunk, dev):
        """Chunks of X_batch as fp32 device tensors.  Host arrays go through two pinned staging buffers and a copy
        stream: the H2D copy of chunk i+1 runs while chunk i is in the LSTM kernels (06:346 does a synchronous
        ``torch.FloatTensor(X_batch[i:batch_end]).to(DEVICE)`` per chunk)."""
        if torch.is_tensor(X_batch) and X_batch.is_cuda:
            for i in range(0, n, chunk):
                yield X_batch[i:i + chunk]
            return
        if torch.is_tensor(X_batch):
            X_batch = X_batch.numpy()
        shape = tuple(X_batch.shape[1:])
        nb = min(chunk, n)
        main = torch.cuda.current_stream(dev)
        # page-locked staging buffers and the copy stream are kept on the object: pinning 2 x 256 MB costs more than
        # moving it (hipHostMalloc + first touch), and predict_batch is called once per evaluation pass (06:461, 537)
        key = (nb,) + shape + (str(dev),)
        if getattr(self, "_h2d_key", None) != key:
            self._h2d_key = key
            # high priority: HIP maps streams onto a few hardware queues, and a copy stream that lands on the compute
            # stream's queue has its copies dispatched BEHIND the kernels already queued there (seen as the GPU idling for
            # one H2D per chunk, tools/api_probe.py); priority streams get queues of their own
            self._h2d_stream = torch.cuda.Stream(device=dev, priority=-1)
            self._h2d_stage = [torch.empty((nb,) + shape, dtype=torch.float32).pin_memory() for _ in range(2)]
            self._h2d_dbuf = [torch.empty((nb,) + shape, dtype=torch.float32, device=dev) for _ in range(2)]
        copy_stream, stage, dbuf = self._h2d_stream, self._h2d_stage, self._h2d_dbuf
        torch.cuda.current_stream(dev).synchronize()      # a previous call's kernels may still read dbuf
        done = [None, None]          # events: device buffer b may be overwritten (its consumer kernels finished)
        # Chunk schedule.  With several chunks per call what the caller waits for beyond the kernels is the pipeline's FILL
        # (staging + H2D of the first chunk: 2.2 + 4.4 ms for 4096 windows, tools/api_probe.py) and its DRAIN (download +
        # host copy of the last chunk's results): the first chunks are therefore small and double in size -- each one's
        # upload (1.6 us per window) fits under its predecessor's kernels (2.7-3.1 us per window) -- and what remains after
        # the last full chunk is split the same way in reverse.  Every window's result is independent of the chunking.
        sizes = _chunk_schedule(n, chunk, self.ramp_chunk)
        starts = [0]
        for sz in sizes[:-1]:
            starts.append(starts[-1] + sz)

        pool = _copy_pool()
        piece = max(int(self.stage_piece), 1)      # staging granule: the H2D of a piece runs while the next piece is staged

        def upload(k):
            b, i = k & 1, starts[k]
            m = sizes[k]
            # host-side cast + copy into pinned memory (float64 .npz arrays: half the PCIe bytes, 04:346), sliced
            # over a few threads (numpy releases the GIL in copyto; one thread moves only ~5 GB/s)
            dst = stage[b].numpy()
            with torch.cuda.stream(copy_stream):
                if done[b] is not None:
                    copy_stream.wait_event(done[b])
            for p0 in range(0, m, piece):
                pm = min(piece, m - p0)
                step = max(1, (pm + _COPY_THREADS - 1) // _COPY_THREADS)
                list(pool.map(lambda s: np.copyto(dst[p0 + s:p0 + min(pm, s + step)],
                                                  X_batch[i + p0 + s:i + p0 + min(pm, s + step)], casting="same_kind"),
                              range(0, pm, step)))
                with torch.cuda.stream(copy_stream):
                    dbuf[b][p0:p0 + pm].copy_(stage[b][p0:p0 + pm], non_blocking=True)
            with torch.cuda.stream(copy_stream):
                ev = torch.cuda.Event()
                ev.record(copy_stream)
            return ev, m
        ev, m = upload(0)
        prev_ev = None
        for k in range(len(starts)):
            main.wait_event(ev)
            b = k & 1
            yield dbuf[b][:m]            # the consumer enqueues chunk k's kernels BEFORE the host stages chunk k+1
            done[b] = torch.cuda.Event()
            done[b].record(main)
            if k + 1 < len(starts):
                if prev_ev is not None:
                    # staging buffer (k+1)&1 was last read by the copy of chunk k-1: make sure that copy has finished
                    prev_ev.synchronize()
                prev_ev = ev
                ev, m = upload(k + 1)    # host-side staging + H2D of chunk k+1 overlap chunk k's kernels

    def predict_batch(self, X_batch, forecast_steps=20, batch_size=512, show_progress=True, use_amp=None,
                      respect_batch_size=False):
        """(trajectories (N,steps,3) f64, probs (N,2) f32, predictions (N,) int64) as numpy
        (06:308-406).  ``show_progress`` is accepted and ignored (no per-sample loop to show).  ``use_amp`` mirrors
        the reference's flag of 06:340 (see the class attribute); ``respect_batch_size`` makes ``batch_size`` an
        upper bound on the windows per device pass.

        Host arrays in, host arrays out: per device chunk the LSTM pass, the ODE kernel and the download of that chunk's
        results are queued back to back; the download runs on a side stream into page-locked staging buffers while the
        NEXT chunk is in the LSTM kernels, and the host moves the previous chunk's staged results into the returned
        arrays meanwhile (the trajectories are the larger message: 7.2 KB per window at 300 points)."""
        n = len(X_batch)
        if n == 0 or (torch.is_tensor(X_batch) and X_batch.is_cuda):
            traj, probs, pred = self.predict_batch_device(X_batch, forecast_steps, batch_size, use_amp=use_amp,
                                                          respect_batch_size=respect_batch_size)
            self.ode_model.params = self.base_params.copy()
            return traj.cpu().numpy(), probs.cpu().numpy(), pred.cpu().numpy()
        _stamp("enter")
        chunk = self._chunk(batch_size, respect_batch_size)
        self.lstm_model.eval()
        dev = self._device()
        steps = int(forecast_steps)
        traj = _RESULTS.empty((n, steps, 3), np.float64)
        probs = np.empty((n, 2), dtype=np.float32)
        pred = np.empty((n,), dtype=np.int64)
        nb = min(chunk, n)
        key = (nb, steps, str(dev))
        if getattr(self, "_d2h_key", None) != key:
            self._d2h_key = key
            self._d2h_stream = torch.cuda.Stream(device=dev, priority=-1)
            self._d2h_stage = [(torch.empty((nb, steps, 3), dtype=torch.float64).pin_memory(),
                                torch.empty((nb, 2), dtype=torch.float32).pin_memory(),
                                torch.empty((nb,), dtype=torch.int64).pin_memory()) for _ in range(2)]
        side, stage = self._d2h_stream, self._d2h_stage
        pool = _copy_pool()

        def drain(p):
            """Chunk p's results: page-locked staging -> the returned arrays (sliced over the copy threads)."""
            i0, m, b, ev, _keep = p
            _stamp("drain: wait for the download")
            ev.synchronize()
            _stamp("drain: host copy")
            st, sp, sd = (t.numpy() for t in stage[b])
            step = max(1, (m + _COPY_THREADS - 1) // _COPY_THREADS)
            list(pool.map(lambda s0: np.copyto(traj[i0 + s0:i0 + min(m, s0 + step)], st[s0:min(m, s0 + step)]),
                          range(0, m, step)))
            probs[i0:i0 + m] = sp[:m]
            pred[i0:i0 + m] = sd[:m]
            _stamp("drain: done")

        amp = self._resolve_amp(use_amp)
        with torch.no_grad(), ops.on_device(dev):
            main = torch.cuda.current_stream(dev)
            pending, i0, k = None, 0, 0
            _stamp("buffers ready")
            for Xc in self._device_chunks(X_batch, n, chunk, dev):
                _stamp("chunk uploaded (enqueued)")
                m = Xc.shape[0]
                probs_d, _ = self._probs_device(Xc, amp)
                traj_d, _, pred_d = ops.ode_rk4(self._base_rates(), steps, 0.0, float(steps), self._substeps(),
                                                probs=probs_d, alpha=self.coupling_strength, want_traj=True,
                                                want_pred=True)
                ready = torch.cuda.Event()
                ready.record(main)
                b = k & 1               # staging set b was drained when chunk k-1 was queued
                with torch.cuda.stream(side):
                    side.wait_event(ready)
                    stage[b][0][:m].copy_(traj_d, non_blocking=True)
                    stage[b][1][:m].copy_(probs_d, non_blocking=True)
                    stage[b][2][:m].copy_(pred_d, non_blocking=True)
                    ev = torch.cuda.Event()
                    ev.record(side)
                if pending is not None:
                    drain(pending)      # the host copy of chunk k-1 runs while the GPU is in chunk k
                # the device results stay referenced until their download has been waited for (they were
                # allocated on the compute stream and are read on the side stream)
                pending = (i0, m, b, ev, (traj_d, probs_d, pred_d))
                i0 += m
                k += 1
            _stamp("all chunks enqueued")
            if pending is not None:
                drain(pending)
        self.ode_model.params = self.base_params.copy()
        _stamp("return")
        return traj, probs, pred
