"""Drop-in ``LSTMODEIntegration`` (reference: 06_lstm_ode_integration.py:183-406).

Same constructor and method signatures and return types.  The reference runs the LSTM in
chunks of ``batch_size`` on the device and then a per-sample Python loop of ``odeint`` solves
on the host (06:372-401); here both stages run on the GPU: logits -> softmax -> rate
modulation -> initial-state rule -> RK4 -> clip/renormalise -> decision rule are one batched
kernel launch over all windows.
"""
from __future__ import annotations

import numpy as np
import torch

from . import ops
from .synthetic import RATE_KEYS


class LSTMODEIntegration:
    #: windows per device pass.  ``predict_batch``'s ``batch_size`` argument is honoured as a
    #: lower bound; results do not depend on the chunking (every window is independent).
    min_device_chunk = 4096

    def __init__(self, lstm_model, ode_model, coupling_strength=0.5):
        self.lstm_model = lstm_model
        self.ode_model = ode_model
        self.coupling_strength = coupling_strength
        self.base_params = ode_model.params.copy()

    # ---------------------------------------------------------------------------------
    def _device(self):
        return next(self.lstm_model.parameters()).device

    def _probs_device(self, X):
        """(probs, attention) device tensors for a batch-first (B,T,C) array/tensor."""
        if isinstance(X, np.ndarray):
            X = torch.from_numpy(np.ascontiguousarray(X, dtype=np.float32))
        X = X.to(self._device(), dtype=torch.float32)
        logits, attention = self.lstm_model(X, return_attention=True)
        return ops.softmax_rows(logits.contiguous()), attention

    def get_lstm_probabilities(self, X):
        """(probs (B,2) [P(open), P(closed)], attention (B,T)) as numpy (06:216-234)."""
        self.lstm_model.eval()
        with torch.no_grad():
            probs, attention = self._probs_device(X)
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
    def predict_trajectory(self, X, initial_state=None, forecast_steps=10):
        """(trajectory (steps,3), probs (1,2), attention (1,T)) for one window (06:266-306)."""
        self.lstm_model.eval()
        with torch.no_grad():
            probs, attention = self._probs_device(X)
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

    def predict_batch_device(self, X_batch, forecast_steps=20, batch_size=512, want_traj=True):
        """Device-resident result tensors: (traj (N,steps,3) f64 | None, probs (N,2) f32, pred (N,) i64)."""
        n = len(X_batch)
        chunk = max(int(batch_size), self.min_device_chunk)
        self.lstm_model.eval()
        probs_all = []
        with torch.no_grad():
            for i in range(0, n, chunk):
                probs, _ = self._probs_device(X_batch[i:i + chunk])
                probs_all.append(probs)
            probs = torch.cat(probs_all, 0) if len(probs_all) > 1 else probs_all[0]
            traj, _, pred = ops.ode_rk4(self._base_rates(), forecast_steps, 0.0, float(forecast_steps),
                                        self._substeps(), probs=probs, alpha=self.coupling_strength,
                                        want_traj=want_traj, want_pred=True)
        return traj, probs, pred

    def predict_batch(self, X_batch, forecast_steps=20, batch_size=512, show_progress=True):
        """(trajectories (N,steps,3) f64, probs (N,2) f32, predictions (N,) int64) as numpy
        (06:308-406).  ``show_progress`` is accepted and ignored (no per-sample loop to show)."""
        traj, probs, pred = self.predict_batch_device(X_batch, forecast_steps, batch_size)
        self.ode_model.params = self.base_params.copy()
        return traj.cpu().numpy(), probs.cpu().numpy(), pred.cpu().numpy()
