"""Batched device versions of the other consumers of the LSTM->ODE path.

* ``get_three_state_probabilities`` -- 10_three_state_probabilities.py:204-290: the loop of
  06's predict_batch that keeps only the final ODE state (so the kernel never writes the
  trajectory) and a 3-way decision rule.
* ``predict_trajectory`` / ``multistep_forecast`` -- 08_forecasting.py:149-153, 215-234, 252-289:
  un-modulated ODE started from ``prob_to_ode_state(P(closed))``, read out at several horizons
  as ``F + 0.5 P``.

Same names, arguments and return values as the reference functions; the per-sample Python /
``odeint`` loops become one kernel launch over all windows.
"""
from __future__ import annotations

import numpy as np
import torch

from . import ops
from .synthetic import RATE_KEYS

RK4_SUBSTEPS = 16


def _rates(params):
    return [float(params[k]) for k in RATE_KEYS]


def _device_of(model):
    return next(model.parameters()).device


def get_lstm_probabilities(lstm_model, X_data, batch_size=256):
    """(N,2) float32 softmax probabilities (08_forecasting.py:198-212)."""
    lstm_model.eval()
    dev = _device_of(lstm_model)
    chunk = max(int(batch_size), 4096)
    out = []
    with torch.no_grad():
        for i in range(0, len(X_data), chunk):
            xb = X_data[i:i + chunk]
            if isinstance(xb, np.ndarray):
                xb = torch.from_numpy(np.ascontiguousarray(xb, dtype=np.float32))
            out.append(ops.softmax_rows(lstm_model(xb.to(dev, dtype=torch.float32)).contiguous()))
    return torch.cat(out, 0).cpu().numpy()


def get_three_state_probabilities(lstm_model, ode_model, X, batch_size=512):
    """(lstm_probs (N,2) f32, three_state_probs (N,3) f64 [Active, Passive, Fatigued],
    predictions (N,) int: 2 if F > .5 else 0 if A > .5 else 1)  -- 10:204-290."""
    from .integration import LSTMODEIntegration
    integ = LSTMODEIntegration(lstm_model, ode_model, coupling_strength=0.5)       # 10:243
    n = len(X)
    chunk = max(int(batch_size), integ.min_device_chunk)
    lstm_model.eval()
    probs_all = []
    with torch.no_grad():
        for i in range(0, n, chunk):
            probs_all.append(integ._probs_device(X[i:i + chunk])[0])
        probs = torch.cat(probs_all, 0) if len(probs_all) > 1 else probs_all[0]
        _, final, _ = ops.ode_rk4(integ._base_rates(), 20, 0.0, 20.0, integ._substeps(), probs=probs, alpha=0.5,
                                  want_traj=False, want_final=True, want_pred=False)          # 10:270
    three = final.cpu().numpy()
    pred = np.where(three[:, 2] > 0.5, 2, np.where(three[:, 0] > 0.5, 0, 1))                    # 10:281-288
    return probs.cpu().numpy(), three, pred


def prob_to_ode_state(prob_closed):
    """Host scalar version (08:215-234), kept for callers that use it directly."""
    A = 1.0 - prob_closed
    if prob_closed > 0.5:
        F, P = prob_closed * 0.6, prob_closed * 0.4
    else:
        F, P = prob_closed * 0.3, prob_closed * 0.3
    total = A + P + F
    return np.array([A / total, P / total, F / total])


def predict_trajectory(initial_state, params, n_steps, dt=1.0, device=None):
    """(n_steps+1, 3) float64 raw trajectory (08:149-153: no clamp, no clip/renormalise)."""
    dev = device or torch.device("cuda", torch.cuda.current_device())
    y0 = torch.as_tensor(np.asarray(initial_state, np.float64).reshape(1, 3), device=dev)
    traj, _, _ = ops.ode_rk4(_rates(params), n_steps + 1, 0.0, n_steps * dt, RK4_SUBSTEPS, y0=y0, want_pred=False,
                             raw=True)
    return traj[0].cpu().numpy()


def forecast_trajectories(probs, ode_params, max_horizon, device=None):
    """Device tensor (N, max_horizon+1, 3) f64: one raw trajectory per window, started from
    prob_to_ode_state(probs[i, 1])."""
    dev = device or torch.device("cuda", torch.cuda.current_device())
    p = torch.from_numpy(np.ascontiguousarray(probs, dtype=np.float32)).to(dev) if isinstance(probs, np.ndarray) \
        else probs.to(dev, dtype=torch.float32).contiguous()
    y0 = ops.prob_to_state(p)
    traj, _, _ = ops.ode_rk4(_rates(ode_params), max_horizon + 1, 0.0, float(max_horizon), RK4_SUBSTEPS, y0=y0,
                             want_pred=False, raw=True)
    return traj


def multistep_forecast(probs, ode_params, horizons=[5, 10, 20]):
    """{h: {'predictions': (N-max_h,), 'actuals': (N-max_h,)}}  (08:252-289)."""
    probs = np.asarray(probs)
    max_h = max(horizons)
    n = len(probs) - max_h
    results = {h: {"predictions": np.zeros(0), "actuals": np.zeros(0, probs.dtype)} for h in horizons}
    if n <= 0:
        return results
    traj = forecast_trajectories(probs[:n], ode_params, max_h).cpu().numpy()
    for h in horizons:
        results[h]["predictions"] = np.clip(traj[:, h, 2] + traj[:, h, 1] * 0.5, 0, 1)
        results[h]["actuals"] = probs[h:h + n, 1].copy()
    return results


def rolling_forecast_evaluation(probs, ode_params, window_size=50, horizon=10):
    """08_forecasting.py:346-392: accuracy and MAE of the `horizon`-step forecast per rolling window; returns the same
    pandas DataFrame (columns window, accuracy, mae).  All windows' trajectories come from ONE batched device solve
    (the reference solves one odeint per sample)."""
    import pandas as pd
    probs = np.asarray(probs)
    n_windows = (len(probs) - window_size - horizon) // window_size
    rows = []
    if n_windows <= 0:
        return pd.DataFrame(rows)
    n = n_windows * window_size                   # every i < n has i + horizon < len(probs) (n_windows' definition)
    traj = forecast_trajectories(probs[:n], ode_params, horizon).cpu().numpy()
    preds = np.clip(traj[:, horizon, 2] + traj[:, horizon, 1] * 0.5, 0, 1)
    actuals = probs[horizon:horizon + n, 1]
    for w in range(n_windows):
        sl = slice(w * window_size, (w + 1) * window_size)
        rows.append({"window": w, "accuracy": np.mean((preds[sl] > 0.5) == (actuals[sl] > 0.5)),
                     "mae": np.mean(np.abs(preds[sl] - actuals[sl]))})
    return pd.DataFrame(rows)
