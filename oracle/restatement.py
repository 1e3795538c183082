"""ORACLE -- TEST INFRASTRUCTURE ONLY.  Never imported by the product package.

CPU restatement (numpy, explicit loops, fp64-capable) of the reference's
LSTM-ODE inner loop.  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import this module, and only as the
checker.

Parity pin: the reference has no tests or golden vectors of its own
(SURVEY.md §4).  This restatement is pinned by fixtures generated in the build
container by exec-importing the reference classes
(``tests/golden/make_goldens.py`` -> ``tests/golden/*.npz``); see
``tests/test_oracle_vs_golden.py``.

Each function cites the reference lines it restates (paths relative to
/root/reference).  The arithmetic of the LSTM cell and of the ODE integrator
lives in third-party code the reference calls (torch ``nn.LSTM`` ->
oneDNN; ``scipy.integrate.odeint`` -> ODEPACK LSODA); their published
semantics are restated here.
"""
from __future__ import annotations

import math

import numpy as np

RATE_KEYS = ("k_ap", "k_af", "k_pa", "k_pf", "k_fa", "k_fp")

try:  # vectorised erf; scipy is in the image, math.erf is the fallback
    from scipy.special import erf as _erf
except Exception:  # pragma: no cover
    _erf = np.vectorize(math.erf)


# --------------------------------------------------------------------------- #
# elementary pieces
# --------------------------------------------------------------------------- #
def gelu_erf(x):
    """nn.GELU() default = exact erf form (04_lstm_model.py:176, 198, 201)."""
    return 0.5 * x * (1.0 + _erf(x / math.sqrt(2.0)))


def sigmoid(x):
    return 1.0 / (1.0 + np.exp(-x))


def layer_norm(x, w, b, eps=1e-5):
    """nn.LayerNorm over the last axis, biased variance (04_lstm_model.py:175, 192)."""
    mu = x.mean(axis=-1, keepdims=True)
    var = ((x - mu) ** 2).mean(axis=-1, keepdims=True)
    return (x - mu) / np.sqrt(var + eps) * w + b


def softmax(x, axis):
    m = x.max(axis=axis, keepdims=True)
    e = np.exp(x - m)
    return e / e.sum(axis=axis, keepdims=True)


# --------------------------------------------------------------------------- #
# LSTM (torch.nn.LSTM semantics; call site 04_lstm_model.py:181-188, 211)
# --------------------------------------------------------------------------- #
def lstm_direction(x, w_ih, w_hh, b_ih, b_hh, reverse):
    """One direction of one layer.  x: (B,T,K) -> h: (B,T,H).

    Gate rows are stacked [i | f | g | o]; both biases are added; h_{-1} =
    c_{-1} = 0; the reverse direction walks t = T-1..0 and stores h_t at t.
    """
    B, T, _ = x.shape
    H = w_hh.shape[1]
    h = np.zeros((B, H), x.dtype)
    c = np.zeros((B, H), x.dtype)
    out = np.zeros((B, T, H), x.dtype)
    steps = range(T - 1, -1, -1) if reverse else range(T)
    for t in steps:
        z = x[:, t, :] @ w_ih.T + b_ih + h @ w_hh.T + b_hh
        i = sigmoid(z[:, 0 * H:1 * H])
        f = sigmoid(z[:, 1 * H:2 * H])
        g = np.tanh(z[:, 2 * H:3 * H])
        o = sigmoid(z[:, 3 * H:4 * H])
        c = f * c + i * g
        h = o * np.tanh(c)
        out[:, t, :] = h
    return out


def lstm_stack(x, sd, num_layers, bidirectional):
    """All layers; eval-mode (no inter-layer dropout)."""
    per_layer = []
    for layer in range(num_layers):
        outs = []
        for sfx, rev in (("", False), ("_reverse", True)):
            if rev and not bidirectional:
                continue
            outs.append(lstm_direction(
                x, sd[f"lstm.weight_ih_l{layer}{sfx}"], sd[f"lstm.weight_hh_l{layer}{sfx}"],
                sd[f"lstm.bias_ih_l{layer}{sfx}"], sd[f"lstm.bias_hh_l{layer}{sfx}"], rev))
        x = np.concatenate(outs, axis=-1)
        per_layer.append(x)
    return x, per_layer


# --------------------------------------------------------------------------- #
# EnhancedLSTMModel.forward, eval mode (04_lstm_model.py:206-222)
# --------------------------------------------------------------------------- #
def model_forward(sd, x, num_layers=3, bidirectional=True, dtype=np.float64):
    """Returns a dict with every intermediate the goldens pin.

    input_proj 04:173-178/208; lstm 04:211; layer_norm 04:212; attention
    04:123-128/215; classifier 04:196-204/218.
    """
    sd = {k: np.asarray(v, dtype=dtype) for k, v in sd.items()}
    x = np.asarray(x, dtype=dtype)
    res = {}
    a = x @ sd["input_proj.0.weight"].T + sd["input_proj.0.bias"]
    a = gelu_erf(layer_norm(a, sd["input_proj.1.weight"], sd["input_proj.1.bias"]))
    res["input_proj"] = a
    y, per_layer = lstm_stack(a, sd, num_layers, bidirectional)
    res["lstm_layers"] = per_layer
    res["lstm"] = y
    v = layer_norm(y, sd["layer_norm.weight"], sd["layer_norm.bias"])
    res["layer_norm"] = v
    u = np.tanh(v @ sd["attention.attention.0.weight"].T + sd["attention.attention.0.bias"])
    s = u @ sd["attention.attention.2.weight"].T + sd["attention.attention.2.bias"]  # (B,T,1)
    w = softmax(s, axis=1)                                                            # over TIME
    ctx = (w * v).sum(axis=1)
    res["context"] = ctx
    res["attn"] = w[..., 0]
    z = gelu_erf(ctx @ sd["classifier.0.weight"].T + sd["classifier.0.bias"])
    z = gelu_erf(z @ sd["classifier.3.weight"].T + sd["classifier.3.bias"])
    res["logits"] = z @ sd["classifier.6.weight"].T + sd["classifier.6.bias"]
    res["probs"] = softmax(res["logits"], axis=1)          # 06_lstm_ode_integration.py:232
    return res


# --------------------------------------------------------------------------- #
# ODE (06_lstm_ode_integration.py:146-180; full variant 05_ode_model.py:58-169)
# --------------------------------------------------------------------------- #
def q_matrix(p):
    """Rows = from-state (05_ode_model.py:236-240)."""
    return np.array([
        [-(p["k_ap"] + p["k_af"]), p["k_ap"], p["k_af"]],
        [p["k_pa"], -(p["k_pa"] + p["k_pf"]), p["k_pf"]],
        [p["k_fa"], p["k_fp"], -(p["k_fa"] + p["k_fp"])],
    ], dtype=np.float64)


def ode_rhs(y, t, p):
    """06_lstm_ode_integration.py:158-172 (clamp at 0, then the three linear equations)."""
    A, P, F = max(0.0, y[0]), max(0.0, y[1]), max(0.0, y[2])
    dA = -p["k_ap"] * A - p["k_af"] * A + p["k_pa"] * P + p["k_fa"] * F
    dP = p["k_ap"] * A - p["k_pa"] * P - p["k_pf"] * P + p["k_fp"] * F
    dF = p["k_af"] * A + p["k_pf"] * P - p["k_fa"] * F - p["k_fp"] * F
    return [dA, dP, dF]


def _post(sol):
    """clip to [0,1] then row-renormalise (06_lstm_ode_integration.py:178-179)."""
    sol = np.clip(sol, 0.0, 1.0)
    return sol / sol.sum(axis=1, keepdims=True)


def solve_odeint(initial_state, t_span, n_points, p):
    """The reference's own call: scipy odeint == ODEPACK LSODA (06:174-180)."""
    from scipy.integrate import odeint
    t = np.linspace(t_span[0], t_span[1], n_points)
    y0 = np.array(initial_state, dtype=np.float64)
    y0 = y0 / y0.sum()
    sol = odeint(ode_rhs, y0, t, args=(p,))
    return t, _post(sol)


def solve_ivp_rk45(initial_state, t_span, n_points, p):
    """The reference's other branch: scipy solve_ivp, RK45, default tolerances, t_eval = the output grid
    (05_ode_model.py:157-163), then the same clip + renormalise."""
    from scipy.integrate import solve_ivp
    t = np.linspace(t_span[0], t_span[1], n_points)
    y0 = np.array(initial_state, dtype=np.float64)
    y0 = y0 / y0.sum()
    sol = solve_ivp(lambda tt, y: ode_rhs(y, tt, p), t_span, y0, t_eval=t, method="RK45")
    return t, _post(sol.y.T)


def solve_expm(initial_state, t_span, n_points, p):
    """Closed form y(t) = expm(Q^T t) y0 (the clamp is inactive inside the simplex)."""
    from scipy.linalg import expm
    t = np.linspace(t_span[0], t_span[1], n_points)
    y0 = np.array(initial_state, dtype=np.float64)
    y0 = y0 / y0.sum()
    QT = q_matrix(p).T
    sol = np.stack([expm(QT * (ti - t[0])) @ y0 for ti in t])
    return t, _post(sol)


def solve_rk4(initial_state, t_span, n_points, p, substeps=16, dtype=np.float64):
    """Fixed-step RK4, ``substeps`` per output interval -- the algorithm of the HIP kernel."""
    t = np.linspace(t_span[0], t_span[1], n_points)
    y = np.array(initial_state, dtype=dtype)
    y = y / y.sum()
    QT = q_matrix(p).T.astype(dtype)
    out = np.zeros((n_points, 3), dtype)
    out[0] = y
    if n_points > 1:
        h = dtype((t_span[1] - t_span[0]) / (n_points - 1) / substeps)
        f = lambda v: QT @ np.maximum(v, 0)
        for n in range(1, n_points):
            for _ in range(substeps):
                k1 = f(y)
                k2 = f(y + dtype(0.5) * h * k1)
                k3 = f(y + dtype(0.5) * h * k2)
                k4 = f(y + h * k3)
                y = y + h / dtype(6.0) * (k1 + dtype(2.0) * k2 + dtype(2.0) * k3 + k4)
            out[n] = y
    return t, _post(out.astype(np.float64))


# --------------------------------------------------------------------------- #
# coupling (06_lstm_ode_integration.py:236-264, 285-292, 372-401)
# --------------------------------------------------------------------------- #
def modulate_rates(base, alpha, p_closed, p_open):
    """06:249-264.  k_af,k_pf *= 1+alpha*p_closed; k_fa,k_pa *= 1+alpha*p_open; floor 0.001."""
    p = dict(base)
    p["k_af"] = p["k_af"] * (1 + alpha * p_closed)
    p["k_pf"] = p["k_pf"] * (1 + alpha * p_closed)
    p["k_fa"] = p["k_fa"] * (1 + alpha * p_open)
    p["k_pa"] = p["k_pa"] * (1 + alpha * p_open)
    for k in p:
        p[k] = max(0.001, p[k])
    return p


def initial_state_rule(p_closed, p_open):
    """06:377-382 (strict '>' comparisons, p_closed tested first)."""
    if p_closed > 0.6:
        return [0.2, 0.2, 0.6]
    if p_open > 0.6:
        return [0.6, 0.2, 0.2]
    return [0.33, 0.34, 0.33]


def predict_from_probs(probs, base, alpha, forecast_steps, solver=solve_odeint):
    """Step 2 of predict_batch (06:372-401).  probs: (N,2) float32 [P(open), P(closed)]."""
    trajs, preds = [], []
    for i in range(len(probs)):
        p_open, p_closed = probs[i, 0], probs[i, 1]
        y0 = initial_state_rule(p_closed, p_open)
        mp = modulate_rates(base, alpha, p_closed, p_open)
        _, traj = solver(y0, (0, forecast_steps), forecast_steps, mp)
        trajs.append(traj)
        preds.append(1 if traj[-1][2] > 0.5 else 0)
    return np.array(trajs), np.array(preds)


# --------------------------------------------------------------------------- #
# other consumers of the path: 08_forecasting.py, 10_three_state_probabilities.py
# --------------------------------------------------------------------------- #
def prob_to_ode_state(prob_closed):
    """08_forecasting.py:215-234."""
    A = 1.0 - prob_closed
    if prob_closed > 0.5:
        F, P = prob_closed * 0.6, prob_closed * 0.4
    else:
        F, P = prob_closed * 0.3, prob_closed * 0.3
    total = A + P + F
    return np.array([A / total, P / total, F / total])


def raw_rhs(y, t, p):
    """08_forecasting.py:132-146 (no clamp)."""
    A, P, F = y
    return [-p["k_ap"] * A - p["k_af"] * A + p["k_pa"] * P + p["k_fa"] * F,
            p["k_ap"] * A - p["k_pa"] * P - p["k_pf"] * P + p["k_fp"] * F,
            p["k_af"] * A + p["k_pf"] * P - p["k_fa"] * F - p["k_fp"] * F]


def forecast_raw(initial_state, p, n_steps, dt=1.0):
    """08_forecasting.py:149-153: raw odeint trajectory, n_steps+1 points, no post-processing."""
    from scipy.integrate import odeint
    t = np.linspace(0, n_steps * dt, n_steps + 1)
    return odeint(raw_rhs, initial_state, t, args=(p,))


def multistep_forecast(probs, p, horizons=(5, 10, 20)):
    """08_forecasting.py:252-289."""
    max_h = max(horizons)
    res = {h: {"predictions": [], "actuals": []} for h in horizons}
    for i in range(len(probs) - max_h):
        traj = forecast_raw(prob_to_ode_state(probs[i, 1]), p, max_h)
        for h in horizons:
            res[h]["predictions"].append(np.clip(traj[h, 2] + traj[h, 1] * 0.5, 0, 1))
            res[h]["actuals"].append(probs[i + h, 1])
    return {h: {k: np.array(v) for k, v in r.items()} for h, r in res.items()}


def rolling_forecast(probs, p, window_size=50, horizon=10):
    """08_forecasting.py:346-392: [(window, accuracy, mae)] per rolling window."""
    out = []
    for w in range((len(probs) - window_size - horizon) // window_size):
        preds, acts = [], []
        for i in range(w * window_size, (w + 1) * window_size):
            if i + horizon >= len(probs):
                break
            traj = forecast_raw(prob_to_ode_state(probs[i, 1]), p, horizon)
            preds.append(np.clip(traj[horizon, 2] + traj[horizon, 1] * 0.5, 0, 1))
            acts.append(probs[i + horizon, 1])
        if preds:
            preds, acts = np.array(preds), np.array(acts)
            out.append((w, np.mean((preds > 0.5) == (acts > 0.5)), np.mean(np.abs(preds - acts))))
    return out


def three_state_from_probs(probs, base):
    """Step 2 of 10_three_state_probabilities.py:239-290 (alpha = 0.5, 20 points over [0,20])."""
    traj, _ = predict_from_probs(probs, base, 0.5, 20)
    final = traj[:, -1]
    pred = np.where(final[:, 2] > 0.5, 2, np.where(final[:, 0] > 0.5, 0, 1))
    return final, pred
