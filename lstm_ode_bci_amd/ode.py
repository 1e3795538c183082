"""Drop-in ``CognitiveStateODE`` (reference: 06_lstm_ode_integration.py:146-180; the fuller
05_ode_model.py:58-169 variant has the same ``ode_system`` / ``solve``).

``solve`` integrates on the GPU with the batched fp64 RK4 kernel (``lob_ode_rk4_f64``); the
reference calls ``scipy.integrate.odeint`` (LSODA) per sample on the host.  With the default
16 sub-steps per output interval the two agree to < 1e-6 on every admissible rate set
(tests/test_gpu_parity.py).  ``params`` stays a plain mutable dict attribute because callers
read and re-assign it (06:214, 296, 304, 386, 404; 10:242, 267, 276).
"""
from __future__ import annotations

import numpy as np
import torch

from . import ops
from .synthetic import DEFAULT_RATES, RATE_KEYS


class CognitiveStateODE:
    """Three-state (Active / Passive / Fatigued) compartmental model."""

    rk4_substeps = 16

    def __init__(self, params=None):
        if params is None:
            self.params = dict(DEFAULT_RATES)
        else:
            self.params = params
        self.state_names = ["Active", "Passive", "Fatigued"]
        self.state_labels = ["A", "P", "F"]
        self.device = None          # None -> current CUDA device

    # -- host-side scalar helpers (pure bookkeeping, no integration) -------------------
    def ode_system(self, y, t, params=None):
        """Right-hand side, list of 3 floats (06:158-172)."""
        if params is None:
            params = self.params
        A, P, F = max(0, y[0]), max(0, y[1]), max(0, y[2])
        k_ap, k_af = params["k_ap"], params["k_af"]
        k_pa, k_pf = params["k_pa"], params["k_pf"]
        k_fa, k_fp = params["k_fa"], params["k_fp"]
        return [-k_ap * A - k_af * A + k_pa * P + k_fa * F,
                k_ap * A - k_pa * P - k_pf * P + k_fp * F,
                k_af * A + k_pf * P - k_fa * F - k_fp * F]

    def get_transition_matrix(self):
        """Q matrix, rows = from-state (05:223-242)."""
        p = self.params
        return np.array([[-(p["k_ap"] + p["k_af"]), p["k_ap"], p["k_af"]],
                         [p["k_pa"], -(p["k_pa"] + p["k_pf"]), p["k_pf"]],
                         [p["k_fa"], p["k_fp"], -(p["k_fa"] + p["k_fp"])]])

    def _rates(self, params=None):
        p = self.params if params is None else params
        return [float(p[k]) for k in RATE_KEYS]

    def _dev(self):
        return torch.device("cuda", torch.cuda.current_device()) if self.device is None else self.device

    # -- integration ------------------------------------------------------------------
    def solve(self, initial_state, t_span, n_points=100, method="odeint"):
        """(t (n,), solution (n,3) float64), clipped to [0,1] and row-normalised (06:174-180; 05:137-169).
        ``method='odeint'`` (the default and the only value the reference's scripts ever pass; 06:174 has no such
        argument at all) integrates with the fixed-step fp64 RK4 kernel, within 1e-6 of LSODA.  ANY other value takes the
        reference's ``else:`` branch (05:154-163: ``solve_ivp``, scipy's adaptive RK45 at its default rtol 1e-3 / atol
        1e-6) and runs the SAME kernel here: the system is linear and the kernel's result is the more accurate of the
        two, so it differs from scipy's RK45 output by that solver's own error (<= 2e-3 measured,
        tests/test_gpu_parity.py) -- inside the tolerance the reference asked its solver for."""
        t = np.linspace(t_span[0], t_span[1], n_points)
        y0 = torch.as_tensor(np.asarray(initial_state, dtype=np.float64).reshape(1, 3), device=self._dev())
        traj, _, _ = ops.ode_rk4(self._rates(), n_points, t_span[0], t_span[1], self.rk4_substeps,
                                 y0=y0, want_pred=False)
        return t, traj[0].cpu().numpy()

    def solve_batch(self, initial_states, t_span, n_points=100):
        """Batched ``solve``: initial_states (B,3) -> (t, (B,n,3) float64 numpy)."""
        t = np.linspace(t_span[0], t_span[1], n_points)
        y0 = torch.as_tensor(np.ascontiguousarray(initial_states, dtype=np.float64), device=self._dev())
        traj, _, _ = ops.ode_rk4(self._rates(), n_points, t_span[0], t_span[1], self.rk4_substeps,
                                 y0=y0, want_pred=False)
        return t, traj.cpu().numpy()

    def get_steady_state(self):
        """Long-horizon numerical steady state (05:198-221)."""
        _, sol = self.solve([0.33, 0.33, 0.34], (0, 1000), 1000)
        return {"Active": sol[-1][0], "Passive": sol[-1][1], "Fatigued": sol[-1][2]}
