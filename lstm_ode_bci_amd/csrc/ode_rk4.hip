// Batched three-state (Active/Passive/Fatigued) compartmental ODE, one window per lane.
//
// dy/dt = Q^T max(y, 0), six rates, linear; fp64 fixed-step RK4 with `substeps` sub-steps
// per output interval.  Write-bandwidth bound: 24 B per window per output point (fp64
// trajectory, the dtype the reference returns).  Each 64-lane wave stages CH output points
// per window in LDS and writes them out as 24*CH-byte contiguous runs per window instead
// of 64 scattered 24-B stores per step.
#include "lob_common.h"

namespace {

constexpr int CH = 16;   // output points staged per window before a coalesced flush

struct OdeArgs {
    const float* probs; const double* y0_in;
    double k_ap, k_af, k_pa, k_pf, k_fa, k_fp, alpha;
    int n_points, substeps; double t0, t1;
    double* traj; double* final_state; int64_t* pred; int B; int flags;
};

template <bool RAW>
__device__ __forceinline__ void rhs(const double (&k)[6], double a, double p, double f,
                                    double& da, double& dp, double& df) {
    if (!RAW) { a = fmax(a, 0.0); p = fmax(p, 0.0); f = fmax(f, 0.0); }
    // k = {k_ap, k_af, k_pa, k_pf, k_fa, k_fp}; same operation order as the reference rhs
    da = -k[0] * a - k[1] * a + k[2] * p + k[4] * f;
    dp = k[0] * a - k[2] * p - k[3] * p + k[5] * f;
    df = k[1] * a + k[3] * p - k[4] * f - k[5] * f;
}

template <bool RAW>
__device__ __forceinline__ void post(double a, double p, double f, double (&o)[3]) {
    if (RAW) { o[0] = a; o[1] = p; o[2] = f; return; }
    a = fmin(fmax(a, 0.0), 1.0); p = fmin(fmax(p, 0.0), 1.0); f = fmin(fmax(f, 0.0), 1.0);
    const double s = a + p + f;
    o[0] = a / s; o[1] = p / s; o[2] = f / s;
}

template <bool RAW>
__global__ __launch_bounds__(64) void ode_rk4_kernel(OdeArgs g) {
    __shared__ double stage[64 * CH * 3];
    const int lane = threadIdx.x;
    const int b0 = blockIdx.x * 64;
    const int b = b0 + lane;
    const bool live = b < g.B;

    double k[6] = {g.k_ap, g.k_af, g.k_pa, g.k_pf, g.k_fa, g.k_fp};
    double a = 1.0 / 3, p = 1.0 / 3, f = 1.0 / 3;
    if (live) {
        if (g.probs) {
            const float po32 = g.probs[2 * (size_t)b + 0], pc32 = g.probs[2 * (size_t)b + 1];
            const double p_open = (double)po32, p_closed = (double)pc32;
            // strict '>' on the float32 probabilities (np.float32 > 0.6 compares in float32
            // under NumPy >= 2; differs from the float64 compare only at p == float32(0.6))
            if (pc32 > 0.6f) { a = 0.2; p = 0.2; f = 0.6; }
            else if (po32 > 0.6f) { a = 0.6; p = 0.2; f = 0.2; }
            else { a = 0.33; p = 0.34; f = 0.33; }
            k[1] = k[1] * (1.0 + g.alpha * p_closed);
            k[3] = k[3] * (1.0 + g.alpha * p_closed);
            k[4] = k[4] * (1.0 + g.alpha * p_open);
            k[2] = k[2] * (1.0 + g.alpha * p_open);
#pragma unroll
            for (int i = 0; i < 6; ++i) k[i] = fmax(0.001, k[i]);
        } else {
            a = g.y0_in[3 * (size_t)b + 0]; p = g.y0_in[3 * (size_t)b + 1]; f = g.y0_in[3 * (size_t)b + 2];
        }
        if (!RAW) { const double s = a + p + f; a /= s; p /= s; f /= s; }
    }
    const int n = g.n_points;
    const double h = n > 1 ? (g.t1 - g.t0) / (double)(n - 1) / (double)g.substeps : 0.0;
    double o[3];

    // Coupled mode: the system is linear with constant coefficients and (rates floored at 0.001,
    // y0 on the simplex) the trajectory stays inside the simplex, where the clamp of the rhs is
    // the identity.  `substeps` RK4 steps of size h are then exactly one multiplication by
    // M = R(hA)^substeps, R(z) = I + z + z^2/2 + z^3/6 + z^4/24, A = Q^T -- the same arithmetic as
    // stepping, up to fp64 rounding, at 1/60 of the serial work per output point.
    const bool propagate = !RAW && g.probs != nullptr;
    double M[3][3];
    if (propagate) {
        const double A_[3][3] = {{-(k[0] + k[1]) * h, k[2] * h, k[4] * h},
                                 {k[0] * h, -(k[2] + k[3]) * h, k[5] * h},
                                 {k[1] * h, k[3] * h, -(k[4] + k[5]) * h}};
        double P2[3][3], P3[3][3], P4[3][3];
        auto mm = [](const double (&x)[3][3], const double (&y)[3][3], double (&z)[3][3]) {
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j) z[i][j] = x[i][0] * y[0][j] + x[i][1] * y[1][j] + x[i][2] * y[2][j];
        };
        mm(A_, A_, P2); mm(P2, A_, P3); mm(P3, A_, P4);
        double R[3][3];
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j)
                R[i][j] = (i == j ? 1.0 : 0.0) + A_[i][j] + P2[i][j] * 0.5 + P3[i][j] * (1.0 / 6.0) + P4[i][j] * (1.0 / 24.0);
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) M[i][j] = R[i][j];
        for (int ss = 1; ss < g.substeps; ++ss) {
            double Tm[3][3];
            mm(M, R, Tm);
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j) M[i][j] = Tm[i][j];
        }
    }

    for (int s0 = 0; s0 < n; s0 += CH) {
        const int cnt = min(CH, n - s0);
        for (int j = 0; j < cnt; ++j) {
            if (s0 + j > 0) {
                if (propagate) {
                    const double na = M[0][0] * a + M[0][1] * p + M[0][2] * f;
                    const double np_ = M[1][0] * a + M[1][1] * p + M[1][2] * f;
                    const double nf = M[2][0] * a + M[2][1] * p + M[2][2] * f;
                    a = na; p = np_; f = nf;
                } else {
                    for (int ss = 0; ss < g.substeps; ++ss) {
                        double k1a, k1p, k1f, k2a, k2p, k2f, k3a, k3p, k3f, k4a, k4p, k4f;
                        rhs<RAW>(k, a, p, f, k1a, k1p, k1f);
                        rhs<RAW>(k, a + 0.5 * h * k1a, p + 0.5 * h * k1p, f + 0.5 * h * k1f, k2a, k2p, k2f);
                        rhs<RAW>(k, a + 0.5 * h * k2a, p + 0.5 * h * k2p, f + 0.5 * h * k2f, k3a, k3p, k3f);
                        rhs<RAW>(k, a + h * k3a, p + h * k3p, f + h * k3f, k4a, k4p, k4f);
                        a += h / 6.0 * (k1a + 2.0 * k2a + 2.0 * k3a + k4a);
                        p += h / 6.0 * (k1p + 2.0 * k2p + 2.0 * k3p + k4p);
                        f += h / 6.0 * (k1f + 2.0 * k2f + 2.0 * k3f + k4f);
                    }
                }
            }
            if (g.traj) {
                post<RAW>(a, p, f, o);
                double* st = stage + (lane * CH + j) * 3;
                st[0] = o[0]; st[1] = o[1]; st[2] = o[2];
            }
        }
        if (g.traj) {
            __syncthreads();
            const int per = cnt * 3;                       // doubles per window in this chunk
            const int total = 64 * per;
            for (int e = lane; e < total; e += 64) {
                const int w = e / per, j = e - w * per;
                if (b0 + w < g.B)
                    g.traj[((size_t)(b0 + w) * n + s0) * 3 + j] = stage[w * CH * 3 + j];
            }
            __syncthreads();
        }
    }
    if (live) {
        post<RAW>(a, p, f, o);
        if (g.final_state) {
            g.final_state[3 * (size_t)b + 0] = o[0]; g.final_state[3 * (size_t)b + 1] = o[1];
            g.final_state[3 * (size_t)b + 2] = o[2];
        }
        if (g.pred) g.pred[b] = o[2] > 0.5 ? 1 : 0;
    }
}

// y0 = prob_to_ode_state(P(closed)) of 08_forecasting.py:215-234, one window per thread.
__global__ void prob_to_state_kernel(const float* __restrict__ probs, double* __restrict__ y0, int B) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const double pc = (double)probs[2 * (size_t)b + 1];
    const double A = 1.0 - pc;
    double F, P;
    if (probs[2 * (size_t)b + 1] > 0.5f) { F = pc * 0.6; P = pc * 0.4; } else { F = pc * 0.3; P = pc * 0.3; }
    const double tot = A + P + F;
    y0[3 * (size_t)b + 0] = A / tot; y0[3 * (size_t)b + 1] = P / tot; y0[3 * (size_t)b + 2] = F / tot;
}

}  // namespace

extern "C" int lob_ode_rk4_f64(const float* probs, const double* y0_in, const double* base_rates,
                               double alpha, int n_points, double t0, double t1, int substeps,
                               double* traj, double* final_state, int64_t* pred, int B, int flags,
                               void* stream) {
    if (!base_rates || B <= 0 || n_points <= 0 || substeps <= 0) return LOB_E_ARG;
    if (!probs && !y0_in) return LOB_E_ARG;
    if (!traj && !final_state && !pred) return LOB_E_ARG;
    OdeArgs g{probs, y0_in, base_rates[0], base_rates[1], base_rates[2], base_rates[3], base_rates[4],
              base_rates[5], alpha, n_points, substeps, t0, t1, traj, final_state, pred, B, flags};
    if (flags & LOB_ODE_RAW)
        hipLaunchKernelGGL(ode_rk4_kernel<true>, dim3((B + 63) / 64), dim3(64), 0, (hipStream_t)stream, g);
    else
        hipLaunchKernelGGL(ode_rk4_kernel<false>, dim3((B + 63) / 64), dim3(64), 0, (hipStream_t)stream, g);
    LOB_CHECK_LAUNCH();
    return 0;
}

extern "C" int lob_prob_to_state_f64(const float* probs, double* y0, int B, void* stream) {
    if (!probs || !y0 || B <= 0) return LOB_E_ARG;
    hipLaunchKernelGGL(prob_to_state_kernel, dim3((B + 255) / 256), dim3(256), 0, (hipStream_t)stream, probs, y0, B);
    LOB_CHECK_LAUNCH();
    return 0;
}
