// HBM-bound row-wise pieces of the forward pass: LayerNorm(+GELU+dropout, + the
// (b,t)->(t,b) relayout), additive-attention pooling over time, row softmax.
#include <stdlib.h>
#include "lob_common.h"

namespace {

constexpr int LN_MAX_PER_LANE = 16;   // width <= 1024

// One wave per row; lanes stride the row (coalesced 256 B per wave-load).
__global__ __launch_bounds__(256) void layernorm_act_kernel(
    const float* __restrict__ in, const float* __restrict__ gamma, const float* __restrict__ beta,
    float* __restrict__ out, int rows, int width, float eps, int act,
    int remap_T, int remap_B, int remap_Bp, float drop_p, uint64_t seed) {
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int nwaves = (gridDim.x * blockDim.x) >> 6;
    const int per = (width + 63) >> 6;
    const bool norm = !(act & LOB_LN_IDENTITY);      // identity: y = act(x) (the no-LayerNorm ablation, 09:190)
    act &= 0xff;
    for (int row = wave; row < rows; row += nwaves) {
        const float* x = in + (size_t)row * width;
        float v[LN_MAX_PER_LANE];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < LN_MAX_PER_LANE; ++i) {
            const int c = lane + 64 * i;
            v[i] = (i < per && c < width) ? x[c] : 0.f;
            s += v[i];
        }
        const float mean = norm ? wave_sum(s) / (float)width : 0.f;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < LN_MAX_PER_LANE; ++i) {
            const int c = lane + 64 * i;
            const float dlt = (i < per && c < width) ? v[i] - mean : 0.f;
            q += dlt * dlt;
        }
        const float rstd = norm ? rsqrtf(wave_sum(q) / (float)width + eps) : 1.f;
        int orow = row;
        if (remap_T > 0) { const int b = row / remap_T, t = row % remap_T; orow = t * remap_Bp + b; }
        float* y = out + (size_t)orow * width;
#pragma unroll
        for (int i = 0; i < LN_MAX_PER_LANE; ++i) {
            const int c = lane + 64 * i;
            if (i < per && c < width) {
                float o = norm ? (v[i] - mean) * rstd * gamma[c] + beta[c] : v[i];
                o = apply_act(o, act);
                if (drop_p > 0.f) o *= lob_dropout_scale(seed, (uint64_t)orow * width + c, drop_p);
                y[c] = o;
            }
        }
    }
}

// One workgroup per window b.
template <typename VE>
__global__ __launch_bounds__(256) void attn_pool_fwd_kernel(
    const VE* __restrict__ V, const float* __restrict__ U, const float* __restrict__ w2,
    const float* __restrict__ b2, float* __restrict__ ctx, float* __restrict__ attn,
    int T, int Bp, int W, int W2) {
    extern __shared__ __attribute__((aligned(16))) float sc[];   // [T] scores, then weights
    __shared__ float red[8];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float bias2 = b2 ? b2[0] : 0.f;
    if (U && W2 == 0) {      // U = finished scores S[B][T], bias included (lob_attn_scores_f32)
        for (int t = tid; t < T; t += 256) sc[t] = U[(size_t)b * T + t];
    } else {
        for (int t = wave; t < T; t += 4) {
            float s = 0.f;
            if (U) {         // U == NULL: all scores equal -> weights 1/T (mean pooling over time, 09:236)
                const float* u = U + ((size_t)t * Bp + b) * W2;
                for (int j = lane; j < W2; j += 64) s = fmaf(u[j], w2[j], s);
                s = wave_sum(s);
            }
            if (lane == 0) sc[t] = s + bias2;
        }
    }
    __syncthreads();
    float m = -INFINITY;
    for (int t = tid; t < T; t += 256) m = fmaxf(m, sc[t]);
    m = wave_max(m);
    if (lane == 0) red[wave] = m;
    __syncthreads();
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    float l = 0.f;
    for (int t = tid; t < T; t += 256) { const float e = expf(sc[t] - m); sc[t] = e; l += e; }
    l = wave_sum(l);
    if (lane == 0) red[4 + wave] = l;
    __syncthreads();
    const float inv = 1.0f / (red[4] + red[5] + red[6] + red[7]);
    for (int t = tid; t < T; t += 256) { const float a = sc[t] * inv; sc[t] = a; attn[(size_t)b * T + t] = a; }
    __syncthreads();
    for (int c = tid; c < W; c += 256) {
        const VE* v = V + (size_t)b * W + c;
        float acc = 0.f;
        for (int t = 0; t < T; ++t) acc = fmaf(sc[t], (float)v[(size_t)t * Bp * W], acc);
        ctx[(size_t)b * W + c] = acc;
    }
}

__global__ void softmax_rows_kernel(const float* __restrict__ in, float* __restrict__ out, int rows, int cols) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= rows) return;
    const float* x = in + (size_t)r * cols;
    float m = x[0];
    for (int c = 1; c < cols; ++c) m = fmaxf(m, x[c]);
    float l = 0.f;
    for (int c = 0; c < cols; ++c) l += expf(x[c] - m);
    for (int c = 0; c < cols; ++c) out[(size_t)r * cols + c] = expf(x[c] - m) / l;
}

// out[i] = in[i] * keep(seed, i) / (1-p): forward AND backward of nn.Dropout (same seed).
__global__ __launch_bounds__(256) void dropout_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                      size_t n, float p, uint64_t seed) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        out[i] = in[i] * lob_dropout_scale(seed, i, p);
}


__device__ __forceinline__ float gelu_grad(float x) {
    // d/dx [0.5 x (1 + erf(x/sqrt2))] = 0.5 (1 + erf(x/sqrt2)) + x exp(-x^2/2) / sqrt(2 pi)
    float e;                   // e^{-x^2/2}: shared between the erf and the density term
    const float er = erf_as(x * 0.70710678118654752440f, &e);
    return 0.5f * (1.0f + er) + x * e * 0.39894228040143267794f;
}

__global__ __launch_bounds__(256) void act_kernel(const float* __restrict__ in, float* __restrict__ out, size_t n, int act) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) out[i] = apply_act(in[i], act);
}

__global__ __launch_bounds__(256) void act_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ pre,
                                                      float* __restrict__ dx, size_t n, int act) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const float x = pre[i];
        float g = 1.f;
        if (act == LOB_ACT_GELU) g = gelu_grad(x);
        else if (act == LOB_ACT_TANH) { const float t = fast_tanh(x); g = 1.f - t * t; }
        dx[i] = dy[i] * g;
    }
}

// Backward of layernorm_act_kernel.  One wave per row; per-wave register partials of
// dgamma / dbeta, one atomicAdd per column per wave at the end.
__global__ __launch_bounds__(256) void layernorm_act_bwd_kernel(
    const float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
    const float* __restrict__ dy, float* __restrict__ dx, float* __restrict__ dgamma, float* __restrict__ dbeta,
    int rows, int width, float eps, int act, int remap_T, int remap_B, int remap_Bp, float drop_p, uint64_t seed) {
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int nwaves = (gridDim.x * blockDim.x) >> 6;
    const int per = (width + 63) >> 6;
    float dga[LN_MAX_PER_LANE], dba[LN_MAX_PER_LANE], gm[LN_MAX_PER_LANE], bt[LN_MAX_PER_LANE];
    const bool norm = !(act & LOB_LN_IDENTITY);
    act &= 0xff;
#pragma unroll
    for (int i = 0; i < LN_MAX_PER_LANE; ++i) {
        const int c = lane + 64 * i;
        dga[i] = 0.f; dba[i] = 0.f;
        gm[i] = (i < per && c < width) ? (norm ? gamma[c] : 1.f) : 0.f;
        bt[i] = (i < per && c < width && norm) ? beta[c] : 0.f;
    }
    const float invw = 1.0f / (float)width;
    for (int row = wave; row < rows; row += nwaves) {
        const float* xr = x + (size_t)row * width;
        int orow = row;
        if (remap_T > 0) { const int b = row / remap_T, t = row % remap_T; orow = t * remap_Bp + b; }
        const float* dyr = dy + (size_t)orow * width;
        float v[LN_MAX_PER_LANE], go[LN_MAX_PER_LANE];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < LN_MAX_PER_LANE; ++i) {
            const int c = lane + 64 * i;
            const bool ok = (i < per && c < width);
            v[i] = ok ? xr[c] : 0.f;
            go[i] = ok ? dyr[c] : 0.f;
            s += v[i];
        }
        const float mean = norm ? wave_sum(s) * invw : 0.f;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < LN_MAX_PER_LANE; ++i) {
            const int c = lane + 64 * i;
            const float dl = (i < per && c < width) ? v[i] - mean : 0.f;
            q += dl * dl;
        }
        const float rstd = norm ? rsqrtf(wave_sum(q) * invw + eps) : 1.f;
        float m1 = 0.f, m2 = 0.f;
#pragma unroll
        for (int i = 0; i < LN_MAX_PER_LANE; ++i) {
            const int c = lane + 64 * i;
            if (i < per && c < width) {
                const float xh = (v[i] - mean) * rstd;
                float g = go[i];
                if (drop_p > 0.f) g *= lob_dropout_scale(seed, (uint64_t)orow * width + c, drop_p);
                if (act == LOB_ACT_GELU) g *= gelu_grad(xh * gm[i] + bt[i]);
                dga[i] += g * xh;
                dba[i] += g;
                const float dxh = g * gm[i];
                v[i] = xh; go[i] = dxh;
                m1 += dxh; m2 += dxh * xh;
            } else { v[i] = 0.f; go[i] = 0.f; }
        }
        m1 = norm ? wave_sum(m1) * invw : 0.f;
        m2 = norm ? wave_sum(m2) * invw : 0.f;
        float* dxr = dx + (size_t)row * width;
#pragma unroll
        for (int i = 0; i < LN_MAX_PER_LANE; ++i) {
            const int c = lane + 64 * i;
            if (i < per && c < width) dxr[c] = rstd * (go[i] - m1 - v[i] * m2);
        }
    }
#pragma unroll
    for (int i = 0; i < LN_MAX_PER_LANE; ++i) {
        const int c = lane + 64 * i;
        if (norm && i < per && c < width) { atomicAdd(dgamma + c, dga[i]); atomicAdd(dbeta + c, dba[i]); }
    }
}


// Vectorised variants for width = 64 * VPL (VPL = 2, 4, 8: widths 128, 256, 512): lane l owns the
// VPL contiguous columns [l*VPL, (l+1)*VPL), so a row is ONE 8/16/32-byte load per lane (a wave reads
// the whole row as one contiguous 512 B..2 KB burst) instead of VPL strided dword loads.
template <int VPL>
__device__ __forceinline__ void ldv(const float* p, float (&v)[VPL]) {
    if (VPL == 2) { const float2 t = *reinterpret_cast<const float2*>(p); v[0] = t.x; v[1] = t.y; }
    else {
#pragma unroll
        for (int i = 0; i < VPL / 4; ++i) {
            const f32x4 t = *reinterpret_cast<const f32x4*>(p + 4 * i);
            v[4 * i] = t[0]; v[4 * i + 1] = t[1]; v[4 * i + 2] = t[2]; v[4 * i + 3] = t[3];
        }
    }
}
template <int VPL>
__device__ __forceinline__ void stv(float* p, const float (&v)[VPL]) {
    if (VPL == 2) { *reinterpret_cast<float2*>(p) = make_float2(v[0], v[1]); }
    else {
#pragma unroll
        for (int i = 0; i < VPL / 4; ++i) {
            f32x4 t = {v[4 * i], v[4 * i + 1], v[4 * i + 2], v[4 * i + 3]};
            *reinterpret_cast<f32x4*>(p + 4 * i) = t;
        }
    }
}

template <int VPL>
__device__ __forceinline__ void stv_bf16(__bf16* p, const float (&v)[VPL]) {
    typedef __bf16 bvec __attribute__((ext_vector_type(VPL)));
    bvec t;
#pragma unroll
    for (int i = 0; i < VPL; ++i) t[i] = (__bf16)v[i];
    *reinterpret_cast<bvec*>(p) = t;
}

// typed row-segment access for the backward kernel: fp32 or bf16 storage, fp32 in registers
template <int VPL>
__device__ __forceinline__ void ldv_bf16(const __bf16* p, float (&v)[VPL]) {
    typedef __bf16 bvec __attribute__((ext_vector_type(VPL)));
    const bvec t = *reinterpret_cast<const bvec*>(p);
#pragma unroll
    for (int i = 0; i < VPL; ++i) v[i] = (float)t[i];
}
template <int VPL, typename E>
__device__ __forceinline__ void ldv_t(const E* p, float (&v)[VPL]) {
    if constexpr (sizeof(E) == 4) ldv<VPL>(reinterpret_cast<const float*>(p), v);
    else                          ldv_bf16<VPL>(reinterpret_cast<const __bf16*>(p), v);
}
template <int VPL, typename E>
__device__ __forceinline__ void stv_t(E* p, const float (&v)[VPL]) {
    if constexpr (sizeof(E) == 4) stv<VPL>(reinterpret_cast<float*>(p), v);
    else                          stv_bf16<VPL>(reinterpret_cast<__bf16*>(p), v);
}

// Row enumeration for the (b,t) -> (t,b) relayout: index g walks 8 x 8 (window, time) tiles, so that 64 consecutive
// waves touch 8 runs of 8 consecutive rows on the (b,t)-ordered side AND 8 runs of 8 consecutive rows on the
// time-major side (a plain row-by-row walk reads one side in 512-B pieces 2 MB apart).
__device__ __forceinline__ bool tiled_row(int g, int T, int B, int Bp, int& row, int& orow) {
    const int ntt = (T + 7) >> 3;
    const int tile = g >> 6, r = g & 63;
    const int b = (tile / ntt) * 8 + (r >> 3), t = (tile % ntt) * 8 + (r & 7);
    if (b >= B || t >= T) return false;
    row = b * T + t;
    orow = t * Bp + b;
    return true;
}
__device__ __forceinline__ int tiled_count(int T, int B) { return ((B + 7) >> 3) * ((T + 7) >> 3) * 64; }

// LPR = lanes per row: 64 (one row per wave pass), 16 (four rows per pass, VPL = 8: width 128 then moves 32 B per
// lane in and 16 B of bf16 out, instead of 8 B / 4 B with 64 lanes on the row) or 32 (two rows per pass, VPL = 8:
// width 256 with bf16 rows in and out -- 16 B per lane and stream instead of 8).
template <int LPR>
__device__ __forceinline__ float row_sum(float v) {
    if (LPR == 64) return wave_sum(v);
#define LOB_DPP_ADD(CTRL)                                                                                   \
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false))
    LOB_DPP_ADD(0xB1);       // quad_perm [1,0,3,2]
    LOB_DPP_ADD(0x4E);       // quad_perm [2,3,0,1]
    LOB_DPP_ADD(0x141);      // row_half_mirror
    LOB_DPP_ADD(0x140);      // row_mirror: every lane of the 16-lane row now holds the row's sum
#undef LOB_DPP_ADD
    if (LPR == 32) v += __shfl_xor(v, 16, 64);      // two 16-lane halves of a 32-lane row
    return v;
}

// XE: storage type of the input rows (LOB_X_BF16: the last LSTM layer's output handed over as bf16 only)
template <int VPL, bool OUT_BF16, int LPR = 64, typename XE = float>
__global__ __launch_bounds__(256) void layernorm_act_vec_kernel(
    const XE* __restrict__ in, const float* __restrict__ gamma, const float* __restrict__ beta,
    void* __restrict__ outv, int rows, float eps, int act,
    int remap_T, int remap_B, int remap_Bp, float drop_p, uint64_t seed) {
    constexpr int width = LPR * VPL, GPW = 64 / LPR;       // GPW rows per wave pass
    const int lane = threadIdx.x & 63, sub = lane / LPR, sl = lane % LPR;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int nwaves = (gridDim.x * blockDim.x) >> 6;
    float gm[VPL], bt[VPL];
    const bool norm = !(act & LOB_LN_IDENTITY);
    act &= 0xff;
    if (norm) { ldv<VPL>(gamma + sl * VPL, gm); ldv<VPL>(beta + sl * VPL, bt); }
    else {
#pragma unroll
        for (int i = 0; i < VPL; ++i) { gm[i] = 1.f; bt[i] = 0.f; }
    }
    const float invw = 1.0f / (float)width;
    const int count = remap_T > 0 ? tiled_count(remap_T, remap_B) : rows;
    // RPW rows per wave in flight: with one 512-B / 1-KB row per wave the CU has too few bytes outstanding to cover the
    // HBM latency (measured 2.1 TB/s at width 128 against 4.6 at width 256 with the same code)
    constexpr int RPW = VPL <= 2 ? 4 : 2;
    for (int g0 = wave * RPW * GPW; g0 < count; g0 += nwaves * RPW * GPW) {
        float vv[RPW][VPL];
        int rowv[RPW], orowv[RPW];
        bool ok[RPW];
#pragma unroll
        for (int r = 0; r < RPW; ++r) {
            const int g = g0 + r * GPW + sub;
            rowv[r] = g; orowv[r] = g;
            ok[r] = g < count && (remap_T <= 0 || tiled_row(g, remap_T, remap_B, remap_Bp, rowv[r], orowv[r]));
            if (ok[r]) ldv_t<VPL, XE>(in + (size_t)rowv[r] * width + sl * VPL, vv[r]);
        }
#pragma unroll
        for (int r = 0; r < RPW; ++r) {
            if (!ok[r]) continue;
            float (&v)[VPL] = vv[r];
            const int orow = orowv[r];
            float s = 0.f;
#pragma unroll
            for (int i = 0; i < VPL; ++i) s += v[i];
            const float mean = norm ? row_sum<LPR>(s) * invw : 0.f;
            float q = 0.f;             // explicit fma: the fused kernels below (input_proj_ln / attn_score) must round alike
#pragma unroll
            for (int i = 0; i < VPL; ++i) { const float dl = v[i] - mean; q = __builtin_fmaf(dl, dl, q); }
            const float rstd = norm ? rsqrtf(__builtin_fmaf(row_sum<LPR>(q), invw, eps)) : 1.f;
            float ds[VPL];              // dropout scales: one hash per pair of neighbouring columns
#pragma unroll
            for (int i = 0; i < VPL; i += 2) {
                if (drop_p > 0.f) lob_dropout_scale2(seed, (uint64_t)orow * width + sl * VPL + i, drop_p, ds[i], ds[i + 1]);
                else { ds[i] = 1.f; ds[i + 1] = 1.f; }
            }
#pragma unroll
            for (int i = 0; i < VPL; ++i) {
                float o = __builtin_fmaf((v[i] - mean) * rstd, gm[i], bt[i]);
                v[i] = apply_act(o, act) * ds[i];
            }
            if (OUT_BF16) stv_bf16<VPL>(reinterpret_cast<__bf16*>(outv) + (size_t)orow * width + sl * VPL, v);
            else          stv<VPL>(reinterpret_cast<float*>(outv) + (size_t)orow * width + sl * VPL, v);
        }
    }
}

template <int N, int W>
__device__ __forceinline__ float red_sum(const float (&r)[N][W], int c) {
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < N; ++i) t += r[i][c];
    return t;
}

// DYE / DXE: storage type of the incoming gradient dy / the outgoing dx (mixed path: the gradient carried between the
// LSTM layers and into / out of the LayerNorms is a bf16 stream; LOB_DY_BF16 / LOB_OUT_BF16)
template <int VPL, int LPR = 64, typename DYE = float, typename DXE = float, typename XE = float>
__global__ __launch_bounds__(256) void layernorm_act_bwd_vec_kernel(
    const XE* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
    const DYE* __restrict__ dy, DXE* __restrict__ dx, float* __restrict__ dgamma, float* __restrict__ dbeta,
    int rows, float eps, int act, int remap_T, int remap_B, int remap_Bp, float drop_p, uint64_t seed,
    const float* __restrict__ pool_attn, const float* __restrict__ pool_dctx, int pool_T, int pool_B, int pool_Bp,
    float* __restrict__ dx_colsum) {
    constexpr int width = LPR * VPL, GPW = 64 / LPR, NRED = 4 * GPW;
    __shared__ float red[2][NRED][width];
    const int lane = threadIdx.x & 63, sub = lane / LPR, sl = lane % LPR, wib = (threadIdx.x >> 6) * GPW + sub;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int nwaves = (gridDim.x * blockDim.x) >> 6;
    float gm[VPL], bt[VPL], dga[VPL], dba[VPL], dxs[VPL];
    const bool norm = !(act & LOB_LN_IDENTITY);
    act &= 0xff;
    if (norm) { ldv<VPL>(gamma + sl * VPL, gm); ldv<VPL>(beta + sl * VPL, bt); }
#pragma unroll
    for (int i = 0; i < VPL; ++i) { dga[i] = 0.f; dba[i] = 0.f; dxs[i] = 0.f; if (!norm) { gm[i] = 1.f; bt[i] = 0.f; } }
    const float invw = 1.0f / (float)width;
    const int count = remap_T > 0 ? tiled_count(remap_T, remap_B) : rows;
    constexpr int RPW = VPL <= 4 ? 4 : 2;       // rows per wave in flight (see the forward kernel)
    for (int g0 = wave * RPW * GPW; g0 < count; g0 += nwaves * RPW * GPW) {
        float vv[RPW][VPL], gov[RPW][VPL];
        int rowv[RPW], orowv[RPW];
        bool ok[RPW];
#pragma unroll
        for (int r = 0; r < RPW; ++r) {
            const int g = g0 + r * GPW + sub;
            rowv[r] = g; orowv[r] = g;
            ok[r] = g < count && (remap_T <= 0 || tiled_row(g, remap_T, remap_B, remap_Bp, rowv[r], orowv[r]));
            if (ok[r]) {
                ldv_t<VPL, XE>(x + (size_t)rowv[r] * width + sl * VPL, vv[r]);
                ldv_t<VPL, DYE>(dy + (size_t)orowv[r] * width + sl * VPL, gov[r]);
            }
        }
#pragma unroll
        for (int r = 0; r < RPW; ++r) {
            if (!ok[r]) continue;
            float (&v)[VPL] = vv[r];
            float (&go)[VPL] = gov[r];
            const int row = rowv[r], orow = orowv[r];
            if (pool_attn) {     // rows are (t, b) time-major: dy += attn[b][t] * dctx[b][:] (Attention's context path)
                const int t = row / pool_Bp, b = row % pool_Bp;
                if (b < pool_B) {
                    const float a = pool_attn[(size_t)b * pool_T + t];
                    float dcv[VPL];
                    ldv<VPL>(pool_dctx + (size_t)b * width + sl * VPL, dcv);
#pragma unroll
                    for (int i = 0; i < VPL; ++i) go[i] = fmaf(a, dcv[i], go[i]);
                }
            }
            float s = 0.f;
#pragma unroll
            for (int i = 0; i < VPL; ++i) s += v[i];
            const float mean = norm ? row_sum<LPR>(s) * invw : 0.f;
            float q = 0.f;             // explicit fma throughout: attn_ln_bwd_kernel below must round alike
#pragma unroll
            for (int i = 0; i < VPL; ++i) { const float dl = v[i] - mean; q = __builtin_fmaf(dl, dl, q); }
            const float rstd = norm ? rsqrtf(__builtin_fmaf(row_sum<LPR>(q), invw, eps)) : 1.f;
            float m1 = 0.f, m2 = 0.f;
            float ds[VPL];
#pragma unroll
            for (int i = 0; i < VPL; i += 2) {
                if (drop_p > 0.f) lob_dropout_scale2(seed, (uint64_t)orow * width + sl * VPL + i, drop_p, ds[i], ds[i + 1]);
                else { ds[i] = 1.f; ds[i + 1] = 1.f; }
            }
#pragma unroll
            for (int i = 0; i < VPL; ++i) {
                const float xh = (v[i] - mean) * rstd;
                float g = go[i] * ds[i];
                if (act == LOB_ACT_GELU) g *= gelu_grad(xh * gm[i] + bt[i]);
                dga[i] = __builtin_fmaf(g, xh, dga[i]);
                dba[i] += g;
                const float dxh = g * gm[i];
                v[i] = xh; go[i] = dxh;
                m1 += dxh; m2 = __builtin_fmaf(dxh, xh, m2);
            }
            m1 = norm ? row_sum<LPR>(m1) * invw : 0.f;
            m2 = norm ? row_sum<LPR>(m2) * invw : 0.f;
#pragma unroll
            for (int i = 0; i < VPL; ++i) { v[i] = rstd * __builtin_fmaf(-v[i], m2, go[i] - m1); dxs[i] += v[i]; }
            stv_t<VPL, DXE>(dx + (size_t)row * width + sl * VPL, v);
        }
    }
    if (dx_colsum) {      // column sums of dx = the bias gradient of the Linear that feeds this LayerNorm (04:174-175)
        __syncthreads();
#pragma unroll
        for (int i = 0; i < VPL; ++i) red[0][wib][sl * VPL + i] = dxs[i];
        __syncthreads();
        for (int c = threadIdx.x; c < width; c += 256)
            atomicAdd(dx_colsum + c, red_sum<NRED, width>(red[0], c));
        __syncthreads();
    }
    // block-level reduction of the affine gradients, then ONE atomic per column per block
#pragma unroll
    for (int i = 0; i < VPL; ++i) { red[0][wib][sl * VPL + i] = dga[i]; red[1][wib][sl * VPL + i] = dba[i]; }
    __syncthreads();
    if (!norm) return;
    for (int c = threadIdx.x; c < width; c += 256) {
        atomicAdd(dgamma + c, red_sum<NRED, width>(red[0], c));
        atomicAdd(dbeta + c, red_sum<NRED, width>(red[1], c));
    }
}

// Backward of attn_pool_fwd_kernel, one workgroup per window.
//   dV[t,b,:]    = a[t] * dctx[b,:]                      (the W1 path is added by a GEMM afterwards)
//   ds[t]        = a[t] * (da[t] - sum_t a da),  da[t] = dctx . V[t,b,:] (+ dattn[b,t]: the gradient that arrives through
//                  the WEIGHTS output of a stand-alone Attention module, 04_lstm_model.py:112-128)
//   dPreU[t,b,j] = ds[t] * w2[j] * (1 - U^2);   dw2[j] += sum_t ds[t] U[t,b,j]
template <typename VE, typename UE>
__global__ __launch_bounds__(256) void attn_pool_bwd_kernel(
    const VE* __restrict__ V, const float* __restrict__ U, const float* __restrict__ attn,
    const float* __restrict__ dctx, const float* __restrict__ w2, float* __restrict__ dV,
    UE* __restrict__ dPreU, float* __restrict__ dw2, int T, int Bp, int W, int W2, const float* __restrict__ dattn) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* a = sm;            // [T]
    float* ds = sm + T;       // [T]  (da, then ds)
    float* dc = sm + 2 * T;   // [W]
    __shared__ float red[4];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int t = tid; t < T; t += 256) a[t] = attn[(size_t)b * T + t];
    for (int c = tid; c < W; c += 256) dc[c] = dctx[(size_t)b * W + c];
    __syncthreads();
    if (!U) {            // mean pooling: the weights do not depend on anything -> dV = a[t] * dctx only
        for (int c = tid; c < W; c += 256) {
            const float g = dc[c];
            float* o = dV + (size_t)b * W + c;
            for (int t = 0; t < T; ++t) o[(size_t)t * Bp * W] = a[t] * g;
        }
        return;
    }
    for (int t = wave; t < T; t += 4) {
        const VE* v = V + ((size_t)t * Bp + b) * W;
        float s = 0.f;
        for (int c = lane; c < W; c += 64) s = fmaf(dc[c], (float)v[c], s);
        s = wave_sum(s);
        if (lane == 0) ds[t] = s + (dattn ? dattn[(size_t)b * T + t] : 0.f);
    }
    __syncthreads();
    float dot = 0.f;
    for (int t = tid; t < T; t += 256) dot = fmaf(a[t], ds[t], dot);
    dot = wave_sum(dot);
    if (lane == 0) red[wave] = dot;
    __syncthreads();
    dot = red[0] + red[1] + red[2] + red[3];
    __syncthreads();
    for (int t = tid; t < T; t += 256) ds[t] = a[t] * (ds[t] - dot);
    __syncthreads();
    if (dV) {      // (NULL: the caller adds a[t] * dctx inside the LayerNorm backward instead)
        for (int c = tid; c < W; c += 256) {
            const float g = dc[c];
            float* o = dV + (size_t)b * W + c;
            for (int t = 0; t < T; ++t) o[(size_t)t * Bp * W] = a[t] * g;
        }
    }
    for (int j = tid; j < W2; j += 256) {
        const float wj = w2[j];
        const float* u = U + (size_t)b * W2 + j;
        UE* o = dPreU + (size_t)b * W2 + j;
        float acc = 0.f;
        for (int t = 0; t < T; ++t) {
            const float uv = u[(size_t)t * Bp * W2];
            acc = fmaf(ds[t], uv, acc);
            o[(size_t)t * Bp * W2] = (UE)(ds[t] * wj * (1.f - uv * uv));
        }
        atomicAdd(dw2 + j, acc);
    }
}


// ------------------------------------------------------------------------------------------
// Vectorised attention pooling for the mixed path at H = 128, D = 2 (V bf16 [.,256], U fp32 [.,128]): the generic
// kernels above walk a window's 256 rows with 2- and 4-byte scalar accesses from one thread per column (1.9 TB/s in
// the backward); here a wave takes every 4th time step and moves whole rows per instruction (V: 8 B per lane, U:
// 8 B per lane, dU: 4 B per lane), partial sums are combined through LDS at the end.
// ------------------------------------------------------------------------------------------
typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));

// F: width factor -- W = 256 F columns of V, W2 = 128 F columns of U (F = 1: H = 128, F = 2: H = 256, both bidirectional):
// a lane holds 4 F consecutive columns of V and 2 F of U
// SCORES: U holds the finished scores S[b][t] (lob_attn_scores_bf16) instead of the score layer's hidden activations
template <int F, bool SCORES = false>
__global__ __launch_bounds__(256) void attn_pool_fwd_vec_kernel(
    const __bf16* __restrict__ V, const float* __restrict__ U, const float* __restrict__ w2,
    const float* __restrict__ b2, float* __restrict__ ctx, float* __restrict__ attn, int T, int Bp) {
    constexpr int W = 256 * F, W2 = 128 * F, NV = 4 * F, NU = 2 * F;
    typedef __bf16 bvec __attribute__((ext_vector_type(NV)));
    extern __shared__ __attribute__((aligned(16))) float sc[];   // [T] scores -> weights, then [4][W] partial contexts
    __shared__ float red[8];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const size_t rs = (size_t)Bp;
    if constexpr (SCORES) {
        for (int t = tid; t < T; t += 256) sc[t] = U[(size_t)b * T + t];
    } else {
        const float bias2 = b2 ? b2[0] : 0.f;
        float wv[NU];
        ldv<NU>(w2 + NU * lane, wv);
#pragma unroll 4
        for (int t = wave; t < T; t += 4) {
            float u[NU];
            ldv<NU>(U + ((size_t)t * rs + b) * W2 + NU * lane, u);
            float s = u[0] * wv[0];
#pragma unroll
            for (int i = 1; i < NU; ++i) s = fmaf(u[i], wv[i], s);
            s = wave_sum(s);
            if (lane == 0) sc[t] = s + bias2;
        }
    }
    __syncthreads();
    float m = -INFINITY;
    for (int t = tid; t < T; t += 256) m = fmaxf(m, sc[t]);
    m = wave_max(m);
    if (lane == 0) red[wave] = m;
    __syncthreads();
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    float l = 0.f;
    for (int t = tid; t < T; t += 256) { const float e = expf(sc[t] - m); sc[t] = e; l += e; }
    l = wave_sum(l);
    if (lane == 0) red[4 + wave] = l;
    __syncthreads();
    const float inv = 1.0f / (red[4] + red[5] + red[6] + red[7]);
    for (int t = tid; t < T; t += 256) { const float a = sc[t] * inv; sc[t] = a; attn[(size_t)b * T + t] = a; }
    __syncthreads();
    float acc[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) acc[i] = 0.f;
#pragma unroll 4
    for (int t = wave; t < T; t += 4) {
        const bvec v = *reinterpret_cast<const bvec*>(V + ((size_t)t * rs + b) * W + NV * lane);
        const float a = sc[t];
#pragma unroll
        for (int i = 0; i < NV; ++i) acc[i] = fmaf(a, (float)v[i], acc[i]);
    }
    float* part = sc + ((T + 3) & ~3);              // [4][W]
#pragma unroll
    for (int i = 0; i < NV; ++i) part[wave * W + NV * lane + i] = acc[i];
    __syncthreads();
#pragma unroll
    for (int c = tid; c < W; c += 256) ctx[(size_t)b * W + c] = part[c] + part[W + c] + part[2 * W + c] + part[3 * W + c];
}

// Backward, fused form (dV is NOT materialised: the a[t] dctx term goes into the LayerNorm backward): writes dPreU
// (bf16) and accumulates dw2.
template <int F>
__global__ __launch_bounds__(256) void attn_pool_bwd_vec_kernel(
    const __bf16* __restrict__ V, const float* __restrict__ U, const float* __restrict__ attn,
    const float* __restrict__ dctx, const float* __restrict__ w2, __bf16* __restrict__ dPreU,
    float* __restrict__ dw2, float* __restrict__ du_colsum, int T, int Bp) {
    constexpr int W = 256 * F, W2 = 128 * F, NV = 4 * F, NU = 2 * F;
    typedef __bf16 bvec __attribute__((ext_vector_type(NV)));
    typedef __bf16 buvec __attribute__((ext_vector_type(NU)));
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* a = sm;                         // [T]
    float* ds = sm + T;                    // [T]  (da, then ds)
    float* part = sm + 2 * T;              // [4][W2] partial dw2
    __shared__ float red[4];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const size_t rs = (size_t)Bp;
    for (int t = tid; t < T; t += 256) a[t] = attn[(size_t)b * T + t];
    float dc[NV];
    ldv<NV>(dctx + (size_t)b * W + NV * lane, dc);
#pragma unroll 4
    for (int t = wave; t < T; t += 4) {
        const bvec v = *reinterpret_cast<const bvec*>(V + ((size_t)t * rs + b) * W + NV * lane);
        float s = dc[0] * (float)v[0];
#pragma unroll
        for (int i = 1; i < NV; ++i) s = fmaf(dc[i], (float)v[i], s);
        s = wave_sum(s);
        if (lane == 0) ds[t] = s;
    }
    __syncthreads();
    float dot = 0.f;
    for (int t = tid; t < T; t += 256) dot = fmaf(a[t], ds[t], dot);
    dot = wave_sum(dot);
    if (lane == 0) red[wave] = dot;
    __syncthreads();
    dot = red[0] + red[1] + red[2] + red[3];
    __syncthreads();
    for (int t = tid; t < T; t += 256) ds[t] = a[t] * (ds[t] - dot);
    __syncthreads();
    float wv[NU], acc[NU], cs[NU];
    ldv<NU>(w2 + NU * lane, wv);
#pragma unroll
    for (int i = 0; i < NU; ++i) { acc[i] = 0.f; cs[i] = 0.f; }
#pragma unroll 4
    for (int t = wave; t < T; t += 4) {
        const size_t ro = ((size_t)t * rs + b) * W2 + NU * lane;
        float u[NU];
        ldv<NU>(U + ro, u);
        const float d = ds[t];
        buvec o;
#pragma unroll
        for (int i = 0; i < NU; ++i) {
            acc[i] = fmaf(d, u[i], acc[i]);
            const float oi = d * wv[i] * (1.f - u[i] * u[i]);
            cs[i] += oi;
            o[i] = (__bf16)oi;
        }
        *reinterpret_cast<buvec*>(dPreU + ro) = o;
    }
#pragma unroll
    for (int i = 0; i < NU; ++i) part[wave * W2 + NU * lane + i] = acc[i];
    __syncthreads();
    if (tid < W2) atomicAdd(dw2 + tid, part[tid] + part[W2 + tid] + part[2 * W2 + tid] + part[3 * W2 + tid]);
    if (du_colsum) {      // column sums of dPreU = the gradient of the score MLP's first bias (04:118)
        __syncthreads();
#pragma unroll
        for (int i = 0; i < NU; ++i) part[wave * W2 + NU * lane + i] = cs[i];
        __syncthreads();
        if (tid < W2) atomicAdd(du_colsum + tid, part[tid] + part[W2 + tid] + part[2 * W2 + tid] + part[3 * W2 + tid]);
    }
}

// ------------------------------------------------------------------------------------------
// Fused input projection of the mixed path at width 128 (round 3): input_proj = Linear(C -> 128) -> LayerNorm ->
// GELU -> Dropout (04_lstm_model.py:173-178), one pass from the fp32 windows to the bf16 time-major activations of
// the first LSTM layer.  Unfused it was three kernels and 1.9 GB: pad + cast of the windows to bf16 (lob_pad_cast_bf16),
// the K = 64 GEMM writing fp32 pre-activations (lob_gemm_nt_bf16), the LayerNorm kernel reading them back.
// A wave owns 32 consecutive (b, t) rows: their C floats each are one contiguous run (16-byte aligned: 128 C bytes per
// tile), copied flat into the wave's LDS block (next tile's copy requested first), picked up as MFMA A fragments
// (bf16(x), zero beyond C), multiplied with the whole weight matrix kept in registers as B fragments (16 x
// v_mfma_f32_32x32x16_bf16: the SAME instruction and k order as the unfused GEMM, so the pre-activations are
// bit-identical), + bias back into the LDS block row-major, and from there the LayerNorm walks them with the unfused
// kernel's lane assignment (16 lanes x 8 columns per row, same reductions, same dropout hash): the activations are
// bit-identical too.  SAVE (training): the bf16 padded windows (the dW GEMM's operand) and the fp32 pre-activations
// (the LayerNorm backward's input) are written on the way; inference writes neither.
// ------------------------------------------------------------------------------------------
typedef __bf16 ip_bf16x8 __attribute__((ext_vector_type(8)));
constexpr int IP_LD = 132;             // fp32 row stride of a wave's output tile in LDS (528 B)

template <bool SAVE>
__global__ __launch_bounds__(256, 2) void input_proj_ln_kernel(
    const float* __restrict__ x, int C, int Cp, const float* __restrict__ W, int ldw, const float* __restrict__ bias,
    const float* __restrict__ gamma, const float* __restrict__ beta, float* __restrict__ pre, __bf16* __restrict__ xb,
    __bf16* __restrict__ out, long rows, int T, int Bp, float eps, int act, float drop_p, uint64_t seed) {
    constexpr int width = 128;
    __shared__ __attribute__((aligned(16))) float tile[4][32 * IP_LD];
    const int lane = threadIdx.x & 63, wib = threadIdx.x >> 6;
    float* tw = tile[wib];
    const int l31 = lane & 31, hi = lane >> 5;
    const int sub = lane >> 4, sl = lane & 15;
    // B fragments: W[32 cb + l31][16 ks + 8 hi + j], zero beyond C
    ip_bf16x8 wf[4][4];
    float bv[4];
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) {
        const float* wrow = W + (size_t)(32 * cb + l31) * ldw;
        bv[cb] = bias ? bias[32 * cb + l31] : 0.f;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int k = 16 * ks + 8 * hi + j;
                wf[cb][ks][j] = (__bf16)(k < C ? wrow[k] : 0.f);
            }
    }
    float gm[8], bt[8];
    const bool norm = !(act & LOB_LN_IDENTITY);
    act &= 0xff;
    if (norm) { ldv<8>(gamma + sl * 8, gm); ldv<8>(beta + sl * 8, bt); }
    else {
#pragma unroll
        for (int i = 0; i < 8; ++i) { gm[i] = 1.f; bt[i] = 0.f; }
    }
    const float invw = 1.0f / (float)width;
    const long ntile = (rows + 31) >> 5;
    const long total = rows * (long)C;                 // floats in x
    const int nch = 8 * C;                             // 16-byte chunks of a full tile (32 rows x C floats)
    const long gw = (long)blockIdx.x * 4 + wib, nw = (long)gridDim.x * 4;

    f32x4 pf[8];                                       // the next tile's chunks: lane + 64 i
    auto fetch = [&](long tl) {
        const long f0 = tl * 32 * C;                   // first float of the tile
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int ch = lane + 64 * i;
            const long f = f0 + 4L * ch;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (ch < nch) {
                if (f + 4 <= total) v = *reinterpret_cast<const f32x4*>(x + f);
                else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) if (f + e < total) v[e] = x[f + e];
                }
            }
            pf[i] = v;
        }
    };
    long tl = gw;
    if (tl < ntile) fetch(tl);
    for (; tl < ntile; tl += nw) {
        const long r0 = tl * 32;
        // the tile's floats, flat, into the wave's LDS block
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int ch = lane + 64 * i;
            if (ch < nch) *reinterpret_cast<f32x4*>(tw + 4 * ch) = pf[i];
        }
        if (tl + nw < ntile) fetch(tl + nw);
        // A fragments: x[r0 + l31][16 ks + 8 hi + j] as bf16, zero beyond C
        f32x16 acc[4];
#pragma unroll
        for (int cb = 0; cb < 4; ++cb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[cb][r] = 0.f;
        ip_bf16x8 af[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int k = 16 * ks + 8 * hi + j;
                af[ks][j] = (__bf16)(k < C ? tw[l31 * C + k] : 0.f);
            }
        if (SAVE && r0 + l31 < rows) {
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
                if (16 * ks + 8 * hi < Cp)
                    *reinterpret_cast<ip_bf16x8*>(xb + (size_t)(r0 + l31) * Cp + 16 * ks + 8 * hi) = af[ks];
        }
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int cb = 0; cb < 4; ++cb)
                acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ks], wf[cb][ks], acc[cb], 0, 0, 0);
        // + bias, row-major into the same LDS block (this wave's reads of it are done: LDS operations of a wave are in order)
#pragma unroll
        for (int cb = 0; cb < 4; ++cb)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                tw[((r & 3) + 8 * (r >> 2) + 4 * hi) * IP_LD + 32 * cb + l31] = acc[cb][r] + bv[cb];
        // LayerNorm + activation + dropout, four rows per pass on 16 lanes each (layernorm_act_vec_kernel<8, true, 16>)
        // (window, time) of a pass's row by increments from one 32-bit division per tile (round 4: `r / T` on the 64-bit row
        // index was ~70 of a pass's ~250 instructions; the entry point refuses rows >= 2^31)
        int bwi = (int)((unsigned)(r0 + sub) / (unsigned)T), tti = (int)((unsigned)(r0 + sub) - (unsigned)bwi * (unsigned)T);
#pragma unroll 2
        for (int ps = 0; ps < 8; ++ps) {
            const int rt = 4 * ps + sub;
            const long r = r0 + rt;
            const int orow = tti * Bp + bwi;
            tti += 4;
            while (tti >= T) { tti -= T; ++bwi; }
            if (r >= rows) continue;
            float v[8];
            {
                const f32x4 a = *reinterpret_cast<const f32x4*>(tw + rt * IP_LD + 8 * sl);
                const f32x4 b = *reinterpret_cast<const f32x4*>(tw + rt * IP_LD + 8 * sl + 4);
                v[0] = a[0]; v[1] = a[1]; v[2] = a[2]; v[3] = a[3]; v[4] = b[0]; v[5] = b[1]; v[6] = b[2]; v[7] = b[3];
                if (SAVE) {
                    *reinterpret_cast<f32x4*>(pre + (size_t)r * width + 8 * sl) = a;
                    *reinterpret_cast<f32x4*>(pre + (size_t)r * width + 8 * sl + 4) = b;
                }
            }
            float s = 0.f;
#pragma unroll
            for (int i = 0; i < 8; ++i) s += v[i];
            const float mean = norm ? row_sum<16>(s) * invw : 0.f;
            float q = 0.f;
#pragma unroll
            for (int i = 0; i < 8; ++i) { const float dl = v[i] - mean; q = __builtin_fmaf(dl, dl, q); }
            const float rstd = norm ? rsqrtf(__builtin_fmaf(row_sum<16>(q), invw, eps)) : 1.f;
            float ds[8];
#pragma unroll
            for (int i = 0; i < 8; i += 2) {
                if (drop_p > 0.f) lob_dropout_scale2(seed, (uint64_t)orow * width + sl * 8 + i, drop_p, ds[i], ds[i + 1]);
                else { ds[i] = 1.f; ds[i + 1] = 1.f; }
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float o = __builtin_fmaf((v[i] - mean) * rstd, gm[i], bt[i]);
                v[i] = apply_act(o, act) * ds[i];
            }
            stv_bf16<8>(out + (size_t)orow * width + sl * 8, v);
        }
    }
}

// (t, b) of the time-major row r = t * Bp + b.  The fused tail kernels below need it for every row (the score / attention
// element of window b at time t); as `r / Bp` on a 64-bit row index that was ~70 instructions per row in kernels whose time
// IS their vector instruction count (round 4).  One 32-bit division per tile and wave, then steps of 1 or 2 rows
// (step <= 32 <= Bp).  The entry points refuse T * Bp >= 2^31.
struct RowTB {
    int t, b;
    __device__ RowTB(long r, int Bp) { t = (int)((unsigned)r / (unsigned)Bp); b = (int)((unsigned)r - (unsigned)t * (unsigned)Bp); }
    __device__ void step(int inc, int Bp) { b += inc; if (b >= Bp) { b -= Bp; ++t; } }
};

// ------------------------------------------------------------------------------------------
// Fused tail of the mixed forward at H = 128 (round 3): post-LSTM LayerNorm (04_lstm_model.py:192) + the attention's
// score layer u = tanh(W1 v + b1), s = w2 . u + b2 (Attention.forward, 04:123-125) in one pass over the last layer's bf16
// output.  Unfused: LayerNorm (Y16 -> v), the K = 256 GEMM (v -> fp32 u), and the pooling kernel reading u and v again:
// 3 KB per row; here 1.5 KB (inference: Y16 in, v out, one score per row) or 2 KB (training: + fp32 u for the backward).
// A workgroup (4 waves) owns 128 consecutive time-major rows:
//   1. wave w normalises rows 32 w .. + 31 with the LayerNorm kernel's lane assignment (32 lanes x 8 columns per row, two
//      rows per pass; all 16 passes' loads issued up front), writes v to HBM and into the shared LDS tile;
//   2. wave w owns the score layer's columns 32 w .. + 31: its 16 B fragments of W1 stay in registers for the whole
//      launch; 4 row blocks x 16 k-steps of v_mfma_f32_32x32x16_bf16 in the unfused GEMM's k order;
//   3. tanh(acc + b1) goes back into the same LDS block row-major (fp32), and wave w forms the scores of rows 32 w ..
//      with the pooling kernel's lane assignment (2 columns per lane, the same wave reduction) -> S[b][t]; training also
//      writes u.
// Same instructions in the same order as the three kernels it replaces: v, u and the scores are bit-identical.
// ------------------------------------------------------------------------------------------
constexpr int AS_LDA = 264;            // bf16 row stride of the v tile (528 B)
constexpr int AS_LDU = 132;            // fp32 row stride of the u tile (528 B): the two tiles share the LDS block

template <bool SAVE>
__global__ __launch_bounds__(256, 2) void attn_score_kernel(
    const __bf16* __restrict__ Y, const float* __restrict__ gamma, const float* __restrict__ beta,
    const __bf16* __restrict__ W1, const float* __restrict__ b1, const float* __restrict__ w2, const float* __restrict__ b2,
    __bf16* __restrict__ V, float* __restrict__ U, float* __restrict__ S, int T, int B, int Bp, float eps) {
    constexpr int W = 256, W2 = 128;
    __shared__ __attribute__((aligned(16))) float lds[128 * AS_LDU];
    __bf16* at = reinterpret_cast<__bf16*>(lds);
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    // B fragments of this wave's 32 score columns: W1[32 w + l31][16 ks + 8 hi + j]
    ip_bf16x8 wf[16];
#pragma unroll
    for (int ks = 0; ks < 16; ++ks)
        wf[ks] = *reinterpret_cast<const ip_bf16x8*>(W1 + (size_t)(32 * w + l31) * W + 16 * ks + 8 * hi);
    const float b1v = b1 ? b1[32 * w + l31] : 0.f;
    float gm[8], bt[8];
    ldv<8>(gamma + l31 * 8, gm);
    ldv<8>(beta + l31 * 8, bt);
    float w2v[2];
    ldv<2>(w2 + 2 * lane, w2v);
    const float bias2 = b2 ? b2[0] : 0.f;
    const long rows = (long)T * Bp;
    const long ntile = (rows + 127) >> 7;
    const float invw = 1.0f / (float)W;
    // The rows run one tile AHEAD (round 4): pass p's registers are re-requested for the next tile as soon as the pass has
    // consumed them, so a tile no longer opens with an HBM round trip.  Rows past the end: clamped address, zero value.
    ip_bf16x8 raw[16];
    auto load_row = [&](long tl, int p) {
        const long r = tl * 128 + 32 * w + 2 * p + hi;
        raw[p] = *reinterpret_cast<const ip_bf16x8*>(Y + (size_t)(r < rows ? r : rows - 1) * W + l31 * 8);
    };
    if ((long)blockIdx.x < ntile) {
#pragma unroll
        for (int p = 0; p < 16; ++p) load_row(blockIdx.x, p);
    }
    for (long tl = blockIdx.x; tl < ntile; tl += gridDim.x) {
        const long r0 = tl * 128 + 32 * w;
        const long nx = tl + gridDim.x < ntile ? tl + gridDim.x : tl;
        // ---- 1. LayerNorm of this wave's 32 rows (two per pass on 32 lanes each)
#pragma unroll
        for (int p = 0; p < 16; ++p) {
            const long r = r0 + 2 * p + hi;
            float v[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = r < rows ? (float)raw[p][i] : 0.f;
            load_row(nx, p);
            float s = 0.f;
#pragma unroll
            for (int i = 0; i < 8; ++i) s += v[i];
            const float mean = row_sum<32>(s) * invw;
            float q = 0.f;
#pragma unroll
            for (int i = 0; i < 8; ++i) { const float dl = v[i] - mean; q = __builtin_fmaf(dl, dl, q); }
            const float rstd = rsqrtf(__builtin_fmaf(row_sum<32>(q), invw, eps));
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = __builtin_fmaf((v[i] - mean) * rstd, gm[i], bt[i]);
            __bf16* arow = at + (32 * w + 2 * p + hi) * AS_LDA + l31 * 8;
            stv_bf16<8>(arow, v);                              // the LayerNorm kernel's own conversion
            if (r < rows) *reinterpret_cast<ip_bf16x8*>(V + (size_t)r * W + l31 * 8) = *reinterpret_cast<const ip_bf16x8*>(arow);
        }
        __syncthreads();
        // ---- 2. pre-activations of the score layer: this wave's 32 columns for all 128 rows
        f32x16 acc[4];
#pragma unroll
        for (int rb = 0; rb < 4; ++rb)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[rb][i] = 0.f;
#pragma unroll
        for (int ks = 0; ks < 16; ++ks)
#pragma unroll
            for (int rb = 0; rb < 4; ++rb) {
                const ip_bf16x8 a = *reinterpret_cast<const ip_bf16x8*>(at + (32 * rb + l31) * AS_LDA + 16 * ks + 8 * hi);
                acc[rb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, wf[ks], acc[rb], 0, 0, 0);
            }
        __syncthreads();                       // every wave is done with the v tile: the block now takes u (fp32)
#pragma unroll
        for (int rb = 0; rb < 4; ++rb)
#pragma unroll
            for (int i = 0; i < 16; ++i)
                lds[(32 * rb + (i & 3) + 8 * (i >> 2) + 4 * hi) * AS_LDU + 32 * w + l31] = fast_tanh(acc[rb][i] + b1v);
        __syncthreads();
        // ---- 3. scores of this wave's 32 rows (the pooling kernel's order: two columns per lane, wave reduction)
        RowTB tb(r0, Bp);
        for (int rr = 0; rr < 32; ++rr, tb.step(1, Bp)) {
            const long r = r0 + rr;
            if (r >= rows) break;
            const float u0 = lds[(32 * w + rr) * AS_LDU + 2 * lane], u1 = lds[(32 * w + rr) * AS_LDU + 2 * lane + 1];
            if (SAVE) { float uu[2] = {u0, u1}; stv<2>(U + (size_t)r * W2 + 2 * lane, uu); }
            float sc = u0 * w2v[0];
            sc = fmaf(u1, w2v[1], sc);
            sc = wave_sum(sc);
            if (lane == 0 && tb.b < B) S[(size_t)tb.b * T + tb.t] = sc + bias2;
        }
        __syncthreads();                       // the block is free for the next tile's v
    }
}

// ------------------------------------------------------------------------------------------
// Fused tail of the mixed BACKWARD at H = 128 (round 3): dV = dU W1 (the score layer's input gradient, 04:123) and the
// post-LSTM LayerNorm's backward (04:192) with the attention's context term a[t] dctx folded in, in one pass.  Unfused,
// the K = 128 GEMM wrote dV (bf16, 512 B per row) and the LayerNorm backward read it back.  A workgroup owns 128
// time-major rows (8 waves): (0) its dU tile (128 x 128 bf16) into LDS; (1) each wave 32 of dV's 256 columns: 8 B
// fragments of W1^T in registers for the whole launch, 4 row blocks x 8 k-steps of v_mfma_f32_32x32x16_bf16; (2) dV, rounded
// to bf16 as the unfused GEMM stores it, row-major into the same LDS block; (3) wave w runs the LayerNorm backward of
// rows 16 w .. + 15 with the unfused kernel's lane assignment and arithmetic (32 lanes x 8 columns, two rows per pass):
// dx equals the unfused pair's except where a dV element rounds the other way (another k order: ~1e-5 of the elements);
// dgamma / dbeta are summed in another order (they go through fp32 atomics either way).
// ------------------------------------------------------------------------------------------
constexpr int AB_LDA = 136;            // bf16 row stride of the dU tile (272 B = 17 x 16 B)
constexpr int AB_LDV = 264;            // bf16 row stride of the dV tile (528 B)

__global__ __launch_bounds__(512) void attn_ln_bwd_kernel(
    const __bf16* __restrict__ X, const float* __restrict__ gamma, const float* __restrict__ beta,
    const __bf16* __restrict__ dU, const __bf16* __restrict__ W1T, __bf16* __restrict__ dX,
    float* __restrict__ dgamma, float* __restrict__ dbeta, const float* __restrict__ attn, const float* __restrict__ dctx,
    int T, int B, int Bp, float eps) {
    constexpr int W = 256, W2 = 128;
    // EIGHT waves: wave w owns 32 of dV's columns (8 B fragments + 64 accumulator registers: two waves per SIMD fit) and
    // the LayerNorm backward of 16 of the tile's rows.
    // Round 4: the kernel ran at 2.8 TB/s with one workgroup per CU in lock-step phases, every phase opening with an exposed
    // round trip (dU at the top of a tile, the attention weight and the dctx row of every pass).  Now
    //  * a tile is 16 windows x 8 time steps instead of 128 consecutive rows: a wave's lane half keeps ONE window for the
    //    tile's eight passes, so its dctx row and its eight attention weights are loaded once per tile, not per pass;
    //  * every load runs one tile ahead: the LayerNorm input of pass p is re-requested for the NEXT tile as soon as pass p
    //    has consumed it (same registers), the next tile's dU rows, dctx row and attention weights during this tile's passes;
    //  * the dU tile has an LDS region of its own: two barriers per tile instead of four.
    // dx is bit-identical to the old kernel's (same arithmetic per row); dgamma / dbeta sum the rows in another order.
    __shared__ __attribute__((aligned(16))) __bf16 lds[128 * AB_LDV];          // dV tile (the affine partials at the end)
    __shared__ __attribute__((aligned(16))) __bf16 ldu[128 * AB_LDA];          // dU tile
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    // B fragments of this wave's 32 dV columns: W1^T[32 w + l31][16 ks + 8 hi + j]
    ip_bf16x8 wf[8];
#pragma unroll
    for (int ks = 0; ks < 8; ++ks)
        wf[ks] = *reinterpret_cast<const ip_bf16x8*>(W1T + (size_t)(32 * w + l31) * W2 + 16 * ks + 8 * hi);
    float gm[8], dga[8], dba[8];
    ldv<8>(gamma + l31 * 8, gm);
#pragma unroll
    for (int i = 0; i < 8; ++i) { dga[i] = 0.f; dba[i] = 0.f; }
    const int nbb = Bp >> 4;                             // window blocks (Bp is a multiple of 32)
    const long ntile = (long)((T + 7) >> 3) * nbb;
    const float invw = 1.0f / (float)W;
    // tile -> (first time step, first window); rows of the tile: (t0 + p) * Bp + b0 + j, p < 8, j < 16; local index 16 p + j
    auto origin = [&](long tl, int& t0, int& b0) {
        const int tb = (int)((unsigned)tl / (unsigned)nbb);
        t0 = 8 * tb;
        b0 = 16 * ((int)tl - tb * nbb);
    };
    auto row_of = [&](int t0, int b0, int p, int j) -> size_t {      // clamped to the last time step (results masked)
        const int t = t0 + p < T ? t0 + p : T - 1;
        return (size_t)t * Bp + b0 + j;
    };
    ip_bf16x8 du[4], xr[8];
    float dcv[8], av[8];
    // wave w brings in the dU rows of time step t0 + w (16 consecutive rows, 4 KB): lane -> (row l / 16 + 4 i, chunk l % 16)
    auto load_du = [&](int t0, int b0) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
            du[i] = *reinterpret_cast<const ip_bf16x8*>(dU + row_of(t0, b0, w, (lane >> 4) + 4 * i) * W2 + 8 * (lane & 15));
    };
    // this lane half's window: b0 + 2 w + hi; its dctx row and its eight attention weights (0 for a padding window)
    auto load_ctx = [&](int t0, int b0) {
        const int b = b0 + 2 * w + hi;
        const int bb = b < B ? b : 0;
        ldv<8>(dctx + (size_t)bb * W + l31 * 8, dcv);
#pragma unroll
        for (int p = 0; p < 8; ++p) av[p] = attn[(size_t)bb * T + (t0 + p < T ? t0 + p : T - 1)];
    };
    auto load_x = [&](int t0, int b0, int p) {
        xr[p] = *reinterpret_cast<const ip_bf16x8*>(X + row_of(t0, b0, p, 2 * w + hi) * W + l31 * 8);
    };
    if ((long)blockIdx.x < ntile) {
        int t0, b0;
        origin(blockIdx.x, t0, b0);
        load_du(t0, b0);
        load_ctx(t0, b0);
#pragma unroll
        for (int p = 0; p < 8; ++p) load_x(t0, b0, p);
    }
    for (long tl = blockIdx.x; tl < ntile; tl += gridDim.x) {
        int t0, b0, nt0, nb0;
        origin(tl, t0, b0);
        {
            const long nx = tl + gridDim.x;
            origin(nx < ntile ? nx : tl, nt0, nb0);
        }
        // ---- 0. dU rows (requested a tile ago) into their LDS tile
#pragma unroll
        for (int i = 0; i < 4; ++i)
            *reinterpret_cast<ip_bf16x8*>(ldu + (16 * w + (lane >> 4) + 4 * i) * AB_LDA + 8 * (lane & 15)) = du[i];
        __syncthreads();
        load_du(nt0, nb0);
        // ---- 1. dV = dU W1: this wave's 32 columns for all 128 rows
        f32x16 acc[4];
#pragma unroll
        for (int rb = 0; rb < 4; ++rb)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[rb][i] = 0.f;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks)
#pragma unroll
            for (int rb = 0; rb < 4; ++rb) {
                const ip_bf16x8 a = *reinterpret_cast<const ip_bf16x8*>(ldu + (32 * rb + l31) * AB_LDA + 16 * ks + 8 * hi);
                // operands SWAPPED like the unfused weight-stationary GEMM: D[n][r] -- this lane holds row r = l31 of the
                // block, columns n = (i & 3) + 8 (i >> 2) + 4 hi
                acc[rb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[ks], a, acc[rb], 0, 0, 0);
            }
        typedef __bf16 ab_bf16x4 __attribute__((ext_vector_type(4)));
#pragma unroll
        for (int rb = 0; rb < 4; ++rb)
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {
                ab_bf16x4 pk = {(__bf16)acc[rb][4 * q4], (__bf16)acc[rb][4 * q4 + 1], (__bf16)acc[rb][4 * q4 + 2],
                                (__bf16)acc[rb][4 * q4 + 3]};
                *reinterpret_cast<ab_bf16x4*>(lds + (32 * rb + l31) * AB_LDV + 32 * w + 8 * q4 + 4 * hi) = pk;
            }
        __syncthreads();
        // ---- 2. LayerNorm backward of this lane half's window, time steps t0 .. t0 + 7
        //         (layernorm_act_bwd_vec_kernel<8, 32, bf16, bf16, bf16>'s arithmetic)
        const bool real = b0 + 2 * w + hi < B;
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            if (t0 + p < T) {
                float v[8], go[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = (float)xr[p][i];
                ldv_bf16<8>(lds + (16 * p + 2 * w + hi) * AB_LDV + l31 * 8, go);
                if (real) {      // context path of the attention pooling: dy += attn[b][t] * dctx[b][:]
                    const float a = av[p];
#pragma unroll
                    for (int i = 0; i < 8; ++i) go[i] = fmaf(a, dcv[i], go[i]);
                }
                float s = 0.f;
#pragma unroll
                for (int i = 0; i < 8; ++i) s += v[i];
                const float mean = row_sum<32>(s) * invw;
                float q = 0.f;
#pragma unroll
                for (int i = 0; i < 8; ++i) { const float dl = v[i] - mean; q = __builtin_fmaf(dl, dl, q); }
                const float rstd = rsqrtf(__builtin_fmaf(row_sum<32>(q), invw, eps));
                float m1 = 0.f, m2 = 0.f;
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const float xh = (v[i] - mean) * rstd;
                    const float g = go[i];
                    dga[i] = __builtin_fmaf(g, xh, dga[i]);
                    dba[i] += g;
                    const float dxh = g * gm[i];
                    v[i] = xh; go[i] = dxh;
                    m1 += dxh; m2 = __builtin_fmaf(dxh, xh, m2);
                }
                m1 = row_sum<32>(m1) * invw;
                m2 = row_sum<32>(m2) * invw;
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = rstd * __builtin_fmaf(-v[i], m2, go[i] - m1);
                stv_bf16<8>(dX + ((size_t)(t0 + p) * Bp + b0 + 2 * w + hi) * W + l31 * 8, v);
            }
            load_x(nt0, nb0, p);                 // the same pass of the next tile, into the registers just released
        }
        load_ctx(nt0, nb0);
        // no barrier here: the next tile's dU writes go to `ldu`, last read before the barrier above; its dV writes come
        // after its own first barrier, which every wave reaches only when it has finished reading this tile's dV rows
    }
    __syncthreads();
    // block-level reduction of the affine gradients (in the tile's LDS block), then ONE atomic per column per block
    float* red = reinterpret_cast<float*>(lds);          // [2][16][W] floats = 32 KB
    const int wib = 2 * w + hi;
#pragma unroll
    for (int i = 0; i < 8; ++i) { red[(0 * 16 + wib) * W + l31 * 8 + i] = dga[i]; red[(1 * 16 + wib) * W + l31 * 8 + i] = dba[i]; }
    __syncthreads();
    if (tid < W) {
        float sg = 0.f, sb = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) { sg += red[(0 * 16 + k) * W + tid]; sb += red[(1 * 16 + k) * W + tid]; }
        atomicAdd(dgamma + tid, sg);
        atomicAdd(dbeta + tid, sb);
    }
}

// ------------------------------------------------------------------------------------------
// Fused head of the mixed BACKWARD at width 128 (round 3): backward of input_proj's LayerNorm + GELU + dropout
// (04_lstm_model.py:175-177) AND the Linear's weight gradient dW = dpre^T xb (04:174) in one pass: dpre (the gradient
// w.r.t. the Linear's output) never goes to HBM.  Unfused, the LayerNorm backward wrote it (bf16, 256 B per row) and the TN
// GEMM read it back next to the bf16 windows.  A wave owns 32 consecutive (b, t) rows: LayerNorm backward with the
// unfused kernel's lane assignment and arithmetic (16 lanes x 8 columns, four rows per pass) -> bf16(dpre) TRANSPOSED into
// the wave's LDS block, the rows' padded bf16 windows transposed next to it, then dW[128][64] += dpre^T xb as 8 blocks x
// 2 k-steps of v_mfma_f32_32x32x16_bf16 into accumulators that live across the wave's tiles; at the end the four waves'
// accumulators are summed through LDS and added to dW with one atomic per element and workgroup.  dgamma, dbeta and the
// bias gradient (column sums of the fp32 dpre) as in the unfused kernel.
// ------------------------------------------------------------------------------------------
constexpr int HB_LDD = 160;            // bf16 row stride of the dpre tile ([row][128 columns + pad]: 320 B, see gemm_bf16.hip's LDK)
constexpr int HB_LDX = 96;             // bf16 row stride of the window tile ([row][64 + pad]: 192 B -- four rows 64 B apart mod 256)

typedef __bf16 hb_bf16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) hb_bf16x4 hb_lds_bf16x4;
// MFMA fragment of the 32-column block starting at column `cb`, k-step s (16 rows) of a [row][column] LDS image: gfx950's
// transposing LDS read hands every lane 8 consecutive ROWS of one column (gemm_bf16.hip: tr_frag)
template <int LD>
__device__ __forceinline__ ip_bf16x8 hb_tr_frag(const __bf16* S, int cb, int ks, int lane) {
    const int h = lane >> 5, mh = (lane >> 4) & 1, q = (lane >> 2) & 3, p = lane & 3;
    const __bf16* a = S + (16 * ks + 8 * h + q) * LD + cb + 16 * mh + 4 * p;
    const hb_bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((hb_lds_bf16x4*)a);
    const hb_bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((hb_lds_bf16x4*)(a + 4 * LD));
    ip_bf16x8 f = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return f;
}

__global__ __launch_bounds__(256, 2) void input_proj_bwd_kernel(
    const float* __restrict__ pre, const float* __restrict__ gamma, const float* __restrict__ beta,
    const __bf16* __restrict__ dA, const __bf16* __restrict__ xb, int Cp, float* __restrict__ dW, int lddw,
    float* __restrict__ dgamma, float* __restrict__ dbeta, float* __restrict__ dbias,
    long rows, int T, int Bp, float eps, int act, float drop_p, uint64_t seed) {
    constexpr int width = 128;
    constexpr int WAVE_LDS = 32 * HB_LDD + 32 * HB_LDX;       // bf16 elements per wave: dpre tile + window tile = 16 KB
    __shared__ __attribute__((aligned(16))) __bf16 tt[4 * WAVE_LDS];
    const int tid = threadIdx.x, lane = tid & 63, wib = tid >> 6;
    __bf16* dt = tt + wib * WAVE_LDS;
    __bf16* xt = dt + 32 * HB_LDD;
    const int l31 = lane & 31, hi = lane >> 5;
    const int sub = lane >> 4, sl = lane & 15;
    float gm[8], bt[8], dga[8], dba[8], dxs[8];
    const bool norm = !(act & LOB_LN_IDENTITY);
    act &= 0xff;
    if (norm) { ldv<8>(gamma + sl * 8, gm); ldv<8>(beta + sl * 8, bt); }
#pragma unroll
    for (int i = 0; i < 8; ++i) { dga[i] = 0.f; dba[i] = 0.f; dxs[i] = 0.f; if (!norm) { gm[i] = 1.f; bt[i] = 0.f; } }
    f32x16 acc[4][2];              // dW blocks [32 nb .. + 31][32 kb .. + 31]
#pragma unroll
    for (int nb = 0; nb < 4; ++nb)
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[nb][kb][i] = 0.f;
    // window columns beyond Cp: zero once (their products must be zero, not stale LDS)
    for (int i = lane; i < 32 * HB_LDX; i += 64) xt[i] = (__bf16)0.f;
    const float invw = 1.0f / (float)width;
    const long ntile = (rows + 31) >> 5;
    const long gw = (long)blockIdx.x * 4 + wib, nw = (long)gridDim.x * 4;
    const int cpc = Cp >> 3;                                   // 16-byte chunks per window row
    for (long tl = gw; tl < ntile; tl += nw) {
        const long r0 = tl * 32;
        // ---- the tile's padded bf16 windows as they are: xt[r][k]
        for (int i = lane; i < 32 * cpc; i += 64) {
            const int rr = i / cpc, ch = i - rr * cpc;
            ip_bf16x8 z;
#pragma unroll
            for (int e = 0; e < 8; ++e) z[e] = (__bf16)0.f;
            if (r0 + rr < rows) z = *reinterpret_cast<const ip_bf16x8*>(xb + (size_t)(r0 + rr) * Cp + 8 * ch);
            *reinterpret_cast<ip_bf16x8*>(xt + rr * HB_LDX + 8 * ch) = z;
        }
        // ---- LayerNorm backward, four rows per pass (layernorm_act_bwd_vec_kernel<8, 16, bf16, ., float>)
#pragma unroll 2
        for (int ps = 0; ps < 8; ++ps) {
            const int rt = 4 * ps + sub;
            const long r = r0 + rt;
            float v[8], go[8];
            if (r < rows) {
                const long bw = r / T;
                const long orow = (r - bw * T) * Bp + bw;
                ldv<8>(pre + (size_t)r * width + sl * 8, v);
                ldv_bf16<8>(dA + (size_t)orow * width + sl * 8, go);
                float s = 0.f;
#pragma unroll
                for (int i = 0; i < 8; ++i) s += v[i];
                const float mean = norm ? row_sum<16>(s) * invw : 0.f;
                float q = 0.f;
#pragma unroll
                for (int i = 0; i < 8; ++i) { const float dl = v[i] - mean; q = __builtin_fmaf(dl, dl, q); }
                const float rstd = norm ? rsqrtf(__builtin_fmaf(row_sum<16>(q), invw, eps)) : 1.f;
                float m1 = 0.f, m2 = 0.f;
                float ds[8];
#pragma unroll
                for (int i = 0; i < 8; i += 2) {
                    if (drop_p > 0.f) lob_dropout_scale2(seed, (uint64_t)orow * width + sl * 8 + i, drop_p, ds[i], ds[i + 1]);
                    else { ds[i] = 1.f; ds[i + 1] = 1.f; }
                }
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const float xh = (v[i] - mean) * rstd;
                    float g = go[i] * ds[i];
                    if (act == LOB_ACT_GELU) g *= gelu_grad(xh * gm[i] + bt[i]);
                    dga[i] = __builtin_fmaf(g, xh, dga[i]);
                    dba[i] += g;
                    const float dxh = g * gm[i];
                    v[i] = xh; go[i] = dxh;
                    m1 += dxh; m2 = __builtin_fmaf(dxh, xh, m2);
                }
                m1 = norm ? row_sum<16>(m1) * invw : 0.f;
                m2 = norm ? row_sum<16>(m2) * invw : 0.f;
#pragma unroll
                for (int i = 0; i < 8; ++i) { v[i] = rstd * __builtin_fmaf(-v[i], m2, go[i] - m1); dxs[i] += v[i]; }
            } else {
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = 0.f;
            }
            stv_bf16<8>(dt + rt * HB_LDD + 8 * sl, v);        // dpre[row][columns], bf16 as the unfused GEMM reads it
        }
        // ---- dW += dpre^T xb: A = dpre tile (block of 32 output rows n), B = window tile (block of 32 output columns k)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const ip_bf16x8 b0 = hb_tr_frag<HB_LDX>(xt, 0, ks, lane), b1 = hb_tr_frag<HB_LDX>(xt, 32, ks, lane);
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) {
                const ip_bf16x8 af = hb_tr_frag<HB_LDD>(dt, 32 * nb, ks, lane);
                acc[nb][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, b0, acc[nb][0], 0, 0, 0);
                acc[nb][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, b1, acc[nb][1], 0, 0, 0);
            }
        }
    }
    // ---- workgroup reductions in the tiles' LDS block: dW (four waves' accumulators), then dgamma / dbeta / dbias
    __syncthreads();
    float* wsum = reinterpret_cast<float*>(tt);                 // [128][64] fp32 = 32 KB
    float* red = wsum + 128 * 64;                               // [3][16][128] fp32 = 24 KB  (block: 64 KB)
    for (int i = tid; i < 128 * 64; i += 256) wsum[i] = 0.f;
    const int wr = wib * 4 + sub;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        red[(0 * 16 + wr) * width + sl * 8 + i] = dga[i];
        red[(1 * 16 + wr) * width + sl * 8 + i] = dba[i];
        red[(2 * 16 + wr) * width + sl * 8 + i] = dxs[i];
    }
    __syncthreads();
#pragma unroll
    for (int nb = 0; nb < 4; ++nb)
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int i = 0; i < 16; ++i)
                atomicAdd(&wsum[(32 * nb + (i & 3) + 8 * (i >> 2) + 4 * hi) * 64 + 32 * kb + l31], acc[nb][kb][i]);
    __syncthreads();
    for (int i = tid; i < 128 * Cp; i += 256) {
        const int n = i / Cp, k = i - n * Cp;
        atomicAdd(dW + (size_t)n * lddw + k, wsum[n * 64 + k]);
    }
    if (tid < width) {
        float s0 = 0.f, s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            s0 += red[(0 * 16 + k) * width + tid]; s1 += red[(1 * 16 + k) * width + tid]; s2 += red[(2 * 16 + k) * width + tid];
        }
        if (norm) { atomicAdd(dgamma + tid, s0); atomicAdd(dbeta + tid, s1); }
        if (dbias) atomicAdd(dbias + tid, s2);
    }
}

// ------------------------------------------------------------------------------------------
// The forward tail of the FP32 path at H = 128 (round 4): post-LSTM LayerNorm (04:192) + score layer u = tanh(W1 v + b1),
// s = w2 . u + b2 (04:123-125) in one pass over the last layer's fp32 output.  Unfused (layernorm_act_vec_kernel<4, false, 64>,
// the exact-fp32 NT GEMM with its tanh epilogue, attn_pool_fwd_kernel<float> reading u and v): 5 KB per row; here 2 KB
// (inference: Y in, v out) or 2.5 KB (training: + u for the backward), and the pooling reads v once more.
// The LayerNorm is the unfused kernel's, lane for lane (64 lanes x 4 columns, one row per pass): v is bit-identical.  The
// score layer runs on the 16-bit matrix pipe like the fp32 path's gate GEMMs: both operands as two-way fp16 splits
// (hi = fp16(x 2^k), lo = fp16((x 2^k - hi) 2^11); hi hi and the two cross terms in separate fp32 accumulators: 22-bit
// products), W1's halves in registers for the whole launch (128 VGPRs), v's halves written to LDS as the rows are
// normalised.  Pre-scales on the device: |v| <= 16 max|gamma| + max|beta| (|x_hat| < sqrt(256)), max|W1| by a reduction in
// every workgroup's prologue (128 KB out of L2).  A workgroup (4 waves) owns 64 consecutive time-major rows; wave w
// normalises rows 16 w .. + 15 and owns score columns 32 w .. + 31.
// ------------------------------------------------------------------------------------------
typedef _Float16 af_f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 af_f16x4 __attribute__((ext_vector_type(4)));
constexpr int AF_LDA = 264;            // fp16 row stride of the v images (528 B)
constexpr int AF_LDU = 132;            // fp32 row stride of the u tile (528 B)
constexpr float AF_LO = 2048.f;        // residual scale 2^11

template <bool SAVE>
__global__ __launch_bounds__(256, 2) void attn_score_f32_kernel(
    const float* __restrict__ Y, const float* __restrict__ gamma, const float* __restrict__ beta,
    const float* __restrict__ W1, const float* __restrict__ b1, const float* __restrict__ w2, const float* __restrict__ b2,
    float* __restrict__ V, float* __restrict__ U, float* __restrict__ S, int T, int B, int Bp, float eps) {
    constexpr int W = 256, W2 = 128;
    __shared__ __attribute__((aligned(16))) _Float16 img[2 * 64 * AF_LDA];    // v hi | v lo; the fp32 u tile aliases it
    __shared__ float red[12];
    float* ut = reinterpret_cast<float*>(img);
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hi = lane >> 5;
    // ---- operand ranges -> power-of-two pre-scales (every workgroup derives the same values)
    {
        float gmx = fabsf(gamma[tid]), bmx = fabsf(beta[tid]), wmx = 0.f;
        for (int i = tid; i < W2 * W; i += 256) wmx = fmaxf(wmx, fabsf(W1[i]));
        gmx = wave_max(gmx); bmx = wave_max(bmx); wmx = wave_max(wmx);
        if (lane == 0) { red[w] = gmx; red[4 + w] = bmx; red[8 + w] = wmx; }
        __syncthreads();
    }
    const float amax_v = 16.f * fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])) + fmaxf(fmaxf(red[4], red[5]), fmaxf(red[6], red[7]));
    const float sv = lob_split_scale(amax_v), sw = lob_split_scale(fmaxf(fmaxf(red[8], red[9]), fmaxf(red[10], red[11])));
    const float r_hh = 1.f / (sv * sw), r_sm = r_hh * (1.f / AF_LO);
    // B fragments of this wave's 32 score columns, split: W1[32 w + l31][16 ks + 8 hi + j]
    af_f16x8 whi[16], wlo[16];
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) {
        const float* src = W1 + (size_t)(32 * w + l31) * W + 16 * ks + 8 * hi;
        const f32x4 a = *reinterpret_cast<const f32x4*>(src), b = *reinterpret_cast<const f32x4*>(src + 4);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float x = (j < 4 ? a[j & 3] : b[j & 3]) * sw;
            const _Float16 h = (_Float16)x;
            whi[ks][j] = h;
            wlo[ks][j] = (_Float16)((x - (float)h) * AF_LO);
        }
    }
    const float b1v = b1 ? b1[32 * w + l31] : 0.f;
    float gm[4], bt[4];
    ldv<4>(gamma + lane * 4, gm);
    ldv<4>(beta + lane * 4, bt);
    const float w2a = w2[lane], w2b = w2[lane + 64];
    const float bias2 = b2 ? b2[0] : 0.f;
    const long rows = (long)T * Bp;
    const long ntile = (rows + 63) >> 6;
    const float invw = 1.0f / (float)W;
    for (long tl = blockIdx.x; tl < ntile; tl += gridDim.x) {
        const long r0 = tl * 64 + 16 * w;
        // ---- 1. LayerNorm of this wave's 16 rows (layernorm_act_vec_kernel<4, false, 64, float>), all loads up front
        f32x4 raw[16];
#pragma unroll
        for (int p = 0; p < 16; ++p) {
            const long r = r0 + p < rows ? r0 + p : rows - 1;
            raw[p] = *reinterpret_cast<const f32x4*>(Y + (size_t)r * W + lane * 4);
        }
#pragma unroll
        for (int p = 0; p < 16; ++p) {
            const long r = r0 + p;
            float v[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = r < rows ? raw[p][i] : 0.f;
            float s = 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i) s += v[i];
            const float mean = row_sum<64>(s) * invw;
            float q = 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i) { const float dl = v[i] - mean; q = __builtin_fmaf(dl, dl, q); }
            const float rstd = rsqrtf(__builtin_fmaf(row_sum<64>(q), invw, eps));
            af_f16x4 h4, l4;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                v[i] = __builtin_fmaf((v[i] - mean) * rstd, gm[i], bt[i]);
                const float x = v[i] * sv;
                const _Float16 h = (_Float16)x;
                h4[i] = h;
                l4[i] = (_Float16)((x - (float)h) * AF_LO);
            }
            if (r < rows) stv<4>(V + (size_t)r * W + lane * 4, v);
            *reinterpret_cast<af_f16x4*>(img + (16 * w + p) * AF_LDA + lane * 4) = h4;
            *reinterpret_cast<af_f16x4*>(img + 64 * AF_LDA + (16 * w + p) * AF_LDA + lane * 4) = l4;
        }
        __syncthreads();
        // ---- 2. this wave's 32 score columns for the 64 rows: three MFMAs per fragment pair, small terms apart
        float uv[2][16];
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) {
            f32x16 ahh, asm_;
#pragma unroll
            for (int i = 0; i < 16; ++i) { ahh[i] = 0.f; asm_[i] = 0.f; }
            const _Float16* ap = img + (32 * rb + l31) * AF_LDA + 8 * hi;
#pragma unroll
            for (int ks = 0; ks < 16; ++ks) {
                const af_f16x8 ah = *reinterpret_cast<const af_f16x8*>(ap + 16 * ks);
                const af_f16x8 al = *reinterpret_cast<const af_f16x8*>(ap + 64 * AF_LDA + 16 * ks);
                ahh = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, whi[ks], ahh, 0, 0, 0);
                asm_ = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, wlo[ks], asm_, 0, 0, 0);
                asm_ = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, whi[ks], asm_, 0, 0, 0);
            }
#pragma unroll
            for (int i = 0; i < 16; ++i) uv[rb][i] = apply_act((ahh[i] * r_hh + asm_[i] * r_sm) + b1v, LOB_ACT_TANH);
        }
        __syncthreads();                       // every wave is done with the v images: the block now takes u (fp32)
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
#pragma unroll
            for (int i = 0; i < 16; ++i)
                ut[(32 * rb + (i & 3) + 8 * (i >> 2) + 4 * hi) * AF_LDU + 32 * w + l31] = uv[rb][i];
        __syncthreads();
        // ---- 3. scores of this wave's 16 rows (attn_pool_fwd_kernel<float>'s order: columns lane, lane + 64; wave reduction)
        RowTB tb(r0, Bp);
        for (int rr = 0; rr < 16; ++rr, tb.step(1, Bp)) {
            const long r = r0 + rr;
            if (r >= rows) break;
            const float u0 = ut[(16 * w + rr) * AF_LDU + lane], u1 = ut[(16 * w + rr) * AF_LDU + lane + 64];
            if (SAVE) { U[(size_t)r * W2 + lane] = u0; U[(size_t)r * W2 + lane + 64] = u1; }
            float sc = fmaf(u0, w2a, 0.f);
            sc = fmaf(u1, w2b, sc);
            sc = wave_sum(sc);
            if (lane == 0 && tb.b < B) S[(size_t)tb.b * T + tb.t] = sc + bias2;
        }
        __syncthreads();                       // the block is free for the next tile's v images
    }
}

// The same fusion at H = 256 (post-LSTM width 512, score layer 512 -> 256): a workgroup of EIGHT waves owns 64 rows;
// wave w normalises rows 8 w .. + 7 with the width-512 LayerNorm kernel's lane assignment (64 lanes x 8 columns, one row
// per pass), owns the score layer's columns 32 w .. + 31 (32 B fragments of W1 = 128 registers, 2 row blocks x 32 k-steps)
// and forms the scores of its 8 rows with the pooling kernel's order (4 columns per lane).
constexpr int AS2_LDA = 520;           // bf16 row stride of the v tile (1040 B = 65 x 16 B)
constexpr int AS2_LDU = 260;           // fp32 row stride of the u tile (1040 B)

template <bool SAVE>
__global__ __launch_bounds__(512) void attn_score_h256_kernel(
    const __bf16* __restrict__ Y, const float* __restrict__ gamma, const float* __restrict__ beta,
    const __bf16* __restrict__ W1, const float* __restrict__ b1, const float* __restrict__ w2, const float* __restrict__ b2,
    __bf16* __restrict__ V, float* __restrict__ U, float* __restrict__ S, int T, int B, int Bp, float eps) {
    constexpr int W = 512, W2 = 256;
    __shared__ __attribute__((aligned(16))) float lds[64 * AS2_LDU];
    __bf16* at = reinterpret_cast<__bf16*>(lds);
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    ip_bf16x8 wf[32];
#pragma unroll
    for (int ks = 0; ks < 32; ++ks)
        wf[ks] = *reinterpret_cast<const ip_bf16x8*>(W1 + (size_t)(32 * w + l31) * W + 16 * ks + 8 * hi);
    const float b1v = b1 ? b1[32 * w + l31] : 0.f;
    float gm[8], bt[8];
    ldv<8>(gamma + lane * 8, gm);
    ldv<8>(beta + lane * 8, bt);
    float w2v[4];
    ldv<4>(w2 + 4 * lane, w2v);
    const float bias2 = b2 ? b2[0] : 0.f;
    const long rows = (long)T * Bp;
    const long ntile = (rows + 63) >> 6;
    const float invw = 1.0f / (float)W;
    // The rows of a tile are requested one tile AHEAD (round 4): the loads of tile q + 1 are issued right after the barrier
    // that ends tile q's LayerNorm phase and land during its 64 MFMAs; before, every tile started with an exposed HBM
    // round trip (one workgroup per CU, all eight waves in the same phase: 0.82 ms for 3.2 GB).  Rows past the end are
    // clamped to the last row (their results are never stored).
    ip_bf16x8 raw[8];
    auto load_rows = [&](long tl) {
        const long r0 = tl * 64 + 8 * w;
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            const long r = r0 + p < rows ? r0 + p : rows - 1;
            raw[p] = *reinterpret_cast<const ip_bf16x8*>(Y + (size_t)r * W + lane * 8);
        }
    };
    if ((long)blockIdx.x < ntile) load_rows(blockIdx.x);
    for (long tl = blockIdx.x; tl < ntile; tl += gridDim.x) {
        const long r0 = tl * 64 + 8 * w;
        // ---- 1. LayerNorm of this wave's 8 rows (layernorm_act_vec_kernel<8, true, 64, bf16>)
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            const long r = r0 + p;
            float v[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = r < rows ? (float)raw[p][i] : 0.f;
            float s = 0.f;
#pragma unroll
            for (int i = 0; i < 8; ++i) s += v[i];
            const float mean = row_sum<64>(s) * invw;
            float q = 0.f;
#pragma unroll
            for (int i = 0; i < 8; ++i) { const float dl = v[i] - mean; q = __builtin_fmaf(dl, dl, q); }
            const float rstd = rsqrtf(__builtin_fmaf(row_sum<64>(q), invw, eps));
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = __builtin_fmaf((v[i] - mean) * rstd, gm[i], bt[i]);
            __bf16* arow = at + (8 * w + p) * AS2_LDA + lane * 8;
            stv_bf16<8>(arow, v);
            if (r < rows) *reinterpret_cast<ip_bf16x8*>(V + (size_t)r * W + lane * 8) = *reinterpret_cast<const ip_bf16x8*>(arow);
        }
        __syncthreads();
        {
            const long nx = tl + gridDim.x;
            load_rows(nx < ntile ? nx : tl);
        }
        // ---- 2. this wave's 32 score columns for the 64 rows
        f32x16 acc[2];
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[rb][i] = 0.f;
#pragma unroll
        for (int ks = 0; ks < 32; ++ks)
#pragma unroll
            for (int rb = 0; rb < 2; ++rb) {
                const ip_bf16x8 a = *reinterpret_cast<const ip_bf16x8*>(at + (32 * rb + l31) * AS2_LDA + 16 * ks + 8 * hi);
                acc[rb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, wf[ks], acc[rb], 0, 0, 0);
            }
        __syncthreads();
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
#pragma unroll
            for (int i = 0; i < 16; ++i)
                lds[(32 * rb + (i & 3) + 8 * (i >> 2) + 4 * hi) * AS2_LDU + 32 * w + l31] = fast_tanh(acc[rb][i] + b1v);
        __syncthreads();
        // ---- 3. scores of this wave's 8 rows (the pooling kernel's order: four columns per lane, wave reduction)
        RowTB tb(r0, Bp);
#pragma unroll
        for (int rr = 0; rr < 8; ++rr, tb.step(1, Bp)) {
            const long r = r0 + rr;
            if (r >= rows) break;
            float uu[4];
            {
                const f32x4 t4 = *reinterpret_cast<const f32x4*>(lds + (8 * w + rr) * AS2_LDU + 4 * lane);
                uu[0] = t4[0]; uu[1] = t4[1]; uu[2] = t4[2]; uu[3] = t4[3];
            }
            if (SAVE) stv<4>(U + (size_t)r * W2 + 4 * lane, uu);
            float sc = uu[0] * w2v[0];
#pragma unroll
            for (int i = 1; i < 4; ++i) sc = fmaf(uu[i], w2v[i], sc);
            sc = wave_sum(sc);
            if (lane == 0 && tb.b < B) S[(size_t)tb.b * T + tb.t] = sc + bias2;
        }
        __syncthreads();
    }
}

// The backward tail at H = 256 (width 512, dU 256 wide): eight waves per 64-row tile; wave w owns 64 of dV's 512 columns
// (2 column blocks x 16 k-steps = 32 B fragments of W1^T) and the LayerNorm backward of eight rows with the width-512
// kernel's lane assignment (64 lanes x 8 columns, one row per pass).
// Round 4 (as attn_ln_bwd_kernel above; 0.99 -> see HISTORY.md): a tile is 8 windows x 8 time steps and wave w keeps window
// b0 + w, so its dctx row is loaded once per tile and its attention weights are scalar loads; the LayerNorm inputs, dctx
// and the weights are requested BEFORE the tile's MFMAs and land behind them, the dU rows of the next tile during the
// LayerNorm passes; the dU tile has an LDS region of its own (two barriers per tile instead of four); the two 32-row blocks
// are multiplied one after the other (32 accumulator registers live instead of 64: room for the rows in flight); gamma
// sits in LDS.  Same arithmetic per row: bit-identical dx; dgamma / dbeta sum the rows in another order.
constexpr int AB2_LDA = 264;           // bf16 row stride of the dU tile (528 B)
constexpr int AB2_LDV = 520;           // bf16 row stride of the dV tile (1040 B)

__global__ __launch_bounds__(512) void attn_ln_bwd_h256_kernel(
    const __bf16* __restrict__ X, const float* __restrict__ gamma, const float* __restrict__ beta,
    const __bf16* __restrict__ dU, const __bf16* __restrict__ W1T, __bf16* __restrict__ dX,
    float* __restrict__ dgamma, float* __restrict__ dbeta, const float* __restrict__ attn, const float* __restrict__ dctx,
    int T, int B, int Bp, float eps) {
    constexpr int W = 512, W2 = 256;
    __shared__ __attribute__((aligned(16))) __bf16 lds[64 * AB2_LDV];          // dV tile; the dgamma / dbeta partials at the end
    __shared__ __attribute__((aligned(16))) __bf16 ldu[64 * AB2_LDA];          // dU tile
    __shared__ __attribute__((aligned(16))) float lgb[W];                       // gamma
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hi = lane >> 5;
    ip_bf16x8 wf[2][16];
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int ks = 0; ks < 16; ++ks)
            wf[cb][ks] = *reinterpret_cast<const ip_bf16x8*>(W1T + (size_t)(64 * w + 32 * cb + l31) * W2 + 16 * ks + 8 * hi);
    lgb[tid] = gamma[tid];
    float dga[8], dba[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { dga[i] = 0.f; dba[i] = 0.f; }
    const int nbb = Bp >> 3;                             // window blocks
    const long ntile = (long)((T + 7) >> 3) * nbb;
    const float invw = 1.0f / (float)W;
    // tile -> (first time step, first window); its rows: (t0 + p) * Bp + b0 + j, p < 8, j < 8; local index 8 p + j
    auto origin = [&](long tl, int& t0, int& b0) {
        const int tb = (int)((unsigned)tl / (unsigned)nbb);
        t0 = 8 * tb;
        b0 = 8 * ((int)tl - tb * nbb);
    };
    auto row_of = [&](int t0, int b0, int p, int j) -> size_t {      // clamped to the last time step (results masked)
        const int t = t0 + p < T ? t0 + p : T - 1;
        return (size_t)t * Bp + b0 + j;
    };
    // wave w brings in the dU rows of time step t0 + w (8 consecutive rows, 4 KB): lane -> (row l / 32 + 2 i, chunk l % 32)
    ip_bf16x8 du[4];
    auto load_du = [&](int t0, int b0) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
            du[i] = *reinterpret_cast<const ip_bf16x8*>(dU + row_of(t0, b0, w, (lane >> 5) + 2 * i) * W2 + 8 * (lane & 31));
    };
    if ((long)blockIdx.x < ntile) {
        int t0, b0;
        origin(blockIdx.x, t0, b0);
        load_du(t0, b0);
    }
    for (long tl = blockIdx.x; tl < ntile; tl += gridDim.x) {
        int t0, b0;
        origin(tl, t0, b0);
        // ---- 0. dU rows (requested a tile ago) into their LDS tile
#pragma unroll
        for (int i = 0; i < 4; ++i)
            *reinterpret_cast<ip_bf16x8*>(ldu + (8 * w + (lane >> 5) + 2 * i) * AB2_LDA + 8 * (lane & 31)) = du[i];
        // this wave's window b0 + w: its eight LayerNorm inputs, its dctx row, its attention weights (0 for a padding
        // window): in flight during the MFMAs
        ip_bf16x8 xr[8];
#pragma unroll
        for (int p = 0; p < 8; ++p) xr[p] = *reinterpret_cast<const ip_bf16x8*>(X + row_of(t0, b0, p, w) * W + lane * 8);
        const bool real = b0 + w < B;
        const int bb = real ? b0 + w : 0;
        float dcv[8], av[8];
        ldv<8>(dctx + (size_t)bb * W + lane * 8, dcv);
#pragma unroll
        for (int p = 0; p < 8; ++p) av[p] = attn[(size_t)bb * T + (t0 + p < T ? t0 + p : T - 1)];
        __syncthreads();
        // ---- 1. dV = dU W1: this wave's 64 columns for the 64 rows (operands swapped like the unfused GEMM), rounded to
        //         bf16 as the unfused GEMM stores it, row-major into the dV tile
        typedef __bf16 ab_bf16x4 __attribute__((ext_vector_type(4)));
#pragma nounroll
        for (int rb = 0; rb < 2; ++rb) {
            f32x16 acc[2];
#pragma unroll
            for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[cb][i] = 0.f;
#pragma unroll
            for (int ks = 0; ks < 16; ++ks) {
                const ip_bf16x8 a = *reinterpret_cast<const ip_bf16x8*>(ldu + (32 * rb + l31) * AB2_LDA + 16 * ks + 8 * hi);
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[0][ks], a, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[1][ks], a, acc[1], 0, 0, 0);
            }
#pragma unroll
            for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4) {
                    ab_bf16x4 pk = {(__bf16)acc[cb][4 * q4], (__bf16)acc[cb][4 * q4 + 1], (__bf16)acc[cb][4 * q4 + 2],
                                    (__bf16)acc[cb][4 * q4 + 3]};
                    *reinterpret_cast<ab_bf16x4*>(lds + (32 * rb + l31) * AB2_LDV + 64 * w + 32 * cb + 8 * q4 + 4 * hi) = pk;
                }
        }
        {   // the next tile's dU rows: in flight during the LayerNorm passes
            const long nx = tl + gridDim.x;
            int nt0, nb0;
            origin(nx < ntile ? nx : tl, nt0, nb0);
            load_du(nt0, nb0);
        }
        __syncthreads();
        // ---- 2. LayerNorm backward of window b0 + w, time steps t0 .. t0 + 7
        //         (layernorm_act_bwd_vec_kernel<8, 64, bf16, bf16, bf16>'s arithmetic)
        float gm[8];
        ldv<8>(lgb + lane * 8, gm);
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            if (t0 + p >= T) continue;
            float v[8], go[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = (float)xr[p][i];
            ldv_bf16<8>(lds + (8 * p + w) * AB2_LDV + lane * 8, go);
            if (real) {
                const float a = av[p];
#pragma unroll
                for (int i = 0; i < 8; ++i) go[i] = fmaf(a, dcv[i], go[i]);
            }
            float s = 0.f;
#pragma unroll
            for (int i = 0; i < 8; ++i) s += v[i];
            const float mean = row_sum<64>(s) * invw;
            float q = 0.f;
#pragma unroll
            for (int i = 0; i < 8; ++i) { const float dl = v[i] - mean; q = __builtin_fmaf(dl, dl, q); }
            const float rstd = rsqrtf(__builtin_fmaf(row_sum<64>(q), invw, eps));
            float m1 = 0.f, m2 = 0.f;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float xh = (v[i] - mean) * rstd;
                const float g = go[i];
                dga[i] = __builtin_fmaf(g, xh, dga[i]);
                dba[i] += g;
                const float dxh = g * gm[i];
                v[i] = xh; go[i] = dxh;
                m1 += dxh; m2 = __builtin_fmaf(dxh, xh, m2);
            }
            m1 = row_sum<64>(m1) * invw;
            m2 = row_sum<64>(m2) * invw;
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = rstd * __builtin_fmaf(-v[i], m2, go[i] - m1);
            stv_bf16<8>(dX + ((size_t)(t0 + p) * Bp + b0 + w) * W + lane * 8, v);
        }
        // no barrier here: the next tile's dU writes go to `ldu`, last read before the barrier above; its dV writes come
        // after its own first barrier, which every wave reaches only when it has finished reading this tile's dV rows
    }
    __syncthreads();
    float* red = reinterpret_cast<float*>(lds);          // [2][8][W] floats = 32 KB
#pragma unroll
    for (int i = 0; i < 8; ++i) { red[(0 * 8 + w) * W + lane * 8 + i] = dga[i]; red[(1 * 8 + w) * W + lane * 8 + i] = dba[i]; }
    __syncthreads();
    {
        float sg = 0.f, sb = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) { sg += red[(0 * 8 + k) * W + tid]; sb += red[(1 * 8 + k) * W + tid]; }
        atomicAdd(dgamma + tid, sg);
        atomicAdd(dbeta + tid, sb);
    }
}

// The input projection decomposed by COLUMNS (round 3; width 256, and width 128 as a test twin): a workgroup of
// WIDTH / 32 waves owns 64 consecutive (b, t) rows; every wave multiplies the whole tile with ITS 32 output columns (4 B
// fragments, 2 row blocks x 4 k-steps of v_mfma_f32_32x32x16_bf16: 32 accumulator registers instead of 64-128), the fp32
// pre-activations meet in one shared LDS tile, and the LayerNorm rows are dealt to the waves.  ~100 registers per wave: 16
// waves per CU instead of 8.  Same instruction, k order, lane assignment and arithmetic as input_proj_ln_kernel and the
// unfused sequence: bit-identical results.  At width 128 the wave-per-tile kernel above is faster (0.34 against 0.38 ms: four
// barriers per 64 rows here); at width 256 its accumulators and B fragments (2 x 128 registers) do not fit, this one does.
template <bool SAVE, int WIDTH>
__global__ __launch_bounds__(WIDTH * 2) void input_proj_ln2_kernel(
    const float* __restrict__ x, int C, int Cp, const float* __restrict__ W, int ldw, const float* __restrict__ bias,
    const float* __restrict__ gamma, const float* __restrict__ beta, float* __restrict__ pre, __bf16* __restrict__ xb,
    __bf16* __restrict__ out, long rows, int T, int Bp, float eps, int act, float drop_p, uint64_t seed) {
    constexpr int NW = WIDTH / 32, NT = NW * 64;           // waves / threads per workgroup
    constexpr int LDT = WIDTH + 4;                         // fp32 row stride of the output tile
    constexpr int LPR = WIDTH == 128 ? 16 : 64, VPL = WIDTH / LPR, GPW = 64 / LPR;    // the unfused LayerNorm kernel's lanes
    constexpr int RPWAVE = 64 / NW;                        // LayerNorm rows per wave and tile
    __shared__ __attribute__((aligned(16))) float tile[64 * LDT];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const int sub = lane / LPR, sl = lane % LPR;
    // B fragments of this wave's 32 columns: W[32 w + l31][16 ks + 8 hi + j], zero beyond C
    ip_bf16x8 wf[4];
    {
        const float* wrow = W + (size_t)(32 * w + l31) * ldw;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int k = 16 * ks + 8 * hi + j;
                wf[ks][j] = (__bf16)(k < C ? wrow[k] : 0.f);
            }
    }
    const float bv = bias ? bias[32 * w + l31] : 0.f;
    float gm[VPL], bt[VPL];
    const bool norm = !(act & LOB_LN_IDENTITY);
    act &= 0xff;
    if (norm) { ldv<VPL>(gamma + sl * VPL, gm); ldv<VPL>(beta + sl * VPL, bt); }
    else {
#pragma unroll
        for (int i = 0; i < VPL; ++i) { gm[i] = 1.f; bt[i] = 0.f; }
    }
    const float invw = 1.0f / (float)WIDTH;
    const long ntile = (rows + 63) >> 6;
    const long total = rows * (long)C;
    const int nch = 16 * C;                                // 16-byte chunks of a full tile (64 rows x C floats)
    constexpr int CPT = (16 * 64 + NT - 1) / NT;           // chunks per thread (C <= 64)
    f32x4 pf[CPT];
    auto fetch = [&](long tl) {
        const long f0 = tl * 64 * C;
#pragma unroll
        for (int i = 0; i < CPT; ++i) {
            const int ch = tid + NT * i;
            const long f = f0 + 4L * ch;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (ch < nch) {
                if (f + 4 <= total) v = *reinterpret_cast<const f32x4*>(x + f);
                else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) if (f + e < total) v[e] = x[f + e];
                }
            }
            pf[i] = v;
        }
    };
    long tl = blockIdx.x;
    if (tl < ntile) fetch(tl);
    for (; tl < ntile; tl += gridDim.x) {
        const long r0 = tl * 64;
        // ---- 0. the tile's floats, flat, into the block
#pragma unroll
        for (int i = 0; i < CPT; ++i) {
            const int ch = tid + NT * i;
            if (ch < nch) *reinterpret_cast<f32x4*>(tile + 4 * ch) = pf[i];
        }
        if (tl + gridDim.x < ntile) fetch(tl + gridDim.x);
        __syncthreads();
        // ---- 1. this wave's 32 columns of the 64 rows
        f32x16 acc[2];
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[rb][r] = 0.f;
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) {
            ip_bf16x8 af[4];
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int k = 16 * ks + 8 * hi + j;
                    af[ks][j] = (__bf16)(k < C ? tile[(32 * rb + l31) * C + k] : 0.f);
                }
            if (SAVE && w == rb && r0 + 32 * rb + l31 < rows) {          // waves 0 and 1 write the padded bf16 windows
#pragma unroll
                for (int ks = 0; ks < 4; ++ks)
                    if (16 * ks + 8 * hi < Cp)
                        *reinterpret_cast<ip_bf16x8*>(xb + (size_t)(r0 + 32 * rb + l31) * Cp + 16 * ks + 8 * hi) = af[ks];
            }
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
                acc[rb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ks], wf[ks], acc[rb], 0, 0, 0);
        }
        __syncthreads();                       // every wave has read the windows: the block now takes the pre-activations
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                tile[(32 * rb + (r & 3) + 8 * (r >> 2) + 4 * hi) * LDT + 32 * w + l31] = acc[rb][r] + bv;
        __syncthreads();
        // ---- 2. LayerNorm + activation + dropout of this wave's rows, GPW rows per pass (the unfused kernel's lanes)
        int bwi = (int)((unsigned)(r0 + RPWAVE * w + sub) / (unsigned)T);
        int tti = (int)((unsigned)(r0 + RPWAVE * w + sub) - (unsigned)bwi * (unsigned)T);
#pragma unroll 2
        for (int ps = 0; ps < RPWAVE / GPW; ++ps) {
            const int rt = RPWAVE * w + GPW * ps + sub;
            const long r = r0 + rt;
            const int orow = tti * Bp + bwi;             // (window, time) by increments: see input_proj_ln_kernel
            tti += GPW;
            while (tti >= T) { tti -= T; ++bwi; }
            if (r >= rows) continue;
            float v[VPL];
            ldv<VPL>(tile + rt * LDT + VPL * sl, v);
            if (SAVE) stv<VPL>(pre + (size_t)r * WIDTH + VPL * sl, v);
            float s = 0.f;
#pragma unroll
            for (int i = 0; i < VPL; ++i) s += v[i];
            const float mean = norm ? row_sum<LPR>(s) * invw : 0.f;
            float q = 0.f;
#pragma unroll
            for (int i = 0; i < VPL; ++i) { const float dl = v[i] - mean; q = __builtin_fmaf(dl, dl, q); }
            const float rstd = norm ? rsqrtf(__builtin_fmaf(row_sum<LPR>(q), invw, eps)) : 1.f;
            float ds[VPL];
#pragma unroll
            for (int i = 0; i < VPL; i += 2) {
                if (drop_p > 0.f) lob_dropout_scale2(seed, (uint64_t)orow * WIDTH + sl * VPL + i, drop_p, ds[i], ds[i + 1]);
                else { ds[i] = 1.f; ds[i + 1] = 1.f; }
            }
#pragma unroll
            for (int i = 0; i < VPL; ++i) {
                const float o = __builtin_fmaf((v[i] - mean) * rstd, gm[i], bt[i]);
                v[i] = apply_act(o, act) * ds[i];
            }
            stv_bf16<VPL>(out + (size_t)orow * WIDTH + sl * VPL, v);
        }
        __syncthreads();                       // the block is free for the next tile's windows
    }
}

// The same head for the FP32 path at width 128 (round 4), column-decomposed like the kernel above (the wave-per-tile shape
// needs 128 registers of fp32 B fragments next to 64 accumulators: it spilled): exact fp32 products (v_mfma_f32_32x32x2_f32,
// k ascending: the contraction is 61 long, the matrix work is nothing), fp32 activations out.  Unfused, lob_gemm_nt_f32 wrote
// the pre-activations and the LayerNorm kernel read them back: 1.8 KB per row against 0.76 (inference) / 1.27 (training,
// which keeps the pre-activations for the backward).
template <bool SAVE>
__global__ __launch_bounds__(256) void input_proj_ln_f32_kernel(
    const float* __restrict__ x, int C, const float* __restrict__ W, int ldw, const float* __restrict__ bias,
    const float* __restrict__ gamma, const float* __restrict__ beta, float* __restrict__ pre,
    float* __restrict__ out, long rows, int T, int Bp, float eps, int act, float drop_p, uint64_t seed) {
    constexpr int WIDTH = 128;
    constexpr int NW = WIDTH / 32, NT = NW * 64;           // waves / threads per workgroup
    constexpr int LDT = WIDTH + 4;                         // fp32 row stride of the output tile
    constexpr int LPR = WIDTH == 128 ? 16 : 64, VPL = WIDTH / LPR, GPW = 64 / LPR;    // the unfused LayerNorm kernel's lanes
    constexpr int RPWAVE = 64 / NW;                        // LayerNorm rows per wave and tile
    __shared__ __attribute__((aligned(16))) float tile[64 * LDT];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const int sub = lane / LPR, sl = lane % LPR;
    // B fragments of this wave's 32 columns: W[32 w + l31][2 ks + hi], zero beyond C
    float wf[32];
    {
        const float* wrow = W + (size_t)(32 * w + l31) * ldw;
#pragma unroll
        for (int ks = 0; ks < 32; ++ks) wf[ks] = 2 * ks + hi < C ? wrow[2 * ks + hi] : 0.f;
    }
    const float bv = bias ? bias[32 * w + l31] : 0.f;
    float gm[VPL], bt[VPL];
    const bool norm = !(act & LOB_LN_IDENTITY);
    act &= 0xff;
    if (norm) { ldv<VPL>(gamma + sl * VPL, gm); ldv<VPL>(beta + sl * VPL, bt); }
    else {
#pragma unroll
        for (int i = 0; i < VPL; ++i) { gm[i] = 1.f; bt[i] = 0.f; }
    }
    const float invw = 1.0f / (float)WIDTH;
    const long ntile = (rows + 63) >> 6;
    const long total = rows * (long)C;
    const int nch = 16 * C;                                // 16-byte chunks of a full tile (64 rows x C floats)
    constexpr int CPT = (16 * 64 + NT - 1) / NT;           // chunks per thread (C <= 64)
    f32x4 pf[CPT];
    auto fetch = [&](long tl) {
        const long f0 = tl * 64 * C;
#pragma unroll
        for (int i = 0; i < CPT; ++i) {
            const int ch = tid + NT * i;
            const long f = f0 + 4L * ch;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (ch < nch) {
                if (f + 4 <= total) v = *reinterpret_cast<const f32x4*>(x + f);
                else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) if (f + e < total) v[e] = x[f + e];
                }
            }
            pf[i] = v;
        }
    };
    long tl = blockIdx.x;
    if (tl < ntile) fetch(tl);
    for (; tl < ntile; tl += gridDim.x) {
        const long r0 = tl * 64;
        // ---- 0. the tile's floats, flat, into the block
#pragma unroll
        for (int i = 0; i < CPT; ++i) {
            const int ch = tid + NT * i;
            if (ch < nch) *reinterpret_cast<f32x4*>(tile + 4 * ch) = pf[i];
        }
        if (tl + gridDim.x < ntile) fetch(tl + gridDim.x);
        __syncthreads();
        // ---- 1. this wave's 32 columns of the 64 rows
        f32x16 acc[2];
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[rb][r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < 32; ++ks) {
            const int k = 2 * ks + hi;
#pragma unroll
            for (int rb = 0; rb < 2; ++rb) {
                const float af = k < C ? tile[(32 * rb + l31) * C + k] : 0.f;
                acc[rb] = __builtin_amdgcn_mfma_f32_32x32x2f32(af, wf[ks], acc[rb], 0, 0, 0);
            }
        }
        __syncthreads();                       // every wave has read the windows: the block now takes the pre-activations
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                tile[(32 * rb + (r & 3) + 8 * (r >> 2) + 4 * hi) * LDT + 32 * w + l31] = acc[rb][r] + bv;
        __syncthreads();
        // ---- 2. LayerNorm + activation + dropout of this wave's rows, GPW rows per pass (the unfused kernel's lanes)
        int bwi = (int)((unsigned)(r0 + RPWAVE * w + sub) / (unsigned)T);
        int tti = (int)((unsigned)(r0 + RPWAVE * w + sub) - (unsigned)bwi * (unsigned)T);
#pragma unroll 2
        for (int ps = 0; ps < RPWAVE / GPW; ++ps) {
            const int rt = RPWAVE * w + GPW * ps + sub;
            const long r = r0 + rt;
            const int orow = tti * Bp + bwi;             // (window, time) by increments: see input_proj_ln_kernel
            tti += GPW;
            while (tti >= T) { tti -= T; ++bwi; }
            if (r >= rows) continue;
            float v[VPL];
            ldv<VPL>(tile + rt * LDT + VPL * sl, v);
            if (SAVE) stv<VPL>(pre + (size_t)r * WIDTH + VPL * sl, v);
            float s = 0.f;
#pragma unroll
            for (int i = 0; i < VPL; ++i) s += v[i];
            const float mean = norm ? row_sum<LPR>(s) * invw : 0.f;
            float q = 0.f;
#pragma unroll
            for (int i = 0; i < VPL; ++i) { const float dl = v[i] - mean; q = __builtin_fmaf(dl, dl, q); }
            const float rstd = norm ? rsqrtf(__builtin_fmaf(row_sum<LPR>(q), invw, eps)) : 1.f;
            float ds[VPL];
#pragma unroll
            for (int i = 0; i < VPL; i += 2) {
                if (drop_p > 0.f) lob_dropout_scale2(seed, (uint64_t)orow * WIDTH + sl * VPL + i, drop_p, ds[i], ds[i + 1]);
                else { ds[i] = 1.f; ds[i + 1] = 1.f; }
            }
#pragma unroll
            for (int i = 0; i < VPL; ++i) {
                const float o = __builtin_fmaf((v[i] - mean) * rstd, gm[i], bt[i]);
                v[i] = apply_act(o, act) * ds[i];
            }
            stv<VPL>(out + (size_t)orow * WIDTH + sl * VPL, v);
        }
        __syncthreads();                       // the block is free for the next tile's windows
    }
}

}  // namespace

extern "C" int lob_dropout_f32(const float* in, float* out, int64_t n, float p, uint64_t seed, void* stream) {
    if (!in || !out || n <= 0 || p < 0.f || p >= 1.f) return LOB_E_ARG;
    int64_t blocks = (n + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL(dropout_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, in, out, (size_t)n, p, seed);
    LOB_CHECK_LAUNCH();
    return 0;
}

// width 128: 16 lanes per row (LOB_LN_LPR=64 selects the one-row-per-wave form)
static bool ln_lpr16() {
    const bool v = lob_variant(LOB_VAR_LN_LPR) != 64;
    return v;
}

extern "C" int lob_input_proj_bwd_bf16(const float* pre, const float* gamma, const float* beta, const void* dA16,
                                       const void* xb16, int Cp, float* dW, int lddw, float* dgamma, float* dbeta,
                                       float* dbias, int B, int T, int Bp, int H, float eps, int act, float drop_p,
                                       uint64_t seed, void* stream) {
    if (!pre || !dA16 || !xb16 || !dW || B <= 0 || T <= 0 || Bp < B) return LOB_E_ARG;
    if (!(act & LOB_LN_IDENTITY) && (!gamma || !beta || !dgamma || !dbeta)) return LOB_E_ARG;
    if (H != 128 || Cp <= 0 || Cp > 64 || (Cp & 7) || lddw < Cp) return LOB_E_SHAPE;
    if (drop_p < 0.f || drop_p >= 1.f) return LOB_E_ARG;
    if ((reinterpret_cast<uintptr_t>(pre) | reinterpret_cast<uintptr_t>(dA16) | reinterpret_cast<uintptr_t>(xb16) |
         reinterpret_cast<uintptr_t>(gamma) | reinterpret_cast<uintptr_t>(beta)) & 15) return LOB_E_ALIGN;
    const long rows = (long)B * T;
    const long ntile = (rows + 31) / 32;
    const int nb = (int)((ntile + 3) / 4 < 512 ? (ntile + 3) / 4 : 512);
    hipLaunchKernelGGL(input_proj_bwd_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, pre, gamma, beta,
                       reinterpret_cast<const __bf16*>(dA16), reinterpret_cast<const __bf16*>(xb16), Cp, dW, lddw, dgamma, dbeta,
                       dbias, rows, T, Bp, eps, act, drop_p, seed);
    LOB_CHECK_LAUNCH();
    return 0;
}

extern "C" int lob_attn_ln_bwd_bf16(const void* X16, const float* gamma, const float* beta, const void* dU16,
                                    const void* W1T_16, void* dX16, float* dgamma, float* dbeta, const float* attn,
                                    const float* dctx, int T, int B, int Bp, int H, int D, float eps, void* stream) {
    if (!X16 || !gamma || !beta || !dU16 || !W1T_16 || !dX16 || !dgamma || !dbeta || !attn || !dctx || T <= 0 || B <= 0 ||
        Bp < B) return LOB_E_ARG;
    if ((H != 128 && H != 256) || D != 2 || (Bp % 32) || (long)T * Bp >= (1L << 31)) return LOB_E_SHAPE;
    if ((reinterpret_cast<uintptr_t>(X16) | reinterpret_cast<uintptr_t>(gamma) | reinterpret_cast<uintptr_t>(beta) |
         reinterpret_cast<uintptr_t>(dU16) | reinterpret_cast<uintptr_t>(W1T_16) | reinterpret_cast<uintptr_t>(dX16) |
         reinterpret_cast<uintptr_t>(dctx)) & 15) return LOB_E_ALIGN;
    if (H == 256) {
        const long nt = (long)((T + 7) / 8) * (Bp / 8);            // tiles of 8 windows x 8 time steps
        const int nb2 = (int)(nt < 256 ? nt : 256);
        hipLaunchKernelGGL(attn_ln_bwd_h256_kernel, dim3(nb2), dim3(512), 0, (hipStream_t)stream,
                           reinterpret_cast<const __bf16*>(X16), gamma, beta, reinterpret_cast<const __bf16*>(dU16),
                           reinterpret_cast<const __bf16*>(W1T_16), reinterpret_cast<__bf16*>(dX16), dgamma, dbeta, attn, dctx,
                           T, B, Bp, eps);
        LOB_CHECK_LAUNCH();
        return 0;
    }
    const long ntile = (long)((T + 7) / 8) * (Bp / 16);         // tiles of 16 windows x 8 time steps
    const int nb = (int)(ntile < 256 ? ntile : 256);      // one 8-wave workgroup per CU
    hipLaunchKernelGGL(attn_ln_bwd_kernel, dim3(nb), dim3(512), 0, (hipStream_t)stream,
                       reinterpret_cast<const __bf16*>(X16), gamma, beta, reinterpret_cast<const __bf16*>(dU16),
                       reinterpret_cast<const __bf16*>(W1T_16), reinterpret_cast<__bf16*>(dX16), dgamma, dbeta, attn, dctx,
                       T, B, Bp, eps);
    LOB_CHECK_LAUNCH();
    return 0;
}

extern "C" int lob_attn_scores_f32(const float* Y, const float* gamma, const float* beta, const float* W1, const float* b1,
                                   const float* w2, const float* b2, float* V, float* U, float* S, int T, int B, int Bp,
                                   int H, int D, float eps, void* stream) {
    if (!Y || !gamma || !beta || !W1 || !w2 || !V || !S || T <= 0 || B <= 0 || Bp < B) return LOB_E_ARG;
    if (H != 128 || D != 2 || (Bp % 32) || (long)T * Bp >= (1L << 31)) return LOB_E_SHAPE;
    if ((reinterpret_cast<uintptr_t>(Y) | reinterpret_cast<uintptr_t>(gamma) | reinterpret_cast<uintptr_t>(beta) |
         reinterpret_cast<uintptr_t>(W1) | reinterpret_cast<uintptr_t>(V)) & 15) return LOB_E_ALIGN;
    const long ntile = ((long)T * Bp + 63) / 64;
    const int nb = (int)(ntile < 512 ? ntile : 512);          // two 4-wave workgroups per CU
    if (U) hipLaunchKernelGGL((attn_score_f32_kernel<true>), dim3(nb), dim3(256), 0, (hipStream_t)stream, Y, gamma, beta, W1, b1,
                              w2, b2, V, U, S, T, B, Bp, eps);
    else   hipLaunchKernelGGL((attn_score_f32_kernel<false>), dim3(nb), dim3(256), 0, (hipStream_t)stream, Y, gamma, beta, W1, b1,
                              w2, b2, V, U, S, T, B, Bp, eps);
    LOB_CHECK_LAUNCH();
    return 0;
}

extern "C" int lob_attn_scores_bf16(const void* Y16, const float* gamma, const float* beta, const void* W1_16,
                                    const float* b1, const float* w2, const float* b2, void* V, float* U, float* S,
                                    int T, int B, int Bp, int H, int D, float eps, void* stream) {
    if (!Y16 || !gamma || !beta || !W1_16 || !w2 || !V || !S || T <= 0 || B <= 0 || Bp < B) return LOB_E_ARG;
    if ((H != 128 && H != 256) || D != 2 || (Bp % 32) || (long)T * Bp >= (1L << 31)) return LOB_E_SHAPE;
    if ((reinterpret_cast<uintptr_t>(Y16) | reinterpret_cast<uintptr_t>(gamma) | reinterpret_cast<uintptr_t>(beta) |
         reinterpret_cast<uintptr_t>(W1_16) | reinterpret_cast<uintptr_t>(w2) | reinterpret_cast<uintptr_t>(V) |
         reinterpret_cast<uintptr_t>(U)) & 15) return LOB_E_ALIGN;
    if (H == 256) {
        const long nt = ((long)T * Bp + 63) / 64;
        const int nb2 = (int)(nt < 256 ? nt : 256);           // one 8-wave workgroup per CU
        if (U)
            hipLaunchKernelGGL((attn_score_h256_kernel<true>), dim3(nb2), dim3(512), 0, (hipStream_t)stream,
                               reinterpret_cast<const __bf16*>(Y16), gamma, beta, reinterpret_cast<const __bf16*>(W1_16), b1, w2,
                               b2, reinterpret_cast<__bf16*>(V), U, S, T, B, Bp, eps);
        else
            hipLaunchKernelGGL((attn_score_h256_kernel<false>), dim3(nb2), dim3(512), 0, (hipStream_t)stream,
                               reinterpret_cast<const __bf16*>(Y16), gamma, beta, reinterpret_cast<const __bf16*>(W1_16), b1, w2,
                               b2, reinterpret_cast<__bf16*>(V), U, S, T, B, Bp, eps);
        LOB_CHECK_LAUNCH();
        return 0;
    }
    const long ntile = ((long)T * Bp + 127) / 128;
    const int nb = (int)(ntile < 512 ? ntile : 512);
    if (U)
        hipLaunchKernelGGL((attn_score_kernel<true>), dim3(nb), dim3(256), 0, (hipStream_t)stream,
                           reinterpret_cast<const __bf16*>(Y16), gamma, beta, reinterpret_cast<const __bf16*>(W1_16), b1, w2, b2,
                           reinterpret_cast<__bf16*>(V), U, S, T, B, Bp, eps);
    else
        hipLaunchKernelGGL((attn_score_kernel<false>), dim3(nb), dim3(256), 0, (hipStream_t)stream,
                           reinterpret_cast<const __bf16*>(Y16), gamma, beta, reinterpret_cast<const __bf16*>(W1_16), b1, w2, b2,
                           reinterpret_cast<__bf16*>(V), U, S, T, B, Bp, eps);
    LOB_CHECK_LAUNCH();
    return 0;
}

extern "C" int lob_input_proj_ln_bf16(const float* x, int C, const float* W, int ldw, const float* bias,
                                      const float* gamma, const float* beta, float* pre, void* xb, int Cp, void* out,
                                      int B, int T, int Bp, int H, float eps, int act, float drop_p, uint64_t seed,
                                      void* stream) {
    if (!x || !W || !out || B <= 0 || T <= 0 || Bp < B || C <= 0 || ldw < C) return LOB_E_ARG;
    if (!(act & LOB_LN_IDENTITY) && (!gamma || !beta)) return LOB_E_ARG;
    if ((pre == nullptr) != (xb == nullptr)) return LOB_E_ARG;
    if ((H != 128 && H != 256) || C > 64 || (xb && (Cp < C || Cp > 64 || (Cp & 7)))) return LOB_E_SHAPE;
    if (drop_p < 0.f || drop_p >= 1.f) return LOB_E_ARG;
    if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(pre) | reinterpret_cast<uintptr_t>(xb) |
         reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(gamma) | reinterpret_cast<uintptr_t>(beta)) & 15)
        return LOB_E_ALIGN;
    const long rows = (long)B * T;
    if (rows + 64 >= (1L << 31) || (long)T * Bp >= (1L << 31)) return LOB_E_SHAPE;       // 32-bit row arithmetic in the kernels
    __bf16* xbb = reinterpret_cast<__bf16*>(xb);
    __bf16* outb = reinterpret_cast<__bf16*>(out);
    // H = 128 runs the wave-per-tile kernel (0.34 against 0.38 ms at B = 4096); LOB_IP_COLWAVE in act selects the
    // column-decomposed one there as well (test twin).  H = 256: column-decomposed only (0.56 against 0.87 ms unfused)
    const bool colwave = (act & LOB_IP_COLWAVE) != 0;
    act &= ~LOB_IP_COLWAVE;
    if (H == 256 || colwave) {
        const long nt = (rows + 63) / 64;
        const int wgs = H == 128 ? 1024 : 512;              // 4 (2) workgroups of 4 (8) waves per CU
        const int nb2 = (int)(nt < wgs ? nt : wgs);
#define LOB_IP2(SV, WD) hipLaunchKernelGGL((input_proj_ln2_kernel<SV, WD>), dim3(nb2), dim3(WD * 2), 0, (hipStream_t)stream, x, C, \
                                           Cp, W, ldw, bias, gamma, beta, pre, xbb, outb, rows, T, Bp, eps, act, drop_p, seed)
        if (H == 128) { if (pre) LOB_IP2(true, 128); else LOB_IP2(false, 128); }
        else          { if (pre) LOB_IP2(true, 256); else LOB_IP2(false, 256); }
#undef LOB_IP2
        LOB_CHECK_LAUNCH();
        return 0;
    }
    const long ntile = (rows + 31) / 32;
    const int nb = (int)((ntile + 3) / 4 < 512 ? (ntile + 3) / 4 : 512);      // 2 workgroups per CU, persistent waves
    if (pre)
        hipLaunchKernelGGL((input_proj_ln_kernel<true>), dim3(nb), dim3(256), 0, (hipStream_t)stream, x, C, Cp, W, ldw, bias,
                           gamma, beta, pre, xbb, outb, rows, T, Bp, eps, act, drop_p, seed);
    else
        hipLaunchKernelGGL((input_proj_ln_kernel<false>), dim3(nb), dim3(256), 0, (hipStream_t)stream, x, C, Cp, W, ldw, bias,
                           gamma, beta, pre, xbb, outb, rows, T, Bp, eps, act, drop_p, seed);
    LOB_CHECK_LAUNCH();
    return 0;
}

extern "C" int lob_input_proj_ln_f32(const float* x, int C, const float* W, int ldw, const float* bias, const float* gamma,
                                    const float* beta, float* pre, float* out, int B, int T, int Bp, int H, float eps, int act,
                                    float drop_p, uint64_t seed, void* stream) {
    if (!x || !W || !out || B <= 0 || T <= 0 || Bp < B || C <= 0 || ldw < C) return LOB_E_ARG;
    if (!(act & LOB_LN_IDENTITY) && (!gamma || !beta)) return LOB_E_ARG;
    if (H != 128 || C > 64) return LOB_E_SHAPE;
    if (drop_p < 0.f || drop_p >= 1.f) return LOB_E_ARG;
    if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(pre) | reinterpret_cast<uintptr_t>(out) |
         reinterpret_cast<uintptr_t>(gamma) | reinterpret_cast<uintptr_t>(beta)) & 15) return LOB_E_ALIGN;
    const long rows = (long)B * T;
    if (rows + 64 >= (1L << 31) || (long)T * Bp >= (1L << 31)) return LOB_E_SHAPE;       // 32-bit row arithmetic in the kernel
    const long ntile = (rows + 63) / 64;
    const int nb = (int)(ntile < 512 ? ntile : 512);                           // 2 workgroups of 4 waves per CU, persistent
    if (pre)
        hipLaunchKernelGGL((input_proj_ln_f32_kernel<true>), dim3(nb), dim3(256), 0, (hipStream_t)stream, x, C, W, ldw, bias,
                           gamma, beta, pre, out, rows, T, Bp, eps, act, drop_p, seed);
    else
        hipLaunchKernelGGL((input_proj_ln_f32_kernel<false>), dim3(nb), dim3(256), 0, (hipStream_t)stream, x, C, W, ldw, bias,
                           gamma, beta, pre, out, rows, T, Bp, eps, act, drop_p, seed);
    LOB_CHECK_LAUNCH();
    return 0;
}

extern "C" int lob_layernorm_act_f32(const float* in, const float* gamma, const float* beta,
                                     void* out, int out_bf16, int rows, int width, float eps, int act,
                                     int remap_T, int remap_B, int remap_Bp,
                                     float drop_p, uint64_t seed, void* stream) {
    const bool ident = (act & LOB_LN_IDENTITY) != 0;
    const bool x16 = (act & LOB_X_BF16) != 0;
    act &= ~LOB_X_BF16;
    if (!in || !out || rows <= 0 || width <= 0 || (!ident && (!gamma || !beta))) return LOB_E_ARG;
    if (width > 64 * LN_MAX_PER_LANE) return LOB_E_SHAPE;
    if (drop_p < 0.f || drop_p >= 1.f) return LOB_E_ARG;
    if (remap_T > 0 && (remap_B <= 0 || remap_Bp < remap_B || rows != remap_T * remap_B)) return LOB_E_SHAPE;
    if (x16) {       // bf16 input rows: the post-LSTM LayerNorm of the mixed path (width 256 / 512), vectorised kernel only
        if ((width != 256 && width != 512) || remap_T > 0 ||
            ((reinterpret_cast<uintptr_t>(in) | reinterpret_cast<uintptr_t>(out) |
              reinterpret_cast<uintptr_t>(gamma) | reinterpret_cast<uintptr_t>(beta)) & 15)) return LOB_E_SHAPE;
        const __bf16* inb = reinterpret_cast<const __bf16*>(in);
        if (width == 512) {                // H = 256: 64 lanes x 8 values: 16 B per lane in, 16 B (bf16) / 32 B out
            int nb = (rows + 3) / 4;
            if (nb > 256 * 16) nb = 256 * 16;
            if (out_bf16) hipLaunchKernelGGL((layernorm_act_vec_kernel<8, true, 64, __bf16>), dim3(nb), dim3(256), 0, (hipStream_t)stream,
                                             inb, gamma, beta, out, rows, eps, act, remap_T, remap_B, remap_Bp, drop_p, seed);
            else hipLaunchKernelGGL((layernorm_act_vec_kernel<8, false, 64, __bf16>), dim3(nb), dim3(256), 0, (hipStream_t)stream,
                                    inb, gamma, beta, out, rows, eps, act, remap_T, remap_B, remap_Bp, drop_p, seed);
            LOB_CHECK_LAUNCH();
            return 0;
        }
        if (out_bf16 && ln_lpr16()) {      // bf16 in and out: 32 lanes per row, 16 B per lane and stream
            int nb = (rows + 15) / 16;     // RPW 2 x GPW 2 rows per wave pass, 4 waves
            if (nb > 256 * 16) nb = 256 * 16;
            hipLaunchKernelGGL((layernorm_act_vec_kernel<8, true, 32, __bf16>), dim3(nb), dim3(256), 0, (hipStream_t)stream,
                               inb, gamma, beta, out, rows, eps, act, remap_T, remap_B, remap_Bp, drop_p, seed);
            LOB_CHECK_LAUNCH();
            return 0;
        }
        int nb = (rows + 3) / 4;
        if (nb > 256 * 16) nb = 256 * 16;
        if (out_bf16) hipLaunchKernelGGL((layernorm_act_vec_kernel<4, true, 64, __bf16>), dim3(nb), dim3(256), 0, (hipStream_t)stream,
                                         inb, gamma, beta, out, rows, eps, act, remap_T, remap_B, remap_Bp, drop_p, seed);
        else hipLaunchKernelGGL((layernorm_act_vec_kernel<4, false, 64, __bf16>), dim3(nb), dim3(256), 0, (hipStream_t)stream,
                                inb, gamma, beta, out, rows, eps, act, remap_T, remap_B, remap_Bp, drop_p, seed);
        LOB_CHECK_LAUNCH();
        return 0;
    }
    const int waves_per_block = 4;
    // the walk a wave strides over: rows, or (remap) the 8 x 8 (b, t) tiles INCLUDING their out-of-range corners -- with
    // one window that is 8 x the rows, and a grid sized from `rows` made every wave loop 8 times (28 us for 256 rows)
    const int work = remap_T > 0 ? ((remap_B + 7) >> 3) * ((remap_T + 7) >> 3) * 64 : rows;
    int blocks = (work + waves_per_block - 1) / waves_per_block;
    if (blocks > 256 * 16) blocks = 256 * 16;
    const bool al = ((reinterpret_cast<uintptr_t>(in) | reinterpret_cast<uintptr_t>(out) |
                      reinterpret_cast<uintptr_t>(gamma) | reinterpret_cast<uintptr_t>(beta)) & 15) == 0;
#define LOB_LN_VEC(V) do { if (out_bf16) hipLaunchKernelGGL((layernorm_act_vec_kernel<V, true>), dim3(blocks), dim3(256), 0, \
        (hipStream_t)stream, in, gamma, beta, out, rows, eps, act, remap_T, remap_B, remap_Bp, drop_p, seed); \
    else hipLaunchKernelGGL((layernorm_act_vec_kernel<V, false>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, \
        in, gamma, beta, out, rows, eps, act, remap_T, remap_B, remap_Bp, drop_p, seed); } while (0)
    if (al && width == 128 && ln_lpr16()) {
        // four rows per wave pass: a pass covers 4 rows, RPW = 2 passes in flight
        blocks = (work + 31) / 32;
        if (blocks > 256 * 16) blocks = 256 * 16;
        if (out_bf16) hipLaunchKernelGGL((layernorm_act_vec_kernel<8, true, 16>), dim3(blocks), dim3(256), 0, (hipStream_t)stream,
                                         in, gamma, beta, out, rows, eps, act, remap_T, remap_B, remap_Bp, drop_p, seed);
        else hipLaunchKernelGGL((layernorm_act_vec_kernel<8, false, 16>), dim3(blocks), dim3(256), 0, (hipStream_t)stream,
                                in, gamma, beta, out, rows, eps, act, remap_T, remap_B, remap_Bp, drop_p, seed);
    }
    else if (al && width == 128) LOB_LN_VEC(2);
    else if (al && width == 256) LOB_LN_VEC(4);
    else if (al && width == 512) LOB_LN_VEC(8);
    else {
        if (out_bf16) return LOB_E_SHAPE;       // bf16 output exists on the vectorised path only
        hipLaunchKernelGGL(layernorm_act_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, in, gamma, beta,
                           reinterpret_cast<float*>(out), rows, width, eps, act, remap_T, remap_B, remap_Bp, drop_p, seed);
    }
#undef LOB_LN_VEC
    LOB_CHECK_LAUNCH();
    return 0;
}

extern "C" int lob_attn_pool_fwd_f32(const void* V, int v_bf16, const float* U, const float* w2, const float* b2,
                                     float* ctx, float* attn, int T, int B, int Bp, int W, int W2,
                                     void* stream) {
    if (!V || !ctx || !attn || T <= 0 || B <= 0 || Bp < B || W <= 0) return LOB_E_ARG;
    if ((size_t)T * sizeof(float) > 60 * 1024) return LOB_E_SHAPE;
    if (U && W2 == 0 && !v_bf16) {      // fp32 v with finished scores S[B][T] (lob_attn_scores_f32): the generic kernel
        hipLaunchKernelGGL((attn_pool_fwd_kernel<float>), dim3(B), dim3(256), (size_t)T * sizeof(float), (hipStream_t)stream,
                           reinterpret_cast<const float*>(V), U, w2, b2, ctx, attn, T, Bp, W, 0);
        LOB_CHECK_LAUNCH();
        return 0;
    }
    if (U && W2 == 0) {          // U = finished scores S[B][T] (lob_attn_scores_bf16)
        if (!v_bf16 || (W != 256 && W != 512) || (reinterpret_cast<uintptr_t>(V) & 15)) return LOB_E_SHAPE;
        const size_t smem = ((size_t)((T + 3) & ~3) + 4 * W) * sizeof(float);
        if (W == 256) hipLaunchKernelGGL((attn_pool_fwd_vec_kernel<1, true>), dim3(B), dim3(256), smem, (hipStream_t)stream,
                                         reinterpret_cast<const __bf16*>(V), U, w2, b2, ctx, attn, T, Bp);
        else          hipLaunchKernelGGL((attn_pool_fwd_vec_kernel<2, true>), dim3(B), dim3(256), smem, (hipStream_t)stream,
                                         reinterpret_cast<const __bf16*>(V), U, w2, b2, ctx, attn, T, Bp);
        LOB_CHECK_LAUNCH();
        return 0;
    }
    if (U && (!w2 || W2 <= 0)) return LOB_E_ARG;
    const bool al8 = ((reinterpret_cast<uintptr_t>(V) | reinterpret_cast<uintptr_t>(U) | reinterpret_cast<uintptr_t>(w2)) & 7) == 0;
    const bool al16v = ((reinterpret_cast<uintptr_t>(V) | reinterpret_cast<uintptr_t>(U) | reinterpret_cast<uintptr_t>(w2)) & 15) == 0;
    if (v_bf16 && U && ((W == 256 && W2 == 128 && al8) || (W == 512 && W2 == 256 && al16v))) {
        const size_t smem = ((size_t)((T + 3) & ~3) + 4 * W) * sizeof(float);
        if (W == 256) hipLaunchKernelGGL(attn_pool_fwd_vec_kernel<1>, dim3(B), dim3(256), smem, (hipStream_t)stream,
                                         reinterpret_cast<const __bf16*>(V), U, w2, b2, ctx, attn, T, Bp);
        else          hipLaunchKernelGGL(attn_pool_fwd_vec_kernel<2>, dim3(B), dim3(256), smem, (hipStream_t)stream,
                                         reinterpret_cast<const __bf16*>(V), U, w2, b2, ctx, attn, T, Bp);
        LOB_CHECK_LAUNCH();
        return 0;
    }
    if (v_bf16)
        hipLaunchKernelGGL((attn_pool_fwd_kernel<__bf16>), dim3(B), dim3(256), (size_t)T * sizeof(float), (hipStream_t)stream,
                           reinterpret_cast<const __bf16*>(V), U, w2, b2, ctx, attn, T, Bp, W, W2);
    else
        hipLaunchKernelGGL((attn_pool_fwd_kernel<float>), dim3(B), dim3(256), (size_t)T * sizeof(float), (hipStream_t)stream,
                           reinterpret_cast<const float*>(V), U, w2, b2, ctx, attn, T, Bp, W, W2);
    LOB_CHECK_LAUNCH();
    return 0;
}

extern "C" int lob_softmax_rows_f32(const float* in, float* out, int rows, int cols, void* stream) {
    if (!in || !out || rows <= 0 || cols <= 0) return LOB_E_ARG;
    hipLaunchKernelGGL(softmax_rows_kernel, dim3((rows + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                       in, out, rows, cols);
    LOB_CHECK_LAUNCH();
    return 0;
}

extern "C" int lob_act_f32(const float* in, float* out, int64_t n, int act, void* stream) {
    if (!in || !out || n <= 0) return LOB_E_ARG;
    int64_t blocks = (n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(act_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, in, out, (size_t)n, act);
    LOB_CHECK_LAUNCH();
    return 0;
}

extern "C" int lob_act_bwd_f32(const float* dy, const float* pre, float* dx, int64_t n, int act, void* stream) {
    if (!dy || !pre || !dx || n <= 0) return LOB_E_ARG;
    int64_t blocks = (n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(act_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, dy, pre, dx, (size_t)n, act);
    LOB_CHECK_LAUNCH();
    return 0;
}

extern "C" int lob_layernorm_act_bwd_f32(const float* x, const float* gamma, const float* beta, const float* dy,
                                         float* dx, float* dgamma, float* dbeta, int rows, int width, float eps,
                                         int act, int remap_T, int remap_B, int remap_Bp, float drop_p,
                                         uint64_t seed, const float* pool_attn, const float* pool_dctx,
                                         int pool_T, int pool_B, int pool_Bp, float* dx_colsum, void* stream) {
    const bool ident = (act & LOB_LN_IDENTITY) != 0;
    const bool dy16 = (act & LOB_DY_BF16) != 0, dx16 = (act & LOB_OUT_BF16) != 0, x16 = (act & LOB_X_BF16) != 0;
    act &= ~(LOB_DY_BF16 | LOB_OUT_BF16 | LOB_X_BF16);
    if (x16 && !(dy16 && dx16 && (width == 256 || width == 512))) return LOB_E_SHAPE;   // bf16 x: with bf16 dy / dx, width 256 / 512
    if (!x || !dy || !dx || rows <= 0 || width <= 0) return LOB_E_ARG;
    if (!ident && (!gamma || !beta || !dgamma || !dbeta)) return LOB_E_ARG;
    if (width > 64 * LN_MAX_PER_LANE) return LOB_E_SHAPE;
    if (drop_p < 0.f || drop_p >= 1.f) return LOB_E_ARG;
    if (remap_T > 0 && (remap_B <= 0 || remap_Bp < remap_B || rows != remap_T * remap_B)) return LOB_E_SHAPE;
    int blocks = (rows + 3) / 4;
    if (dy16 || dx16) {
        // bf16 gradient streams: the shapes of the mixed path (post-LSTM LayerNorm, width 256 / 512 at H = 128 / 256: dx bf16,
        // dy fp32 or bf16; input-projection LayerNorm, width 128 / 256: dy bf16, dx fp32), vectorised kernels only
        const bool al16b = ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(dy) | reinterpret_cast<uintptr_t>(dx) |
                             reinterpret_cast<uintptr_t>(gamma) | reinterpret_cast<uintptr_t>(beta)) & 15) == 0;
        if (!al16b) return LOB_E_ALIGN;
        if (pool_attn && (!pool_dctx || pool_T <= 0 || pool_Bp <= 0 || rows != pool_T * pool_Bp || remap_T)) return LOB_E_SHAPE;
        const __bf16* dyb = reinterpret_cast<const __bf16*>(dy);
        __bf16* dxb = reinterpret_cast<__bf16*>(dx);
#define LOB_LNB_T(V, L, DYT, DXT, DYP, DXP) hipLaunchKernelGGL((layernorm_act_bwd_vec_kernel<V, L, DYT, DXT>), dim3(blocks), \
                       dim3(256), 0, (hipStream_t)stream, x, gamma, beta, DYP, DXP, dgamma, dbeta, rows, eps, act, remap_T,    \
                       remap_B, remap_Bp, drop_p, seed, pool_attn, pool_dctx, pool_T, pool_B, pool_Bp, dx_colsum)
        if (width == 256) {
            if (blocks > 256 * 8) blocks = 256 * 8;
            if (x16 && ln_lpr16()) {     // all three streams bf16: 32 lanes per row, 16 B per lane and stream
                blocks = (rows + 31) / 32;
                if (blocks > 256 * 8) blocks = 256 * 8;
                hipLaunchKernelGGL((layernorm_act_bwd_vec_kernel<8, 32, __bf16, __bf16, __bf16>), dim3(blocks), dim3(256), 0,
                                   (hipStream_t)stream, reinterpret_cast<const __bf16*>(x), gamma, beta, dyb, dxb, dgamma, dbeta,
                                   rows, eps, act, remap_T, remap_B, remap_Bp, drop_p, seed, pool_attn, pool_dctx, pool_T, pool_B,
                                   pool_Bp, dx_colsum);
            } else if (x16)
                hipLaunchKernelGGL((layernorm_act_bwd_vec_kernel<4, 64, __bf16, __bf16, __bf16>), dim3(blocks), dim3(256), 0,
                                   (hipStream_t)stream, reinterpret_cast<const __bf16*>(x), gamma, beta, dyb, dxb, dgamma, dbeta,
                                   rows, eps, act, remap_T, remap_B, remap_Bp, drop_p, seed, pool_attn, pool_dctx, pool_T, pool_B,
                                   pool_Bp, dx_colsum);
            else if (dy16 && dx16)  LOB_LNB_T(4, 64, __bf16, __bf16, dyb, dxb);
            else if (dx16)          LOB_LNB_T(4, 64, float, __bf16, dy, dxb);
            else                    LOB_LNB_T(4, 64, __bf16, float, dyb, dx);
        } else if (width == 128) {
            blocks = (rows + 31) / 32;
            if (blocks > 256 * 8) blocks = 256 * 8;
            if (dy16 && dx16)       LOB_LNB_T(8, 16, __bf16, __bf16, dyb, dxb);
            else if (dx16)          LOB_LNB_T(8, 16, float, __bf16, dy, dxb);
            else                    LOB_LNB_T(8, 16, __bf16, float, dyb, dx);
        } else if (width == 512) {       // post-LSTM LayerNorm at H = 256
            if (blocks > 256 * 8) blocks = 256 * 8;
            if (x16)                // all three streams bf16 (the last layer's output arrives as bf16 only)
                hipLaunchKernelGGL((layernorm_act_bwd_vec_kernel<8, 64, __bf16, __bf16, __bf16>), dim3(blocks), dim3(256), 0,
                                   (hipStream_t)stream, reinterpret_cast<const __bf16*>(x), gamma, beta, dyb, dxb, dgamma, dbeta,
                                   rows, eps, act, remap_T, remap_B, remap_Bp, drop_p, seed, pool_attn, pool_dctx, pool_T, pool_B,
                                   pool_Bp, dx_colsum);
            else if (dy16 && dx16)  LOB_LNB_T(8, 64, __bf16, __bf16, dyb, dxb);
            else if (dx16)          LOB_LNB_T(8, 64, float, __bf16, dy, dxb);
            else                    LOB_LNB_T(8, 64, __bf16, float, dyb, dx);
        } else {
            return LOB_E_SHAPE;
        }
#undef LOB_LNB_T
        LOB_CHECK_LAUNCH();
        return 0;
    }
    const bool al = ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(dy) | reinterpret_cast<uintptr_t>(dx) |
                      reinterpret_cast<uintptr_t>(gamma) | reinterpret_cast<uintptr_t>(beta)) & 15) == 0;
    if (al && (width == 128 || width == 256 || width == 512)) {
        if (blocks > 256 * 8) blocks = 256 * 8;
#define LOB_LNB_VEC(V) hipLaunchKernelGGL((layernorm_act_bwd_vec_kernel<V>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, \
                       x, gamma, beta, dy, dx, dgamma, dbeta, rows, eps, act, remap_T, remap_B, remap_Bp, drop_p, seed, \
                       pool_attn, pool_dctx, pool_T, pool_B, pool_Bp, dx_colsum)
        if (pool_attn && (!pool_dctx || pool_T <= 0 || pool_Bp <= 0 || rows != pool_T * pool_Bp || remap_T)) return LOB_E_SHAPE;
        if (width == 128 && ln_lpr16()) {
            blocks = (rows + 31) / 32;
            if (blocks > 256 * 8) blocks = 256 * 8;
            hipLaunchKernelGGL((layernorm_act_bwd_vec_kernel<8, 16>), dim3(blocks), dim3(256), 0, (hipStream_t)stream,
                               x, gamma, beta, dy, dx, dgamma, dbeta, rows, eps, act, remap_T, remap_B, remap_Bp, drop_p, seed,
                               pool_attn, pool_dctx, pool_T, pool_B, pool_Bp, dx_colsum);
        }
        else if (width == 128) LOB_LNB_VEC(2); else if (width == 256) LOB_LNB_VEC(4); else LOB_LNB_VEC(8);
#undef LOB_LNB_VEC
        LOB_CHECK_LAUNCH();
        return 0;
    }
    if (pool_attn || dx_colsum) return LOB_E_SHAPE;        // the fused pooling term / column sums exist on the vectorised path only
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(layernorm_act_bwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, gamma, beta, dy,
                       dx, dgamma, dbeta, rows, width, eps, act, remap_T, remap_B, remap_Bp, drop_p, seed);
    LOB_CHECK_LAUNCH();
    return 0;
}

extern "C" int lob_attn_pool_bwd_f32(const void* V, int v_bf16, const float* U, const float* attn, const float* dctx,
                                     const float* w2, float* dV, void* dPreU, int du_bf16, float* dw2,
                                     float* du_colsum, int T, int B, int Bp, int W, int W2, const float* dattn,
                                     void* stream) {
    if (!V || !attn || !dctx) return LOB_E_ARG;
    if (dattn && (!U || du_colsum)) return LOB_E_SHAPE;      // the weights' gradient: generic kernel with a score MLP only
    if (U ? (!w2 || !dPreU || !dw2 || W2 <= 0) : !dV) return LOB_E_ARG;
    if (T <= 0 || B <= 0 || Bp < B || W <= 0) return LOB_E_ARG;
    const size_t smem = ((size_t)2 * T + W) * sizeof(float);
    if (smem > 60 * 1024) return LOB_E_SHAPE;
    if (v_bf16 && du_bf16 && U && !dV && !dattn && ((W == 256 && W2 == 128) || (W == 512 && W2 == 256)) &&
        ((reinterpret_cast<uintptr_t>(V) | reinterpret_cast<uintptr_t>(U) | reinterpret_cast<uintptr_t>(w2) |
          reinterpret_cast<uintptr_t>(dctx)) & 15) == 0 && (reinterpret_cast<uintptr_t>(dPreU) & 7) == 0) {
        const size_t sm2 = ((size_t)2 * T + 4 * W2) * sizeof(float);
        if (W == 256) hipLaunchKernelGGL(attn_pool_bwd_vec_kernel<1>, dim3(B), dim3(256), sm2, (hipStream_t)stream,
                           reinterpret_cast<const __bf16*>(V), U, attn, dctx, w2, reinterpret_cast<__bf16*>(dPreU), dw2, du_colsum, T, Bp);
        else          hipLaunchKernelGGL(attn_pool_bwd_vec_kernel<2>, dim3(B), dim3(256), sm2, (hipStream_t)stream,
                           reinterpret_cast<const __bf16*>(V), U, attn, dctx, w2, reinterpret_cast<__bf16*>(dPreU), dw2, du_colsum, T, Bp);
        LOB_CHECK_LAUNCH();
        return 0;
    }
    if (du_colsum) return LOB_E_SHAPE;        // fused column sums exist on the vectorised path only
#define LOB_APB(VE, UE) hipLaunchKernelGGL((attn_pool_bwd_kernel<VE, UE>), dim3(B), dim3(256), smem, (hipStream_t)stream, \
        reinterpret_cast<const VE*>(V), U, attn, dctx, w2, dV, reinterpret_cast<UE*>(dPreU), dw2, T, Bp, W, W2, dattn)
    if (v_bf16 && du_bf16) LOB_APB(__bf16, __bf16);
    else if (v_bf16) LOB_APB(__bf16, float);
    else if (du_bf16) LOB_APB(float, __bf16);
    else LOB_APB(float, float);
#undef LOB_APB
    LOB_CHECK_LAUNCH();
    return 0;
}
