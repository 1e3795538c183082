// HBM-bound row-wise pieces of the forward pass: LayerNorm(+GELU+dropout, + the
// (b,t)->(t,b) relayout), additive-attention pooling over time, row softmax.
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
        const float mean = wave_sum(s) / (float)width;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < LN_MAX_PER_LANE; ++i) {
            const int c = lane + 64 * i;
            const float dlt = (i < per && c < width) ? v[i] - mean : 0.f;
            q += dlt * dlt;
        }
        const float rstd = rsqrtf(wave_sum(q) / (float)width + eps);
        int orow = row;
        if (remap_T > 0) { const int b = row / remap_T, t = row % remap_T; orow = t * remap_Bp + b; }
        float* y = out + (size_t)orow * width;
#pragma unroll
        for (int i = 0; i < LN_MAX_PER_LANE; ++i) {
            const int c = lane + 64 * i;
            if (i < per && c < width) {
                float o = (v[i] - mean) * rstd * gamma[c] + beta[c];
                o = apply_act(o, act);
                if (drop_p > 0.f) o *= lob_dropout_scale(seed, (uint64_t)orow * width + c, drop_p);
                y[c] = o;
            }
        }
    }
}

// One workgroup per window b.
__global__ __launch_bounds__(256) void attn_pool_fwd_kernel(
    const float* __restrict__ V, const float* __restrict__ U, const float* __restrict__ w2,
    const float* __restrict__ b2, float* __restrict__ ctx, float* __restrict__ attn,
    int T, int Bp, int W, int W2) {
    extern __shared__ __attribute__((aligned(16))) float sc[];   // [T] scores, then weights
    __shared__ float red[8];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float bias2 = b2 ? b2[0] : 0.f;
    for (int t = wave; t < T; t += 4) {
        const float* u = U + ((size_t)t * Bp + b) * W2;
        float s = 0.f;
        for (int j = lane; j < W2; j += 64) s = fmaf(u[j], w2[j], s);
        s = wave_sum(s);
        if (lane == 0) sc[t] = s + bias2;
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
        const float* v = V + (size_t)b * W + c;
        float acc = 0.f;
        for (int t = 0; t < T; ++t) acc = fmaf(sc[t], v[(size_t)t * Bp * W], acc);
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

}  // namespace

extern "C" int lob_dropout_f32(const float* in, float* out, int64_t n, float p, uint64_t seed, void* stream) {
    if (!in || !out || n <= 0 || p < 0.f || p >= 1.f) return LOB_E_ARG;
    int64_t blocks = (n + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL(dropout_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, in, out, (size_t)n, p, seed);
    LOB_CHECK_LAUNCH();
    return 0;
}

extern "C" int lob_layernorm_act_f32(const float* in, const float* gamma, const float* beta,
                                     float* out, int rows, int width, float eps, int act,
                                     int remap_T, int remap_B, int remap_Bp,
                                     float drop_p, uint64_t seed, void* stream) {
    if (!in || !gamma || !beta || !out || rows <= 0 || width <= 0) return LOB_E_ARG;
    if (width > 64 * LN_MAX_PER_LANE) return LOB_E_SHAPE;
    if (drop_p < 0.f || drop_p >= 1.f) return LOB_E_ARG;
    if (remap_T > 0 && (remap_B <= 0 || remap_Bp < remap_B || rows != remap_T * remap_B)) return LOB_E_SHAPE;
    const int waves_per_block = 4;
    int blocks = (rows + waves_per_block - 1) / waves_per_block;
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL(layernorm_act_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream,
                       in, gamma, beta, out, rows, width, eps, act, remap_T, remap_B, remap_Bp, drop_p, seed);
    LOB_CHECK_LAUNCH();
    return 0;
}

extern "C" int lob_attn_pool_fwd_f32(const float* V, const float* U, const float* w2, const float* b2,
                                     float* ctx, float* attn, int T, int B, int Bp, int W, int W2,
                                     void* stream) {
    if (!V || !U || !w2 || !ctx || !attn || T <= 0 || B <= 0 || Bp < B || W <= 0 || W2 <= 0) return LOB_E_ARG;
    if ((size_t)T * sizeof(float) > 60 * 1024) return LOB_E_SHAPE;
    hipLaunchKernelGGL(attn_pool_fwd_kernel, dim3(B), dim3(256), (size_t)T * sizeof(float), (hipStream_t)stream,
                       V, U, w2, b2, ctx, attn, T, Bp, W, W2);
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
