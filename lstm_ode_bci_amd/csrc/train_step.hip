// The steps either side of fwd+bwd in the reference's training / attribution loops, as device kernels over
// FLAT parameter / gradient buffers (one launch per step instead of one per tensor):
//   weighted cross-entropy + its gradient + the accuracy count   (04_lstm_model.py:430-435, 488-509)
//   global gradient norm, clip, AdamW                            (04_lstm_model.py:438, 495-505)
//   |d logit / d x| averaged over time, summed over windows      (07_explainability.py:254-258)
// All HBM / launch-latency bound: 1.14 M parameters = 4.5 MB per stream.
#include "lob_common.h"
#include <math.h>

namespace {

__device__ __forceinline__ float block_sum(float v, float* red) {     // 256..1024 threads, red[16]
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    float s = 0.f;
    for (int i = 0; i < nw; ++i) s += red[i];
    return s;
}

// One workgroup.  loss = sum_i w[y_i] nll_i / sum_i w[y_i];  dlogits_i = scale * w[y_i]/sum_w * (softmax_i - onehot).
__global__ __launch_bounds__(1024) void weighted_ce_kernel(
    const float* __restrict__ logits, const int64_t* __restrict__ target, const float* __restrict__ cw,
    float* __restrict__ loss_out, float* __restrict__ dlogits, int* __restrict__ correct_out,
    int B, int C, float scale) {
    __shared__ float red[16];
    float sw = 0.f, sl = 0.f, sc = 0.f;
    for (int i = threadIdx.x; i < B; i += blockDim.x) {
        const float* z = logits + (size_t)i * C;
        const int y = (int)target[i];
        float m = z[0];
        int am = 0;
        for (int c = 1; c < C; ++c) if (z[c] > m) { m = z[c]; am = c; }
        float l = 0.f;
        for (int c = 0; c < C; ++c) l += expf(z[c] - m);
        if (y >= 0 && y < C) {
            const float w = cw ? cw[y] : 1.f;
            sw += w;
            sl += w * (logf(l) + m - z[y]);
            sc += (am == y) ? 1.f : 0.f;
        }
    }
    sw = block_sum(sw, red);
    sl = block_sum(sl, red);
    sc = block_sum(sc, red);
    if (threadIdx.x == 0) {
        loss_out[0] = sl / sw;
        if (correct_out) correct_out[0] = (int)(sc + 0.5f);
    }
    if (!dlogits) return;
    const float k = scale / sw;
    for (int i = threadIdx.x; i < B; i += blockDim.x) {
        const float* z = logits + (size_t)i * C;
        float* dz = dlogits + (size_t)i * C;
        const int y = (int)target[i];
        const bool ok = (y >= 0 && y < C);
        float m = z[0];
        for (int c = 1; c < C; ++c) m = fmaxf(m, z[c]);
        float l = 0.f;
        for (int c = 0; c < C; ++c) l += expf(z[c] - m);
        const float wk = ok ? (cw ? cw[y] : 1.f) * k : 0.f;
        const float inv = 1.0f / l;
        for (int c = 0; c < C; ++c) dz[c] = wk * (expf(z[c] - m) * inv - (c == y ? 1.f : 0.f));
    }
}

__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ x, size_t n, float* __restrict__ out,
                                                    float* __restrict__ partial) {
    __shared__ float red[16];
    const size_t n4 = n >> 2, stride = (size_t)gridDim.x * blockDim.x;
    const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    float s = 0.f;
    for (size_t i = gid; i < n4; i += stride) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(x + 4 * i);
        s += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
    }
    if (gid < (n & 3)) { const float t = x[4 * n4 + gid]; s += t * t; }      // ragged tail
    s = block_sum(s, red);
    if (threadIdx.x == 0) {
        if (partial) partial[blockIdx.x] = s;      // deterministic path: summed in block order by sumsq_finish_kernel
        else         atomicAdd(out, s);
    }
}

// out[0] += partial[0] + ... + partial[nb-1], always in the same order: the squared gradient norm (and with it the
// clip coefficient and every weight) is then bit-identical on all data-parallel ranks that hold the same gradient.
__global__ __launch_bounds__(512) void sumsq_finish_kernel(const float* __restrict__ partial, int nb, float* __restrict__ out) {
    __shared__ float buf[512];
    buf[threadIdx.x] = (int)threadIdx.x < nb ? partial[threadIdx.x] : 0.f;
    __syncthreads();
    for (int w = 256; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) buf[threadIdx.x] += buf[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] += buf[0];
}

__device__ __forceinline__ float clip_coef_of(const float* normsq, float max_norm, float grad_scale) {
    // torch.nn.utils.clip_grad_norm_: clamp(max_norm / (total_norm + 1e-6), max = 1); normsq = sum g^2 of the
    // UNSCALED gradient, the norm that is clipped is that of grad_scale * g
    return normsq ? fminf(1.0f, max_norm / (grad_scale * sqrtf(normsq[0]) + 1e-6f)) : 1.0f;
}

__global__ __launch_bounds__(256) void clip_scale_kernel(float* __restrict__ g, size_t n, const float* __restrict__ normsq,
                                                         float max_norm) {
    const float k = clip_coef_of(normsq, max_norm, 1.0f);
    if (k >= 1.0f) return;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) g[i] *= k;
}

// torch.optim.AdamW (decoupled decay first), on g_eff = g * grad_scale * clip.
__global__ __launch_bounds__(256) void adamw_kernel(
    float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, size_t n,
    float lr, float beta1, float beta2, float eps, float weight_decay, float step_size, float inv_sqrt_bc2,
    const float* __restrict__ normsq, float max_norm, float grad_scale) {
    const float gs = grad_scale * clip_coef_of(normsq, max_norm, grad_scale);
    const float decay = 1.0f - lr * weight_decay;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const float gi = g[i] * gs;
        const float pi = p[i] * decay;
        const float mi = beta1 * m[i] + (1.0f - beta1) * gi;
        const float vi = beta2 * v[i] + (1.0f - beta2) * gi * gi;
        m[i] = mi;
        v[i] = vi;
        p[i] = pi - step_size * (mi / (sqrtf(vi) * inv_sqrt_bc2 + eps));
    }
}

// out[c] += scale * sum over rows of |gx[row][c]|; block = (64 columns, 4 rows in flight).
__global__ __launch_bounds__(256) void abs_colsum_kernel(const float* __restrict__ gx, size_t rows, int C, float scale,
                                                         float* __restrict__ out) {
    __shared__ float red[4][64];
    const int cx = threadIdx.x & 63, ry = threadIdx.x >> 6;
    for (int c0 = 0; c0 < C; c0 += 64) {
        const int c = c0 + cx;
        float s = 0.f;
        if (c < C)
            for (size_t r = (size_t)blockIdx.x * 4 + ry; r < rows; r += (size_t)gridDim.x * 4) s += fabsf(gx[r * C + c]);
        red[ry][cx] = s;
        __syncthreads();
        if (ry == 0 && c < C) atomicAdd(out + c, scale * (red[0][cx] + red[1][cx] + red[2][cx] + red[3][cx]));
        __syncthreads();
    }
}

// out[row][0..Cp) = bf16(in[row][0..C)), zero-padded: one thread per 8 output elements (one 16-B store).
__global__ __launch_bounds__(256) void pad_cast_bf16_kernel(const float* __restrict__ in, __bf16* __restrict__ out,
                                                            size_t rows, int C, int Cp) {
    typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
    const int cpr = Cp >> 3;                                   // chunks per row
    const size_t total = rows * cpr, stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const size_t row = i / cpr;
        const int c0 = (int)(i % cpr) * 8;
        const float* src = in + row * C + c0;
        bf16x8_t v;
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = (__bf16)(c0 + j < C ? src[j] : 0.f);
        *reinterpret_cast<bf16x8_t*>(out + row * Cp + c0) = v;
    }
}

}  // namespace

extern "C" int lob_pad_cast_bf16(const float* in, void* out, int64_t rows, int C, int Cp, void* stream) {
    if (!in || !out || rows <= 0 || C <= 0 || Cp < C || (Cp & 7)) return LOB_E_ARG;
    if (reinterpret_cast<uintptr_t>(out) & 15) return LOB_E_ALIGN;
    int64_t blocks = (rows * (Cp >> 3) + 255) / 256;
    if (blocks > 256 * 32) blocks = 256 * 32;
    hipLaunchKernelGGL(pad_cast_bf16_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, in,
                       reinterpret_cast<__bf16*>(out), (size_t)rows, C, Cp);
    LOB_CHECK_LAUNCH();
    return 0;
}

extern "C" int lob_weighted_ce_f32(const float* logits, const int64_t* target, const float* class_weight,
                                   float* loss, float* dlogits, int* correct, int B, int C, float scale,
                                   void* stream) {
    if (!logits || !target || !loss || B <= 0 || C <= 0) return LOB_E_ARG;
    if (C > 4096) return LOB_E_SHAPE;
    const int threads = B >= 1024 ? 1024 : (B > 256 ? 512 : 256);
    hipLaunchKernelGGL(weighted_ce_kernel, dim3(1), dim3(threads), 0, (hipStream_t)stream, logits, target,
                       class_weight, loss, dlogits, correct, B, C, scale);
    LOB_CHECK_LAUNCH();
    return 0;
}

extern "C" int lob_sumsq_f32(const float* x, int64_t n, float* out, float* scratch, void* stream) {
    if (!x || !out || n <= 0) return LOB_E_ARG;
    if (reinterpret_cast<uintptr_t>(x) & 15) return LOB_E_ALIGN;
    int64_t blocks = (n + 1023) / 1024;
    if (blocks > 512) blocks = 512;
    hipLaunchKernelGGL(sumsq_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, (size_t)n, out, scratch);
    if (scratch)
        hipLaunchKernelGGL(sumsq_finish_kernel, dim3(1), dim3(512), 0, (hipStream_t)stream, scratch, (int)blocks, out);
    LOB_CHECK_LAUNCH();
    return 0;
}

extern "C" int lob_clip_scale_f32(float* g, int64_t n, const float* normsq, float max_norm, void* stream) {
    if (!g || !normsq || n <= 0 || !(max_norm > 0.f)) return LOB_E_ARG;
    int64_t blocks = (n + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(clip_scale_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, g, (size_t)n,
                       normsq, max_norm);
    LOB_CHECK_LAUNCH();
    return 0;
}

extern "C" int lob_adamw_f32(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1,
                             float beta2, float eps, float weight_decay, int64_t step, const float* normsq,
                             float max_norm, float grad_scale, void* stream) {
    if (!p || !g || !m || !v || n <= 0 || step <= 0) return LOB_E_ARG;
    if (!(beta1 >= 0.f && beta1 < 1.f && beta2 >= 0.f && beta2 < 1.f) || (normsq && !(max_norm > 0.f))) return LOB_E_ARG;
    const double bc1 = 1.0 - pow((double)beta1, (double)step);
    const double bc2 = 1.0 - pow((double)beta2, (double)step);
    int64_t blocks = (n + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(adamw_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, p, g, m, v, (size_t)n,
                       lr, beta1, beta2, eps, weight_decay, (float)((double)lr / bc1), (float)(1.0 / sqrt(bc2)),
                       normsq, max_norm, grad_scale);
    LOB_CHECK_LAUNCH();
    return 0;
}

extern "C" int lob_abs_colsum_f32(const float* gx, int64_t rows, int C, float scale, float* out, void* stream) {
    if (!gx || !out || rows <= 0 || C <= 0) return LOB_E_ARG;
    int64_t blocks = (rows + 3) / 4;
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(abs_colsum_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, gx, (size_t)rows,
                       C, scale, out);
    LOB_CHECK_LAUNCH();
    return 0;
}
