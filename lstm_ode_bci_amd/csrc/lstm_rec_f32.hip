// Recurrent half of one LSTM layer: all T steps inside ONE persistent kernel.
//
// Fast path (H = 128), exact fp32:
//   grid  = (Bp/32 batch tiles) x (D directions), 256 threads = 4 waves, 1 wave / SIMD.
//   wave w owns hidden units [32w, 32w+32) for all four gates, so the cell update is
//   wave-local.  Its slice of W_hh (4 gates x 32 units x K=128 fp32 = 64 KB) lives in
//   256 VGPRs per lane for the whole kernel (the register file is the only on-chip
//   store large enough: W_hh is 256 KB, LDS is 160 KB).
//   h_{t-1} (32 rows x 128) is exchanged through a double-buffered LDS tile: one
//   s_barrier per time step.  The gate pre-activations P_t (x-part + biases, produced
//   by lob_gate_gemm_x_f32 in accumulator-fragment order) are loaded straight INTO the
//   MFMA accumulators one step ahead, so the recurrent GEMM accumulates on top of them.
//   Per step and wave: 256 x v_mfma_f32_32x32x2_f32 (16,384 cycles) vs ~2.5k cycles of
//   VALU for the 5 transcendentals x 16 elements per lane.
//
// Generic path (any H): VALU dot products, W_hh streamed from L2.  Correct, not fast;
// it exists so that every (H, num_layers, bidirectional) the reference accepts runs.
#include "lob_common.h"

namespace {

constexpr int HS_LD = 132;   // LDS row stride of the h tile (floats): 33 x 16 B -> conflict-free b128

template <bool SAVE>
__global__ __launch_bounds__(256, 1) void lstm_rec_fwd_h128_kernel(
    float* __restrict__ P, const float* __restrict__ Whh, float* __restrict__ Y,
    float* __restrict__ Csave, int T, int Bp) {
    constexpr int H = 128;
    __shared__ __attribute__((aligned(16))) float hs[2 * 32 * HS_LD];

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int bt = blockIdx.x, d = blockIdx.y, D = gridDim.y, NBT = gridDim.x;
    const int l31 = lane & 31, hi = lane >> 5;

    // ---- W_hh slice -> registers.  B operand of step (kb, q, e): W[n][k = 32kb + 16hi + 4q + e]
    f32x4 wr[4][4][4];
    {
        const float* wbase = Whh + (size_t)d * 4 * H * H;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float* row = wbase + (size_t)(g * H + 32 * w + l31) * H + 16 * hi;
#pragma unroll
            for (int kb = 0; kb < 4; ++kb)
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    wr[g][kb][q] = *reinterpret_cast<const f32x4*>(row + 32 * kb + 4 * q);
        }
    }
    for (int i = tid; i < 2 * 32 * HS_LD; i += 256) hs[i] = 0.f;

    float c[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) c[r] = 0.f;

    // per-(d,t,bt) block of P: [w][g][q][lane][4]
    const size_t pstep = (size_t)NBT * 4 * 4 * 1024;               // floats per t
    float* pblk = P + ((size_t)d * T * NBT + bt) * 4 * 4 * 1024 + (size_t)w * 4 * 1024 + lane * 4;
    const size_t cstep = (size_t)NBT * 4 * 1024;
    float* cblk = SAVE ? Csave + ((size_t)d * T * NBT + bt) * 4 * 1024 + (size_t)w * 1024 + lane * 4 : nullptr;

    const int t_first = d ? T - 1 : 0, dt = d ? -1 : 1;
    f32x16 pn[4];
    {
        const float* p = pblk + (size_t)t_first * pstep;
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                f32x4 v = *reinterpret_cast<const f32x4*>(p + g * 1024 + q * 256);
                pn[g][4 * q + 0] = v[0]; pn[g][4 * q + 1] = v[1]; pn[g][4 * q + 2] = v[2]; pn[g][4 * q + 3] = v[3];
            }
    }
    __syncthreads();

    int cur = 0;
    for (int step = 0; step < T; ++step) {
        const int t = t_first + dt * step;
        f32x16 acc[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) acc[g] = pn[g];
        if (step + 1 < T) {
            const float* p = pblk + (size_t)(t + dt) * pstep;
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    f32x4 v = *reinterpret_cast<const f32x4*>(p + g * 1024 + q * 256);
                    pn[g][4 * q + 0] = v[0]; pn[g][4 * q + 1] = v[1]; pn[g][4 * q + 2] = v[2]; pn[g][4 * q + 3] = v[3];
                }
        }
        // ---- z += h_{t-1} * W_hh^T
        const float* hrow = hs + cur * 32 * HS_LD + l31 * HS_LD + 16 * hi;
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
            f32x4 a[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) a[q] = *reinterpret_cast<const f32x4*>(hrow + 32 * kb + 4 * q);
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int g = 0; g < 4; ++g)
                        acc[g] = mfma32(a[q][e], wr[g][kb][q][e], acc[g]);
        }
        // ---- cell update (wave-local), h_t -> LDS (other buffer) and HBM
        float* hnext = hs + (cur ^ 1) * 32 * HS_LD + 32 * w + l31;
        float* yrow = Y + ((size_t)t * Bp + bt * 32) * (D * H) + d * H + 32 * w + l31;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float ig = fast_sigmoid(acc[0][r]);
            const float fg = fast_sigmoid(acc[1][r]);
            const float gg = fast_tanh(acc[2][r]);
            const float og = fast_sigmoid(acc[3][r]);
            c[r] = fg * c[r] + ig * gg;
            const float h = og * fast_tanh(c[r]);
            const int row = acc_row(r, lane);
            hnext[row * HS_LD] = h;
            yrow[(size_t)row * (D * H)] = h;
            if (SAVE) { acc[0][r] = ig; acc[1][r] = fg; acc[2][r] = gg; acc[3][r] = og; }
        }
        if (SAVE) {
            float* p = pblk + (size_t)t * pstep;
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    f32x4 v = {acc[g][4 * q + 0], acc[g][4 * q + 1], acc[g][4 * q + 2], acc[g][4 * q + 3]};
                    *reinterpret_cast<f32x4*>(p + g * 1024 + q * 256) = v;
                }
            float* cp = cblk + (size_t)t * cstep;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                f32x4 v = {c[4 * q + 0], c[4 * q + 1], c[4 * q + 2], c[4 * q + 3]};
                *reinterpret_cast<f32x4*>(cp + q * 256) = v;
            }
        }
        __syncthreads();
        cur ^= 1;
    }
}

// ------------------------------------------------------------------------------------
// Generic path: P row-major (T*Bp, D*4H); RB batch rows per workgroup.
// ------------------------------------------------------------------------------------
constexpr int RB = 4;

template <bool SAVE>
__global__ __launch_bounds__(256) void lstm_rec_fwd_generic_kernel(
    float* __restrict__ P, const float* __restrict__ Whh, float* __restrict__ Y,
    float* __restrict__ Csave, int T, int Bp, int H) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* hcur = sm;               // [RB][H]
    float* hnew = sm + RB * H;      // [RB][H]
    float* cst = sm + 2 * RB * H;   // [RB][H]
    const int tid = threadIdx.x, d = blockIdx.y, D = gridDim.y;
    const int b0 = blockIdx.x * RB;
    const float* W = Whh + (size_t)d * 4 * H * H;
    for (int i = tid; i < RB * H; i += blockDim.x) { hcur[i] = 0.f; cst[i] = 0.f; }
    __syncthreads();
    const int t_first = d ? T - 1 : 0, dt = d ? -1 : 1;
    for (int step = 0; step < T; ++step) {
        const int t = t_first + dt * step;
        for (int idx = tid; idx < RB * H; idx += blockDim.x) {
            const int r = idx / H, u = idx % H;
            const int b = b0 + r;
            if (b >= Bp) continue;
            float* prow = P + ((size_t)t * Bp + b) * (D * 4 * H) + (size_t)d * 4 * H;
            float z[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float* wrow = W + (size_t)(g * H + u) * H;
                float s = prow[g * H + u];
                for (int k = 0; k < H; ++k) s = fmaf(hcur[r * H + k], wrow[k], s);
                z[g] = s;
            }
            const float ig = fast_sigmoid(z[0]), fg = fast_sigmoid(z[1]);
            const float gg = fast_tanh(z[2]), og = fast_sigmoid(z[3]);
            const float cn = fg * cst[idx] + ig * gg;
            const float h = og * fast_tanh(cn);
            cst[idx] = cn;
            hnew[idx] = h;
            Y[((size_t)t * Bp + b) * (D * H) + d * H + u] = h;
            if (SAVE) {
                prow[0 * H + u] = ig; prow[1 * H + u] = fg; prow[2 * H + u] = gg; prow[3 * H + u] = og;
                Csave[(((size_t)d * T + t) * Bp + b) * H + u] = cn;
            }
        }
        __syncthreads();
        float* tmp = hcur; hcur = hnew; hnew = tmp;
    }
}

}  // namespace

extern "C" int lob_lstm_rec_fwd_f32(float* P, const float* Whh, float* Y, float* Csave,
                                    int T, int Bp, int H, int D, int save, void* stream) {
    if (!P || !Whh || !Y || T <= 0 || Bp <= 0 || H <= 0 || (D != 1 && D != 2)) return LOB_E_ARG;
    if (save && !Csave) return LOB_E_ARG;
    hipStream_t s = (hipStream_t)stream;
    if (H == 128) {
        if (Bp % 32) return LOB_E_SHAPE;
        if ((reinterpret_cast<uintptr_t>(P) | reinterpret_cast<uintptr_t>(Whh) |
             reinterpret_cast<uintptr_t>(Csave)) & 15) return LOB_E_ALIGN;
        const dim3 grid(Bp / 32, D), block(256);
        if (save) hipLaunchKernelGGL((lstm_rec_fwd_h128_kernel<true>), grid, block, 0, s, P, Whh, Y, Csave, T, Bp);
        else      hipLaunchKernelGGL((lstm_rec_fwd_h128_kernel<false>), grid, block, 0, s, P, Whh, Y, Csave, T, Bp);
    } else {
        const dim3 grid((Bp + RB - 1) / RB, D), block(256);
        const size_t smem = (size_t)3 * RB * H * sizeof(float);
        if (smem > 64 * 1024) return LOB_E_SHAPE;
        if (save) hipLaunchKernelGGL((lstm_rec_fwd_generic_kernel<true>), grid, block, smem, s, P, Whh, Y, Csave, T, Bp, H);
        else      hipLaunchKernelGGL((lstm_rec_fwd_generic_kernel<false>), grid, block, smem, s, P, Whh, Y, Csave, T, Bp, H);
    }
    LOB_CHECK_LAUNCH();
    return 0;
}
