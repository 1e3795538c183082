// Exact-fp32 recurrent forward kernel (H = 128) on 16-row sub-tiles and v_mfma_f32_16x16x4_f32.
//
// Why a second fp32 forward kernel: at B = 1024 (BASELINE configs[1]) the 32-row kernel of lstm_rec_f32.hip
// has a grid of 64 workgroups on 256 CUs.  Here a workgroup owns 16 batch rows: twice the workgroups (the
// recurrent part of a B = 1024 forward drops from 2.8 to 1.4 ms per layer), and at B = 4096 it is still
// 2-8 % faster than the 32-row kernel (two rounds of half-length steps).
// (Tried and dropped: two 16-row sub-tiles per workgroup, software-pipelined so that one sub-tile's cell
//  update overlaps the other's MFMAs.  hipcc keeps the 256 MFMAs and the ~330 VALU ops of a segment as two
//  blocks, and sched_group_barrier pipelines of that size did not change the emitted order: no gain.)
// Same W_hh-in-256-VGPRs residency, same fragment-order P / saved-gates / c layouts as the 32-row kernel
// (a 16-row sub-tile is one q-half of the 32-row fragment block), so the gate GEMM and the BPTT kernel
// are unchanged.
//
// MFMA 16x16x4: lane l feeds A[row = l&15][k = l>>4] and B[k = l>>4][col = l&15]; D: 4 registers,
// col = l&15, row = 4*(l>>4) + reg.  k-permutation: step s (0..31) contracts k = 32*(l>>4) + s, so a lane
// reads 32 CONTIGUOUS floats of its h row (8 x ds_read_b128) and of its W_hh row.
#include "lob_common.h"

namespace {

constexpr int H = 128;
constexpr int HLD = 132;       // h tile row stride (floats): 33 x 16 B, odd -> conflict-free b128

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

template <bool SAVE>
__global__ __launch_bounds__(256, 1) void lstm_rec_fwd_h128_s16_kernel(
    float* __restrict__ P, const float* __restrict__ Whh, float* __restrict__ Y,
    float* __restrict__ Csave, int T, int Bp) {
    constexpr int NSUB = 1;
    __shared__ __attribute__((aligned(16))) float hs[2 * 16 * HLD];      // [buf][16 rows][HLD]
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int d = blockIdx.y, D = gridDim.y, NBT = Bp >> 5;
    const int c16 = lane & 15, rq = lane >> 4;
    // 16-row tile index -> (32-row fragment block bt, half s0)
    const int tile16 = blockIdx.x * NSUB;
    const int bt = tile16 >> 1, s0 = tile16 & 1;

    // ---- W_hh slice -> registers: wr[g][cbu][s] = W[g*128 + 32w + 16cbu + c16][32*rq + s]
    float wr[4][2][32];
    {
        const float* wbase = Whh + (size_t)d * 4 * H * H;
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int cbu = 0; cbu < 2; ++cbu) {
                const float* row = wbase + (size_t)(g * H + 32 * w + 16 * cbu + c16) * H + 32 * rq;
#pragma unroll
                for (int s4 = 0; s4 < 8; ++s4) {
                    const f32x4 v = *reinterpret_cast<const f32x4*>(row + 4 * s4);
                    wr[g][cbu][4 * s4] = v[0]; wr[g][cbu][4 * s4 + 1] = v[1];
                    wr[g][cbu][4 * s4 + 2] = v[2]; wr[g][cbu][4 * s4 + 3] = v[3];
                }
            }
    }
    for (int i = tid; i < 2 * NSUB * 16 * HLD; i += 256) hs[i] = 0.f;

    // fragment addressing (see header): for sub-tile half s, gate g, unit block cbu the 4 accumulator rows
    // of a lane are ONE float4 at [w][g][q = 2s + (rq>>1)][lane' = (rq&1)*32 + 16cbu + c16][0..3]
    const size_t pstep = (size_t)NBT * 16 * 1024, cstep = (size_t)NBT * 4096;
    float* pblk = P + ((size_t)d * T * NBT + bt) * 16 * 1024 + (size_t)w * 4096;
    float* cblk = SAVE ? Csave + ((size_t)d * T * NBT + bt) * 4096 + (size_t)w * 1024 : nullptr;
    const unsigned lane_p = (unsigned)(((rq >> 1) * 256 + ((rq & 1) * 32 + c16) * 4));   // + q0*256 + cbu*64
    const int DH = D * H;
    const unsigned y_off = (unsigned)(4 * rq * DH + c16);

    const int t_first = d ? T - 1 : 0, dt = d ? -1 : 1;
    f32x4 pn[NSUB][4][2];          // P of the next step, per sub-tile / gate / unit block
    float c[NSUB][2][4];
#pragma unroll
    for (int s = 0; s < NSUB; ++s)
#pragma unroll
        for (int cbu = 0; cbu < 2; ++cbu)
#pragma unroll
            for (int j = 0; j < 4; ++j) c[s][cbu][j] = 0.f;

    auto load_p = [&](int t, int s, f32x4 (&dst)[4][2]) {
        const float* p = pblk + (size_t)t * pstep + (s0 + s) * 512 + lane_p;
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int cbu = 0; cbu < 2; ++cbu) dst[g][cbu] = *reinterpret_cast<const f32x4*>(p + g * 1024 + cbu * 64);
    };
#pragma unroll
    for (int s = 0; s < NSUB; ++s) load_p(t_first, s, pn[s]);
    __syncthreads();

    // z += h_{t-1} W_hh^T for one sub-tile: 8 column blocks x 32 k-steps
    auto gemm_sub = [&](const float* hsub, f32x4 (&acc)[4][2]) {
        const float* hrow = hsub + c16 * HLD + 32 * rq;
#pragma unroll
        for (int s4 = 0; s4 < 8; ++s4) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(hrow + 4 * s4);
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int g = 0; g < 4; ++g)
#pragma unroll
                    for (int cbu = 0; cbu < 2; ++cbu)
                        acc[g][cbu] = mfma16(a[e], wr[g][cbu][4 * s4 + e], acc[g][cbu]);
        }
    };
    // gates + cell update for one sub-tile; writes h_t to LDS (next buffer), Y, and the saved activations
    auto cell_sub = [&](int t, int s, f32x4 (&acc)[4][2], float* hsub_next) {
        float* yrow = Y + ((size_t)t * Bp + bt * 32 + (s0 + s) * 16) * DH + d * H + 32 * w;
#pragma unroll
        for (int cbu = 0; cbu < 2; ++cbu) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float ig = fast_sigmoid(acc[0][cbu][j]);
                const float fg = fast_sigmoid(acc[1][cbu][j]);
                const float gg = fast_tanh(acc[2][cbu][j]);
                const float og = fast_sigmoid(acc[3][cbu][j]);
                c[s][cbu][j] = __builtin_fmaf(fg, c[s][cbu][j], ig * gg);
                const float h = og * fast_tanh(c[s][cbu][j]);
                hsub_next[(4 * rq + j) * HLD + 32 * w + 16 * cbu + c16] = h;
                (yrow + (size_t)j * DH + 16 * cbu)[y_off] = h;
                if (SAVE) { acc[0][cbu][j] = ig; acc[1][cbu][j] = fg; acc[2][cbu][j] = gg; acc[3][cbu][j] = og; }
            }
        }
        if (SAVE) {
            float* p = pblk + (size_t)t * pstep + (s0 + s) * 512 + lane_p;
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int cbu = 0; cbu < 2; ++cbu) *reinterpret_cast<f32x4*>(p + g * 1024 + cbu * 64) = acc[g][cbu];
            float* cp = cblk + (size_t)t * cstep + (s0 + s) * 512 + lane_p;
#pragma unroll
            for (int cbu = 0; cbu < 2; ++cbu) {
                f32x4 v = {c[s][cbu][0], c[s][cbu][1], c[s][cbu][2], c[s][cbu][3]};
                *reinterpret_cast<f32x4*>(cp + cbu * 64) = v;
            }
        }
    };

    int cur = 0;
    for (int step = 0; step < T; ++step) {
        const int t = t_first + dt * step;
        f32x4 acc[4][2];
#pragma unroll
        for (int g = 0; g < 4; ++g) { acc[g][0] = pn[0][g][0]; acc[g][1] = pn[0][g][1]; }
        if (step + 1 < T) load_p(t + dt, 0, pn[0]);
        gemm_sub(hs + cur * 16 * HLD, acc);
        cell_sub(t, 0, acc, hs + (cur ^ 1) * 16 * HLD);
        __syncthreads();
        cur ^= 1;
    }
}

}  // namespace

// Internal entry point used by lob_lstm_rec_fwd_f32 (lstm_rec_f32.hip): 16-row tiles, grid Bp/16 x D.
int lob_rec_fwd_s16(float* P, const float* Whh, float* Y, float* Csave, int T, int Bp, int D, int save, hipStream_t s) {
    const dim3 grid(Bp / 16, D), block(256);
    if (save) hipLaunchKernelGGL((lstm_rec_fwd_h128_s16_kernel<true>), grid, block, 0, s, P, Whh, Y, Csave, T, Bp);
    else      hipLaunchKernelGGL((lstm_rec_fwd_h128_s16_kernel<false>), grid, block, 0, s, P, Whh, Y, Csave, T, Bp);
    LOB_CHECK_LAUNCH();
    return 0;
}
