// Mixed-precision recurrent FORWARD kernel (H = 128) for small batches: 16 rows per workgroup, EIGHT waves.
//
// Below about one workgroup per CU (B <= 2048) the recurrent kernels are not bandwidth-bound but LATENCY-bound: a layer is
// 256 dependent steps, and a serving-style call of B = 1..512 windows costs 3 x 256 x (time of one step) whatever the
// batch (tools/latency_probe.py: 1.20 ms per mixed forward from B = 1 to B = 128 with the four-wave kernel of
// lstm_rec_bf16_s16.hip, 1.55 us per step).  Inside a step a wave's work is serial -- 32 MFMAs, then the gate
// activations of its 8 elements per lane (40 transcendentals), the LDS hand-over, the barrier -- so the step shrinks
// when the SAME tile is spread over more waves: here wave w8 owns 16 hidden columns (32 (w8 >> 1) + 16 (w8 & 1)) of all
// four gates: 16 MFMAs and 4 elements per lane per step, W_hh in 64 VGPRs.  Inference only (no saved gates, no dropout);
// same fragment-order bf16 P as the other H = 128 kernels (include/lob.h).  Outputs: fp32 Y (last layer, for the
// LayerNorm) or bf16 Y (the next layer's GEMM operand).
#include "lob_common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

constexpr int H = 128;
constexpr int HB_LD = 136;     // bf16 h tile row stride (272 B = 17 x 16 B, odd -> conflict-free b128)
constexpr int YF_LD = 132;     // fp32 h staging row stride

template <bool YF32>
__global__ __launch_bounds__(512, 2) void lstm_rec_fwd_h128_bf16_w8_kernel(
    const __bf16* __restrict__ P, const float* __restrict__ Whh, float* __restrict__ Y, __bf16* __restrict__ Y16p,
    int T, int Bp) {
    __shared__ __attribute__((aligned(16))) __bf16 hs[2 * 16 * HB_LD];
    __shared__ __attribute__((aligned(16))) float yfs[YF32 ? 2 * 16 * YF_LD : 4];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w8 = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wcol = w8 >> 1, cbu = w8 & 1;
    const int d = blockIdx.y, D = gridDim.y, NBT = Bp >> 5;
    const int c16 = lane & 15, rq = lane >> 4;
    const int bt = blockIdx.x >> 1, s0 = blockIdx.x & 1;
    const int col = 32 * wcol + 16 * cbu + c16;

    // B fragments: wr[g][ks] = W_hh[g*128 + col][32 ks + 8 rq .. + 7]
    bf16x8 wr[4][4];
    {
        const float* wbase = Whh + (size_t)d * 4 * H * H;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float* row = wbase + (size_t)(g * H + col) * H + 8 * rq;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const f32x4 a = *reinterpret_cast<const f32x4*>(row + 32 * ks), b = *reinterpret_cast<const f32x4*>(row + 32 * ks + 4);
                bf16x8 r = {(__bf16)a[0], (__bf16)a[1], (__bf16)a[2], (__bf16)a[3], (__bf16)b[0], (__bf16)b[1], (__bf16)b[2], (__bf16)b[3]};
                wr[g][ks] = r;
            }
        }
    }
    for (int i = tid; i < 2 * 16 * HB_LD; i += 512) hs[i] = (__bf16)0.f;
    float c[4] = {0.f, 0.f, 0.f, 0.f};

    // bf16 fragment order: gate g of this 16-row half = [q pair s0][lane' = (rq & 1) * 32 + 16 cbu + c16][8]; this lane's
    // rows 4 rq .. + 3 are the 4 elements at (rq >> 1) * 4
    const size_t pstep = (size_t)NBT * 16 * 1024;
    const __bf16* pblk = P + ((size_t)d * T * NBT + bt) * 16 * 1024 + (size_t)wcol * 4096 + s0 * 512 +
                         ((rq & 1) * 32 + 16 * cbu + c16) * 8 + (rq >> 1) * 4;
    const int DH = D * H;
    const int t_first = d ? T - 1 : 0, dt = d ? -1 : 1;
    const int row0 = bt * 32 + s0 * 16;

    bf16x4 pa[4], pb[4];
    auto load_p = [&](int t, bf16x4 (&dst)[4]) {
        const __bf16* p = pblk + (size_t)t * pstep;
#pragma unroll
        for (int g = 0; g < 4; ++g) dst[g] = *reinterpret_cast<const bf16x4*>(p + g * 1024);
    };
    load_p(t_first, pa);
    if (T > 1) load_p(t_first + dt, pb);
    __syncthreads();

    auto one_step = [&](int step, bf16x4 (&praw)[4], int cur) {
        const int t = t_first + dt * step;
        f32x4 acc[4];
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[g][e] = (float)praw[g][e];
        if (step + 2 < T) load_p(t + 2 * dt, praw);
        const __bf16* hrow = hs + cur * 16 * HB_LD + c16 * HB_LD + 8 * rq;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const bf16x8 a = *reinterpret_cast<const bf16x8*>(hrow + 32 * ks);
#pragma unroll
            for (int g = 0; g < 4; ++g) acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, wr[g][ks], acc[g], 0, 0, 0);
        }
        __bf16* hnext = hs + (cur ^ 1) * 16 * HB_LD + 4 * rq * HB_LD + col;
        float* ynext = yfs + (YF32 ? (cur ^ 1) * 16 * YF_LD + 4 * rq * YF_LD + col : 0);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float ig = fast_sigmoid(acc[0][j]), fg = fast_sigmoid(acc[1][j]);
            const float gg = fast_tanh(acc[2][j]), og = fast_sigmoid(acc[3][j]);
            c[j] = fg * c[j] + ig * gg;
            const float h = og * fast_tanh(c[j]);
            hnext[j * HB_LD] = (__bf16)h;
            if (YF32) ynext[j * YF_LD] = h;
        }
        __syncthreads();
        if (YF32) {             // 16 rows x 512 B: one 16-B store per thread
            const int row = tid >> 5, c4 = (tid & 31) * 4;
            const f32x4 v = *reinterpret_cast<const f32x4*>(yfs + (cur ^ 1) * 16 * YF_LD + row * YF_LD + c4);
            *reinterpret_cast<f32x4*>(Y + ((size_t)t * Bp + row0 + row) * DH + d * H + c4) = v;
        } else if (tid < 256) { // 16 rows x 256 B of bf16: one 16-B store per thread of the first four waves
            const int row = tid >> 4, c8 = (tid & 15) * 8;
            const bf16x8 hv = *reinterpret_cast<const bf16x8*>(hs + (cur ^ 1) * 16 * HB_LD + row * HB_LD + c8);
            *reinterpret_cast<bf16x8*>(Y16p + ((size_t)t * Bp + row0 + row) * DH + d * H + c8) = hv;
        }
    };

    for (int step = 0; step < T; step += 2) {
        one_step(step, pa, 0);
        if (step + 1 < T) one_step(step + 1, pb, 1);
    }
}

}  // namespace

// Internal entry point used by lob_lstm_rec_fwd_bf16 (lstm_rec_bf16.hip): inference, bf16 P, exactly one of Y (fp32) / Y16.
int lob_rec_fwd_bf16_w8(const void* P, const float* Whh, float* Y, void* Y16, int T, int Bp, int D, hipStream_t s) {
    const dim3 grid(Bp / 16, D), block(512);
    if (Y) hipLaunchKernelGGL((lstm_rec_fwd_h128_bf16_w8_kernel<true>), grid, block, 0, s, reinterpret_cast<const __bf16*>(P), Whh, Y,
                              reinterpret_cast<__bf16*>(Y16), T, Bp);
    else   hipLaunchKernelGGL((lstm_rec_fwd_h128_bf16_w8_kernel<false>), grid, block, 0, s, reinterpret_cast<const __bf16*>(P), Whh, Y,
                              reinterpret_cast<__bf16*>(Y16), T, Bp);
    LOB_CHECK_LAUNCH();
    return 0;
}
