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
#include <stdlib.h>
#include "lob_common.h"

// H = 128 on 16-row sub-tiles / v_mfma_f32_16x16x4_f32 (lstm_rec_f32_s16.hip)
int lob_rec_fwd_s16(float* P, const float* Whh, float* Y, float* Csave, int T, int Bp, int D, int save, hipStream_t s);
int lob_rec_fwd_split(float* P, const float* Whh, float* Y, float* Csave, int T, int Bp, int D, int save, const float* range,
                      float* Yd, float drop_p, uint64_t seed, hipStream_t s);
// H = 32 / 64 / 256: W_hh streamed from L2 (lstm_rec_stream.hip)
int lob_stream_supports(int H);
int lob_stream_fwd(float* P, const float* Whh, float* Y, float* Csave, int T, int Bp, int H, int D, int save, hipStream_t s);
int lob_stream_bwd(const float* G, const float* Cs, const float* Whh, const float* dY, void* dP, int bf, float* dbias,
                   int T, int Bp, int H, int D, hipStream_t s);

namespace {

#ifndef LOB_SAVE_WLDS
#define LOB_SAVE_WLDS 0
#endif
constexpr bool SAVE_WLDS = LOB_SAVE_WLDS;
constexpr int HS_LD = 132;   // LDS row stride of the h tile (floats): 33 x 16 B -> conflict-free b128

template <bool SAVE, bool WLDS>
__global__ __launch_bounds__(256, 1) void lstm_rec_fwd_h128_kernel(
    float* __restrict__ P, const float* __restrict__ Whh, float* __restrict__ Y,
    float* __restrict__ Csave, int T, int Bp) {
    constexpr int H = 128;
    __shared__ __attribute__((aligned(16))) float hs[2 * 32 * HS_LD];
    // k-block 3 of the W_hh slice lives in LDS ([wave][gate][16 steps][64 lanes], lane-contiguous =
    // conflict-free ds_read_b32): 192 + 64 = the 256 values per lane, 64 VGPRs freed for the
    // P prefetch and the cell update (no scratch spills).
    __shared__ float wls[WLDS ? 4 * 4 * 16 * 64 : 64];
    constexpr int KBR = WLDS ? 3 : 4;          // k-blocks of W_hh held in registers

    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);          // wave index, provably uniform
    const int bt = blockIdx.x, d = blockIdx.y, D = gridDim.y, NBT = gridDim.x;
    const int l31 = lane & 31, hi = lane >> 5;

    // ---- W_hh slice -> registers.  B operand of step (kb, q, e): W[n][k = 32kb + 16hi + 4q + e]
    f32x4 wr[4][KBR][4];
    {
        const float* wbase = Whh + (size_t)d * 4 * H * H;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float* row = wbase + (size_t)(g * H + 32 * w + l31) * H + 16 * hi;
#pragma unroll
            for (int kb = 0; kb < KBR; ++kb)
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    wr[g][kb][q] = *reinterpret_cast<const f32x4*>(row + 32 * kb + 4 * q);
            if (WLDS) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 v = *reinterpret_cast<const f32x4*>(row + 96 + 4 * q);
#pragma unroll
                    for (int e = 0; e < 4; ++e) wls[((w * 4 + g) * 16 + 4 * q + e) * 64 + lane] = v[e];
                }
            }
        }
    }
    for (int i = tid; i < 2 * 32 * HS_LD; i += 256) hs[i] = 0.f;

    float c[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) c[r] = 0.f;

    // per-(d,t,bt) block of P: [w][g][q][lane][4]
    const size_t pstep = (size_t)NBT * 4 * 4 * 1024;               // floats per t
    // (wave-uniform pointer) + (one 32-bit lane offset) addressing: see the backward kernel
    float* pblk = P + ((size_t)d * T * NBT + bt) * 4 * 4 * 1024 + (size_t)w * 4 * 1024;
    const size_t cstep = (size_t)NBT * 4 * 1024;
    float* cblk = SAVE ? Csave + ((size_t)d * T * NBT + bt) * 4 * 1024 + (size_t)w * 1024 : nullptr;
    const unsigned frag_off = lane * 4;
    const unsigned y_off = (unsigned)(4 * hi * (D * H) + l31);

    const int t_first = d ? T - 1 : 0, dt = d ? -1 : 1;
    f32x16 pn[4];
    {
        const float* p = pblk + (size_t)t_first * pstep;
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                f32x4 v = *reinterpret_cast<const f32x4*>((p + g * 1024 + q * 256) + frag_off);
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
                    f32x4 v = *reinterpret_cast<const f32x4*>((p + g * 1024 + q * 256) + frag_off);
                    pn[g][4 * q + 0] = v[0]; pn[g][4 * q + 1] = v[1]; pn[g][4 * q + 2] = v[2]; pn[g][4 * q + 3] = v[3];
                }
        }
        // ---- z += h_{t-1} * W_hh^T
        const float* hrow = hs + cur * 32 * HS_LD + l31 * HS_LD + 16 * hi;
#pragma unroll
        for (int kb = 0; kb < KBR; ++kb) {
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
        if (WLDS) {
            f32x4 a[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) a[q] = *reinterpret_cast<const f32x4*>(hrow + 96 + 4 * q);
            const float* wl = wls + (size_t)w * 4 * 16 * 64 + lane;
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int g = 0; g < 4; ++g)
                        acc[g] = mfma32(a[q][e], wl[(g * 16 + 4 * q + e) * 64], acc[g]);
        }
        // ---- cell update (wave-local), h_t -> LDS (other buffer) and HBM
        float* hnext = hs + (cur ^ 1) * 32 * HS_LD + 32 * w + l31 + 4 * hi * HS_LD;
        float* yrow = Y + ((size_t)t * Bp + bt * 32) * (D * H) + d * H + 32 * w;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float ig = fast_sigmoid(acc[0][r]);
            const float fg = fast_sigmoid(acc[1][r]);
            const float gg = fast_tanh(acc[2][r]);
            const float og = fast_sigmoid(acc[3][r]);
            c[r] = __builtin_fmaf(fg, c[r], ig * gg);
            const float h = og * fast_tanh(c[r]);
            const int row = (r & 3) + 8 * (r >> 2);          // + 4*hi folded into the lane offsets
            hnext[row * HS_LD] = h;
            (yrow + (size_t)row * (D * H))[y_off] = h;
            if (SAVE) { acc[0][r] = ig; acc[1][r] = fg; acc[2][r] = gg; acc[3][r] = og; }
        }
        if (SAVE) {
            float* p = pblk + (size_t)t * pstep;
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    f32x4 v = {acc[g][4 * q + 0], acc[g][4 * q + 1], acc[g][4 * q + 2], acc[g][4 * q + 3]};
                    *reinterpret_cast<f32x4*>((p + g * 1024 + q * 256) + frag_off) = v;
                }
            float* cp = cblk + (size_t)t * cstep;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                f32x4 v = {c[4 * q + 0], c[4 * q + 1], c[4 * q + 2], c[4 * q + 3]};
                *reinterpret_cast<f32x4*>((cp + q * 256) + frag_off) = v;
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
            const float cn = __builtin_fmaf(fg, cst[idx], ig * gg);
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


// ------------------------------------------------------------------------------------
// BPTT, H = 128 fast path.  Mirror of the forward kernel: wave w owns hidden units
// [32w, 32w+32); W_hh^T fragments (contraction over the 512 gate rows) live in 256 VGPRs.
// Per step: cell backward (wave-local, VALU) -> dgates tile in LDS (32 x 512) -> barrier ->
//   dh_{t-1} = dgates * W_hh  (256 MFMAs per wave)  +  coalesced copy of the tile to dP (HBM)
// -> barrier.  Saved activations (G, Csave) are read-only: backward is re-entrant.
// ------------------------------------------------------------------------------------
typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
constexpr int DG_LD = 516;   // 512 + 4: row stride = 129 x 16 B, odd -> conflict-free ds_read_b128

template <bool DP_BF16>
__global__ __launch_bounds__(256, 1) void lstm_rec_bwd_h128_kernel(
    const float* __restrict__ G, const float* __restrict__ Csave, const float* __restrict__ Whh,
    const float* __restrict__ dY, void* __restrict__ dPv, float* __restrict__ dbias, int T, int Bp) {
    constexpr int H = 128;
    __shared__ __attribute__((aligned(16))) float dgs[32 * DG_LD];
    __shared__ float wls[4 * 4 * 16 * 64];      // gate-row blocks 12..15 of the W_hh^T slice (see forward)
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);          // wave index, provably uniform
    const int bt = blockIdx.x, d = blockIdx.y, D = gridDim.y, NBT = gridDim.x;
    const int l31 = lane & 31, hi = lane >> 5;

    float wt[12][16];
    {
        const float* wb = Whh + (size_t)d * 4 * H * H + 32 * w + l31;
#pragma unroll
        for (int nb = 0; nb < 12; ++nb)
#pragma unroll
            for (int s = 0; s < 16; ++s) wt[nb][s] = wb[(size_t)(32 * nb + 16 * hi + s) * H];
#pragma unroll
        for (int nb = 12; nb < 16; ++nb)
#pragma unroll
            for (int s = 0; s < 16; ++s)
                wls[((w * 4 + nb - 12) * 16 + s) * 64 + lane] = wb[(size_t)(32 * nb + 16 * hi + s) * H];
    }
    // Addressing: every access is (wave-uniform pointer, SGPRs) + (ONE 32-bit per-lane offset), so
    // the 20 fragment loads, 16 dY loads and 16 dP stores of a step share three offset VGPRs.
    const size_t gstep = (size_t)NBT * 16 * 1024, cstep = (size_t)NBT * 4096;
    const float* gwave = G + ((size_t)d * T * NBT + bt) * 16 * 1024 + (size_t)w * 4096;
    const float* cwave = Csave + ((size_t)d * T * NBT + bt) * 4096 + (size_t)w * 1024;
    const unsigned frag_off = lane * 4;
    const int DH = D * H, D4H = D * 4 * H;
    const float* dywave = dY + (size_t)(bt * 32) * DH + d * H + 32 * w;
    const unsigned dy_off = (unsigned)(4 * hi * DH + l31);
    const unsigned dp_off = (unsigned)((tid >> 7) * D4H + (tid & 127) * 4);

    const int t_first = d ? 0 : T - 1, dt = d ? 1 : -1;   // backward walk; c_{prev} lives at t + dt
    f32x16 gt[4], ct, cp, dhrec;
    float dy[16], dcarry[16];
    float dbsum[4] = {0.f, 0.f, 0.f, 0.f};      // bias gradient of this lane's unit, summed over rows and time
#pragma unroll
    for (int r = 0; r < 16; ++r) { dcarry[r] = 0.f; dhrec[r] = 0.f; }

    auto load_step = [&](int t) {
        const float* gp = gwave + (size_t)t * gstep;
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 v = *reinterpret_cast<const f32x4*>((gp + g * 1024 + q * 256) + frag_off);
                gt[g][4 * q] = v[0]; gt[g][4 * q + 1] = v[1]; gt[g][4 * q + 2] = v[2]; gt[g][4 * q + 3] = v[3];
            }
        const int tp = t + dt;
        if (tp >= 0 && tp < T) {
            const float* cq = cwave + (size_t)tp * cstep;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 v = *reinterpret_cast<const f32x4*>((cq + q * 256) + frag_off);
                cp[4 * q] = v[0]; cp[4 * q + 1] = v[1]; cp[4 * q + 2] = v[2]; cp[4 * q + 3] = v[3];
            }
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) cp[r] = 0.f;
        }
        const float* dp = dywave + (size_t)t * Bp * DH;
#pragma unroll
        for (int r = 0; r < 16; ++r) dy[r] = (dp + ((r & 3) + 8 * (r >> 2)) * DH)[dy_off];
    };
    {
        const float* cq = cwave + (size_t)t_first * cstep;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 v = *reinterpret_cast<const f32x4*>((cq + q * 256) + frag_off);
            ct[4 * q] = v[0]; ct[4 * q + 1] = v[1]; ct[4 * q + 2] = v[2]; ct[4 * q + 3] = v[3];
        }
    }
    load_step(t_first);

    for (int step = 0; step < T; ++step) {
        const int t = t_first + dt * step;
        float* dgw = dgs + 32 * w + l31 + 4 * hi * DG_LD;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float ig = gt[0][r], fg = gt[1][r], gg = gt[2][r], og = gt[3][r];
            const float dh = dy[r] + dhrec[r];
            const float tc = fast_tanh(ct[r]);
            const float dc = dcarry[r] + dh * og * (1.f - tc * tc);
            dcarry[r] = dc * fg;
            float* p = dgw + ((r & 3) + 8 * (r >> 2)) * DG_LD;
            const float v0 = dc * gg * ig * (1.f - ig), v1 = dc * cp[r] * fg * (1.f - fg);
            const float v2 = dc * ig * (1.f - gg * gg), v3 = dh * tc * og * (1.f - og);
            p[0 * H] = v0; p[1 * H] = v1; p[2 * H] = v2; p[3 * H] = v3;
            dbsum[0] += v0; dbsum[1] += v1; dbsum[2] += v2; dbsum[3] += v3;
        }
        ct = cp;
        __syncthreads();
        if (step + 1 < T) load_step(t + dt);
        // ---- dh_{t-1} = dgates * W_hh   (contraction over 512 gate rows)
#pragma unroll
        for (int r = 0; r < 16; ++r) dhrec[r] = 0.f;
        const float* arow = dgs + l31 * DG_LD + 16 * hi;
#pragma unroll
        for (int nb = 0; nb < 12; ++nb) {
            f32x4 a[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) a[q] = *reinterpret_cast<const f32x4*>(arow + 32 * nb + 4 * q);
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int e = 0; e < 4; ++e) dhrec = mfma32(a[q][e], wt[nb][4 * q + e], dhrec);
        }
        {
            const float* wl = wls + (size_t)w * 4 * 16 * 64 + lane;
#pragma unroll
            for (int nb = 12; nb < 16; ++nb) {
                f32x4 a[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) a[q] = *reinterpret_cast<const f32x4*>(arow + 32 * nb + 4 * q);
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        dhrec = mfma32(a[q][e], wl[((nb - 12) * 16 + 4 * q + e) * 64], dhrec);
            }
        }
        // ---- dgates tile -> dP (row-major [T*Bp][D*4H]), one full row segment per 128 threads
        const float* src = dgs + (tid >> 7) * DG_LD + (tid & 127) * 4;
        if (DP_BF16) {
            __bf16* dpb = reinterpret_cast<__bf16*>(dPv) + ((size_t)t * Bp + bt * 32) * D4H + d * 4 * H;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(src + 2 * i * DG_LD);
                bf16x4_t hv = {(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
                *reinterpret_cast<bf16x4_t*>((dpb + (size_t)(2 * i) * D4H) + dp_off) = hv;
            }
        } else {
            float* dpb = reinterpret_cast<float*>(dPv) + ((size_t)t * Bp + bt * 32) * D4H + d * 4 * H;
#pragma unroll
            for (int i = 0; i < 16; ++i)
                *reinterpret_cast<f32x4*>((dpb + (size_t)(2 * i) * D4H) + dp_off) =
                    *reinterpret_cast<const f32x4*>(src + 2 * i * DG_LD);
        }
        __syncthreads();
    }
    if (dbias) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float v = dbsum[g] + __shfl_xor(dbsum[g], 32, 64);
            if (hi == 0) atomicAdd(dbias + (size_t)d * 4 * H + g * H + 32 * w + l31, v);
        }
    }
}

// Generic BPTT (any H): G row-major activated gates, Csave [D][T][Bp][H].
template <bool DP_BF16>
__global__ __launch_bounds__(256) void lstm_rec_bwd_generic_kernel(
    const float* __restrict__ G, const float* __restrict__ Csave, const float* __restrict__ Whh,
    const float* __restrict__ dY, void* __restrict__ dPv, int T, int Bp, int H) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* dg = sm;                    // [RB][4H]
    float* dhrec = sm + RB * 4 * H;    // [RB][H]
    float* dcar = dhrec + RB * H;      // [RB][H]
    const int tid = threadIdx.x, d = blockIdx.y, D = gridDim.y, b0 = blockIdx.x * RB;
    const float* W = Whh + (size_t)d * 4 * H * H;
    for (int i = tid; i < RB * H; i += blockDim.x) { dhrec[i] = 0.f; dcar[i] = 0.f; }
    __syncthreads();
    const int t_first = d ? 0 : T - 1, dt = d ? 1 : -1;
    for (int step = 0; step < T; ++step) {
        const int t = t_first + dt * step, tp = t + dt;
        for (int idx = tid; idx < RB * H; idx += blockDim.x) {
            const int r = idx / H, u = idx % H, b = b0 + r;
            if (b >= Bp) { for (int g = 0; g < 4; ++g) dg[r * 4 * H + g * H + u] = 0.f; continue; }
            const float* grow = G + ((size_t)t * Bp + b) * (D * 4 * H) + (size_t)d * 4 * H;
            const float ig = grow[u], fg = grow[H + u], gg = grow[2 * H + u], og = grow[3 * H + u];
            const float ctv = Csave[(((size_t)d * T + t) * Bp + b) * H + u];
            const float cpv = (tp >= 0 && tp < T) ? Csave[(((size_t)d * T + tp) * Bp + b) * H + u] : 0.f;
            const float dh = dY[((size_t)t * Bp + b) * (D * H) + d * H + u] + dhrec[idx];
            const float tc = fast_tanh(ctv);
            const float dc = dcar[idx] + dh * og * (1.f - tc * tc);
            dcar[idx] = dc * fg;
            const float v0 = dc * gg * ig * (1.f - ig), v1 = dc * cpv * fg * (1.f - fg);
            const float v2 = dc * ig * (1.f - gg * gg), v3 = dh * tc * og * (1.f - og);
            float* o = dg + r * 4 * H;
            o[u] = v0; o[H + u] = v1; o[2 * H + u] = v2; o[3 * H + u] = v3;
            const size_t po = ((size_t)t * Bp + b) * (D * 4 * H) + (size_t)d * 4 * H;
            if (DP_BF16) {
                __bf16* prow = reinterpret_cast<__bf16*>(dPv) + po;
                prow[u] = (__bf16)v0; prow[H + u] = (__bf16)v1; prow[2 * H + u] = (__bf16)v2; prow[3 * H + u] = (__bf16)v3;
            } else {
                float* prow = reinterpret_cast<float*>(dPv) + po;
                prow[u] = v0; prow[H + u] = v1; prow[2 * H + u] = v2; prow[3 * H + u] = v3;
            }
        }
        __syncthreads();
        for (int idx = tid; idx < RB * H; idx += blockDim.x) {
            const int r = idx / H, k = idx % H;
            float s = 0.f;
            for (int n = 0; n < 4 * H; ++n) s = fmaf(dg[r * 4 * H + n], W[(size_t)n * H + k], s);
            dhrec[idx] = s;
        }
        __syncthreads();
    }
}

}  // namespace

// The saving forward of the fp16-split kernel (H = 128, LOB_VAR_F32_SPLIT != 0) with nn.LSTM's inter-layer dropout fused into
// the producer: Yd = dropout(Y), the stand-alone kernel's mask (lob_dropout_f32 with the same p / seed on Y).
extern "C" int lob_lstm_rec_fwd_f32_drop(float* P, const float* Whh, float* Y, float* Yd, float drop_p, uint64_t seed,
                                         float* Csave, int T, int Bp, int H, int D, const float* range, void* stream) {
    if (!P || !Whh || !Y || !Yd || !Csave || T <= 0 || Bp <= 0 || (D != 1 && D != 2) || drop_p <= 0.f || drop_p >= 1.f)
        return LOB_E_ARG;
    if (H != 128 || (Bp % 32) || lob_variant(LOB_VAR_F32_SPLIT) == 0) return LOB_E_SHAPE;
    if ((reinterpret_cast<uintptr_t>(P) | reinterpret_cast<uintptr_t>(Whh) | reinterpret_cast<uintptr_t>(Csave) |
         reinterpret_cast<uintptr_t>(Y) | reinterpret_cast<uintptr_t>(Yd)) & 15) return LOB_E_ALIGN;
    return lob_rec_fwd_split(P, Whh, Y, Csave, T, Bp, D, 1, range, Yd, drop_p, seed, (hipStream_t)stream);
}

extern "C" int lob_lstm_rec_fwd_f32(float* P, const float* Whh, float* Y, float* Csave,
                                    int T, int Bp, int H, int D, int save, const float* range, void* stream) {
    if (!P || !Whh || !Y || T <= 0 || Bp <= 0 || H <= 0 || (D != 1 && D != 2)) return LOB_E_ARG;
    if (save && !Csave) return LOB_E_ARG;
    hipStream_t s = (hipStream_t)stream;
    if (H == 128) {
        if (Bp % 32) return LOB_E_SHAPE;
        if ((reinterpret_cast<uintptr_t>(P) | reinterpret_cast<uintptr_t>(Whh) |
             reinterpret_cast<uintptr_t>(Csave)) & 15) return LOB_E_ALIGN;
        // 16-row tiles (v_mfma_f32_16x16x4_f32) by default: twice the workgroups for small batches and 2-8 % faster at
        // B = 4096; LOB_REC_FWD=32 selects the 32-row kernel below (kept for A/B measurements)
        // default: the fp32-accurate split kernel on the 16-bit matrix pipe (lstm_rec_f32_split.hip);
        // LOB_VAR_F32_SPLIT = 0 selects the exact-fp32 MFMA kernels (16-row, or 32-row with LOB_VAR_REC_FWD_ROWS = 32)
        if (lob_variant(LOB_VAR_F32_SPLIT) != 0) return lob_rec_fwd_split(P, Whh, Y, Csave, T, Bp, D, save, range, nullptr, 0.f, 0, s);
        const bool rows32 = lob_variant(LOB_VAR_REC_FWD_ROWS) == 32;
        if (!rows32) return lob_rec_fwd_s16(P, Whh, Y, Csave, T, Bp, D, save, s);
        const dim3 grid(Bp / 32, D), block(256);
        if (save) hipLaunchKernelGGL((lstm_rec_fwd_h128_kernel<true, SAVE_WLDS>), grid, block, 0, s, P, Whh, Y, Csave, T, Bp);
        else      hipLaunchKernelGGL((lstm_rec_fwd_h128_kernel<false, false>), grid, block, 0, s, P, Whh, Y, Csave, T, Bp);
    } else if (lob_stream_supports(H) && Bp % 32 == 0) {
        if ((reinterpret_cast<uintptr_t>(P) | reinterpret_cast<uintptr_t>(Whh) |
             reinterpret_cast<uintptr_t>(Csave)) & 15) return LOB_E_ALIGN;
        return lob_stream_fwd(P, Whh, Y, Csave, T, Bp, H, D, save, s);
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

extern "C" int lob_lstm_rec_bwd_f32(const float* G, const float* Csave, const float* Whh,
                                    const float* dY, void* dP, int dp_bf16, float* dbias,
                                    int T, int Bp, int H, int D, void* stream) {
    if (!G || !Csave || !Whh || !dY || !dP || T <= 0 || Bp <= 0 || H <= 0 || (D != 1 && D != 2)) return LOB_E_ARG;
    hipStream_t s = (hipStream_t)stream;
    if (H == 128) {
        if (Bp % 32) return LOB_E_SHAPE;
        if ((reinterpret_cast<uintptr_t>(G) | reinterpret_cast<uintptr_t>(Csave) |
             reinterpret_cast<uintptr_t>(dP)) & 15) return LOB_E_ALIGN;
        if (dp_bf16) hipLaunchKernelGGL((lstm_rec_bwd_h128_kernel<true>), dim3(Bp / 32, D), dim3(256), 0, s, G, Csave, Whh, dY, dP, dbias, T, Bp);
        else         hipLaunchKernelGGL((lstm_rec_bwd_h128_kernel<false>), dim3(Bp / 32, D), dim3(256), 0, s, G, Csave, Whh, dY, dP, dbias, T, Bp);
    } else if (lob_stream_supports(H) && Bp % 32 == 0) {
        if ((reinterpret_cast<uintptr_t>(G) | reinterpret_cast<uintptr_t>(Csave) |
             reinterpret_cast<uintptr_t>(dP)) & 15) return LOB_E_ALIGN;
        return lob_stream_bwd(G, Csave, Whh, dY, dP, dp_bf16, dbias, T, Bp, H, D, s);
    } else {
        if (dbias) return LOB_E_SHAPE;      // the generic path leaves the bias gradient to lob_colsum_f32
        const size_t smem = (size_t)6 * RB * H * sizeof(float);
        if (smem > 64 * 1024) return LOB_E_SHAPE;
        const dim3 grid((Bp + RB - 1) / RB, D);
        if (dp_bf16) hipLaunchKernelGGL((lstm_rec_bwd_generic_kernel<true>), grid, dim3(256), smem, s, G, Csave, Whh, dY, dP, T, Bp, H);
        else         hipLaunchKernelGGL((lstm_rec_bwd_generic_kernel<false>), grid, dim3(256), smem, s, G, Csave, Whh, dY, dP, T, Bp, H);
    }
    LOB_CHECK_LAUNCH();
    return 0;
}

int lob_rec_bwd_split(const float* G, const float* Csave, const float* Whh, const float* dY, void* dP, int dp_bf16, float* dbias,
                      float* amax_out, int T, int Bp, int D, const float* range, hipStream_t s);      // lstm_rec_f32_split.hip

// lob_lstm_rec_bwd_f32 on the fp16-split arithmetic of the fp32 path (H == 128, Bp % 32 == 0, LOB_VAR_F32_SPLIT != 0;
// anything else: LOB_E_SHAPE -- the caller keeps lob_lstm_rec_bwd_f32).  amax_out (may be NULL): a zeroed device float that
// receives max|dP| of the launch (atomic max); range (may be NULL): D device floats, max|W_hh| per direction.
extern "C" int lob_lstm_rec_bwd_f32_x(const float* G, const float* Csave, const float* Whh, const float* dY, void* dP,
                                      int dp_bf16, float* dbias, float* amax_out, const float* range,
                                      int T, int Bp, int H, int D, void* stream) {
    if (!G || !Csave || !Whh || !dY || !dP || T <= 0 || Bp <= 0 || H <= 0 || (D != 1 && D != 2)) return LOB_E_ARG;
    if (H != 128 || (Bp % 32) || lob_variant(LOB_VAR_F32_SPLIT) == 0) return LOB_E_SHAPE;
    if ((reinterpret_cast<uintptr_t>(G) | reinterpret_cast<uintptr_t>(Csave) | reinterpret_cast<uintptr_t>(dP)) & 15) return LOB_E_ALIGN;
    return lob_rec_bwd_split(G, Csave, Whh, dY, dP, dp_bf16, dbias, amax_out, T, Bp, D, range, (hipStream_t)stream);
}

// 1 when the recurrent kernels for this H consume / produce the accumulator-fragment layout
// (P from lob_gate_gemm_x_* with frag = 1, fused bias gradient), 0 for the row-major generic path.
extern "C" int lob_lstm_uses_fragment_layout(int H) { return (H == 128 || lob_stream_supports(H)) ? 1 : 0; }
