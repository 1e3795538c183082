// Mixed-precision recurrent kernels (H = 128): the hidden-state gate GEMM h_{t-1} W_hh^T (and its
// transpose in BPTT) runs on bf16 MFMA (v_mfma_f32_32x32x16_bf16) with fp32 accumulation ON TOP of
// the fp32 pre-activations; cell state, activations, gradients-through-time carry and everything
// stored for the backward pass stay fp32 ("bf16 gate-GEMMs + fp32 recurrence", BASELINE.json
// configs[2]).  Same persistent structure as lstm_rec_f32.hip (grid = batch tiles x directions,
// wave w owns hidden units [32w, 32w+32), one barrier per step forward / two backward), but the
// per-step matrix work drops from 16,384 to 1,024 MFMA cycles, which leaves the kernels bound by
// the HBM streams (P in, gates/c/Y out; gates/c/dY in, dP out): P is prefetched TWO steps ahead.
#include "lob_common.h"
#include <stdlib.h>

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

constexpr int H = 128;
constexpr int HB_LD = 136;     // h tile row stride in bf16 (272 B = 17 x 16 B, odd -> conflict-free b128)
constexpr int DGB_LD = 520;    // dgates tile row stride in bf16 (1040 B = 65 x 16 B)

__device__ __forceinline__ f32x16 mfma_bf16(bf16x8 a, bf16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ bf16x8 cvt8(const float* p) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(p), b = *reinterpret_cast<const f32x4*>(p + 4);
    bf16x8 r = {(__bf16)a[0], (__bf16)a[1], (__bf16)a[2], (__bf16)a[3],
                (__bf16)b[0], (__bf16)b[1], (__bf16)b[2], (__bf16)b[3]};
    return r;
}

// Fragment blocks: fp32 as [gate][q][lane][4]; bf16 as [gate][q pair][lane][8] (the same elements, the two q of a pair
// side by side), so that a lane moves 16 B per access either way (the gate GEMM's epilogue is store-issue bound: half
// the store instructions for the bf16 image).  `off` is lane * 4.
template <typename E>
__device__ __forceinline__ void load_frag4(const E* p, unsigned off, f32x16 (&dst)[4]) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        if constexpr (sizeof(E) == 4) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 v = *reinterpret_cast<const f32x4*>((p + g * 1024 + q * 256) + off);
                dst[g][4 * q] = v[0]; dst[g][4 * q + 1] = v[1]; dst[g][4 * q + 2] = v[2]; dst[g][4 * q + 3] = v[3];
            }
        } else {
#pragma unroll
            for (int pq = 0; pq < 2; ++pq) {
                const bf16x8 v = *reinterpret_cast<const bf16x8*>((p + g * 1024 + pq * 512) + 2 * off);
#pragma unroll
                for (int e = 0; e < 8; ++e) dst[g][8 * pq + e] = (float)v[e];
            }
        }
    }
}
// Raw (unconverted) fragment block: prefetched two steps ahead and only converted to fp32 when it is
// consumed -- a conversion at load time would force the wait for the data right at the prefetch.
template <typename E> struct RawFrag;
template <> struct RawFrag<float> { f32x4 v[16]; };
template <> struct RawFrag<__bf16> { bf16x8 v[8]; };

template <typename E>
__device__ __forceinline__ void load_raw(const E* p, unsigned off, RawFrag<E>& r) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        if constexpr (sizeof(E) == 4) {
#pragma unroll
            for (int q = 0; q < 4; ++q) r.v[4 * g + q] = *reinterpret_cast<const f32x4*>((p + g * 1024 + q * 256) + off);
        } else {
#pragma unroll
            for (int pq = 0; pq < 2; ++pq)
                r.v[2 * g + pq] = *reinterpret_cast<const bf16x8*>((p + g * 1024 + pq * 512) + 2 * off);
        }
    }
}
template <typename E>
__device__ __forceinline__ void raw_to_acc(const RawFrag<E>& r, f32x16 (&dst)[4]) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        if constexpr (sizeof(E) == 4) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int e = 0; e < 4; ++e) dst[g][4 * q + e] = r.v[4 * g + q][e];
        } else {
#pragma unroll
            for (int pq = 0; pq < 2; ++pq)
#pragma unroll
                for (int e = 0; e < 8; ++e) dst[g][8 * pq + e] = (float)r.v[2 * g + pq][e];
        }
    }
}

template <typename E>
__device__ __forceinline__ void store_frag4(E* p, unsigned off, const f32x16 (&src)[4]) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        if constexpr (sizeof(E) == 4) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                f32x4 v = {src[g][4 * q + 0], src[g][4 * q + 1], src[g][4 * q + 2], src[g][4 * q + 3]};
                *reinterpret_cast<f32x4*>((p + g * 1024 + q * 256) + off) = v;
            }
        } else {
#pragma unroll
            for (int pq = 0; pq < 2; ++pq) {
                bf16x8 v;
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = (__bf16)src[g][8 * pq + e];
                *reinterpret_cast<bf16x8*>((p + g * 1024 + pq * 512) + 2 * off) = v;
            }
        }
    }
}

// Outputs: Y (fp32, scattered 128-B segments straight from the accumulator layout) when YF32; the bf16
// copies -- Y16 (h_t as the next GEMMs' bf16 operand) and Yd (the same with nn.LSTM's inter-layer dropout
// applied) -- are produced from the bf16 h tile in LDS one barrier later, as whole 256-B row segments
// (16 B per lane), instead of 2-byte stores from the accumulator layout.
template <bool SAVE, bool YF32, bool Y16, bool DROP, typename PE>
__global__ __launch_bounds__(256, 1) void lstm_rec_fwd_h128_bf16_kernel(
    PE* __restrict__ P, const float* __restrict__ Whh, float* __restrict__ Y,
    float* __restrict__ Csave, __bf16* __restrict__ Y16p, __bf16* __restrict__ Yd, float drop_p, uint64_t seed,
    int T, int Bp) {
    __shared__ __attribute__((aligned(16))) __bf16 hs[2 * 32 * HB_LD];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int bt = blockIdx.x, d = blockIdx.y, D = gridDim.y, NBT = gridDim.x;
    const int l31 = lane & 31, hi = lane >> 5;

    // B fragments: W_hh[n = g*128 + 32w + l31][k = 16 ks + 8 hi + j]
    bf16x8 wr[4][8];
    {
        const float* wbase = Whh + (size_t)d * 4 * H * H;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float* row = wbase + (size_t)(g * H + 32 * w + l31) * H + 8 * hi;
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) wr[g][ks] = cvt8(row + 16 * ks);
        }
    }
    for (int i = tid; i < 2 * 32 * HB_LD; i += 256) hs[i] = (__bf16)0.f;
    float c[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) c[r] = 0.f;

    const size_t pstep = (size_t)NBT * 16 * 1024, cstep = (size_t)NBT * 4096;
    PE* pblk = P + ((size_t)d * T * NBT + bt) * 16 * 1024 + (size_t)w * 4096;
    float* cblk = SAVE ? Csave + ((size_t)d * T * NBT + bt) * 4096 + (size_t)w * 1024 : nullptr;
    const unsigned frag_off = lane * 4;
    const unsigned y_off = (unsigned)(4 * hi * (D * H) + l31);
    const int t_first = d ? T - 1 : 0, dt = d ? -1 : 1;

    RawFrag<PE> pa, pb;           // P two steps ahead (raw): pa = step s, pb = step s+1
    load_raw(pblk + (size_t)t_first * pstep, frag_off, pa);
    if (T > 1) load_raw(pblk + (size_t)(t_first + dt) * pstep, frag_off, pb);
    __syncthreads();

    auto one_step = [&](int step, RawFrag<PE>& praw, int cur) {
        const int t = t_first + dt * step;
        f32x16 acc[4];
        raw_to_acc(praw, acc);
        // refill this raw set with P of step + 2 (it is consumed two steps from now)
        if (step + 2 < T) load_raw(pblk + (size_t)(t + 2 * dt) * pstep, frag_off, praw);
        // ---- z = P_t + h_{t-1} W_hh^T
        const __bf16* hrow = hs + cur * 32 * HB_LD + l31 * HB_LD + 8 * hi;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            const bf16x8 a = *reinterpret_cast<const bf16x8*>(hrow + 16 * ks);
#pragma unroll
            for (int g = 0; g < 4; ++g) acc[g] = mfma_bf16(a, wr[g][ks], acc[g]);
        }
        __bf16* hnext = hs + (cur ^ 1) * 32 * HB_LD + 32 * w + l31 + 4 * hi * HB_LD;
        const size_t ybase = ((size_t)t * Bp + bt * 32) * (D * H) + d * H + 32 * w;
        float* yrow = Y + ybase;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float ig = fast_sigmoid(acc[0][r]);
            const float fg = fast_sigmoid(acc[1][r]);
            const float gg = fast_tanh(acc[2][r]);
            const float og = fast_sigmoid(acc[3][r]);
            c[r] = __builtin_fmaf(fg, c[r], ig * gg);
            const float h = og * fast_tanh(c[r]);
            const int row = (r & 3) + 8 * (r >> 2);
            hnext[row * HB_LD] = (__bf16)h;
            if (YF32) (yrow + (size_t)row * (D * H))[y_off] = h;
            if (SAVE) { acc[0][r] = ig; acc[1][r] = fg; acc[2][r] = gg; acc[3][r] = og; }
        }
        if (SAVE) {
            store_frag4(pblk + (size_t)t * pstep, frag_off, acc);
            float* cp = cblk + (size_t)t * cstep;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                f32x4 v = {c[4 * q + 0], c[4 * q + 1], c[4 * q + 2], c[4 * q + 3]};
                *reinterpret_cast<f32x4*>((cp + q * 256) + frag_off) = v;
            }
        }
        __syncthreads();
        if (Y16 || DROP) {       // h_t is now complete in hs[cur ^ 1]: emit the bf16 row segments
            const __bf16* hsrc = hs + (cur ^ 1) * 32 * HB_LD;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int idx = tid + 256 * i, row = idx >> 4, c8 = (idx & 15) * 8;
                const bf16x8 hv = *reinterpret_cast<const bf16x8*>(hsrc + row * HB_LD + c8);
                const size_t o = ((size_t)t * Bp + bt * 32 + row) * (D * H) + d * H + c8;
                if (Y16) *reinterpret_cast<bf16x8*>(Y16p + o) = hv;
                if (DROP) {
                    bf16x8 dv;
#pragma unroll
                    for (int j = 0; j < 8; j += 2) {          // o is a multiple of 8: (o+j, o+j+1) share one hash
                        float s0, s1;
                        lob_dropout_scale2(seed, (uint64_t)o + j, drop_p, s0, s1);
                        dv[j] = (__bf16)((float)hv[j] * s0);
                        dv[j + 1] = (__bf16)((float)hv[j + 1] * s1);
                    }
                    *reinterpret_cast<bf16x8*>(Yd + o) = dv;
                }
            }
        }
    };

    for (int step = 0; step < T; step += 2) {
        one_step(step, pa, 0);
        if (step + 1 < T) one_step(step + 1, pb, 1);
    }
}

// ------------------------------------------------------------------------------------------
// BPTT.  dgates are rounded to bf16 once: the LDS tile feeds the MFMA A operand AND is the dP
// image copied to HBM (dP is bf16 in mixed mode).
// ------------------------------------------------------------------------------------------
template <typename PE>
__global__ __launch_bounds__(256, 1) void lstm_rec_bwd_h128_bf16_kernel(
    const PE* __restrict__ G, const float* __restrict__ Csave, const float* __restrict__ Whh,
    const float* __restrict__ dY, __bf16* __restrict__ dP, float* __restrict__ dbias, float* __restrict__ dbias2, int T, int Bp) {
    __shared__ __attribute__((aligned(16))) __bf16 dgs[32 * DGB_LD];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int bt = blockIdx.x, d = blockIdx.y, D = gridDim.y, NBT = gridDim.x;
    const int l31 = lane & 31, hi = lane >> 5;

    // B fragments of dh = dgates * W_hh: B[kc = n][col = 32w + l31], n = 16 ks + 8 hi + j, ks < 32
    bf16x8 wt[32];
    {
        const float* wb = Whh + (size_t)d * 4 * H * H + 32 * w + l31;
#pragma unroll
        for (int ks = 0; ks < 32; ++ks) {
            bf16x8 f;
#pragma unroll
            for (int j = 0; j < 8; ++j) f[j] = (__bf16)wb[(size_t)(16 * ks + 8 * hi + j) * H];
            wt[ks] = f;
        }
    }
    const size_t gstep = (size_t)NBT * 16 * 1024, cstep = (size_t)NBT * 4096;
    const PE* gwave = G + ((size_t)d * T * NBT + bt) * 16 * 1024 + (size_t)w * 4096;
    const float* cwave = Csave + ((size_t)d * T * NBT + bt) * 4096 + (size_t)w * 1024;
    const unsigned frag_off = lane * 4;
    const int DH = D * H, D4H = D * 4 * H;
    const float* dywave = dY + (size_t)(bt * 32) * DH + d * H + 32 * w;
    const unsigned dy_off = (unsigned)(4 * hi * DH + l31);
    const unsigned dp_off = (unsigned)((tid >> 6) * D4H + (tid & 63) * 8);

    const int t_first = d ? 0 : T - 1, dt = d ? 1 : -1;
    f32x16 gt[4], ct, cp, dhrec;
    float dy[16], dcarry[16];
    float dbsum[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int r = 0; r < 16; ++r) { dcarry[r] = 0.f; dhrec[r] = 0.f; }

    auto load_c = [&](int t, f32x16& dst) {
        if (t >= 0 && t < T) {
            const float* cq = cwave + (size_t)t * cstep;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 v = *reinterpret_cast<const f32x4*>((cq + q * 256) + frag_off);
                dst[4 * q] = v[0]; dst[4 * q + 1] = v[1]; dst[4 * q + 2] = v[2]; dst[4 * q + 3] = v[3];
            }
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) dst[r] = 0.f;
        }
    };
    RawFrag<PE> graw;             // saved gates of the next step, raw until the cell backward consumes them
    auto load_step = [&](int t) {
        load_raw(gwave + (size_t)t * gstep, frag_off, graw);
        load_c(t + dt, cp);
        const float* dp = dywave + (size_t)t * Bp * DH;
#pragma unroll
        for (int r = 0; r < 16; ++r) dy[r] = (dp + ((r & 3) + 8 * (r >> 2)) * DH)[dy_off];
    };
    load_c(t_first, ct);
    load_step(t_first);

    for (int step = 0; step < T; ++step) {
        const int t = t_first + dt * step;
        raw_to_acc(graw, gt);
        __bf16* dgw = dgs + 32 * w + l31 + 4 * hi * DGB_LD;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float ig = gt[0][r], fg = gt[1][r], gg = gt[2][r], og = gt[3][r];
            const float dh = dy[r] + dhrec[r];
            const float tc = fast_tanh(ct[r]);
            const float dc = dcarry[r] + dh * og * (1.f - tc * tc);
            dcarry[r] = dc * fg;
            __bf16* p = dgw + ((r & 3) + 8 * (r >> 2)) * DGB_LD;
            const float v0 = dc * gg * ig * (1.f - ig), v1 = dc * cp[r] * fg * (1.f - fg);
            const float v2 = dc * ig * (1.f - gg * gg), v3 = dh * tc * og * (1.f - og);
            p[0 * H] = (__bf16)v0; p[1 * H] = (__bf16)v1; p[2 * H] = (__bf16)v2; p[3 * H] = (__bf16)v3;
            dbsum[0] += v0; dbsum[1] += v1; dbsum[2] += v2; dbsum[3] += v3;
        }
        ct = cp;
        __syncthreads();
        if (step + 1 < T) load_step(t + dt);
#pragma unroll
        for (int r = 0; r < 16; ++r) dhrec[r] = 0.f;
        const __bf16* arow = dgs + l31 * DGB_LD + 8 * hi;
#pragma unroll
        for (int ks = 0; ks < 32; ++ks)
            dhrec = mfma_bf16(*reinterpret_cast<const bf16x8*>(arow + 16 * ks), wt[ks], dhrec);
        // ---- the bf16 tile IS the dP image: 32 rows x 1 KB, one row per wave per pass
        __bf16* dpb = dP + ((size_t)t * Bp + bt * 32) * D4H + d * 4 * H;
        const __bf16* src = dgs + (tid >> 6) * DGB_LD + (tid & 63) * 8;
#pragma unroll
        for (int i = 0; i < 8; ++i)
            *reinterpret_cast<bf16x8*>((dpb + (size_t)(4 * i) * D4H) + dp_off) =
                *reinterpret_cast<const bf16x8*>(src + 4 * i * DGB_LD);
        __syncthreads();
    }
    if (dbias) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float v = dbsum[g] + __shfl_xor(dbsum[g], 32, 64);
            if (hi == 0) {
                const size_t bi = (size_t)d * 4 * H + g * H + 32 * w + l31;
                atomicAdd(dbias + bi, v);
                if (dbias2) atomicAdd(dbias2 + bi, v);
            }
        }
    }
}

}  // namespace

// 16-row tiles, two workgroups per CU (lstm_rec_bf16_s16.hip): the default.  LOB_REC_BF16=32 selects the 32-row
// kernels of this file.
int lob_rec_fwd_bf16_s16(void* P, int pg_bf16, const float* Whh, float* Y, void* Csave, int c_bf16, void* Y16, void* Yd,
                         float drop_p, uint64_t seed, int T, int Bp, int D, int save, int nvalid, hipStream_t s);
int lob_rec_bwd_bf16_s16(const void* G, int pg_bf16, const void* Csave, int c_bf16, const float* Whh, const void* dY,
                         int dy_bf16, void* dP, float* dbias, float* dbias2, int T, int Bp, int D, hipStream_t s);
// H = 256: W_hh streamed from L2 (lstm_rec_h256_bf16.hip); bf16 P / saved gates only
int lob_rec_fwd_h256_bf16(void* P, const void* Whh16, float* Y, void* Csave, int c_bf16, void* Y16, void* Yd, float drop_p,
                          uint64_t seed, int T, int Bp, int D, int save, hipStream_t s);
int lob_rec_bwd_h256_bf16(const void* G, const void* Csave, int c_bf16, const void* WhhT16, const void* dY, int dy_bf16,
                          void* dP, float* dbias, float* dbias2, int T, int Bp, int D, hipStream_t s);
static bool use_s16() {
    const bool v = lob_variant(LOB_VAR_REC_BF16_ROWS) != 32;
    return v;
}

extern "C" int lob_lstm_rec_fwd_bf16(void* P, int pg_bf16, const float* Whh, const void* Whh16, float* Y, void* Csavev,
                                     int c_bf16, void* Y16, void* Yd, float drop_p, uint64_t seed,
                                     int T, int Bp, int Hh, int D, int save, int nvalid, void* stream) {
    if (!P || !Whh || T <= 0 || Bp <= 0 || (D != 1 && D != 2)) return LOB_E_ARG;
    if (!Y && !Y16) return LOB_E_ARG;
    if (nvalid < 0 || nvalid > Bp) return LOB_E_ARG;
    if (save && !Csavev) return LOB_E_ARG;
    // bf16 cell-state storage: the 16-row H = 128 kernels and the H = 256 kernels
    if (c_bf16 && !((Hh == 128 && use_s16()) || Hh == 256)) return LOB_E_SHAPE;
    float* Csave = reinterpret_cast<float*>(Csavev);
    if (Yd && (drop_p <= 0.f || drop_p >= 1.f)) return LOB_E_ARG;
    if (Hh == 256) {
        if (!pg_bf16 || (Bp % 32)) return LOB_E_SHAPE;
        if (!Whh16) return LOB_E_ARG;
        if ((reinterpret_cast<uintptr_t>(P) | reinterpret_cast<uintptr_t>(Whh16) | reinterpret_cast<uintptr_t>(Csave) |
             reinterpret_cast<uintptr_t>(Y16) | reinterpret_cast<uintptr_t>(Yd)) & 15) return LOB_E_ALIGN;
        return lob_rec_fwd_h256_bf16(P, Whh16, Y, Csavev, c_bf16, Y16, Yd, drop_p, seed, T, Bp, D, save, (hipStream_t)stream);
    }
    if (Hh != 128 || (Bp % 32)) return LOB_E_SHAPE;
    if ((reinterpret_cast<uintptr_t>(P) | reinterpret_cast<uintptr_t>(Whh) | reinterpret_cast<uintptr_t>(Csave) |
         reinterpret_cast<uintptr_t>(Y16) | reinterpret_cast<uintptr_t>(Yd)) & 15) return LOB_E_ALIGN;
    hipStream_t s = (hipStream_t)stream;
    if (use_s16()) return lob_rec_fwd_bf16_s16(P, pg_bf16, Whh, Y, Csavev, c_bf16, Y16, Yd, drop_p, seed, T, Bp, D, save, nvalid, s);
    const dim3 grid(Bp / 32, D), block(256);
    __bf16* y16 = reinterpret_cast<__bf16*>(Y16);
    __bf16* yd = reinterpret_cast<__bf16*>(Yd);
#define LOB_FWD(SV, YF, Y6, DR, PE) hipLaunchKernelGGL((lstm_rec_fwd_h128_bf16_kernel<SV, YF, Y6, DR, PE>), grid, block, \
        0, s, reinterpret_cast<PE*>(P), Whh, Y, Csave, y16, yd, drop_p, seed, T, Bp)
#define LOB_FWD_OUT(SV, PE) do {                                                     \
        if (Y && !y16 && !yd) LOB_FWD(SV, true, false, false, PE);                   \
        else if (Y && y16 && !yd) LOB_FWD(SV, true, true, false, PE);                \
        else if (Y && !y16 && yd) LOB_FWD(SV, true, false, true, PE);                \
        else if (Y && y16 && yd) LOB_FWD(SV, true, true, true, PE);                  \
        else if (!Y && y16 && !yd) LOB_FWD(SV, false, true, false, PE);              \
        else LOB_FWD(SV, false, true, true, PE); } while (0)
    if (!Y && !y16) return LOB_E_ARG;
    if (pg_bf16) { if (save) LOB_FWD_OUT(true, __bf16); else LOB_FWD_OUT(false, __bf16); }
    else         { if (save) LOB_FWD_OUT(true, float); else LOB_FWD_OUT(false, float); }
#undef LOB_FWD_OUT
#undef LOB_FWD
    LOB_CHECK_LAUNCH();
    return 0;
}

extern "C" int lob_lstm_rec_bwd_bf16(const void* G, int pg_bf16, const void* Csavev, int c_bf16, const float* Whh,
                                     const void* WhhT16, const void* dYv, int dy_bf16, void* dP, float* dbias, float* dbias2, int T, int Bp,
                                     int Hh, int D, void* stream) {
    if (!G || !Csavev || !Whh || !dYv || !dP || T <= 0 || Bp <= 0 || (D != 1 && D != 2)) return LOB_E_ARG;
    // bf16 dY / bf16 cell state: the 16-row H = 128 kernels and the H = 256 kernel
    if ((dy_bf16 || c_bf16) && !((Hh == 128 && use_s16()) || Hh == 256)) return LOB_E_SHAPE;
    const float* Csave = reinterpret_cast<const float*>(Csavev);
    const float* dY = reinterpret_cast<const float*>(dYv);
    if (Hh == 256) {
        if (!pg_bf16 || (Bp % 32)) return LOB_E_SHAPE;
        if (!WhhT16) return LOB_E_ARG;
        if ((reinterpret_cast<uintptr_t>(G) | reinterpret_cast<uintptr_t>(Csave) | reinterpret_cast<uintptr_t>(WhhT16) |
             reinterpret_cast<uintptr_t>(dP)) & 15) return LOB_E_ALIGN;
        return lob_rec_bwd_h256_bf16(G, Csavev, c_bf16, WhhT16, dYv, dy_bf16, dP, dbias, dbias2, T, Bp, D, (hipStream_t)stream);
    }
    if (Hh != 128 || (Bp % 32)) return LOB_E_SHAPE;
    if ((reinterpret_cast<uintptr_t>(G) | reinterpret_cast<uintptr_t>(Csave) |
         reinterpret_cast<uintptr_t>(dP)) & 15) return LOB_E_ALIGN;
    if (use_s16()) return lob_rec_bwd_bf16_s16(G, pg_bf16, Csavev, c_bf16, Whh, dYv, dy_bf16, dP, dbias, dbias2, T, Bp, D, (hipStream_t)stream);
    if (pg_bf16)
        hipLaunchKernelGGL((lstm_rec_bwd_h128_bf16_kernel<__bf16>), dim3(Bp / 32, D), dim3(256), 0, (hipStream_t)stream,
                           reinterpret_cast<const __bf16*>(G), Csave, Whh, dY, reinterpret_cast<__bf16*>(dP), dbias, dbias2, T, Bp);
    else
        hipLaunchKernelGGL((lstm_rec_bwd_h128_bf16_kernel<float>), dim3(Bp / 32, D), dim3(256), 0, (hipStream_t)stream,
                           reinterpret_cast<const float*>(G), Csave, Whh, dY, reinterpret_cast<__bf16*>(dP), dbias, dbias2, T, Bp);
    LOB_CHECK_LAUNCH();
    return 0;
}
