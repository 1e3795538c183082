// Mixed-precision recurrent kernels (H = 128) on 16-row tiles, TWO workgroups per CU.
//
// The 32-row kernels of lstm_rec_bf16.hip run one workgroup (4 waves) per CU: inside a time step the cell
// update (VALU), the LDS hand-over and the MFMAs of a wave are serial, so nothing overlaps them and the
// HBM streams are pulled at 4.4-4.9 TB/s.  With bf16 operands the W_hh slice of a wave is 128 VGPRs (not 256
// as in fp32), so a 16-row tile fits in 256 registers and two workgroups share a CU: while one waits at its
// barrier or runs its cell update the other issues MFMAs and loads.  Same fragment-order P / saved-gates / c
// layouts as the 32-row kernels (a 16-row tile is one q-half of a 32-row fragment block), so the GEMMs on either
// side are unchanged; also twice the workgroups at small batch (B = 1024: 128 instead of 64).
//
// MFMA v_mfma_f32_16x16x32_bf16: lane l feeds A[row = l&15][k = 8*(l>>4) + j] and B[k = 8*(l>>4) + j][col = l&15]
// (j < 8); D: 4 registers, col = l&15, row = 4*(l>>4) + reg.
#include "lob_common.h"
#include <stdlib.h>

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

constexpr int H = 128;
constexpr int YF_LD = 132;     // fp32 h staging row stride in floats (528 B = 33 x 16 B)
constexpr int HB_LD = 136;     // h tile row stride in bf16 (272 B = 17 x 16 B, odd -> conflict-free b128)
constexpr int DGB_LD = 520;    // dgates tile row stride in bf16 (1040 B = 65 x 16 B)

__device__ __forceinline__ f32x4 mfma16_bf16(bf16x8 a, bf16x8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ bf16x8 cvt8(const float* p) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(p), b = *reinterpret_cast<const f32x4*>(p + 4);
    bf16x8 r = {(__bf16)a[0], (__bf16)a[1], (__bf16)a[2], (__bf16)a[3],
                (__bf16)b[0], (__bf16)b[1], (__bf16)b[2], (__bf16)b[3]};
    return r;
}

// One 16-row tile of a fragment block, per lane: [gate][cbu] groups of 4 consecutive elements (rows 4rq..4rq+3 of
// one column).  Raw storage type (fp32 or bf16), converted when consumed.
// Element offsets inside a wave's block (see lstm_rec_bf16.hip): fp32 [gate][q][lane'][4], bf16 [gate][q pair][lane'][8].
// A 16-row tile s0 is q in {2 s0, 2 s0 + 1} = q pair s0; this lane's rows are q = 2 s0 + (rq >> 1), lane' =
// (rq & 1) * 32 + 16 cbu + c16.  frag_lane<E>() = the lane's offset without the s0 / gate / cbu terms,
// FRAG_CBU<E> = the step between the two unit blocks cbu.
template <typename E> __device__ __forceinline__ unsigned frag_lane(int rq, int c16) {
    if constexpr (sizeof(E) == 4) return (unsigned)((rq >> 1) * 256 + ((rq & 1) * 32 + c16) * 4);
    else                          return (unsigned)(((rq & 1) * 32 + c16) * 8 + (rq >> 1) * 4);
}
template <typename E> constexpr int FRAG_CBU = sizeof(E) == 4 ? 64 : 128;
template <typename E> struct Raw16;
template <> struct Raw16<float> { f32x4 v[8]; };
template <> struct Raw16<__bf16> { bf16x4 v[8]; };

// Cache-policy experiment (tools/h256_ablate.sh with ABL_SRC=lstm_rec_bf16_s16 ABL_DEF=LOB_NT128): bit 0 = non-temporal P
// loads, bit 1 = non-temporal saved-gate / cell-state stores, bit 2 = non-temporal Y16 / Yd stores of the forward kernel
#ifndef LOB_NT128
#define LOB_NT128 0
#endif
template <typename E>
__device__ __forceinline__ void load_raw16(const E* p, unsigned off, Raw16<E>& r) {
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int cbu = 0; cbu < 2; ++cbu) {
            if constexpr (sizeof(E) == 4) r.v[2 * g + cbu] = *reinterpret_cast<const f32x4*>((p + g * 1024 + cbu * 64) + off);
            else if constexpr (LOB_NT128 & 1) r.v[2 * g + cbu] = __builtin_nontemporal_load(reinterpret_cast<const bf16x4*>((p + g * 1024 + cbu * 128) + off));
            else                          r.v[2 * g + cbu] = *reinterpret_cast<const bf16x4*>((p + g * 1024 + cbu * 128) + off);
        }
}
template <typename E>
__device__ __forceinline__ void store_frag16(E* p, unsigned off, const f32x4 (&src)[4][2]) {
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int cbu = 0; cbu < 2; ++cbu) {
            if constexpr (sizeof(E) == 4) {
                *reinterpret_cast<f32x4*>((p + g * 1024 + cbu * 64) + off) = src[g][cbu];
            } else {
                bf16x4 v = {(__bf16)src[g][cbu][0], (__bf16)src[g][cbu][1], (__bf16)src[g][cbu][2], (__bf16)src[g][cbu][3]};
                if constexpr (LOB_NT128 & 2) __builtin_nontemporal_store(v, reinterpret_cast<bf16x4*>((p + g * 1024 + cbu * 128) + off));
                else *reinterpret_cast<bf16x4*>((p + g * 1024 + cbu * 128) + off) = v;
            }
        }
}

// CE: storage type of the saved cell state c_t (fp32, or bf16 in the same [q pair][lane][8] order as one gate of the
// saved gates: BPTT only ever uses c_t inside tanh(c_t) and as the factor of the forget-gate gradient, next to gate
// values that are bf16 already -- half the bytes of that stream, forward and backward).  The state carried through
// time stays fp32 in registers.
//
// FEW (inference, nvalid < 4 windows in the whole call -- the single-window serving call of
// 06_lstm_ode_integration.py:340-360 with one (256, 61) window): a call that small is one 16-row tile per direction on
// one CU, 768 dependent steps long, and a step is bound by what ONE wave issues: 32 MFMAs, then the gate activations of
// its 8 elements per lane (80 transcendentals).  The MFMA's D layout puts tile row 4 rq + j in register j, so with
// nvalid rows only registers j < nvalid hold a window: the cell update of the other registers is skipped (their h
// rows stay zero in LDS, so padding rows leave as zeros), and a tile without any window only writes zeros.
//
// PARTS = 2 / 4 (round 4, inference, LOB_VAR_REC_HALF): that many workgroups share each 16-row tile; workgroup jh runs the
// cell update of the registers j in [jh * 4 / PARTS, (jh + 1) * 4 / PARTS) only (its rows of the h tile; the others stay zero in
// ITS tile, the MFMAs on them are wasted) and stores only those rows.  For batches whose full tiles would occupy a quarter
// of the CUs or less: PARTS times the workgroups, 1 / PARTS of the activations on each step's serial chain (-14 % per mixed
// forward at B <= 512 with PARTS = 4, -5 % at B = 1024 with PARTS = 2; tools/half_tile_ab.py).  The accumulator register of
// a row is picked by selects on the workgroup-uniform jh, not by a branch per register as in FEW (a first version that
// extended FEW's branches made the one-window call 16 % slower).  Bit-identical to full tiles.
template <bool SAVE, bool YF32, bool Y16, bool DROP, typename PE, typename CE, bool FEW = false, int PARTS = 1>
__global__ __launch_bounds__(256, 2) void lstm_rec_fwd_h128_bf16_s16_kernel(
    PE* __restrict__ P, const float* __restrict__ Whh, float* __restrict__ Y,
    CE* __restrict__ Csave, __bf16* __restrict__ Y16p, __bf16* __restrict__ Yd, float drop_p, uint64_t seed,
    int T, int Bp, int nvalid) {
    static_assert(!FEW || (!SAVE && !DROP), "FEW: inference only");
    static_assert(PARTS == 1 || (!SAVE && !DROP && !FEW), "PARTS: inference only, full batches");
    constexpr int RPP = 4 / PARTS;                             // registers (tile rows j) per workgroup
    __shared__ __attribute__((aligned(16))) __bf16 hs[2 * 16 * HB_LD];
    // fp32 h of the step (last layer only), staged so that it leaves as 32-B-per-lane row segments instead of eight
    // 4-byte stores per lane; double-buffered like hs (one barrier per step)
    __shared__ __attribute__((aligned(16))) float yfs[YF32 ? 2 * 16 * YF_LD : 4];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int d = blockIdx.y, D = gridDim.y, NBT = Bp >> 5;
    const int c16 = lane & 15, rq = lane >> 4;
    const int bxx = (int)blockIdx.x / PARTS, jh = (int)blockIdx.x % PARTS, jlo = RPP * jh;
    const int bt = bxx >> 1, s0 = bxx & 1;                    // 32-row fragment block, 16-row half
    const int nj = FEW ? nvalid - (bt * 32 + s0 * 16) : 4;   // FEW: registers j < nj hold windows (nvalid < 4)
    const bool mine = PARTS == 1 || ((((tid >> 4) & 3) / RPP) == jh);      // PARTS: only this workgroup's rows leave
    if (FEW && nj <= 0) {                                     // a tile of padding rows: zeros for every step
        const int row = tid >> 4, c8 = (tid & 15) * 8;
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
        for (int t = 0; t < T; ++t) {
            const size_t o = ((size_t)t * Bp + bt * 32 + s0 * 16 + row) * (D * H) + d * H + c8;
            if (YF32) { *reinterpret_cast<f32x4*>(Y + o) = z; *reinterpret_cast<f32x4*>(Y + o + 4) = z; }
            if (Y16) *reinterpret_cast<f32x4*>(Y16p + o) = z;
        }
        return;
    }

    // B fragments: wr[g][cbu][ks] = W_hh[g*128 + 32w + 16cbu + c16][32ks + 8rq .. +7]
    bf16x8 wr[4][2][4];
    {
        const float* wbase = Whh + (size_t)d * 4 * H * H;
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int cbu = 0; cbu < 2; ++cbu) {
                const float* row = wbase + (size_t)(g * H + 32 * w + 16 * cbu + c16) * H + 8 * rq;
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) { wr[g][cbu][ks] = cvt8(row + 32 * ks); __builtin_amdgcn_sched_barrier(0); }
            }
    }
    for (int i = tid; i < 2 * 16 * HB_LD; i += 256) hs[i] = (__bf16)0.f;
    if ((FEW || PARTS > 1) && YF32)
        for (int i = tid; i < 2 * 16 * YF_LD; i += 256) yfs[i] = 0.f;
    float c[2][4];
#pragma unroll
    for (int cbu = 0; cbu < 2; ++cbu)
#pragma unroll
        for (int j = 0; j < 4; ++j) c[cbu][j] = 0.f;

    const size_t pstep = (size_t)NBT * 16 * 1024, cstep = (size_t)NBT * 4096;
    PE* pblk = P + ((size_t)d * T * NBT + bt) * 16 * 1024 + (size_t)w * 4096 + s0 * 512;
    CE* cblk = SAVE ? Csave + ((size_t)d * T * NBT + bt) * 4096 + (size_t)w * 1024 + s0 * 512 : nullptr;
    const unsigned lane_p = frag_lane<PE>(rq, c16);        // P / saved gates (storage type PE)
    const unsigned lane_c = frag_lane<CE>(rq, c16);        // c (storage type CE)
    const int DH = D * H;
    const int t_first = d ? T - 1 : 0, dt = d ? -1 : 1;
    const int row0 = bt * 32 + s0 * 16;

    Raw16<PE> pa, pb;             // P two steps ahead (raw): pa = step s, pb = step s+1
    load_raw16(pblk + (size_t)t_first * pstep, lane_p, pa);
    if (T > 1) load_raw16(pblk + (size_t)(t_first + dt) * pstep, lane_p, pb);
    __syncthreads();

    auto one_step = [&](int step, Raw16<PE>& praw, int cur) {
        const int t = t_first + dt * step;
        f32x4 acc[4][2];
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int cbu = 0; cbu < 2; ++cbu)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[g][cbu][e] = (float)praw.v[2 * g + cbu][e];
        if (step + 2 < T) load_raw16(pblk + (size_t)(t + 2 * dt) * pstep, lane_p, praw);
        // ---- z = P_t + h_{t-1} W_hh^T
        const __bf16* hrow = hs + cur * 16 * HB_LD + c16 * HB_LD + 8 * rq;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const bf16x8 a = *reinterpret_cast<const bf16x8*>(hrow + 32 * ks);
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int cbu = 0; cbu < 2; ++cbu) acc[g][cbu] = mfma16_bf16(a, wr[g][cbu][ks], acc[g][cbu]);
        }
        __bf16* hnext = hs + (cur ^ 1) * 16 * HB_LD + 4 * rq * HB_LD + 32 * w + c16;
        float* ynext = yfs + (YF32 ? (cur ^ 1) * 16 * YF_LD + 4 * rq * YF_LD + 32 * w + c16 : 0);
        if constexpr (PARTS > 1) {
            auto pick = [&](const f32x4& v, int jj) -> float {      // register jlo + jj (jlo is workgroup-uniform: selects)
                if constexpr (PARTS == 2) return jlo ? v[2 + jj] : v[jj];
                else return jlo == 0 ? v[0] : (jlo == 1 ? v[1] : (jlo == 2 ? v[2] : v[3]));
            };
#pragma unroll
            for (int cbu = 0; cbu < 2; ++cbu)
#pragma unroll
                for (int jj = 0; jj < RPP; ++jj) {
                    const float ig = fast_sigmoid(pick(acc[0][cbu], jj));
                    const float fg = fast_sigmoid(pick(acc[1][cbu], jj));
                    const float gg = fast_tanh(pick(acc[2][cbu], jj));
                    const float og = fast_sigmoid(pick(acc[3][cbu], jj));
                    c[cbu][jj] = __builtin_fmaf(fg, c[cbu][jj], ig * gg);
                    const float h = og * fast_tanh(c[cbu][jj]);
                    hnext[(jlo + jj) * HB_LD + 16 * cbu] = (__bf16)h;
                    if (YF32) ynext[(jlo + jj) * YF_LD + 16 * cbu] = h;
                }
        } else
#pragma unroll
        for (int cbu = 0; cbu < 2; ++cbu)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (FEW && j >= nj) continue;              // wave-uniform: a register of padding rows only
                const float ig = fast_sigmoid(acc[0][cbu][j]);
                const float fg = fast_sigmoid(acc[1][cbu][j]);
                const float gg = fast_tanh(acc[2][cbu][j]);
                const float og = fast_sigmoid(acc[3][cbu][j]);
                c[cbu][j] = __builtin_fmaf(fg, c[cbu][j], ig * gg);
                const float h = og * fast_tanh(c[cbu][j]);
                hnext[j * HB_LD + 16 * cbu] = (__bf16)h;
                if (YF32) ynext[j * YF_LD + 16 * cbu] = h;
                if (SAVE) { acc[0][cbu][j] = ig; acc[1][cbu][j] = fg; acc[2][cbu][j] = gg; acc[3][cbu][j] = og; }
            }
        if (SAVE) {
            store_frag16(pblk + (size_t)t * pstep, lane_p, acc);
            CE* cp = cblk + (size_t)t * cstep;
#pragma unroll
            for (int cbu = 0; cbu < 2; ++cbu) {
                if constexpr (sizeof(CE) == 4) {
                    f32x4 v = {c[cbu][0], c[cbu][1], c[cbu][2], c[cbu][3]};
                    *reinterpret_cast<f32x4*>((cp + cbu * FRAG_CBU<CE>) + lane_c) = v;
                } else {
                    bf16x4 v = {(__bf16)c[cbu][0], (__bf16)c[cbu][1], (__bf16)c[cbu][2], (__bf16)c[cbu][3]};
                    if constexpr (LOB_NT128 & 2) __builtin_nontemporal_store(v, reinterpret_cast<bf16x4*>((cp + cbu * FRAG_CBU<CE>) + lane_c));
                    else *reinterpret_cast<bf16x4*>((cp + cbu * FRAG_CBU<CE>) + lane_c) = v;
                }
            }
        }
        __syncthreads();
        if (YF32 && mine) {
            const int row = tid >> 4, c8 = (tid & 15) * 8;
            const float* ysrc = yfs + (cur ^ 1) * 16 * YF_LD + row * YF_LD + c8;
            float* dst = Y + ((size_t)t * Bp + row0 + row) * DH + d * H + c8;
            *reinterpret_cast<f32x4*>(dst) = *reinterpret_cast<const f32x4*>(ysrc);
            *reinterpret_cast<f32x4*>(dst + 4) = *reinterpret_cast<const f32x4*>(ysrc + 4);
        }
        if ((Y16 || DROP) && mine) {       // h_t is complete in hs[cur ^ 1]: emit the bf16 row segments (16 rows x 256 B)
            const __bf16* hsrc = hs + (cur ^ 1) * 16 * HB_LD;
            const int row = tid >> 4, c8 = (tid & 15) * 8;
            const bf16x8 hv = *reinterpret_cast<const bf16x8*>(hsrc + row * HB_LD + c8);
            const size_t o = ((size_t)t * Bp + row0 + row) * DH + d * H + c8;
            if (Y16) { if constexpr (LOB_NT128 & 4) __builtin_nontemporal_store(hv, reinterpret_cast<bf16x8*>(Y16p + o)); else *reinterpret_cast<bf16x8*>(Y16p + o) = hv; }
            if (DROP) {
                bf16x8 dv;
#pragma unroll
                for (int j = 0; j < 8; j += 2) {          // o is a multiple of 8: (o+j, o+j+1) share one hash
                    float s0, s1;
                    lob_dropout_scale2(seed, (uint64_t)o + j, drop_p, s0, s1);
                    dv[j] = (__bf16)((float)hv[j] * s0);
                    dv[j + 1] = (__bf16)((float)hv[j + 1] * s1);
                }
                if constexpr (LOB_NT128 & 4) __builtin_nontemporal_store(dv, reinterpret_cast<bf16x8*>(Yd + o));
                else *reinterpret_cast<bf16x8*>(Yd + o) = dv;
            }
        }
    };

    for (int step = 0; step < T; step += 2) {
        one_step(step, pa, 0);
        if (step + 1 < T) one_step(step + 1, pb, 1);
    }
}

// ------------------------------------------------------------------------------------------
// BPTT on 16-row tiles.  dgates are rounded to bf16 once: the LDS tile feeds the MFMA A operand AND is the dP
// image copied to HBM.
// ------------------------------------------------------------------------------------------
template <typename PE, typename DE, typename CE>
__global__ __launch_bounds__(256, 2) void lstm_rec_bwd_h128_bf16_s16_kernel(
    const PE* __restrict__ G, const CE* __restrict__ Csave, const float* __restrict__ Whh,
    const DE* __restrict__ dY, __bf16* __restrict__ dP, float* __restrict__ dbias, float* __restrict__ dbias2, int T, int Bp) {
    __shared__ __attribute__((aligned(16))) __bf16 dgs[16 * DGB_LD];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int d = blockIdx.y, D = gridDim.y, NBT = Bp >> 5;
    const int c16 = lane & 15, rq = lane >> 4;
    const int bt = blockIdx.x >> 1, s0 = blockIdx.x & 1;

    // B fragments of dh = dgates * W_hh: wt[cbu][ks] = W_hh[n = 32ks + 8rq + j][col = 32w + 16cbu + c16]
    bf16x8 wt[2][16];
    {
        const float* wb = Whh + (size_t)d * 4 * H * H + 32 * w + c16;
#pragma unroll
        for (int cbu = 0; cbu < 2; ++cbu)
#pragma unroll
            for (int ks = 0; ks < 16; ++ks) {
                bf16x8 f;
#pragma unroll
                for (int j = 0; j < 8; ++j) f[j] = (__bf16)wb[(size_t)(32 * ks + 8 * rq + j) * H + 16 * cbu];
                wt[cbu][ks] = f;
                __builtin_amdgcn_sched_barrier(0);      // one fragment at a time: keeps the prologue's register peak low
            }
    }
    const size_t gstep = (size_t)NBT * 16 * 1024, cstep = (size_t)NBT * 4096;
    const PE* gwave = G + ((size_t)d * T * NBT + bt) * 16 * 1024 + (size_t)w * 4096 + s0 * 512;
    const CE* cwave = Csave + ((size_t)d * T * NBT + bt) * 4096 + (size_t)w * 1024 + s0 * 512;
    const unsigned lane_p = frag_lane<PE>(rq, c16);        // saved gates (storage type PE)
    const unsigned lane_c = frag_lane<CE>(rq, c16);        // c (storage type CE)
    const int DH = D * H, D4H = D * 4 * H;
    const int row0 = bt * 32 + s0 * 16;
    const DE* dywave = dY + (size_t)row0 * DH + d * H + 32 * w;
    const unsigned dy_off = (unsigned)(4 * rq * DH + c16);
    const unsigned dp_off = (unsigned)((tid >> 6) * D4H + (tid & 63) * 8);

    const int t_first = d ? 0 : T - 1, dt = d ? 1 : -1;
    f32x4 ct[2], dhrec[2];
    float dcarry[2][4];
    float dbsum[4][2];
#pragma unroll
    for (int cbu = 0; cbu < 2; ++cbu) {
#pragma unroll
        for (int j = 0; j < 4; ++j) { dcarry[cbu][j] = 0.f; dhrec[cbu][j] = 0.f; }
#pragma unroll
        for (int g = 0; g < 4; ++g) dbsum[g][cbu] = 0.f;
    }

    auto load_c = [&](int t, f32x4 (&dst)[2]) {
        if (t >= 0 && t < T) {
            const CE* cq = cwave + (size_t)t * cstep;
#pragma unroll
            for (int cbu = 0; cbu < 2; ++cbu) {
                if constexpr (sizeof(CE) == 4) dst[cbu] = *reinterpret_cast<const f32x4*>((cq + cbu * FRAG_CBU<CE>) + lane_c);
                else {
                    const bf16x4 v = *reinterpret_cast<const bf16x4*>((cq + cbu * FRAG_CBU<CE>) + lane_c);
                    f32x4 f = {(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
                    dst[cbu] = f;
                }
            }
        } else {
#pragma unroll
            for (int cbu = 0; cbu < 2; ++cbu) { f32x4 z = {0.f, 0.f, 0.f, 0.f}; dst[cbu] = z; }
        }
    };
    // (Tried: a second register set to prefetch two steps ahead -- the kernel then spills inside the loop and
    //  runs 1.4 -> 2.1 ms.)
    struct Pre { Raw16<PE> g; f32x4 cp[2]; float dy[2][4]; };
    Pre pa;
    auto load_step = [&](int t, Pre& s) {
        load_raw16(gwave + (size_t)t * gstep, lane_p, s.g);
        load_c(t + dt, s.cp);
        const DE* dp = dywave + (size_t)t * Bp * DH;
#pragma unroll
        for (int cbu = 0; cbu < 2; ++cbu)
#pragma unroll
            for (int j = 0; j < 4; ++j) s.dy[cbu][j] = (float)(dp + (size_t)j * DH + 16 * cbu)[dy_off];
    };
    load_c(t_first, ct);
    load_step(t_first, pa);

    for (int step = 0; step < T; ++step) {
        Pre& s = pa;
        const int t = t_first + dt * step;
        __bf16* dgw = dgs + 4 * rq * DGB_LD + 32 * w + c16;
#pragma unroll
        for (int cbu = 0; cbu < 2; ++cbu)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float ig = (float)s.g.v[0 + cbu][j], fg = (float)s.g.v[2 + cbu][j];
                const float gg = (float)s.g.v[4 + cbu][j], og = (float)s.g.v[6 + cbu][j];
                const float dh = s.dy[cbu][j] + dhrec[cbu][j];
                const float tc = fast_tanh(ct[cbu][j]);
                const float dc = dcarry[cbu][j] + dh * og * (1.f - tc * tc);
                dcarry[cbu][j] = dc * fg;
                __bf16* p = dgw + j * DGB_LD + 16 * cbu;
                const float v0 = dc * gg * ig * (1.f - ig), v1 = dc * s.cp[cbu][j] * fg * (1.f - fg);
                const float v2 = dc * ig * (1.f - gg * gg), v3 = dh * tc * og * (1.f - og);
                p[0 * H] = (__bf16)v0; p[1 * H] = (__bf16)v1; p[2 * H] = (__bf16)v2; p[3 * H] = (__bf16)v3;
                dbsum[0][cbu] += v0; dbsum[1][cbu] += v1; dbsum[2][cbu] += v2; dbsum[3][cbu] += v3;
            }
        ct[0] = s.cp[0]; ct[1] = s.cp[1];
        __syncthreads();
        if (step + 1 < T) load_step(t + dt, s);
#pragma unroll
        for (int cbu = 0; cbu < 2; ++cbu) { f32x4 z = {0.f, 0.f, 0.f, 0.f}; dhrec[cbu] = z; }
        const __bf16* arow = dgs + c16 * DGB_LD + 8 * rq;
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) {
            const bf16x8 a = *reinterpret_cast<const bf16x8*>(arow + 32 * ks);
            dhrec[0] = mfma16_bf16(a, wt[0][ks], dhrec[0]);
            dhrec[1] = mfma16_bf16(a, wt[1][ks], dhrec[1]);
            if ((ks & 3) == 3) __builtin_amdgcn_sched_barrier(0);     // keep at most 4 A fragments in flight
        }
        // ---- the bf16 tile IS the dP image: 16 rows x 1 KB, one row per wave per pass
        __bf16* dpb = dP + ((size_t)t * Bp + row0) * D4H + d * 4 * H;
        const __bf16* src = dgs + (tid >> 6) * DGB_LD + (tid & 63) * 8;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            *reinterpret_cast<bf16x8*>((dpb + (size_t)(4 * i) * D4H) + dp_off) =
                *reinterpret_cast<const bf16x8*>(src + 4 * i * DGB_LD);
        __syncthreads();
    }
    if (dbias) {
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int cbu = 0; cbu < 2; ++cbu) {
                float v = dbsum[g][cbu];
                v += __shfl_xor(v, 16, 64);
                v += __shfl_xor(v, 32, 64);
                if (rq == 0) {
                    const size_t bi = (size_t)d * 4 * H + g * H + 32 * w + 16 * cbu + c16;
                    atomicAdd(dbias + bi, v);
                    if (dbias2) atomicAdd(dbias2 + bi, v);          // b_ih and b_hh: the same gradient, two destinations
                }
            }
    }
}


// ------------------------------------------------------------------------------------------
// BPTT, 16-row tiles, operands streamed by LDS-DMA.
//
// Measured on the register-prefetch kernel above (B = 4096): 1.43 ms per launch with its loads, 0.79 ms with
// the loads removed -- one step of prefetch distance (the registers are busy until the cell backward has
// consumed them) does not cover the HBM latency under load, and a second register set spills (2.1 ms).
// This kernel: 1.31 ms = 4.9 TB/s of a 2-reads-per-write stream (torch's add, the same mix, reaches 6.0).  Here the saved gates and c_{t-1} of
// step s+2 are DMA'd (global_load_lds_dwordx4: no VGPRs) into a wave-private two-slot LDS ring while step s
// computes; dY is prefetched two steps ahead in registers.  Waits on the ring are counted by hand
// (s_waitcnt vmcnt(N): VMEM operations of a wave retire in order, N = operations issued after the DMA that is
// needed); barriers are raw s_barrier.
// ------------------------------------------------------------------------------------------
typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void gbl_cvoid;

template <bool C16> constexpr int RING_WAVE = 4096 + (C16 ? 1024 : 2048);   // bytes per wave per slot: G (4 gates x 1 KB bf16) + c (1 KB bf16 / 2 KB fp32)
template <bool C16> constexpr int RING_SLOT = 4 * RING_WAVE<C16>;          // 20 / 24 KB

// DY16: the incoming gradient dY is stored as bf16 (the dX GEMM / LayerNorm backward above write it so): the eight
// hand-issued loads per step become global_load_ushort (same count: the vmcnt bookkeeping is unchanged) and the
// 16 bits are shifted into an fp32 after the wait.
// C16: the saved cell state is bf16 ([q pair][lane][8], one 1-KB DMA instruction per step instead of two fp32 ones):
// every step then issues 8 + 5 + 4 = 17 VMEM operations instead of 18, and both counted waits drop by one.
template <int D, bool DY16, bool C16>
__global__ __launch_bounds__(256, 2) void lstm_rec_bwd_h128_bf16_s16_dma_kernel(
    const __bf16* __restrict__ G, const void* __restrict__ Csavev, const float* __restrict__ Whh,
    const void* __restrict__ dYv, __bf16* __restrict__ dP, float* __restrict__ dbias, float* __restrict__ dbias2, int T, int Bp) {
    constexpr int NCD = C16 ? 1 : 2;                         // DMA instructions for c per step
    constexpr int VM_FIRST = 8 + 4 + NCD, VM_LOOP = VM_FIRST + 4;     // see the wait before the loop
    constexpr int CB = C16 ? 2 : 4;                          // bytes per stored c element
    __shared__ __attribute__((aligned(16))) __bf16 dgs[16 * DGB_LD];
    __shared__ __attribute__((aligned(1024))) unsigned char ring[2 * RING_SLOT<C16>];
    __shared__ float dbs[8 * 256];                 // bias-gradient partial sums, lane-private: [g][cbu][tid]
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int d = blockIdx.y, NBT = Bp >> 5;
    const int c16 = lane & 15, rq = lane >> 4;
    const int bt = blockIdx.x >> 1, s0 = blockIdx.x & 1;

    bf16x8 wt[2][16];
    {
        const float* wb = Whh + (size_t)d * 4 * H * H + 32 * w + c16;
#pragma unroll
        for (int cbu = 0; cbu < 2; ++cbu)
#pragma unroll
            for (int ks = 0; ks < 16; ++ks) {
                bf16x8 f;
#pragma unroll
                for (int j = 0; j < 8; ++j) f[j] = (__bf16)wb[(size_t)(32 * ks + 8 * rq + j) * H + 16 * cbu];
                wt[cbu][ks] = f;
                __builtin_amdgcn_sched_barrier(0);      // one fragment at a time: keeps the prologue's register peak low
            }
    }
    const size_t gstep = (size_t)NBT * 16 * 1024, cstep = (size_t)NBT * 4096;
    const __bf16* gwave = G + ((size_t)d * T * NBT + bt) * 16 * 1024 + (size_t)w * 4096 + s0 * 512;
    const char* cwave = reinterpret_cast<const char*>(Csavev) +
                        (((size_t)d * T * NBT + bt) * 4096 + (size_t)w * 1024 + s0 * 512) * CB;
    const unsigned lane_g = 2 * frag_lane<__bf16>(rq, c16);   // byte offset of this lane's gates in a 1-KB gate chunk
    const unsigned lane_c = frag_lane<float>(rq, c16);        // element offset of this lane's c values (fp32 c)
    constexpr int DH = D * H, D4H = D * 4 * H;
    const int row0 = bt * 32 + s0 * 16;
    constexpr int DYB = DY16 ? 2 : 4;                          // bytes per dY element
    const char* dywave = reinterpret_cast<const char*>(dYv) + ((size_t)row0 * DH + d * H + 32 * w) * DYB;
    const unsigned dy_off = (unsigned)(4 * rq * DH + c16) * DYB;
    const unsigned dp_off = (unsigned)((tid >> 6) * D4H + (tid & 63) * 8);
    const int t_first = d ? 0 : T - 1, dt = d ? 1 : -1;

    f32x4 ct[2], dhrec[2];
    float dcarry[2][4];
#pragma unroll
    for (int cbu = 0; cbu < 2; ++cbu) {
#pragma unroll
        for (int j = 0; j < 4; ++j) { dcarry[cbu][j] = 0.f; dhrec[cbu][j] = 0.f; }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) dbs[i * 256 + tid] = 0.f;
    {   // c of the first step: plain load
        const char* cq = cwave + (size_t)t_first * cstep * CB;
#pragma unroll
        for (int cbu = 0; cbu < 2; ++cbu) {
            if constexpr (C16) {
                const bf16x4 v = *reinterpret_cast<const bf16x4*>(cq + lane_g + cbu * 256);
                f32x4 f = {(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
                ct[cbu] = f;
            } else {
                ct[cbu] = *reinterpret_cast<const f32x4*>(cq + 4 * (cbu * 64 + lane_c));
            }
        }
    }
    unsigned char* wring = ring + w * RING_WAVE<C16>;     // this wave's part of slot 0; slot 1 at + RING_SLOT
    const unsigned ring_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)wring;
    // DMA of step u into slot u & 1: 4 x 1 KB of saved gates, NCD x 1 KB of c_{t-1} (absent at the last step)
    // (u is clamped to the last step: the tail re-fetches it into a slot nobody reads again, so that every step
    //  issues the same number of VMEM operations and one counted wait is valid for all of them)
    auto dma_step = [&](int u_) {
        const int u = u_ < T ? u_ : T - 1;
        const int t = t_first + dt * u;
        unsigned char* dst = wring + (u_ & 1) * RING_SLOT<C16>;
        const __bf16* gsrc = gwave + (size_t)t * gstep + lane * 8;
#pragma unroll
        for (int g = 0; g < 4; ++g)
            __builtin_amdgcn_global_load_lds((gbl_cvoid*)(gsrc + g * 1024), (lds_void*)(dst + g * 1024), 16, 0, 0);
        const int tc = u + 1 < T ? t + dt : t;            // c_{t-1}; the last step has none (cp = 0 there): any valid block
        const char* csrc = cwave + (size_t)tc * cstep * CB + lane * 16;
#pragma unroll
        for (int i = 0; i < NCD; ++i)
            __builtin_amdgcn_global_load_lds((gbl_cvoid*)(csrc + i * 1024), (lds_void*)(dst + 4096 + i * 1024), 16, 0, 0);
    };
    float dya[2][4], dyb[2][4];
    // dY loads are issued by hand (inline asm), so that the compiler's own wait-count bookkeeping never sees a
    // pending register load in this loop -- it would drain the whole queue (vmcnt(0)) at the loop header.  The
    // matching wait is the s_waitcnt below, which names the registers so that no use can move above it.
    const char* dylane = dywave + dy_off;
    auto load_dy = [&](int u_, float (&dy)[2][4]) {
        const int u = u_ < T ? u_ : T - 1;
        const char* dp = dylane + (size_t)(t_first + dt * u) * Bp * DH * DYB;
#define LOB_DY_LOADS(OP)                                                                                    \
        asm volatile(                                                                                       \
            OP " %0, %8, off offset:%9\n\t"                                                                 \
            OP " %1, %8, off offset:%10\n\t"                                                                \
            OP " %2, %8, off offset:%11\n\t"                                                                \
            OP " %3, %8, off offset:%12\n\t"                                                                \
            OP " %4, %8, off offset:%13\n\t"                                                                \
            OP " %5, %8, off offset:%14\n\t"                                                                \
            OP " %6, %8, off offset:%15\n\t"                                                                \
            OP " %7, %8, off offset:%16"                                                                    \
            : "=&v"(dy[0][0]), "=&v"(dy[0][1]), "=&v"(dy[0][2]), "=&v"(dy[0][3]),                           \
              "=&v"(dy[1][0]), "=&v"(dy[1][1]), "=&v"(dy[1][2]), "=&v"(dy[1][3])                            \
            : "v"(dp), "n"(0 * DH * DYB), "n"(1 * DH * DYB), "n"(2 * DH * DYB), "n"(3 * DH * DYB),          \
              "n"(0 * DH * DYB + 16 * DYB), "n"(1 * DH * DYB + 16 * DYB), "n"(2 * DH * DYB + 16 * DYB),      \
              "n"(3 * DH * DYB + 16 * DYB)                                                                  \
            : "memory")
        if constexpr (DY16) LOB_DY_LOADS("global_load_ushort");
        else                LOB_DY_LOADS("global_load_dword");
#undef LOB_DY_LOADS
    };
    load_dy(0, dya);
    dma_step(0);
    load_dy(1, dyb);
    dma_step(1);
    // VMEM operations of this wave younger than DMA(s) when step s starts (fp32 c: 6 DMA instructions per step, bf16
    // c: 5): step 0: dy(1) 8 + DMA(1) 6|5 = 14|13 (waited for here); step 1: dy(2) 8 + DMA(2) 6|5 + stores(0) 4 = 18|17;
    // later steps 22|21 -> one in-loop wait, vmcnt(18|17)
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(VM_FIRST) : "memory");
    auto one_step = [&](int step, float (&dy)[2][4]) {
        const int t = t_first + dt * step;
        asm volatile("s_waitcnt vmcnt(%8)"
                     : "+v"(dy[0][0]), "+v"(dy[0][1]), "+v"(dy[0][2]), "+v"(dy[0][3]),
                       "+v"(dy[1][0]), "+v"(dy[1][1]), "+v"(dy[1][2]), "+v"(dy[1][3]) : "n"(VM_LOOP) : "memory");
        if constexpr (DY16) {               // zero-extended bf16 bits -> fp32
#pragma unroll
            for (int cbu = 0; cbu < 2; ++cbu)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    dy[cbu][j] = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, dy[cbu][j]) << 16);
        }
        // ring reads through inline asm: an ordinary LDS read of a DMA target makes hipcc wait for ALL outstanding
        // DMAs (vmcnt(0)), i.e. also for the slot that is being filled for the step after next
        const unsigned ring_a = ring_base + (unsigned)((step & 1) * RING_SLOT<C16>);
        __bf16* dgw = dgs + 4 * rq * DGB_LD + 32 * w + c16;
        f32x4 cp[2];
#pragma unroll
        for (int cbu = 0; cbu < 2; ++cbu) {
            u32x2 ri, rf, rc, ro;
            f32x4 cpl;
            if constexpr (C16) {            // c_{t-1} as bf16, same lane mapping as one gate chunk
                u32x2 rcp;
                if (cbu == 0)
                    asm volatile("ds_read_b64 %0, %5 offset:0\n\tds_read_b64 %1, %5 offset:1024\n\t"
                                 "ds_read_b64 %2, %5 offset:2048\n\tds_read_b64 %3, %5 offset:3072\n\t"
                                 "ds_read_b64 %4, %5 offset:4096\n\ts_waitcnt lgkmcnt(0)"
                                 : "=&v"(ri), "=&v"(rf), "=&v"(rc), "=&v"(ro), "=&v"(rcp)
                                 : "v"(ring_a + lane_g) : "memory");
                else
                    asm volatile("ds_read_b64 %0, %5 offset:256\n\tds_read_b64 %1, %5 offset:1280\n\t"
                                 "ds_read_b64 %2, %5 offset:2304\n\tds_read_b64 %3, %5 offset:3328\n\t"
                                 "ds_read_b64 %4, %5 offset:4352\n\ts_waitcnt lgkmcnt(0)"
                                 : "=&v"(ri), "=&v"(rf), "=&v"(rc), "=&v"(ro), "=&v"(rcp)
                                 : "v"(ring_a + lane_g) : "memory");
                const bf16x4 cb4 = __builtin_bit_cast(bf16x4, rcp);
                f32x4 f = {(float)cb4[0], (float)cb4[1], (float)cb4[2], (float)cb4[3]};
                cpl = f;
            } else if (cbu == 0)
                asm volatile("ds_read_b64 %0, %5 offset:0\n\tds_read_b64 %1, %5 offset:1024\n\t"
                             "ds_read_b64 %2, %5 offset:2048\n\tds_read_b64 %3, %5 offset:3072\n\t"
                             "ds_read_b128 %4, %6 offset:4096\n\ts_waitcnt lgkmcnt(0)"
                             : "=&v"(ri), "=&v"(rf), "=&v"(rc), "=&v"(ro), "=&v"(cpl)
                             : "v"(ring_a + lane_g), "v"(ring_a + 4 * lane_c) : "memory");
            else
                asm volatile("ds_read_b64 %0, %5 offset:256\n\tds_read_b64 %1, %5 offset:1280\n\t"
                             "ds_read_b64 %2, %5 offset:2304\n\tds_read_b64 %3, %5 offset:3328\n\t"
                             "ds_read_b128 %4, %6 offset:4352\n\ts_waitcnt lgkmcnt(0)"
                             : "=&v"(ri), "=&v"(rf), "=&v"(rc), "=&v"(ro), "=&v"(cpl)
                             : "v"(ring_a + lane_g), "v"(ring_a + 4 * lane_c) : "memory");
            const bf16x4 gi = __builtin_bit_cast(bf16x4, ri), gf = __builtin_bit_cast(bf16x4, rf);
            const bf16x4 gc = __builtin_bit_cast(bf16x4, rc), go = __builtin_bit_cast(bf16x4, ro);
            if (step + 1 < T) cp[cbu] = cpl;
            else { f32x4 z = {0.f, 0.f, 0.f, 0.f}; cp[cbu] = z; }
            float s0v = 0.f, s1v = 0.f, s2v = 0.f, s3v = 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float ig = (float)gi[j], fg = (float)gf[j], gg = (float)gc[j], og = (float)go[j];
                const float dh = dy[cbu][j] + dhrec[cbu][j];
                const float tc = fast_tanh(ct[cbu][j]);
                const float dc = dcarry[cbu][j] + dh * og * (1.f - tc * tc);
                dcarry[cbu][j] = dc * fg;
                __bf16* p = dgw + j * DGB_LD + 16 * cbu;
                const float v0 = dc * gg * ig * (1.f - ig), v1 = dc * cp[cbu][j] * fg * (1.f - fg);
                const float v2 = dc * ig * (1.f - gg * gg), v3 = dh * tc * og * (1.f - og);
                p[0 * H] = (__bf16)v0; p[1 * H] = (__bf16)v1; p[2 * H] = (__bf16)v2; p[3 * H] = (__bf16)v3;
                s0v += v0; s1v += v1; s2v += v2; s3v += v3;
            }
            // four rows summed in registers, then one LDS accumulate per (gate, column block)
            dbs[(0 + cbu) * 256 + tid] += s0v; dbs[(2 + cbu) * 256 + tid] += s1v;
            dbs[(4 + cbu) * 256 + tid] += s2v; dbs[(6 + cbu) * 256 + tid] += s3v;
        }
        ct[0] = cp[0]; ct[1] = cp[1];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        load_dy(step + 2, dy);             // this wave is done with slot step & 1 and with this dy set
        dma_step(step + 2);
#pragma unroll
        for (int cbu = 0; cbu < 2; ++cbu) { f32x4 z = {0.f, 0.f, 0.f, 0.f}; dhrec[cbu] = z; }
        const __bf16* arow = dgs + c16 * DGB_LD + 8 * rq;
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) {
            const bf16x8 a = *reinterpret_cast<const bf16x8*>(arow + 32 * ks);
            dhrec[0] = mfma16_bf16(a, wt[0][ks], dhrec[0]);
            dhrec[1] = mfma16_bf16(a, wt[1][ks], dhrec[1]);
            if ((ks & 3) == 3) __builtin_amdgcn_sched_barrier(0);
        }
        __bf16* dpb = dP + ((size_t)t * Bp + row0) * D4H + d * 4 * H;
        const __bf16* src = dgs + (tid >> 6) * DGB_LD + (tid & 63) * 8;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            *reinterpret_cast<bf16x8*>((dpb + (size_t)(4 * i) * D4H) + dp_off) =
                *reinterpret_cast<const bf16x8*>(src + 4 * i * DGB_LD);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    };
    for (int step = 0; step < T; step += 2) {
        one_step(step, dya);
        if (step + 1 < T) one_step(step + 1, dyb);
    }
    if (dbias) {
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int cbu = 0; cbu < 2; ++cbu) {
                float v = dbs[(2 * g + cbu) * 256 + tid];
                v += __shfl_xor(v, 16, 64);
                v += __shfl_xor(v, 32, 64);
                if (rq == 0) {
                    const size_t bi = (size_t)d * 4 * H + g * H + 32 * w + 16 * cbu + c16;
                    atomicAdd(dbias + bi, v);
                    if (dbias2) atomicAdd(dbias2 + bi, v);          // b_ih and b_hh: the same gradient, two destinations
                }
            }
    }
}

}  // namespace

// Internal entry points used by lob_lstm_rec_fwd_bf16 / lob_lstm_rec_bwd_bf16 (lstm_rec_bf16.hip).
int lob_rec_fwd_bf16_s16(void* P, int pg_bf16, const float* Whh, float* Y, void* Csave, int c_bf16, void* Y16, void* Yd,
                         float drop_p, uint64_t seed, int T, int Bp, int D, int save, int nvalid, hipStream_t s) {
    const dim3 grid(Bp / 16, D), block(256);
    __bf16* y16 = reinterpret_cast<__bf16*>(Y16);
    __bf16* yd = reinterpret_cast<__bf16*>(Yd);
    if (c_bf16 && !(pg_bf16 && save)) return LOB_E_SHAPE;          // bf16 c: with bf16 saved gates only
    // FEW: one window per call (with two or three windows four workgroups per tile -- PARTS = 4 below -- are faster: 0.85 ms per
    // forward against 0.89 / 1.00; LOB_VAR_REC_HALF = 0 keeps FEW for them)
    if (!save && !yd && pg_bf16 && nvalid > 0 && (nvalid < 2 || (nvalid < 4 && !lob_variant(LOB_VAR_REC_HALF))) &&
        lob_variant(LOB_VAR_REC_FEW)) {
#define LOB_FEW(YF, Y6) hipLaunchKernelGGL((lstm_rec_fwd_h128_bf16_s16_kernel<false, YF, Y6, false, __bf16, float, true>), grid, block, \
        0, s, reinterpret_cast<__bf16*>(P), Whh, Y, (float*)nullptr, y16, yd, 0.f, (uint64_t)0, T, Bp, nvalid)
        if (Y && y16) LOB_FEW(true, true); else if (Y) LOB_FEW(true, false); else LOB_FEW(false, true);
#undef LOB_FEW
        LOB_CHECK_LAUNCH();
        return 0;
    }
    const int hv = lob_variant(LOB_VAR_REC_HALF), htiles = (Bp / 16) * D;
    if (!save && !yd && pg_bf16 && hv && htiles <= 128) {
        // LOB_VAR_REC_HALF: 1 = four workgroups per tile up to 64 tiles (B <= 512: -14 % per forward), two up to 128 tiles
        // (B = 1024: -5 %; four there: +9 %); 2 / 4 force that split (A/B, tools/half_tile_ab.py)
        const int parts = (hv == 2 || hv == 4) ? hv : (htiles <= 64 ? 4 : 2);
        const dim3 gridp((Bp / 16) * parts, D);
#define LOB_PART(YF, Y6, NP) hipLaunchKernelGGL((lstm_rec_fwd_h128_bf16_s16_kernel<false, YF, Y6, false, __bf16, float, false, NP>), gridp, \
        block, 0, s, reinterpret_cast<__bf16*>(P), Whh, Y, (float*)nullptr, y16, yd, 0.f, (uint64_t)0, T, Bp, Bp)
#define LOB_PART_OUT(NP) do { if (Y && y16) LOB_PART(true, true, NP); else if (Y) LOB_PART(true, false, NP); else LOB_PART(false, true, NP); } while (0)
        if (parts == 4) LOB_PART_OUT(4); else LOB_PART_OUT(2);
#undef LOB_PART_OUT
#undef LOB_PART
        LOB_CHECK_LAUNCH();
        return 0;
    }
#define LOB_FWD(SV, YF, Y6, DR, PE, CE) hipLaunchKernelGGL((lstm_rec_fwd_h128_bf16_s16_kernel<SV, YF, Y6, DR, PE, CE>), grid, block, \
        0, s, reinterpret_cast<PE*>(P), Whh, Y, reinterpret_cast<CE*>(Csave), y16, yd, drop_p, seed, T, Bp, Bp)
#define LOB_FWD_OUT(SV, PE, CE) do {                                                     \
        if (Y && !y16 && !yd) LOB_FWD(SV, true, false, false, PE, CE);                   \
        else if (Y && y16 && !yd) LOB_FWD(SV, true, true, false, PE, CE);                \
        else if (Y && !y16 && yd) LOB_FWD(SV, true, false, true, PE, CE);                \
        else if (Y && y16 && yd) LOB_FWD(SV, true, true, true, PE, CE);                  \
        else if (!Y && y16 && !yd) LOB_FWD(SV, false, true, false, PE, CE);              \
        else LOB_FWD(SV, false, true, true, PE, CE); } while (0)
    if (pg_bf16 && c_bf16) LOB_FWD_OUT(true, __bf16, __bf16);
    else if (pg_bf16) { if (save) LOB_FWD_OUT(true, __bf16, float); else LOB_FWD_OUT(false, __bf16, float); }
    else              { if (save) LOB_FWD_OUT(true, float, float); else LOB_FWD_OUT(false, float, float); }
#undef LOB_FWD_OUT
#undef LOB_FWD
    LOB_CHECK_LAUNCH();
    return 0;
}

int lob_rec_bwd_bf16_s16(const void* G, int pg_bf16, const void* Csave, int c_bf16, const float* Whh, const void* dY,
                         int dy_bf16, void* dP, float* dbias, float* dbias2, int T, int Bp, int D, hipStream_t s) {
    const dim3 grid(Bp / 16, D), block(256);
    // LOB_VAR_REC_BWD_DMA = 0 selects the register-prefetch kernel (also the only one for fp32 saved gates)
    const bool dma = lob_variant(LOB_VAR_REC_BWD_DMA) != 0;
    const __bf16* g16 = reinterpret_cast<const __bf16*>(G);
    __bf16* dp16 = reinterpret_cast<__bf16*>(dP);
    if ((dy_bf16 || c_bf16) && !pg_bf16) return LOB_E_SHAPE;       // bf16 dY / c only with bf16 saved gates
    if (pg_bf16 && dma) {
#define LOB_BWD_DMA(DD, Y16, C16) hipLaunchKernelGGL((lstm_rec_bwd_h128_bf16_s16_dma_kernel<DD, Y16, C16>), grid, block, 0, s, \
                                                     g16, Csave, Whh, dY, dp16, dbias, dbias2, T, Bp)
#define LOB_BWD_DMA_D(DD) do {                                                               \
        if (dy_bf16 && c_bf16) LOB_BWD_DMA(DD, true, true);                                  \
        else if (dy_bf16)      LOB_BWD_DMA(DD, true, false);                                 \
        else if (c_bf16)       LOB_BWD_DMA(DD, false, true);                                 \
        else                   LOB_BWD_DMA(DD, false, false); } while (0)
        if (D == 2) LOB_BWD_DMA_D(2); else LOB_BWD_DMA_D(1);
#undef LOB_BWD_DMA_D
#undef LOB_BWD_DMA
    }
    else if (pg_bf16) {
#define LOB_BWD_REG(DE, CE) hipLaunchKernelGGL((lstm_rec_bwd_h128_bf16_s16_kernel<__bf16, DE, CE>), grid, block, 0, s, g16, \
                            reinterpret_cast<const CE*>(Csave), Whh, reinterpret_cast<const DE*>(dY), dp16, dbias, dbias2, T, Bp)
        if (dy_bf16 && c_bf16) LOB_BWD_REG(__bf16, __bf16);
        else if (dy_bf16)      LOB_BWD_REG(__bf16, float);
        else if (c_bf16)       LOB_BWD_REG(float, __bf16);
        else                   LOB_BWD_REG(float, float);
#undef LOB_BWD_REG
    }
    else
        hipLaunchKernelGGL((lstm_rec_bwd_h128_bf16_s16_kernel<float, float, float>), grid, block, 0, s,
                           reinterpret_cast<const float*>(G), reinterpret_cast<const float*>(Csave), Whh,
                           reinterpret_cast<const float*>(dY), dp16, dbias, dbias2, T, Bp);
    LOB_CHECK_LAUNCH();
    return 0;
}
