// fp32-accurate recurrent forward kernel (H = 128) on the 16-bit matrix pipe: two-way fp16 operand splits.
//
// The exact-fp32 MFMA (v_mfma_f32_16x16x4_f32) runs at the fp32 VECTOR rate, 1/16 of the 16-bit MFMA rate
// (MI355X_MICROARCH.md, Matrix cores), and the fp32 recurrent forward (lstm_rec_f32_s16.hip) is bound by it: 5.3 us per
// step, 2.4 ms per layer at B = 4096, and at B = 1024 (BASELINE.json configs[1]) its 128 workgroups leave half the
// chip idle.  Here every fp32 operand x is carried as TWO fp16 numbers
//     s = x * 2^8,   hi = fp16(s),   lo = fp16((s - hi) * 2^11)          (s - hi is exact in fp32)
// i.e. 22 significant bits (fp32 has 24), and the product of two such pairs is three 16-bit MFMAs:
//     x y  =  2^-16 hi_x hi_y  +  2^-27 (hi_x lo_y + lo_x hi_y)  [+ 2^-38 lo_x lo_y, dropped: 2^-22 relative]
// fp16 x fp16 products are exact in the fp32 accumulator (11 + 11 bits), and the two groups of terms are summed in
// SEPARATE fp32 accumulators (the small terms do not lose their low bits against the large ones); the pre-activation
// is z = P + 2^-16 acc_hh + 2^-27 acc_small.  The power-of-two pre-scalings keep both halves of every operand in
// fp16's normal range (|h| <= 1, |w| <~ 10^2), so nothing depends on how the matrix pipe treats fp16 denormals.
// Measured against the fp32 reference goldens the logits move by < 1e-6 (the parity bar of configs[1] is 1e-5).
//
// Shape: 16 batch rows per workgroup, EIGHT waves (lstm_rec_f32_s16.hip: four): wave w8 owns the 16 hidden columns
// 32 (w8 >> 1) + 16 (w8 & 1) of all four gates, W_hh of those columns as 2 x 4 x 4 fp16x8 B fragments = 128 VGPRs.
// Per step and wave 48 v_mfma_f32_16x16x32_f16 (16 cycles each) replace 256 v_mfma_f32_16x16x4_f32 (32 cycles each):
// 0.73 us of matrix time per step per SIMD (two waves) instead of 3.9.  Same fragment-order P / saved gates / c
// layouts as the other H = 128 kernels (include/lob.h), so the gate GEMM and BPTT on either side are unchanged.
#include "lob_common.h"

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

constexpr int H = 128;
constexpr int HB_LD = 136;     // fp16 h tile row stride (272 B = 17 x 16 B, odd -> conflict-free b128)
constexpr int YF_LD = 132;     // fp32 h staging row stride (528 B = 33 x 16 B)
constexpr float S_OP = 256.f;              // operand pre-scale 2^8
constexpr float S_LO = 2048.f;             // residual scale 2^11

__device__ __forceinline__ void split2(float x, _Float16& hi, _Float16& lo, float scale = S_OP) {
    const float s = x * scale;
    hi = (_Float16)s;
    lo = (_Float16)((s - (float)hi) * S_LO);
}

__device__ __forceinline__ f32x4 mfma16_f16(f16x8 a, f16x8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}

// HALF (round 4): when the 16-row tiles of a batch fill no more than a quarter of the CUs (Bp <= 512 with two directions) TWO
// workgroups share each tile: a lane of the MFMA's D layout holds rows 4 rq + j, j < 4, of its column; workgroup jh takes
// j in {2 jh, 2 jh + 1}, i.e. tile rows {0,1,4,5,8,9,12,13} or {2,3,6,7,...}.  Twice the workgroups (every CU busy), half the
// cell-update work per lane and step (the step is a serial chain: MFMAs -> activations -> LDS -> barrier, and the
// activations are a third of it); the other half of its h tile stays zero, the MFMAs on it are wasted.  Same arithmetic
// per row: bit-identical to the full-tile kernel (tests/test_gpu_twins.py).  LOB_VAR_REC_HALF = 0 keeps full tiles.
// PARTS = 4 (quarter tiles: four workgroups per tile, one row j each) where full tiles would fill an eighth of the CUs.
// DROP (round 4): Yd = dropout(Y) (nn.LSTM's inter-layer dropout, 04:186: the next layer's input) written next to Y from the
// same staged rows: the stand-alone dropout kernel's mask (element index row * D * H + column), without its read pass.
template <bool SAVE, int PARTS = 1, bool DROP = false>
__global__ __launch_bounds__(512, 2) void lstm_rec_fwd_h128_split_kernel(
    float* __restrict__ P, const float* __restrict__ Whh, float* __restrict__ Y, float* __restrict__ Csave, int T, int Bp,
    const float* __restrict__ range, float* __restrict__ Yd, float drop_p, uint64_t seed) {
    constexpr bool HALF = PARTS > 1;
    constexpr int NJ = 4 / PARTS;
    typedef float fvec __attribute__((ext_vector_type(NJ)));
    __shared__ __attribute__((aligned(16))) _Float16 hs[2 * 2 * 16 * HB_LD];      // [buf][split][16 rows][HB_LD]
    __shared__ __attribute__((aligned(16))) float yfs[2 * 16 * YF_LD];            // fp32 h of the step, for wide stores
    const int tid = threadIdx.x, lane = tid & 63;
    const int w8 = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wcol = w8 >> 1, cbu = w8 & 1;
    const int d = blockIdx.y, D = gridDim.y, NBT = Bp >> 5;
    const int c16 = lane & 15, rq = lane >> 4;
    const int bx = (int)blockIdx.x / PARTS;
    const int jb = NJ * ((int)blockIdx.x % PARTS);            // first of this workgroup's rows j
    const int bt = bx >> 1, s0 = bx & 1;                      // 32-row fragment block, 16-row half
    const int col = 32 * wcol + 16 * cbu + c16;               // this lane's hidden column

    // weight pre-scale from the range of this direction's W_hh (lob.h; h keeps 2^8: |h| < 1);
    // h w = R_HH hi hi + R_SM (hi lo + lo hi)
    const float sw = range ? lob_split_scale(range[d]) : S_OP;
    const float R_HH = 1.f / (S_OP * sw), R_SM = R_HH * (1.f / S_LO);
    // B fragments: W_hh[g*128 + col][32 ks + 8 rq .. + 7], split
    f16x8 whi[4][4], wlo[4][4];
    {
        const float* wbase = Whh + (size_t)d * 4 * H * H;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float* row = wbase + (size_t)(g * H + col) * H + 8 * rq;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const f32x4 a = *reinterpret_cast<const f32x4*>(row + 32 * ks), b = *reinterpret_cast<const f32x4*>(row + 32 * ks + 4);
                f16x8 h8, l8;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    _Float16 hh, ll;
                    split2(a[j], hh, ll, sw); h8[j] = hh; l8[j] = ll;
                    split2(b[j], hh, ll, sw); h8[4 + j] = hh; l8[4 + j] = ll;
                }
                whi[g][ks] = h8; wlo[g][ks] = l8;
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    for (int i = tid; i < 2 * 2 * 16 * HB_LD; i += 512) hs[i] = (_Float16)0.f;
    float c[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) c[j] = 0.f;

    // fragment addressing (lob.h): gate g, 16-row half s0: rows 4 rq .. + 3 of column `col` are ONE float4 at
    // [wcol][g][q = 2 s0 + (rq >> 1)][lane' = (rq & 1) * 32 + 16 cbu + c16][0..3]  (HALF: its components jb, jb + 1)
    const size_t pstep = (size_t)NBT * 16 * 1024, cstep = (size_t)NBT * 4096;
    const unsigned lane_p = (unsigned)((2 * s0 + (rq >> 1)) * 256 + ((rq & 1) * 32 + 16 * cbu + c16) * 4 + jb);
    float* pblk = P + ((size_t)d * T * NBT + bt) * 16 * 1024 + (size_t)wcol * 4096 + lane_p;
    float* cblk = SAVE ? Csave + ((size_t)d * T * NBT + bt) * 4096 + (size_t)wcol * 1024 + lane_p : nullptr;
    const int DH = D * H;
    const int t_first = d ? T - 1 : 0, dt = d ? -1 : 1;
    const int row0 = bt * 32 + s0 * 16;

    fvec pa[4], pb[4];            // P two steps ahead
    auto load_p = [&](int t, fvec (&dst)[4]) {
        const float* p = pblk + (size_t)t * pstep;
#pragma unroll
        for (int g = 0; g < 4; ++g) dst[g] = *reinterpret_cast<const fvec*>(p + g * 1024);
    };
    load_p(t_first, pa);
    if (T > 1) load_p(t_first + dt, pb);
    __syncthreads();

    auto one_step = [&](int step, fvec (&praw)[4], int cur) {
        const int t = t_first + dt * step;
        fvec pz[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) pz[g] = praw[g];
        if (step + 2 < T) load_p(t + 2 * dt, praw);
        f32x4 ahh[4], asm_[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) { f32x4 z = {0.f, 0.f, 0.f, 0.f}; ahh[g] = z; asm_[g] = z; }
        const _Float16* hrow = hs + cur * 2 * 16 * HB_LD + c16 * HB_LD + 8 * rq;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const f16x8 ah = *reinterpret_cast<const f16x8*>(hrow + 32 * ks);
            const f16x8 al = *reinterpret_cast<const f16x8*>(hrow + 16 * HB_LD + 32 * ks);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                ahh[g] = mfma16_f16(ah, whi[g][ks], ahh[g]);
                asm_[g] = mfma16_f16(ah, wlo[g][ks], asm_[g]);
                asm_[g] = mfma16_f16(al, whi[g][ks], asm_[g]);
            }
        }
        _Float16* hnext = hs + (cur ^ 1) * 2 * 16 * HB_LD + (4 * rq + jb) * HB_LD + col;
        float* ynext = yfs + (cur ^ 1) * 16 * YF_LD + (4 * rq + jb) * YF_LD + col;
        fvec gi, gf, gg_, go;
#pragma unroll
        for (int jj = 0; jj < NJ; ++jj) {
            // accumulator component of row 4 rq + jb + jj (HALF: jb is 0 or 2 -- a select, not a dynamic register index)
            const int ja = jj;
            float a_hh[4], a_sm[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                if constexpr (PARTS == 2) {
                    a_hh[g] = jb ? ahh[g][2 + ja] : ahh[g][ja];
                    a_sm[g] = jb ? asm_[g][2 + ja] : asm_[g][ja];
                } else if constexpr (PARTS == 4) {
                    a_hh[g] = jb == 0 ? ahh[g][0] : (jb == 1 ? ahh[g][1] : (jb == 2 ? ahh[g][2] : ahh[g][3]));
                    a_sm[g] = jb == 0 ? asm_[g][0] : (jb == 1 ? asm_[g][1] : (jb == 2 ? asm_[g][2] : asm_[g][3]));
                } else {
                    a_hh[g] = ahh[g][ja]; a_sm[g] = asm_[g][ja];
                }
            }
            const float zi = pz[0][jj] + (a_hh[0] * R_HH + a_sm[0] * R_SM);
            const float zf = pz[1][jj] + (a_hh[1] * R_HH + a_sm[1] * R_SM);
            const float zg = pz[2][jj] + (a_hh[2] * R_HH + a_sm[2] * R_SM);
            const float zo = pz[3][jj] + (a_hh[3] * R_HH + a_sm[3] * R_SM);
            const float ig = fast_sigmoid(zi), fg = fast_sigmoid(zf), gg = fast_tanh(zg), og = fast_sigmoid(zo);
            c[jj] = __builtin_fmaf(fg, c[jj], ig * gg);
            const float h = og * fast_tanh(c[jj]);
            _Float16 hh, hl;
            split2(h, hh, hl);
            hnext[jj * HB_LD] = hh;
            hnext[16 * HB_LD + jj * HB_LD] = hl;
            ynext[jj * YF_LD] = h;
            if (SAVE) { gi[jj] = ig; gf[jj] = fg; gg_[jj] = gg; go[jj] = og; }
        }
        if (SAVE) {
            float* p = pblk + (size_t)t * pstep;
            *reinterpret_cast<fvec*>(p) = gi;
            *reinterpret_cast<fvec*>(p + 1024) = gf;
            *reinterpret_cast<fvec*>(p + 2048) = gg_;
            *reinterpret_cast<fvec*>(p + 3072) = go;
            fvec cv;
#pragma unroll
            for (int jj = 0; jj < NJ; ++jj) cv[jj] = c[jj];
            *reinterpret_cast<fvec*>(cblk + (size_t)t * cstep) = cv;
        }
        __syncthreads();
        {   // h_t is complete in yfs[cur ^ 1]: the tile's rows (HALF: this workgroup's 8) x 512 B leave as one 16-B store per thread
            const int r8 = tid >> 5, c4 = (tid & 31) * 4;
            const int row = PARTS == 1 ? r8 : (PARTS == 2 ? 4 * (r8 >> 1) + jb + (r8 & 1) : 4 * r8 + jb);
            if (tid < 512 / PARTS) {
                f32x4 v = *reinterpret_cast<const f32x4*>(yfs + (cur ^ 1) * 16 * YF_LD + row * YF_LD + c4);
                const size_t e0 = ((size_t)t * Bp + row0 + row) * DH + d * H + c4;
                *reinterpret_cast<f32x4*>(Y + e0) = v;
                if (DROP) {
                    float s0, s1, s2, s3;
                    lob_dropout_scale2(seed, e0, drop_p, s0, s1);
                    lob_dropout_scale2(seed, e0 + 2, drop_p, s2, s3);
                    v[0] *= s0; v[1] *= s1; v[2] *= s2; v[3] *= s3;
                    *reinterpret_cast<f32x4*>(Yd + e0) = v;
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
// BPTT on the same arithmetic (round 4).  dh_rec = dgates[16 x 512] W_hh[512 x 128] per step: the exact-fp32 kernel
// (lstm_rec_f32.hip) spends 7.8 of its 12.7 us per step in 256 v_mfma_f32_32x32x2_f32; here W_hh^T sits in registers as
// split fp16 fragments (128 VGPRs per wave: eight waves x 16 hidden columns, as in the forward) and the step costs 48
// v_mfma_f32_16x16x32_f16 per wave.  The forward's operand h is bounded by 1; dgates are not bounded by anything known in
// advance, so their pre-scale is taken from the tile itself EVERY step: each wave publishes the |max| of the 16 values
// per lane it has just computed, and after a barrier every wave derives the same power of two from the eight maxima
// (lob_split_scale: hi and lo stay inside fp16's normal range whatever the gradient's magnitude).  Three barriers per
// step: maxima, split tile, dP staging.  dP leaves as fp32 rows (or bf16) through an fp32 LDS image; the kernel also
// returns max|dP| over the launch (atomic max on the bit pattern into a zeroed word): the pre-scale of the dX and dW
// GEMMs that consume dP (lob_gemm_nt_f32_split / lob_gemm_tn_f32_split).
// Layouts: lob.h (fragment-order fp32 saved gates / cell states, row-major dY and dP), 16-row tiles like the forward.
// ------------------------------------------------------------------------------------------
constexpr int DGH_LD = 520;    // fp16 dgates tile row stride (1040 B = 65 x 16 B, odd -> conflict-free b128)
constexpr int DPF_LD = 516;    // fp32 dP image row stride (2064 B = 129 x 16 B)

template <bool DP_BF16>
__global__ __launch_bounds__(512, 2) void lstm_rec_bwd_h128_split_kernel(
    const float* __restrict__ G, const float* __restrict__ Csave, const float* __restrict__ Whh, const float* __restrict__ dY,
    void* __restrict__ dPv, float* __restrict__ dbias, float* __restrict__ amax_out, int T, int Bp, const float* __restrict__ range) {
    __shared__ __attribute__((aligned(16))) _Float16 dgs[2 * 16 * DGH_LD];      // [split][16 rows][DGH_LD]
    __shared__ __attribute__((aligned(16))) float dpf[16 * DPF_LD];             // fp32 dgates = the dP image
    __shared__ float wmax[2][8];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w8 = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wcol = w8 >> 1, cbu = w8 & 1;
    const int d = blockIdx.y, D = gridDim.y, NBT = Bp >> 5;
    const int c16 = lane & 15, rq = lane >> 4;
    const int bt = blockIdx.x >> 1, s0 = blockIdx.x & 1;
    const int col = 32 * wcol + 16 * cbu + c16;               // this lane's hidden unit

    const float sw = range ? lob_split_scale(range[d]) : S_OP;
    // B fragments of dh = dgates W_hh: B[k][unit], k = 32 ks + 8 rq + j  (W_hh is [4H][H]: element (k, unit))
    f16x8 whi[16], wlo[16];
    {
        const float* wb = Whh + (size_t)d * 4 * H * H + col;
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) {
            f16x8 h8, l8;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                _Float16 hh, ll;
                split2(wb[(size_t)(32 * ks + 8 * rq + j) * H], hh, ll, sw);
                h8[j] = hh; l8[j] = ll;
            }
            whi[ks] = h8; wlo[ks] = l8;
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const size_t gstep = (size_t)NBT * 16 * 1024, cstep = (size_t)NBT * 4096;
    const unsigned lane_p = (unsigned)((2 * s0 + (rq >> 1)) * 256 + ((rq & 1) * 32 + 16 * cbu + c16) * 4);
    const float* gblk = G + ((size_t)d * T * NBT + bt) * 16 * 1024 + (size_t)wcol * 4096 + lane_p;
    const float* cblk = Csave + ((size_t)d * T * NBT + bt) * 4096 + (size_t)wcol * 1024 + lane_p;
    const int DH = D * H, D4H = D * 4 * H;
    const int row0 = bt * 32 + s0 * 16;
    const float* dyl = dY + (size_t)(row0 + 4 * rq) * DH + d * H + col;
    const int t_first = d ? 0 : T - 1, dt = d ? 1 : -1;

    f32x4 gt[4], ct, cp, dyv;
    float dhrec[4] = {0.f, 0.f, 0.f, 0.f}, dcarry[4] = {0.f, 0.f, 0.f, 0.f};
    float dbsum[4] = {0.f, 0.f, 0.f, 0.f};
    float runmax = 0.f;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    auto load_c = [&](int t) -> f32x4 {
        const int tt = (t >= 0 && t < T) ? t : t_first;
        return *reinterpret_cast<const f32x4*>(cblk + (size_t)tt * cstep);
    };
    auto load_step = [&](int t) {
        const float* gp = gblk + (size_t)t * gstep;
#pragma unroll
        for (int g = 0; g < 4; ++g) gt[g] = *reinterpret_cast<const f32x4*>(gp + g * 1024);
        cp = load_c(t + dt);
        const float* dp = dyl + (size_t)t * Bp * DH;
#pragma unroll
        for (int j = 0; j < 4; ++j) dyv[j] = dp[(size_t)j * DH];
    };
    ct = load_c(t_first);
    load_step(t_first);

    for (int step = 0; step < T; ++step) {
        const int t = t_first + dt * step;
        const bool cp_ok = (t + dt) >= 0 && (t + dt) < T;       // c of the step before the first one is zero
        float v[4][4];
        float m = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float ig = gt[0][j], fg = gt[1][j], gg = gt[2][j], og = gt[3][j];
            const float dh = dyv[j] + dhrec[j];
            const float tc = fast_tanh(ct[j]);
            const float dc = dcarry[j] + dh * og * (1.f - tc * tc);
            dcarry[j] = dc * fg;
            const float cpv = cp_ok ? cp[j] : 0.f;
            v[0][j] = dc * gg * ig * (1.f - ig);
            v[1][j] = dc * cpv * fg * (1.f - fg);
            v[2][j] = dc * ig * (1.f - gg * gg);
            v[3][j] = dh * tc * og * (1.f - og);
#pragma unroll
            for (int g = 0; g < 4; ++g) { dbsum[g] += v[g][j]; m = fmaxf(m, fabsf(v[g][j])); }
        }
        ct = cp;
        runmax = fmaxf(runmax, m);
        // the dP image (fp32): rows 4 rq + j, columns g * 128 + col
        float* dpw = dpf + 4 * rq * DPF_LD + col;
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g) dpw[j * DPF_LD + g * H] = v[g][j];
        m = wave_max(m);
        if (lane == 0) wmax[step & 1][w8] = m;
        __syncthreads();                                         // (1) the eight maxima; the previous step's dP image was stored
        float tm = wmax[step & 1][0];
#pragma unroll
        for (int i = 1; i < 8; ++i) tm = fmaxf(tm, wmax[step & 1][i]);
        const float sg = lob_split_scale(tm);
        const float r_hh = 1.f / (sg * sw), r_sm = r_hh * (1.f / S_LO);
        _Float16* dgw = dgs + 4 * rq * DGH_LD + col;
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                _Float16 hh, ll;
                split2(v[g][j], hh, ll, sg);
                dgw[j * DGH_LD + g * H] = hh;
                dgw[16 * DGH_LD + j * DGH_LD + g * H] = ll;
            }
        __syncthreads();                                         // (2) the split tile and the fp32 image are complete
        if (step + 1 < T) load_step(t + dt);
        f32x4 ahh = zero4, asm_ = zero4;
        const _Float16* arow = dgs + c16 * DGH_LD + 8 * rq;
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) {
            const f16x8 ah = *reinterpret_cast<const f16x8*>(arow + 32 * ks);
            const f16x8 al = *reinterpret_cast<const f16x8*>(arow + 16 * DGH_LD + 32 * ks);
            ahh = mfma16_f16(ah, whi[ks], ahh);
            asm_ = mfma16_f16(ah, wlo[ks], asm_);
            asm_ = mfma16_f16(al, whi[ks], asm_);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) dhrec[j] = ahh[j] * r_hh + asm_[j] * r_sm;
        // dP rows: 16 rows x 2 KB of fp32 (1 KB of bf16), 16 B per thread and pass
        {
            const int row = tid >> 5, c4 = (tid & 31) * 4;
            const size_t o = ((size_t)t * Bp + row0 + row) * D4H + d * 4 * H;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const f32x4 x = *reinterpret_cast<const f32x4*>(dpf + row * DPF_LD + c4 + 128 * i);
                if constexpr (DP_BF16) {
                    typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
                    bf16x4 b = {(__bf16)x[0], (__bf16)x[1], (__bf16)x[2], (__bf16)x[3]};
                    *reinterpret_cast<bf16x4*>(reinterpret_cast<__bf16*>(dPv) + o + c4 + 128 * i) = b;
                } else {
                    *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(dPv) + o + c4 + 128 * i) = x;
                }
            }
        }
        __syncthreads();                                         // (3) the image and the tile may be overwritten
    }
    if (dbias) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float x = dbsum[g];
            x += __shfl_xor(x, 16, 64);
            x += __shfl_xor(x, 32, 64);
            if (rq == 0) atomicAdd(dbias + (size_t)d * 4 * H + g * H + col, x);
        }
    }
    if (amax_out) {
        runmax = wave_max(runmax);
        if (lane == 0) atomicMax(reinterpret_cast<unsigned*>(amax_out), __builtin_bit_cast(unsigned, runmax));
    }
}

}  // namespace

int lob_rec_bwd_split(const float* G, const float* Csave, const float* Whh, const float* dY, void* dP, int dp_bf16, float* dbias,
                      float* amax_out, int T, int Bp, int D, const float* range, hipStream_t s) {
    const dim3 grid(Bp / 16, D), block(512);
    if (dp_bf16) hipLaunchKernelGGL((lstm_rec_bwd_h128_split_kernel<true>), grid, block, 0, s, G, Csave, Whh, dY, dP, dbias, amax_out, T, Bp, range);
    else         hipLaunchKernelGGL((lstm_rec_bwd_h128_split_kernel<false>), grid, block, 0, s, G, Csave, Whh, dY, dP, dbias, amax_out, T, Bp, range);
    LOB_CHECK_LAUNCH();
    return 0;
}

namespace {
}  // namespace

// Internal entry point used by lob_lstm_rec_fwd_f32 / lob_lstm_rec_fwd_f32_drop (lstm_rec_f32.hip): 16-row tiles, eight waves,
// grid Bp/16 x D.  Yd != NULL: the dropped copy of Y as well (saving launches only).
int lob_rec_fwd_split(float* P, const float* Whh, float* Y, float* Csave, int T, int Bp, int D, int save, const float* range,
                      float* Yd, float drop_p, uint64_t seed, hipStream_t s) {
    // part tiles while full tiles would occupy at most a quarter of the 256 CUs: four workgroups per 16-row tile, measured
    // -13 % (B = 512), -25 % (B = 256), -31 % (B = 32) per fp32 forward; at B = 1024 (128 tiles) two per tile change nothing and
    // four cost +24 % (two 8-wave workgroups per CU), so the switch-over sits at 64 tiles (tools/half_tile_ab.py)
    const int v = lob_variant(LOB_VAR_REC_HALF);
    const int tiles = (Bp / 16) * D;
    // LOB_VAR_REC_HALF: 1 = four workgroups per tile up to 64 tiles; 2 / 4 force that split (up to 128 tiles: A/B)
    const int parts = (v == 2 || v == 4) ? (tiles <= 128 ? v : 1) : ((v == 0 || tiles > 64) ? 1 : 4);
    const dim3 grid((Bp / 16) * parts, D), block(512);
    if (Yd && !save) return LOB_E_ARG;
#define LOB_RFS(SV, PT, DR) hipLaunchKernelGGL((lstm_rec_fwd_h128_split_kernel<SV, PT, DR>), grid, block, 0, s, P, Whh, Y, Csave, T, Bp, \
                                               range, Yd, drop_p, seed)
    if (Yd) {
        if (parts == 2) LOB_RFS(true, 2, true); else if (parts == 4) LOB_RFS(true, 4, true); else LOB_RFS(true, 1, true);
    } else if (parts == 2) {
        if (save) LOB_RFS(true, 2, false); else LOB_RFS(false, 2, false);
    } else if (parts == 4) {
        if (save) LOB_RFS(true, 4, false); else LOB_RFS(false, 4, false);
    } else {
        if (save) LOB_RFS(true, 1, false); else LOB_RFS(false, 1, false);
    }
#undef LOB_RFS
    LOB_CHECK_LAUNCH();
    return 0;
}
