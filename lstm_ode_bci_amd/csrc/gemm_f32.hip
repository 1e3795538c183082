// Exact-fp32 MFMA GEMMs for the dense layers and the input-side LSTM gate GEMM.
//
// C[M,N] = act(A[M,K] * W[N,K]^T + bias)            (gemm_nt: both operands K-contiguous)
//
// Tiling: 128x128 output tile per 256-thread workgroup (4 waves as 2x2, each wave a 2x2
// block of 32x32 v_mfma_f32_32x32x2_f32 accumulators = 64 VGPRs), K in steps of 32 staged
// through double-buffered LDS (2 workgroups per CU).
//
// k-permutation: inside each 32-wide K block, MFMA step s (0..15) contracts k = s (lane
// half 0) and k = 16 + s (lane half 1).  A dot product does not care about the order of
// its terms, and with this order every lane reads 16 CONTIGUOUS floats of its row from
// LDS (4 x ds_read_b128) instead of 16 scalar reads.  LDS rows are padded to 36 floats
// (144 B): for ds_read_b128 the 16-lane groups then hit 16 distinct 16-B slots.
#include <stdlib.h>
#include "lob_common.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 32, LDT = 36;

struct GemmNT {
    const float* A; const float* W; const float* bias; float* C;
    int lda, ldw, ldc, M, N, K, act, accumulate;
    // fragment epilogue (gate pre-activations): N = D*4H, M = T*Bp
    int T, Bp, H, D;
    // SPLIT kernels: device floats, an upper bound of |A| and of |W| (the operands' power-of-two pre-scales derive from them)
    const float* amax_a; const float* amax_w;
    // gemm_nt_split_kernel: C *= dropout mask of element (row * ldc + col) -- the backward of a dropout fused into the producer
    // of the layer below's output (lob_lstm_rec_fwd_f32_drop), as the bf16 NT kernels' epilogue
    float drop_p; uint64_t seed;
};

// ---- fp32 products on the 16-bit matrix pipe (round 4: the backward GEMMs of the fp32 path) ---------------------------
// As gate_gemm_ws_split.hip / lstm_rec_f32_split.hip: every fp32 operand x is carried as hi = fp16(s), lo = fp16((s - hi) 2^11),
// s = x 2^k with k from the tensor's |max| (lob_split_scale: both halves stay inside fp16's normal range for any finite
// operand); x y = r_hh hi hi + r_sm (hi lo + lo hi), the two groups in SEPARATE fp32 accumulators: three
// v_mfma_f32_32x32x16_f16 (96 cycles) per 16-deep k-step of a 32x32 block instead of eight v_mfma_f32_32x32x2_f32 (512).
// Here the split happens where the fragments are read: a lane's 16 fp32 values of one k-tile (the same 16 k for the A lane
// and the B lane of a lane half, in the same order: a dot product does not care which k a slot carries) become two hi
// and two lo fragments.  The kernels are then bound by that conversion (7 vector operations per element), not by the MFMAs.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
constexpr float SPLIT_LO = 2048.f;

__device__ __forceinline__ void split8(const f32x4& a, const f32x4& b, float scale, f16x8& hi, f16x8& lo) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float s0 = a[e] * scale, s1 = b[e] * scale;
        const _Float16 h0 = (_Float16)s0, h1 = (_Float16)s1;
        hi[e] = h0; hi[4 + e] = h1;
        lo[e] = (_Float16)((s0 - (float)h0) * SPLIT_LO);
        lo[4 + e] = (_Float16)((s1 - (float)h1) * SPLIT_LO);
    }
}
__device__ __forceinline__ f32x16 mfma_h(f16x8 a, f16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}

template <bool VEC>
__device__ __forceinline__ void load_tile(const float* __restrict__ G, int ld, int row0, int rows,
                                          int k0, int K, int tid, f32x4 (&r)[4]) {
    const int rr = tid >> 3, c4 = (tid & 7) * 4;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = row0 + rr + 32 * i;
        const int k = k0 + c4;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (row < rows) {
            const float* p = G + (size_t)row * ld + k;
            if (VEC) {
                if (k < K) v = *reinterpret_cast<const f32x4*>(p);
            } else {
                if (k + 0 < K) v[0] = p[0];
                if (k + 1 < K) v[1] = p[1];
                if (k + 2 < K) v[2] = p[2];
                if (k + 3 < K) v[3] = p[3];
            }
        }
        r[i] = v;
    }
}

__device__ __forceinline__ void store_tile(float* S, int tid, const f32x4 (&r)[4]) {
    const int rr = tid >> 3, c4 = (tid & 7) * 4;
#pragma unroll
    for (int i = 0; i < 4; ++i)
        *reinterpret_cast<f32x4*>(S + (rr + 32 * i) * LDT + c4) = r[i];
}

template <bool VEC, int EPI>
__global__ __launch_bounds__(256, 2) void gemm_nt_kernel(GemmNT g) {
    __shared__ __attribute__((aligned(16))) float lds[2 * 2 * BM * LDT];
    float* As = lds;                    // [2][BM][LDT]
    float* Ws = lds + 2 * BM * LDT;     // [2][BN][LDT]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    // XCD-aware tile order (see gemm_bf16.hip): one row panel's N tiles run back to back on one XCD
    const int ntn = (g.N + BN - 1) / BN, ntm = (g.M + BM - 1) / BM;
    const int xcd = blockIdx.x & 7, jb = blockIdx.x >> 3;
    const int mt = (jb / ntn) * 8 + xcd;
    if (mt >= ntm) return;
    const int m0 = mt * BM, n0 = (jb % ntn) * BN;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    f32x4 ra[4], rb[4];
    const int nk = (g.K + BK - 1) / BK;
    load_tile<VEC>(g.A, g.lda, m0, g.M, 0, g.K, tid, ra);
    load_tile<VEC>(g.W, g.ldw, n0, g.N, 0, g.K, tid, rb);
    store_tile(As, tid, ra);
    store_tile(Ws, tid, rb);
    __syncthreads();

    int buf = 0;
    for (int kt = 0; kt < nk; ++kt) {
        if (kt + 1 < nk) {
            load_tile<VEC>(g.A, g.lda, m0, g.M, (kt + 1) * BK, g.K, tid, ra);
            load_tile<VEC>(g.W, g.ldw, n0, g.N, (kt + 1) * BK, g.K, tid, rb);
        }
        const float* as = As + buf * BM * LDT + (64 * wr + (lane & 31)) * LDT + 16 * (lane >> 5);
        const float* ws = Ws + buf * BN * LDT + (64 * wc + (lane & 31)) * LDT + 16 * (lane >> 5);
        f32x4 af[2][4], bf[2][4];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                af[i][q] = *reinterpret_cast<const f32x4*>(as + i * 32 * LDT + 4 * q);
                bf[i][q] = *reinterpret_cast<const f32x4*>(ws + i * 32 * LDT + 4 * q);
            }
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = mfma32(af[i][q][e], bf[j][q][e], acc[i][j]);
        if (kt + 1 < nk) {
            store_tile(As + (buf ^ 1) * BM * LDT, tid, ra);
            store_tile(Ws + (buf ^ 1) * BN * LDT, tid, rb);
        }
        __syncthreads();
        buf ^= 1;
    }

    if (EPI == 0) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int col = n0 + 64 * wc + 32 * j + (lane & 31);
                if (col >= g.N) continue;
                const float bv = g.bias ? g.bias[col] : 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = m0 + 64 * wr + 32 * i + acc_row(r, lane);
                    if (row < g.M) {
                        float* dst = g.C + (size_t)row * g.ldc + col;
                        const float val = apply_act(acc[i][j][r] + bv, g.act);
                        *dst = g.accumulate ? *dst + val : val;
                    }
                }
            }
    } else {
        // accumulator-fragment order for the persistent recurrent kernel:
        // [D][T][Bp/32][H/32][4 gates][q=4][64 lanes][4]
        const int NBT = g.Bp >> 5, NW = g.H >> 5, H4 = 4 * g.H;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int mrow = m0 + 64 * wr + 32 * i;
            if (mrow >= g.M) continue;
            const int t = mrow / g.Bp, bt = (mrow % g.Bp) >> 5;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int ncol = n0 + 64 * wc + 32 * j;
                if (ncol >= g.N) continue;
                const int d = ncol / H4, gg = (ncol % H4) / g.H, w = (ncol % g.H) >> 5;
                const float bv = g.bias ? g.bias[ncol + (lane & 31)] : 0.f;
                float* dst = g.C + ((((size_t)(d * g.T + t) * NBT + bt) * NW + w) * 4 + gg) * 1024 + lane * 4;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    f32x4 v = {acc[i][j][4 * q + 0] + bv, acc[i][j][4 * q + 1] + bv,
                               acc[i][j][4 * q + 2] + bv, acc[i][j][4 * q + 3] + bv};
                    *reinterpret_cast<f32x4*>(dst + q * 256) = v;
                }
            }
        }
    }
}


// ------------------------------------------------------------------------------------------
// LDS-DMA variant of the exact-fp32 NT GEMM (K % 32 == 0, K >= 128).  Same 128x128 tile and MFMA
// schedule as gemm_nt_kernel, but the fp32 operand tiles go HBM -> LDS by global_load_lds_dwordx4:
// no staging VGPRs, no ds_write pass, and a 4-slot ring keeps THREE 32-deep k-tiles (5 us of MFMA
// work) in flight across output-tile boundaries, so a persistent workgroup never waits on memory at a
// tile seam.  Counted s_waitcnt vmcnt + raw s_barrier, as in gemm_bf16.hip.
//
// LDS image of a slot: [128 rows][32 floats] = 128-B rows, lane-linear (8 rows per DMA instruction).
// A lane's 16 contiguous k (4 x ds_read_b128) would be 8-way conflicted on unpadded rows, so 16-B
// chunk c of row r is stored at chunk slot c ^ ((r >> 1) & 7) -- on the DMA's per-lane source address
// and again on the fragment read; the 16 lanes of a b128 group then hit 16 distinct slots.
// ------------------------------------------------------------------------------------------
constexpr int FDS = 2, FTK = 32, FSLOT = 128 * FTK;     // floats per operand per slot (16 KB); 2 slots = 64 KB
// (2 slots -> 2 workgroups per CU: with the 64-cycle fp32 MFMA one k-tile is 1.7 us of work, enough to cover the
//  DMA of the next one, and the second resident workgroup covers the ds_read -> MFMA latency at each k-tile
//  start; 4 slots / 1 workgroup per CU measured 13 % slower than the register-staged kernel)

typedef __attribute__((address_space(3))) void lds_void_f;
typedef __attribute__((address_space(1))) const void gbl_cvoid_f;

__device__ __forceinline__ void dma_rows8_f32(const float* G, int ld, int row0, int nrows, int k0, float* lds_rows,
                                              int lane) {
    const int r = row0 + (lane >> 3), p = lane & 7;
    const int c = p ^ ((r >> 1) & 7);
    const int rr = r < nrows ? r : nrows - 1;                    // clamp: padded rows are never stored
    __builtin_amdgcn_global_load_lds((gbl_cvoid_f*)(G + (size_t)rr * ld + k0 + c * 4), (lds_void_f*)lds_rows, 16, 0, 0);
}

template <int EPI, bool SPLIT = false>
__global__ __launch_bounds__(256, 2) void gemm_nt_dma_f32_kernel(GemmNT g) {
    __shared__ __attribute__((aligned(1024))) float ring[FDS * 2 * FSLOT + 2048];     // 128 KB ring + 8 KB bias
    float* bias_s = ring + FDS * 2 * FSLOT;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    const int ntn = (g.N + BN - 1) / BN, ntm = (g.M + BM - 1) / BM;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, nslot = gridDim.x >> 3;
    const int panels = (ntm - xcd + 7) / 8, ntile = panels * ntn;
    const int nk = g.K / FTK;
    if (slot >= ntile) return;
    const int total = ((ntile - slot + nslot - 1) / nslot) * nk;
    for (int i = tid; i < g.N && i < 2048; i += 256) bias_s[i] = g.bias ? g.bias[i] : 0.f;
    __syncthreads();

    int p_q = 0, p_it = slot, p_kt = 0;
    auto issue = [&]() {
        const int m0 = ((p_it / ntn) * 8 + xcd) * BM, n0 = (p_it % ntn) * BN;
        float* as = ring + (p_q % FDS) * 2 * FSLOT;
        float* ws = as + FSLOT;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int rb = (wave * 4 + j) * 8;
            dma_rows8_f32(g.A, g.lda, m0 + rb, g.M, p_kt * FTK, as + rb * FTK, lane);
            dma_rows8_f32(g.W, g.ldw, n0 + rb, g.N, p_kt * FTK, ws + rb * FTK, lane);
        }
        ++p_q;
        if (++p_kt == nk) { p_kt = 0; p_it += nslot; }
    };
#pragma unroll 1
    for (int i = 0; i < FDS - 1 && i < total; ++i) issue();

    int it = slot, kt = 0, since_epi = 99;
    f32x16 acc[2][2];
    f32x16 asm_[SPLIT ? 2 : 1][SPLIT ? 2 : 1];       // SPLIT: the small terms (hi lo + lo hi); acc holds hi hi
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc[i][j][r] = 0.f; if (SPLIT) asm_[i][j][r] = 0.f; }
    float sa = 1.f, sw_ = 1.f, r_hh = 1.f, r_sm = 0.f;
    if constexpr (SPLIT) {
        sa = lob_split_scale(*g.amax_a); sw_ = lob_split_scale(*g.amax_w);
        r_hh = 1.f / (sa * sw_); r_sm = r_hh * (1.f / SPLIT_LO);
    }

    for (int q = 0; q < total; ++q) {
        // younger operations of this wave that may stay in flight: (FDS-2) k-tiles x 8 DMAs, plus the 16
        // fragment-epilogue stores if they were issued after DMA(q)
        constexpr int VM_STEADY = (FDS - 2) * 8, VM_EPI = VM_STEADY + 16;
        if (q + FDS - 1 > total)                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (EPI == 1 && since_epi < FDS - 1)   asm volatile("s_waitcnt vmcnt(%0)" ::"n"(VM_EPI) : "memory");
        else if (since_epi < FDS - 1)               asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else                                        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(VM_STEADY) : "memory");
        __builtin_amdgcn_s_barrier();
        if (p_q < total) issue();
        ++since_epi;

        const float* as = ring + (q % FDS) * 2 * FSLOT;
        const float* ws = as + FSLOT;
        const int r31 = lane & 31, hi = lane >> 5, sw = (r31 >> 1) & 7;      // (64wr + 32i) >> 1 is a multiple of 8
        f32x4 af[2][4], bf[2][4];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int qq = 0; qq < 4; ++qq) {
                const int pc = ((4 * hi + qq) ^ sw) * 4;
                af[i][qq] = *reinterpret_cast<const f32x4*>(as + (64 * wr + 32 * i + r31) * FTK + pc);
                bf[i][qq] = *reinterpret_cast<const f32x4*>(ws + (64 * wc + 32 * i + r31) * FTK + pc);
            }
        if constexpr (SPLIT) {
            f16x8 ah[2][2], al[2][2], bh[2][2], bl[2][2];      // [block][k-step]
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    split8(af[i][2 * ks], af[i][2 * ks + 1], sa, ah[i][ks], al[i][ks]);
                    split8(bf[i][2 * ks], bf[i][2 * ks + 1], sw_, bh[i][ks], bl[i][ks]);
                }
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        acc[i][j] = mfma_h(ah[i][ks], bh[j][ks], acc[i][j]);
                        asm_[i][j] = mfma_h(ah[i][ks], bl[j][ks], asm_[i][j]);
                        asm_[i][j] = mfma_h(al[i][ks], bh[j][ks], asm_[i][j]);
                    }
        } else {
#pragma unroll
        for (int qq = 0; qq < 4; ++qq)
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = mfma32(af[i][qq][e], bf[j][qq][e], acc[i][j]);
        }
        if (++kt < nk) continue;
        if constexpr (SPLIT) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) { acc[i][j][r] = acc[i][j][r] * r_hh + asm_[i][j][r] * r_sm; asm_[i][j][r] = 0.f; }
        }

        kt = 0;
        const int cm0 = ((it / ntn) * 8 + xcd) * BM, cn0 = (it % ntn) * BN;
        it += nslot;
        since_epi = 0;
        if (EPI == 0) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int col = cn0 + 64 * wc + 32 * j + (lane & 31);
                    const float bv = SPLIT ? 0.f : lds_read_f32_opaque(bias_s + (col < 2048 ? col : 0));
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = cm0 + 64 * wr + 32 * i + acc_row(r, lane);
                        // (SPLIT: no bias, no activation -- the generic epilogue's transcendentals would be a quarter of this
                        // kernel's vector work)
                        const float val = SPLIT ? acc[i][j][r] : apply_act(acc[i][j][r] + bv, g.act);
                        if (row < g.M && col < g.N) g.C[(size_t)row * g.ldc + col] = val;
                        acc[i][j][r] = 0.f;
                    }
                }
        } else {
            const int NBT = g.Bp >> 5, NW = g.H >> 5, H4 = 4 * g.H;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int mrow = cm0 + 64 * wr + 32 * i;
                const int t = mrow / g.Bp, bt = (mrow % g.Bp) >> 5;
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int ncol = cn0 + 64 * wc + 32 * j;
                    const int d = ncol / H4, gg = (ncol % H4) / g.H, w = (ncol % g.H) >> 5;
                    const float bv = lds_read_f32_opaque(bias_s + ncol + (lane & 31));
                    float* dst = g.C + ((((size_t)(d * g.T + t) * NBT + bt) * NW + w) * 4 + gg) * 1024 + lane * 4;
#pragma unroll
                    for (int qq = 0; qq < 4; ++qq) {
                        f32x4 v = {acc[i][j][4 * qq + 0] + bv, acc[i][j][4 * qq + 1] + bv,
                                   acc[i][j][4 * qq + 2] + bv, acc[i][j][4 * qq + 3] + bv};
                        *reinterpret_cast<f32x4*>(dst + qq * 256) = v;
                    }
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
                }
            }
        }
    }
}


// ------------------------------------------------------------------------------------
// C[M,N] += A[Kc,M]^T * B[Kc,N]  -- weight gradients.  The contraction index is the ROW index
// of both operands (time x batch, ~1M), the output is small, so the grid splits Kc and every
// workgroup adds its 128x128 partial tile with fp32 atomics (each wave-instruction covers two
// 128-B row segments = the full-rate shape).  Operands are staged [k][m] so that an MFMA
// operand read is one conflict-free ds_read_b32 (consecutive lanes, consecutive addresses).
// ------------------------------------------------------------------------------------
constexpr int TLD = 132;   // LDS row stride (floats) of a [32 k][128 m] tile

struct GemmTN {
    const float* A; const float* B; float* C;
    int lda, ldb, ldc, M, N, Kc, kchunk;
    const float* amax_a; const float* amax_b;      // SPLIT kernels (see GemmNT)
};

template <bool VEC>
__device__ __forceinline__ void load_tile_k(const float* __restrict__ G, int ld, int k0, int kend,
                                            int c0, int cols, int tid, f32x4 (&r)[4]) {
    // tile [32 k][128 cols]; thread -> (k = tid/32 + 8i, c4 = (tid%32)*4)
    const int kk = tid >> 5, c4 = (tid & 31) * 4;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int k = k0 + kk + 8 * i, c = c0 + c4;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (k < kend) {
            const float* p = G + (size_t)k * ld + c;
            if (VEC) {
                if (c < cols) v = *reinterpret_cast<const f32x4*>(p);
            } else {
                if (c + 0 < cols) v[0] = p[0];
                if (c + 1 < cols) v[1] = p[1];
                if (c + 2 < cols) v[2] = p[2];
                if (c + 3 < cols) v[3] = p[3];
            }
        }
        r[i] = v;
    }
}

__device__ __forceinline__ void store_tile_k(float* S, int tid, const f32x4 (&r)[4]) {
    const int kk = tid >> 5, c4 = (tid & 31) * 4;
#pragma unroll
    for (int i = 0; i < 4; ++i) *reinterpret_cast<f32x4*>(S + (kk + 8 * i) * TLD + c4) = r[i];
}

template <bool VEC, bool SPLIT = false>
__global__ __launch_bounds__(256, 2) void gemm_tn_kernel(GemmTN g) {
    __shared__ __attribute__((aligned(16))) float lds[2 * 2 * 32 * TLD];
    float* As = lds;
    float* Bs = lds + 2 * 32 * TLD;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int ntm = (g.M + 127) / 128, ntn = (g.N + 127) / 128;
    const int tile = blockIdx.x % (ntm * ntn), chunk = blockIdx.x / (ntm * ntn);
    const int m0 = (tile / ntn) * 128, n0 = (tile % ntn) * 128;
    const int kbeg = chunk * g.kchunk, kend = min(g.Kc, kbeg + g.kchunk);

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    f32x16 asm_[SPLIT ? 2 : 1][SPLIT ? 2 : 1];
    float sa = 1.f, sb = 1.f;
    if constexpr (SPLIT) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) asm_[i][j][r] = 0.f;
        sa = lob_split_scale(*g.amax_a); sb = lob_split_scale(*g.amax_b);
    }
    f32x4 ra[4], rb[4];
    load_tile_k<VEC>(g.A, g.lda, kbeg, kend, m0, g.M, tid, ra);
    load_tile_k<VEC>(g.B, g.ldb, kbeg, kend, n0, g.N, tid, rb);
    store_tile_k(As, tid, ra);
    store_tile_k(Bs, tid, rb);
    __syncthreads();
    int buf = 0;
    for (int k0 = kbeg; k0 < kend; k0 += 32) {
        if (k0 + 32 < kend) {
            load_tile_k<VEC>(g.A, g.lda, k0 + 32, kend, m0, g.M, tid, ra);
            load_tile_k<VEC>(g.B, g.ldb, k0 + 32, kend, n0, g.N, tid, rb);
        }
        const float* as = As + buf * 32 * TLD + (16 * (lane >> 5)) * TLD + 64 * wr + (lane & 31);
        const float* bs = Bs + buf * 32 * TLD + (16 * (lane >> 5)) * TLD + 64 * wc + (lane & 31);
        if constexpr (SPLIT) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                f16x8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    f32x4 x0, x1, y0, y1;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        x0[e] = as[(8 * ks + e) * TLD + 32 * i]; x1[e] = as[(8 * ks + 4 + e) * TLD + 32 * i];
                        y0[e] = bs[(8 * ks + e) * TLD + 32 * i]; y1[e] = bs[(8 * ks + 4 + e) * TLD + 32 * i];
                    }
                    split8(x0, x1, sa, ah[i], al[i]);
                    split8(y0, y1, sb, bh[i], bl[i]);
                }
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        acc[i][j] = mfma_h(ah[i], bh[j], acc[i][j]);
                        asm_[i][j] = mfma_h(ah[i], bl[j], asm_[i][j]);
                        asm_[i][j] = mfma_h(al[i], bh[j], asm_[i][j]);
                    }
            }
        } else {
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const float a0 = as[s * TLD], a1 = as[s * TLD + 32];
            const float b0 = bs[s * TLD], b1 = bs[s * TLD + 32];
            acc[0][0] = mfma32(a0, b0, acc[0][0]);
            acc[0][1] = mfma32(a0, b1, acc[0][1]);
            acc[1][0] = mfma32(a1, b0, acc[1][0]);
            acc[1][1] = mfma32(a1, b1, acc[1][1]);
        }
        }
        if (k0 + 32 < kend) {
            store_tile_k(As + (buf ^ 1) * 32 * TLD, tid, ra);
            store_tile_k(Bs + (buf ^ 1) * 32 * TLD, tid, rb);
        }
        __syncthreads();
        buf ^= 1;
    }
    if constexpr (SPLIT) {
        const float r_hh = 1.f / (sa * sb), r_sm = r_hh * (1.f / SPLIT_LO);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = acc[i][j][r] * r_hh + asm_[i][j][r] * r_sm;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = n0 + 64 * wc + 32 * j + (lane & 31);
            if (col >= g.N) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + 64 * wr + 32 * i + acc_row(r, lane);
                if (row < g.M) atomicAdd(g.C + (size_t)row * g.ldc + col, acc[i][j][r]);
            }
        }
}

// ------------------------------------------------------------------------------------
// The same product, fp16-split, with the split done ONCE per element while the tile is staged (the structure of
// gemm_tn_bf16_kernel in gemm_bf16.hip, which rounds fp32 sources to bf16 on their way into LDS): every [32 k][128 cols]
// source tile becomes a hi and a lo fp16 image ([k][col], 320-B rows), the MFMA fragments (8 consecutive k of one column per
// lane) come out of gfx950's transposing LDS read ds_read_b64_tr_b16, three v_mfma_f32_32x32x16_f16 per fragment pair.
// Against gemm_tn_kernel<.., SPLIT = true> (split at every fragment read: each element converted by two waves): half the
// conversions, 16-B LDS stores instead of 64 scalar fragment reads per k-tile and lane.
// ------------------------------------------------------------------------------------
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef __fp16 fp16x4v __attribute__((__vector_size__(4 * sizeof(__fp16))));      // the transposing read's builtin type
typedef __attribute__((address_space(3))) fp16x4v lds_f16x4;
constexpr int STK = 32, SLDK = 160;          // contraction rows per stage; LDS row stride in fp16 elements (320 B)

__device__ __forceinline__ void ldk_f32s(const float* __restrict__ G, int ld, int k0, int kend, int c0, int cols, int tid,
                                         f32x4 (&r)[4]) {
    const int kk = tid >> 5, c4 = (tid & 31) * 4;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int k = k0 + kk + 8 * i, c = c0 + c4;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (k < kend && c < cols) v = *reinterpret_cast<const f32x4*>(G + (size_t)k * ld + c);
        r[i] = v;
    }
}
__device__ __forceinline__ void stk_split(_Float16* Shi, _Float16* Slo, int tid, const f32x4 (&r)[4], float scale) {
    const int kk = tid >> 5, c4 = (tid & 31) * 4;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        f16x4 h, l;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float sv = r[i][e] * scale;
            const _Float16 hh = (_Float16)sv;
            h[e] = hh;
            l[e] = (_Float16)((sv - (float)hh) * SPLIT_LO);
        }
        *reinterpret_cast<f16x4*>(Shi + (kk + 8 * i) * SLDK + c4) = h;
        *reinterpret_cast<f16x4*>(Slo + (kk + 8 * i) * SLDK + c4) = l;
    }
}
// fragment of the 32-column block starting at column `cb`, k-step s (16 rows) of a [k][col] fp16 LDS image
__device__ __forceinline__ f16x8 tr_frag_h(const _Float16* S, int cb, int s, int lane) {
    const int h = lane >> 5, mh = (lane >> 4) & 1, q = (lane >> 2) & 3, p = lane & 3;
    const _Float16* a = S + (16 * s + 8 * h + q) * SLDK + cb + 16 * mh + 4 * p;
    const f16x4 lo = __builtin_bit_cast(f16x4, __builtin_amdgcn_ds_read_tr16_b64_v4f16((lds_f16x4*)a));
    const f16x4 hi = __builtin_bit_cast(f16x4, __builtin_amdgcn_ds_read_tr16_b64_v4f16((lds_f16x4*)(a + 4 * SLDK)));
    f16x8 f = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return f;
}

struct GemmTNS {
    const float* A; const float* B; float* C;
    int lda, ldb, ldc, M, N, Kc, kchunk, tiles;
    const float* amax_a; const float* amax_b;
};

// One row piece (8 k-rows apart) of the staging: load / split + store.  Pieces let the pipelined kernel below spread the
// conversion of a tile between the MFMA groups of the previous one.
__device__ __forceinline__ f32x4 ldk_piece(const float* __restrict__ G, int ld, int k0, int c0, int tid, int i) {
    return *reinterpret_cast<const f32x4*>(G + (size_t)(k0 + (tid >> 5) + 8 * i) * ld + c0 + (tid & 31) * 4);
}
__device__ __forceinline__ void stk_piece(_Float16* Shi, _Float16* Slo, int tid, const f32x4& r, float scale, int i) {
    const int kk = tid >> 5, c4 = (tid & 31) * 4;
    f16x4 h, l;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float sv = r[e] * scale;
        const _Float16 hh = (_Float16)sv;
        h[e] = hh;
        l[e] = (_Float16)((sv - (float)hh) * SPLIT_LO);
    }
    *reinterpret_cast<f16x4*>(Shi + (kk + 8 * i) * SLDK + c4) = h;
    *reinterpret_cast<f16x4*>(Slo + (kk + 8 * i) * SLDK + c4) = l;
}

// FAST (round 4, second half): M, N multiples of 128 and every contraction chunk a whole, even number of 32-row tiles.
// The general kernel (FAST = false) guards each of its 8 loads with a branch, requests a tile only ONE step ahead -- its
// 24 MFMAs (0.4 us) do not cover an HBM round trip -- and converts after the MFMAs, behind a `if (more)` the scheduler
// cannot move code across: the matrix pipe sat idle ~70 % of a step.  Here tile t + 2 is requested while tile t is
// multiplied (two register sets), the loads are unguarded, and the split + LDS store of tile t + 1 is written piece by piece
// BETWEEN the MFMA groups of tile t in one basic block.  Same products, same accumulation order: bit-identical results.
template <bool FAST>
__global__ __launch_bounds__(256, 2) void gemm_tn_split_kernel(GemmTNS g) {
    __shared__ __attribute__((aligned(16))) _Float16 lds[2 * 4 * STK * SLDK];      // [buf][A hi, A lo, B hi, B lo][32][160]: 80 KB
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int ntn = (g.N + 127) / 128;
    const int xcd = blockIdx.x & 7, rest = blockIdx.x >> 3;
    const int tile = rest % g.tiles, chunk = (rest / g.tiles) * 8 + xcd;
    const int m0 = (tile / ntn) * 128, n0 = (tile % ntn) * 128;
    const int kbeg = chunk * g.kchunk, kend = min(g.Kc, kbeg + g.kchunk);
    if (kbeg >= kend) return;
    const float sa = lob_split_scale(*g.amax_a), sb = lob_split_scale(*g.amax_b);

    f32x16 acc[2][2], asm_[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc[i][j][r] = 0.f; asm_[i][j][r] = 0.f; }

    if constexpr (FAST) {
        // ONE register set: piece q of the next tile is split + stored after MFMA triple q of this tile, and the same
        // registers are re-requested at once for the tile after next (two sets spilled next to the 128 accumulators)
        f32x4 ra[4], rb[4];
        auto ldp = [&](int k0, int q) {
            k0 = k0 < kend ? k0 : kend - STK;              // past the end: the last tile again (L2 hit, never multiplied)
            if (q < 4) ra[q] = ldk_piece(g.A, g.lda, k0, m0, tid, q);
            else       rb[q - 4] = ldk_piece(g.B, g.ldb, k0, n0, tid, q - 4);
        };
        auto stp = [&](_Float16* nb, int q) {
            if (q < 4) stk_piece(nb, nb + STK * SLDK, tid, ra[q], sa, q);
            else       stk_piece(nb + 2 * STK * SLDK, nb + 3 * STK * SLDK, tid, rb[q - 4], sb, q - 4);
        };
        // multiply the tile in `buf` (contraction rows k0 ..); the registers hold tile k0 + STK, refilled with k0 + 2 STK
        auto step = [&](int buf, int k0) {
            const _Float16* ah_s = lds + buf * 4 * STK * SLDK;
            const _Float16* al_s = ah_s + STK * SLDK;
            const _Float16* bh_s = ah_s + 2 * STK * SLDK;
            const _Float16* bl_s = ah_s + 3 * STK * SLDK;
            _Float16* nb = lds + (buf ^ 1) * 4 * STK * SLDK;
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                f16x8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    ah[i] = tr_frag_h(ah_s, 64 * wr + 32 * i, s2, lane);
                    al[i] = tr_frag_h(al_s, 64 * wr + 32 * i, s2, lane);
                    bh[i] = tr_frag_h(bh_s, 64 * wc + 32 * i, s2, lane);
                    bl[i] = tr_frag_h(bl_s, 64 * wc + 32 * i, s2, lane);
                }
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        acc[i][j] = mfma_h(ah[i], bh[j], acc[i][j]);
                        asm_[i][j] = mfma_h(ah[i], bl[j], asm_[i][j]);
                        asm_[i][j] = mfma_h(al[i], bh[j], asm_[i][j]);
                        const int q = 4 * s2 + 2 * i + j;
                        stp(nb, q);
                        ldp(k0 + 2 * STK, q);
                    }
            }
        };
#pragma unroll
        for (int q = 0; q < 8; ++q) ldp(kbeg, q);
#pragma unroll
        for (int q = 0; q < 8; ++q) { stp(lds, q); ldp(kbeg + STK, q); }
        __syncthreads();
        for (int k0 = kbeg; k0 < kend; k0 += 2 * STK) {
            step(0, k0);
            __syncthreads();
            step(1, k0 + STK);
            __syncthreads();
        }
    } else {
    f32x4 ra[4], rb[4];
    auto load = [&](int k0) {
        ldk_f32s(g.A, g.lda, k0, kend, m0, g.M, tid, ra);
        ldk_f32s(g.B, g.ldb, k0, kend, n0, g.N, tid, rb);
    };
    auto store = [&](int b) {
        _Float16* base = lds + b * 4 * STK * SLDK;
        stk_split(base, base + STK * SLDK, tid, ra, sa);
        stk_split(base + 2 * STK * SLDK, base + 3 * STK * SLDK, tid, rb, sb);
    };
    load(kbeg);
    store(0);
    __syncthreads();
    int buf = 0;
    for (int k0 = kbeg; k0 < kend; k0 += STK) {
        const bool more = k0 + STK < kend;
        if (more) load(k0 + STK);
        const _Float16* ah_s = lds + buf * 4 * STK * SLDK;
        const _Float16* al_s = ah_s + STK * SLDK;
        const _Float16* bh_s = ah_s + 2 * STK * SLDK;
        const _Float16* bl_s = ah_s + 3 * STK * SLDK;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            f16x8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                ah[i] = tr_frag_h(ah_s, 64 * wr + 32 * i, s, lane);
                al[i] = tr_frag_h(al_s, 64 * wr + 32 * i, s, lane);
                bh[i] = tr_frag_h(bh_s, 64 * wc + 32 * i, s, lane);
                bl[i] = tr_frag_h(bl_s, 64 * wc + 32 * i, s, lane);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    acc[i][j] = mfma_h(ah[i], bh[j], acc[i][j]);
                    asm_[i][j] = mfma_h(ah[i], bl[j], asm_[i][j]);
                    asm_[i][j] = mfma_h(al[i], bh[j], asm_[i][j]);
                }
        }
        if (more) store(buf ^ 1);
        __syncthreads();
        buf ^= 1;
    }
    }
    const float r_hh = 1.f / (sa * sb), r_sm = r_hh * (1.f / SPLIT_LO);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = n0 + 64 * wc + 32 * j + (lane & 31);
            if (col >= g.N) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + 64 * wr + 32 * i + acc_row(r, lane);
                if (row < g.M) atomicAdd(g.C + (size_t)row * g.ldc + col, acc[i][j][r] * r_hh + asm_[i][j][r] * r_sm);
            }
        }
}

// ------------------------------------------------------------------------------------
// C[M,N] = A[M,K] W[N,K]^T, fp16-split, the split done once per element while the tiles are staged (the structure of
// gemm_nt_bf16_kernel: persistent workgroups, register-staged 128 x 32 operand tiles, the next tile's first loads issued
// before the epilogue's stores; XCD-aware tile order).  hi and lo images [row][k] with 80-B rows (conflict-free
// ds_read_b128 fragments).  Row-major fp32 output, no bias / activation: dX = dP W_ih of the fp32 training step.
// ------------------------------------------------------------------------------------
constexpr int NKT = 32, NLD = NKT + 8;

__device__ __forceinline__ void ld_rows_s(const float* __restrict__ G, int ld, int row0, int rows, int k0, int K, int tid,
                                          f32x4 (&r)[4]) {
    const int rr = tid >> 3, c4 = (tid & 7) * 4;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = row0 + rr + 32 * i, k = k0 + c4;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (row < rows && k < K) v = *reinterpret_cast<const f32x4*>(G + (size_t)row * ld + k);
        r[i] = v;
    }
}
__device__ __forceinline__ void st_rows_s(_Float16* Shi, _Float16* Slo, int tid, const f32x4 (&r)[4], float scale) {
    const int rr = tid >> 3, c4 = (tid & 7) * 4;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        f16x4 h, l;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float sv = r[i][e] * scale;
            const _Float16 hh = (_Float16)sv;
            h[e] = hh;
            l[e] = (_Float16)((sv - (float)hh) * SPLIT_LO);
        }
        *reinterpret_cast<f16x4*>(Shi + (rr + 32 * i) * NLD + c4) = h;
        *reinterpret_cast<f16x4*>(Slo + (rr + 32 * i) * NLD + c4) = l;
    }
}

// FAST (round 4, as gemm_tn_split_kernel<true>): M, N multiples of 128, K a multiple of 64: unguarded loads two k-tiles
// ahead (two register sets), the split + LDS store of k-tile q + 1 written piece by piece between the MFMA groups of
// k-tile q.  Bit-identical to the general kernel.
template <bool FAST>
__global__ __launch_bounds__(256, 2) void gemm_nt_split_kernel(GemmNT g) {
    __shared__ __attribute__((aligned(16))) _Float16 lds[2 * 4 * 128 * NLD];       // [buf][A hi, A lo, W hi, W lo][128][40]: 80 KB
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int ntn = (g.N + BN - 1) / BN, ntm = (g.M + BM - 1) / BM;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, nslot = gridDim.x >> 3;
    const int panels = (ntm - xcd + 7) / 8, ntile = panels * ntn;
    const int nk = (g.K + NKT - 1) / NKT;
    const float sa = lob_split_scale(*g.amax_a), sw_ = lob_split_scale(*g.amax_w);
    const float r_hh = 1.f / (sa * sw_), r_sm = r_hh * (1.f / SPLIT_LO);

    f32x4 ra[1][4], rw[1][4];
    auto load = [&](int m0, int n0, int k0) {
        ld_rows_s(g.A, g.lda, m0, g.M, k0, g.K, tid, ra[0]);
        ld_rows_s(g.W, g.ldw, n0, g.N, k0, g.K, tid, rw[0]);
    };
    auto store = [&](int b) {
        _Float16* base = lds + b * 4 * 128 * NLD;
        st_rows_s(base, base + 128 * NLD, tid, ra[0], sa);
        st_rows_s(base + 2 * 128 * NLD, base + 3 * 128 * NLD, tid, rw[0], sw_);
    };
    // FAST: one register set, piece-wise (see gemm_tn_split_kernel<true>); unguarded; a k-tile past the end is the last
    // one again (L2 hit, never multiplied)
    auto ldp = [&](int m0, int n0, int k0, int q) {
        k0 = k0 < g.K ? k0 : g.K - NKT;
        const int rr = tid >> 3, c4 = (tid & 7) * 4;
        if (q < 4) ra[0][q] = *reinterpret_cast<const f32x4*>(g.A + (size_t)(m0 + rr + 32 * q) * g.lda + k0 + c4);
        else       rw[0][q - 4] = *reinterpret_cast<const f32x4*>(g.W + (size_t)(n0 + rr + 32 * (q - 4)) * g.ldw + k0 + c4);
    };
    auto stp = [&](_Float16* nb, int q) {
        const int rr = tid >> 3, c4 = (tid & 7) * 4, i = q & 3;
        const f32x4& r = q < 4 ? ra[0][i] : rw[0][i];
        const float scale = q < 4 ? sa : sw_;
        _Float16* Shi = nb + (q < 4 ? 0 : 2 * 128 * NLD);
        f16x4 h, l;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float sv = r[e] * scale;
            const _Float16 hh = (_Float16)sv;
            h[e] = hh;
            l[e] = (_Float16)((sv - (float)hh) * SPLIT_LO);
        }
        *reinterpret_cast<f16x4*>(Shi + (rr + 32 * i) * NLD + c4) = h;
        *reinterpret_cast<f16x4*>(Shi + 128 * NLD + (rr + 32 * i) * NLD + c4) = l;
    };
    int it = slot;
    if (it >= ntile) return;
    int m0 = ((it / ntn) * 8 + xcd) * BM, n0 = (it % ntn) * BN;
    if constexpr (FAST) {
#pragma unroll
        for (int q = 0; q < 8; ++q) ldp(m0, n0, 0, q);
    } else load(m0, n0, 0);
    while (true) {
        f32x16 acc[2][2], asm_[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) { acc[i][j][r] = 0.f; asm_[i][j][r] = 0.f; }
        if constexpr (FAST) {
            // multiply the k-tile in `buf` (kt); the registers hold k-tile kt + 1, refilled with kt + 2
            auto step = [&](int buf, int kt) {
                const _Float16* base = lds + buf * 4 * 128 * NLD;
                _Float16* nb = lds + (buf ^ 1) * 4 * 128 * NLD;
                const _Float16* ap = base + (64 * wr + (lane & 31)) * NLD + 8 * (lane >> 5);
                const _Float16* bp = base + 2 * 128 * NLD + (64 * wc + (lane & 31)) * NLD + 8 * (lane >> 5);
#pragma unroll
                for (int ks = 0; ks < NKT / 16; ++ks) {
                    f16x8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        ah[i] = *reinterpret_cast<const f16x8*>(ap + 32 * i * NLD + 16 * ks);
                        al[i] = *reinterpret_cast<const f16x8*>(ap + 128 * NLD + 32 * i * NLD + 16 * ks);
                        bh[i] = *reinterpret_cast<const f16x8*>(bp + 32 * i * NLD + 16 * ks);
                        bl[i] = *reinterpret_cast<const f16x8*>(bp + 128 * NLD + 32 * i * NLD + 16 * ks);
                    }
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j) {
                            acc[i][j] = mfma_h(ah[i], bh[j], acc[i][j]);
                            asm_[i][j] = mfma_h(ah[i], bl[j], asm_[i][j]);
                            asm_[i][j] = mfma_h(al[i], bh[j], asm_[i][j]);
                            const int q = 4 * ks + 2 * i + j;
                            stp(nb, q);
                            ldp(m0, n0, (kt + 2) * NKT, q);
                        }
                }
            };
#pragma unroll
            for (int q = 0; q < 8; ++q) { stp(lds, q); ldp(m0, n0, NKT, q); }
            __syncthreads();
            for (int kt = 0; kt < nk; kt += 2) {
                step(0, kt);
                __syncthreads();
                step(1, kt + 1);
                __syncthreads();
            }
        } else {
        store(0);
        __syncthreads();
        int buf = 0;
        for (int kt = 0; kt < nk; ++kt) {
            if (kt + 1 < nk) load(m0, n0, (kt + 1) * NKT);
            const _Float16* base = lds + buf * 4 * 128 * NLD;
            const _Float16* ap = base + (64 * wr + (lane & 31)) * NLD + 8 * (lane >> 5);
            const _Float16* bp = base + 2 * 128 * NLD + (64 * wc + (lane & 31)) * NLD + 8 * (lane >> 5);
#pragma unroll
            for (int ks = 0; ks < NKT / 16; ++ks) {
                f16x8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    ah[i] = *reinterpret_cast<const f16x8*>(ap + 32 * i * NLD + 16 * ks);
                    al[i] = *reinterpret_cast<const f16x8*>(ap + 128 * NLD + 32 * i * NLD + 16 * ks);
                    bh[i] = *reinterpret_cast<const f16x8*>(bp + 32 * i * NLD + 16 * ks);
                    bl[i] = *reinterpret_cast<const f16x8*>(bp + 128 * NLD + 32 * i * NLD + 16 * ks);
                }
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        acc[i][j] = mfma_h(ah[i], bh[j], acc[i][j]);
                        asm_[i][j] = mfma_h(ah[i], bl[j], asm_[i][j]);
                        asm_[i][j] = mfma_h(al[i], bh[j], asm_[i][j]);
                    }
            }
            if (kt + 1 < nk) store(buf ^ 1);
            __syncthreads();
            buf ^= 1;
        }
        }
        const int nit = it + nslot;
        const bool more = nit < ntile;
        const int cm0 = m0, cn0 = n0;
        if (more) {                                   // the next tile's first operand tiles are requested before this tile's stores
            m0 = ((nit / ntn) * 8 + xcd) * BM; n0 = (nit % ntn) * BN;
            if constexpr (FAST) {
#pragma unroll
                for (int q = 0; q < 8; ++q) ldp(m0, n0, 0, q);
            } else load(m0, n0, 0);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int col = cn0 + 64 * wc + 32 * j + (lane & 31);
                if (col >= g.N) continue;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = cm0 + 64 * wr + 32 * i + acc_row(r, lane);
                    if (row < g.M) {
                        float val = acc[i][j][r] * r_hh + asm_[i][j][r] * r_sm;
                        if (g.drop_p > 0.f) val *= lob_dropout_scale(g.seed, (uint64_t)row * g.ldc + col, g.drop_p);
                        g.C[(size_t)row * g.ldc + col] = val;
                    }
                }
            }
        if (!more) break;
        it = nit;
    }
}

// out[n] += sum_m A[m][n]  (bias gradients).  Block = 256 threads = 64 column-lanes x 4 row-groups.
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ A, int lda, int M, int N,
                                                     int rows_per_block, float* __restrict__ out) {
    __shared__ float red[4][64];
    const int cl = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const int ncb = (N + 63) / 64;
    const int cb = blockIdx.x % ncb, rb = blockIdx.x / ncb;
    const int col = cb * 64 + cl;
    const int r0 = rb * rows_per_block, r1 = min(M, r0 + rows_per_block);
    float s = 0.f;
    if (col < N) {
        // four independent loads in flight per thread (a dependent chain of 256 L2/HBM round trips made the three
        // classifier-bias column sums of a training step cost 49 us each)
        float s1 = 0.f, s2 = 0.f, s3 = 0.f;
        int r = r0 + rg;
        for (; r + 12 < r1; r += 16) {
            s += A[(size_t)r * lda + col];
            s1 += A[(size_t)(r + 4) * lda + col];
            s2 += A[(size_t)(r + 8) * lda + col];
            s3 += A[(size_t)(r + 12) * lda + col];
        }
        for (; r < r1; r += 4) s += A[(size_t)r * lda + col];
        s += s1 + s2 + s3;
    }
    red[rg][cl] = s;
    __syncthreads();
    if (rg == 0 && col < N) atomicAdd(out + col, red[0][cl] + red[1][cl] + red[2][cl] + red[3][cl]);
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

inline bool f32_dma_enabled() {
    const bool v = lob_variant(LOB_VAR_F32_DMA) != 0;
    return v;
}

int launch_nt(const GemmNT& g, int epi, hipStream_t s) {
    const int ntm = (g.M + BM - 1) / BM, ntn = (g.N + BN - 1) / BN;
    // LDS-DMA kernel: 16-B aligned K-contiguous fp32 operands, K a multiple of 32 and >= 4 k-tiles, no accumulate;
    // fragment epilogue needs whole tiles
    if (f32_dma_enabled() && aligned16(g.A) && aligned16(g.W) && g.lda % 4 == 0 && g.ldw % 4 == 0 && g.K % FTK == 0 &&
        g.K / FTK >= 4 && !g.accumulate && g.N <= 2048 && (epi == 0 || (g.M % BM == 0 && g.N % BN == 0))) {
        long gsz = 512;
        const long tiles = (long)ntm * ntn;
        if (gsz > tiles) gsz = ((tiles + 7) / 8) * 8;
        if (epi == 0 && g.amax_a && g.amax_w) hipLaunchKernelGGL((gemm_nt_dma_f32_kernel<0, true>), dim3((unsigned)gsz), dim3(256), 0, s, g);
        else if (epi == 0) hipLaunchKernelGGL((gemm_nt_dma_f32_kernel<0>), dim3((unsigned)gsz), dim3(256), 0, s, g);
        else          hipLaunchKernelGGL((gemm_nt_dma_f32_kernel<1>), dim3((unsigned)gsz), dim3(256), 0, s, g);
        LOB_CHECK_LAUNCH();
        return 0;
    }
    const dim3 grid((unsigned)(((ntm + 7) / 8) * 8 * ntn)), block(256);
    const bool vec = aligned16(g.A) && aligned16(g.W) && (g.lda % 4 == 0) && (g.ldw % 4 == 0) && (g.K % 4 == 0);
    if (epi == 0) {
        if (vec) hipLaunchKernelGGL((gemm_nt_kernel<true, 0>), grid, block, 0, s, g);
        else     hipLaunchKernelGGL((gemm_nt_kernel<false, 0>), grid, block, 0, s, g);
    } else {
        if (vec) hipLaunchKernelGGL((gemm_nt_kernel<true, 1>), grid, block, 0, s, g);
        else     hipLaunchKernelGGL((gemm_nt_kernel<false, 1>), grid, block, 0, s, g);
    }
    LOB_CHECK_LAUNCH();
    return 0;
}

}  // namespace

extern "C" int lob_gemm_nt_f32(const float* A, int lda, const float* W, int ldw, const float* bias,
                               float* C, int ldc, int M, int N, int K, int act, void* stream) {
    if (!A || !W || !C || M <= 0 || N <= 0 || K <= 0) return LOB_E_ARG;
    if (lda < K || ldw < K || ldc < N) return LOB_E_SHAPE;
    if ((act & 0xff) > LOB_ACT_GELU || act < 0) return LOB_E_ARG;
    GemmNT g{A, W, bias, C, lda, ldw, ldc, M, N, K, act & 0xff, (act >> 8) & 1, 0, 0, 0, 0, nullptr, nullptr};
    return launch_nt(g, 0, (hipStream_t)stream);
}

// C = A W^T with the fp32 products carried as two-way fp16 splits (see above).  amax_a / amax_w: device floats >= max|A| /
// max|W|.  Shapes the LDS-DMA kernel takes (16-B aligned operands, lda % 4 == ldw % 4 == 0, K % 32 == 0, K >= 128,
// N <= 2048); anything else: LOB_E_SHAPE (the caller keeps lob_gemm_nt_f32).
extern "C" int lob_gemm_nt_f32_split(const float* A, int lda, const float* W, int ldw, float* C, int ldc, int M, int N, int K,
                                     const float* amax_a, const float* amax_w, float drop_p, uint64_t seed, void* stream) {
    if (!A || !W || !C || !amax_a || !amax_w || M <= 0 || N <= 0 || K <= 0 || drop_p < 0.f || drop_p >= 1.f) return LOB_E_ARG;
    if (lda < K || ldw < K || ldc < N) return LOB_E_SHAPE;
    if (!(aligned16(A) && aligned16(W) && lda % 4 == 0 && ldw % 4 == 0 && K % FTK == 0 && K / FTK >= 4 && N <= 2048)) return LOB_E_SHAPE;
    if (drop_p > 0.f && lob_variant(LOB_VAR_F32_SPLIT) == 2) return LOB_E_SHAPE;        // the fragment-read twin has no mask epilogue
    GemmNT g{A, W, nullptr, C, lda, ldw, ldc, M, N, K, LOB_ACT_NONE, 0, 0, 0, 0, 0, amax_a, amax_w, drop_p, seed};
    const int ntm = (M + BM - 1) / BM, ntn = (N + BN - 1) / BN;
    long gsz = 512;
    const long tiles = (long)ntm * ntn;
    if (gsz > tiles) gsz = ((tiles + 7) / 8) * 8;
    // LOB_VAR_F32_SPLIT = 2: the twin that splits at every fragment read (LDS-DMA'd fp32 tiles)
    if (lob_variant(LOB_VAR_F32_SPLIT) == 2)
        hipLaunchKernelGGL((gemm_nt_dma_f32_kernel<0, true>), dim3((unsigned)gsz), dim3(256), 0, (hipStream_t)stream, g);
    else if (M % BM == 0 && N % BN == 0 && K % (2 * NKT) == 0 && lob_variant(LOB_VAR_F32_SPLIT) != 3)
        hipLaunchKernelGGL((gemm_nt_split_kernel<true>), dim3((unsigned)gsz), dim3(256), 0, (hipStream_t)stream, g);
    else                                              // F32_SPLIT = 3: the general kernel, the pipelined one's bit-identical twin
        hipLaunchKernelGGL((gemm_nt_split_kernel<false>), dim3((unsigned)gsz), dim3(256), 0, (hipStream_t)stream, g);
    LOB_CHECK_LAUNCH();
    return 0;
}

int lob_gate_gemm_ws_split(const float* X, int ldx, const float* Wih, const float* bias, float* P, int T, int Bp, int D,
                           int K, const float* range, hipStream_t s);        // gate_gemm_ws_split.hip

extern "C" int lob_gate_gemm_x_f32(const float* X, int ldx, const float* Wih, const float* bias,
                                   float* P, int T, int Bp, int H, int D, int K, int frag,
                                   const float* range, void* stream) {
    if (!X || !Wih || !P || T <= 0 || Bp <= 0 || H <= 0 || K <= 0 || (D != 1 && D != 2)) return LOB_E_ARG;
    if (ldx < K) return LOB_E_SHAPE;
    const int N = D * 4 * H;
    if (frag) {
        if ((H % 32) || (Bp % 32)) return LOB_E_SHAPE;
        if (!aligned16(P)) return LOB_E_ALIGN;
    }
    // H = 128: fp32-accurate two-way fp16 split on the 16-bit matrix pipe, weights stationary (gate_gemm_ws_split.hip);
    // LOB_VAR_F32_SPLIT = 0 keeps the exact-fp32 MFMA kernels of this file
    // (frag & 2: the caller has no bound on |X| -- exact kernels)
    if ((frag & 1) && !(frag & 2) && H == 128 && (K == 128 || K == 256) && (ldx % 4) == 0 && aligned16(X) && aligned16(Wih) &&
        lob_variant(LOB_VAR_F32_SPLIT) != 0)
        return lob_gate_gemm_ws_split(X, ldx, Wih, bias, P, T, Bp, D, K, range, (hipStream_t)stream);
    frag &= 1;
    GemmNT g{X, Wih, bias, P, ldx, K, N, T * Bp, N, K, LOB_ACT_NONE, 0, T, Bp, H, D, nullptr, nullptr};
    return launch_nt(g, frag ? 1 : 0, (hipStream_t)stream);
}

extern "C" int lob_gemm_tn_f32(const float* A, int lda, const float* B, int ldb, float* C, int ldc,
                               int M, int N, int Kc, void* stream) {
    if (!A || !B || !C || M <= 0 || N <= 0 || Kc <= 0) return LOB_E_ARG;
    if (lda < M || ldb < N || ldc < N) return LOB_E_SHAPE;
    const int tiles = ((M + 127) / 128) * ((N + 127) / 128);
    // ~2048 workgroups in flight; chunk a multiple of 32 rows, at least 256
    int nchunk = (2048 + tiles - 1) / tiles;
    int kchunk = (Kc + nchunk - 1) / nchunk;
    kchunk = ((kchunk + 31) / 32) * 32;
    if (kchunk < 256) kchunk = 256;
    nchunk = (Kc + kchunk - 1) / kchunk;
    GemmTN g{A, B, C, lda, ldb, ldc, M, N, Kc, kchunk, nullptr, nullptr};
    const bool vec = aligned16(A) && aligned16(B) && (lda % 4 == 0) && (ldb % 4 == 0) && (M % 4 == 0) && (N % 4 == 0);
    const dim3 grid((unsigned)(tiles * nchunk)), block(256);
    if (vec) hipLaunchKernelGGL((gemm_tn_kernel<true>), grid, block, 0, (hipStream_t)stream, g);
    else     hipLaunchKernelGGL((gemm_tn_kernel<false>), grid, block, 0, (hipStream_t)stream, g);
    LOB_CHECK_LAUNCH();
    return 0;
}

// C += A^T B with two-way fp16 split products (see lob_gemm_nt_f32_split).  16-B aligned operands, lda % 4 == ldb % 4 == 0,
// M % 4 == N % 4 == 0; anything else: LOB_E_SHAPE.
extern "C" int lob_gemm_tn_f32_split(const float* A, int lda, const float* B, int ldb, float* C, int ldc, int M, int N, int Kc,
                                     const float* amax_a, const float* amax_b, void* stream) {
    if (!A || !B || !C || !amax_a || !amax_b || M <= 0 || N <= 0 || Kc <= 0) return LOB_E_ARG;
    if (lda < M || ldb < N || ldc < N) return LOB_E_SHAPE;
    if (!(aligned16(A) && aligned16(B) && (lda % 4 == 0) && (ldb % 4 == 0) && (M % 4 == 0) && (N % 4 == 0))) return LOB_E_SHAPE;
    const int tiles = ((M + 127) / 128) * ((N + 127) / 128);
    if (lob_variant(LOB_VAR_F32_SPLIT) == 2) {          // the twin: split at every fragment read (tests, A/B)
        int nchunk = (2048 + tiles - 1) / tiles;
        int kchunk = (Kc + nchunk - 1) / nchunk;
        kchunk = ((kchunk + 31) / 32) * 32;
        if (kchunk < 256) kchunk = 256;
        nchunk = (Kc + kchunk - 1) / kchunk;
        GemmTN g{A, B, C, lda, ldb, ldc, M, N, Kc, kchunk, amax_a, amax_b};
        hipLaunchKernelGGL((gemm_tn_kernel<true, true>), dim3((unsigned)(tiles * nchunk)), dim3(256), 0, (hipStream_t)stream, g);
        LOB_CHECK_LAUNCH();
        return 0;
    }
    // split once while staging; the chunks of one contraction range sit 8 apart in blockIdx (one XCD: shared source tiles)
    int nchunk = (2048 + tiles - 1) / tiles;
    int kchunk = (Kc + nchunk - 1) / nchunk;
    kchunk = ((kchunk + 2 * STK - 1) / (2 * STK)) * (2 * STK);
    if (kchunk < 512) kchunk = 512;
    nchunk = (Kc + kchunk - 1) / kchunk;
    const int nchunk8 = ((nchunk + 7) / 8) * 8;
    GemmTNS g{A, B, C, lda, ldb, ldc, M, N, Kc, kchunk, tiles, amax_a, amax_b};
    // the pipelined kernel wants whole 128 x 128 tiles and an even number of 32-row stages per chunk; F32_SPLIT = 3 forces
    // the general kernel on the same chunks (its bit-identical twin)
    const bool fast = M % 128 == 0 && N % 128 == 0 && Kc % (2 * STK) == 0 && lob_variant(LOB_VAR_F32_SPLIT) != 3;
    if (fast) hipLaunchKernelGGL((gemm_tn_split_kernel<true>), dim3((unsigned)(tiles * nchunk8)), dim3(256), 0, (hipStream_t)stream, g);
    else      hipLaunchKernelGGL((gemm_tn_split_kernel<false>), dim3((unsigned)(tiles * nchunk8)), dim3(256), 0, (hipStream_t)stream, g);
    LOB_CHECK_LAUNCH();
    return 0;
}

extern "C" int lob_colsum_f32(const float* A, int lda, int M, int N, float* out, void* stream) {
    if (!A || !out || M <= 0 || N <= 0 || lda < N) return LOB_E_ARG;
    const int ncb = (N + 63) / 64;
    int nrb = (M + 127) / 128;                  // 128 rows per workgroup (32 per thread), at most 2048 workgroups of rows
    if (nrb > 2048) nrb = 2048;
    const int rpb = (M + nrb - 1) / nrb;
    nrb = (M + rpb - 1) / rpb;
    hipLaunchKernelGGL(colsum_kernel, dim3((unsigned)(ncb * nrb)), dim3(256), 0, (hipStream_t)stream,
                       A, lda, M, N, rpb, out);
    LOB_CHECK_LAUNCH();
    return 0;
}
