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
#include "lob_common.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 32, LDT = 36;

struct GemmNT {
    const float* A; const float* W; const float* bias; float* C;
    int lda, ldw, ldc, M, N, K, act;
    // fragment epilogue (gate pre-activations): N = D*4H, M = T*Bp
    int T, Bp, H, D;
};

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
    const int ntn = (g.N + BN - 1) / BN;
    const int m0 = (blockIdx.x / ntn) * BM, n0 = (blockIdx.x % ntn) * BN;

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
                    if (row < g.M) g.C[(size_t)row * g.ldc + col] = apply_act(acc[i][j][r] + bv, g.act);
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

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

int launch_nt(const GemmNT& g, int epi, hipStream_t s) {
    const int ntm = (g.M + BM - 1) / BM, ntn = (g.N + BN - 1) / BN;
    const dim3 grid((unsigned)(ntm * ntn)), block(256);
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
    if (act < LOB_ACT_NONE || act > LOB_ACT_GELU) return LOB_E_ARG;
    GemmNT g{A, W, bias, C, lda, ldw, ldc, M, N, K, act, 0, 0, 0, 0};
    return launch_nt(g, 0, (hipStream_t)stream);
}

extern "C" int lob_gate_gemm_x_f32(const float* X, int ldx, const float* Wih, const float* bias,
                                   float* P, int T, int Bp, int H, int D, int K, int frag,
                                   void* stream) {
    if (!X || !Wih || !P || T <= 0 || Bp <= 0 || H <= 0 || K <= 0 || (D != 1 && D != 2)) return LOB_E_ARG;
    if (ldx < K) return LOB_E_SHAPE;
    const int N = D * 4 * H;
    if (frag) {
        if ((H % 32) || (Bp % 32)) return LOB_E_SHAPE;
        if (!aligned16(P)) return LOB_E_ALIGN;
    }
    GemmNT g{X, Wih, bias, P, ldx, K, N, T * Bp, N, K, LOB_ACT_NONE, T, Bp, H, D};
    return launch_nt(g, frag ? 1 : 0, (hipStream_t)stream);
}
