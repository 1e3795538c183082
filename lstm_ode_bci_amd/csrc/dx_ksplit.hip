// dX = dP W_ih for the mixed path (the gradient handed to the layer below; torch autograd of nn.LSTM's input-side
// Linear, training step 04_lstm_model.py:482-512): C[M, N] = A[M, K] * Wt[N, K]^T with K = D*4H in {512, 1024} (the wide
// contraction), N in {128, 256}, bf16 operands, fp32 accumulate, bf16 or fp32 C, optional fused dropout-backward mask.
//
// Why not the tiled LDS-DMA NT GEMM (gemm_nt_dma_kernel<0,256,256,...,ADEEP>): there both operands go through LDS and
// the 512-KB weight matrix is re-staged from L2 for every 256 rows with one k-tile of look-ahead -- 44 % of its wave
// cycles were parked in s_waitcnt / barriers, 0.80-0.87 ms for 2.7-3.2 GB of HBM traffic (3.1-4.0 TB/s).  The weights do
// not fit one wave's registers for a whole output column block (K = 1024), so this kernel splits the CONTRACTION over
// the eight waves of a workgroup instead:
//   * a workgroup owns 128 output columns; wave w keeps W^T[all 128 columns][k in its 1/8 slice] as MFMA B fragments
//     in registers (128 VGPRs at K = 1024) for the whole launch -- the weights never move again;
//   * per 16-row tile each wave loads ONLY its k-slice of the 16 dP rows (256 B per row) straight global -> VGPR as A
//     fragments (no LDS, no sharing: nobody else needs those bytes), three tiles ahead in four register sets;
//   * 32 v_mfma_f32_16x16x32_bf16 per wave give a 16 x 128 fp32 PARTIAL sum over the wave's k-slice; the eight partials
//     meet in LDS (double-buffered, one barrier per tile) and 512 threads add them up, apply the dropout mask and
//     store 16 rows x 256 B (bf16) / 512 B (fp32) as whole row segments.
// Bytes through LDS per tile: 64 KB of partials in, 64 KB out -- about the matrix time (2 x 512 cycles per SIMD), both
// well under the tile's share of HBM time.
// The A-fragment loads are issued by hand (inline asm) and retired by ONE counted s_waitcnt per tile: with
// compiler-visible loads pending across the loop's back edge hipcc drains the whole queue (vmcnt(0)) at the loop header,
// i.e. once per four tiles the three-tile look-ahead collapsed.  Per tile a wave issues KS loads and one store, in the
// order load(q+3), [wait for load(q)], MFMA, store(q): when the wait of tile q runs, the operations younger than load(q)
// are 3 x KS loads and 3 stores (fewer stores in the first three tiles, where the smaller count 3 x KS is used).
#include "lob_common.h"
#include <type_traits>

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

constexpr int RLD = 132;            // partial-sum row stride in floats (odd multiple of 4: conflict-free b32 writes)

struct DXArgs {
    const __bf16* A; const __bf16* Wt; void* C;
    int lda, ldc, M, N, out_bf16;
    float drop_p; uint64_t seed;
};

// KS = k-steps of 32 per wave (K = 8 * 32 * KS)
template <int KS>
__global__ __launch_bounds__(512, 2) void dx_ksplit_kernel(DXArgs g) {
    constexpr int K = 8 * 32 * KS;
    __shared__ __attribute__((aligned(16))) float red[2 * 8 * 16 * RLD];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c16 = lane & 15, rq = lane >> 4;
    const int ncg = g.N >> 7;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int cg = slot % ncg, rest = slot / ncg, nrest = (gridDim.x >> 3) / ncg;
    const int ntile = g.M >> 4;
    const int panels = (ntile - xcd + 7) / 8;
    if (rest >= panels) return;
    const int total = (panels - rest + nrest - 1) / nrest;
    const int kbase = w * 32 * KS;

    // stationary B fragments: wt[cb][ks] = Wt[128 cg + 16 cb + c16][kbase + 32 ks + 8 rq .. + 7]
    bf16x8 wt[8][KS];
    {
        const __bf16* wb = g.Wt + (size_t)(128 * cg + c16) * K + kbase + 8 * rq;
#pragma unroll
        for (int cb = 0; cb < 8; ++cb)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) wt[cb][ks] = *reinterpret_cast<const bf16x8*>(wb + (size_t)(16 * cb) * K + 32 * ks);
    }
#pragma unroll
    for (int cb = 0; cb < 8; ++cb)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) asm volatile("" : "+v"(wt[cb][ks]));

    // A fragments of tile u: A[m0 + c16][kbase + 32 ks + 8 rq .. + 7]
    const __bf16* alane = g.A + (size_t)c16 * g.lda + kbase + 8 * rq;
    auto tile_row0 = [&](int u) { return ((rest + nrest * u) * 8 + xcd) * 16; };
    auto load_a = [&](int u, bf16x8 (&dst)[KS]) {
        const int uu = u < total ? u : total - 1;          // past the end: re-read the last tile (never used): the
        const __bf16* p = alane + (size_t)tile_row0(uu) * g.lda;     // operation count per tile stays constant
        if constexpr (KS == 4)
            asm volatile("global_load_dwordx4 %0, %4, off\n\tglobal_load_dwordx4 %1, %4, off offset:64\n\t"
                         "global_load_dwordx4 %2, %4, off offset:128\n\tglobal_load_dwordx4 %3, %4, off offset:192"
                         : "=&v"(dst[0]), "=&v"(dst[1]), "=&v"(dst[2]), "=&v"(dst[3]) : "v"(p) : "memory");
        else
            asm volatile("global_load_dwordx4 %0, %2, off\n\tglobal_load_dwordx4 %1, %2, off offset:64"
                         : "=&v"(dst[0]), "=&v"(dst[1]) : "v"(p) : "memory");
    };
    // the wait names the registers (no use can be scheduled above it).  STEADY is a compile-time constant per call
    // site: with a run-time choice between the two counts hipcc merged the two asm statements' outputs through register
    // COPIES placed in front of the wait -- reads of in-flight load destinations (tools/isa_check.py pins this)
    auto wait_a = [&](auto steady, bf16x8 (&a)[KS]) {
        constexpr int N = decltype(steady)::value ? 3 * KS + 3 : 3 * KS;
        if constexpr (KS == 4) asm volatile("s_waitcnt vmcnt(%4)" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]) : "n"(N));
        else                   asm volatile("s_waitcnt vmcnt(%2)" : "+v"(a[0]), "+v"(a[1]) : "n"(N));
    };
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the weights have landed: from here on the queue is counted by hand
    bf16x8 a0[KS], a1[KS], a2[KS], a3[KS];
    load_a(0, a0); load_a(1, a1); load_a(2, a2);

    const int rrow = tid >> 5, rc4 = (tid & 31) * 4;       // reduction: this thread's row / first of its 4 columns

    auto one_tile = [&](int q, bf16x8 (&a)[KS], bf16x8 (&anext)[KS], auto steady) {
        load_a(q + 3, anext);                               // three tiles ahead of the one being consumed
        wait_a(steady, a);
        f32x4 acc[8];
#pragma unroll
        for (int cb = 0; cb < 8; ++cb) { f32x4 z = {0.f, 0.f, 0.f, 0.f}; acc[cb] = z; }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int cb = 0; cb < 8; ++cb)
                acc[cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[ks], wt[cb][ks], acc[cb], 0, 0, 0);
        // partial sums of this wave's k-slice -> LDS: D layout of the 16x16 MFMA: rows 4 rq + r, column 16 cb + c16
        float* rw = red + ((q & 1) * 8 + w) * 16 * RLD + 4 * rq * RLD + c16;
#pragma unroll
        for (int cb = 0; cb < 8; ++cb)
#pragma unroll
            for (int r = 0; r < 4; ++r) rw[r * RLD + 16 * cb] = acc[cb][r];
        __syncthreads();
        // every thread adds the eight partials of 4 neighbouring columns of one row
        const float* rr = red + (q & 1) * 8 * 16 * RLD + rrow * RLD + rc4;
        f32x4 s = *reinterpret_cast<const f32x4*>(rr);
#pragma unroll
        for (int ww = 1; ww < 8; ++ww) {
            const f32x4 t = *reinterpret_cast<const f32x4*>(rr + ww * 16 * RLD);
            s[0] += t[0]; s[1] += t[1]; s[2] += t[2]; s[3] += t[3];
        }
        const size_t o = (size_t)(tile_row0(q) + rrow) * g.ldc + 128 * cg + rc4;
        if (g.drop_p > 0.f) {                               // the mask of element (row * ldc + col), as lob_dropout_f32
            float d0, d1, d2, d3;
            lob_dropout_scale2(g.seed, (uint64_t)o, g.drop_p, d0, d1);
            lob_dropout_scale2(g.seed, (uint64_t)o + 2, g.drop_p, d2, d3);
            s[0] *= d0; s[1] *= d1; s[2] *= d2; s[3] *= d3;
        }
        if (g.out_bf16) {
            bf16x4 v = {(__bf16)s[0], (__bf16)s[1], (__bf16)s[2], (__bf16)s[3]};
            *reinterpret_cast<bf16x4*>(reinterpret_cast<__bf16*>(g.C) + o) = v;
        } else {
            *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(g.C) + o) = s;
        }
    };

    constexpr std::integral_constant<bool, false> first{};
    constexpr std::integral_constant<bool, true> steady{};
    one_tile(0, a0, a3, first);                            // the first three tiles: no stores older than the awaited loads yet
    if (1 < total) one_tile(1, a1, a0, first);
    if (2 < total) one_tile(2, a2, a1, first);
    for (int q = 3; q < total; q += 4) {
        one_tile(q, a3, a2, steady);
        if (q + 1 < total) one_tile(q + 1, a0, a3, steady);
        if (q + 2 < total) one_tile(q + 2, a1, a0, steady);
        if (q + 3 < total) one_tile(q + 3, a2, a1, steady);
    }
}

}  // namespace

// Internal entry point used by lob_gemm_nt_bf16 (gemm_bf16.hip).  Preconditions checked by the caller: bf16 A / Wt, no
// bias / activation / accumulate, K in {512, 1024}, N in {128, 256}, M % 16 == 0, lda % 8 == 0, ldw == K, ldc % 4 == 0
// (and even: the dropout pairs), 16-B aligned bases.
int lob_dx_ksplit(const void* A, int lda, const void* Wt, void* C, int ldc, int M, int N, int K, int out_bf16, float drop_p,
                  uint64_t seed, hipStream_t s) {
    const int ncg = N / 128;
    const int ntile = M / 16;
    int nrx = (ntile + 7) / 8;                         // row-tile walkers per XCD; one workgroup per CU at most
    const int cap = 32 / ncg;
    if (nrx > cap) nrx = cap;
    DXArgs g{(const __bf16*)A, (const __bf16*)Wt, C, lda, ldc, M, N, out_bf16, drop_p, seed};
    const dim3 grid((unsigned)(8 * ncg * nrx)), block(512);
    if (K == 1024) hipLaunchKernelGGL(dx_ksplit_kernel<4>, grid, block, 0, s, g);
    else           hipLaunchKernelGGL(dx_ksplit_kernel<2>, grid, block, 0, s, g);
    LOB_CHECK_LAUNCH();
    return 0;
}
