// Ping-pong bf16 GEMMs for the H = 256 mixed training step (the reference's real checkpoint size, hidden_size = 256 for
// 61 channels, 04_lstm_model.py:877; training step 04:482-512 under autocast 04:487).
//
// At H = 256 the three tiled GEMM families of a layer are MATRIX-bound, not HBM-bound: per launch at B = 4096
//     gate GEMM   P  = X  W_ih^T      1M x 2048 x 512   2.2 TFLOP for 5.4 GB  (410 FLOP/B; machine balance ~310)
//     dX          dX = dP W_ih        1M x 512 x 2048   2.2 TFLOP for 5.4 GB
//     dW          dW = dP^T [X | Y]   2048 x 768 x 1M   3.3 TFLOP for 6.3 GB
// and the one-barrier-per-k-tile kernels of gemm_bf16.hip (16 waves, 64 x 64 per wave) sat at 0.9-0.95 PFLOP/s with 42 % of
// their wave cycles parked in s_waitcnt / s_barrier (profiles/r02_train_mixed_B4096_H256_sq_counters.csv).
//
// Structure (cdna_hip_programming.md section 5, the 256^2 template, rebuilt around v_mfma_f32_32x32x16_bf16 because the
// recurrent kernels' P layout is that instruction's accumulator order):
//   * 256 x 256 output tile, 64-deep k-tiles, EIGHT waves as 2 x 4, 128 x 64 per wave (128 accumulator registers): half
//     the LDS fragment bytes per MFMA of the 64 x 64 wave tile;
//   * the waves form two groups (wr = 0 / 1), one wave of each on every SIMD, running the SAME instruction stream ONE
//     BARRIER APART: every interval between two barriers one group issues 8 MFMAs (256 cycles of its SIMD's matrix
//     pipe) while the other reads its next fragments from LDS and issues its share of the operand DMA -- the matrix pipe
//     always has a wave with operands in registers;
//   * two LDS buffers per operand (4 x 32 KB).  An operand region is re-staged as soon as BOTH groups have read it, so
//     the DMA of k-tile q + 2 starts in the middle of k-tile q: 2 global_load_lds_dwordx4 per wave and interval, one
//     counted s_waitcnt vmcnt per k-tile and wave, never 0 inside the loop; raw s_barrier;
//   * fragment reads are inline asm (a compiler-visible LDS read of a DMA target gets an s_waitcnt vmcnt(0) in front);
//     the s_waitcnt lgkmcnt(0) that ends a read segment names the fragment registers, so no MFMA can move above it.
//
// NT kernel (A[M,K] W[N,K]^T, both k-contiguous), per k-tile q and wave (buffer b = q & 1):
//     L1  A rows 0-63 (of the wave's 128) k 0-31, B k 0-31            DMA: A-hi(q+1) -> b^1
//     M1  acc[rb 0,1][cb 0,1] += k 0-31
//     L2  A rows 0-63 k 32-63, B k 32-63                              DMA: B-h1(q+1) -> b^1
//     M2  acc[rb 0,1][..] += k 32-63
//     L3  A rows 64-127 k 0-31 (B stays in registers)                 DMA: A-lo(q+2) -> b    (A-lo(q): free after L2)
//     M3  acc[rb 2,3][..] += k 0-31
//     L4  A rows 64-127 k 32-63                                       DMA: B-h0(q+2) -> b    (B(q): free after L2)
//         s_waitcnt vmcnt(4): everything of k-tile q + 1 has landed (younger: the 4 DMAs of L3 / L4)
//     M4  acc[rb 2,3][..] += k 32-63
// (A-lo = rows 0-63 and 128-191 of the tile: the first halves of both groups' rows; "free after L2" means after the
//  barrier that follows the SECOND group's L2, one interval later.)  Reads per segment 8 / 8 / 4 / 4 ds_read_b128.
//
// TN kernel (C += A[Kc,M]^T B[Kc,N], both k-major: the weight gradients): the [k][cols] images need no transformation
// on the way in, ds_read_b64_tr_b16 transposes on the read side; the phases split the k-tile by k-STEP (a k-quarter of
// both operands is consumed per phase and re-staged right away: 6 fragments per segment, operand DMA three quarters
// ahead + one k-tile: vmcnt(6)).
#include "lob_common.h"
#include <type_traits>

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void gbl_cvoid;

constexpr int PP_OP = 32768;          // bytes of one operand tile in LDS (256 rows x 64 k, or 64 k x 256 columns)
constexpr int PP_BIAS = 4 * PP_OP;    // byte offset of the bias image (NT fragment epilogue)

struct PPArgs {
    const __bf16* A; const __bf16* W; void* C; const float* bias;
    int lda, ldw, ldc, M, N, K;
    int T, Bp, H;                     // fragment epilogue
    int out_bf16; float drop_p; uint64_t seed;
};

__device__ __forceinline__ f32x16 pp_mfma(bf16x8 a, bf16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

#define PP_RD128(dst, addr, OFF) \
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=&v"(dst) : "v"(addr), "n"(OFF) : "memory")
#define PP_BAR()                                   \
    do {                                           \
        __builtin_amdgcn_sched_barrier(0);         \
        __builtin_amdgcn_s_barrier();              \
        __builtin_amdgcn_sched_barrier(0);         \
    } while (0)

// EPI 0: row-major C[M, N] (bf16 or fp32), optional dropout mask of element (row * ldc + col): the MFMA runs with its
//        operands SWAPPED (D[n][m]), so a lane holds 4 consecutive columns of one row per accumulator quad.
// EPI 1: the gate GEMM: bf16 P in the recurrent kernels' fragment order [d][t][bt][w H/32][gate 4][q pair 2][lane 64][8]
//        (include/lob.h) + bias.
// ABL (diagnostic builds only, results are garbage): 1 = no operand DMA inside the loop, 2 = no fragment reads,
// 4 = no MFMAs -- what each component costs the interval (tools/pp_bench.py abl)
// PRIO (A/B of the priority protocol): 0 = s_setprio 1 around every MFMA cluster, 1 = no priority changes,
// 2 = the READ / DMA segment runs at priority 1 instead
template <int EPI, int ABL = 0, int PRIO = 0>
__global__ __launch_bounds__(512, 2) void gemm_nt_pp_kernel(PPArgs g) {
    constexpr bool SWAP = EPI == 0;
    __shared__ __attribute__((aligned(1024))) unsigned char lds[4 * PP_OP + (EPI == 1 ? 8192 : 0)];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int r31 = lane & 31, hi = lane >> 5;
    const int ntn = g.N >> 8, ntm = g.M >> 8;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, nslot = gridDim.x >> 3;
    const int panels = (ntm - xcd + 7) / 8, ntile = panels * ntn;      // XCD x owns the row panels mt = x (mod 8)
    if (slot >= ntile) return;
    const int nk = g.K >> 6;                                             // even (checked by the host)
    const int my_tiles = (ntile - slot + nslot - 1) / nslot;
    const int last_it = slot + (my_tiles - 1) * nslot;
    const int total = my_tiles * nk;

    if constexpr (EPI == 1) {          // bias image (read in the epilogue through an asm-opaque LDS load)
        float* bs = reinterpret_cast<float*>(lds + PP_BIAS);
        for (int i = tid; i < g.N && i < 2048; i += 512) bs[i] = g.bias ? g.bias[i] : 0.f;
        __syncthreads();
    }

    // ---- producer side: operand DMA.  One wave instruction = 8 rows x 128 B; 16-B chunk c of row r lands at chunk
    //      slot c ^ ((r >> 1) & 7) (applied on the per-lane SOURCE address; the same XOR on the fragment read).
    struct Cur { int it, kt; const char* pa; const char* pb; };
    auto set_cur = [&](Cur& c) {
        const int m0 = ((c.it / ntn) * 8 + xcd) << 8, n0 = (c.it % ntn) << 8;
        c.pa = reinterpret_cast<const char*>(g.A) + ((size_t)m0 * g.lda + (size_t)c.kt * 64) * 2;
        c.pb = reinterpret_cast<const char*>(g.W) + ((size_t)n0 * g.ldw + (size_t)c.kt * 64) * 2;
    };
    auto advance = [&](Cur& c) {       // next k-tile of this workgroup's sequence; past the end: stay on the last one
        if (c.kt + 1 < nk) { ++c.kt; c.pa += 128; c.pb += 128; }
        else if (c.it < last_it) { c.it += nslot; c.kt = 0; set_cur(c); }
    };
    const int l3 = lane >> 3, l7 = lane & 7;
    unsigned asrc[2], bsrc[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int ra = wr * 128 + wc * 16 + j * 8 + l3;                  // A-lo row of this lane (A-hi: + 64)
        asrc[j] = 2u * (unsigned)(ra * g.lda + ((l7 ^ ((ra >> 1) & 7)) << 3));
        const int rb = wave * 16 + j * 8 + l3;                           // B row inside a 128-row half
        bsrc[j] = 2u * (unsigned)(rb * g.ldw + ((l7 ^ ((rb >> 1) & 7)) << 3));
    }
    const size_t a_hi_b = (size_t)64 * g.lda * 2, b_h1_b = (size_t)128 * g.ldw * 2;
    unsigned char* const a_dst = lds + wr * 16384 + wc * 2048;           // + buf * PP_OP (+ 8192: A-hi)
    unsigned char* const b_dst = lds + 2 * PP_OP + wave * 2048;          // + buf * PP_OP + half * 16384
    auto dma2 = [&](const char* src, unsigned o0, unsigned o1, unsigned char* dst) {
        __builtin_amdgcn_global_load_lds((gbl_cvoid*)(src + o0), (lds_void*)dst, 16, 0, 0);
        __builtin_amdgcn_global_load_lds((gbl_cvoid*)(src + o1), (lds_void*)(dst + 1024), 16, 0, 0);
    };
    bool in_loop = false;
    auto stage_a = [&](const Cur& c, int buf, int hi_half) {
        if ((ABL & 1) && in_loop) return;
        dma2(c.pa + (hi_half ? a_hi_b : 0), asrc[0], asrc[1], a_dst + buf * PP_OP + hi_half * 8192);
    };
    auto stage_b = [&](const Cur& c, int buf, int half) {
        if ((ABL & 1) && in_loop) return;
        dma2(c.pb + (half ? b_h1_b : 0), bsrc[0], bsrc[1], b_dst + buf * PP_OP + half * 16384);
    };

    // ---- consumer side: fragment read addresses: row wr*128 + 32 rb + r31 (rb, buffer: immediates), k-step ks
    const unsigned lds_b = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)lds;
    const int sw = (r31 >> 1) & 7;
    unsigned aoff[4], boff[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        aoff[ks] = lds_b + (unsigned)((wr * 128 + r31) * 128 + (((2 * ks + hi) ^ sw) << 4));
        boff[ks] = lds_b + (unsigned)(2 * PP_OP + (wc * 64 + r31) * 128 + (((2 * ks + hi) ^ sw) << 4));
    }

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][c][r] = 0.f;

    // ---- prologue: all of k-tile 0, A-lo / B-h0 of k-tile 1
    Cur c1{slot, 0, nullptr, nullptr};
    set_cur(c1);
    stage_a(c1, 0, 0); stage_b(c1, 0, 0); stage_a(c1, 0, 1); stage_b(c1, 0, 1);
    advance(c1);
    stage_a(c1, 1, 0); stage_b(c1, 1, 0);
    Cur c2 = c1;
    advance(c2);
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    PP_BAR();
    if (wr == 1) PP_BAR();             // the second group runs one barrier behind

    bf16x8 af[2][2], bf[2][4];
    if constexpr (ABL != 0) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int j = 0; j < 2; ++j) { for (int e = 0; e < 8; ++e) af[i][j][e] = (__bf16)0.f; asm volatile("" : "+v"(af[i][j])); }
#pragma unroll
            for (int j = 0; j < 4; ++j) { for (int e = 0; e < 8; ++e) bf[i][j][e] = (__bf16)0.f; asm volatile("" : "+v"(bf[i][j])); }
        }
        in_loop = true;
    }
#define PP_MMA(ACC, AF, BF) do { if constexpr (!(ABL & 4)) ACC = SWAP ? pp_mfma(BF, AF, ACC) : pp_mfma(AF, BF, ACC); } while (0)
#define PP_WAIT4()                                                                                              \
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(af[0][0]), "+v"(af[0][1]), "+v"(af[1][0]), "+v"(af[1][1]))
#define PP_RDA(dst, addr, OFF) do { if constexpr (!(ABL & 2)) PP_RD128(dst, addr, OFF); } while (0)
    auto ktile = [&](auto bufc) {      // one k-tile out of buffer BUF; c1 / c2 = k-tiles q + 1 / q + 2
        constexpr int BUF = decltype(bufc)::value, BO = BUF * PP_OP, NB = BUF ^ 1;
        // ---- L1 / M1
        PP_RDA(af[0][0], aoff[0], BO);        PP_RDA(af[0][1], aoff[1], BO);
        PP_RDA(af[1][0], aoff[0], BO + 4096); PP_RDA(af[1][1], aoff[1], BO + 4096);
        PP_RDA(bf[0][0], boff[0], BO);        PP_RDA(bf[0][1], boff[1], BO);
        PP_RDA(bf[1][0], boff[0], BO + 4096); PP_RDA(bf[1][1], boff[1], BO + 4096);
        stage_a(c1, NB, 1);
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(af[0][0]), "+v"(af[0][1]), "+v"(af[1][0]), "+v"(af[1][1]),
                     "+v"(bf[0][0]), "+v"(bf[0][1]), "+v"(bf[1][0]), "+v"(bf[1][1]));
        PP_BAR();
        if constexpr (PRIO == 0) __builtin_amdgcn_s_setprio(1); else if constexpr (PRIO == 2) __builtin_amdgcn_s_setprio(0);
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int c = 0; c < 2; ++c) PP_MMA(acc[i][c], af[i][s], bf[c][s]);
        if constexpr (PRIO == 0) __builtin_amdgcn_s_setprio(0); else if constexpr (PRIO == 2) __builtin_amdgcn_s_setprio(1);
        PP_BAR();
        // ---- L2 / M2
        PP_RDA(af[0][0], aoff[2], BO);        PP_RDA(af[0][1], aoff[3], BO);
        PP_RDA(af[1][0], aoff[2], BO + 4096); PP_RDA(af[1][1], aoff[3], BO + 4096);
        PP_RDA(bf[0][2], boff[2], BO);        PP_RDA(bf[0][3], boff[3], BO);
        PP_RDA(bf[1][2], boff[2], BO + 4096); PP_RDA(bf[1][3], boff[3], BO + 4096);
        stage_b(c1, NB, 1);
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(af[0][0]), "+v"(af[0][1]), "+v"(af[1][0]), "+v"(af[1][1]),
                     "+v"(bf[0][2]), "+v"(bf[0][3]), "+v"(bf[1][2]), "+v"(bf[1][3]));
        PP_BAR();
        if constexpr (PRIO == 0) __builtin_amdgcn_s_setprio(1); else if constexpr (PRIO == 2) __builtin_amdgcn_s_setprio(0);
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int c = 0; c < 2; ++c) PP_MMA(acc[i][c], af[i][s], bf[c][2 + s]);
        if constexpr (PRIO == 0) __builtin_amdgcn_s_setprio(0); else if constexpr (PRIO == 2) __builtin_amdgcn_s_setprio(1);
        PP_BAR();
        // ---- L3 / M3
        PP_RDA(af[0][0], aoff[0], BO + 8192);  PP_RDA(af[0][1], aoff[1], BO + 8192);
        PP_RDA(af[1][0], aoff[0], BO + 12288); PP_RDA(af[1][1], aoff[1], BO + 12288);
        stage_a(c2, BUF, 0);
        PP_WAIT4();
        PP_BAR();
        if constexpr (PRIO == 0) __builtin_amdgcn_s_setprio(1); else if constexpr (PRIO == 2) __builtin_amdgcn_s_setprio(0);
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int c = 0; c < 2; ++c) PP_MMA(acc[2 + i][c], af[i][s], bf[c][s]);
        if constexpr (PRIO == 0) __builtin_amdgcn_s_setprio(0); else if constexpr (PRIO == 2) __builtin_amdgcn_s_setprio(1);
        PP_BAR();
        // ---- L4 / M4
        PP_RDA(af[0][0], aoff[2], BO + 8192);  PP_RDA(af[0][1], aoff[3], BO + 8192);
        PP_RDA(af[1][0], aoff[2], BO + 12288); PP_RDA(af[1][1], aoff[3], BO + 12288);
        stage_b(c2, BUF, 0);
        PP_WAIT4();
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");     // k-tile q + 1 complete (this wave's share)
        PP_BAR();
        if constexpr (PRIO == 0) __builtin_amdgcn_s_setprio(1); else if constexpr (PRIO == 2) __builtin_amdgcn_s_setprio(0);
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int c = 0; c < 2; ++c) PP_MMA(acc[2 + i][c], af[i][s], bf[c][2 + s]);
        if constexpr (PRIO == 0) __builtin_amdgcn_s_setprio(0); else if constexpr (PRIO == 2) __builtin_amdgcn_s_setprio(1);
        PP_BAR();
        c1 = c2;
        advance(c2);
    };

    int it = slot, kt = 0;
    for (int q = 0; q < total; q += 2) {
        ktile(std::integral_constant<int, 0>{});
        ktile(std::integral_constant<int, 1>{});
        kt += 2;
        if (kt < nk) continue;

        // ---- epilogue of output tile `it` (the other group keeps the matrix pipe for one more interval)
        kt = 0;
        const int m0 = ((it / ntn) * 8 + xcd) << 8, n0 = (it % ntn) << 8;
        it += nslot;
        if constexpr (EPI == 0) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = m0 + wr * 128 + 32 * i + r31;
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    const size_t o = (size_t)row * g.ldc + n0 + wc * 64 + 32 * c + 4 * hi;
#pragma unroll
                    for (int qd = 0; qd < 4; ++qd) {
                        float v0 = acc[i][c][4 * qd], v1 = acc[i][c][4 * qd + 1], v2 = acc[i][c][4 * qd + 2],
                              v3 = acc[i][c][4 * qd + 3];
                        if (g.drop_p > 0.f) {
                            float d0, d1, d2, d3;
                            lob_dropout_scale2(g.seed, (uint64_t)(o + 8 * qd), g.drop_p, d0, d1);
                            lob_dropout_scale2(g.seed, (uint64_t)(o + 8 * qd) + 2, g.drop_p, d2, d3);
                            v0 *= d0; v1 *= d1; v2 *= d2; v3 *= d3;
                        }
                        if (g.out_bf16) {
                            bf16x4 v = {(__bf16)v0, (__bf16)v1, (__bf16)v2, (__bf16)v3};
                            *reinterpret_cast<bf16x4*>(reinterpret_cast<__bf16*>(g.C) + o + 8 * qd) = v;
                        } else {
                            f32x4 v = {v0, v1, v2, v3};
                            *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(g.C) + o + 8 * qd) = v;
                        }
                    }
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[i][c][r] = 0.f;
                }
            }
        } else {
            const int NBT = g.Bp >> 5, NW = g.H >> 5, H4 = 4 * g.H;
            const float* bs = reinterpret_cast<const float*>(lds + PP_BIAS);
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const int ncol = n0 + wc * 64 + 32 * c;
                const int d = ncol / H4, gate = (ncol % H4) / g.H, w = (ncol % g.H) >> 5;
                const float bv = lds_read_f32_opaque(bs + ncol + r31);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int mrow = m0 + wr * 128 + 32 * i;
                    const int t = mrow / g.Bp, bt = (mrow - t * g.Bp) >> 5;
                    const size_t fo = ((((size_t)(d * g.T + t) * NBT + bt) * NW + w) * 4 + gate) * 1024;
                    __bf16* dst = reinterpret_cast<__bf16*>(g.C) + fo + lane * 8;
#pragma unroll
                    for (int pq = 0; pq < 2; ++pq) {
                        bf16x8 v;
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] = (__bf16)(acc[i][c][8 * pq + e] + bv);
                        __builtin_nontemporal_store(v, reinterpret_cast<bf16x8*>(dst + pq * 512));
                    }
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[i][c][r] = 0.f;
                }
            }
        }
    }
    if (wr == 0) PP_BAR();             // barrier counts of the two groups match
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#undef PP_MMA
#undef PP_WAIT4
#undef PP_RDA
}

// ------------------------------------------------------------------------------------------------------------------
// The same NT schedule on v_mfma_f32_16x16x32_bf16 (row-major epilogue only).  Why: these kernels run the chip into
// its power limit -- SQ_WAVE_CYCLES / wall time of the 32x32x16 kernel is 1.37-1.40 GHz (profiles/r03_*), i.e. the
// matrix pipe is ~78 % busy at the clock the chip holds -- and MI355X_MICROARCH.md ('DVFS give-back' item 7) measures
// the 16x16x32 shape at ~1.15 x the FLOP/s of 32x32x16 at equal cycles per FLOP (it holds a higher clock).  Same tile,
// same LDS images, same DMA, same segment structure (8 / 8 / 4 / 4 ds_read_b128, 16 MFMAs of 16 cycles per matrix
// segment); a fragment is 16 rows x 32 k (lane: row l & 15, 16-B chunk 4 kstep + (l >> 4)), and the same source-side
// swizzle (chunk ^ ((row >> 1) & 7)) is conflict-free for that read pattern too.
// ------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(512, 2) void gemm_nt_pp16_kernel(PPArgs g) {
    __shared__ __attribute__((aligned(1024))) unsigned char lds[4 * PP_OP];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int r15 = lane & 15, c4 = lane >> 4;
    const int ntn = g.N >> 8, ntm = g.M >> 8;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, nslot = gridDim.x >> 3;
    const int panels = (ntm - xcd + 7) / 8, ntile = panels * ntn;
    if (slot >= ntile) return;
    const int nk = g.K >> 6;
    const int my_tiles = (ntile - slot + nslot - 1) / nslot;
    const int last_it = slot + (my_tiles - 1) * nslot;
    const int total = my_tiles * nk;

    struct Cur { int it, kt; const char* pa; const char* pb; };
    auto set_cur = [&](Cur& c) {
        const int m0 = ((c.it / ntn) * 8 + xcd) << 8, n0 = (c.it % ntn) << 8;
        c.pa = reinterpret_cast<const char*>(g.A) + ((size_t)m0 * g.lda + (size_t)c.kt * 64) * 2;
        c.pb = reinterpret_cast<const char*>(g.W) + ((size_t)n0 * g.ldw + (size_t)c.kt * 64) * 2;
    };
    auto advance = [&](Cur& c) {
        if (c.kt + 1 < nk) { ++c.kt; c.pa += 128; c.pb += 128; }
        else if (c.it < last_it) { c.it += nslot; c.kt = 0; set_cur(c); }
    };
    const int l3 = lane >> 3, l7 = lane & 7;
    unsigned asrc[2], bsrc[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int ra = wr * 128 + wc * 16 + j * 8 + l3;
        asrc[j] = 2u * (unsigned)(ra * g.lda + ((l7 ^ ((ra >> 1) & 7)) << 3));
        const int rb = wave * 16 + j * 8 + l3;
        bsrc[j] = 2u * (unsigned)(rb * g.ldw + ((l7 ^ ((rb >> 1) & 7)) << 3));
    }
    const size_t a_hi_b = (size_t)64 * g.lda * 2, b_h1_b = (size_t)128 * g.ldw * 2;
    unsigned char* const a_dst = lds + wr * 16384 + wc * 2048;
    unsigned char* const b_dst = lds + 2 * PP_OP + wave * 2048;
    auto dma2 = [&](const char* src, unsigned o0, unsigned o1, unsigned char* dst) {
        __builtin_amdgcn_global_load_lds((gbl_cvoid*)(src + o0), (lds_void*)dst, 16, 0, 0);
        __builtin_amdgcn_global_load_lds((gbl_cvoid*)(src + o1), (lds_void*)(dst + 1024), 16, 0, 0);
    };
    auto stage_a = [&](const Cur& c, int buf, int hi_half) {
        dma2(c.pa + (hi_half ? a_hi_b : 0), asrc[0], asrc[1], a_dst + buf * PP_OP + hi_half * 8192);
    };
    auto stage_b = [&](const Cur& c, int buf, int half) {
        dma2(c.pb + (half ? b_h1_b : 0), bsrc[0], bsrc[1], b_dst + buf * PP_OP + half * 16384);
    };

    const unsigned lds_b = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)lds;
    const int sw = (r15 >> 1) & 7;
    unsigned aoff[2], boff[2];       // k-step 0 / 1 (32 k each): 16-B chunk 4 kstep + c4
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        aoff[ks] = lds_b + (unsigned)((wr * 128 + r15) * 128 + (((4 * ks + c4) ^ sw) << 4));
        boff[ks] = lds_b + (unsigned)(2 * PP_OP + (wc * 64 + r15) * 128 + (((4 * ks + c4) ^ sw) << 4));
    }

    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int c = 0; c < 4; ++c) { f32x4 z = {0.f, 0.f, 0.f, 0.f}; acc[i][c] = z; }

    Cur c1{slot, 0, nullptr, nullptr};
    set_cur(c1);
    stage_a(c1, 0, 0); stage_b(c1, 0, 0); stage_a(c1, 0, 1); stage_b(c1, 0, 1);
    advance(c1);
    stage_a(c1, 1, 0); stage_b(c1, 1, 0);
    Cur c2 = c1;
    advance(c2);
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    PP_BAR();
    if (wr == 1) PP_BAR();

    bf16x8 af[4], bf[4][2];
#define PP_W4A() asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(af[0]), "+v"(af[1]), "+v"(af[2]), "+v"(af[3]))
#define PP_M16(R0, KS)                                                                                       \
    do {                                                                                                     \
        __builtin_amdgcn_s_setprio(1);                                                                       \
        _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_)                                                     \
            _Pragma("unroll") for (int c_ = 0; c_ < 4; ++c_)                                                 \
                acc[R0 + i_][c_] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[c_][KS], af[i_], acc[R0 + i_][c_], 0, 0, 0);   \
        __builtin_amdgcn_s_setprio(0);                                                                       \
    } while (0)
    auto ktile = [&](auto bufc) {
        constexpr int BUF = decltype(bufc)::value, BO = BUF * PP_OP, NB = BUF ^ 1;
        // ---- L1 / M1: rows 0-63, k 0-31
#pragma unroll
        for (int i = 0; i < 4; ++i) PP_RD128(af[i], aoff[0], BO + i * 2048);
#pragma unroll
        for (int c = 0; c < 4; ++c) PP_RD128(bf[c][0], boff[0], BO + c * 2048);
        stage_a(c1, NB, 1);
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(af[0]), "+v"(af[1]), "+v"(af[2]), "+v"(af[3]),
                     "+v"(bf[0][0]), "+v"(bf[1][0]), "+v"(bf[2][0]), "+v"(bf[3][0]));
        PP_BAR();
        PP_M16(0, 0);
        PP_BAR();
        // ---- L2 / M2: rows 0-63, k 32-63
#pragma unroll
        for (int i = 0; i < 4; ++i) PP_RD128(af[i], aoff[1], BO + i * 2048);
#pragma unroll
        for (int c = 0; c < 4; ++c) PP_RD128(bf[c][1], boff[1], BO + c * 2048);
        stage_b(c1, NB, 1);
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(af[0]), "+v"(af[1]), "+v"(af[2]), "+v"(af[3]),
                     "+v"(bf[0][1]), "+v"(bf[1][1]), "+v"(bf[2][1]), "+v"(bf[3][1]));
        PP_BAR();
        PP_M16(0, 1);
        PP_BAR();
        // ---- L3 / M3: rows 64-127, k 0-31
#pragma unroll
        for (int i = 0; i < 4; ++i) PP_RD128(af[i], aoff[0], BO + 8192 + i * 2048);
        stage_a(c2, BUF, 0);
        PP_W4A();
        PP_BAR();
        PP_M16(4, 0);
        PP_BAR();
        // ---- L4 / M4: rows 64-127, k 32-63
#pragma unroll
        for (int i = 0; i < 4; ++i) PP_RD128(af[i], aoff[1], BO + 8192 + i * 2048);
        stage_b(c2, BUF, 0);
        PP_W4A();
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        PP_BAR();
        PP_M16(4, 1);
        PP_BAR();
        c1 = c2;
        advance(c2);
    };

    int it = slot, kt = 0;
    for (int q = 0; q < total; q += 2) {
        ktile(std::integral_constant<int, 0>{});
        ktile(std::integral_constant<int, 1>{});
        kt += 2;
        if (kt < nk) continue;
        kt = 0;
        const int m0 = ((it / ntn) * 8 + xcd) << 8, n0 = (it % ntn) << 8;
        it += nslot;
        // operands swapped (D[n][m]): lane = row m0 + .. + r15, columns 4 c4 .. + 3 of each 16-column block
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int row = m0 + wr * 128 + 16 * i + r15;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const size_t o = (size_t)row * g.ldc + n0 + wc * 64 + 16 * c + 4 * c4;
                float v0 = acc[i][c][0], v1 = acc[i][c][1], v2 = acc[i][c][2], v3 = acc[i][c][3];
                if (g.drop_p > 0.f) {
                    float d0, d1, d2, d3;
                    lob_dropout_scale2(g.seed, (uint64_t)o, g.drop_p, d0, d1);
                    lob_dropout_scale2(g.seed, (uint64_t)o + 2, g.drop_p, d2, d3);
                    v0 *= d0; v1 *= d1; v2 *= d2; v3 *= d3;
                }
                if (g.out_bf16) {
                    bf16x4 v = {(__bf16)v0, (__bf16)v1, (__bf16)v2, (__bf16)v3};
                    *reinterpret_cast<bf16x4*>(reinterpret_cast<__bf16*>(g.C) + o) = v;
                } else {
                    f32x4 v = {v0, v1, v2, v3};
                    *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(g.C) + o) = v;
                }
                f32x4 z = {0.f, 0.f, 0.f, 0.f};
                acc[i][c] = z;
            }
        }
    }
    if (wr == 0) PP_BAR();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#undef PP_W4A
#undef PP_M16
}

// ------------------------------------------------------------------------------------------------------------------
// TN: C[i][j] += sum_k A[k][i] B[k][j] over this workgroup's contraction chunk, fp32 atomics at the end (split-k).
// One (256 x 256 output tile, chunk) per workgroup.  LDS images [64 k][256 columns] (512-B rows), 64-B granule g of
// k-row kr stored at granule g ^ (kr & 3) (on the DMA source address and on the tr read: the four k-rows of a
// ds_read_b64_tr_b16 block then cover the 256-B bank row).
// B may be a time-shifted source (the h_prev operand of dW_hh): rows [ex_lo, ex_hi) have no predecessor -- their k-tiles
// are fetched unshifted and their products skipped (Bp % 64 == 0: a k-tile never straddles two time steps).
// ------------------------------------------------------------------------------------------------------------------
struct TNPPArgs {
    const __bf16* A; const __bf16* B; float* C;
    int lda, ldb, ldc, M, N, Kc, kchunk, tiles;
    int shift, ex_lo, ex_hi;          // B row = k + shift outside [ex_lo, ex_hi)
};

#define PP_TR(dst, addr, OFF) \
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=&v"(dst) : "v"(addr), "n"(OFF) : "memory")

__global__ __launch_bounds__(512, 2) void gemm_tn_pp_kernel(TNPPArgs g) {
    __shared__ __attribute__((aligned(1024))) unsigned char lds[4 * PP_OP];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int ntn = g.N >> 8;
    // workgroups of one contraction chunk are 8 apart in blockIdx: one XCD under round-robin placement, so the source
    // tiles they share are fetched from HBM once (speed only)
    const int xcd = blockIdx.x & 7, rest = blockIdx.x >> 3;
    const int tile = rest % g.tiles, chunk = (rest / g.tiles) * 8 + xcd;
    const int m0 = (tile / ntn) << 8, n0 = (tile % ntn) << 8;
    const int kbeg = chunk * g.kchunk, kend = min(g.Kc, kbeg + g.kchunk);
    if (kbeg >= kend) return;
    const int total = (kend - kbeg) >> 6;                                // even (host)

    // ---- producer: one wave instruction = 2 k-rows x 512 B; per k-quarter (16 k-rows) one instruction per wave and operand
    const int kr2 = 2 * wave + (lane >> 5), ch = lane & 31;
    const unsigned asrc = 2u * (unsigned)(kr2 * g.lda + ((ch ^ ((kr2 & 3) << 2)) << 3));
    const unsigned bsrc = 2u * (unsigned)(kr2 * g.ldb + ((ch ^ ((kr2 & 3) << 2)) << 3));
    const char* const a_base = reinterpret_cast<const char*>(g.A + m0);
    const char* const b_base = reinterpret_cast<const char*>(g.B + n0);
    unsigned char* const a_dst = lds + wave * 1024;                       // + buf * PP_OP + kq * 8192
    unsigned char* const b_dst = lds + 2 * PP_OP + wave * 1024;
    auto stage = [&](int p, int buf, int kq) {     // k-quarter kq of k-tile p (clamped: past the end the last one again)
        const int pp = p < total ? p : total - 1;
        const int k0 = kbeg + 64 * pp;
        const bool ex = k0 >= g.ex_lo && k0 < g.ex_hi;
        const char* pa = a_base + ((size_t)(k0 + 16 * kq) * g.lda) * 2;
        const char* pb = b_base + ((size_t)(k0 + 16 * kq + (ex ? 0 : g.shift)) * g.ldb) * 2;
        __builtin_amdgcn_global_load_lds((gbl_cvoid*)(pa + asrc), (lds_void*)(a_dst + buf * PP_OP + kq * 8192), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((gbl_cvoid*)(pb + bsrc), (lds_void*)(b_dst + buf * PP_OP + kq * 8192), 16, 0, 0);
    };

    // ---- consumer: tr-read addresses.  Element (k-row 8 h + q, column c + 16 mh + 4 p) of a block sits at column
    //      (c ^ 32 q) + 16 mh + 4 p; the 32-column block index meets the swizzle in its two low bits only.
    const int fh = lane >> 5, fmh = (lane >> 4) & 1, fq = (lane >> 2) & 3, fp = lane & 3;
    const unsigned lds_b = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)lds;
    const unsigned frow = (unsigned)((8 * fh + fq) * 512 + 2 * (16 * fmh + 4 * fp));
    const unsigned a0 = lds_b + frow + 2u * (unsigned)(wr * 128 + 32 * fq);                 // block rb: ^ (rb << 6)
    const unsigned b0 = lds_b + 2 * PP_OP + frow + 2u * (unsigned)((wc * 64) ^ (32 * fq));  // block cb: ^ (cb << 6)
    unsigned av[4], bv[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) av[i] = a0 ^ (unsigned)(i << 6);
    bv[0] = b0; bv[1] = b0 ^ 64u;

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][c][r] = 0.f;

    // ---- prologue: k-tile 0 and k-quarters 0..2 of k-tile 1
#pragma unroll
    for (int kq = 0; kq < 4; ++kq) stage(0, 0, kq);
#pragma unroll
    for (int kq = 0; kq < 3; ++kq) stage(1, 1, kq);
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    PP_BAR();
    if (wr == 1) PP_BAR();

    bf16x4 al[4], ah[4], bl[2], bh[2];
#define PP_FRAG(l, h) bf16x8{l[0], l[1], l[2], l[3], h[0], h[1], h[2], h[3]}
    auto phase = [&](auto bufc, auto ksc, int q, bool on) {
        constexpr int BUF = decltype(bufc)::value, KS = decltype(ksc)::value, O = BUF * PP_OP + KS * 8192;
#pragma unroll
        for (int i = 0; i < 4; ++i) { PP_TR(al[i], av[i], O); PP_TR(ah[i], av[i], O + 2048); }
#pragma unroll
        for (int c = 0; c < 2; ++c) { PP_TR(bl[c], bv[c], O); PP_TR(bh[c], bv[c], O + 2048); }
        // DMA: the k-quarter both groups finished reading one phase ago
        if constexpr (KS == 0) stage(q + 1, BUF ^ 1, 3);
        else                   stage(q + 2, BUF, KS - 1);
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(al[0]), "+v"(ah[0]), "+v"(al[1]), "+v"(ah[1]), "+v"(al[2]), "+v"(ah[2]),
                     "+v"(al[3]), "+v"(ah[3]), "+v"(bl[0]), "+v"(bh[0]), "+v"(bl[1]), "+v"(bh[1]));
        if constexpr (KS == 3) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");      // k-tile q + 1 complete
        PP_BAR();
        if (on) {
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int c = 0; c < 2; ++c) acc[i][c] = pp_mfma(PP_FRAG(al[i], ah[i]), PP_FRAG(bl[c], bh[c]), acc[i][c]);
            __builtin_amdgcn_s_setprio(0);
        }
        PP_BAR();
    };
    auto ktile = [&](auto bufc, int q) {
        const int k0 = kbeg + 64 * q;
        const bool on = !(k0 >= g.ex_lo && k0 < g.ex_hi);
        phase(bufc, std::integral_constant<int, 0>{}, q, on);
        phase(bufc, std::integral_constant<int, 1>{}, q, on);
        phase(bufc, std::integral_constant<int, 2>{}, q, on);
        phase(bufc, std::integral_constant<int, 3>{}, q, on);
    };
    for (int q = 0; q < total; q += 2) {
        ktile(std::integral_constant<int, 0>{}, q);
        ktile(std::integral_constant<int, 1>{}, q + 1);
    }
    if (wr == 0) PP_BAR();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#undef PP_FRAG
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int col = n0 + wc * 64 + 32 * c + (lane & 31);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wr * 128 + 32 * i + acc_row(r, lane);
                atomicAdd(g.C + (size_t)row * g.ldc + col, acc[i][c][r]);
            }
        }
}


// ------------------------------------------------------------------------------------------------------------------
// Ring schedule.  Measured on the first schedule (tools/pp_bench.py abl, dX at B = 4096; full kernel 1.91 ms): MFMAs +
// barriers alone 1.21 ms, fragment reads + barriers alone 0.81 ms, the operand DMA + barriers ALONE 1.66 ms -- the
// kernel is bound by how fast 17 GB cross L2 -> LDS, and that rate was set by bytes in flight: with two buffers per
// operand the last-staged quarter of a k-tile is requested only 4 intervals (~0.8 us) before the wait that retires
// it.  (A schedule with 16 MFMAs per segment, i.e. 2 intervals of lead, was built and measured: 2.49 ms -- slower.)
// Here ALL 160 KB of LDS form one ring of 16-KB slots (10: two and a half k-tiles) and every quarter of k-tile q + 2
// is requested during k-tile q, as soon as its slot has been read: 10-14 intervals ahead, ~96 KB in flight per CU.
//   chunks of a k-tile, in ring order: A-lo | B rows 0-127 | B rows 128-255 | A-hi  (the first three are read in the
//   k-tile's first two segments, A-hi in the last two)
//   L1(q): DMA A-lo(q+2)   L2(q): B-h0(q+2), then vmcnt(12): A-hi(q) landed      (read in L3 / L4)
//   L3(q): DMA B-h1(q+2)   L4(q): A-hi(q+2), then vmcnt(10): A-lo, B of q+1 landed (read from L1(q+1) on)
// Slot offsets rotate at run time (scalar registers; the fragment-read address registers are rebuilt once per k-tile).
// ------------------------------------------------------------------------------------------------------------------
constexpr int RS = 16384;            // ring slot bytes

__global__ __launch_bounds__(512, 2) void gemm_nt_ring_kernel(PPArgs g) {
    constexpr int NS = 10;
    __shared__ __attribute__((aligned(1024))) unsigned char lds[NS * RS];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int r31 = lane & 31, hi = lane >> 5;
    const int ntn = g.N >> 8, ntm = g.M >> 8;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, nslot = gridDim.x >> 3;
    const int panels = (ntm - xcd + 7) / 8, ntile = panels * ntn;
    if (slot >= ntile) return;
    const int nk = g.K >> 6;
    const int my_tiles = (ntile - slot + nslot - 1) / nslot;
    const int last_it = slot + (my_tiles - 1) * nslot;
    const int total = my_tiles * nk;

    struct Cur { int it, kt; const char* pa; const char* pb; };
    auto set_cur = [&](Cur& c) {
        const int m0 = ((c.it / ntn) * 8 + xcd) << 8, n0 = (c.it % ntn) << 8;
        c.pa = reinterpret_cast<const char*>(g.A) + ((size_t)m0 * g.lda + (size_t)c.kt * 64) * 2;
        c.pb = reinterpret_cast<const char*>(g.W) + ((size_t)n0 * g.ldw + (size_t)c.kt * 64) * 2;
    };
    auto advance = [&](Cur& c) {       // next k-tile of this workgroup's sequence; past the end: stay on the last one
        if (c.kt + 1 < nk) { ++c.kt; c.pa += 128; c.pb += 128; }
        else if (c.it < last_it) { c.it += nslot; c.kt = 0; set_cur(c); }
    };
    const int l3 = lane >> 3, l7 = lane & 7;
    unsigned asrc[2], bsrc[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int ra = wr * 128 + wc * 16 + j * 8 + l3;                  // A-lo row of this lane (A-hi: + 64)
        asrc[j] = 2u * (unsigned)(ra * g.lda + ((l7 ^ ((ra >> 1) & 7)) << 3));
        const int rb = wave * 16 + j * 8 + l3;                           // B row inside a 128-row half
        bsrc[j] = 2u * (unsigned)(rb * g.ldw + ((l7 ^ ((rb >> 1) & 7)) << 3));
    }
    const size_t a_hi_b = (size_t)64 * g.lda * 2, b_h1_b = (size_t)128 * g.ldw * 2;
    // destination of this wave's two DMA instructions inside a slot: A chunks [wr][64 rows][128 B], B chunks [128 rows]
    const int a_wo = wr * 8192 + wc * 2048, b_wo = wave * 2048;
    auto dma2 = [&](const char* src, unsigned o0, unsigned o1, int slot_off, int wo) {
        unsigned char* dst = lds + slot_off + wo;
        __builtin_amdgcn_global_load_lds((gbl_cvoid*)(src + o0), (lds_void*)dst, 16, 0, 0);
        __builtin_amdgcn_global_load_lds((gbl_cvoid*)(src + o1), (lds_void*)(dst + 1024), 16, 0, 0);
    };
    auto wrap = [&](int so) { return so >= NS * RS ? so - NS * RS : so; };

    const unsigned lds_b = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)lds;
    const int sw = (r31 >> 1) & 7;
    unsigned abase[4], bbase[4];      // lane addresses inside a chunk (k-step ks); + slot offset + immediate
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        abase[ks] = lds_b + (unsigned)(wr * 8192 + r31 * 128 + (((2 * ks + hi) ^ sw) << 4));
        bbase[ks] = lds_b + (unsigned)(((wc & 1) * 64 + r31) * 128 + (((2 * ks + hi) ^ sw) << 4));
    }

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][c][r] = 0.f;

    // ---- prologue: k-tiles 0 and 1 in ring order (slots 0..7)
    Cur c2{slot, 0, nullptr, nullptr};
    set_cur(c2);
#pragma unroll 1
    for (int t = 0; t < 2; ++t) {
        dma2(c2.pa, asrc[0], asrc[1], (4 * t) * RS, a_wo);
        dma2(c2.pb, bsrc[0], bsrc[1], (4 * t + 1) * RS, b_wo);
        dma2(c2.pb + b_h1_b, bsrc[0], bsrc[1], (4 * t + 2) * RS, b_wo);
        dma2(c2.pa + a_hi_b, asrc[0], asrc[1], (4 * t + 3) * RS, a_wo);
        advance(c2);
    }
    int s0 = 0;                        // slot byte offset of the consumed k-tile's first chunk; k-tile q + 2: + 8 slots
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    PP_BAR();
    if (wr == 1) PP_BAR();

    bf16x8 af[2][2], bf[2][4];
    unsigned alo[4], ahi[4], bb[4];
#define PP_WAIT4()                                                                                              \
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(af[0][0]), "+v"(af[0][1]), "+v"(af[1][0]), "+v"(af[1][1]))
#define PP_M8(R0, K0)                                                                     \
    do {                                                                                  \
        __builtin_amdgcn_s_setprio(1);                                                    \
        _Pragma("unroll") for (int s_ = 0; s_ < 2; ++s_)                                  \
            _Pragma("unroll") for (int i_ = 0; i_ < 2; ++i_)                              \
                _Pragma("unroll") for (int c_ = 0; c_ < 2; ++c_)                          \
                    acc[R0 + i_][c_] = pp_mfma(bf[c_][K0 + s_], af[i_][s_], acc[R0 + i_][c_]);   \
        __builtin_amdgcn_s_setprio(0);                                                    \
    } while (0)

    int it = slot, kt = 0;
    for (int q = 0; q < total; ++q) {
        const int sB = wrap(s0 + (1 + (wc >> 1)) * RS), sH = wrap(s0 + 3 * RS);   // this wave's B half, A-hi
        const int p0 = wrap(s0 + 8 * RS);                                         // first chunk of k-tile q + 2
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) { alo[ks] = abase[ks] + (unsigned)s0; ahi[ks] = abase[ks] + (unsigned)sH; bb[ks] = bbase[ks] + (unsigned)sB; }
        // ---- L1 / M1
        PP_RD128(af[0][0], alo[0], 0);    PP_RD128(af[0][1], alo[1], 0);
        PP_RD128(af[1][0], alo[0], 4096); PP_RD128(af[1][1], alo[1], 4096);
        PP_RD128(bf[0][0], bb[0], 0);     PP_RD128(bf[0][1], bb[1], 0);
        PP_RD128(bf[1][0], bb[0], 4096);  PP_RD128(bf[1][1], bb[1], 4096);
        dma2(c2.pa, asrc[0], asrc[1], p0, a_wo);
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(af[0][0]), "+v"(af[0][1]), "+v"(af[1][0]), "+v"(af[1][1]),
                     "+v"(bf[0][0]), "+v"(bf[0][1]), "+v"(bf[1][0]), "+v"(bf[1][1]));
        PP_BAR();
        PP_M8(0, 0);
        PP_BAR();
        // ---- L2 / M2
        PP_RD128(af[0][0], alo[2], 0);    PP_RD128(af[0][1], alo[3], 0);
        PP_RD128(af[1][0], alo[2], 4096); PP_RD128(af[1][1], alo[3], 4096);
        PP_RD128(bf[0][2], bb[2], 0);     PP_RD128(bf[0][3], bb[3], 0);
        PP_RD128(bf[1][2], bb[2], 4096);  PP_RD128(bf[1][3], bb[3], 4096);
        dma2(c2.pb, bsrc[0], bsrc[1], wrap(p0 + RS), b_wo);
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(af[0][0]), "+v"(af[0][1]), "+v"(af[1][0]), "+v"(af[1][1]),
                     "+v"(bf[0][2]), "+v"(bf[0][3]), "+v"(bf[1][2]), "+v"(bf[1][3]));
        asm volatile("s_waitcnt vmcnt(12)" ::: "memory");    // A-hi of THIS k-tile (and everything older) has landed
        PP_BAR();
        PP_M8(0, 2);
        PP_BAR();
        // ---- L3 / M3
        PP_RD128(af[0][0], ahi[0], 0);    PP_RD128(af[0][1], ahi[1], 0);
        PP_RD128(af[1][0], ahi[0], 4096); PP_RD128(af[1][1], ahi[1], 4096);
        dma2(c2.pb + b_h1_b, bsrc[0], bsrc[1], wrap(p0 + 2 * RS), b_wo);
        PP_WAIT4();
        PP_BAR();
        PP_M8(2, 0);
        PP_BAR();
        // ---- L4 / M4
        PP_RD128(af[0][0], ahi[2], 0);    PP_RD128(af[0][1], ahi[3], 0);
        PP_RD128(af[1][0], ahi[2], 4096); PP_RD128(af[1][1], ahi[3], 4096);
        dma2(c2.pa + a_hi_b, asrc[0], asrc[1], wrap(p0 + 3 * RS), a_wo);
        PP_WAIT4();
        asm volatile("s_waitcnt vmcnt(10)" ::: "memory");    // A-lo and both B halves of k-tile q + 1 have landed
        PP_BAR();
        PP_M8(2, 2);
        PP_BAR();
        advance(c2);
        s0 = wrap(s0 + 4 * RS);
        if (++kt < nk) continue;

        // ---- epilogue of output tile `it`: row-major C, MFMA operands swapped (a lane holds 4 consecutive columns)
        kt = 0;
        const int m0 = ((it / ntn) * 8 + xcd) << 8, n0 = (it % ntn) << 8;
        it += nslot;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = m0 + wr * 128 + 32 * i + r31;
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const size_t o = (size_t)row * g.ldc + n0 + wc * 64 + 32 * c + 4 * hi;
#pragma unroll
                for (int qd = 0; qd < 4; ++qd) {
                    float v0 = acc[i][c][4 * qd], v1 = acc[i][c][4 * qd + 1], v2 = acc[i][c][4 * qd + 2],
                          v3 = acc[i][c][4 * qd + 3];
                    if (g.drop_p > 0.f) {
                        float d0, d1, d2, d3;
                        lob_dropout_scale2(g.seed, (uint64_t)(o + 8 * qd), g.drop_p, d0, d1);
                        lob_dropout_scale2(g.seed, (uint64_t)(o + 8 * qd) + 2, g.drop_p, d2, d3);
                        v0 *= d0; v1 *= d1; v2 *= d2; v3 *= d3;
                    }
                    if (g.out_bf16) {
                        bf16x4 v = {(__bf16)v0, (__bf16)v1, (__bf16)v2, (__bf16)v3};
                        *reinterpret_cast<bf16x4*>(reinterpret_cast<__bf16*>(g.C) + o + 8 * qd) = v;
                    } else {
                        f32x4 v = {v0, v1, v2, v3};
                        *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(g.C) + o + 8 * qd) = v;
                    }
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][c][r] = 0.f;
            }
        }
    }
    if (wr == 0) PP_BAR();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#undef PP_WAIT4
#undef PP_M8
}

// TN ring: 20 slots of 8 KB (a k-quarter of one operand: 16 k-rows x 512 B); the pair (A, B) of k-quarter kq of k-tile
// q + 2 is requested in segment kq of k-tile q (its slots were read two and a half k-tiles earlier); every segment ends
// with the same s_waitcnt vmcnt(14): the pair the NEXT segment reads was issued 8 segments ago, 7 x 2 DMAs are younger.
constexpr int TS = 8192;

__global__ __launch_bounds__(512, 2) void gemm_tn_ring_kernel(TNPPArgs g) {
    constexpr int NS = 20;
    __shared__ __attribute__((aligned(1024))) unsigned char lds[NS * TS];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int ntn = g.N >> 8;
    const int xcd = blockIdx.x & 7, rest = blockIdx.x >> 3;
    const int tile = rest % g.tiles, chunk = (rest / g.tiles) * 8 + xcd;
    const int m0 = (tile / ntn) << 8, n0 = (tile % ntn) << 8;
    const int kbeg = chunk * g.kchunk, kend = min(g.Kc, kbeg + g.kchunk);
    if (kbeg >= kend) return;
    const int total = (kend - kbeg) >> 6;

    const int kr2 = 2 * wave + (lane >> 5), ch = lane & 31;
    const unsigned asrc = 2u * (unsigned)(kr2 * g.lda + ((ch ^ ((kr2 & 3) << 2)) << 3));
    const unsigned bsrc = 2u * (unsigned)(kr2 * g.ldb + ((ch ^ ((kr2 & 3) << 2)) << 3));
    const char* const a_base = reinterpret_cast<const char*>(g.A + m0);
    const char* const b_base = reinterpret_cast<const char*>(g.B + n0);
    unsigned char* const w_dst = lds + wave * 1024;
    auto wrap = [&](int so) { return so >= NS * TS ? so - NS * TS : so; };
    auto stage = [&](int p, int kq, int slot_off) {        // (A, B) k-quarter kq of k-tile p -> slots slot_off, slot_off + TS
        const int pp = p < total ? p : total - 1;
        const int k0 = kbeg + 64 * pp;
        const bool ex = k0 >= g.ex_lo && k0 < g.ex_hi;
        const char* pa = a_base + ((size_t)(k0 + 16 * kq) * g.lda) * 2;
        const char* pb = b_base + ((size_t)(k0 + 16 * kq + (ex ? 0 : g.shift)) * g.ldb) * 2;
        __builtin_amdgcn_global_load_lds((gbl_cvoid*)(pa + asrc), (lds_void*)(w_dst + slot_off), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((gbl_cvoid*)(pb + bsrc), (lds_void*)(w_dst + wrap(slot_off + TS)), 16, 0, 0);
    };

    const int fh = lane >> 5, fmh = (lane >> 4) & 1, fq = (lane >> 2) & 3, fp = lane & 3;
    const unsigned lds_b = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)lds;
    const unsigned frow = (unsigned)((8 * fh + fq) * 512 + 2 * (16 * fmh + 4 * fp));
    const unsigned a0 = lds_b + frow + 2u * (unsigned)(wr * 128 + 32 * fq);
    const unsigned b0 = lds_b + frow + 2u * (unsigned)((wc * 64) ^ (32 * fq));

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][c][r] = 0.f;

    // prologue: k-tiles 0 and 1 (slots 0..15), then the pair the first segment reads must have landed
#pragma unroll 1
    for (int t = 0; t < 2; ++t)
#pragma unroll 1
        for (int kq = 0; kq < 4; ++kq) stage(t, kq, (8 * t + 2 * kq) * TS);
    int s0 = 0;
    asm volatile("s_waitcnt vmcnt(14)" ::: "memory");
    PP_BAR();
    if (wr == 1) PP_BAR();

    bf16x4 al[4], ah[4], bl[2], bh[2];
#define PP_FRAG(l, h) bf16x8{l[0], l[1], l[2], l[3], h[0], h[1], h[2], h[3]}
    for (int q = 0; q < total; ++q) {
        const int k0 = kbeg + 64 * q;
        const bool on = !(k0 >= g.ex_lo && k0 < g.ex_hi);
#pragma unroll
        for (int kq = 0; kq < 4; ++kq) {
            const int sa = wrap(s0 + 2 * kq * TS), sb = wrap(sa + TS);
            const unsigned va = a0 + (unsigned)sa, vb = b0 + (unsigned)sb;
            unsigned av[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) av[i] = va ^ (unsigned)(i << 6);
            const unsigned bv0 = vb, bv1 = vb ^ 64u;
#pragma unroll
            for (int i = 0; i < 4; ++i) { PP_TR(al[i], av[i], 0); PP_TR(ah[i], av[i], 2048); }
            PP_TR(bl[0], bv0, 0); PP_TR(bh[0], bv0, 2048);
            PP_TR(bl[1], bv1, 0); PP_TR(bh[1], bv1, 2048);
            stage(q + 2, kq, wrap(s0 + (16 + 2 * kq) * TS));
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(al[0]), "+v"(ah[0]), "+v"(al[1]), "+v"(ah[1]), "+v"(al[2]), "+v"(ah[2]),
                         "+v"(al[3]), "+v"(ah[3]), "+v"(bl[0]), "+v"(bh[0]), "+v"(bl[1]), "+v"(bh[1]));
            asm volatile("s_waitcnt vmcnt(14)" ::: "memory");
            PP_BAR();
            if (on) {
                __builtin_amdgcn_s_setprio(1);
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int c = 0; c < 2; ++c) acc[i][c] = pp_mfma(PP_FRAG(al[i], ah[i]), PP_FRAG(bl[c], bh[c]), acc[i][c]);
                __builtin_amdgcn_s_setprio(0);
            }
            PP_BAR();
        }
        s0 = wrap(s0 + 8 * TS);
    }
    if (wr == 0) PP_BAR();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#undef PP_FRAG
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int col = n0 + wc * 64 + 32 * c + (lane & 31);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wr * 128 + 32 * i + acc_row(r, lane);
                atomicAdd(g.C + (size_t)row * g.ldc + col, acc[i][c][r]);
            }
        }
}

// ------------------------------------------------------------------------------------------------------------------
// Both weight gradients of one H = 256 LSTM layer from ONE pass over dP (04_lstm_model.py:490 loss.backward() through
// nn.LSTM, hidden_size = 256: 04:877):
//     dW_ih[d] (4H x NX)  = sum_rows dP[row, d]^T X[row]
//     dW_hh[d] (4H x H)   = sum_rows dP[row, d]^T h_prev_d[row],   h_prev_0[t] = Y[t-1, 0:H], h_prev_1[t] = Y[t+1, H:2H]
// The ping-pong TN schedule above with a WIDER output tile: 256 dP-columns x 32 NCB x 4 columns of [X | h_prev_d]
// (NCB = 3: 384 columns at NX = 512, NCB = 2: 256 at NX = 256), 8 waves as 2 x 4, 128 x 32 NCB accumulators per wave
// (192 registers at NCB = 3).  Per dP-column tile the operand [X | h_prev_d] (NX + 256 columns) is exactly TWO such
// tiles, so a layer is 8 x 2 = 16 tiles x 16 contraction chunks = 256 workgroups, the two chunks' worth of tiles that
// share an XCD read each dP tile twice and each [X | h_prev] tile eight times out of ONE L2.  Against three launches of
// the 256 x 256 kernel: dP crosses HBM once instead of twice, and a workgroup moves (256 + 384) x 2 B through L2 -> LDS
// per 256 x 384 MFMA columns instead of (256 + 256) per 256 x 256: -17 % of the bytes that bound these kernels
// (the CU's 39 B/clk LDS-DMA path, tools/pp_bench.py dma).
// The B tile is two LDS images: 128 columns (256-B rows: NCB = 3 only) + 256 columns (512-B rows); tile 1 of a
// dP-column tile takes its 256-column image from Y, shifted by one time step (Bp rows) -- k-tiles of the step without
// a predecessor are fetched unshifted and their h_prev products skipped (Bp % 64 == 0: a k-tile never straddles steps).
// LDS: A[2] 2 x 32 KB | B256[2] 2 x 32 KB | B128[2] 2 x 16 KB = 160 KB.
// DMA per k-quarter and wave: one instruction for A, one for B256, and (waves 0-3, NCB = 3) one for B128: the one
// counted wait per k-tile is vmcnt(9) for waves 0-3 and vmcnt(6) for waves 4-7.
// ------------------------------------------------------------------------------------------------------------------
struct DWPPArgs {
    const __bf16* dP; const __bf16* X; const __bf16* Y; float* dWih; float* dWhh;
    int ldp, ldx, ldy, nx, H, T, Bp, kchunk;
};

template <int NCB>
__global__ __launch_bounds__(512, 2) void lstm_dw_pp_kernel(DWPPArgs g) {
    constexpr int OA = 0, OB = 65536, OS = 131072;                    // LDS byte offsets: A[2], B256[2], B128[2]
    __shared__ __attribute__((aligned(1024))) unsigned char lds[NCB == 3 ? 163840 : 131072];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int xcd = blockIdx.x & 7, rest = blockIdx.x >> 3;
    const int tile = rest & 15, chunk = (rest >> 4) * 8 + xcd;        // 8 dP-column tiles x 2 operand tiles
    const int mt = tile >> 1, nt = tile & 1;
    const int m0 = mt << 8, d = m0 / (4 * g.H);
    const int Kc = g.T * g.Bp;
    const int kbeg = chunk * g.kchunk, kend = min(Kc, kbeg + g.kchunk);
    if (kbeg >= kend) return;
    const int total = (kend - kbeg) >> 6;
    // rows [ex_lo, ex_hi) of dP have no h_prev
    const int ex_lo = d == 0 ? 0 : (g.T - 1) * g.Bp, ex_hi = ex_lo + g.Bp;
    const int yshift = d == 0 ? -g.Bp : g.Bp;

    // ---- the two B images of this operand tile: source, leading dimension, time shift; and where their products go
    const bool y256 = nt == 1;                                          // the 256-column image comes from h_prev
    const __bf16* s256 = y256 ? g.Y + d * g.H : g.X + (NCB == 3 ? 128 : 0);
    const int ld256 = y256 ? g.ldy : g.ldx;
    const __bf16* s128 = g.X + (nt == 0 ? 0 : 384);                    // NCB = 3 only: X columns 0-127 / 384-511
    float* o256 = y256 ? g.dWhh : g.dWih + (NCB == 3 ? 128 : 0);
    const int lo256 = y256 ? g.H : g.nx;
    float* o128 = g.dWih + (nt == 0 ? 0 : 384);

    // ---- producer
    const int kr2 = 2 * wave + (lane >> 5), ch = lane & 31;            // 512-B rows: 2 k-rows per instruction
    const unsigned asrc = 2u * (unsigned)(kr2 * g.ldp + ((ch ^ ((kr2 & 3) << 2)) << 3));
    const unsigned bsrc = 2u * (unsigned)(kr2 * ld256 + ((ch ^ ((kr2 & 3) << 2)) << 3));
    const int kr4 = 4 * (wave & 3) + (lane >> 4), c16 = lane & 15;      // 256-B rows: 4 k-rows per instruction (waves 0-3)
    const unsigned ssrc = 2u * (unsigned)(kr4 * g.ldx + ((c16 ^ ((kr4 & 3) << 2)) << 3));
    const char* const a_base = reinterpret_cast<const char*>(g.dP + m0);
    const char* const b_base = reinterpret_cast<const char*>(s256);
    const char* const s_base = reinterpret_cast<const char*>(s128);
    unsigned char* const w_dst = lds + wave * 1024;
    const bool seg_wave = NCB == 3 && wave < 4;
    auto stage = [&](int p, int buf, int kq) {
        const int pp = p < total ? p : total - 1;
        const int k0 = kbeg + 64 * pp;
        const bool ex = k0 >= ex_lo && k0 < ex_hi;
        const size_t kr = (size_t)(k0 + 16 * kq);
        const char* pa = a_base + kr * g.ldp * 2;
        const char* pb = b_base + (size_t)((long)kr + ((y256 && !ex) ? yshift : 0)) * ld256 * 2;
        __builtin_amdgcn_global_load_lds((gbl_cvoid*)(pa + asrc), (lds_void*)(w_dst + OA + buf * 32768 + kq * 8192), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((gbl_cvoid*)(pb + bsrc), (lds_void*)(w_dst + OB + buf * 32768 + kq * 8192), 16, 0, 0);
        if (seg_wave) {
            const char* ps = s_base + kr * g.ldx * 2;
            __builtin_amdgcn_global_load_lds((gbl_cvoid*)(ps + ssrc), (lds_void*)(w_dst + OS + buf * 16384 + kq * 4096), 16, 0, 0);
        }
    };

    // ---- consumer: tr-read addresses.  256-column images: element (k-row 8 h + q, column 32 b + 16 mh + 4 p) at byte
    //      (8 h + q) 512 + 64 (b ^ q) + 32 mh + 8 p; 128-column image: (8 h + q) 256 + 64 (b ^ q) + 32 mh + 8 p
    const int fh = lane >> 5, fmh = (lane >> 4) & 1, fq = (lane >> 2) & 3, fp = lane & 3;
    const unsigned lds_b = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)lds;
    const unsigned in512 = (unsigned)((8 * fh + fq) * 512 + 32 * fmh + 8 * fp), in256 = (unsigned)((8 * fh + fq) * 256 + 32 * fmh + 8 * fp);
    unsigned av[4], bv[NCB];
    bool b_small[NCB];                      // this wave's column block cb lives in the 128-column image
#pragma unroll
    for (int i = 0; i < 4; ++i) av[i] = lds_b + OA + in512 + 64u * (unsigned)((4 * wr + i) ^ fq);
#pragma unroll
    for (int c = 0; c < NCB; ++c) {
        const int jb = wc * NCB + c;                                   // 32-column block of the B tile
        b_small[c] = NCB == 3 && jb < 4;
        if (b_small[c]) bv[c] = lds_b + OS + in256 + 64u * (unsigned)(jb ^ fq);
        else            bv[c] = lds_b + OB + in512 + 64u * (unsigned)((jb - (NCB == 3 ? 4 : 0)) ^ fq);
    }

    f32x16 acc[4][NCB];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int c = 0; c < NCB; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][c][r] = 0.f;

#pragma unroll
    for (int kq = 0; kq < 4; ++kq) stage(0, 0, kq);
#pragma unroll
    for (int kq = 0; kq < 3; ++kq) stage(1, 1, kq);
    if (seg_wave) asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    PP_BAR();
    if (wr == 1) PP_BAR();

    bf16x4 al[4], ah[4], bl[NCB], bh[NCB];
#define PP_FRAG(l, h) bf16x8{l[0], l[1], l[2], l[3], h[0], h[1], h[2], h[3]}
    auto phase = [&](auto bufc, auto ksc, int q, bool on) {
        constexpr int BUF = decltype(bufc)::value, KS = decltype(ksc)::value;
        constexpr int O5 = BUF * 32768 + KS * 8192, O2 = BUF * 16384 + KS * 4096;
#pragma unroll
        for (int i = 0; i < 4; ++i) { PP_TR(al[i], av[i], O5); PP_TR(ah[i], av[i], O5 + 2048); }
#pragma unroll
        for (int c = 0; c < NCB; ++c) {
            if (b_small[c]) { PP_TR(bl[c], bv[c], O2); PP_TR(bh[c], bv[c], O2 + 1024); }
            else            { PP_TR(bl[c], bv[c], O5); PP_TR(bh[c], bv[c], O5 + 2048); }
        }
        if constexpr (KS == 0) stage(q + 1, BUF ^ 1, 3);
        else                   stage(q + 2, BUF, KS - 1);
        if constexpr (NCB == 3)
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(al[0]), "+v"(ah[0]), "+v"(al[1]), "+v"(ah[1]), "+v"(al[2]), "+v"(ah[2]),
                         "+v"(al[3]), "+v"(ah[3]), "+v"(bl[0]), "+v"(bh[0]), "+v"(bl[1]), "+v"(bh[1]), "+v"(bl[NCB - 1]),
                         "+v"(bh[NCB - 1]));
        else
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(al[0]), "+v"(ah[0]), "+v"(al[1]), "+v"(ah[1]), "+v"(al[2]), "+v"(ah[2]),
                         "+v"(al[3]), "+v"(ah[3]), "+v"(bl[0]), "+v"(bh[0]), "+v"(bl[1]), "+v"(bh[1]));
        if constexpr (KS == 3) {          // k-tile q + 1 complete (this wave's share)
            if (seg_wave) asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        }
        PP_BAR();
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int c = 0; c < NCB; ++c) {
            if (!on && y256 && !b_small[c]) continue;                  // h_prev products of the step without a predecessor
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i][c] = pp_mfma(PP_FRAG(al[i], ah[i]), PP_FRAG(bl[c], bh[c]), acc[i][c]);
        }
        __builtin_amdgcn_s_setprio(0);
        PP_BAR();
    };
    auto ktile = [&](auto bufc, int q) {
        const int k0 = kbeg + 64 * q;
        const bool on = !(k0 >= ex_lo && k0 < ex_hi);
        phase(bufc, std::integral_constant<int, 0>{}, q, on);
        phase(bufc, std::integral_constant<int, 1>{}, q, on);
        phase(bufc, std::integral_constant<int, 2>{}, q, on);
        phase(bufc, std::integral_constant<int, 3>{}, q, on);
    };
    for (int q = 0; q < total; q += 2) {
        ktile(std::integral_constant<int, 0>{}, q);
        ktile(std::integral_constant<int, 1>{}, q + 1);
    }
    if (wr == 0) PP_BAR();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#undef PP_FRAG
#pragma unroll
    for (int c = 0; c < NCB; ++c) {
        const int jb = wc * NCB + c;
        float* out = b_small[c] ? o128 : o256;
        const int ldo = b_small[c] ? g.nx : lo256;
        const int col = 32 * (b_small[c] ? jb : jb - (NCB == 3 ? 4 : 0)) + (lane & 31);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wr * 128 + 32 * i + acc_row(r, lane);
                atomicAdd(out + (size_t)row * ldo + col, acc[i][c][r]);
            }
    }
}

#ifdef LOB_PP_DIAG
// Diagnostic: the operand DMA of gemm_nt_pp_kernel ALONE -- same tile walk, same source addresses, same 1-KB
// global_load_lds_dwordx4 instructions into LDS, but no barriers, no reads, no MFMAs, DEPTH instructions in flight per wave.
// What the L2 -> LDS path delivers for this traffic mix (A once from HBM and once from L2 / MALL, B from L2) is the floor
// of any 256 x 256 tiling of these GEMMs (tools/pp_bench.py dma).
template <int DEPTH>
__global__ __launch_bounds__(512, 2) void dma_probe_kernel(PPArgs g) {
    __shared__ __attribute__((aligned(1024))) unsigned char lds[4 * PP_OP];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int ntn = g.N >> 8, ntm = g.M >> 8;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, nslot = gridDim.x >> 3;
    const int panels = (ntm - xcd + 7) / 8, ntile = panels * ntn;
    if (slot >= ntile) return;
    const int nk = g.K >> 6;
    const int l3 = lane >> 3, l7 = lane & 7;
    unsigned asrc[2], bsrc[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int ra = wr * 128 + wc * 16 + j * 8 + l3;
        asrc[j] = 2u * (unsigned)(ra * g.lda + ((l7 ^ ((ra >> 1) & 7)) << 3));
        const int rb = wave * 16 + j * 8 + l3;
        bsrc[j] = 2u * (unsigned)(rb * g.ldw + ((l7 ^ ((rb >> 1) & 7)) << 3));
    }
    const size_t a_hi_b = (size_t)64 * g.lda * 2, b_h1_b = (size_t)128 * g.ldw * 2;
    unsigned char* const dst = lds + wave * 16384;
    int n = 0;
    for (int it = slot; it < ntile; it += nslot) {
        const int m0 = ((it / ntn) * 8 + xcd) << 8, n0 = (it % ntn) << 8;
        const char* pa = reinterpret_cast<const char*>(g.A) + (size_t)m0 * g.lda * 2;
        const char* pb = reinterpret_cast<const char*>(g.W) + (size_t)n0 * g.ldw * 2;
        for (int kt = 0; kt < nk; ++kt, pa += 128, pb += 128) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                __builtin_amdgcn_global_load_lds((gbl_cvoid*)(pa + asrc[j]), (lds_void*)(dst + ((n++ & 15) << 10)), 16, 0, 0);
                __builtin_amdgcn_global_load_lds((gbl_cvoid*)(pa + a_hi_b + asrc[j]), (lds_void*)(dst + ((n++ & 15) << 10)), 16, 0, 0);
                __builtin_amdgcn_global_load_lds((gbl_cvoid*)(pb + bsrc[j]), (lds_void*)(dst + ((n++ & 15) << 10)), 16, 0, 0);
                __builtin_amdgcn_global_load_lds((gbl_cvoid*)(pb + b_h1_b + bsrc[j]), (lds_void*)(dst + ((n++ & 15) << 10)), 16, 0, 0);
            }
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DEPTH) : "memory");
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
#endif

inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

inline int pp_grid(int M, int N) {          // persistent: one workgroup per CU, a multiple of 8 for the XCD map
    const long tiles = (long)(M >> 8) * (N >> 8);
    long gsz = 256;
    if (gsz > tiles) gsz = ((tiles + 7) / 8) * 8;
    // an XCD's panels may be fewer than its share of workgroups when M is small: the surplus exits at once
    return (int)gsz;
}

}  // namespace

// Shapes the ping-pong NT kernels take (checked again by the entry points).
bool lob_pp_nt_ok(int M, int N, int K) {
    return M > 0 && (M % 256) == 0 && (N % 256) == 0 && N >= 256 && (K % 128) == 0 && K >= 256;
}

// Row-major C[M, N] = A[M, K] Wt[N, K]^T (bf16 operands, bf16 or fp32 C, optional dropout mask).  Preconditions beyond
// lob_pp_nt_ok (checked by the caller, lob_gemm_nt_bf16): 16-B aligned bases, lda % 8 == 0, ldw % 8 == 0, ldc % 4 == 0.
int lob_gemm_nt_pp(const void* A, int lda, const void* Wt, int ldw, void* C, int ldc, int M, int N, int K, int out_bf16,
                   float drop_p, uint64_t seed, hipStream_t s) {
    if (!lob_pp_nt_ok(M, N, K)) return LOB_E_SHAPE;
    // 32-bit lane offsets of the DMA sources
    if ((long)256 * lda * 2 >= (1L << 31) || (long)256 * ldw * 2 >= (1L << 31)) return LOB_E_SHAPE;
    PPArgs g{(const __bf16*)A, (const __bf16*)Wt, C, nullptr, lda, ldw, ldc, M, N, K, 0, 0, 0, out_bf16, drop_p, seed};
    if (lob_variant(LOB_VAR_GEMM_PP) & 8) hipLaunchKernelGGL(gemm_nt_ring_kernel, dim3((unsigned)pp_grid(M, N)), dim3(512), 0, s, g);
    else {
        const dim3 gr((unsigned)pp_grid(M, N)), bl(512);
#ifdef LOB_PP_DIAG      // garbage-result instantiations: diagnostic builds only (tools/h256_ablate.sh, ABL_SRC=gemm_pp ABL_DEF=LOB_PP_DIAG)
        if (lob_variant(LOB_VAR_GEMM_PP) & 1024) {                 // the operand DMA alone (tools/pp_bench.py dma)
            if (lob_variant(LOB_VAR_GEMM_PP) & 2048) hipLaunchKernelGGL(dma_probe_kernel<24>, gr, bl, 0, s, g);
            else                                     hipLaunchKernelGGL(dma_probe_kernel<8>, gr, bl, 0, s, g);
            LOB_CHECK_LAUNCH();
            return 0;
        }
#define LOB_PP_ABL_CASES                                                                        \
            case 1: hipLaunchKernelGGL((gemm_nt_pp_kernel<0, 1>), gr, bl, 0, s, g); break;      \
            case 2: hipLaunchKernelGGL((gemm_nt_pp_kernel<0, 2>), gr, bl, 0, s, g); break;      \
            case 3: hipLaunchKernelGGL((gemm_nt_pp_kernel<0, 3>), gr, bl, 0, s, g); break;      \
            case 4: hipLaunchKernelGGL((gemm_nt_pp_kernel<0, 4>), gr, bl, 0, s, g); break;      \
            case 5: hipLaunchKernelGGL((gemm_nt_pp_kernel<0, 5>), gr, bl, 0, s, g); break;      \
            case 6: hipLaunchKernelGGL((gemm_nt_pp_kernel<0, 6>), gr, bl, 0, s, g); break;      \
            case 7: hipLaunchKernelGGL((gemm_nt_pp_kernel<0, 7>), gr, bl, 0, s, g); break;
        const int abl = (lob_variant(LOB_VAR_GEMM_PP) >> 4) & 7;   // ablations (tools/pp_bench.py abl)
#else
#define LOB_PP_ABL_CASES
        const int abl = 0;      // the product library holds no kernel that computes garbage: bits 16..64 / 1024 / 2048 are ignored
#endif
        switch (abl) {
            LOB_PP_ABL_CASES
            default:
                if (lob_variant(LOB_VAR_GEMM_PP) & 512) { hipLaunchKernelGGL(gemm_nt_pp16_kernel, gr, bl, 0, s, g); break; }
                switch ((lob_variant(LOB_VAR_GEMM_PP) >> 7) & 3) {
                    case 1: hipLaunchKernelGGL((gemm_nt_pp_kernel<0, 0, 1>), gr, bl, 0, s, g); break;
                    case 2: hipLaunchKernelGGL((gemm_nt_pp_kernel<0, 0, 2>), gr, bl, 0, s, g); break;
                    default: hipLaunchKernelGGL((gemm_nt_pp_kernel<0>), gr, bl, 0, s, g);
                }
        }
#undef LOB_PP_ABL_CASES
    }
    LOB_CHECK_LAUNCH();
    return 0;
}

// The gate GEMM: bf16 fragment-order P[T*Bp, D*4H] = X[T*Bp, K] W_ih[D*4H, K]^T + bias.  N = D*4H <= 2048.
int lob_gate_gemm_pp(const void* X, int ldx, const void* Wih, const float* bias, void* P, int T, int Bp, int H, int D, int K,
                     hipStream_t s) {
    const int M = T * Bp, N = D * 4 * H;
    if (!lob_pp_nt_ok(M, N, K) || N > 2048 || (H % 32) || (Bp % 32)) return LOB_E_SHAPE;
    if ((long)256 * ldx * 2 >= (1L << 31)) return LOB_E_SHAPE;
    PPArgs g{(const __bf16*)X, (const __bf16*)Wih, P, bias, ldx, K, N, M, N, K, T, Bp, H, 1, 0.f, 0};
    hipLaunchKernelGGL((gemm_nt_pp_kernel<1>), dim3((unsigned)pp_grid(M, N)), dim3(512), 0, s, g);
    LOB_CHECK_LAUNCH();
    return 0;
}

bool lob_pp_tn_ok(int M, int N, int Kc) {
    return (M % 256) == 0 && (N % 256) == 0 && M >= 256 && N >= 256 && (Kc % 128) == 0 && Kc >= 256;
}

// C[M, N] (fp32, += by atomics) = A[Kc, M]^T B'[Kc, N], B'[k] = B[k + shift] outside [ex_lo, ex_hi), 0 inside.
// shift != 0 needs ex_hi - ex_lo and ex_lo multiples of 64 covering every k with k + shift outside [0, Kc).
int lob_gemm_tn_pp(const void* A, int lda, const void* B, int ldb, float* C, int ldc, int M, int N, int Kc, int shift,
                   int ex_lo, int ex_hi, hipStream_t s) {
    if (!lob_pp_tn_ok(M, N, Kc)) return LOB_E_SHAPE;
    if ((ex_lo % 64) || (ex_hi % 64)) return LOB_E_SHAPE;
    if ((long)16 * lda * 2 >= (1L << 31) || (long)16 * ldb * 2 >= (1L << 31)) return LOB_E_SHAPE;
    const int tiles = (M >> 8) * (N >> 8);
    int nchunk = (256 + tiles - 1) / tiles;                  // one workgroup per CU
    long kchunk = ((long)Kc + nchunk - 1) / nchunk;
    kchunk = ((kchunk + 127) / 128) * 128;
    if (kchunk < 1024) kchunk = 1024;
    if (kchunk > Kc) kchunk = Kc;
    nchunk = (int)((Kc + kchunk - 1) / kchunk);
    const int nchunk8 = ((nchunk + 7) / 8) * 8;
    TNPPArgs g{(const __bf16*)A, (const __bf16*)B, C, lda, ldb, ldc, M, N, Kc, (int)kchunk, tiles, shift, ex_lo, ex_hi};
    if (lob_variant(LOB_VAR_GEMM_PP) & 8) hipLaunchKernelGGL(gemm_tn_ring_kernel, dim3((unsigned)(tiles * nchunk8)), dim3(512), 0, s, g);
    else                                  hipLaunchKernelGGL(gemm_tn_pp_kernel, dim3((unsigned)(tiles * nchunk8)), dim3(512), 0, s, g);
    LOB_CHECK_LAUNCH();
    return 0;
}

// dW_ih (D*4H x nx) and dW_hh (D x 4H x H) of one H = 256 layer from one pass over dP (outputs zeroed / accumulated into
// by the caller; fp32 atomics).  nx in {256, 512}, D = 2, Bp % 64 == 0, (T * Bp) % 128 == 0, T >= 2.
bool lob_dw_pp_ok(int T, int Bp, int H, int D, int nx) {
    return H == 256 && D == 2 && (nx == 256 || nx == 512) && T >= 2 && (Bp % 64) == 0 && ((long)T * Bp) % 128 == 0 &&
           (long)T * Bp >= 128;
}
int lob_lstm_dw_pp(const void* dP, int ldp, const void* X, int ldx, int nx, const void* Y, int ldy, float* dWih, float* dWhh,
                   int T, int Bp, int H, int D, hipStream_t s) {
    if (!lob_dw_pp_ok(T, Bp, H, D, nx)) return LOB_E_SHAPE;
    if ((long)16 * ldp * 2 >= (1L << 31)) return LOB_E_SHAPE;
    const long Kc = (long)T * Bp;
    int nchunk = 16;                                             // 16 tiles x 16 chunks = one workgroup per CU
    long kchunk = (Kc + nchunk - 1) / nchunk;
    kchunk = ((kchunk + 127) / 128) * 128;
    if (kchunk < 1024) kchunk = 1024;
    if (kchunk > Kc) kchunk = Kc;
    nchunk = (int)((Kc + kchunk - 1) / kchunk);
    const int nchunk8 = ((nchunk + 7) / 8) * 8;
    DWPPArgs g{(const __bf16*)dP, (const __bf16*)X, (const __bf16*)Y, dWih, dWhh, ldp, ldx, ldy, nx, H, T, Bp, (int)kchunk};
    const dim3 grid((unsigned)(16 * nchunk8)), block(512);
    if (nx == 512) hipLaunchKernelGGL(lstm_dw_pp_kernel<3>, grid, block, 0, s, g);
    else           hipLaunchKernelGGL(lstm_dw_pp_kernel<2>, grid, block, 0, s, g);
    LOB_CHECK_LAUNCH();
    return 0;
}
