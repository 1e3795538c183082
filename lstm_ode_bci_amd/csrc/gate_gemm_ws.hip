// Weight-stationary input-side gate GEMM for the mixed path at H = 128 (nn.LSTM's W_ih x_t + b_ih + b_hh,
// 04_lstm_model.py:181-188, 211, under autocast 04:487):
//     P[T*Bp, D*512] = X[T*Bp, K] * W_ih[D*512, K]^T + bias,   K = 128 (layer 0) or 256 (layers 1..),
// bf16 operands, fp32 accumulate (v_mfma_f32_32x32x16_bf16), P written as bf16 in the accumulator-fragment order the
// recurrent kernel consumes (lob.h).
//
// Why a second kernel next to gemm_nt_dma_kernel<1,256,256>: with K = 256 an output tile has only four 64-deep k-tiles,
// and the tiled kernel re-stages the 128-KB W tile through L2 -> LDS for every 256 rows -- 6.3 GB cross L2 -> LDS for
// 2.7 GB of HBM traffic, and that path (about 7 TB/s with one k-tile in flight), not HBM, set its 0.85-0.96 ms.  Here
// the weights never move: a workgroup (8 waves) owns the 512 gate columns of ONE direction, each wave keeps the B
// fragments of its 64 columns for the whole contraction in registers (2 x K/16 x 4 = 128 VGPRs at K = 256), and only
// the activations stream: 64-row tiles (32 KB at K = 256) HBM -> LDS by global_load_lds_dwordx4 through a 4-slot ring,
// THREE tiles in flight, one workgroup barrier per 64 rows (64 MFMAs per wave) instead of one per 64-deep k-tile.
// L2 -> LDS traffic is 2 x the bytes of X (once per direction), 1.1 GB.  The two workgroups that share a row tile (one
// per direction) sit 8 apart in blockIdx -- same XCD under round-robin placement -- and walk the same tile sequence,
// so the second read of a tile is an L2 hit (speed only, never correctness).
//
// vmcnt bookkeeping (VMEM operations of a wave retire in order).  Per iteration q a wave issues, after the barrier,
// NDMA DMA instructions for tile q + LA and, after the MFMAs, NST = 8 fragment stores of tile q.  When iteration q
// starts, the operations younger than DMA(q) (issued in iteration q - LA) are LA x NST stores and (LA - 1) x NDMA
// DMAs: one counted s_waitcnt vmcnt(VM_STEADY) serves every steady-state iteration.  The first LA iterations wait
// with the smaller VM_PRO (no stores older than the awaited DMA exist yet: waiting longer is safe, shorter is not);
// past the end the DMA of the last tile is re-issued into a slot nobody reads again, so the count stays valid.
//
// LDS image of a slot: [64 rows][K] bf16, rows of 2K bytes, lane-linear as the DMA writes it (1 KB per wave
// instruction = 2 rows at K = 256, 4 at K = 128).  16-B chunk c of row r is stored at chunk slot c ^ (r & 15): applied
// on the per-lane GLOBAL source address of the DMA and again on the fragment read (cdna guide rule 21); the 16 lanes a
// ds_read_b128 services together hold 16 different r & 15, i.e. the 16 different 16-B pieces of the 256-B bank row.
#include "lob_common.h"
// H = 256: eight column groups re-read every activation tile from L2, and P (4.3 GB, written once) would evict
// them: non-temporal fragment stores there (K = 512: 2.33 -> 2.31 ms, K = 256: 1.15 -> 1.12)
#ifndef LOB_NT_WSP
#define LOB_NT_WSP true
#endif

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void gbl_cvoid;

struct WSArgs {
    const __bf16* A; const __bf16* W; const float* bias; __bf16* P;
    int lda, M, T, Bp, D;
    int ldc, out_bf16;           // EPI = 1 (row-major C): leading dimension in elements, bf16 or fp32 C
};

// EPI = 0: the gate GEMM -- bf16 fragment-order P, bias.  K <= 256: 64 columns per wave (512 per workgroup), 64-row tiles.
//          K = 512 (layers 1.. at H = 256, the reference's checkpoint size, 04_lstm_model.py:877): the B fragments of 64
//          columns would be the whole 256-register budget, so a wave keeps 32 columns (128 VGPRs), a workgroup 256, and
//          the tiles are 32 rows (32 KB per ring slot as at K = 256).  Every MFMA then needs its own 1-KB A fragment from
//          LDS (the 64-column shape shares one between two), i.e. LDS reads run at the MFMA rate -- still 1.7 x faster
//          than re-staging the weight tile per 256 rows.  HID = hidden size: only the fragment-order address
//          ([d][t][bt][w HID/32][gate 4][q pair 2][lane 64][8], include/lob.h) depends on it.
// EPI = 1 (bf16 C) / 2 (fp32 C): a plain row-major C[M, N] = A W^T (no bias) for narrow contractions -- the attention pooling's
//          dV = dPreU W1 (K = 128, N = 256; 04_lstm_model.py:118 in the backward of 04:482-512): 32 columns per wave,
//          256 per workgroup, `D` = number of 256-column groups.  The MFMA runs with its operands SWAPPED (the weights as
//          the A operand), so a lane holds 4 consecutive columns of one row per accumulator quad: 8-B (bf16) / 16-B
//          (fp32) stores -- and, as in the fragment epilogue, exactly 8 store instructions per wave and tile.
template <int K, int EPI, int HID = 128>
__global__ __launch_bounds__(512, 2) void gate_gemm_ws_kernel(WSArgs g) {
    constexpr int MT = K == 512 ? 32 : 64;          // rows per tile
    constexpr int MI = MT / 32;                     // 32-row blocks per tile
    constexpr int NCB = (EPI == 0 && K <= 256) ? 2 : 1;   // 32-column blocks per wave
    constexpr int WGN = 8 * 32 * NCB;               // columns per workgroup
    constexpr int NSLOT = K == 128 ? 6 : 4, LA = NSLOT - 1;
    constexpr int ROWB = 2 * K, SLOTB = MT * ROWB;  // bytes per LDS row / slot
    constexpr int RPI = 1024 / ROWB;                // rows per DMA instruction
    constexpr int NDMA = MT / RPI / 8;              // DMA instructions per wave per tile
    constexpr int NST = EPI == 0 ? MI * NCB * 2 : 8;   // stores per wave per tile
    static_assert(EPI == 0 || K <= 256, "row-major epilogue: K in {128, 256}");
    constexpr int VM_STEADY = LA * NST + (LA - 1) * NDMA, VM_PRO = (LA - 1) * NDMA;
    constexpr int KS = K / 16;                      // MFMA k-steps
    static_assert(VM_STEADY < 64 && NDMA >= 1, "vmcnt is a 6-bit counter");
    __shared__ __attribute__((aligned(1024))) unsigned char ring[NSLOT * SLOTB];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r31 = lane & 31, hi = lane >> 5;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int d = slot % g.D, pair = slot / g.D, npair = (gridDim.x >> 3) / g.D;
    const int ntile = (g.M + MT - 1) / MT;
    const int panels = (ntile - xcd + 7) / 8;       // row tiles owned by this XCD: mt = 8 p + xcd
    if (pair >= panels) return;
    const int total = (panels - pair + npair - 1) / npair;

    // ---- stationary B fragments: wreg[cb][s] = W[d*WGN + 32 NCB wv + 32 cb + r31][16 s + 8 hi .. + 7]
    bf16x8 wreg[NCB][KS];
    float bv[NCB];
    const int ncol0 = d * WGN + 32 * NCB * wv;
    {
        const __bf16* wb = g.W + (size_t)(ncol0 + r31) * K + 8 * hi;
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) {
#pragma unroll
            for (int s = 0; s < KS; ++s) wreg[cb][s] = *reinterpret_cast<const bf16x8*>(wb + (size_t)(32 * cb) * K + 16 * s);
            bv[cb] = (EPI == 0 && g.bias) ? g.bias[ncol0 + 32 * cb + r31] : 0.f;
        }
    }
    // make the weights opaque: hipcc otherwise feels free to re-load them inside the loop (256 VGPRs are tight)
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
        for (int s = 0; s < KS; ++s) asm volatile("" : "+v"(wreg[cb][s]));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    // ---- DMA source addressing: instruction j of this wave covers tile rows (8 wv... ) see RPI
    const int drow = lane / (ROWB / 16), dsl = lane % (ROWB / 16);        // row inside the instruction, chunk slot
    unsigned dsrc[NDMA];                                                  // element offsets inside a tile (rows clamped later)
    int drw[NDMA];
#pragma unroll
    for (int j = 0; j < NDMA; ++j) {
        const int r = (wv * NDMA + j) * RPI + drow;                        // row inside the tile (0..63)
        drw[j] = r;
        dsrc[j] = (unsigned)((dsl ^ (r & 15)) * 8);
    }
    auto issue = [&](int u) {
        const int ut = u < total ? u : total - 1;
        const int m0 = ((pair + npair * ut) * 8 + xcd) * MT;
        unsigned char* dst = ring + (u % NSLOT) * SLOTB + wv * NDMA * 1024;
#pragma unroll
        for (int j = 0; j < NDMA; ++j) {
            int r = m0 + drw[j];
            r = r < g.M ? r : g.M - 1;                                      // rows past the end are never stored
            const __bf16* src = g.A + (size_t)r * g.lda + dsrc[j];
            __builtin_amdgcn_global_load_lds((gbl_cvoid*)src, (lds_void*)(dst + j * 1024), 16, 0, 0);
        }
    };
#pragma unroll 1
    for (int u = 0; u < LA; ++u) issue(u);

    // ---- fragment read addressing: row 32 i + r31, chunk 2 s + hi at slot (2 s + hi) ^ (r31 & 15)
    const unsigned ring_b = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)ring;
    unsigned aoff[8];
#pragma unroll
    for (int s = 0; s < 8; ++s) aoff[s] = (unsigned)(r31 * ROWB + (((2 * s + hi) ^ (r31 & 15)) * 16));

    f32x16 acc[MI][NCB];
    const int NBT = EPI == 0 ? g.Bp >> 5 : 1;

    for (int q = 0; q < total; ++q) {
        if (q < LA) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(VM_PRO) : "memory");
        else        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(VM_STEADY) : "memory");
        __builtin_amdgcn_s_barrier();
        issue(q + LA);                                  // refills the slot read in iteration q - 1

        const unsigned sb = ring_b + (unsigned)((q % NSLOT) * SLOTB);
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][cb][r] = 0.f;
        // fragment reads as inline asm (compiler-visible LDS reads of a DMA target may get an s_waitcnt vmcnt(0) in
        // front, which would drain the ring); LDS operations return in order: each wait names the registers it frees
#define LOB_RD(dstv, ADDR, OFF) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=&v"(dstv) : "v"(ADDR), "n"(OFF) : "memory")
        bf16x8 a0, a1, n0, n1;
        if constexpr (MI == 1) {
            // one 32-row block: one fragment per k-step, TWO k-steps ahead (a0 = step s, a1 = s + 1, n0 = s + 2)
            LOB_RD(a0, sb + aoff[0], 0);
            LOB_RD(a1, sb + aoff[1], 0);
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                if (s + 2 < KS) {
                    LOB_RD(n0, sb + aoff[(s + 2) & 7], ((s + 2) >> 3) * 256);
                    asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(a0));
                } else if (s + 1 < KS) {
                    asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(a0));
                } else {
                    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a0));
                }
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, wreg[0][s], acc[0][0], 0, 0, 0);
                a0 = a1; a1 = n0;
            }
        } else {
        LOB_RD(a0, sb + aoff[0], 0);
        LOB_RD(a1, sb + aoff[0], 32 * ROWB);
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            if (s + 1 < KS) {
                LOB_RD(n0, sb + aoff[(s + 1) & 7], ((s + 1) >> 3) * 256);
                LOB_RD(n1, sb + aoff[(s + 1) & 7], ((s + 1) >> 3) * 256 + 32 * ROWB);
                asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(a0), "+v"(a1));
            } else {
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a0), "+v"(a1));
            }
            if constexpr (EPI == 0) {
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, wreg[0][s], acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, wreg[1][s], acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, wreg[0][s], acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, wreg[1][s], acc[1][1], 0, 0, 0);
            } else {        // operands swapped: D[n][m] -- this lane: row m = r31, columns acc_row(r)
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wreg[0][s], a0, acc[0][0], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wreg[0][s], a1, acc[1][0], 0, 0, 0);
            }
            a0 = n0; a1 = n1;
        }
        }
#undef LOB_RD

        // ---- epilogue: + bias, bf16, fragment order [d][t][bt][w HID/32][gate 4][q pair 2][lane 64][8]
        const int m0 = ((pair + npair * q) * 8 + xcd) * MT;
        if constexpr (EPI >= 1) {
            // row-major C: lane = row m0 + 32 i + r31; accumulator quad qd = columns ncol0 + 8 qd + 4 hi .. + 3
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int row = m0 + 32 * i + r31;
                const bool ok = m0 + 32 * i < g.M;            // wave-uniform (M % 32 == 0): see below
                const size_t o = (size_t)row * g.ldc + ncol0 + 4 * hi;
#pragma unroll
                for (int qd = 0; qd < 4; ++qd) {
                    if (!ok) continue;
                    if constexpr (EPI == 1) {
                        typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
                        bf16x4 v = {(__bf16)acc[i][0][4 * qd], (__bf16)acc[i][0][4 * qd + 1], (__bf16)acc[i][0][4 * qd + 2],
                                    (__bf16)acc[i][0][4 * qd + 3]};
                        *reinterpret_cast<bf16x4*>(g.P + o + 8 * qd) = v;
                    } else {
                        f32x4 v = {acc[i][0][4 * qd], acc[i][0][4 * qd + 1], acc[i][0][4 * qd + 2], acc[i][0][4 * qd + 3]};
                        *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(g.P) + o + 8 * qd) = v;
                    }
                }
            }
            continue;
        }
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            const int mrow = m0 + 32 * i;
            const bool ok = mrow < g.M;
            const int mr = ok ? mrow : 0;
            const int t = mr / g.Bp, bt = (mr - t * g.Bp) >> 5;
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb) {
                // column block ncol0 + 32 cb of the [D][gate 4][HID] gate axis
                const int nc = ncol0 + 32 * cb;
                const int dd = nc / (4 * HID), gate = (nc % (4 * HID)) / HID, wq = (nc % HID) >> 5;
                const size_t fo = ((((size_t)(dd * g.T + t) * NBT + bt) * (HID / 32) + wq) * 4 + gate) * 1024;
                __bf16* dst = g.P + fo + lane * 8;
#pragma unroll
                for (int pq = 0; pq < 2; ++pq) {
                    bf16x8 v;
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = (__bf16)(acc[i][cb][8 * pq + e] + bv[cb]);
                    // rows past the end (M % 64 == 32) exist only in the LAST iteration of the workgroup that owns the
                    // last tile: skipping their stores changes no later vmcnt count
                    if (ok) {
                        if constexpr (LOB_NT_WSP && HID == 256) __builtin_nontemporal_store(v, reinterpret_cast<bf16x8*>(dst + pq * 512));
                        else *reinterpret_cast<bf16x8*>(dst + pq * 512) = v;
                    }
                }
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

}  // namespace

// Internal entry point used by lob_gate_gemm_x_bf16 (gemm_bf16.hip).  Preconditions checked by the caller: X, W bf16,
// P bf16 fragment order, (H, K) in {(128, 128), (128, 256), (256, 256), (256, 512)}, ldx % 8 == 0, 16-B aligned bases,
// Bp % 32 == 0.
int lob_gate_gemm_ws(const void* X, int ldx, const void* Wih, const float* bias, void* P, int T, int Bp, int H, int D, int K,
                     hipStream_t s) {
    const int M = T * Bp;
    const int mt = K == 512 ? 32 : 64;
    const int wgn = K == 512 ? 256 : 512;            // gate columns per workgroup
    const int ncg = D * 4 * H / wgn;                 // column groups: 1 per direction at H = 128; 4 (K = 256) / 8 (K = 512) at H = 256
    const int ntile = (M + mt - 1) / mt;
    int npx = (ntile + 7) / 8;                       // row walkers per XCD, one workgroup per CU at most
    const int cap = 32 / ncg;
    if (npx > cap) npx = cap;
    WSArgs g{(const __bf16*)X, (const __bf16*)Wih, bias, (__bf16*)P, ldx, M, T, Bp, ncg, 0, 0};
    const dim3 grid((unsigned)(8 * ncg * npx)), block(512);
    if (H == 256) {
        if (K == 512) hipLaunchKernelGGL((gate_gemm_ws_kernel<512, 0, 256>), grid, block, 0, s, g);
        else          hipLaunchKernelGGL((gate_gemm_ws_kernel<256, 0, 256>), grid, block, 0, s, g);
    } else {
        if (K == 256) hipLaunchKernelGGL((gate_gemm_ws_kernel<256, 0>), grid, block, 0, s, g);
        else          hipLaunchKernelGGL((gate_gemm_ws_kernel<128, 0>), grid, block, 0, s, g);
    }
    LOB_CHECK_LAUNCH();
    return 0;
}

// Row-major C[M, N] = A[M, K] W[N, K]^T, bf16 operands, K in {128, 256}, N a multiple of 256 (<= 1024), M % 32 == 0, no
// bias / activation; C bf16 or fp32.  Used by lob_gemm_nt_bf16 for the narrow-contraction GEMMs of the backward.
int lob_gemm_nt_ws(const void* A, int lda, const void* W, void* C, int ldc, int M, int N, int K, int out_bf16, hipStream_t s) {
    const int ncg = N / 256;
    const int ntile = (M + 63) / 64;
    int npx = (ntile + 7) / 8;
    const int cap = 32 / ncg;
    if (npx > cap) npx = cap;
    WSArgs g{(const __bf16*)A, (const __bf16*)W, nullptr, (__bf16*)C, lda, M, 1, 32, ncg, ldc, out_bf16};
    const dim3 grid((unsigned)(8 * ncg * npx)), block(512);
    if (K == 256) { if (out_bf16) hipLaunchKernelGGL((gate_gemm_ws_kernel<256, 1>), grid, block, 0, s, g);
                    else          hipLaunchKernelGGL((gate_gemm_ws_kernel<256, 2>), grid, block, 0, s, g); }
    else          { if (out_bf16) hipLaunchKernelGGL((gate_gemm_ws_kernel<128, 1>), grid, block, 0, s, g);
                    else          hipLaunchKernelGGL((gate_gemm_ws_kernel<128, 2>), grid, block, 0, s, g); }
    LOB_CHECK_LAUNCH();
    return 0;
}
