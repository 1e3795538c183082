// bf16-input / fp32-accumulate MFMA GEMMs (v_mfma_f32_32x32x16_bf16) for the mixed-precision
// mode (BASELINE.json configs[2]: "bf16 gate-GEMMs + fp32 recurrence"; the reference's own
// training loop runs the model under autocast, 04_lstm_model.py:487).
//
// Operands live in HBM as fp32 (activations, weights) or bf16 (dP written by the BPTT kernel)
// and are rounded to bf16 (RNE, v_cvt_pk_bf16_f32) while being staged into LDS; products are
// exact, accumulation is fp32.  At 16x the fp32 MFMA rate these kernels are HBM-bound
// (arithmetic intensity ~110 FLOP/B vs a machine balance of ~400): what matters is coalesced
// 16-B global accesses and enough of them in flight, not the MFMA schedule.
//
// Tiles: 128x128 output per 256-thread workgroup (4 waves as 2x2, 2x2 32x32 accumulators each),
// contraction in steps of 64 through double-buffered LDS; LDS rows are 64 bf16 + 16 B pad =
// 144 B = 9 x 16-B slots (odd), so a wave's ds_read_b128 fragment reads are conflict-free.
#include <stdlib.h>
#include "lob_common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

constexpr int TM = 128, TN_ = 128;

__device__ __forceinline__ f32x16 mfma_bf16(bf16x8 a, bf16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

struct NTArgs {
    const void* A; const float* W; const float* bias; float* C;
    int lda, ldw, ldc, M, N, K, act, accumulate;
    int T, Bp, H, D;     // fragment epilogue
    int out_bf16;                  // fragment epilogue: P stored as bf16
    float drop_p; uint64_t seed;   // row-major epilogue: C *= dropout mask of element (row*ldc + col)
    int stagger;                   // LDS-DMA kernel: start delay unit (see the kernel), 0 = none
};

// ---- staging: [128 rows][TKT k] tile, source rows are K-contiguous -------------------------
// LDS rows are TKT bf16 + 16 B pad (TKT=64: 144 B = 9 slots, TKT=32: 80 B = 5 slots; odd -> the
// ds_read_b128 fragment reads of a wave are conflict-free).
template <int TKT>
__device__ __forceinline__ void ld_rows_f32(const float* __restrict__ G, int ld, int row0, int rows,
                                            int k0, int K, int tid, f32x4 (&r)[TKT / 8]) {
    constexpr int TPR = TKT / 4, RPP = 256 / TPR, NP = 128 / RPP;
    const int rr = tid / TPR, c4 = (tid % TPR) * 4;
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        const int row = row0 + rr + RPP * i, k = k0 + c4;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (row < rows && k < K) v = *reinterpret_cast<const f32x4*>(G + (size_t)row * ld + k);
        r[i] = v;
    }
}
template <int TKT>
__device__ __forceinline__ void st_rows_f32(__bf16* S, int tid, const f32x4 (&r)[TKT / 8]) {
    constexpr int TPR = TKT / 4, RPP = 256 / TPR, NP = 128 / RPP, LD = TKT + 8;
    const int rr = tid / TPR, c4 = (tid % TPR) * 4;
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        bf16x4 h = {(__bf16)r[i][0], (__bf16)r[i][1], (__bf16)r[i][2], (__bf16)r[i][3]};
        *reinterpret_cast<bf16x4*>(S + (rr + RPP * i) * LD + c4) = h;
    }
}
template <int TKT>
__device__ __forceinline__ void ld_rows_bf16(const __bf16* __restrict__ G, int ld, int row0, int rows,
                                             int k0, int K, int tid, bf16x8 (&r)[TKT / 16]) {
    constexpr int TPR = TKT / 8, RPP = 256 / TPR, NP = 128 / RPP;
    const int rr = tid / TPR, c8 = (tid % TPR) * 8;
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        const int row = row0 + rr + RPP * i, k = k0 + c8;
        bf16x8 v;
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = (__bf16)0.f;
        if (row < rows && k < K) v = *reinterpret_cast<const bf16x8*>(G + (size_t)row * ld + k);
        r[i] = v;
    }
}
template <int TKT>
__device__ __forceinline__ void st_rows_bf16(__bf16* S, int tid, const bf16x8 (&r)[TKT / 16]) {
    constexpr int TPR = TKT / 8, RPP = 256 / TPR, NP = 128 / RPP, LD = TKT + 8;
    const int rr = tid / TPR, c8 = (tid % TPR) * 8;
#pragma unroll
    for (int i = 0; i < NP; ++i) *reinterpret_cast<bf16x8*>(S + (rr + RPP * i) * LD + c8) = r[i];
}

template <int TKT>
__device__ __forceinline__ void mma_tile(const __bf16* As, const __bf16* Bs, int wr, int wc, int lane,
                                         f32x16 (&acc)[2][2]) {
    constexpr int LD = TKT + 8;
    const __bf16* ap = As + (64 * wr + (lane & 31)) * LD + 8 * (lane >> 5);
    const __bf16* bp = Bs + (64 * wc + (lane & 31)) * LD + 8 * (lane >> 5);
#pragma unroll
    for (int s = 0; s < TKT / 16; ++s) {
        const bf16x8 a0 = *reinterpret_cast<const bf16x8*>(ap + 16 * s);
        const bf16x8 a1 = *reinterpret_cast<const bf16x8*>(ap + 32 * LD + 16 * s);
        const bf16x8 b0 = *reinterpret_cast<const bf16x8*>(bp + 16 * s);
        const bf16x8 b1 = *reinterpret_cast<const bf16x8*>(bp + 32 * LD + 16 * s);
        acc[0][0] = mfma_bf16(a0, b0, acc[0][0]);
        acc[0][1] = mfma_bf16(a0, b1, acc[0][1]);
        acc[1][0] = mfma_bf16(a1, b0, acc[1][0]);
        acc[1][1] = mfma_bf16(a1, b1, acc[1][1]);
    }
}

// PERSISTENT workgroups: the grid is one wave of resident workgroups (3 per CU at TKT = 32), each
// walking its own sequence of output tiles.  Two things a one-tile-per-workgroup grid cannot do:
// the first operand tile of the NEXT output tile is requested before the epilogue of the current
// one (its HBM latency hides behind 64 KB of stores), and a workgroup slot is never idle while
// stores drain at wave exit.  Measured on the K=256 gate GEMM: loads and stores were additive
// (0.67 ms compute + 0.6 loads + 0.75 stores); overlapped they approach the HBM floor.
//
// XCD-aware tile order: workgroups are dealt round-robin over the 8 XCDs, so XCD x owns the row
// panels mt = x (mod 8) and sweeps the N tiles of one panel back to back: an A panel is pulled
// through the fabric once, into one L2.
template <bool A_BF16, int EPI, int TKT>
__global__ __launch_bounds__(256, TKT == 32 ? 3 : 2) void gemm_nt_bf16_kernel(NTArgs g) {
    constexpr int LD = TKT + 8;
    __shared__ __attribute__((aligned(16))) __bf16 lds[2 * 2 * TM * LD];
    __bf16* As = lds;
    __bf16* Ws = lds + 2 * TM * LD;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int ntn = (g.N + TN_ - 1) / TN_, ntm = (g.M + TM - 1) / TM;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, nslot = gridDim.x >> 3;
    const int panels = (ntm - xcd + 7) / 8;            // row panels owned by this XCD
    const int ntile = panels * ntn;                    // tiles of this XCD, panel-major
    const int nk = (g.K + TKT - 1) / TKT;

    f32x4 ra[TKT / 8], rw[TKT / 8];
    bf16x8 rab[TKT / 16];
    auto load = [&](int m0, int n0, int k0) {
        if (A_BF16) ld_rows_bf16<TKT>(reinterpret_cast<const __bf16*>(g.A), g.lda, m0, g.M, k0, g.K, tid, rab);
        else        ld_rows_f32<TKT>(reinterpret_cast<const float*>(g.A), g.lda, m0, g.M, k0, g.K, tid, ra);
        ld_rows_f32<TKT>(g.W, g.ldw, n0, g.N, k0, g.K, tid, rw);
    };
    auto store = [&](int b) {
        if (A_BF16) st_rows_bf16<TKT>(As + b * TM * LD, tid, rab); else st_rows_f32<TKT>(As + b * TM * LD, tid, ra);
        st_rows_f32<TKT>(Ws + b * TM * LD, tid, rw);
    };

    int it = slot;
    if (it >= ntile) return;
    int m0 = ((it / ntn) * 8 + xcd) * TM, n0 = (it % ntn) * TN_;
    load(m0, n0, 0);
    while (true) {
        f32x16 acc[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        store(0);
        __syncthreads();
        int buf = 0;
        for (int kt = 0; kt < nk; ++kt) {
            if (kt + 1 < nk) load(m0, n0, (kt + 1) * TKT);
            mma_tile<TKT>(As + buf * TM * LD, Ws + buf * TM * LD, wr, wc, lane, acc);
            if (kt + 1 < nk) store(buf ^ 1);
            __syncthreads();
            buf ^= 1;
        }
        // request the next tile's first operand tiles BEFORE this tile's stores
        const int nit = it + nslot;
        const bool more = nit < ntile;
        const int cm0 = m0, cn0 = n0;
        if (more) {
            m0 = ((nit / ntn) * 8 + xcd) * TM; n0 = (nit % ntn) * TN_;
            load(m0, n0, 0);
        }
        {
            if (EPI == 0) {
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const int col = cn0 + 64 * wc + 32 * j + (lane & 31);
                        if (col >= g.N) continue;
                        const float bv = g.bias ? g.bias[col] : 0.f;
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int row = cm0 + 64 * wr + 32 * i + acc_row(r, lane);
                            if (row < g.M) {
                                float* dst = g.C + (size_t)row * g.ldc + col;
                                float val = apply_act(acc[i][j][r] + bv, g.act);
                                if (g.drop_p > 0.f) val *= lob_dropout_scale(g.seed, (uint64_t)row * g.ldc + col, g.drop_p);
                                *dst = g.accumulate ? *dst + val : val;
                            }
                        }
                    }
            } else {
                const int NBT = g.Bp >> 5, NW = g.H >> 5, H4 = 4 * g.H;
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int mrow = cm0 + 64 * wr + 32 * i;
                    if (mrow >= g.M) continue;
                    const int t = mrow / g.Bp, bt = (mrow % g.Bp) >> 5;
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const int ncol = cn0 + 64 * wc + 32 * j;
                        if (ncol >= g.N) continue;
                        const int d = ncol / H4, gg = (ncol % H4) / g.H, w = (ncol % g.H) >> 5;
                        const float bv = g.bias ? g.bias[ncol + (lane & 31)] : 0.f;
                        size_t fo = ((((size_t)(d * g.T + t) * NBT + bt) * NW + w) * 4 + gg) * 1024 + lane * 4;
                        if (g.out_bf16) {        // bf16 fragment storage: [gate][q pair][lane][8] -> 16-B stores
                            __bf16* dst = reinterpret_cast<__bf16*>(g.C) + (fo - lane * 4) + lane * 8;
#pragma unroll
                            for (int pq = 0; pq < 2; ++pq) {
                                bf16x8 v;
#pragma unroll
                                for (int e = 0; e < 8; ++e) v[e] = (__bf16)(acc[i][j][8 * pq + e] + bv);
                                *reinterpret_cast<bf16x8*>(dst + pq * 512) = v;
                            }
                        } else {
                            float* dst = g.C + fo;
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
        }
        if (!more) break;
        it = nit;
    }
}


// ------------------------------------------------------------------------------------------
// LDS-DMA variant of the NT GEMM for bf16 operands in HBM (A = dP / dropped layer outputs, W =
// bf16 copy of the weights).  The register-staged kernel above is latency-bound: one k-tile of
// loads per workgroup in flight, measured 2.5-3 TB/s.  Here the operand tiles go HBM -> LDS with
// global_load_lds_dwordx4 (no VGPRs, no conversion, no ds_write), a 4-slot ring keeps THREE k-tiles
// per workgroup in flight across tile boundaries, waits are counted (s_waitcnt vmcnt(N), never 0 in
// steady state) and the barrier is a raw s_barrier (a __syncthreads() would drain the DMA queue).
//
// LDS image of a slot: [128 rows][32 bf16] = 64-B rows, lane-linear as the DMA writes it (16 rows per
// wave-instruction).  Unpadded 64-B rows would make the ds_read_b128 fragment reads 4-way conflicted,
// so chunk c (16 B) of row r is stored at chunk slot c ^ ((r >> 2) & 3): the swizzle is applied on
// the per-lane GLOBAL source address of the DMA and again on the fragment read (cdna guide rule 21).
// ------------------------------------------------------------------------------------------
constexpr int DS = 4;            // ring slots
constexpr int DTK = 32;          // k per slot

typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void gbl_cvoid;

__device__ __forceinline__ void dma_rows16(const __bf16* G, int ld, int row0, int nrows, int k0, __bf16* lds_rows,
                                           int lane) {
    // one wave-instruction: 16 rows x 4 chunks of 16 B -> 1 KB of LDS starting at lds_rows (wave-uniform)
    const int r = row0 + (lane >> 2), p = lane & 3;
    const int c = p ^ ((r >> 2) & 3);
    const int rr = r < nrows ? r : nrows - 1;                     // clamp: padded rows are never stored
    const __bf16* src = G + (size_t)rr * ld + k0 + c * 8;
    __builtin_amdgcn_global_load_lds((gbl_cvoid*)src, (lds_void*)lds_rows, 16, 0, 0);
}

// 64-wide k-slots: one wave-instruction = 8 rows x 8 chunks of 16 B = 8 WHOLE 128-B cache lines (the 32-wide form
// above fetches half a line per row, and the other half one k-tile later, when L1 has lost it).  LDS rows are 128 B;
// chunk c of row r is stored at chunk slot c ^ ((r >> 1) & 7): the bank of a 16-B piece is 32 (r & 1) + 4 slot, and
// the 16 lanes that a ds_read_b128 services together hold rows whose (r >> 1) & 7 takes every value twice, once per
// parity -> 16 distinct pieces of the 256-B bank row.  (c ^ (r & 7) was 2-way conflicted: SQ_LDS_BANK_CONFLICT
// equal to the active LDS cycles.)
__device__ __forceinline__ void dma_rows8(const __bf16* G, int ld, int row0, int nrows, int k0, __bf16* lds_rows,
                                          int lane) {
    const int r = row0 + (lane >> 3), p = lane & 7;
    const int c = p ^ ((r >> 1) & 7);
    const int rr = r < nrows ? r : nrows - 1;
    const __bf16* src = G + (size_t)rr * ld + k0 + c * 8;
    __builtin_amdgcn_global_load_lds((gbl_cvoid*)src, (lds_void*)lds_rows, 16, 0, 0);
}

// Output tile BTM x BTN per workgroup, one wave per 64x64 block of it (4 waves at 128x128, 16 at
// 256x256).  The bigger tile halves the bytes that cross the L2 -> LDS path per FLOP (every operand
// tile is re-read once per tile of the OTHER dimension) -- at 128x128 that path carried 3x the
// algorithmic bytes and, not HBM, set the rate.
// NDS = ring depth, BIASF = floats of the bias image.  <256,128,3,1024>: 76 KB of LDS -> TWO workgroups per CU, so
// one workgroup's epilogue (stores at HBM rate) overlaps the other's main loop.
// ADEEP: the A operand (the one that comes from HBM: dP in the dX GEMMs) gets a ring one slot deeper than W's and
// runs one k-tile further ahead -- with 64-wide slots only one (A, W) pair fits in flight otherwise, i.e. 32 KB of HBM
// reads per CU.  3 x 32 KB of A + 2 x 32 KB of W is all 160 KB of LDS: no bias image (BIASF = 0), so no-bias GEMMs only.
template <int EPI, int BTM, int BTN, int NDS = DS, int BIASF = 2048, int KT = DTK, bool ADEEP = false>
__global__ __launch_bounds__((BTM / 64) * (BTN / 64) * 64, (BTM * BTN > 128 * 128) ? 4 : 2)
void gemm_nt_dma_kernel(NTArgs g) {
    constexpr int WR = BTM / 64, WC = BTN / 64, NWV = WR * WC, NTHR = NWV * 64;
    constexpr int RPI = KT == 32 ? 16 : 8;                        // rows per 1-KB DMA instruction
    constexpr int NA = BTM / RPI / NWV, NB = BTN / RPI / NWV;     // 1-KB DMA instructions per wave per k-tile
    constexpr int ASLOT = BTM * KT, WSLOT = BTN * KT;
    constexpr int NAS = ADEEP ? NDS + 1 : NDS;                    // A ring slots
    constexpr int VM_STEADY = (NDS - 2) * (NA + NB) + (ADEEP ? NA : 0), VM_EPI = VM_STEADY + 8;
    // VM_EPI: the wait right after an epilogue may leave only operations YOUNGER than the awaited DMA in flight, i.e. the
    // epilogue's stores: 8 per wave with bf16 P (16-B fragment stores), 16 with fp32 P -- the smaller count is safe for both
    // (a larger one would let the wait pass with the awaited DMA itself still outstanding)
    static_assert(NA >= 1 && NB >= 1 && VM_EPI < 64, "tile / wave configuration");
    static_assert(!ADEEP || (NDS == 2 && BIASF == 0 && EPI == 0), "deep-A variant: two W slots, no bias image");
    __shared__ __attribute__((aligned(1024))) __bf16 ring[NAS * ASLOT + NDS * WSLOT + 2 * BIASF];     // A ring, W ring, bias image
    __bf16* wring = ring + NAS * ASLOT;
    float* bias_s = reinterpret_cast<float*>(wring + NDS * WSLOT);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave / WC, wc = wave % WC;
    const __bf16* A = reinterpret_cast<const __bf16*>(g.A);
    const __bf16* W = reinterpret_cast<const __bf16*>(g.W);
    const int ntn = (g.N + BTN - 1) / BTN, ntm = (g.M + BTM - 1) / BTM;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, nslot = gridDim.x >> 3;
    const int panels = (ntm - xcd + 7) / 8, ntile = panels * ntn;
    const int nk = g.K / KT;
    if (slot >= ntile) return;
    const int my_tiles = (ntile - slot + nslot - 1) / nslot;
    const int total = my_tiles * nk;
    if (BIASF > 0) {       // bias via LDS: an ordinary global load inside the loop would make hipcc drain the DMA queue
        for (int i = tid; i < g.N && i < BIASF; i += NTHR) bias_s[i] = g.bias ? g.bias[i] : 0.f;
        __syncthreads();
    }
    // Optional start stagger (LOB_NT_STAGGER, default 0 = off; measured: no effect at 1/2/4/8 units, so the
    // workgroups' epilogues are not what synchronises them -- the epilogue cost is store issue, see there).
    for (int i = ((blockIdx.x >> 3) & 7) * g.stagger; i > 0; --i) __builtin_amdgcn_s_sleep(32);

    // producer cursors (W runs NDS-1 k-tiles ahead of the consumer, A one more with ADEEP; across tile boundaries)
    int pa_q = 0, pa_it = slot, pa_kt = 0, pw_q = 0, pw_it = slot, pw_kt = 0;
    auto issueA = [&]() {
        const int m0 = ((pa_it / ntn) * 8 + xcd) * BTM;
        __bf16* as = ring + (pa_q % NAS) * ASLOT;
#pragma unroll
        for (int j = 0; j < NA; ++j) {
            const int rb = (wave * NA + j) * RPI;
            if constexpr (KT == 32) dma_rows16(A, g.lda, m0 + rb, g.M, pa_kt * KT, as + rb * KT, lane);
            else                    dma_rows8(A, g.lda, m0 + rb, g.M, pa_kt * KT, as + rb * KT, lane);
        }
        ++pa_q;
        if (++pa_kt == nk) { pa_kt = 0; pa_it += nslot; }
    };
    auto issueW = [&]() {
        const int n0 = (pw_it % ntn) * BTN;
        __bf16* ws = wring + (pw_q % NDS) * WSLOT;
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const int rb = (wave * NB + j) * RPI;
            if constexpr (KT == 32) dma_rows16(W, g.ldw, n0 + rb, g.N, pw_kt * KT, ws + rb * KT, lane);
            else                    dma_rows8(W, g.ldw, n0 + rb, g.N, pw_kt * KT, ws + rb * KT, lane);
        }
        ++pw_q;
        if (++pw_kt == nk) { pw_kt = 0; pw_it += nslot; }
    };
    if (ADEEP) {            // issue order A(0) W(0) A(1): every wait below leaves exactly the youngest A tile in flight
        issueA();
        issueW();
        if (total > 1) issueA();
    } else {
#pragma unroll 1
        for (int i = 0; i < NDS - 1 && i < total; ++i) { issueA(); issueW(); }
    }

    int it = slot, kt = 0, since_epi = 99;
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    for (int q = 0; q < total; ++q) {
        // ---- wait for slot q: all but the younger operations of THIS wave may still be in flight:
        //      (DS-2) k-tiles x (NA+NB) DMAs, plus the 16 epilogue stores if they were issued after DMA(q)
        if (q + NDS - 1 + (ADEEP ? 1 : 0) > total) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else if (EPI == 1 && since_epi < NDS - 1) {
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(VM_EPI) : "memory");
        } else if (since_epi < NDS - 1) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(VM_STEADY) : "memory");
        }
        __builtin_amdgcn_s_barrier();
        if (ADEEP) {                                    // W(q+1) first, then A(q+2): the next wait counts past the A tile
            if (pw_q < total) issueW();
            if (pa_q < total) issueA();
        } else if (pa_q < total) {                      // refills the slots read in iteration q-1
            issueA();
            issueW();
        }
        ++since_epi;

        const __bf16* as = ring + (q % NAS) * ASLOT;
        const __bf16* ws = wring + (q % NDS) * WSLOT;
        const int r31 = lane & 31, hi = lane >> 5, sw = KT == 32 ? (r31 >> 2) & 3 : (r31 >> 1) & 7;
#pragma unroll
        for (int s = 0; s < KT / 16; ++s) {
            const int pc = ((2 * s + hi) ^ sw) * 8;
            const bf16x8 a0 = *reinterpret_cast<const bf16x8*>(as + (64 * wr + r31) * KT + pc);
            const bf16x8 a1 = *reinterpret_cast<const bf16x8*>(as + (64 * wr + 32 + r31) * KT + pc);
            const bf16x8 b0 = *reinterpret_cast<const bf16x8*>(ws + (64 * wc + r31) * KT + pc);
            const bf16x8 b1 = *reinterpret_cast<const bf16x8*>(ws + (64 * wc + 32 + r31) * KT + pc);
            acc[0][0] = mfma_bf16(a0, b0, acc[0][0]);
            acc[0][1] = mfma_bf16(a0, b1, acc[0][1]);
            acc[1][0] = mfma_bf16(a1, b0, acc[1][0]);
            acc[1][1] = mfma_bf16(a1, b1, acc[1][1]);
        }
        if (++kt < nk) continue;

        // ---- epilogue of tile `it`
        kt = 0;
        const int cm0 = ((it / ntn) * 8 + xcd) * BTM, cn0 = (it % ntn) * BTN;
        it += nslot;
        since_epi = 0;
        if (EPI == 0) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int col = cn0 + 64 * wc + 32 * j + (lane & 31);
                    const float bv = BIASF > 0 ? lds_read_f32_opaque(bias_s + (col < BIASF ? col : 0)) : 0.f;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = cm0 + 64 * wr + 32 * i + acc_row(r, lane);
                        if (row < g.M && col < g.N) {
                            float val = apply_act(acc[i][j][r] + bv, g.act);
                            if (g.drop_p > 0.f) val *= lob_dropout_scale(g.seed, (uint64_t)row * g.ldc + col, g.drop_p);
                            // no accumulate here: a read-modify-write would drain the DMA queue
                            if (g.out_bf16) reinterpret_cast<__bf16*>(g.C)[(size_t)row * g.ldc + col] = (__bf16)val;
                            else            g.C[(size_t)row * g.ldc + col] = val;
                        }
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
                    size_t fo = ((((size_t)(d * g.T + t) * NBT + bt) * NW + w) * 4 + gg) * 1024 + lane * 4;
                    if (g.out_bf16) {
                        // bf16 fragment storage: [gate][q pair][lane][8]: two 16-B stores per 32x32 block.
                        // Measured on this epilogue (gate GEMM, K = 256, 0.96 ms): without the stores 0.66 ms, with the
                        // stores kept in L2 0.77 ms, non-temporal stores 0.96 ms -- 0.2 ms is HBM write-back that the
                        // main loop does not hide: VMEM operations of a wave retire in order, so the DMA issued after the
                        // epilogue cannot be counted complete before the stores are acknowledged
                        __bf16* dst = reinterpret_cast<__bf16*>(g.C) + (fo - lane * 4) + lane * 8;
#pragma unroll
                        for (int pq = 0; pq < 2; ++pq) {
                            bf16x8 v;
#pragma unroll
                            for (int e = 0; e < 8; ++e) v[e] = (__bf16)(acc[i][j][8 * pq + e] + bv);
                            *reinterpret_cast<bf16x8*>(dst + pq * 512) = v;
                        }
                    } else {
                        float* dst = g.C + fo;
#pragma unroll
                        for (int qq = 0; qq < 4; ++qq) {
                            f32x4 v = {acc[i][j][4 * qq + 0] + bv, acc[i][j][4 * qq + 1] + bv,
                                       acc[i][j][4 * qq + 2] + bv, acc[i][j][4 * qq + 3] + bv};
                            *reinterpret_cast<f32x4*>(dst + qq * 256) = v;
                        }
                    }
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// C[M,N] += A[Kc,M]^T B[Kc,N]  (weight gradients), bf16 MFMA.  The contraction index is the ROW
// index of both sources, i.e. both operands are "K-major".  The source tiles are copied into LDS
// as they are ([k][m], 16-B coalesced global loads, 8/16-B LDS stores) and the MFMA fragments
// (8 consecutive k for one m per lane) are produced by gfx950's transposing LDS read
// ds_read_b64_tr_b16: per 16-lane group it reads a 4(k) x 16(m) block and hands lane i column i.
// LDS rows are 128 bf16 + 32 pad = 320 B: the four k-rows of a block then sit 64 B apart modulo
// the 256-B bank row and the 32 lanes of a half-wave cover it exactly once (conflict-free).
// ------------------------------------------------------------------------------------------
constexpr int TK2 = 32;      // contraction rows per stage
constexpr int LDK = 160;     // LDS row stride in bf16 elements (320 B)

struct TNArgs {
    const void* A; const void* B; float* C;
    int lda, ldb, ldc, M, N, Kc, kchunk, tiles;
};

// fp32 source: thread -> k = tid/32 + 8 i (i<4), 4 floats at col (tid%32)*4
__device__ __forceinline__ void ldk_f32(const float* __restrict__ G, int ld, int k0, int kend, int c0, int cols,
                                        int tid, f32x4 (&r)[4]) {
    const int kk = tid >> 5, c4 = (tid & 31) * 4;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int k = k0 + kk + 8 * i, c = c0 + c4;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (k < kend && c < cols) v = *reinterpret_cast<const f32x4*>(G + (size_t)k * ld + c);
        r[i] = v;
    }
}
__device__ __forceinline__ void stk_f32(__bf16* S, int tid, const f32x4 (&r)[4]) {
    const int kk = tid >> 5, c4 = (tid & 31) * 4;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        bf16x4 h = {(__bf16)r[i][0], (__bf16)r[i][1], (__bf16)r[i][2], (__bf16)r[i][3]};
        *reinterpret_cast<bf16x4*>(S + (kk + 8 * i) * LDK + c4) = h;
    }
}
// bf16 source: thread -> k = tid/16 + 16 i (i<2), 8 bf16 at col (tid%16)*8
__device__ __forceinline__ void ldk_bf16(const __bf16* __restrict__ G, int ld, int k0, int kend, int c0, int cols,
                                         int tid, bf16x8 (&r)[2]) {
    const int kk = tid >> 4, c8 = (tid & 15) * 8;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int k = k0 + kk + 16 * i, c = c0 + c8;
        bf16x8 v;
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = (__bf16)0.f;
        if (k < kend && c < cols) v = *reinterpret_cast<const bf16x8*>(G + (size_t)k * ld + c);
        r[i] = v;
    }
}
__device__ __forceinline__ void stk_bf16(__bf16* S, int tid, const bf16x8 (&r)[2]) {
    const int kk = tid >> 4, c8 = (tid & 15) * 8;
#pragma unroll
    for (int i = 0; i < 2; ++i) *reinterpret_cast<bf16x8*>(S + (kk + 16 * i) * LDK + c8) = r[i];
}

typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;

// fragment of the 32-column block starting at column `cb`, k-step s (16 rows) of a [k][col] LDS image
__device__ __forceinline__ bf16x8 tr_frag(const __bf16* S, int cb, int s, int lane) {
    const int h = lane >> 5, mh = (lane >> 4) & 1, q = (lane >> 2) & 3, p = lane & 3;
    const __bf16* a = S + (16 * s + 8 * h + q) * LDK + cb + 16 * mh + 4 * p;
    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)a);
    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(a + 4 * LDK));
    bf16x8 f = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return f;
}

template <bool A_BF16, bool B_BF16>
__global__ __launch_bounds__(256, 2) void gemm_tn_bf16_kernel(TNArgs g) {
    __shared__ __attribute__((aligned(16))) __bf16 lds[2 * 2 * TK2 * LDK];
    __bf16* As = lds;
    __bf16* Bs = lds + 2 * TK2 * LDK;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int ntn = (g.N + 127) / 128;
    // workgroups of one contraction chunk are 8 apart in blockIdx -> they share an XCD (L2) and run
    // together, so the source tiles they have in common are fetched from HBM once (speed only).
    const int xcd = blockIdx.x & 7, rest = blockIdx.x >> 3;
    const int tile = rest % g.tiles, chunk = (rest / g.tiles) * 8 + xcd;
    const int m0 = (tile / ntn) * 128, n0 = (tile % ntn) * 128;
    const int kbeg = chunk * g.kchunk, kend = min(g.Kc, kbeg + g.kchunk);
    if (kbeg >= kend) return;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    f32x4 raf[4], rbf[4];
    bf16x8 rah[2], rbh[2];
    auto load = [&](int k0) {
        if (A_BF16) ldk_bf16(reinterpret_cast<const __bf16*>(g.A), g.lda, k0, kend, m0, g.M, tid, rah);
        else        ldk_f32(reinterpret_cast<const float*>(g.A), g.lda, k0, kend, m0, g.M, tid, raf);
        if (B_BF16) ldk_bf16(reinterpret_cast<const __bf16*>(g.B), g.ldb, k0, kend, n0, g.N, tid, rbh);
        else        ldk_f32(reinterpret_cast<const float*>(g.B), g.ldb, k0, kend, n0, g.N, tid, rbf);
    };
    auto store = [&](int b) {
        if (A_BF16) stk_bf16(As + b * TK2 * LDK, tid, rah); else stk_f32(As + b * TK2 * LDK, tid, raf);
        if (B_BF16) stk_bf16(Bs + b * TK2 * LDK, tid, rbh); else stk_f32(Bs + b * TK2 * LDK, tid, rbf);
    };
    load(kbeg);
    store(0);
    __syncthreads();
    int buf = 0;
    for (int k0 = kbeg; k0 < kend; k0 += TK2) {
        const bool more = k0 + TK2 < kend;
        if (more) load(k0 + TK2);
        const __bf16* as = As + buf * TK2 * LDK;
        const __bf16* bs = Bs + buf * TK2 * LDK;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const bf16x8 a0 = tr_frag(as, 64 * wr, s, lane), a1 = tr_frag(as, 64 * wr + 32, s, lane);
            const bf16x8 b0 = tr_frag(bs, 64 * wc, s, lane), b1 = tr_frag(bs, 64 * wc + 32, s, lane);
            acc[0][0] = mfma_bf16(a0, b0, acc[0][0]);
            acc[0][1] = mfma_bf16(a0, b1, acc[0][1]);
            acc[1][0] = mfma_bf16(a1, b0, acc[1][0]);
            acc[1][1] = mfma_bf16(a1, b1, acc[1][1]);
        }
        if (more) store(buf ^ 1);
        __syncthreads();
        buf ^= 1;
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



// ------------------------------------------------------------------------------------------
// LDS-DMA variant of the TN (weight-gradient) GEMM for bf16 sources, BTM x BTN output tiles (256 x 256
// with 16 waves, 256 x 128 with 8).  The [k][cols] source tiles need no transformation on their way into
// LDS (ds_read_b64_tr_b16 transposes on the read side), so they are DMA'd: 4-slot ring, three k-tiles in
// flight, counted vmcnt, raw barrier; no stores until the final atomics.  (At 128 x 128 this variant was
// 12-20 % slower than the register-staged kernel -- half the resident workgroups; the big tile halves the
// operand re-reads instead.)  Rows are unpadded: the 64-B granule of k-row k is XOR-ed with (k & 3), on
// the DMA source address and on the tr-read address, which spreads the four k-rows of a tr block over the
// 256-B bank row exactly like the +64-B row pad of the register-staged kernel.
// ------------------------------------------------------------------------------------------
template <int COLS>
__device__ __forceinline__ void dma_krows(const __bf16* G, int ld, int k0, int krow0, int c0, __bf16* lds_rows, int lane) {
    // one wave-instruction = 1 KB = (512 / COLS) k-rows of COLS bf16
    constexpr int CPR = COLS / 8;                       // 16-B chunks per row
    const int kr = krow0 + lane / CPR, ch = lane % CPR;
    const __bf16* src = G + (size_t)(k0 + kr) * ld + c0 + ((ch ^ ((kr & 3) << 2)) * 8);
    __builtin_amdgcn_global_load_lds((gbl_cvoid*)src, (lds_void*)lds_rows, 16, 0, 0);
}

template <int COLS>
__device__ __forceinline__ bf16x8 tr_frag_sw(const __bf16* S, int cb, int s, int lane) {
    const int h = lane >> 5, mh = (lane >> 4) & 1, q = (lane >> 2) & 3, p = lane & 3;
    const __bf16* a = S + (16 * s + 8 * h + q) * COLS + ((cb + 16 * mh + 4 * p) ^ (q * 32));
    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)a);
    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(a + 4 * COLS));
    bf16x8 f = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return f;
}

template <int BTM, int BTN>
__global__ __launch_bounds__((BTM / 64) * (BTN / 64) * 64, (BTM / 64) * (BTN / 64) / 4) void gemm_tn_dma_kernel(TNArgs g) {
    constexpr int WC = BTN / 64, NWV = (BTM / 64) * WC;
    constexpr int ASLOT = 32 * BTM, BSLOT = 32 * BTN, SLOT = ASLOT + BSLOT;
    constexpr int RPA = 512 / BTM, RPB = 512 / BTN;                  // k-rows per DMA instruction
    constexpr int NA = 32 / RPA / NWV, NB = 32 / RPB / NWV;          // DMA instructions per wave per k-tile
    static_assert(NA >= 1 && NB >= 1, "tile / wave configuration");
    constexpr int VM_STEADY = (DS - 2) * (NA + NB);
    __shared__ __attribute__((aligned(1024))) __bf16 ring[DS * SLOT];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave / WC, wc = wave % WC;
    const __bf16* A = reinterpret_cast<const __bf16*>(g.A);
    const __bf16* B = reinterpret_cast<const __bf16*>(g.B);
    const int ntn = g.N / BTN;
    const int xcd = blockIdx.x & 7, rest = blockIdx.x >> 3;
    const int tile = rest % g.tiles, chunk = (rest / g.tiles) * 8 + xcd;
    const int m0 = (tile / ntn) * BTM, n0 = (tile % ntn) * BTN;
    const int kbeg = chunk * g.kchunk, kend = min(g.Kc, kbeg + g.kchunk);
    if (kbeg >= kend) return;
    const int total = (kend - kbeg) / 32;

    int p_q = 0;
    auto issue = [&]() {
        __bf16* as = ring + (p_q % DS) * SLOT;
        __bf16* bs = as + ASLOT;
        const int k0 = kbeg + p_q * 32;
#pragma unroll
        for (int j = 0; j < NA; ++j) {
            const int kr0 = (wave * NA + j) * RPA;
            dma_krows<BTM>(A, g.lda, k0, kr0, m0, as + kr0 * BTM, lane);
        }
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const int kr0 = (wave * NB + j) * RPB;
            dma_krows<BTN>(B, g.ldb, k0, kr0, n0, bs + kr0 * BTN, lane);
        }
        ++p_q;
    };
#pragma unroll 1
    for (int i = 0; i < DS - 1 && i < total; ++i) issue();

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // Fragment reads as inline asm with hand-counted lgkmcnt (see lstm_dw_h128_kernel below: compiler-visible
    // ds_read_b64_tr_b16 of the ring get an s_waitcnt vmcnt(0) in front, which waits for the DMA just issued).
    // Lane addresses: block 64 wr (+32: address ^ 64 B) of the A tile, block 64 wc of the B tile.
    const int fh = lane >> 5, fmh = (lane >> 4) & 1, fq = (lane >> 2) & 3, fp = lane & 3;
    const unsigned ring_b = (unsigned)(uintptr_t)(__attribute__((address_space(3))) const __bf16*)ring;
    const unsigned a_off = ring_b + 2 * ((8 * fh + fq) * BTM + ((64 * wr) ^ (32 * fq)) + 16 * fmh + 4 * fp);
    const unsigned b_off = ring_b + 2 * ((8 * fh + fq) * BTN + ((64 * wc) ^ (32 * fq)) + 16 * fmh + 4 * fp);
#define LOB_TR2(f, addr, OFF, HI)                                                                              \
    asm volatile("ds_read_b64_tr_b16 %0, %2 offset:%3\n\tds_read_b64_tr_b16 %1, %2 offset:%4"                   \
                 : "=&v"(f##l), "=&v"(f##h) : "v"(addr), "n"(OFF), "n"((OFF) + (HI)) : "memory")
#define LOB_FRAG(f) bf16x8{f##l[0], f##l[1], f##l[2], f##l[3], f##h[0], f##h[1], f##h[2], f##h[3]}
    for (int q = 0; q < total; ++q) {
        if (q + DS - 1 > total) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else                    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(VM_STEADY) : "memory");
        __builtin_amdgcn_s_barrier();
        if (p_q < total) issue();
        const unsigned sb = (q % DS) * (2 * SLOT);
        const unsigned va0 = sb + a_off, va1 = va0 ^ 64, vb0 = sb + b_off, vb1 = vb0 ^ 64;
        bf16x4 a0l, a0h, a1l, a1h, b0l, b0h, b1l, b1h;
#define LOB_KSTEP(S)                                                                                           \
        LOB_TR2(a0, va0, 2 * 16 * BTM * S, 2 * 4 * BTM);                                                       \
        LOB_TR2(a1, va1, 2 * 16 * BTM * S, 2 * 4 * BTM);                                                       \
        LOB_TR2(b0, vb0, 2 * ASLOT + 2 * 16 * BTN * S, 2 * 4 * BTN);                                           \
        LOB_TR2(b1, vb1, 2 * ASLOT + 2 * 16 * BTN * S, 2 * 4 * BTN);                                           \
        asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(a0l), "+v"(a0h), "+v"(a1l), "+v"(a1h), "+v"(b0l), "+v"(b0h)); \
        acc[0][0] = mfma_bf16(LOB_FRAG(a0), LOB_FRAG(b0), acc[0][0]);                                          \
        acc[1][0] = mfma_bf16(LOB_FRAG(a1), LOB_FRAG(b0), acc[1][0]);                                          \
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(b1l), "+v"(b1h));                                           \
        acc[0][1] = mfma_bf16(LOB_FRAG(a0), LOB_FRAG(b1), acc[0][1]);                                          \
        acc[1][1] = mfma_bf16(LOB_FRAG(a1), LOB_FRAG(b1), acc[1][1]);
        LOB_KSTEP(0)
        LOB_KSTEP(1)
#undef LOB_KSTEP
    }
#undef LOB_TR2
#undef LOB_FRAG
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = n0 + 64 * wc + 32 * j + (lane & 31);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + 64 * wr + 32 * i + acc_row(r, lane);
                atomicAdd(g.C + (size_t)row * g.ldc + col, acc[i][j][r]);
            }
        }
}

// ------------------------------------------------------------------------------------------
// Both weight gradients of one LSTM layer in ONE pass over dP (H = 128):
//     dW_ih[d] (4H x NX)  = sum_rows dP[row, d]^T X[row]
//     dW_hh[d] (4H x H)   = sum_rows dP[row, d]^T h_prev_d[row],   h_prev_0[t] = Y[t-1, 0:H], h_prev_1[t] = Y[t+1, H:2H]
// dP (bf16, 2 KB per row) is by far the widest operand; as two GEMM launches it was read from HBM twice
// per layer.  Output tile 128 (dP columns) x (NX + 128): the contraction tile of dP is paired with the same
// rows of X and with the rows of Y one time step (Bp rows) earlier / later; k-tiles that lie in the step
// without a predecessor (t = 0 forward, t = T-1 reverse) skip the Y products (Bp % 32 == 0: a k-tile never
// straddles two steps).  16 waves as 4 x 4, wave tile 32 x (NX + 128) / 4; same ring / swizzle / split-k /
// XCD map as gemm_tn_dma_kernel.
// ------------------------------------------------------------------------------------------
constexpr int DWDS = 4;          // ring slots of the fused dW kernel (40 KB each at NX = 256: all 160 KB of LDS)
struct DWArgs {
    const __bf16* dP; const __bf16* X; const __bf16* Y; float* dWih; float* dWhh;
    int ldp, ldx, ldy, T, Bp, kchunk, tiles;
};

template <int COLS>
__device__ __forceinline__ bf16x8 tr_pair(const __bf16* a) {       // k-rows r .. r+3 and r+4 .. r+7 of one fragment
    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)a);
    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(a + 4 * COLS));
    bf16x8 f = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return f;
}

template <int NX>
__global__ __launch_bounds__(512, 2) void lstm_dw_h128_kernel(DWArgs g) {
    constexpr int BTM = 256, NBW = (NX + 128) / 64, XB = NX / 32;       // 32-col blocks per wave; X blocks in all
    constexpr int ASLOT = 32 * BTM, XSLOT = 32 * NX, YSLOT = 32 * 128, SLOT = ASLOT + XSLOT + YSLOT;
    constexpr int NDS = DWDS;
    constexpr int NDMA = 2 + NX / 128 + 1;                              // DMA instructions per wave per k-tile
    constexpr int VM_STEADY = (NDS - 2) * NDMA;
    __shared__ __attribute__((aligned(1024))) __bf16 ring[NDS * SLOT];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    const int xcd = blockIdx.x & 7, rest = blockIdx.x >> 3;
    const int tile = rest % g.tiles, chunk = (rest / g.tiles) * 8 + xcd;
    const int m0 = tile * BTM, d = m0 / 512;
    const int Kc = g.T * g.Bp;
    const int kbeg = chunk * g.kchunk, kend = min(Kc, kbeg + g.kchunk);
    if (kbeg >= kend) return;
    const int total = (kend - kbeg) / 32;
    // rows [ex_lo, ex_hi) of dP have no h_prev
    const int ex_lo = d == 0 ? 0 : (g.T - 1) * g.Bp, ex_hi = ex_lo + g.Bp;
    const int yshift = d == 0 ? -g.Bp : g.Bp;

    // DMA sources: wave-uniform 64-bit base (advances by k-tile) + 32-bit lane byte offsets, so that the instruction
    // takes the base from SGPRs.  One instruction covers 2 k-rows of a 256-column tile (4 of a 128-column one); k-row
    // kr's 16-B chunk ch comes from source chunk ch ^ 4 (kr & 3) (dma_krows).
    const int l5 = lane >> 5, c5 = lane & 31, l4 = lane >> 4, c4 = lane & 15;
    unsigned da_off[2], dx_off[2], dy_off;
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const int kr = wave * 4 + 2 * e + l5;
        da_off[e] = 2 * (kr * g.ldp + ((c5 ^ ((kr & 3) << 2)) * 8));
        dx_off[e] = 2 * (kr * g.ldx + ((c5 ^ ((kr & 3) << 2)) * 8));
    }
    {
        const int kr = wave * 4 + l4;
        dy_off = 2 * (kr * g.ldy + ((c4 ^ ((kr & 3) << 2)) * 8));
        if (NX == 128) dx_off[0] = 2 * (kr * g.ldx + ((c4 ^ ((kr & 3) << 2)) * 8));
    }
    const char* dP_b = reinterpret_cast<const char*>(g.dP + m0);
    const char* X_b = reinterpret_cast<const char*>(g.X);
    const char* Y_b = reinterpret_cast<const char*>(g.Y + d * 128);
    int p_q = 0;
    auto issue = [&]() {            // per wave and k-tile: 4 k-rows of dP (2 instructions), 4 of X (2 or 1), 4 of h_prev (1)
        __bf16* as = ring + (p_q % NDS) * SLOT;
        __bf16* xs = as + ASLOT;
        __bf16* ys = xs + XSLOT;
        const int k0 = kbeg + p_q * 32;
        const char* pa = dP_b + (size_t)k0 * g.ldp * 2;
        const char* px = X_b + (size_t)k0 * g.ldx * 2;
        __builtin_amdgcn_global_load_lds((gbl_cvoid*)(pa + da_off[0]), (lds_void*)(as + wave * 4 * 256), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((gbl_cvoid*)(pa + da_off[1]), (lds_void*)(as + (wave * 4 + 2) * 256), 16, 0, 0);
        if (NX == 256) {
            __builtin_amdgcn_global_load_lds((gbl_cvoid*)(px + dx_off[0]), (lds_void*)(xs + wave * 4 * 256), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((gbl_cvoid*)(px + dx_off[1]), (lds_void*)(xs + (wave * 4 + 2) * 256), 16, 0, 0);
        } else {
            __builtin_amdgcn_global_load_lds((gbl_cvoid*)(px + dx_off[0]), (lds_void*)(xs + wave * 4 * 128), 16, 0, 0);
        }
        // rows of the step without a predecessor are fetched unshifted and never used
        const int ky = (k0 >= ex_lo && k0 < ex_hi) ? k0 : k0 + yshift;
        const char* py = Y_b + (size_t)ky * g.ldy * 2;
        __builtin_amdgcn_global_load_lds((gbl_cvoid*)(py + dy_off), (lds_void*)(ys + wave * 4 * 128), 16, 0, 0);
        ++p_q;
    };
#pragma unroll 1
    for (int i = 0; i < NDS - 1 && i < total; ++i) issue();

    f32x16 acc[2][NBW];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NBW; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // tr-read addressing (see tr_frag_sw): element (k-row 8 h + q, column cb + 16 mh + 4 p) of a [k][COLS] tile sits at
    // column (cb ^ 32 q) + 16 mh + 4 p.  Of a 32-column block index only the two low bits meet the swizzle: this wave's
    // row blocks 2 wr + i and column blocks 2 j + wc need two lane offsets per operand, the rest is a compile-time
    // offset in the DS instruction.  Byte addresses, relative to the slot.
    const int fh = lane >> 5, fmh = (lane >> 4) & 1, fq = (lane >> 2) & 3, fp = lane & 3;
    const int frow = 8 * fh + fq, fcol = 16 * fmh + 4 * fp;
    const unsigned ring_b = (unsigned)(uintptr_t)(__attribute__((address_space(3))) const __bf16*)ring;
    // (block + 1 of an even block = address ^ 64 B, block + 2 of a block < 2 = address ^ 128 B; slots are 1-KB aligned)
    const unsigned a_off = ring_b + 2 * (frow * 256 + 32 * ((((2 * wr) & 3) ^ fq) + ((2 * wr) & ~3)) + fcol);
    const unsigned x_off = ring_b + 2 * (frow * NX + 32 * (wc ^ fq) + fcol);
    const unsigned y_off = ring_b + 2 * (frow * 128 + 32 * (wc ^ fq) + fcol);
    // The fragment reads are inline asm: hipcc cannot tell which ring slot a ds_read_b64_tr_b16 touches and puts an
    // s_waitcnt vmcnt(0) in front of compiler-visible ones -- that waits for the DMA issued a moment ago, i.e. it
    // serialises the ring.  Asm reads are invisible to that pass; their own completion is counted here (LDS operations
    // return in order): the s_waitcnt lgkmcnt(n) that releases a fragment carries its registers as "+v" operands, so no
    // MFMA can be scheduled above it.  (hipcc sinks the MFMAs of a k-step below its last wait; pinning them between
    // the waits with sched_barrier changed nothing: 0.952 vs 0.954 ms.)
#define LOB_TR2(f, addr, OFF, HI)                                                                              \
    asm volatile("ds_read_b64_tr_b16 %0, %2 offset:%3\n\tds_read_b64_tr_b16 %1, %2 offset:%4"                   \
                 : "=&v"(f##l), "=&v"(f##h) : "v"(addr), "n"(OFF), "n"((OFF) + (HI)) : "memory")
#define LOB_LGKM(n, f) asm volatile("s_waitcnt lgkmcnt(" #n ")" : "+v"(f##l), "+v"(f##h))
#define LOB_FRAG(f) bf16x8{f##l[0], f##l[1], f##l[2], f##l[3], f##h[0], f##h[1], f##h[2], f##h[3]}
    constexpr int XO = 2 * ASLOT, YO = 2 * (ASLOT + XSLOT);             // byte offsets of the X / h_prev tiles in a slot
    constexpr int XS = 2 * 16 * NX, XH = 2 * 4 * NX;                    // bytes per k-step / to the upper 4 k-rows

    for (int q = 0; q < total; ++q) {
        if (q + NDS - 1 > total) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else                     asm volatile("s_waitcnt vmcnt(%0)" ::"n"(VM_STEADY) : "memory");
        __builtin_amdgcn_s_barrier();
        if (p_q < total) issue();
        const unsigned sb = (q % NDS) * (2 * SLOT);
        const unsigned va0 = sb + a_off, va1 = va0 ^ 64;
        const unsigned vx0 = sb + x_off, vx1 = vx0 ^ 128, vy0 = sb + y_off, vy1 = vy0 ^ 128;
        const int k0 = kbeg + q * 32;
        const bool has_prev = !(k0 >= ex_lo && k0 < ex_hi);
        bf16x4 a0l, a0h, a1l, a1h, b0l, b0h, b1l, b1h, b2l, b2h, b3l, b3h;
#define LOB_KSTEP(S)                                                                                           \
        LOB_TR2(a0, va0, 8192 * S, 2048);                                                                      \
        LOB_TR2(a1, va1, 8192 * S, 2048);                                                                      \
        LOB_TR2(b0, vx0, XO + XS * S, XH);                                                                     \
        LOB_TR2(b1, vx1, XO + XS * S, XH);                                                                     \
        if constexpr (NX == 256) {                                                                             \
            LOB_TR2(b2, vx0, XO + XS * S + 256, XH);                                                           \
            LOB_TR2(b3, vx1, XO + XS * S + 256, XH);                                                           \
            asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(a0l), "+v"(a0h), "+v"(a1l), "+v"(a1h), "+v"(b0l), "+v"(b0h)); \
        } else {                                                                                               \
            asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(a0l), "+v"(a0h), "+v"(a1l), "+v"(a1h), "+v"(b0l), "+v"(b0h)); \
        }                                                                                                      \
        acc[0][0] = mfma_bf16(LOB_FRAG(a0), LOB_FRAG(b0), acc[0][0]);                                          \
        acc[1][0] = mfma_bf16(LOB_FRAG(a1), LOB_FRAG(b0), acc[1][0]);                                          \
        if constexpr (NX == 256) { LOB_LGKM(4, b1); } else { LOB_LGKM(0, b1); }                                \
        acc[0][1] = mfma_bf16(LOB_FRAG(a0), LOB_FRAG(b1), acc[0][1]);                                          \
        acc[1][1] = mfma_bf16(LOB_FRAG(a1), LOB_FRAG(b1), acc[1][1]);                                          \
        if constexpr (NX == 256) {                                                                             \
            LOB_LGKM(2, b2);                                                                                   \
            acc[0][2] = mfma_bf16(LOB_FRAG(a0), LOB_FRAG(b2), acc[0][2]);                                      \
            acc[1][2] = mfma_bf16(LOB_FRAG(a1), LOB_FRAG(b2), acc[1][2]);                                      \
            LOB_LGKM(0, b3);                                                                                   \
            acc[0][3] = mfma_bf16(LOB_FRAG(a0), LOB_FRAG(b3), acc[0][3]);                                      \
            acc[1][3] = mfma_bf16(LOB_FRAG(a1), LOB_FRAG(b3), acc[1][3]);                                      \
        }                                                                                                      \
        if (has_prev) {                                                                                        \
            LOB_TR2(b0, vy0, YO + 4096 * S, 1024);                                                             \
            LOB_TR2(b1, vy1, YO + 4096 * S, 1024);                                                             \
            LOB_LGKM(2, b0);                                                                                   \
            acc[0][XB / 2] = mfma_bf16(LOB_FRAG(a0), LOB_FRAG(b0), acc[0][XB / 2]);                            \
            acc[1][XB / 2] = mfma_bf16(LOB_FRAG(a1), LOB_FRAG(b0), acc[1][XB / 2]);                            \
            LOB_LGKM(0, b1);                                                                                   \
            acc[0][XB / 2 + 1] = mfma_bf16(LOB_FRAG(a0), LOB_FRAG(b1), acc[0][XB / 2 + 1]);                    \
            acc[1][XB / 2 + 1] = mfma_bf16(LOB_FRAG(a1), LOB_FRAG(b1), acc[1][XB / 2 + 1]);                    \
        }
        LOB_KSTEP(0)
        LOB_KSTEP(1)
#undef LOB_KSTEP
    }
#undef LOB_TR2
#undef LOB_LGKM
#undef LOB_FRAG
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NBW; ++j) {
            const bool isx = j < XB / 2;
            float* C = isx ? g.dWih : g.dWhh;
            const int ldc = isx ? NX : 128;
            const int col = 32 * (2 * j + wc - (isx ? 0 : XB)) + (lane & 31);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + 64 * wr + 32 * i + acc_row(r, lane);
                atomicAdd(C + (size_t)row * ldc + col, acc[i][j][r]);
            }
        }
}

__global__ __launch_bounds__(256) void colsum_bf16_kernel(const __bf16* __restrict__ A, int lda, int M, int N,
                                                          int rows_per_block, float* __restrict__ out) {
    __shared__ float red[4][64];
    const int cl = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const int ncb = (N + 63) / 64;
    const int cb = blockIdx.x % ncb, rb = blockIdx.x / ncb;
    const int col = cb * 64 + cl;
    const int r0 = rb * rows_per_block, r1 = min(M, r0 + rows_per_block);
    float s = 0.f;
    if (col < N)
        for (int r = r0 + rg; r < r1; r += 4) s += (float)A[(size_t)r * lda + col];
    red[rg][cl] = s;
    __syncthreads();
    if (rg == 0 && col < N) atomicAdd(out + col, red[0][cl] + red[1][cl] + red[2][cl] + red[3][cl]);
}

inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// persistent grid of the NT kernel: one resident wave of workgroups (multiple of 8 for the XCD map)
inline int nt_grid(int M, int N) {
    const long tiles = (long)((M + TM - 1) / TM) * ((N + TN_ - 1) / TN_);
    const int per_cu = lob_variant(LOB_VAR_NT_WGS) > 0 ? lob_variant(LOB_VAR_NT_WGS) : 3;
    long gsz = 256L * per_cu;
    if (gsz > tiles) gsz = ((tiles + 7) / 8) * 8;
    return (int)gsz;
}

// contraction depth per LDS stage of the NT kernel (tuning knob; LOB_NT_TK=32|64)
inline int nt_tk() {
    const int v = lob_variant(LOB_VAR_NT_TK) == 64 ? 64 : 32;
    return v;
}

}  // namespace

int lob_gemm_nt_ws(const void* A, int lda, const void* W, void* C, int ldc, int M, int N, int K, int out_bf16,
                   hipStream_t s);                        // gate_gemm_ws.hip
int lob_dx_ksplit(const void* A, int lda, const void* Wt, void* C, int ldc, int M, int N, int K, int out_bf16, float drop_p,
                  uint64_t seed, hipStream_t s);         // dx_ksplit.hip
// gemm_pp.hip: the 8-wave ping-pong 256 x 256 x 64 kernels (the matrix-bound GEMMs of the H = 256 step)
bool lob_pp_nt_ok(int M, int N, int K);
bool lob_pp_tn_ok(int M, int N, int Kc);
int lob_gemm_nt_pp(const void* A, int lda, const void* Wt, int ldw, void* C, int ldc, int M, int N, int K, int out_bf16,
                   float drop_p, uint64_t seed, hipStream_t s);
int lob_gate_gemm_pp(const void* X, int ldx, const void* Wih, const float* bias, void* P, int T, int Bp, int H, int D, int K,
                     hipStream_t s);
int lob_gemm_tn_pp(const void* A, int lda, const void* B, int ldb, float* C, int ldc, int M, int N, int Kc, int shift,
                   int ex_lo, int ex_hi, hipStream_t s);
bool lob_dw_pp_ok(int T, int Bp, int H, int D, int nx);
int lob_lstm_dw_pp(const void* dP, int ldp, const void* X, int ldx, int nx, const void* Y, int ldy, float* dWih, float* dWhh,
                   int T, int Bp, int H, int D, hipStream_t s);

// A: fp32 (a_bf16 = 0) or bf16 (a_bf16 = 1) row-major [M][lda]; W fp32 [N][ldw]; C fp32.
inline int nt_stagger() {       // tuning knob LOB_NT_STAGGER (units of s_sleep(32) = 2048 clocks per group step)
    const int v = lob_variant(LOB_VAR_NT_STAGGER);
    return v;
}
inline bool nt_dma_enabled() {
    const bool v = lob_variant(LOB_VAR_NT_DMA) != 0;
    return v;
}
inline int nt_dma_tile() {      // output tile of the LDS-DMA kernel (tuning knob; LOB_DMA_TILE=128|256|2 (= 256x128, 2 WGs/CU))
    const int x = lob_variant(LOB_VAR_DMA_TILE);
    const int v = (x == 128 || x == 2) ? x : 256;
    return v;
}
inline int nt_dma_grid(int M, int N, int tile) {
    const long tiles = (long)((M + tile - 1) / tile) * ((N + tile - 1) / tile);
    long gsz = 256L * (tile == 128 ? 2 : 1);
    if (gsz > tiles) gsz = ((tiles + 7) / 8) * 8;
    return (int)gsz;
}
inline int nt_dma_grid2(int M, int N) {     // 256 x 128 tiles, one workgroup per CU
    const long tiles = (long)((M + 255) / 256) * ((N + 127) / 128);
    long gsz = 256L;
    if (gsz > tiles) gsz = ((tiles + 7) / 8) * 8;
    return (int)gsz;
}
template <int EPI>
inline void launch_nt_dma(const NTArgs& g_, hipStream_t s) {
    NTArgs g = g_;
    g.stagger = nt_stagger();
    // 256x256 tiles (16 waves, one workgroup per CU) wherever N is a multiple of 256: fewest operand bytes through
    // LDS per FLOP.  Measured at M = 1M: gate GEMM K=256 0.955 ms vs 1.053 with 256x128 tiles on two workgroups per
    // CU, dX N=256 0.835 vs 0.946; dX N=128 (layer 0) 0.648 (256x128, one workgroup) vs 0.591 (two workgroups).
    const bool two_wg = g.N <= 1024 && g.K / DTK >= 3;
    auto launch_2wg = [&] {
        const long tiles = (long)((g.M + 255) / 256) * ((g.N + 127) / 128);
        long gsz = 512;
        if (gsz > tiles) gsz = ((tiles + 7) / 8) * 8;
        hipLaunchKernelGGL((gemm_nt_dma_kernel<EPI, 256, 128, 3, 1024>), dim3((unsigned)gsz), dim3(512), 0, s, g);
    };
    // 64-wide k-slots (two of them) by default: every DMA instruction fetches whole cache lines.  Measured against
    // four 32-wide slots (half a line per row per k-tile): gate GEMM K=256 0.96 -> 0.92 ms, dX N=256 0.84 -> 0.77 ms.
    const int kt64 = lob_variant(LOB_VAR_DMA_KT);
    const bool adeep = lob_variant(LOB_VAR_NT_ADEEP) != 0;
    if (EPI == 0 && adeep && kt64 == 64 && !g.bias && g.N % 256 == 0 && g.K % 64 == 0 && g.K / 64 >= 3) {
        // no bias (dX = dP W_ih, dV = dU W1): deeper ring for the operand that comes from HBM
        if constexpr (EPI == 0)
            hipLaunchKernelGGL((gemm_nt_dma_kernel<0, 256, 256, 2, 0, 64, true>), dim3((unsigned)nt_dma_grid(g.M, g.N, 256)), dim3(1024), 0, s, g);
    }
    else if (kt64 == 64 && g.N % 256 == 0 && g.K % 64 == 0 && g.K / 64 >= 2)
        hipLaunchKernelGGL((gemm_nt_dma_kernel<EPI, 256, 256, 2, 2048, 64>), dim3((unsigned)nt_dma_grid(g.M, g.N, 256)), dim3(1024), 0, s, g);
    else if (nt_dma_tile() == 2 && two_wg)
        launch_2wg();
    else if (nt_dma_tile() != 128 && g.N % 256 == 0)
        hipLaunchKernelGGL((gemm_nt_dma_kernel<EPI, 256, 256>), dim3((unsigned)nt_dma_grid(g.M, g.N, 256)), dim3(1024), 0, s, g);
    else if (nt_dma_tile() != 128 && two_wg)      // N = 128 (dX of layer 0)
        launch_2wg();
    else if (nt_dma_tile() != 128)
        hipLaunchKernelGGL((gemm_nt_dma_kernel<EPI, 256, 128>), dim3((unsigned)nt_dma_grid2(g.M, g.N)), dim3(512), 0, s, g);
    else
        hipLaunchKernelGGL((gemm_nt_dma_kernel<EPI, 128, 128>), dim3((unsigned)nt_dma_grid(g.M, g.N, 128)), dim3(256), 0, s, g);
}

extern "C" int lob_gemm_nt_bf16(const void* A, int a_bf16, int lda, const void* W, int w_bf16, int ldw,
                                const float* bias, float* C, int ldc, int M, int N, int K, int act, float drop_p,
                                uint64_t seed, void* stream) {
    if (!A || !W || !C || M <= 0 || N <= 0 || K <= 0) return LOB_E_ARG;
    if (lda < K || ldw < K || ldc < N) return LOB_E_SHAPE;
    if ((act & 0xff) > LOB_ACT_GELU || act < 0) return LOB_E_ARG;
    if (!al16(A) || !al16(W) || (K % 8) || (lda % 8) || (ldw % 4)) return LOB_E_ALIGN;
    const int out16 = (act & LOB_OUT_BF16) ? 1 : 0;
    NTArgs g{A, reinterpret_cast<const float*>(W), bias, C, lda, ldw, ldc, M, N, K, act & 0xff, (act >> 8) & 1,
             0, 0, 0, 0, out16, drop_p, seed, 0};
    if (drop_p < 0.f || drop_p >= 1.f) return LOB_E_ARG;
    if (out16 && !(a_bf16 && w_bf16)) return LOB_E_SHAPE;      // bf16 C: the LDS-DMA kernel's row-major epilogue only
    if (w_bf16) {      // both operands bf16 in HBM: LDS-DMA kernel (no bias / activation in its row-major epilogue)
        if (!a_bf16 || (act & LOB_ACCUMULATE) || N > 2048 || (K % DTK) || K / DTK < DS || (ldw % 8)) return LOB_E_SHAPE;
        // dX = dP W_ih (wide contraction, narrow output): weights stationary, contraction split over the waves
        if (!bias && (act & 0xff) == LOB_ACT_NONE && (K == 512 || K == 1024) && (N == 128 || N == 256) && (M % 16) == 0 &&
            ldw == K && (ldc % 4) == 0 && al16(C) && lob_variant(LOB_VAR_DX_KSPLIT) != 0)
            return lob_dx_ksplit(A, lda, W, C, ldc, M, N, K, out16, drop_p, seed, (hipStream_t)stream);
        // narrow contraction, no bias / activation / dropout (dV = dPreU W1 of the attention pooling): weights stationary
        if (!bias && (act & 0xff) == LOB_ACT_NONE && drop_p == 0.f && (K == 128 || K == 256) && (N % 256) == 0 && N <= 1024 &&
            (M % 32) == 0 && ldw == K && (ldc % 4) == 0 && al16(C) && lob_variant(LOB_VAR_GATE_WS) != 0)
            return lob_gemm_nt_ws(A, lda, W, C, ldc, M, N, K, out16, (hipStream_t)stream);
        // wide contraction AND wide output (dX at H = 256: K = 2048, N = 512 / 256): matrix-bound -> ping-pong kernel
        if (!bias && (act & 0xff) == LOB_ACT_NONE && K >= 1024 && lob_pp_nt_ok(M, N, K) && (ldc % 4) == 0 && al16(C) &&
            (lob_variant(LOB_VAR_GEMM_PP) & 1))
            return lob_gemm_nt_pp(A, lda, W, ldw, C, ldc, M, N, K, out16, drop_p, seed, (hipStream_t)stream);
        launch_nt_dma<0>(g, (hipStream_t)stream);
        LOB_CHECK_LAUNCH();
        return 0;
    }
    const dim3 grid((unsigned)nt_grid(M, N)), block(256);
    const bool tk32 = nt_tk() == 32;
    if (a_bf16) {
        if (tk32) hipLaunchKernelGGL((gemm_nt_bf16_kernel<true, 0, 32>), grid, block, 0, (hipStream_t)stream, g);
        else      hipLaunchKernelGGL((gemm_nt_bf16_kernel<true, 0, 64>), grid, block, 0, (hipStream_t)stream, g);
    } else {
        if (tk32) hipLaunchKernelGGL((gemm_nt_bf16_kernel<false, 0, 32>), grid, block, 0, (hipStream_t)stream, g);
        else      hipLaunchKernelGGL((gemm_nt_bf16_kernel<false, 0, 64>), grid, block, 0, (hipStream_t)stream, g);
    }
    LOB_CHECK_LAUNCH();
    return 0;
}

int lob_gate_gemm_ws(const void* X, int ldx, const void* Wih, const float* bias, void* P, int T, int Bp, int H, int D, int K,
                     hipStream_t s);       // gate_gemm_ws.hip

extern "C" int lob_gate_gemm_x_bf16(const void* X, int x_bf16, int ldx, const void* Wih, int w_bf16, const float* bias,
                                    void* P, int p_bf16, int T, int Bp, int H, int D, int K, void* stream) {
    if (!X || !Wih || !P || T <= 0 || Bp <= 0 || H <= 0 || K <= 0 || (D != 1 && D != 2)) return LOB_E_ARG;
    if (ldx < K || (H % 32) || (Bp % 32)) return LOB_E_SHAPE;
    if (!al16(X) || !al16(Wih) || !al16(P) || (K % 8) || (ldx % 8)) return LOB_E_ALIGN;
    const int N = D * 4 * H, M = T * Bp;
    NTArgs g{X, reinterpret_cast<const float*>(Wih), bias, reinterpret_cast<float*>(P), ldx, K, N, M, N, K, LOB_ACT_NONE, 0,
             T, Bp, H, D, p_bf16, 0.f, 0, 0};
    if (w_bf16) {
        // H = 256: matrix-bound (410 FLOP per HBM byte) -> ping-pong kernel
        if (x_bf16 && p_bf16 && H == 256 && lob_pp_nt_ok(M, N, K) && N <= 2048 && (lob_variant(LOB_VAR_GEMM_PP) & 2))
            return lob_gate_gemm_pp(X, ldx, Wih, bias, P, T, Bp, H, D, K, (hipStream_t)stream);
        // bf16 P: the weight-stationary kernel (gate_gemm_ws.hip) -- only the activations stream
        if (x_bf16 && p_bf16 && ((H == 128 && (K == 128 || K == 256)) || (H == 256 && (K == 256 || K == 512))) &&
            lob_variant(LOB_VAR_GATE_WS) != 0)
            return lob_gate_gemm_ws(X, ldx, Wih, bias, P, T, Bp, H, D, K, (hipStream_t)stream);
        if (!x_bf16 || (K % DTK) || K / DTK < DS || N > 2048 || (N % 128) || (M % 256)) return LOB_E_SHAPE;
        launch_nt_dma<1>(g, (hipStream_t)stream);
        LOB_CHECK_LAUNCH();
        return 0;
    }
    const dim3 grid((unsigned)nt_grid(M, N)), block(256);
    if (x_bf16) {
        if (nt_tk() == 32) hipLaunchKernelGGL((gemm_nt_bf16_kernel<true, 1, 32>), grid, block, 0, (hipStream_t)stream, g);
        else               hipLaunchKernelGGL((gemm_nt_bf16_kernel<true, 1, 64>), grid, block, 0, (hipStream_t)stream, g);
    } else {
        if (nt_tk() == 32) hipLaunchKernelGGL((gemm_nt_bf16_kernel<false, 1, 32>), grid, block, 0, (hipStream_t)stream, g);
        else               hipLaunchKernelGGL((gemm_nt_bf16_kernel<false, 1, 64>), grid, block, 0, (hipStream_t)stream, g);
    }
    LOB_CHECK_LAUNCH();
    return 0;
}

extern "C" int lob_gemm_tn_bf16(const void* A, int a_bf16, int lda, const void* B, int b_bf16, int ldb,
                                float* C, int ldc, int M, int N, int Kc, void* stream) {
    if (!A || !B || !C || M <= 0 || N <= 0 || Kc <= 0) return LOB_E_ARG;
    if (lda < M || ldb < N || ldc < N) return LOB_E_SHAPE;
    // 16-byte vector loads along the column index of both sources
    const int am = a_bf16 ? 8 : 4, bm = b_bf16 ? 8 : 4;
    if ((M % am) || (lda % am) || (N % bm) || (ldb % bm) || !al16(A) || !al16(B)) return LOB_E_ALIGN;
    hipStream_t s = (hipStream_t)stream;
    // wide outputs over a long contraction (the H = 256 weight gradients): ping-pong kernel
    if (a_bf16 && b_bf16 && lob_pp_tn_ok(M, N, Kc) && (lob_variant(LOB_VAR_GEMM_PP) & 4))
        return lob_gemm_tn_pp(A, lda, B, ldb, C, ldc, M, N, Kc, 0, 0, 0, s);
    // big-tile LDS-DMA kernel: both sources bf16, output a multiple of 256 x 256, contraction a multiple of 32
    // (measured: 0.93 vs 1.02 ms on dW_ih 1024 x 256; on 256 x 128 tiles it LOSES to the register-staged kernel,
    //  0.33 vs 0.30 ms on dW_hh, so those shapes stay there)
    const bool dma = a_bf16 && b_bf16 && nt_dma_enabled() && M % 256 == 0 && N % 256 == 0 && Kc % 32 == 0;
    const int tm = dma ? 256 : 128, tn = dma ? 256 : 128;
    const int tiles = ((M + tm - 1) / tm) * ((N + tn - 1) / tn);
    const int target = dma ? 256 : 1024;                  // workgroups: one per CU for the big tiles (fewer, larger
                                                          // partial sums = fewer atomics), 4 per CU otherwise (round 4: the
                                                          // two small weight gradients of the mixed step, 128 x 256 and
                                                          // 128 x 64 over 2^20 rows, 0.214 + 0.113 -> 0.176 + 0.094 ms
                                                          // against 8 per CU; 2 per CU: the same; 1 per CU: slower)
    int nchunk = (target + tiles - 1) / tiles;
    int kchunk = (Kc + nchunk - 1) / nchunk;
    kchunk = ((kchunk + TK2 - 1) / TK2) * TK2;
    if (kchunk < 512) kchunk = 512;
    nchunk = (Kc + kchunk - 1) / kchunk;
    const int nchunk8 = ((nchunk + 7) / 8) * 8;
    TNArgs g{A, B, C, lda, ldb, ldc, M, N, Kc, kchunk, tiles};
    const dim3 grid((unsigned)(tiles * nchunk8));
    if (dma)                    hipLaunchKernelGGL((gemm_tn_dma_kernel<256, 256>), grid, dim3(1024), 0, s, g);
    else if (a_bf16 && b_bf16)  hipLaunchKernelGGL((gemm_tn_bf16_kernel<true, true>), grid, dim3(256), 0, s, g);
    else if (a_bf16)            hipLaunchKernelGGL((gemm_tn_bf16_kernel<true, false>), grid, dim3(256), 0, s, g);
    else if (b_bf16)            hipLaunchKernelGGL((gemm_tn_bf16_kernel<false, true>), grid, dim3(256), 0, s, g);
    else                        hipLaunchKernelGGL((gemm_tn_bf16_kernel<false, false>), grid, dim3(256), 0, s, g);
    LOB_CHECK_LAUNCH();
    return 0;
}

// dW_ih (D*4H x nx) and dW_hh (D x 4H x H) of one layer from one pass over dP; both outputs must be zeroed by the
// caller (split-k partial sums are added atomically).  H = 128, nx in {128, 256}, Bp % 32 == 0, T >= 2; or H = 256,
// D = 2, nx in {256, 512}, Bp % 64 == 0, (T * Bp) % 128 == 0.
extern "C" int lob_lstm_dw_bf16(const void* dP, int ldp, const void* X, int ldx, int nx, const void* Y, int ldy,
                                float* dWih, float* dWhh, int T, int Bp, int H, int D, void* stream) {
    if (!dP || !X || !Y || !dWih || !dWhh || T < 2 || Bp <= 0 || (D != 1 && D != 2)) return LOB_E_ARG;
    if (H == 256) {        // the reference's checkpoint size: 256 x 384 / 256 x 256 ping-pong tiles (gemm_pp.hip)
        if (!lob_dw_pp_ok(T, Bp, H, D, nx)) return LOB_E_SHAPE;
        if (ldp < D * 4 * H || ldx < nx || ldy < D * H) return LOB_E_SHAPE;
        if ((ldp % 8) || (ldx % 8) || (ldy % 8) || !al16(dP) || !al16(X) || !al16(Y)) return LOB_E_ALIGN;
        return lob_lstm_dw_pp(dP, ldp, X, ldx, nx, Y, ldy, dWih, dWhh, T, Bp, H, D, (hipStream_t)stream);
    }
    if (H != 128 || (nx != 128 && nx != 256) || (Bp % 32)) return LOB_E_SHAPE;
    if (ldp < D * 4 * H || ldx < nx || ldy < D * H) return LOB_E_SHAPE;
    if ((ldp % 8) || (ldx % 8) || (ldy % 8) || !al16(dP) || !al16(X) || !al16(Y)) return LOB_E_ALIGN;
    const int tiles = D * 2;                                  // 256 columns of dP each
    const long Kc = (long)T * Bp;
    int nchunk = (256 + tiles - 1) / tiles;                   // one workgroup per CU
    long kchunk = (Kc + nchunk - 1) / nchunk;
    kchunk = ((kchunk + 31) / 32) * 32;
    if (kchunk < 512) kchunk = 512;
    nchunk = (int)((Kc + kchunk - 1) / kchunk);
    const int nchunk8 = ((nchunk + 7) / 8) * 8;
    DWArgs g{(const __bf16*)dP, (const __bf16*)X, (const __bf16*)Y, dWih, dWhh, ldp, ldx, ldy, T, Bp, (int)kchunk, tiles};
    const dim3 grid((unsigned)(tiles * nchunk8));
    if (nx == 256) hipLaunchKernelGGL((lstm_dw_h128_kernel<256>), grid, dim3(512), 0, (hipStream_t)stream, g);
    else           hipLaunchKernelGGL((lstm_dw_h128_kernel<128>), grid, dim3(512), 0, (hipStream_t)stream, g);
    LOB_CHECK_LAUNCH();
    return 0;
}

extern "C" int lob_colsum_bf16(const void* A, int lda, int M, int N, float* out, void* stream) {
    if (!A || !out || M <= 0 || N <= 0 || lda < N) return LOB_E_ARG;
    const int ncb = (N + 63) / 64;
    int nrb = (M + 1023) / 1024;
    if (nrb > 1024) nrb = 1024;
    const int rpb = (M + nrb - 1) / nrb;
    nrb = (M + rpb - 1) / rpb;
    hipLaunchKernelGGL(colsum_bf16_kernel, dim3((unsigned)(ncb * nrb)), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const __bf16*>(A), lda, M, N, rpb, out);
    LOB_CHECK_LAUNCH();
    return 0;
}
