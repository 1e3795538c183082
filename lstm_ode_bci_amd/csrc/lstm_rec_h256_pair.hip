// Mixed-precision recurrent FORWARD for H = 256 on PAIRS of workgroups (round 3).
//
// The single-workgroup kernel (lstm_rec_h256_bf16.hip) streams W_hh of its direction -- 512 KB in bf16, more than a
// CU can hold -- out of L2 every step: measured (tools/h256_ablate.sh, bit 0) that stream is 0.77 ms of its 2.6 ms per
// launch at B = 4096.  Here TWO workgroups on two CUs share 64 batch rows of one direction: side s owns the hidden
// units [128 s, 128 s + 128), i.e. 512 of the 1024 gate columns, and keeps ITS half of W_hh (256 KB) in registers
// (+ LDS) for the whole launch -- nothing is streamed.  What the sides must exchange each step is the other half of
// h_t (64 rows x 128 units, bf16: 16 KB).  It goes through a small buffer with agent-scope 8-byte accesses in which
// every word carries its own sequence tag (4 bytes of payload + the step number: the "LL" idea of collective
// libraries): the writer just stores -- no completion wait, no separate flag -- and the reader knows a word is the one
// it needs when the tag matches.  (A first version with a counter per publish -- stores, s_waitcnt vmcnt(0), atomic
// add, partner polls, then loads -- drained the kernel's HBM streams every phase and put two memory round trips on
// the critical path: 4.6 ms per launch against 2.7 for the single-workgroup kernel.)  The latency is hidden by
// splitting the 64 rows into two 32-row tiles that are one phase apart: while tile A's half travels, both sides work
// on tile B.
//
// One phase (tile X, step) of a workgroup = 8 waves, wave w owns units 128 s + 16 w .. + 16 (64 gate columns):
//   b1. barrier    c. publish the OWN half of the other tile's newest h (complete in LDS since the barrier): 4 tagged
//      words per thread to the exchange buffer
//   e1. the half of the contraction over the OWN units' h_X(step-1) (32 of the 64 v_mfma_f32_16x16x32_bf16 per wave):
//      needs nothing from the partner
//   e2. tags of the partner's 4 words (loaded during the previous phase) checked -- a thread whose words are not there
//      yet polls (bounded) --, partner half -> LDS, barrier b2, the other half of the contraction
//   f. the kernel's bf16 outputs of the other tile (Y16 / dropped copy), 16 B per thread
//   g. first half of the cell update, then the loads of the partner's 4 words of the other tile are issued (the
//      partner published them at ITS step c, a matrix phase ago), second half of the cell update
// No deadlock without a co-residency guarantee: partners are 8 workgroup ids apart (same XCD under round-robin
// dispatch), ids are dispatched in order, so every resident set is a prefix of the grid whose complete pairs always
// make progress and free their CUs; a wait that still exceeds its bound (seconds) writes the workspace's error word
// and the thread stops waiting for the rest of the launch (garbage results, no hang): the host reads that word back
// asynchronously and raises at its next call (ops.py).
//
// Every global layout is the single-workgroup kernel's (fragment-order P / saved gates / cell states, the host's
// fragment-order bf16 W_hh image, row-major Y / Y16 / Yd, the same dropout mask): the BPTT kernel, the gate GEMM and
// the host do not change, and LOB_VAR_H256_PAIR = 0 keeps the old kernel as the twin (tests/test_gpu_pair.py: equal
// within fp32 rounding of the different MFMA shape, 16x16x32 here against 32x32x16 there).
#include "lob_common.h"
#include <type_traits>

// Diagnostic builds only (garbage results): bit 0 = no waiting for the partner, bit 1 = no exchange traffic at all
#ifndef LOB_ABL_PAIR
#define LOB_ABL_PAIR 0
#endif
#ifndef LOB_PAIR_NRF
#define LOB_PAIR_NRF 22        // W_hh fragments of a wave kept in registers (of 32; the rest in its private LDS block)
#endif

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int HH = 256, NW = 8;
constexpr int HB_LD = 264;             // h tile row stride in bf16 (528 B = 33 x 16 B)
constexpr int TILE = 32 * HB_LD;       // one 32-row h tile
constexpr int SPIN_LIMIT = 1 << 21;    // polls of ~2 us (one memory round trip each): seconds

__device__ __forceinline__ f32x4 mfma16(bf16x8 a, bf16x8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}

// agent-scope (cross-CU, cross-XCD) 8-byte words of the exchange buffer: {payload, tag}
__device__ __forceinline__ void st_word(unsigned long long* p, unsigned payload, unsigned tag) {
    __hip_atomic_store(p, ((unsigned long long)tag << 32) | payload, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned long long ld_word(const unsigned long long* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// NRF: how many of a wave's 32 W_hh fragments (k-step 8 x gate 4, 1 KB each) live in registers; the rest in the wave's
// private LDS block.
template <bool SAVE, bool YF32, bool Y16, bool DROP, typename CE, int NRF>
__global__ __launch_bounds__(512) void lstm_rec_fwd_h256_pair_kernel(
    __bf16* __restrict__ P, const __bf16* __restrict__ Wb, float* __restrict__ Y, CE* __restrict__ Csave,
    __bf16* __restrict__ Y16p, __bf16* __restrict__ Yd, float drop_p, uint64_t seed, int T, int Bp, int D,
    unsigned* __restrict__ err, unsigned long long* __restrict__ xbuf) {
    __shared__ __attribute__((aligned(16))) __bf16 hs[2 * 2 * TILE];               // [tile][parity][32][HB_LD]
    __shared__ __attribute__((aligned(16))) __bf16 wl[NRF < 32 ? NW * (32 - NRF) * 512 : 8];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c16 = lane & 15, rq = lane >> 4;
    // workgroup id -> (pair, side): partners 8 ids apart
    const int id = blockIdx.x, side = (id >> 3) & 1, pair = (id >> 4) * 8 + (id & 7);
    const int NPB = Bp >> 6, npairs = NPB * D;          // 64-row blocks per direction
    if (pair >= npairs) return;
    const int d = pair / NPB, pb = pair - d * NPB;
    const int NBT = Bp >> 5;                             // 32-row tiles of the global layouts
    const int DH = D * HH;

    for (int i = tid; i < 2 * 2 * TILE; i += 512) hs[i] = (__bf16)0.f;

    // ---- W_hh fragments from the host's fragment-order image [D][wave 8][ks 16][gate 4][lane 64][8] (32-unit waves,
    // 16-deep k-steps): this wave's B operand of v_mfma_16x16x32 for (gate g, k-step ks) is, per lane (c16, rq),
    // W_hh[g H + 128 s + 16 w + c16][32 ks + 8 rq .. + 7] = old wave 4 s + (w >> 1), old k-step 2 ks + (rq >> 1), old lane
    // 16 (w & 1) + c16 + 32 (rq & 1)
    const int wold = 4 * side + (w >> 1);
    const int lane_old = 16 * (w & 1) + c16 + 32 * (rq & 1);
    bf16x8 wr[NRF > 0 ? NRF : 1];
    __bf16* wlw = wl + (size_t)w * ((32 - NRF) * 512) + lane * 8;
    {
        const __bf16* wsrc = Wb + ((size_t)d * NW + wold) * (16 * 4 * 512) + (size_t)lane_old * 8;
#pragma unroll
        for (int f = 0; f < 32; ++f) {
            // slot f >> 2 -> k-step: slots 0..3 are the k-steps of the OWN units' h (k = 128 s ..), slots 4..7 the partner's
            const int slot = f >> 2, g = f & 3;
            const int ks = slot < 4 ? 4 * side + slot : 4 * (side ^ 1) + slot - 4;
            const bf16x8 v = *reinterpret_cast<const bf16x8*>(wsrc + (size_t)((2 * ks + (rq >> 1)) * 4 + g) * 512);
            if (f < NRF) wr[f < NRF ? f : 0] = v;
            else *reinterpret_cast<bf16x8*>(wlw + (f - NRF) * 512) = v;
        }
    }

    // ---- global addressing (the single-workgroup kernel's layouts)
    const size_t pstep = (size_t)NBT * NW * 4096, cstep = (size_t)NBT * NW * 1024;
    // element offset of this lane inside an old wave's 4096-element block, without gate / row-block terms
    const unsigned p_lane = (unsigned)(lane_old * 8 + 4 * (rq >> 1));
    const int t_first = d ? T - 1 : 0, dt = d ? -1 : 1;
    auto pblock = [&](int X, int t) -> __bf16* {
        return P + ((((size_t)d * T + t) * NBT + (2 * pb + X)) * NW + wold) * 4096 + p_lane;
    };
    // publish / staging geometry: thread -> (row, 16-B chunk of the 128-unit half)
    const int prow = tid >> 4, pch = tid & 15;
    // exchange buffer [pair][side 2][tile 2][step parity 2][word 4][thread 512]: word i of a thread's 16 bytes (a wave's
    // store is 512 B).  Two slots per tile: the reader checks (and, when it polls, re-reads) the words of h_X(step)
    // as late as the middle of its phase (X, step+1), by when a writer that runs ahead may already publish
    // h_X(step+1); it cannot reach h_X(step+2) before the reader is through (it waits for the reader's own publish)
    unsigned long long* x_own = xbuf + (((size_t)pair * 2 + side) * 4) * 2048 + tid;
    const unsigned long long* x_oth = xbuf + (((size_t)pair * 2 + (side ^ 1)) * 4) * 2048 + tid;

    float c[2][2][4];
#pragma unroll
    for (int X = 0; X < 2; ++X)
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
#pragma unroll
            for (int j = 0; j < 4; ++j) c[X][rb][j] = 0.f;

    bf16x4 praw[2][4][2];          // [tile][gate][row block]: P of the tile's next step, unconverted
    auto load_p = [&](int X, int t) {
        const __bf16* pp = pblock(X, t);
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int rb = 0; rb < 2; ++rb)
                praw[X][g][rb] = *reinterpret_cast<const bf16x4*>(pp + g * 1024 + rb * 512);
    };
    load_p(0, t_first);
    load_p(1, t_first);
    unsigned long long lw[4] = {0ull, 0ull, 0ull, 0ull};     // the partner's 4 tagged words of the tile handled next
    bool dead = false;             // a wait timed out: no further waits (the launch is garbage and says so in *err)
    __syncthreads();

    // own half of h_O(step_o) (complete in LDS) -> exchange buffer (if the partner still needs it) and bf16 outputs
    auto publish = [&](int O, int step_o, bool exchange, bool outputs) {
        const int t_o = t_first + dt * step_o;
        const __bf16* src = hs + (O * 2 + (step_o & 1)) * TILE + prow * HB_LD + 128 * side + pch * 8;
        const bf16x8 hv = *reinterpret_cast<const bf16x8*>(src);
        if (exchange) {
            const u32x4 pv = __builtin_bit_cast(u32x4, hv);
#pragma unroll
            for (int i = 0; i < 4; ++i) st_word(x_own + (O * 2 + (step_o & 1)) * 2048 + i * 512, pv[i], (unsigned)(step_o + 1));
        }
        if ((Y16 || DROP) && outputs) {
            const size_t o = ((size_t)t_o * Bp + (2 * pb + O) * 32 + prow) * DH + d * HH + 128 * side + pch * 8;
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
    };

    auto phase = [&](int X, int step) {
        const int O = X ^ 1, par = step & 1;
        const int t = t_first + dt * step;
        const int step_o = X ? step : step - 1;            // newest finished step of the other tile
        __bf16* hx_prev = hs + (X * 2 + (par ^ 1)) * TILE;
        // b1. own halves in LDS are complete (the other tile's newest h for the publish; h_X(step-1) since long)
        __syncthreads();
        // c.
        const bool have_o = step_o >= 0;
        const bool exch_o = have_o && step_o + 1 < T && !(LOB_ABL_PAIR & 2);
        if (exch_o) publish(O, step_o, true, false);
        // d. accumulators start from P(X, step); then the registers take P(X, step+1)
        f32x4 acc[4][2];
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int rb = 0; rb < 2; ++rb)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[g][rb][j] = (float)praw[X][g][rb][j];
        // e. z = P + h_X(step-1) W_hh^T in two halves of the contraction: the OWN units' k-range needs nothing from the
        // partner and runs first; then the partner's words (loaded during the previous phase) are checked, staged, and
        // the second half follows after a barrier
        const __bf16* arow = hx_prev + c16 * HB_LD + 8 * rq;
        auto mfma_half = [&](int slot0, int ks0) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const bf16x8 a0 = *reinterpret_cast<const bf16x8*>(arow + 32 * (ks0 + i));
                const bf16x8 a1 = *reinterpret_cast<const bf16x8*>(arow + 16 * HB_LD + 32 * (ks0 + i));
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int f = (slot0 + i) * 4 + g;
                    bf16x8 b;
                    if (f < NRF) b = wr[f < NRF ? f : 0];
                    else b = *reinterpret_cast<const bf16x8*>(wlw + (f - NRF) * 512);
                    acc[g][0] = mfma16(a0, b, acc[g][0]);
                    acc[g][1] = mfma16(a1, b, acc[g][1]);
                }
            }
        };
        mfma_half(0, 4 * side);
        if (step > 0 && !(LOB_ABL_PAIR & 3)) {      // every word must carry the tag of h_X(step-1)
            const unsigned need = (unsigned)step;
            int spins = 0;
            while (!dead && ((unsigned)(lw[0] >> 32) != need || (unsigned)(lw[1] >> 32) != need ||
                             (unsigned)(lw[2] >> 32) != need || (unsigned)(lw[3] >> 32) != need)) {
                __builtin_amdgcn_s_sleep(1);
#pragma unroll
                for (int i = 0; i < 4; ++i) lw[i] = ld_word(x_oth + (X * 2 + ((step - 1) & 1)) * 2048 + i * 512);
                if (++spins > SPIN_LIMIT) {
                    dead = true;
                    __hip_atomic_store(err, (unsigned)step, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
        }
        if (step > 0) {
            const u32x4 stage = {(unsigned)lw[0], (unsigned)lw[1], (unsigned)lw[2], (unsigned)lw[3]};
            *reinterpret_cast<u32x4*>(hx_prev + prow * HB_LD + 128 * (side ^ 1) + pch * 8) = stage;
        }
        // b2.
        __syncthreads();
        mfma_half(4, 4 * (side ^ 1));
        // the kernel's bf16 outputs of the other tile's newest h (off the exchange's critical path)
        if (have_o) publish(O, step_o, false, true);
        // g. cell update of (X, step); between its halves the loads of the partner's words of the other tile go out
        __bf16* hx = hs + (X * 2 + par) * TILE + 4 * rq * HB_LD + 128 * side + 16 * w + c16;
        float* yrow = YF32 ? Y + ((size_t)t * Bp + (2 * pb + X) * 32 + 4 * rq) * DH + d * HH + 128 * side + 16 * w + c16 : nullptr;
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) {
            if (rb == 1 && exch_o && !(LOB_ABL_PAIR & 1)) {
                asm volatile("" ::: "memory");
#pragma unroll
                for (int i = 0; i < 4; ++i) lw[i] = ld_word(x_oth + (O * 2 + (step_o & 1)) * 2048 + i * 512);
                asm volatile("" ::: "memory");
            }
            // P of (X, step+1): issued AFTER the exchange loads -- a wave's loads return in order, and behind this HBM
            // stream the partner's words (an L2 hit) would arrive several microseconds late
            if (rb == 1 && step + 1 < T) { load_p(X, t + dt); asm volatile("" ::: "memory"); }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float ig = fast_sigmoid(acc[0][rb][j]);
                const float fg = fast_sigmoid(acc[1][rb][j]);
                const float gg = fast_tanh(acc[2][rb][j]);
                const float og = fast_sigmoid(acc[3][rb][j]);
                c[X][rb][j] = __builtin_fmaf(fg, c[X][rb][j], ig * gg);
                const float h = og * fast_tanh(c[X][rb][j]);
                hx[(16 * rb + j) * HB_LD] = (__bf16)h;
                if (YF32) yrow[(size_t)(16 * rb + j) * DH] = h;
                if (SAVE) { acc[0][rb][j] = ig; acc[1][rb][j] = fg; acc[2][rb][j] = gg; acc[3][rb][j] = og; }
            }
        }
        if (SAVE) {
            __bf16* gp = pblock(X, t);
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int rb = 0; rb < 2; ++rb) {
                    bf16x4 v = {(__bf16)acc[g][rb][0], (__bf16)acc[g][rb][1], (__bf16)acc[g][rb][2], (__bf16)acc[g][rb][3]};
                    __builtin_nontemporal_store(v, reinterpret_cast<bf16x4*>(gp + g * 1024 + rb * 512));
                }
            CE* cp = Csave + ((((size_t)d * T + t) * NBT + (2 * pb + X)) * NW + wold) * 1024;
#pragma unroll
            for (int rb = 0; rb < 2; ++rb) {
                if constexpr (sizeof(CE) == 4) {     // fp32 [q 4][lane 64][4]: q = 2 rb + (rq >> 1)
                    f32x4 v = {c[X][rb][0], c[X][rb][1], c[X][rb][2], c[X][rb][3]};
                    *reinterpret_cast<f32x4*>(cp + (2 * rb + (rq >> 1)) * 256 + lane_old * 4) = v;
                } else {                             // bf16 [q pair 2][lane 64][8]: the element order of one saved gate
                    bf16x4 v = {(__bf16)c[X][rb][0], (__bf16)c[X][rb][1], (__bf16)c[X][rb][2], (__bf16)c[X][rb][3]};
                    __builtin_nontemporal_store(v, reinterpret_cast<bf16x4*>(cp + rb * 512 + p_lane));
                }
            }
        }
        (void)cstep; (void)pstep;
    };

    for (int step = 0; step < T; ++step) {
        phase(0, step);
        phase(1, step);
    }
    // the last step of tile B: its bf16 outputs (tile A's went out in the last phase)
    __syncthreads();
    publish(1, T - 1, false, true);
}

}  // namespace

// Workspace of the pair kernel: a 256-byte header (word 0 = error word), then the exchange buffer
// [npairs][side 2][tile 2][parity 2][word 4][thread 512] of 8-byte {payload, tag} words; all of it zeroed by the entry point (a
// tag of a previous launch must not look valid).
constexpr size_t PAIR_HDR = 256;

extern "C" size_t lob_rec_pair_ws_bytes(int Hh, int Bp, int D) {
    if (Hh != 256 || Bp <= 0 || (Bp % 64) || (D != 1 && D != 2)) return 0;
    return PAIR_HDR + (size_t)(Bp / 64) * D * 2 * 2 * 2 * 2048 * sizeof(unsigned long long);
}

// Internal entry point used by lob_lstm_rec_fwd_bf16_ws (lstm_rec_bf16.hip) at H = 256 when a workspace is given.
int lob_rec_fwd_h256_pair(void* P, const void* Whh16, float* Y, void* Csave, int c_bf16, void* Y16, void* Yd, float drop_p,
                          uint64_t seed, int T, int Bp, int D, int save, void* ws, size_t ws_bytes, hipStream_t s) {
    if ((Bp % 64) || ws_bytes < lob_rec_pair_ws_bytes(256, Bp, D) || (reinterpret_cast<uintptr_t>(ws) & 15)) return LOB_E_SHAPE;
    if (c_bf16 && !save) return LOB_E_SHAPE;
    { hipError_t e = hipMemsetAsync(ws, 0, lob_rec_pair_ws_bytes(256, Bp, D), s); if (e != hipSuccess) return (int)e; }
    unsigned* err = reinterpret_cast<unsigned*>(ws);
    unsigned long long* xbuf = reinterpret_cast<unsigned long long*>(reinterpret_cast<char*>(ws) + PAIR_HDR);
    const int npairs = (Bp / 64) * D;
    const dim3 grid((npairs + 7) / 8 * 16), block(512);
    __bf16* y16 = reinterpret_cast<__bf16*>(Y16);
    __bf16* yd = reinterpret_cast<__bf16*>(Yd);
#define LOB_FWD(SV, YF, Y6, DR, CE) hipLaunchKernelGGL((lstm_rec_fwd_h256_pair_kernel<SV, YF, Y6, DR, CE, LOB_PAIR_NRF>), grid, block, 0, s, \
        reinterpret_cast<__bf16*>(P), reinterpret_cast<const __bf16*>(Whh16), Y, reinterpret_cast<CE*>(Csave), y16, yd,      \
        drop_p, seed, T, Bp, D, err, xbuf)
#define LOB_FWD_OUT(SV, CE) do {                                                     \
        if (Y && !y16 && !yd) LOB_FWD(SV, true, false, false, CE);                   \
        else if (Y && y16 && !yd) LOB_FWD(SV, true, true, false, CE);                \
        else if (Y && !y16 && yd) LOB_FWD(SV, true, false, true, CE);                \
        else if (Y && y16 && yd) LOB_FWD(SV, true, true, true, CE);                  \
        else if (!Y && y16 && !yd) LOB_FWD(SV, false, true, false, CE);              \
        else LOB_FWD(SV, false, true, true, CE); } while (0)
    if (save && c_bf16) LOB_FWD_OUT(true, __bf16); else if (save) LOB_FWD_OUT(true, float); else LOB_FWD_OUT(false, float);
#undef LOB_FWD_OUT
#undef LOB_FWD
    LOB_CHECK_LAUNCH();
    return 0;
}
