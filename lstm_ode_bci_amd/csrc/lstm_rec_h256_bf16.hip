// Mixed-precision recurrent kernels for H = 256 (the size of the reference's real checkpoints, 04_lstm_model.py:877).
//
// W_hh of one direction is 1024 x 256: 512 KB in bf16 -- more than a CU's registers + LDS can hold beside the
// accumulators, so (unlike H = 128) the B operand of h W_hh^T is STREAMED from L2 every step: a workgroup is 32 batch
// rows x one direction, 8 waves (two per SIMD), wave w owns hidden units [32w, 32w+32) = 128 gate columns, and per
// step each lane loads its 64 fragments (16 B each, straight global -> VGPR, no LDS) in 16 groups of 4, three groups  The weights arrive pre-converted to bf16 (the host
// casts them once per call), in the orientation each kernel reads with 16 contiguous bytes per lane: [4H][H] for
// the forward (k = h index contiguous), [H][4H] (W_hh^T) for BPTT (k = gate index contiguous).
// The fp32 streaming kernels this replaces in mixed mode (lstm_rec_stream.hip) moved 1 MB per step per CU with one
// k-block in flight and ran 12.9 ms (forward) / 17.1 ms (BPTT) per launch at B = 4096.
// Everything else -- fragment-order P / saved gates (bf16, [gate][q pair][lane][8]) / c (fp32), fp32 cell state and
// gradient carries, the bf16 dgates tile that is both MFMA operand and dP image, fused dropout copy -- is as in
// lstm_rec_bf16.hip.
#include "lob_common.h"
#include <type_traits>
// Cache policy: the W_hh stream of these kernels lives in L2 (every workgroup of a direction re-reads the same 512 KB each
// step), while the saved gates, cell states and dY are read ONCE and dP is written once -- left at the default policy
// those streams keep evicting the weights.  Non-temporal hints on them (BPTT): 3.40 -> 2.95 ms per launch, same-box A/B
// (G loads alone: 3.05; the same hint on the weight loads: 4.83).
#ifndef LOB_NT_G
#define LOB_NT_G true
#endif
#ifndef LOB_NT_CDY
#define LOB_NT_CDY true
#endif
#ifndef LOB_NT_DP
#define LOB_NT_DP true
#endif
#ifndef LOB_NT_FST
#define LOB_NT_FST true      // forward: saved gates / cell states written once: 2.88 -> 2.79 ms per launch
#endif
#ifndef LOB_NT_P
#define LOB_NT_P false
#endif
// Diagnostic builds only (tools/h256_ablate.sh; results are garbage): bit 0 = the per-step W_hh stream is not issued
// (the MFMAs run on whatever the ring registers hold), bit 1 = the cell update's transcendentals are skipped.
#ifndef LOB_ABL_H256
#define LOB_ABL_H256 0
#endif
// Issue order (round 4).  A wave's vector-memory operations retire IN ORDER (one vmcnt counter for loads, stores and
// LDS-DMA): a W_hh fragment load (an L2 hit, ~0.2 us) issued behind an HBM load or a store cannot be waited for before
// that older operation has completed (~1-2 us under load), so every HBM operation issued inside or just in front of the
// per-step W stream stalls the stream for its own latency.  Both kernels therefore issue their HBM loads at the START of
// a W-free stretch that is about an HBM latency long:
//   forward: P of the next step is loaded right AFTER the step's MFMA loop and lands during the cell update (it used to
//            be loaded in front of the MFMA loop: 2.64 -> 2.29 ms per launch, same-box A/B, bit-identical);
//   BPTT:    the dgates phase runs in two halves (tile rows r < 8 / r >= 8 of a lane) and the next step's saved gates,
//            cell states and dY of a half are loaded into the registers that half has just released -- the first half's
//            in the middle of the dgates phase, the second half's after the MFMA loop (2.84 -> 2.71 ms).
// Measured and dropped (profiles/r04_h256_issue_order.txt): the forward's bf16 row outputs deferred into the next cell
// update (slower: their dropout hashes then compete with the cell update instead of covering W latency); touching the
// next step's HBM lines ahead of the real loads (slower: +25 % line fills through the CU's one memory pipe); refilling
// the BPTT ring fragment by fragment, 448 instead of 256 MFMA cycles ahead (no change: BPTT is not latency-bound on W).
// LOB_H256_NQR: register-resident W groups of the forward next to the NQL groups in LDS (fragment ring, see the kernel)
#ifndef LOB_H256_NQR
#define LOB_H256_NQR 5
#endif

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int HH = 256, NW = 8;
constexpr int HB_LD = 264;         // h tile row stride in bf16 (528 B = 33 x 16 B, odd -> conflict-free b128)
constexpr int DGB_LD = 1032;       // dgates tile row stride in bf16 (2064 B = 129 x 16 B)

__device__ __forceinline__ f32x16 mfma_bf16(bf16x8 a, bf16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

// compile-time loop: the body sees its index as a constant (register arrays stay registers)
template <int B, int E, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (B < E) { f(std::integral_constant<int, B>{}); static_for<B + 1, E>(f); }
}

// Cell backward of one element, every contraction written out (explicit fma / plain products): hipcc fuses an unpinned
// expression differently in different kernels, and the full-tile, the part-tile and the fp32-cell-state instantiations must
// round alike (tests compare them bit for bit).
__device__ __forceinline__ void lob_dgate(float ig, float fg, float gg, float og, float dh, float tc, float cpv, float& dcarry,
                                          float& v0, float& v1, float& v2, float& v3) {
    const float dc = __builtin_fmaf(dh * og, __builtin_fmaf(-tc, tc, 1.f), dcarry);
    dcarry = dc * fg;
    v0 = (dc * gg) * __builtin_fmaf(-ig, ig, ig);
    v1 = (dc * cpv) * __builtin_fmaf(-fg, fg, fg);
    v2 = (dc * ig) * __builtin_fmaf(-gg, gg, 1.f);
    v3 = (dh * tc) * __builtin_fmaf(-og, og, og);
}

struct Raw { bf16x8 v[8]; };       // one wave's [4 gates][2 q pairs] x 8 elements per lane, unconverted

template <bool NT = false>
__device__ __forceinline__ void load_raw(const __bf16* p, unsigned off8, Raw& r) {
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int pq = 0; pq < 2; ++pq) {
            if constexpr (NT) r.v[2 * g + pq] = __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>((p + g * 1024 + pq * 512) + off8));
            else              r.v[2 * g + pq] = *reinterpret_cast<const bf16x8*>((p + g * 1024 + pq * 512) + off8);
        }
}

// CE: storage type of the saved cell states (fp32 [q 4][lane 64][4], or bf16 in the element order of one saved gate,
// [q pair 2][lane 64][8]: ops.C_BF16, as in the H = 128 kernels -- BPTT only uses c_t inside tanh and as a factor next
// to bf16 gate values; the state carried through time stays fp32 in registers)
// NQL: how many of the 16 k-step groups of a wave's W_hh fragments stay in LDS for the whole launch (a wave's
// private 4 KB per group): the free LDS next to the h tiles holds 3 of them (96 KB), i.e. 3/16 of the per-step weight
// stream never leaves the CU again.  The other groups are streamed as before, three ahead; with 13 streamed groups
// the cyclic order needs a FIFTH ring buffer (positions 0..11 use buffer pos % 4, position 12 buffer 4: any four
// consecutive positions of ..., 11, 12, 0, 1, ... then sit in different buffers).  Same MFMAs in the same order.
// PARTS = 4 (round 4, LOB_VAR_REC_HALF; few tiles: the reference's own training batch of 512 windows is 16 tiles per direction):
// FOUR workgroups share each 32-row tile.  Register r of the 32x32 D layout is tile row (r & 3) + 8 (r >> 2) + 4 hi, so part p =
// r >> 2 is the eight rows 8 p .. 8 p + 7: workgroup p loads / saves / updates only the four elements r = 4 p + e of each lane
// and stores only its rows; the other rows of ITS h tile stay zero (their MFMA work is wasted), the W stream is the same per
// workgroup.  A quarter of the activations on every step's serial chain; same arithmetic per row: bit-identical to full tiles.
template <bool SAVE, bool YF32, bool Y16, bool DROP, typename CE = float, int NQL = 0, int PARTS = 1>
__global__ __launch_bounds__(512, 2) void lstm_rec_fwd_h256_bf16_kernel(
    __bf16* __restrict__ P, const __bf16* __restrict__ Wb, float* __restrict__ Y, CE* __restrict__ Csave,
    __bf16* __restrict__ Y16p, __bf16* __restrict__ Yd, float drop_p, uint64_t seed, int T, int Bp) {
    static_assert(PARTS == 1 || PARTS == 4, "full tiles or four workgroups per tile");
    typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
    __shared__ __attribute__((aligned(16))) __bf16 hs[2 * 32 * HB_LD];
    __shared__ __attribute__((aligned(16))) __bf16 wl[NQL > 0 ? NW * NQL * 4 * 512 : 8];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int bt = (int)blockIdx.x / PARTS, part = (int)blockIdx.x % PARTS;
    const int d = blockIdx.y, D = gridDim.y, NBT = (int)gridDim.x / PARTS;
    const int l31 = lane & 31, hi = lane >> 5;
    // PARTS == 4: this workgroup's elements sit in q pair part >> 1 of the fragment-order blocks, at 4 (part & 1) .. + 3
    const unsigned poff = (unsigned)((part >> 1) * 512 + lane * 8 + (part & 1) * 4);

    for (int i = tid; i < 2 * 32 * HB_LD; i += 512) hs[i] = (__bf16)0.f;
    float c[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) c[r] = 0.f;

    const size_t pstep = (size_t)NBT * NW * 4096, cstep = (size_t)NBT * NW * 1024;
    __bf16* pblk = P + (((size_t)d * T * NBT + bt) * NW + w) * 4096;
    CE* cblk = SAVE ? Csave + (((size_t)d * T * NBT + bt) * NW + w) * 1024 : nullptr;
    const unsigned off8 = lane * 8, off4 = lane * 4;
    const int DH = D * HH;
    const unsigned y_off = (unsigned)(4 * hi * DH + l31);
    // B fragments, pre-arranged by the host in FRAGMENT ORDER [D][wave][ks 16][gate 4][lane 64][8]: element =
    // W_hh[g*H + 32w + (lane & 31)][16 ks + 8 (lane >> 5) + j].  A wave-load is then 1 KB contiguous (8 whole cache
    // lines); read straight from the [4H][H] matrix each load touched 32 lines for 32 B each, the lines did not
    // survive in L1 until their next k-step, and L2 delivered 4x the bytes (18 us per step)
    const __bf16* wwave = Wb + ((size_t)d * NW + w) * (16 * 4 * 64 * 8);
    unsigned w_off = (unsigned)(lane * 8);
    const int t_first = d ? T - 1 : 0, dt = d ? -1 : 1;

    Raw pn;
    bf16x4 pn4[4];                                     // PARTS == 4: the lane's four elements of each gate
    auto load_p4 = [&](int t) {
#pragma unroll
        for (int g = 0; g < 4; ++g) pn4[g] = *reinterpret_cast<const bf16x4*>((pblk + (size_t)t * pstep + g * 1024) + poff);
    };
    if constexpr (PARTS == 4) load_p4(t_first);
    else load_raw<LOB_NT_P>(pblk + (size_t)t_first * pstep, off8, pn);
    // W stream: group q = k-step q x 4 gates = 4 fragments; buffer q & 3; THREE groups in flight ahead of the one
    // being consumed (cyclic: the weights are the same every step, so the tail of a step prefetches the head of the
    // next).  The stream is latency-bound: what counts is bytes in flight per CU (12 KB per wave here).
    constexpr int NSQ = 16 - NQL;                      // streamed groups per step
    constexpr int NRB = (NSQ % 4 == 0) ? 4 : 5;        // ring buffers (see above)
    static_assert(NSQ % 4 == 0 || NSQ % 4 == 1, "ring assignment: pos % 4, last position in a fifth buffer");
    // NQL == 3 (the product configuration): the ring is refilled fragment by fragment -- 8 buffers = 32 registers, 7
    // fragments = 448 MFMA cycles ahead, instead of 5 x 4 buffers = 80 registers three groups ahead -- and the registers
    // this frees hold NQR MORE groups of the wave's fragments for the whole launch: 32 of its 64 fragments are streamed
    // per step instead of 52 (saving forward 2.28 -> 2.13 ms, inference 1.91 -> 1.72).  NQL == 0 keeps the group ring: the
    // all-streamed twin (LOB_VAR_H256_LDSW = 0)
    constexpr bool FRAG_RING = NQL == 3;
    constexpr int NQR = FRAG_RING ? LOB_H256_NQR : 0;
    constexpr int NSF = (16 - NQL - NQR) * 4;          // streamed fragments per step (fragment ring)
    static_assert(!FRAG_RING || NSF % 8 == 0, "fragment ring: a fragment must keep its buffer across steps");
    bf16x8 wb[FRAG_RING ? 2 : NRB][4];
    bf16x8 wr[NQR > 0 ? NQR : 1][4];
    auto load_w = [&](int q, bf16x8 (&dst)[4]) {
#pragma unroll
        for (int g = 0; g < 4; ++g)
            dst[g] = *reinterpret_cast<const bf16x8*>((wwave + (q * 4 + g) * 512) + w_off);
    };
    auto load_wf = [&](int i) {                        // streamed fragment i -> buffer i % 8
        wb[(i & 7) >> 2][i & 3] = *reinterpret_cast<const bf16x8*>((wwave + ((NQL + NQR) * 4 + i) * 512) + w_off);
    };
    // streamed position p (0 .. NSQ-1) = group NQL + p, ring buffer rb(p)
    auto rb = [](int p) { return (NSQ % 4 == 1 && p == NSQ - 1) ? 4 : (p & 3); };
    __bf16* wlw = wl + (size_t)w * (NQL * 4 * 512) + lane * 8;      // this wave's stationary fragments
    if constexpr (NQL > 0) {
#pragma unroll
        for (int f = 0; f < NQL * 4; ++f)
            *reinterpret_cast<bf16x8*>(wlw + f * 512) = *reinterpret_cast<const bf16x8*>((wwave + f * 512) + w_off);
    }
    if constexpr (FRAG_RING) {
#pragma unroll
        for (int r = 0; r < NQR; ++r)
#pragma unroll
            for (int g = 0; g < 4; ++g)
                wr[r][g] = *reinterpret_cast<const bf16x8*>((wwave + ((NQL + r) * 4 + g) * 512) + lane * 8);
#pragma unroll
        for (int i = 0; i < 7; ++i) load_wf(i);
    } else {
        load_w(NQL + 0, wb[rb(0)]);
        load_w(NQL + 1, wb[rb(1)]);
        load_w(NQL + 2, wb[rb(2)]);
    }
    __syncthreads();

    // the bf16 row segments of a finished step (32 rows x 512 B) from its h tile: Y16 = bf16(h), Yd = bf16(dropout(h))
    auto emit_rows = [&](int buf, int t) {
        const __bf16* hsrc = hs + buf * 32 * HB_LD;
#pragma unroll
        for (int i = 0; i < (PARTS == 4 ? 1 : 2); ++i) {
            const int idx = tid + 512 * i, c8 = (idx & 31) * 8;
            const int row = PARTS == 4 ? 8 * part + (tid >> 5) : idx >> 5;      // PARTS == 4: this workgroup's eight rows
            if (PARTS == 4 && tid >= 256) continue;
            const bf16x8 hv = *reinterpret_cast<const bf16x8*>(hsrc + row * HB_LD + c8);
            const size_t o = ((size_t)t * Bp + bt * 32 + row) * DH + d * HH + c8;
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
    int cur = 0;
    for (int step = 0; step < T; ++step) {
        const int t = t_first + dt * step;
        f32x16 acc[4];
        if constexpr (PARTS == 4) {      // P into this part's four registers (selects on the uniform part index), zero elsewhere
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int k = 0; k < 4; ++k)
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[g][4 * k + e] = part == k ? (float)pn4[g][e] : 0.f;
        } else {
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int pq = 0; pq < 2; ++pq)
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[g][8 * pq + e] = (float)pn.v[2 * g + pq][e];
        }
        const __bf16* hrow = hs + cur * 32 * HB_LD + l31 * HB_LD + 8 * hi;
        // the weights are loop-invariant, and hipcc would hoist all 64 fragment loads out of the time loop (256
        // registers -> scratch); an opaque no-op on the lane offset ties every step's loads to that step
        asm volatile("" : "+v"(w_off));
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            if (q < NQL) {                              // stationary group: B fragments from this wave's LDS block
                const bf16x8 a = *reinterpret_cast<const bf16x8*>(hrow + 16 * q);
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    acc[g] = mfma_bf16(a, *reinterpret_cast<const bf16x8*>(wlw + (q * 4 + g) * 512), acc[g]);
                continue;
            }
            if constexpr (FRAG_RING) {
                const bf16x8 a = *reinterpret_cast<const bf16x8*>(hrow + 16 * q);
                if (q < NQL + NQR) {                    // register-resident group
#pragma unroll
                    for (int g = 0; g < 4; ++g) acc[g] = mfma_bf16(a, wr[q - NQL][g], acc[g]);
                    continue;
                }
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int i = (q - NQL - NQR) * 4 + g;
                    if constexpr (!(LOB_ABL_H256 & 1)) load_wf((i + 7) % NSF);      // into the buffer MFMA i - 1 has just read
                    __builtin_amdgcn_sched_barrier(0);
                    acc[g] = mfma_bf16(a, wb[(i & 7) >> 2][i & 3], acc[g]);
                    __builtin_amdgcn_sched_barrier(0);
                }
                continue;
            }
            const int p = q - NQL, pnx = (p + 3) % NSQ; // three streamed groups ahead, cyclic over the steps
            if constexpr (!(LOB_ABL_H256 & 1)) load_w(NQL + pnx, wb[rb(pnx)]);
            __builtin_amdgcn_sched_barrier(0);          // pin the order: issue the prefetch, then consume group q
            const bf16x8 a = *reinterpret_cast<const bf16x8*>(hrow + 16 * q);
#pragma unroll
            for (int g = 0; g < 4; ++g) acc[g] = mfma_bf16(a, wb[rb(p)][g], acc[g]);
            __builtin_amdgcn_sched_barrier(0);
        }
        // ---- the W-free window of the step starts here (the last W loads issued are the prefetch of the next step's
        // first fragments): P of the next step is requested now and lands during the cell update
        if constexpr (PARTS == 4) { if (step + 1 < T) load_p4(t + dt); }
        else { if (step + 1 < T) load_raw<LOB_NT_P>(pblk + (size_t)(t + dt) * pstep, off8, pn); }
        __bf16* hnext = hs + (cur ^ 1) * 32 * HB_LD + 32 * w + l31 + 4 * hi * HB_LD;
        float* yrow = Y + ((size_t)t * Bp + bt * 32) * DH + d * HH + 32 * w;
        if constexpr (PARTS == 4) {
            auto pick = [&](const f32x16& v, int e) -> float {
                return part == 0 ? v[e] : (part == 1 ? v[4 + e] : (part == 2 ? v[8 + e] : v[12 + e]));
            };
            bf16x4 sg[4], sc;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float ig = fast_sigmoid(pick(acc[0], e));
                const float fg = fast_sigmoid(pick(acc[1], e));
                const float gg = fast_tanh(pick(acc[2], e));
                const float og = fast_sigmoid(pick(acc[3], e));
                c[e] = __builtin_fmaf(fg, c[e], ig * gg);
                const float h = og * fast_tanh(c[e]);
                const int row = e + 8 * part;
                hnext[row * HB_LD] = (__bf16)h;
                if (YF32) (yrow + (size_t)row * DH)[y_off] = h;
                if (SAVE) { sg[0][e] = (__bf16)ig; sg[1][e] = (__bf16)fg; sg[2][e] = (__bf16)gg; sg[3][e] = (__bf16)og; sc[e] = (__bf16)c[e]; }
            }
            if (SAVE) {
                __bf16* gp = pblk + (size_t)t * pstep;
#pragma unroll
                for (int g = 0; g < 4; ++g) *reinterpret_cast<bf16x4*>((gp + g * 1024) + poff) = sg[g];
                CE* cp = cblk + (size_t)t * cstep;
                if constexpr (sizeof(CE) == 4) {
                    f32x4 v = {c[0], c[1], c[2], c[3]};
                    *reinterpret_cast<f32x4*>((cp + part * 256) + off4) = v;
                } else {
                    *reinterpret_cast<bf16x4*>(cp + poff) = sc;
                }
            }
        } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
#if LOB_ABL_H256 & 2
            const float ig = acc[0][r] * 0.25f, fg = acc[1][r] * 0.25f, gg = acc[2][r] * 0.25f, og = acc[3][r] * 0.25f;
            c[r] = __builtin_fmaf(fg, c[r], ig * gg);
            const float h = og * c[r];
#else
            const float ig = fast_sigmoid(acc[0][r]);
            const float fg = fast_sigmoid(acc[1][r]);
            const float gg = fast_tanh(acc[2][r]);
            const float og = fast_sigmoid(acc[3][r]);
            c[r] = __builtin_fmaf(fg, c[r], ig * gg);
            const float h = og * fast_tanh(c[r]);
#endif
            const int row = (r & 3) + 8 * (r >> 2);
            hnext[row * HB_LD] = (__bf16)h;
            if (YF32) (yrow + (size_t)row * DH)[y_off] = h;
            if (SAVE) { acc[0][r] = ig; acc[1][r] = fg; acc[2][r] = gg; acc[3][r] = og; }
        }
        if (SAVE) {
            __bf16* gp = pblk + (size_t)t * pstep;
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int pq = 0; pq < 2; ++pq) {
                    bf16x8 v;
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = (__bf16)acc[g][8 * pq + e];
                    if constexpr (LOB_NT_FST) __builtin_nontemporal_store(v, reinterpret_cast<bf16x8*>((gp + g * 1024 + pq * 512) + off8));
                    else                      *reinterpret_cast<bf16x8*>((gp + g * 1024 + pq * 512) + off8) = v;
                }
            CE* cp = cblk + (size_t)t * cstep;
            if constexpr (sizeof(CE) == 4) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    f32x4 v = {c[4 * q + 0], c[4 * q + 1], c[4 * q + 2], c[4 * q + 3]};
                    *reinterpret_cast<f32x4*>((cp + q * 256) + off4) = v;
                }
            } else {
#pragma unroll
                for (int pq = 0; pq < 2; ++pq) {
                    bf16x8 v;
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = (__bf16)c[8 * pq + e];
                    if constexpr (LOB_NT_FST) __builtin_nontemporal_store(v, reinterpret_cast<bf16x8*>((cp + pq * 512) + off8));
                    else                      *reinterpret_cast<bf16x8*>((cp + pq * 512) + off8) = v;
                }
            }
        }
        }       // (full tiles)
        __syncthreads();
        if (Y16 || DROP) emit_rows(cur ^ 1, t);        // h_t is complete in hs[cur ^ 1]
        cur ^= 1;
    }
}

// ------------------------------------------------------------------------------------------
// BPTT.  dh = dgates[32 x 1024] * W_hh[1024 x 256]: wave w owns the 32 hidden units [32w, 32w+32) of dh; its B
// fragments W_hh^T[unit][n = 16 ks + 8 hi .. +7] (ks < 64) are streamed from the pre-transposed bf16 copy, 8 per
// group, one group ahead.  The bf16 dgates tile (32 x 1024) is the MFMA A operand and the dP image.
// ------------------------------------------------------------------------------------------
// CE / DE: storage types of the saved cell states and of the incoming gradient dY (fp32, or bf16: ops.C_BF16 /
// ops.DY_BF16_CARRY -- the gradient carried from layer to layer is a bf16 stream like dP)
// NGL: how many of the 16 groups (of four k-steps) of a wave's W_hh^T fragments stay in LDS for the whole launch (a
// wave's private 4 KB per group; up to 2 fit next to the 66-KB dgates tile).  NAH: how many streamed groups are in
// flight ahead of the one being consumed (ring of NAH + 1 register buffers; the streamed-group count must be a multiple
// of it so that the cyclic order over the steps keeps one buffer per position).
// NGR (round 4): groups NGL .. NGL + NGR - 1 stay in REGISTERS for the whole launch (16 registers each): with the cell backward's
// contractions pinned the kernel needs 221 registers, and two groups fit into the rest -- 12 of 16 groups are streamed per step
// instead of 14.
template <typename CE, typename DE, int NGL = 0, int NAH = 1, int NGR = 0>
__global__ __launch_bounds__(512, 2) void lstm_rec_bwd_h256_bf16_kernel(
    const __bf16* __restrict__ G, const CE* __restrict__ Csave, const __bf16* __restrict__ WTb,
    const DE* __restrict__ dY, __bf16* __restrict__ dP, float* __restrict__ dbias, float* __restrict__ dbias2, int T, int Bp) {
    __shared__ __attribute__((aligned(16))) __bf16 dgs[32 * DGB_LD];
    __shared__ __attribute__((aligned(16))) __bf16 wl[NGL > 0 ? NW * NGL * 4 * 512 : 8];
    constexpr int NSG = 16 - NGL - NGR, NRB = NAH + 1;       // streamed groups per step, ring buffers
    static_assert(NSG % NRB == 0, "ring: the streamed-group count must be a multiple of the buffer count");
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int bt = blockIdx.x, d = blockIdx.y, D = gridDim.y, NBT = gridDim.x;
    const int l31 = lane & 31, hi = lane >> 5;

    const size_t gstep = (size_t)NBT * NW * 4096, cstep = (size_t)NBT * NW * 1024;
    const __bf16* gwave = G + (((size_t)d * T * NBT + bt) * NW + w) * 4096;
    const CE* cwave = Csave + (((size_t)d * T * NBT + bt) * NW + w) * 1024;
    const unsigned off8 = lane * 8, off4 = lane * 4;
    const int DH = D * HH, D4H = D * 4 * HH;
    const DE* dywave = dY + (size_t)(bt * 32) * DH + d * HH + 32 * w;
    const unsigned dy_off = (unsigned)(4 * hi * DH + l31);
    // B fragments in fragment order [D][wave][ks 64][lane 64][8]: element = W_hh[n = 16 ks + 8 (lane >> 5) + j][32w + (lane & 31)]
    const __bf16* wtwave = WTb + ((size_t)d * NW + w) * (64 * 64 * 8);
    unsigned wt_off = (unsigned)(lane * 8);

    const int t_first = d ? 0 : T - 1, dt = d ? 1 : -1;
    // cell states of this step / of the step before it in the direction's time: kept as loaded (bf16: 8 registers each
    // instead of 16) and widened where the cell update reads them -- the MFMA phase needs every register it can get
    struct CRaw16 { bf16x8 v[2]; };
    typedef typename std::conditional<sizeof(CE) == 2, CRaw16, f32x16>::type CS;
    CS ct, cp;
    // the two-halves schedule of the header for every storage type (with the cell backward's contractions pinned -- lob_dgate --
    // hipcc allocates the kernel in 221-232 registers; the unpinned expression needed 250 and spilled at fp32 cell states)
    constexpr bool HALVES = true;
    auto cval = [](const CS& cs, int r) -> float {
        if constexpr (sizeof(CE) == 2) return (float)cs.v[r >> 3][r & 7];
        else return cs[r];
    };
    f32x16 dhrec;
    float dy[16], dcarry[16];
    float dbsum[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int r = 0; r < 16; ++r) { dcarry[r] = 0.f; dhrec[r] = 0.f; }

    auto load_c = [&](int t, CS& dst) {
        if (t >= 0 && t < T) {
            const CE* cq = cwave + (size_t)t * cstep;
            if constexpr (sizeof(CE) == 4) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 v = *reinterpret_cast<const f32x4*>((cq + q * 256) + off4);
                    dst[4 * q] = v[0]; dst[4 * q + 1] = v[1]; dst[4 * q + 2] = v[2]; dst[4 * q + 3] = v[3];
                }
            } else {
#pragma unroll
                for (int pq = 0; pq < 2; ++pq) {
                    if constexpr (LOB_NT_CDY) dst.v[pq] = __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>((cq + pq * 512) + off8));
                    else                      dst.v[pq] = *reinterpret_cast<const bf16x8*>((cq + pq * 512) + off8);
                }
            }
        } else {
            if constexpr (sizeof(CE) == 4) {
#pragma unroll
                for (int r = 0; r < 16; ++r) dst[r] = 0.f;
            } else {
#pragma unroll
                for (int pq = 0; pq < 2; ++pq)
#pragma unroll
                    for (int e = 0; e < 8; ++e) dst.v[pq][e] = (__bf16)0.f;
            }
        }
    };
    Raw graw;
    // (HALVES) the same loads by halves of the lane's 16 tile elements (pq = r >> 3); t is in range (the caller clamps),
    // the cell states of an out-of-range step are never used (cp_ok below)
    auto load_half = [&](int t, int pq) {
        const __bf16* gq = gwave + (size_t)t * gstep;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            if constexpr (LOB_NT_G) graw.v[2 * g + pq] = __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>((gq + g * 1024 + pq * 512) + off8));
            else                    graw.v[2 * g + pq] = *reinterpret_cast<const bf16x8*>((gq + g * 1024 + pq * 512) + off8);
        }
        const int tc = t + dt;
        const CE* cq = cwave + (size_t)((tc >= 0 && tc < T) ? tc : t) * cstep;
        if constexpr (sizeof(CE) == 4) {
#pragma unroll
            for (int q = 2 * pq; q < 2 * pq + 2; ++q) {
                const f32x4 v = *reinterpret_cast<const f32x4*>((cq + q * 256) + off4);
                cp[4 * q] = v[0]; cp[4 * q + 1] = v[1]; cp[4 * q + 2] = v[2]; cp[4 * q + 3] = v[3];
            }
        } else {
            if constexpr (LOB_NT_CDY) cp.v[pq] = __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>((cq + pq * 512) + off8));
            else                      cp.v[pq] = *reinterpret_cast<const bf16x8*>((cq + pq * 512) + off8);
        }
        const DE* dp = dywave + (size_t)t * Bp * DH;
#pragma unroll
        for (int r = 8 * pq; r < 8 * pq + 8; ++r) {
            if constexpr (LOB_NT_CDY) dy[r] = (float)__builtin_nontemporal_load((dp + ((r & 3) + 8 * (r >> 2)) * DH) + dy_off);
            else                      dy[r] = (float)(dp + ((r & 3) + 8 * (r >> 2)) * DH)[dy_off];
        }
    };
    bf16x8 wb[NRB][4];            // group q = k-steps 4q .. 4q+3 (64 k-steps of 16 gate rows in 16 groups); streamed position
                                  // p = q - NGL lives in buffer p % NRB
    auto load_w = [&](int q, bf16x8 (&dst)[4]) {
#pragma unroll
        for (int j = 0; j < 4; ++j) dst[j] = *reinterpret_cast<const bf16x8*>((wtwave + (4 * q + j) * 512) + wt_off);
    };
    __bf16* wlw = wl + (size_t)w * (NGL * 4 * 512) + lane * 8;      // this wave's stationary fragments
    if constexpr (NGL > 0) {
#pragma unroll
        for (int f = 0; f < NGL * 4; ++f)
            *reinterpret_cast<bf16x8*>(wlw + f * 512) = *reinterpret_cast<const bf16x8*>((wtwave + f * 512) + wt_off);
    }
    load_c(t_first, ct);
    load_half(t_first, 0);
    load_half(t_first, 1);
    bf16x8 wr[NGR > 0 ? NGR : 1][4];                                 // register-resident groups
    if constexpr (NGR > 0) {
#pragma unroll
        for (int r = 0; r < NGR; ++r)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                wr[r][j] = *reinterpret_cast<const bf16x8*>((wtwave + (4 * (NGL + r) + j) * 512) + lane * 8);
    }
#pragma unroll
    for (int a = 0; a < NAH; ++a) load_w(NGL + NGR + a, wb[a % NRB]);

    for (int step = 0; step < T; ++step) {
        const int t = t_first + dt * step;
        __bf16* dgw = dgs + 32 * w + l31 + 4 * hi * DGB_LD;
        const int tcp = t + dt;
        const bool cp_ok = tcp >= 0 && tcp < T;            // (HALVES) c of the step before the first one is zero
        auto dgate_elem = [&](auto rc) {
            constexpr int r = decltype(rc)::value;
            constexpr int g8 = r >> 3, e8 = r & 7;     // element r of the 32x32 block = q pair r>>3, position r&7
            const float ig = (float)graw.v[0 + g8][e8], fg = (float)graw.v[2 + g8][e8];
            const float gg = (float)graw.v[4 + g8][e8], og = (float)graw.v[6 + g8][e8];
            const float dh = dy[r] + dhrec[r];
            const float tc = (LOB_ABL_H256 & 2) ? cval(ct, r) * 0.5f : fast_tanh(cval(ct, r));
            __bf16* p = dgw + ((r & 3) + 8 * (r >> 2)) * DGB_LD;
            const float cpv = (HALVES && !cp_ok) ? 0.f : cval(cp, r);
            float v0, v1, v2, v3;
            lob_dgate(ig, fg, gg, og, dh, tc, cpv, dcarry[r], v0, v1, v2, v3);
            p[0 * HH] = (__bf16)v0; p[1 * HH] = (__bf16)v1; p[2 * HH] = (__bf16)v2; p[3 * HH] = (__bf16)v3;
            dbsum[0] += v0; dbsum[1] += v1; dbsum[2] += v2; dbsum[3] += v3;
        };
        {
            const int tn = step + 1 < T ? t + dt : t;      // clamped: the last step re-reads its own lines, unused
            static_for<0, 8>(dgate_elem);
            if constexpr (sizeof(CE) == 2) ct.v[0] = cp.v[0];
            else {
#pragma unroll
                for (int r = 0; r < 8; ++r) ct[r] = cp[r];
            }
            load_half(tn, 0);
            static_for<8, 16>(dgate_elem);
            if constexpr (sizeof(CE) == 2) ct.v[1] = cp.v[1];
            else {
#pragma unroll
                for (int r = 8; r < 16; ++r) ct[r] = cp[r];
            }
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 16; ++r) dhrec[r] = 0.f;
        const __bf16* arow = dgs + l31 * DGB_LD + 8 * hi;
        asm volatile("" : "+v"(wt_off));           // see the forward kernel: keeps the W stream inside the time loop
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            if (q < NGL) {                              // stationary group: B fragments from this wave's LDS block
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    dhrec = mfma_bf16(*reinterpret_cast<const bf16x8*>(arow + 16 * (4 * q + j)),
                                      *reinterpret_cast<const bf16x8*>(wlw + (4 * q + j) * 512), dhrec);
                continue;
            }
            if (q < NGL + NGR) {                        // register-resident group
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    dhrec = mfma_bf16(*reinterpret_cast<const bf16x8*>(arow + 16 * (4 * q + j)), wr[q - NGL][j], dhrec);
                continue;
            }
            const int p = q - NGL - NGR, pn = (p + NAH) % NSG;     // NAH streamed groups ahead, cyclic over the steps
            if constexpr (!(LOB_ABL_H256 & 1)) load_w(NGL + NGR + pn, wb[pn % NRB]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < 4; ++j)
                dhrec = mfma_bf16(*reinterpret_cast<const bf16x8*>(arow + 16 * (4 * q + j)), wb[p % NRB][j], dhrec);
            __builtin_amdgcn_sched_barrier(0);
        }
        // the W-free stretch (dP stores, barrier, the first half of the next dgates phase) starts here.  No branch around
        // these loads: a block boundary at this point costs ~25 registers (the last step re-reads its own lines, unused)
        load_half(step + 1 < T ? t + dt : t, 1);
        // ---- the bf16 tile IS the dP image: 32 rows x 2 KB; 128 lanes x 16 B per row, 4 rows per pass
        __bf16* dpb = dP + ((size_t)t * Bp + bt * 32) * D4H + d * 4 * HH;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int row = 4 * i + (tid >> 7), c8 = (tid & 127) * 8;
            if constexpr (LOB_NT_DP) __builtin_nontemporal_store(*reinterpret_cast<const bf16x8*>(dgs + row * DGB_LD + c8),
                                                                 reinterpret_cast<bf16x8*>(dpb + (size_t)row * D4H + c8));
            else *reinterpret_cast<bf16x8*>(dpb + (size_t)row * D4H + c8) = *reinterpret_cast<const bf16x8*>(dgs + row * DGB_LD + c8);
        }
        __syncthreads();
    }
    if (dbias) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float v = dbsum[g] + __shfl_xor(dbsum[g], 32, 64);
            if (hi == 0) {
                const size_t bi = (size_t)d * 4 * HH + g * HH + 32 * w + l31;
                atomicAdd(dbias + bi, v);
                if (dbias2) atomicAdd(dbias2 + bi, v);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// BPTT on part tiles (round 4, LOB_VAR_REC_HALF): FOUR workgroups per 32-row tile, workgroup p owns tile rows 8 p .. 8 p + 7, i.e.
// the four elements r = 4 p + e of each lane (see the forward kernel).  It loads saved gates / cell states / dY of those elements
// only (8-byte fragments), computes their dgates into ITS tile (the other rows stay zero: their MFMA work is wasted), keeps
// only its rows of dh and stores only its rows of dP.  With four elements per lane there are registers to spare: the HBM
// loads of step s + 2 are issued right after the MFMA loop of step s (a W-free stretch, and a whole step ahead of their use)
// into one of two register sets.  Same arithmetic per row as the full-tile kernel: bit-identical.  For the few-tile regime
// (B <= 512 with two directions -- the reference's own training batch, 04_lstm_model.py:866): a quarter of the dgates work on
// every step's serial chain.
// ------------------------------------------------------------------------------------------
template <typename CE, typename DE>
__global__ __launch_bounds__(512, 2) void lstm_rec_bwd_h256_part_kernel(
    const __bf16* __restrict__ G, const CE* __restrict__ Csave, const __bf16* __restrict__ WTb,
    const DE* __restrict__ dY, __bf16* __restrict__ dP, float* __restrict__ dbias, float* __restrict__ dbias2, int T, int Bp) {
    typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
    constexpr int NGL = 2, NGR = 6, NSG = 16 - NGL - NGR;      // groups in LDS / in registers (96 of the ~100 this kernel leaves free) / streamed
    __shared__ __attribute__((aligned(16))) __bf16 dgs[32 * DGB_LD];
    __shared__ __attribute__((aligned(16))) __bf16 wl[NW * NGL * 4 * 512];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int bt = (int)blockIdx.x >> 2, part = (int)blockIdx.x & 3;
    const int d = blockIdx.y, D = gridDim.y, NBT = (int)gridDim.x >> 2;
    const int l31 = lane & 31, hi = lane >> 5;
    const unsigned poff = (unsigned)((part >> 1) * 512 + lane * 8 + (part & 1) * 4), off4 = lane * 4;

    const size_t gstep = (size_t)NBT * NW * 4096, cstep = (size_t)NBT * NW * 1024;
    const __bf16* gwave = G + (((size_t)d * T * NBT + bt) * NW + w) * 4096;
    const CE* cwave = Csave + (((size_t)d * T * NBT + bt) * NW + w) * 1024;
    const int DH = D * HH, D4H = D * 4 * HH;
    // this lane's rows: e + 8 part + 4 hi, column 32 w + l31 of direction d
    const DE* dylane = dY + (size_t)(bt * 32 + 8 * part + 4 * hi) * DH + d * HH + 32 * w + l31;
    const __bf16* wtwave = WTb + ((size_t)d * NW + w) * (64 * 64 * 8);
    unsigned wt_off = (unsigned)(lane * 8);
    const int t_first = d ? 0 : T - 1, dt = d ? 1 : -1;

    for (int i = tid; i < 32 * DGB_LD; i += 512) dgs[i] = (__bf16)0.f;      // rows of the other parts stay zero

    struct Set { bf16x4 g[4]; float cp[4]; float dy[4]; };
    Set sa, sb;
    auto load_c4 = [&](int t, float (&dst)[4]) {       // c of step t (t clamped by the caller)
        const CE* cq = cwave + (size_t)t * cstep;
        if constexpr (sizeof(CE) == 4) {
            const f32x4 v = *reinterpret_cast<const f32x4*>((cq + part * 256) + off4);
#pragma unroll
            for (int e = 0; e < 4; ++e) dst[e] = v[e];
        } else {
            const bf16x4 v = *reinterpret_cast<const bf16x4*>(cq + poff);
#pragma unroll
            for (int e = 0; e < 4; ++e) dst[e] = (float)v[e];
        }
    };
    auto load_set = [&](int t, Set& st) {              // t in range
        const __bf16* gq = gwave + (size_t)t * gstep;
#pragma unroll
        for (int g = 0; g < 4; ++g) st.g[g] = *reinterpret_cast<const bf16x4*>((gq + g * 1024) + poff);
        const int tc = t + dt;
        load_c4((tc >= 0 && tc < T) ? tc : t, st.cp);
        const DE* dp = dylane + (size_t)t * Bp * DH;
#pragma unroll
        for (int e = 0; e < 4; ++e) st.dy[e] = (float)dp[(size_t)e * DH];
    };
    float ct[4], dcarry[4] = {0.f, 0.f, 0.f, 0.f}, dbsum[4] = {0.f, 0.f, 0.f, 0.f};
    f32x16 dhrec;
#pragma unroll
    for (int r = 0; r < 16; ++r) dhrec[r] = 0.f;
    load_c4(t_first, ct);
    load_set(t_first, sa);
    if (T > 1) load_set(t_first + dt, sb);

    bf16x8 wb[2][4];
    auto load_w = [&](int q, bf16x8 (&dst)[4]) {
#pragma unroll
        for (int j = 0; j < 4; ++j) dst[j] = *reinterpret_cast<const bf16x8*>((wtwave + (4 * q + j) * 512) + wt_off);
    };
    __bf16* wlw = wl + (size_t)w * (NGL * 4 * 512) + lane * 8;
#pragma unroll
    for (int f = 0; f < NGL * 4; ++f)
        *reinterpret_cast<bf16x8*>(wlw + f * 512) = *reinterpret_cast<const bf16x8*>((wtwave + f * 512) + wt_off);
    bf16x8 wr[NGR][4];
#pragma unroll
    for (int r = 0; r < NGR; ++r)
#pragma unroll
        for (int j = 0; j < 4; ++j) wr[r][j] = *reinterpret_cast<const bf16x8*>((wtwave + (4 * (NGL + r) + j) * 512) + lane * 8);
    load_w(NGL + NGR, wb[0]);
    __syncthreads();

    auto one_step = [&](int step, Set& st) {
        const int t = t_first + dt * step;
        const int tcp = t + dt;
        const bool cp_ok = tcp >= 0 && tcp < T;
        __bf16* dgw = dgs + 32 * w + l31 + (8 * part + 4 * hi) * DGB_LD;
        float cnext[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float ig = (float)st.g[0][e], fg = (float)st.g[1][e], gg = (float)st.g[2][e], og = (float)st.g[3][e];
            const float rec = part == 0 ? dhrec[e] : (part == 1 ? dhrec[4 + e] : (part == 2 ? dhrec[8 + e] : dhrec[12 + e]));
            const float dh = st.dy[e] + rec;
            const float tc = fast_tanh(ct[e]);
            __bf16* p = dgw + e * DGB_LD;
            const float cpv = cp_ok ? st.cp[e] : 0.f;
            float v0, v1, v2, v3;
            lob_dgate(ig, fg, gg, og, dh, tc, cpv, dcarry[e], v0, v1, v2, v3);
            p[0 * HH] = (__bf16)v0; p[1 * HH] = (__bf16)v1; p[2 * HH] = (__bf16)v2; p[3 * HH] = (__bf16)v3;
            dbsum[0] += v0; dbsum[1] += v1; dbsum[2] += v2; dbsum[3] += v3;
            cnext[e] = st.cp[e];
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) ct[e] = cnext[e];
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 16; ++r) dhrec[r] = 0.f;
        const __bf16* arow = dgs + l31 * DGB_LD + 8 * hi;
        asm volatile("" : "+v"(wt_off));
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            if (q < NGL) {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    dhrec = mfma_bf16(*reinterpret_cast<const bf16x8*>(arow + 16 * (4 * q + j)),
                                      *reinterpret_cast<const bf16x8*>(wlw + (4 * q + j) * 512), dhrec);
                continue;
            }
            if (q < NGL + NGR) {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    dhrec = mfma_bf16(*reinterpret_cast<const bf16x8*>(arow + 16 * (4 * q + j)), wr[q - NGL][j], dhrec);
                continue;
            }
            const int p = q - NGL - NGR, pn = (p + 1) % NSG;
            load_w(NGL + NGR + pn, wb[pn & 1]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < 4; ++j)
                dhrec = mfma_bf16(*reinterpret_cast<const bf16x8*>(arow + 16 * (4 * q + j)), wb[p & 1][j], dhrec);
            __builtin_amdgcn_sched_barrier(0);
        }
        // W-free stretch: the loads of step s + 2 (into the set this step has just consumed), then this part's rows of dP
        load_set(step + 2 < T ? t + 2 * dt : t, st);
        __bf16* dpb = dP + ((size_t)t * Bp + bt * 32 + 8 * part) * D4H + d * 4 * HH;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = 4 * i + (tid >> 7), c8 = (tid & 127) * 8;
            *reinterpret_cast<bf16x8*>(dpb + (size_t)row * D4H + c8) =
                *reinterpret_cast<const bf16x8*>(dgs + (8 * part + row) * DGB_LD + c8);
        }
        __syncthreads();
    };
    for (int step = 0; step < T; step += 2) {
        one_step(step, sa);
        if (step + 1 < T) one_step(step + 1, sb);
    }
    if (dbias) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float v = dbsum[g] + __shfl_xor(dbsum[g], 32, 64);
            if (hi == 0) {
                const size_t bi = (size_t)d * 4 * HH + g * HH + 32 * w + l31;
                atomicAdd(dbias + bi, v);
                if (dbias2) atomicAdd(dbias2 + bi, v);
            }
        }
    }
}

}  // namespace

// Internal entry points used by lob_lstm_rec_fwd_bf16 / lob_lstm_rec_bwd_bf16 (lstm_rec_bf16.hip) at H = 256.
int lob_rec_fwd_h256_bf16(void* P, const void* Whh16, float* Y, void* Csave, int c_bf16, void* Y16, void* Yd, float drop_p,
                          uint64_t seed, int T, int Bp, int D, int save, hipStream_t s) {
    __bf16* y16 = reinterpret_cast<__bf16*>(Y16);
    __bf16* yd = reinterpret_cast<__bf16*>(Yd);
    if (c_bf16 && !save) return LOB_E_SHAPE;
    const bool ldsw = lob_variant(LOB_VAR_H256_LDSW) != 0;      // 0: every weight fragment streamed (the twin)
    // few tiles (B <= 512 with two directions: the reference's own training batch): four workgroups per tile
    // (LOB_VAR_REC_HALF: 1 = up to 32 tiles, 4 = up to 64 tiles, 0 = never)
    const int hv = lob_variant(LOB_VAR_REC_HALF), tiles = (Bp / 32) * D;
    const bool parts4 = ldsw && hv != 0 && tiles <= (hv == 4 ? 64 : 32);
    const dim3 grid(parts4 ? 4 * (Bp / 32) : Bp / 32, D), block(512);
#define LOB_FWD(SV, YF, Y6, DR, CE) do {                                                                                      \
        if (parts4) hipLaunchKernelGGL((lstm_rec_fwd_h256_bf16_kernel<SV, YF, Y6, DR, CE, 3, 4>), grid, block, 0, s,         \
            reinterpret_cast<__bf16*>(P), reinterpret_cast<const __bf16*>(Whh16), Y, reinterpret_cast<CE*>(Csave), y16, yd,  \
            drop_p, seed, T, Bp);                                                                                            \
        else if (ldsw) hipLaunchKernelGGL((lstm_rec_fwd_h256_bf16_kernel<SV, YF, Y6, DR, CE, 3>), grid, block, 0, s,         \
            reinterpret_cast<__bf16*>(P), reinterpret_cast<const __bf16*>(Whh16), Y, reinterpret_cast<CE*>(Csave), y16, yd,  \
            drop_p, seed, T, Bp);                                                                                            \
        else hipLaunchKernelGGL((lstm_rec_fwd_h256_bf16_kernel<SV, YF, Y6, DR, CE, 0>), grid, block, 0, s,                   \
            reinterpret_cast<__bf16*>(P), reinterpret_cast<const __bf16*>(Whh16), Y, reinterpret_cast<CE*>(Csave), y16, yd,  \
            drop_p, seed, T, Bp); } while (0)
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

int lob_rec_bwd_h256_bf16(const void* G, const void* Csave, int c_bf16, const void* WhhT16, const void* dY, int dy_bf16,
                          void* dP, float* dbias, float* dbias2, int T, int Bp, int D, hipStream_t s) {
    // LOB_VAR_H256_LDSW: 0 = every weight fragment streamed (the twin); 1 = two groups resident in LDS.  One streamed
    // group ahead in both: a deeper ring (one resident group, two streamed groups ahead: fits without spills once the
    // cell states are kept as loaded) measured 3.48 ms against 3.38 -- BPTT is not bound by the weights in flight
    const int mode = lob_variant(LOB_VAR_H256_LDSW);
    // few tiles: four workgroups per tile (see the forward's entry point)
    const int hv = lob_variant(LOB_VAR_REC_HALF), tiles = (Bp / 32) * D;
    if (mode != 0 && hv != 0 && tiles <= (hv == 4 ? 64 : 32)) {
#define LOB_BWD_P(CE, DE) hipLaunchKernelGGL((lstm_rec_bwd_h256_part_kernel<CE, DE>), dim3(4 * (Bp / 32), D), dim3(512), 0, s,   \
                       reinterpret_cast<const __bf16*>(G), reinterpret_cast<const CE*>(Csave),                               \
                       reinterpret_cast<const __bf16*>(WhhT16), reinterpret_cast<const DE*>(dY),                            \
                       reinterpret_cast<__bf16*>(dP), dbias, dbias2, T, Bp)
        if (c_bf16) { if (dy_bf16) LOB_BWD_P(__bf16, __bf16); else LOB_BWD_P(__bf16, float); }
        else        { if (dy_bf16) LOB_BWD_P(float, __bf16);  else LOB_BWD_P(float, float); }
#undef LOB_BWD_P
        LOB_CHECK_LAUNCH();
        return 0;
    }
#define LOB_BWD_R(CE, DE) hipLaunchKernelGGL((lstm_rec_bwd_h256_bf16_kernel<CE, DE, 2, 1, (sizeof(CE) == 2 ? 2 : 0)>), dim3(Bp / 32, D),  \
                       dim3(512), 0, s, reinterpret_cast<const __bf16*>(G), reinterpret_cast<const CE*>(Csave),              \
                       reinterpret_cast<const __bf16*>(WhhT16), reinterpret_cast<const DE*>(dY),                            \
                       reinterpret_cast<__bf16*>(dP), dbias, dbias2, T, Bp)
#define LOB_BWD_L(CE, DE, NGL, NAH) hipLaunchKernelGGL((lstm_rec_bwd_h256_bf16_kernel<CE, DE, NGL, NAH>), dim3(Bp / 32, D),  \
                       dim3(512), 0, s, reinterpret_cast<const __bf16*>(G), reinterpret_cast<const CE*>(Csave),              \
                       reinterpret_cast<const __bf16*>(WhhT16), reinterpret_cast<const DE*>(dY),                            \
                       reinterpret_cast<__bf16*>(dP), dbias, dbias2, T, Bp)
#define LOB_BWD(CE, DE, DEEP) do { if (mode == 0) LOB_BWD_L(CE, DE, 0, 1); else LOB_BWD_R(CE, DE); } while (0)
    if (c_bf16) { if (dy_bf16) LOB_BWD(__bf16, __bf16, true); else LOB_BWD(__bf16, float, true); }
    else        { if (dy_bf16) LOB_BWD(float, __bf16, false);  else LOB_BWD(float, float, false); }
#undef LOB_BWD
#undef LOB_BWD_L
#undef LOB_BWD_R
    LOB_CHECK_LAUNCH();
    return 0;
}
