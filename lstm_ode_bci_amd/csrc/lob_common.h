// Shared device helpers for the lob kernels (gfx950 / CDNA4 only, wave = 64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "lob.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// current value of a LOB_VAR_* kernel-variant switch (lob_api.hip)
int lob_variant(int which);

#define LOB_CHECK_LAUNCH()                                   \
    do {                                                     \
        hipError_t e__ = hipGetLastError();                  \
        if (e__ != hipSuccess) return (int)e__;              \
    } while (0)

// exact-f32 matrix core op: D(32x32) += A(32x2) * B(2x32); lane l feeds
// A[l&31][l>>5] and B[l>>5][l&31]  (cdna_hip_programming.md §3).
__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

// Row of accumulator register r (0..15) inside a 32x32 tile for this lane:
// row = (r&3) + 8*(r>>2) + 4*(lane>>5); column = lane & 31.
__device__ __forceinline__ int acc_row(int r, int lane) {
    return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
}

// Activations.  v_exp_f32 / v_rcp_f32 are ~1 ulp; the absolute error of the results
// (<= ~1.5e-7) is what the 1e-5 logit parity budget sees.  The reciprocal is the raw v_rcp_f32
// (__builtin_amdgcn_rcpf): __frcp_rn expands to the full IEEE division sequence (v_div_scale x2, v_rcp, four FMAs,
// v_div_fmas, v_div_fixup -- a third of all instructions of a recurrent step, and those kernels are issue-bound).
__device__ __forceinline__ float fast_sigmoid(float x) {
    return __builtin_amdgcn_rcpf(1.0f + __expf(-x));
}
__device__ __forceinline__ float fast_tanh(float x) {
    // 1 - 2/(1+e^{2x}): saturates cleanly to +-1, no inf/inf.
    return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __expf(2.0f * x));
}
// erf for the GELUs (04_lstm_model.py:176, 198, 201: nn.GELU() = the erf form).  The device library's erff is ~100
// instructions with two data-dependent branches, and the input projection applies it to 134 M elements per step (the
// LayerNorm kernels around it were VALU-bound on it).  Abramowitz & Stegun 7.1.26, branch-free: 1 - (a1 t + ... + a5 t^5)
// e^{-z^2}, t = 1 / (1 + p |z|): |error| <= 1.5e-7 absolute (+ ~1e-7 from v_exp / v_rcp), i.e. <= 4e-7 |x| on a GELU
// output -- fp32 rounding of values of order one; saturates exactly to +-1.  e2 (optional) returns e^{-z^2}, which
// the GELU derivative needs as well.
__device__ __forceinline__ float erf_as(float z, float* e2 = nullptr) {
    const float az = fabsf(z);
    const float t = __builtin_amdgcn_rcpf(__builtin_fmaf(0.3275911f, az, 1.0f));
    float p = __builtin_fmaf(t, 1.061405429f, -1.453152027f);
    p = __builtin_fmaf(p, t, 1.421413741f);
    p = __builtin_fmaf(p, t, -0.284496736f);
    p = __builtin_fmaf(p, t, 0.254829592f);
    const float e = __expf(-az * az);
    if (e2) *e2 = e;
    return copysignf(1.0f - p * t * e, z);
}
__device__ __forceinline__ float gelu_erf(float x) {
    return 0.5f * x * (1.0f + erf_as(x * 0.70710678118654752440f));
}
__device__ __forceinline__ float apply_act(float x, int act) {
    if (act == LOB_ACT_TANH) return fast_tanh(x);     // the cell update's tanh (|error| <= 1.2e-7): the library tanhf is ~50
                                                      // instructions and the score layer applies it to 134 M elements per step
    if (act == LOB_ACT_GELU) return gelu_erf(x);
    return x;
}

// Sum over the 64 lanes, result in every lane.  DPP (data-parallel primitives: the cross-lane operand rides on the
// VALU instruction) instead of six ds_bpermute round trips through the LDS pipeline -- the row-wise kernels do 2-4 of
// these per 512-B row and were bound by them.  Steps: quad swaps, half-row / row mirrors (every lane of a 16-lane
// row then holds the row's sum), row_bcast15 / row_bcast31 carry the running sum into the later rows, lane 63 ends
// with the total, which is read back as a scalar.
__device__ __forceinline__ float wave_sum(float v) {
#define LOB_DPP_ADD(CTRL, ROWMASK)                                                                          \
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROWMASK, 0xf, false))
    LOB_DPP_ADD(0xB1, 0xf);      // quad_perm [1,0,3,2]
    LOB_DPP_ADD(0x4E, 0xf);      // quad_perm [2,3,0,1]
    LOB_DPP_ADD(0x141, 0xf);     // row_half_mirror
    LOB_DPP_ADD(0x140, 0xf);     // row_mirror
    LOB_DPP_ADD(0x142, 0xa);     // row_bcast:15 -> rows 1 and 3
    LOB_DPP_ADD(0x143, 0xc);     // row_bcast:31 -> rows 2 and 3
#undef LOB_DPP_ADD
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// Counter-based RNG for dropout masks: the keep/drop decision of element `idx` depends
// only on (seed, idx), so the backward pass regenerates the mask instead of storing it.
// 32-bit mixing (xorshift-multiply rounds): the first version mixed in 64 bits (three 64-bit multiplies per
// element, ~100 VALU lane-operations) and made the dropout-carrying kernels VALU-bound -- 1.3 G hashes per training
// step at B = 4096; 32-bit integer multiplies are quarter rate on CDNA, so every multiply counts.
// The seed halves are wave-uniform (scalar registers); idx >> 32 is non-zero only beyond 4 G elements.
__device__ __forceinline__ uint32_t lob_hash32(uint64_t seed, uint64_t idx) {
    // two xorshift-multiply rounds (the "lowbias32" constants: avalanche bias 0.17 bits) over idx ^ seed; the high halves of
    // the seed (wave-uniform: scalar arithmetic) and of the index (< 2^24: a full-rate 24-bit multiply) are folded in
    // before the first round.  Round 3 dropped the third multiply round and the 32-bit multiply of idx >> 32: two
    // quarter-rate multiplies per hash instead of four (the dropout-carrying recurrent forward spent ~10 % of its time here).
    uint32_t x = (uint32_t)idx ^ (uint32_t)seed;
    x += __umul24((uint32_t)(idx >> 32), 0x9E3779u) + (uint32_t)(seed >> 32) * 0x85EBCA6Bu;
    x ^= x >> 16; x *= 0x7feb352du;
    x ^= x >> 15; x *= 0x846ca68bu;
    x ^= x >> 16;
    return x;
}
// One 32-bit hash decides TWO neighbouring elements (idx >> 1; the even element reads bits 0..15, the odd one bits
// 16..31), against a 16-bit threshold: p is quantised to 1/65536 (0.4 -> 0.399994).  Kernels that own both elements
// of a pair call lob_dropout_scale2 and pay for one hash; everything else calls lob_dropout_scale -- the same mask.
__device__ __forceinline__ uint32_t lob_dropout_thr16(float p) { return (uint32_t)(p * 65536.0f); }
__device__ __forceinline__ float lob_dropout_scale(uint64_t seed, uint64_t idx, float p) {
    // returns 0 (dropped) or 1/(1-p) (kept); p in [0,1)
    const uint32_t h = lob_hash32(seed, idx >> 1);
    const uint32_t bits = (idx & 1) ? (h >> 16) : (h & 0xffffu);
    return bits >= lob_dropout_thr16(p) ? 1.0f / (1.0f - p) : 0.0f;
}
// idx_even must be even: scales of elements idx_even and idx_even + 1
__device__ __forceinline__ void lob_dropout_scale2(uint64_t seed, uint64_t idx_even, float p, float& s0, float& s1) {
    const uint32_t h = lob_hash32(seed, idx_even >> 1), thr = lob_dropout_thr16(p);
    const float keep = 1.0f / (1.0f - p);
    s0 = (h & 0xffffu) >= thr ? keep : 0.0f;
    s1 = (h >> 16) >= thr ? keep : 0.0f;
}

// Power-of-two pre-scale of an fp32 operand tensor for the two-way fp16 split (gate_gemm_ws_split.hip,
// lstm_rec_f32_split.hip): the largest 2^k with amax * 2^k < 2^14, k clamped to [-60, 20].  hi = fp16(s) then has
// |hi| <= 2^14 and lo = (s - hi) 2^11 has |lo| <= 2^14 as well: both halves inside fp16's range for ANY finite operand.
__device__ __forceinline__ float lob_split_scale(float amax) {
    const int e = (int)((__builtin_bit_cast(unsigned, amax) >> 23) & 255u) - 127;      // floor(log2(amax)), normal numbers
    int k = 13 - e;
    k = k < -60 ? -60 : (k > 20 ? 20 : k);
    return __builtin_bit_cast(float, (unsigned)(k + 127) << 23);
}

// LDS read that the compiler cannot see (inline asm): used next to in-flight LDS-DMA (global_load_lds),
// where an ordinary LDS read of ANOTHER region makes hipcc drain the whole DMA queue (s_waitcnt vmcnt(0))
// because it cannot prove the regions disjoint.  The caller guarantees the word is not a DMA target.
__device__ __forceinline__ float lds_read_f32_opaque(const float* p) {
    const unsigned addr = (unsigned)(uintptr_t)(__attribute__((address_space(3))) const float*)p;
    float v;
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
    return v;
}
