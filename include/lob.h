/*
 * lob.h -- C ABI of the MI355X-native LSTM-ODE inner loop ("lob" = LSTM-ODE-BCI).
 *
 * The reference (khurrameycon/LSTM-ODE-BCI) is 100 % Python and has no FFI layer:
 * its boundary for this path is three Python classes.  This header
 * is the build-defined C ABI underneath those classes; every entry point cites the
 * reference lines whose arithmetic it replaces (paths relative to the reference
 * repository root).
 *
 * Conventions
 *   - every pointer is a caller-owned DEVICE buffer (hipMalloc / torch allocator);
 *   - `stream` is a hipStream_t passed as void*; nothing synchronises, nothing
 *     allocates, no global mutable state;
 *   - return value: 0 = ok; < 0 = argument error (LOB_E_*); > 0 = hipError_t of the
 *     launch.  Functions never throw and never exit.
 *   - internal activation layout is TIME-MAJOR: row index = t * Bp + b, where Bp is
 *     the batch padded to a multiple of 32 (pad rows hold finite garbage/zeros and
 *     are never read back by the caller).
 */
#ifndef LOB_H_
#define LOB_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LOB_VERSION 207

#define LOB_E_ARG   (-1)   /* null pointer / non-positive size                      */
#define LOB_E_SHAPE (-2)   /* shape not supported by this kernel (see each entry)   */
#define LOB_E_ALIGN (-3)   /* pointer / leading dimension alignment                 */

#define LOB_ACT_NONE 0
#define LOB_ACT_TANH 1     /* nn.Tanh  (04_lstm_model.py:119)                        */
#define LOB_ACT_GELU 2     /* nn.GELU, exact erf form (04_lstm_model.py:176,198,201) */
#define LOB_ACCUMULATE 0x100  /* OR into `act` of lob_gemm_nt_f32: C += result instead of C = */
#define LOB_OUT_BF16 0x400    /* OR into `act` of lob_gemm_nt_bf16 (bf16 x bf16 operands only) / lob_layernorm_act_bwd_f32:
                               * the result C / dx is stored as bf16 (row-major, same leading dimension in elements)   */
#define LOB_DY_BF16 0x800     /* OR into `act` of lob_layernorm_act_bwd_f32: dy is read as bf16                          */
#define LOB_X_BF16 0x1000     /* OR into `act` of lob_layernorm_act_f32 / lob_layernorm_act_bwd_f32: the input rows x are
                               * bf16 (widths 256 / 512; the backward then also needs LOB_DY_BF16 | LOB_OUT_BF16): in the mixed
                               * path the last LSTM layer hands its output over as bf16 only                             */
#define LOB_IP_COLWAVE 0x4000 /* OR into `act` of lob_input_proj_ln_bf16 (H == 128): the column-decomposed kernel (the one H == 256
                               * always runs) instead of the wave-per-tile one -- same results, test twin                         */
#define LOB_LN_IDENTITY 0x200 /* OR into `act` of lob_layernorm_act(_bwd)_f32: skip the normalisation and the
                               * affine (nn.Identity in place of nn.LayerNorm: the no-LayerNorm ablation,
                               * 09_sensitivity_analysis.py:190,209); gamma/beta/dgamma/dbeta may be NULL */

/* The ABI is VERSIONED, not append-only: entry points have gained parameters between versions (e.g. `range` of
 * lob_gate_gemm_x_f32 / lob_lstm_rec_fwd_f32 and `dattn` of lob_attn_pool_bwd_f32 in 202) and 205 removed two symbols.  A caller
 * must check, once after loading the library, that it was compiled against the same header -- lob_abi_ok() below, or the
 * build-id check of the Python loader -- and refuse to call anything otherwise.                                          */
int lob_version(void);
#define lob_abi_ok() (lob_version() == LOB_VERSION)

/* Identity of the build: a hash of every csrc source, this header and the compile flags, injected by
 * lstm_ode_bci_amd/build.py (-DLOB_BUILD_ID).  The Python loader compares it with the hash of the sources it
 * sees and refuses a stale library.                                                                      */
const char* lob_build_id(void);

/* ------------------------------------------------------------------------------------
 * TEST-ONLY: kernel variants.  Several hot kernels have a slower twin that computes the same values by a
 * simpler route (register-staged instead of LDS-DMA with hand-counted waits, the tiled GEMM instead of the
 * weight-stationary one, ...).  The parity tests run both at the bench's full shapes and compare them
 * (bit-exact where the arithmetic order is the same).  The table is process-global and read at every launch;
 * each entry starts from its default; only when LOB_DEBUG_VARIANTS=1 is set in the environment is an entry seeded
 * from the environment variable of the same name (A/B runs), so a stray LOB_* variable cannot re-route product
 * kernels.  Product code never calls the setter.  lob_debug_set_variant returns the previous value, LOB_E_ARG for an unknown index.
 * ---------------------------------------------------------------------------------- */
#define LOB_VAR_REC_BWD_DMA   0  /* 1: BPTT (H=128, mixed) streams G/c by LDS-DMA; 0: register-prefetch twin      */
#define LOB_VAR_NT_DMA        1  /* 1: bf16 TN (weight-gradient) GEMM on the LDS-DMA kernel; 0: register-staged    */
#define LOB_VAR_DMA_TILE      2  /* NT LDS-DMA GEMM output tile: 256 (256x256), 2 (256x128 x 2 WGs/CU), 128        */
#define LOB_VAR_DMA_KT        3  /* NT LDS-DMA GEMM k-slot width: 64 or 32                                        */
#define LOB_VAR_NT_ADEEP      4  /* 1: deeper A ring for the no-bias NT GEMMs (dX)                                */
#define LOB_VAR_GATE_WS       5  /* 1: weight-stationary gate GEMM (H=128, mixed); 0: tiled LDS-DMA NT GEMM         */
#define LOB_VAR_REC_BF16_ROWS 6  /* 16 (two workgroups per CU) or 32: row tile of the H=128 bf16 recurrent kernels */
#define LOB_VAR_F32_DMA       7  /* 1: fp32 NT GEMM on its LDS-DMA kernel; 0: register-staged                      */
#define LOB_VAR_REC_FWD_ROWS  8  /* 16 or 32: row tile of the fp32 recurrent forward (H=128)                      */
#define LOB_VAR_LN_LPR        9  /* 16: several rows per wave in the vectorised LayerNorm kernels; 64: one         */
#define LOB_VAR_NT_WGS       10  /* persistent workgroups per CU of the register-staged bf16 NT GEMM               */
#define LOB_VAR_NT_TK        11  /* its k-tile: 32 or 64                                                          */
#define LOB_VAR_NT_STAGGER   12  /* start stagger of the NT LDS-DMA GEMM (units of s_sleep(32)); 0 = off           */
#define LOB_VAR_FUSED_DW     13  /* read by the Python host: 1 = dW_ih and dW_hh from one pass (lob_lstm_dw_bf16)  */
#define LOB_VAR_F32_SPLIT    14  /* 1: fp32 gate GEMMs / recurrent kernels / backward GEMMs (H=128) carry each operand as two
                                  *    fp16 halves (22 bits) on the 16-bit matrix pipe; 0: exact-fp32 MFMA; 2: as 1, with
                                  *    lob_gemm_nt_f32_split / lob_gemm_tn_f32_split on their twins that split at every fragment
                                  *    read instead of once while staging (same arithmetic up to the summation order); 3: as 1,
                                  *    with those two on their general kernels where the pipelined ones would run (whole
                                  *    128 x 128 tiles; bit-identical twins)                                                    */
#define LOB_VAR_H256_LDSW    15  /* H=256 recurrent kernels: 0 = all W_hh fragments streamed; 1 = part of them resident in LDS      */
#define LOB_VAR_DX_KSPLIT    16  /* 1: dX = dP W_ih on the k-split weight-stationary kernel; 0: tiled LDS-DMA NT GEMM          */
#define LOB_VAR_REC_FEW      17  /* 1: mixed inference forward with fewer than 4 windows skips the padding registers' cell update */
#define LOB_VAR_GEMM_PP      18  /* bit mask, H=256 mixed step: 1 = dX, 2 = gate GEMM, 4 = weight gradients on the 8-wave
                                  *    ping-pong 256x256x64 kernels (csrc/gemm_pp.hip); 0 bits: the tiled / weight-stationary
                                  *    twins.  512 = dX on v_mfma_f32_16x16x32_bf16 (default: 1 | 4 | 512).  Slower but CORRECT
                                  *    schedules kept as twins: 8 = ring schedule, 128 / 256 = priority protocol.  The
                                  *    instantiations that compute garbage (16..64 = ablations, 1024 / 2048 = operand DMA alone,
                                  *    tools/pp_bench.py) exist only in builds with -DLOB_PP_DIAG; the product library ignores
                                  *    those bits                                                                                */
#define LOB_VAR_REC_HALF     19  /* recurrent forward kernels at H=128 (fp32 split; mixed: inference only) on few tiles: 1 = FOUR
                                  *    workgroups share each 16-row tile (one row j of the MFMA's D layout each) up to 64 tiles
                                  *    (B <= 512 with two directions), the mixed kernel two per tile up to 128 tiles; 2 / 4 force
                                  *    that split up to 128 tiles (A/B); 0: always full tiles (the twin, bit-identical)       */
#define LOB_VAR_COUNT        20
int lob_debug_set_variant(int which, int value);
int lob_debug_get_variant(int which);

/* ------------------------------------------------------------------------------------
 * Dense layers.  C[M,N] = act(A[M,K] * W[N,K]^T + bias[N]); fp32 in, exact-fp32 MFMA
 * (v_mfma_f32_32x32x2_f32), fp32 out.  A row-major with leading dimension lda, W is
 * the torch nn.Linear weight (N,K) row-major with leading dimension ldw, bias may be
 * NULL.  Replaces nn.Linear of input_proj (04_lstm_model.py:174), of the attention
 * score MLP (04:118) and of the classifier (04:197,200,203).
 * ---------------------------------------------------------------------------------- */
int lob_gemm_nt_f32(const float* A, int lda, const float* W, int ldw, const float* bias,
                    float* C, int ldc, int M, int N, int K, int act, void* stream);

/* The same product with every fp32 multiply carried as a two-way fp16 split on the 16-bit matrix pipe (22-bit products,
 * fp32 accumulate; as lob_gate_gemm_x_f32's default arithmetic): the backward GEMMs of the fp32 training step
 * (dX = dP W_ih, 04_lstm_model.py:490).  amax_a / amax_w: DEVICE floats, upper bounds of max|A| / max|W| (power-of-two
 * pre-scales are derived from them; lob_lstm_rec_bwd_f32_x produces the one of dP).  No bias / activation.  16-B aligned
 * operands, lda % 4 == ldw % 4 == 0, K % 32 == 0, K >= 128, N <= 2048; anything else LOB_E_SHAPE.                        */
int lob_gemm_nt_f32_split(const float* A, int lda, const float* W, int ldw, float* C, int ldc, int M, int N, int K,
                          const float* amax_a, const float* amax_w, float drop_p, uint64_t seed, void* stream);
/*   drop_p > 0 (207): C[row][col] *= the dropout mask of element row * ldc + col (lob_dropout_f32's, same p / seed): the
 *   backward of a dropout that was fused into the producer of the layer below's output (lob_lstm_rec_fwd_f32_drop);
 *   LOB_E_SHAPE under LOB_VAR_F32_SPLIT = 2 (the fragment-read twin has no such epilogue).                               */

/* C[M,N] (+)= A[Kc,M]^T * B[Kc,N]  (contraction over the leading/row index; weight
 * gradients dW = dY^T X).  Split over Kc across workgroups, fp32 atomics into C, so C
 * must be zeroed (or hold the value to accumulate into) before the call.             */
int lob_gemm_tn_f32(const float* A, int lda, const float* B, int ldb, float* C, int ldc,
                    int M, int N, int Kc, void* stream);
/* lob_gemm_tn_f32 with two-way fp16 split products (see lob_gemm_nt_f32_split): dW = dP^T X of the fp32 training step.
 * 16-B aligned operands, lda % 4 == ldb % 4 == 0, M % 4 == N % 4 == 0; anything else LOB_E_SHAPE.                      */
int lob_gemm_tn_f32_split(const float* A, int lda, const float* B, int ldb, float* C, int ldc, int M, int N, int Kc,
                          const float* amax_a, const float* amax_b, void* stream);

/* ------------------------------------------------------------------------------------
 * Input-side gate GEMM of one LSTM layer, both directions at once:
 *   P = X[T*Bp, K] * Wih[D*4H, K]^T + bias[D*4H]       (bias = b_ih + b_hh)
 * nn.LSTM call site 04_lstm_model.py:181-188, 211 (W_ih x_t + b_ih + b_hh of the cell).
 * frag = 1: P is written in the accumulator-fragment order the persistent recurrent
 *   kernel consumes, [D][T][Bp/32][H/32][4 gates][4][64 lanes][4] floats (requires
 *   H % 32 == 0, Bp % 32 == 0);  frag = 0: P is row-major (T*Bp, D*4H).
 * ---------------------------------------------------------------------------------- */
int lob_gate_gemm_x_f32(const float* X, int ldx, const float* Wih, const float* bias,
                        float* P, int T, int Bp, int H, int D, int K, int frag,
                        const float* range, void* stream);
/*   frag & 2: never take the fp16-split kernel (exact-fp32 MFMA; for activations without a known bound).
 *   range (device, may be NULL; read by the H = 128 fp16-split kernel only): range[d] = max |W_ih| of direction d,
 *   range[D] = an upper bound on |X|.  The split kernel carries every fp32 operand as two fp16 halves of s = x 2^k; with
 *   `range` it chooses k per operand tensor so that max|s| lies in [2^13, 2^14) -- any finite weights / activations stay
 *   inside fp16's range (a fixed k = 8 / 6 overflowed to inf/NaN beyond |w| >= 256 or |x| >= 1024).  NULL: k = 8 / 6. */

/* ------------------------------------------------------------------------------------
 * Recurrent part of one LSTM layer, all T steps inside one persistent kernel, both
 * directions (grid = batch tiles x D):
 *   z = P_t + W_hh h_{t-1};  i,f,g,o = split(z);  c = s(f) c + s(i) tanh(g);
 *   h = s(o) tanh(c);   h_-1 = c_-1 = 0;  the reverse direction walks t = T-1..0.
 * (torch nn.LSTM semantics; call site 04_lstm_model.py:211.)
 *   P     gate pre-activations from lob_gate_gemm_x_f32 (frag=1 iff H == 128)
 *   Whh   [D][4H][H]   (weight_hh_l{k}, weight_hh_l{k}_reverse stacked)
 *   Y     [T][Bp][D*H] layer output, direction d in columns [d*H, (d+1)*H)
 *   save  0: inference.  1: training -- P is overwritten in place with the ACTIVATED
 *         gates (same layout) and Csave receives c_t ([D][T][Bp][H], fragment order
 *         when frag) for the backward kernel.
 * ---------------------------------------------------------------------------------- */
int lob_lstm_rec_fwd_f32(float* P, const float* Whh, float* Y, float* Csave,
                         int T, int Bp, int H, int D, int save, const float* range, void* stream);
/*   range (device, may be NULL; H = 128 fp16-split kernel only): range[d] = max |W_hh| of direction d -- the weight
 *   pre-scale is chosen from it (see lob_gate_gemm_x_f32); h needs none (|h| < 1).                                   */
/* 207: the saving forward (save = 1) with nn.LSTM's inter-layer dropout (04_lstm_model.py:186) fused into the producer:
 * Yd [T*Bp][D*H] = dropout(Y), the mask lob_dropout_f32 applies to Y with the same drop_p / seed.  H == 128, Bp % 32 == 0,
 * LOB_VAR_F32_SPLIT != 0, 0 < drop_p < 1; anything else LOB_E_SHAPE / LOB_E_ARG (run lob_lstm_rec_fwd_f32 + lob_dropout_f32). */
int lob_lstm_rec_fwd_f32_drop(float* P, const float* Whh, float* Y, float* Yd, float drop_p, uint64_t seed, float* Csave,
                              int T, int Bp, int H, int D, const float* range, void* stream);

/* 1 if the recurrent kernels for hidden size H use the fragment layout (H = 32, 64, 128, 256:
 * MFMA kernels; W_hh register-resident at 128, streamed from L2 otherwise), 0 for the generic
 * row-major VALU path.  Pass this as `frag` to lob_gate_gemm_x_f32.                        */
int lob_lstm_uses_fragment_layout(int H);

/* Backward of the above (BPTT through time), one persistent kernel, both directions:
 *   G      activated gates saved by the forward (save = 1), same layout as P; READ-ONLY
 *   Csave  cell states saved by the forward; READ-ONLY (backward is re-entrant, 07:254)
 *   dY     [T][Bp][D*H] gradient w.r.t. the layer output
 *   dP     [T*Bp][D*4H] ROW-MAJOR gradient w.r.t. the gate pre-activations (output), fp32 or bf16
 * dW_ih = dP^T X, dW_hh[d] = dP[:,d]^T H_prev, db = colsum(dP) (lob_gemm_tn_f32,
 * lob_colsum_f32) and dX = dP W_ih (lob_gemm_nt_f32 with the transposed weight) follow.
 * (torch autograd of nn.LSTM; training step 04_lstm_model.py:482-512.)                */
int lob_lstm_rec_bwd_f32(const float* G, const float* Csave, const float* Whh,
                         const float* dY, void* dP, int dp_bf16, float* dbias,
                         int T, int Bp, int H, int D, void* stream);
/* The same on the fp32 path's fp16-split arithmetic (two-way operand splits on the 16-bit matrix pipe, 22-bit products,
 * fp32 accumulate; the dgates' pre-scale is re-derived from each tile every step): H == 128, Bp % 32 == 0 and
 * LOB_VAR_F32_SPLIT != 0, else LOB_E_SHAPE (keep lob_lstm_rec_bwd_f32).  amax_out (may be NULL): a ZEROED device float that
 * receives max|dP| of the launch -- the pre-scale input of lob_gemm_nt_f32_split / lob_gemm_tn_f32_split on dP.
 * range (may be NULL): D device floats, max|W_hh| per direction (LOB_PREP_ABSMAX).                                      */
int lob_lstm_rec_bwd_f32_x(const float* G, const float* Csave, const float* Whh, const float* dY, void* dP,
                           int dp_bf16, float* dbias, float* amax_out, const float* range,
                           int T, int Bp, int H, int D, void* stream);
/*   dp_bf16  1: dP is written as bf16 (mixed-precision mode: its consumers are bf16 MFMA GEMMs)
 *   dbias    [D*4H] bias gradient, accumulated in-kernel with fp32 atomics (initialise it); may be
 *            NULL; must be NULL on the generic (H != 128) path -- use lob_colsum_f32/_bf16 there. */

/* out[n] += sum_m A[m][n]  (bias gradients; fp32 atomics, `out` must be initialised). */
int lob_colsum_f32(const float* A, int lda, int M, int N, float* out, void* stream);

int lob_colsum_bf16(const void* A, int lda, int M, int N, float* out, void* stream);

/* ------------------------------------------------------------------------------------
 * Mixed-precision variants (BASELINE.json configs[2] "bf16 gate-GEMMs + fp32 recurrence"; the
 * reference's training loop runs under autocast, 04_lstm_model.py:487): operands are rounded to
 * bf16 on their way into LDS, products are exact, accumulation and outputs are fp32
 * (v_mfma_f32_32x32x16_bf16).  `a_bf16` / `b_bf16` = 1 when that operand is stored as bf16 in HBM
 * (dP from lob_lstm_rec_bwd_f32), 0 for fp32 storage.  Shapes as the _f32 entry points; K % 8 == 0
 * and 16-byte aligned bases are required (LOB_E_ALIGN otherwise: use the fp32 entry point).
 * ---------------------------------------------------------------------------------- */
int lob_gemm_nt_bf16(const void* A, int a_bf16, int lda, const void* W, int w_bf16, int ldw,
                     const float* bias, float* C, int ldc, int M, int N, int K, int act, float drop_p,
                     uint64_t seed, void* stream);
/*   a_bf16 = w_bf16 = 1 (both operands bf16 in HBM, no bias / activation, K % 32 == 0, K >= 128):
 *   LDS-DMA kernel -- operand tiles go HBM -> LDS by global_load_lds, three k-tiles in flight.    */
/*   drop_p > 0: the result is multiplied by the nn.Dropout mask of element (row*ldc + col) (same
 *   counter-based hash as lob_dropout_f32): fuses the backward of nn.LSTM's inter-layer dropout
 *   (04_lstm_model.py:186) into dX = dP W_ih.                                                */
int lob_gate_gemm_x_bf16(const void* X, int x_bf16, int ldx, const void* Wih, int w_bf16, const float* bias,
                         void* P, int p_bf16, int T, int Bp, int H, int D, int K, void* stream);
/*   p_bf16 = 1: the fragment-order pre-activations are STORED as bf16 (same element order, half the
 *   bytes); the bf16 recurrent kernels take the matching pg_bf16 flag and also keep the activated gates
 *   they save for BPTT in bf16.  Accumulation, cell state and everything else stay fp32.          */
int lob_gemm_tn_bf16(const void* A, int a_bf16, int lda, const void* B, int b_bf16, int ldb,
                     float* C, int ldc, int M, int N, int Kc, void* stream);
/*   Both weight gradients of one LSTM layer (the dW_ih / dW_hh that loss.backward() accumulates for
 *   lstm.weight_ih_l* / weight_hh_l*, 04_lstm_model.py:490) from ONE pass over the bf16 gate gradients:
 *   dWih[D*4H][nx] += dP^T X and dWhh[D][4H][H] += dP[:, d]^T h_prev_d, where h_prev is Y one step (Bp rows)
 *   earlier (d = 0) or later (d = 1).  dP [T*Bp][ldp], X [T*Bp][ldx], Y [T*Bp][ldy] bf16 time-major; outputs
 *   fp32, zeroed by the caller.  H == 128, nx in {128, 256}, Bp % 32 == 0, T >= 2; or H == 256 (the size the reference
 *   trains at 61 channels, 04_lstm_model.py:877), D == 2, nx in {256, 512}, Bp % 64 == 0, (T * Bp) % 128 == 0; else
 *   LOB_E_SHAPE.                                                                                        */
int lob_lstm_dw_bf16(const void* dP, int ldp, const void* X, int ldx, int nx, const void* Y, int ldy,
                     float* dWih, float* dWhh, int T, int Bp, int H, int D, void* stream);

/* Recurrent kernels with the hidden-state gate GEMM on bf16 MFMA (H == 128 and H == 256; cell state and
 * everything carried through time stay fp32).  Same arguments as lob_lstm_rec_fwd_f32 / lob_lstm_rec_bwd_f32;
 * dP is always bf16 here.  H == 128: W_hh (fp32, [D][4H][H]) is converted and kept in registers.  H == 256: the
 * weights are streamed from L2 every step and must ALSO be handed over as bf16 in MFMA FRAGMENT ORDER (every
 * wave-load 1 KB contiguous), and P / saved gates must be bf16 (pg_bf16 = 1):
 *   Whh16  [D][w 8][ks 16][gate 4][lane 64][8]: W_hh[d][gate*H + 32w + (lane&31)][16ks + 8(lane>>5) + j]   (forward)
 *   WhhT16 [D][w 8][ks 64][lane 64][8]:         W_hh[d][16ks + 8(lane>>5) + j][32w + (lane&31)]            (BPTT)
 * Whh16 / WhhT16 are ignored (may be NULL) at H == 128.                                                  */
int lob_lstm_rec_fwd_bf16(void* P, int pg_bf16, const float* Whh, const void* Whh16, float* Y, void* Csave, int c_bf16,
                          void* Y16, void* Yd, float drop_p, uint64_t seed,
                          int T, int Bp, int H, int D, int save, int nvalid, void* stream);
/*   nvalid: how many of the Bp rows carry windows (rows >= nvalid are padding); 0 = unknown / all.  Only a hint: with
 *   save == 0, bf16 P, H == 128 and nvalid < 4 the padding rows' cell update is skipped and their outputs are zeros.  */
/*   c_bf16 = 1 (H == 128, 16-row kernels, bf16 saved gates): the cell states saved for BPTT are stored as bf16, in the
 *   element order of one gate of the saved gates ([q pair][lane][8]); BPTT takes the matching flag.  The state carried
 *   through time is fp32 in both kernels; only the saved copy is rounded.                                          */
/*   Outputs, any non-empty subset with Y or Y16 present: Y (fp32 [T*Bp][D*H]), Y16 = bf16(Y) and
 *   Yd = bf16(dropout(Y; drop_p, seed)) (element index = position in Y) -- nn.LSTM's inter-layer dropout
 *   (04_lstm_model.py:186) fused into the producer.  In mixed mode the bf16 GEMMs of the next layer read
 *   Yd (or Y16 without dropout) and dW_hh reads Y16, so layers below the last never write fp32 Y.  */
int lob_lstm_rec_bwd_bf16(const void* G, int pg_bf16, const void* Csave, int c_bf16, const float* Whh, const void* WhhT16,
                          const void* dY, int dy_bf16,
                          void* dP, float* dbias, float* dbias2, int T, int Bp, int H, int D, void* stream);
/*   dbias2 (may be NULL; needs dbias): a second [D*4H] destination that receives the same atomic adds -- b_ih and b_hh
 *   have the same gradient (04_lstm_model.py:181-188 keeps both), and an optimizer that owns one flat gradient buffer
 *   hands the two parameters' slices over directly instead of adding a temporary into each afterwards.             */
/*   dy_bf16 = 1 (H == 128, 16-row kernels): dY is stored as bf16 -- in the mixed path the gradient carried from layer
 *   to layer (dX of the layer above, written by lob_gemm_nt_bf16 with LOB_OUT_BF16, or the LayerNorm backward's dx)
 *   is a bf16 stream like dP; it is widened to fp32 on load and everything carried through time stays fp32.      */

/* Element-wise activation and its backward (dx = dy * act'(pre)); classifier GELUs
 * (04_lstm_model.py:198, 201) in training mode.                                       */
int lob_act_f32(const float* in, float* out, int64_t n, int act, void* stream);
int lob_act_bwd_f32(const float* dy, const float* pre, float* dx, int64_t n, int act, void* stream);

/* Backward of lob_layernorm_act_f32 (same act / dropout seed / remap arguments).  x is the
 * LayerNorm INPUT (rows in input order), dy the gradient of the output (rows in output
 * order).  dgamma / dbeta are ACCUMULATED with fp32 atomics (initialise them).         */
int lob_layernorm_act_bwd_f32(const float* x, const float* gamma, const float* beta, const float* dy,
                              float* dx, float* dgamma, float* dbeta, int rows, int width, float eps,
                              int act, int remap_T, int remap_B, int remap_Bp, float drop_p,
                              uint64_t seed, const float* pool_attn, const float* pool_dctx,
                              int pool_T, int pool_B, int pool_Bp, float* dx_colsum, void* stream);
/*   pool_attn != NULL (time-major rows, widths 128/256/512): dy[t*Bp+b][:] += pool_attn[b][t] * pool_dctx[b][:]
 *   before the LayerNorm backward -- the context path of the attention pooling, fused here instead of
 *   being written to HBM by lob_attn_pool_bwd_f32.
 *   dx_colsum != NULL (same widths): dx_colsum[c] += sum_rows dx[row][c] -- the bias gradient of the Linear that
 *   feeds this LayerNorm (input_proj.0.bias, 04:174), instead of a separate pass over dx.             */

/* Backward of lob_attn_pool_fwd_f32.  dV [T*Bp][W] and dPreU [T*Bp][W2] are WRITTEN for rows
 * b < B (pad rows untouched); dw2 [W2] is accumulated (atomics).  dPreU is the gradient w.r.t.
 * the pre-tanh hidden W1 v + b1; the caller adds dPreU W1 into dV with lob_gemm_nt_f32.   */
int lob_attn_pool_bwd_f32(const void* V, int v_bf16, const float* U, const float* attn, const float* dctx,
                          const float* w2, float* dV, void* dPreU, int du_bf16, float* dw2,
                          float* du_colsum, int T, int B, int Bp, int W, int W2, const float* dattn, void* stream);
/*   dattn [B][T] (may be NULL): gradient w.r.t. the attention WEIGHTS themselves, added to dctx . V[t] before the softmax
 *   backward -- a stand-alone Attention module (04_lstm_model.py:112-128) returns (context, weights) and a caller may
 *   differentiate through either; inside EnhancedLSTMModel the weights carry no gradient (NULL).
 *   v_bf16 / du_bf16: V read / dPreU written as bf16 (mixed mode).  dV == NULL: the direct term
 *   a[t] * dctx is not materialised; pass pool_attn / pool_dctx to lob_layernorm_act_bwd_f32 instead.
 *   U == NULL (mean pooling): only dV = a[t] * dctx is written; w2, dPreU, dw2 are ignored.
 *   du_colsum != NULL (bf16 V and dPreU, dV == NULL, W = 256, W2 = 128 only): du_colsum[j] += sum_t dPreU[t][j]
 *   -- the gradient of the score MLP's first bias (04:118) -- instead of a separate pass over dPreU.   */

/* ------------------------------------------------------------------------------------
 * Row-wise LayerNorm (biased variance, eps) with affine, optional GELU, optional
 * dropout, optional (b,t)->(t,b) row remap.  out[row'] = drop(act(LN(in[row]))).
 *   remap_T > 0: in rows are ordered (b, t) with t < remap_T; out row = t*Bp + b.
 *   remap_T = 0: out row = in row.
 *   out_bf16 = 1 (widths 128/256/512 only): the result is stored as bf16 (mixed mode: it only feeds
 *   bf16 MFMA GEMMs).
 * nn.LayerNorm + nn.GELU + nn.Dropout of input_proj (04_lstm_model.py:175-177) and
 * the post-LSTM nn.LayerNorm (04:192, 212).
 * ---------------------------------------------------------------------------------- */
int lob_layernorm_act_f32(const float* in, const float* gamma, const float* beta,
                          void* out, int out_bf16, int rows, int width, float eps, int act,
                          int remap_T, int remap_B, int remap_Bp,
                          float drop_p, uint64_t seed, void* stream);

/* Fused input projection of the mixed path, H == 128 or 256 (input_proj = Linear(C -> H) -> LayerNorm -> GELU -> Dropout,
 * 04_lstm_model.py:173-178), C <= 64: reads the fp32 windows x[B*T][C] (rows (b,t), 16-byte aligned base) once and writes
 * the bf16 activations out[T*Bp][H] time-major (row t*Bp + b; pad rows b >= B untouched) -- the same numbers, bit for
 * bit, as lob_pad_cast_bf16 + lob_gemm_nt_bf16(bias) + lob_layernorm_act_f32(remap, dropout) in a row.  W: fp32 [H][ldw]
 * (nn.Linear weight), rounded to bf16 like the windows.  Training passes pre (fp32 [B*T][H], the LayerNorm's input, rows
 * (b,t)) AND xb (bf16 [B*T][Cp], the padded windows: operand of the weight-gradient GEMM; Cp = C rounded up to 8); inference
 * passes NULL for both.  act / drop_p / seed as lob_layernorm_act_f32 (LOB_LN_IDENTITY accepted).                     */
int lob_input_proj_ln_bf16(const float* x, int C, const float* W, int ldw, const float* bias,
                           const float* gamma, const float* beta, float* pre, void* xb, int Cp, void* out,
                           int B, int T, int Bp, int H, float eps, int act, float drop_p, uint64_t seed, void* stream);
/* The same head for the FP32 path, H == 128 (207): exact fp32 products (the contraction is C <= 64 long), fp32 activations
 * out [T*Bp][128] (time-major; rows of padding windows are left untouched: zero them once), pre (fp32 [B*T][128], rows (b,t)) for
 * the backward or NULL.  Same results as lob_gemm_nt_f32 + lob_layernorm_act_f32(remap) (to fp32 summation order of the 61-term
 * dot products at most).                                                                                              */
int lob_input_proj_ln_f32(const float* x, int C, const float* W, int ldw, const float* bias, const float* gamma,
                         const float* beta, float* pre, float* out, int B, int T, int Bp, int H, float eps, int act,
                         float drop_p, uint64_t seed, void* stream);

/* nn.Dropout (04_lstm_model.py:177,186,199,202): out[i] = in[i] * keep_i / (1-p), where
 * keep_i is a counter-based hash of (seed, i): the backward pass applies the same call to the
 * gradient with the same seed instead of storing a mask.  in == out is allowed.           */
int lob_dropout_f32(const float* in, float* out, int64_t n, float p, uint64_t seed, void* stream);

/* Fused head of the mixed backward, H == 128 (round 3): backward of input_proj's LayerNorm + GELU + dropout
 * (04_lstm_model.py:175-177) and the Linear's weight gradient (04:174) in one pass; dpre is never written:
 *   pre [B*T][128] fp32 (the LayerNorm's input, rows (b,t)); dA16 [T*Bp][128] bf16 (gradient of the block's output, time-major);
 *   xb16 [B*T][Cp] bf16 (the padded windows saved by lob_input_proj_ln_bf16 / lob_pad_cast_bf16), Cp <= 64, Cp % 8 == 0;
 *   dW [128][lddw] fp32 += dpre^T xb (bf16 operands, fp32 accumulate, atomics); dgamma / dbeta [128] accumulated;
 *   dbias [128] (may be NULL) += column sums of the fp32 dpre.  act / drop_p / seed: those of the forward.
 * Equals lob_layernorm_act_bwd_f32 (bf16 dx) + lob_gemm_tn_bf16 up to fp32 summation order.                         */
int lob_input_proj_bwd_bf16(const float* pre, const float* gamma, const float* beta, const void* dA16, const void* xb16,
                            int Cp, float* dW, int lddw, float* dgamma, float* dbeta, float* dbias,
                            int B, int T, int Bp, int H, float eps, int act, float drop_p, uint64_t seed, void* stream);

/* Fused tail of the mixed backward, H == 128 or 256, bidirectional (round 3): dV = dU W1 (+ attn[b][t] dctx[b], the context path
 * of the pooling) and the backward of the post-LSTM LayerNorm in one pass:
 *   X16 [T*Bp][2H] bf16: the LayerNorm's input (the last LSTM layer's output); dU16 [T*Bp][H] bf16: gradient w.r.t. the
 *   score layer's pre-activations (lob_attn_pool_bwd_f32); W1T_16 bf16 [2H][H] = attention.attention.0.weight^T;
 *   dX16 [T*Bp][2H] bf16 out; dgamma / dbeta accumulated (fp32 atomics); attn [B][T], dctx [B][2H] fp32.
 * dX16 is bit-identical to lob_gemm_nt_bf16 (bf16 dV) + lob_layernorm_act_bwd_f32(pool_attn, pool_dctx).             */
int lob_attn_ln_bwd_bf16(const void* X16, const float* gamma, const float* beta, const void* dU16, const void* W1T_16,
                         void* dX16, float* dgamma, float* dbeta, const float* attn, const float* dctx,
                         int T, int B, int Bp, int H, int D, float eps, void* stream);

/* Fused tail of the mixed forward, H == 128 or 256, bidirectional (round 3): post-LSTM LayerNorm (04_lstm_model.py:192) and the
 * attention's score layer (04:123-125) in one pass over the last LSTM layer's bf16 output Y16 [T*Bp][2H] (time-major):
 *   v = LN(Y16) (bf16, written to V [T*Bp][2H]);  u = tanh(W1 v + b1);  S[b][t] = w2 . u + b2   (rows b < B)
 * W1_16: bf16 [H][2H] (attention.attention.0.weight); U: fp32 [T*Bp][H] for the backward, or NULL (inference).
 * v, u and the scores are bit-identical to lob_layernorm_act_f32 + lob_gemm_nt_bf16(tanh) + the score sums of
 * lob_attn_pool_fwd_f32.  The scores go to lob_attn_pool_fwd_f32 as U with W2 == 0.                                */
int lob_attn_scores_bf16(const void* Y16, const float* gamma, const float* beta, const void* W1_16, const float* b1,
                         const float* w2, const float* b2, void* V, float* U, float* S, int T, int B, int Bp, int H, int D,
                         float eps, void* stream);

/* The same fusion for the FP32 path, H == 128, bidirectional (207): Y, V fp32 [T*Bp][256], W1 fp32 [128][256], U fp32 [T*Bp][128]
 * or NULL.  v is bit-identical to lob_layernorm_act_f32; the score layer's products run as two-way fp16 splits on the 16-bit
 * matrix pipe (22-bit products, fp32 accumulate, pre-scales derived on the device from gamma / beta / W1: as
 * lob_gate_gemm_x_f32's default arithmetic), so u and S equal lob_gemm_nt_f32(tanh) + the score sums of lob_attn_pool_fwd_f32 to
 * ~1e-6.  The scores go to lob_attn_pool_fwd_f32 as U with W2 == 0 (v_bf16 = 0).                                          */
int lob_attn_scores_f32(const float* Y, const float* gamma, const float* beta, const float* W1, const float* b1,
                        const float* w2, const float* b2, float* V, float* U, float* S, int T, int B, int Bp, int H, int D,
                        float eps, void* stream);

/* Additive attention pooling over time (Attention.forward, 04_lstm_model.py:123-128):
 *   s[t,b] = U[t*Bp+b,:] . w2 + b2   (U = tanh(W1 v + b1), computed by lob_gemm_nt_f32)
 *   a[b,:] = softmax_t(s[:,b]);   ctx[b,:] = sum_t a[b,t] * V[t*Bp+b,:]
 *   V [T*Bp][W], U [T*Bp][W2], attn [B][T], ctx [B][W].
 *   U == NULL: all scores equal, a = 1/T -- mean pooling over time, the no-attention ablation
 *   (torch.mean(lstm_out, dim=1), 09_sensitivity_analysis.py:236); w2, b2, W2 are then ignored.
 *   U != NULL with W2 == 0 (bf16 V, W == 256 or 512): U holds the finished scores S [B][T] of lob_attn_scores_bf16.   */
int lob_attn_pool_fwd_f32(const void* V, int v_bf16, const float* U, const float* w2, const float* b2,
                          float* ctx, float* attn, int T, int B, int Bp, int W, int W2,
                          void* stream);

/* Row softmax over `cols` (torch.softmax(logits, dim=1), 06_lstm_ode_integration.py:232). */
int lob_softmax_rows_f32(const float* in, float* out, int rows, int cols, void* stream);

/* ------------------------------------------------------------------------------------
 * Coupled LSTM -> ODE step 2 of predict_batch (06_lstm_ode_integration.py:372-401), one
 * window per lane, fp64 arithmetic:
 *   p_open = probs[b][0], p_closed = probs[b][1]                          (06:373-374)
 *   y0 = [.2,.2,.6] if p_closed > .6 else [.6,.2,.2] if p_open > .6 else [.33,.34,.33]
 *        then y0 /= sum(y0)                                    (06:377-382, 06:176)
 *   k_af,k_pf *= 1 + alpha p_closed; k_fa,k_pa *= 1 + alpha p_open; all six max(.001,.)
 *                                                                         (06:249-264)
 *   t = linspace(t0, t1, n_points); fixed-step RK4 of dy/dt = Q^T max(y,0) with
 *   `substeps` sub-steps per output interval (the reference calls LSODA)
 *                                                              (06:158-172, 06:177)
 *   traj = clip(traj,0,1); traj /= rowsum                                (06:178-179)
 *   pred = traj[-1][2] > 0.5                                             (06:396-401)
 * probs == NULL: un-modulated mode -- y0 is read from `y0_in` ([B][3] f64), rates are
 * base_rates for every window (CognitiveStateODE.solve, 06:174-180 / 05:137-169).
 * traj [B][n_points][3] f64 (may be NULL: only final/pred are written),
 * final_state [B][3] f64 (may be NULL), pred [B] int64 (may be NULL).
 * base_rates: HOST pointer to 6 doubles in the order k_ap,k_af,k_pa,k_pf,k_fa,k_fp.
 * ---------------------------------------------------------------------------------- */
#define LOB_ODE_RAW 1   /* 08_forecasting.py:132-153 variant: no clamp in the rhs, y0 used as given,
                         * no clip / renormalise of the output (predict_trajectory, 08:149-153)  */
int lob_ode_rk4_f64(const float* probs, const double* y0_in, const double* base_rates,
                    double alpha, int n_points, double t0, double t1, int substeps,
                    double* traj, double* final_state, int64_t* pred, int B, int flags,
                    void* stream);

/* y0[b] = prob_to_ode_state(P(closed)) (08_forecasting.py:215-234): A = 1-p; (F,P) = (.6p,.4p) if
 * p > .5 else (.3p,.3p); normalised.  probs [B][2] f32 -> y0 [B][3] f64.                   */
int lob_prob_to_state_f64(const float* probs, double* y0, int B, void* stream);

/* ------------------------------------------------------------------------------------
 * The steps either side of fwd+bwd in the reference's training loop (04_lstm_model.py:482-512),
 * over FLAT fp32 parameter / gradient / moment buffers (all tensors of the model back to back).
 * ---------------------------------------------------------------------------------- */

/* nn.CrossEntropyLoss(weight=class_weight) (04:430-435), reduction 'mean':
 *   loss[0]  = sum_i w[y_i] * nll_i / sum_i w[y_i]
 *   dlogits  = scale * d loss / d logits        (may be NULL; scale = 1/gradient_accumulation_steps, 04:489)
 *   correct[0] = #{i : argmax_c logits[i][c] == y_i}   (04:508-509; may be NULL)
 * logits [B][C] f32, target [B] int64, class_weight [C] f32 or NULL (all ones).  One workgroup.      */
int lob_weighted_ce_f32(const float* logits, const int64_t* target, const float* class_weight,
                        float* loss, float* dlogits, int* correct, int B, int C, float scale,
                        void* stream);

/* out[0] += sum x[i]^2 (the squared global gradient norm of clip_grad_norm_, 04:501).  x 16-B aligned;
 * the caller zeroes out[0].  scratch: 512 floats of workspace -> the per-workgroup partial sums are added in a
 * fixed order by a second launch, so the result is bit-reproducible (data-parallel ranks that hold the same
 * all-reduced gradient then take bit-identical optimizer steps); NULL -> fp32 atomics (order varies).        */
int lob_sumsq_f32(const float* x, int64_t n, float* out, float* scratch, void* stream);

/* g *= min(1, max_norm / (sqrt(normsq[0]) + 1e-6)): torch.nn.utils.clip_grad_norm_ in place (04:501),
 * normsq read on the device (no host round trip).                                                   */
int lob_clip_scale_f32(float* g, int64_t n, const float* normsq, float max_norm, void* stream);

/* One torch.optim.AdamW step (04:438, 502) on flat buffers, with the clip folded in:
 *   g' = grad_scale * g * min(1, max_norm / (grad_scale * sqrt(normsq[0]) + 1e-6))   (normsq NULL: no clip)
 *   p *= 1 - lr * weight_decay;  m = b1 m + (1-b1) g';  v = b2 v + (1-b2) g'^2
 *   p -= lr / (1 - b1^step) * m / (sqrt(v) / sqrt(1 - b2^step) + eps)                                */
int lob_adamw_f32(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1,
                  float beta2, float eps, float weight_decay, int64_t step, const float* normsq,
                  float max_norm, float grad_scale, void* stream);

/* out[row][0..Cp) = bf16(in[row][0..C)), columns C..Cp zero: the (B*T, 61) input windows as a 16-B-aligned bf16
 * operand (Cp = 64) for the mixed path's projection GEMMs (nn.Linear under autocast, 04:174, 04:487).   */
int lob_pad_cast_bf16(const float* in, void* out, int64_t rows, int C, int Cp, void* stream);

/* out[c] += scale * sum_{rows} |gx[row][c]|: the |input gradient| reduction of the gradient attribution
 * (07_explainability.py:257-258: X.grad[i].abs().mean(dim=0), summed over windows), gx [rows][C].    */
int lob_abs_colsum_f32(const float* gx, int64_t rows, int C, float scale, float* out, void* stream);

/* ------------------------------------------------------------------------------------
 * Derived weight images of one forward (+ backward) in ONE launch: what the reference's nn.LSTM / nn.Linear keep
 * internally (cuDNN's packed weights, 04_lstm_model.py:181-188) and what this build's kernels take as operands --
 * the two directions' W_ih concatenated, bf16 copies, transposes for the dX GEMMs, W_hh stacked, b_ih + b_hh.
 * `ops` is a HOST array of `nop` <= LOB_PREP_MAX descriptors (copied into the kernel arguments):
 *   dst[r][c] = src[r][c] (+ src2[r][c])          r < rows, c < cols; columns cols..pad_to-1 are written as 0
 *   LOB_PREP_TRANSPOSE: dst[c][r] = src[r][c]     (no src2 / pad_to)
 *   LOB_PREP_BF16: dst is bf16 (else fp32)
 * ld_src / ld_dst are leading dimensions in elements.
 * ---------------------------------------------------------------------------------- */
#define LOB_PREP_MAX 64
#define LOB_PREP_TRANSPOSE 1
#define LOB_PREP_BF16 2
#define LOB_PREP_ABSMAX 4   /* dst[0] (fp32, ZEROED by the caller) = max(dst[0], max |src[r][c]|): the operand range of the
                             * fp16-split kernels, see `range` of lob_gate_gemm_x_f32 / lob_lstm_rec_fwd_f32 (atomic max
                             * over blocks of 8192 elements; no other field of the op is used)                          */
#define LOB_PREP_LNBOUND 8  /* dst[0] = (sqrt(cols) max|src| + max|src2|) * reserved / 1000: an upper bound on
                             * |dropout(GELU(LayerNorm(.)))| from the LayerNorm's gain (src) and bias (src2) -- what the
                             * first LSTM layer's gate GEMM can see as an activation; src == NULL: dst[0] = reserved / 1000 */
typedef struct LobPrepOp {
    const float* src; const float* src2; void* dst;
    int rows, cols, ld_src, ld_dst, pad_to, kind, blk0 /* internal */, reserved;
} LobPrepOp;
int lob_prep_weights(const LobPrepOp* ops, int nop, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* LOB_H_ */
