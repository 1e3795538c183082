// One launch builds every derived weight image a forward (and its backward) needs: the per-layer concatenation of the
// two directions' W_ih, its bf16 copy, its transpose, the stacked W_hh, b_ih + b_hh, the zero-padded projection weight,
// the transposed classifier / attention weights.  torch did this with ~50 tiny launches per training step (cat, stack,
// add, to(bf16), t().contiguous(), zeros + slice copy); at B <= 1024 those launches are a visible share of the step.
// The images are derived from the live parameters EVERY forward (no cache to invalidate: an optimizer that writes the
// parameters through raw pointers, as FusedAdamW does, never bumps a torch version counter).
#include "lob_common.h"

namespace {

constexpr int MAXOPS = LOB_PREP_MAX;
struct PrepArgs { LobPrepOp op[MAXOPS]; int nop; };

__global__ __launch_bounds__(256) void prep_weights_kernel(PrepArgs a) {
    // find this block's operation (blk0 is ascending)
    int o = 0;
#pragma unroll 1
    for (int i = 1; i < a.nop; ++i) if ((int)blockIdx.x >= a.op[i].blk0) o = i;
    const LobPrepOp& p = a.op[o];
    const int b = blockIdx.x - p.blk0;
    if (p.kind & LOB_PREP_LNBOUND) {      // one workgroup: the LayerNorm-derived activation bound (or a constant)
        __shared__ float red[2][4];
        float m1 = 0.f, m2 = 0.f;
        if (p.src)
            for (int i = threadIdx.x; i < p.cols; i += 256) m1 = fmaxf(m1, fabsf(p.src[i]));
        if (p.src2)
            for (int i = threadIdx.x; i < p.cols; i += 256) m2 = fmaxf(m2, fabsf(p.src2[i]));
        m1 = wave_max(m1); m2 = wave_max(m2);
        if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = m1; red[1][threadIdx.x >> 6] = m2; }
        __syncthreads();
        if (threadIdx.x == 0) {
            m1 = fmaxf(fmaxf(red[0][0], red[0][1]), fmaxf(red[0][2], red[0][3]));
            m2 = fmaxf(fmaxf(red[1][0], red[1][1]), fmaxf(red[1][2], red[1][3]));
            reinterpret_cast<float*>(p.dst)[0] = (p.src ? sqrtf((float)p.cols) * m1 + m2 : 1.0f) * (p.reserved * 1e-3f);
        }
        return;
    }
    if (p.kind & LOB_PREP_ABSMAX) {       // max |src| of this block's 8192 elements -> atomic max on the (zeroed) slot:
        __shared__ float red[4];           // non-negative floats order like their bit patterns
        const long n = (long)p.rows * p.cols;
        float m = 0.f;
#pragma unroll 8
        for (int k = 0; k < 32; ++k) {
            const long i = (long)b * 8192 + k * 256 + threadIdx.x;
            if (i < n) m = fmaxf(m, fabsf(p.src[(i / p.cols) * p.ld_src + i % p.cols]));
        }
        m = wave_max(m);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
        __syncthreads();
        if (threadIdx.x == 0) {
            m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
            atomicMax(reinterpret_cast<unsigned*>(p.dst), __builtin_bit_cast(unsigned, m));
        }
        return;
    }
    const int out_cols = p.pad_to > p.cols ? p.pad_to : p.cols;       // columns written per output row (non-transposing)
    const bool tr = (p.kind & LOB_PREP_TRANSPOSE) != 0, bf = (p.kind & LOB_PREP_BF16) != 0;
    const long n = tr ? (long)p.rows * p.cols : (long)p.rows * out_cols;
    for (long i = (long)b * 1024 + threadIdx.x; i < n && i < (long)(b + 1) * 1024; i += 256) {
        float v;
        long di;
        if (tr) {                      // dst[c][r] = src[r][c]: dst-linear order (coalesced writes)
            const int c = (int)(i / p.rows), r = (int)(i % p.rows);
            v = p.src[(long)r * p.ld_src + c];
            di = (long)c * p.ld_dst + r;
        } else {
            const int r = (int)(i / out_cols), c = (int)(i % out_cols);
            v = c < p.cols ? p.src[(long)r * p.ld_src + c] : 0.f;
            if (p.src2 && c < p.cols) v += p.src2[(long)r * p.ld_src + c];
            di = (long)r * p.ld_dst + c;
        }
        if (bf) reinterpret_cast<__bf16*>(p.dst)[di] = (__bf16)v;
        else    reinterpret_cast<float*>(p.dst)[di] = v;
    }
}

}  // namespace

extern "C" int lob_prep_weights(const LobPrepOp* ops, int nop, void* stream) {
    if (!ops || nop <= 0 || nop > MAXOPS) return LOB_E_ARG;
    PrepArgs a;
    int blocks = 0;
    for (int i = 0; i < nop; ++i) {
        a.op[i] = ops[i];
        const LobPrepOp& p = ops[i];
        if (p.kind & (LOB_PREP_ABSMAX | LOB_PREP_LNBOUND)) {
            if (!p.dst || ((p.kind & LOB_PREP_ABSMAX) && !p.src)) return LOB_E_ARG;
            if (p.src && (p.rows <= 0 || p.cols <= 0 || p.ld_src < p.cols)) return LOB_E_ARG;
            a.op[i].blk0 = blocks;
            blocks += (p.kind & LOB_PREP_ABSMAX) ? (int)(((long)p.rows * p.cols + 8191) / 8192) : 1;
            continue;
        }
        if (!p.src || !p.dst || p.rows <= 0 || p.cols <= 0 || p.ld_src < p.cols) return LOB_E_ARG;
        const bool tr = (p.kind & LOB_PREP_TRANSPOSE) != 0;
        if (tr && (p.pad_to || p.src2 || p.ld_dst < p.rows)) return LOB_E_SHAPE;
        const int out_cols = p.pad_to > p.cols ? p.pad_to : p.cols;
        if (!tr && p.ld_dst < out_cols) return LOB_E_SHAPE;
        a.op[i].blk0 = blocks;
        const long n = tr ? (long)p.rows * p.cols : (long)p.rows * out_cols;
        blocks += (int)((n + 1023) / 1024);
    }
    a.nop = nop;
    hipLaunchKernelGGL(prep_weights_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a);
    LOB_CHECK_LAUNCH();
    return 0;
}
