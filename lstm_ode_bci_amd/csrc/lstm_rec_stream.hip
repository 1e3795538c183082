// Recurrent kernels for hidden sizes whose W_hh does not fit one CU's register file (H = 256: the
// size the reference's real checkpoints use, 04_lstm_model.py:877; W_hh is 1 MB per direction) and
// for small multiples of 32.  Same persistent structure, fragment-order P/G/C layout and exact-fp32
// MFMA as the H = 128 kernels (lstm_rec_f32.hip), but H/32 waves per workgroup (wave w owns hidden
// units [32w, 32w+32)) and the W_hh operand is STREAMED from L2 every step instead of living in
// VGPRs: 1 MB per step per CU at H = 256 against 27 us of MFMA time = 37 GB/s per CU, well inside
// the L2 rate; W_hh (2 MB for both directions) stays L2-resident on every XCD.
#include "lob_common.h"

namespace {

typedef __bf16 bf16x4_s __attribute__((ext_vector_type(4)));

template <int HH, bool SAVE>
__global__ __launch_bounds__(HH * 2, (HH > 128) ? 2 : 1) void lstm_rec_fwd_stream_kernel(
    float* __restrict__ P, const float* __restrict__ Whh, float* __restrict__ Y,
    float* __restrict__ Csave, int T, int Bp) {
    constexpr int NW = HH / 32, HLD = HH + 4, KB_UNROLL = (NW > 4 ? 1 : NW);
    __shared__ __attribute__((aligned(16))) float hs[2 * 32 * HLD];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int bt = blockIdx.x, d = blockIdx.y, D = gridDim.y, NBT = gridDim.x;
    const int l31 = lane & 31, hi = lane >> 5;

    for (int i = tid; i < 2 * 32 * HLD; i += HH * 2) hs[i] = 0.f;
    float c[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) c[r] = 0.f;

    const size_t pstep = (size_t)NBT * NW * 4096, cstep = (size_t)NBT * NW * 1024;
    float* pblk = P + (((size_t)d * T * NBT + bt) * NW + w) * 4096;
    float* cblk = SAVE ? Csave + (((size_t)d * T * NBT + bt) * NW + w) * 1024 : nullptr;
    const unsigned frag_off = lane * 4;
    const unsigned y_off = (unsigned)(4 * hi * (D * HH) + l31);
    // B operand rows: W_hh[g*H + 32w + l31][...], 16 contiguous k per lane half
    const float* wwave = Whh + ((size_t)d * 4 * HH + 32 * w) * HH;
    const unsigned w_off = (unsigned)(l31 * HH + 16 * hi);

    const int t_first = d ? T - 1 : 0, dt = d ? -1 : 1;
    f32x16 pn[4];
    {
        const float* p = pblk + (size_t)t_first * pstep;
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 v = *reinterpret_cast<const f32x4*>((p + g * 1024 + q * 256) + frag_off);
                pn[g][4 * q] = v[0]; pn[g][4 * q + 1] = v[1]; pn[g][4 * q + 2] = v[2]; pn[g][4 * q + 3] = v[3];
            }
    }
    __syncthreads();

    int cur = 0;
    for (int step = 0; step < T; ++step) {
        const int t = t_first + dt * step;
        f32x16 acc[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) acc[g] = pn[g];
        if (step + 1 < T) {
            const float* p = pblk + (size_t)(t + dt) * pstep;
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 v = *reinterpret_cast<const f32x4*>((p + g * 1024 + q * 256) + frag_off);
                    pn[g][4 * q] = v[0]; pn[g][4 * q + 1] = v[1]; pn[g][4 * q + 2] = v[2]; pn[g][4 * q + 3] = v[3];
                }
        }
        const float* hrow = hs + cur * 32 * HLD + l31 * HLD + 16 * hi;
        // one k-block (16 float4 of W per lane) in flight at a time at H = 256: the register budget is
        // 256 per wave (2 waves per SIMD), and the partner wave's MFMAs cover the L2 latency
#pragma unroll KB_UNROLL
        for (int kb = 0; kb < NW; ++kb) {
            f32x4 a[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) a[q] = *reinterpret_cast<const f32x4*>(hrow + 32 * kb + 4 * q);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 wv[4];
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    wv[q] = *reinterpret_cast<const f32x4*>((wwave + (size_t)g * HH * HH + 32 * kb + 4 * q) + w_off);
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[g] = mfma32(a[q][e], wv[q][e], acc[g]);
            }
        }
        float* hnext = hs + (cur ^ 1) * 32 * HLD + 32 * w + l31 + 4 * hi * HLD;
        float* yrow = Y + ((size_t)t * Bp + bt * 32) * (D * HH) + d * HH + 32 * w;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float ig = fast_sigmoid(acc[0][r]);
            const float fg = fast_sigmoid(acc[1][r]);
            const float gg = fast_tanh(acc[2][r]);
            const float og = fast_sigmoid(acc[3][r]);
            c[r] = __builtin_fmaf(fg, c[r], ig * gg);
            const float h = og * fast_tanh(c[r]);
            const int row = (r & 3) + 8 * (r >> 2);
            hnext[row * HLD] = h;
            (yrow + (size_t)row * (D * HH))[y_off] = h;
            if (SAVE) { acc[0][r] = ig; acc[1][r] = fg; acc[2][r] = gg; acc[3][r] = og; }
        }
        if (SAVE) {
            float* p = pblk + (size_t)t * pstep;
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    f32x4 v = {acc[g][4 * q + 0], acc[g][4 * q + 1], acc[g][4 * q + 2], acc[g][4 * q + 3]};
                    *reinterpret_cast<f32x4*>((p + g * 1024 + q * 256) + frag_off) = v;
                }
            float* cp = cblk + (size_t)t * cstep;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                f32x4 v = {c[4 * q + 0], c[4 * q + 1], c[4 * q + 2], c[4 * q + 3]};
                *reinterpret_cast<f32x4*>((cp + q * 256) + frag_off) = v;
            }
        }
        __syncthreads();
        cur ^= 1;
    }
}

template <int HH, bool DP_BF16>
__global__ __launch_bounds__(HH * 2, (HH > 128) ? 2 : 1) void lstm_rec_bwd_stream_kernel(
    const float* __restrict__ G, const float* __restrict__ Csave, const float* __restrict__ Whh,
    const float* __restrict__ dY, void* __restrict__ dPv, float* __restrict__ dbias, int T, int Bp) {
    constexpr int NW = HH / 32, DGLD = 4 * HH + 4, NB_UNROLL = (NW > 4 ? 1 : 4);
    __shared__ __attribute__((aligned(16))) float dgs[32 * DGLD];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int bt = blockIdx.x, d = blockIdx.y, D = gridDim.y, NBT = gridDim.x;
    const int l31 = lane & 31, hi = lane >> 5;

    const size_t gstep = (size_t)NBT * NW * 4096, cstep = (size_t)NBT * NW * 1024;
    const float* gwave = G + (((size_t)d * T * NBT + bt) * NW + w) * 4096;
    const float* cwave = Csave + (((size_t)d * T * NBT + bt) * NW + w) * 1024;
    const unsigned frag_off = lane * 4;
    const int DH = D * HH, D4H = D * 4 * HH;
    const float* dywave = dY + (size_t)(bt * 32) * DH + d * HH + 32 * w;
    const unsigned dy_off = (unsigned)(4 * hi * DH + l31);
    // B operand of dh = dgates * W_hh: W_hh[n = 32nb + 16hi + s][32w + l31] (column walk, 128-B segments)
    const float* wwave = Whh + (size_t)d * 4 * HH * HH + 32 * w;
    const unsigned w_off = (unsigned)(16 * hi * HH + l31);

    const int t_first = d ? 0 : T - 1, dt = d ? 1 : -1;
    f32x16 gt[4], ct, cp, dhrec;
    float dy[16], dcarry[16];
    float dbsum[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int r = 0; r < 16; ++r) { dcarry[r] = 0.f; dhrec[r] = 0.f; }

    auto load_c = [&](int t, f32x16& dst) {
        if (t >= 0 && t < T) {
            const float* cq = cwave + (size_t)t * cstep;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 v = *reinterpret_cast<const f32x4*>((cq + q * 256) + frag_off);
                dst[4 * q] = v[0]; dst[4 * q + 1] = v[1]; dst[4 * q + 2] = v[2]; dst[4 * q + 3] = v[3];
            }
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) dst[r] = 0.f;
        }
    };
    auto load_step = [&](int t) {
        const float* gp = gwave + (size_t)t * gstep;
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 v = *reinterpret_cast<const f32x4*>((gp + g * 1024 + q * 256) + frag_off);
                gt[g][4 * q] = v[0]; gt[g][4 * q + 1] = v[1]; gt[g][4 * q + 2] = v[2]; gt[g][4 * q + 3] = v[3];
            }
        load_c(t + dt, cp);
        const float* dp = dywave + (size_t)t * Bp * DH;
#pragma unroll
        for (int r = 0; r < 16; ++r) dy[r] = (dp + ((r & 3) + 8 * (r >> 2)) * DH)[dy_off];
    };
    load_c(t_first, ct);
    load_step(t_first);

    for (int step = 0; step < T; ++step) {
        const int t = t_first + dt * step;
        float* dgw = dgs + 32 * w + l31 + 4 * hi * DGLD;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float ig = gt[0][r], fg = gt[1][r], gg = gt[2][r], og = gt[3][r];
            const float dh = dy[r] + dhrec[r];
            const float tc = fast_tanh(ct[r]);
            const float dc = dcarry[r] + dh * og * (1.f - tc * tc);
            dcarry[r] = dc * fg;
            float* p = dgw + ((r & 3) + 8 * (r >> 2)) * DGLD;
            const float v0 = dc * gg * ig * (1.f - ig), v1 = dc * cp[r] * fg * (1.f - fg);
            const float v2 = dc * ig * (1.f - gg * gg), v3 = dh * tc * og * (1.f - og);
            p[0 * HH] = v0; p[1 * HH] = v1; p[2 * HH] = v2; p[3 * HH] = v3;
            dbsum[0] += v0; dbsum[1] += v1; dbsum[2] += v2; dbsum[3] += v3;
        }
        ct = cp;
        __syncthreads();
        if (step + 1 < T) load_step(t + dt);
#pragma unroll
        for (int r = 0; r < 16; ++r) dhrec[r] = 0.f;
        const float* arow = dgs + l31 * DGLD + 16 * hi;
#pragma unroll NB_UNROLL
        for (int nb = 0; nb < 4 * NW; ++nb) {
            f32x4 a[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) a[q] = *reinterpret_cast<const f32x4*>(arow + 32 * nb + 4 * q);
            float wv[16];
#pragma unroll
            for (int s = 0; s < 16; ++s) wv[s] = (wwave + (size_t)(32 * nb + s) * HH)[w_off];
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int e = 0; e < 4; ++e) dhrec = mfma32(a[q][e], wv[4 * q + e], dhrec);
        }
        // dgates tile -> dP rows (4H floats each)
        for (int idx = tid; idx < 32 * HH; idx += HH * 2) {
            const int row = idx / HH, c4 = (idx % HH) * 4;
            const f32x4 v = *reinterpret_cast<const f32x4*>(dgs + row * DGLD + c4);
            const size_t o = ((size_t)t * Bp + bt * 32 + row) * D4H + d * 4 * HH + c4;
            if (DP_BF16) {
                bf16x4_s hv = {(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
                *reinterpret_cast<bf16x4_s*>(reinterpret_cast<__bf16*>(dPv) + o) = hv;
            } else {
                *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(dPv) + o) = v;
            }
        }
        __syncthreads();
    }
    if (dbias) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float v = dbsum[g] + __shfl_xor(dbsum[g], 32, 64);
            if (hi == 0) atomicAdd(dbias + (size_t)d * 4 * HH + g * HH + 32 * w + l31, v);
        }
    }
}

template <int HH>
int launch_fwd(float* P, const float* Whh, float* Y, float* Csave, int T, int Bp, int D, int save, hipStream_t s) {
    const dim3 grid(Bp / 32, D), block(HH * 2);
    if (save) hipLaunchKernelGGL((lstm_rec_fwd_stream_kernel<HH, true>), grid, block, 0, s, P, Whh, Y, Csave, T, Bp);
    else      hipLaunchKernelGGL((lstm_rec_fwd_stream_kernel<HH, false>), grid, block, 0, s, P, Whh, Y, Csave, T, Bp);
    LOB_CHECK_LAUNCH();
    return 0;
}
template <int HH>
int launch_bwd(const float* G, const float* Cs, const float* Whh, const float* dY, void* dP, int bf, float* dbias,
               int T, int Bp, int D, hipStream_t s) {
    const dim3 grid(Bp / 32, D), block(HH * 2);
    if (bf) hipLaunchKernelGGL((lstm_rec_bwd_stream_kernel<HH, true>), grid, block, 0, s, G, Cs, Whh, dY, dP, dbias, T, Bp);
    else    hipLaunchKernelGGL((lstm_rec_bwd_stream_kernel<HH, false>), grid, block, 0, s, G, Cs, Whh, dY, dP, dbias, T, Bp);
    LOB_CHECK_LAUNCH();
    return 0;
}

}  // namespace

// Internal entry points used by lob_lstm_rec_fwd_f32 / lob_lstm_rec_bwd_f32 (lstm_rec_f32.hip).
int lob_stream_supports(int H) { return H == 32 || H == 64 || H == 256; }

int lob_stream_fwd(float* P, const float* Whh, float* Y, float* Csave, int T, int Bp, int H, int D, int save,
                   hipStream_t s) {
    switch (H) {
        case 32: return launch_fwd<32>(P, Whh, Y, Csave, T, Bp, D, save, s);
        case 64: return launch_fwd<64>(P, Whh, Y, Csave, T, Bp, D, save, s);
        case 256: return launch_fwd<256>(P, Whh, Y, Csave, T, Bp, D, save, s);
    }
    return LOB_E_SHAPE;
}

int lob_stream_bwd(const float* G, const float* Cs, const float* Whh, const float* dY, void* dP, int bf,
                   float* dbias, int T, int Bp, int H, int D, hipStream_t s) {
    switch (H) {
        case 32: return launch_bwd<32>(G, Cs, Whh, dY, dP, bf, dbias, T, Bp, D, s);
        case 64: return launch_bwd<64>(G, Cs, Whh, dY, dP, bf, dbias, T, Bp, D, s);
        case 256: return launch_bwd<256>(G, Cs, Whh, dY, dP, bf, dbias, T, Bp, D, s);
    }
    return LOB_E_SHAPE;
}
