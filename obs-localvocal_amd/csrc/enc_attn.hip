// enc_attn.hip — fused encoder self-attention (bf16): softmax(Q K^T / sqrt(dh)) V per head without the score matrix
// ever leaving the chip.
//
// Replaces, for the bf16 mode, the unfused sequence of whisper.cpp's encoder graph (mul_mat KQ -> soft_max -> mul_mat
// KQV) that moved ~370 MB per layer (fp32 scores written and re-read, bf16 probabilities written and
// re-read).  Bound: MFMA / exp throughput (11.8 GFLOP per layer at n_ctx 1500).
//
// One 256-thread workgroup per (128 queries, head); wave w owns 32 queries (two 16-query tiles).  Keys go by in steps
// of 64: the K rows [key][dim] and the V^T rows [dim][key] (the V projection already stores V transposed) are
// double-buffered through LDS with coalesced 128-byte rows.  Everything is computed transposed so that no operand
// ever changes lanes:
//   S^T[key][query] = K . Q^T       A = K tile (lane: key l&15, 8 dims), B = Q (lane: query l&15, 8 dims; registers)
//                                   -> lane (q = l&15, g = l>>4) holds keys 4g..4g+3 of each 16-key tile
//   O^T[dim][query] += V^T . P^T    B = P^T needs 8 contraction entries per lane: the lane's 4+4 probabilities of two
//                                   key tiles ARE such a fragment if the contraction index of the 32-key chunk is
//                                   ordered (g, tile, r); the V^T fragment is read from LDS in the same order (two
//                                   8-byte reads).  The running maximum / sum of a query live in the lanes that hold
//                                   its column (l&15), replicated over g; the sum is completed across g at the end.
#include "kernels.hpp"

namespace mw {

namespace {

using bf16x8b = __attribute__((ext_vector_type(8))) __bf16;
using f32x4 = __attribute__((ext_vector_type(4))) float;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;

constexpr int FQ = 128;      // queries per workgroup
constexpr int FK = 64;       // keys per step
constexpr int FLD = 64 + 8;  // LDS row stride (elements): conflict-free 16-byte fragment reads

using bf16x2b = __attribute__((ext_vector_type(2))) __bf16;
using f32x2 = __attribute__((ext_vector_type(2))) float;
__device__ __forceinline__ unsigned pack_bf16(float a, float b) {
    // round to nearest even, like the conversions of the unfused path: one v_cvt_pk_bf16_f32 (as integer arithmetic the
    // pair cost seven VALU instructions, 112 per key step — the kernel is bound by its softmax VALU work, not by the MFMAs)
    return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{a, b}, bf16x2b));
}

// MXO: the output leaves as MX-fp8 (e4m3, one E8M0 scale per 32 dims of a row, in the A-operand scale order of gemm_mx.hip)
// for the attention-output product of MW_PRECISION_FP8_MX, rows numbered through the windows of the pass.
// (Registers: 132 -> three waves per SIMD.  Held to 128 for four — amdgpu_waves_per_eu(4, 4), no spills — one window's launch
// was slower, 34 against 29 us, and 16 windows' equal within the run-to-run spread.)
template <bool MXO>
__global__ __launch_bounds__(256) void enc_flash_attn_kernel(const bf16_t *__restrict__ qk, const bf16_t *__restrict__ vt,
                                                             bf16_t *__restrict__ out, int n_ctx, int tp, int d, float scale_log2,
                                                             int ldq_, unsigned char *__restrict__ out8, unsigned char *__restrict__ osc,
                                                             int MB) {
    // blockIdx.z: window of a batched encoder pass (each window has its own Q|K, V^T and output slab, back to back)
    qk += (long)blockIdx.z * n_ctx * ldq_;
    vt += (long)blockIdx.z * d * tp;
    if constexpr (!MXO) out += (long)blockIdx.z * n_ctx * d;
    __shared__ __attribute__((aligned(16))) bf16_t kt[2][FK * FLD];
    __shared__ __attribute__((aligned(16))) bf16_t vs[2][64 * FLD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, fq = lane >> 4;
    const int h = blockIdx.y, q0 = blockIdx.x * FQ + wave * 32;
    const long ldq = ldq_;  // 2 d (Q | K rows) or 3 d (Q | K | V rows of the MX-fp8 mode)
    // Q fragments of the wave's two query tiles (loop invariant)
    uint4 qf[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int s = 0; s < 2; ++s)
            qf[i][s] = *reinterpret_cast<const uint4 *>(qk + (long)min(q0 + 16 * i + fr, n_ctx - 1) * ldq + h * 64 + 32 * s + 8 * fq);
    // tile loader: thread -> row tid>>2 (a key of K / a dim of V^T), 32 bytes at element (tid&3)*16
    const int lrow = tid >> 2, lcol = (tid & 3) * 16;
    // The next step's K / V^T pieces travel in four first-class vector registers, loaded unconditionally (the step index
    // is clamped: the last step reloads its own tile).  As `uint4 rk[2], rv[2]` filled by a lambda under `if (step + 1 <
    // nsteps)` hipcc kept the arrays in scratch memory: every step waited for its loads at once to copy them there
    // (vmcnt right behind the loads), i.e. one full memory latency per step, 62 us per layer.
    u32x4 rk0, rk1, rv0, rv1;
#define ENC_GLOAD(step_)                                                                                  \
    do {                                                                                                  \
        const int s_ = (step_);                                                                           \
        const bf16_t *kp = qk + (long)min(s_ * FK + lrow, n_ctx - 1) * ldq + d + h * 64 + lcol;           \
        const bf16_t *vp = vt + (long)(h * 64 + lrow) * tp + s_ * FK + lcol;                              \
        rk0 = *reinterpret_cast<const u32x4 *>(kp);                                                       \
        rk1 = *reinterpret_cast<const u32x4 *>(kp + 8);                                                   \
        rv0 = *reinterpret_cast<const u32x4 *>(vp);                                                       \
        rv1 = *reinterpret_cast<const u32x4 *>(vp + 8);                                                   \
    } while (0)
#define ENC_LSTORE(buf_)                                                                  \
    do {                                                                                  \
        *reinterpret_cast<u32x4 *>(&kt[buf_][lrow * FLD + lcol]) = rk0;                   \
        *reinterpret_cast<u32x4 *>(&kt[buf_][lrow * FLD + lcol + 8]) = rk1;               \
        *reinterpret_cast<u32x4 *>(&vs[buf_][lrow * FLD + lcol]) = rv0;                   \
        *reinterpret_cast<u32x4 *>(&vs[buf_][lrow * FLD + lcol + 8]) = rv1;               \
    } while (0)
    f32x4 o[2][4];
    float m[2] = {-INFINITY, -INFINITY}, l[2] = {0.0f, 0.0f};
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int t = 0; t < 4; ++t) o[i][t] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    const int nsteps = tp / FK;
    ENC_GLOAD(0);
    ENC_LSTORE(0);
    __syncthreads();
    for (int step = 0; step < nsteps; ++step) {
        const int cur = step & 1;
        ENC_GLOAD(min(step + 1, nsteps - 1));
        const bf16_t *kb = kt[cur], *vb = vs[cur];
        // ---- S^T tiles ----
        f32x4 sa[2][4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const uint4 a0 = *reinterpret_cast<const uint4 *>(kb + (16 * t + fr) * FLD + 8 * fq);
            const uint4 a1 = *reinterpret_cast<const uint4 *>(kb + (16 * t + fr) * FLD + 32 + 8 * fq);
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8b, a0), __builtin_bit_cast(bf16x8b, qf[i][0]), acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8b, a1), __builtin_bit_cast(bf16x8b, qf[i][1]), acc, 0, 0, 0);
                sa[i][t] = acc;
            }
        }
        // ---- online softmax (base 2), probabilities as P^T fragments ----
        // VALU diet (the MFMAs of a key step take 512 cycles, the softmax took three times that): the running maximum is
        // kept on the RAW scores and the scale rides in the exponent's fma; keys beyond n_ctx exist only in the last step;
        // the accumulators are rescaled only when some query's maximum moved (corr is exactly 1 otherwise).
        uint4 pb[2][2];
        const int key_base = step * FK + 4 * fq;
        const bool tail = step == nsteps - 1;  // wave-uniform
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            if (tail) {
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (key_base + 16 * t + r >= n_ctx) sa[i][t][r] = -INFINITY;
            }
            float mx = m[i];
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) mx = fmaxf(mx, sa[i][t][r]);
            mx = xor_max<32>(xor_max<16>(mx));  // (common.hpp: row / half-wave swaps, no LDS round trip)
            const bool moved = __builtin_amdgcn_ballot_w64(mx > m[i]) != 0;  // wave-uniform
            const float nms = -mx * scale_log2;
            float ps = 0.0f;
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float p = __builtin_amdgcn_exp2f(fmaf(sa[i][t][r], scale_log2, nms));
                    sa[i][t][r] = p;
                    ps += p;
                }
            if (moved) {
                const float corr = __builtin_amdgcn_exp2f((m[i] - mx) * scale_log2);  // first step: exp2(-inf) = 0
                m[i] = mx;
                l[i] *= corr;
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) o[i][t][r] *= corr;
            }
            l[i] += ps;  // this lane's share (its 16 keys of the step); completed across g at the end
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                pb[i][c].x = pack_bf16(sa[i][2 * c][0], sa[i][2 * c][1]);
                pb[i][c].y = pack_bf16(sa[i][2 * c][2], sa[i][2 * c][3]);
                pb[i][c].z = pack_bf16(sa[i][2 * c + 1][0], sa[i][2 * c + 1][1]);
                pb[i][c].w = pack_bf16(sa[i][2 * c + 1][2], sa[i][2 * c + 1][3]);
            }
        }
        // ---- O^T += V^T . P^T ----
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const uint2 v0 = *reinterpret_cast<const uint2 *>(vb + (16 * t + fr) * FLD + 32 * c + 4 * fq);
                const uint2 v1 = *reinterpret_cast<const uint2 *>(vb + (16 * t + fr) * FLD + 32 * c + 16 + 4 * fq);
                const uint4 af = make_uint4(v0.x, v0.y, v1.x, v1.y);
#pragma unroll
                for (int i = 0; i < 2; ++i)
                    o[i][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8b, af), __builtin_bit_cast(bf16x8b, pb[i][c]),
                                                                      o[i][t], 0, 0, 0);
            }
        // the other buffer was last read before the previous barrier (after the last step: a harmless rewrite)
        ENC_LSTORE(cur ^ 1);
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        float ls = l[i];
        ls = xor_sum<32>(xor_sum<16>(ls));
        const float inv = 1.0f / ls;
        const int q = q0 + 16 * i + fr;
        if constexpr (MXO) {
            // a row's 64 dims of this head: 16 t + 4 fq + r over the four lanes fq of the row; a block of 32 dims is t = 0, 1 or 2, 3
            const long row = (long)blockIdx.z * n_ctx + q;
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                float x[8];
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) x[4 * t + r] = o[i][2 * b + t][r] * inv;
                float amax = 0.0f;
#pragma unroll
                for (int r = 0; r < 8; ++r) amax = fmaxf(amax, fabsf(x[r]));
                amax = xor_max<32>(xor_max<16>(amax));
                const unsigned sb = mx_scale_byte(amax);
                const float is = __uint_as_float((254u - sb) << 23);
                if (q < n_ctx) {
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
                        int pk = __builtin_amdgcn_cvt_pk_fp8_f32(x[4 * t] * is, x[4 * t + 1] * is, 0, false);
                        pk = __builtin_amdgcn_cvt_pk_fp8_f32(x[4 * t + 2] * is, x[4 * t + 3] * is, pk, true);
                        *reinterpret_cast<int *>(out8 + row * d + h * 64 + 16 * (2 * b + t) + 4 * fq) = pk;
                    }
                    if (fq == 0) osc[mx_sa_off((int)row, 2 * h + b, MB)] = (unsigned char)sb;
                }
            }
        } else if (q < n_ctx) {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                uint2 pk;
                pk.x = pack_bf16(o[i][t][0] * inv, o[i][t][1] * inv);
                pk.y = pack_bf16(o[i][t][2] * inv, o[i][t][3] * inv);
                *reinterpret_cast<uint2 *>(out + (long)q * d + h * 64 + 16 * t + 4 * fq) = pk;
            }
        }
    }
}

#undef ENC_GLOAD
#undef ENC_LSTORE

}  // namespace

void launch_enc_flash_attn(const void *qk, const void *vt, void *out, int n_ctx, int tp, int d, int H, hipStream_t st, int n_windows,
                           int ldq, unsigned char *out_mx, unsigned char *out_mx_scales) {
    if (ldq == 0) ldq = 2 * d;
    MW_CHECK(qk && vt && (out || out_mx) && n_ctx > 0 && d == H * 64 && tp % FK == 0 && tp >= n_ctx && tp - n_ctx < FK && ldq >= 2 * d &&
                 ldq % 8 == 0,
             "enc_flash_attn: shape");
    const float scale_log2 = (1.0f / sqrtf(64.0f)) * 1.4426950408889634f;
    MW_CHECK(n_windows >= 1 && n_windows <= 65535, "enc_flash_attn: window count");
    const dim3 grid(ceil_div(n_ctx, FQ), H, n_windows);
    if (out_mx) {
        MW_CHECK(out_mx_scales && ((uintptr_t)out_mx % 4) == 0 && (long)n_windows * n_ctx < (1L << 30), "enc_flash_attn: MX output");
        hipLaunchKernelGGL(enc_flash_attn_kernel<true>, grid, dim3(256), 0, st, (const bf16_t *)qk, (const bf16_t *)vt, (bf16_t *)nullptr, n_ctx,
                           tp, d, scale_log2, ldq, out_mx, out_mx_scales, mx_row_blocks_a(n_windows * n_ctx));
    } else {
        hipLaunchKernelGGL(enc_flash_attn_kernel<false>, grid, dim3(256), 0, st, (const bf16_t *)qk, (const bf16_t *)vt, (bf16_t *)out, n_ctx, tp,
                           d, scale_log2, ldq, (unsigned char *)nullptr, (unsigned char *)nullptr, 0);
    }
    MW_HIP(hipGetLastError());
}

}  // namespace mw
