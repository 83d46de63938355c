// dec.hip — K7/K8/K11: the KV-cached decoder step for gfx950.
//
// Replaces whisper.cpp's whisper_decode_internal graph (token+position embedding, per layer
// {LN, self-attention over the KV cache, LN, cross-attention over the encoder K/V, LN, MLP}, final LN,
// tied-embedding logits), reached once per sampled token from whisper_full
// (/root/reference/src/whisper-utils/whisper-processing.cpp:218-219).
//
// Bound: HBM.  A decode step multiplies a handful of activation rows (beams) by every weight of the
// decoder, so each weight byte is used once per step: the projections are weight-STREAMING kernels —
// weights go straight from HBM to VGPRs in 16-byte loads (no LDS round trip for an operand nobody
// shares), the few activation rows sit in LDS (LayerNorm fused into the prologue), each wave owns R
// output features and reduces with wave shuffles.  The attention kernels read the HBM-resident K/V
// caches with one 64-lane wavefront per (row, head), softmax by wavefront reductions.  Beam search
// never copies KV: a [slot][position] -> physical-slot table is reshuffled instead.
#include "kernels.hpp"

#include <hip/hip_bf16.h>

namespace mw {

namespace {

__device__ __forceinline__ float wave_sum(float v) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ float gelu_f(float x) {
    const float c = 0.79788456080286535587989211986876f, a = 0.044715f;
    return 0.5f * x * (1.0f + tanhf(c * x * (1.0f + a * x * x)));
}
__device__ __forceinline__ float bf16_lo(unsigned u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bf16_hi(unsigned u) { return __uint_as_float(u & 0xffff0000u); }
__device__ __forceinline__ bf16_t to_bf16(float v) {
    __hip_bfloat16 h = __float2bfloat16(v);
    return *reinterpret_cast<bf16_t *>(&h);
}
__device__ __forceinline__ void store_kv(float *p, float v) { *p = v; }
__device__ __forceinline__ void store_kv(bf16_t *p, float v) { *p = to_bf16(v); }

// 16-byte streaming (non-temporal) load: weights are read once per step
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
__device__ __forceinline__ uint4 ld_stream16(const void *p) {
    const u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(p));
    return make_uint4(v.x, v.y, v.z, v.w);
}

// elements per 16-byte chunk
template <typename T> struct Epc { static constexpr int v = 16 / (int)sizeof(T); };

// 16 bytes of T -> floats
__device__ __forceinline__ void unpack(const uint4 &u, float (&f)[8], bf16_t) {
    f[0] = bf16_lo(u.x); f[1] = bf16_hi(u.x); f[2] = bf16_lo(u.y); f[3] = bf16_hi(u.y);
    f[4] = bf16_lo(u.z); f[5] = bf16_hi(u.z); f[6] = bf16_lo(u.w); f[7] = bf16_hi(u.w);
}
__device__ __forceinline__ void unpack(const uint4 &u, float (&f)[4], float) {
    f[0] = __uint_as_float(u.x); f[1] = __uint_as_float(u.y); f[2] = __uint_as_float(u.z); f[3] = __uint_as_float(u.w);
}

// ------------------------------------------------------------------------------------------
// embedding
// ------------------------------------------------------------------------------------------
template <typename WT>
__global__ __launch_bounds__(256) void embed_kernel(const int *__restrict__ tokens, const int *__restrict__ pos,
                                                    const WT *__restrict__ te, const float *__restrict__ pe,
                                                    float *__restrict__ x, int d) {
    const int i = blockIdx.x;
    const long t = tokens[i], p = pos[i];
    for (int e = threadIdx.x; e < d; e += 256) {
        float v;
        if constexpr (sizeof(WT) == 2) v = __uint_as_float((unsigned)te[t * d + e] << 16);
        else v = te[t * d + e];
        x[(long)i * d + e] = v + pe[p * d + e];
    }
}

// ------------------------------------------------------------------------------------------
// weight-streaming projection: out[b][n] = R + act(LN(x)[b] . W[n] + bias[n]),  NB rows at once
// ------------------------------------------------------------------------------------------
// LDS index of activation element k inside a KC-chunk: for 2-byte weights a lane consumes 8
// consecutive k per 16-byte weight chunk; its two float4 halves live in separate planes so that
// consecutive lanes read consecutive 16-byte LDS slots (conflict-free ds_read_b128).
template <typename WT> __device__ __forceinline__ int xs_index(int k, int KC) {
    if constexpr (sizeof(WT) == 2) return ((k >> 2) & 1) * (KC >> 1) + ((k >> 3) << 2) + (k & 3);
    else return k;
}

template <typename WT, typename KVT, int NB, int R>
__global__ __launch_bounds__(256) void gemv_kernel(GemvArgs a, int row0, int KC) {
    constexpr int EPC = Epc<WT>::v;
    extern __shared__ __attribute__((aligned(16))) float xs[];  // [NB][KC]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int K = a.K, N = a.N;
    const int nb_valid = min(NB, a.n - row0);
    const int n_base = (blockIdx.x * 4 + wave) * R;

    float acc[R][NB];
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int b = 0; b < NB; ++b) acc[r][b] = 0.0f;

    const WT *wrow[R];
#pragma unroll
    for (int r = 0; r < R; ++r) wrow[r] = (const WT *)a.W + (long)min(n_base + r, N - 1) * K;

    for (int kc = 0; kc < K; kc += KC) {
        if (kc > 0) __syncthreads();
        // stage the activation chunk (rows beyond the valid count replicate the last row; never stored)
        for (int idx = tid; idx < NB * KC; idx += 256) {
            const int b = idx / KC, k = idx - b * KC;
            int row = row0 + min(b, nb_valid - 1);
            if (a.d_rows) row = a.d_rows[row];
            xs[b * KC + xs_index<WT>(k, KC)] = a.x[(long)row * a.ldx + kc + k];
        }
        __syncthreads();
        if (a.ln_w) {  // fused LayerNorm (requires K == KC): wave w normalises rows w, w+4
            for (int b = wave; b < NB; b += 4) {
                float *xr = xs + b * KC;
                float s = 0.0f;
                for (int k = lane; k < KC; k += 64) s += xr[k];
                const float mean = wave_sum(s) / (float)KC;
                float s2 = 0.0f;
                for (int k = lane; k < KC; k += 64) {
                    const float v = xr[k] - mean;
                    s2 += v * v;
                }
                const float rstd = 1.0f / sqrtf(wave_sum(s2) / (float)KC + 1e-5f);
                for (int k = lane; k < KC; k += 64) {
                    const int i = xs_index<WT>(k, KC);
                    xr[i] = (xr[i] - mean) * rstd * a.ln_w[k] + a.ln_b[k];
                }
            }
            __syncthreads();
        }
        if (n_base < N) {
            for (int k = lane * EPC; k < KC; k += 64 * EPC) {
                uint4 wv[R];
#pragma unroll
                for (int r = 0; r < R; ++r)
                    wv[r] = ld_stream16(wrow[r] + kc + k);
                float wf[R][EPC];
#pragma unroll
                for (int r = 0; r < R; ++r) unpack(wv[r], wf[r], WT{});
#pragma unroll
                for (int b = 0; b < NB; ++b) {
                    float xv[EPC];
                    if constexpr (EPC == 8) {
                        const float4 lo = *reinterpret_cast<const float4 *>(xs + b * KC + ((k >> 3) << 2));
                        const float4 hi = *reinterpret_cast<const float4 *>(xs + b * KC + (KC >> 1) + ((k >> 3) << 2));
                        xv[0] = lo.x; xv[1] = lo.y; xv[2] = lo.z; xv[3] = lo.w;
                        xv[4] = hi.x; xv[5] = hi.y; xv[6] = hi.z; xv[7] = hi.w;
                    } else {
                        const float4 lo = *reinterpret_cast<const float4 *>(xs + b * KC + k);
                        xv[0] = lo.x; xv[1] = lo.y; xv[2] = lo.z; xv[3] = lo.w;
                    }
#pragma unroll
                    for (int r = 0; r < R; ++r)
#pragma unroll
                        for (int e = 0; e < EPC; ++e) acc[r][b] = fmaf(wf[r][e], xv[e], acc[r][b]);
                }
            }
        }
    }
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int b = 0; b < NB; ++b) acc[r][b] = wave_sum(acc[r][b]);
    if (lane == 0) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int n = n_base + r;
            if (n >= N) continue;
            const float bias = a.bias ? a.bias[n] : 0.0f;
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                if (b >= nb_valid) continue;
                const int row = row0 + b;
                float v = acc[r][b] + bias;
                if (a.qkv) {
                    const int d = N / 3;
                    if (n < d) {
                        a.out[(long)row * a.ldo + n] = v * a.qk_scale;
                    } else {
                        const long off = ((long)a.d_slot[row] * a.n_text_ctx + a.d_pos[row]) * d;
                        if (n < 2 * d) store_kv((KVT *)a.kcache + off + (n - d), v * a.qk_scale);
                        else store_kv((KVT *)a.vcache + off + (n - 2 * d), v);
                    }
                } else {
                    v *= a.out_scale;
                    if (a.act == ACT_GELU) v = gelu_f(v);
                    if (a.R) v += a.R[(long)row * a.ldo + n];
                    a.out[(long)row * a.ldo + n] = v;
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// self-attention over the indirected KV cache: one wavefront per (row, head), dh = 64
// ------------------------------------------------------------------------------------------
template <typename KVT>
__global__ __launch_bounds__(64) void self_attn_kernel(const float *__restrict__ q, const KVT *__restrict__ kc,
                                                       const KVT *__restrict__ vc, const int *__restrict__ slot,
                                                       const int *__restrict__ pos, const int *__restrict__ kv_idx,
                                                       float *__restrict__ out, int d, int n_text_ctx) {
    constexpr int EPC = Epc<KVT>::v;
    constexpr int NCH = 64 / EPC;  // 16-byte chunks per head row
    constexpr int MAXIT = 7;       // 7 * 64 = 448 positions
    __shared__ float pr[MAXIT * 64];
    const int h = blockIdx.x, i = blockIdx.y, lane = threadIdx.x;
    const int np = pos[i] + 1;
    const int *idx = kv_idx + (long)slot[i] * n_text_ctx;
    float qv[64];
#pragma unroll
    for (int e = 0; e < 64; e += 4) {
        const float4 t = *reinterpret_cast<const float4 *>(q + (long)i * d + h * 64 + e);
        qv[e] = t.x; qv[e + 1] = t.y; qv[e + 2] = t.z; qv[e + 3] = t.w;
    }
    float s[MAXIT];
    float mx = -INFINITY;
#pragma unroll
    for (int it = 0; it < MAXIT; ++it) {
        const int p = it * 64 + lane;
        s[it] = -INFINITY;
        if (p < np) {
            const KVT *kr = kc + ((long)idx[p] * n_text_ctx + p) * d + h * 64;
            float dot = 0.0f;
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                const uint4 u = *reinterpret_cast<const uint4 *>(kr + c * EPC);
                float f[EPC];
                unpack(u, f, KVT{});
#pragma unroll
                for (int e = 0; e < EPC; ++e) dot = fmaf(qv[c * EPC + e], f[e], dot);
            }
            s[it] = dot;
            mx = fmaxf(mx, dot);
        }
    }
    mx = wave_max(mx);
    float sum = 0.0f;
#pragma unroll
    for (int it = 0; it < MAXIT; ++it) {
        const int p = it * 64 + lane;
        const float e = p < np ? expf(s[it] - mx) : 0.0f;
        pr[p] = e;
        sum += e;
    }
    const float inv = 1.0f / wave_sum(sum);
    __syncthreads();
    // P.V: lane = (position group jj, 8-dim chunk e8)
    const int jj = lane >> 3, e8 = lane & 7;
    float o[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int p0 = 0; p0 < np; p0 += 8) {
        const int p = p0 + jj;
        if (p < np) {
            const float w = pr[p];
            const KVT *vr = vc + ((long)idx[p] * n_text_ctx + p) * d + h * 64 + e8 * 8;
            if constexpr (EPC == 8) {
                float f[8];
                unpack(*reinterpret_cast<const uint4 *>(vr), f, KVT{});
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = fmaf(w, f[e], o[e]);
            } else {
                float f0[4], f1[4];
                unpack(*reinterpret_cast<const uint4 *>(vr), f0, KVT{});
                unpack(*reinterpret_cast<const uint4 *>(vr + 4), f1, KVT{});
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    o[e] = fmaf(w, f0[e], o[e]);
                    o[4 + e] = fmaf(w, f1[e], o[4 + e]);
                }
            }
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        o[e] += __shfl_xor(o[e], 8);
        o[e] += __shfl_xor(o[e], 16);
        o[e] += __shfl_xor(o[e], 32);
    }
    if (jj == 0) {
        float *op = out + (long)i * d + h * 64 + e8 * 8;
#pragma unroll
        for (int e = 0; e < 8; ++e) op[e] = o[e] * inv;
    }
}

// ------------------------------------------------------------------------------------------
// cross-attention, split over keys: partial (m, l, o[64]) per (row, head, split), then combine
// ------------------------------------------------------------------------------------------
constexpr int CROSS_PART = 66;  // floats per partial record

template <typename KVT, int NR>
__global__ __launch_bounds__(256) void cross_attn_kernel(const float *__restrict__ q, const KVT *__restrict__ kc,
                                                         const KVT *__restrict__ vc, float *__restrict__ part, int n,
                                                         int H, int d, int n_ctx, int per) {
    constexpr int EPC = Epc<KVT>::v;
    constexpr int NCH = 64 / EPC;
    __shared__ __attribute__((aligned(16))) float qs[NR][64];
    __shared__ float sc[NR][256];
    __shared__ float ml[NR][2];
    __shared__ float red[4][NR][64];
    const int h = blockIdx.x, sp = blockIdx.y, row0 = blockIdx.z * NR;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nr = min(NR, n - row0);
    const int j0 = sp * per, nk = min(per, n_ctx - j0);  // nk may be <= 0 for a trailing split
    for (int idx = tid; idx < NR * 64; idx += 256) {
        const int r = idx >> 6, e = idx & 63;
        qs[r][e] = r < nr ? q[(long)(row0 + r) * d + h * 64 + e] : 0.0f;
    }
    __syncthreads();
    // scores: one key per thread
    {
        float s[NR];
#pragma unroll
        for (int r = 0; r < NR; ++r) s[r] = 0.0f;
        if (tid < nk) {
            const KVT *kr = kc + (long)(j0 + tid) * d + h * 64;
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                float f[EPC];
                unpack(*reinterpret_cast<const uint4 *>(kr + c * EPC), f, KVT{});
#pragma unroll
                for (int r = 0; r < NR; ++r)
#pragma unroll
                    for (int e = 0; e < EPC; ++e) s[r] = fmaf(qs[r][c * EPC + e], f[e], s[r]);
            }
        }
#pragma unroll
        for (int r = 0; r < NR; ++r) sc[r][tid] = tid < nk ? s[r] : -INFINITY;
    }
    __syncthreads();
    // per-row max / exp / sum over this split (wave w: rows w, w+4)
    for (int r = wave; r < NR; r += 4) {
        float mx = -INFINITY;
        for (int j = lane; j < 256; j += 64) mx = fmaxf(mx, sc[r][j]);
        mx = wave_max(mx);
        float sum = 0.0f;
        for (int j = lane; j < 256; j += 64) {
            const float e = (j < nk) ? expf(sc[r][j] - mx) : 0.0f;
            sc[r][j] = e;
            sum += e;
        }
        sum = wave_sum(sum);
        if (lane == 0) {
            ml[r][0] = mx;
            ml[r][1] = sum;
        }
    }
    __syncthreads();
    // P.V: thread = (key group jj of 32, 8-dim chunk e8)
    const int jj = tid >> 3, e8 = tid & 7;
    float o[NR][8];
#pragma unroll
    for (int r = 0; r < NR; ++r)
#pragma unroll
        for (int e = 0; e < 8; ++e) o[r][e] = 0.0f;
    for (int j = jj; j < nk; j += 32) {
        const KVT *vr = vc + (long)(j0 + j) * d + h * 64 + e8 * 8;
        float f[8];
        if constexpr (EPC == 8) {
            unpack(*reinterpret_cast<const uint4 *>(vr), f, KVT{});
        } else {
            float f0[4], f1[4];
            unpack(*reinterpret_cast<const uint4 *>(vr), f0, KVT{});
            unpack(*reinterpret_cast<const uint4 *>(vr + 4), f1, KVT{});
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                f[e] = f0[e];
                f[4 + e] = f1[e];
            }
        }
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            const float w = sc[r][j];
#pragma unroll
            for (int e = 0; e < 8; ++e) o[r][e] = fmaf(w, f[e], o[r][e]);
        }
    }
    // reduce the 8 key groups of a wave (lane bits 3..5), then the 4 waves through LDS
#pragma unroll
    for (int r = 0; r < NR; ++r)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float v = o[r][e];
            v += __shfl_xor(v, 8);
            v += __shfl_xor(v, 16);
            v += __shfl_xor(v, 32);
            o[r][e] = v;
        }
    if ((lane >> 3) == 0) {
#pragma unroll
        for (int r = 0; r < NR; ++r)
#pragma unroll
            for (int e = 0; e < 8; ++e) red[wave][r][e8 * 8 + e] = o[r][e];
    }
    __syncthreads();
    for (int idx = tid; idx < NR * 64; idx += 256) {
        const int r = idx >> 6, e = idx & 63;
        if (r < nr) {
            const float v = red[0][r][e] + red[1][r][e] + red[2][r][e] + red[3][r][e];
            float *pp = part + (((long)(row0 + r) * H + h) * gridDim.y + sp) * CROSS_PART;
            pp[2 + e] = v;
            if (e == 0) {
                pp[0] = ml[r][0];
                pp[1] = ml[r][1];
            }
        }
    }
}

__global__ __launch_bounds__(64) void cross_combine_kernel(const float *__restrict__ part, float *__restrict__ out,
                                                           int H, int d, int splits) {
    const int h = blockIdx.x, i = blockIdx.y, e = threadIdx.x;
    const float *pp = part + ((long)i * H + h) * splits * CROSS_PART;
    float M = -INFINITY;
    for (int s = 0; s < splits; ++s) M = fmaxf(M, pp[s * CROSS_PART]);
    float L = 0.0f, o = 0.0f;
    for (int s = 0; s < splits; ++s) {
        const float m = pp[s * CROSS_PART];
        const float w = m > -INFINITY ? expf(m - M) : 0.0f;
        L += w * pp[s * CROSS_PART + 1];
        o += w * pp[s * CROSS_PART + 2 + e];
    }
    out[(long)i * d + h * 64 + e] = o / L;
}

__global__ __launch_bounds__(256) void kv_reorder_kernel(const int *__restrict__ old_idx, int *__restrict__ new_idx,
                                                         BeamSrc src, int n_past, int n_text_ctx) {
    const int j = blockIdx.x;
    const int s = src.src[j];
    if (s < 0) return;
    for (int p = threadIdx.x; p <= n_past; p += 256)
        new_idx[j * n_text_ctx + p] = p < n_past ? old_idx[s * n_text_ctx + p] : j;
}

template <typename WT, typename KVT, int NB>
void gemv_dispatch_r(const GemvArgs &a, int row0, int KC, int R, hipStream_t st) {
    const size_t lds = (size_t)NB * KC * sizeof(float);
    if (R == 1) {
        dim3 grid(ceil_div(a.N, 4));
        hipLaunchKernelGGL((gemv_kernel<WT, KVT, NB, 1>), grid, dim3(256), lds, st, a, row0, KC);
    } else if (R == 2) {
        dim3 grid(ceil_div(a.N, 8));
        hipLaunchKernelGGL((gemv_kernel<WT, KVT, NB, 2>), grid, dim3(256), lds, st, a, row0, KC);
    } else {
        dim3 grid(ceil_div(a.N, 16));
        hipLaunchKernelGGL((gemv_kernel<WT, KVT, NB, 4>), grid, dim3(256), lds, st, a, row0, KC);
    }
}

template <typename WT, typename KVT>
void gemv_dispatch(const GemvArgs &a, int row0, int nb, int KC, int R, hipStream_t st) {
    switch (nb) {
        case 1: gemv_dispatch_r<WT, KVT, 1>(a, row0, KC, R, st); break;
        case 2: gemv_dispatch_r<WT, KVT, 2>(a, row0, KC, R, st); break;
        case 3: gemv_dispatch_r<WT, KVT, 3>(a, row0, KC, R, st); break;
        case 4: gemv_dispatch_r<WT, KVT, 4>(a, row0, KC, R, st); break;
        case 5: gemv_dispatch_r<WT, KVT, 5>(a, row0, KC, R, st); break;
        case 6: gemv_dispatch_r<WT, KVT, 6>(a, row0, KC, R, st); break;
        case 7: gemv_dispatch_r<WT, KVT, 7>(a, row0, KC, R, st); break;
        default: gemv_dispatch_r<WT, KVT, 8>(a, row0, KC, R, st); break;
    }
}

template <typename KVT>
void cross_dispatch(const float *q, const void *kc, const void *vc, float *partial, int n, int H, int d, int n_ctx,
                    int splits, int per, hipStream_t st) {
    // rows in groups of up to 8 sharing one K/V read; the tail group picks the tightest instantiation
    const int full = n / 8, tail = n % 8;
    if (full > 0) {
        dim3 grid(H, splits, full);
        hipLaunchKernelGGL((cross_attn_kernel<KVT, 8>), grid, dim3(256), 0, st, q, (const KVT *)kc, (const KVT *)vc,
                           partial, full * 8, H, d, n_ctx, per);
    }
    if (tail > 0) {
        const float *q2 = q + (long)full * 8 * d;
        float *p2 = partial + (long)full * 8 * H * splits * CROSS_PART;
        dim3 grid(H, splits, 1);
#define MW_CROSS_CASE(NR)                                                                                          \
    case NR:                                                                                                       \
        hipLaunchKernelGGL((cross_attn_kernel<KVT, NR>), grid, dim3(256), 0, st, q2, (const KVT *)kc, (const KVT *)vc, \
                           p2, tail, H, d, n_ctx, per);                                                            \
        break;
        switch (tail) {
            MW_CROSS_CASE(1)
            MW_CROSS_CASE(2)
            MW_CROSS_CASE(3)
            MW_CROSS_CASE(4)
            MW_CROSS_CASE(5)
            MW_CROSS_CASE(6)
            MW_CROSS_CASE(7)
        }
#undef MW_CROSS_CASE
    }
}

}  // namespace

void launch_embed(const int *d_tokens, const int *d_pos, const void *tok_emb, DType wt, const float *pos_emb, float *x,
                  int n, int d, hipStream_t st) {
    MW_CHECK(n > 0 && d > 0, "embed: empty");
    if (wt == DType::F32)
        hipLaunchKernelGGL(embed_kernel<float>, dim3(n), dim3(256), 0, st, d_tokens, d_pos, (const float *)tok_emb,
                           pos_emb, x, d);
    else
        hipLaunchKernelGGL(embed_kernel<bf16_t>, dim3(n), dim3(256), 0, st, d_tokens, d_pos, (const bf16_t *)tok_emb,
                           pos_emb, x, d);
    MW_HIP(hipGetLastError());
}

void launch_gemv(const GemvArgs &a, DType wt, hipStream_t st) {
    MW_CHECK(a.x && a.W && a.out && a.n > 0 && a.N > 0 && a.K > 0, "gemv: bad arguments");
    const int epc = wt == DType::F32 ? 4 : 8;
    // activation chunk held in LDS: the model width for K = d or 4d, capped at 1280 floats per row
    int KC = a.K;
    if (KC > 1280) {
        KC = 0;
        for (int c = 1280; c >= 64; c -= 64)
            if (a.K % c == 0) {
                KC = c;
                break;
            }
        MW_CHECK(KC > 0, "gemv: K has no chunk size <= 1280 that is a multiple of 64");
    }
    MW_CHECK(KC % (2 * epc) == 0 && a.K % KC == 0, "gemv: K must be a multiple of 16 elements");
    MW_CHECK(((uintptr_t)a.W % 16) == 0 && ((uintptr_t)a.x % 16) == 0 && a.ldx % 4 == 0, "gemv: alignment");
    if (a.ln_w) MW_CHECK(a.K == KC && a.ln_b, "gemv: fused LayerNorm needs K <= 1280");
    if (a.qkv) {
        MW_CHECK(a.N % 3 == 0 && a.kcache && a.vcache && a.d_slot && a.d_pos && a.n_text_ctx > 0, "gemv: qkv epilogue");
        MW_CHECK(a.kv_t == wt, "gemv: KV cache type must equal the weight type");
    }
    // rows per wave: enough workgroups to cover 256 CUs, more rows per wave once N is large
    const int R = a.N >= 16384 ? 4 : (a.N >= 2560 ? 2 : 1);
    for (int row0 = 0; row0 < a.n; row0 += DEC_MAX_ROWS) {
        const int nb = std::min(DEC_MAX_ROWS, a.n - row0);
        if (wt == DType::F32) gemv_dispatch<float, float>(a, row0, nb, KC, R, st);
        else gemv_dispatch<bf16_t, bf16_t>(a, row0, nb, KC, R, st);
    }
    MW_HIP(hipGetLastError());
}

void launch_self_attn(const float *q, const void *kcache, const void *vcache, DType kv_t, const int *d_slot,
                      const int *d_pos, const int *d_kv_idx, float *out, int n, int H, int d, int n_text_ctx,
                      hipStream_t st) {
    MW_CHECK(n > 0 && H > 0 && d == H * 64, "self-attn: head size must be 64");
    MW_CHECK(n_text_ctx <= 448, "self-attn: n_text_ctx > 448");
    dim3 grid(H, n);
    if (kv_t == DType::F32)
        hipLaunchKernelGGL(self_attn_kernel<float>, grid, dim3(64), 0, st, q, (const float *)kcache, (const float *)vcache,
                           d_slot, d_pos, d_kv_idx, out, d, n_text_ctx);
    else
        hipLaunchKernelGGL(self_attn_kernel<bf16_t>, grid, dim3(64), 0, st, q, (const bf16_t *)kcache,
                           (const bf16_t *)vcache, d_slot, d_pos, d_kv_idx, out, d, n_text_ctx);
    MW_HIP(hipGetLastError());
}

void launch_cross_attn(const float *q, const void *kc, const void *vc, DType kv_t, float *partial, float *out, int n,
                       int H, int d, int n_ctx, int splits, hipStream_t st) {
    MW_CHECK(n > 0 && H > 0 && d == H * 64 && n_ctx > 0, "cross-attn: head size must be 64");
    MW_CHECK(splits >= 1 && splits <= CROSS_SPLITS_MAX, "cross-attn: split count");
    const int per = ceil_div(n_ctx, splits);
    MW_CHECK(per <= 256, "cross-attn: more than 256 keys per split");
    if (kv_t == DType::F32) cross_dispatch<float>(q, kc, vc, partial, n, H, d, n_ctx, splits, per, st);
    else cross_dispatch<bf16_t>(q, kc, vc, partial, n, H, d, n_ctx, splits, per, st);
    hipLaunchKernelGGL(cross_combine_kernel, dim3(H, n), dim3(64), 0, st, partial, out, H, d, splits);
    MW_HIP(hipGetLastError());
}

void launch_kv_reorder(const int *old_idx, int *new_idx, BeamSrc src, int n_past, int n_text_ctx, hipStream_t st) {
    MW_CHECK(n_past >= 0 && n_past < n_text_ctx, "kv reorder: position out of range");
    hipLaunchKernelGGL(kv_reorder_kernel, dim3(KV_SLOTS), dim3(256), 0, st, old_idx, new_idx, src, n_past, n_text_ctx);
    MW_HIP(hipGetLastError());
}

}  // namespace mw
