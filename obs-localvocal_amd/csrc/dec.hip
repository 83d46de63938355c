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

// wave64 sum through DPP (VALU cross-lane moves, no LDS round trips): quad xor-1/xor-2, half-row and
// row mirrors give every lane its 16-lane row total; row_bcast15 / row_bcast31 chain the four rows so
// lane 63 holds the wave total, which v_readlane broadcasts as a scalar.
template <int CTRL, int ROW_MASK> __device__ __forceinline__ float dpp_f(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xf, false));
}
__device__ __forceinline__ float wave_sum(float v) {
    v += dpp_f<0xB1, 0xf>(v);   // quad_perm [1,0,3,2]
    v += dpp_f<0x4E, 0xf>(v);   // quad_perm [2,3,0,1]
    v += dpp_f<0x141, 0xf>(v);  // row_half_mirror
    v += dpp_f<0x140, 0xf>(v);  // row_mirror
    v += dpp_f<0x142, 0xa>(v);  // row_bcast15 into rows 1 and 3
    v += dpp_f<0x143, 0xc>(v);  // row_bcast31 into rows 2 and 3
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
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

// TAG (GEMV_* in kernels.hpp) names the projection class: each class is its own kernel symbol, so
// rocprofv3 statistics and the live HIP-event measurement in bench.py address the same launches.  It
// also fixes the K-chunking at compile time: classes with K = d (<= 1280) hold the whole activation row
// set in LDS (one chunk, LayerNorm fusable); GEMV_FC2 (K = 4d) walks chunks of 1024.
//
// Memory-level parallelism is the whole game here (a wave that waits for each 16-byte load before
// issuing the next one runs at ~4 % of HBM): EVERY weight load of the wave's R rows is issued before
// anything else — they do not depend on the activations — and lands in registers while the block
// stages / normalises the activation rows.
template <int TAG, int EPC> struct GemvShape {
    static constexpr int KC_FIXED = TAG == GEMV_FC2 ? 1024 : 0;                // 0: chunk = K (runtime)
    static constexpr int NCH = TAG == GEMV_FC2 ? 5 : 1;                        // max chunks
    static constexpr int NLC = TAG == GEMV_FC2 ? 1024 / (64 * EPC) : (1280 + 64 * EPC - 1) / (64 * EPC);  // loads per lane per chunk
};

template <typename WT, typename KVT, int NB, int R, int TAG>
__global__ __launch_bounds__(256) void gemv_kernel(GemvArgs a, int row0) {
    constexpr int EPC = Epc<WT>::v;
    using S = GemvShape<TAG, EPC>;
    constexpr int NCH = S::NCH, NLC = S::NLC;
    constexpr int G = NCH == 1 ? 2 : 1;  // float4 groups of one activation row chunk per thread (KC/4 <= 256 G)
    extern __shared__ __attribute__((aligned(16))) float xs[];  // [NCH > 1 ? 2 : 1][NB][KC]
    __shared__ float red[4][NB];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int K = a.K, N = a.N;
    const int KC = S::KC_FIXED ? S::KC_FIXED : K;
    const int nch = S::KC_FIXED ? (K + KC - 1) / KC : 1;
    const int nb_valid = min(NB, a.n - row0);
    const int n_base = (blockIdx.x * 4 + wave) * R;
    const int kc4 = KC >> 2;

    // (weight loads are issued below, right after the activation loads: vmcnt retires in order, so the
    //  L2-resident activations must be OLDER than the HBM weight stream for LayerNorm to overlap it)
    // 2. epilogue operands: lane r*NB+b owns output (row b, feature n_base+r); fetch its bias / residual now
    const int my_r = lane / NB, my_b = lane - my_r * NB;
    const bool my_out = lane < R * NB && my_b < nb_valid && n_base + my_r < N;
    float my_bias = 0.0f, my_res = 0.0f;
    if (my_out) {
        if (a.bias) my_bias = a.bias[n_base + my_r];
        if (a.R && !a.qkv) my_res = a.R[(long)(row0 + my_b) * a.ldo + n_base + my_r];
    }
    // 3. activation rows (first chunk) and LayerNorm parameters
    int xrow[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        const int row = row0 + min(b, nb_valid - 1);  // rows beyond the valid count replicate the last; never stored
        xrow[b] = a.d_rows ? a.d_rows[row] : row;
    }
    float4 xr[NB][G];
    auto load_x = [&](int c) {
#pragma unroll
        for (int j = 0; j < G; ++j) {
            const int g = tid + 256 * j, k = c * KC + 4 * g;
            const bool ok = g < kc4 && k < K;
            const int kk = ok ? k : 0;  // clamped address + select instead of an exec-masked branch per load
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                const float4 t = *reinterpret_cast<const float4 *>(a.x + (long)xrow[b] * a.ldx + kk);
                xr[b][j] = ok ? t : make_float4(0, 0, 0, 0);
            }
        }
    };
    auto store_x = [&](float *dst) {
#pragma unroll
        for (int j = 0; j < G; ++j) {
            const int g = tid + 256 * j;
            if (g < kc4) {
                int slot;
                if constexpr (EPC == 8) slot = (g & 1) * (KC >> 1) + ((g >> 1) << 2);
                else slot = 4 * g;
#pragma unroll
                for (int b = 0; b < NB; ++b) *reinterpret_cast<float4 *>(dst + b * KC + slot) = xr[b][j];
            }
        }
    };
    load_x(0);
    float4 lw[G], lb[G];
    const bool do_ln = NCH == 1 && a.ln_w != nullptr;
    if (do_ln) {
#pragma unroll
        for (int j = 0; j < G; ++j) {
            const int g = min(tid + 256 * j, kc4 - 1);
            lw[j] = *reinterpret_cast<const float4 *>(a.ln_w + 4 * g);
            lb[j] = *reinterpret_cast<const float4 *>(a.ln_b + 4 * g);
        }
    }
    // every weight load of this wave, up front: independent of the activations, in flight during LayerNorm
    uint4 wv[NCH][R][NLC];
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const WT *wrow = (const WT *)a.W + (long)min(n_base + r, N - 1) * K;
#pragma unroll
            for (int i = 0; i < NLC; ++i) {
                const int kin = lane * EPC + i * 64 * EPC, k = c * KC + kin;
                const bool ok = c < nch && kin < KC && k < K;
                const uint4 t = ld_stream16(wrow + (ok ? k : 0));
                wv[c][r][i] = ok ? t : make_uint4(0, 0, 0, 0);
            }
        }
    if (do_ln) {
        // fused LayerNorm on the register copy: mean, then centred variance (two passes, like ggml_norm)
        float part[NB];
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            float s = 0.0f;
#pragma unroll
            for (int j = 0; j < G; ++j) s += (xr[b][j].x + xr[b][j].y) + (xr[b][j].z + xr[b][j].w);
            part[b] = wave_sum(s);
        }
        if (lane == 0)
#pragma unroll
            for (int b = 0; b < NB; ++b) red[wave][b] = part[b];
        __syncthreads();
        float mean[NB];
#pragma unroll
        for (int b = 0; b < NB; ++b) mean[b] = ((red[0][b] + red[1][b]) + (red[2][b] + red[3][b])) / (float)K;
        __syncthreads();
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            float s = 0.0f;
#pragma unroll
            for (int j = 0; j < G; ++j) {
                if (tid + 256 * j < kc4) {
                    xr[b][j].x -= mean[b]; xr[b][j].y -= mean[b]; xr[b][j].z -= mean[b]; xr[b][j].w -= mean[b];
                    s += (xr[b][j].x * xr[b][j].x + xr[b][j].y * xr[b][j].y) + (xr[b][j].z * xr[b][j].z + xr[b][j].w * xr[b][j].w);
                }
            }
            part[b] = wave_sum(s);
        }
        if (lane == 0)
#pragma unroll
            for (int b = 0; b < NB; ++b) red[wave][b] = part[b];
        __syncthreads();
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            const float rstd = 1.0f / sqrtf(((red[0][b] + red[1][b]) + (red[2][b] + red[3][b])) / (float)K + 1e-5f);
#pragma unroll
            for (int j = 0; j < G; ++j) {
                xr[b][j].x = xr[b][j].x * rstd * lw[j].x + lb[j].x;
                xr[b][j].y = xr[b][j].y * rstd * lw[j].y + lb[j].y;
                xr[b][j].z = xr[b][j].z * rstd * lw[j].z + lb[j].z;
                xr[b][j].w = xr[b][j].w * rstd * lw[j].w + lb[j].w;
            }
        }
    }
    float acc[R][NB];
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int b = 0; b < NB; ++b) acc[r][b] = 0.0f;

#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        if (c < nch) {
            float *xc = xs + (NCH > 1 ? (c & 1) * NB * KC : 0);
            store_x(xc);
            if (NCH > 1 && c + 1 < nch) load_x(c + 1);  // next chunk's loads fly during this chunk's FMAs
            __syncthreads();  // double-buffered: chunk c-1's readers are past the previous barrier
#pragma unroll
            for (int i = 0; i < NLC; ++i) {
                const int kin = lane * EPC + i * 64 * EPC;
                if (kin < KC) {
                    float wf[R][EPC];
#pragma unroll
                    for (int r = 0; r < R; ++r) unpack(wv[c][r][i], wf[r], WT{});
#pragma unroll
                    for (int b = 0; b < NB; ++b) {
                        float xv[EPC];
                        if constexpr (EPC == 8) {
                            const float4 lo = *reinterpret_cast<const float4 *>(xc + b * KC + ((kin >> 3) << 2));
                            const float4 hi = *reinterpret_cast<const float4 *>(xc + b * KC + (KC >> 1) + ((kin >> 3) << 2));
                            xv[0] = lo.x; xv[1] = lo.y; xv[2] = lo.z; xv[3] = lo.w;
                            xv[4] = hi.x; xv[5] = hi.y; xv[6] = hi.z; xv[7] = hi.w;
                        } else {
                            const float4 lo = *reinterpret_cast<const float4 *>(xc + b * KC + kin);
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
    }
    // reduce over the wave; every lane ends with every sum, lane r*NB+b keeps its own
    float mine = 0.0f;
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            const float v = wave_sum(acc[r][b]);
            if (lane == r * NB + b) mine = v;
        }
    if (my_out) {
        const int n = n_base + my_r, row = row0 + my_b;
        float v = mine + my_bias;
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
            if (a.R) v += my_res;
            a.out[(long)row * a.ldo + n] = v;
        }
    }
}

// ------------------------------------------------------------------------------------------
// self-attention over the indirected KV cache: one wavefront per (row, head), dh = 64
// ------------------------------------------------------------------------------------------
template <typename KVT, typename OT>
__global__ __launch_bounds__(64) void self_attn_kernel(const float *__restrict__ q, const KVT *__restrict__ kc,
                                                       const KVT *__restrict__ vc, const int *__restrict__ slot,
                                                       const int *__restrict__ pos, const int *__restrict__ kv_idx,
                                                       OT *__restrict__ out, int d, int n_text_ctx) {
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
        OT *op = out + (long)i * d + h * 64 + e8 * 8;
#pragma unroll
        for (int e = 0; e < 8; ++e) store_kv(op + e, o[e] * inv);
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

template <typename OT>
__global__ __launch_bounds__(64) void cross_combine_kernel(const float *__restrict__ part, OT *__restrict__ out,
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
    store_kv(out + (long)i * d + h * 64 + e, o / L);
}

__global__ __launch_bounds__(256) void kv_reorder_kernel(const int *__restrict__ old_idx, int *__restrict__ new_idx,
                                                         BeamSrc src, int n_past, int n_text_ctx) {
    const int j = blockIdx.x;
    const int s = src.src[j];
    if (s < 0) return;
    for (int p = threadIdx.x; p <= n_past; p += 256)
        new_idx[j * n_text_ctx + p] = p < n_past ? old_idx[s * n_text_ctx + p] : j;
}

__global__ __launch_bounds__(256) void kv_reorder_dev_kernel(const int *__restrict__ old_idx, int *__restrict__ new_idx,
                                                             const int *__restrict__ args, int n_text_ctx) {
    const int j = blockIdx.x;
    const int s = args[j], n_past = args[KV_SLOTS];
    if (args[KV_SLOTS + 1] == 0 || s < 0) return;
    for (int p = threadIdx.x; p <= n_past; p += 256)
        new_idx[j * n_text_ctx + p] = p < n_past ? old_idx[s * n_text_ctx + p] : j;
}

// First kernel of a decode call: takes the step block straight from pinned host memory.  Blocks 0..KV_SLOTS-1 apply the
// deferred index-table update (kv_reorder_dev_kernel), the next n_rows blocks embed one token each (embed_kernel), the
// last block copies the block into device memory for the kernels behind (positions, slots, logits rows, sampler rows).
// One launch instead of a copy node and two kernels at the head of every beam step (~8 us of a 1.4 ms step).
template <typename WT>
__global__ __launch_bounds__(256) void step_begin_kernel(const int *__restrict__ hblk, int *__restrict__ dblk, int n_ints, int n_text_ctx,
                                                         int args_off, int n_rows, const int *__restrict__ old_idx,
                                                         int *__restrict__ new_idx, const WT *__restrict__ te,
                                                         const float *__restrict__ pe, float *__restrict__ x, int d) {
    const int b = blockIdx.x;
    if (b < KV_SLOTS) {
        const int *args = hblk + args_off;
        const int s = args[b], n_past = args[KV_SLOTS];
        if (args[KV_SLOTS + 1] == 0 || s < 0) return;
        for (int p = threadIdx.x; p <= n_past; p += 256) new_idx[b * n_text_ctx + p] = p < n_past ? old_idx[s * n_text_ctx + p] : b;
    } else if (b < KV_SLOTS + n_rows) {
        const int i = b - KV_SLOTS;
        const long t = hblk[i], p = hblk[n_text_ctx + i];
        for (int e = threadIdx.x; e < d; e += 256) {
            float v;
            if constexpr (sizeof(WT) == 2) v = __uint_as_float((unsigned)te[t * d + e] << 16);
            else v = te[t * d + e];
            x[(long)i * d + e] = v + pe[p * d + e];
        }
    } else {
        for (int k = threadIdx.x; k < n_ints; k += 256) dblk[k] = hblk[k];
    }
}

// rows-per-wave R is a property of the projection class: more output features per wave once N is large
template <int TAG> struct TagR { static constexpr int v = (TAG == GEMV_LOGITS) ? 4 : ((TAG == GEMV_QKV || TAG == GEMV_FC1) ? 2 : 1); };

template <typename WT, typename KVT, int NB, int TAG>
void gemv_launch(const GemvArgs &a, int row0, int KC, hipStream_t st) {
    constexpr int R = TagR<TAG>::v;
    const size_t lds = (size_t)NB * KC * sizeof(float) * (TAG == GEMV_FC2 ? 2 : 1);
    dim3 grid(ceil_div(a.N, 4 * R));
    hipLaunchKernelGGL((gemv_kernel<WT, KVT, NB, R, TAG>), grid, dim3(256), lds, st, a, row0);
}

template <typename WT, typename KVT, int NB>
void gemv_dispatch_tag(const GemvArgs &a, int row0, int KC, hipStream_t st) {
    switch (a.tag) {
        case GEMV_QKV: gemv_launch<WT, KVT, NB, GEMV_QKV>(a, row0, KC, st); break;
        case GEMV_ATTN_OUT: gemv_launch<WT, KVT, NB, GEMV_ATTN_OUT>(a, row0, KC, st); break;
        case GEMV_CROSS_Q: gemv_launch<WT, KVT, NB, GEMV_CROSS_Q>(a, row0, KC, st); break;
        case GEMV_FC1: gemv_launch<WT, KVT, NB, GEMV_FC1>(a, row0, KC, st); break;
        case GEMV_FC2: gemv_launch<WT, KVT, NB, GEMV_FC2>(a, row0, KC, st); break;
        default: gemv_launch<WT, KVT, NB, GEMV_LOGITS>(a, row0, KC, st); break;
    }
}

template <typename WT, typename KVT>
void gemv_dispatch(const GemvArgs &a, int row0, int nb, int KC, hipStream_t st) {
    switch (nb) {
        case 1: gemv_dispatch_tag<WT, KVT, 1>(a, row0, KC, st); break;
        case 2: gemv_dispatch_tag<WT, KVT, 2>(a, row0, KC, st); break;
        case 3: gemv_dispatch_tag<WT, KVT, 3>(a, row0, KC, st); break;
        case 4: gemv_dispatch_tag<WT, KVT, 4>(a, row0, KC, st); break;
        case 5: gemv_dispatch_tag<WT, KVT, 5>(a, row0, KC, st); break;
        case 6: gemv_dispatch_tag<WT, KVT, 6>(a, row0, KC, st); break;
        case 7: gemv_dispatch_tag<WT, KVT, 7>(a, row0, KC, st); break;
        default: gemv_dispatch_tag<WT, KVT, 8>(a, row0, KC, st); break;
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

void launch_step_begin(const int *h_block_dev, int *d_block, int n_ints, int n_text_ctx, int args_off, int n_rows, const int *old_idx,
                       int *new_idx, const void *tok_emb, DType wt, const float *pos_emb, float *x, int d, hipStream_t st) {
    MW_CHECK(h_block_dev && d_block && n_ints > 0 && n_rows > 0 && d > 0, "step begin: empty");
    MW_CHECK(args_off + KV_SLOTS + 2 <= n_ints && n_text_ctx + n_rows <= n_ints, "step begin: block layout");
    const dim3 grid(KV_SLOTS + n_rows + 1), block(256);
    if (wt == DType::F32)
        hipLaunchKernelGGL(step_begin_kernel<float>, grid, block, 0, st, h_block_dev, d_block, n_ints, n_text_ctx, args_off, n_rows, old_idx,
                           new_idx, (const float *)tok_emb, pos_emb, x, d);
    else
        hipLaunchKernelGGL(step_begin_kernel<bf16_t>, grid, block, 0, st, h_block_dev, d_block, n_ints, n_text_ctx, args_off, n_rows, old_idx,
                           new_idx, (const bf16_t *)tok_emb, pos_emb, x, d);
    MW_HIP(hipGetLastError());
}

void launch_gemv(const GemvArgs &a, DType wt, hipStream_t st) {
    MW_CHECK(a.x && a.W && a.out && a.n > 0 && a.N > 0 && a.K > 0, "gemv: bad arguments");
    const int epc = wt == DType::F32 ? 4 : 8;
    // K-chunking is a property of the class (GemvShape): single chunk of K <= 1280, or 1024-chunks for FC2
    const int KC = a.tag == GEMV_FC2 ? 1024 : a.K;
    if (a.tag == GEMV_FC2) MW_CHECK(a.K <= 5 * 1024, "gemv: FC2 needs K <= 5120");
    else MW_CHECK(a.K <= 1280, "gemv: this projection class needs K <= 1280");
    MW_CHECK(a.K % (2 * epc) == 0 && a.K % 64 == 0, "gemv: K must be a multiple of 64");
    MW_CHECK(((uintptr_t)a.W % 16) == 0 && ((uintptr_t)a.x % 16) == 0 && a.ldx % 4 == 0, "gemv: alignment");
    if (a.ln_w) MW_CHECK(a.tag != GEMV_FC2 && a.ln_b, "gemv: fused LayerNorm needs a single-chunk class");
    if (a.qkv) {
        MW_CHECK(a.N % 3 == 0 && a.kcache && a.vcache && a.d_slot && a.d_pos && a.n_text_ctx > 0, "gemv: qkv epilogue");
        MW_CHECK(a.kv_t == wt, "gemv: KV cache type must equal the weight type");
    }
    MW_CHECK(a.tag >= 0 && a.tag < GEMV_N_TAGS, "gemv: bad projection class");
    for (int row0 = 0; row0 < a.n; row0 += DEC_MAX_ROWS) {
        const int nb = std::min(DEC_MAX_ROWS, a.n - row0);
        if (wt == DType::F32) gemv_dispatch<float, float>(a, row0, nb, KC, st);
        else gemv_dispatch<bf16_t, bf16_t>(a, row0, nb, KC, st);
    }
    MW_HIP(hipGetLastError());
}

void launch_self_attn(const float *q, const void *kcache, const void *vcache, DType kv_t, const int *d_slot,
                      const int *d_pos, const int *d_kv_idx, void *out, DType out_t, int n, int H, int d, int n_text_ctx,
                      hipStream_t st) {
    MW_CHECK(n > 0 && H > 0 && d == H * 64, "self-attn: head size must be 64");
    MW_CHECK(n_text_ctx <= 448, "self-attn: n_text_ctx > 448");
    dim3 grid(H, n);
    MW_CHECK(kv_t == out_t || out_t == DType::F32, "self-attn: output type");
    if (kv_t == DType::F32)
        hipLaunchKernelGGL((self_attn_kernel<float, float>), grid, dim3(64), 0, st, q, (const float *)kcache,
                           (const float *)vcache, d_slot, d_pos, d_kv_idx, (float *)out, d, n_text_ctx);
    else if (out_t == DType::F32)
        hipLaunchKernelGGL((self_attn_kernel<bf16_t, float>), grid, dim3(64), 0, st, q, (const bf16_t *)kcache,
                           (const bf16_t *)vcache, d_slot, d_pos, d_kv_idx, (float *)out, d, n_text_ctx);
    else
        hipLaunchKernelGGL((self_attn_kernel<bf16_t, bf16_t>), grid, dim3(64), 0, st, q, (const bf16_t *)kcache,
                           (const bf16_t *)vcache, d_slot, d_pos, d_kv_idx, (bf16_t *)out, d, n_text_ctx);
    MW_HIP(hipGetLastError());
}

void launch_cross_attn(const float *q, const void *kc, const void *vc, DType kv_t, float *partial, void *out, DType out_t,
                       int n, int H, int d, int n_ctx, int splits, hipStream_t st) {
    MW_CHECK(n > 0 && H > 0 && d == H * 64 && n_ctx > 0, "cross-attn: head size must be 64");
    MW_CHECK(splits >= 1 && splits <= CROSS_SPLITS_MAX, "cross-attn: split count");
    const int per = ceil_div(n_ctx, splits);
    MW_CHECK(per <= 256, "cross-attn: more than 256 keys per split");
    if (kv_t == DType::F32) cross_dispatch<float>(q, kc, vc, partial, n, H, d, n_ctx, splits, per, st);
    else cross_dispatch<bf16_t>(q, kc, vc, partial, n, H, d, n_ctx, splits, per, st);
    if (out_t == DType::F32)
        hipLaunchKernelGGL(cross_combine_kernel<float>, dim3(H, n), dim3(64), 0, st, partial, (float *)out, H, d, splits);
    else
        hipLaunchKernelGGL(cross_combine_kernel<bf16_t>, dim3(H, n), dim3(64), 0, st, partial, (bf16_t *)out, H, d, splits);
    MW_HIP(hipGetLastError());
}

void launch_kv_reorder_dev(const int *old_idx, int *new_idx, const int *args, int n_text_ctx, hipStream_t st) {
    hipLaunchKernelGGL(kv_reorder_dev_kernel, dim3(KV_SLOTS), dim3(256), 0, st, old_idx, new_idx, args, n_text_ctx);
    MW_HIP(hipGetLastError());
}

void launch_kv_reorder(const int *old_idx, int *new_idx, BeamSrc src, int n_past, int n_text_ctx, hipStream_t st) {
    MW_CHECK(n_past >= 0 && n_past < n_text_ctx, "kv reorder: position out of range");
    hipLaunchKernelGGL(kv_reorder_kernel, dim3(KV_SLOTS), dim3(256), 0, st, old_idx, new_idx, src, n_past, n_text_ctx);
    MW_HIP(hipGetLastError());
}

}  // namespace mw
