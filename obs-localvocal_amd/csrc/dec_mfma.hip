// dec_mfma.hip — K7: the bf16 decode-step projections on the matrix cores.
//
// Same contract as gemv_kernel in dec.hip (out = R + act(scale * (LN(x) W^T + b)), Q|K|V epilogue into the
// KV cache; replaces the decoder mul_mat nodes of whisper.cpp's whisper_decode_internal, SURVEY.md A.4,
// reached from whisper_full at /root/reference/src/whisper-utils/whisper-processing.cpp:218-219), but the
// handful of activation rows (beams) ride in the M dimension of v_mfma_f32_16x16x32_bf16 instead of being
// multiplied lane by lane.  Still HBM-bound weight streaming — the MFMA is used because it removes the
// per-lane bf16->f32 unpacking, the FMA chains and all cross-lane reductions (the K reduction happens
// inside the instruction), which made the VALU version instruction-issue bound at these tiny shapes:
//   * a workgroup owns 16 output features; its waves split K; a lane's B fragment is 16 contiguous bytes
//     of one weight row, loaded straight from HBM (non-temporal), ALL of a wave's loads issued up front;
//   * LayerNorm classes: the <= 8 activation rows are normalised in registers (two-pass, like ggml_norm),
//     rounded to bf16 and staged once in LDS (padded rows: conflict-free ds_read_b128 A fragments);
//   * direct classes (attention output projection, FC2): the producer already wrote bf16 activations, the
//     A fragment is loaded from global memory — no LDS staging, one barrier in the whole kernel;
//   * K-split partial tiles meet in LDS; 16 x rows threads run the fused epilogue with prefetched
//     bias / residual.
#include "kernels.hpp"

#include <hip/hip_bf16.h>

namespace mw {

namespace {

using bf16x8b = __attribute__((ext_vector_type(8))) __bf16;
using f32x4 = __attribute__((ext_vector_type(4))) float;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;

__device__ __forceinline__ uint4 ld_stream16(const void *p) {
    const u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(p));
    return make_uint4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ bf16_t to_bf16(float v) {
    __hip_bfloat16 h = __float2bfloat16(v);
    return *reinterpret_cast<bf16_t *>(&h);
}
__device__ __forceinline__ float gelu_f(float x) {
    const float c = 0.79788456080286535587989211986876f, a = 0.044715f;
    return 0.5f * x * (1.0f + tanhf(c * x * (1.0f + a * x * x)));
}
template <int CTRL, int ROW_MASK> __device__ __forceinline__ float dpp_f(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xf, false));
}
// wave64 sums of N independent values, step-major so consecutive DPP ops never depend on each other
template <int N> __device__ __forceinline__ void wave_sum_n(float (&v)[N]) {
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] += dpp_f<0xB1, 0xf>(v[i]);   // quad_perm [1,0,3,2]
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] += dpp_f<0x4E, 0xf>(v[i]);   // quad_perm [2,3,0,1]
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] += dpp_f<0x141, 0xf>(v[i]);  // row_half_mirror
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] += dpp_f<0x140, 0xf>(v[i]);  // row_mirror
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] += dpp_f<0x142, 0xa>(v[i]);  // row_bcast15
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] += dpp_f<0x143, 0xc>(v[i]);  // row_bcast31
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v[i]), 63));
}

template <int TAG> struct MShape {
    static constexpr int WAVES = TAG == GEMV_FC2 ? 8 : 4;  // K split
    static constexpr int NS = TAG == GEMV_FC2 ? 20 : 10;   // max 32-deep k-steps per wave (K <= 5120 / 1280)
};

// NB = 0: direct class (bf16 activations from global memory); NB = 1..8: LayerNorm class with NB rows
template <int TAG, int NB>
__global__ __launch_bounds__(MShape<TAG>::WAVES * 64) void gemv_mfma_kernel(GemvArgs a, int row0) {
    constexpr bool DIRECT = NB == 0;
    constexpr int WAVES = MShape<TAG>::WAVES, NS = MShape<TAG>::NS, NT = WAVES * 64;
    constexpr int NBX = DIRECT ? 1 : NB;
    constexpr int G = (320 + NT - 1) / NT;  // float4 groups of one activation row per thread (K <= 1280)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float *red = reinterpret_cast<float *>(smem);                                 // [WAVES][8][16]
    float *stat = red + WAVES * 8 * 16;                                           // [WAVES][8]
    bf16_t *xsb = reinterpret_cast<bf16_t *>(stat + WAVES * 8);                   // [NB][K + 8]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, fq = lane >> 4;
    const int K = a.K, N = a.N;
    const int nb = min(DIRECT ? 8 : NB, a.n - row0);
    const int n0 = blockIdx.x * 16;
    const int kw = K / WAVES, ns = kw >> 5, kw0 = wave * kw;
    const int ldxs = K + 8;

    // epilogue operands: thread t -> (row t>>4, feature n0 + (t&15))
    const int e_row = tid >> 4, e_n = n0 + (tid & 15);
    const bool e_ok = tid < nb * 16 && e_n < N;
    float e_bias = 0.0f, e_res = 0.0f;
    if (e_ok) {
        if (a.bias) e_bias = a.bias[e_n];
        if (a.R && !a.qkv) e_res = a.R[(long)(row0 + e_row) * a.ldo + e_n];
    }

    // activations first (L2-resident, must be older than the weight stream: vmcnt retires in order)
    float4 xr[NBX][G], lw[G], lb[G];
    uint4 av[DIRECT ? NS : 1];
    if constexpr (DIRECT) {
        const bf16_t *xrow = (const bf16_t *)a.x + (long)(row0 + min(fr, nb - 1)) * a.ldx + kw0 + 8 * fq;
#pragma unroll
        for (int s = 0; s < NS; ++s)
            if (s < ns) {
                const uint4 t = *reinterpret_cast<const uint4 *>(xrow + 32 * s);
                av[s] = fr < nb ? t : make_uint4(0, 0, 0, 0);
            }
    } else {
        const int kc4 = K >> 2;
#pragma unroll
        for (int j = 0; j < G; ++j) {
            const int g = tid + NT * j;
            const bool ok = g < kc4;
            const int gg = ok ? g : 0;
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                int row = row0 + min(b, nb - 1);
                if (a.d_rows) row = a.d_rows[row];
                const float4 t = *reinterpret_cast<const float4 *>(a.x + (long)row * a.ldx + 4 * gg);
                xr[b][j] = ok ? t : make_float4(0, 0, 0, 0);
            }
            if (a.ln_w) {
                lw[j] = *reinterpret_cast<const float4 *>(a.ln_w + 4 * gg);
                lb[j] = *reinterpret_cast<const float4 *>(a.ln_b + 4 * gg);
            }
        }
    }
    // every weight load of this wave, up front
    uint4 wv[NS];
    {
        const bf16_t *wrow = (const bf16_t *)a.W + (long)min(n0 + fr, N - 1) * K + kw0 + 8 * fq;
#pragma unroll
        for (int s = 0; s < NS; ++s)
            if (s < ns) wv[s] = ld_stream16(wrow + 32 * s);
    }

    if constexpr (!DIRECT) {
        const int kc4 = K >> 2;
        if (a.ln_w) {
            const float inv_k = 1.0f / (float)K;
            float part[NB];
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                float s = 0.0f;
#pragma unroll
                for (int j = 0; j < G; ++j) s += (xr[b][j].x + xr[b][j].y) + (xr[b][j].z + xr[b][j].w);
                part[b] = s;
            }
            wave_sum_n<NB>(part);
            if (lane == 0)
#pragma unroll
                for (int b = 0; b < NB; ++b) stat[wave * 8 + b] = part[b];
            __syncthreads();
            float mean[NB];
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                float s = 0.0f;
#pragma unroll
                for (int w = 0; w < WAVES; ++w) s += stat[w * 8 + b];
                mean[b] = s * inv_k;
            }
            __syncthreads();
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                float s = 0.0f;
#pragma unroll
                for (int j = 0; j < G; ++j)
                    if (tid + NT * j < kc4) {
                        xr[b][j].x -= mean[b]; xr[b][j].y -= mean[b]; xr[b][j].z -= mean[b]; xr[b][j].w -= mean[b];
                        s += (xr[b][j].x * xr[b][j].x + xr[b][j].y * xr[b][j].y) +
                             (xr[b][j].z * xr[b][j].z + xr[b][j].w * xr[b][j].w);
                    }
                part[b] = s;
            }
            wave_sum_n<NB>(part);
            if (lane == 0)
#pragma unroll
                for (int b = 0; b < NB; ++b) stat[wave * 8 + b] = part[b];
            __syncthreads();
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                float s = 0.0f;
#pragma unroll
                for (int w = 0; w < WAVES; ++w) s += stat[w * 8 + b];
                const float rstd = 1.0f / sqrtf(s * inv_k + 1e-5f);
#pragma unroll
                for (int j = 0; j < G; ++j) {
                    xr[b][j].x = xr[b][j].x * rstd * lw[j].x + lb[j].x;
                    xr[b][j].y = xr[b][j].y * rstd * lw[j].y + lb[j].y;
                    xr[b][j].z = xr[b][j].z * rstd * lw[j].z + lb[j].z;
                    xr[b][j].w = xr[b][j].w * rstd * lw[j].w + lb[j].w;
                }
            }
        }
        // bf16 rows into LDS (row stride K + 8 elements: consecutive rows shift by one 16-byte slot)
#pragma unroll
        for (int j = 0; j < G; ++j) {
            const int g = tid + NT * j;
            if (g < kc4) {
#pragma unroll
                for (int b = 0; b < NB; ++b) {
                    uint2 pk;
                    pk.x = (unsigned)to_bf16(xr[b][j].x) | ((unsigned)to_bf16(xr[b][j].y) << 16);
                    pk.y = (unsigned)to_bf16(xr[b][j].z) | ((unsigned)to_bf16(xr[b][j].w) << 16);
                    *reinterpret_cast<uint2 *>(xsb + b * ldxs + 4 * g) = pk;
                }
            }
        }
        __syncthreads();
    }

    f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int s = 0; s < NS; ++s)
        if (s < ns) {
            uint4 af;
            if constexpr (DIRECT) {
                af = av[s];
            } else {
                af = make_uint4(0, 0, 0, 0);
                if (fr < NB) af = *reinterpret_cast<const uint4 *>(xsb + fr * ldxs + kw0 + 32 * s + 8 * fq);
            }
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8b, af), __builtin_bit_cast(bf16x8b, wv[s]),
                                                          acc, 0, 0, 0);
        }
    // K-split partial tiles meet in LDS: lane holds rows 4*fq + r (only rows < 8 are ever valid)
    if (fq < 2) {
#pragma unroll
        for (int r = 0; r < 4; ++r) red[(wave * 8 + 4 * fq + r) * 16 + fr] = acc[r];
    }
    __syncthreads();
    if (e_ok) {
        float v = e_bias;
#pragma unroll
        for (int w = 0; w < WAVES; ++w) v += red[(w * 8 + e_row) * 16 + (tid & 15)];
        const int row = row0 + e_row, n = e_n;
        if (a.qkv) {
            const int d = N / 3;
            if (n < d) {
                a.out[(long)row * a.ldo + n] = v * a.qk_scale;
            } else {
                const long off = ((long)a.d_slot[row] * a.n_text_ctx + a.d_pos[row]) * d;
                if (n < 2 * d) ((bf16_t *)a.kcache)[off + (n - d)] = to_bf16(v * a.qk_scale);
                else ((bf16_t *)a.vcache)[off + (n - 2 * d)] = to_bf16(v);
            }
        } else {
            v *= a.out_scale;
            if (a.act == ACT_GELU) v = gelu_f(v);
            if (a.R) v += e_res;
            if (a.out_bf16) ((bf16_t *)a.out)[(long)row * a.ldo + n] = to_bf16(v);
            else a.out[(long)row * a.ldo + n] = v;
        }
    }
}

template <int TAG, int NB> void launch_one(const GemvArgs &a, int row0, hipStream_t st) {
    constexpr int WAVES = MShape<TAG>::WAVES;
    size_t lds = (size_t)WAVES * 8 * 16 * 4 + WAVES * 8 * 4;
    if (NB > 0) lds += (size_t)NB * (a.K + 8) * 2;
    hipLaunchKernelGGL((gemv_mfma_kernel<TAG, NB>), dim3(ceil_div(a.N, 16)), dim3(WAVES * 64), lds, st, a, row0);
}

template <int TAG> void launch_ln_class(const GemvArgs &a, int row0, int nb, hipStream_t st) {
    switch (nb) {
        case 1: launch_one<TAG, 1>(a, row0, st); break;
        case 2: launch_one<TAG, 2>(a, row0, st); break;
        case 3: launch_one<TAG, 3>(a, row0, st); break;
        case 4: launch_one<TAG, 4>(a, row0, st); break;
        case 5: launch_one<TAG, 5>(a, row0, st); break;
        case 6: launch_one<TAG, 6>(a, row0, st); break;
        case 7: launch_one<TAG, 7>(a, row0, st); break;
        default: launch_one<TAG, 8>(a, row0, st); break;
    }
}

}  // namespace

void launch_gemv_mfma(const GemvArgs &a, hipStream_t st) {
    MW_CHECK(a.x && a.W && a.out && a.n > 0 && a.N > 0 && a.K > 0, "gemv_mfma: bad arguments");
    MW_CHECK(a.tag >= 0 && a.tag < GEMV_N_TAGS, "gemv_mfma: bad projection class");
    const bool direct = a.tag == GEMV_ATTN_OUT || a.tag == GEMV_FC2;
    const int waves = a.tag == GEMV_FC2 ? 8 : 4;
    MW_CHECK(a.K % (32 * waves) == 0, "gemv_mfma: K must be a multiple of 32 x waves");
    MW_CHECK(a.K / (32 * waves) <= (a.tag == GEMV_FC2 ? 20 : 10), "gemv_mfma: K too large for this class");
    MW_CHECK(((uintptr_t)a.W % 16) == 0 && ((uintptr_t)a.x % 16) == 0, "gemv_mfma: alignment");
    if (direct) {
        MW_CHECK(a.x_bf16 && a.ldx % 8 == 0 && !a.ln_w && !a.d_rows, "gemv_mfma: direct classes take bf16 activations");
    } else {
        MW_CHECK(!a.x_bf16 && a.ldx % 4 == 0 && a.K <= 1280, "gemv_mfma: LayerNorm classes take fp32 rows, K <= 1280");
        if (a.ln_w) MW_CHECK(a.ln_b != nullptr, "gemv_mfma: LayerNorm bias missing");
    }
    if (a.qkv) {
        MW_CHECK(a.N % 3 == 0 && a.kcache && a.vcache && a.d_slot && a.d_pos && a.n_text_ctx > 0, "gemv_mfma: qkv epilogue");
        MW_CHECK(a.kv_t == DType::BF16, "gemv_mfma: KV cache must be bf16");
    }
    for (int row0 = 0; row0 < a.n; row0 += DEC_MAX_ROWS) {
        const int nb = std::min(DEC_MAX_ROWS, a.n - row0);
        switch (a.tag) {
            case GEMV_ATTN_OUT: launch_one<GEMV_ATTN_OUT, 0>(a, row0, st); break;
            case GEMV_FC2: launch_one<GEMV_FC2, 0>(a, row0, st); break;
            case GEMV_QKV: launch_ln_class<GEMV_QKV>(a, row0, nb, st); break;
            case GEMV_CROSS_Q: launch_ln_class<GEMV_CROSS_Q>(a, row0, nb, st); break;
            case GEMV_FC1: launch_ln_class<GEMV_FC1>(a, row0, nb, st); break;
            default: launch_ln_class<GEMV_LOGITS>(a, row0, nb, st); break;
        }
    }
    MW_HIP(hipGetLastError());
}

}  // namespace mw
