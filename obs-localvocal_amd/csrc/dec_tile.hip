// dec_tile.hip — the beam step (<= 5 rows, bf16 weights) as straight-line matrix-core tile kernels for gfx950.
//
// With <= 5 rows every projection of a decode step is a weight-streaming matrix-vector product bound by latency, not
// bandwidth: each kernel here is one short dependent link of the step's chain (8 per decoder layer + the logits
// projection), written so that every global load is unconditional and issued before the first wait, whole cache
// lines are read per instruction, and the rows ride in M of v_mfma_f32_16x16x32_bf16.  Replaces the decoder mul_mat /
// attention graph whisper.cpp runs per token inside whisper_full
// (/root/reference/src/whisper-utils/whisper-processing.cpp:218-219).  DESIGN.md §4 has the rules each
// kernel follows and the measurements behind them.
#include "kernels.hpp"

#include <hip/hip_bf16.h>

namespace mw {

namespace {

constexpr int MT = TILE_THREADS;

template <int CTRL> __device__ __forceinline__ float dpp_add(float v) {
    return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false));
}
// total of the 16-lane DPP row a lane belongs to, in every lane of the row
__device__ __forceinline__ float row16_sum(float v) {
    v = dpp_add<0xB1>(v);   // quad_perm [1,0,3,2]
    v = dpp_add<0x4E>(v);   // quad_perm [2,3,0,1]
    v = dpp_add<0x141>(v);  // row_half_mirror
    v = dpp_add<0x140>(v);  // row_mirror
    return v;
}
// Wave-wide sum / maximum in every lane, on DPP moves only (as ds_bpermute shuffles — what __shfl_xor compiles to — each of
// the six steps was an LDS round trip behind its own s_waitcnt, in the middle of the attention kernels' critical paths).
// Rows of 16 first, then row_bcast15 / row_bcast31 carry the row totals up to lane 63, whose value is broadcast as a scalar.
// The sum adds the same pairs in the same order as the shuffle form did ((r0 + r1) + (r2 + r3)): same bits.
__device__ __forceinline__ float wave_sum_all(float v) {
    v = row16_sum(v);
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x142, 0xa, 0xf, false));  // row_bcast15
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x143, 0xc, 0xf, false));  // row_bcast31
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
template <int CTRL, int ROWS> __device__ __forceinline__ float dpp_max(float v) {  // (lanes outside ROWS keep their value)
    return fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), CTRL, ROWS, 0xf, false)));
}
__device__ __forceinline__ float wave_max_all(float v) {
    v = dpp_max<0xB1, 0xf>(v);   // quad_perm [1,0,3,2]
    v = dpp_max<0x4E, 0xf>(v);   // quad_perm [2,3,0,1]
    v = dpp_max<0x141, 0xf>(v);  // row_half_mirror
    v = dpp_max<0x140, 0xf>(v);  // row_mirror
    v = dpp_max<0x142, 0xa>(v);  // row_bcast15
    v = dpp_max<0x143, 0xc>(v);  // row_bcast31
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
// v + v[lane ^ 8] (DPP row rotate by 8), v + v[lane ^ 16], v + v[lane ^ 32] (gfx950's row / half-wave swaps: both copies of v
// go in, one comes back with the partner's value in every lane): the partners and the order of each addition are those
// of the __shfl_xor form, without its LDS round trips
__device__ __forceinline__ float xor8_add(float v) { return dpp_add<0x128>(v); }
__device__ __forceinline__ float xor16_add(float v) {
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float xor32_add(float v) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
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
__device__ __forceinline__ void unpack8(const uint4 &u, float (&f)[8]) {
    f[0] = bf16_lo(u.x); f[1] = bf16_hi(u.x); f[2] = bf16_lo(u.y); f[3] = bf16_hi(u.y);
    f[4] = bf16_lo(u.z); f[5] = bf16_hi(u.z); f[6] = bf16_lo(u.w); f[7] = bf16_hi(u.w);
}
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
__device__ __forceinline__ uint4 ld_stream16(const void *p) {  // weights: read once per step
    const u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(p));
    return make_uint4(v.x, v.y, v.z, v.w);
}
// agent-scope (write-through, sc1) stores: what other workgroups read after the next barrier
__device__ __forceinline__ void st_agent(float *p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_agent(bf16_t *p, bf16_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// workgroup barrier that orders LDS traffic only: unlike __syncthreads() it does not wait for outstanding
// global loads (vmcnt), so weight prefetches stay in flight across the intra-phase synchronisations
__device__ __forceinline__ void lds_sync() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }


struct Proj {
    const bf16_t *W;
    const float *bias;
    int N, K;
    // fp8 mode (template parameter W8): the same matrix as OCP e4m3 bytes, one power-of-two scale per output row
    const unsigned char *W8;
    const float *ws;
};

// 16 e4m3 bytes -> 16 bf16 (exact: 4 significant bits): v_cvt_pk_f32_fp8 per byte pair, the high halves of the two floats
// packed by one v_perm_b32.  The row scale is applied to the finished sum (a power of two: the result is the same to the bit
// as multiplying bf16 weights that already carry it).
__device__ __forceinline__ unsigned fp8x2_to_bf16x2(unsigned u, bool hi) {
    const auto f = hi ? __builtin_amdgcn_cvt_pk_f32_fp8((int)u, true) : __builtin_amdgcn_cvt_pk_f32_fp8((int)u, false);
    return __builtin_amdgcn_perm(__float_as_uint(f[1]), __float_as_uint(f[0]), 0x07060302u);
}
__device__ __forceinline__ void fp8x16_to_bf16(const uint4 &u, uint4 &lo, uint4 &hi) {
    lo = make_uint4(fp8x2_to_bf16x2(u.x, false), fp8x2_to_bf16x2(u.x, true), fp8x2_to_bf16x2(u.y, false), fp8x2_to_bf16x2(u.y, true));
    hi = make_uint4(fp8x2_to_bf16x2(u.z, false), fp8x2_to_bf16x2(u.z, true), fp8x2_to_bf16x2(u.w, false), fp8x2_to_bf16x2(u.w, true));
}

using bf16x8b = __attribute__((ext_vector_type(8))) __bf16;
using f32x4 = __attribute__((ext_vector_type(4))) float;
using u32x4v = __attribute__((ext_vector_type(4))) unsigned;
// 16 bytes through a pointer that came out of a step descriptor in memory: hipcc cannot know such a pointer's address space
// and emits flat_load (aperture check, counted on lgkmcnt as well as vmcnt — entangled with the LDS traffic behind it);
// the caches it points to are device memory, so the load is stated as a global one
// (A/B on one box, headline workload: 158.5 / 158.6 -> 159.1 / 159.3 audio-s/s, profiles/r03_ab_runs.txt)
__device__ __forceinline__ u32x4v ldg16(const void *p) { return *(const __attribute__((address_space(1))) u32x4v *)p; }
__device__ __forceinline__ int ldg_i32(const int *p) { return *(const __attribute__((address_space(1))) int *)p; }


enum { EPI_QKV = 0, EPI_RES = 1, EPI_SCALE = 2, EPI_GELU_BF16 = 3, EPI_GELU_F32 = 4 };

struct Epi {
    // EPI_QKV
    float *q;
    bf16_t *kc, *vc;
    const int *slot, *pos;
    int TC, d;
    float qk_scale;
    // EPI_RES
    const float *res;
    float *xout;
    // EPI_SCALE
    float *fout;
    float scale;
    // EPI_GELU_BF16
    bf16_t *hout;
    int ldo;
    // batched step (EPI_QKV): when set, row i writes K / V through rows[i] (its context's cache + kv_layer_off elements)
    const StepRow *rows;
    long kv_layer_off;
};

__device__ __forceinline__ float wave_total(float v) {  // DPP reduction, result broadcast as a scalar
    v = row16_sum(v);
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x142, 0xa, 0xf, false));  // row_bcast15
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x143, 0xc, 0xf, false));  // row_bcast31
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

constexpr int LN_MAX_F4 = 5;  // float4 per lane of a LayerNorm row: K <= 1280


// ---- the LayerNorm-fused projection as a kernel of its own: one 16-feature tile per workgroup, K <= 1280 ----
// Same arithmetic as proj_phase<LN>, written so that the compiler emits straight-line code: every global load
// is unconditional (clamped address, value selected afterwards) and issued before the first wait.  A load
// under a predicate becomes a branch with `s_waitcnt vmcnt(0)` at its join; in proj_phase that put the weight
// loads first and then five serialised (row, gamma, beta) round trips in front of the LayerNorm — 3 to 4 us
// of a 5 us kernel (wall_clock64 stamps through the `stamp` argument below).
template <int EPI>
__device__ __forceinline__ void proj_finish(float v, int erow, int n_e, const Proj &p, const Epi &e, float res_e) {
    if constexpr (EPI == EPI_QKV) {
        const int d = e.d;
        if (n_e < d) {
            st_agent(e.q + (long)erow * d + n_e, v * e.qk_scale);
        } else {
            const long off = ((long)e.slot[erow] * e.TC + e.pos[erow]) * d;
            if (n_e < 2 * d) st_agent(e.kc + off + (n_e - d), to_bf16(v * e.qk_scale));
            else st_agent(e.vc + off + (n_e - 2 * d), to_bf16(v));
        }
    } else if constexpr (EPI == EPI_RES) {
        st_agent(e.xout + (long)erow * p.N + n_e, v + res_e);
    } else if constexpr (EPI == EPI_SCALE) {
        st_agent(e.fout + (long)erow * p.N + n_e, v * e.scale);
    } else if constexpr (EPI == EPI_GELU_BF16) {
        st_agent(e.hout + (long)erow * e.ldo + n_e, to_bf16(gelu_f(v)));
    } else {
        st_agent(e.fout + (long)erow * e.ldo + n_e, gelu_f(v));
    }
}

// The tile's weights (16 rows of K bf16, contiguous in memory) are loaded fully coalesced — wave w takes rows
// 2w and 2w+1 as consecutive 1 KB pieces — and re-read from LDS in the MFMA B-fragment layout.  Loading the
// fragments straight from global memory touches 64 cache lines per instruction (16 bytes of each) instead of
// 8 and was 1.3-1.4 us slower per launch (tools/microbench/mb_latency, "B-fragment pattern").
template <int NB, int EPI, int FR, bool DESC, bool W8>
__device__ __forceinline__ void proj_ln_tile(const Proj &p, const float *xin, const float *ln_w, const float *ln_b, int n, const Epi &e,
                                             bf16_t *xs, bf16_t *wt, float *red, unsigned long long *stamp) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, fq = lane >> 4;
    const int K = p.K, nf4 = K >> 2, ksteps = K >> 5, ldxs = K + 8, tile = blockIdx.x;
    constexpr int RW = (NB + 7) / 8;  // rows a wave normalises: wave w takes rows w, w + 8 (batched step: up to 16 rows)
    // ---- every global load of the kernel, in the order their values are needed ----
    // (a wave without a row issues the same loads with every lane on one address: a single cache-line request each,
    // still straight-line code — a branch around the loads made the compiler wait for them inside the branch)
    float4 v[RW][LN_MAX_F4], gw[LN_MAX_F4], gb[LN_MAX_F4];
#pragma unroll
    for (int rw = 0; rw < RW; ++rw) {
        const bool has_row = wave + 8 * rw < n;
        const float *xr = xin + (long)(has_row ? wave + 8 * rw : 0) * K;
#pragma unroll
        for (int i = 0; i < LN_MAX_F4; ++i) v[rw][i] = *reinterpret_cast<const float4 *>(xr + (has_row ? 4 * min(lane + 64 * i, nf4 - 1) : 0));
    }
#pragma unroll
    for (int i = 0; i < LN_MAX_F4; ++i) {
        const int c = wave < n ? 4 * min(lane + 64 * i, nf4 - 1) : 0;
        gw[i] = *reinterpret_cast<const float4 *>(ln_w + c);
        gb[i] = *reinterpret_cast<const float4 *>(ln_b + c);
    }
    // weight rows per wave, load instructions per wave (K <= 1280: 160 sixteen-byte pieces per bf16 row, 80 per e4m3 row)
    constexpr int RPW = FR / 8, WJ = (RPW * (W8 ? 80 : 160) + 63) / 64;
    constexpr int TL = FR > 16 ? FR / 16 : 1;                // 16-feature MFMA tiles per workgroup
    constexpr int MTL = (NB + 15) / 16;                      // 16-row MFMA tiles (2 for the 32-row step: the weights are read once)
    static_assert(TL * MTL <= 2, "the finishing pass covers two 16 x 16 tiles (32 rows: FR <= 16)");
    uint4 w[WJ];
    const int wpieces = RPW * (K >> (W8 ? 4 : 3));  // 16-byte pieces in the wave's row(s)
    if constexpr (W8) {
        const unsigned char *wbase = p.W8 + (long)(tile * FR + RPW * wave) * K;
#pragma unroll
        for (int j = 0; j < WJ; ++j) w[j] = ld_stream16(wbase + 16 * min(lane + 64 * j, wpieces - 1));
    } else {
        const bf16_t *wbase = p.W + (long)(tile * FR + RPW * wave) * K;
#pragma unroll
        for (int j = 0; j < WJ; ++j) w[j] = ld_stream16(wbase + 8 * min(lane + 64 * j, wpieces - 1));
    }
    // finisher: tile (feature tile when TL = 2, row tile when MTL = 2), lane, register
    const int et = tid >> 8, el = (tid & 255) >> 2, er = tid & 3;
    const int etl = TL > 1 ? et : 0, emt = MTL > 1 ? et : 0, erow = 16 * emt + 4 * (el >> 4) + er;
    const int n_e = tile * FR + min(16 * etl + (el & 15), FR - 1);
    const float bias_e = p.bias ? p.bias[n_e] : 0.0f;
    float ws_e = 1.0f;
    if constexpr (W8) ws_e = p.ws[n_e];
    long kv_off = 0;
    bf16_t *kc_e = e.kc, *vc_e = e.vc;
    if constexpr (EPI == EPI_QKV) {
        const int rr = erow < n ? erow : 0;
        if constexpr (DESC) {  // batched step: the row's own context (compile-time choice: a run-time branch around these
                               // loads would end in a wait for every load issued before it, the weights included)
            const StepRow *R = e.rows + rr;
            kv_off = ((long)R->slot * e.TC + R->pos) * e.d + e.kv_layer_off;
            kc_e = R->kself;
            vc_e = R->vself;
        } else {
            kv_off = ((long)e.slot[rr] * e.TC + e.pos[rr]) * e.d;
        }
    }
    // ---- LayerNorm of the wave's row(s) (wave reductions only) -> bf16 in LDS ----
#pragma unroll
    for (int rw = 0; rw < RW; ++rw) {
        const int row = wave + 8 * rw;
        float s = 0.0f;
#pragma unroll
        for (int i = 0; i < LN_MAX_F4; ++i) s += lane + 64 * i < nf4 ? (v[rw][i].x + v[rw][i].y) + (v[rw][i].z + v[rw][i].w) : 0.0f;
        const float mean = wave_total(s) * (1.0f / (float)K);
        float q = 0.0f;
#pragma unroll
        for (int i = 0; i < LN_MAX_F4; ++i) {
            const float a0 = v[rw][i].x - mean, a1 = v[rw][i].y - mean, a2 = v[rw][i].z - mean, a3 = v[rw][i].w - mean;
            q += lane + 64 * i < nf4 ? (a0 * a0 + a1 * a1) + (a2 * a2 + a3 * a3) : 0.0f;
        }
        const float rstd = rsqrtf(wave_total(q) * (1.0f / (float)K) + 1e-5f);
#pragma unroll
        for (int i = 0; i < LN_MAX_F4; ++i) {
            const int c = lane + 64 * i;
            const float y0 = (v[rw][i].x - mean) * rstd * gw[i].x + gb[i].x, y1 = (v[rw][i].y - mean) * rstd * gw[i].y + gb[i].y;
            const float y2 = (v[rw][i].z - mean) * rstd * gw[i].z + gb[i].z, y3 = (v[rw][i].w - mean) * rstd * gw[i].w + gb[i].w;
            uint2 pk;
            pk.x = (unsigned)to_bf16(y0) | ((unsigned)to_bf16(y1) << 16);
            pk.y = (unsigned)to_bf16(y2) | ((unsigned)to_bf16(y3) << 16);
            if (c < nf4 && row < n) *reinterpret_cast<uint2 *>(xs + row * ldxs + 4 * c) = pk;
        }
    }
    if (stamp && tid == 0) stamp[0] = wall_clock64();
    // ---- the wave's two weight rows -> LDS ----
#pragma unroll
    for (int j = 0; j < WJ; ++j) {
        const int pc = lane + 64 * j, half = K >> (W8 ? 4 : 3);  // pieces per row
        const int r = (pc >= half ? 1 : 0) + (pc >= 2 * half ? 1 : 0) + (pc >= 3 * half ? 1 : 0);
        if constexpr (W8) {
            uint4 lo, hi;
            fp8x16_to_bf16(w[j], lo, hi);
            if (pc < wpieces) {
                *reinterpret_cast<uint4 *>(wt + (RPW * wave + r) * ldxs + 16 * (pc - r * half)) = lo;
                *reinterpret_cast<uint4 *>(wt + (RPW * wave + r) * ldxs + 16 * (pc - r * half) + 8) = hi;
            }
        } else {
            if (pc < wpieces) *reinterpret_cast<uint4 *>(wt + (RPW * wave + r) * ldxs + 8 * (pc - r * half)) = w[j];
        }
    }
    lds_sync();
    if (stamp && tid == 0) stamp[1] = wall_clock64();
    // ---- matrix-core products ----
    {
        const bf16_t *arow[MTL];  // rows >= n: any valid row, result ignored
#pragma unroll
        for (int mt = 0; mt < MTL; ++mt) arow[mt] = xs + (16 * mt + fr < n ? 16 * mt + fr : 0) * ldxs + 8 * fq;
        const bf16_t *brow = wt + fr * ldxs + 8 * fq;
        f32x4 acc[MTL][TL];
#pragma unroll
        for (int mt = 0; mt < MTL; ++mt)
#pragma unroll
            for (int tl = 0; tl < TL; ++tl) acc[mt][tl] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            const int step = wave + 8 * j, sc = min(step, ksteps - 1);
            uint4 af[MTL];
#pragma unroll
            for (int mt = 0; mt < MTL; ++mt) af[mt] = *reinterpret_cast<const uint4 *>(arow[mt] + 32 * sc);
#pragma unroll
            for (int tl = 0; tl < TL; ++tl) {
                uint4 wv = *reinterpret_cast<const uint4 *>(brow + 16 * tl * ldxs + 32 * sc);
                if (step >= ksteps) wv = make_uint4(0, 0, 0, 0);
#pragma unroll
                for (int mt = 0; mt < MTL; ++mt)
                    acc[mt][tl] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8b, af[mt]), __builtin_bit_cast(bf16x8b, wv),
                                                                          acc[mt][tl], 0, 0, 0);
            }
        }
#pragma unroll
        for (int mt = 0; mt < MTL; ++mt)
#pragma unroll
            for (int tl = 0; tl < TL; ++tl) *reinterpret_cast<f32x4 *>(red + (((mt * TL + tl) * 8 + wave) * 64 + lane) * 4) = acc[mt][tl];
    }
    lds_sync();
    if (stamp && tid == 0) stamp[2] = wall_clock64();
    // ---- finish: fixed-order sum of the 8 partial tiles, bias, epilogue ----
    if (et < TL * MTL && erow < n && 16 * etl + (el & 15) < FR) {  // tile columns beyond FR come from unwritten LDS rows: ignored
        float r = 0.0f;
#pragma unroll
        for (int k = 0; k < 8; ++k) r += red[((et * 8 + k) * 64 + el) * 4 + er];
        if constexpr (W8) r *= ws_e;
        r += bias_e;
        if constexpr (EPI == EPI_QKV) {
            const int d = e.d;
            if (n_e < d) st_agent(e.q + (long)erow * d + n_e, r * e.qk_scale);
            else if (n_e < 2 * d) st_agent(kc_e + kv_off + (n_e - d), to_bf16(r * e.qk_scale));
            else st_agent(vc_e + kv_off + (n_e - 2 * d), to_bf16(r));
        } else {
            proj_finish<EPI>(r, erow, n_e, p, e, 0.0f);
        }
    }
    if (stamp && tid == 0) stamp[3] = wall_clock64();
}

// ---- residual-class projection (attention outputs, fc2) as a kernel of its own: x rows are bf16, no LayerNorm ----
// One 16-feature tile per workgroup; wave w owns the k-range [w*K/8, (w+1)*K/8) of all 16 weight rows and of the
// x rows, in CH chunks of SPC MFMA steps.  Every load is issued up front, coalesced (a row segment of a chunk is
// 64*SPC contiguous bytes); each chunk is transposed into the fragment layout through a wave-private LDS tile
// (LDS operations of one wave execute in order: no barrier), so the only workgroup barrier is the one before
// the 8 partial tiles are summed.
template <int NB, int SPC, int CH, int FR, bool W8>
__device__ __forceinline__ void proj_res_tile(const Proj &p, const bf16_t *xin, int n, const Epi &e, bf16_t *lds_t, float *red,
                                              unsigned long long *stamp) {
    constexpr int KC = 32 * SPC, LDT = KC + 8, PPR = 4 * SPC;  // k per chunk, LDS row stride, 16-byte pieces per row segment
    // the wave's x tile: its whole k-range at once for the beam step (<= 8 rows), chunk by chunk for the batched step
    // (16 rows x the whole range of fc2 would not fit the LDS)
    constexpr bool XCH = NB > 8;
    constexpr int XST = XCH ? CH : 1;                            // x stages
    constexpr int XLD = (XCH ? KC : CH * KC) + 8, XPR = XCH ? PPR : PPR * CH;  // row stride, pieces per row of a stage
    constexpr int XI = (NB * XPR + 63) / 64;                     // x load instructions per stage
    constexpr int PPW = W8 ? 2 * SPC : PPR;                      // 16-byte pieces per WEIGHT row segment (e4m3: 16 k per piece)
    constexpr int WI = (FR * PPW + 63) / 64;                     // weight load instructions per chunk (FR feature rows per workgroup)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, fq = lane >> 4;
    const int K = p.K, tile = blockIdx.x, kw = wave * CH * KC;
    u32x4v xv[XST][XI];  // first-class vectors: as HIP uint4 structs this array was not split into registers and went through
                         // LDS scratch, whose indexing reads the workgroup size from the dispatch packet in host memory (2-6 us)
    uint4 w[CH][WI];
#pragma unroll
    for (int s2 = 0; s2 < XST; ++s2)
#pragma unroll
        for (int i = 0; i < XI; ++i) {
            const int pc = min(lane + 64 * i, NB * XPR - 1);
            xv[s2][i] = *reinterpret_cast<const u32x4v *>(xin + (long)min(pc / XPR, n - 1) * K + kw + s2 * KC + 8 * (pc % XPR));
        }
    constexpr int MTL = (NB + 15) / 16;  // 16-row MFMA tiles (32-row step: two; threads 256.. finish the second)
    const int el = (tid & 255) >> 2, er = tid & 3, emt = MTL > 1 ? tid >> 8 : 0, erow = 16 * emt + 4 * (el >> 4) + er;
    const int n_e = tile * FR + min(el & 15, FR - 1);
    const float res_e = e.res[(long)(erow < n ? erow : 0) * p.N + n_e];
    const float bias_v = (p.bias ? p.bias : e.res)[n_e];  // unconditional load: a branch here would split the load block
    const float bias_e = p.bias ? bias_v : 0.0f;
    float ws_e = 1.0f;
    if constexpr (W8) ws_e = p.ws[n_e];
#pragma unroll
    for (int c = 0; c < CH; ++c)
#pragma unroll
        for (int j = 0; j < WI; ++j) {
            const int pc = min(lane + 64 * j, FR * PPW - 1);
            if constexpr (W8) w[c][j] = ld_stream16(p.W8 + (long)(tile * FR + pc / PPW) * K + kw + c * KC + 16 * (pc % PPW));
            else w[c][j] = ld_stream16(p.W + (long)(tile * FR + pc / PPW) * K + kw + c * KC + 8 * (pc % PPW));
        }
    __builtin_amdgcn_sched_barrier(0);  // every load above is issued before anything below (no sinking next to the use)
    bf16_t *wt = lds_t + wave * (16 * LDT + NB * XLD), *xt = wt + 16 * LDT;
    if constexpr (!XCH) {
#pragma unroll
        for (int i = 0; i < XI; ++i) {
            const int pc = min(lane + 64 * i, NB * XPR - 1);  // surplus lanes rewrite the last piece (same data): no predicate
            *reinterpret_cast<u32x4v *>(xt + (pc / XPR) * XLD + 8 * (pc % XPR)) = xv[0][i];
        }
    }
    f32x4 acc[MTL];
#pragma unroll
    for (int mt = 0; mt < MTL; ++mt) acc[mt] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    if (stamp && tid == 0) stamp[0] = wall_clock64();
#pragma unroll
    for (int c = 0; c < CH; ++c) {
        if constexpr (XCH) {  // this chunk's x rows (the wave's LDS operations execute in order: no barrier)
#pragma unroll
            for (int i = 0; i < XI; ++i) {
                const int pc = min(lane + 64 * i, NB * XPR - 1);
                *reinterpret_cast<u32x4v *>(xt + (pc / XPR) * XLD + 8 * (pc % XPR)) = xv[c][i];
            }
        }
#pragma unroll
        for (int j = 0; j < WI; ++j) {
            const int pc = min(lane + 64 * j, FR * PPW - 1);  // surplus lanes rewrite the last piece
            if constexpr (W8) {
                uint4 lo, hi;
                fp8x16_to_bf16(w[c][j], lo, hi);
                *reinterpret_cast<uint4 *>(wt + (pc / PPW) * LDT + 16 * (pc % PPW)) = lo;
                *reinterpret_cast<uint4 *>(wt + (pc / PPW) * LDT + 16 * (pc % PPW) + 8) = hi;
            } else {
                *reinterpret_cast<uint4 *>(wt + (pc / PPW) * LDT + 8 * (pc % PPW)) = w[c][j];
            }
        }
#pragma unroll
        for (int s2 = 0; s2 < SPC; ++s2) {
            const uint4 bfr = *reinterpret_cast<const uint4 *>(wt + fr * LDT + 32 * s2 + 8 * fq);
#pragma unroll
            for (int mt = 0; mt < MTL; ++mt) {
                const int xr = 16 * mt + fr < n ? 16 * mt + fr : 0;
                const uint4 af = *reinterpret_cast<const uint4 *>(xt + xr * XLD + (XCH ? 0 : c * KC) + 32 * s2 + 8 * fq);
                acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8b, af), __builtin_bit_cast(bf16x8b, bfr), acc[mt], 0,
                                                                  0, 0);
            }
        }
    }
    if (stamp && tid == 0) stamp[1] = wall_clock64();
#pragma unroll
    for (int mt = 0; mt < MTL; ++mt) *reinterpret_cast<f32x4 *>(red + ((mt * 8 + wave) * 64 + lane) * 4) = acc[mt];
    lds_sync();
    if (stamp && tid == 0) stamp[2] = wall_clock64();
    if (tid < 256 * MTL && erow < n && (el & 15) < FR) {  // tile columns beyond FR come from unwritten LDS rows: ignored
        float r = 0.0f;
#pragma unroll
        for (int k = 0; k < 8; ++k) r += red[((emt * 8 + k) * 64 + el) * 4 + er];
        if constexpr (W8) r *= ws_e;
        st_agent(e.xout + (long)erow * p.N + n_e, r + bias_e + res_e);
    }
    if (stamp && tid == 0) stamp[3] = wall_clock64();
}

// ---- self-attention over the indirected KV cache: one workgroup per (row, head), thread = key.  Every
// global load of the phase (index table, K row, the thread's V chunks) is issued before anything waits. ----
__device__ __forceinline__ void st_out(bf16_t *p, float v) { st_agent(p, to_bf16(v)); }
__device__ __forceinline__ void st_out(float *p, float v) { st_agent(p, v); }


// ---- the logits projection (final LayerNorm + tied token embedding, N = n_vocab) as tile kernel ----
// 128 vocabulary rows per workgroup: the LayerNorm of the <= 5 sampled rows is done once (wave w = row w) and
// shared through LDS; wave w then owns the 16-feature tile w over the whole K and needs nobody else's sums (no
// cross-wave reduction).  Its weights stream in chunks of 128 k (16 rows x 256 contiguous bytes per chunk, 4 loads
// per lane) through a ring of LRING chunks in registers and a wave-private LDS tile (fragment layout).
constexpr int LKC = 128, LLD = LKC + 8, LRING = 3;

template <int NB, int NCH, bool W8>
__device__ __forceinline__ void logits_tile(const bf16_t *W, const unsigned char *W8p, const float *wsc, const float *xin, const int *d_rows, const float *ln_w, const float *ln_b,
                                            float *out, long ldo, int n, int N, bf16_t *xs, bf16_t *wtile) {
    constexpr int K = NCH * LKC, nf4 = K / 4, ldxs = K + 8;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, fq = lane >> 4;
    const int f0 = (blockIdx.x * 8 + wave) * 16;  // first vocabulary row of the wave's tile
    // ---- loads: the wave's LayerNorm row(s) (or one address), then the first LRING weight chunks ----
    constexpr int RW = (NB + 7) / 8;  // wave w normalises rows w, w + 8
    float4 v[RW][LN_MAX_F4], gw[LN_MAX_F4], gb[LN_MAX_F4];
#pragma unroll
    for (int rw = 0; rw < RW; ++rw) {
        const bool has_row = wave + 8 * rw < n;
        const float *xr = xin + (long)d_rows[has_row ? wave + 8 * rw : 0] * K;
#pragma unroll
        for (int i = 0; i < LN_MAX_F4; ++i) v[rw][i] = *reinterpret_cast<const float4 *>(xr + (has_row ? 4 * min(lane + 64 * i, nf4 - 1) : 0));
    }
#pragma unroll
    for (int i = 0; i < LN_MAX_F4; ++i) {
        const int c = wave < n ? 4 * min(lane + 64 * i, nf4 - 1) : 0;
        gw[i] = *reinterpret_cast<const float4 *>(ln_w + c);
        gb[i] = *reinterpret_cast<const float4 *>(ln_b + c);
    }
    // weight piece (chunk c, load j): row = (lane + 64 j) >> 4, 16-byte column = (lane + 64 j) & 15
    // (e4m3: 8 pieces of 16 k per row and chunk: row = (lane + 64 j) >> 3, column = (lane + 64 j) & 7, two loads per lane)
    constexpr int WL = W8 ? 2 : 4;
    const unsigned char *wl[WL];
#pragma unroll
    for (int j = 0; j < WL; ++j) {
        const int pc = lane + 64 * j;
        // rows beyond the vocabulary: clamped, never stored
        if constexpr (W8) wl[j] = W8p + (long)min(f0 + (pc >> 3), N - 1) * K + 16 * (pc & 7);
        else wl[j] = reinterpret_cast<const unsigned char *>(W + (long)min(f0 + (pc >> 4), N - 1) * K + 8 * (pc & 15));
    }
    constexpr int CB = W8 ? LKC : 2 * LKC;  // bytes of a chunk in a weight row
    float ws_e = 1.0f;
    if constexpr (W8) ws_e = wsc[min(f0 + fr, N - 1)];
    uint4 w[NCH][WL];
#pragma unroll
    for (int c = 0; c < LRING && c < NCH; ++c)
#pragma unroll
        for (int j = 0; j < WL; ++j) w[c][j] = ld_stream16(wl[j] + c * CB);
    __builtin_amdgcn_sched_barrier(0);
    // ---- LayerNorm of the wave's row(s) -> bf16 in LDS ----
#pragma unroll
    for (int rw = 0; rw < RW; ++rw) {
        const int row = wave + 8 * rw;
        float s = 0.0f;
#pragma unroll
        for (int i = 0; i < LN_MAX_F4; ++i) s += lane + 64 * i < nf4 ? (v[rw][i].x + v[rw][i].y) + (v[rw][i].z + v[rw][i].w) : 0.0f;
        const float mean = wave_total(s) * (1.0f / (float)K);
        float q = 0.0f;
#pragma unroll
        for (int i = 0; i < LN_MAX_F4; ++i) {
            const float a0 = v[rw][i].x - mean, a1 = v[rw][i].y - mean, a2 = v[rw][i].z - mean, a3 = v[rw][i].w - mean;
            q += lane + 64 * i < nf4 ? (a0 * a0 + a1 * a1) + (a2 * a2 + a3 * a3) : 0.0f;
        }
        const float rstd = rsqrtf(wave_total(q) * (1.0f / (float)K) + 1e-5f);
#pragma unroll
        for (int i = 0; i < LN_MAX_F4; ++i) {
            const int c = lane + 64 * i;
            const float y0 = (v[rw][i].x - mean) * rstd * gw[i].x + gb[i].x, y1 = (v[rw][i].y - mean) * rstd * gw[i].y + gb[i].y;
            const float y2 = (v[rw][i].z - mean) * rstd * gw[i].z + gb[i].z, y3 = (v[rw][i].w - mean) * rstd * gw[i].w + gb[i].w;
            uint2 pk;
            pk.x = (unsigned)to_bf16(y0) | ((unsigned)to_bf16(y1) << 16);
            pk.y = (unsigned)to_bf16(y2) | ((unsigned)to_bf16(y3) << 16);
            if (c < nf4 && row < n) *reinterpret_cast<uint2 *>(xs + row * ldxs + 4 * c) = pk;
        }
    }
    lds_sync();
    // ---- the wave's tile over all of K: chunk c from the ring -> private LDS tile -> 4 MFMA steps ----
    bf16_t *wt = wtile + wave * 16 * LLD;
    constexpr int MTL = (NB + 15) / 16;  // 16-row MFMA tiles
    const bf16_t *arow[MTL];
    f32x4 acc[MTL];
#pragma unroll
    for (int mt = 0; mt < MTL; ++mt) {
        arow[mt] = xs + (16 * mt + fr < n ? 16 * mt + fr : 0) * ldxs + 8 * fq;
        acc[mt] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    }
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
#pragma unroll
        for (int j = 0; j < WL; ++j) {
            const int pc = lane + 64 * j;
            if constexpr (W8) {
                uint4 lo, hi;
                fp8x16_to_bf16(w[c][j], lo, hi);
                *reinterpret_cast<uint4 *>(wt + (pc >> 3) * LLD + 16 * (pc & 7)) = lo;
                *reinterpret_cast<uint4 *>(wt + (pc >> 3) * LLD + 16 * (pc & 7) + 8) = hi;
            } else {
                *reinterpret_cast<uint4 *>(wt + (pc >> 4) * LLD + 8 * (pc & 15)) = w[c][j];
            }
        }
        if (c + LRING < NCH) {
#pragma unroll
            for (int j = 0; j < WL; ++j) w[c + LRING][j] = ld_stream16(wl[j] + (c + LRING) * CB);
        }
#pragma unroll
        for (int s2 = 0; s2 < 4; ++s2) {
            const uint4 bfr = *reinterpret_cast<const uint4 *>(wt + fr * LLD + 32 * s2 + 8 * fq);
#pragma unroll
            for (int mt = 0; mt < MTL; ++mt) {
                const uint4 af = *reinterpret_cast<const uint4 *>(arow[mt] + c * LKC + 32 * s2);
                acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8b, af), __builtin_bit_cast(bf16x8b, bfr), acc[mt], 0,
                                                                  0, 0);
            }
        }
    }
    // lane holds rows 16*mt + 4*fq + r of vocabulary entry f0 + fr
#pragma unroll
    for (int mt = 0; mt < MTL; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = 16 * mt + 4 * fq + r;
            if (row < n && f0 + fr < N) out[(long)row * ldo + f0 + fr] = W8 ? acc[mt][r] * ws_e : acc[mt][r];
        }
}

// ---- self-attention as a kernel of its own, straight-line: three dependent load levels and no more ----
//   1. pos / slot of the row (scalar) and the thread's 8 query values        2. the index-table entries of its keys
//   3. its 16-byte K and V chunks (thread (jj, e8): keys jj, jj+64, ...; 8 lanes cover a 128-byte head row, so an
//      instruction reads whole cache lines)
// Loads beyond the row's length are clamped to its last position (one line for the whole wave) and masked in the
// arithmetic; nothing is predicated, so the compiler keeps each level's loads in one batch.  Probabilities stay
// in registers (the thread that scored a key also weights its V chunk).
// Batched step: `rows` set -> the row's cache, index table, slot and position come from its descriptor (one more scalar
// load level than the single-context form; the rows of a step may belong to different contexts).
template <typename OT, bool DESC>
__device__ __forceinline__ void self_attn_lean(const float *q, const bf16_t *kc, const bf16_t *vc, const int *slot, const int *pos,
                                               const int *kv_idx, const StepRow *rows, const int *phys, long kv_layer_off, OT *out, int H,
                                               int d, int TC, float *red, float *sm) {
    constexpr int PVI = TILE_TC_MAX / 64;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, jj = tid >> 3, e8 = tid & 7;
    const int r = blockIdx.x / H, h = blockIdx.x - r * H;
    int np, sl;
    if constexpr (DESC) {
        const StepRow *R = rows + r;
        np = R->pos + 1;
        sl = R->slot;
        kc = R->kself + kv_layer_off;
        vc = R->vself + kv_layer_off;
        kv_idx = R->kv_idx;
    } else {
        np = pos[r] + 1;
        sl = slot[r];
    }
    const float4 qa = *reinterpret_cast<const float4 *>(q + (long)r * d + h * 64 + e8 * 8);
    const float4 qb = *reinterpret_cast<const float4 *>(q + (long)r * d + h * 64 + e8 * 8 + 4);
    int iv[PVI];
    if constexpr (DESC) {
        // the row's physical slot per position, laid out by the step's first kernel (step_begin_batch_kernel: the index-table
        // entries this loop used to fetch behind the row's descriptor, padded with the last valid one): its address depends
        // on nothing loaded here, so descriptor, q and these travel together and the K / V loads are the second round trip,
        // not the third
#pragma unroll
        for (int i = 0; i < PVI; ++i) iv[i] = phys[(long)r * TILE_TC_MAX + jj + 64 * i];
    } else {
        const int *idx = kv_idx + (long)sl * TC;
#pragma unroll
        for (int i = 0; i < PVI; ++i) iv[i] = idx[min(jj + 64 * i, np - 1)];
    }
    uint4 kq[PVI], vq[PVI];
#pragma unroll
    for (int i = 0; i < PVI; ++i) {
        const long off = ((long)iv[i] * TC + min(jj + 64 * i, np - 1)) * d + h * 64 + e8 * 8;
        const u32x4v k4 = ldg16(kc + off), v4 = ldg16(vc + off);
        kq[i] = make_uint4(k4[0], k4[1], k4[2], k4[3]);
        vq[i] = make_uint4(v4[0], v4[1], v4[2], v4[3]);
    }
    __builtin_amdgcn_sched_barrier(0);  // the V loads stay in this batch (the scheduler would sink them next to their use)
    float sc[PVI], mx = -INFINITY;
#pragma unroll
    for (int i = 0; i < PVI; ++i) {
        float f[8];
        unpack8(kq[i], f);
        float dot = qa.x * f[0];
        dot = fmaf(qa.y, f[1], dot); dot = fmaf(qa.z, f[2], dot); dot = fmaf(qa.w, f[3], dot);
        dot = fmaf(qb.x, f[4], dot); dot = fmaf(qb.y, f[5], dot); dot = fmaf(qb.z, f[6], dot); dot = fmaf(qb.w, f[7], dot);
        dot = dpp_add<0xB1>(dot);   // the 8 lanes of a key: lane ^ 1, lane ^ 2 (quad permutes), then the other quad of the
        dot = dpp_add<0x4E>(dot);   // eight (row_half_mirror: every lane of a quad holds the quad's sum by then)
        dot = dpp_add<0x141>(dot);
        sc[i] = jj + 64 * i < np ? dot : -INFINITY;
        mx = fmaxf(mx, sc[i]);
    }
    mx = wave_max_all(mx);
    if (lane == 0) sm[wave] = mx;
    lds_sync();
    mx = sm[0];
#pragma unroll
    for (int w = 1; w < MT / 64; ++w) mx = fmaxf(mx, sm[w]);
    float o[8] = {0, 0, 0, 0, 0, 0, 0, 0}, psum = 0.0f;
#pragma unroll
    for (int i = 0; i < PVI; ++i) {
        const float w = jj + 64 * i < np ? expf(sc[i] - mx) : 0.0f;
        psum += w;
        float f[8];
        unpack8(vq[i], f);
#pragma unroll
        for (int q8 = 0; q8 < 8; ++q8) o[q8] = fmaf(w, f[q8], o[q8]);
    }
    const float ws = wave_sum_all(e8 == 0 ? psum : 0.0f);  // the 8 lanes of a key hold the same weight: count it once
#pragma unroll
    for (int q8 = 0; q8 < 8; ++q8) {
        o[q8] = xor32_add(xor16_add(xor8_add(o[q8])));
    }
    if (lane == 0) sm[8 + wave] = ws;
    if ((lane >> 3) == 0) {
#pragma unroll
        for (int q8 = 0; q8 < 8; ++q8) red[wave * 64 + e8 * 8 + q8] = o[q8];
    }
    lds_sync();
    if (tid < 64) {
        float sum = 0.0f, v = 0.0f;
#pragma unroll
        for (int w = 0; w < MT / 64; ++w) {
            sum += sm[8 + w];
            v += red[w * 64 + tid];
        }
        st_out(out + (long)r * d + h * 64 + tid, v / sum);
    }
}

// One partial softmax record = (max, sum, 64 partial outputs) of one (row, head, split): 66 floats, padded to 96 = three
// 128-byte lines of its own (the buffer comes from hipMalloc: 256-byte aligned).  No line is ever shared by the records of two
// workgroups, so a combiner that runs early on some XCD cannot pull a line into that XCD's L2 that another split has yet to
// write (at 66 floats the last split of head h and split 0 of head h + 1 shared a line).
constexpr int CPART = CROSS_PART_STRIDE;
static_assert(CPART >= 66 && CPART % 32 == 0, "a partial record starts on a 128-byte line");

// tail of a cross-attention work item: count this split of head h; the last one to arrive combines the partial
// softmax records of all splits (fixed split order) into the output rows.  Rows row0 .. row0 + n - 1 of the step
// (row0 = 0, cnt = h for a single context; the request's rows and counter in a batched step).
//
// Hand-off (MI355X_MICROARCH.md, "Valid forms"; cdna_hip_programming.md Guideline 16, R1 + consumer recipe):
//   producers  — every record byte leaves by an sc1 (write-through) store (st_agent), EVERY storing wave drains its stores
//                (s_waitcnt vmcnt(0) as inline asm: the compiler cannot drop it), the workgroup barrier, then ONE lane's
//                returning agent-scope fetch_add on the head's counter;
//   consumer   — the workgroup whose add returned splits - 1: that lane runs ONE agent-scope acquire (buffer_inv sc1: this
//                CU's L1 may hold lines of the record buffer from an earlier workgroup), waits vmcnt(0) so the barrier
//                below is not passed before the invalidate has completed, raises the LDS flag; after the barrier every
//                wave reads the records with plain loads.
// Measured against the other valid form for one workgroup per CU (every record load an sc1 load, no invalidate; the
// kernel's LDS raised above 80 KB to guarantee the residency): 157.0 / 157.5 vs 157.2 / 157.1 audio-s/s, 9.49 / 9.68 vs
// 9.47 / 9.66 us per launch, interleaved runs on one box — no difference, so the form that holds for any residency is
// the only one kept.  The counter is reset by the combiner for the next launch AFTER its loads were issued (nobody adds
// to it again in this launch; the kernel boundary publishes the store).
template <int NB, typename OT>
__device__ __forceinline__ void cross_count_and_combine(const float *part, OT *out, unsigned *head_cnt, unsigned *flag, int row0, int n,
                                                        int H, int d, int h, int cnt, int splits) {
    const int tid = threadIdx.x;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's record stores are acknowledged
    __syncthreads();                                   // ... and so are every other wave's, before the split is counted
    if (tid == 0) {
        const unsigned old = __hip_atomic_fetch_add(&head_cnt[cnt], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned last = old == (unsigned)splits - 1;
        if (last) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        *flag = last;
    }
    lds_sync();
    if (*flag) {  // every split of head h is in memory: combine in fixed split order
        // one batch of unconditional loads per thread (row r, dim e): the splits' (max, sum) pairs — the same
        // address for the 64 threads of a row — and the thread's own partial outputs; surplus slots repeat the
        // last split and get weight zero
#pragma unroll
        for (int it = 0; it < (NB * 64 + MT - 1) / MT; ++it) {
            const int i0 = tid + MT * it;
            const int i = min(i0, NB * 64 - 1), r = min(i >> 6, n - 1), e = i & 63;
            const float *pp = part + ((long)(row0 + r) * H + h) * splits * CPART;
            float2 mlv[CROSS_SPLITS_MAX];
            float ov[CROSS_SPLITS_MAX];
#pragma unroll
            for (int sidx = 0; sidx < CROSS_SPLITS_MAX; ++sidx) {
                const float *ps = pp + min(sidx, splits - 1) * CPART;
                mlv[sidx] = *reinterpret_cast<const float2 *>(ps);
                ov[sidx] = ps[2 + e];
            }
            if (it == 0 && tid == 0) __hip_atomic_store(&head_cnt[cnt], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // next launch
            float M = -INFINITY;
#pragma unroll
            for (int sidx = 0; sidx < CROSS_SPLITS_MAX; ++sidx) M = fmaxf(M, sidx < splits ? mlv[sidx].x : -INFINITY);
            float Lsum = 0.0f, acc = 0.0f;
#pragma unroll
            for (int sidx = 0; sidx < CROSS_SPLITS_MAX; ++sidx) {
                const float mm = sidx < splits ? mlv[sidx].x : -INFINITY;
                const float w = mm > -INFINITY ? expf(mm - M) : 0.0f;
                Lsum += w * mlv[sidx].y;
                acc += w * ov[sidx];
            }
            if (i0 < NB * 64 && (i0 >> 6) < n) st_out(out + (long)(row0 + r) * d + h * 64 + e, acc / Lsum);
        }
    }
}

// ---- cross-attention on the matrix cores as a straight-line kernel (one (head, split) of one context per workgroup) ----
// scores  S[row][key] = q[row] . K[key]:  A = q rows as bf16 (LDS), B = K rows, one 16-key tile per wave, 2 MFMA steps;
// P . V   O[row][dim] = sum_key P[row][key] V[key][dim]:  B needs 8 consecutive KEYS of one dim per lane, so the V
//         tile is transposed on its way into LDS; wave w takes dims 16*(w&3).. and key half w>>2, 2 MFMAs each.
// Every global load is unconditional (keys beyond the split are clamped to its last key and masked in the scores) and
// issued before anything else; K and V are loaded 8 lanes per 128-byte head row (whole cache lines per instruction)
// and re-read from LDS in the B-fragment layout; no LDS zero-fill (operand rows beyond the real rows only feed
// result rows nobody reads).  Batched step: `groups` set -> blockIdx.x also selects the request, whose rows
// row0 .. row0 + nrows - 1 attend to ITS context's K / V.
constexpr int XK = 128;      // keys per split
constexpr int XLDP = XK + 8; // LDS row stride (elements) of the [*][key] tiles: conflict-free 16-byte reads
// Row stride of the transposed V tile: 65 dwords.  The tile is written by 2-byte scatter stores — a lane holds 8 dims of one
// key, the 8 lanes of a key are 8 dims apart — and with 68 dwords per row (XLDP) the 64 lanes of a store instruction fell
// into 8 banks; an odd dword stride spreads them over 32.  Rows are then only 4-byte aligned: the B fragments are read as
// four dwords instead of one 16-byte piece.  Interleaved A/B on one box (profiles/r03_ab_runs.txt, `vt_*`): 9.38 / 9.58 ->
// 8.85 / 8.79 us per launch inside graph replay, headline 159.9 -> 160.4 audio-s/s.  Dword stores after a lane swap (keys
// 2m and 2m + 1 are 8 lanes apart: half the store instructions, none sub-dword) measured 9.09 / 8.98: not kept.
constexpr int XVLD = XK + 2;
static_assert(XVLD % 2 == 0 && XVLD >= XK && XVLD <= XLDP, "transposed V rows: dword-aligned, inside their region");
constexpr int XKD = 64 + 8;  // LDS row stride of the [key][dim] K tile

template <int NB, typename OT, bool DESC>
__device__ __forceinline__ void cross_attn_lean(const float *q, const bf16_t *kc, const bf16_t *vc, const StepGroup *groups, long k_off,
                                                long v_off, float *part, OT *out, unsigned *head_cnt, int n, int H, int d, int n_ctx,
                                                int splits, float *lds) {
    // LDS, under 40 KB for NB <= 5 so that FOUR workgroups share a CU (53 KB and three before): the q and probability tiles
    // hold their NB real rows only — the MFMA A operand reads 16, and rows beyond NB run on into the tile behind (K, V: any
    // bytes will do, they feed result rows nobody reads) — and the PV partial sums reuse the K tile, dead since the scores.
    bf16_t *qb = reinterpret_cast<bf16_t *>(lds);  // [NB][72]    q rows (+ 16 - NB rows of whatever follows)
    bf16_t *kt = qb + NB * 72;                     // [XK][XKD]   K rows
    bf16_t *pb = kt + XK * XKD;                    // [NB][XLDP]  probabilities (+ 16 - NB rows of whatever follows)
    bf16_t *vt = pb + NB * XLDP;                   // [64][XVLD]  V transposed: [dim][key]
    float *sc = reinterpret_cast<float *>(vt + 64 * XVLD);  // [NB][XK] scores
    float *ml = sc + NB * XK;                      // [NB][2]
    unsigned *flag = reinterpret_cast<unsigned *>(ml + NB * 2);
    float *red = reinterpret_cast<float *>(kt);    // [8][64][4]  PV partial sums, written behind the softmax barrier
    static_assert(8 * 64 * 4 * 4 <= XK * XKD * 2, "the partial sums fit the K tile");
    static_assert((16 - NB) * 72 <= XK * XKD && (16 - NB) * XLDP <= 64 * XVLD, "operand rows beyond NB stay inside the next tile");
    static_assert((NB * 72 * 2) % 16 == 0 && (NB * XLDP * 2) % 16 == 0 && (64 * XVLD * 2) % 16 == 0, "16-byte aligned tiles");
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, fq = lane >> 4;
    const int hs = H * splits, g = blockIdx.x / hs, rem = blockIdx.x - g * hs;
    const int h = rem / splits, sp = rem - h * splits;
    int row0 = 0;
    if constexpr (DESC) {
        const StepGroup *G = groups + g;
        row0 = G->row0;
        n = G->nrows;
        n_ctx = G->n_ctx;
        kc = G->cross + k_off;
        vc = G->cross + v_off;
    }
    const int per = (n_ctx + splits - 1) / splits;
    const int j0 = sp * per, nk = min(per, n_ctx - j0);  // may be <= 0 for a trailing split
    const int last = max(nk - 1, 0) + min(j0, n_ctx - 1);  // a valid key for clamped loads
    // ---- loads ----
    constexpr int QI = (NB * 64 + MT - 1) / MT;
    float qv[QI];
#pragma unroll
    for (int it = 0; it < QI; ++it) {
        const int qi = min(tid + MT * it, NB * 64 - 1);
        qv[it] = q[(long)(row0 + min(qi >> 6, n - 1)) * d + h * 64 + (qi & 63)];
    }
    const int kk = tid >> 3, e8 = tid & 7;
    u32x4v kq[2], vq[2];  // first-class vectors: component access stays in registers
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int key = kk + 64 * i;
        const long off = (long)(key < nk ? j0 + key : last) * d + h * 64 + e8 * 8;
        kq[i] = ldg16(kc + off);
        vq[i] = ldg16(vc + off);
    }
    __builtin_amdgcn_sched_barrier(0);
    // ---- stage q, K and V^T ----
#pragma unroll
    for (int it = 0; it < QI; ++it) {
        const int qi = min(tid + MT * it, NB * 64 - 1);
        qb[(qi >> 6) * 72 + (qi & 63)] = to_bf16(qv[it]);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int key = kk + 64 * i;
        *reinterpret_cast<u32x4v *>(kt + key * XKD + e8 * 8) = kq[i];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            vt[(e8 * 8 + 2 * c) * XVLD + key] = (bf16_t)(vq[i][c] & 0xffffu);
            vt[(e8 * 8 + 2 * c + 1) * XVLD + key] = (bf16_t)(vq[i][c] >> 16);
        }
    }
    lds_sync();
    // ---- scores: wave w -> keys 16w .. 16w+15 ----
    {
        const int kkey = 16 * wave + fr;
        f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            const uint4 af = *reinterpret_cast<const uint4 *>(qb + fr * 72 + 32 * s2 + 8 * fq);
            const uint4 bfr = *reinterpret_cast<const uint4 *>(kt + kkey * XKD + 32 * s2 + 8 * fq);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8b, af), __builtin_bit_cast(bf16x8b, bfr), acc, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = 4 * fq + r;
            if (row < NB) sc[row * XK + kkey] = kkey < nk ? acc[r] : -INFINITY;
        }
    }
    lds_sync();
    // ---- per-row softmax over the split (wave w: rows w, w + 8) ----
#pragma unroll
    for (int rw = 0; rw < (NB + 7) / 8; ++rw) {
        const int r = wave + 8 * rw;
        if (r < NB) {
            const float v0 = sc[r * XK + lane], v1 = sc[r * XK + 64 + lane];
            const float mx = wave_max_all(fmaxf(v0, v1));
            const float e0 = lane < nk ? expf(v0 - mx) : 0.0f, e1 = (64 + lane) < nk ? expf(v1 - mx) : 0.0f;
            const float sum = wave_sum_all(e0 + e1);
            pb[r * XLDP + lane] = to_bf16(e0);
            pb[r * XLDP + 64 + lane] = to_bf16(e1);
            if (lane == 0) {
                ml[r * 2] = mx;
                ml[r * 2 + 1] = sum;
            }
        }
    }
    lds_sync();
    // ---- P . V ----
    {
        const int nt = wave & 3, kh = wave >> 2;
        f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            const int ks = 2 * kh + s2;
            const uint4 af = *reinterpret_cast<const uint4 *>(pb + fr * XLDP + 32 * ks + 8 * fq);
            const unsigned *bp = reinterpret_cast<const unsigned *>(vt + (16 * nt + fr) * XVLD + 32 * ks + 8 * fq);
            const uint4 bfr = XVLD % 8 == 0 ? *reinterpret_cast<const uint4 *>(bp) : make_uint4(bp[0], bp[1], bp[2], bp[3]);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8b, af), __builtin_bit_cast(bf16x8b, bfr), acc, 0, 0, 0);
        }
        *reinterpret_cast<f32x4 *>(red + (wave * 64 + lane) * 4) = acc;
    }
    lds_sync();
#pragma unroll
    for (int it = 0; it < QI; ++it) {
        const int i = tid + MT * it, r = i >> 6, e = i & 63;
        if (i < NB * 64 && r < n) {
            const int nt = e >> 4, src_lane = (r >> 2) * 16 + (e & 15), rr = r & 3;
            const float v = red[(nt * 64 + src_lane) * 4 + rr] + red[((4 + nt) * 64 + src_lane) * 4 + rr];
            float *pp = part + (((long)(row0 + r) * H + h) * splits + sp) * CPART;
            st_agent(pp + 2 + e, v);
            if (e == 0) {
                st_agent(pp, ml[r * 2]);
                st_agent(pp + 1, ml[r * 2 + 1]);
            }
        }
    }
    cross_count_and_combine<NB, OT>(part, out, head_cnt, flag, row0, n, H, d, h, g * H + h, splits);
}

template <int NB, int NCH, bool W8>
__global__ __launch_bounds__(MT) void logits_tile_kernel(const bf16_t *W, const unsigned char *W8p, const float *wsc, const float *xin,
                                                         const int *d_rows, const float *ln_w, const float *ln_b, float *out, long ldo,
                                                         int n, int N) {
    __shared__ __attribute__((aligned(16))) bf16_t xs[NB * (NCH * LKC + 8)];
    __shared__ __attribute__((aligned(16))) bf16_t wtile[8 * 16 * LLD];
    logits_tile<NB, NCH, W8>(W, W8p, wsc, xin, d_rows, ln_w, ln_b, out, ldo, n, N, xs, wtile);
}

template <typename OT, bool DESC>
__global__ __launch_bounds__(MT) void self_attn_lean_kernel(const float *q, const bf16_t *kc, const bf16_t *vc, const int *slot,
                                                                       const int *pos, const int *kv_idx, const StepRow *rows,
                                                                       const int *phys, long kv_layer_off, OT *out, int H, int d, int TC) {
    __shared__ __attribute__((aligned(16))) float red[8 * 64];
    __shared__ float sm[16];
    self_attn_lean<OT, DESC>(q, kc, vc, slot, pos, kv_idx, rows, phys, kv_layer_off, out, H, d, TC, red, sm);
}


template <int NB, int EPI, int FR, bool DESC, bool W8>
__global__ __launch_bounds__(MT) void proj_ln_tile_kernel(Proj p, const float *xin, const float *ln_w, const float *ln_b, int n,
                                                                     Epi e) {
    __shared__ __attribute__((aligned(16))) bf16_t xs[NB * (256 * LN_MAX_F4 + 8)];
    __shared__ __attribute__((aligned(16))) bf16_t wt[FR * (256 * LN_MAX_F4 + 8)];
    __shared__ __attribute__((aligned(16))) float red[(FR > 16 ? FR / 16 : 1) * ((NB + 15) / 16) * 8 * 64 * 4];
    proj_ln_tile<NB, EPI, FR, DESC, W8>(p, xin, ln_w, ln_b, n, e, xs, wt, red, nullptr);
}

template <int NB, int SPC, int CH, int FR, bool W8>
__global__ __launch_bounds__(MT) void proj_res_tile_kernel(Proj p, const bf16_t *xin, int n, Epi e) {
    __shared__ __attribute__((aligned(16))) bf16_t lds_t[8 * (16 * (32 * SPC + 8) + NB * ((NB > 8 ? 1 : CH) * 32 * SPC + 8))];
    __shared__ __attribute__((aligned(16))) float red[((NB + 15) / 16) * 8 * 64 * 4];
    proj_res_tile<NB, SPC, CH, FR, W8>(p, xin, n, e, lds_t, red, nullptr);
}

template <int NB, typename OT, bool DESC>
__global__ __launch_bounds__(MT) void cross_attn_lean_kernel(const float *q, const bf16_t *kc, const bf16_t *vc, const StepGroup *groups,
                                                              long k_off, long v_off, float *part, OT *out, unsigned *head_cnt, int n, int H,
                                                              int d, int n_ctx, int splits) {
    __shared__ __attribute__((aligned(16))) float lds[(NB * 72 + XK * XKD + NB * XLDP + 64 * XVLD) / 2 + NB * XK + NB * 2 + 16];
    cross_attn_lean<NB, OT, DESC>(q, kc, vc, groups, k_off, v_off, part, out, head_cnt, n, H, d, n_ctx, splits, lds);
}

// Measurement only (MI355_WHISPER_CROSS_PREFETCH): reads exactly the K / V bytes that workgroup blockIdx.x of
// cross_attn_lean_kernel<.., DESC = true> will read, with the same grid, and drops them — run on a parallel branch of the
// step's graph, ahead of the cross-attention of the same layer.
__global__ __launch_bounds__(MT) void cross_touch_kernel(const StepGroup *groups, long k_off, long v_off, int H, int d, int splits, int *sink) {
    const int tid = threadIdx.x;
    const int hs = H * splits, g = blockIdx.x / hs, rem = blockIdx.x - g * hs;
    const int h = rem / splits, sp = rem - h * splits;
    const StepGroup *G = groups + g;
    const int n_ctx = G->n_ctx;
    const bf16_t *kc = G->cross + k_off, *vc = G->cross + v_off;
    const int per = (n_ctx + splits - 1) / splits;
    const int j0 = sp * per, nk = min(per, n_ctx - j0);
    const int last = max(nk - 1, 0) + min(j0, n_ctx - 1);
    const int kk = tid >> 3, e8 = tid & 7;
    unsigned acc = 0;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int key = kk + 64 * i;
        const long off = (long)(key < nk ? j0 + key : last) * d + h * 64 + e8 * 8;
        const u32x4v a = *reinterpret_cast<const u32x4v *>(kc + off), b = *reinterpret_cast<const u32x4v *>(vc + off);
        acc ^= a[0] ^ a[1] ^ a[2] ^ a[3] ^ b[0] ^ b[1] ^ b[2] ^ b[3];
    }
    if (acc == 0x9e3779b9u && sink) sink[0] = 1;  // (keeps the loads; practically never taken)
}

// ---- head of a batched step: the StepBlock straight from pinned host memory ----
// Blocks [0, n_groups * KV_SLOTS): the requests' deferred beam reshuffles of their index tables; the next n_rows blocks
// embed one token each; the last block copies the StepBlock into device memory for the kernels behind.
constexpr int STEP_BEGIN_MAX_D = 1280;  // text width the step head's straight-line embedding covers (launch_step_begin_batch checks)
__global__ __launch_bounds__(256) void step_begin_batch_kernel(const StepBlock *__restrict__ hblk, StepBlock *__restrict__ dblk, int n_rows,
                                                               int n_groups, int n_text_ctx, const bf16_t *__restrict__ te,
                                                               const float *__restrict__ pe, float *__restrict__ x, int d,
                                                               int *__restrict__ phys) {
    const int b = blockIdx.x;
    if (b < n_groups * KV_SLOTS) {
        const int g = b / KV_SLOTS, j = b - g * KV_SLOTS;
        const StepGroup *G = &hblk->groups[g];
        const int s = G->src[j], n_past = G->reorder_n_past;
        if (G->reorder_flag == 0 || s < 0) return;
        const int *old_idx = G->idx_old;
        int *new_idx = G->idx_new;
        for (int p = threadIdx.x; p <= n_past; p += 256) new_idx[j * n_text_ctx + p] = p < n_past ? ldg_i32(old_idx + s * n_text_ctx + p) : j;
    } else if (b < n_groups * KV_SLOTS + n_rows) {
        // Straight-line on purpose: as loops over a run-time d the embedding and the table reads waited for one load per
        // iteration (five and two dependent round trips behind the two host reads).  Here the row's descriptor is one host
        // round trip, the embedding loads and the read of the row's group (host again) then travel together, the table reads
        // follow, and the stores come last.
        const int i = b - n_groups * KV_SLOTS;
        const StepRow *Rw = &hblk->rows[i];
        const long t = Rw->token, p = Rw->pos;
        const int sl = Rw->slot, grp = Rw->group;
        const int *kvi = Rw->kv_idx;
        constexpr int EI = STEP_BEGIN_MAX_D / 256;
        unsigned tv[EI];
        float pv[EI];
#pragma unroll
        for (int j = 0; j < EI; ++j) {
            const int e = min((int)threadIdx.x + 256 * j, d - 1);
            tv[j] = te[t * d + e];
            pv[j] = pe[p * d + e];
        }
        // the row's physical slot per position for the self-attention kernels of every layer (self_attn_lean): what they
        // would read from the row's index table AFTER this kernel's reshuffle blocks — taken from the reshuffle's source
        // where the reshuffle writes (those blocks run beside this one), positions beyond the row's own padded with its last
        const StepGroup *G = &hblk->groups[grp];
        const int np = (int)p + 1, src = G->src[sl], rnp = G->reorder_n_past;
        const bool re = G->reorder_flag != 0 && src >= 0;
        const int *cur = kvi + (long)sl * n_text_ctx, *old = G->idx_old + (long)(src < 0 ? 0 : src) * n_text_ctx;
        constexpr int PI = (TILE_TC_MAX + 255) / 256;
        int pvv[PI];
#pragma unroll
        for (int j = 0; j < PI; ++j) {
            const int pp = min((int)threadIdx.x + 256 * j, np - 1);
            const int v = ldg_i32(re && pp < rnp ? old + pp : cur + pp);
            pvv[j] = re && pp == rnp ? sl : v;
        }
#pragma unroll
        for (int j = 0; j < EI; ++j) {
            const int e = (int)threadIdx.x + 256 * j;
            if (e < d) x[(long)i * d + e] = __uint_as_float(tv[j] << 16) + pv[j];
        }
#pragma unroll
        for (int j = 0; j < PI; ++j) {
            const int k = (int)threadIdx.x + 256 * j;
            if (k < TILE_TC_MAX) phys[(long)i * TILE_TC_MAX + k] = pvv[j];
        }
    } else {
        // the whole block in one round trip to host memory: every load is issued before the first store waits for one
        const int *src = reinterpret_cast<const int *>(hblk);
        int *dst = reinterpret_cast<int *>(dblk);
        constexpr int NI = (int)(sizeof(StepBlock) / sizeof(int)), IT = (NI + 255) / 256;
        int v[IT];
#pragma unroll
        for (int j = 0; j < IT; ++j) v[j] = src[min((int)threadIdx.x + 256 * j, NI - 1)];
#pragma unroll
        for (int j = 0; j < IT; ++j) {
            const int k = (int)threadIdx.x + 256 * j;
            if (k < NI) dst[k] = v[j];
        }
    }
}

}  // namespace

// Tile sizes per class (measured, DESIGN.md §4 rule 3): 8-feature tiles (twice the workgroups) for the projections of
// at most d features (cross-q, attention outputs, fc2), 16 for qkv, 32-feature workgroups (two tiles, half the LayerNorm
// passes) for fc1.  Rows: kernels are instantiated for 1, 5 (beam step) and 16 (batched step) rows.
namespace {
void launch_proj_impl(const GemvArgs &a, const StepRow *rows, long kv_layer_off, hipStream_t st) {
    MW_CHECK(a.x && a.W && a.out && a.n >= 1 && a.n <= STEP_MAX_ROWS, "proj_wg: 1..32 rows");
    MW_CHECK(a.N % 32 == 0 && a.K % 32 == 0 && a.ldx == a.K && !a.d_rows, "proj_wg: shape");
    const bool ln = a.ln_w != nullptr;
    Proj p{(const bf16_t *)a.W, a.bias, a.N, a.K, (const unsigned char *)a.W8, a.w_scale};
    const bool w8 = a.W8 != nullptr;
    if (w8) MW_CHECK(a.w_scale && rows && a.K % 16 == 0, "proj_wg: e4m3 weights need row scales and the batched step");
    Epi e{};
    e.rows = rows;
    e.kv_layer_off = kv_layer_off;
    const dim3 block(MT);
#define MW_PROJ_LN(NBv, EPIv, DESCv, W8v)                                                                                           \
    do {                                                                                                                            \
        if (a.N <= 1280)                                                                                                            \
            hipLaunchKernelGGL((proj_ln_tile_kernel<NBv, EPIv, 8, DESCv, W8v>), dim3(a.N / 8), block, 0, st, p, (const float *)a.x, \
                               a.ln_w, a.ln_b, a.n, e);                                                                             \
        else if (a.N > 4096 && NBv <= 16)                                                                                           \
            hipLaunchKernelGGL((proj_ln_tile_kernel<(NBv <= 16 ? NBv : 16), EPIv, 32, DESCv, W8v>), dim3(a.N / 32), block, 0, st, p, \
                               (const float *)a.x, a.ln_w, a.ln_b, a.n, e);                                                         \
        else                                                                                                                        \
            hipLaunchKernelGGL((proj_ln_tile_kernel<NBv, EPIv, 16, DESCv, W8v>), dim3(a.N / 16), block, 0, st, p,                   \
                               (const float *)a.x, a.ln_w, a.ln_b, a.n, e);                                                         \
    } while (0)
#define MW_PROJ_LN_W(EPIv, DESCv, W8v)                       \
    do {                                                      \
        if (a.n == 1) MW_PROJ_LN(1, EPIv, DESCv, W8v);        \
        else if (a.n <= 5) MW_PROJ_LN(5, EPIv, DESCv, W8v);   \
        else if (a.n <= 16) MW_PROJ_LN(16, EPIv, DESCv, W8v); \
        else MW_PROJ_LN(32, EPIv, DESCv, W8v);                \
    } while (0)
#define MW_PROJ_LN_NB(EPIv, DESCv) MW_PROJ_LN_W(EPIv, DESCv, false)
    if (ln) MW_CHECK(a.ln_b && a.K <= 256 * LN_MAX_F4, "proj_wg: LayerNorm classes need K <= 1280");
    if (a.qkv) {
        MW_CHECK(ln && a.N == 3 * a.K && a.ldo == a.K && a.kv_t == DType::BF16, "proj_wg: qkv class");
        MW_CHECK(rows || (a.kcache && a.vcache && a.d_slot && a.d_pos), "proj_wg: qkv class needs a cache");
        e.q = a.out; e.kc = (bf16_t *)a.kcache; e.vc = (bf16_t *)a.vcache; e.slot = a.d_slot; e.pos = a.d_pos; e.TC = a.n_text_ctx;
        e.d = a.K; e.qk_scale = a.qk_scale;
        if (w8) MW_PROJ_LN_W(EPI_QKV, true, true);
        else if (rows) MW_PROJ_LN_NB(EPI_QKV, true);
        else MW_PROJ_LN_NB(EPI_QKV, false);
    } else if (ln && a.act == ACT_GELU) {
        MW_CHECK(!a.R, "proj_wg: fc1 class");
        e.hout = (bf16_t *)a.out; e.fout = a.out; e.ldo = (int)a.ldo;
        if (w8) {
            MW_CHECK(a.out_bf16, "proj_wg: e4m3 fc1 writes bf16");
            MW_PROJ_LN_W(EPI_GELU_BF16, false, true);
        } else if (a.out_bf16) MW_PROJ_LN_NB(EPI_GELU_BF16, false);
        else MW_PROJ_LN_NB(EPI_GELU_F32, false);
    } else if (ln) {
        MW_CHECK(!a.R && a.ldo == a.N && a.act == ACT_NONE, "proj_wg: cross-q class");
        e.fout = a.out; e.scale = a.out_scale;
        if (w8) MW_PROJ_LN_W(EPI_SCALE, false, true);
        else MW_PROJ_LN_NB(EPI_SCALE, false);
    } else {
        MW_CHECK(a.x_bf16 && a.R && a.ldo == a.N && a.act == ACT_NONE && a.out_scale == 1.0f, "proj_wg: residual class");
        e.res = a.R; e.xout = a.out;
        const int ch = a.K > 1280 ? 4 : 1, spc = a.K / (256 * ch);  // K = 8 waves x CH chunks x SPC steps x 32
        MW_CHECK(a.K == 256 * ch * spc && spc >= 2 && spc <= 5, "proj_wg: residual class needs K = 256 * {1,4} * {2..5}");
#define MW_RES(NBv, SPCv, CHv)                                                                                                        \
    do {                                                                                                                              \
        if (w8) hipLaunchKernelGGL((proj_res_tile_kernel<NBv, SPCv, CHv, 8, true>), dim3(a.N / 8), block, 0, st, p, (const bf16_t *)a.x, a.n, e); \
        else hipLaunchKernelGGL((proj_res_tile_kernel<NBv, SPCv, CHv, 8, false>), dim3(a.N / 8), block, 0, st, p, (const bf16_t *)a.x, a.n, e);  \
    } while (0)
#define MW_RES_CH(NBv, SPCv)              \
    do {                                   \
        if (ch == 1) MW_RES(NBv, SPCv, 1); \
        else MW_RES(NBv, SPCv, 4);         \
    } while (0)
#define MW_RES_SPC(SPCv)                         \
    do {                                          \
        if (a.n == 1) MW_RES_CH(1, SPCv);         \
        else if (a.n <= 5) MW_RES_CH(5, SPCv);    \
        else if (a.n <= 16) MW_RES_CH(16, SPCv);  \
        else MW_RES_CH(32, SPCv);                 \
    } while (0)
        switch (spc) {
        case 2: MW_RES_SPC(2); break;
        case 3: MW_RES_SPC(3); break;
        case 4: MW_RES_SPC(4); break;
        default: MW_RES_SPC(5); break;
        }
#undef MW_RES_SPC
#undef MW_RES_CH
#undef MW_RES
    }
#undef MW_PROJ_LN_NB
#undef MW_PROJ_LN_W
#undef MW_PROJ_LN
    MW_HIP(hipGetLastError());
}
}  // namespace

void launch_proj_wg(const GemvArgs &a, hipStream_t st) {
    MW_CHECK(a.n <= 5, "proj_wg: 1..5 rows");
    launch_proj_impl(a, nullptr, 0, st);
}

void launch_proj_batch(const GemvArgs &a, const StepBlock *blk, long kv_layer_off, hipStream_t st) {
    MW_CHECK(blk != nullptr, "proj_batch: no step block");
    launch_proj_impl(a, blk->rows, kv_layer_off, st);
}

void launch_step_begin_batch(const StepBlock *h_block_dev, StepBlock *d_block, int n_rows, int n_groups, int n_text_ctx,
                             const void *tok_emb, const float *pos_emb, float *x, int d, int *row_phys, hipStream_t st) {
    MW_CHECK(h_block_dev && d_block && row_phys && n_rows >= 1 && n_rows <= STEP_MAX_ROWS && n_groups >= 1 && n_groups <= STEP_MAX_GROUPS &&
                 n_text_ctx <= TILE_TC_MAX && d >= 1 && d <= STEP_BEGIN_MAX_D,
             "step_begin_batch: rows / groups / width");
    static_assert(sizeof(StepBlock) % sizeof(int) == 0, "StepBlock is copied int by int");
    hipLaunchKernelGGL(step_begin_batch_kernel, dim3(n_groups * KV_SLOTS + n_rows + 1), dim3(256), 0, st, h_block_dev, d_block, n_rows,
                       n_groups, n_text_ctx, (const bf16_t *)tok_emb, pos_emb, x, d, row_phys);
    MW_HIP(hipGetLastError());
}

void launch_self_attn_batch(const float *q, const StepBlock *blk, const int *row_phys, long kv_layer_off, void *out, int n, int H, int d,
                            int n_text_ctx, hipStream_t st) {
    MW_CHECK(blk && row_phys && n >= 1 && n <= STEP_MAX_ROWS && H >= 1 && d == H * 64 && n_text_ctx <= TILE_TC_MAX, "self_attn_batch: shape");
    hipLaunchKernelGGL((self_attn_lean_kernel<bf16_t, true>), dim3(n * H), dim3(MT), 0, st, q, (const bf16_t *)nullptr, (const bf16_t *)nullptr,
                       (const int *)nullptr, (const int *)nullptr, (const int *)nullptr, blk->rows, row_phys, kv_layer_off, (bf16_t *)out,
                       H, d, n_text_ctx);
    MW_HIP(hipGetLastError());
}

void launch_cross_attn_batch(const float *q, const StepBlock *blk, long k_off, long v_off, float *partial, void *out,
                             unsigned *head_cnt, int n_rows, int max_group_rows, int n_groups, int H, int d, int n_ctx_max, int splits,
                             hipStream_t st) {
    MW_CHECK(blk && n_rows >= 1 && n_rows <= STEP_MAX_ROWS && n_groups >= 1 && n_groups <= STEP_MAX_GROUPS && d == H * 64,
             "cross_attn_batch: shape");
    MW_CHECK(splits >= 1 && splits <= CROSS_SPLITS_MAX && ceil_div(n_ctx_max, splits) <= XK, "cross_attn_batch: at most 128 keys per split");
    const dim3 grid(n_groups * H * splits), block(MT);
    // the kernel variant goes by the rows of ONE request (a workgroup serves one request): <= 5 (the beams of a stream, one
    // row of a greedy stream) unless a prompt of up to 16 tokens rides in the step
    MW_CHECK(max_group_rows >= 1 && max_group_rows <= n_rows && max_group_rows <= STEP_MAX_REQ_ROWS, "cross_attn_batch: rows per request");
    if (max_group_rows <= 5)
        hipLaunchKernelGGL((cross_attn_lean_kernel<5, bf16_t, true>), grid, block, 0, st, q, (const bf16_t *)nullptr, (const bf16_t *)nullptr,
                           blk->groups, k_off, v_off, partial, (bf16_t *)out, head_cnt, 0, H, d, 0, splits);
    else
        hipLaunchKernelGGL((cross_attn_lean_kernel<16, bf16_t, true>), grid, block, 0, st, q, (const bf16_t *)nullptr, (const bf16_t *)nullptr,
                           blk->groups, k_off, v_off, partial, (bf16_t *)out, head_cnt, 0, H, d, 0, splits);
    MW_HIP(hipGetLastError());
}

void launch_cross_touch_batch(const StepBlock *blk, long k_off, long v_off, int n_groups, int H, int d, int splits, int *sink, hipStream_t st) {
    MW_CHECK(blk && n_groups >= 1 && n_groups <= STEP_MAX_GROUPS && d == H * 64 && splits >= 1 && splits <= CROSS_SPLITS_MAX, "cross_touch: shape");
    hipLaunchKernelGGL(cross_touch_kernel, dim3(n_groups * H * splits), dim3(MT), 0, st, blk->groups, k_off, v_off, H, d, splits, sink);
    MW_HIP(hipGetLastError());
}

bool logits_tile_supported(const GemvArgs &a) {
    const int nch = a.K / LKC;
    return a.n >= 1 && a.n <= STEP_MAX_ROWS && a.ln_w && a.ln_b && a.d_rows && a.K % LKC == 0 && a.ldx == a.K && !a.bias && !a.R &&
           a.act == ACT_NONE && a.out_scale == 1.0f && (nch == 3 || nch == 4 || nch == 6 || nch == 8 || nch == 10);
}

void launch_logits_tile(const GemvArgs &a, hipStream_t st) {
    MW_CHECK(logits_tile_supported(a), "logits_tile: shape");
    const dim3 grid(ceil_div(a.N, 128)), block(MT);
    const bool w8 = a.W8 != nullptr;
    if (w8) MW_CHECK(a.w_scale != nullptr, "logits_tile: e4m3 weights need row scales");
#define MW_LT(NBv, NCHv)                                                                                                          \
    do {                                                                                                                          \
        if (w8)                                                                                                                   \
            hipLaunchKernelGGL((logits_tile_kernel<NBv, NCHv, true>), grid, block, 0, st, (const bf16_t *)a.W,                    \
                               (const unsigned char *)a.W8, a.w_scale, a.x, a.d_rows, a.ln_w, a.ln_b, a.out, a.ldo, a.n, a.N);    \
        else                                                                                                                      \
            hipLaunchKernelGGL((logits_tile_kernel<NBv, NCHv, false>), grid, block, 0, st, (const bf16_t *)a.W,                   \
                               (const unsigned char *)nullptr, (const float *)nullptr, a.x, a.d_rows, a.ln_w, a.ln_b, a.out,      \
                               a.ldo, a.n, a.N);                                                                                  \
    } while (0)
#define MW_LT_NB(NCHv)                     \
    do {                                    \
        if (a.n == 1) MW_LT(1, NCHv);       \
        else if (a.n <= 5) MW_LT(5, NCHv);  \
        else if (a.n <= 16) MW_LT(16, NCHv); \
        else MW_LT(32, NCHv);               \
    } while (0)
    switch (a.K / LKC) {
    case 3: MW_LT_NB(3); break;
    case 4: MW_LT_NB(4); break;
    case 6: MW_LT_NB(6); break;
    case 8: MW_LT_NB(8); break;
    default: MW_LT_NB(10); break;
    }
#undef MW_LT_NB
#undef MW_LT
    MW_HIP(hipGetLastError());
}

void launch_self_attn_wg(const float *q, const void *kcache, const void *vcache, const int *d_slot, const int *d_pos,
                         const int *d_kv_idx, void *out, DType out_t, int n, int H, int d, int n_text_ctx, hipStream_t st) {
    MW_CHECK(n >= 1 && H >= 1 && d == H * 64 && n_text_ctx <= TILE_TC_MAX, "self_attn_wg: shape");
    if (out_t == DType::F32)
        hipLaunchKernelGGL((self_attn_lean_kernel<float, false>), dim3(n * H), dim3(MT), 0, st, q, (const bf16_t *)kcache, (const bf16_t *)vcache,
                           d_slot, d_pos, d_kv_idx, (const StepRow *)nullptr, (const int *)nullptr, 0L, (float *)out, H, d, n_text_ctx);
    else
        hipLaunchKernelGGL((self_attn_lean_kernel<bf16_t, false>), dim3(n * H), dim3(MT), 0, st, q, (const bf16_t *)kcache, (const bf16_t *)vcache,
                           d_slot, d_pos, d_kv_idx, (const StepRow *)nullptr, (const int *)nullptr, 0L, (bf16_t *)out, H, d, n_text_ctx);
    MW_HIP(hipGetLastError());
}

void launch_cross_attn_wg(const float *q, const void *kc, const void *vc, float *partial, void *out, DType out_t, unsigned *head_cnt,
                          int n, int H, int d, int n_ctx, int splits, hipStream_t st) {
    MW_CHECK(n >= 1 && n <= 5 && H >= 1 && d == H * 64, "cross_attn_wg: 1..5 rows, 64-wide heads");
    MW_CHECK(splits >= 1 && splits <= CROSS_SPLITS_MAX, "cross_attn_wg: split count");
    const int per = ceil_div(n_ctx, splits);
    MW_CHECK(per <= XK, "cross_attn_wg: at most 128 keys per split");
    const dim3 grid(H * splits), block(MT);
#define MW_CROSS(NBv, OTv)                                                                                                \
    hipLaunchKernelGGL((cross_attn_lean_kernel<NBv, OTv, false>), grid, block, 0, st, q, (const bf16_t *)kc, (const bf16_t *)vc, \
                       (const StepGroup *)nullptr, 0L, 0L, partial, (OTv *)out, head_cnt, n, H, d, n_ctx, splits)
    if (out_t == DType::F32) {
        if (n == 1) MW_CROSS(1, float);
        else MW_CROSS(5, float);
    } else {
        if (n == 1) MW_CROSS(1, bf16_t);
        else MW_CROSS(5, bf16_t);
    }
#undef MW_CROSS
    MW_HIP(hipGetLastError());
}

}  // namespace mw
