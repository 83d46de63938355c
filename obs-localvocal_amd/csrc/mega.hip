// mega.hip — the decoder stack of ONE decode step (all layers, up to 5 rows) as a single persistent
// cooperative kernel for gfx950.
//
// Why: with <= 5 rows every projection of the step is a weight-streaming matrix-vector product that is
// bound by latency, not bandwidth: as separate kernels (even replayed from a hipGraph) each of the ~290
// launches costs 5-12 us of which ~1 us is the weight stream itself (profiles/r01_*).  Here 256 workgroups
// (one per CU) stay resident for the whole step, each owning a fixed slice of every weight matrix:
//   * phases (LN+QKV | self-attn | out-proj | LN+cross-Q | cross-attn | cross-out | LN+FC1 | FC2) are
//     separated by a grid barrier (two-level arrival tree + per-workgroup release flags, ~1.9 us);
//   * a workgroup issues the 16-byte weight loads of its NEXT projection before it computes the current
//     one, so the HBM latency of the weight stream hides behind the barrier and the activation loads;
//   * activations cross workgroups through buffers that are written exactly once per launch with
//     agent-scope (write-through) stores and read with ordinary cached loads after the barrier — no line is
//     rewritten inside a launch, so no L2 (one per XCD) can hold a stale copy.  tools/microbench/mb_barrier
//     checks exactly this pattern across launches.
// Every wait is bounded: a workgroup that waits longer than MEGA_TIMEOUT_TICKS raises the abort word, every
// later barrier returns at once, the kernel drains, and the host reports the failure (engine.hip).
//
// Projection mapping (per workgroup b of G): output features [b*F, (b+1)*F) with F = ceil(N/G); inside the
// workgroup thread t owns ONE 16-byte chunk position c of the K axis (8 consecutive k) for every feature
// of its group, so its activation slice (rows x 8) stays in registers for the whole phase; partial sums
// are reduced over 16-lane DPP rows, combined through LDS in a fixed order (deterministic), and finished
// with bias / scale / GELU / residual by one thread per (feature, row).
#include "kernels.hpp"

#include <hip/hip_bf16.h>

namespace mw {

namespace {

constexpr int MT = MEGA_THREADS;
constexpr unsigned long long MEGA_TIMEOUT_TICKS = 5000000ull;  // 50 ms of the 100 MHz wall clock

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
__device__ __forceinline__ float wave_sum_all(float v) {
    v = row16_sum(v);
    v += __shfl_xor(v, 16);
    v += __shfl_xor(v, 32);
    return v;
}
__device__ __forceinline__ float wave_max_all(float v) {
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

// ---- sync area layout (unsigned words) ----
constexpr int SY_ABORT = 32, SY_ROOT = 64, SY_GROUP = 128 /* + g*64 */, SY_FLAGS = 2048 /* + b*32 */,
              SY_HEADS = 2048 + 1024 * 32 /* + layer*64 + head */;

struct Barrier {
    unsigned *sync;
    unsigned epoch;
    bool ok;
};

// Grid barrier.  Arrival: 16 group counters on separate lines, then a root counter; the last arriver writes
// one release flag per workgroup (each on its own 128-byte line); everybody else polls only its own flag.
// Counters run monotonically across launches (epoch_base), so nothing is reset between steps.
// Split in two so that the caller can issue weight prefetches between the halves without a callback (the
// register sets must stay in registers): part 1 = stores acknowledged + arrival/wait by wave 0 only.
__device__ __forceinline__ void grid_barrier_service(Barrier &bar, bool *s_ok) {
    // runs on wave 0 only
    const unsigned lane = threadIdx.x, nb = gridDim.x, g = blockIdx.x & 15, epoch = bar.epoch;
    const unsigned members = (nb + 15 - g) / 16, ngroups = nb < 16 ? nb : 16;
    unsigned *sy = bar.sync;
    bool ok = bar.ok;
    if (ok) {
        unsigned last = 0;
        if (lane == 0) {
            unsigned old = __hip_atomic_fetch_add(&sy[SY_GROUP + g * 64], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (old == members * epoch - 1) {
                old = __hip_atomic_fetch_add(&sy[SY_ROOT], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                last = old == ngroups * epoch - 1;
            }
        }
        last = __builtin_amdgcn_readfirstlane(last);
        if (last) {
            for (unsigned b = lane; b < nb; b += 64)
                __hip_atomic_store(&sy[SY_FLAGS + b * 32], epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            const unsigned long long t0 = wall_clock64();
            while (true) {
                const unsigned v = __hip_atomic_load(&sy[SY_FLAGS + blockIdx.x * 32], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if ((int)(v - epoch) >= 0) break;
                __builtin_amdgcn_s_sleep(1);
                if (wall_clock64() - t0 > MEGA_TIMEOUT_TICKS ||
                    __hip_atomic_load(&sy[SY_ABORT], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) {
                    __hip_atomic_store(&sy[SY_ABORT], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    ok = false;
                    break;
                }
            }
        }
    }
    if (lane == 0) *s_ok = ok;
}

struct Proj {
    const bf16_t *W;
    const float *bias;
    int N, K;
    int rot;  // tile t of this projection belongs to workgroup (t + rot) % gridDim.x
};
// weight registers: TPB tiles (16 features each) x J k-steps (32 k each) per wave
template <int TPB, int J> struct WRegs {
    uint4 w[TPB][J];
};
// one register set per projection family, so that every set is completely redefined by each of its prefetches
// (a partially refilled set would keep its stale half alive around the whole layer loop)
using WRegsQ = WRegs<1, 5>;   // qkv, cross-q      (K = d <= 1280, one tile per workgroup)
using WRegsO = WRegs<1, 5>;   // attention outputs (K = d)
using WRegsF = WRegs<2, 5>;   // fc1               (K = d, up to two tiles per workgroup)
using WRegsC = WRegs<1, 20>;  // fc2               (K = 4d <= 5120)

using bf16x8b = __attribute__((ext_vector_type(8))) __bf16;
using f32x4 = __attribute__((ext_vector_type(4))) float;
using u32x4v = __attribute__((ext_vector_type(4))) unsigned;

// tiles of a projection owned by this workgroup: first, first + gridDim.x, ...
struct TileGeo {
    int tiles, ksteps, first, ntile;
};
__device__ __forceinline__ TileGeo make_tiles(const Proj &p) {
    TileGeo g;
    const int nb = gridDim.x;
    g.tiles = p.N >> 4;
    g.ksteps = p.K >> 5;
    g.first = ((int)blockIdx.x - p.rot % nb + nb) % nb;
    g.ntile = g.first < g.tiles ? (g.tiles - g.first + nb - 1) / nb : 0;
    return g;
}

// B operand of v_mfma_f32_16x16x32_bf16: lane l holds W[feature 16*tile + (l & 15)][32*step + 8*(l >> 4) ..+8];
// the 8 waves of the workgroup split the k-steps (wave w takes steps w, w+8, ...)
template <int TPB, int J> __device__ __forceinline__ void load_weights(WRegs<TPB, J> &wr, const Proj &p) {
    const TileGeo g = make_tiles(p);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int fr = lane & 15, fq = lane >> 4;
#pragma unroll
    for (int k = 0; k < TPB; ++k) {
        if (k < g.ntile) {
            const bf16_t *row = p.W + (long)((g.first + k * (int)gridDim.x) * 16 + fr) * p.K + 8 * fq;
#pragma unroll
            for (int j = 0; j < J; ++j) {
                const int step = wave + 8 * j;
                if (j * 8 < g.ksteps) wr.w[k][j] = ld_stream16(row + 32 * (step < g.ksteps ? step : 0));  // clamped, unused
            }
        }
    }
}

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
};

__device__ __forceinline__ float wave_total(float v) {  // DPP reduction, result broadcast as a scalar
    v = row16_sum(v);
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x142, 0xa, 0xf, false));  // row_bcast15
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x143, 0xc, 0xf, false));  // row_bcast31
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

constexpr int LN_MAX_F4 = 5;  // float4 per lane of a LayerNorm row: K <= 1280

// One projection phase on the matrix cores.  The rows (LayerNorm'ed fp32 or plain bf16) are staged as bf16 in
// LDS (row stride K + 8 elements: conflict-free 16-byte reads), each wave multiplies its k-steps of the
// workgroup's tile(s), the 8 partial tiles meet in LDS and one thread per (feature, row) finishes.
// `wr` holds this phase's weights, issued at a grid barrier two or more phases ago.
template <int NB, bool LN, int EPI, int TPB, int J, bool LOADW = false>
__device__ __forceinline__ void proj_phase(unsigned long long *stamp, WRegs<TPB, J> &wr, const Proj &p, const void *xin,
                                           const float *ln_w, const float *ln_b, int n, const Epi &e, float *lds) {
    const TileGeo g = make_tiles(p);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int K = p.K, ldxs = K + 8;
    bf16_t *xs = reinterpret_cast<bf16_t *>(lds);                         // [n][K + 8] bf16
    float *red = lds + (MEGA_XS_BYTES / 4);                                // [TPB][8 waves][64 lanes][4]

    // ---- finisher operands first (tiny loads): thread t -> tile k, lane l, register r of the result tile ----
    const int ek = tid >> 8, el = (tid & 255) >> 2, er = tid & 3;
    const int erow = 4 * (el >> 4) + er;
    const bool finisher = ek < TPB && ek < g.ntile && erow < n;
    const int n_e = finisher ? (g.first + ek * (int)gridDim.x) * 16 + (el & 15) : 0;
    const float bias_e = (finisher && p.bias) ? p.bias[n_e] : 0.0f;
    float res_e = 0.0f;
    if constexpr (EPI == EPI_RES) res_e = finisher ? e.res[(long)erow * p.N + n_e] : 0.0f;

    // ---- stage the activation rows ----
    if constexpr (LN) {
        // wave w normalises row w: statistics by wave reductions only, no workgroup synchronisation
        if (wave < n) {
            const float *xr = (const float *)xin + (long)wave * K;
            const int nf4 = K >> 2;
            float4 v[LN_MAX_F4], gw[LN_MAX_F4], gb[LN_MAX_F4];
#pragma unroll
            for (int i = 0; i < LN_MAX_F4; ++i) {
                const int c = lane + 64 * i;
                const bool on = c < nf4;
                v[i] = on ? *reinterpret_cast<const float4 *>(xr + 4 * c) : make_float4(0, 0, 0, 0);
                gw[i] = on ? *reinterpret_cast<const float4 *>(ln_w + 4 * c) : make_float4(0, 0, 0, 0);
                gb[i] = on ? *reinterpret_cast<const float4 *>(ln_b + 4 * c) : make_float4(0, 0, 0, 0);
            }
            if constexpr (LOADW) load_weights(wr, p);  // after the row loads: vector loads return in order
            float s = 0.0f;
#pragma unroll
            for (int i = 0; i < LN_MAX_F4; ++i) s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
            const float mean = wave_total(s) * (1.0f / (float)K);
            float q = 0.0f;
#pragma unroll
            for (int i = 0; i < LN_MAX_F4; ++i) {
                if (lane + 64 * i < nf4) {
                    const float a0 = v[i].x - mean, a1 = v[i].y - mean, a2 = v[i].z - mean, a3 = v[i].w - mean;
                    q += (a0 * a0 + a1 * a1) + (a2 * a2 + a3 * a3);
                }
            }
            const float rstd = rsqrtf(wave_total(q) * (1.0f / (float)K) + 1e-5f);
#pragma unroll
            for (int i = 0; i < LN_MAX_F4; ++i) {
                const int c = lane + 64 * i;
                if (c < nf4) {
                    const float y0 = (v[i].x - mean) * rstd * gw[i].x + gb[i].x, y1 = (v[i].y - mean) * rstd * gw[i].y + gb[i].y;
                    const float y2 = (v[i].z - mean) * rstd * gw[i].z + gb[i].z, y3 = (v[i].w - mean) * rstd * gw[i].w + gb[i].w;
                    uint2 pk;
                    pk.x = (unsigned)to_bf16(y0) | ((unsigned)to_bf16(y1) << 16);
                    pk.y = (unsigned)to_bf16(y2) | ((unsigned)to_bf16(y3) << 16);
                    *reinterpret_cast<uint2 *>(xs + wave * ldxs + 4 * c) = pk;
                }
            }
        } else {
            if constexpr (LOADW) load_weights(wr, p);
        }
    } else {
        const int nu = K >> 3;  // 16-byte pieces per row
        uint4 t[NB][2];
#pragma unroll
        for (int r = 0; r < NB; ++r)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int c = tid + MT * i;
                if (r < n && c < nu) t[r][i] = *reinterpret_cast<const uint4 *>((const bf16_t *)xin + (long)r * K + 8 * c);
            }
        if constexpr (LOADW) load_weights(wr, p);
#pragma unroll
        for (int r = 0; r < NB; ++r)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int c = tid + MT * i;
                if (r < n && c < nu) *reinterpret_cast<uint4 *>(xs + r * ldxs + 8 * c) = t[r][i];
            }
    }
    if (stamp && tid == 0) stamp[0] = wall_clock64();
    lds_sync();
    if (stamp && tid == 0) stamp[1] = wall_clock64();

    // ---- matrix-core products: D[x-row 4*(l>>4)+r][feature l&15] += A(x rows) . B(weights) ----
    const int fr = lane & 15, fq = lane >> 4;
    const bf16_t *arow = xs + (fr < n ? fr : 0) * ldxs + 8 * fq;  // rows >= n: any valid row, result ignored
#pragma unroll
    for (int k = 0; k < TPB; ++k) {
        if (k < g.ntile) {
            f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
            for (int j = 0; j < J; ++j) {
                const int step = wave + 8 * j;
                if (j * 8 < g.ksteps) {
                    const uint4 af = *reinterpret_cast<const uint4 *>(arow + 32 * (step < g.ksteps ? step : 0));
                    const uint4 wv = step < g.ksteps ? wr.w[k][j] : make_uint4(0, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8b, af), __builtin_bit_cast(bf16x8b, wv), acc,
                                                                  0, 0, 0);
                }
            }
            *reinterpret_cast<f32x4 *>(red + ((k * 8 + wave) * 64 + lane) * 4) = acc;
        }
    }
    lds_sync();
    if (stamp && tid == 0) stamp[2] = wall_clock64();

    // ---- finish: fixed-order sum of the 8 partial tiles, bias, epilogue, agent-scope store ----
    if (finisher) {
        float v = 0.0f;
#pragma unroll
        for (int w = 0; w < 8; ++w) v += red[((ek * 8 + w) * 64 + el) * 4 + er];
        v += bias_e;
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
    if (stamp && tid == 0) stamp[3] = wall_clock64();
}

// ---- the LayerNorm-fused projection as a kernel of its own: one 16-feature tile per workgroup, K <= 1280 ----
// Same arithmetic as proj_phase<LN>, written so that the compiler emits straight-line code: every global load
// is unconditional (clamped address, value selected afterwards) and issued before the first wait.  A load
// under a predicate becomes a branch with `s_waitcnt vmcnt(0)` at its join; in proj_phase that put the weight
// loads first and then five serialised (row, gamma, beta) round trips in front of the LayerNorm — 3 to 4 us
// of a 5 us kernel (tools/microbench/mb_phases).
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
template <int NB, int EPI, int FR>
__device__ __forceinline__ void proj_ln_tile(const Proj &p, const float *xin, const float *ln_w, const float *ln_b, int n, const Epi &e,
                                             bf16_t *xs, bf16_t *wt, float *red, unsigned long long *stamp) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, fq = lane >> 4;
    const int K = p.K, nf4 = K >> 2, ksteps = K >> 5, ldxs = K + 8, tile = blockIdx.x;
    // ---- every global load of the kernel, in the order their values are needed ----
    // (a wave without a row issues the same loads with every lane on one address: a single cache-line request each,
    // still straight-line code — a branch around the loads made the compiler wait for them inside the branch)
    const bool has_row = wave < n;
    const float *xr = xin + (long)(has_row ? wave : 0) * K;
    float4 v[LN_MAX_F4], gw[LN_MAX_F4], gb[LN_MAX_F4];
#pragma unroll
    for (int i = 0; i < LN_MAX_F4; ++i) v[i] = *reinterpret_cast<const float4 *>(xr + (has_row ? 4 * min(lane + 64 * i, nf4 - 1) : 0));
#pragma unroll
    for (int i = 0; i < LN_MAX_F4; ++i) {
        const int c = has_row ? 4 * min(lane + 64 * i, nf4 - 1) : 0;
        gw[i] = *reinterpret_cast<const float4 *>(ln_w + c);
        gb[i] = *reinterpret_cast<const float4 *>(ln_b + c);
    }
    constexpr int RPW = FR / 8, WJ = (RPW * 160 + 63) / 64;  // weight rows per wave, load instructions per wave (K <= 1280)
    constexpr int TL = FR > 16 ? FR / 16 : 1;                // 16-feature MFMA tiles per workgroup
    uint4 w[WJ];
    const int wpieces = RPW * (K >> 3);  // 16-byte pieces in the wave's row(s)
    {
        const bf16_t *wbase = p.W + (long)(tile * FR + RPW * wave) * K;
#pragma unroll
        for (int j = 0; j < WJ; ++j) w[j] = ld_stream16(wbase + 8 * min(lane + 64 * j, wpieces - 1));
    }
    const int et = tid >> 8, el = (tid & 255) >> 2, er = tid & 3, erow = 4 * (el >> 4) + er;  // finisher: tile, lane, register
    const int n_e = tile * FR + min(16 * et + (el & 15), FR - 1);
    const float bias_e = p.bias ? p.bias[n_e] : 0.0f;
    long kv_off = 0;
    if constexpr (EPI == EPI_QKV) {
        const int rr = erow < n ? erow : 0;
        kv_off = ((long)e.slot[rr] * e.TC + e.pos[rr]) * e.d;
    }
    // ---- LayerNorm of the wave's row (wave reductions only) -> bf16 in LDS ----
    float s = 0.0f;
#pragma unroll
    for (int i = 0; i < LN_MAX_F4; ++i) s += lane + 64 * i < nf4 ? (v[i].x + v[i].y) + (v[i].z + v[i].w) : 0.0f;
    const float mean = wave_total(s) * (1.0f / (float)K);
    float q = 0.0f;
#pragma unroll
    for (int i = 0; i < LN_MAX_F4; ++i) {
        const float a0 = v[i].x - mean, a1 = v[i].y - mean, a2 = v[i].z - mean, a3 = v[i].w - mean;
        q += lane + 64 * i < nf4 ? (a0 * a0 + a1 * a1) + (a2 * a2 + a3 * a3) : 0.0f;
    }
    const float rstd = rsqrtf(wave_total(q) * (1.0f / (float)K) + 1e-5f);
#pragma unroll
    for (int i = 0; i < LN_MAX_F4; ++i) {
        const int c = lane + 64 * i;
        const float y0 = (v[i].x - mean) * rstd * gw[i].x + gb[i].x, y1 = (v[i].y - mean) * rstd * gw[i].y + gb[i].y;
        const float y2 = (v[i].z - mean) * rstd * gw[i].z + gb[i].z, y3 = (v[i].w - mean) * rstd * gw[i].w + gb[i].w;
        uint2 pk;
        pk.x = (unsigned)to_bf16(y0) | ((unsigned)to_bf16(y1) << 16);
        pk.y = (unsigned)to_bf16(y2) | ((unsigned)to_bf16(y3) << 16);
        if (c < nf4 && wave < n) *reinterpret_cast<uint2 *>(xs + wave * ldxs + 4 * c) = pk;
    }
    if (stamp && tid == 0) stamp[0] = wall_clock64();
    // ---- the wave's two weight rows -> LDS ----
#pragma unroll
    for (int j = 0; j < WJ; ++j) {
        const int pc = lane + 64 * j, half = K >> 3;  // pieces per row
        const int r = (pc >= half ? 1 : 0) + (pc >= 2 * half ? 1 : 0) + (pc >= 3 * half ? 1 : 0);
        if (pc < wpieces) *reinterpret_cast<uint4 *>(wt + (RPW * wave + r) * ldxs + 8 * (pc - r * half)) = w[j];
    }
    lds_sync();
    if (stamp && tid == 0) stamp[1] = wall_clock64();
    // ---- matrix-core products ----
    {
        const bf16_t *arow = xs + (fr < n ? fr : 0) * ldxs + 8 * fq;  // rows >= n: any valid row, result ignored
        const bf16_t *brow = wt + fr * ldxs + 8 * fq;
        f32x4 acc[TL];
#pragma unroll
        for (int tl = 0; tl < TL; ++tl) acc[tl] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            const int step = wave + 8 * j, sc = min(step, ksteps - 1);
            const uint4 af = *reinterpret_cast<const uint4 *>(arow + 32 * sc);
#pragma unroll
            for (int tl = 0; tl < TL; ++tl) {
                uint4 wv = *reinterpret_cast<const uint4 *>(brow + 16 * tl * ldxs + 32 * sc);
                if (step >= ksteps) wv = make_uint4(0, 0, 0, 0);
                acc[tl] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8b, af), __builtin_bit_cast(bf16x8b, wv), acc[tl], 0,
                                                                  0, 0);
            }
        }
#pragma unroll
        for (int tl = 0; tl < TL; ++tl) *reinterpret_cast<f32x4 *>(red + ((tl * 8 + wave) * 64 + lane) * 4) = acc[tl];
    }
    lds_sync();
    if (stamp && tid == 0) stamp[2] = wall_clock64();
    // ---- finish: fixed-order sum of the 8 partial tiles, bias, epilogue ----
    if (et < TL && erow < n && 16 * et + (el & 15) < FR) {  // tile columns beyond FR come from unwritten LDS rows: ignored
        float r = 0.0f;
#pragma unroll
        for (int k = 0; k < 8; ++k) r += red[((et * 8 + k) * 64 + el) * 4 + er];
        r += bias_e;
        if constexpr (EPI == EPI_QKV) {
            const int d = e.d;
            if (n_e < d) st_agent(e.q + (long)erow * d + n_e, r * e.qk_scale);
            else if (n_e < 2 * d) st_agent(e.kc + kv_off + (n_e - d), to_bf16(r * e.qk_scale));
            else st_agent(e.vc + kv_off + (n_e - 2 * d), to_bf16(r));
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
template <int NB, int SPC, int CH, int FR>
__device__ __forceinline__ void proj_res_tile(const Proj &p, const bf16_t *xin, int n, const Epi &e, bf16_t *lds_t, float *red,
                                              unsigned long long *stamp) {
    constexpr int KC = 32 * SPC, LDT = KC + 8, PPR = 4 * SPC;  // k per chunk, LDS row stride, 16-byte pieces per row segment
    constexpr int XLD = CH * KC + 8, XPR = PPR * CH;           // the wave's x tile: row stride, pieces per row
    constexpr int XI = (NB * XPR + 63) / 64;                   // x load instructions
    constexpr int WI = (FR * PPR + 63) / 64;                   // weight load instructions per chunk (FR feature rows per workgroup)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, fq = lane >> 4;
    const int K = p.K, tile = blockIdx.x, kw = wave * CH * KC;
    u32x4v xv[XI];  // first-class vectors: as HIP uint4 structs this array was not split into registers and went through
                    // LDS scratch, whose indexing reads the workgroup size from the dispatch packet in host memory (2-6 us)
    uint4 w[CH][WI];
#pragma unroll
    for (int i = 0; i < XI; ++i) {
        const int pc = min(lane + 64 * i, NB * XPR - 1);
        xv[i] = *reinterpret_cast<const u32x4v *>(xin + (long)min(pc / XPR, n - 1) * K + kw + 8 * (pc % XPR));
    }
    const int el = (tid & 255) >> 2, er = tid & 3, erow = 4 * (el >> 4) + er, n_e = tile * FR + min(el & 15, FR - 1);
    const float res_e = e.res[(long)(erow < n ? erow : 0) * p.N + n_e];
    const float bias_v = (p.bias ? p.bias : e.res)[n_e];  // unconditional load: a branch here would split the load block
    const float bias_e = p.bias ? bias_v : 0.0f;
#pragma unroll
    for (int c = 0; c < CH; ++c)
#pragma unroll
        for (int j = 0; j < WI; ++j) {
            const int pc = min(lane + 64 * j, FR * PPR - 1);
            w[c][j] = ld_stream16(p.W + (long)(tile * FR + pc / PPR) * K + kw + c * KC + 8 * (pc % PPR));
        }
    __builtin_amdgcn_sched_barrier(0);  // every load above is issued before anything below (no sinking next to the use)
    bf16_t *wt = lds_t + wave * (16 * LDT + NB * XLD), *xt = wt + 16 * LDT;
#pragma unroll
    for (int i = 0; i < XI; ++i) {
        const int pc = min(lane + 64 * i, NB * XPR - 1);  // surplus lanes rewrite the last piece (same data): no predicate
        *reinterpret_cast<u32x4v *>(xt + (pc / XPR) * XLD + 8 * (pc % XPR)) = xv[i];
    }
    f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
    if (stamp && tid == 0) stamp[0] = wall_clock64();
#pragma unroll
    for (int c = 0; c < CH; ++c) {
#pragma unroll
        for (int j = 0; j < WI; ++j) {
            const int pc = min(lane + 64 * j, FR * PPR - 1);  // surplus lanes rewrite the last piece
            *reinterpret_cast<uint4 *>(wt + (pc / PPR) * LDT + 8 * (pc % PPR)) = w[c][j];
        }
#pragma unroll
        for (int s2 = 0; s2 < SPC; ++s2) {
            const uint4 af = *reinterpret_cast<const uint4 *>(xt + (fr < n ? fr : 0) * XLD + c * KC + 32 * s2 + 8 * fq);
            const uint4 bfr = *reinterpret_cast<const uint4 *>(wt + fr * LDT + 32 * s2 + 8 * fq);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8b, af), __builtin_bit_cast(bf16x8b, bfr), acc, 0, 0, 0);
        }
    }
    if (stamp && tid == 0) stamp[1] = wall_clock64();
    *reinterpret_cast<f32x4 *>(red + (wave * 64 + lane) * 4) = acc;
    lds_sync();
    if (stamp && tid == 0) stamp[2] = wall_clock64();
    if (tid < 256 && erow < n && (el & 15) < FR) {  // tile columns beyond FR come from unwritten LDS rows: ignored
        float r = 0.0f;
#pragma unroll
        for (int k = 0; k < 8; ++k) r += red[(k * 64 + el) * 4 + er];
        st_agent(e.xout + (long)erow * p.N + n_e, r + bias_e + res_e);
    }
    if (stamp && tid == 0) stamp[3] = wall_clock64();
}

// ---- self-attention over the indirected KV cache: one workgroup per (row, head), thread = key.  Every
// global load of the phase (index table, K row, the thread's V chunks) is issued before anything waits. ----
__device__ __forceinline__ void st_out(bf16_t *p, float v) { st_agent(p, to_bf16(v)); }
__device__ __forceinline__ void st_out(float *p, float v) { st_agent(p, v); }

template <typename OT>
__device__ __forceinline__ void self_attn_phase(const float *q, const bf16_t *kc, const bf16_t *vc, const int *slot,
                                                const int *pos, const int *kv_idx, OT *out, int n, int H, int d, int TC,
                                                float *lds) {
    float *pr = lds;             // [MT] exp(scores)
    float *red = lds + MT;       // [8][64]
    float *sm = lds + MT + 512;  // [16] scalars
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr int PVI = MEGA_TC_MAX / 64;  // P.V iterations: thread (jj, e8) walks positions jj, jj+64, ...
    const int jj = tid >> 3, e8 = tid & 7;
    for (int it = blockIdx.x; it < n * H; it += gridDim.x) {
        const int r = it / H, h = it - r * H;
        const int np = pos[r] + 1;
        const int *idx = kv_idx + (long)slot[r] * TC;
        // index loads first, then everything that depends on them
        const int ik = tid < np ? idx[tid] : 0;
        int iv[PVI];
#pragma unroll
        for (int i = 0; i < PVI; ++i) iv[i] = (jj + 64 * i) < np ? idx[jj + 64 * i] : 0;
        uint4 kq[8];
        {
            const bf16_t *kr = kc + ((long)ik * TC + (tid < np ? tid : 0)) * d + h * 64;
#pragma unroll
            for (int c = 0; c < 8; ++c) kq[c] = *reinterpret_cast<const uint4 *>(kr + c * 8);
        }
        uint4 vq[PVI];
#pragma unroll
        for (int i = 0; i < PVI; ++i) {
            const int p = jj + 64 * i;
            vq[i] = p < np ? *reinterpret_cast<const uint4 *>(vc + ((long)iv[i] * TC + p) * d + h * 64 + e8 * 8) : make_uint4(0, 0, 0, 0);
        }
        float s = -INFINITY;
        {
            const float *qr = q + (long)r * d + h * 64;
            float dot = 0.0f;
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                float f[8];
                unpack8(kq[c], f);
                const float4 a = *reinterpret_cast<const float4 *>(qr + c * 8), b = *reinterpret_cast<const float4 *>(qr + c * 8 + 4);
                dot = fmaf(a.x, f[0], dot); dot = fmaf(a.y, f[1], dot); dot = fmaf(a.z, f[2], dot); dot = fmaf(a.w, f[3], dot);
                dot = fmaf(b.x, f[4], dot); dot = fmaf(b.y, f[5], dot); dot = fmaf(b.z, f[6], dot); dot = fmaf(b.w, f[7], dot);
            }
            if (tid < np) s = dot;
        }
        const float wm = wave_max_all(s);
        if (lane == 0) sm[wave] = wm;
        lds_sync();
        float mx = sm[0];
#pragma unroll
        for (int w = 1; w < MT / 64; ++w) mx = fmaxf(mx, sm[w]);
        const float ex = tid < np ? expf(s - mx) : 0.0f;
        pr[tid] = ex;
        const float ws = wave_sum_all(ex);
        if (lane == 0) sm[8 + wave] = ws;
        lds_sync();
        float sum = 0.0f;
#pragma unroll
        for (int w = 0; w < MT / 64; ++w) sum += sm[8 + w];
        float o[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int i = 0; i < PVI; ++i) {
            const float w = pr[jj + 64 * i];  // zero beyond np
            float f[8];
            unpack8(vq[i], f);
#pragma unroll
            for (int q8 = 0; q8 < 8; ++q8) o[q8] = fmaf(w, f[q8], o[q8]);
        }
#pragma unroll
        for (int q8 = 0; q8 < 8; ++q8) {
            float v = o[q8];
            v += __shfl_xor(v, 8);
            v += __shfl_xor(v, 16);
            v += __shfl_xor(v, 32);
            o[q8] = v;
        }
        if ((lane >> 3) == 0) {
#pragma unroll
            for (int q8 = 0; q8 < 8; ++q8) red[wave * 64 + e8 * 8 + q8] = o[q8];
        }
        lds_sync();
        if (tid < 64) {
            float v = 0.0f;
#pragma unroll
            for (int w = 0; w < MT / 64; ++w) v += red[w * 64 + tid];
            st_out(out + (long)r * d + h * 64 + tid, v / sum);
        }
        lds_sync();
    }
}

// ---- the logits projection (final LayerNorm + tied token embedding, N = n_vocab) as tile kernel ----
// 128 vocabulary rows per workgroup: the LayerNorm of the <= 5 sampled rows is done once (wave w = row w) and
// shared through LDS; wave w then owns the 16-feature tile w over the whole K and needs nobody else's sums (no
// cross-wave reduction).  Its weights stream in chunks of 128 k (16 rows x 256 contiguous bytes per chunk, 4 loads
// per lane) through a ring of LRING chunks in registers and a wave-private LDS tile (fragment layout).
constexpr int LKC = 128, LLD = LKC + 8, LRING = 3;

template <int NB, int NCH>
__device__ __forceinline__ void logits_tile(const bf16_t *W, const float *xin, const int *d_rows, const float *ln_w, const float *ln_b,
                                            float *out, long ldo, int n, int N, bf16_t *xs, bf16_t *wtile) {
    constexpr int K = NCH * LKC, nf4 = K / 4, ldxs = K + 8;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, fq = lane >> 4;
    const int f0 = (blockIdx.x * 8 + wave) * 16;  // first vocabulary row of the wave's tile
    // ---- loads: the wave's LayerNorm row (or one address), then the first LRING weight chunks ----
    const bool has_row = wave < n;
    const float *xr = xin + (long)d_rows[has_row ? wave : 0] * K;
    float4 v[LN_MAX_F4], gw[LN_MAX_F4], gb[LN_MAX_F4];
#pragma unroll
    for (int i = 0; i < LN_MAX_F4; ++i) v[i] = *reinterpret_cast<const float4 *>(xr + (has_row ? 4 * min(lane + 64 * i, nf4 - 1) : 0));
#pragma unroll
    for (int i = 0; i < LN_MAX_F4; ++i) {
        const int c = has_row ? 4 * min(lane + 64 * i, nf4 - 1) : 0;
        gw[i] = *reinterpret_cast<const float4 *>(ln_w + c);
        gb[i] = *reinterpret_cast<const float4 *>(ln_b + c);
    }
    // weight piece (chunk c, load j): row = (lane + 64 j) >> 4, 16-byte column = (lane + 64 j) & 15
    const bf16_t *wl[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int pc = lane + 64 * j;
        wl[j] = W + (long)min(f0 + (pc >> 4), N - 1) * K + 8 * (pc & 15);  // rows beyond the vocabulary: clamped, never stored
    }
    uint4 w[NCH][4];
#pragma unroll
    for (int c = 0; c < LRING && c < NCH; ++c)
#pragma unroll
        for (int j = 0; j < 4; ++j) w[c][j] = ld_stream16(wl[j] + c * LKC);
    __builtin_amdgcn_sched_barrier(0);
    // ---- LayerNorm of the wave's row -> bf16 in LDS ----
    {
        float s = 0.0f;
#pragma unroll
        for (int i = 0; i < LN_MAX_F4; ++i) s += lane + 64 * i < nf4 ? (v[i].x + v[i].y) + (v[i].z + v[i].w) : 0.0f;
        const float mean = wave_total(s) * (1.0f / (float)K);
        float q = 0.0f;
#pragma unroll
        for (int i = 0; i < LN_MAX_F4; ++i) {
            const float a0 = v[i].x - mean, a1 = v[i].y - mean, a2 = v[i].z - mean, a3 = v[i].w - mean;
            q += lane + 64 * i < nf4 ? (a0 * a0 + a1 * a1) + (a2 * a2 + a3 * a3) : 0.0f;
        }
        const float rstd = rsqrtf(wave_total(q) * (1.0f / (float)K) + 1e-5f);
#pragma unroll
        for (int i = 0; i < LN_MAX_F4; ++i) {
            const int c = lane + 64 * i;
            const float y0 = (v[i].x - mean) * rstd * gw[i].x + gb[i].x, y1 = (v[i].y - mean) * rstd * gw[i].y + gb[i].y;
            const float y2 = (v[i].z - mean) * rstd * gw[i].z + gb[i].z, y3 = (v[i].w - mean) * rstd * gw[i].w + gb[i].w;
            uint2 pk;
            pk.x = (unsigned)to_bf16(y0) | ((unsigned)to_bf16(y1) << 16);
            pk.y = (unsigned)to_bf16(y2) | ((unsigned)to_bf16(y3) << 16);
            if (c < nf4 && wave < n) *reinterpret_cast<uint2 *>(xs + wave * ldxs + 4 * c) = pk;
        }
    }
    lds_sync();
    // ---- the wave's tile over all of K: chunk c from the ring -> private LDS tile -> 4 MFMA steps ----
    bf16_t *wt = wtile + wave * 16 * LLD;
    const bf16_t *arow = xs + (fr < n ? fr : 0) * ldxs + 8 * fq;
    f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int pc = lane + 64 * j;
            *reinterpret_cast<uint4 *>(wt + (pc >> 4) * LLD + 8 * (pc & 15)) = w[c][j];
        }
        if (c + LRING < NCH) {
#pragma unroll
            for (int j = 0; j < 4; ++j) w[c + LRING][j] = ld_stream16(wl[j] + (c + LRING) * LKC);
        }
#pragma unroll
        for (int s2 = 0; s2 < 4; ++s2) {
            const uint4 af = *reinterpret_cast<const uint4 *>(arow + c * LKC + 32 * s2);
            const uint4 bfr = *reinterpret_cast<const uint4 *>(wt + fr * LLD + 32 * s2 + 8 * fq);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8b, af), __builtin_bit_cast(bf16x8b, bfr), acc, 0, 0, 0);
        }
    }
    // lane holds rows 4*fq + r of vocabulary entry f0 + fr
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = 4 * fq + r;
        if (row < n && f0 + fr < N) out[(long)row * ldo + f0 + fr] = acc[r];
    }
}

// ---- self-attention as a kernel of its own, straight-line: three dependent load levels and no more ----
//   1. pos / slot of the row (scalar) and the thread's 8 query values        2. the index-table entries of its keys
//   3. its 16-byte K and V chunks (thread (jj, e8): keys jj, jj+64, ...; 8 lanes cover a 128-byte head row, so an
//      instruction reads whole cache lines)
// Loads beyond the row's length are clamped to its last position (one line for the whole wave) and masked in the
// arithmetic; nothing is predicated, so the compiler keeps each level's loads in one batch.  Probabilities stay
// in registers (the thread that scored a key also weights its V chunk).
template <typename OT>
__device__ __forceinline__ void self_attn_lean(const float *q, const bf16_t *kc, const bf16_t *vc, const int *slot, const int *pos,
                                               const int *kv_idx, OT *out, int H, int d, int TC, float *red, float *sm) {
    constexpr int PVI = MEGA_TC_MAX / 64;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, jj = tid >> 3, e8 = tid & 7;
    const int r = blockIdx.x / H, h = blockIdx.x - r * H;
    const int np = pos[r] + 1;
    const int *idx = kv_idx + (long)slot[r] * TC;
    const float4 qa = *reinterpret_cast<const float4 *>(q + (long)r * d + h * 64 + e8 * 8);
    const float4 qb = *reinterpret_cast<const float4 *>(q + (long)r * d + h * 64 + e8 * 8 + 4);
    int iv[PVI];
#pragma unroll
    for (int i = 0; i < PVI; ++i) iv[i] = idx[min(jj + 64 * i, np - 1)];
    uint4 kq[PVI], vq[PVI];
#pragma unroll
    for (int i = 0; i < PVI; ++i) {
        const long off = ((long)iv[i] * TC + min(jj + 64 * i, np - 1)) * d + h * 64 + e8 * 8;
        kq[i] = *reinterpret_cast<const uint4 *>(kc + off);
        vq[i] = *reinterpret_cast<const uint4 *>(vc + off);
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
        dot += __shfl_xor(dot, 1);
        dot += __shfl_xor(dot, 2);
        dot += __shfl_xor(dot, 4);
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
        float v = o[q8];
        v += __shfl_xor(v, 8);
        v += __shfl_xor(v, 16);
        v += __shfl_xor(v, 32);
        o[q8] = v;
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

constexpr int CPART = 66;

// tail of a cross-attention work item: count this split of head h; the last one to arrive combines the partial
// softmax records of all splits (fixed split order) into the output rows
template <int NB, typename OT>
__device__ __forceinline__ void cross_count_and_combine(const float *part, OT *out, unsigned *head_cnt, float *sc, unsigned *flag,
                                                        int n, int H, int d, int h, int splits) {
    const int tid = threadIdx.x;
        __syncthreads();  // partial stores acknowledged (vmcnt(0)) before this split is counted
        if (tid == 0) {
            const unsigned old = __hip_atomic_fetch_add(&head_cnt[h], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned last = old == (unsigned)splits - 1;
            if (last) __hip_atomic_store(&head_cnt[h], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // ready for the next launch
            *flag = last;
        }
        lds_sync();
        if (*flag) {  // every split of head h is in memory: combine in fixed split order
            // one batch of unconditional loads per thread (row r, dim e): the splits' (max, sum) pairs — the same
            // address for the 64 threads of a row — and the thread's own partial outputs; surplus slots repeat the
            // last split and get weight zero
            const int i = min(tid, NB * 64 - 1), r = min(i >> 6, n - 1), e = i & 63;
            const float *pp = part + ((long)r * H + h) * splits * CPART;
            float2 mlv[CROSS_SPLITS_MAX];
            float ov[CROSS_SPLITS_MAX];
#pragma unroll
            for (int sidx = 0; sidx < CROSS_SPLITS_MAX; ++sidx) {
                const float *ps = pp + min(sidx, splits - 1) * CPART;
                mlv[sidx] = *reinterpret_cast<const float2 *>(ps);
                ov[sidx] = ps[2 + e];
            }
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
            if (tid < NB * 64 && (tid >> 6) < n) st_out(out + (long)r * d + h * 64 + e, acc / Lsum);
        }
}

// ---- cross-attention: workgroup = (head, key split); all rows share each K/V read; the last split of a head
// to finish combines the partial softmaxes (no waiting: an arrival counter per head) ----
template <int NB, typename OT>
__device__ __forceinline__ void cross_attn_phase(const float *q, const bf16_t *kc, const bf16_t *vc, float *part, OT *out,
                                                 unsigned *head_cnt, int n, int H, int d, int n_ctx, int splits, int per,
                                                 float *lds) {
    float *qs = lds;                   // [NB][64]
    float *sc = lds + NB * 64;         // [NB][MT]
    float *red = sc + NB * MT;         // [8][NB][64]
    float *ml = red + 8 * NB * 64;     // [NB][2]
    unsigned *flag = reinterpret_cast<unsigned *>(ml + NB * 2);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr int PVI = MEGA_CROSS_PER_MAX / 64;  // a split holds at most MEGA_CROSS_PER_MAX keys
    const int jj = tid >> 3, e8 = tid & 7;
    for (int it = blockIdx.x; it < H * splits; it += gridDim.x) {
        const int h = it / splits, sp = it - h * splits;
        const int j0 = sp * per, nk = min(per, n_ctx - j0);  // may be <= 0 for a trailing split
        // all global loads of the split up front: q, this thread's K row, its V chunks
        float qreg[(NB * 64 + MT - 1) / MT];
#pragma unroll
        for (int u = 0; u < (NB * 64 + MT - 1) / MT; ++u) {
            const int i = tid + u * MT, r = i >> 6, e = i & 63;
            qreg[u] = (i < NB * 64 && r < n) ? q[(long)r * d + h * 64 + e] : 0.0f;
        }
        uint4 kq[8];
        {
            const bf16_t *kr = kc + (long)(j0 + (tid < nk ? tid : 0)) * d + h * 64;
#pragma unroll
            for (int c = 0; c < 8; ++c) kq[c] = *reinterpret_cast<const uint4 *>(kr + c * 8);
        }
        uint4 vq[PVI];
#pragma unroll
        for (int i = 0; i < PVI; ++i) {
            const int j = jj + 64 * i;
            vq[i] = (i * 64 < per && j < nk) ? *reinterpret_cast<const uint4 *>(vc + (long)(j0 + j) * d + h * 64 + e8 * 8)
                                             : make_uint4(0, 0, 0, 0);
        }
#pragma unroll
        for (int u = 0; u < (NB * 64 + MT - 1) / MT; ++u) {
            const int i = tid + u * MT;
            if (i < NB * 64) qs[i] = qreg[u];
        }
        lds_sync();
        {
            float s[NB];
#pragma unroll
            for (int r = 0; r < NB; ++r) s[r] = 0.0f;
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                float f[8];
                unpack8(kq[c], f);
#pragma unroll
                for (int r = 0; r < NB; ++r) {
                    const float4 a = *reinterpret_cast<const float4 *>(qs + r * 64 + c * 8);
                    const float4 b = *reinterpret_cast<const float4 *>(qs + r * 64 + c * 8 + 4);
                    float t = s[r];
                    t = fmaf(a.x, f[0], t); t = fmaf(a.y, f[1], t); t = fmaf(a.z, f[2], t); t = fmaf(a.w, f[3], t);
                    t = fmaf(b.x, f[4], t); t = fmaf(b.y, f[5], t); t = fmaf(b.z, f[6], t); t = fmaf(b.w, f[7], t);
                    s[r] = t;
                }
            }
#pragma unroll
            for (int r = 0; r < NB; ++r) sc[r * MT + tid] = tid < nk ? s[r] : -INFINITY;
        }
        lds_sync();
        for (int r = wave; r < NB; r += MT / 64) {
            float v[MT / 64];
            float mx = -INFINITY;
#pragma unroll
            for (int u = 0; u < MT / 64; ++u) {
                v[u] = sc[r * MT + lane + 64 * u];
                mx = fmaxf(mx, v[u]);
            }
            mx = wave_max_all(mx);
            float sum = 0.0f;
#pragma unroll
            for (int u = 0; u < MT / 64; ++u) {
                const float ex = (lane + 64 * u) < nk ? expf(v[u] - mx) : 0.0f;
                sc[r * MT + lane + 64 * u] = ex;
                sum += ex;
            }
            sum = wave_sum_all(sum);
            if (lane == 0) {
                ml[r * 2] = mx;
                ml[r * 2 + 1] = sum;
            }
        }
        lds_sync();
        float o[NB][8];
#pragma unroll
        for (int r = 0; r < NB; ++r)
#pragma unroll
            for (int q8 = 0; q8 < 8; ++q8) o[r][q8] = 0.0f;
#pragma unroll
        for (int i = 0; i < PVI; ++i) {
            if (i * 64 < per) {
                float f[8];
                unpack8(vq[i], f);
#pragma unroll
                for (int r = 0; r < NB; ++r) {
                    const float w = sc[r * MT + jj + 64 * i];  // zero beyond nk
#pragma unroll
                    for (int q8 = 0; q8 < 8; ++q8) o[r][q8] = fmaf(w, f[q8], o[r][q8]);
                }
            }
        }
#pragma unroll
        for (int r = 0; r < NB; ++r)
#pragma unroll
            for (int q8 = 0; q8 < 8; ++q8) {
                float v = o[r][q8];
                v += __shfl_xor(v, 8);
                v += __shfl_xor(v, 16);
                v += __shfl_xor(v, 32);
                o[r][q8] = v;
            }
        if ((lane >> 3) == 0) {
#pragma unroll
            for (int r = 0; r < NB; ++r)
#pragma unroll
                for (int q8 = 0; q8 < 8; ++q8) red[(wave * NB + r) * 64 + e8 * 8 + q8] = o[r][q8];
        }
        lds_sync();
        for (int i = tid; i < NB * 64; i += MT) {
            const int r = i >> 6, e = i & 63;
            if (r < n) {
                float v = 0.0f;
#pragma unroll
                for (int w = 0; w < MT / 64; ++w) v += red[(w * NB + r) * 64 + e];
                float *pp = part + (((long)r * H + h) * splits + sp) * CPART;
                st_agent(pp + 2 + e, v);
                if (e == 0) {
                    st_agent(pp, ml[r * 2]);
                    st_agent(pp + 1, ml[r * 2 + 1]);
                }
            }
        }
        cross_count_and_combine<NB, OT>(part, out, head_cnt, sc, flag, n, H, d, h, splits);
        lds_sync();
    }
}

// ---- cross-attention on the matrix cores (splits of at most 128 keys) ----
// scores  S[row][key] = q[row] . K[key]:  A = q rows as bf16 (LDS), B = K rows straight from global memory (a lane's
//         16 bytes are 8 consecutive head dims of its key), one 16-key tile per wave, 2 x v_mfma_f32_16x16x32_bf16;
// P . V   O[row][dim] = sum_key P[row][key] V[key][dim]:  B needs 8 consecutive KEYS of one dim per lane, so the V
//         tile is transposed on its way into LDS; wave w takes dims 16*(w&3).. and key half w>>2, 2 MFMAs each.
constexpr int XK = 128;      // keys per split
constexpr int XLDP = XK + 8; // LDS row stride (elements) of the [*][key] tiles: conflict-free 16-byte reads

template <int NB, typename OT>
__device__ __forceinline__ void cross_attn_phase_mfma(const float *q, const bf16_t *kc, const bf16_t *vc, float *part, OT *out,
                                                      unsigned *head_cnt, int n, int H, int d, int n_ctx, int splits, int per,
                                                      float *lds) {
    bf16_t *qb = reinterpret_cast<bf16_t *>(lds);  // [16][72]   q rows (rows >= n zero)
    bf16_t *vt = qb + 16 * 72;                     // [64][XLDP] V transposed: [dim][key]
    bf16_t *pb = vt + 64 * XLDP;                   // [16][XLDP] probabilities (rows >= n zero)
    float *sc = reinterpret_cast<float *>(pb + 16 * XLDP);  // [NB][XK] scores; reused by the combine
    float *red = sc + (NB * XK > 2 * NB * CROSS_SPLITS_MAX ? NB * XK : 2 * NB * CROSS_SPLITS_MAX);  // [8][64][4]
    float *ml = red + 8 * 64 * 4;                  // [NB][2]
    unsigned *flag = reinterpret_cast<unsigned *>(ml + NB * 2);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, fq = lane >> 4;
    // zero the operand tiles once: rows >= n and keys >= nk must contribute nothing
    for (int i = tid; i < (16 * 72 + 16 * XLDP) / 2; i += MT) {
        unsigned *z = i < 16 * 72 / 2 ? reinterpret_cast<unsigned *>(qb) + i : reinterpret_cast<unsigned *>(pb) + (i - 16 * 72 / 2);
        *z = 0u;
    }
    for (int it = blockIdx.x; it < H * splits; it += gridDim.x) {
        const int h = it / splits, sp = it - h * splits;
        const int j0 = sp * per, nk = min(per, n_ctx - j0);  // may be <= 0 for a trailing split
        // ---- every global load of the item up front ----
        float qreg[(NB * 64 + MT - 1) / MT];
#pragma unroll
        for (int u = 0; u < (NB * 64 + MT - 1) / MT; ++u) {
            const int i = tid + u * MT, r = i >> 6, e = i & 63;
            qreg[u] = (i < NB * 64 && r < n) ? q[(long)r * d + h * 64 + e] : 0.0f;
        }
        const int kkey = 16 * wave + fr;  // this lane's key in the score tile of its wave
        uint4 kb[2];
        {
            const bf16_t *kr = kc + (long)(j0 + (kkey < nk ? kkey : 0)) * d + h * 64 + 8 * fq;
            kb[0] = *reinterpret_cast<const uint4 *>(kr);
            kb[1] = *reinterpret_cast<const uint4 *>(kr + 32);
        }
        const int vk = tid >> 3, e8 = tid & 7;  // V: thread loads 8 dims of keys vk and vk + 64
        uint4 vq[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int j = vk + 64 * i;
            vq[i] = j < nk ? *reinterpret_cast<const uint4 *>(vc + (long)(j0 + j) * d + h * 64 + e8 * 8) : make_uint4(0, 0, 0, 0);
        }
        // ---- stage q (bf16) and V^T in LDS ----
#pragma unroll
        for (int u = 0; u < (NB * 64 + MT - 1) / MT; ++u) {
            const int i = tid + u * MT;
            if (i < NB * 64) qb[(i >> 6) * 72 + (i & 63)] = to_bf16(qreg[u]);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int j = vk + 64 * i;
            const unsigned w4[4] = {vq[i].x, vq[i].y, vq[i].z, vq[i].w};
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                vt[(e8 * 8 + 2 * c) * XLDP + j] = (bf16_t)(w4[c] & 0xffffu);
                vt[(e8 * 8 + 2 * c + 1) * XLDP + j] = (bf16_t)(w4[c] >> 16);
            }
        }
        lds_sync();
        // ---- scores ----
        {
            f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const uint4 af = *reinterpret_cast<const uint4 *>(qb + fr * 72 + 32 * s2 + 8 * fq);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8b, af), __builtin_bit_cast(bf16x8b, kb[s2]), acc, 0,
                                                              0, 0);
            }
            // lane holds rows 4*fq + r of key kkey
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 4 * fq + r;
                if (row < NB) sc[row * XK + kkey] = kkey < nk ? acc[r] : -INFINITY;
            }
        }
        lds_sync();
        // ---- per-row softmax over the split (wave w: row w) ----
        if (wave < NB) {
            const int r = wave;
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
        lds_sync();
        // ---- P . V ----
        {
            const int nt = wave & 3, kh = wave >> 2;
            f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const int ks = 2 * kh + s2;
                const uint4 af = *reinterpret_cast<const uint4 *>(pb + fr * XLDP + 32 * ks + 8 * fq);
                const uint4 bfr = *reinterpret_cast<const uint4 *>(vt + (16 * nt + fr) * XLDP + 32 * ks + 8 * fq);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8b, af), __builtin_bit_cast(bf16x8b, bfr), acc, 0, 0,
                                                              0);
            }
            *reinterpret_cast<f32x4 *>(red + (wave * 64 + lane) * 4) = acc;  // rows 4*fq + r, dim 16*nt + fr
        }
        lds_sync();
        for (int i = tid; i < NB * 64; i += MT) {
            const int r = i >> 6, e = i & 63;
            if (r < n) {
                const int nt = e >> 4, src_lane = (r >> 2) * 16 + (e & 15), rr = r & 3;
                const float v = red[(nt * 64 + src_lane) * 4 + rr] + red[((4 + nt) * 64 + src_lane) * 4 + rr];
                float *pp = part + (((long)r * H + h) * splits + sp) * CPART;
                st_agent(pp + 2 + e, v);
                if (e == 0) {
                    st_agent(pp, ml[r * 2]);
                    st_agent(pp + 1, ml[r * 2 + 1]);
                }
            }
        }
        cross_count_and_combine<NB, OT>(part, out, head_cnt, sc, flag, n, H, d, h, splits);
        lds_sync();
    }
}

// ---- cross-attention on the matrix cores as a straight-line kernel of its own (one (head, split) per workgroup) ----
// Same arithmetic as cross_attn_phase_mfma; differences: every global load is unconditional (keys beyond the split
// are clamped to its last key and masked in the scores) and issued before anything else; K is loaded like V, 8 lanes
// per 128-byte head row (whole cache lines per instruction), and re-read from LDS in the B-fragment layout; no LDS
// zero-fill (operand rows beyond the real rows only feed result rows nobody reads).
constexpr int XKD = 64 + 8;  // LDS row stride of the [key][dim] K tile

template <int NB, typename OT>
__device__ __forceinline__ void cross_attn_lean(const float *q, const bf16_t *kc, const bf16_t *vc, float *part, OT *out,
                                                unsigned *head_cnt, int n, int H, int d, int n_ctx, int splits, int per, float *lds) {
    bf16_t *qb = reinterpret_cast<bf16_t *>(lds);  // [16][72]    q rows
    bf16_t *kt = qb + 16 * 72;                     // [XK][XKD]   K rows
    bf16_t *vt = kt + XK * XKD;                    // [64][XLDP]  V transposed: [dim][key]
    bf16_t *pb = vt + 64 * XLDP;                   // [16][XLDP]  probabilities
    float *sc = reinterpret_cast<float *>(pb + 16 * XLDP);  // [NB][XK] scores; reused by the combine
    float *red = sc + (NB * XK > 2 * NB * CROSS_SPLITS_MAX ? NB * XK : 2 * NB * CROSS_SPLITS_MAX);  // [8][64][4]
    float *ml = red + 8 * 64 * 4;                  // [NB][2]
    unsigned *flag = reinterpret_cast<unsigned *>(ml + NB * 2);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, fq = lane >> 4;
    const int h = blockIdx.x / splits, sp = blockIdx.x - h * splits;
    const int j0 = sp * per, nk = min(per, n_ctx - j0);  // may be <= 0 for a trailing split
    const int last = max(nk - 1, 0) + min(j0, n_ctx - 1);  // a valid key for clamped loads
    // ---- loads ----
    const int qi = min(tid, NB * 64 - 1);
    const float qv = q[(long)min(qi >> 6, n - 1) * d + h * 64 + (qi & 63)];
    const int kk = tid >> 3, e8 = tid & 7;
    u32x4v kq[2], vq[2];  // first-class vectors: component access stays in registers
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int key = kk + 64 * i;
        const long off = (long)(key < nk ? j0 + key : last) * d + h * 64 + e8 * 8;
        kq[i] = *reinterpret_cast<const u32x4v *>(kc + off);
        vq[i] = *reinterpret_cast<const u32x4v *>(vc + off);
    }
    __builtin_amdgcn_sched_barrier(0);
    // ---- stage q, K and V^T ----
    qb[(qi >> 6) * 72 + (qi & 63)] = to_bf16(qv);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int key = kk + 64 * i;
        *reinterpret_cast<u32x4v *>(kt + key * XKD + e8 * 8) = kq[i];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            vt[(e8 * 8 + 2 * c) * XLDP + key] = (bf16_t)(vq[i][c] & 0xffffu);
            vt[(e8 * 8 + 2 * c + 1) * XLDP + key] = (bf16_t)(vq[i][c] >> 16);
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
    // ---- per-row softmax over the split (wave w: row w) ----
    if (wave < NB) {
        const int r = wave;
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
    lds_sync();
    // ---- P . V ----
    {
        const int nt = wave & 3, kh = wave >> 2;
        f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            const int ks = 2 * kh + s2;
            const uint4 af = *reinterpret_cast<const uint4 *>(pb + fr * XLDP + 32 * ks + 8 * fq);
            const uint4 bfr = *reinterpret_cast<const uint4 *>(vt + (16 * nt + fr) * XLDP + 32 * ks + 8 * fq);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8b, af), __builtin_bit_cast(bf16x8b, bfr), acc, 0, 0, 0);
        }
        *reinterpret_cast<f32x4 *>(red + (wave * 64 + lane) * 4) = acc;
    }
    lds_sync();
    if (tid < NB * 64) {
        const int r = tid >> 6, e = tid & 63;
        if (r < n) {
            const int nt = e >> 4, src_lane = (r >> 2) * 16 + (e & 15), rr = r & 3;
            const float v = red[(nt * 64 + src_lane) * 4 + rr] + red[((4 + nt) * 64 + src_lane) * 4 + rr];
            float *pp = part + (((long)r * H + h) * splits + sp) * CPART;
            st_agent(pp + 2 + e, v);
            if (e == 0) {
                st_agent(pp, ml[r * 2]);
                st_agent(pp + 1, ml[r * 2 + 1]);
            }
        }
    }
    cross_count_and_combine<NB, OT>(part, out, head_cnt, sc, flag, n, H, d, h, splits);
}

// optional per-phase timestamps (a.prof != null): MEGA_PROF_STRIDE wall-clock samples per (layer, workgroup)
#define MEGA_STAMP(i)                                                                                              \
    do {                                                                                                           \
        if (a.prof && threadIdx.x == 0) a.prof[((size_t)l * gridDim.x + blockIdx.x) * MEGA_PROF_STRIDE + (i)] = wall_clock64();  \
    } while (0)
// Grid barrier with a weight prefetch folded in: waves 1..7 issue the loads while wave 0 runs the barrier
// protocol (its atomics' return values would otherwise queue behind its own prefetch loads); wave 0 issues its
// share after the release.  The leading __syncthreads() waits for this workgroup's stores (vmcnt(0)).
#define MEGA_PSTAMP(i) (a.prof ? a.prof + ((size_t)l * gridDim.x + blockIdx.x) * MEGA_PROF_STRIDE + (i) : nullptr)
#define MEGA_BARRIER(stamp_before, prefetch_stmt)                 \
    MEGA_STAMP(stamp_before);                                     \
    __syncthreads();                                              \
    bar.epoch += 1;                                               \
    if (threadIdx.x >= 64) {                                      \
        prefetch_stmt;                                            \
    } else {                                                      \
        grid_barrier_service(bar, &s_ok);                         \
    }                                                             \
    lds_sync();                                                   \
    bar.ok = s_ok;                                                \
    if (!bar.ok) return;                                          \
    if (threadIdx.x < 64) {                                       \
        prefetch_stmt;                                            \
    }                                                             \
    MEGA_STAMP(stamp_before + 1)

template <int NB>
__global__ __launch_bounds__(MEGA_THREADS) void decode_mega_kernel(MegaArgs a) {
    // projection phases: bf16 rows + partial tiles; the attention phases reuse the same area
    __shared__ __attribute__((aligned(16))) float lds[MEGA_XS_BYTES / 4 + 2 * 8 * 64 * 4];
    static_assert(sizeof(lds) >= sizeof(float) * (NB * 64 + NB * MT + 8 * NB * 64 + NB * 2 + 16), "attention scratch");
    __shared__ bool s_ok;
    float *att = lds;
    Barrier bar{a.sync, a.epoch_base, true};
    const int d = a.d, n = a.n;
    // weight register sets and where they are refilled (B_k = grid barrier after phase k):
    //   wQ: qkv (P1) -> cq at B1 (P4) -> next layer's qkv at B4        wO: out (P3) -> co at B3 (P6) -> next out at B6
    //   wF: fc1 at B5 (P7)                                              wC: fc2 at B7 (P8)
    // i.e. every projection's weights are requested one to three phases before they are used.
    WRegsQ wQ;
    WRegsO wO;
    WRegsF wF;
    WRegsC wC;
    // tile -> workgroup rotations: the three d x d projections and fc2 use disjoint workgroups where the grid allows
    const int t1 = d >> 4;
    const int rot_out = 0, rot_cq = t1, rot_co = 2 * t1, rot_fc2 = t1 / 2;
    {
        const MegaLayerW &L0 = a.layers[0];
        load_weights(wQ, Proj{L0.qkv_w, L0.qkv_b, 3 * d, d, 0});
        load_weights(wO, Proj{L0.out_w, L0.out_b, d, d, rot_out});
    }
    const MegaActLayout lay = mega_act_layout(d, a.H);
    for (int l = 0; l < a.n_layer; ++l) {
        // layer pointers are re-read from the (constant) layer table right where they are used: keeping all 22
        // of them in scalar registers for the whole layer spills
        const MegaLayerW *Lp = a.layers + l;
        char *base = a.act + (size_t)l * a.act_layer_stride;
        const bool more = l + 1 < a.n_layer;
        const MegaLayerW *Ln = a.layers + (more ? l + 1 : l);
#define ACT_F32(off) reinterpret_cast<float *>(base + (off))
#define ACT_BF16(off) reinterpret_cast<bf16_t *>(base + (off))
        const float *x_in = l == 0 ? a.x0 : reinterpret_cast<const float *>(a.act + (size_t)(l - 1) * a.act_layer_stride + lay.xc);
        Epi e{};
        // P1: LN1 + Q|K|V (K and V rows go straight into the cache)
        MEGA_STAMP(0);
        e.q = ACT_F32(lay.q1); e.kc = Lp->ks; e.vc = Lp->vs; e.slot = a.slot; e.pos = a.pos; e.TC = a.TC; e.d = d; e.qk_scale = a.qk_scale;
        proj_phase<NB, true, EPI_QKV>(MEGA_PSTAMP(17), wQ, Proj{Lp->qkv_w, Lp->qkv_b, 3 * d, d, 0}, x_in, Lp->ln1_w, Lp->ln1_b, n, e, lds);
        MEGA_BARRIER(1, load_weights(wQ, Proj{Lp->cq_w, Lp->cq_b, d, d, rot_cq}));
        // P2: self-attention
        self_attn_phase<bf16_t>(ACT_F32(lay.q1), Lp->ks, Lp->vs, a.slot, a.pos, a.kv_idx, ACT_BF16(lay.at1), n, a.H, d, a.TC, att);
        MEGA_BARRIER(3, (void)0);
        // P3: attention output projection + residual
        e.res = x_in; e.xout = ACT_F32(lay.xa);
        proj_phase<NB, false, EPI_RES>(MEGA_PSTAMP(21), wO, Proj{Lp->out_w, Lp->out_b, d, d, rot_out}, ACT_BF16(lay.at1), nullptr, nullptr, n, e, lds);
        MEGA_BARRIER(5, load_weights(wO, Proj{Lp->co_w, Lp->co_b, d, d, rot_co}));
        // P4: LNc + cross query
        e.fout = ACT_F32(lay.q2); e.scale = a.qk_scale;
        proj_phase<NB, true, EPI_SCALE>(MEGA_PSTAMP(25), wQ, Proj{Lp->cq_w, Lp->cq_b, d, d, rot_cq}, ACT_F32(lay.xa), Lp->lnc_w, Lp->lnc_b, n, e, lds);
        MEGA_BARRIER(7, if (more) load_weights(wQ, Proj{Ln->qkv_w, Ln->qkv_b, 3 * d, d, 0}));
        // P5: cross-attention (+ combine by the last split of each head)
        cross_attn_phase<NB, bf16_t>(ACT_F32(lay.q2), Lp->ck, Lp->cv, ACT_F32(lay.part), ACT_BF16(lay.at2), a.sync + SY_HEADS + l * 64, n, a.H, d,
                             a.n_ctx, a.splits, a.per, att);
        MEGA_BARRIER(9, load_weights(wF, Proj{Lp->fc1_w, Lp->fc1_b, 4 * d, d, 0}));
        // P6: cross output projection + residual
        e.res = ACT_F32(lay.xa); e.xout = ACT_F32(lay.xb);
        proj_phase<NB, false, EPI_RES>(MEGA_PSTAMP(29), wO, Proj{Lp->co_w, Lp->co_b, d, d, rot_co}, ACT_BF16(lay.at2), nullptr, nullptr, n, e, lds);
        MEGA_BARRIER(11, if (more) load_weights(wO, Proj{Ln->out_w, Ln->out_b, d, d, rot_out}));
        // P7: LN2 + FC1 + GELU
        e.hout = ACT_BF16(lay.ffd); e.ldo = 4 * d;
        proj_phase<NB, true, EPI_GELU_BF16>(MEGA_PSTAMP(33), wF, Proj{Lp->fc1_w, Lp->fc1_b, 4 * d, d, 0}, ACT_F32(lay.xb), Lp->ln2_w, Lp->ln2_b, n, e, lds);
        MEGA_BARRIER(13, load_weights(wC, Proj{Lp->fc2_w, Lp->fc2_b, d, 4 * d, rot_fc2}));
        // P8: FC2 + residual
        e.res = ACT_F32(lay.xb); e.xout = more ? ACT_F32(lay.xc) : a.x_out;
        proj_phase<NB, false, EPI_RES>(MEGA_PSTAMP(37), wC, Proj{Lp->fc2_w, Lp->fc2_b, d, 4 * d, rot_fc2}, ACT_BF16(lay.ffd), nullptr, nullptr, n, e, lds);
        if (more) {
            MEGA_BARRIER(15, (void)0);
        }
#undef ACT_F32
#undef ACT_BF16
    }
}

// ---- the two attention phases as kernels of their own (the graph-replayed decode step uses these) ----
template <typename OT>
__global__ __launch_bounds__(MEGA_THREADS) void self_attn_wg_kernel(const float *q, const bf16_t *kc, const bf16_t *vc, const int *slot,
                                                                     const int *pos, const int *kv_idx, OT *out, int n, int H, int d,
                                                                     int TC) {
    __shared__ __attribute__((aligned(16))) float lds[MT + 512 + 16];
    self_attn_phase<OT>(q, kc, vc, slot, pos, kv_idx, out, n, H, d, TC, lds);
}

template <int NB, int NCH>
__global__ __launch_bounds__(MEGA_THREADS) void logits_tile_kernel(const bf16_t *W, const float *xin, const int *d_rows, const float *ln_w,
                                                                    const float *ln_b, float *out, long ldo, int n, int N) {
    __shared__ __attribute__((aligned(16))) bf16_t xs[NB * (NCH * LKC + 8)];
    __shared__ __attribute__((aligned(16))) bf16_t wtile[8 * 16 * LLD];
    logits_tile<NB, NCH>(W, xin, d_rows, ln_w, ln_b, out, ldo, n, N, xs, wtile);
}

template <typename OT>
__global__ __launch_bounds__(MEGA_THREADS) void self_attn_lean_kernel(const float *q, const bf16_t *kc, const bf16_t *vc, const int *slot,
                                                                       const int *pos, const int *kv_idx, OT *out, int H, int d, int TC) {
    __shared__ __attribute__((aligned(16))) float red[8 * 64];
    __shared__ float sm[16];
    self_attn_lean<OT>(q, kc, vc, slot, pos, kv_idx, out, H, d, TC, red, sm);
}

template <int NB, typename OT>
__global__ __launch_bounds__(MEGA_THREADS) void cross_attn_wg_kernel(const float *q, const bf16_t *kc, const bf16_t *vc, float *part,
                                                                      OT *out, unsigned *head_cnt, int n, int H, int d, int n_ctx,
                                                                      int splits, int per) {
    __shared__ __attribute__((aligned(16))) float lds[NB * 64 + NB * MT + 8 * NB * 64 + NB * 2 + 16];
    cross_attn_phase<NB, OT>(q, kc, vc, part, out, head_cnt, n, H, d, n_ctx, splits, per, lds);
}

// ---- one projection as a kernel of its own: workgroup = one 16-feature tile, 8 waves split K ----
template <int NB, bool LN, int EPI, int J>
__global__ __launch_bounds__(MEGA_THREADS) void proj_wg_kernel(Proj p, const void *xin, const float *ln_w, const float *ln_b, int n,
                                                                Epi e) {
    __shared__ __attribute__((aligned(16))) float lds[MEGA_XS_BYTES / 4 + 8 * 64 * 4];
    WRegs<1, J> w;
    proj_phase<NB, LN, EPI, 1, J, true>(nullptr, w, p, xin, ln_w, ln_b, n, e, lds);
}

template <int NB, int EPI, int FR>
__global__ __launch_bounds__(MEGA_THREADS) void proj_ln_tile_kernel(Proj p, const float *xin, const float *ln_w, const float *ln_b, int n,
                                                                     Epi e) {
    __shared__ __attribute__((aligned(16))) bf16_t xs[NB * (256 * LN_MAX_F4 + 8)];
    __shared__ __attribute__((aligned(16))) bf16_t wt[FR * (256 * LN_MAX_F4 + 8)];
    __shared__ __attribute__((aligned(16))) float red[(FR > 16 ? FR / 16 : 1) * 8 * 64 * 4];
    proj_ln_tile<NB, EPI, FR>(p, xin, ln_w, ln_b, n, e, xs, wt, red, nullptr);
}

template <int NB, int SPC, int CH, int FR>
__global__ __launch_bounds__(MEGA_THREADS) void proj_res_tile_kernel(Proj p, const bf16_t *xin, int n, Epi e) {
    __shared__ __attribute__((aligned(16))) bf16_t lds_t[8 * (16 * (32 * SPC + 8) + NB * (CH * 32 * SPC + 8))];
    __shared__ __attribute__((aligned(16))) float red[8 * 64 * 4];
    proj_res_tile<NB, SPC, CH, FR>(p, xin, n, e, lds_t, red, nullptr);
}

template <int NB, typename OT>
__global__ __launch_bounds__(MEGA_THREADS) void cross_attn_lean_kernel(const float *q, const bf16_t *kc, const bf16_t *vc, float *part,
                                                                        OT *out, unsigned *head_cnt, int n, int H, int d, int n_ctx,
                                                                        int splits, int per) {
    __shared__ __attribute__((aligned(16))) float lds[(16 * 72 + XK * XKD + 64 * XLDP + 16 * XLDP) / 2 + NB * XK +
                                                      2 * NB * CROSS_SPLITS_MAX + 8 * 64 * 4 + NB * 2 + 16];
    cross_attn_lean<NB, OT>(q, kc, vc, part, out, head_cnt, n, H, d, n_ctx, splits, per, lds);
}

template <int NB, typename OT>
__global__ __launch_bounds__(MEGA_THREADS) void cross_attn_mfma_kernel(const float *q, const bf16_t *kc, const bf16_t *vc, float *part,
                                                                        OT *out, unsigned *head_cnt, int n, int H, int d, int n_ctx,
                                                                        int splits, int per) {
    __shared__ __attribute__((aligned(16))) float lds[(16 * 72 + 64 * XLDP + 16 * XLDP) / 2 + NB * XK + 2 * NB * CROSS_SPLITS_MAX +
                                                      8 * 64 * 4 + NB * 2 + 16];
    cross_attn_phase_mfma<NB, OT>(q, kc, vc, part, out, head_cnt, n, H, d, n_ctx, splits, per, lds);
}

}  // namespace

void launch_proj_wg(const GemvArgs &a, hipStream_t st) {
    MW_CHECK(a.x && a.W && a.out && a.n >= 1 && a.n <= 5, "proj_wg: 1..5 rows");
    MW_CHECK(a.N % 16 == 0 && a.K % 32 == 0 && a.ldx == a.K && !a.d_rows, "proj_wg: shape");
    MW_CHECK((size_t)a.n * (a.K + 8) * 2 <= MEGA_XS_BYTES, "proj_wg: rows exceed the LDS staging area");
    const bool ln = a.ln_w != nullptr;
    const int spw = ceil_div(a.K / 32, 8);
    Proj p{(const bf16_t *)a.W, a.bias, a.N, a.K, 0};
    Epi e{};
    const dim3 grid(a.N / 16), block(MT);
    // 8-feature tiles (twice the workgroups) for projections of at most this many features (env for experiments)
    static const int ln8_max_n = getenv("MI355_WHISPER_LN8_MAX_N") ? atoi(getenv("MI355_WHISPER_LN8_MAX_N")) : 1280;
    // 32-feature workgroups (two tiles, half the LayerNorm passes) for projections of at least this many features
    static const int ln32_min_n = getenv("MI355_WHISPER_LN32_MIN_N") ? atoi(getenv("MI355_WHISPER_LN32_MIN_N")) : 4097;
#define MW_PROJ(NBv, LNv, EPIv, Jv) \
    hipLaunchKernelGGL((proj_wg_kernel<NBv, LNv, EPIv, Jv>), grid, block, 0, st, p, (const void *)a.x, a.ln_w, a.ln_b, a.n, e)
#define MW_PROJ_LN(NBv, EPIv)                                                                                                       \
    do {                                                                                                                            \
        if (a.N <= ln8_max_n)                                                                                                       \
            hipLaunchKernelGGL((proj_ln_tile_kernel<NBv, EPIv, 8>), dim3(a.N / 8), block, 0, st, p, (const float *)a.x, a.ln_w, a.ln_b, \
                               a.n, e);                                                                                             \
        else if (a.N >= ln32_min_n && a.N % 32 == 0)                                                                                \
            hipLaunchKernelGGL((proj_ln_tile_kernel<NBv, EPIv, 32>), dim3(a.N / 32), block, 0, st, p, (const float *)a.x, a.ln_w,   \
                               a.ln_b, a.n, e);                                                                                     \
        else                                                                                                                        \
            hipLaunchKernelGGL((proj_ln_tile_kernel<NBv, EPIv, 16>), grid, block, 0, st, p, (const float *)a.x, a.ln_w, a.ln_b, a.n, e); \
    } while (0)
#define MW_PROJ_NB(LNv, EPIv, Jv)                \
    do {                                          \
        if (a.n == 1) MW_PROJ(1, LNv, EPIv, Jv);  \
        else MW_PROJ(5, LNv, EPIv, Jv);           \
    } while (0)
#define MW_PROJ_LN_NB(EPIv)               \
    do {                                   \
        if (a.n == 1) MW_PROJ_LN(1, EPIv); \
        else MW_PROJ_LN(5, EPIv);          \
    } while (0)
    if (a.qkv) {
        MW_CHECK(ln && a.ln_b && spw <= 5 && a.K <= 256 * LN_MAX_F4 && a.N == 3 * a.K && a.ldo == a.K && a.kv_t == DType::BF16,
                 "proj_wg: qkv class");
        e.q = a.out; e.kc = (bf16_t *)a.kcache; e.vc = (bf16_t *)a.vcache; e.slot = a.d_slot; e.pos = a.d_pos; e.TC = a.n_text_ctx;
        e.d = a.K; e.qk_scale = a.qk_scale;
        MW_PROJ_LN_NB(EPI_QKV);
    } else if (ln && a.act == ACT_GELU) {
        MW_CHECK(a.ln_b && spw <= 5 && a.K <= 256 * LN_MAX_F4 && !a.R, "proj_wg: fc1 class");
        e.hout = (bf16_t *)a.out; e.fout = a.out; e.ldo = (int)a.ldo;
        if (a.out_bf16) MW_PROJ_LN_NB(EPI_GELU_BF16);
        else MW_PROJ_LN_NB(EPI_GELU_F32);
    } else if (ln) {
        MW_CHECK(a.ln_b && spw <= 5 && a.K <= 256 * LN_MAX_F4 && !a.R && a.ldo == a.N && a.act == ACT_NONE, "proj_wg: cross-q class");
        e.fout = a.out; e.scale = a.out_scale;
        MW_PROJ_LN_NB(EPI_SCALE);
    } else {
        MW_CHECK(a.x_bf16 && a.R && a.ldo == a.N && a.act == ACT_NONE && a.out_scale == 1.0f && spw <= 20, "proj_wg: residual class");
        e.res = a.R; e.xout = a.out;
        const int ch = a.K > 1280 ? 4 : 1, spc = a.K / (256 * ch);  // K = 8 waves x CH chunks x SPC steps x 32
        static const bool res16 = getenv("MI355_WHISPER_RES16") != nullptr;  // 16-feature tiles (default: 8)
#define MW_RES(NBv, SPCv, CHv)                                                                                                  \
    do {                                                                                                                        \
        if (res16)                                                                                                              \
            hipLaunchKernelGGL((proj_res_tile_kernel<NBv, SPCv, CHv, 16>), grid, block, 0, st, p, (const bf16_t *)a.x, a.n, e); \
        else                                                                                                                    \
            hipLaunchKernelGGL((proj_res_tile_kernel<NBv, SPCv, CHv, 8>), dim3(a.N / 8), block, 0, st, p, (const bf16_t *)a.x, a.n, e);  \
    } while (0)
#define MW_RES_SPC(SPCv)                                    \
    do {                                                     \
        if (a.n == 1 && ch == 1) MW_RES(1, SPCv, 1);         \
        else if (a.n == 1) MW_RES(1, SPCv, 4);               \
        else if (ch == 1) MW_RES(5, SPCv, 1);                \
        else MW_RES(5, SPCv, 4);                             \
    } while (0)
        if (a.K == 256 * ch * spc && spc >= 2 && spc <= 5) {
            switch (spc) {
            case 2: MW_RES_SPC(2); break;
            case 3: MW_RES_SPC(3); break;
            case 4: MW_RES_SPC(4); break;
            default: MW_RES_SPC(5); break;
            }
        } else if (spw <= 5) {
            MW_PROJ_NB(false, EPI_RES, 5);
        } else {
            MW_PROJ_NB(false, EPI_RES, 20);
        }
#undef MW_RES_SPC
#undef MW_RES
    }
#undef MW_PROJ_LN_NB
#undef MW_PROJ_NB
#undef MW_PROJ_LN
#undef MW_PROJ
    MW_HIP(hipGetLastError());
}

bool logits_tile_supported(const GemvArgs &a) {
    const int nch = a.K / LKC;
    return a.n >= 1 && a.n <= 5 && a.ln_w && a.ln_b && a.d_rows && a.K % LKC == 0 && a.ldx == a.K && !a.bias && !a.R &&
           a.act == ACT_NONE && a.out_scale == 1.0f && (nch == 3 || nch == 4 || nch == 6 || nch == 8 || nch == 10);
}

void launch_logits_tile(const GemvArgs &a, hipStream_t st) {
    MW_CHECK(logits_tile_supported(a), "logits_tile: shape");
    const dim3 grid(ceil_div(a.N, 128)), block(MT);
#define MW_LT(NBv, NCHv)                                                                                                     \
    hipLaunchKernelGGL((logits_tile_kernel<NBv, NCHv>), grid, block, 0, st, (const bf16_t *)a.W, a.x, a.d_rows, a.ln_w, a.ln_b, \
                       a.out, a.ldo, a.n, a.N)
#define MW_LT_NB(NCHv)              \
    do {                             \
        if (a.n == 1) MW_LT(1, NCHv); \
        else MW_LT(5, NCHv);          \
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
    MW_CHECK(n >= 1 && H >= 1 && d == H * 64 && n_text_ctx <= MEGA_TC_MAX, "self_attn_wg: shape");
    if (out_t == DType::F32)
        hipLaunchKernelGGL(self_attn_lean_kernel<float>, dim3(n * H), dim3(MT), 0, st, q, (const bf16_t *)kcache, (const bf16_t *)vcache,
                           d_slot, d_pos, d_kv_idx, (float *)out, H, d, n_text_ctx);
    else
        hipLaunchKernelGGL(self_attn_lean_kernel<bf16_t>, dim3(n * H), dim3(MT), 0, st, q, (const bf16_t *)kcache, (const bf16_t *)vcache,
                           d_slot, d_pos, d_kv_idx, (bf16_t *)out, H, d, n_text_ctx);
    MW_HIP(hipGetLastError());
}

void launch_cross_attn_wg(const float *q, const void *kc, const void *vc, float *partial, void *out, DType out_t, unsigned *head_cnt,
                          int n, int H, int d, int n_ctx, int splits, hipStream_t st) {
    MW_CHECK(n >= 1 && n <= 5 && H >= 1 && d == H * 64, "cross_attn_wg: 1..5 rows, 64-wide heads");
    MW_CHECK(splits >= 1 && splits <= CROSS_SPLITS_MAX, "cross_attn_wg: split count");
    const int per = ceil_div(n_ctx, splits);
    MW_CHECK(per <= MEGA_CROSS_PER_MAX, "cross_attn_wg: too many keys per split");
    const dim3 grid(H * splits), block(MT);
    static const bool no_mfma = getenv("MI355_WHISPER_NO_CROSS_MFMA") != nullptr;
    const bool mfma = per <= XK && !no_mfma;  // matrix-core variant for splits of at most 128 keys
#define MW_CROSS(NBv, OTv)                                                                                                          \
    do {                                                                                                                            \
        if (mfma)                                                                                                                   \
            hipLaunchKernelGGL((cross_attn_lean_kernel<NBv, OTv>), grid, block, 0, st, q, (const bf16_t *)kc, (const bf16_t *)vc,   \
                               partial, (OTv *)out, head_cnt, n, H, d, n_ctx, splits, per);                                        \
        else                                                                                                                        \
            hipLaunchKernelGGL((cross_attn_wg_kernel<NBv, OTv>), grid, block, 0, st, q, (const bf16_t *)kc, (const bf16_t *)vc,     \
                               partial, (OTv *)out, head_cnt, n, H, d, n_ctx, splits, per);                                        \
    } while (0)
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

int mega_barriers_per_launch(int n_layer) { return 8 * n_layer - 1; }

size_t mega_sync_bytes() { return (size_t)(SY_HEADS + 64 * 64 + 64) * sizeof(unsigned); }

bool mega_supported(int d, int H, int TC, int n_layer, int grid, std::string *why) {
    auto fail = [&](const char *m) {
        if (why) *why = m;
        return false;
    };
    if (d % 128 != 0 || d / 64 != H) return fail("d must be a multiple of 128 with 64-wide heads");
    if (d > 256 * LN_MAX_F4) return fail("d exceeds the LayerNorm row registers");
    if (ceil_div(d / 32, 8) > 5 || ceil_div(4 * d / 32, 8) > 20) return fail("too many k-steps per wave");
    if (d / 16 > grid || 3 * d / 16 > 2 * grid || 4 * d / 16 > 2 * grid) return fail("grid too small for the tile counts");
    if ((size_t)5 * (4 * d + 8) * 2 > MEGA_XS_BYTES) return fail("activation rows exceed the LDS staging area");
    if (TC > MEGA_TC_MAX) return fail("n_text_ctx exceeds MEGA_TC_MAX");
    if (n_layer > 64 || H > 64 || grid > 1024 || grid < 1) return fail("layer/head/grid count");
    return true;
}

void launch_decode_mega(const MegaArgs &a, int grid, hipStream_t st) {
    MW_CHECK(a.n >= 1 && a.n <= 5, "decode_mega: 1..5 rows");
    MW_CHECK(a.splits >= 1 && a.splits <= CROSS_SPLITS_MAX && a.per <= MEGA_CROSS_PER_MAX && a.per * a.splits >= a.n_ctx, "decode_mega: cross split");
    MW_CHECK(a.H * a.splits <= 4 * grid, "decode_mega: cross work items");
    MegaArgs args = a;
    void *params[] = {&args};
    const void *fn = a.n == 1 ? (const void *)decode_mega_kernel<1> : (const void *)decode_mega_kernel<5>;
    MW_HIP(hipLaunchCooperativeKernel(fn, dim3(grid), dim3(MT), params, 0, st));
}

int mega_max_grid(int device) {
    int per_cu1 = 0, per_cu5 = 0, cus = 0;
    MW_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu1, decode_mega_kernel<1>, MT, 0));
    MW_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu5, decode_mega_kernel<5>, MT, 0));
    MW_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device));
    if (per_cu1 < 1 || per_cu5 < 1) return 0;
    return cus;  // one workgroup per CU
}

}  // namespace mw
