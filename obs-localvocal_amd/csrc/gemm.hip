// gemm.hip — K2/K4/K5(unfused)/K6: LDS-tiled MFMA GEMM for gfx950, "NT" form
//     C[m][n] = act(alpha * sum_k A[m][k] * W[n][k] + bias[n]) + R[m][n]
// Both operands are K-contiguous (activations [rows][features], weights as stored in the GGML file
// [out][in]), so A and B fragments are the same 16-byte K-runs.  Used for the conv stem (the stem is a
// GEMM over a strided VIEW of the frame-major mel / conv1 output: row t = rows t-1,t,t+1 concatenated),
// the encoder projections and MLP, QK^T / PV of the unfused attention path, and the cross-K/V
// precompute.  Replaces the ggml mul_mat / conv_1d graph nodes whisper.cpp runs inside whisper_full
// (/root/reference/src/whisper-utils/whisper-processing.cpp:218-219).
//
// Bound: MFMA (measured: per-launch fixed cost and staging).  bf16 path (gemm_bf16_k64_kernel): 128 (or 64)
// x 128 x 64 tile, 4 waves, v_mfma_f32_16x16x32_bf16, operands staged global -> register ring -> LDS (XOR-swizzled
// 128-byte rows: conflict-free ds_read_b128 fragment reads), tiles assigned in an XCD-aware order; an LDS-direct
// (global_load_lds) variant of the same tile is kept as an experiment switch (gemm_bf16_glds_kernel).  fp32 path
// (bit-exact-token mode for tiny/base, gemm_nt_kernel): 128x128x16 tile on v_mfma_f32_32x32x2_f32 (exact fp32 fma chain),
// LDS tile stored k-major.
#include "kernels.hpp"

#include <type_traits>

#include <hip/hip_bf16.h>

#include <cmath>

namespace mw {

namespace {

using bf16x8 = __attribute__((ext_vector_type(8))) short;   // 8 bf16 = 4 VGPRs (MFMA A/B fragment)
using bf16x8b = __attribute__((ext_vector_type(8))) __bf16;
using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x16 = __attribute__((ext_vector_type(16))) float;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;

constexpr int BN = 128;  // BM (128 or 64 rows) is a template parameter: 64 doubles the tiles of narrow-N shapes
constexpr int ROW_BYTES = 64;  // K-extent of one LDS tile row: 32 bf16 or 16 f32
constexpr int F32_LD = 128 + 4; // k-major f32 tile: row length (floats) incl. padding

__device__ __forceinline__ float gelu_f(float x) {
    const float c = 0.79788456080286535587989211986876f, a = 0.044715f;
    return 0.5f * x * (1.0f + tanhf(c * x * (1.0f + a * x * x)));
}

// bf16 paths: the same function as x * sigmoid(2u), u = c x (1 + a x^2), on the hardware exp2 and reciprocal (a handful of
// instructions instead of tanhf's ~40; fc1's epilogue at M = 12000 cost 55 us of 300 with tanhf).  Relative error ~1e-6,
// far below the bf16 rounding of the result; the fp32 (token-exact) kernel keeps tanhf.
__device__ __forceinline__ float gelu_fast(float x) {
    const float c2 = 2.0f * 0.79788456080286535587989211986876f, a = 0.044715f;
    const float u2 = c2 * x * fmaf(a * x, x, 1.0f);
    // x * rcp(1 + e): __fdividef is a full IEEE division on this compiler (two v_div_scale, v_rcp, four fma, v_div_fmas,
    // v_div_fixup: 20 instructions per element instead of 9, found in the ISA of the 256 x 256 kernel's epilogue)
    return x * __builtin_amdgcn_rcpf(1.0f + __expf(-u2));
}

__device__ __forceinline__ void store_out(float *p, float v) { *p = v; }
__device__ __forceinline__ void store_out(bf16_t *p, float v) {
    __hip_bfloat16 h = __float2bfloat16(v);
    *p = *reinterpret_cast<bf16_t *>(&h);
}

// swizzled byte offset of 16-byte chunk c (0..3) of tile row r (64-byte rows)
__device__ __forceinline__ int swz(int r, int c) {
    const int g = (0x1320 >> (((r >> 2) & 3) * 4)) & 3;  // g = [0,2,3,1][(r>>2)&3]
    return r * ROW_BYTES + ((c ^ g) << 4);
}

// ---- XCD-aware tile order ----
// Workgroups are dealt round-robin over the 8 XCDs in dispatch order (ids b and b + 8 share an XCD and its 4 MiB L2), so with
// tile = blockIdx every XCD touches every row panel of A and of W: each L2 pulls both operands whole from the Infinity Cache,
// and a CU reads from there at half the rate it reads from its L2 (the M = 1500 encoder GEMMs ran at the per-CU Infinity
// Cache rate, ~27 GB/s per CU, not at an MFMA or LDS limit).  The remap gives the workgroups that share an XCD one contiguous
// range of a tile order made of super-rows of hb M-tiles, N-tile outer, M-tile inner: with hb chosen by launch_gemm a range
// is a compact hb x w rectangle of tiles (or whole batch slices), whose operand panels are fetched into that L2 once.
// Bijective for any grid and any hb >= 1; which XCD a label lands on does not matter (performance only, never correctness).
struct TileCoord {
    int tm, tn, z;
};
__device__ __forceinline__ TileCoord xcd_tile_coord(int hb) {
    const unsigned gx = gridDim.x, gy = gridDim.y;
    if (hb <= 0) return {(int)blockIdx.y, (int)blockIdx.x, (int)blockIdx.z};
    const unsigned T = gx * gy, G = T * gridDim.z;
    const unsigned id = blockIdx.x + gx * (blockIdx.y + gy * blockIdx.z);
    const unsigned xcd = id & 7, q = G >> 3, r = G & 7;
    const unsigned nl = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (id >> 3);
    const unsigned z = nl / T, lin = nl - z * T;
    const unsigned per = (unsigned)hb * gx, s = lin / per, rem = lin - s * per;
    const unsigned rows = min((unsigned)hb, gy - s * (unsigned)hb);
    const unsigned tn = rem / rows;
    return {(int)(s * (unsigned)hb + (rem - tn * rows)), (int)tn, (int)z};
}

template <typename TIN, typename TOUT, int BM>
__global__ __launch_bounds__(256) void gemm_nt_kernel(GemmArgs a) {
    constexpr bool F32 = sizeof(TIN) == 4;
    static_assert(BM == 128 || (BM == 64 && !F32), "64-row tiles: bf16 path only");
    constexpr int WJ = BM == 128 ? 4 : 2;  // 16-column MFMA tiles per wave: waves are 2 x 2 (64 x 64 each) or 1 x 4 (64 x 32 each)
    constexpr int BK = ROW_BYTES / (int)sizeof(TIN);
    // LDS: [2 buffers][A | W]; bf16: 128 rows x 64 B swizzled; f32: 16 k-rows x F32_LD floats
    constexpr int TILE_BYTES = F32 ? 16 * F32_LD * 4 : 128 * ROW_BYTES;
    __shared__ __attribute__((aligned(16))) unsigned char lds[2 * 2 * TILE_BYTES];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = BM == 128 ? wave >> 1 : 0, wc = BM == 128 ? wave & 1 : wave;
    constexpr int WCOLS = BM == 128 ? 64 : 32;  // columns per wave
    const TileCoord tc = xcd_tile_coord(a.hb);
    const int z0 = tc.z / a.nz1, z1 = tc.z % a.nz1;
    const TIN *A = (const TIN *)a.A + z0 * a.sA0 + z1 * a.sA1;
    const TIN *W = (const TIN *)a.W + z0 * a.sW0 + z1 * a.sW1;
    TOUT *C = (TOUT *)a.C + z0 * a.sC0 + z1 * a.sC1;
    const float *bias = a.bias ? a.bias + z0 * a.sB0 + z1 * a.sB1 : nullptr;
    const float *R = a.R ? a.R + z0 * a.sR0 + z1 * a.sR1 : nullptr;
    const int m0 = tc.tm * BM, n0 = tc.tn * BN;
    const int M = a.M, N = a.N, K = a.K;

    // this thread's two 16-byte chunks per operand tile: idx = tid + 256 q -> row idx>>2, chunk idx&3
    int lrow[2], lchk[2];
    const TIN *ga[2], *gw[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int idx = tid + 256 * q;
        lrow[q] = idx >> 2;
        lchk[q] = idx & 3;
        const int ra = min(m0 + lrow[q], M - 1), rw = min(n0 + lrow[q], N - 1);
        ga[q] = A + (long)ra * a.lda + lchk[q] * (16 / (int)sizeof(TIN));
        gw[q] = W + (long)rw * a.ldw + lchk[q] * (16 / (int)sizeof(TIN));
    }
    const int nk = (K + BK - 1) / BK;
    uint4 ra_[2], rw_[2];
    auto gload = [&](int kt) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int kk = kt * BK + lchk[q] * (16 / (int)sizeof(TIN));
            if (kk < K) {
                if (BM == 128 || q == 0) ra_[q] = *reinterpret_cast<const uint4 *>(ga[q] + (long)kt * BK);
                rw_[q] = *reinterpret_cast<const uint4 *>(gw[q] + (long)kt * BK);
            } else {
                ra_[q] = make_uint4(0, 0, 0, 0);
                rw_[q] = make_uint4(0, 0, 0, 0);
            }
        }
    };
    auto lstore = [&](int buf) {
        unsigned char *la = lds + buf * 2 * TILE_BYTES, *lw = la + TILE_BYTES;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            if constexpr (F32) {
                float *fa = (float *)la, *fw = (float *)lw;
                const int kb = lchk[q] * 4;
                fa[(kb + 0) * F32_LD + lrow[q]] = __uint_as_float(ra_[q].x);
                fa[(kb + 1) * F32_LD + lrow[q]] = __uint_as_float(ra_[q].y);
                fa[(kb + 2) * F32_LD + lrow[q]] = __uint_as_float(ra_[q].z);
                fa[(kb + 3) * F32_LD + lrow[q]] = __uint_as_float(ra_[q].w);
                fw[(kb + 0) * F32_LD + lrow[q]] = __uint_as_float(rw_[q].x);
                fw[(kb + 1) * F32_LD + lrow[q]] = __uint_as_float(rw_[q].y);
                fw[(kb + 2) * F32_LD + lrow[q]] = __uint_as_float(rw_[q].z);
                fw[(kb + 3) * F32_LD + lrow[q]] = __uint_as_float(rw_[q].w);
            } else {
                if (BM == 128 || q == 0) *reinterpret_cast<uint4 *>(la + swz(lrow[q], lchk[q])) = ra_[q];
                *reinterpret_cast<uint4 *>(lw + swz(lrow[q], lchk[q])) = rw_[q];
            }
        }
    };

    // accumulators: bf16 path 4x4 tiles of 16x16 (f32x4); fp32 path 2x2 tiles of 32x32 (f32x16)
    f32x4 acc16[F32 ? 1 : 4][F32 ? 1 : WJ];
    f32x16 acc32[F32 ? 2 : 1][F32 ? 2 : 1];
    if constexpr (F32) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc32[i][j][r] = 0.0f;
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < WJ; ++j) acc16[i][j] = f32x4{0, 0, 0, 0};
    }

    gload(0);
    lstore(0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) gload(kt + 1);
        const unsigned char *la = lds + cur * 2 * TILE_BYTES, *lw = la + TILE_BYTES;
        if constexpr (F32) {
            const float *fa = (const float *)la, *fw = (const float *)lw;
            const int li = lane & 31, lk = lane >> 5;
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                float av[2], bv[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    av[i] = fa[(ks * 2 + lk) * F32_LD + wr * 64 + i * 32 + li];
                    bv[i] = fw[(ks * 2 + lk) * F32_LD + wc * 64 + i * 32 + li];
                }
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc32[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[j], acc32[i][j], 0, 0, 0);
            }
        } else {
            const int fr = lane & 15, fq = lane >> 4;
            bf16x8 af[4], bfr[WJ];
#pragma unroll
            for (int i = 0; i < 4; ++i) af[i] = *reinterpret_cast<const bf16x8 *>(la + swz(wr * 64 + i * 16 + fr, fq));
#pragma unroll
            for (int j = 0; j < WJ; ++j) bfr[j] = *reinterpret_cast<const bf16x8 *>(lw + swz(wc * WCOLS + j * 16 + fr, fq));
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < WJ; ++j)
                    acc16[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                        __builtin_bit_cast(bf16x8b, af[i]), __builtin_bit_cast(bf16x8b, bfr[j]), acc16[i][j], 0, 0, 0);
        }
        if (kt + 1 < nk) {
            lstore(cur ^ 1);  // the other buffer was last read before the previous barrier
            __syncthreads();
        }
    }

    // epilogue
    auto emit = [&](int m, int n, float v) {
        if (m >= M || n >= N) return;
        v *= a.alpha;
        if (bias) v += bias[n];
        if (a.act == ACT_GELU) v = gelu_f(v);
        if (R) v += R[(long)m * a.ldr + n];
        if (a.trans_c) store_out(C + (long)n * a.ldc + m, v);
        else store_out(C + (long)m * a.ldc + n, v);
    };
    if constexpr (F32) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                    emit(m0 + wr * 64 + i * 32 + row, n0 + wc * 64 + j * 32 + (lane & 31), acc32[i][j][r]);
                }
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < WJ; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    emit(m0 + wr * 64 + i * 16 + (lane >> 4) * 4 + r, n0 + wc * WCOLS + j * 16 + (lane & 15), acc16[i][j][r]);
    }
}

// epilogue shared by the bf16 kernels: one vector store of 4 consecutive output elements per MFMA tile and lane.
// Its loads are vectors too: the bias of a lane's 4 columns is one 16-byte load per column tile (hoisted out of the row
// tiles), the residual one 16-byte load per tile.  As 4 + 4 clamped dword loads per tile (64-128 tiles' worth of dword
// loads per lane, issued by every wave of the CU at the same moment) they were what a tile spent its time on outside the
// k-loop: ~20 us per round of 256 x 256 tiles, no matter whether anything was stored.
template <typename TOUT, int BM, bool TR, int WJ>
__device__ __forceinline__ void gemm_bf16_epilogue(const GemmArgs &a, f32x4 (&acc)[4][WJ], TOUT *C, const float *bias, const float *R,
                                                   int m0, int n0, int wr, int wc, int lane) {
    constexpr int WCOLS = BM == 128 ? 64 : 32;
    const int M = a.M, N = a.N;
    if (!TR && a.epi_vec) {
        // vector form (launch_gemm checked N % 4 == 0 and the 16-byte alignment of bias, residual and output rows): every
        // load is unconditional — out-of-range lanes read a clamped address — because a branch around a load ends in a wait
        // for ALL loads in flight at its join, the next tile's operand loads included; only the store is predicated
        // Tiles that lie wholly inside the matrix (all but the last row of tiles at M = 1500) take the FULL form: no store
        // predicate at all.  With the predicate the compiler sinks the conversions under it and opens every predicated block
        // with a wait for whatever is in flight (found in the 256 x 256 kernel's listing: one store acknowledgement per block).
        auto run = [&](auto full_tag) {
        constexpr bool FULL = decltype(full_tag)::value;
        float bj[WJ][4];
#pragma unroll
        for (int j = 0; j < WJ; ++j) {
            const int nb = min(n0 + wc * WCOLS + j * 16 + (lane >> 4) * 4, N - 4);
            const float4 b4 = *reinterpret_cast<const float4 *>((bias ? bias : a.epi_zero) + (bias ? nb : 0));
            bj[j][0] = b4.x; bj[j][1] = b4.y; bj[j][2] = b4.z; bj[j][3] = b4.w;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            // one row tile at a time: left alone the scheduler hoists every residual load of the epilogue to its top
            __builtin_amdgcn_sched_barrier(0);
            const int mb = m0 + wr * 64 + i * 16 + (lane & 15), mc = FULL ? mb : min(mb, M - 1);
            float4 rv[WJ];
#pragma unroll
            for (int j = 0; j < WJ; ++j) {
                const int nb = min(n0 + wc * WCOLS + j * 16 + (lane >> 4) * 4, N - 4);
                rv[j] = R ? *reinterpret_cast<const float4 *>(R + (long)mc * a.ldr + nb) : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            }
#pragma unroll
            for (int j = 0; j < WJ; ++j) {
                const int nb = n0 + wc * WCOLS + j * 16 + (lane >> 4) * 4;
                const float rr[4] = {rv[j].x, rv[j].y, rv[j].z, rv[j].w};
                float v[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float x = acc[i][j][r] * a.alpha;
                    if (bias) x += bj[j][r];
                    if (a.act == ACT_GELU) x = gelu_fast(x);
                    if (R) x += rr[r];
                    v[r] = x;
                }
                TOUT *dst = C + (long)mc * a.ldc + (FULL ? nb : min(nb, N - 4));
                if (FULL || (mb < M && nb < N)) {
                    if constexpr (sizeof(TOUT) == 4) {
                        *reinterpret_cast<float4 *>(dst) = make_float4(v[0], v[1], v[2], v[3]);
                    } else {
                        __hip_bfloat16 h0 = __float2bfloat16(v[0]), h1 = __float2bfloat16(v[1]), h2 = __float2bfloat16(v[2]),
                                       h3 = __float2bfloat16(v[3]);
                        uint2 pk;
                        pk.x = (unsigned)*reinterpret_cast<bf16_t *>(&h0) | ((unsigned)*reinterpret_cast<bf16_t *>(&h1) << 16);
                        pk.y = (unsigned)*reinterpret_cast<bf16_t *>(&h2) | ((unsigned)*reinterpret_cast<bf16_t *>(&h3) << 16);
                        *reinterpret_cast<uint2 *>(dst) = pk;
                    }
                }
            }
        }
        };
        if (m0 + BM <= M && n0 + BN <= N) run(std::true_type{});
        else run(std::false_type{});
        return;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < WJ; ++j) {
            // TR: lane = (n = lane & 15, m = 4*(lane >> 4) + r); else lane = (m = lane & 15, n = 4*(lane >> 4) + r)
            const int mb = m0 + wr * 64 + i * 16 + (TR ? (lane >> 4) * 4 : (lane & 15));
            const int nb = n0 + wc * WCOLS + j * 16 + (TR ? (lane & 15) : (lane >> 4) * 4);
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = TR ? mb + r : mb, n = TR ? nb : nb + r;
                float x = acc[i][j][r] * a.alpha;
                if (bias) x += bias[min(n, N - 1)];
                if (a.act == ACT_GELU) x = gelu_fast(x);
                if (R) x += R[(long)min(m, M - 1) * a.ldr + min(n, N - 1)];
                v[r] = x;
            }
            TOUT *dst = TR ? C + (long)nb * a.ldc + mb : C + (long)mb * a.ldc + nb;
            const bool full = TR ? (nb < N && mb + 3 < M) : (mb < M && nb + 3 < N);
            if (full && (((uintptr_t)dst) % (4 * sizeof(TOUT))) == 0) {
                if constexpr (sizeof(TOUT) == 4) {
                    *reinterpret_cast<float4 *>(dst) = make_float4(v[0], v[1], v[2], v[3]);
                } else {
                    __hip_bfloat16 h0 = __float2bfloat16(v[0]), h1 = __float2bfloat16(v[1]), h2 = __float2bfloat16(v[2]),
                                   h3 = __float2bfloat16(v[3]);
                    uint2 pk;
                    pk.x = (unsigned)*reinterpret_cast<bf16_t *>(&h0) | ((unsigned)*reinterpret_cast<bf16_t *>(&h1) << 16);
                    pk.y = (unsigned)*reinterpret_cast<bf16_t *>(&h2) | ((unsigned)*reinterpret_cast<bf16_t *>(&h3) << 16);
                    *reinterpret_cast<uint2 *>(dst) = pk;
                }
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int m = TR ? mb + r : mb, n = TR ? nb : nb + r;
                    if (m < M && n < N) store_out(TR ? C + (long)n * a.ldc + m : C + (long)m * a.ldc + n, v[r]);
                }
            }
        }
}

// ---- bf16 path with 64-deep k-steps ----
// Same tiling as gemm_nt_kernel<bf16> (128 or 64 x 128 tile, 4 waves, 16x16x32 MFMA) with twice the k-extent per
// iteration: the loop is bound by the latency of the next tile's global loads (one barrier and one wait per
// iteration), so halving the iterations nearly halves the time.  LDS rows are 128 bytes, 16-byte chunk c of row r
// lives at chunk c ^ (r & 7); every load is unconditional (clamped address, zero selected afterwards).
//
// PIN: the ring's loads are pinned to the top of the k-step (scheduling barrier behind them).  Pinned, three steps of
// loads are really in flight, at the price of ~280 registers (one workgroup per CU): the choice for grids of at most one
// tile per CU (the encoder's M = 1500 shapes).  Unpinned, hipcc moves the loads behind the step's products and keeps two
// workgroups per CU: better when a CU has several tiles to overlap (cross-K/V over 32 layers, fc1).
template <typename TOUT, int BM, bool TR, bool PIN>
__global__ __launch_bounds__(256) void gemm_bf16_k64_kernel(GemmArgs a) {
    constexpr int BK = 64, RB = 128, TILE = 128 * RB;
    constexpr int WJ = BM == 128 ? 4 : 2, WCOLS = BM == 128 ? 64 : 32, QA = BM == 128 ? 4 : 2;
    __shared__ __attribute__((aligned(16))) unsigned char lds[2 * 2 * TILE];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = BM == 128 ? wave >> 1 : 0, wc = BM == 128 ? wave & 1 : wave;
    const TileCoord tc = xcd_tile_coord(a.hb);
    const int z0 = tc.z / a.nz1, z1 = tc.z % a.nz1;
    const bf16_t *A = (const bf16_t *)a.A + z0 * a.sA0 + z1 * a.sA1;
    const bf16_t *W = (const bf16_t *)a.W + z0 * a.sW0 + z1 * a.sW1;
    TOUT *C = (TOUT *)a.C + z0 * a.sC0 + z1 * a.sC1;
    const float *bias = a.bias ? a.bias + z0 * a.sB0 + z1 * a.sB1 : nullptr;
    const float *R = a.R ? a.R + z0 * a.sR0 + z1 * a.sR1 : nullptr;
    const int m0 = tc.tm * BM, n0 = tc.tn * BN;
    const int M = a.M, N = a.N, K = a.K;
    auto swz8 = [](int r, int c) { return r * RB + ((c ^ (r & 7)) << 4); };
    // this thread's 16-byte chunks: idx = tid + 256 q -> row idx >> 3, chunk idx & 7
    const bf16_t *ga[QA], *gw[4];
    int lo[4];  // LDS offset of chunk q (same for both operands)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int idx = tid + 256 * q, row = idx >> 3, chk = idx & 7;
        lo[q] = swz8(row, chk);
        if (q < QA) ga[q] = A + (long)min(m0 + row, M - 1) * a.lda + chk * 8;
        gw[q] = W + (long)min(n0 + row, N - 1) * a.ldw + chk * 8;
    }
    const int kchk = (tid & 7) * 8;  // k offset of this thread's chunks inside a k-step
    const int nk = (K + BK - 1) / BK;
    // Register ring of NS k-steps: the loads of step kt + NS - 1 are issued while step kt is multiplied, so a load has
    // NS - 2 iterations (not one) to arrive before its tile goes to LDS.  The k-loop is unrolled by NS so that every
    // ring index is a constant; steps beyond nk reload the last tile (clamped) and store zeros: no branch in the loop.
    constexpr int NS = PIN ? 4 : 2;  // unpinned variant: one step ahead, pinned too (see PIN)
    // Every load is unconditional (clamped address); a chunk beyond K or a step beyond nk is zeroed by an AND with the ring
    // slot's mask when the tile goes to LDS.  A select on the loaded value made hipcc sink the loads under an exec-masked
    // branch, and behind the joins of those branches it waited for vmcnt(0) in every k-step: the ring never had more than
    // one step in flight and a k-step took a full load latency (~1 us for 0.2 us of matrix work).  The mask passes
    // through an empty asm so that it stays an opaque register and the AND is not folded back into a select.
    // (The loads stay compiler-visible on purpose: as inline asm with hand-counted vmcnt, hipcc parked ring registers in
    // AGPRs across the back-edge, copying registers whose loads were still in flight: wrong tiles under memory load.)
    u32x4 ra[NS][QA], rw[NS][4];
    unsigned rm[NS];
    auto gload = [&](int kt, u32x4(&pa)[QA], u32x4(&pw)[4], unsigned &pm) {
        const int kc = min(kt, nk - 1);
        const bool ok = kt < nk && kc * BK + kchk < K;
        const long off = kc * BK + kchk < K ? (long)kc * BK : 0;
        unsigned m = ok ? 0xffffffffu : 0u;
        asm volatile("" : "+v"(m));
        pm = m;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (q < QA) pa[q] = *reinterpret_cast<const u32x4 *>(ga[q] + off);
            pw[q] = *reinterpret_cast<const u32x4 *>(gw[q] + off);
        }
    };
    auto lstore = [&](int buf, const u32x4(&pa)[QA], const u32x4(&pw)[4], unsigned m) {
        unsigned char *la = lds + buf * 2 * TILE, *lw = la + TILE;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (q < QA) *reinterpret_cast<u32x4 *>(la + lo[q]) = pa[q] & m;
            *reinterpret_cast<u32x4 *>(lw + lo[q]) = pw[q] & m;
        }
    };
    f32x4 acc[4][WJ];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < WJ; ++j) acc[i][j] = f32x4{0, 0, 0, 0};
#pragma unroll
    for (int u = 0; u < NS - 1; ++u) gload(u, ra[u], rw[u], rm[u]);
    lstore(0, ra[0], rw[0], rm[0]);
    __syncthreads();
    const int fr = lane & 15, fq = lane >> 4;
    for (int kt0 = 0; kt0 < nk; kt0 += NS) {
#pragma unroll
        for (int u = 0; u < NS; ++u) {
            const int kt = kt0 + u, cur = u & 1;  // NS is even: the LDS buffer parity of step kt is that of u
            gload(kt + NS - 1, ra[(u + NS - 1) % NS], rw[(u + NS - 1) % NS], rm[(u + NS - 1) % NS]);
            // the loads are issued HERE, NS - 1 steps ahead of their use: left alone, the scheduler moved them behind the
            // products of the step (shorter live ranges) and the ring was one step deep again
            __builtin_amdgcn_sched_barrier(0);
            const unsigned char *la = lds + cur * 2 * TILE, *lw = la + TILE;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                bf16x8 af[4], bfr[WJ];
#pragma unroll
                for (int i = 0; i < 4; ++i) af[i] = *reinterpret_cast<const bf16x8 *>(la + swz8(wr * 64 + i * 16 + fr, 4 * ks + fq));
#pragma unroll
                for (int j = 0; j < WJ; ++j)
                    bfr[j] = *reinterpret_cast<const bf16x8 *>(lw + swz8(wc * WCOLS + j * 16 + fr, 4 * ks + fq));
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < WJ; ++j)
                        // the operand that is contiguous in the OUTPUT goes first: a lane then holds 4 consecutive output
                        // elements (C row-major: consecutive n -> W first; transposed store: consecutive m -> A first)
                        acc[i][j] = TR ? __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8b, af[i]),
                                                                                 __builtin_bit_cast(bf16x8b, bfr[j]), acc[i][j], 0, 0, 0)
                                       : __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8b, bfr[j]),
                                                                                 __builtin_bit_cast(bf16x8b, af[i]), acc[i][j], 0, 0, 0);
            }
            // tile kt + 1 (zeros beyond nk) -> the other buffer, last read before the previous barrier
            lstore(cur ^ 1, ra[(u + 1) % NS], rw[(u + 1) % NS], rm[(u + 1) % NS]);
            // LDS-only barrier: __syncthreads() would also wait for the NS - 2 tiles of loads still in flight
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        }
    }
    gemm_bf16_epilogue<TOUT, BM, TR, WJ>(a, acc, C, bias, R, m0, n0, wr, wc, lane);
}

// ---- bf16 path, operands loaded straight into LDS (global_load_lds, 16 bytes per lane) ----
// Same tile, fragment reads, MFMA order and epilogue as gemm_bf16_k64_kernel; what changes is the staging: no register
// ring, no ds_write pass (79 B/clk against 256 B/clk for the reads), no masks.  A wave-instruction of the LDS DMA writes
// 64 x 16 bytes to a wave-uniform base + lane * 16, i.e. 8 tile rows of 128 bytes in row order; the XOR swizzle the
// fragment reads expect (chunk c of row r at slot c ^ (r & 7)) is therefore applied to the SOURCE address: the lane that
// fills slot s of row r loads global chunk s ^ (r & 7) — still whole 128-byte lines per 8 lanes.
// Pipeline: GS LDS stages, the loads of step kt + GS - 1 are issued right after the barrier of step kt (every wave has
// then finished reading that stage in step kt - 1); a step's data are complete for the whole workgroup after each
// wave's counted `s_waitcnt vmcnt` (the DMAs of the two younger steps stay in flight) followed by that barrier — one
// barrier per k-step, a raw s_barrier (__syncthreads() would drain the DMAs).  Steps beyond the last reload the last
// tile into a free stage so that the count is the same in every iteration.  Needs K % 64 == 0 (no zero-fill of a tail).
//
// GS = 1 is the kernel of the many-tile grids (encoder pass over the windows of several streams, M = windows x 1500): one
// 32 KB stage, two barriers per k-step, registers capped at 128 — FOUR workgroups per CU, which cover each other's load
// latency and, above all, each other's prologue and epilogue: with one or two workgroups per CU every CU runs its tiles in
// lockstep (all load, then all multiply, then all store) and a K = 1280 tile spent 15 of its 33 us outside the k-loop.
// Measured at M = 12000 (tools/microbench/mb_gemm): Q|K 143 -> 106 us (739 TFLOP/s), fc1 324 -> 222, fc2 237 -> 196,
// attention output 102 -> 79 against the register-ring kernel; 256 x 128 tiles and 2- or 3-stage rings measured no better.
template <typename TOUT, int BM, bool TR, int GS>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(GS == 1 ? 4 : 1, GS == 1 ? 4 : 2))) void gemm_bf16_glds_kernel(GemmArgs a) {
    constexpr int BK = 64, RB = 128, TILE_A = BM * RB, TILE_W = 128 * RB, STAGE = TILE_A + TILE_W;
    constexpr int WJ = BM == 128 ? 4 : 2, WCOLS = BM == 128 ? 64 : 32, QA = BM == 128 ? 4 : 2, LPS = QA + 4;
    __shared__ __attribute__((aligned(1024))) unsigned char lds[GS * STAGE];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = BM == 128 ? wave >> 1 : 0, wc = BM == 128 ? wave & 1 : wave;
    const TileCoord tc = xcd_tile_coord(a.hb);
    const int z0 = tc.z / a.nz1, z1 = tc.z % a.nz1;
    const bf16_t *A = (const bf16_t *)a.A + z0 * a.sA0 + z1 * a.sA1;
    const bf16_t *W = (const bf16_t *)a.W + z0 * a.sW0 + z1 * a.sW1;
    TOUT *C = (TOUT *)a.C + z0 * a.sC0 + z1 * a.sC1;
    const float *bias = a.bias ? a.bias + z0 * a.sB0 + z1 * a.sB1 : nullptr;
    const float *R = a.R ? a.R + z0 * a.sR0 + z1 * a.sR1 : nullptr;
    const int m0 = tc.tm * BM, n0 = tc.tn * BN;
    const int M = a.M, N = a.N, K = a.K;
    auto swz8 = [](int r, int c) { return r * RB + ((c ^ (r & 7)) << 4); };
    // piece q of this wave: tile rows 8 * wave + 32 * q .. + 7; lane -> row lane >> 3, slot lane & 7, global chunk slot ^ (row & 7)
    const bf16_t *ga[QA], *gw[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int row = 8 * wave + 32 * q + (lane >> 3), chk = (lane & 7) ^ (row & 7);
        if (q < QA) ga[q] = A + (long)min(m0 + row, M - 1) * a.lda + chk * 8;
        gw[q] = W + (long)min(n0 + row, N - 1) * a.ldw + chk * 8;
    }
    const int nk = K / BK;
    using gptr_t = const __attribute__((address_space(1))) void *;
    using lptr_t = __attribute__((address_space(3))) void *;
    auto issue = [&](int kt, int stage) {
        const long off = (long)min(kt, nk - 1) * BK;
        unsigned char *sa = lds + stage * STAGE + (8 * wave) * RB, *sw = sa + TILE_A;  // wave-uniform
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (q < QA) __builtin_amdgcn_global_load_lds((gptr_t)(ga[q] + off), (lptr_t)(sa + 32 * q * RB), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((gptr_t)(gw[q] + off), (lptr_t)(sw + 32 * q * RB), 16, 0, 0);
        }
    };
    f32x4 acc[4][WJ];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < WJ; ++j) acc[i][j] = f32x4{0, 0, 0, 0};
#pragma unroll
    for (int u = 0; u < GS - 1; ++u) issue(u, u);
    const int fr = lane & 15, fq = lane >> 4;
    for (int kt = 0; kt < nk; ++kt) {
        if constexpr (GS == 1) {
            // one stage, two barriers per k-step: the latency of a step's loads is covered by the other workgroups of the CU
            // (32 KB of LDS and <= 128 registers: four of them)
            if (kt > 0) __builtin_amdgcn_s_barrier();  // every wave has read step kt - 1
            issue(kt, 0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        } else {
        // this wave's DMAs of step kt are done when at most the 2 * LPS younger ones are outstanding ...
        static_assert(GS >= 1 && GS <= 4, "stages");
        if constexpr ((GS - 2) * LPS == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        else if constexpr ((GS - 2) * LPS == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        else if constexpr ((GS - 2) * LPS == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if constexpr ((GS - 2) * LPS == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();  // ... and every wave's are, behind the barrier; stage (kt - 1) % GS is free again
        issue(kt + GS - 1, (kt + GS - 1) % GS);
        }
        const unsigned char *la = lds + (kt % GS) * STAGE, *lw = la + TILE_A;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 af[4], bfr[WJ];
#pragma unroll
            for (int i = 0; i < 4; ++i) af[i] = *reinterpret_cast<const bf16x8 *>(la + swz8(wr * 64 + i * 16 + fr, 4 * ks + fq));
#pragma unroll
            for (int j = 0; j < WJ; ++j) bfr[j] = *reinterpret_cast<const bf16x8 *>(lw + swz8(wc * WCOLS + j * 16 + fr, 4 * ks + fq));
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < WJ; ++j)
                    acc[i][j] = TR ? __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8b, af[i]),
                                                                             __builtin_bit_cast(bf16x8b, bfr[j]), acc[i][j], 0, 0, 0)
                                   : __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8b, bfr[j]),
                                                                             __builtin_bit_cast(bf16x8b, af[i]), acc[i][j], 0, 0, 0);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the reloads of the last tile: nothing may land in LDS after the kernel's waves move on
    gemm_bf16_epilogue<TOUT, BM, TR, WJ>(a, acc, C, bias, R, m0, n0, wr, wc, lane);
}

// ---- bf16 path for the many-window encoder pass: 256 x 256 tile, persistent workgroups ----
// 8 waves as 2 (M) x 4 (N), each 128 x 64 of the output (acc = 128 registers): a k-step moves 64 KB for 8.4 MFLOP, twice the
// flops per byte of the 128 x 128 tile, and a wave reads 24 fragments for 64 MFMAs instead of 16 for 32; the k-loop runs at
// ~1.4 PFLOP/s (measured from two K extents of one grid).  LDS: two k-step buffers of four 16 KB half-tiles (A rows 0-127 |
// A rows 128-255 | W rows 0-127 | W rows 128-255), the 128-byte swizzled rows and fragment reads of the kernels above,
// filled by global_load_lds; the loads of step kt + 1 are issued behind the barrier of step kt (A before the first half of
// the products, W before the second) and waited for at the top of step kt + 1.  One workgroup per CU.
// With one workgroup per CU nothing hides a tile's prologue and epilogue (~20 us per round of tiles, as much as a K = 1280
// k-loop): the workgroups are PERSISTENT — min(tiles, 256) of them, each walking tiles of an XCD-contiguous range — and the
// first k-step of a workgroup's next tile is issued before the epilogue of the current one, so the stores of tile i and the
// cold loads of tile i + 1 overlap.  Batch members (GemmArgs::nz0 x nz1) are further tiles of the same walk; needs
// K % 64 == 0, N % 256 == 0; no transposed store.
// The epilogue's shape is a compile-time choice (HAS_R: fp32 residual added, GELU: tanh-GELU): as run-time branches around
// its loads and stores every slice ended in a wait at the branch's join.
template <typename TOUT, bool HAS_R, bool GELU>
__global__ __launch_bounds__(512) void gemm_bf16_256_kernel(GemmArgs a) {
    constexpr int BK = 64, RB = 128, HALF = 128 * RB, STAGE = 4 * HALF;
    __shared__ __attribute__((aligned(1024))) unsigned char lds[2 * STAGE];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 2, wc = wave & 3;
    const bf16_t *A = (const bf16_t *)a.A, *W = (const bf16_t *)a.W;
    TOUT *C = (TOUT *)a.C;
    const int M = a.M, N = a.N, K = a.K;
    const int tiles_n = N / 256, tiles_m = (M + 255) / 256, tiles_z = tiles_m * tiles_n, ntiles = tiles_z * a.nz0 * a.nz1, per_xcd = (ntiles + 7) / 8;
    auto swz8 = [](int r, int c) { return r * RB + ((c ^ (r & 7)) << 4); };
    // tile of round r: workgroups that share an XCD (ids equal mod 8) walk one contiguous range of the tile order (N-tile
    // fastest, batch member slowest), so what an XCD's workgroups hold at one time is a compact block of A panels x all W
    // panels of one or two batch members
    auto tile_of = [&](int r, int &m0, int &n0, int &z) -> bool {
        const int L = blockIdx.x + r * gridDim.x, j = L >> 3, p = (L & 7) * per_xcd + j;
        if (j >= per_xcd || p >= ntiles) return false;
        z = p / tiles_z;
        const int q = p - z * tiles_z;
        m0 = (q / tiles_n) * 256;
        n0 = (q % tiles_n) * 256;
        return true;
    };
    // piece q of this wave in a half-tile: rows 8 * wave + 64 * q .. + 7; lane -> row lane >> 3, slot lane & 7, global chunk slot ^ (row & 7)
    const bf16_t *ga[2][2], *gw[2][2];
    auto point = [&](int m0, int n0, int z) {
        const int z0 = z / a.nz1, z1 = z - z0 * a.nz1;
        const bf16_t *Az = A + z0 * a.sA0 + z1 * a.sA1, *Wz = W + z0 * a.sW0 + z1 * a.sW1;
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int row = 8 * wave + 64 * q + (lane >> 3), chk = (lane & 7) ^ (row & 7);
                ga[h][q] = Az + (long)min(m0 + 128 * h + row, M - 1) * a.lda + chk * 8;
                gw[h][q] = Wz + (long)min(n0 + 128 * h + row, N - 1) * a.ldw + chk * 8;
            }
    };
    const int nk = K / BK;
    using gptr_t = const __attribute__((address_space(1))) void *;
    using lptr_t = __attribute__((address_space(3))) void *;
    auto issue_a = [&](int kt, int stage) {
        const long off = (long)kt * BK;
        unsigned char *sa = lds + stage * STAGE + (8 * wave) * RB;  // wave-uniform
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int q = 0; q < 2; ++q)
                __builtin_amdgcn_global_load_lds((gptr_t)(ga[h][q] + off), (lptr_t)(sa + h * HALF + 64 * q * RB), 16, 0, 0);
    };
    auto issue_w = [&](int kt, int stage) {
        const long off = (long)kt * BK;
        unsigned char *sw = lds + stage * STAGE + 2 * HALF + (8 * wave) * RB;
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int q = 0; q < 2; ++q)
                __builtin_amdgcn_global_load_lds((gptr_t)(gw[h][q] + off), (lptr_t)(sw + h * HALF + 64 * q * RB), 16, 0, 0);
    };
    const int fr = lane & 15, fq = lane >> 4;
    int m0 = 0, n0 = 0, zz = 0, round = 0;
    bool have = tile_of(0, m0, n0, zz);  // workgroup-uniform throughout
    if (have) {
        point(m0, n0, zz);
        issue_a(0, 0);
        issue_w(0, 0);
    }
    int par = 0;  // LDS buffer of the step about to be multiplied (runs on across tiles)
    while (have) {
        f32x4 acc[8][4];
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0, 0, 0, 0};
        int m1 = 0, n1 = 0, z1n = 0;
        const bool have_next = tile_of(round + 1, m1, n1, z1n);
        for (int kt = 0; kt < nk; ++kt) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();  // this step is in LDS for every wave; every wave is done with the other buffer
            const unsigned char *la = lds + par * STAGE + wr * HALF, *lw = lds + par * STAGE + (2 + (wc >> 1)) * HALF;
            const int wcol = (wc & 1) * 64;
            const bool last = kt + 1 == nk;
            if (last && have_next) point(m1, n1, z1n);  // the loads below then fetch step 0 of the next tile
            const bool more = !last || have_next;
            const int nkt = last ? 0 : kt + 1;
            if (more) issue_a(nkt, par ^ 1);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                bf16x8 bfr[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) bfr[j] = *reinterpret_cast<const bf16x8 *>(lw + swz8(wcol + j * 16 + fr, 4 * ks + fq));
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) {
                    bf16x8 af[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) af[i] = *reinterpret_cast<const bf16x8 *>(la + swz8(hh * 64 + i * 16 + fr, 4 * ks + fq));
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            acc[hh * 4 + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                                __builtin_bit_cast(bf16x8b, bfr[j]), __builtin_bit_cast(bf16x8b, af[i]), acc[hh * 4 + i][j], 0, 0, 0);
                }
                if (ks == 0 && more) issue_w(nkt, par ^ 1);
            }
            par ^= 1;
        }
        // Epilogue of this tile while step 0 of the next one is in flight.  Written straight from the MFMA layout the tile went
        // out as 32-byte pieces of 16 different rows per store instruction, all 256 CUs at the same moment: 16-17 us per tile
        // in which the matrix cores idle.  Here every 16-row slice of the wave's 128 x 64 block is turned through a
        // wave-private corner of the k-step buffer that was read last (free behind the barrier below; LDS operations of
        // one wave execute in order, so no further barrier), and leaves as whole lines: 16 lanes cover one row's 64 columns
        // (256 bytes of fp32 / 128 of bf16), the residual is read the same way, the lane's four bias values once.
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // every wave has read the last k-step
        {
            constexpr int SROW = 64 + 4;  // floats per staged row (padding: conflict-free 16-byte column accesses)
            float *stg = reinterpret_cast<float *>(lds + (par ^ 1) * STAGE + wave * (16 * SROW * 4 + 256));
            const int ncol0 = n0 + wc * 64;
            const int zb0 = zz / a.nz1, zb1 = zz - zb0 * a.nz1;  // this tile's batch member
            TOUT *const Cz = C + zb0 * a.sC0 + zb1 * a.sC1;
            const float *const bias_z = a.bias ? a.bias + zb0 * a.sB0 + zb1 * a.sB1 : nullptr;
            const float *const R_z = HAS_R ? a.R + zb0 * a.sR0 + zb1 * a.sR1 : nullptr;
            // A row predicate around the stores made the compiler sink each slice's work under it and open every one of the
            // 16 predicated blocks with s_waitcnt vmcnt(0): every store waited for the acknowledgement of the one before
            // (and the first for the next tile's DMA).  Only the last row of tiles can be partial (M = windows x 1500), so the
            // choice is made once per tile, workgroup-uniform: FULL tiles run straight-line code without any predicate.
            auto epilogue = [&](auto full_tag) {
            constexpr bool FULL = decltype(full_tag)::value;
            if constexpr (sizeof(TOUT) == 4) {
                // fp32 out (the residual stream): a lane takes 4 columns, 16 lanes one row's 64 columns (256 bytes), 4 rows per
                // instruction; the residual is read the same way.  The row predicate stays on these stores ON PURPOSE: the
                // compiler then waits for each store's acknowledgement before the next slice's loads, and that throttle is
                // FASTER here — the straight-line form (residual rows prefetched a slice ahead, 32 stores in flight per wave)
                // measured 109 vs 92 us (attention output) and 290 vs 273 us (fc2) at M = 24 000: all 256 CUs read and write
                // the same fp32 stream at once, and interleaved bursts of both cost more than they hide.
                const int erow = lane >> 4, ecol = 4 * (lane & 15), nb = ncol0 + ecol;
                const float4 b4 = *reinterpret_cast<const float4 *>(bias_z ? bias_z + nb : a.epi_zero);
#pragma unroll
                for (int i8 = 0; i8 < 8; ++i8) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        f32x4 v = acc[i8][j];
                        v *= a.alpha;
                        *reinterpret_cast<f32x4 *>(stg + (lane & 15) * SROW + 16 * j + 4 * (lane >> 4)) = v;
                    }
                    const int mrow0 = m0 + wr * 128 + i8 * 16;
                    float4 rv[4];
#pragma unroll
                    for (int it = 0; it < 4; ++it) {
                        const int m = min(mrow0 + erow + 4 * it, M - 1);
                        if constexpr (HAS_R) rv[it] = *reinterpret_cast<const float4 *>(R_z + (long)m * a.ldr + nb);
                        else rv[it] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                    }
#pragma unroll
                    for (int it = 0; it < 4; ++it) {
                        const int mr = mrow0 + erow + 4 * it;
                        const f32x4 t = *reinterpret_cast<const f32x4 *>(stg + (erow + 4 * it) * SROW + ecol);
                        float x[4] = {t[0] + b4.x, t[1] + b4.y, t[2] + b4.z, t[3] + b4.w};
                        if constexpr (GELU) {
#pragma unroll
                            for (int r = 0; r < 4; ++r) x[r] = gelu_fast(x[r]);
                        }
                        x[0] += rv[it].x; x[1] += rv[it].y; x[2] += rv[it].z; x[3] += rv[it].w;
                        if (mr < M) *reinterpret_cast<float4 *>(Cz + (long)mr * a.ldc + nb) = make_float4(x[0], x[1], x[2], x[3]);
                    }
                }
            } else {
                // bf16 out: a lane takes 8 columns (one 16-byte store), 8 lanes one row's 64 columns (128 bytes = one line), 8 rows
                // per instruction — half the store instructions of the 8-byte form, and the stores, not the bytes, bound this tail
                static_assert(!HAS_R, "bf16 outputs carry no residual");
                const int erow = lane >> 3, ecol = 8 * (lane & 7), nb = ncol0 + ecol;
                const float *bp = bias_z ? bias_z + nb : a.epi_zero;
                const float4 b0 = *reinterpret_cast<const float4 *>(bp), b1 = *reinterpret_cast<const float4 *>(bias_z ? bp + 4 : bp);
                const float bb[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
                for (int i8 = 0; i8 < 8; ++i8) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        f32x4 v = acc[i8][j];
                        v *= a.alpha;
                        *reinterpret_cast<f32x4 *>(stg + (lane & 15) * SROW + 16 * j + 4 * (lane >> 4)) = v;
                    }
                    const int mrow0 = m0 + wr * 128 + i8 * 16;
#pragma unroll
                    for (int it = 0; it < 2; ++it) {
                        const int mr = mrow0 + erow + 8 * it;
                        const f32x4 t0 = *reinterpret_cast<const f32x4 *>(stg + (erow + 8 * it) * SROW + ecol);
                        const f32x4 t1 = *reinterpret_cast<const f32x4 *>(stg + (erow + 8 * it) * SROW + ecol + 4);
                        float x[8] = {t0[0], t0[1], t0[2], t0[3], t1[0], t1[1], t1[2], t1[3]};
                        unsigned pk[4];
#pragma unroll
                        for (int r = 0; r < 8; ++r) {
                            x[r] += bb[r];
                            if constexpr (GELU) x[r] = gelu_fast(x[r]);
                        }
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            __hip_bfloat16 h0 = __float2bfloat16(x[2 * r]), h1 = __float2bfloat16(x[2 * r + 1]);
                            pk[r] = (unsigned)*reinterpret_cast<bf16_t *>(&h0) | ((unsigned)*reinterpret_cast<bf16_t *>(&h1) << 16);
                        }
                        using u32x4n = __attribute__((ext_vector_type(4))) unsigned;
                        const u32x4n out4 = {pk[0], pk[1], pk[2], pk[3]};
                        if (FULL || mr < M)  // 60-120 MB read once by the next kernel: streaming stores
                            __builtin_nontemporal_store(out4, reinterpret_cast<u32x4n *>(Cz + (long)mr * a.ldc + nb));
                    }
                }
            }
            };
            if (m0 + 256 <= M) epilogue(std::true_type{});
            else epilogue(std::false_type{});
        }
        m0 = m1;
        n0 = n1;
        zz = z1n;
        have = have_next;
        ++round;
    }
}

}  // namespace

// 16 zero bytes in device memory: what the vector epilogue reads when a GEMM has no bias (an unconditional load)
namespace {
__device__ __attribute__((aligned(16))) float g_epi_zero[4] = {0.0f, 0.0f, 0.0f, 0.0f};
}  // namespace
const float *epi_zero_device() {
    thread_local const float *ptr[64] = {};
    int dev = 0;
    MW_HIP(hipGetDevice(&dev));
    MW_CHECK(dev >= 0 && dev < 64, "gemm: device index");
    if (!ptr[dev]) MW_HIP(hipGetSymbolAddress((void **)&ptr[dev], HIP_SYMBOL(g_epi_zero)));
    return ptr[dev];
}


void launch_gemm(const GemmArgs &args, DType in, DType out, hipStream_t st) {
    GemmArgs a = args;
    MW_CHECK(a.A && a.W && a.C, "gemm: null operand");
    MW_CHECK(a.M > 0 && a.N > 0 && a.K > 0, "gemm: empty shape");
    const int epc = in == DType::F32 ? 4 : 8;  // elements per 16-byte chunk
    MW_CHECK(a.K % epc == 0, "gemm: K must be a multiple of the 16-byte chunk");
    MW_CHECK(a.lda % epc == 0 && a.ldw % epc == 0, "gemm: row strides must keep 16-byte alignment");
    MW_CHECK(((uintptr_t)a.A % 16) == 0 && ((uintptr_t)a.W % 16) == 0, "gemm: operands must be 16-byte aligned");
    MW_CHECK(a.sA0 % epc == 0 && a.sA1 % epc == 0 && a.sW0 % epc == 0 && a.sW1 % epc == 0, "gemm: batch strides");
    // note: lda < K is legal — the conv stem passes overlapping row views (row t = rows t-1..t+1)
    MW_CHECK(a.nz0 >= 1 && a.nz1 >= 1 && (long)a.nz0 * a.nz1 <= 65535, "gemm: batch count");
    if (a.R) MW_CHECK(a.ldr > 0, "gemm: residual stride");
    {   // vector epilogue (gemm_bf16_epilogue): 4 consecutive columns per lane as 16-byte loads / 8- or 16-byte stores
        const size_t oa = out == DType::F32 ? 16 : 8, os = out == DType::F32 ? 4 : 2;
        const auto al = [](const void *p, size_t n) { return ((uintptr_t)p % n) == 0; };
        a.epi_vec = in == DType::BF16 && !a.trans_c && a.N % 4 == 0 && a.N >= 4 && al(a.C, oa) && a.ldc % 4 == 0 &&
                    (a.sC0 * os) % oa == 0 && (a.sC1 * os) % oa == 0 && (!a.bias || (al(a.bias, 16) && a.sB0 % 4 == 0 && a.sB1 % 4 == 0)) &&
                    (!a.R || (al(a.R, 16) && a.ldr % 4 == 0 && a.sR0 % 4 == 0 && a.sR1 % 4 == 0));
        a.epi_zero = epi_zero_device();
    }
    // 64-row tiles when 128-row tiles would give the 256 CUs at most about one tile each (N = d and N = 2d at M = 1500):
    // Q|K (240 tiles of 128 rows) 29.6 us, as 480 tiles of 64 rows 26.4; fc1 (480 tiles): 48 vs 56
    const long tiles128 = (long)ceil_div(a.N, BN) * ceil_div(a.M, 128) * a.nz0 * a.nz1;
    const bool bm64 = in == DType::BF16 && tiles128 < 300 && a.M > 64;
    // many-tile grids (the encoder pass over the windows of several streams, M = windows x 1500): the single-stage LDS-direct
    // kernel, four workgroups per CU
    const bool many = in == DType::BF16 && a.K % 64 == 0 && a.K >= 256 && tiles128 >= 900;  // (also the transposed V projection)
    // 256 x 256 persistent tiles once they fill the chip (>= 200 of them for 256 CUs): fc2 / attention output at M = 12000,
    // Q|K and fc1 from M = 6000 on
    // (batch members are tiles like any other: the cross K / V of all decoder layers for one window, 960 tiles)
    const long tiles256 = (long)(a.N / 256) * ceil_div(a.M, 256) * a.nz0 * a.nz1;
    const bool big = in == DType::BF16 && !a.trans_c && a.K % 64 == 0 && a.K >= 256 && a.N % 256 == 0 && tiles256 < (1L << 28) &&
                     a.epi_vec && (a.glds == 8 || (a.glds == 0 && tiles256 >= 200)) &&
                     // the epilogue shapes the 256 x 256 kernel is instantiated for; its bf16 rows leave as 16-byte stores
                     ((out == DType::F32 && a.act == ACT_NONE) || (out == DType::BF16 && !a.R && a.ldc % 8 == 0 && ((uintptr_t)a.C % 16) == 0));
    const int bm = big ? 256 : bm64 ? 64 : 128, bn = big ? 256 : BN;
    dim3 grid(ceil_div(a.N, bn), ceil_div(a.M, bm), a.nz0 * a.nz1);
    MW_CHECK(grid.y <= 65535, "gemm: M too large for grid.y");
    {   // XCD-aware tile order (xcd_tile_coord): super-row height for which an XCD's share of the tiles is a compact rectangle
        const long T = (long)grid.x * grid.y, G = T * grid.z, share = (G + 7) / 8;
        a.hb = 0;
        if (G >= 16 && G < (1L << 31)) {
            if (share >= T) {
                a.hb = (int)grid.y;  // an XCD gets whole batch slices: any order inside a slice
            } else {
                // rectangle of `share` tiles with equal operand bytes on both sides: h * BM = w * BN  ->  h = sqrt(share * BN / BM)
                const double h = sqrt((double)share * bn / bm);
                double best = 1e30;
                for (int xm = 1; xm <= 8; xm *= 2) {
                    const int hb = ceil_div((int)grid.y, xm);
                    if (fabs(hb - h) < best) best = fabs(hb - h), a.hb = hb;
                }
            }
        }
    }
    // pinned load ring (one workgroup per CU) when the grid has at most one tile per CU anyway
    // (64-row tiles keep two workgroups per CU pinned or not: always pinned)
    const bool pin = bm64 || (long)grid.x * grid.y * grid.z <= 256;
#define K64_LAUNCH(TOv, BMv)                                                                                                \
    do {                                                                                                                    \
        if (pin && a.trans_c) hipLaunchKernelGGL((gemm_bf16_k64_kernel<TOv, BMv, true, true>), grid, dim3(256), 0, st, a);  \
        else if (pin) hipLaunchKernelGGL((gemm_bf16_k64_kernel<TOv, BMv, false, true>), grid, dim3(256), 0, st, a);         \
        else if (a.trans_c) hipLaunchKernelGGL((gemm_bf16_k64_kernel<TOv, BMv, true, false>), grid, dim3(256), 0, st, a);   \
        else hipLaunchKernelGGL((gemm_bf16_k64_kernel<TOv, BMv, false, false>), grid, dim3(256), 0, st, a);                 \
    } while (0)
    // LDS-direct staging (GemmArgs::glds, set from the context's switch MI355_WHISPER_GEMM_GLDS read at context creation)
    const bool glds = a.glds && in == DType::BF16 && a.K % 64 == 0 && a.K >= 256;
#define GLDS_LAUNCH(TOv, BMv)                                                                                           \
    do {                                                                                                                \
        if (a.trans_c) hipLaunchKernelGGL((gemm_bf16_glds_kernel<TOv, BMv, true, 4>), grid, dim3(256), 0, st, a);       \
        else hipLaunchKernelGGL((gemm_bf16_glds_kernel<TOv, BMv, false, 4>), grid, dim3(256), 0, st, a);                \
    } while (0)
    if (big) {
        // persistent: one workgroup per CU at most, a multiple of 8 so that every XCD's range is walked in step
        const dim3 pgrid((unsigned)std::min<long>(ceil_div<long>(tiles256, 8) * 8, 256));
        const bool gelu = a.act == ACT_GELU;
        if (out == DType::F32 && a.R) hipLaunchKernelGGL((gemm_bf16_256_kernel<float, true, false>), pgrid, dim3(512), 0, st, a);
        else if (out == DType::F32) hipLaunchKernelGGL((gemm_bf16_256_kernel<float, false, false>), pgrid, dim3(512), 0, st, a);
        else if (gelu) hipLaunchKernelGGL((gemm_bf16_256_kernel<bf16_t, false, true>), pgrid, dim3(512), 0, st, a);
        else hipLaunchKernelGGL((gemm_bf16_256_kernel<bf16_t, false, false>), pgrid, dim3(512), 0, st, a);
    } else if (many && out == DType::F32) {
        if (a.trans_c) hipLaunchKernelGGL((gemm_bf16_glds_kernel<float, 128, true, 1>), grid, dim3(256), 0, st, a);
        else hipLaunchKernelGGL((gemm_bf16_glds_kernel<float, 128, false, 1>), grid, dim3(256), 0, st, a);
    } else if (many) {
        if (a.trans_c) hipLaunchKernelGGL((gemm_bf16_glds_kernel<bf16_t, 128, true, 1>), grid, dim3(256), 0, st, a);
        else hipLaunchKernelGGL((gemm_bf16_glds_kernel<bf16_t, 128, false, 1>), grid, dim3(256), 0, st, a);
    } else if (in == DType::F32 && out == DType::F32)
        hipLaunchKernelGGL((gemm_nt_kernel<float, float, 128>), grid, dim3(256), 0, st, a);
    else if (glds && out == DType::F32 && bm64)
        GLDS_LAUNCH(float, 64);
    else if (glds && out == DType::F32)
        GLDS_LAUNCH(float, 128);
    else if (glds && out == DType::BF16 && bm64)
        GLDS_LAUNCH(bf16_t, 64);
    else if (glds && out == DType::BF16)
        GLDS_LAUNCH(bf16_t, 128);
    else if (in == DType::BF16 && out == DType::F32 && bm64)
        K64_LAUNCH(float, 64);
    else if (in == DType::BF16 && out == DType::F32)
        K64_LAUNCH(float, 128);
    else if (in == DType::BF16 && out == DType::BF16 && bm64)
        K64_LAUNCH(bf16_t, 64);
    else if (in == DType::BF16 && out == DType::BF16)
        K64_LAUNCH(bf16_t, 128);
    else
        MW_CHECK(false, "gemm: unsupported dtype combination");
#undef K64_LAUNCH
#undef GLDS_LAUNCH
    MW_HIP(hipGetLastError());
}

}  // namespace mw
