// gemm_mx.hip — the encoder's large projections in MX-fp8 (MW_PRECISION_FP8_MX): OCP e4m3 elements with one E8M0 scale per
// block of 32 consecutive k, multiplied by v_mfma_scale_f32_16x16x128_f8f6f4 — twice the bf16 MFMA rate on half the
// operand bytes.  Replaces, inside whisper_full's encoder (/root/reference/src/whisper-utils/whisper-processing.cpp:218-219;
// whisper.cpp's whisper_build_graph_encoder mul_mat nodes), the Q|K|V, attention-output, fc1 and fc2 products of every block and the cross K / V projections.
//
// Operand maps of the instruction, found with exact data by tools/microbench/mb_mxfp8_probe (neither is documented in
// the guides): lane l = (r = l & 15, g = l >> 4) supplies 32 operand bytes — bytes 0-15 are k = 16 g .. 16 g + 15, bytes
// 16-31 are k = 64 + 16 g .. 64 + 16 g + 15 of its row r — and ONE scale byte (selected by op_sel from the lane's scale
// register): the scale of row r, k-block g (k = 32 g .. 32 g + 31).  With 128-byte LDS rows (one k-step of 128 e4m3) a
// lane's two 16-byte pieces are chunks g and 4 + g of its row: exactly the two fragment reads the bf16 kernel of gemm.hip
// does for its two 32-deep MFMAs, so staging, swizzle and fragment addressing are that kernel's, byte for byte; what
// changes is that each pair of reads feeds one instruction of twice the depth, and a k-step covers K = 128.
//
// Quantisation rule (the oracle's twin: oracle/whisper_oracle.cpp mx_qdq): per block of 32, scale = the smallest power of
// two s with absmax / s <= 448 (e4m3's largest finite value; E8M0 byte = exponent + 127, clamped at 0), element =
// round-to-nearest-even(value / s) by v_cvt_pk_fp8_f32.  Nothing saturates; a block of zeros gets scale byte 0.
#include "kernels.hpp"

#include <hip/hip_bf16.h>

#include <cstdlib>
#include <type_traits>

namespace mw {

namespace {

typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float mx_inv_scale(unsigned byte) { return __uint_as_float((254u - byte) << 23); }  // 2^(127 - byte)
__device__ __forceinline__ unsigned pack4_e4m3(float a, float b, float c, float d) {
    int pk = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false);
    pk = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, pk, true);
    return (unsigned)pk;
}
__device__ __forceinline__ float gelu_mx(float x) {  // as gemm.hip's bf16 epilogue: x * sigmoid(2u) on the hardware exp / rcp
    const float u2 = 1.5957691216057308f * x * (1.0f + 0.044715f * x * x);
    return x * __builtin_amdgcn_rcpf(1.0f + __expf(-u2));
}
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F &&f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// quantisers
// ---------------------------------------------------------------------------------------------------------------------
// one thread per (row, block of 32); ACT: activation scale order, else weight order
template <typename TIN, bool ACT>
__global__ __launch_bounds__(256) void quantize_mx_kernel(const TIN *__restrict__ src, long lds_, unsigned char *__restrict__ q, long ldq,
                                                          unsigned char *__restrict__ s, int rows, int K, int RBLK) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    const int nkb = K / 32;
    const int row = (int)(idx / nkb), kb = (int)(idx % nkb);
    if (row >= rows) return;
    const TIN *p = src + (long)row * lds_ + 32 * kb;
    float v[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) {
        if constexpr (sizeof(TIN) == 2) v[i] = __uint_as_float((unsigned)p[i] << 16);
        else v[i] = p[i];
    }
    float amax = 0.0f;
#pragma unroll
    for (int i = 0; i < 32; ++i) amax = fmaxf(amax, fabsf(v[i]));
    const unsigned sb = mx_scale_byte(amax);
    const float inv = mx_inv_scale(sb);
    unsigned *qo = reinterpret_cast<unsigned *>(q + (long)row * ldq + 32 * kb);
#pragma unroll
    for (int i = 0; i < 8; ++i) qo[i] = pack4_e4m3(v[4 * i] * inv, v[4 * i + 1] * inv, v[4 * i + 2] * inv, v[4 * i + 3] * inv);
    s[ACT ? mx_sa_off(row, kb, RBLK) : mx_sw_off(row, kb, RBLK)] = (unsigned char)sb;
}

// LayerNorm of enc_misc.hip (one wave per row, the same statistics in the same order: the bf16 copy equals the bf16 path's)
// with d = 256 NV; a lane holds columns (64 i + lane) 4 .. + 3, so a block of 32 columns is 8 neighbouring lanes of one i
template <int NV, bool WITH_BF16>
__global__ __launch_bounds__(256) void layernorm_mx_kernel(const float *__restrict__ x, long ldx, const float *__restrict__ w,
                                                           const float *__restrict__ b, bf16_t *__restrict__ out, long ldo,
                                                           unsigned char *__restrict__ q, long ldq, unsigned char *__restrict__ s, int MB,
                                                           int rows) {
    constexpr int d = 256 * NV;
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float *xr = x + (long)row * ldx;
    float4 v[NV], wv[NV], bv[NV];  // (weights and biases with the row: loaded in the output loop each slice waited for its own)
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        v[i] = *reinterpret_cast<const float4 *>(xr + (i * 64 + lane) * 4);
        wv[i] = *reinterpret_cast<const float4 *>(w + (i * 64 + lane) * 4);
        bv[i] = *reinterpret_cast<const float4 *>(b + (i * 64 + lane) * 4);
    }
    float sum = 0.0f;
#pragma unroll
    for (int i = 0; i < NV; ++i) sum += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    sum = wave_sum_desc(sum);  // (common.hpp: the shuffle form's partners and order, without its LDS round trips)
    const float mean = sum / (float)d;
    float s2 = 0.0f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        v[i].x -= mean; v[i].y -= mean; v[i].z -= mean; v[i].w -= mean;
        s2 += (v[i].x * v[i].x + v[i].y * v[i].y) + (v[i].z * v[i].z + v[i].w * v[i].w);
    }
    s2 = wave_sum_desc(s2);
    const float rstd = 1.0f / sqrtf(s2 / (float)d + 1e-5f);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = (i * 64 + lane) * 4;
        const float4 ww = wv[i], bb = bv[i];
        const float y0 = v[i].x * rstd * ww.x + bb.x, y1 = v[i].y * rstd * ww.y + bb.y, y2 = v[i].z * rstd * ww.z + bb.z,
                    y3 = v[i].w * rstd * ww.w + bb.w;
        if constexpr (WITH_BF16) {
            __hip_bfloat16 h0 = __float2bfloat16(y0), h1 = __float2bfloat16(y1), h2 = __float2bfloat16(y2), h3 = __float2bfloat16(y3);
            uint2 pk;
            pk.x = (unsigned)*reinterpret_cast<bf16_t *>(&h0) | ((unsigned)*reinterpret_cast<bf16_t *>(&h1) << 16);
            pk.y = (unsigned)*reinterpret_cast<bf16_t *>(&h2) | ((unsigned)*reinterpret_cast<bf16_t *>(&h3) << 16);
            *reinterpret_cast<uint2 *>(out + (long)row * ldo + c) = pk;
        }
        float amax = fmaxf(fmaxf(fabsf(y0), fabsf(y1)), fmaxf(fabsf(y2), fabsf(y3)));
        amax = xor_max<4>(xor_max<2>(xor_max<1>(amax)));
        const unsigned sb = mx_scale_byte(amax);
        const float inv = mx_inv_scale(sb);
        *reinterpret_cast<unsigned *>(q + (long)row * ldq + c) = pack4_e4m3(y0 * inv, y1 * inv, y2 * inv, y3 * inv);
        if ((lane & 7) == 0) s[mx_sa_off(row, i * 8 + (lane >> 3), MB)] = (unsigned char)sb;
    }
}

constexpr int MX_EPI_PATCH = 16 * (64 + 4) * 4 + 256;  // bytes of LDS a wave uses in the epilogue

// The epilogue of one output tile, shared by the two MX GEMM kernels below: the accumulators of a wave (RT row tiles of
// 16 x 64) take a turn through the wave's own LDS patch `stg` so that a lane stores whole 16- or 32-byte runs of a row.
template <int EPI, int BM>
__device__ __forceinline__ void mx_tile_epilogue(const GemmMxArgs &a, f32x4 (&acc)[BM / 32][4], float *stg, int m0, int n0, int z0, int lane,
                                                 int wr, int wc, int M, int MB) {
    constexpr int RT = BM / 32, SROW = 64 + 4;
    const int ncol0 = n0 + wc * 64;
    const float *bias_z = a.bias ? a.bias + z0 * a.sBias : nullptr;
    const long c_off = z0 * a.sC;  // elements of the output type
    auto epilogue = [&](auto full_tag) {
        constexpr bool FULL = decltype(full_tag)::value;
        if constexpr (EPI == MXE_F32_RES) {
            // (the row predicate stays on these stores on purpose: gemm.hip, fp32 epilogue)
            float *C = reinterpret_cast<float *>(a.C) + c_off;
            const int erow = lane >> 4, ecol = 4 * (lane & 15), nb = ncol0 + ecol;
            const float4 b4 = *reinterpret_cast<const float4 *>(bias_z ? bias_z + nb : a.epi_zero);
#pragma unroll
            for (int i8 = 0; i8 < RT; ++i8) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    f32x4 v = acc[i8][j];
                    v *= a.alpha;
                    *reinterpret_cast<f32x4 *>(stg + (lane & 15) * SROW + 16 * j + 4 * (lane >> 4)) = v;
                }
                const int mrow0 = m0 + wr * (BM / 2) + i8 * 16;
                float4 rv[4];
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    const int m = min(mrow0 + erow + 4 * it, M - 1);
                    rv[it] = *reinterpret_cast<const float4 *>(a.R + (long)m * a.ldr + nb);
                }
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    const int mr = mrow0 + erow + 4 * it;
                    const f32x4 t = *reinterpret_cast<const f32x4 *>(stg + (erow + 4 * it) * SROW + ecol);
                    const float4 o = make_float4(t[0] + b4.x + rv[it].x, t[1] + b4.y + rv[it].y, t[2] + b4.z + rv[it].z,
                                                 t[3] + b4.w + rv[it].w);
                    if (mr < M) *reinterpret_cast<float4 *>(C + (long)mr * a.ldc + nb) = o;
                }
            }
        } else {
            // a lane takes 8 columns of a row, 8 lanes one row's 64 columns, 8 rows per instruction
            const int erow = lane >> 3, ecol = 8 * (lane & 7), nb = ncol0 + ecol;
            const float *bp = bias_z ? bias_z + nb : a.epi_zero;
            const float4 b0 = *reinterpret_cast<const float4 *>(bp), b1 = *reinterpret_cast<const float4 *>(bias_z ? bp + 4 : bp);
            const float bb[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
            unsigned long long scl[2] = {0ull, 0ull};  // MXE_MX_GELU: the scale bytes of rows erow, erow + 8 of the RT slices
#pragma unroll
            for (int i8 = 0; i8 < RT; ++i8) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    f32x4 v = acc[i8][j];
                    v *= a.alpha;
                    *reinterpret_cast<f32x4 *>(stg + (lane & 15) * SROW + 16 * j + 4 * (lane >> 4)) = v;
                }
                const int mrow0 = m0 + wr * (BM / 2) + i8 * 16;
#pragma unroll
                for (int it = 0; it < 2; ++it) {
                    const int mr = mrow0 + erow + 8 * it;
                    const f32x4 t0 = *reinterpret_cast<const f32x4 *>(stg + (erow + 8 * it) * SROW + ecol);
                    const f32x4 t1 = *reinterpret_cast<const f32x4 *>(stg + (erow + 8 * it) * SROW + ecol + 4);
                    float x[8] = {t0[0], t0[1], t0[2], t0[3], t1[0], t1[1], t1[2], t1[3]};
#pragma unroll
                    for (int r = 0; r < 8; ++r) {
                        x[r] += bb[r];
                        if constexpr (EPI == MXE_MX_GELU) x[r] = gelu_mx(x[r]);
                    }
                    if constexpr (EPI == MXE_BF16) {
                        bf16_t *C = reinterpret_cast<bf16_t *>(a.C) + c_off;
                        unsigned pk[4];
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            __hip_bfloat16 h0 = __float2bfloat16(x[2 * r]), h1 = __float2bfloat16(x[2 * r + 1]);
                            pk[r] = (unsigned)*reinterpret_cast<bf16_t *>(&h0) | ((unsigned)*reinterpret_cast<bf16_t *>(&h1) << 16);
                        }
                        using u32x4n = __attribute__((ext_vector_type(4))) unsigned;
                        const u32x4n out4 = {pk[0], pk[1], pk[2], pk[3]};
                        if (FULL || mr < M) __builtin_nontemporal_store(out4, reinterpret_cast<u32x4n *>(C + (long)mr * a.ldc + nb));
                    } else {
                        // block of 32 output columns = lanes 4 b .. 4 b + 3 of the row's 8: absmax over the quad, one scale
                        float amax = 0.0f;
#pragma unroll
                        for (int r = 0; r < 8; ++r) amax = fmaxf(amax, fabsf(x[r]));
                        amax = fmaxf(amax, __uint_as_float(__builtin_amdgcn_update_dpp(0, __float_as_uint(amax), 0xB1, 0xf, 0xf, false)));
                        amax = fmaxf(amax, __uint_as_float(__builtin_amdgcn_update_dpp(0, __float_as_uint(amax), 0x4E, 0xf, 0xf, false)));
                        const unsigned sb = mx_scale_byte(amax);
                        const float inv = mx_inv_scale(sb);
                        using u32x2n = __attribute__((ext_vector_type(2))) unsigned;
                        const u32x2n out2 = {pack4_e4m3(x[0] * inv, x[1] * inv, x[2] * inv, x[3] * inv),
                                             pack4_e4m3(x[4] * inv, x[5] * inv, x[6] * inv, x[7] * inv)};
                        unsigned char *C = reinterpret_cast<unsigned char *>(a.C);
                        if (FULL || mr < M) __builtin_nontemporal_store(out2, reinterpret_cast<u32x2n *>(C + (long)mr * a.ldc + nb));
                        scl[it] |= (unsigned long long)sb << (8 * i8);
                    }
                }
            }
            if constexpr (EPI == MXE_MX_GELU) {
                // one 8-byte record per (128-row block, row & 15, output k-block), of which this wave's RT row tiles are
                // bytes base .. base + RT - 1: lane (erow, quad leader) holds rows erow and erow + 8
                if ((lane & 3) == 0) {
                    const int kb = (ncol0 >> 5) + ((lane & 7) >> 2);  // block of 32 output columns
                    const int mw = m0 + wr * (BM / 2), mblk = mw >> 7, base = (mw >> 4) & 7;
#pragma unroll
                    for (int it = 0; it < 2; ++it) {
                        unsigned char *rec = a.SC + (((size_t)(kb >> 2) * MB + mblk) * 64 + ((kb & 3) * 16 + erow + 8 * it)) * 8 + base;
                        if constexpr (RT == 8) *reinterpret_cast<unsigned long long *>(rec) = scl[it];
                        else if constexpr (RT == 4) *reinterpret_cast<unsigned *>(rec) = (unsigned)scl[it];
                        else *reinterpret_cast<unsigned short *>(rec) = (unsigned short)scl[it];
                    }
                }
            }
        }
    };
    if (m0 + BM <= M) epilogue(std::true_type{});
    else epilogue(std::false_type{});
}

// ---------------------------------------------------------------------------------------------------------------------
// 256 x 256 persistent MX-fp8 GEMM.  Structure, staging, tile order and the LDS turn of the epilogue are those of
// gemm_bf16_256_kernel (gemm.hip; its comments give the measurements behind them): 8 waves as 2 (M) x 4 (N), each 128 x 64
// of the output, two k-step buffers of four 16 KB half-tiles filled by global_load_lds, the next step's A loads issued
// before the first half of the products and its W loads before the second.  A k-step is K = 128 here: 64 KB staged for
// 16.8 MFLOP, 24 fragment reads for 32 MFMAs of 16 x 16 x 128.  The block scales of a step (8 bytes for the wave's 8 row
// tiles of A, 4 for its 4 row tiles of W, per lane) travel in registers: loaded a step ahead by ordinary loads, and copied
// behind the k-step's own wait BEFORE the step's DMA is issued — hipcc waits for every load in flight, DMA included,
// where a loaded register is first used.
//   EPI: MXE_BF16 (bf16 + bias), MXE_F32_RES (fp32 + bias + residual), MXE_MX_GELU (GELU(acc + bias) re-quantised to MX-fp8
//   for the next product: blocks of 32 output columns, 4 lanes of the bf16 row layout per block).
// ---------------------------------------------------------------------------------------------------------------------
// BM (256, 128 or 64 rows per tile): the wave grid stays 2 x 4, a wave owns RT = BM / 32 row tiles of 16 x 64; the smaller
// tiles are for M = 1500 (one window), where 256-row tiles would leave most CUs without work.  An output element is the
// same chain of MFMAs over k whatever BM is: results do not depend on it (a window encodes to the same bits alone and
// inside a batched pass).
template <int EPI, int BM>
__global__ __launch_bounds__(512) void gemm_mxfp8_256_kernel(GemmMxArgs a) {
    constexpr int BK = 128, RB = 128, HALF = 128 * RB, ATILE = BM * RB, STAGE = ATILE + 2 * HALF;  // A rows | W rows 0-127 | W rows 128-255
    constexpr int RT = BM / 32, AP = BM / 64;  // row tiles per wave; 8-row pieces of the A tile each wave stages per k-step
    __shared__ __attribute__((aligned(1024))) unsigned char lds[2 * STAGE];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 2, wc = wave & 3;
    const unsigned char *A = a.A, *W = a.W;
    const int M = a.M, N = a.N, K = a.K;
    const int tiles_n = N / 256, tiles_m = (M + BM - 1) / BM, tiles_z = tiles_m * tiles_n, ntiles = tiles_z * a.nz, per_xcd = (ntiles + 7) / 8;
    const int MB = 2 * ((M + 255) / 256), NB = N / 64;  // (mx_row_blocks_a)
    auto swz8 = [](int r, int c) { return r * RB + ((c ^ (r & 7)) << 4); };
    // (batch z slowest: what an XCD's workgroups hold at one time is a few W panels of one or two batch members and A)
    auto tile_of = [&](int r, int &m0, int &n0, int &z) -> bool {
        const int L = blockIdx.x + r * gridDim.x, j = L >> 3, p = (L & 7) * per_xcd + j;
        if (j >= per_xcd || p >= ntiles) return false;
        z = p / tiles_z;
        const int q = p - z * tiles_z;
        m0 = (q / tiles_n) * BM;
        n0 = (q % tiles_n) * 256;
        return true;
    };
    // piece p of this wave in the A tile / the W tile: rows 8 * wave + 64 * p .. + 7; lane -> row lane >> 3, slot lane & 7,
    // global chunk slot ^ (row & 7)
    const unsigned char *ga[AP], *gw[4];
    const uint2 *gsa = nullptr;
    const unsigned *gsw = nullptr;
    int sa_shift = 0;  // BM < 256: where the wave's RT scale bytes start in the 8-byte record of its 128-row block (bits)
    auto point = [&](int m0, int n0, int z) {
#pragma unroll
        for (int pc = 0; pc < 4; ++pc) {
            const int row = 8 * wave + 64 * pc + (lane >> 3), chk = (lane & 7) ^ (row & 7);
            if (pc < AP) ga[pc] = A + (long)min(m0 + row, M - 1) * a.lda + chk * 16;
            gw[pc] = W + z * a.sW + (long)min(n0 + row, N - 1) * a.ldw + chk * 16;
        }
        const int mw = m0 + wr * (BM / 2);  // first row of the wave
        gsa = reinterpret_cast<const uint2 *>(a.SA) + ((long)(mw >> 7) * 64 + lane);
        gsw = reinterpret_cast<const unsigned *>(a.SW + z * a.sSW) + ((long)((n0 >> 6) + wc) * 64 + lane);
        sa_shift = 8 * ((mw >> 4) & 7);
    };
    const int nk = K / BK;
    using gptr_t = const __attribute__((address_space(1))) void *;
    using lptr_t = __attribute__((address_space(3))) void *;
    auto issue_a = [&](int kt, int stage) {
        const long off = (long)kt * BK;
        unsigned char *sa = lds + stage * STAGE + (8 * wave) * RB;  // wave-uniform
#pragma unroll
        for (int pc = 0; pc < AP; ++pc) __builtin_amdgcn_global_load_lds((gptr_t)(ga[pc] + off), (lptr_t)(sa + 64 * pc * RB), 16, 0, 0);
    };
    auto issue_w = [&](int kt, int stage) {
        const long off = (long)kt * BK;
        unsigned char *sw = lds + stage * STAGE + ATILE + (8 * wave) * RB;
#pragma unroll
        for (int pc = 0; pc < 4; ++pc) __builtin_amdgcn_global_load_lds((gptr_t)(gw[pc] + off), (lptr_t)(sw + 64 * pc * RB), 16, 0, 0);
    };
    const int fr = lane & 15, fq = lane >> 4;
    int m0 = 0, n0 = 0, z0 = 0, round = 0;
    bool have = tile_of(0, m0, n0, z0);  // workgroup-uniform throughout
    uint2 sa_n = make_uint2(0, 0);   // the scales of the step in flight
    unsigned sw_n = 0;
    if (have) {
        point(m0, n0, z0);
        issue_a(0, 0);
        issue_w(0, 0);
        sa_n = gsa[0];
        sw_n = gsw[0];
    }
    int par = 0;
    while (have) {
        f32x4 acc[RT][4];
#pragma unroll
        for (int i = 0; i < RT; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0, 0, 0, 0};
        int m1 = 0, n1 = 0, z1 = 0;
        const bool have_next = tile_of(round + 1, m1, n1, z1);
        for (int kt = 0; kt < nk; ++kt) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            int sa_lo = (int)sa_n.x, sa_hi = (int)sa_n.y, sw = (int)sw_n;
            asm volatile("" : "+v"(sa_lo), "+v"(sa_hi), "+v"(sw));  // this step's scales are taken here, before any new load
            if constexpr (RT < 8)  // the wave's RT bytes to the bottom of sa_lo (op_sel is an immediate)
                sa_lo = (int)((((unsigned long long)(unsigned)sa_hi << 32) | (unsigned)sa_lo) >> sa_shift);
            __builtin_amdgcn_s_barrier();
            const unsigned char *la = lds + par * STAGE + wr * (ATILE / 2), *lw = lds + par * STAGE + ATILE + (wc >> 1) * HALF;
            const int wcol = (wc & 1) * 64;
            const bool last = kt + 1 == nk;
            if (last && have_next) point(m1, n1, z1);  // the loads below then fetch step 0 of the next tile
            const bool more = !last || have_next;
            const int nkt = last ? 0 : kt + 1;
            if (more) {
                issue_a(nkt, par ^ 1);
                sa_n = gsa[(long)nkt * MB * 64];
                sw_n = gsw[(long)nkt * NB * 64];
            }
            i32x8 wf[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const i32x4 lo = *reinterpret_cast<const i32x4 *>(lw + swz8(wcol + j * 16 + fr, fq));
                const i32x4 hi = *reinterpret_cast<const i32x4 *>(lw + swz8(wcol + j * 16 + fr, 4 + fq));
                wf[j] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
            }
            // the wave's rows in groups of two row tiles: 4 fragment reads, 8 products
            static_for<0, RT / 2>([&](auto qq_c) {
                constexpr int qq = decltype(qq_c)::value;
                i32x8 af[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const i32x4 lo = *reinterpret_cast<const i32x4 *>(la + swz8(qq * 32 + i * 16 + fr, fq));
                    const i32x4 hi = *reinterpret_cast<const i32x4 *>(la + swz8(qq * 32 + i * 16 + fr, 4 + fq));
                    af[i] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                }
                static_for<0, 2>([&](auto i_c) {
                    constexpr int i = decltype(i_c)::value, it = qq * 2 + i;  // row tile 0..7 of the wave
                    static_for<0, 4>([&](auto j_c) {
                        constexpr int j = decltype(j_c)::value;
                        acc[it][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(wf[j], af[i], acc[it][j], 0, 0, j, sw, it & 3,
                                                                                     it >= 4 ? sa_hi : sa_lo);
                    });
                });
                if (qq == (RT >= 4 ? RT / 4 - 1 : 0) && more) issue_w(nkt, par ^ 1);  // behind half of the products
            });
            par ^= 1;
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // every wave has read the last k-step
        mx_tile_epilogue<EPI, BM>(a, acc, reinterpret_cast<float *>(lds + (par ^ 1) * STAGE + wave * MX_EPI_PATCH), m0, n0, z0, lane, wr, wc, M, MB);
        m0 = m1;
        n0 = n1;
        z0 = z1;
        have = have_next;
        ++round;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// The same product for few rows (one to a few windows: BM = 128 or 64), where a workgroup has one or two tiles and its
// time is the latency of its own k-steps: THREE k-step buffers, each step's loads issued two steps ahead, and the wait at
// the top of a step counted (`vmcnt(NL)`: the step after this one may still be in flight).  For the count to be the
// kernel's own, no load of the k-loop has a register destination (hipcc waits for every load in flight where a loaded
// register is first used): the block scales of a step travel through LDS like the tiles, one more 16-byte DMA per lane —
// the 512 bytes of the tile's 128-row block of A and the 1 KB of its four 64-row blocks of W, contiguous in the scale
// layouts — and are read from there at the top of the step.  Tile order, fragment reads, the chain of MFMAs of an output
// element and the epilogue are those of gemm_mxfp8_256_kernel: same bits.
// ---------------------------------------------------------------------------------------------------------------------
template <int EPI, int BM>
__global__ __launch_bounds__(512) void gemm_mxfp8_s3_kernel(GemmMxArgs a) {
    static_assert(BM == 128 || BM == 64, "the three-buffer kernel is for the small tiles");
    constexpr int BK = 128, RB = 128, HALF = 128 * RB, ATILE = BM * RB, SCL = 2048, STAGE = ATILE + 2 * HALF + SCL, NST = 3;
    constexpr int RT = BM / 32, AP = BM / 64, NL = AP + 4 + 1;  // DMA instructions per wave and k-step
    static_assert(8 * MX_EPI_PATCH <= STAGE, "the epilogue's patches fit one buffer");
    __shared__ __attribute__((aligned(1024))) unsigned char lds[NST * STAGE];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 2, wc = wave & 3;
    const unsigned char *A = a.A, *W = a.W;
    const int M = a.M, N = a.N, K = a.K;
    const int tiles_n = N / 256, tiles_m = (M + BM - 1) / BM, tiles_z = tiles_m * tiles_n, ntiles = tiles_z * a.nz, per_xcd = (ntiles + 7) / 8;
    const int MB = 2 * ((M + 255) / 256), NB = N / 64;
    auto swz8 = [](int r, int c) { return r * RB + ((c ^ (r & 7)) << 4); };
    auto tile_of = [&](int r, int &m0, int &n0, int &z) -> bool {
        const int L = blockIdx.x + r * gridDim.x, j = L >> 3, p = (L & 7) * per_xcd + j;
        if (j >= per_xcd || p >= ntiles) return false;
        z = p / tiles_z;
        const int q = p - z * tiles_z;
        m0 = (q / tiles_n) * BM;
        n0 = (q % tiles_n) * 256;
        return true;
    };
    // the loads run two k-steps ahead of the products, across tile ends: they have a cursor (tile round, k-step) of their own
    const unsigned char *ga[AP], *gw[4], *gs = nullptr;
    // a step's scale area, 16 bytes per lane: even waves bring A's 512 bytes and the first half of W's 1 KB, odd waves the
    // second half of W's and a repeat into the padding (every wave issues the same count; the four copies are the same bytes)
    const bool s_of_a = !(wave & 1) && lane < 32;
    const long gs_step = s_of_a ? (long)MB * 512 : (long)NB * 256;  // bytes from one k-step's scales to the next
    const unsigned char *const gs0 = s_of_a ? a.SA + lane * 16 : a.SW + (!(wave & 1) ? (lane - 32) * 16 : lane < 32 ? 512 + lane * 16 : (lane - 32) * 16);
    auto point = [&](int m0, int n0, int z) {
#pragma unroll
        for (int pc = 0; pc < 4; ++pc) {
            const int row = 8 * wave + 64 * pc + (lane >> 3), chk = (lane & 7) ^ (row & 7);
            if (pc < AP) ga[pc] = A + (long)min(m0 + row, M - 1) * a.lda + chk * 16;
            gw[pc] = W + z * a.sW + (long)min(n0 + row, N - 1) * a.ldw + chk * 16;
        }
        gs = gs0 + (s_of_a ? (long)(m0 >> 7) * 512 : z * a.sSW + (long)(n0 >> 6) * 256);
    };
    const int nk = K / BK;
    using gptr_t = const __attribute__((address_space(1))) void *;
    using lptr_t = __attribute__((address_space(3))) void *;
    auto issue = [&](int kt, int stage) {
        const long off = (long)kt * BK;
        unsigned char *sb = lds + stage * STAGE;
        unsigned char *sa = sb + (8 * wave) * RB, *sw = sb + ATILE + (8 * wave) * RB;  // wave-uniform
#pragma unroll
        for (int pc = 0; pc < AP; ++pc) __builtin_amdgcn_global_load_lds((gptr_t)(ga[pc] + off), (lptr_t)(sa + 64 * pc * RB), 16, 0, 0);
#pragma unroll
        for (int pc = 0; pc < 4; ++pc) __builtin_amdgcn_global_load_lds((gptr_t)(gw[pc] + off), (lptr_t)(sw + 64 * pc * RB), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((gptr_t)(gs + kt * gs_step), (lptr_t)(sb + ATILE + 2 * HALF + (wave & 1) * 1024), 16, 0, 0);
    };
    int c_round = 0, c_kt = 0, cm = 0, cn = 0, cz = 0;
    bool c_have = tile_of(0, cm, cn, cz);  // workgroup-uniform throughout
    if (c_have) point(cm, cn, cz);
    auto issue_next = [&](int stage) -> bool {
        if (!c_have) return false;
        issue(c_kt, stage);
        if (++c_kt == nk) {
            c_kt = 0;
            c_have = tile_of(++c_round, cm, cn, cz);
            if (c_have) point(cm, cn, cz);
        }
        return true;
    };
    const int fr = lane & 15, fq = lane >> 4;
    int m0 = 0, n0 = 0, z0 = 0, round = 0;
    bool have = tile_of(0, m0, n0, z0);
    int inflight = 0, si = 0, sc = 0;  // k-steps issued and not yet multiplied; the buffers of the next issue and the next products
    for (; si < 2; ++si)
        if (issue_next(si)) ++inflight;
    while (have) {
        f32x4 acc[RT][4];
#pragma unroll
        for (int i = 0; i < RT; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0, 0, 0, 0};
        const int sa_shift = 8 * (((m0 + wr * (BM / 2)) >> 4) & 7);  // the wave's RT bytes in the record of its 128-row block (bits)
        for (int kt = 0; kt < nk; ++kt) {
            if (inflight >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NL) : "memory");  // this step has landed; the next may be on its way
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();  // ... for every wave's part of it, and every wave is past the step before
            if (issue_next(si)) {          // into the buffer of the step before
                ++inflight;
                si = si == NST - 1 ? 0 : si + 1;
            }
            const unsigned char *sb = lds + sc * STAGE;
            const unsigned char *la = sb + wr * (ATILE / 2), *lw = sb + ATILE + (wc >> 1) * HALF, *ls = sb + ATILE + 2 * HALF;
            const int wcol = (wc & 1) * 64;
            typedef int i32x2 __attribute__((ext_vector_type(2)));
            typedef int i32x1 __attribute__((ext_vector_type(1)));
            const i32x2 sa2 = *reinterpret_cast<const i32x2 *>(ls + lane * 8);
            const int sw = (*reinterpret_cast<const i32x1 *>(ls + 512 + (wc * 64 + lane) * 4))[0];
            const int sa_lo = (int)((((unsigned long long)(unsigned)sa2[1] << 32) | (unsigned)sa2[0]) >> sa_shift);
            i32x8 wf[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const i32x4 lo = *reinterpret_cast<const i32x4 *>(lw + swz8(wcol + j * 16 + fr, fq));
                const i32x4 hi = *reinterpret_cast<const i32x4 *>(lw + swz8(wcol + j * 16 + fr, 4 + fq));
                wf[j] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
            }
            static_for<0, RT / 2>([&](auto qq_c) {
                constexpr int qq = decltype(qq_c)::value;
                i32x8 af[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const i32x4 lo = *reinterpret_cast<const i32x4 *>(la + swz8(qq * 32 + i * 16 + fr, fq));
                    const i32x4 hi = *reinterpret_cast<const i32x4 *>(la + swz8(qq * 32 + i * 16 + fr, 4 + fq));
                    af[i] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                }
                static_for<0, 2>([&](auto i_c) {
                    constexpr int i = decltype(i_c)::value, it = qq * 2 + i;
                    static_for<0, 4>([&](auto j_c) {
                        constexpr int j = decltype(j_c)::value;
                        acc[it][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(wf[j], af[i], acc[it][j], 0, 0, j, sw, it, sa_lo);
                    });
                });
            });
            --inflight;
            sc = sc == NST - 1 ? 0 : sc + 1;
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // every wave has read the last k-step: its buffer is free
        const int sfree = sc == 0 ? NST - 1 : sc - 1;
        mx_tile_epilogue<EPI, BM>(a, acc, reinterpret_cast<float *>(lds + sfree * STAGE + wave * MX_EPI_PATCH), m0, n0, z0, lane, wr, wc, M, MB);
        have = tile_of(++round, m0, n0, z0);
    }
}

}  // namespace

void launch_quantize_mx_weights(const void *w_bf16, int N, int K, unsigned char *q, unsigned char *s, hipStream_t st) {
    MW_CHECK(N > 0 && K > 0 && K % 128 == 0 && N % 64 == 0, "mx weights: N % 64, K % 128");
    const long n = (long)N * (K / 32);
    hipLaunchKernelGGL((quantize_mx_kernel<bf16_t, false>), dim3((unsigned)ceil_div<long>(n, 256)), dim3(256), 0, st, (const bf16_t *)w_bf16,
                       (long)K, q, (long)K, s, N, K, N / 64);
    MW_HIP(hipGetLastError());
}

void launch_quantize_mx_rows(const float *x, long ldx, int M, int K, unsigned char *q, long ldq, unsigned char *s, hipStream_t st) {
    MW_CHECK(M > 0 && K > 0 && K % 128 == 0 && ldq % 4 == 0 && ((uintptr_t)q % 4) == 0, "mx rows: K % 128, 4-byte aligned rows");
    const long n = (long)M * (K / 32);
    hipLaunchKernelGGL((quantize_mx_kernel<float, true>), dim3((unsigned)ceil_div<long>(n, 256)), dim3(256), 0, st, x, ldx, q, ldq, s, M, K,
                       mx_row_blocks_a(M));
    MW_HIP(hipGetLastError());
}

void launch_quantize_mx_rows_bf16(const void *x_bf16, long ldx, int M, int K, unsigned char *q, long ldq, unsigned char *s, hipStream_t st) {
    MW_CHECK(M > 0 && K > 0 && K % 128 == 0 && ldq % 4 == 0 && ((uintptr_t)q % 4) == 0, "mx rows: K % 128, 4-byte aligned rows");
    const long n = (long)M * (K / 32);
    hipLaunchKernelGGL((quantize_mx_kernel<bf16_t, true>), dim3((unsigned)ceil_div<long>(n, 256)), dim3(256), 0, st, (const bf16_t *)x_bf16, ldx, q,
                       ldq, s, M, K, mx_row_blocks_a(M));
    MW_HIP(hipGetLastError());
}

void launch_layernorm_mx(const float *x, long ldx, const float *w, const float *b, void *out_bf16, long ldo, unsigned char *q, long ldq,
                         unsigned char *s, int rows, int d, hipStream_t st) {
    MW_CHECK(rows > 0 && d % 256 == 0 && d >= 256 && d <= 2048, "layernorm (mx): d must be a multiple of 256, <= 2048");
    MW_CHECK(ldx % 4 == 0 && ((uintptr_t)x % 16) == 0 && ((uintptr_t)w % 16) == 0 && ((uintptr_t)b % 16) == 0 && ldq % 4 == 0 &&
                 ((uintptr_t)q % 4) == 0 && (!out_bf16 || (ldo % 4 == 0 && ((uintptr_t)out_bf16 % 8) == 0)),
             "layernorm (mx): alignment");
    const dim3 grid(ceil_div(rows, 4));
    const int MB = mx_row_blocks_a(rows);
#define LN_MX(NVv)                                                                                                                     \
    case NVv:                                                                                                                          \
        if (out_bf16)                                                                                                                  \
            hipLaunchKernelGGL((layernorm_mx_kernel<NVv, true>), grid, dim3(256), 0, st, x, ldx, w, b, (bf16_t *)out_bf16, ldo, q, ldq, s, MB, rows); \
        else                                                                                                                           \
            hipLaunchKernelGGL((layernorm_mx_kernel<NVv, false>), grid, dim3(256), 0, st, x, ldx, w, b, (bf16_t *)nullptr, ldo, q, ldq, s, MB, rows); \
        break;
    switch (d / 256) {
        LN_MX(1) LN_MX(2) LN_MX(3) LN_MX(4) LN_MX(5) LN_MX(6) LN_MX(7) LN_MX(8)
    default: MW_CHECK(false, "layernorm (mx): d");
    }
#undef LN_MX
    MW_HIP(hipGetLastError());
}

void launch_gemm_mx(const GemmMxArgs &args, int epi, hipStream_t st) {
    GemmMxArgs a = args;
    MW_CHECK(a.A && a.SA && a.W && a.SW && a.C, "gemm (mx): null operand");
    MW_CHECK(a.M > 0 && a.N > 0 && a.K > 0 && a.K % 128 == 0 && a.N % 256 == 0, "gemm (mx): K % 128, N % 256");
    MW_CHECK(a.lda % 16 == 0 && a.ldw % 16 == 0 && ((uintptr_t)a.A % 16) == 0 && ((uintptr_t)a.W % 16) == 0 && a.lda >= a.K && a.ldw >= a.K,
             "gemm (mx): operand rows must keep 16-byte alignment");
    MW_CHECK(((uintptr_t)a.SA % 8) == 0 && ((uintptr_t)a.SW % 4) == 0, "gemm (mx): scale alignment");
    MW_CHECK(!a.bias || ((uintptr_t)a.bias % 16) == 0, "gemm (mx): bias alignment");
    MW_CHECK(a.ldc >= a.N, "gemm (mx): output stride");
    a.epi_zero = epi_zero_device();
    // the largest tile height that still gives the 256 CUs about a tile each (M = 1500: 64 or 128 rows; batched passes: 256)
    MW_CHECK(a.nz >= 1 && (a.nz == 1 || (epi == MXE_BF16 && a.sW % 16 == 0 && a.sSW % 4 == 0 && a.sC % 8 == 0 && a.sBias % 4 == 0)),
             "gemm (mx): batch strides (bf16 rows only)");
    // Tile height: the one with the shortest estimated launch, rounds of <= 256 resident tiles x (k-steps x time of a k-step +
    // epilogue), with the k-step and epilogue times measured per height (us): M = 1500 runs fc2 on 64-row tiles (120 of
    // them), fc1 and Q|K|V on 128-row tiles (240 / 180: one round), batched passes on 256-row tiles.  The small tiles run on
    // the three-buffer kernel: inside the encoder, with each layer's weights arriving from HBM, it is the faster one at
    // every shape (one window: fc2 53 -> 39 us, fc1 21.0 -> 19.2, Q|K|V 18.6 -> 16.5; four windows' fc2 at 128 rows
    // 65 -> 58.5); in a loop over one product, with the operands resident in L2, the two-buffer form was level or ahead
    // at K = 1280.  Any choice gives the same bits.
    const long nt = (long)(a.N / 256) * a.nz;
    const int nk = a.K / 128;
    int bm = 256;
    double best = 1e30;
    for (int cand : {256, 128, 64}) {
        const double t_k = cand == 256 ? 1.9 : cand == 128 ? 1.3 : 0.9, t_e = cand == 256 ? 6.0 : cand == 128 ? 4.0 : 2.5;
        const double cost = (double)ceil_div<long>(nt * ceil_div(a.M, cand), 256) * (nk * t_k + t_e);
        if (cost < best - 1e-9) best = cost, bm = cand;
    }
    if (const char *e = getenv("MI355_WHISPER_MX_TILE_ROWS")) {  // development / tests: a given tile height (same bits by construction)
        const int v = atoi(e);
        if (v == 256 || v == 128 || v == 64) bm = v;
    }
    const long tiles = nt * ceil_div(a.M, bm);
    MW_CHECK(tiles < (1L << 28), "gemm (mx): tile count");
    const dim3 pgrid((unsigned)std::min<long>(ceil_div<long>(tiles, 8) * 8, 256));
#define MX_LAUNCH(EPIv)                                                                                                     \
    do {                                                                                                                    \
        if (bm == 256) hipLaunchKernelGGL((gemm_mxfp8_256_kernel<EPIv, 256>), pgrid, dim3(512), 0, st, a);                  \
        else if (bm == 128) hipLaunchKernelGGL((gemm_mxfp8_s3_kernel<EPIv, 128>), pgrid, dim3(512), 0, st, a);              \
        else hipLaunchKernelGGL((gemm_mxfp8_s3_kernel<EPIv, 64>), pgrid, dim3(512), 0, st, a);                              \
    } while (0)
    if (epi == MXE_BF16) {
        MW_CHECK(!a.R && a.ldc % 8 == 0 && ((uintptr_t)a.C % 16) == 0, "gemm (mx): bf16 rows leave as 16-byte stores");
        MX_LAUNCH(MXE_BF16);
    } else if (epi == MXE_F32_RES) {
        MW_CHECK(a.R && a.ldr % 4 == 0 && ((uintptr_t)a.R % 16) == 0 && a.ldc % 4 == 0 && ((uintptr_t)a.C % 16) == 0,
                 "gemm (mx): fp32 rows and residual leave as 16-byte accesses");
        MX_LAUNCH(MXE_F32_RES);
    } else if (epi == MXE_MX_GELU) {
        MW_CHECK(!a.R && a.SC && a.ldc % 8 == 0 && ((uintptr_t)a.C % 8) == 0 && ((uintptr_t)a.SC % 8) == 0 && a.N % 128 == 0,
                 "gemm (mx): e4m3 rows leave as 8-byte stores");
        MX_LAUNCH(MXE_MX_GELU);
    } else {
        MW_CHECK(false, "gemm (mx): unknown epilogue");
    }
#undef MX_LAUNCH
    MW_HIP(hipGetLastError());
}

}  // namespace mw
