// enc_misc.hip — K3 LayerNorm, row softmax of the unfused attention path, small utility kernels.
// HBM-bound streaming kernels: one wave per row, 16-byte loads, statistics by wave shuffles.
// LayerNorm follows ggml_norm as whisper.cpp uses it (eps 1e-5, mean then centred variance;
// SURVEY.md A.3), reached from whisper_full (/root/reference/src/whisper-utils/whisper-processing.cpp:218).
#include "kernels.hpp"

#include <hip/hip_bf16.h>

namespace mw {

namespace {

__device__ __forceinline__ float wave_sum(float v) { return wave_sum_desc(v); }  // (common.hpp: no LDS shuffles, same order)
__device__ __forceinline__ float wave_max(float v) { return wave_max_desc(v); }
__device__ __forceinline__ void store_out(float *p, float v) { *p = v; }
__device__ __forceinline__ void store_out(bf16_t *p, float v) {
    __hip_bfloat16 h = __float2bfloat16(v);
    *p = *reinterpret_cast<bf16_t *>(&h);
}

// one wave per row; d <= 64 * 4 * MAXV (MAXV = ceil(d / 256): the launcher's choice).  Every load is unconditional — a clamped
// address, the value used only where the column exists — and issued before the first use: under `if (c < d)` each of
// the row's loads sat in a predicated block that opened with s_waitcnt vmcnt(0), i.e. one memory round trip per 1 KB of a
// row (6.1 us for 1500 rows of 1280).  The sums run over the same columns in the same order: same bits.
template <typename TOUT, int MAXV>
__global__ __launch_bounds__(256) void layernorm_kernel(const float *__restrict__ x, long ldx,
                                                        const float *__restrict__ w, const float *__restrict__ b,
                                                        TOUT *__restrict__ out, long ldo, int rows, int d) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float *xr = x + (long)row * ldx;
    float4 v[MAXV], ww[MAXV], bb[MAXV];
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = min((i * 64 + lane) * 4, d - 4);
        v[i] = *reinterpret_cast<const float4 *>(xr + c);
        ww[i] = *reinterpret_cast<const float4 *>(w + c);
        bb[i] = *reinterpret_cast<const float4 *>(b + c);
    }
    float s = 0.0f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const bool in = (i * 64 + lane) * 4 < d;
        s += in ? (v[i].x + v[i].y) + (v[i].z + v[i].w) : 0.0f;
    }
    const float mean = wave_sum(s) / (float)d;
    float s2 = 0.0f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const bool in = (i * 64 + lane) * 4 < d;
        v[i].x -= mean; v[i].y -= mean; v[i].z -= mean; v[i].w -= mean;
        s2 += in ? (v[i].x * v[i].x + v[i].y * v[i].y) + (v[i].z * v[i].z + v[i].w * v[i].w) : 0.0f;
    }
    const float rstd = 1.0f / sqrtf(wave_sum(s2) / (float)d + 1e-5f);
    TOUT *o = out + (long)row * ldo;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = (i * 64 + lane) * 4;
        if (c < d) {
            store_out(o + c + 0, v[i].x * rstd * ww[i].x + bb[i].x);
            store_out(o + c + 1, v[i].y * rstd * ww[i].y + bb[i].y);
            store_out(o + c + 2, v[i].z * rstd * ww[i].z + bb[i].z);
            store_out(o + c + 3, v[i].w * rstd * ww[i].w + bb[i].w);
        }
    }
}

// one wave per row of n (<= 64*MAXE) scores
template <typename TOUT, int MAXE>
__global__ __launch_bounds__(256) void softmax_rows_kernel(const float *__restrict__ s, long lds_, TOUT *__restrict__ out,
                                                           long ldo, long rows, int n, int n_pad, float scale) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float *sr = s + row * lds_;
    float v[MAXE];
    float mx = -INFINITY;
#pragma unroll
    for (int i = 0; i < MAXE; ++i) {
        const int c = i * 64 + lane;
        v[i] = c < n ? sr[c] * scale : -INFINITY;
        mx = fmaxf(mx, v[i]);
    }
    mx = wave_max(mx);
    float sum = 0.0f;
#pragma unroll
    for (int i = 0; i < MAXE; ++i) {
        const int c = i * 64 + lane;
        v[i] = c < n ? expf(v[i] - mx) : 0.0f;
        sum += v[i];
    }
    const float inv = 1.0f / wave_sum(sum);
    TOUT *o = out + row * ldo;
#pragma unroll
    for (int i = 0; i < MAXE; ++i) {
        const int c = i * 64 + lane;
        if (c < n_pad) store_out(o + c, v[i] * inv);
    }
}

__global__ __launch_bounds__(256) void convert_kernel(const float *__restrict__ src, bf16_t *__restrict__ dst, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) store_out(dst + i, src[i]);
}

// 64 x 64 tiles through LDS, 16-byte accesses on both sides
__global__ __launch_bounds__(256) void transpose_v_kernel(const bf16_t *__restrict__ src, long lds_, bf16_t *__restrict__ vt, int n_ctx, int tp,
                                                          int d) {
    __shared__ bf16_t tile[64][64 + 8];
    const int t0 = blockIdx.x * 64, n0 = blockIdx.y * 64, w = blockIdx.z;
    const int tid = threadIdx.x, r = tid >> 3, c8 = (tid & 7) * 8;
    src += (long)w * n_ctx * lds_;
    vt += (long)w * d * tp;
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int t = t0 + r + 32 * it;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (t < n_ctx) v = *reinterpret_cast<const uint4 *>(src + (long)t * lds_ + n0 + c8);
        const bf16_t *e = reinterpret_cast<const bf16_t *>(&v);
#pragma unroll
        for (int k = 0; k < 8; ++k) tile[c8 + k][r + 32 * it] = e[k];
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int n = n0 + r + 32 * it;
        *reinterpret_cast<uint4 *>(vt + (long)n * tp + t0 + c8) = *reinterpret_cast<const uint4 *>(&tile[r + 32 * it][c8]);
    }
}

}  // namespace

void launch_transpose_v(const void *src, long lds_, void *vt, int n_ctx, int tp, int d, int n_windows, hipStream_t st) {
    MW_CHECK(src && vt && n_ctx > 0 && tp % 64 == 0 && tp >= n_ctx && d % 64 == 0 && lds_ % 8 == 0 && ((uintptr_t)src % 16) == 0 &&
                 ((uintptr_t)vt % 16) == 0 && n_windows >= 1 && n_windows <= 65535,
             "transpose_v: shape / alignment");
    hipLaunchKernelGGL(transpose_v_kernel, dim3(tp / 64, d / 64, n_windows), dim3(256), 0, st, (const bf16_t *)src, lds_, (bf16_t *)vt, n_ctx, tp, d);
    MW_HIP(hipGetLastError());
}

void launch_layernorm(const float *x, long ldx, const float *w, const float *b, void *out, long ldo, DType out_t,
                      int rows, int d, hipStream_t st) {
    MW_CHECK(rows > 0 && d > 0 && d % 4 == 0 && d <= 64 * 4 * 8, "layernorm: d must be a multiple of 4, <= 2048");
    MW_CHECK(ldx % 4 == 0 && ((uintptr_t)x % 16) == 0 && ((uintptr_t)w % 16) == 0 && ((uintptr_t)b % 16) == 0,
             "layernorm: alignment");
    dim3 grid(ceil_div(rows, 4));
#define MW_LN(Vv)                                                                                                              \
    case Vv:                                                                                                                   \
        if (out_t == DType::F32)                                                                                               \
            hipLaunchKernelGGL((layernorm_kernel<float, Vv>), grid, dim3(256), 0, st, x, ldx, w, b, (float *)out, ldo, rows, d); \
        else                                                                                                                   \
            hipLaunchKernelGGL((layernorm_kernel<bf16_t, Vv>), grid, dim3(256), 0, st, x, ldx, w, b, (bf16_t *)out, ldo, rows, d); \
        break;
    switch (ceil_div(d, 256)) {  // float4 per lane
        MW_LN(1) MW_LN(2) MW_LN(3) MW_LN(4) MW_LN(5) MW_LN(6) MW_LN(7) MW_LN(8)
    default: MW_CHECK(false, "layernorm: d");
    }
#undef MW_LN
    MW_HIP(hipGetLastError());
}

void launch_softmax_rows(const float *s, long lds_, void *out, long ldo, DType out_t, long rows, int n, int n_pad,
                         float scale, hipStream_t st) {
    MW_CHECK(rows > 0 && n > 0 && n <= n_pad && n_pad <= 64 * 24, "softmax: row length out of range");
    MW_CHECK(ldo >= n_pad && lds_ >= n, "softmax: strides");
    dim3 grid((unsigned)ceil_div<long>(rows, 4));
    if (out_t == DType::F32)
        hipLaunchKernelGGL((softmax_rows_kernel<float, 24>), grid, dim3(256), 0, st, s, lds_, (float *)out, ldo, rows, n,
                           n_pad, scale);
    else
        hipLaunchKernelGGL((softmax_rows_kernel<bf16_t, 24>), grid, dim3(256), 0, st, s, lds_, (bf16_t *)out, ldo, rows, n,
                           n_pad, scale);
    MW_HIP(hipGetLastError());
}

void launch_fill_zero(void *p, size_t bytes, hipStream_t st) { MW_HIP(hipMemsetAsync(p, 0, bytes, st)); }

void launch_convert(const float *src, void *dst, DType dt, size_t n, hipStream_t st) {
    if (dt == DType::F32) {
        MW_HIP(hipMemcpyAsync(dst, src, n * 4, hipMemcpyDeviceToDevice, st));
        return;
    }
    dim3 grid((unsigned)std::min<size_t>(ceil_div<size_t>(n, 256), 4096));
    hipLaunchKernelGGL(convert_kernel, grid, dim3(256), 0, st, src, (bf16_t *)dst, n);
    MW_HIP(hipGetLastError());
}

}  // namespace mw
