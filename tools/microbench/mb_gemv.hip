// Micro-benchmark of the decode-step weight-streaming projections (development tool, not product).
// Build: make -C tools/microbench ; run on the GPU box: tools/microbench/mb_gemv
#include "../../obs-localvocal_amd/csrc/kernels.hpp"

#include <vector>

using namespace mw;
namespace mw {
void log_msg(int, const char *, ...) {}
}  // namespace mw

using u32x4 = __attribute__((ext_vector_type(4))) unsigned;

// floor: stream N x K bf16 with the same decomposition (4 waves x R rows per block), no activations
template <int R, int NL>
__global__ __launch_bounds__(256) void stream_floor(const unsigned short *W, int N, int K, float *out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n_base = (blockIdx.x * 4 + wave) * R;
    u32x4 v[R][NL];
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            const int k = lane * 8 + i * 512;
            v[r][i] = u32x4{0, 0, 0, 0};
            if (k < K && n_base + r < N) v[r][i] = __builtin_nontemporal_load((const u32x4 *)(W + (long)(n_base + r) * K + k));
        }
    unsigned s = 0;
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int i = 0; i < NL; ++i) s += v[r][i].x ^ v[r][i].y ^ v[r][i].z ^ v[r][i].w;
    if (s == 0x12345678u) out[0] = 1.0f;
}

template <typename F> float time_it(hipStream_t st, int iters, F &&f) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int i = 0; i < 5; ++i) f(i);
    hipStreamSynchronize(st);
    hipEventRecord(e0, st);
    for (int i = 0; i < iters; ++i) f(i);
    hipEventRecord(e1, st);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return ms * 1000.0f / iters;
}

// few workgroups, each streaming a large contiguous slab: how fast can ONE CU pull from HBM?
template <int T, int NL>
__global__ __launch_bounds__(T) void slab_stream(const unsigned short *W, long slab_elems, float *out) {
    const unsigned short *base = W + (long)blockIdx.x * slab_elems;
    u32x4 v[NL];
#pragma unroll
    for (int i = 0; i < NL; ++i) {
        const long e = ((long)i * T + threadIdx.x) * 8;
        v[i] = u32x4{0, 0, 0, 0};
        if (e < slab_elems) v[i] = __builtin_nontemporal_load((const u32x4 *)(base + e));
    }
    unsigned s = 0;
#pragma unroll
    for (int i = 0; i < NL; ++i) s += v[i].x ^ v[i].y ^ v[i].z ^ v[i].w;
    if (s == 0x12345678u) out[threadIdx.x] = 1.0f;
}

int main(int argc, char **argv) {
    const int d = 1280, L = 32, rows = argc > 1 ? atoi(argv[1]) : 5;
    hipStream_t st;
    hipStreamCreate(&st);
    // one big weight pool: L matrices of 4d x d (fc1 / fc2 share the size), cycled so nothing stays cached
    const size_t per = (size_t)4 * d * d;
    unsigned short *W;
    hipMalloc(&W, per * L * 2);
    hipMemset(W, 0x11, per * L * 2);
    float *x, *out, *bias, *lnw, *lnb, *ffd;
    hipMalloc(&x, 8 * 4 * d * 4);
    hipMalloc(&ffd, 8 * 4 * d * 4);
    hipMalloc(&out, 8 * 4 * d * 4);
    hipMalloc(&bias, 4 * d * 4);
    hipMalloc(&lnw, d * 4);
    hipMalloc(&lnb, d * 4);
    hipMemset(x, 0, 8 * 4 * d * 4);
    hipMemset(ffd, 0, 8 * 4 * d * 4);
    hipMemset(out, 0, 8 * 4 * d * 4);
    hipMemset(bias, 0, 4 * d * 4);
    hipMemset(lnw, 0, d * 4);
    hipMemset(lnb, 0, d * 4);
    struct Case { const char *name; int tag, N, K; bool ln, res; };
    const Case cases[] = {{"attn_out", GEMV_ATTN_OUT, d, d, false, true}, {"cross_q(LN)", GEMV_CROSS_Q, d, d, true, false},
                          {"qkv(LN)", GEMV_QKV, 3 * d, d, true, false}, {"fc1(LN)", GEMV_FC1, 4 * d, d, true, false},
                          {"fc2", GEMV_FC2, d, 4 * d, false, true}};
    void *kc, *vc;
    hipMalloc(&kc, (size_t)8 * 448 * d * 2);
    hipMalloc(&vc, (size_t)8 * 448 * d * 2);
    int *slot, *pos;
    hipMalloc(&slot, 64);
    hipMalloc(&pos, 64);
    hipMemset(slot, 0, 64);
    hipMemset(pos, 0, 64);
    for (const Case &c : cases) {
        const double bytes = (double)c.N * c.K * 2;
        float us = time_it(st, 200, [&](int i) {
            GemvArgs g;
            g.tag = c.tag; g.x = c.K == d ? x : ffd; g.ldx = c.K; g.n = rows; g.W = W + per * (i % L); g.bias = bias;
            g.N = c.N; g.K = c.K; g.out = out; g.ldo = c.tag == GEMV_QKV ? d : c.N;
            if (c.ln) { g.ln_w = lnw; g.ln_b = lnb; }
            if (c.res) g.R = out;
            if (c.tag == GEMV_QKV) { g.qkv = 1; g.kcache = kc; g.vcache = vc; g.kv_t = DType::BF16; g.d_slot = slot; g.d_pos = pos; g.n_text_ctx = 448; }
            launch_gemv(g, DType::BF16, st);
        });
        printf("%-12s N=%5d K=%5d rows=%d : %7.2f us/launch  %7.1f GB/s\n", c.name, c.N, c.K, rows, us, bytes / us / 1e3);
    }
    // matrix-core path (dec_mfma.hip)
    unsigned short *xb;
    hipMalloc(&xb, 8 * 4 * d * 2);
    hipMemset(xb, 0, 8 * 4 * d * 2);
    for (const Case &c : cases) {
        const double bytes = (double)c.N * c.K * 2;
        const bool direct = c.tag == GEMV_ATTN_OUT || c.tag == GEMV_FC2;
        float us = time_it(st, 200, [&](int i) {
            GemvArgs g;
            g.tag = c.tag; g.x = direct ? (const float *)xb : x; g.x_bf16 = direct; g.ldx = c.K; g.n = rows; g.W = W + per * (i % L); g.bias = bias;
            g.N = c.N; g.K = c.K; g.out = out; g.ldo = c.tag == GEMV_QKV ? d : c.N;
            if (c.ln) { g.ln_w = lnw; g.ln_b = lnb; }
            if (c.res) g.R = out;
            if (c.tag == GEMV_QKV) { g.qkv = 1; g.kcache = kc; g.vcache = vc; g.kv_t = DType::BF16; g.d_slot = slot; g.d_pos = pos; g.n_text_ctx = 448; }
            launch_gemv_mfma(g, st);
        });
        printf("mfma %-12s N=%5d K=%5d rows=%d : %7.2f us/launch  %7.1f GB/s\n", c.name, c.N, c.K, rows, us, bytes / us / 1e3);
    }
    // floors
    {
        float us = time_it(st, 200, [&](int i) { hipLaunchKernelGGL((stream_floor<1, 3>), dim3(d / 4), dim3(256), 0, st, W + per * (i % L), d, d, out); });
        printf("floor N=d K=d R=1        : %7.2f us  %7.1f GB/s\n", us, (double)d * d * 2 / us / 1e3);
        us = time_it(st, 200, [&](int i) { hipLaunchKernelGGL((stream_floor<2, 3>), dim3(4 * d / 8), dim3(256), 0, st, W + per * (i % L), 4 * d, d, out); });
        printf("floor N=4d K=d R=2       : %7.2f us  %7.1f GB/s\n", us, (double)4 * d * d * 2 / us / 1e3);
        us = time_it(st, 200, [&](int i) { hipLaunchKernelGGL((stream_floor<4, 3>), dim3(4 * d / 16), dim3(256), 0, st, W + per * (i % L), 4 * d, d, out); });
        printf("floor N=4d K=d R=4       : %7.2f us  %7.1f GB/s\n", us, (double)4 * d * d * 2 / us / 1e3);
        us = time_it(st, 200, [&](int i) { hipLaunchKernelGGL((stream_floor<1, 10>), dim3(d / 4), dim3(256), 0, st, W + per * (i % L), d, 4 * d, out); });
        printf("floor N=d K=4d R=1       : %7.2f us  %7.1f GB/s\n", us, (double)4 * d * d * 2 / us / 1e3);
        {
            const long slab = 384 * 1024 / 2;  // 384 KB per workgroup (one head's cross K+V)
            float u2 = time_it(st, 200, [&](int i) { hipLaunchKernelGGL((slab_stream<1024, 24>), dim3(20), dim3(1024), 0, st, W + per * (i % L), slab, out); });
            printf("slab 20 wg x 384 KB (1024 thr): %7.2f us  %7.1f GB/s per wg\n", u2, 384.0 * 1024 / u2 / 1e3);
            u2 = time_it(st, 200, [&](int i) { hipLaunchKernelGGL((slab_stream<1024, 12>), dim3(40), dim3(1024), 0, st, W + per * (i % L), slab / 2, out); });
            printf("slab 40 wg x 192 KB (1024 thr): %7.2f us  %7.1f GB/s per wg\n", u2, 192.0 * 1024 / u2 / 1e3);
            u2 = time_it(st, 200, [&](int i) { hipLaunchKernelGGL((slab_stream<512, 12>), dim3(80), dim3(512), 0, st, W + per * (i % L), slab / 4, out); });
            printf("slab 80 wg x  96 KB ( 512 thr): %7.2f us  %7.1f GB/s per wg\n", u2, 96.0 * 1024 / u2 / 1e3);
            u2 = time_it(st, 200, [&](int i) { hipLaunchKernelGGL((slab_stream<512, 4>), dim3(240), dim3(512), 0, st, W + per * (i % L), slab / 12, out); });
            printf("slab 240 wg x 32 KB ( 512 thr): %7.2f us  %7.1f GB/s per wg\n", u2, 32.0 * 1024 / u2 / 1e3);
        }
        us = time_it(st, 200, [&](int i) { hipLaunchKernelGGL((stream_floor<1, 1>), dim3(1), dim3(64), 0, st, W, 1, 8, out); });
        printf("empty-ish kernel         : %7.2f us\n", us);
    }
    return 0;
}
