// Phase timing of the beam-step cross-attention kernel (development tool, not product): the product's
// cross_attn_lean body with wall_clock64 stamps kept in LDS; launched behind a producer that rewrites q.
// Build: make -C tools/microbench mb_cross ; run on the GPU box: tools/microbench/mb_cross
#include "../../obs-localvocal_amd/csrc/mega.hip"

#include <algorithm>
#include <cstdio>
#include <vector>

namespace mw {
void log_msg(int, const char *, ...) {}
}  // namespace mw
using namespace mw;

__global__ void producer(float *x, int cnt, float v) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < cnt) x[i] = 0.01f * (float)((i + (int)v) % 97);
}

template <int NB>
__global__ __launch_bounds__(MEGA_THREADS) void cross_probe(const float *q, const bf16_t *kc, const bf16_t *vc, float *part, bf16_t *out,
                                                            unsigned *head_cnt, int n, int H, int d, int n_ctx, int splits, int per,
                                                            unsigned long long *stamps) {
    __shared__ __attribute__((aligned(16))) float lds[(16 * 72 + XK * XKD + 64 * XLDP + 16 * XLDP) / 2 + NB * XK +
                                                      2 * NB * CROSS_SPLITS_MAX + 8 * 64 * 4 + NB * 2 + 16];
    __shared__ unsigned long long s[2];
    if (threadIdx.x == 0) s[0] = wall_clock64();
    cross_attn_lean<NB, bf16_t>(q, kc, vc, part, out, head_cnt, n, H, d, n_ctx, splits, per, lds);
    __syncthreads();
    if (threadIdx.x == 0) {
        s[1] = wall_clock64();
        stamps[blockIdx.x * 2] = s[0];
        stamps[blockIdx.x * 2 + 1] = s[1];
    }
}

int main() {
    hipStream_t st;
    hipStreamCreate(&st);
    const int d = 1280, H = 20, n = 5, n_ctx = 1500, splits = 12, per = 125, L = 32;
    float *q, *part;
    bf16_t *kv, *out;
    unsigned *cnt;
    unsigned long long *stamps;
    hipMalloc(&q, n * d * 4);
    hipMalloc(&part, (size_t)n * H * CROSS_SPLITS_MAX * 66 * 4);
    hipMalloc(&kv, (size_t)L * 2 * n_ctx * d * 2);
    hipMemset(kv, 0x3c, (size_t)L * 2 * n_ctx * d * 2);
    hipMalloc(&out, n * d * 2);
    hipMalloc(&cnt, 64 * 4);
    hipMemset(cnt, 0, 64 * 4);
    hipMalloc(&stamps, 1024 * 16);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int iters = 200, wgs = H * splits;
    for (int it = 0; it < iters + 10; ++it) {
        if (it == 10) hipEventRecord(e0, st);
        const bf16_t *k = kv + (size_t)(it % L) * 2 * n_ctx * d;
        hipLaunchKernelGGL(producer, dim3((n * d + 255) / 256), dim3(256), 0, st, q, n * d, (float)it);
        hipLaunchKernelGGL((cross_probe<5>), dim3(wgs), dim3(MEGA_THREADS), 0, st, q, k, k + (size_t)n_ctx * d, part, out, cnt, n, H, d, n_ctx,
                           splits, per, stamps);
    }
    hipEventRecord(e1, st);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(wgs * 2);
    hipMemcpy(h.data(), stamps, h.size() * 8, hipMemcpyDeviceToHost);
    unsigned long long t0 = ~0ull;
    for (int b = 0; b < wgs; ++b) t0 = std::min(t0, h[b * 2]);
    std::vector<double> ex(wgs);
    for (int b = 0; b < wgs; ++b) ex[b] = (double)(long long)(h[b * 2 + 1] - t0) / 100.0;
    std::sort(ex.begin(), ex.end());
    printf("%.2f us per (producer + cross-attention) pair; exit since first entry: min %.2f p50 %.2f p90 %.2f max %.2f us\n",
           ms * 1000.0 / iters, ex[0], ex[wgs / 2], ex[wgs * 9 / 10], ex[wgs - 1]);
    // the last-arriving split of a head exits last: the 20 largest exits are the combiners
    printf("combiners (20 largest): %.2f .. %.2f us; others p50 %.2f us\n", ex[wgs - 20], ex[wgs - 1], ex[(wgs - 20) / 2]);
    return 0;
}
