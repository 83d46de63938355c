// Where does the time of a 5-row decode projection go?  (development tool, not product)
// Runs the product's proj_phase inside an instrumented kernel: every workgroup stamps wall_clock64 (100 MHz)
// at entry, after staging, after the barrier, after the products and after its stores.  The kernel is
// launched behind a small producer kernel that rewrites the activation rows (so they arrive from another
// XCD as in the decode step) and over 32 rotating weight buffers (so nothing is cache resident).
// Build: make -C tools/microbench mb_phases ; run on the GPU box: tools/microbench/mb_phases
#include "../../obs-localvocal_amd/csrc/mega.hip"

#include <algorithm>
#include <cstdio>
#include <vector>

namespace mw {
void log_msg(int, const char *, ...) {}
}  // namespace mw
using namespace mw;

constexpr int NST = 8;

template <int NB, bool LN, int EPI, int J>
__global__ __launch_bounds__(MEGA_THREADS) void proj_probe(Proj p, const void *xin, const float *ln_w, const float *ln_b, int n, Epi e,
                                                           unsigned long long *stamps) {
    __shared__ __attribute__((aligned(16))) float lds[MEGA_XS_BYTES / 4 + 8 * 64 * 4];
    __shared__ unsigned long long s[NST];  // stamps stay in LDS until the end: a global store at entry would be
                                           // waited for (vmcnt) together with the first loads and distort them
    if (threadIdx.x == 0) s[4] = wall_clock64();
    WRegs<1, J> w;
    proj_phase<NB, LN, EPI, 1, J, true>(s, w, p, xin, ln_w, ln_b, n, e, lds);
    __syncthreads();  // waits for the stores of the whole workgroup
    if (threadIdx.x == 0) s[5] = wall_clock64();
    if (threadIdx.x == 0)
        for (int k = 0; k < 6; ++k) stamps[(size_t)blockIdx.x * NST + k] = s[k];
}

template <int NB, int EPI>
__global__ __launch_bounds__(MEGA_THREADS) void proj_probe_ln(Proj p, const void *xin, const float *ln_w, const float *ln_b, int n, Epi e,
                                                              unsigned long long *stamps) {
    __shared__ __attribute__((aligned(16))) bf16_t xs[NB * (256 * LN_MAX_F4 + 8)];
    __shared__ __attribute__((aligned(16))) bf16_t wt[16 * (256 * LN_MAX_F4 + 8)];
    __shared__ __attribute__((aligned(16))) float red[8 * 64 * 4];
    __shared__ unsigned long long s[NST];  // stamps stay in LDS until the end: a global store at entry would be
                                           // waited for (vmcnt) together with the first loads and distort them
    if (threadIdx.x == 0) s[4] = wall_clock64();
    proj_ln_tile<NB, EPI, 16>(p, (const float *)xin, ln_w, ln_b, n, e, xs, wt, red, s);
    __syncthreads();
    if (threadIdx.x == 0) s[5] = wall_clock64();
    if (threadIdx.x == 0)
        for (int k = 0; k < 6; ++k) stamps[(size_t)blockIdx.x * NST + k] = s[k];
}

template <int NB, int SPC, int CH, int FR>
__global__ __launch_bounds__(MEGA_THREADS) void proj_probe_res(Proj p, const bf16_t *xin, int n, Epi e, unsigned long long *stamps) {
    __shared__ __attribute__((aligned(16))) bf16_t lds_t[8 * (16 * (32 * SPC + 8) + NB * (CH * 32 * SPC + 8))];
    __shared__ __attribute__((aligned(16))) float red[8 * 64 * 4];
    __shared__ unsigned long long s[NST];  // stamps stay in LDS until the end: a global store at entry would be
                                           // waited for (vmcnt) together with the first loads and distort them
    if (threadIdx.x == 0) s[4] = wall_clock64();
    proj_res_tile<NB, SPC, CH, FR>(p, xin, n, e, lds_t, red, s);
    __syncthreads();
    if (threadIdx.x == 0) s[5] = wall_clock64();
    if (threadIdx.x == 0)
        for (int k = 0; k < 6; ++k) stamps[(size_t)blockIdx.x * NST + k] = s[k];
}

// warms the cache levels with the weights of the next projection: workgroup b reads the rows workgroup b of the
// consumer will read (same blockIdx -> same XCD, if the dispatcher's round-robin holds), `mode` 2 reads them rotated
// by one workgroup (lands in another XCD's L2: memory-side cache only)
__global__ __launch_bounds__(512) void prefetch_rows(const bf16_t *W, long tile_elems, int mode, float *sink) {
    const int b = mode == 2 ? (blockIdx.x + 1) % gridDim.x : blockIdx.x;
    const uint4 *src = reinterpret_cast<const uint4 *>(W + (long)b * tile_elems);
    unsigned acc = 0;
    for (long i = threadIdx.x; i < tile_elems / 8; i += 512) {
        const uint4 v = src[i];
        acc += v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345u) sink[0] = 1.0f;
}

__global__ void producer(float *x, int cnt, float v, unsigned long long *t) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < cnt) x[i] = v + 0.001f * (float)(i % 97);
    if (i == 0) t[0] = wall_clock64();
}
__global__ void producer_bf16(bf16_t *x, int cnt, unsigned long long *t) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
#ifdef WIDE_PRODUCER
    if (i < cnt / 8) reinterpret_cast<uint4 *>(x)[i] = make_uint4(0x3c003c01u + i, 0x3c003c01u, 0x3c003c01u, 0x3c003c01u);
#else
    if (i < cnt) x[i] = (bf16_t)(0x3c00 + (i % 64));
#endif
    if (i == 0) t[0] = wall_clock64();
}

static void report(const char *name, const std::vector<unsigned long long> &st, int wgs, unsigned long long tprod) {
    // all times in microseconds relative to the earliest workgroup entry
    unsigned long long t0 = ~0ull;
    for (int b = 0; b < wgs; ++b) t0 = std::min(t0, st[(size_t)b * NST + 4]);
    auto pct = [&](int idx, int rel, double q) {
        std::vector<double> v(wgs);
        for (int b = 0; b < wgs; ++b) {
            const unsigned long long a = st[(size_t)b * NST + idx], r = rel < 0 ? t0 : st[(size_t)b * NST + rel];
            v[b] = (double)(long long)(a - r) / 100.0;
        }
        std::sort(v.begin(), v.end());
        return v[std::min(wgs - 1, (int)(q * (wgs - 1) + 0.5))];
    };
    printf("%s: %d workgroups; producer stamp -> first entry %.2f us\n", name, wgs, (double)(long long)(t0 - tprod) / 100.0);
    const char *lbl[] = {"entry      (since first entry)", "staged     (since own entry)", "barrier    (since staged)",
                         "products   (since barrier)", "stores out (since products)", "stores ack (since issue)", "exit       (since first entry)"};
    const int idx[] = {4, 0, 1, 2, 3, 5, 5}, rel[] = {-1, 4, 0, 1, 2, 3, -1};
    for (int k = 0; k < 7; ++k)
        printf("  %-32s min %6.2f  p50 %6.2f  p90 %6.2f  max %6.2f\n", lbl[k], pct(idx[k], rel[k], 0.0), pct(idx[k], rel[k], 0.5),
               pct(idx[k], rel[k], 0.9), pct(idx[k], rel[k], 1.0));
}

int main() {
    hipStream_t st;
    hipStreamCreate(&st);
    const int d = 1280, n = 5, NBUF = 32;
    struct Case {
        const char *name;
        int N, K;
        bool ln;
        int epi;
        bool lean = false;
    } cases[] = {{"fc1 lean (LN, gelu) N=5120 K=1280", 5120, 1280, true, EPI_GELU_BF16, true},
                 {"qkv-like lean (LN, scale) N=3840 K=1280", 3840, 1280, true, EPI_SCALE, true},
                 {"cq lean (LN, scale) N=1280 K=1280", 1280, 1280, true, EPI_SCALE, true},
                 {"fc1  (LN, gelu) N=5120 K=1280", 5120, 1280, true, EPI_GELU_BF16},
                 {"qkv-like (LN, scale) N=3840 K=1280", 3840, 1280, true, EPI_SCALE},
                 {"cq   (LN, scale) N=1280 K=1280", 1280, 1280, true, EPI_SCALE},
                 {"out lean (res) N=1280 K=1280", 1280, 1280, false, EPI_RES, true},
                 {"fc2 lean (res) N=1280 K=5120", 1280, 5120, false, EPI_RES, true},
                 {"out  (res) N=1280 K=1280", 1280, 1280, false, EPI_RES},
                 {"fc2  (res) N=1280 K=5120", 1280, 5120, false, EPI_RES}};
    bf16_t *W;
    const size_t wmax = (size_t)5120 * 1280;
    hipMalloc(&W, wmax * 2 * NBUF);
    hipMemset(W, 0x11, wmax * 2 * NBUF);
    float *x, *lnw, *lnb, *bias, *res, *outf;
    bf16_t *xb, *outb;
    hipMalloc(&x, n * 5120 * 4);
    hipMalloc(&xb, n * 5120 * 2);
    hipMalloc(&lnw, 5120 * 4);
    hipMalloc(&lnb, 5120 * 4);
    hipMalloc(&bias, 5120 * 4);
    hipMalloc(&res, n * 5120 * 4);
    hipMalloc(&outf, n * 5120 * 4);
    hipMalloc(&outb, n * 5120 * 2);
    hipMemset(lnw, 0, 5120 * 4);
    hipMemset(lnb, 0, 5120 * 4);
    hipMemset(bias, 0, 5120 * 4);
    hipMemset(res, 0, n * 5120 * 4);
    unsigned long long *stamps, *tprod;
    hipMalloc(&stamps, 512 * NST * 8);
    hipMalloc(&tprod, 8);
    (void)d;
    for (const Case &c : cases) {
        const int wgs = c.N / 16;
        Epi e{};
        e.fout = outf;
        e.hout = outb;
        e.ldo = c.N;
        e.scale = 1.0f;
        e.res = res;
        e.xout = outf;
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        const int iters = 200;
        float ms = 0;
        for (int it = 0; it < iters + 10; ++it) {
            if (it == 10) hipEventRecord(e0, st);
            Proj p{W + (size_t)(it % NBUF) * wmax, bias, c.N, c.K, 0};
#ifdef PREFETCH
            if (c.lean) {
                const int fr = (!c.ln && c.K > 1280) ? 8 : 16;
                hipLaunchKernelGGL(prefetch_rows, dim3(c.N / fr), dim3(512), 0, st, p.W, (long)fr * c.K, PREFETCH, outf + 4096);
            }
#endif
            if (c.ln) {
                hipLaunchKernelGGL(producer, dim3((n * c.K + 255) / 256), dim3(256), 0, st, x, n * c.K, (float)it, tprod);
                if (c.lean && c.epi == EPI_GELU_BF16)
                    hipLaunchKernelGGL((proj_probe_ln<5, EPI_GELU_BF16>), dim3(wgs), dim3(MEGA_THREADS), 0, st, p, (const void *)x, lnw, lnb, n,
                                       e, stamps);
                else if (c.lean)
                    hipLaunchKernelGGL((proj_probe_ln<5, EPI_SCALE>), dim3(wgs), dim3(MEGA_THREADS), 0, st, p, (const void *)x, lnw, lnb, n, e,
                                       stamps);
                else if (c.epi == EPI_GELU_BF16)
                    hipLaunchKernelGGL((proj_probe<5, true, EPI_GELU_BF16, 5>), dim3(wgs), dim3(MEGA_THREADS), 0, st, p, (const void *)x, lnw,
                                       lnb, n, e, stamps);
                else
                    hipLaunchKernelGGL((proj_probe<5, true, EPI_SCALE, 5>), dim3(wgs), dim3(MEGA_THREADS), 0, st, p, (const void *)x, lnw, lnb,
                                       n, e, stamps);
            } else {
#ifdef F32_PRODUCER
                hipLaunchKernelGGL(producer, dim3((n * c.K / 2 + 255) / 256), dim3(256), 0, st, (float *)xb, n * c.K / 2, (float)it, tprod);
#else
                hipLaunchKernelGGL(producer_bf16, dim3((n * c.K + 255) / 256), dim3(256), 0, st, xb, n * c.K, tprod);
#endif
                if (c.lean && c.K <= 1280)
                    hipLaunchKernelGGL((proj_probe_res<5, 5, 1, 16>), dim3(wgs), dim3(MEGA_THREADS), 0, st, p, (const bf16_t *)xb, n, e, stamps);
                else if (c.lean)
                    hipLaunchKernelGGL((proj_probe_res<5, 5, 4, 8>), dim3(2 * wgs), dim3(MEGA_THREADS), 0, st, p, (const bf16_t *)xb, n, e, stamps);
                else if (c.K <= 1280)
                    hipLaunchKernelGGL((proj_probe<5, false, EPI_RES, 5>), dim3(wgs), dim3(MEGA_THREADS), 0, st, p, (const void *)xb, nullptr,
                                       nullptr, n, e, stamps);
                else
                    hipLaunchKernelGGL((proj_probe<5, false, EPI_RES, 20>), dim3(wgs), dim3(MEGA_THREADS), 0, st, p, (const void *)xb, nullptr,
                                       nullptr, n, e, stamps);
            }
        }
        hipEventRecord(e1, st);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> h((size_t)wgs * NST);
        unsigned long long tp = 0;
        hipMemcpy(h.data(), stamps, h.size() * 8, hipMemcpyDeviceToHost);
        hipMemcpy(&tp, tprod, 8, hipMemcpyDeviceToHost);
        printf("\n%.2f us per (producer + projection) pair, eager back-to-back\n", ms * 1000.0 / iters);
        report(c.name, h, wgs, tp);
    }
    return 0;
}
