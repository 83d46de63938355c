// Development microbenchmark + exactness check of the MX-fp8 encoder GEMM (gemm_mx.hip) at M = windows x 1500 rows:
//   make -C tools/microbench mb_gemm_mx && ./mb_gemm_mx [windows ...]
// Check: operands are quantised by the library's own kernels, read back, decoded on the host (e4m3 byte x 2^(scale - 127),
// scale found through mx_sa_off / mx_sw_off) and a sample of output elements is recomputed in double from those decoded
// values — so the product, the operand maps and the scale order are checked independently of the quantiser's rounding.
#include "../../obs-localvocal_amd/csrc/gemm.hip"
#include "../../obs-localvocal_amd/csrc/gemm_mx.hip"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

using namespace mw;

static float e4m3_to_float(unsigned char v) {
    const int s = v >> 7, e = (v >> 3) & 15, m = v & 7;
    float x;
    if (e == 0) x = ldexpf((float)m, -9);
    else if (e == 15 && m == 7) x = NAN;
    else x = ldexpf(1.0f + m / 8.0f, e - 7);
    return s ? -x : x;
}
static double gelu_ref(double x) { return 0.5 * x * (1.0 + tanh(0.79788456080286535588 * x * (1.0 + 0.044715 * x * x))); }

int main(int argc, char **argv) {
    const int d = 1280;
    struct Shape { const char *name; int N, K, epi; };
    const Shape shapes[] = {{"qk", 2 * d, d, MXE_BF16}, {"fc1", 4 * d, d, MXE_MX_GELU}, {"fc2", d, 4 * d, MXE_F32_RES}, {"ckv", d, d, MXE_BF16}};
    std::vector<int> Ws;
    for (int i = 1; i < argc; ++i) Ws.push_back(atoi(argv[i]));
    if (Ws.empty()) Ws = {1, 8, 16};
    size_t maxM = 1500;
    for (int w : Ws) maxM = std::max<size_t>(maxM, (size_t)w * 1500);
    const size_t maxMp = (maxM + 255) / 256 * 256;
    float *Af, *bias, *R;
    void *Wb, *C;
    unsigned char *A8, *SA, *W8, *SW, *SC;
    MW_HIP(hipMalloc((void **)&Af, maxM * 4 * d * 4));
    MW_HIP(hipMalloc(&Wb, (size_t)4 * d * d * 2));
    MW_HIP(hipMalloc(&C, maxMp * 4 * d * 4));
    MW_HIP(hipMalloc((void **)&R, maxMp * d * 4));
    MW_HIP(hipMalloc((void **)&bias, 4 * d * 4));
    MW_HIP(hipMalloc((void **)&A8, maxM * 4 * d));
    MW_HIP(hipMalloc((void **)&W8, (size_t)4 * d * d));
    MW_HIP(hipMalloc((void **)&SA, mx_scale_bytes_a((int)maxM, 4 * d)));
    MW_HIP(hipMalloc((void **)&SC, mx_scale_bytes_a((int)maxM, 4 * d)));
    MW_HIP(hipMalloc((void **)&SW, mx_scale_bytes_w(4 * d, 4 * d)));
    unsigned rng = 777;
    auto rnd = [&]() { rng = rng * 1664525u + 1013904223u; return (float)(rng >> 8) / 8388608.0f - 1.0f; };  // [-1, 1)
    std::vector<float> hA(maxM * 4 * d), hbias(4 * d), hR(maxM * d);
    std::vector<unsigned short> hW((size_t)4 * d * d);
    for (size_t i = 0; i < hA.size(); ++i) hA[i] = rnd() * (1.0f + 7.0f * ((i / 32) % 5 == 0)) * ((i % 977) == 0 ? 30.0f : 1.0f);  // blocks of mixed range
    for (auto &v : hW) v = f32_to_bf16_host(0.05f * rnd());
    for (auto &v : hbias) v = 0.3f * rnd();
    for (auto &v : hR) v = rnd();
    MW_HIP(hipMemcpy(Af, hA.data(), hA.size() * 4, hipMemcpyHostToDevice));
    MW_HIP(hipMemcpy(Wb, hW.data(), hW.size() * 2, hipMemcpyHostToDevice));
    MW_HIP(hipMemcpy(bias, hbias.data(), hbias.size() * 4, hipMemcpyHostToDevice));
    MW_HIP(hipMemcpy(R, hR.data(), hR.size() * 4, hipMemcpyHostToDevice));
    hipStream_t st;
    MW_HIP(hipStreamCreate(&st));
    hipEvent_t e0, e1;
    MW_HIP(hipEventCreate(&e0));
    MW_HIP(hipEventCreate(&e1));
    int rc = 0;
    for (int W : Ws)
        for (const Shape &s : shapes) {
            const int M = W * 1500, MB = mx_row_blocks_a(M), NB = s.N / 64;
            launch_quantize_mx_rows(Af, s.K, M, s.K, A8, s.K, SA, st);
            launch_quantize_mx_weights(Wb, s.N, s.K, W8, SW, st);
            GemmMxArgs g;
            g.A = A8; g.SA = SA; g.lda = s.K; g.W = W8; g.SW = SW; g.ldw = s.K; g.C = C; g.ldc = s.N; g.bias = bias;
            g.M = M; g.N = s.N; g.K = s.K;
            if (s.epi == MXE_F32_RES) { g.R = R; g.ldr = s.N; }
            if (s.epi == MXE_MX_GELU) g.SC = SC;
            const size_t obytes = (size_t)M * s.N * (s.epi == MXE_F32_RES ? 4 : s.epi == MXE_BF16 ? 2 : 1);
            MW_HIP(hipMemsetAsync(C, 0xff, obytes, st));
            launch_gemm_mx(g, s.epi, st);
            MW_HIP(hipStreamSynchronize(st));
            if (W == Ws[0] || W == Ws.back()) {  // exactness on a sample
                std::vector<unsigned char> a8((size_t)M * s.K), w8((size_t)s.N * s.K), sa(mx_scale_bytes_a(M, s.K)), sw(mx_scale_bytes_w(s.N, s.K)),
                    out(obytes), sc(s.epi == MXE_MX_GELU ? mx_scale_bytes_a(M, s.N) : 0);
                MW_HIP(hipMemcpy(a8.data(), A8, a8.size(), hipMemcpyDeviceToHost));
                MW_HIP(hipMemcpy(w8.data(), W8, w8.size(), hipMemcpyDeviceToHost));
                MW_HIP(hipMemcpy(sa.data(), SA, sa.size(), hipMemcpyDeviceToHost));
                MW_HIP(hipMemcpy(sw.data(), SW, sw.size(), hipMemcpyDeviceToHost));
                MW_HIP(hipMemcpy(out.data(), C, obytes, hipMemcpyDeviceToHost));
                if (!sc.empty()) MW_HIP(hipMemcpy(sc.data(), SC, sc.size(), hipMemcpyDeviceToHost));
                // the quantiser against the rule, on a sample of blocks
                int qbad = 0;
                for (int t = 0; t < 4000; ++t) {
                    const int m = (int)((unsigned)(t * 2654435761u) % (unsigned)M), kb = (t * 7) % (s.K / 32);
                    float amax = 0;
                    for (int k = 0; k < 32; ++k) amax = std::max(amax, fabsf(hA[(size_t)m * s.K + 32 * kb + k]));
                    int e = 0;
                    if (amax > 0) { (void)frexpf(amax / 448.0f, &e); if (ldexpf(448.0f, e - 1) >= amax) e -= 1; }
                    const int want = std::max(0, e + 127), got = sa[mx_sa_off(m, kb, MB)];
                    if (amax > 0 && want != got) { if (qbad < 3) printf("   scale of (%d, %d): %d, rule says %d (amax %g)\n", m, kb, got, want, amax); ++qbad; }
                    for (int k = 0; k < 32; ++k) {
                        const float x = hA[(size_t)m * s.K + 32 * kb + k], y = e4m3_to_float(a8[(size_t)m * s.K + 32 * kb + k]) * ldexpf(1.0f, got - 127);
                        if (fabsf(x - y) > ldexpf(1.0f, got - 127) * std::max(fabsf(x) * ldexpf(1.0f, 127 - got) / 16.0f, ldexpf(1.0f, -10)) * 1.0001f) {
                            if (qbad < 3) printf("   element (%d, %d): %g decoded %g\n", m, 32 * kb + k, x, y);
                            ++qbad;
                        }
                    }
                }
                int bad = 0;
                double worst = 0;
                for (int t = 0; t < 3000; ++t) {
                    int m = (int)((unsigned)(t * 2246822519u) % (unsigned)M), n = (int)((unsigned)(t * 3266489917u) % (unsigned)s.N);
                    if (t < 64) m = M - 1 - t;  // the partial last tile
                    double ref = 0, mag = 0;
                    for (int k = 0; k < s.K; ++k) {
                        const double av = (double)e4m3_to_float(a8[(size_t)m * s.K + k]) * ldexp(1.0, sa[mx_sa_off(m, k >> 5, MB)] - 127);
                        const double wv = (double)e4m3_to_float(w8[(size_t)n * s.K + k]) * ldexp(1.0, sw[mx_sw_off(n, k >> 5, NB)] - 127);
                        ref += av * wv;
                        mag += fabs(av * wv);
                    }
                    ref += hbias[n];
                    double got, tol = 1e-5 * mag + 1e-6;  // fp32 accumulation over up to 40 k-steps
                    if (s.epi == MXE_F32_RES) {
                        ref += hR[(size_t)m * s.N + n];
                        got = reinterpret_cast<const float *>(out.data())[(size_t)m * s.N + n];
                    } else if (s.epi == MXE_BF16) {
                        got = bf16_to_f32_host(reinterpret_cast<const bf16_t *>(out.data())[(size_t)m * s.N + n]);
                        tol += fabs(ref) / 256.0;
                    } else {
                        ref = gelu_ref(ref);
                        const int sb = sc[mx_sa_off(m, n >> 5, MB)];
                        got = (double)e4m3_to_float(out[(size_t)m * s.N + n]) * ldexp(1.0, sb - 127);
                        tol += std::max(fabs(ref) / 16.0, ldexp(1.0, sb - 127 - 10)) * 1.001 + 2e-6;
                    }
                    const double err = fabs(got - ref);
                    worst = std::max(worst, err / (tol + 1e-30));
                    if (!(err <= tol)) {
                        if (bad < 5) printf("   C[%d][%d] = %.7g, recomputed %.7g (tol %.3g)\n", m, n, got, ref, tol);
                        ++bad;
                    }
                }
                printf("check W=%d %-4s: %d of 3000 sampled outputs outside tolerance (worst err / tol %.3f), quantiser: %d findings\n", W, s.name, bad,
                       worst, qbad);
                rc |= bad != 0 || qbad != 0;
            }
            const int iters = 20;
            for (int i = 0; i < 3; ++i) launch_gemm_mx(g, s.epi, st);
            MW_HIP(hipEventRecord(e0, st));
            for (int i = 0; i < iters; ++i) launch_gemm_mx(g, s.epi, st);
            MW_HIP(hipEventRecord(e1, st));
            MW_HIP(hipEventSynchronize(e1));
            float ms = 0;
            MW_HIP(hipEventElapsedTime(&ms, e0, e1));
            const double us = ms * 1000.0 / iters;
            printf("W=%d %-4s M=%d N=%d K=%d  %8.1f us  %7.1f TFLOP/s\n", W, s.name, M, s.N, s.K, us, 2.0 * M * s.N * s.K / us * 1e-6);
        }
    // the LayerNorm producer
    {
        float *x, *lw, *lb;
        const int M = (int)maxM;
        MW_HIP(hipMalloc((void **)&x, (size_t)M * d * 4));
        MW_HIP(hipMalloc((void **)&lw, d * 4));
        MW_HIP(hipMalloc((void **)&lb, d * 4));
        MW_HIP(hipMemcpy(x, hA.data(), (size_t)M * d * 4, hipMemcpyHostToDevice));
        MW_HIP(hipMemcpy(lw, hbias.data(), d * 4, hipMemcpyHostToDevice));
        MW_HIP(hipMemcpy(lb, hbias.data() + d, d * 4, hipMemcpyHostToDevice));
        for (int with = 0; with < 2; ++with) {
            for (int i = 0; i < 3; ++i) launch_layernorm_mx(x, d, lw, lb, with ? C : nullptr, d, A8, d, SA, M, d, st);
            MW_HIP(hipEventRecord(e0, st));
            for (int i = 0; i < 20; ++i) launch_layernorm_mx(x, d, lw, lb, with ? C : nullptr, d, A8, d, SA, M, d, st);
            MW_HIP(hipEventRecord(e1, st));
            MW_HIP(hipEventSynchronize(e1));
            float ms = 0;
            MW_HIP(hipEventElapsedTime(&ms, e0, e1));
            printf("layernorm_mx M=%d %s: %.1f us\n", M, with ? "bf16 + mx" : "mx only", ms * 50.0);
        }
    }
    return rc;
}
