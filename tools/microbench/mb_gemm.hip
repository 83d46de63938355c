// Development microbenchmark: the encoder's GEMM shapes through launch_gemm at M = windows x 1500 rows.
//   make -C tools/microbench mb_gemm && ./mb_gemm [variant [windows ...]]     (variant = GemmArgs::glds: 0 launch_gemm's own choice, 8 the 256 x 256 kernel)
#include "../../obs-localvocal_amd/csrc/gemm.hip"

#include <cstdio>
#include <cstdlib>
#include <vector>

using namespace mw;

int main(int argc, char **argv) {
    const int variant = argc > 1 ? atoi(argv[1]) : 0;
    const int d = 1280;
    struct Shape { const char *name; int N, K; bool f32out, gelu, res; };
    const Shape shapes[] = {{"qk", 2 * d, d, false, false, false}, {"out", d, d, true, false, true},
                            {"fc1", 4 * d, d, false, true, false}, {"fc2", d, 4 * d, true, false, true},
                            {"fc1n", 4 * d, d, false, false, false}, {"outn", d, d, true, false, false},
                            {"outb", d, d, false, false, false}, {"fc2b", d, 4 * d, false, false, false}};
    std::vector<int> Ws;
    for (int i = 2; i < argc; ++i) Ws.push_back(atoi(argv[i]));
    if (Ws.empty()) Ws = {1, 4, 8};
    size_t maxM = 1500;
    for (int w : Ws) maxM = std::max<size_t>(maxM, (size_t)w * 1500);
    void *A, *Wt, *C; float *bias;
    MW_HIP(hipMalloc(&A, maxM * 4 * d * 2)); MW_HIP(hipMalloc(&Wt, (size_t)4 * d * d * 2)); MW_HIP(hipMalloc(&C, maxM * 4 * d * 4));
    MW_HIP(hipMalloc((void **)&bias, 4 * d * 4));
    std::vector<unsigned short> h(maxM * 4 * d);
    for (size_t i = 0; i < h.size(); ++i) h[i] = 0x3c00 + (unsigned short)((i * 2654435761u) >> 24);  // small finite bf16
    MW_HIP(hipMemcpy(A, h.data(), h.size() * 2, hipMemcpyHostToDevice));
    MW_HIP(hipMemcpy(Wt, h.data(), (size_t)4 * d * d * 2, hipMemcpyHostToDevice));
    MW_HIP(hipMemset(bias, 0, 4 * d * 4)); MW_HIP(hipMemset(C, 0, maxM * 4 * d * 4));
    hipStream_t st; MW_HIP(hipStreamCreate(&st));
    hipEvent_t e0, e1; MW_HIP(hipEventCreate(&e0)); MW_HIP(hipEventCreate(&e1));
    // correctness of the variant against variant 0 (same k order: bit-identical), on the W = 1 shapes without residual aliasing
    if (variant != 0) {
        void *C0;
        MW_HIP(hipMalloc(&C0, (size_t)1500 * 4 * d * 4));
        for (const Shape &s : shapes) {
            GemmArgs g;
            g.A = A; g.lda = s.K; g.W = Wt; g.ldw = s.K; g.ldc = s.N; g.bias = bias;
            g.M = 1500; g.N = s.N; g.K = s.K; g.act = s.gelu ? ACT_GELU : ACT_NONE;
            const size_t bytes = (size_t)g.M * s.N * (s.f32out ? 4 : 2);
            std::vector<unsigned char> h0(bytes), h1(bytes);
            g.C = C0; g.glds = 0;
            launch_gemm(g, DType::BF16, s.f32out ? DType::F32 : DType::BF16, st);
            MW_HIP(hipStreamSynchronize(st));
            MW_HIP(hipMemcpy(h0.data(), C0, bytes, hipMemcpyDeviceToHost));
            MW_HIP(hipMemset(C0, 0xff, bytes));
            g.glds = variant;
            launch_gemm(g, DType::BF16, s.f32out ? DType::F32 : DType::BF16, st);
            MW_HIP(hipStreamSynchronize(st));
            MW_HIP(hipMemcpy(h1.data(), C0, bytes, hipMemcpyDeviceToHost));
            size_t bad = 0;
            for (size_t i = 0; i < bytes; ++i) bad += h0[i] != h1[i];
            printf("check %-6s variant %d vs 0: %zu differing bytes of %zu\n", s.name, variant, bad, bytes);
        }
        MW_HIP(hipFree(C0));
    }
    for (int W : Ws)
        for (const Shape &s : shapes) {
            GemmArgs g;
            g.A = A; g.lda = s.K; g.W = Wt; g.ldw = s.K; g.C = C; g.ldc = s.N; g.bias = bias;
            g.M = W * 1500; g.N = s.N; g.K = s.K; g.act = s.gelu ? ACT_GELU : ACT_NONE;
            if (s.res) { g.R = (float *)C; g.ldr = s.N; }
            g.glds = variant;
            const int it = 20;
            for (int i = 0; i < 3; ++i) launch_gemm(g, DType::BF16, s.f32out ? DType::F32 : DType::BF16, st);
            MW_HIP(hipEventRecord(e0, st));
            for (int i = 0; i < it; ++i) launch_gemm(g, DType::BF16, s.f32out ? DType::F32 : DType::BF16, st);
            MW_HIP(hipEventRecord(e1, st));
            MW_HIP(hipEventSynchronize(e1));
            float ms; MW_HIP(hipEventElapsedTime(&ms, e0, e1));
            const double us = ms * 1e3 / it, tf = 2.0 * g.M * s.N * s.K / (us * 1e-6) / 1e12;
            printf("W=%d %-10s M=%5d N=%4d K=%4d  %8.1f us  %7.1f TFLOP/s\n", W, s.name, g.M, s.N, s.K, us, tf);
        }
    return 0;
}
