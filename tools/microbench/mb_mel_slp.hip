// mb_mel_slp — what makes mel_frames_kernel return wrong values beside other kernels when hipcc's SLP vectoriser
// (packed-f32 VALU, v_pk_*_f32) is on?  
//
// The product kernel is compiled into this probe (#include of mel.hip) in several variants (Makefile: SLP on / off, the
// opaque() hooks of the FFT stages and of the power-spectrum loop on / off).  A sparse synthetic filter table makes mel bin
// j the sum of power bins 4j .. 4j+3 only, so a wrong output value names the four power bins it came from.  The kernel
// runs once alone (reference), then `reps` times beside one kind of neighbour kernel on other streams:
//   none   nothing else on the GPU
//   mfma   bf16 matrix-core loops (what the encoders of other contexts run)
//   valu   fp32 VALU loops that leave NaN bit patterns in every VGPR they can reach when they exit
//   lds    LDS read / write loops over a 64 KB allocation filled with NaN patterns
// and every output word is compared with the reference.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../obs-localvocal_amd/csrc/mel.hip"

using namespace mw;

#define CK(x)                                                                              \
    do {                                                                                   \
        hipError_t e_ = (x);                                                               \
        if (e_ != hipSuccess) {                                                            \
            fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));      \
            return 2;                                                                      \
        }                                                                                  \
    } while (0)

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

__global__ __launch_bounds__(256) void noise_mfma(float *sink, int iters) {
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) {
        a[i] = (__bf16)(0.001f * (threadIdx.x + i));
        b[i] = (__bf16)(0.002f * (threadIdx.x - i));
    }
    f32x4 acc[4] = {};
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[j], 0, 0, 0);
    float s = 0.0f;
    for (int j = 0; j < 4; ++j) s += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
    if (s == 123.456f) sink[0] = s;
}

// leaves NaN patterns in many VGPRs: 96 live values that are all quiet / signalling NaNs at exit
__global__ __launch_bounds__(256) void noise_valu(float *sink, int iters) {
    float v[96];
#pragma unroll
    for (int i = 0; i < 96; ++i) v[i] = __uint_as_float(0x7fc00000u + threadIdx.x * 97u + i);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 96; ++i) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(v[i]) : "v"(it & 0xffff));
    }
    unsigned x = 0;
#pragma unroll
    for (int i = 0; i < 96; ++i) x ^= __float_as_uint(v[i]);
    if (x == 0x12345678u) sink[1] = 1.0f;
}

__global__ __launch_bounds__(256) void noise_lds(float *sink, int iters) {
    __shared__ unsigned buf[16384];
    for (int i = threadIdx.x; i < 16384; i += 256) buf[i] = 0x7fc00000u + i;
    __syncthreads();
    unsigned x = 0;
    for (int it = 0; it < iters; ++it) {
        const int i = (threadIdx.x * 33 + it * 257) & 16383;
        x ^= buf[i];
        buf[(i + 4099) & 16383] = x | 0x7f800001u;
    }
    if (x == 0x12345678u) sink[2] = 1.0f;
}

int main(int argc, char **argv) {
    const int reps = argc > 1 ? atoi(argv[1]) : 200;
    const char *label = argc > 2 ? argv[2] : "variant";
    const int n_mel = 51, n_samples = 8 * 16000;
    MelTables tab;
    memset(&tab, 0, sizeof(tab));
    for (int i = 0; i < MEL_N_FFT; ++i) {
        tab.tw_cos[i] = (float)cos(2.0 * M_PI * i / MEL_N_FFT);
        tab.tw_sin[i] = (float)sin(2.0 * M_PI * i / MEL_N_FFT);
        tab.hann[i] = (float)(0.5 * (1.0 - cos(2.0 * M_PI * i / MEL_N_FFT)));
    }
    tab.n_mel = n_mel;
    for (int j = 0; j < n_mel; ++j) {
        tab.k0[j] = 4 * j;
        tab.ngrp[j] = 1;
        tab.voff[j] = 4 * j;
        tab.last[j] = 0.0f;
        for (int t = 0; t < 4; ++t) tab.vals[4 * j + t] = 1.0f;
    }
    std::vector<float> pcm(n_samples);
    unsigned s = 12345;
    for (int i = 0; i < n_samples; ++i) {
        s = s * 1664525u + 1013904223u;
        pcm[i] = 0.2f * ((float)(s >> 8) / 8388608.0f - 1.0f) * (0.3f + 0.7f * (float)sin(i * 0.0007));
    }
    const int n_len = (n_samples + 30 * 16000) / MEL_HOP;
    const int n_active = std::min((n_samples + MEL_N_FFT / 2) / MEL_HOP + 1, n_len);
    float *d_pcm, *d_raw, *d_sink;
    MelTables *d_tab;
    unsigned *d_gmax;
    CK(hipMalloc((void **)&d_pcm, n_samples * 4));
    CK(hipMalloc((void **)&d_raw, (size_t)n_active * n_mel * 4));
    CK(hipMalloc((void **)&d_tab, sizeof(tab)));
    CK(hipMalloc((void **)&d_gmax, 64));
    CK(hipMalloc((void **)&d_sink, 64));
    CK(hipMemcpy(d_pcm, pcm.data(), n_samples * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_tab, &tab, sizeof(tab), hipMemcpyHostToDevice));
    hipStream_t st, ns[3];
    CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    for (auto &q : ns) CK(hipStreamCreateWithFlags(&q, hipStreamNonBlocking));
    const size_t nw = (size_t)n_active * n_mel;
    std::vector<float> ref(nw), got(nw);
    auto run_mel = [&](std::vector<float> &out) -> int {
        CK(hipMemsetAsync(d_raw, 0xff, nw * 4, st));
        launch_mel_frames(d_pcm, 0, n_samples, 1, d_tab, d_raw, (long)nw, n_active, n_mel, d_gmax, st);
        CK(hipMemcpyAsync(out.data(), d_raw, nw * 4, hipMemcpyDeviceToHost, st));
        CK(hipStreamSynchronize(st));
        return 0;
    };
    if (run_mel(ref)) return 2;
    for (int k = 0; k < 3; ++k) {  // the kernel alone is deterministic
        if (run_mel(got)) return 2;
        if (memcmp(ref.data(), got.data(), nw * 4) != 0) printf("%s: NOT deterministic alone\n", label);
    }
    const char *kinds[4] = {"none", "mfma", "valu", "lds"};
    for (int kind = 0; kind < 4; ++kind) {
        int bad_runs = 0, bad_words = 0, shown = 0;
        int hist_k[51] = {0}, hist_lane16[4] = {0};
        for (int r = 0; r < reps; ++r) {
            // neighbours: ~1 ms of work on three other streams, re-issued every repetition so the mel launch lands among them
            if (kind == 1)
                for (auto &q : ns) hipLaunchKernelGGL(noise_mfma, dim3(512), dim3(256), 0, q, d_sink, 20000);
            if (kind == 2)
                for (auto &q : ns) hipLaunchKernelGGL(noise_valu, dim3(1024), dim3(256), 0, q, d_sink, 400);
            if (kind == 3)
                for (auto &q : ns) hipLaunchKernelGGL(noise_lds, dim3(512), dim3(256), 0, q, d_sink, 20000);
            if (run_mel(got)) return 2;
            int bw = 0;
            for (size_t i = 0; i < nw; ++i) {
                if (memcmp(&ref[i], &got[i], 4) != 0) {
                    ++bw;
                    const int frame = (int)(i / n_mel), bin = (int)(i % n_mel);
                    hist_k[bin]++;
                    hist_lane16[((4 * bin) & 63) >> 4]++;
                    if (shown < 6) {
                        printf("  %s/%s rep %d: frame %d (in block %d) bin %d (power bins %d..%d): %.9g != %.9g\n", label, kinds[kind], r,
                               frame, frame % 16, bin, 4 * bin, 4 * bin + 3, got[i], ref[i]);
                        ++shown;
                    }
                }
            }
            bad_runs += bw > 0;
            bad_words += bw;
        }
        for (auto &q : ns) CK(hipStreamSynchronize(q));
        printf("%s beside %-4s: %d of %d launches differ, %d words", label, kinds[kind], bad_runs, reps, bad_words);
        if (bad_words) {
            printf("; by lane quarter of the power loop (k & 63) >> 4: %d %d %d %d; bins:", hist_lane16[0], hist_lane16[1], hist_lane16[2],
                   hist_lane16[3]);
            for (int j = 0; j < 51; ++j)
                if (hist_k[j]) printf(" %d:%d", j, hist_k[j]);
        }
        printf("\n");
    }
    return 0;
}
