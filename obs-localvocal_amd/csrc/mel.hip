// mel.hip — K1: log-mel front end for gfx950.
//
// Replaces whisper.cpp's log_mel_spectrogram (reached from whisper_full,
// /root/reference/src/whisper-utils/whisper-processing.cpp:218-219): reflect-pad 200 at
// the start, zeros at the end, periodic Hann-400, 400-point real DFT -> 201 power bins, mel filterbank
// (taps accumulated in double, groups of 4 in float like the reference), log10(max(.,1e-10)),
// global max-8 clamp, (x+4)/4.
//
// Bound: HBM (3.46 MB algorithmic per 30 s window at 128 bins).  Structure: one 256-thread workgroup
// stages the PCM span of 16 consecutive frames in LDS with coalesced 4-byte loads (each sample is read
// from HBM once per workgroup; the 240-sample overlap between neighbouring workgroups comes from L2),
// then each wave turns 4 frames into mel rows: the real 400-point transform is done as a 200-point
// complex Stockham FFT (radix 5,5,8 butterflies through LDS ping-pong buffers) plus the real-input
// split, so no transcendental is evaluated on the device (twiddles/Hann come from a 4.8 KB table).
// Output rows are frame-major ([frame][bin]) so the conv stem reads them as a GEMM operand.
#include "kernels.hpp"

#include <hip/hip_bf16.h>

#include <mutex>

namespace mw {

namespace {

// Scratch of the allocation-free log-mel entry point (mel_noalloc below; mw_pcm_to_mel_noalloc in
// include/mi355_whisper.h): the global-maximum word (first 256-byte line) followed by the raw log10 mel rows of the
// frames that overlap real samples.  One copy per GPU the library runs on, held until the process ends; every call on
// a GPU uses that GPU's copy, so calls that overlap on one GPU race.  Host code never uses the array's name as an
// address: mel_noalloc_init() asks the runtime for the device address once per device.
constexpr int MEL_NOALLOC_MAX_ACTIVE = (MEL_NOALLOC_MAX_SAMPLES + MEL_N_FFT / 2) / MEL_HOP + 1;
constexpr size_t MEL_NOALLOC_RAW_OFF = 64;  // floats: the raw rows start one 256-byte line behind the maximum word
constexpr size_t MEL_NOALLOC_FLOATS = MEL_NOALLOC_RAW_OFF + (size_t)MEL_NOALLOC_MAX_ACTIVE * MEL_MAX_BINS;
static __device__ __align__(256) float scratch[MEL_NOALLOC_FLOATS];  // 1.54 MB
constexpr int MEL_NOALLOC_MAX_DEVICES = 64;
float *g_scratch_ptr[MEL_NOALLOC_MAX_DEVICES] = {nullptr};
std::mutex g_scratch_mutex;

constexpr int FRAMES_PER_BLOCK = 16;
constexpr int SPAN = (FRAMES_PER_BLOCK - 1) * MEL_HOP + MEL_N_FFT;  // 2800 samples
constexpr int NH = MEL_N_FFT / 2;                                   // 200-point complex transform

struct cpx {
    float x, y;
};
// An empty asm statement the optimiser cannot look through: the re and im parts of a complex value that passed through it
// are no longer "adjacent scalar operations" for hipcc's SLP vectoriser, so no packed-f32 instruction (v_pk_*_f32) is
// formed from them whatever flags the file is built with (see the note on mel_frames_kernel below).
#ifndef MEL_OPQ_FFT
#define MEL_OPQ_FFT 1
#endif
#ifndef MEL_OPQ_PW
#define MEL_OPQ_PW 1
#endif
__device__ __forceinline__ float opaque(float v) {
    asm volatile("" : "+v"(v));
    return v;
}
__device__ __forceinline__ cpx fft_opq(cpx a) {
#if MEL_OPQ_FFT
    a.y = opaque(a.y);
#endif
    return a;
}
__device__ __forceinline__ cpx pw_opq(cpx a) {
#if MEL_OPQ_PW
    a.x = opaque(a.x);
    a.y = opaque(a.y);
#endif
    return a;
}
__device__ __forceinline__ cpx cadd(cpx a, cpx b) { return fft_opq({a.x + b.x, a.y + b.y}); }
__device__ __forceinline__ cpx csub(cpx a, cpx b) { return fft_opq({a.x - b.x, a.y - b.y}); }
__device__ __forceinline__ cpx cmul(cpx a, cpx b) {
#if MEL_OPQ_FFT || MEL_OPQ_PW
    const float p2 = opaque(a.y * b.y), p4 = opaque(a.y * b.x);  // (the other two products contract into the fma)
    return fft_opq({a.x * b.x - p2, a.x * b.y + p4});
#else
    return {a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x};
#endif
}
__device__ __forceinline__ cpx cscale(cpx a, float s) { return {a.x * s, a.y * s}; }
__device__ __forceinline__ cpx mul_neg_i(cpx a) { return {a.y, -a.x}; }  // a * (-i)
__device__ __forceinline__ cpx mul_pos_i(cpx a) { return {-a.y, a.x}; }  // a * (+i)

// forward 5-point DFT (w = exp(-2*pi*i/5))
__device__ __forceinline__ void dft5(cpx *v) {
    const float c1 = 0.30901699437494742f, c2 = -0.80901699437494742f;
    const float s1 = 0.95105651629515357f, s2 = 0.58778525229247313f;
    const cpx t1 = cadd(v[1], v[4]), t2 = cadd(v[2], v[3]), t3 = csub(v[1], v[4]), t4 = csub(v[2], v[3]);
    const cpx m1 = {v[0].x + c1 * t1.x + c2 * t2.x, v[0].y + c1 * t1.y + c2 * t2.y};
    const cpx m2 = {v[0].x + c2 * t1.x + c1 * t2.x, v[0].y + c2 * t1.y + c1 * t2.y};
    const cpx n1 = {s1 * t3.x + s2 * t4.x, s1 * t3.y + s2 * t4.y};
    const cpx n2 = {s2 * t3.x - s1 * t4.x, s2 * t3.y - s1 * t4.y};
    v[0] = cadd(v[0], cadd(t1, t2));
    v[1] = cadd(m1, mul_neg_i(n1));
    v[4] = cadd(m1, mul_pos_i(n1));
    v[2] = cadd(m2, mul_neg_i(n2));
    v[3] = cadd(m2, mul_pos_i(n2));
}

__device__ __forceinline__ void dft4(cpx b0, cpx b1, cpx b2, cpx b3, cpx &y0, cpx &y1, cpx &y2, cpx &y3) {
    const cpx c0 = cadd(b0, b2), c2 = csub(b0, b2), c1 = cadd(b1, b3), c3 = mul_neg_i(csub(b1, b3));
    y0 = cadd(c0, c1);
    y2 = csub(c0, c1);
    y1 = cadd(c2, c3);
    y3 = csub(c2, c3);
}

// forward 8-point DFT
__device__ __forceinline__ void dft8(cpx *v) {
    const float r = 0.70710678118654752f;
    const cpx a0 = cadd(v[0], v[4]), a4 = csub(v[0], v[4]);
    const cpx a1 = cadd(v[1], v[5]), a5 = csub(v[1], v[5]);
    const cpx a2 = cadd(v[2], v[6]), a6 = csub(v[2], v[6]);
    const cpx a3 = cadd(v[3], v[7]), a7 = csub(v[3], v[7]);
    const cpx a5w = cmul(a5, cpx{r, -r});
    const cpx a6w = mul_neg_i(a6);
    const cpx a7w = cmul(a7, cpx{-r, -r});
    cpx e0, e1, e2, e3, o0, o1, o2, o3;
    dft4(a0, a1, a2, a3, e0, e1, e2, e3);
    dft4(a4, a5w, a6w, a7w, o0, o1, o2, o3);
    v[0] = e0; v[2] = e1; v[4] = e2; v[6] = e3;
    v[1] = o0; v[3] = o1; v[5] = o2; v[7] = o3;
}

__device__ __forceinline__ unsigned f32_to_ordered(float f) {
    unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ordered_to_f32(unsigned u) {
    return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u);
}

struct __attribute__((aligned(16))) MelLds {
    float pcm[SPAN];
    float tw_cos[MEL_N_FFT];
    float tw_sin[MEL_N_FFT];
    float hann[MEL_N_FFT];
    cpx buf[4][2][NH];   // per wave: ping-pong FFT buffers
    float pw[4][MEL_N_FREQ + 3];
    float vals[MEL_MAX_VALS];
    float wmax[4];
};

__global__ __launch_bounds__(256) void mel_frames_kernel(const float *__restrict__ pcm, long pcm_stride, int n_samples,
                                                         const MelTables *__restrict__ tab, float *__restrict__ raw,
                                                         long raw_stride, int n_active, int n_mel,
                                                         unsigned *__restrict__ gmax) {
    __shared__ MelLds L;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int stream = blockIdx.y;
    const int f0 = blockIdx.x * FRAMES_PER_BLOCK;
    pcm += (long)stream * pcm_stride;
    raw += (long)stream * raw_stride;

    // stage tables and the PCM span [f0*160 - 200, ...) of the padded signal
    for (int i = tid; i < MEL_N_FFT; i += 256) {
        L.tw_cos[i] = tab->tw_cos[i];
        L.tw_sin[i] = tab->tw_sin[i];
        L.hann[i] = tab->hann[i];
    }
    for (int i = tid; i < MEL_MAX_VALS; i += 256) L.vals[i] = tab->vals[i];
    const long base = (long)f0 * MEL_HOP - MEL_N_FFT / 2;
    for (int i = tid; i < SPAN; i += 256) {
        long s = base + i;
        if (s < 0) s = -s;  // reflect pad at the start only
        L.pcm[i] = (s < n_samples) ? pcm[s] : 0.0f;
    }
    __syncthreads();

    float my_max = -1e30f;
    for (int fi = 0; fi < FRAMES_PER_BLOCK / 4; ++fi) {
        const int fl = wave * (FRAMES_PER_BLOCK / 4) + fi;  // frame within block
        const int frame = f0 + fl;
        cpx *A = L.buf[wave][0], *B = L.buf[wave][1];
        // windowed samples packed as z[n] = x[2n] + i x[2n+1]
        for (int n = lane; n < NH; n += 64) {
            const int o = fl * MEL_HOP + 2 * n;
            A[n] = fft_opq(cpx{L.pcm[o] * L.hann[2 * n], L.pcm[o + 1] * L.hann[2 * n + 1]});
        }
        __syncthreads();
        // stage 1: radix 5, Ns = 1 (no twiddles)
        if (lane < 40) {
            cpx v[5];
#pragma unroll
            for (int t = 0; t < 5; ++t) v[t] = A[lane + 40 * t];
            dft5(v);
#pragma unroll
            for (int t = 0; t < 5; ++t) B[lane * 5 + t] = v[t];
        }
        __syncthreads();
        // stage 2: radix 5, Ns = 5, twiddle exp(-2 pi i t k / 25) = table[16 t k]
        if (lane < 40) {
            const int k = lane % 5;
            cpx v[5];
#pragma unroll
            for (int t = 0; t < 5; ++t) {
                const int m = (16 * t * k) % MEL_N_FFT;
                v[t] = cmul(B[lane + 40 * t], cpx{L.tw_cos[m], -L.tw_sin[m]});
            }
            dft5(v);
            const int j0 = (lane / 5) * 25 + k;
#pragma unroll
            for (int t = 0; t < 5; ++t) A[j0 + 5 * t] = v[t];
        }
        __syncthreads();
        // stage 3: radix 8, Ns = 25, twiddle exp(-2 pi i t k / 200) = table[2 t k]
        if (lane < 25) {
            const int k = lane;
            cpx v[8];
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const int m = (2 * t * k) % MEL_N_FFT;
                v[t] = cmul(A[lane + 25 * t], cpx{L.tw_cos[m], -L.tw_sin[m]});
            }
            dft8(v);
#pragma unroll
            for (int t = 0; t < 8; ++t) B[k + 25 * t] = v[t];
        }
        __syncthreads();
        // real-input split: X[k] = E[k] + exp(-2 pi i k / 400) O[k]; power spectrum
        for (int k = lane; k < MEL_N_FREQ + 3; k += 64) {
            float p = 0.0f;
            if (k < MEL_N_FREQ) {
                cpx zk = B[k % NH];
                cpx zn = B[(NH - k) % NH];
#if MEL_OPQ_PW
                zk.y = opaque(zk.y);
                zn.y = opaque(zn.y);
#endif
                const cpx e = pw_opq({0.5f * (zk.x + zn.x), 0.5f * (zk.y - zn.y)});
                const cpx d = pw_opq({zk.x - zn.x, zk.y + zn.y});   // Zk - conj(Zn)
                const cpx o = pw_opq({0.5f * d.y, -0.5f * d.x});    // d / (2i)
                const cpx w = {L.tw_cos[k % MEL_N_FFT], -L.tw_sin[k % MEL_N_FFT]};
                const cpx x = pw_opq(cadd(e, pw_opq(cmul(w, o))));
#if MEL_OPQ_PW
                p = opaque(x.x * x.x) + x.y * x.y;
#else
                p = x.x * x.x + x.y * x.y;
#endif
            }
            L.pw[wave][k] = p;
        }
        __syncthreads();
        if (frame < n_active) {
            for (int j = lane; j < n_mel; j += 64) {
                double sum = 0.0;
                const float *pv = &L.pw[wave][tab->k0[j]];
                const float *fv = &L.vals[tab->voff[j]];
                const int ng = tab->ngrp[j];
                for (int g = 0; g < ng; ++g) {
#if MEL_OPQ_PW
                    const float part = opaque(pv[4 * g] * fv[4 * g]) + opaque(pv[4 * g + 1] * fv[4 * g + 1]) +
                                       opaque(pv[4 * g + 2] * fv[4 * g + 2]) + opaque(pv[4 * g + 3] * fv[4 * g + 3]);
#else
                    const float part = pv[4 * g] * fv[4 * g] + pv[4 * g + 1] * fv[4 * g + 1] +
                                       pv[4 * g + 2] * fv[4 * g + 2] + pv[4 * g + 3] * fv[4 * g + 3];
#endif
                    sum += (double)part;
                }
                sum += (double)(L.pw[wave][MEL_N_FREQ - 1] * tab->last[j]);
                const float v = (float)log10(sum > 1e-10 ? sum : 1e-10);
                raw[(long)frame * n_mel + j] = v;
                my_max = fmaxf(my_max, v);
            }
        }
        __syncthreads();
    }
    // workgroup max -> one ordered-int atomic per workgroup
    for (int o = 32; o > 0; o >>= 1) my_max = fmaxf(my_max, __shfl_xor(my_max, o));
    if (lane == 0) L.wmax[wave] = my_max;
    __syncthreads();
    if (tid == 0) {
        const float m = fmaxf(fmaxf(L.wmax[0], L.wmax[1]), fmaxf(L.wmax[2], L.wmax[3]));
        if (m > -1e29f) atomicMax(&gmax[stream], f32_to_ordered(m));
    }
}

__global__ __launch_bounds__(256) void mel_normalize_kernel(const float *__restrict__ raw, long raw_stride,
                                                            float *__restrict__ out, long out_stride, int n_len,
                                                            int n_active, int n_mel, const unsigned *__restrict__ gmax) {
    const int stream = blockIdx.y;
    raw += (long)stream * raw_stride;
    out += (long)stream * out_stride;
    // the padding frames are log10(1e-10) = -10 everywhere, so they never raise the maximum
    const double mmax = (double)fmaxf(ordered_to_f32(gmax[stream]), n_active < n_len ? -10.0f : -1e30f) - 8.0;
    const long total = (long)n_len * n_mel;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long frame = i / n_mel;
        float v = frame < n_active ? raw[i] : -10.0f;
        if ((double)v < mmax) v = (float)mmax;
        out[i] = (float)(((double)v + 4.0) / 4.0);
    }
}

template <typename T> __device__ __forceinline__ T cvt_out(float v);
template <> __device__ __forceinline__ float cvt_out<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16_t cvt_out<bf16_t>(float v) {
    __hip_bfloat16 h = __float2bfloat16(v);
    return *reinterpret_cast<bf16_t *>(&h);
}

template <typename T>
__global__ __launch_bounds__(256) void mel_window_kernel(const float *__restrict__ mel, long mel_stride, int n_len,
                                                         int seek, int n_rows, int n_mel, T *__restrict__ dst,
                                                         long dst_stride) {
    const int stream = blockIdx.y;
    mel += (long)stream * mel_stride;
    dst += (long)stream * dst_stride;
    const long total = (long)(n_rows + 2) * n_mel;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long r = i / n_mel - 1;
        const int c = (int)(i % n_mel);
        float v = 0.0f;
        if (r >= 0 && r < n_rows && seek + r < n_len) v = mel[(seek + r) * n_mel + c];
        dst[i] = cvt_out<T>(v);
    }
}

__global__ __launch_bounds__(256) void transpose_f32_kernel(const float *__restrict__ in, float *__restrict__ out,
                                                            int rows, int cols) {
    __shared__ float tile[32][33];
    const int bx = blockIdx.x * 32, by = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int j = ty; j < 32; j += 8)
        if (by + j < rows && bx + tx < cols) tile[j][tx] = in[(long)(by + j) * cols + bx + tx];
    __syncthreads();
    for (int j = ty; j < 32; j += 8)
        if (bx + j < cols && by + tx < rows) out[(long)(bx + j) * rows + by + tx] = tile[tx][j];
}

}  // namespace

void launch_mel_frames(const float *d_pcm, long pcm_stride, int n_samples, int n_streams, const MelTables *d_tab,
                       float *d_raw, long raw_stride, int n_active, int n_mel, unsigned *d_gmax, hipStream_t st) {
    MW_CHECK(n_samples > 0 && n_streams > 0 && n_active > 0, "mel: empty input");
    MW_CHECK(n_mel > 0 && n_mel <= MEL_MAX_BINS, "mel: bin count out of range");
    MW_CHECK(raw_stride >= (long)n_active * n_mel, "mel: raw buffer too small");
    MW_HIP(hipMemsetAsync(d_gmax, 0, sizeof(unsigned) * n_streams, st));
    dim3 grid(ceil_div(n_active, FRAMES_PER_BLOCK), n_streams);
    hipLaunchKernelGGL(mel_frames_kernel, grid, dim3(256), 0, st, d_pcm, pcm_stride, n_samples, d_tab, d_raw, raw_stride,
                       n_active, n_mel, d_gmax);
    MW_HIP(hipGetLastError());
}

void launch_mel_normalize(const float *d_raw, long raw_stride, float *d_out, long out_stride, int n_len, int n_active,
                          int n_mel, int n_streams, const unsigned *d_gmax, hipStream_t st) {
    MW_CHECK(out_stride >= (long)n_len * n_mel, "mel: out buffer too small");
    const long total = (long)n_len * n_mel;
    dim3 grid((unsigned)std::min<long>(ceil_div<long>(total, 256), 1024), n_streams);
    hipLaunchKernelGGL(mel_normalize_kernel, grid, dim3(256), 0, st, d_raw, raw_stride, d_out, out_stride, n_len,
                       n_active, n_mel, d_gmax);
    MW_HIP(hipGetLastError());
}

void launch_mel_window(const float *d_mel, long mel_stride, int n_len, int seek, int n_rows, int n_mel, int n_streams,
                       void *d_dst, long dst_stride, DType dt, hipStream_t st) {
    MW_CHECK(dst_stride >= (long)(n_rows + 2) * n_mel, "mel window: dst too small");
    MW_CHECK(seek >= 0, "mel window: negative seek");
    const long total = (long)(n_rows + 2) * n_mel;
    dim3 grid((unsigned)std::min<long>(ceil_div<long>(total, 256), 1024), n_streams);
    if (dt == DType::F32)
        hipLaunchKernelGGL(mel_window_kernel<float>, grid, dim3(256), 0, st, d_mel, mel_stride, n_len, seek, n_rows, n_mel,
                           (float *)d_dst, dst_stride);
    else
        hipLaunchKernelGGL(mel_window_kernel<bf16_t>, grid, dim3(256), 0, st, d_mel, mel_stride, n_len, seek, n_rows,
                           n_mel, (bf16_t *)d_dst, dst_stride);
    MW_HIP(hipGetLastError());
}

// The device address of `scratch` on `device` (which must be the current device): looked up once per device.
float *mel_noalloc_init(int device) {
    MW_CHECK(device >= 0 && device < MEL_NOALLOC_MAX_DEVICES, "mel_noalloc: device index out of range");
    std::lock_guard<std::mutex> lk(g_scratch_mutex);
    if (!g_scratch_ptr[device]) {
        int cur = -1;
        MW_HIP(hipGetDevice(&cur));
        MW_CHECK(cur == device, "mel_noalloc: the context's device is not current");
        float *ptr = nullptr;
        MW_HIP(hipGetSymbolAddress((void **)&ptr, HIP_SYMBOL(scratch)));
        MW_CHECK(ptr != nullptr, "mel_noalloc: no device address for the scratch array");
        g_scratch_ptr[device] = ptr;
    }
    return g_scratch_ptr[device];
}

// log-mel of device-resident PCM into the caller's buffer d_out[n_len][n_mel]; the same two kernels as pcm_to_mel
// (engine.hip) with the intermediate rows and the maximum word in the static scratch array.  Allocates nothing.
void mel_noalloc(const float *d_pcm, int n_samples, const MelTables *d_tab, int n_mel, float *d_out, size_t out_floats,
                 int *n_len_out, int device, hipStream_t st) {
    MW_CHECK(d_pcm && d_out && n_samples > 0, "mel_noalloc: empty input");
    MW_CHECK(n_samples <= MEL_NOALLOC_MAX_SAMPLES,
             "mel_noalloc: more than 30 s + 640 samples of PCM (480640); the static scratch array holds one window");
    MW_CHECK(n_mel > 0 && n_mel <= MEL_MAX_BINS, "mel_noalloc: bin count out of range");
    const int n_len = (n_samples + 30 * 16000) / MEL_HOP;
    const int n_active = std::min((n_samples + MEL_N_FFT / 2) / MEL_HOP + 1, n_len);
    MW_CHECK(n_active <= MEL_NOALLOC_MAX_ACTIVE, "mel_noalloc: frame count exceeds the scratch array");
    MW_CHECK(out_floats >= (size_t)n_len * n_mel, "mel_noalloc: output buffer smaller than n_len * n_mel floats");
    float *ptr = mel_noalloc_init(device);
    unsigned *gmax = reinterpret_cast<unsigned *>(ptr);
    float *raw = ptr + MEL_NOALLOC_RAW_OFF;
    // launch_mel_frames clears the maximum word (hipMemsetAsync on ptr); the raw rows of frames < n_active are all
    // rewritten by this call and rows beyond are never read, so the previous call's contents cannot leak
    launch_mel_frames(d_pcm, 0, n_samples, 1, d_tab, raw, (long)MEL_NOALLOC_MAX_ACTIVE * MEL_MAX_BINS, n_active, n_mel, gmax, st);
    launch_mel_normalize(raw, (long)MEL_NOALLOC_MAX_ACTIVE * MEL_MAX_BINS, d_out, (long)n_len * n_mel, n_len, n_active, n_mel, 1,
                         gmax, st);
    if (n_len_out) *n_len_out = n_len;
}

void launch_transpose_f32(const float *d_in, float *d_out, int rows, int cols, hipStream_t st) {
    dim3 grid(ceil_div(cols, 32), ceil_div(rows, 32));
    hipLaunchKernelGGL(transpose_f32_kernel, grid, dim3(256), 0, st, d_in, d_out, rows, cols);
    MW_HIP(hipGetLastError());
}

}  // namespace mw
