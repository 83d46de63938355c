// First-load latency inside a freshly launched dependent kernel (development tool, not product).
// Each workgroup (512 threads) stamps wall_clock64 (100 MHz) at entry, after a load of activation data that
// the previous kernel wrote (or did not), and after J cold 16-byte weight loads per lane, in either order.
// Build: make -C tools/microbench mb_latency ; run on the GPU box: tools/microbench/mb_latency
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <vector>

using u32x4 = __attribute__((ext_vector_type(4))) unsigned;

__global__ void producer(float *x, int cnt, float v) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < cnt) x[i] = v + 0.001f * (float)(i % 97);
}

// MODE 4: weights only, in the MFMA B-fragment pattern (lane -> row l&15, 16 bytes at k = 32*step + 8*(l>>4))
// MODE 0: x only; 1: weights only; 2: x then weights (x consumed first); 3: weights issued first, then x
template <int MODE, int J>
__global__ __launch_bounds__(512) void probe(const float *x, int xcnt, const unsigned short *W, long wg_elems, float *sink,
                                             unsigned long long *stamps) {
    const int tid = threadIdx.x;
    unsigned long long t0 = wall_clock64(), t1 = t0, t2 = t0;
    const unsigned short *wb = W + (long)blockIdx.x * wg_elems;
    float acc = 0.0f;
    u32x4 w[J];
    if constexpr (MODE == 3) {
#pragma unroll
        for (int j = 0; j < J; ++j) w[j] = __builtin_nontemporal_load((const u32x4 *)(wb + ((long)j * 512 + tid) * 8));
    }
    if constexpr (MODE == 7) {  // x and W (row-segment pattern) issued together; x waited first
        const int lane = tid & 63, wave = tid >> 6, PPR = 4 * J, K = 8 * 32 * J;
        const float4 v = *reinterpret_cast<const float4 *>(x + (4 * tid) % xcnt);
#pragma unroll
        for (int j = 0; j < J; ++j) {
            const int pc = lane + 64 * j;
            w[j] = __builtin_nontemporal_load((const u32x4 *)(wb + (long)(pc / PPR) * K + wave * 32 * J + 8 * (pc % PPR)));
        }
        acc += v.x + v.y + v.z + v.w;
        asm volatile("" : "+v"(acc));
        __syncthreads();
        t1 = wall_clock64();
    }
    if constexpr (MODE != 1 && MODE != 7) {
        const float4 v = *reinterpret_cast<const float4 *>(x + (4 * tid) % xcnt);
        acc += v.x + v.y + v.z + v.w;
        asm volatile("" : "+v"(acc));
        __syncthreads();
        t1 = wall_clock64();
    }
    if constexpr (MODE == 4) {
        const int lane = tid & 63, wave = tid >> 6, K = J * 256;
#pragma unroll
        for (int j = 0; j < J; ++j)
            w[j] = __builtin_nontemporal_load((const u32x4 *)(wb + (long)(lane & 15) * K + 32 * (wave + 8 * j) + 8 * (lane >> 4)));
    }
    if constexpr (MODE == 6) {  // x first (above), then W as 16 rows x (J*64)-byte segments per wave (row stride 8 waves x that)
        const int lane = tid & 63, wave = tid >> 6, PPR = 4 * J, K = 8 * 32 * J;
#pragma unroll
        for (int j = 0; j < J; ++j) {
            const int pc = lane + 64 * j;
            w[j] = __builtin_nontemporal_load((const u32x4 *)(wb + (long)(pc / PPR) * K + wave * 32 * J + 8 * (pc % PPR)));
        }
    }
    if constexpr (MODE == 1 || MODE == 2) {
#pragma unroll
        for (int j = 0; j < J; ++j) w[j] = __builtin_nontemporal_load((const u32x4 *)(wb + ((long)j * 512 + tid) * 8));
    }
    if constexpr (MODE != 0) {
        unsigned s = 0;
#pragma unroll
        for (int j = 0; j < J; ++j) s += w[j].x ^ w[j].y ^ w[j].z ^ w[j].w;
        acc += (float)s;
        asm volatile("" : "+v"(acc));
        __syncthreads();
        t2 = wall_clock64();
    }
    if (acc == 1.2345f) sink[0] = acc;
    if (tid == 0) {
        stamps[blockIdx.x * 4 + 0] = t0;
        stamps[blockIdx.x * 4 + 1] = t1;
        stamps[blockIdx.x * 4 + 2] = t2;
    }
}

template <int MODE, int J> void run(const char *name, int wgs, bool rewrite, hipStream_t st, float *x, int xcnt, unsigned short *W,
                                    size_t wtotal, float *sink, unsigned long long *stamps) {
    const long wg_elems = (long)J * 512 * 8;
    const size_t per_launch = (size_t)wgs * wg_elems;
    const int nbuf = (int)(wtotal / per_launch);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int iters = 100;
    for (int it = 0; it < iters + 5; ++it) {
        if (it == 5) hipEventRecord(e0, st);
        if (rewrite) hipLaunchKernelGGL(producer, dim3((xcnt + 255) / 256), dim3(256), 0, st, x, xcnt, (float)it);
        hipLaunchKernelGGL((probe<MODE, J>), dim3(wgs), dim3(512), 0, st, x, xcnt, W + (size_t)(it % nbuf) * per_launch, wg_elems, sink, stamps);
    }
    hipEventRecord(e1, st);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h((size_t)wgs * 4);
    hipMemcpy(h.data(), stamps, h.size() * 8, hipMemcpyDeviceToHost);
    unsigned long long first = ~0ull, last = 0;
    std::vector<double> a(wgs), b(wgs);
    for (int i = 0; i < wgs; ++i) {
        first = std::min(first, h[i * 4]);
        last = std::max(last, std::max(h[i * 4 + 1], h[i * 4 + 2]));
        a[i] = (double)(long long)(h[i * 4 + 1] - h[i * 4]) / 100.0;
        b[i] = (double)(long long)(h[i * 4 + 2] - std::max(h[i * 4 + 1], h[i * 4])) / 100.0;
    }
    std::sort(a.begin(), a.end());
    std::sort(b.begin(), b.end());
    printf("%-52s wgs %3d  %6.2f us/iter | x load: p50 %5.2f max %5.2f | weights after: p50 %5.2f max %5.2f | first entry -> last done %5.2f\n",
           name, wgs, ms * 1000.0 / iters, a[wgs / 2], a[wgs - 1], b[wgs / 2], b[wgs - 1], (double)(long long)(last - first) / 100.0);
}

int main() {
    hipStream_t st;
    hipStreamCreate(&st);
    const int xcnt = 5 * 1280;
    float *x, *sink;
    unsigned short *W;
    unsigned long long *stamps;
    const size_t wtotal = (size_t)256 << 20;  // elements (512 MB)
    hipMalloc(&x, xcnt * 4);
    hipMalloc(&sink, 64);
    hipMalloc(&W, wtotal * 2);
    hipMemset(W, 0x11, wtotal * 2);
    hipMalloc(&stamps, 1024 * 4 * 8);
    hipLaunchKernelGGL(producer, dim3((xcnt + 255) / 256), dim3(256), 0, st, x, xcnt, 1.0f);
    for (int wgs : {80, 320}) {
        run<0, 5>("x only, x rewritten by the previous kernel", wgs, true, st, x, xcnt, W, wtotal, sink, stamps);
        run<0, 5>("x only, x untouched since long ago", wgs, false, st, x, xcnt, W, wtotal, sink, stamps);
        run<1, 5>("weights only (5 x 16 B per lane, cold)", wgs, true, st, x, xcnt, W, wtotal, sink, stamps);
        run<1, 20>("weights only (20 x 16 B per lane, cold)", wgs, true, st, x, xcnt, W, wtotal, sink, stamps);
        run<4, 5>("weights only, B-fragment pattern (5 per lane, cold)", wgs, true, st, x, xcnt, W, wtotal, sink, stamps);
        run<4, 5>("weights only, B-fragment pattern, same 105 MB", wgs, true, st, x, xcnt, W, (size_t)wgs * 5 * 512 * 8 * (wgs == 80 ? 16 : 4), sink, stamps);
        run<7, 5>("x and W (row segments) issued together", wgs, true, st, x, xcnt, W, wtotal, sink, stamps);
        run<7, 20>("x and 20 W (row segments) issued together", wgs, true, st, x, xcnt, W, wtotal, sink, stamps);
        run<6, 5>("x, then W as 16 rows x 320 B segments per wave", wgs, true, st, x, xcnt, W, wtotal, sink, stamps);
        run<2, 5>("x consumed, then 5 weight loads", wgs, true, st, x, xcnt, W, wtotal, sink, stamps);
        run<3, 5>("5 weight loads issued, then x", wgs, true, st, x, xcnt, W, wtotal, sink, stamps);
        run<3, 20>("20 weight loads issued, then x", wgs, true, st, x, xcnt, W, wtotal, sink, stamps);
    }
    return 0;
}
