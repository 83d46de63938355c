// Micro-benchmark: cost of a grid-wide barrier inside one persistent kernel on gfx950 (256 CUs, 8 XCDs).
// Decides whether the beam decode step can run as ONE persistent kernel (phases separated by barriers)
// instead of ~290 graph-replayed launches.  Variants:
//   0  relaxed agent-scope atomics only (no cache maintenance; data would travel via sc1 loads/stores)
//   1  release/acquire fences at agent scope around the counter (what __threadfence() costs)
//   2  variant 0 + each block passes a value to its neighbour through agent-scope relaxed stores/loads
//      (checks cross-XCD visibility without fences; prints mismatches)
// Every spin loop is bounded: a wave that waits too long raises `abort_flag` and everyone drains.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

struct Sync {
    unsigned counter;
    unsigned abort_flag;
};

template <int VARIANT>
__device__ __forceinline__ bool grid_barrier(Sync *s, unsigned &target) {
    __syncthreads();
    bool ok = true;
    if (threadIdx.x == 0) {
        if (VARIANT == 1) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __hip_atomic_fetch_add(&s->counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned spins = 0;
        while (__hip_atomic_load(&s->counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > (1u << 22) || __hip_atomic_load(&s->abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                __hip_atomic_store(&s->abort_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                ok = false;
                break;
            }
        }
        if (VARIANT == 1) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    target += gridDim.x;
    __syncthreads();
    return ok;
}

template <int VARIANT>
__global__ void __launch_bounds__(512) barrier_loop(Sync *s, int iters, float *buf, unsigned *mism) {
    unsigned target = gridDim.x;
    const int b = blockIdx.x, nb = gridDim.x;
    unsigned bad = 0;
    for (int i = 0; i < iters; ++i) {
        if (VARIANT == 2) {
            // each thread < 64 publishes a value; neighbour block (different XCD: +1) checks it after the barrier
            if (threadIdx.x < 64)
                __hip_atomic_store(&buf[(i & 1) * nb * 64 + b * 64 + threadIdx.x], (float)(i * 1000 + b), __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
        }
        if (!grid_barrier<VARIANT == 2 ? 0 : VARIANT>(s, target)) return;
        if (VARIANT == 2) {
            if (threadIdx.x < 64) {
                const int src = (b + 1) % nb;
                float v = __hip_atomic_load(&buf[(i & 1) * nb * 64 + src * 64 + threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (v != (float)(i * 1000 + src)) ++bad;
            }
        }
    }
    if (VARIANT == 2 && bad) atomicAdd(mism, bad);
}

template <int VARIANT>
void run(const char *name, int blocks, int threads, int iters) {
    Sync *s;
    float *buf;
    unsigned *mism;
    CK(hipMalloc(&s, sizeof(Sync)));
    CK(hipMalloc(&buf, 2 * blocks * 64 * sizeof(float)));
    CK(hipMalloc(&mism, 4));
    CK(hipMemset(s, 0, sizeof(Sync)));
    CK(hipMemset(mism, 0, 4));
    CK(hipMemset(buf, 0, 2 * blocks * 64 * sizeof(float)));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    void *args[] = {&s, &iters, &buf, &mism};
    for (int rep = 0; rep < 2; ++rep) {
        CK(hipMemset(s, 0, sizeof(Sync)));
        CK(hipEventRecord(e0, 0));
        CK(hipLaunchCooperativeKernel((const void *)barrier_loop<VARIANT>, dim3(blocks), dim3(threads), args, 0, 0));
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
    }
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    Sync hs;
    unsigned hm;
    CK(hipMemcpy(&hs, s, sizeof(Sync), hipMemcpyDeviceToHost));
    CK(hipMemcpy(&hm, mism, 4, hipMemcpyDeviceToHost));
    printf("%-34s blocks=%4d threads=%3d : %7.3f us/barrier  abort=%u mismatches=%u\n", name, blocks, threads,
           1000.0f * ms / iters, hs.abort_flag, hm);
    CK(hipFree(s));
    CK(hipFree(buf));
    CK(hipFree(mism));
}

int main() {
    const int iters = 2000;
    for (int threads : {256, 512}) {
        for (int blocks : {64, 256, 512}) {
            if (blocks * threads > 256 * 2048) continue;
            run<0>("relaxed atomics", blocks, threads, iters);
            run<1>("release/acquire fences", blocks, threads, iters);
            run<2>("relaxed + neighbour handoff check", blocks, threads, iters);
        }
    }
    return 0;
}
