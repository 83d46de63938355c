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

// variant 3: arrivals spread over 16 counters in one 64-byte line (block b -> counter b & 15); lanes 0..15 of
// wave 0 poll one counter each.  16 atomics per address instead of 256.
__device__ __forceinline__ bool grid_barrier16(unsigned *cnt, unsigned *abort_flag, unsigned epoch) {
    __syncthreads();
    bool ok = true;
    if (threadIdx.x < 64) {
        const unsigned lane = threadIdx.x;
        if (lane == 0) __hip_atomic_fetch_add(&cnt[blockIdx.x & 15], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned members = (gridDim.x + 15 - (lane & 15)) / 16;  // blocks with (b & 15) == lane
        const unsigned target = members * epoch;
        const unsigned long long t0 = wall_clock64();
        while (true) {
            const unsigned v = lane < 16 ? __hip_atomic_load(&cnt[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : target;
            if (__all((int)(v - target) >= 0)) break;
            __builtin_amdgcn_s_sleep(1);
            if (wall_clock64() - t0 > 2000000ull || __hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                if (lane == 0) __hip_atomic_store(abort_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                ok = false;
                break;
            }
        }
    }
    __syncthreads();
    return ok;
}

// variant 4: one arrival counter; the last arriver (fetch_add returns total-1) writes a per-block release flag
// (each on its own 128-byte line), every block polls only its own flag.
__device__ __forceinline__ bool grid_barrier_flags(unsigned *cnt, unsigned *flags, unsigned *abort_flag, unsigned epoch) {
    __syncthreads();
    bool ok = true;
    if (threadIdx.x < 64) {
        const unsigned lane = threadIdx.x, nb = gridDim.x;
        unsigned old = 0;
        if (lane == 0) old = __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        old = __builtin_amdgcn_readfirstlane(old);
        if (old == nb * epoch - 1) {  // last one in: release everybody
            for (unsigned b = lane; b < nb; b += 64)
                __hip_atomic_store(&flags[b * 32], epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            const unsigned long long t0 = wall_clock64();
            while (true) {
                const unsigned v = __hip_atomic_load(&flags[blockIdx.x * 32], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if ((int)(v - epoch) >= 0) break;
                __builtin_amdgcn_s_sleep(1);
                if (wall_clock64() - t0 > 2000000ull || __hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                    if (lane == 0) __hip_atomic_store(abort_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    ok = false;
                    break;
                }
            }
        }
    }
    __syncthreads();
    return ok;
}

// variant 5: two-level arrival (16 group counters on separate 256-byte-spaced lines, then a root counter); the
// last arriver at the root writes the per-block release flags.
__device__ __forceinline__ bool grid_barrier_tree(unsigned *grp, unsigned *root, unsigned *flags, unsigned *abort_flag,
                                                  unsigned epoch) {
    __syncthreads();
    bool ok = true;
    if (threadIdx.x < 64) {
        const unsigned lane = threadIdx.x, nb = gridDim.x, g = blockIdx.x & 15;
        const unsigned members = (nb + 15 - g) / 16, ngroups = nb < 16 ? nb : 16;
        unsigned last = 0;
        if (lane == 0) {
            unsigned old = __hip_atomic_fetch_add(&grp[g * 64], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (old == members * epoch - 1) {
                old = __hip_atomic_fetch_add(root, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                last = old == ngroups * epoch - 1;
            }
        }
        last = __builtin_amdgcn_readfirstlane(last);
        if (last) {
            for (unsigned b = lane; b < nb; b += 64)
                __hip_atomic_store(&flags[b * 32], epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            const unsigned long long t0 = wall_clock64();
            while (true) {
                const unsigned v = __hip_atomic_load(&flags[blockIdx.x * 32], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if ((int)(v - epoch) >= 0) break;
                __builtin_amdgcn_s_sleep(1);
                if (wall_clock64() - t0 > 2000000ull || __hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                    if (lane == 0) __hip_atomic_store(abort_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    ok = false;
                    break;
                }
            }
        }
    }
    __syncthreads();
    return ok;
}

template <int V>
__global__ void __launch_bounds__(512) barrier16_loop(unsigned *cnt, unsigned *abort_flag, int iters, float *buf, unsigned *mism) {
    const int b = blockIdx.x, nb = gridDim.x;
    unsigned bad = 0;
    for (int i = 0; i < iters; ++i) {
        if (threadIdx.x < 64)
            __hip_atomic_store(&buf[(i & 1) * nb * 64 + b * 64 + threadIdx.x], (float)(i * 1000 + b), __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
        if (V == 3 && !grid_barrier16(cnt, abort_flag, (unsigned)(i + 1))) return;
        if (V == 4 && !grid_barrier_flags(cnt + 64, cnt + 2048, abort_flag, (unsigned)(i + 1))) return;
        if (V == 5 && !grid_barrier_tree(cnt + 128, cnt + 64, cnt + 2048, abort_flag, (unsigned)(i + 1))) return;
        if (V == 6) {  // no barrier at all: cost of the handoff traffic + loop alone
            __syncthreads();
        }
        if (threadIdx.x < 64) {
            const int src = (b + 37) % nb;
            float v = __hip_atomic_load(&buf[(i & 1) * nb * 64 + src * 64 + threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (v != (float)(i * 1000 + src)) ++bad;
        }
    }
    if (bad) atomicAdd(mism, bad);
}

// variant 7: what the decode kernel relies on — each phase writes a buffer of its own with agent-scope
// (write-through) stores, readers in other workgroups use PLAIN cached loads after the barrier; the same
// buffers are rewritten by every launch with new values (stale L2 lines from the previous launch would show).
__global__ void __launch_bounds__(512) unique_buf_loop(unsigned *cnt, unsigned *abort_flag, int iters, float *buf, unsigned *mism,
                                                       unsigned epoch0, float base) {
    const int b = blockIdx.x, nb = gridDim.x;
    unsigned bad = 0;
    for (int i = 0; i < iters; ++i) {
        float *pb = buf + (size_t)i * nb * 64;
        if (threadIdx.x < 64)
            __hip_atomic_store(&pb[b * 64 + threadIdx.x], base + (float)(i * 1000 + b), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (!grid_barrier_tree(cnt + 128, cnt + 64, cnt + 2048, abort_flag, epoch0 + (unsigned)(i + 1))) return;
        if (threadIdx.x < 64) {
            const int src = (b + 37) % nb;
            const float v = pb[src * 64 + threadIdx.x];  // plain load
            if (v != base + (float)(i * 1000 + src)) ++bad;
        }
    }
    if (bad) atomicAdd(mism, bad);
}

void run_unique(int blocks, int iters, int launches) {
    unsigned *cnt, *mism;
    float *buf;
    const size_t cbytes = 8192 + (size_t)blocks * 128 + 128;
    CK(hipMalloc(&cnt, cbytes));
    CK(hipMalloc(&buf, (size_t)iters * blocks * 64 * sizeof(float)));
    CK(hipMalloc(&mism, 4));
    CK(hipMemset(mism, 0, 4));
    CK(hipMemset(cnt, 0, cbytes));
    CK(hipMemset(buf, 0, (size_t)iters * blocks * 64 * sizeof(float)));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    unsigned *ab = cnt + 32;
    CK(hipEventRecord(e0, 0));
    for (int l = 0; l < launches; ++l) {
        unsigned epoch0 = (unsigned)(l * iters);
        float base = (float)(l * 7);
        void *args[] = {&cnt, &ab, &iters, &buf, &mism, &epoch0, &base};
        CK(hipLaunchCooperativeKernel((const void *)unique_buf_loop, dim3(blocks), dim3(512), args, 0, 0));
    }
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    unsigned h[64], hm;
    CK(hipMemcpy(h, cnt, 256, hipMemcpyDeviceToHost));
    CK(hipMemcpy(&hm, mism, 4, hipMemcpyDeviceToHost));
    printf("unique buffers, plain loads, %d launches x %d phases, blocks=%d: %7.3f us/phase abort=%u mismatches=%u\n", launches,
           iters, blocks, 1000.0f * ms / (iters * launches), h[32], hm);
}

template <int V> void run16(int blocks, int threads, int iters) {
    unsigned *cnt, *mism;
    float *buf;
    const size_t cbytes = 8192 + (size_t)blocks * 128 + 128;
    CK(hipMalloc(&cnt, cbytes));
    CK(hipMalloc(&buf, 2 * blocks * 64 * sizeof(float)));
    CK(hipMalloc(&mism, 4));
    CK(hipMemset(mism, 0, 4));
    CK(hipMemset(buf, 0, 2 * blocks * 64 * sizeof(float)));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    unsigned *ab = cnt + 32;
    void *args[] = {&cnt, &ab, &iters, &buf, &mism};
    for (int rep = 0; rep < 2; ++rep) {
        CK(hipMemset(cnt, 0, cbytes));
        CK(hipEventRecord(e0, 0));
        CK(hipLaunchCooperativeKernel((const void *)barrier16_loop<V>, dim3(blocks), dim3(threads), args, 0, 0));
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
    }
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    unsigned h[64], hm;
    CK(hipMemcpy(h, cnt, 256, hipMemcpyDeviceToHost));
    CK(hipMemcpy(&hm, mism, 4, hipMemcpyDeviceToHost));
    printf("%-34s blocks=%4d threads=%3d : %7.3f us/barrier  abort=%u mismatches=%u\n", V == 3 ? "16 counters + handoff check" : V == 4 ? "last-arriver flags + handoff" : V == 5 ? "tree arrival + flags + handoff" : "no barrier (handoff traffic only)", blocks, threads,
           1000.0f * ms / iters, h[32], hm);
    CK(hipFree(cnt));
    CK(hipFree(buf));
    CK(hipFree(mism));
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
    for (int blocks : {64, 128, 256, 512}) run16<3>(blocks, 512, iters);
    for (int blocks : {64, 128, 256, 512}) run16<4>(blocks, 512, iters);
    for (int blocks : {64, 128, 256, 512}) run16<5>(blocks, 512, iters);
    for (int blocks : {64, 256}) run16<6>(blocks, 512, iters);
    run_unique(256, 256, 200);
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
