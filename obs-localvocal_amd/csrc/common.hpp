// common.hpp — shared declarations of the MI355X Whisper engine (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <string>

namespace mw {

// ---- logging (whisper_log_set; /root/reference/src/whisper-utils/whisper-processing.cpp:46-57) ----
void log_msg(int level, const char *fmt, ...) __attribute__((format(printf, 2, 3)));
void set_last_error(const std::string &msg);  // thread-local slot behind mw_last_error()
constexpr int LOG_ERROR = 2, LOG_WARN = 3, LOG_INFO = 4, LOG_DEBUG = 5;

struct hip_error : std::runtime_error {
    using std::runtime_error::runtime_error;
};

#define MW_HIP(expr)                                                                                       \
    do {                                                                                                   \
        hipError_t _e = (expr);                                                                            \
        if (_e != hipSuccess) {                                                                            \
            char _b[512];                                                                                  \
            snprintf(_b, sizeof(_b), "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            throw ::mw::hip_error(_b);                                                                     \
        }                                                                                                  \
    } while (0)

#define MW_CHECK(cond, msg)                                                                     \
    do {                                                                                        \
        if (!(cond)) {                                                                          \
            char _b[512];                                                                       \
            snprintf(_b, sizeof(_b), "check failed: %s — %s (%s:%d)", #cond, msg, __FILE__, __LINE__); \
            throw std::runtime_error(_b);                                                       \
        }                                                                                       \
    } while (0)

// ---- element types ----
using bf16_t = uint16_t;  // raw bfloat16 bits

enum class DType : int { F32 = 0, BF16 = 1 };
inline size_t dtype_size(DType t) { return t == DType::F32 ? 4 : 2; }

inline bf16_t f32_to_bf16_host(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (bf16_t)((u >> 16) | 0x40);  // NaN stays NaN
    u += 0x7fffu + ((u >> 16) & 1u);
    return (bf16_t)(u >> 16);
}
inline float bf16_to_f32_host(bf16_t h) {
    uint32_t u = (uint32_t)h << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}

template <typename T> inline T ceil_div(T a, T b) { return (a + b - 1) / b; }

// ---- lane exchanges without LDS -------------------------------------------------------------------------------------
// __shfl_xor compiles to ds_bpermute_b32: an LDS round trip behind its own s_waitcnt.  These give v (+ or max) the value of
// lane ^ N on DPP moves (N = 1, 2: quad permutes; 4: two bank-masked row shifts; 8: a row rotate) and on gfx950's
// v_permlane16_swap / v_permlane32_swap (N = 16, 32: both copies of v go in, every lane ends up with its own row's /
// half's value in one and the partner's in the other).  Same partners, same operation: a reduction written with them in
// the order of its shuffle form gives the same bits.
#ifdef __HIPCC__
template <int N>
__device__ __forceinline__ float lane_xor(float v) {  // N = 1, 2, 4, 8
    static_assert(N == 1 || N == 2 || N == 4 || N == 8, "DPP forms");
    const int x = __float_as_int(v);
    if constexpr (N == 1) return __int_as_float(__builtin_amdgcn_update_dpp(x, x, 0xB1, 0xf, 0xf, false));       // quad_perm [1,0,3,2]
    else if constexpr (N == 2) return __int_as_float(__builtin_amdgcn_update_dpp(x, x, 0x4E, 0xf, 0xf, false));  // quad_perm [2,3,0,1]
    else if constexpr (N == 8) return __int_as_float(__builtin_amdgcn_update_dpp(x, x, 0x128, 0xf, 0xf, false)); // row_ror:8
    else {
        int t = __builtin_amdgcn_update_dpp(x, x, 0x104, 0xf, 0x5, false);  // row_shl:4 into quads 0 and 2: lane + 4
        t = __builtin_amdgcn_update_dpp(t, x, 0x114, 0xf, 0xa, false);      // row_shr:4 into quads 1 and 3: lane - 4
        return __int_as_float(t);
    }
}
template <int N>
__device__ __forceinline__ float xor_sum(float v) {  // v + v[lane ^ N]
    if constexpr (N == 16) {
        const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
        return __uint_as_float(r[0]) + __uint_as_float(r[1]);
    } else if constexpr (N == 32) {
        const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
        return __uint_as_float(r[0]) + __uint_as_float(r[1]);
    } else {
        return v + lane_xor<N>(v);
    }
}
template <int N>
__device__ __forceinline__ float xor_max(float v) {  // max(v, v[lane ^ N])
    if constexpr (N == 16) {
        const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
        return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
    } else if constexpr (N == 32) {
        const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
        return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
    } else {
        return fmaxf(v, lane_xor<N>(v));
    }
}
// the whole wave, partners from lane ^ 32 down to lane ^ 1 (the order of `for (o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o)`)
__device__ __forceinline__ float wave_sum_desc(float v) {
    return xor_sum<1>(xor_sum<2>(xor_sum<4>(xor_sum<8>(xor_sum<16>(xor_sum<32>(v))))));
}
__device__ __forceinline__ float wave_max_desc(float v) {
    return xor_max<1>(xor_max<2>(xor_max<4>(xor_max<8>(xor_max<16>(xor_max<32>(v))))));
}
#endif

}  // namespace mw
