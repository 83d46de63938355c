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

}  // namespace mw
