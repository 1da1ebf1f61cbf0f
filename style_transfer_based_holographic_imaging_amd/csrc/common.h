// Shared host/device helpers for the holo_hip C-ABI library (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>
#include "../../include/holo_hip.h"

namespace holo {

void set_error(const char* fmt, ...);

#define HOLO_REQUIRE(cond, code, ...)          \
    do {                                       \
        if (!(cond)) {                         \
            ::holo::set_error(__VA_ARGS__);    \
            return (code);                     \
        }                                      \
    } while (0)

inline int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: %s", what, hipGetErrorString(e));
        return HOLO_ERR_LAUNCH;
    }
    return HOLO_OK;
}

inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

__host__ __device__ inline int ceil_div(int a, int b) { return (a + b - 1) / b; }

// ReflectionPad2d(1) index: -1 -> 1, n -> n-2 (n >= 2); anything further out is clamped so that
// tile overhang past the image stays a valid (masked-at-store) address.
__device__ __forceinline__ int reflect1(int i, int n) {
    i = i < 0 ? -i : i;
    i = i >= n ? 2 * n - 2 - i : i;
    return i < 0 ? 0 : (i >= n ? n - 1 : i);
}

}  // namespace holo
