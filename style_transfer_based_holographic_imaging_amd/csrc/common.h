// Shared host/device helpers for the holo_hip C-ABI library (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <atomic>
#include <cstdarg>
#include <cstdio>
#include <mutex>
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

// Per-device one-time set-up (hipFuncSetAttribute for > 48 KB of dynamic LDS is a property of the function ON A DEVICE).
// `run(f)` executes f at most once per device and publishes the device's bit only AFTER f has returned (release), under a
// mutex: a second host thread on the same device either sees the bit (acquire: the attributes are set) or waits for the
// first one to finish — it can never launch ahead of the set-up.  One static instance per launcher.
struct DeviceOnce {
    std::atomic<unsigned long long> done{0};
    std::mutex mu;
    template <class F>
    void run(F&& f) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev > 63) {   // unknown device: do the (idempotent) set-up every time
            std::lock_guard<std::mutex> lk(mu);
            f();
            return;
        }
        const unsigned long long bit = 1ull << dev;
        if (done.load(std::memory_order_acquire) & bit) return;
        std::lock_guard<std::mutex> lk(mu);
        if (done.load(std::memory_order_relaxed) & bit) return;
        f();
        done.fetch_or(bit, std::memory_order_release);
    }
};

// compute units of the current device (cached per device; 256 on MI355X): sizes persistent grids
inline int device_cus() {
    static std::atomic<int> n_cu[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev > 63) return 256;
    int cus = n_cu[dev].load(std::memory_order_relaxed);
    if (cus <= 0) {
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
        n_cu[dev].store(cus, std::memory_order_relaxed);
    }
    return cus;
}

// Kernels whose summation grouping differs from the standard family (fp32 rounding of partial sums: not bit-identical to it) are
// chosen from PER-IMAGE quantities, and only for batches in the open-ended batch bucket (engine.Plan.BATCH_TOPS: 1 | 2 | 3-5 | 6-8 |
// 9-12 | 13+): a hologram's arithmetic then does not depend on the batch it arrives in within a bucket, and all batches >= 13 — every
// shard of a multi-GPU job, every micro-batch of a fixed-total pass — take the same kernels.
constexpr int kOpenBatchBucket = 13;

// tile selection of the 3x3 / transposed convolutions: 0 = automatic (shipped), 1..3 force a variant (holo_debug_set_conv_tile,
// used by the tests that hold the variants bit-identical)
int conv_tile_override();

__host__ __device__ inline int ceil_div(int a, int b) { return (a + b - 1) / b; }

// ReflectionPad2d(1) index: -1 -> 1, n -> n-2 (n >= 2); anything further out is clamped so that
// tile overhang past the image stays a valid (masked-at-store) address.
// orders a wave's LDS reads behind its own earlier LDS writes (wave-private regions: the LDS unit executes one wave's DS
// instructions in order, so no block barrier is needed)
__device__ __forceinline__ void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ int reflect1(int i, int n) {
    i = i < 0 ? -i : i;
    i = i >= n ? 2 * n - 2 - i : i;
    return i < 0 ? 0 : (i >= n ? n - 1 : i);
}

// "f32s" activations: an fp32 value as two fp16 halves, a = hi + lo' * 2^-11 (22-23 significant bits), stored planar per
// 16 consecutive elements: [16 x hi | 16 x lo'] = 64 bytes per group, so that a 16-byte read is 8 consecutive channels of
// one half — one operand of v_mfma_f32_32x32x16_f16.  Channel counts are multiples of 16 and buffers 64-byte aligned, so
// the groups tile any NHWC tensor linearly and an element's place follows from its address alone.
struct split16 {
    _Float16 hi, lo;
    split16() = default;
    __host__ __device__ explicit split16(float f) : hi((_Float16)f), lo((_Float16)0.0f) {}   // so that dead store paths compile
};
typedef _Float16 halfx4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void split_halves(float f, _Float16* hi, _Float16* lo) {
    f = fminf(fmaxf(f, -65504.0f), 65504.0f);            // fp16 range; values beyond saturate
    *hi = (_Float16)f;
    *lo = (_Float16)((f - (float)*hi) * 2048.0f);
}
// 4 consecutive elements starting at element pointer p (element index a multiple of 4)
__device__ __forceinline__ float4 split_load4(const split16* p) {
    const unsigned long long a = reinterpret_cast<unsigned long long>(p);
    const unsigned char* g = reinterpret_cast<const unsigned char*>(a & ~63ull) + ((a & 63ull) >> 1);
    const halfx4 hi = *reinterpret_cast<const halfx4*>(g);
    const halfx4 lo = *reinterpret_cast<const halfx4*>(g + 32);
    return make_float4((float)hi[0] + (float)lo[0] * 0.00048828125f, (float)hi[1] + (float)lo[1] * 0.00048828125f,
                       (float)hi[2] + (float)lo[2] * 0.00048828125f, (float)hi[3] + (float)lo[3] * 0.00048828125f);
}
__device__ __forceinline__ void split_store4(split16* p, float4 v) {
    const unsigned long long a = reinterpret_cast<unsigned long long>(p);
    unsigned char* g = reinterpret_cast<unsigned char*>(a & ~63ull) + ((a & 63ull) >> 1);
    halfx4 hi, lo;
    const float f[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        _Float16 h, l;
        split_halves(f[i], &h, &l);
        hi[i] = h;
        lo[i] = l;
    }
    *reinterpret_cast<halfx4*>(g) = hi;
    *reinterpret_cast<halfx4*>(g + 32) = lo;
}

}  // namespace holo
