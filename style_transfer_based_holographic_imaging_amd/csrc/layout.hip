// NCHW <-> NHWC transposes for callers that hand over / expect the reference's plane-major tensors
// (e.g. a (B,512,h,w) feature map passed straight to the decoder).  32x32 LDS tiles, both sides coalesced.
#include "common.h"

namespace holo {

// in: rows x cols (row-major) per batch item -> out: cols x rows
__global__ __launch_bounds__(256) void transpose_kernel(const float* __restrict__ in, float* __restrict__ out, int rows,
                                                        int cols) {
    __shared__ float tile[32][33];
    const size_t base = (size_t)blockIdx.z * rows * cols;
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
#pragma unroll
    for (int i = 0; i < 32; i += 8) {
        const int r = r0 + ty + i, c = c0 + tx;
        if (r < rows && c < cols) tile[ty + i][tx] = in[base + (size_t)r * cols + c];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 32; i += 8) {
        const int c = c0 + ty + i, r = r0 + tx;
        if (r < rows && c < cols) out[base + (size_t)c * rows + r] = tile[tx][ty + i];
    }
}

template <typename TI, typename TO>
__global__ __launch_bounds__(256) void cast_kernel(const TI* __restrict__ in, TO* __restrict__ out, long n) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) out[i] = (TO)(float)in[i];
}

template <typename TI, typename TO>
static int launch_cast(const TI* x, TO* y, long n, hipStream_t s, const char* what) {
    HOLO_REQUIRE(x && y && n > 0, HOLO_ERR_ARG, "%s: null pointer or empty", what);
    long blocks = (n + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL((cast_kernel<TI, TO>), dim3((unsigned)blocks), dim3(256), 0, s, x, y, n);
    return check_launch(what);
}

static int launch_transpose(const float* x, float* y, int batch, int rows, int cols, hipStream_t s, const char* what) {
    HOLO_REQUIRE(x && y, HOLO_ERR_ARG, "%s: null pointer", what);
    HOLO_REQUIRE(batch > 0 && batch <= 65535 && rows > 0 && cols > 0, HOLO_ERR_ARG, "%s: batch=%d rows=%d cols=%d", what,
                 batch, rows, cols);
    HOLO_REQUIRE(ceil_div(rows, 32) <= 65535, HOLO_ERR_UNSUPPORTED, "%s: %d rows too many", what, rows);
    hipLaunchKernelGGL(transpose_kernel, dim3(ceil_div(cols, 32), ceil_div(rows, 32), batch), dim3(256), 0, s, x, y, rows,
                       cols);
    return check_launch(what);
}

// fp32 <-> "f32s" (split halves, planar per 16 elements; common.h): 4 elements per thread, n a multiple of 16
template <bool TO_SPLIT>
__global__ __launch_bounds__(256) void split_cast_kernel(const void* __restrict__ in, void* __restrict__ out, long n4) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        if constexpr (TO_SPLIT) split_store4(static_cast<split16*>(out) + 4 * i, static_cast<const float4*>(in)[i]);
        else static_cast<float4*>(out)[i] = split_load4(static_cast<const split16*>(in) + 4 * i);
    }
}
template <bool TO_SPLIT>
static int launch_split_cast(const void* x, void* y, long n, hipStream_t s, const char* what) {
    HOLO_REQUIRE(x && y && n > 0 && n % 16 == 0, HOLO_ERR_ARG, "%s: null pointer, empty, or length %ld not a multiple of 16", what, n);
    HOLO_REQUIRE(((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 63) == 0, HOLO_ERR_ARG, "%s: buffers must be 64-byte aligned", what);
    long blocks = (n / 4 + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL((split_cast_kernel<TO_SPLIT>), dim3((unsigned)blocks), dim3(256), 0, s, x, y, n / 4);
    return check_launch(what);
}

}  // namespace holo

extern "C" {
int holo_cast_f32_to_f32s(const float* x, void* y, long n, void* stream) {
    return holo::launch_split_cast<true>(x, y, n, holo::as_stream(stream), "holo_cast_f32_to_f32s");
}
int holo_cast_f32s_to_f32(const void* x, float* y, long n, void* stream) {
    return holo::launch_split_cast<false>(x, y, n, holo::as_stream(stream), "holo_cast_f32s_to_f32");
}
int holo_nhwc_to_nchw_f32(const float* x, float* y, int batch, int hw, int c, void* stream) {
    return holo::launch_transpose(x, y, batch, hw, c, holo::as_stream(stream), "holo_nhwc_to_nchw_f32");
}
int holo_nchw_to_nhwc_f32(const float* x, float* y, int batch, int hw, int c, void* stream) {
    return holo::launch_transpose(x, y, batch, c, hw, holo::as_stream(stream), "holo_nchw_to_nhwc_f32");
}
int holo_cast_f32_to_bf16(const float* x, uint16_t* y, long n, void* stream) {
    return holo::launch_cast(x, reinterpret_cast<__bf16*>(y), n, holo::as_stream(stream), "holo_cast_f32_to_bf16");
}
int holo_cast_bf16_to_f32(const uint16_t* x, float* y, long n, void* stream) {
    return holo::launch_cast(reinterpret_cast<const __bf16*>(x), y, n, holo::as_stream(stream), "holo_cast_bf16_to_f32");
}
}
