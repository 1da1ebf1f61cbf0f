// Distance_G (net.py:266-308): three Linear -> per-row normalisation -> ReLU blocks and a sigmoid output.
// 5.2 MFLOP per hologram against 10.5 MB of weights: weight-read bound, plain fp32 VALU dot products.
#include "common.h"

namespace holo {

constexpr int kBT = 8;  // samples per wave

// one wave per (output feature n, tile of 8 samples); lanes split K in float4 steps
__global__ __launch_bounds__(256) void linear_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                     const float* __restrict__ bias, float* __restrict__ y, int batch,
                                                     int k, int n) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int on = blockIdx.x * 4 + wave;
    const int b0 = blockIdx.y * kBT;
    if (on >= n) return;
    float acc[kBT];
#pragma unroll
    for (int i = 0; i < kBT; ++i) acc[i] = 0.0f;
    const float* __restrict__ wr = w + (size_t)on * k;
    for (int kk = lane * 4; kk < k; kk += 256) {
        const float4 wv = *reinterpret_cast<const float4*>(wr + kk);
#pragma unroll
        for (int i = 0; i < kBT; ++i) {
            const int b = b0 + i < batch ? b0 + i : batch - 1;
            const float4 xv = *reinterpret_cast<const float4*>(x + (size_t)b * k + kk);
            // one explicit FMA chain per sample slot: with `acc += x.x*w.x + x.y*w.y + ...` hipcc contracted and paired the eight
            // slots differently, and a sample's result depended (by an ulp) on its position in the batch (found in round 3 by
            // comparing a sharded pass with the unsharded batch)
            acc[i] = fmaf(xv.x, wv.x, acc[i]);
            acc[i] = fmaf(xv.y, wv.y, acc[i]);
            acc[i] = fmaf(xv.z, wv.z, acc[i]);
            acc[i] = fmaf(xv.w, wv.w, acc[i]);
        }
    }
#pragma unroll
    for (int i = 0; i < kBT; ++i) {
        float v = acc[i];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
        acc[i] = v;
    }
    if (lane == 0) {
        const float bv = bias[on];
#pragma unroll
        for (int i = 0; i < kBT; ++i)
            if (b0 + i < batch) y[(size_t)(b0 + i) * n + on] = acc[i] + bv;
    }
}

__device__ __forceinline__ float block_sum(float v, float* red) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

// InstanceNorm1d(num_features) applied to a 2-D (B,N) tensor = each row normalised over its N features,
// biased variance, eps inside the sqrt, no affine (net.py:301-305), followed by ReLU.
__global__ __launch_bounds__(256) void rownorm_relu_kernel(const float* __restrict__ x, float* __restrict__ y, int n,
                                                           float eps) {
    __shared__ float red[4];
    const float* __restrict__ xr = x + (size_t)blockIdx.x * n;
    float* __restrict__ yr = y + (size_t)blockIdx.x * n;
    float s = 0.0f;
    for (int i = threadIdx.x; i < n; i += 256) s += xr[i];
    const float mu = block_sum(s, red) / (float)n;
    float q = 0.0f;
    for (int i = threadIdx.x; i < n; i += 256) {
        const float d = xr[i] - mu;
        q += d * d;
    }
    const float var = block_sum(q, red) / (float)n;
    const float inv = 1.0f / sqrtf(var + eps);
    for (int i = threadIdx.x; i < n; i += 256) yr[i] = fmaxf((xr[i] - mu) * inv, 0.0f);
}

// one wave per sample: sigmoid(bias + x.w)
__global__ __launch_bounds__(256) void linear1_sigmoid_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                              const float* __restrict__ bias, float* __restrict__ y,
                                                              int batch, int k) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b = blockIdx.x * 4 + wave;
    if (b >= batch) return;
    float acc = 0.0f;
    for (int kk = lane; kk < k; kk += 64) acc += x[(size_t)b * k + kk] * w[kk];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
    if (lane == 0) y[b] = 1.0f / (1.0f + expf(-(acc + bias[0])));
}

}  // namespace holo

extern "C" {

int holo_linear_f32(const float* x, const float* w, const float* bias, float* y, int batch, int k, int n, void* stream) {
    using namespace holo;
    HOLO_REQUIRE(x && w && bias && y, HOLO_ERR_ARG, "holo_linear_f32: null pointer");
    HOLO_REQUIRE(batch > 0 && k > 0 && k % 4 == 0 && n > 0, HOLO_ERR_ARG, "holo_linear_f32: batch=%d k=%d n=%d", batch, k, n);
    const int by = ceil_div(batch, kBT);
    HOLO_REQUIRE(by <= 65535, HOLO_ERR_UNSUPPORTED, "holo_linear_f32: batch %d too large for one call", batch);
    hipLaunchKernelGGL(linear_kernel, dim3(ceil_div(n, 4), by), dim3(256), 0, as_stream(stream), x, w, bias, y, batch, k, n);
    return check_launch("holo_linear_f32");
}

int holo_rownorm_relu_f32(const float* x, float* y, int batch, int n, float eps, void* stream) {
    using namespace holo;
    HOLO_REQUIRE(x && y, HOLO_ERR_ARG, "holo_rownorm_relu_f32: null pointer");
    HOLO_REQUIRE(batch > 0 && n > 0, HOLO_ERR_ARG, "holo_rownorm_relu_f32: batch=%d n=%d", batch, n);
    hipLaunchKernelGGL(rownorm_relu_kernel, dim3(batch), dim3(256), 0, as_stream(stream), x, y, n, eps);
    return check_launch("holo_rownorm_relu_f32");
}

int holo_linear1_sigmoid_f32(const float* x, const float* w, const float* bias, float* y, int batch, int k, void* stream) {
    using namespace holo;
    HOLO_REQUIRE(x && w && bias && y, HOLO_ERR_ARG, "holo_linear1_sigmoid_f32: null pointer");
    HOLO_REQUIRE(batch > 0 && k > 0, HOLO_ERR_ARG, "holo_linear1_sigmoid_f32: batch=%d k=%d", batch, k);
    hipLaunchKernelGGL(linear1_sigmoid_kernel, dim3(ceil_div(batch, 4)), dim3(256), 0, as_stream(stream), x, w, bias, y, batch, k);
    return check_launch("holo_linear1_sigmoid_f32");
}

}  // extern "C"
