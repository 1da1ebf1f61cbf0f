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

// ---- the whole head in four launches (holo_distance_head_f32) -----------------------------------------------------------------------
// At small batches the head is launch latency: Linear, norm+ReLU, Linear, norm+ReLU, Linear, norm+ReLU, Linear+sigmoid and the
// concatenation in front are 8 launches of 4-7 us for 10 MB of weights.  Here the normalisation moves into the CONSUMER: a block of
// the next Linear stages its 8 input rows in LDS anyway, so the wave that fetches a row computes its mean and biased variance itself
// (two passes over its registers, lane-sequential partial sums, xor butterfly: every lane ends with the same value) and stores
// (x - mu) / sqrt(var + eps), ReLU'd; the first Linear reads its row from the two statistics tensors.
// Same arithmetic per sample whatever its slot or batch: one fmaf chain per slot, fixed summation trees.
constexpr int kHeadMaxK = 1024;   // input features of a normalising layer (registers: 8 rows x k/64 values)

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

// NORM: the input rows are raw Linear outputs, normalised + ReLU'd on load.  TWO: row b = [x[b][0..k1) | x2[b][0..k-k1)].
// A block = 8 samples x (4 waves x kHeadFeat) output features: every wave requests its weights, then the four waves bring the
// 8 input rows into LDS (two rows per wave, normalised there if NORM) and every wave runs its features against all 8 rows — a row
// is fetched (and normalised) once per block, and the weight latency runs beside the staging instead of behind it.
#ifndef HOLO_HEAD_FEAT
#define HOLO_HEAD_FEAT 2
#endif
constexpr int kHeadFeat = HOLO_HEAD_FEAT;      // output features per wave (2: 128 blocks for the 1024-feature layers)
template <bool NORM, bool TWO>
__global__ __launch_bounds__(256) void head_linear_kernel(const float* __restrict__ x, const float* __restrict__ x2, int k1,
                                                          const float* __restrict__ w, const float* __restrict__ bias,
                                                          float* __restrict__ y, int batch, int k, int n, float eps) {
    __shared__ __attribute__((aligned(16))) float rows[kBT][kHeadMaxK];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b0 = blockIdx.y * kBT;
    constexpr int kQ = kHeadMaxK / 256;    // float4 steps per lane and row
    // the wave's weights first: they do not depend on the rows, so their latency runs beside the staging (and normalising) below
    const int on0 = (blockIdx.x * 4 + wave) * kHeadFeat;
    float4 wv[kQ][kHeadFeat];
#pragma unroll
    for (int q = 0; q < kQ; ++q)
#pragma unroll
        for (int f = 0; f < kHeadFeat; ++f) {
            const int kk = lane * 4 + 256 * q;
            const int on = on0 + f < n ? on0 + f : n - 1;
            wv[q][f] = kk < k ? *reinterpret_cast<const float4*>(w + (size_t)on * k + kk) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    // ---- the block's 8 rows -> LDS (wave w: rows 2w, 2w+1) ----
#pragma unroll
    for (int i2 = 0; i2 < 2; ++i2) {
        const int i = 2 * wave + i2;
        const int b = b0 + i < batch ? b0 + i : batch - 1;
        float4 xv[kQ];
#pragma unroll
        for (int q = 0; q < kQ; ++q) {
            const int kk = lane * 4 + 256 * q;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (kk < k) {
                if (TWO) v = kk < k1 ? *reinterpret_cast<const float4*>(x + (size_t)b * k1 + kk)
                                     : *reinterpret_cast<const float4*>(x2 + (size_t)b * (k - k1) + (kk - k1));
                else v = *reinterpret_cast<const float4*>(x + (size_t)b * k + kk);
            }
            xv[q] = v;
        }
        if (NORM) {
            float s = 0.0f;
#pragma unroll
            for (int q = 0; q < kQ; ++q)
                if (lane * 4 + 256 * q < k) s += (xv[q].x + xv[q].y) + (xv[q].z + xv[q].w);
            const float mu = wave_sum(s) / (float)k;
            float qs = 0.0f;
#pragma unroll
            for (int q = 0; q < kQ; ++q)
                if (lane * 4 + 256 * q < k) {
                    const float dx = xv[q].x - mu, dy = xv[q].y - mu, dz = xv[q].z - mu, dw = xv[q].w - mu;
                    qs += (dx * dx + dy * dy) + (dz * dz + dw * dw);
                }
            const float inv = 1.0f / sqrtf(wave_sum(qs) / (float)k + eps);
#pragma unroll
            for (int q = 0; q < kQ; ++q)
                xv[q] = make_float4(fmaxf((xv[q].x - mu) * inv, 0.0f), fmaxf((xv[q].y - mu) * inv, 0.0f),
                                    fmaxf((xv[q].z - mu) * inv, 0.0f), fmaxf((xv[q].w - mu) * inv, 0.0f));
        }
#pragma unroll
        for (int q = 0; q < kQ; ++q) *reinterpret_cast<float4*>(&rows[i][lane * 4 + 256 * q]) = xv[q];
    }
    __syncthreads();
    // ---- kHeadFeat features per wave against the 8 rows ----
    float acc[kHeadFeat][kBT];
#pragma unroll
    for (int f = 0; f < kHeadFeat; ++f)
#pragma unroll
        for (int i = 0; i < kBT; ++i) acc[f][i] = 0.0f;
#pragma unroll
    for (int q = 0; q < kQ; ++q) {
        const int kk = lane * 4 + 256 * q;
        if (kk < k) {
#pragma unroll
            for (int i = 0; i < kBT; ++i) {
                const float4 xv = *reinterpret_cast<const float4*>(&rows[i][kk]);
#pragma unroll
                for (int f = 0; f < kHeadFeat; ++f) {   // one explicit FMA chain per (feature, sample slot), as in linear_kernel
                    acc[f][i] = fmaf(xv.x, wv[q][f].x, acc[f][i]);
                    acc[f][i] = fmaf(xv.y, wv[q][f].y, acc[f][i]);
                    acc[f][i] = fmaf(xv.z, wv[q][f].z, acc[f][i]);
                    acc[f][i] = fmaf(xv.w, wv[q][f].w, acc[f][i]);
                }
            }
        }
    }
#pragma unroll
    for (int f = 0; f < kHeadFeat; ++f)
#pragma unroll
        for (int i = 0; i < kBT; ++i) acc[f][i] = wave_sum(acc[f][i]);
    if (lane == 0) {
#pragma unroll
        for (int f = 0; f < kHeadFeat; ++f)
            if (on0 + f < n) {
                const float bv = bias[on0 + f];
#pragma unroll
                for (int i = 0; i < kBT; ++i)
                    if (b0 + i < batch) y[(size_t)(b0 + i) * n + on0 + f] = acc[f][i] + bv;
            }
    }
}

// one wave per sample: sigmoid(bias + relu(norm(x)) . w), x the raw output of the third Linear
__global__ __launch_bounds__(256) void head_out_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                       const float* __restrict__ bias, float* __restrict__ y, int batch, int k,
                                                       float eps) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b = blockIdx.x * 4 + wave;
    if (b >= batch) return;
    constexpr int kQ = kHeadMaxK / 256;
    float4 xv[kQ];
    float s = 0.0f;
#pragma unroll
    for (int q = 0; q < kQ; ++q) {
        const int kk = lane * 4 + 256 * q;
        xv[q] = kk < k ? *reinterpret_cast<const float4*>(x + (size_t)b * k + kk) : make_float4(0.f, 0.f, 0.f, 0.f);
        if (kk < k) s += (xv[q].x + xv[q].y) + (xv[q].z + xv[q].w);
    }
    const float mu = wave_sum(s) / (float)k;
    float qs = 0.0f;
#pragma unroll
    for (int q = 0; q < kQ; ++q)
        if (lane * 4 + 256 * q < k) {
            const float dx = xv[q].x - mu, dy = xv[q].y - mu, dz = xv[q].z - mu, dw = xv[q].w - mu;
            qs += (dx * dx + dy * dy) + (dz * dz + dw * dw);
        }
    const float inv = 1.0f / sqrtf(wave_sum(qs) / (float)k + eps);
    float acc = 0.0f;
#pragma unroll
    for (int q = 0; q < kQ; ++q) {
        const int kk = lane * 4 + 256 * q;
        if (kk < k) {
            const float4 wv = *reinterpret_cast<const float4*>(w + kk);
            acc = fmaf(fmaxf((xv[q].x - mu) * inv, 0.0f), wv.x, acc);
            acc = fmaf(fmaxf((xv[q].y - mu) * inv, 0.0f), wv.y, acc);
            acc = fmaf(fmaxf((xv[q].z - mu) * inv, 0.0f), wv.z, acc);
            acc = fmaf(fmaxf((xv[q].w - mu) * inv, 0.0f), wv.w, acc);
        }
    }
    acc = wave_sum(acc);
    if (lane == 0) y[b] = 1.0f / (1.0f + expf(-(acc + bias[0])));
}

}  // namespace holo

extern "C" {

size_t holo_distance_head_workspace_bytes(int batch, int c) {
    return batch > 0 && c > 0 ? (size_t)2 * batch * 2 * c * sizeof(float) : 0;
}

int holo_distance_head_f32(const float* mean, const float* std, const float* w1, const float* b1, const float* w2, const float* b2,
                           const float* w3, const float* b3, const float* wo, const float* bo, float* out, int batch, int c, float eps,
                           float* workspace, size_t workspace_bytes, void* stream) {
    using namespace holo;
    HOLO_REQUIRE(mean && std && w1 && b1 && w2 && b2 && w3 && b3 && wo && bo && out && workspace, HOLO_ERR_ARG,
                 "holo_distance_head_f32: null pointer");
    HOLO_REQUIRE(batch > 0 && c > 0, HOLO_ERR_ARG, "holo_distance_head_f32: batch=%d c=%d", batch, c);
    HOLO_REQUIRE(c % 256 == 0 && 2 * c <= kHeadMaxK, HOLO_ERR_UNSUPPORTED,
                 "holo_distance_head_f32: c=%d must be a multiple of 256 with 2c <= %d (use the per-layer entry points otherwise)", c, kHeadMaxK);
    HOLO_REQUIRE(workspace_bytes >= holo_distance_head_workspace_bytes(batch, c), HOLO_ERR_WORKSPACE,
                 "holo_distance_head_f32: workspace %zu < required %zu", workspace_bytes, holo_distance_head_workspace_bytes(batch, c));
    const int by = ceil_div(batch, kBT);
    HOLO_REQUIRE(by <= 65535, HOLO_ERR_UNSUPPORTED, "holo_distance_head_f32: batch %d too large for one call", batch);
    const int f = 2 * c;
    float* y1 = workspace;
    float* y2 = workspace + (size_t)batch * f;
    hipStream_t st = as_stream(stream);
    constexpr int per_block = 4 * kHeadFeat;     // output features per block
    hipLaunchKernelGGL((head_linear_kernel<false, true>), dim3(ceil_div(f, per_block), by), dim3(256), 0, st, mean, std, c, w1, b1, y1, batch, f, f, eps);
    hipLaunchKernelGGL((head_linear_kernel<true, false>), dim3(ceil_div(f, per_block), by), dim3(256), 0, st, y1, nullptr, 0, w2, b2, y2, batch, f, f, eps);
    hipLaunchKernelGGL((head_linear_kernel<true, false>), dim3(ceil_div(c, per_block), by), dim3(256), 0, st, y2, nullptr, 0, w3, b3, y1, batch, f, c, eps);
    hipLaunchKernelGGL(head_out_kernel, dim3(ceil_div(batch, 4)), dim3(256), 0, st, y1, wo, bo, out, batch, c, eps);
    return check_launch("holo_distance_head_f32");
}

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
