// Per-channel feature statistics and the stored-style AdaIN (function.py:4-23, net.py:235-246), NHWC.
// Both are HBM-bound: one read of the (B,HW,C) map for the statistics, one read + one write for the apply.
#include "common.h"

namespace holo {

typedef __bf16 bf16_t;
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

// 4 consecutive channels as fp32, from fp32 or bf16 storage
template <typename T>
__device__ __forceinline__ float4 load4(const T* p) {
    if constexpr (__is_same(T, split16)) {
        return split_load4(p);
    } else if constexpr (sizeof(T) == 4) {
        return *reinterpret_cast<const float4*>(p);
    } else {
        const bf16x4 v = *reinterpret_cast<const bf16x4*>(p);
        return make_float4((float)v[0], (float)v[1], (float)v[2], (float)v[3]);
    }
}
template <typename T>
__device__ __forceinline__ void store4(T* p, float4 o) {
    if constexpr (__is_same(T, split16)) {
        split_store4(p, o);
    } else if constexpr (sizeof(T) == 4) {
        *reinterpret_cast<float4*>(p) = o;
    } else {
        const bf16x4 v = {(bf16_t)o.x, (bf16_t)o.y, (bf16_t)o.z, (bf16_t)o.w};
        *reinterpret_cast<bf16x4*>(p) = v;
    }
}

// grid (C/64, B); 16 lanes x float4 cover 64 channels of one pixel (256 contiguous bytes),
// 16 pixel groups per block stride over HW; sums are kept in float64 (the CPU reference's var/mean
// accumulate in double as well), reduced through LDS.
template <typename T>
__global__ __launch_bounds__(256) void channel_mean_std_kernel(const T* __restrict__ x, float* __restrict__ mean,
                                                               float* __restrict__ stdv, int hw, int c, float eps) {
    __shared__ double s_sum[16][64];
    __shared__ double s_sq[16][64];
    const int tid = threadIdx.x;
    const int cq = tid & 15, pg = tid >> 4;
    const int c0 = blockIdx.x * 64;
    const int b = blockIdx.y;
    const T* __restrict__ xb = x + (size_t)b * hw * c + c0 + 4 * cq;
    double s[4] = {0, 0, 0, 0}, q[4] = {0, 0, 0, 0};
    for (int p = pg; p < hw; p += 16) {
        const float4 v = load4<T>(xb + (size_t)p * c);
        s[0] += v.x; q[0] += (double)v.x * v.x;
        s[1] += v.y; q[1] += (double)v.y * v.y;
        s[2] += v.z; q[2] += (double)v.z * v.z;
        s[3] += v.w; q[3] += (double)v.w * v.w;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        s_sum[pg][4 * cq + j] = s[j];
        s_sq[pg][4 * cq + j] = q[j];
    }
    __syncthreads();
    if (tid < 64) {
        double ts = 0, tq = 0;
#pragma unroll
        for (int g = 0; g < 16; ++g) {
            ts += s_sum[g][tid];
            tq += s_sq[g][tid];
        }
        const double n = (double)hw;
        const double mu = ts / n;
        double var = (tq - ts * mu) / (n - 1.0);  // unbiased (function.py:9: .var(dim=2))
        var = var > 0 ? var : 0;
        mean[(size_t)b * c + c0 + tid] = (float)mu;
        stdv[(size_t)b * c + c0 + tid] = sqrtf((float)var + eps);  // function.py:9-10: var + eps, then sqrt, in fp32
    }
}

template <typename T>
__global__ __launch_bounds__(256) void adain_apply_kernel(const T* __restrict__ x, const float* __restrict__ mean,
                                                          const float* __restrict__ stdv,
                                                          const float* __restrict__ smean,
                                                          const float* __restrict__ sstd, long sstride, float alpha,
                                                          float one_minus_alpha, T* __restrict__ y, long total4,
                                                          int hw, int c) {
#pragma clang fp contract(off)
    const int c4n = c >> 2;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total4; i += (long)gridDim.x * 256) {
        const int c4 = (int)(i % c4n);
        const long b = i / ((long)c4n * hw);
        const float4 v = load4<T>(x + 4 * i);
        const float4 mu = reinterpret_cast<const float4*>(mean + b * c)[c4];
        const float4 sd = reinterpret_cast<const float4*>(stdv + b * c)[c4];
        const float4 sm = reinterpret_cast<const float4*>(smean + b * sstride)[c4];
        const float4 ss = reinterpret_cast<const float4*>(sstd + b * sstride)[c4];
        float4 o;
        // net.py:241-246, same operation order: ((x-mu)/sd)*ss + sm, then alpha*t + (1-alpha)*x
        o.x = alpha * (((v.x - mu.x) / sd.x) * ss.x + sm.x) + one_minus_alpha * v.x;
        o.y = alpha * (((v.y - mu.y) / sd.y) * ss.y + sm.y) + one_minus_alpha * v.y;
        o.z = alpha * (((v.z - mu.z) / sd.z) * ss.z + sm.z) + one_minus_alpha * v.z;
        o.w = alpha * (((v.w - mu.w) / sd.w) * ss.w + sm.w) + one_minus_alpha * v.w;
        store4<T>(y + 4 * i, o);
    }
}

// calc_mean_std + AdaIN in ONE launch (net.py:240-246 inside field_retrieval; SURVEY S1 + S2): grid (C/64, B) and the very sums of
// channel_mean_std_kernel in the same order — mean / std come out bit-identical to the two-launch form and are still written for
// the distance head — then the block walks its pixels a second time (its 64 channels of one image: L2-resident) and applies the
// expression of adain_apply_kernel.
template <typename T>
__global__ __launch_bounds__(256) void adain_fused_kernel(const T* __restrict__ x, float* __restrict__ mean, float* __restrict__ stdv,
                                                          const float* __restrict__ smean, const float* __restrict__ sstd, long sstride,
                                                          float alpha, float one_minus_alpha, T* __restrict__ y, int hw, int c, float eps) {
#pragma clang fp contract(off)
    __shared__ double s_sum[16][64];
    __shared__ double s_sq[16][64];
    __shared__ __attribute__((aligned(16))) float s_mu[64], s_sd[64];
    const int tid = threadIdx.x;
    const int cq = tid & 15, pg = tid >> 4;
    const int c0 = blockIdx.x * 64;
    const int b = blockIdx.y;
    const T* __restrict__ xb = x + (size_t)b * hw * c + c0 + 4 * cq;
    T* __restrict__ yb = y + (size_t)b * hw * c + c0 + 4 * cq;
    double s[4] = {0, 0, 0, 0}, q[4] = {0, 0, 0, 0};
    for (int p = pg; p < hw; p += 16) {
        const float4 v = load4<T>(xb + (size_t)p * c);
        s[0] += v.x; q[0] += (double)v.x * v.x;
        s[1] += v.y; q[1] += (double)v.y * v.y;
        s[2] += v.z; q[2] += (double)v.z * v.z;
        s[3] += v.w; q[3] += (double)v.w * v.w;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        s_sum[pg][4 * cq + j] = s[j];
        s_sq[pg][4 * cq + j] = q[j];
    }
    __syncthreads();
    if (tid < 64) {
        double ts = 0, tq = 0;
#pragma unroll
        for (int g = 0; g < 16; ++g) {
            ts += s_sum[g][tid];
            tq += s_sq[g][tid];
        }
        const double n = (double)hw;
        const double mu = ts / n;
        double var = (tq - ts * mu) / (n - 1.0);
        var = var > 0 ? var : 0;
        const float mf = (float)mu, sf = sqrtf((float)var + eps);
        mean[(size_t)b * c + c0 + tid] = mf;
        stdv[(size_t)b * c + c0 + tid] = sf;
        s_mu[tid] = mf;
        s_sd[tid] = sf;
    }
    __syncthreads();
    const float4 mu = *reinterpret_cast<const float4*>(&s_mu[4 * cq]);
    const float4 sd = *reinterpret_cast<const float4*>(&s_sd[4 * cq]);
    const float4 sm = *reinterpret_cast<const float4*>(smean + b * sstride + c0 + 4 * cq);
    const float4 ss = *reinterpret_cast<const float4*>(sstd + b * sstride + c0 + 4 * cq);
    for (int p = pg; p < hw; p += 16) {
        const float4 v = load4<T>(xb + (size_t)p * c);
        float4 o;
        o.x = alpha * (((v.x - mu.x) / sd.x) * ss.x + sm.x) + one_minus_alpha * v.x;
        o.y = alpha * (((v.y - mu.y) / sd.y) * ss.y + sm.y) + one_minus_alpha * v.y;
        o.z = alpha * (((v.z - mu.z) / sd.z) * ss.z + sm.z) + one_minus_alpha * v.z;
        o.w = alpha * (((v.w - mu.w) / sd.w) * ss.w + sm.w) + one_minus_alpha * v.w;
        store4<T>(yb + (size_t)p * c, o);
    }
}

template <typename T>
static int adain_fused_entry(const T* x_nhwc, float* mean, float* std, const float* style_mean, const float* style_std,
                             long style_batch_stride, float alpha, T* y_nhwc, int batch, int hw, int c, float eps, void* stream) {
    HOLO_REQUIRE(x_nhwc && mean && std && style_mean && style_std && y_nhwc, HOLO_ERR_ARG, "holo_adain_fused: null pointer");
    HOLO_REQUIRE(batch > 0 && batch <= 65535 && hw >= 1 && c > 0 && c % 64 == 0 && style_batch_stride % 4 == 0, HOLO_ERR_ARG,
                 "holo_adain_fused: batch=%d hw=%d c=%d style_stride=%ld (c must be a multiple of 64, batch <= 65535)", batch, hw, c, style_batch_stride);
    HOLO_REQUIRE(alpha >= 0.0f && alpha <= 1.0f, HOLO_ERR_ARG, "holo_adain_fused: alpha=%f outside [0,1]", alpha);
    const float one_minus = (float)(1.0 - (double)alpha);      // net.py:246: `(1 - alpha)` in Python double precision
    hipLaunchKernelGGL((adain_fused_kernel<T>), dim3(c / 64, batch), dim3(256), 0, as_stream(stream), x_nhwc, mean, std, style_mean,
                       style_std, style_batch_stride, alpha, one_minus, y_nhwc, hw, c, eps);
    return check_launch("holo_adain_fused");
}

template <typename T>
static int mean_std_entry(const T* x_nhwc, float* mean, float* std, int batch, int hw, int c, float eps, void* stream) {
    HOLO_REQUIRE(x_nhwc && mean && std, HOLO_ERR_ARG, "holo_channel_mean_std: null pointer");
    HOLO_REQUIRE(batch > 0 && batch <= 65535 && hw >= 1 && c > 0 && c % 64 == 0, HOLO_ERR_ARG,
                 "holo_channel_mean_std: batch=%d hw=%d c=%d (c must be a multiple of 64, batch <= 65535)", batch, hw, c);
    hipLaunchKernelGGL((channel_mean_std_kernel<T>), dim3(c / 64, batch), dim3(256), 0, as_stream(stream), x_nhwc, mean, std,
                       hw, c, eps);
    return check_launch("holo_channel_mean_std");
}

template <typename T>
static int adain_entry(const T* x_nhwc, const float* mean, const float* std, const float* style_mean,
                       const float* style_std, long style_batch_stride, float alpha, T* y_nhwc, int batch, int hw, int c,
                       void* stream) {
    HOLO_REQUIRE(x_nhwc && mean && std && style_mean && style_std && y_nhwc, HOLO_ERR_ARG, "holo_adain_apply: null pointer");
    HOLO_REQUIRE(batch > 0 && hw >= 1 && c > 0 && c % 4 == 0 && style_batch_stride % 4 == 0, HOLO_ERR_ARG,
                 "holo_adain_apply: batch=%d hw=%d c=%d style_stride=%ld", batch, hw, c, style_batch_stride);
    HOLO_REQUIRE(alpha >= 0.0f && alpha <= 1.0f, HOLO_ERR_ARG, "holo_adain_apply: alpha=%f outside [0,1]", alpha);
    const long total4 = (long)batch * hw * (c / 4);
    long blocks = (total4 + 255) / 256;
    if (blocks > 256 * 8) blocks = 256 * 8;
    // net.py:246: `(1 - alpha)` is evaluated in Python double precision, then meets the fp32 tensor
    const float one_minus = (float)(1.0 - (double)alpha);
    hipLaunchKernelGGL((adain_apply_kernel<T>), dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), x_nhwc, mean, std,
                       style_mean, style_std, style_batch_stride, alpha, one_minus, y_nhwc, total4, hw, c);
    return check_launch("holo_adain_apply");
}

}  // namespace holo

extern "C" {

int holo_channel_mean_std_f32(const float* x_nhwc, float* mean, float* std, int batch, int hw, int c, float eps, void* stream) {
    return holo::mean_std_entry<float>(x_nhwc, mean, std, batch, hw, c, eps, stream);
}
int holo_channel_mean_std_bf16(const uint16_t* x_nhwc, float* mean, float* std, int batch, int hw, int c, float eps, void* stream) {
    return holo::mean_std_entry<holo::bf16_t>(reinterpret_cast<const holo::bf16_t*>(x_nhwc), mean, std, batch, hw, c, eps, stream);
}
int holo_channel_mean_std_f32s(const void* x_nhwc, float* mean, float* std, int batch, int hw, int c, float eps, void* stream) {
    return holo::mean_std_entry<holo::split16>(static_cast<const holo::split16*>(x_nhwc), mean, std, batch, hw, c, eps, stream);
}
int holo_adain_apply_f32s(const void* x_nhwc, const float* mean, const float* std, const float* style_mean,
                          const float* style_std, long style_batch_stride, float alpha, void* y_nhwc, int batch, int hw,
                          int c, void* stream) {
    return holo::adain_entry<holo::split16>(static_cast<const holo::split16*>(x_nhwc), mean, std, style_mean, style_std,
                                            style_batch_stride, alpha, static_cast<holo::split16*>(y_nhwc), batch, hw, c, stream);
}
int holo_adain_apply_f32(const float* x_nhwc, const float* mean, const float* std, const float* style_mean,
                         const float* style_std, long style_batch_stride, float alpha, float* y_nhwc, int batch, int hw,
                         int c, void* stream) {
    return holo::adain_entry<float>(x_nhwc, mean, std, style_mean, style_std, style_batch_stride, alpha, y_nhwc, batch, hw, c, stream);
}
int holo_adain_apply_bf16(const uint16_t* x_nhwc, const float* mean, const float* std, const float* style_mean,
                          const float* style_std, long style_batch_stride, float alpha, uint16_t* y_nhwc, int batch, int hw,
                          int c, void* stream) {
    using holo::bf16_t;
    return holo::adain_entry<bf16_t>(reinterpret_cast<const bf16_t*>(x_nhwc), mean, std, style_mean, style_std,
                                     style_batch_stride, alpha, reinterpret_cast<bf16_t*>(y_nhwc), batch, hw, c, stream);
}

/* calc_mean_std + AdaIN in one launch: mean / std (B,c) written as by holo_channel_mean_std_*, y as by holo_adain_apply_* (bit for bit) */
int holo_adain_fused_f32(const float* x_nhwc, float* mean, float* std, const float* style_mean, const float* style_std,
                         long style_batch_stride, float alpha, float* y_nhwc, int batch, int hw, int c, float eps, void* stream) {
    return holo::adain_fused_entry<float>(x_nhwc, mean, std, style_mean, style_std, style_batch_stride, alpha, y_nhwc, batch, hw, c, eps, stream);
}
int holo_adain_fused_bf16(const uint16_t* x_nhwc, float* mean, float* std, const float* style_mean, const float* style_std,
                          long style_batch_stride, float alpha, uint16_t* y_nhwc, int batch, int hw, int c, float eps, void* stream) {
    using holo::bf16_t;
    return holo::adain_fused_entry<bf16_t>(reinterpret_cast<const bf16_t*>(x_nhwc), mean, std, style_mean, style_std, style_batch_stride,
                                           alpha, reinterpret_cast<bf16_t*>(y_nhwc), batch, hw, c, eps, stream);
}
int holo_adain_fused_f32s(const void* x_nhwc, float* mean, float* std, const float* style_mean, const float* style_std,
                          long style_batch_stride, float alpha, void* y_nhwc, int batch, int hw, int c, float eps, void* stream) {
    return holo::adain_fused_entry<holo::split16>(static_cast<const holo::split16*>(x_nhwc), mean, std, style_mean, style_std,
                                                  style_batch_stride, alpha, static_cast<holo::split16*>(y_nhwc), batch, hw, c, eps, stream);
}

}  // extern "C"
