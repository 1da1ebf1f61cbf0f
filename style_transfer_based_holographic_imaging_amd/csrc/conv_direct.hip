// The two skinny layers that cannot feed the matrix cores (K = 27 and N = 2): direct fp32 VALU kernels.
//
//   holo_conv_first_f32 : vgg[0:4]  Conv2d(1,3,1x1) -> ReflectionPad2d(1) -> Conv2d(3,64,3x3) -> ReLU
//                         (net.py:77-80), single-channel planes in, NHWC 64-channel out.  Output-write bound.
//   holo_conv_last_f32  : decoder[36:38]  ReflectionPad2d(1) -> Conv2d(64,2,3x3), no activation (net.py:72-73),
//                         NHWC in, NCHW amplitude/phase planes out.  Input-read bound.
#include "common.h"

namespace holo {

typedef __bf16 bf16_t;
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// 16 lanes share a pixel (4 output channels each) so one store instruction writes whole 256-B pixels.
// Each thread keeps its 27x4 weights in registers and walks pixels with a grid stride.
template <typename T>
__global__ __launch_bounds__(256) void conv_first_kernel(const float* __restrict__ x, const float* __restrict__ w0,
                                                         const float* __restrict__ b0, const float* __restrict__ w1,
                                                         const float* __restrict__ b1, T* __restrict__ y,
                                                         long total_px, int H, int W) {
    const int tid = threadIdx.x;
    const int cq = tid & 15;   // output channels 4*cq .. 4*cq+3
    const int pl = tid >> 4;   // pixel slot within the block (16 per block)
    float wreg[27][4];
#pragma unroll
    for (int k = 0; k < 27; ++k)
#pragma unroll
        for (int j = 0; j < 4; ++j) wreg[k][j] = w1[(4 * cq + j) * 27 + k];  // OIHW (64,3,3,3): k = c*9 + ky*3 + kx
    float breg[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) breg[j] = b1[4 * cq + j];
    const float w00 = w0[0], w01 = w0[1], w02 = w0[2];
    const float b00 = b0[0], b01 = b0[1], b02 = b0[2];
    const long hw = (long)H * W;
    for (long p = (long)blockIdx.x * 16 + pl; p < total_px; p += (long)gridDim.x * 16) {
        const long img = p / hw;
        const int rem = (int)(p % hw);
        const int py = rem / W, px = rem % W;
        const float* __restrict__ xi = x + img * hw;
        float acc[4] = {breg[0], breg[1], breg[2], breg[3]};
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int ry = reflect1(py + ky - 1, H);
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int rx = reflect1(px + kx - 1, W);
                const float v = xi[ry * W + rx];
                // vgg[0]: per-pixel affine 1 -> 3 (reflection padding commutes with it)
                const float t0 = v * w00 + b00, t1 = v * w01 + b01, t2 = v * w02 + b02;
                const int k = ky * 3 + kx;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    acc[j] += t0 * wreg[k][j];
                    acc[j] += t1 * wreg[9 + k][j];
                    acc[j] += t2 * wreg[18 + k][j];
                }
            }
        }
        if constexpr (sizeof(T) == 4) {
            float4 o;
            o.x = fmaxf(acc[0], 0.0f);
            o.y = fmaxf(acc[1], 0.0f);
            o.z = fmaxf(acc[2], 0.0f);
            o.w = fmaxf(acc[3], 0.0f);
            *reinterpret_cast<float4*>(y + p * 64 + 4 * cq) = o;
        } else {
            const bf16x4 o = {(bf16_t)fmaxf(acc[0], 0.0f), (bf16_t)fmaxf(acc[1], 0.0f), (bf16_t)fmaxf(acc[2], 0.0f),
                              (bf16_t)fmaxf(acc[3], 0.0f)};
            *reinterpret_cast<bf16x4*>(y + p * 64 + 4 * cq) = o;
        }
    }
}

// 16x16 output pixels per block, 64 input channels staged in four 16-channel chunks (18x18 halo each).
constexpr int kLT = 16;
constexpr int kLH = kLT + 2;

template <typename T>
__global__ __launch_bounds__(256) void conv_last_kernel(const T* __restrict__ x, const float* __restrict__ w,
                                                        const float* __restrict__ bias, float* __restrict__ y, int H,
                                                        int W, int tiles_x, int tiles_y) {
    __shared__ __attribute__((aligned(16))) float halo[kLH * kLH * 16];
    __shared__ float wl[9 * 64 * 2];  // [tap][ci][co]
    const int tid = threadIdx.x;
    int bid = blockIdx.x;
    const int tx = bid % tiles_x;
    bid /= tiles_x;
    const int ty = bid % tiles_y;
    const int img = bid / tiles_y;
    const int x0 = tx * kLT, y0 = ty * kLT;
    const T* __restrict__ xi = x + (size_t)img * H * W * 64;
    for (int i = tid; i < 9 * 64 * 2; i += 256) {
        const int co = i & 1, ci = (i >> 1) & 63, tap = i >> 7;
        wl[i] = w[(co * 64 + ci) * 9 + tap];  // OIHW (2,64,3,3)
    }
    const int lx = tid & 15, ly = tid >> 4;
    float acc0 = bias[0], acc1 = bias[1];
    for (int chunk = 0; chunk < 4; ++chunk) {
        __syncthreads();
        if constexpr (sizeof(T) == 4) {
            for (int u = tid; u < kLH * kLH * 4; u += 256) {
                const int q = u >> 2, c = u & 3;
                const int hy = q / kLH, hx = q % kLH;
                const int ry = reflect1(y0 - 1 + hy, H), rx = reflect1(x0 - 1 + hx, W);
                const float4 v = *reinterpret_cast<const float4*>(xi + ((size_t)ry * W + rx) * 64 + chunk * 16 + 4 * c);
                *reinterpret_cast<float4*>(&halo[q * 16 + ((c ^ ((q >> 2) & 3)) << 2)]) = v;
            }
        } else {
            // one 16-byte load = 8 bf16 channels = two fp32 units of the LDS row
            for (int u = tid; u < kLH * kLH * 2; u += 256) {
                const int q = u >> 1, c8 = u & 1;
                const int hy = q / kLH, hx = q % kLH;
                const int ry = reflect1(y0 - 1 + hy, H), rx = reflect1(x0 - 1 + hx, W);
                const bf16x8 v = *reinterpret_cast<const bf16x8*>(xi + ((size_t)ry * W + rx) * 64 + chunk * 16 + 8 * c8);
                const int sw = (q >> 2) & 3;
                *reinterpret_cast<float4*>(&halo[q * 16 + (((2 * c8) ^ sw) << 2)]) =
                    make_float4((float)v[0], (float)v[1], (float)v[2], (float)v[3]);
                *reinterpret_cast<float4*>(&halo[q * 16 + (((2 * c8 + 1) ^ sw) << 2)]) =
                    make_float4((float)v[4], (float)v[5], (float)v[6], (float)v[7]);
            }
        }
        __syncthreads();
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int q = (ly + tap / 3) * kLH + lx + tap % 3;
            const float* wt = &wl[(tap * 64 + chunk * 16) * 2];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float4 v = *reinterpret_cast<const float4*>(&halo[q * 16 + ((c ^ ((q >> 2) & 3)) << 2)]);
                const float* wc = wt + c * 8;
                acc0 += v.x * wc[0]; acc1 += v.x * wc[1];
                acc0 += v.y * wc[2]; acc1 += v.y * wc[3];
                acc0 += v.z * wc[4]; acc1 += v.z * wc[5];
                acc0 += v.w * wc[6]; acc1 += v.w * wc[7];
            }
        }
    }
    const int X = x0 + lx, Y = y0 + ly;
    if (X < W && Y < H) {
        float* __restrict__ yi = y + (size_t)img * 2 * H * W;
        yi[(size_t)Y * W + X] = acc0;
        yi[(size_t)H * W + (size_t)Y * W + X] = acc1;
    }
}

template <typename T>
static int conv_first_entry(const float* x, const float* w0, const float* b0, const float* w1, const float* b1, T* y_nhwc,
                            int batch, int h, int w, void* stream) {
    HOLO_REQUIRE(x && w0 && b0 && w1 && b1 && y_nhwc, HOLO_ERR_ARG, "holo_conv_first: null pointer");
    HOLO_REQUIRE(batch > 0 && h >= 2 && w >= 2, HOLO_ERR_ARG,
                 "holo_conv_first: batch=%d h=%d w=%d (reflection padding needs h,w >= 2)", batch, h, w);
    const long total = (long)batch * h * w;
    long blocks = (total + 15) / 16;
    if (blocks > 256 * 32) blocks = 256 * 32;  // grid-stride the rest
    hipLaunchKernelGGL((conv_first_kernel<T>), dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), x, w0, b0, w1, b1,
                       y_nhwc, total, h, w);
    return check_launch("holo_conv_first");
}

template <typename T>
static int conv_last_entry(const T* x_nhwc, const float* w_oihw, const float* bias, float* y_nchw, int batch, int h, int w,
                           void* stream) {
    HOLO_REQUIRE(x_nhwc && w_oihw && bias && y_nchw, HOLO_ERR_ARG, "holo_conv_last: null pointer");
    HOLO_REQUIRE(batch > 0 && h >= 2 && w >= 2, HOLO_ERR_ARG,
                 "holo_conv_last: batch=%d h=%d w=%d (reflection padding needs h,w >= 2)", batch, h, w);
    const int tiles_x = ceil_div(w, kLT), tiles_y = ceil_div(h, kLT);
    const long blocks = (long)batch * tiles_x * tiles_y;
    HOLO_REQUIRE(blocks < 2147483647L, HOLO_ERR_UNSUPPORTED, "holo_conv_last: grid of %ld blocks", blocks);
    hipLaunchKernelGGL((conv_last_kernel<T>), dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), x_nhwc, w_oihw, bias,
                       y_nchw, h, w, tiles_x, tiles_y);
    return check_launch("holo_conv_last");
}

}  // namespace holo

extern "C" {

int holo_conv_first_f32(const float* x, const float* w0, const float* b0, const float* w1, const float* b1,
                        float* y_nhwc, int batch, int h, int w, void* stream) {
    return holo::conv_first_entry<float>(x, w0, b0, w1, b1, y_nhwc, batch, h, w, stream);
}
int holo_conv_first_bf16(const float* x, const float* w0, const float* b0, const float* w1, const float* b1,
                         uint16_t* y_nhwc, int batch, int h, int w, void* stream) {
    return holo::conv_first_entry<holo::bf16_t>(x, w0, b0, w1, b1, reinterpret_cast<holo::bf16_t*>(y_nhwc), batch, h, w, stream);
}
int holo_conv_last_f32(const float* x_nhwc, const float* w_oihw, const float* bias, float* y_nchw, int batch, int h,
                       int w, void* stream) {
    return holo::conv_last_entry<float>(x_nhwc, w_oihw, bias, y_nchw, batch, h, w, stream);
}
int holo_conv_last_bf16(const uint16_t* x_nhwc, const float* w_oihw, const float* bias, float* y_nchw, int batch, int h,
                        int w, void* stream) {
    return holo::conv_last_entry<holo::bf16_t>(reinterpret_cast<const holo::bf16_t*>(x_nhwc), w_oihw, bias, y_nchw, batch, h, w, stream);
}

}  // extern "C"
