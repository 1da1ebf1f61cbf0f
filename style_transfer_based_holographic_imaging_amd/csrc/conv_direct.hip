// The two skinny layers (K = 27 and N = 2).
//
//   holo_conv_first_f32 : vgg[0:4]  Conv2d(1,3,1x1) -> ReflectionPad2d(1) -> Conv2d(3,64,3x3) -> ReLU
//                         (net.py:77-80), single-channel planes in, NHWC 64-channel out.  Output-write bound
//                         (fp32 MFMA with K = 27).
//   holo_conv_last_f32  : decoder[36:38]  ReflectionPad2d(1) -> Conv2d(64,2,3x3), no activation (net.py:72-73),
//                         NHWC in, NCHW amplitude/phase planes out.  Input-read bound.
#include "common.h"

namespace holo {

typedef __bf16 bf16_t;
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// First layer on the matrix cores.  K = 27 (3 channels x 9 taps, padded to 28) is tiny, but as a direct VALU kernel the
// layer costs 108 FMAs per lane and pixel and was compute-bound (0.48 ms at batch 64 x 256^2, 7.9 ms at 256 x 512^2);
// as 14 steps of v_mfma_f32_32x32x2_f32 per 32 px x 32 co tile it is bound by its 64-channel output write.
// Block = 16x16 output pixels, 4 waves; a wave owns 4 rows = two 32-pixel M tiles (m = 16*(row&1) + x) x 64 co.
//   A[m][k]  = w0[c]*in(pixel + tap) + b0[c], k = 9c + tap: vgg[0] applied on the fly to the reflect-padded halo in LDS
//   B[k][co] = w1[co][k] (OIHW is already [co][27]); k = 27 is a zero column
// fp32 accumulation for both output types; the tile goes through LDS once more so that every pixel's 64 channels
// leave as whole 16-byte units of one contiguous NHWC run.
typedef float floatx16 __attribute__((ext_vector_type(16)));
constexpr int kFT = 16;            // tile side
constexpr int kFH = kFT + 2;       // halo side
constexpr int kFK = 28;            // padded K
constexpr int kFWS = 29;           // LDS row stride of the weights (odd: conflict-free column reads)

template <typename T>
__global__ __launch_bounds__(256) void conv_first_kernel(const float* __restrict__ x, const float* __restrict__ w0,
                                                         const float* __restrict__ b0, const float* __restrict__ w1,
                                                         const float* __restrict__ b1, T* __restrict__ y, int H, int W,
                                                         int tiles_x, int tiles_y, int sqrt_in) {
    __shared__ float halo[kFH * kFH];
    __shared__ float wl[64 * kFWS];
    __shared__ __attribute__((aligned(16))) T stage[4][32 * 64];   // per wave: one M tile as [pixel][co]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int bid = blockIdx.x;
    const int tx = bid % tiles_x;
    bid /= tiles_x;
    const int ty = bid % tiles_y;
    const int img = bid / tiles_y;
    const int x0 = tx * kFT, y0 = ty * kFT;
    const float* __restrict__ xi = x + (size_t)img * H * W;
    for (int i = tid; i < kFH * kFH; i += 256) {
        const int hy = i / kFH, hx = i % kFH;
        const float v = xi[reflect1(y0 - 1 + hy, H) * W + reflect1(x0 - 1 + hx, W)];
        halo[i] = sqrt_in ? sqrtf(v) : v;     // intensity in (sqrtf is correctly rounded here, __fsqrt_rn is not): torch.sqrt(content_holo) of the entry loop (test_field_retrieval_mnist.py:119) in the loader
    }
    for (int i = tid; i < 64 * kFK; i += 256) {
        const int co = i / kFK, k = i % kFK;
        wl[co * kFWS + k] = k < 27 ? w1[co * 27 + k] : 0.0f;
    }
    __syncthreads();

    // this lane's K column per MFMA step: k = 2j + (lane>>5)
    const int hsel = lane >> 5, r = lane & 31;
    int koff[kFK / 2];
    float kw[kFK / 2], kb[kFK / 2];
#pragma unroll
    for (int j = 0; j < kFK / 2; ++j) {
        int k = 2 * j + hsel;
        k = k < 27 ? k : 0;                   // the padded column multiplies a zero weight: any finite value does
        const int c = k / 9, tap = k % 9;
        koff[j] = (tap / 3) * kFH + tap % 3;
        kw[j] = w0[c];
        kb[j] = b0[c];
    }
    const int lx = r & 15, ly = r >> 4;
    float bias[2];
    bias[0] = b1[r], bias[1] = b1[32 + r];
    T* __restrict__ yimg = y + (size_t)img * H * W * 64;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        const int row = 4 * wave + 2 * mt;       // first of the tile's two rows
        floatx16 acc[2];
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[nt][q] = 0.0f;
        const int hbase = (row + ly) * kFH + lx;
#pragma unroll
        for (int j = 0; j < kFK / 2; ++j) {
            const float a = halo[hbase + koff[j]] * kw[j] + kb[j];
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                const float b = wl[(nt * 32 + r) * kFWS + 2 * j + hsel];
                acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[nt], 0, 0, 0);
            }
        }
        // C layout: pixel m = (q&3) + 8*(q>>2) + 4*hsel, channel n = r (+32 for the second N tile)
        T* st = stage[wave];
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int m = (q & 3) + 8 * (q >> 2) + 4 * hsel;
                const float val = fmaxf(acc[nt][q] + bias[nt], 0.0f);
                if constexpr (__is_same(T, split16)) {
                    // pixel m's 64 channels = 4 groups of [16 x hi | 16 x lo']
                    const int c = nt * 32 + r;
                    _Float16* hp = reinterpret_cast<_Float16*>(st + m * 64) + (c / 16) * 32 + (c % 16);
                    split_halves(val, hp, hp + 16);
                } else {
                    st[m * 64 + nt * 32 + r] = (T)val;
                }
            }
        // stage[wave] is private to this wave and the LDS unit executes one wave's DS instructions in order: a wave-local
        // fence orders the transposing reads behind the writes — no block barrier, the four waves run their tiles
        // independently (round 2 had two __syncthreads() per M tile here)
        wave_lds_fence();
        // 32 pixels x 64 channels = 32 * 64 * sizeof(T) / 16 units of 16 bytes; a lane takes every 64th
        constexpr int UPP = 64 * (int)sizeof(T) / 16;    // units per pixel (16 fp32 / 8 bf16)
#pragma unroll
        for (int u = lane; u < 32 * UPP; u += 64) {
            const int m = u / UPP, c = u % UPP;
            const int X = x0 + (m & 15), Y = y0 + row + (m >> 4);
            if (X < W && Y < H)
                *reinterpret_cast<float4*>(reinterpret_cast<unsigned char*>(yimg + ((size_t)Y * W + X) * 64) + 16 * c) =
                    *reinterpret_cast<const float4*>(reinterpret_cast<const unsigned char*>(st + m * 64) + 16 * c);
        }
        wave_lds_fence();   // the next M tile overwrites the stage
    }
}

// 16x16 output pixels per block, 64 input channels staged in four 16-channel chunks (18x18 halo each).
// The 2 x 64 x 9 weights are block-uniform: they are read straight from the OIHW tensor through the SCALAR cache
// (s_load -> SGPR operands of the FMAs), 72 at a time (4 channels x 9 taps x 2 outputs).  Round 2 kept them in LDS: every
// lane then fetched the same 16 bytes with a ds_read_b128, and the LDS return path moves a full 1 KiB per wave for a
// broadcast too — 288 of a thread's 432 LDS reads, which made the layer LDS-bound (0.35 ms at batch 64 x 256^2 for a
// 1.07 GB read) instead of read-bound.
constexpr int kLT = 16;
constexpr int kLH = kLT + 2;

// EARLY (fp32 only): the halo of all four chunks is requested before the first chunk is used — one memory latency per block instead
// of four, and the two halves of every 128-byte line are asked for together (with chunk-by-chunk loads every line was fetched
// twice, a chunk apart).  0.340 -> 0.246 ms at 64 x 256^2 (3.3 -> 4.5 TB/s), 25 -> 21 us at 5 x 128^2.  Same arithmetic, same order.
template <typename T, bool EARLY = false>
__global__ __launch_bounds__(256) void conv_last_kernel(const T* __restrict__ x, const float* __restrict__ w,
                                                        const float* __restrict__ bias, float* __restrict__ y, int H,
                                                        int W, int tiles_x, int tiles_y) {
    __shared__ __attribute__((aligned(16))) float halo[kLH * kLH * 16];
    const int tid = threadIdx.x;
    int bid = blockIdx.x;
    const int tx = bid % tiles_x;
    bid /= tiles_x;
    const int ty = bid % tiles_y;
    const int img = bid / tiles_y;
    const int x0 = tx * kLT, y0 = ty * kLT;
    const T* __restrict__ xi = x + (size_t)img * H * W * 64;
    const int lx = tid & 15, ly = tid >> 4;
    float acc0 = bias[0], acc1 = bias[1];
    constexpr int kRounds = (kLH * kLH * 4 + 255) / 256;     // 6 (the last one partial)
    float4 early[EARLY ? 4 : 1][EARLY ? kRounds : 1];      // [0] is the chunk about to be staged; the sets shift down after each chunk
    if constexpr (EARLY) {
#pragma unroll
        for (int r = 0; r < kRounds; ++r) {
            const int u = tid + 256 * r;
            const int q = (u < kLH * kLH * 4 ? u : 0) >> 2, c = u & 3;
            const int hy = q / kLH, hx = q % kLH;
            const int ry = reflect1(y0 - 1 + hy, H), rx = reflect1(x0 - 1 + hx, W);
            const T* src = xi + ((size_t)ry * W + rx) * 64 + 4 * c;
#pragma unroll
            for (int chunk = 0; chunk < 4; ++chunk) early[chunk][r] = *reinterpret_cast<const float4*>(src + chunk * 16);
        }
    }
    for (int chunk = 0; chunk < 4; ++chunk) {
        __syncthreads();
        if constexpr (EARLY) {
#pragma unroll
            for (int r = 0; r < kRounds; ++r) {
                const int u = tid + 256 * r;
                const int q = u >> 2, c = u & 3;
                if (u < kLH * kLH * 4) *reinterpret_cast<float4*>(&halo[q * 16 + ((c ^ ((q >> 2) & 3)) << 2)]) = early[0][r];
                early[0][r] = early[1][r];
                early[1][r] = early[2][r];
                early[2][r] = early[3][r];
            }
        } else if constexpr (__is_same(T, float)) {
            for (int u = tid; u < kLH * kLH * 4; u += 256) {
                const int q = u >> 2, c = u & 3;
                const int hy = q / kLH, hx = q % kLH;
                const int ry = reflect1(y0 - 1 + hy, H), rx = reflect1(x0 - 1 + hx, W);
                const float4 v = *reinterpret_cast<const float4*>(xi + ((size_t)ry * W + rx) * 64 + chunk * 16 + 4 * c);
                *reinterpret_cast<float4*>(&halo[q * 16 + ((c ^ ((q >> 2) & 3)) << 2)]) = v;
            }
        } else if constexpr (__is_same(T, split16)) {
            // chunk = one 16-channel group: 64 bytes per pixel = hi units 0,1 | lo' units 2,3; a thread turns hi unit c8 and
            // lo' unit 2 + c8 (8 channels) into two fp32 units of the LDS row
            for (int u = tid; u < kLH * kLH * 2; u += 256) {
                const int q = u >> 1, c8 = u & 1;
                const int hy = q / kLH, hx = q % kLH;
                const int ry = reflect1(y0 - 1 + hy, H), rx = reflect1(x0 - 1 + hx, W);
                const unsigned char* gp = reinterpret_cast<const unsigned char*>(xi + ((size_t)ry * W + rx) * 64 + chunk * 16);
                typedef _Float16 halfx8 __attribute__((ext_vector_type(8)));
                const halfx8 hi = *reinterpret_cast<const halfx8*>(gp + 16 * c8);
                const halfx8 lo = *reinterpret_cast<const halfx8*>(gp + 32 + 16 * c8);
                float f[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) f[e] = (float)hi[e] + (float)lo[e] * 0.00048828125f;
                const int sw = (q >> 2) & 3;
                *reinterpret_cast<float4*>(&halo[q * 16 + (((2 * c8) ^ sw) << 2)]) = make_float4(f[0], f[1], f[2], f[3]);
                *reinterpret_cast<float4*>(&halo[q * 16 + (((2 * c8 + 1) ^ sw) << 2)]) = make_float4(f[4], f[5], f[6], f[7]);
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
        for (int c = 0; c < 4; ++c) {
            // OIHW (2,64,3,3): the 9 taps of (co, ci) are contiguous; ci = chunk*16 + 4c + k.  Uniform addresses: scalar loads.
            const float* __restrict__ w0p = w + (size_t)(chunk * 16 + 4 * c) * 9;
            const float* __restrict__ w1p = w0p + 64 * 9;
            float4 v[9];
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int q = (ly + tap / 3) * kLH + lx + tap % 3;
                v[tap] = *reinterpret_cast<const float4*>(&halo[q * 16 + ((c ^ ((q >> 2) & 3)) << 2)]);
            }
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                acc0 += v[tap].x * w0p[tap];      acc1 += v[tap].x * w1p[tap];
                acc0 += v[tap].y * w0p[9 + tap];  acc1 += v[tap].y * w1p[9 + tap];
                acc0 += v[tap].z * w0p[18 + tap]; acc1 += v[tap].z * w1p[18 + tap];
                acc0 += v[tap].w * w0p[27 + tap]; acc1 += v[tap].w * w1p[27 + tap];
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

// conv_bf16.hip: the first layer as one split-bf16 MFMA step for bf16 activations; -1 when it does not take the problem
int conv_first_b16_try(const float* x, const float* w0, const float* b0, const float* w1, const float* b1, void* y, int batch, int h,
                       int w_px, int sqrt_in, hipStream_t stream);

template <typename T>
static int conv_first_entry(const float* x, const float* w0, const float* b0, const float* w1, const float* b1, T* y_nhwc,
                            int batch, int h, int w, void* stream, int sqrt_in = 0) {
    HOLO_REQUIRE(x && w0 && b0 && w1 && b1 && y_nhwc, HOLO_ERR_ARG, "holo_conv_first: null pointer");
    HOLO_REQUIRE(batch > 0 && h >= 2 && w >= 2, HOLO_ERR_ARG,
                 "holo_conv_first: batch=%d h=%d w=%d (reflection padding needs h,w >= 2)", batch, h, w);
    const int tiles_x = ceil_div(w, kFT), tiles_y = ceil_div(h, kFT);
    const long blocks = (long)batch * tiles_x * tiles_y;
    HOLO_REQUIRE(blocks < 2147483647L, HOLO_ERR_UNSUPPORTED, "holo_conv_first: grid of %ld blocks", blocks);
    if constexpr (__is_same(T, bf16_t)) {
        if (conv_tile_override() != 1) {     // holo_debug_set_conv_tile(1) keeps the fp32-MFMA kernel (A/B, tests)
            const int r = conv_first_b16_try(x, w0, b0, w1, b1, y_nhwc, batch, h, w, sqrt_in, as_stream(stream));
            if (r >= 0) return r;
        }
    }
    hipLaunchKernelGGL((conv_first_kernel<T>), dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), x, w0, b0, w1, b1,
                       y_nhwc, h, w, tiles_x, tiles_y, sqrt_in);
    return check_launch("holo_conv_first");
}

// conv_bf16.hip: the matrix-core form of the last layer for bf16 activations; -1 when it does not take the problem
int conv_last_b16_try(const void* x, const float* w, const float* bias, float* y, int batch, int h, int w_px, hipStream_t stream);

template <typename T>
static int conv_last_entry(const T* x_nhwc, const float* w_oihw, const float* bias, float* y_nchw, int batch, int h, int w,
                           void* stream) {
    HOLO_REQUIRE(x_nhwc && w_oihw && bias && y_nchw, HOLO_ERR_ARG, "holo_conv_last: null pointer");
    HOLO_REQUIRE(batch > 0 && h >= 2 && w >= 2, HOLO_ERR_ARG,
                 "holo_conv_last: batch=%d h=%d w=%d (reflection padding needs h,w >= 2)", batch, h, w);
    const int tiles_x = ceil_div(w, kLT), tiles_y = ceil_div(h, kLT);
    const long blocks = (long)batch * tiles_x * tiles_y;
    HOLO_REQUIRE(blocks < 2147483647L, HOLO_ERR_UNSUPPORTED, "holo_conv_last: grid of %ld blocks", blocks);
    if constexpr (__is_same(T, bf16_t)) {
        if (conv_tile_override() != 1) {     // holo_debug_set_conv_tile(1) keeps the VALU kernel (A/B, tests)
            const int r = conv_last_b16_try(x_nhwc, w_oihw, bias, y_nchw, batch, h, w, as_stream(stream));
            if (r >= 0) return r;
        }
    }
    if constexpr (__is_same(T, float)) {
        if (conv_tile_override() != 1) {      // holo_debug_set_conv_tile(1): the chunk-by-chunk loads (A/B, tests); same bits
            hipLaunchKernelGGL((conv_last_kernel<T, true>), dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), x_nhwc, w_oihw, bias,
                               y_nchw, h, w, tiles_x, tiles_y);
            return check_launch("holo_conv_last");
        }
    }
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
/* the same layers fed with the hologram INTENSITY: sqrt (correctly rounded, = torch.sqrt) applied by the loader
 * (test_field_retrieval_mnist.py:119 folded into vgg[0:4]) */
int holo_conv_first_sqrt_f32(const float* x, const float* w0, const float* b0, const float* w1, const float* b1, float* y_nhwc, int batch,
                             int h, int w, void* stream) {
    return holo::conv_first_entry<float>(x, w0, b0, w1, b1, y_nhwc, batch, h, w, stream, 1);
}
int holo_conv_first_sqrt_bf16(const float* x, const float* w0, const float* b0, const float* w1, const float* b1, uint16_t* y_nhwc,
                              int batch, int h, int w, void* stream) {
    return holo::conv_first_entry<holo::bf16_t>(x, w0, b0, w1, b1, reinterpret_cast<holo::bf16_t*>(y_nhwc), batch, h, w, stream, 1);
}
int holo_conv_first_sqrt_f32s(const float* x, const float* w0, const float* b0, const float* w1, const float* b1, void* y_nhwc, int batch,
                              int h, int w, void* stream) {
    return holo::conv_first_entry<holo::split16>(x, w0, b0, w1, b1, static_cast<holo::split16*>(y_nhwc), batch, h, w, stream, 1);
}
int holo_conv_first_bf16(const float* x, const float* w0, const float* b0, const float* w1, const float* b1,
                         uint16_t* y_nhwc, int batch, int h, int w, void* stream) {
    return holo::conv_first_entry<holo::bf16_t>(x, w0, b0, w1, b1, reinterpret_cast<holo::bf16_t*>(y_nhwc), batch, h, w, stream);
}
int holo_conv_first_f32s(const float* x, const float* w0, const float* b0, const float* w1, const float* b1, void* y_nhwc,
                         int batch, int h, int w, void* stream) {
    return holo::conv_first_entry<holo::split16>(x, w0, b0, w1, b1, static_cast<holo::split16*>(y_nhwc), batch, h, w, stream);
}
int holo_conv_last_f32s(const void* x_nhwc, const float* w_oihw, const float* bias, float* y_nchw, int batch, int h, int w,
                        void* stream) {
    return holo::conv_last_entry<holo::split16>(static_cast<const holo::split16*>(x_nhwc), w_oihw, bias, y_nchw, batch, h, w, stream);
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
