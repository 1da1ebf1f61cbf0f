// Angular-spectrum free-space propagation on gfx950.
//
// Replaces the aten op chain of utils/Angular_Spectrum_Method.py:7-36 (ASM) and the field packing /
// abs / angle / |.|^2 epilogues of utils/Forward_model.py:22-39 (Holo_Generator.forward).
//
// out = crop( ifft2( H . fft2( replicate_pad(O) ) ) ),  H[ky][kx] = exp(i*c*sqrt(max(0,1-l^2 f^2))/l)
//
// fftshift/ifftshift (even P) only relabel the bins, so H is evaluated at the signed frequency of
// the un-shifted bin and no shift pass exists.  The P x P complex grid does not fit a CU's LDS, so
// the 2-D transform is three HBM-bound kernels (80*S^2 algorithmic bytes per field):
//
//   K1 rows_fwd : amp,phase (8S^2 B) -> pack a*e^{ip}, replicate-pad the ROW to P, row FFT
//                 -> T[S][P] (16S^2 B).  The S/2 padded rows above/below the field are copies of
//                 its first/last row, so only S distinct row FFTs exist.
//   K2 cols     : T columns (16S^2 B) -> replicate-pad the COLUMN to P in LDS, column FFT,
//                 multiply by H generated on the fly (float64 phase argument, range-reduced, then
//                 fp32 sincos), column IFFT, keep the S centre rows -> U[S][P] (16S^2 B)
//   K3 rows_inv : U rows (16S^2 B) -> row IFFT, keep the S centre columns, scale 1/P^2,
//                 abs/angle | intensity | complex epilogue (8S^2 B)
//
// Transforms run in fp32 (complex64) with float64-rounded twiddles; the reference runs the product
// and the inverse transform in complex128 and casts to float at the end
// (Angular_Spectrum_Method.py:29-33, Forward_model.py:28-29) — measured difference ~2e-7 relative.
#include "common.h"
#include "fft_lds.h"

namespace holo {

constexpr int kThreads = 256;
constexpr int kRowPad = 16;  // complex elements of padding between sequences in the row kernels
constexpr int kColPad = 17;  // column pitch P+17: transposed b64 stores of 16 columns hit 16 bank pairs

__global__ void asm_twiddle_kernel(float2* tw, int p) {
    const int m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m < p) {
        double s, c;
        sincospi(-2.0 * (double)m / (double)p, &s, &c);
        tw[m] = make_float2((float)c, (float)s);
    }
}

__device__ __forceinline__ float2 load_field(const float* __restrict__ in0, const float* __restrict__ in1,
                                             int in_mode, long off0, long off1, float phase_scale) {
    if (in_mode == HOLO_ASM_IN_COMPLEX) {
        return reinterpret_cast<const float2*>(in0)[off0];
    }
    // Forward_model.py:20-22: phase*phase_normalize, then amplitude*exp(1j*phase) in complex64
    const float a = in0[off0];
    const float p = in1[off1] * phase_scale;
    float s, c;
    sincosf(p, &s, &c);
    return make_float2(a * c, a * s);
}

// ---- K1: pack + pad + row FFT ------------------------------------------------------------------
template <int P, bool PAD>
__global__ __launch_bounds__(kThreads) void asm_rows_fwd_kernel(
    const float* __restrict__ in0, const float* __restrict__ in1, int in_mode, long stride0, long stride1,
    float phase_scale, float2* __restrict__ tbuf, const float2* __restrict__ tw_g, long n_rows) {
    constexpr int S = PAD ? P / 2 : P;
    constexpr int G = P / 8;            // threads per sequence
    constexpr int SEQ = kThreads / G;   // sequences per block
    constexpr int PITCH = P + kRowPad;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float2* tw = reinterpret_cast<float2*>(smem_raw);
    float2* buf = tw + P;

    const int tid = threadIdx.x;
    const int seq = tid / G, t = tid % G;
    const long row = (long)blockIdx.x * SEQ + seq;
    const bool valid = row < n_rows;
    float2* my = buf + seq * PITCH;
    for (int i = tid; i < P; i += kThreads) tw[i] = tw_g[i];
    if (valid) {
        const long f = row / S;
        const int y = (int)(row % S);
        for (int x = t; x < S; x += G) {
            const long pix = (long)y * S + x;
            my[x + (PAD ? S / 2 : 0)] =
                load_field(in0, in1, in_mode, f * stride0 + pix, f * stride1 + pix, phase_scale);
        }
    }
    __syncthreads();
    if (PAD) {
        // F.pad(..., mode="replicate") along the row (Angular_Spectrum_Method.py:12)
        const float2 a = my[S / 2], b = my[S / 2 + S - 1];
        for (int c = t; c < S / 2; c += G) {
            my[c] = a;
            my[3 * S / 2 + c] = b;
        }
        __syncthreads();
    }
    fft_inplace<P, 1, G, -1>(my, tw, t);
    if (valid) {
        float2* dst = tbuf + row * P;
        for (int k = t; k < P; k += G) dst[k] = my[k];
    }
}

// ---- K2: column FFT . H . column IFFT -----------------------------------------------------------
// theta = c * sqrt(max(0, 1 - l^2 (fx^2+fy^2))) / l evaluated as numpy does
// (Angular_Spectrum_Method.py:16-23), in float64 with no contraction.
__device__ __forceinline__ float2 transfer_value(double fx2, double fy, double lam2, double lamb, double c) {
#pragma clang fp contract(off)
    const double sum = fx2 + fy * fy;
    const double g = 1.0 - lam2 * sum;
    const double gg = sqrt(g > 0.0 ? g : 0.0) / lamb;
    const double theta = c * gg;
    // reduce to [-pi, pi] in float64, then fp32 sincos
    constexpr double INV_2PI = 0.15915494309189533577;
    constexpr double TWO_PI_HI = 6.283185307179586232;     // double(2*pi)
    constexpr double TWO_PI_LO = 2.4492935982947064e-16;   // 2*pi - double(2*pi)
    const double n = rint(theta * INV_2PI);
    double r = __builtin_fma(-n, TWO_PI_HI, theta);
    r = __builtin_fma(-n, TWO_PI_LO, r);
    float s, co;
    sincosf((float)r, &s, &co);
    return make_float2(co, s);
}

template <int P, bool PAD, int CW>
__global__ __launch_bounds__(kThreads) void asm_cols_kernel(
    const float2* __restrict__ tbuf, float2* __restrict__ ubuf, const float2* __restrict__ tw_g,
    const double* __restrict__ phase_const, int n_dist, double ppx, double lam2, double lamb) {
    constexpr int S = PAD ? P / 2 : P;
    constexpr int OFF = PAD ? S / 2 : 0;
    constexpr int G = kThreads / CW;  // threads per column
    constexpr int PITCH = P + kColPad;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float2* tw = reinterpret_cast<float2*>(smem_raw);
    float2* cols = tw + P;

    const int tid = threadIdx.x;
    const int fo = blockIdx.y;          // output field = field * n_dist + dist
    const int f = fo / n_dist;
    const int kx0 = blockIdx.x * CW;
    for (int i = tid; i < P; i += kThreads) tw[i] = tw_g[i];
    const float2* src = tbuf + (long)f * S * P + kx0;
    for (int i = tid; i < S * CW; i += kThreads) {
        const int y = i / CW, cl = i % CW;
        cols[cl * PITCH + y + OFF] = src[(long)y * P + cl];
    }
    __syncthreads();
    if (PAD) {
        // replicate rows above/below the field: copies of its first/last row spectrum
        for (int i = tid; i < (S / 2) * CW; i += kThreads) {
            const int r = i / CW, cl = i % CW;
            cols[cl * PITCH + r] = cols[cl * PITCH + S / 2];
            cols[cl * PITCH + 3 * S / 2 + r] = cols[cl * PITCH + 3 * S / 2 - 1];
        }
        __syncthreads();
    }
    const int col = tid / G, t = tid % G;
    float2* my = cols + col * PITCH;
    fft_inplace<P, 1, G, -1>(my, tw, t);

    {
        const double c = phase_const[fo];
        const int kx = kx0 + col;
        const int kxs = kx < P / 2 ? kx : kx - P;   // signed frequency index of the un-shifted bin
        double fx2;
        {
#pragma clang fp contract(off)
            const double fx = (double)kxs / ppx;
            fx2 = fx * fx;
        }
        // rows ky and P-ky share fy^2: one transfer value serves both
        for (int m = t; m <= P / 2; m += G) {
            double fy;
            {
#pragma clang fp contract(off)
                fy = (double)m / ppx;
            }
            const float2 h = transfer_value(fx2, fy, lam2, lamb, c);
            my[m] = cmul(my[m], h);
            if (m != 0 && m != P / 2) my[P - m] = cmul(my[P - m], h);
        }
    }
    __syncthreads();
    fft_inplace<P, 1, G, +1>(my, tw, t);

    float2* dst = ubuf + (long)fo * S * P + kx0;
    for (int i = tid; i < S * CW; i += kThreads) {
        const int y = i / CW, cl = i % CW;
        dst[(long)y * P + cl] = cols[cl * PITCH + y + OFF];
    }
}

// ---- K3: row IFFT + crop + epilogue ---------------------------------------------------------------
template <int P, bool PAD>
__global__ __launch_bounds__(kThreads) void asm_rows_inv_kernel(
    const float2* __restrict__ ubuf, float* __restrict__ out0, float* __restrict__ out1, int out_mode,
    long ostride0, long ostride1, const float2* __restrict__ tw_g, long n_rows, float scale) {
    constexpr int S = PAD ? P / 2 : P;
    constexpr int G = P / 8;
    constexpr int SEQ = kThreads / G;
    constexpr int PITCH = P + kRowPad;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float2* tw = reinterpret_cast<float2*>(smem_raw);
    float2* buf = tw + P;

    const int tid = threadIdx.x;
    const int seq = tid / G, t = tid % G;
    const long row = (long)blockIdx.x * SEQ + seq;
    const bool valid = row < n_rows;
    float2* my = buf + seq * PITCH;
    for (int i = tid; i < P; i += kThreads) tw[i] = tw_g[i];
    if (valid) {
        const float2* src = ubuf + row * P;
        for (int k = t; k < P; k += G) my[k] = src[k];
    }
    __syncthreads();
    fft_inplace<P, 1, G, +1>(my, tw, t);
    if (valid) {
        const long fo = row / S;
        const int y = (int)(row % S);
        for (int x = t; x < S; x += G) {
            float2 v = my[x + (PAD ? S / 2 : 0)];
            v.x *= scale;
            v.y *= scale;
            const long pix = (long)y * S + x;
            if (out_mode == HOLO_ASM_OUT_AMP_PHASE) {
                out0[fo * ostride0 + pix] = sqrtf(v.x * v.x + v.y * v.y);
                out1[fo * ostride1 + pix] = atan2f(v.y, v.x);
            } else if (out_mode == HOLO_ASM_OUT_INTENSITY) {
                // Forward_model.py:39: pow(abs(.),2)
                const float a = sqrtf(v.x * v.x + v.y * v.y);
                out0[fo * ostride0 + pix] = a * a;
            } else {
                reinterpret_cast<float2*>(out0)[fo * ostride0 + pix] = v;
            }
        }
    }
}

// ---- host side ---------------------------------------------------------------------------------------
struct AsmArgs {
    const float *in0, *in1;
    int in_mode;
    long s0, s1;
    float *out0, *out1;
    int out_mode;
    long os0, os1;
    const double* phase_const;
    int n_fields, n_dist;
    double ppx, lam2, lamb;
    float phase_scale;
    float2 *tw, *tbuf, *ubuf;
    hipStream_t stream;
};

template <int P, bool PAD, int CW>
static int launch_asm(const AsmArgs& a) {
    constexpr int S = PAD ? P / 2 : P;
    constexpr int G = P / 8;
    constexpr int SEQ = kThreads / G;
    const size_t row_lds = (size_t)(P + SEQ * (P + kRowPad)) * sizeof(float2);
    const size_t col_lds = (size_t)(P + CW * (P + kColPad)) * sizeof(float2);
    static bool attr_done = false;  // idempotent; a race only repeats the call
    if (!attr_done) {
        if (col_lds > 48 * 1024)
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&asm_cols_kernel<P, PAD, CW>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)col_lds);
        if (row_lds > 48 * 1024) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&asm_rows_fwd_kernel<P, PAD>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)row_lds);
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&asm_rows_inv_kernel<P, PAD>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)row_lds);
        }
        attr_done = true;
    }
    hipLaunchKernelGGL(asm_twiddle_kernel, dim3(ceil_div(P, 256)), dim3(256), 0, a.stream, a.tw, P);
    const long rows_in = (long)a.n_fields * S;
    hipLaunchKernelGGL((asm_rows_fwd_kernel<P, PAD>), dim3((unsigned)((rows_in + SEQ - 1) / SEQ)), dim3(kThreads),
                       row_lds, a.stream, a.in0, a.in1, a.in_mode, a.s0, a.s1, a.phase_scale, a.tbuf, a.tw, rows_in);
    hipLaunchKernelGGL((asm_cols_kernel<P, PAD, CW>), dim3(P / CW, a.n_fields * a.n_dist), dim3(kThreads), col_lds,
                       a.stream, a.tbuf, a.ubuf, a.tw, a.phase_const, a.n_dist, a.ppx, a.lam2, a.lamb);
    const long rows_out = rows_in * a.n_dist;
    const float scale = 1.0f / ((float)P * (float)P);  // ifft2 "backward" normalisation; exact (power of two)
    hipLaunchKernelGGL((asm_rows_inv_kernel<P, PAD>), dim3((unsigned)((rows_out + SEQ - 1) / SEQ)), dim3(kThreads),
                       row_lds, a.stream, a.ubuf, a.out0, a.out1, a.out_mode, a.os0, a.os1, a.tw, rows_out, scale);
    return check_launch("holo_asm_propagate");
}

template <int P, int CW>
static int launch_asm_pad(const AsmArgs& a, bool pad) {
    return pad ? launch_asm<P, true, CW>(a) : launch_asm<P, false, CW>(a);
}

}  // namespace holo

extern "C" {

size_t holo_asm_workspace_bytes(int n_fields, int n_dist, int s, int zero_padding) {
    if (n_fields <= 0 || n_dist <= 0 || s <= 0) return 0;
    const size_t p = zero_padding ? 2 * (size_t)s : (size_t)s;
    const size_t slab = (size_t)s * p * sizeof(float2);
    return 2048 * sizeof(float2) + slab * (size_t)n_fields + slab * (size_t)n_fields * (size_t)n_dist;
}

int holo_asm_propagate(const void* in0, const void* in1, int in_mode, long in0_stride, long in1_stride,
                       void* out0, void* out1, int out_mode, long out0_stride, long out1_stride,
                       const double* phase_const, int n_fields, int n_dist, int s, int zero_padding,
                       double wavelength, double pixel_size, float phase_scale, void* workspace,
                       size_t workspace_bytes, void* stream) {
    using namespace holo;
    HOLO_REQUIRE(in0 && out0 && phase_const && workspace, HOLO_ERR_ARG, "holo_asm_propagate: null pointer");
    HOLO_REQUIRE(in_mode == HOLO_ASM_IN_COMPLEX || (in_mode == HOLO_ASM_IN_AMP_PHASE && in1), HOLO_ERR_ARG,
                 "holo_asm_propagate: bad in_mode %d / missing phase plane", in_mode);
    HOLO_REQUIRE(out_mode == HOLO_ASM_OUT_INTENSITY || out_mode == HOLO_ASM_OUT_COMPLEX ||
                     (out_mode == HOLO_ASM_OUT_AMP_PHASE && out1),
                 HOLO_ERR_ARG, "holo_asm_propagate: bad out_mode %d / missing phase output", out_mode);
    HOLO_REQUIRE(n_fields > 0 && n_dist > 0 && s > 0, HOLO_ERR_ARG, "holo_asm_propagate: empty problem (%d,%d,%d)",
                 n_fields, n_dist, s);
    HOLO_REQUIRE((long)n_fields * n_dist <= 65535, HOLO_ERR_UNSUPPORTED,
                 "holo_asm_propagate: at most 65535 output fields per call (got %ld)", (long)n_fields * n_dist);
    HOLO_REQUIRE(wavelength > 0 && pixel_size > 0, HOLO_ERR_ARG, "holo_asm_propagate: wavelength/pixel_size <= 0");
    const int p = zero_padding ? 2 * s : s;
    HOLO_REQUIRE((p & (p - 1)) == 0 && p >= 16 && p <= 2048, HOLO_ERR_UNSUPPORTED,
                 "holo_asm_propagate: FFT side %d (S=%d, zero_padding=%d) must be a power of two in [16,2048]", p, s,
                 zero_padding);
    HOLO_REQUIRE(workspace_bytes >= holo_asm_workspace_bytes(n_fields, n_dist, s, zero_padding), HOLO_ERR_WORKSPACE,
                 "holo_asm_propagate: workspace %zu < required %zu", workspace_bytes,
                 holo_asm_workspace_bytes(n_fields, n_dist, s, zero_padding));
    AsmArgs a;
    a.in0 = static_cast<const float*>(in0);
    a.in1 = static_cast<const float*>(in1);
    a.in_mode = in_mode;
    a.s0 = in0_stride;
    a.s1 = in1_stride;
    a.out0 = static_cast<float*>(out0);
    a.out1 = static_cast<float*>(out1);
    a.out_mode = out_mode;
    a.os0 = out0_stride;
    a.os1 = out1_stride;
    a.phase_const = phase_const;
    a.n_fields = n_fields;
    a.n_dist = n_dist;
    // numpy: fx = (arange(2S)/2 - S//2)/(S*px) == signed_bin/(P*px) bit for bit (scaling by 2 is exact)
    a.ppx = (double)s * pixel_size * (zero_padding ? 2.0 : 1.0);
    a.lam2 = wavelength * wavelength;
    a.lamb = wavelength;
    a.phase_scale = phase_scale;
    a.tw = static_cast<float2*>(workspace);
    a.tbuf = a.tw + 2048;
    a.ubuf = a.tbuf + (size_t)n_fields * s * p;
    a.stream = as_stream(stream);
    const bool pad = zero_padding != 0;
    switch (p) {
        case 16: return launch_asm_pad<16, 16>(a, pad);
        case 32: return launch_asm_pad<32, 16>(a, pad);
        case 64: return launch_asm_pad<64, 16>(a, pad);
        case 128: return launch_asm_pad<128, 16>(a, pad);
        case 256: return launch_asm_pad<256, 16>(a, pad);
        case 512: return launch_asm_pad<512, 16>(a, pad);
        case 1024: return launch_asm_pad<1024, 8>(a, pad);
        case 2048: return launch_asm_pad<2048, 4>(a, pad);
    }
    return HOLO_ERR_UNSUPPORTED;
}

}  // extern "C"
