// In-LDS Stockham autosort FFT for power-of-two lengths 16..2048 (gfx950).
//
// A sequence of P complex64 values lives in LDS; G threads of the block cooperate on it, each
// owning (P/R)/G radix-R butterflies per pass.  Pass p has Ns = product of the earlier radices:
//   v[r]  = buf[j + r*P/R] * W_P^{dir * r * (j mod Ns) * P/(Ns*R)}        (twiddle)
//   v     = DFT_R(v)                                                       (registers)
//   buf[(j - j mod Ns)*R + (j mod Ns) + r*Ns] = v[r]                       (autosort scatter)
// so after the last pass the spectrum is in natural order (validated against numpy.fft for all
// lengths; the schedule is radix 8 while it divides, then 4, then 2).  In place: every thread
// reads all its inputs, the block barriers, then writes.
//
// `tw` is a table of P entries exp(-2*pi*i*m/P) in LDS, correctly rounded from float64.
#pragma once
#include <hip/hip_runtime.h>

namespace holo {

__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ float2 cmul(float2 a, float2 b) {
    return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}
// multiply by DIR*i (DIR = -1 forward, +1 inverse)
template <int DIR>
__device__ __forceinline__ float2 mul_di(float2 a) {
    return DIR > 0 ? make_float2(-a.y, a.x) : make_float2(a.y, -a.x);
}

template <int R, int DIR>
struct Bfly;

template <int DIR>
struct Bfly<2, DIR> {
    static __device__ __forceinline__ void run(float2* v) {
        const float2 a = v[0], b = v[1];
        v[0] = cadd(a, b);
        v[1] = csub(a, b);
    }
};

template <int DIR>
struct Bfly<4, DIR> {
    static __device__ __forceinline__ void run(float2* v) {
        const float2 a0 = cadd(v[0], v[2]), a1 = csub(v[0], v[2]);
        const float2 a2 = cadd(v[1], v[3]), a3 = mul_di<DIR>(csub(v[1], v[3]));
        v[0] = cadd(a0, a2);
        v[1] = cadd(a1, a3);
        v[2] = csub(a0, a2);
        v[3] = csub(a1, a3);
    }
};

template <int DIR>
struct Bfly<8, DIR> {
    static __device__ __forceinline__ void run(float2* v) {
        float2 e[4] = {v[0], v[2], v[4], v[6]};
        float2 o[4] = {v[1], v[3], v[5], v[7]};
        Bfly<4, DIR>::run(e);
        Bfly<4, DIR>::run(o);
        constexpr float s = 0.70710678118654752440f;
        constexpr float d = (float)DIR;
        // o[a] * w8^a, w8 = exp(DIR*2*pi*i/8)
        const float2 t0 = o[0];
        const float2 t1 = make_float2(s * (o[1].x - d * o[1].y), s * (o[1].y + d * o[1].x));
        const float2 t2 = mul_di<DIR>(o[2]);
        const float2 t3 = make_float2(s * (-o[3].x - d * o[3].y), s * (-o[3].y + d * o[3].x));
        v[0] = cadd(e[0], t0); v[4] = csub(e[0], t0);
        v[1] = cadd(e[1], t1); v[5] = csub(e[1], t1);
        v[2] = cadd(e[2], t2); v[6] = csub(e[2], t2);
        v[3] = cadd(e[3], t3); v[7] = csub(e[3], t3);
    }
};

template <int P, int NS, int R, int G, int DIR>
__device__ __forceinline__ void fft_pass(float2* __restrict__ buf, const float2* __restrict__ tw, int t) {
    constexpr int T = P / R;
    constexpr int NB = (T + G - 1) / G;
    constexpr bool FULL = (NB * G == T);
    float2 v[NB][R];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        const int j = t + nb * G;
        if (FULL || j < T) {
#pragma unroll
            for (int r = 0; r < R; ++r) v[nb][r] = buf[j + r * T];
        }
    }
    __syncthreads();
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        const int j = t + nb * G;
        if (FULL || j < T) {
            const int k = j & (NS - 1);
            if constexpr (NS > 1) {
                constexpr int STEP = P / (NS * R);
#pragma unroll
                for (int r = 1; r < R; ++r) {
                    float2 w = tw[r * k * STEP];
                    if (DIR > 0) w.y = -w.y;
                    v[nb][r] = cmul(v[nb][r], w);
                }
            }
            Bfly<R, DIR>::run(v[nb]);
            const int j0 = (j - k) * R + k;
#pragma unroll
            for (int r = 0; r < R; ++r) buf[j0 + r * NS] = v[nb][r];
        }
    }
    __syncthreads();
}

// Full transform; every thread of the block must call it (it contains block barriers).
template <int P, int NS, int G, int DIR>
__device__ __forceinline__ void fft_inplace(float2* __restrict__ buf, const float2* __restrict__ tw, int t) {
    if constexpr (NS < P) {
        constexpr int REM = P / NS;
        constexpr int R = (REM % 8 == 0) ? 8 : ((REM % 4 == 0) ? 4 : 2);
        fft_pass<P, NS, R, G, DIR>(buf, tw, t);
        fft_inplace<P, NS * R, G, DIR>(buf, tw, t);
    }
}

}  // namespace holo
