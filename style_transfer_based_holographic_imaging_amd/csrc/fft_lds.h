// Register-resident Stockham autosort FFT for power-of-two lengths 16..2048 (gfx950).
//
// A sequence of P complex64 values is owned by G threads; each thread keeps E = P/G elements in
// registers, laid out for the pass's radix R as v[nb*R + r], butterfly j = t + nb*G, element index
// j + r*(P/R).  One pass with Ns = product of the earlier radices:
//     v[r] *= W_P^{dir * r * (j mod Ns) * P/(Ns*R)}  ;  v = DFT_R(v)          (fft_compute, registers)
//     lds[(j - j mod Ns)*R + (j mod Ns) + r*Ns] = v[r]                         (fft_scatter)
//     barrier ;  v[r] = lds[j + r*P/R']  for the next radix R'                  (fft_gather)
// After the LAST pass the outputs sit in registers at index j + r*(P/R_last) — the gather pattern
// of that radix — so (a) global stores/epilogues run straight from registers and (b) an inverse
// transform whose FIRST radix equals the forward's LAST radix continues from the same registers:
// the column kernel multiplies by the transfer function between the two without touching LDS.
// The first pass likewise takes its inputs from wherever the caller loaded them (global memory).
//
// LDS indices are padded, pad(i) = i + (i >> 3): the autosort scatter of the first pass has a
// stride of R elements across lanes, which without padding is an 8-way bank conflict on
// ds_write_b64 (measured: the dominant cost of the first version of the column kernel).
//
// The schedule (radices per length) was validated against numpy.fft for every length, forward and
// inverse; the inverse uses the reversed radix list.  `tw` = P entries exp(-2*pi*i*m/P) in LDS,
// correctly rounded from float64.
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

__host__ __device__ constexpr int fft_pad(int i) { return i + (i >> 3); }
__host__ __device__ constexpr int fft_padded_len(int p) { return p + (p >> 3); }

// radix schedule: forward order; the inverse runs it backwards
template <int P> struct FftPlan;
template <> struct FftPlan<16>   { static constexpr int n = 2; static constexpr int r[4] = {4, 4, 1, 1}; };
template <> struct FftPlan<32>   { static constexpr int n = 2; static constexpr int r[4] = {8, 4, 1, 1}; };
template <> struct FftPlan<64>   { static constexpr int n = 2; static constexpr int r[4] = {8, 8, 1, 1}; };
template <> struct FftPlan<128>  { static constexpr int n = 3; static constexpr int r[4] = {8, 4, 4, 1}; };
template <> struct FftPlan<256>  { static constexpr int n = 3; static constexpr int r[4] = {8, 8, 4, 1}; };
template <> struct FftPlan<512>  { static constexpr int n = 3; static constexpr int r[4] = {8, 8, 8, 1}; };
template <> struct FftPlan<1024> { static constexpr int n = 4; static constexpr int r[4] = {8, 8, 4, 4}; };
template <> struct FftPlan<2048> { static constexpr int n = 4; static constexpr int r[4] = {8, 8, 8, 4}; };

// radix of pass `i` (0-based) for direction DIR, and the product of the radices before it
template <int P, int DIR>
__host__ __device__ constexpr int fft_radix(int i) {
    return DIR < 0 ? FftPlan<P>::r[i] : FftPlan<P>::r[FftPlan<P>::n - 1 - i];
}
template <int P, int DIR>
__host__ __device__ constexpr int fft_ns(int i) {
    int ns = 1;
    for (int k = 0; k < i; ++k) ns *= fft_radix<P, DIR>(k);
    return ns;
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

// element index held in register slot e (= nb*R + r) of thread t for a radix-R gather pattern
template <int P, int G, int R>
__device__ __forceinline__ int fft_elem_index(int t, int e) {
    return t + (e / R) * G + (e % R) * (P / R);
}

// twiddle + butterfly on registers.  E = P/G registers; valid only for threads with t < G.
template <int P, int G, int R, int NS, int DIR>
__device__ __forceinline__ void fft_compute(float2* v, int t, const float2* __restrict__ tw) {
    constexpr int E = P / G;
    constexpr int NB = E / R;
    static_assert(NB * R == E, "elements per thread must be a multiple of the radix");
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        if constexpr (NS > 1) {
            const int j = t + nb * G;
            const int k = j & (NS - 1);
            constexpr int STEP = P / (NS * R);
#pragma unroll
            for (int r = 1; r < R; ++r) {
                float2 w = tw[r * k * STEP];
                if (DIR > 0) w.y = -w.y;
                v[nb * R + r] = cmul(v[nb * R + r], w);
            }
        }
        Bfly<R, DIR>::run(v + nb * R);
    }
}

// autosort scatter of a pass's outputs into the (padded) LDS sequence
template <int P, int G, int R, int NS>
__device__ __forceinline__ void fft_scatter(float2* __restrict__ buf, const float2* v, int t) {
    constexpr int NB = (P / G) / R;
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        const int j = t + nb * G;
        const int k = j & (NS - 1);
        const int j0 = (j - k) * R + k;
#pragma unroll
        for (int r = 0; r < R; ++r) buf[fft_pad(j0 + r * NS)] = v[nb * R + r];
    }
}

// gather the inputs of a radix-R pass from the (padded) LDS sequence
template <int P, int G, int R>
__device__ __forceinline__ void fft_gather(const float2* __restrict__ buf, float2* v, int t) {
    constexpr int NB = (P / G) / R;
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        const int j = t + nb * G;
#pragma unroll
        for (int r = 0; r < R; ++r) v[nb * R + r] = buf[fft_pad(j + r * (P / R))];
    }
}

// Passes FIRST..n-1 of the transform in direction DIR, starting from registers laid out for the
// radix of pass FIRST and ending with registers laid out for the last radix (natural order at
// index fft_elem_index<P,G,R_last>).  Every thread of the block must call it (block barriers).
template <int P, int G, int DIR, int I = 0>
__device__ __forceinline__ void fft_run(float2* v, float2* __restrict__ buf, const float2* __restrict__ tw, int t) {
    constexpr int N = FftPlan<P>::n;
    constexpr int R = fft_radix<P, DIR>(I);
    constexpr int NS = fft_ns<P, DIR>(I);
    fft_compute<P, G, R, NS, DIR>(v, t, tw);
    if constexpr (I + 1 < N) {
        __syncthreads();  // everyone has finished reading buf (previous gather / staging reads)
        fft_scatter<P, G, R, NS>(buf, v, t);
        __syncthreads();
        fft_gather<P, G, fft_radix<P, DIR>(I + 1)>(buf, v, t);
        fft_run<P, G, DIR, I + 1>(v, buf, tw, t);
    }
}

}  // namespace holo
