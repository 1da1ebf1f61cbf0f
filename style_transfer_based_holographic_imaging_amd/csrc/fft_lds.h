// Register-resident Stockham autosort FFT for power-of-two lengths 16..2048 (gfx950).
//
// A sequence of P complex64 values is owned by G threads; each thread keeps E = P/G elements in
// registers, laid out for the pass's radix R as v[nb*R + r], butterfly j = t + nb*G, element index
// j + r*(P/R).  One pass with Ns = product of the earlier radices:
//     v[r] *= W^{dir * r * (j mod Ns) / (Ns*R)}  ;  v = DFT_R(v)              (fft_compute, registers)
//     lds[(j - j mod Ns)*R + (j mod Ns) + r*Ns] = v[r]                         (fft_scatter)
//     barrier ;  v[r] = lds[j + r*P/R']  for the next radix R'                  (fft_gather)
// After the LAST pass the outputs sit in registers at index j + r*(P/R_last) — the gather pattern
// of that radix — so (a) global stores/epilogues run straight from registers and (b) an inverse
// transform whose FIRST radix equals the forward's LAST radix continues from the same registers:
// the column kernel multiplies by the transfer function between the two without touching LDS.
// The first pass likewise takes its inputs from wherever the caller loaded them (global memory).
//
// Instruction economy (the column kernel is VALU-bound, not HBM-bound, at P = 512):
//   * complex values are native 2-vectors, so adds/mults are single v_pk_*_f32 without re-pairing moves
//   * LDS indices are padded, pad(i) = i + (i >> 3) (the first autosort scatter is otherwise an 8-way
//     bank conflict), and every gather/scatter is base + r*const so the r-loop uses immediate offsets
//   * twiddles come from per-pass packed tables tw[off(pass) + k*(R-1) + (r-1)]: one address per
//     butterfly instead of one per factor.  Radix plans are palindromes where the length allows it
//     (8.8.8, 8.4.8, 8.4.4.8 ...), so the inverse (reversed plan, conjugated factors) shares the table.
//
// Plans and the linear-address identities were validated against numpy.fft for every length, forward
// and inverse.  Table entries are exp(-2*pi*i*r*k/(Ns*R)), correctly rounded from float64.
#pragma once
#include <hip/hip_runtime.h>

#ifndef HOLO_FFT_SINGLE_READS
#define HOLO_FFT_SINGLE_READS 1   // measured: column kernel 43.9 -> 38.0 us at P = 512 (64 fields)
#endif

namespace holo {

typedef float v2f __attribute__((ext_vector_type(2)));

// Complex arithmetic on (re, im) register pairs as SINGLE packed instructions.  hipcc does not use the VOP3P
// op_sel / neg modifiers for the half-swaps and sign flips of complex products and +-i rotations (it emits
// v_xor + v_mov around the v_pk ops: ~45 % of the FFT's instructions in the first version), so the four
// primitives are written in inline asm.  op_sel[i] / op_sel_hi[i] pick the half of source i that feeds the
// low / high result; neg_lo / neg_hi negate it.  Pure VALU register ops: no waits or hazards involved.
__device__ __forceinline__ v2f cadd_i(v2f a, v2f b) {   // a + i*b = (a.x - b.y, a.y + b.x)
    v2f d;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
__device__ __forceinline__ v2f csub_i(v2f a, v2f b) {   // a - i*b = (a.x + b.y, a.y - b.x)
    v2f d;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
// a + (DIR*i)*b and a - (DIR*i)*b
template <int DIR> __device__ __forceinline__ v2f cadd_di(v2f a, v2f b) { return DIR > 0 ? cadd_i(a, b) : csub_i(a, b); }
template <int DIR> __device__ __forceinline__ v2f csub_di(v2f a, v2f b) { return DIR > 0 ? csub_i(a, b) : cadd_i(a, b); }

__device__ __forceinline__ v2f cmul(v2f a, v2f b) {     // a*b = a.xx*b + a.yy*(-b.y, b.x)
    v2f t, d;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(t) : "v"(a), "v"(b));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]" : "=v"(d) : "v"(a), "v"(b), "v"(t));
    return d;
}
__device__ __forceinline__ v2f cmul_conj(v2f a, v2f b) {  // a*conj(b) = a.xx*(b.x,-b.y) + a.yy*(b.y, b.x)
    v2f t, d;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1] neg_hi:[0,1]" : "=v"(t) : "v"(a), "v"(b));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1]" : "=v"(d) : "v"(a), "v"(b), "v"(t));
    return d;
}
// twiddle product in direction DIR (tables hold the forward factors)
template <int DIR> __device__ __forceinline__ v2f cmul_tw(v2f a, v2f w) { return DIR > 0 ? cmul_conj(a, w) : cmul(a, w); }

__host__ __device__ constexpr int fft_pad(int i) { return i + (i >> 3); }
__host__ __device__ constexpr int fft_padded_len(int p) { return p + (p >> 3); }

// radix schedule: forward order; the inverse runs it backwards
template <int P> struct FftPlan;
template <> struct FftPlan<16>   { static constexpr int n = 2; static constexpr int r[4] = {4, 4, 1, 1}; };
template <> struct FftPlan<32>   { static constexpr int n = 2; static constexpr int r[4] = {8, 4, 1, 1}; };
template <> struct FftPlan<64>   { static constexpr int n = 2; static constexpr int r[4] = {8, 8, 1, 1}; };
template <> struct FftPlan<128>  { static constexpr int n = 3; static constexpr int r[4] = {4, 8, 4, 1}; };
template <> struct FftPlan<256>  { static constexpr int n = 3; static constexpr int r[4] = {8, 4, 8, 1}; };
template <> struct FftPlan<512>  { static constexpr int n = 3; static constexpr int r[4] = {8, 8, 8, 1}; };
template <> struct FftPlan<1024> { static constexpr int n = 4; static constexpr int r[4] = {8, 4, 4, 8}; };
template <> struct FftPlan<2048> { static constexpr int n = 4; static constexpr int r[4] = {8, 8, 8, 4}; };

template <int P>
__host__ __device__ constexpr bool fft_plan_symmetric() {
    for (int i = 0; i < FftPlan<P>::n; ++i)
        if (FftPlan<P>::r[i] != FftPlan<P>::r[FftPlan<P>::n - 1 - i]) return false;
    return true;
}
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
// packed twiddle table of direction DIR: entries before pass i, and in total
template <int P, int DIR>
__host__ __device__ constexpr int fft_tw_offset(int i) {
    int off = 0;
    for (int q = 1; q < i; ++q) off += fft_ns<P, DIR>(q) * (fft_radix<P, DIR>(q) - 1);
    return off;
}
template <int P, int DIR>
__host__ __device__ constexpr int fft_tw_total() { return fft_tw_offset<P, DIR>(FftPlan<P>::n); }

// Fills the packed table of direction DIR (fft_tw_total<P,DIR>() entries) cooperatively, e.g. straight into
// LDS at kernel start.  The angles are dyadic rationals (-2*r*k/(Ns*R) half-turns, exact in fp32), so
// sincospif reduces them exactly and the entries are good to ~1 ulp; a few dozen instructions per thread,
// which removes a table kernel and a launch boundary from every propagation call.
template <int P, int DIR>
__device__ __forceinline__ void fft_fill_twiddles(v2f* tw, int tid, int nthreads) {
#pragma unroll
    for (int i = 1; i < FftPlan<P>::n; ++i) {
        const int R = fft_radix<P, DIR>(i), NS = fft_ns<P, DIR>(i), off = fft_tw_offset<P, DIR>(i);
        for (int e = tid; e < NS * (R - 1); e += nthreads) {
            const int k = e / (R - 1), r = e % (R - 1) + 1;
            float sn, cs;
            sincospif(-2.0f * (float)(r * k) / (float)(NS * R), &sn, &cs);
            tw[off + e] = v2f{cs, sn};
        }
    }
}

template <int R, int DIR>
struct Bfly;

template <int DIR>
struct Bfly<2, DIR> {
    static __device__ __forceinline__ void run(v2f* v) {
        const v2f a = v[0], b = v[1];
        v[0] = a + b;
        v[1] = a - b;
    }
};

template <int DIR>
struct Bfly<4, DIR> {
    static __device__ __forceinline__ void run(v2f* v) {
        const v2f a0 = v[0] + v[2], a1 = v[0] - v[2];
        const v2f a2 = v[1] + v[3], d = v[1] - v[3];
        v[0] = a0 + a2;
        v[1] = cadd_di<DIR>(a1, d);   // a1 + (DIR*i)*d
        v[2] = a0 - a2;
        v[3] = csub_di<DIR>(a1, d);
    }
};

template <int DIR>
struct Bfly<8, DIR> {
    static __device__ __forceinline__ void run(v2f* v) {
        v2f e[4] = {v[0], v[2], v[4], v[6]};
        v2f o[4] = {v[1], v[3], v[5], v[7]};
        Bfly<4, DIR>::run(e);
        Bfly<4, DIR>::run(o);
        const v2f s = {0.70710678118654752440f, 0.70710678118654752440f};
        // o[a] * w8^a, w8 = exp(DIR*2*pi*i/8):  w8 = s*(1 + DIR*i),  w8^2 = DIR*i,  w8^3 = -s*(1 - DIR*i)
        const v2f p1 = cadd_di<DIR>(o[1], o[1]);   // (1 + DIR*i) o1
        const v2f p3 = csub_di<DIR>(o[3], o[3]);   // (1 - DIR*i) o3
        v[0] = e[0] + o[0];            v[4] = e[0] - o[0];
        v[1] = e[1] + p1 * s;          v[5] = e[1] - p1 * s;          // contracts to v_pk_fma_f32
        v[2] = cadd_di<DIR>(e[2], o[2]); v[6] = csub_di<DIR>(e[2], o[2]);
        v[3] = e[3] - p3 * s;          v[7] = e[3] + p3 * s;
    }
};

// element index held in register slot e (= nb*R + r) of thread t for a radix-R gather pattern
template <int P, int G, int R>
__device__ __forceinline__ int fft_elem_index(int t, int e) {
    return t + (e / R) * G + (e % R) * (P / R);
}

// twiddle + butterfly on registers.  E = P/G registers; `twp` = packed table of this direction's plan
// (forward-signed entries; the inverse conjugates them).
template <int P, int G, int DIR, int I>
__device__ __forceinline__ void fft_compute(v2f* v, int t, const v2f* __restrict__ twp) {
    constexpr int R = fft_radix<P, DIR>(I);
    constexpr int NS = fft_ns<P, DIR>(I);
    constexpr int E = P / G;
    constexpr int NB = E / R;
    static_assert(NB * R == E, "elements per thread must be a multiple of the radix");
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        if constexpr (NS > 1) {
            const int j = t + nb * G;
            const int k = j & (NS - 1);
            const v2f* __restrict__ w = twp + fft_tw_offset<P, DIR>(I) + k * (R - 1);
#pragma unroll
            for (int r = 1; r < R; ++r) v[nb * R + r] = cmul_tw<DIR>(v[nb * R + r], w[r - 1]);
        }
        Bfly<R, DIR>::run(v + nb * R);
    }
}

// autosort scatter of a pass's outputs into the (padded) LDS sequence; base + r*const wherever the
// padding is linear in r (validated: Ns >= 8, or Ns == 1 with R | 8)
template <int P, int G, int R, int NS>
__device__ __forceinline__ void fft_scatter(v2f* __restrict__ buf, const v2f* v, int t) {
    constexpr int NB = (P / G) / R;
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        const int j = t + nb * G;
        const int k = j & (NS - 1);
        const int j0 = (j - k) * R + k;
        if constexpr (NS >= 8) {
            v2f* __restrict__ b = buf + fft_pad(j0);
#pragma unroll
            for (int r = 0; r < R; ++r) b[r * (NS + NS / 8)] = v[nb * R + r];
        } else if constexpr (NS == 1 && (8 % R) == 0) {
            v2f* __restrict__ b = buf + fft_pad(j0);
#pragma unroll
            for (int r = 0; r < R; ++r) b[r] = v[nb * R + r];
        } else {
#pragma unroll
            for (int r = 0; r < R; ++r) buf[fft_pad(j0 + r * NS)] = v[nb * R + r];
        }
    }
}

// gather the inputs of a radix-R pass from the (padded) LDS sequence
template <int P, int G, int R>
__device__ __forceinline__ void fft_gather(const v2f* __restrict__ buf, v2f* v, int t) {
    constexpr int NB = (P / G) / R;
    constexpr int T = P / R;
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        const int j = t + nb * G;
        if constexpr (T % 8 == 0) {
            const v2f* __restrict__ b = buf + fft_pad(j);
#pragma unroll
            for (int r = 0; r < R; ++r) {
                v[nb * R + r] = b[r * (T + T / 8)];
#if HOLO_FFT_SINGLE_READS
                // keeps hipcc from pairing the reads into ds_read2_b64, which the LDS serves at half the rate of two
                // ds_read_b64 (8 vs 2 x 2 cycles per wave, MI355X_MICROARCH.md LDS table); no wait is forced, unlike reads
                // issued from inline asm
                asm volatile("" ::: "memory");
#endif
            }
        } else {
#pragma unroll
            for (int r = 0; r < R; ++r) v[nb * R + r] = buf[fft_pad(j + r * T)];
        }
    }
}

// Elements per thread / threads per sequence: a sequence never spans more than one wave (G <= 64 and G
// divides 64), so every gather/scatter exchange is WAVE-LOCAL: the LDS unit executes one wave's DS
// instructions in order, hence a wave's reads see its own earlier writes and its earlier reads are not
// overtaken by its later writes.  No block barrier is needed inside the transform — only a compiler
// fence so the accesses are not reordered in the instruction stream.  (The first version used 13
// __syncthreads() per column block and spent 46 % of its wave-cycles waiting at them.)
// P = 2048 is the exception: 32 elements per thread costs too many registers (measured slower), so there a
// sequence spans two waves (G = 128) and the exchanges fall back to block barriers.
template <int P> struct FftGeom {
    static constexpr int E = P >= 1024 ? 16 : 8;
    static constexpr int G = P / E;                   // 2 .. 64 lanes per sequence (128 for P = 2048)
    static constexpr bool WAVE_LOCAL = G <= 64;
    static_assert(!WAVE_LOCAL || 64 % G == 0, "a wave-local sequence must stay inside one wave");
};

__device__ __forceinline__ void fft_wave_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// exchange synchronisation for sequences of G lanes: wave-local fence, or a block barrier when a sequence
// spans waves (then EVERY thread of the block must reach it)
template <int G>
__device__ __forceinline__ void fft_sync() {
    if constexpr (G <= 64) fft_wave_fence();
    else __syncthreads();
}

// Passes I..n-1 of the transform in direction DIR, starting from registers laid out for the radix of
// pass I and ending with registers laid out for the last radix (natural order at index
// fft_elem_index<P,G,R_last>).  `buf` is private to the G lanes of this sequence; every thread of the
// block must call it when G > 64 (block barriers).
template <int P, int G, int DIR, int I = 0>
__device__ __forceinline__ void fft_run(v2f* v, v2f* __restrict__ buf, const v2f* __restrict__ twp, int t) {
    constexpr int N = FftPlan<P>::n;
    fft_compute<P, G, DIR, I>(v, t, twp);
    if constexpr (I + 1 < N) {
        fft_sync<G>();   // earlier reads of buf (previous gather / staging reads) stay ahead of the writes
        fft_scatter<P, G, fft_radix<P, DIR>(I), fft_ns<P, DIR>(I)>(buf, v, t);
        fft_sync<G>();
        fft_gather<P, G, fft_radix<P, DIR>(I + 1)>(buf, v, t);
        fft_run<P, G, DIR, I + 1>(v, buf, twp, t);
    }
}

}  // namespace holo
