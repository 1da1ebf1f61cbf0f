// Angular-spectrum free-space propagation on gfx950.
//
// Replaces the aten op chain of utils/Angular_Spectrum_Method.py:7-36 (ASM) and the field packing /
// abs / angle / |.|^2 epilogues of utils/Forward_model.py:22-39 (Holo_Generator.forward).
//
// out = crop( ifft2( H . fft2( replicate_pad(O) ) ) ),  H[ky][kx] = exp(i*c*sqrt(max(0,1-l^2 f^2))/l)
//
// fftshift/ifftshift only relabel the bins, so H is evaluated at the signed frequency of the
// un-shifted bin and no shift pass exists.  The P x P complex grid does not fit a CU's LDS, so the
// 2-D transform is three HBM-bound kernels (80*S^2 algorithmic bytes per field):
//
//   K1 rows_fwd : amp,phase (8S^2 B) -> pack a*e^{ip}, replicate-pad the ROW to P, row FFT
//                 -> T (16S^2 B).  The S/2 padded rows above/below the field are copies of
//                 its first/last row, so only S distinct row FFTs exist.  Inputs go straight from
//                 global memory into the first butterfly's registers.
//   K2 cols     : T columns (16S^2 B) -> column FFT (replicate padding applied by the load), multiply
//                 by H IN REGISTERS between the forward's last and the inverse's first pass (same radix,
//                 same register layout), column IFFT, keep the S centre rows -> U (16S^2 B).  H is
//                 generated per column: float64 phase argument, range reduction in float64, fp32
//                 sin/cos; rows ky and P-ky share one value through an LDS table.
//   K3 rows_inv : U rows (16S^2 B) -> row IFFT, keep the S centre columns, scale 1/P^2,
//                 abs/angle | intensity | complex epilogue (8S^2 B), register to global.
//
// Layout of the intermediates (round 2): T and U are stored TRANSPOSED, [field][kx][y] (a column's S
// values contiguous), for 32 <= P <= 1024.  The column kernel is the one limited by VALU + LDS + memory at
// once, the row kernels are purely memory-bound with idle LDS — so the two transposes live in the ROW
// kernels (16 rows per block through a padded LDS tile, 128-byte runs to/from memory) and the column kernel
// moves its data register <-> global with no staging, no block barrier and one wave (or part of one) per
// column: every wave runs load -> FFT -> H -> IFFT -> store on its own and the CU overlaps the phases of its
// 28 waves.  (Round 1 staged 16-column tiles through LDS inside the column kernel: 2 barriers, 2 extra LDS
// round trips per element, blocks of 1024 threads moving in lockstep.)  P = 2048 and the smallest padded
// grid keep the row-major layout and the round-1 kernels.
//
// Transforms run in fp32 (complex64) with float64-rounded twiddles; the reference runs the product
// and the inverse transform in complex128 and casts to float at the end
// (Angular_Spectrum_Method.py:29-33, Forward_model.py:28-29) — measured difference ~2e-7 relative.
#include <cstdlib>
#include "common.h"
#include "fft_lds.h"


namespace holo {

constexpr int kRowThreads = 256;

// timing experiments of the row kernels (-DHOLO_DEBUG builds only, HOLO_ASM_ABLATE=<bits>; wrong results): K1: 4 no sincos,
// 8 no FFT, 16 no transposed store, 32 no input loads; K3: 64 no loads, 128 no FFT, 256 no abs/angle, 512 no stores.
// In the shipped build ASM_ABL() is the constant 0 and leaves no code behind.
#ifndef HOLO_ASM_POLY_SINCOS
#define HOLO_ASM_POLY_SINCOS 0   // 1: the Cody-Waite + polynomial sin/cos of rounds 1-3 for the field packing
#endif
#ifdef HOLO_DEBUG
__device__ int g_asm_abl = 0;
#define ASM_ABL(bit_) (g_asm_abl & (bit_))
#else
#define ASM_ABL(bit_) 0
#endif

// workspace head: tables of the generic-size path (twiddles W_P^m and squared frequencies, P <= 4096); the
// power-of-two kernels build their twiddles in LDS themselves and need no table
constexpr int kTwSlots = 4096;

// sin/cos for arguments of moderate size: Cody-Waite reduction by pi/2 with FMA (three-part constant),
// minimax polynomials on [-pi/4, pi/4]; ~1 ulp.  Large arguments take the library path.
__device__ __forceinline__ void sincos_poly(float r, int q, float* s, float* c) {
    const float z = r * r;
    const float sp = r + r * z * (-1.6666667163e-01f + z * (8.3333337680e-03f + z * (-1.9841270114e-04f + z * 2.7557314297e-06f)));
    const float cp = 1.0f - 0.5f * z + z * z * (4.1666667908e-02f + z * (-1.3888889225e-03f + z * 2.4801587642e-05f));
    const float ss = (q & 1) ? cp : sp;
    const float cc = (q & 1) ? sp : cp;
    *s = (q & 2) ? -ss : ss;
    *c = ((q + 1) & 2) ? -cc : cc;
}

// sin/cos of the field's phase (Forward_model.py:22: amplitude * exp(1j * phase)): the hardware v_sin_f32 / v_cos_f32 take their
// argument in REVOLUTIONS.  x / 2 pi is formed as a two-piece product (the fp32 constant and its remainder, the rounding of the
// first product recovered with an FMA), reduced to [-0.5, 0.5] and handed to the hardware: measured max abs error 4.5e-7 on
// [-8, 8] (scratch/sincos_probe.hip; the Cody-Waite + polynomial form it replaces: 1e-7) at 8 instructions instead of ~30 —
// the forward row kernel's packing was 150 of its 500 instructions per lane.  Large arguments take the library path.
__device__ __forceinline__ void sincos_f32(float x, float* s, float* c) {
    if (fabsf(x) > 8192.0f) {
        sincosf(x, s, c);
        return;
    }
#if HOLO_ASM_POLY_SINCOS
    const float n = rintf(x * 0.63661977236758134f);
    float r = fmaf(-n, 1.5707963705062866f, x);
    r = fmaf(-n, -4.371138828673793e-08f, r);
    r = fmaf(-n, -1.7151245100058819e-15f, r);
    sincos_poly(r, (int)n & 3, s, c);
#else
    constexpr float kHi = 0.15915494309189533577f;                                       // fl32(1 / 2 pi)
    constexpr float kLo = (float)(0.15915494309189533577 - (double)0.15915494309189533577f);
    const float th = x * kHi;
    const float tl = fmaf(x, kHi, -th) + x * kLo;
    const float fr = (th - rintf(th)) + tl;
    *s = __builtin_amdgcn_sinf(fr);
    *c = __builtin_amdgcn_cosf(fr);
#endif
}

// |o| through the hardware square root (v_sqrt_f32, 1 ulp) instead of the correctly rounded sqrtf (~10 instructions)
__device__ __forceinline__ float amp_f32(v2f o) { return __builtin_amdgcn_sqrtf(o.x * o.x + o.y * o.y); }

// atan2 for the phase output: octant reduction + an odd polynomial of degree 15 on [0,1] (Chebyshev-interpolated
// atan(a)/a in a^2; max error 1.5e-7 rad evaluated in fp32 — the fp32 FFT noise on the phase is ~1e-6); the
// library atan2f was about half of the inverse row kernel's VALU instructions.  atan2(0,0) = 0 like the library.
__device__ __forceinline__ float atan2_poly(float y, float x) {
    const float ax = fabsf(x), ay = fabsf(y);
    const float mx = fmaxf(ax, ay), mn = fminf(ax, ay);
    // mn / mx through the hardware reciprocal (1 ulp) instead of the correctly rounded division (~10 instructions): the quotient
    // carries ~1.5 ulp, below the polynomial's own 1.5e-7 rad
    const float a = mx > 0.0f ? mn * __builtin_amdgcn_rcpf(mx) : 0.0f;
    const float s = a * a;
    float r = -0.00455979211255908f;
    r = fmaf(r, s, 0.023780519142746925f);
    r = fmaf(r, s, -0.05882975459098816f);
    r = fmaf(r, s, 0.09868865460157394f);
    r = fmaf(r, s, -0.14003290235996246f);
    r = fmaf(r, s, 0.19966961443424225f);
    r = fmaf(r, s, -0.3333181142807007f);
    r = fmaf(r, s, 0.9999998807907104f);
    r = r * a;
    if (ay > ax) r = 1.57079637f - r;
    if (x < 0.0f) r = 3.14159274f - r;
    return y < 0.0f ? -r : r;
}

__device__ __forceinline__ v2f load_field(const float* __restrict__ in0, const float* __restrict__ in1,
                                          int in_mode, long off0, long off1, float phase_scale) {
    if (in_mode == HOLO_ASM_IN_COMPLEX) return reinterpret_cast<const v2f*>(in0)[off0];
    // Forward_model.py:20-22: phase*phase_normalize, then amplitude*exp(1j*phase) in complex64
    const float a = in0[off0];
    const float p = in1[off1] * phase_scale;
    float s, c;
    sincos_f32(p, &s, &c);
    return v2f{a * c, a * s};
}

// H value for squared frequencies fx2 + fy2: theta = (c/l) * sqrt(max(0, 1 - l^2 (fx^2+fy^2)))
// (Angular_Spectrum_Method.py:22-29).  float64 up to the reduced angle (theta reaches ~1e4 rad and must be
// good to ~1e-6 rad), then fp32 sin/cos of the remainder.
__device__ __forceinline__ v2f transfer_value(double sum_f2, double lam2, double c_over_l) {
    const double g = 1.0 - lam2 * sum_f2;
    double root = 0.0;
    if (g > 0.0) {
        const double y = __builtin_amdgcn_rsq(g);       // ~2^-26 relative
        const double s0 = g * y;
        root = __builtin_fma(__builtin_fma(-s0, s0, g), 0.5 * y, s0);  // one Newton step: ~2^-52
    }
    const double turns = (c_over_l * root) * 0.15915494309189533577;   // theta / 2 pi
    const float fr = (float)(turns - rint(turns));                     // [-0.5, 0.5] revolutions
    const float m = rintf(fr * 4.0f);
    const float r = (fr - m * 0.25f) * 6.28318530717958647692f;        // [-pi/4, pi/4]
    float s, c;
    sincos_poly(r, (int)m & 3, &s, &c);
    return v2f{c, s};
}

// number of LDS twiddle entries a kernel of direction set {fwd, inv} needs
template <int P>
__host__ __device__ constexpr int tw_entries_fwd() { return fft_tw_total<P, -1>() > 0 ? fft_tw_total<P, -1>() : 1; }
template <int P>
__host__ __device__ constexpr int tw_entries_inv() { return fft_tw_total<P, +1>() > 0 ? fft_tw_total<P, +1>() : 1; }

// ---- K1: pack + pad + row FFT ------------------------------------------------------------------
// IN_MODE is a COMPILE-TIME parameter: as a kernel argument every element's fetch sat behind a branch (complex: one 8-byte load,
// amplitude / phase: two 4-byte loads) and hipcc waited vmcnt(0) after each one — a row's loads became E dependent round trips.
template <int P, bool PAD, int IN_MODE>
__global__ __launch_bounds__(kRowThreads) void asm_rows_fwd_kernel(
    const float* __restrict__ in0, const float* __restrict__ in1, long stride0, long stride1,
    float phase_scale, v2f* __restrict__ tbuf, long n_rows) {
    constexpr int in_mode = IN_MODE;
    constexpr int S = PAD ? P / 2 : P;
    constexpr int G = FftGeom<P>::G;       // lanes per row (<= one wave)
    constexpr int SEQ = kRowThreads / G;   // rows per block
    constexpr int E = FftGeom<P>::E;
    constexpr int PITCH = fft_padded_len(P) + 1;
    constexpr int R1 = fft_radix<P, -1>(0);
    constexpr int RL = fft_radix<P, -1>(FftPlan<P>::n - 1);
    constexpr int TWN = tw_entries_fwd<P>();
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    v2f* tw = reinterpret_cast<v2f*>(smem_raw);
    v2f* buf = tw + TWN;

    const int tid = threadIdx.x;
    const int seq = tid / G, t = tid % G;
    const long row = (long)blockIdx.x * SEQ + seq;
    const bool valid = row < n_rows;
    v2f* my = buf + seq * PITCH;
    fft_fill_twiddles<P, -1>(tw, tid, kRowThreads);

    v2f v[E];
    {
        const long f = valid ? row / S : 0;
        const int y = valid ? (int)(row % S) : 0;
        const long base0 = f * stride0 + (long)y * S, base1 = f * stride1 + (long)y * S;
        // element index i = j + r*P/R1 with j < P/R1: the replicate-pad regions (Angular_Spectrum_Method.py:12) are
        // whole r-slices.  Interior elements are loaded and packed first; the two edge values (columns 0 and S-1)
        // are interior elements of the row's first / last lane, so they are broadcast instead of recomputed.
        constexpr bool ONE_BFLY = (E == R1) && (G <= 64);
        v2f edge0 = {0.f, 0.f}, edge1 = edge0;
        if (PAD && !ONE_BFLY) {
            edge0 = load_field(in0, in1, in_mode, base0, base1, phase_scale);
            edge1 = load_field(in0, in1, in_mode, base0 + S - 1, base1 + S - 1, phase_scale);
        }
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const int r = e % R1;
            const bool left = PAD && (r + 1) * 4 <= R1;
            const bool right = PAD && r * 4 >= 3 * R1;
            if (!left && !right) {
                const int i = fft_elem_index<P, G, R1>(t, e);
                const int col = PAD ? i - S / 2 : i;
                v[e] = load_field(in0, in1, in_mode, base0 + col, base1 + col, phase_scale);
            }
        }
        if (PAD && ONE_BFLY) {
            // column 0 = element S/2 = lane 0, slot R1/4;  column S-1 = element 3S/2-1 = lane G-1, slot 3*R1/4-1
            const v2f lo = v[R1 / 4], hi = v[3 * R1 / 4 - 1];
            edge0 = v2f{__shfl(lo.x, 0, G), __shfl(lo.y, 0, G)};
            edge1 = v2f{__shfl(hi.x, G - 1, G), __shfl(hi.y, G - 1, G)};
        }
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const int r = e % R1;
            if (PAD && (r + 1) * 4 <= R1) v[e] = edge0;
            if (PAD && r * 4 >= 3 * R1) v[e] = edge1;
        }
    }
    __syncthreads();  // twiddle table
    fft_run<P, G, -1>(v, my, tw, t);
    if (valid) {
        v2f* dst = tbuf + row * P;
#pragma unroll
        for (int e = 0; e < E; ++e) dst[fft_elem_index<P, G, RL>(t, e)] = v[e];
    }
}

// ---- K2: column FFT . H . column IFFT -----------------------------------------------------------
template <int P>
struct ColCfg {
    // columns per block: 16 x 8 B = one 128-B line per row; fewer only where LDS runs out
    static constexpr int CW = P >= 2048 ? 4 : (P >= 1024 ? 8 : 16);
    static constexpr int G = FftGeom<P>::G;          // lanes per column: one wave (or part of one) owns a column
    static constexpr int E = FftGeom<P>::E;
    static constexpr int THREADS = CW * G;           // 1024 (P=512), 512 (256, 1024), 256 (128, 2048), ...
    static constexpr int PITCH = fft_padded_len(P) + 16 / CW;   // transposed b64 stores spread over the banks
    static constexpr int TWN = tw_entries_fwd<P>() + (fft_plan_symmetric<P>() ? 0 : tw_entries_inv<P>());
    static constexpr size_t LDS = (size_t)(TWN + CW * PITCH) * sizeof(float2);
    // Measured policy: P <= 512 — one tile per block, two blocks per CU (58.5 us at P = 512; persistent variants 65-70 us);
    // P >= 1024 — one PERSISTENT block per CU that loops over tiles and prefetches the next tile's columns into
    // registers while it transforms the current one (+12 % at P = 1024, +11 % at P = 2048).
    static constexpr bool PERSIST = P >= 1024;
    static constexpr int BPC = PERSIST ? 1 : 2;   // blocks per CU the register allocation must allow
    static constexpr int WAVES_PER_SIMD = (BPC * THREADS) / 256 >= 1 ? (BPC * THREADS) / 256 : 1;
};

// One tile (output field x column group) per block: the configuration for P <= 512.
template <int P, bool PAD>
__global__ __launch_bounds__(ColCfg<P>::THREADS, ColCfg<P>::WAVES_PER_SIMD) void asm_cols_kernel(
    const v2f* __restrict__ tbuf, v2f* __restrict__ ubuf, const double* __restrict__ phase_const,
    int n_dist, double inv_ppx, double lam2, double lamb, int ablate) {
    using C = ColCfg<P>;
    constexpr int S = PAD ? P / 2 : P;
    constexpr int CW = C::CW, G = C::G, E = C::E, PITCH = C::PITCH, THREADS = C::THREADS;
    constexpr int R1 = fft_radix<P, -1>(0);
    constexpr int RL = fft_radix<P, -1>(FftPlan<P>::n - 1);  // forward's last = inverse's first radix
    static_assert(fft_radix<P, +1>(0) == RL && fft_radix<P, +1>(FftPlan<P>::n - 1) == R1, "plan symmetry");
    constexpr bool SYM = fft_plan_symmetric<P>();
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    v2f* twf = reinterpret_cast<v2f*>(smem_raw);
    v2f* twi = SYM ? twf : twf + tw_entries_fwd<P>();
    v2f* cols = twf + C::TWN;

    const int tid = threadIdx.x;
    const int fo = blockIdx.y;          // output field = field * n_dist + dist
    const int f = fo / n_dist;
    // Column groups narrower than a 128-B line (CW < 16) share lines with their neighbours.  Blocks are dealt
    // round-robin over the 8 XCDs (private L2s), so give each XCD a contiguous range of groups: the 128/(8*CW)
    // blocks that touch one line then hit the same L2.  Bijective since P/CW is a multiple of 8; speed only.
    int grp = blockIdx.x;
    if (CW < 16 && (P / CW) % 8 == 0) grp = (blockIdx.x % 8) * ((P / CW) / 8) + blockIdx.x / 8;
    const int kx0 = grp * CW;
    fft_fill_twiddles<P, -1>(twf, tid, THREADS);
    if (!SYM) fft_fill_twiddles<P, +1>(twi, tid, THREADS);
    // coalesced read of S rows x CW columns, transposed into per-column sequences (compact: S entries)
    const v2f* src = tbuf + (long)f * S * P + kx0;
    for (int i = tid; i < S * CW; i += THREADS) {
        const int y = i / CW, cl = i % CW;
        cols[cl * PITCH + fft_pad(y)] = src[(long)y * P + cl];
    }
    __syncthreads();
    const int col = tid / G, t = tid % G;
    v2f* my = cols + col * PITCH;
    v2f v[E];
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const int i = fft_elem_index<P, G, R1>(t, e);
        // replicate rows above/below the field are copies of its first/last row spectrum
        int y = PAD ? i - S / 2 : i;
        y = y < 0 ? 0 : (y > S - 1 ? S - 1 : y);
        v[e] = my[fft_pad(y)];
    }
    if (!(ablate & 2)) fft_run<P, G, -1>(v, my, twf, t);   // spectrum in registers at ky = fft_elem_index<P,G,RL>(t,e)

    // transfer function: one value per (column, |ky|), shared through LDS by rows ky and P-ky.  The column's
    // LDS region is private to its G lanes (one wave or less): wave-local ordering, no block barrier.
    fft_sync<G>();
    {
        // squared signed frequencies (k/(P*px))^2: numpy divides by S*px, here k * (1/(P*px)) — 1e-16 relative apart
        const double c_over_l = phase_const[fo] / lamb;
        const int kx = kx0 + col;
        const double fx = (double)(kx < P / 2 ? kx : kx - P) * inv_ppx;
        const double fx2 = fx * fx;
        for (int m = t; m <= P / 2; m += G) {
            const double fy = (double)m * inv_ppx;
            my[m] = (ablate & 1) ? v2f{1.f, 0.f} : transfer_value(fx2 + fy * fy, lam2, c_over_l);
        }
    }
    fft_sync<G>();
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const int ky = fft_elem_index<P, G, RL>(t, e);
        v[e] = cmul(v[e], my[ky <= P / 2 ? ky : P - ky]);
    }
    if (!(ablate & 2)) fft_run<P, G, +1>(v, my, twi, t);

    // keep the S centre rows; transpose back through LDS for a coalesced store
    fft_sync<G>();
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const int i = fft_elem_index<P, G, R1>(t, e);
        const int y = PAD ? i - S / 2 : i;
        if (y >= 0 && y < S) my[fft_pad(y)] = v[e];
    }
    __syncthreads();
    v2f* dst = ubuf + (long)fo * S * P + kx0;
    for (int i = tid; i < S * CW; i += THREADS) {
        const int y = i / CW, cl = i % CW;
        dst[(long)y * P + cl] = cols[cl * PITCH + fft_pad(y)];
    }
}

// A block loops over tiles (output field x column group) with a grid stride; in the persistent configuration
// (ColCfg::PERSIST) it prefetches the NEXT tile's columns into registers while it transforms the current one, so
// the memory phases of a tile overlap the arithmetic of its neighbour.  (At P = 512 the phases of the one-tile
// blocks simply add up — 27 us copy skeleton + 23 us FFT passes + 6.5 us H = 58.5 us, ablation — but the
// persistent form was not faster there.)
// SWEEP (n_dist > 1): a tile is (FIELD, column group); its columns are staged and forward-transformed once and the spectrum kept in
// registers, then  H_d . IFFT . store  runs per distance (see asm_cols_t_kernel).
template <int P, bool PAD, bool SWEEP>
__global__ __launch_bounds__(ColCfg<P>::THREADS, ColCfg<P>::WAVES_PER_SIMD) void asm_cols_persistent_kernel(
    const v2f* __restrict__ tbuf, v2f* __restrict__ ubuf, const double* __restrict__ phase_const,
    int n_dist, int n_tiles, double inv_ppx, double lam2, double lamb, int ablate) {
    using C = ColCfg<P>;
    constexpr int S = PAD ? P / 2 : P;
    constexpr int CW = C::CW, G = C::G, E = C::E, PITCH = C::PITCH, THREADS = C::THREADS;
    constexpr int NGRP = P / CW;                                // column groups per field
    constexpr int LD = (S * CW + THREADS - 1) / THREADS;        // staged elements per thread and tile
    constexpr int R1 = fft_radix<P, -1>(0);
    constexpr int RL = fft_radix<P, -1>(FftPlan<P>::n - 1);  // forward's last = inverse's first radix
    static_assert(fft_radix<P, +1>(0) == RL && fft_radix<P, +1>(FftPlan<P>::n - 1) == R1, "plan symmetry");
    constexpr bool SYM = fft_plan_symmetric<P>();
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    v2f* twf = reinterpret_cast<v2f*>(smem_raw);
    v2f* twi = SYM ? twf : twf + tw_entries_fwd<P>();
    v2f* cols = twf + C::TWN;

    const int tid = threadIdx.x;
    fft_fill_twiddles<P, -1>(twf, tid, THREADS);
    if (!SYM) fft_fill_twiddles<P, +1>(twi, tid, THREADS);
    const int col = tid / G, t = tid % G;
    v2f* my = cols + col * PITCH;

    // tile -> (output field fo = field * n_dist + dist, first column kx0).  Column groups narrower than a 128-B
    // line (CW < 16) share lines with their neighbours; blocks are dealt round-robin over the 8 XCDs (private
    // L2s), so consecutive tile ids of one XCD get adjacent groups.  Bijective since NGRP % 8 == 0; speed only.
    auto tile_fo = [&](int tile) { return tile / NGRP; };
    auto tile_kx0 = [&](int tile) {
        int grp = tile % NGRP;
        if (CW < 16 && NGRP % 8 == 0) grp = (grp % 8) * (NGRP / 8) + grp / 8;
        return grp * CW;
    };

    v2f pre[LD];
    int tile = blockIdx.x;
    if (tile < n_tiles) {
        const v2f* src = tbuf + (long)(SWEEP ? tile_fo(tile) : tile_fo(tile) / n_dist) * S * P + tile_kx0(tile);
#pragma unroll
        for (int l = 0; l < LD; ++l) {
            const int i = tid + l * THREADS;
            if (LD * THREADS == S * CW || i < S * CW) pre[l] = src[(long)(i / CW) * P + (i % CW)];
        }
    }
    for (; tile < n_tiles; tile += gridDim.x) {
        const int fo0 = SWEEP ? tile_fo(tile) * n_dist : tile_fo(tile), kx0 = tile_kx0(tile);
        // staged rows -> per-column sequences (compact: S entries, transposed; coalesced 128-B reads above)
#pragma unroll
        for (int l = 0; l < LD; ++l) {
            const int i = tid + l * THREADS;
            if (LD * THREADS == S * CW || i < S * CW) cols[(i % CW) * PITCH + fft_pad(i / CW)] = pre[l];
        }
        __syncthreads();
        if constexpr (C::PERSIST) {   // prefetch the next tile while this one is transformed
            const int nxt = tile + gridDim.x;
            if (nxt < n_tiles) {
                const v2f* src = tbuf + (long)(SWEEP ? tile_fo(nxt) : tile_fo(nxt) / n_dist) * S * P + tile_kx0(nxt);
#pragma unroll
                for (int l = 0; l < LD; ++l) {
                    const int i = tid + l * THREADS;
                    if (LD * THREADS == S * CW || i < S * CW) pre[l] = src[(long)(i / CW) * P + (i % CW)];
                }
            }
        }
        v2f v[E];
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const int i = fft_elem_index<P, G, R1>(t, e);
            // replicate rows above/below the field are copies of its first/last row spectrum
            int y = PAD ? i - S / 2 : i;
            y = y < 0 ? 0 : (y > S - 1 ? S - 1 : y);
            v[e] = my[fft_pad(y)];
        }
        if (!(ablate & 2)) fft_run<P, G, -1>(v, my, twf, t);   // spectrum in registers at ky = fft_elem_index<P,G,RL>(t,e)

        v2f spec[SWEEP ? E : 1];
        if constexpr (SWEEP) {
#pragma unroll
            for (int e = 0; e < E; ++e) spec[e] = v[e];
        }
        const int d_count = SWEEP ? n_dist : 1;
        for (int d = 0; d < d_count; ++d) {
            const int fo = fo0 + d;
            if (SWEEP && d > 0) {
                __syncthreads();   // the previous distance's rows have left `cols`
#pragma unroll
                for (int e = 0; e < E; ++e) v[e] = spec[e];
            }
            // transfer function: one value per (column, |ky|), shared through LDS by rows ky and P-ky.  The column's
            // LDS region is private to its G lanes (one wave or less): wave-local ordering, no block barrier.
            fft_sync<G>();
            {
                // squared signed frequencies (k/(P*px))^2: numpy divides by S*px, here k * (1/(P*px)) — 1e-16 relative apart
                const double c_over_l = phase_const[fo] / lamb;
                const int kx = kx0 + col;
                const double fx = (double)(kx < P / 2 ? kx : kx - P) * inv_ppx;
                const double fx2 = fx * fx;
                for (int m = t; m <= P / 2; m += G) {
                    const double fy = (double)m * inv_ppx;
                    my[m] = (ablate & 1) ? v2f{1.f, 0.f} : transfer_value(fx2 + fy * fy, lam2, c_over_l);
                }
            }
            fft_sync<G>();
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const int ky = fft_elem_index<P, G, RL>(t, e);
                v[e] = cmul(v[e], my[ky <= P / 2 ? ky : P - ky]);
            }
            if (!(ablate & 2)) fft_run<P, G, +1>(v, my, twi, t);

            // keep the S centre rows; transpose back through LDS for a coalesced store
            fft_sync<G>();
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const int i = fft_elem_index<P, G, R1>(t, e);
                const int y = PAD ? i - S / 2 : i;
                if (y >= 0 && y < S) my[fft_pad(y)] = v[e];
            }
            __syncthreads();
            v2f* dst = ubuf + (long)fo * S * P + kx0;
            for (int i = tid; i < S * CW; i += THREADS) {
                const int y = i / CW, cl = i % CW;
                dst[(long)y * P + cl] = cols[cl * PITCH + fft_pad(y)];
            }
        }
        __syncthreads();   // the next tile's staging overwrites `cols`
    }
}

// ---- K3: row IFFT + crop + epilogue ---------------------------------------------------------------
template <int P, bool PAD, int OUT_MODE>   // OUT_MODE at compile time: the epilogue is straight-line code per element
__global__ __launch_bounds__(kRowThreads) void asm_rows_inv_kernel(
    const v2f* __restrict__ ubuf, float* __restrict__ out0, float* __restrict__ out1,
    long ostride0, long ostride1, long n_rows, float scale) {
    constexpr int out_mode = OUT_MODE;
    constexpr int S = PAD ? P / 2 : P;
    constexpr int G = FftGeom<P>::G;
    constexpr int SEQ = kRowThreads / G;
    constexpr int E = FftGeom<P>::E;
    constexpr int PITCH = fft_padded_len(P) + 1;
    constexpr int R1 = fft_radix<P, +1>(0);
    constexpr int RL = fft_radix<P, +1>(FftPlan<P>::n - 1);
    constexpr int TWN = tw_entries_inv<P>();
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    v2f* tw = reinterpret_cast<v2f*>(smem_raw);
    v2f* buf = tw + TWN;

    const int tid = threadIdx.x;
    const int seq = tid / G, t = tid % G;
    const long row = (long)blockIdx.x * SEQ + seq;
    const bool valid = row < n_rows;
    v2f* my = buf + seq * PITCH;
    fft_fill_twiddles<P, +1>(tw, tid, kRowThreads);
    v2f v[E];
    {
        const v2f* src = ubuf + (valid ? row : 0) * P;
#pragma unroll
        for (int e = 0; e < E; ++e) v[e] = src[fft_elem_index<P, G, R1>(t, e)];
    }
    __syncthreads();
    fft_run<P, G, +1>(v, my, tw, t);
    if (valid) {
        const long fo = row / S;
        const int y = (int)(row % S);
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const int r = e % RL;
            // centre crop (Angular_Spectrum_Method.py:34): with padding only the middle r-slices survive
            if (PAD && ((r + 1) * 4 <= RL || r * 4 >= 3 * RL)) continue;
            const int i = fft_elem_index<P, G, RL>(t, e);
            const int x = PAD ? i - S / 2 : i;
            const v2f o = v[e] * scale;
            const long pix = (long)y * S + x;
            if (out_mode == HOLO_ASM_OUT_AMP_PHASE) {
                out0[fo * ostride0 + pix] = amp_f32(o);
                out1[fo * ostride1 + pix] = atan2_poly(o.y, o.x);
            } else if (out_mode == HOLO_ASM_OUT_INTENSITY) {
                const float a = amp_f32(o);  // Forward_model.py:39: pow(abs(.),2)
                out0[fo * ostride0 + pix] = a * a;
            } else {
                reinterpret_cast<v2f*>(out0)[fo * ostride0 + pix] = o;
            }
        }
    }
}


// ================================================================================================================
// Transposed-intermediate kernels (32 <= P <= 1024, S a multiple of 16): T and U are [field][kx][y].
// ================================================================================================================
#ifndef HOLO_ASM_ABL_CONTIG
#define HOLO_ASM_ABL_CONTIG 0
#endif
#ifndef HOLO_ASM_PERSIST_FWD
#define HOLO_ASM_PERSIST_FWD 1   // forward row kernel of the transposed path: persistent blocks that prefetch the next tile's raw inputs
#endif
#ifndef HOLO_ASM_PERSIST_ROWS
#define HOLO_ASM_PERSIST_ROWS 1   // inverse row kernel of the transposed path at P >= 1024: persistent blocks that prefetch the next tile
#endif
#ifndef HOLO_ASM_TILE_ROWS
#define HOLO_ASM_TILE_ROWS 16
#endif
constexpr int kTileRows = HOLO_ASM_TILE_ROWS;   // rows per row-kernel block: 16 x 8 B = one 128-B run per kx (8: 64-B runs, smaller blocks)
constexpr int kTilePitch = kTileRows + 1;  // LDS tile [kx][17]: conflict-free for 16 consecutive kx at a fixed row

template <int P>
struct RowTCfg {
    static constexpr int G = FftGeom<P>::G, E = FftGeom<P>::E;
    static constexpr int THREADS = kTileRows * G;                 // one row per G lanes
    static constexpr int PITCH = fft_padded_len(P) + 1;
    static constexpr int BUF = kTileRows * PITCH;                 // FFT buffers; the [P][17] tile aliases them
    static_assert(BUF >= P * kTilePitch, "the transpose tile must fit the FFT buffers it aliases");
    static constexpr size_t lds(int twn) { return (size_t)(twn + BUF) * sizeof(float2); }
    static constexpr int bpc(int twn) {                           // blocks per CU that LDS and the 32-wave limit allow
        const int by_lds = (int)((160 * 1024) / lds(twn)), by_waves = 2048 / THREADS;
        const int b = by_lds < by_waves ? by_lds : by_waves;
        return b < 1 ? 1 : (b > 4 ? 4 : b);
    }
    static constexpr int waves_per_simd(int twn) { return (bpc(twn) * THREADS + 255) / 256; }
};

// K1: pack + pad + row FFT of 16 consecutive rows of one field, written transposed
// ---- transfer-function tables -----------------------------------------------------------------------------------
// H depends on (|kx|, |ky|) and on the output field's distance only, and a batch usually shares ONE distance (the entry
// loop back-propagates every hologram by the style distance): evaluating it per column in the column kernel (float64
// argument, ~330 instructions per thread) was 39 % of that kernel's VALU work.  Output fields fo < kHSlots get a table slot;
// fo's slot is uid(fo) = the first g < min(fo, kHSlots) with a bit-identical phase constant, else fo itself (fo >= kHSlots
// without a match: no table, H is evaluated in the column kernel as before).  Only slots that represent themselves are
// filled — by EXTRA BLOCKS of the forward row kernel, which is memory-bound and has the VALU to spare, so the table costs
// no launch and no time on the critical path.  Layout [slot][|kx|][|ky|] (P/2+1 squared, complex64): a column's values are
// contiguous.  Values are produced by the same transfer_value(): results are bit-identical to the inline evaluation.
constexpr int kHSlots = 64;
template <int P>
__host__ __device__ constexpr size_t htab_elems() { return (size_t)(P / 2 + 1) * (P / 2 + 1); }

// slot of output field fo (or -1), evaluated by the first wave of a block: lanes compare the phase constants of the first
// kHSlots fields with fo's
__device__ __forceinline__ int htab_uid_wave(const double* __restrict__ phase_const, int fo, int lane) {
    static_assert(kHSlots == 64, "one lane per candidate");
    const double mine = phase_const[fo];
    const bool eq = lane < fo && phase_const[lane < fo ? lane : 0] == mine;
    const unsigned long long m = __ballot(eq);
    return m ? __ffsll((long long)m) - 1 : (fo < kHSlots ? fo : -1);
}

template <int P, bool PAD, int IN_MODE>
__global__ __launch_bounds__(RowTCfg<P>::THREADS, RowTCfg<P>::waves_per_simd(tw_entries_fwd<P>())) void asm_rows_fwd_t_kernel(
    const float* __restrict__ in0, const float* __restrict__ in1, long stride0, long stride1,
    float phase_scale, v2f* __restrict__ tbuf, int tab_blocks, v2f* __restrict__ htab,
    const double* __restrict__ phase_const, int n_out, double inv_ppx, double lam2, double lamb, int n_row_tiles) {
    using C = RowTCfg<P>;
    constexpr int in_mode = IN_MODE;   // compile-time: see asm_rows_fwd_kernel
    if ((int)blockIdx.x < tab_blocks) {
        // table blocks (dispatched FIRST, so that they run beside the row blocks instead of forming a tail): block j computes
        // entries [j*THREADS, (j+1)*THREADS) of EVERY slot that represents itself (usually one).  The constants of the slots
        // go through LDS once; slot s represents itself unless an earlier slot holds the same bits.  (One block per
        // (slot, piece) would be 64x more blocks, nearly all exiting at once: +4 us on this kernel, measured.)
        constexpr int HN = P / 2 + 1;
        __shared__ double s_pc[kHSlots];
        __shared__ int s_dup[kHSlots];
        const int slots = n_out < kHSlots ? n_out : kHSlots;
        const int tid = threadIdx.x;
        if (tid < kHSlots) {
            s_pc[tid] = phase_const[tid < slots ? tid : 0];
            s_dup[tid] = 0;
        }
        __syncthreads();
        for (int q = tid; q < kHSlots * kHSlots; q += C::THREADS) {
            const int sl = q / kHSlots, g = q % kHSlots;
            if (g < sl && sl < slots && s_pc[g] == s_pc[sl]) s_dup[sl] = 1;   // benign race: every writer stores 1
        }
        __syncthreads();
        const int i = (int)blockIdx.x * C::THREADS + tid;
        if (i >= (int)htab_elems<P>()) return;
        const int akx = i / HN, aky = i % HN;
        const double fx = (double)akx * inv_ppx, fy = (double)aky * inv_ppx;
        const double f2 = fx * fx + fy * fy;
        for (int slot = 0; slot < slots; ++slot)
            if (!s_dup[slot]) htab[(size_t)slot * htab_elems<P>() + i] = transfer_value(f2, lam2, s_pc[slot] / lamb);
        return;
    }
    constexpr int S = PAD ? P / 2 : P;
    constexpr int G = C::G, E = C::E, PITCH = C::PITCH, THREADS = C::THREADS;
    constexpr int R1 = fft_radix<P, -1>(0);
    constexpr int RL = fft_radix<P, -1>(FftPlan<P>::n - 1);
    constexpr int TWN = tw_entries_fwd<P>();
    static_assert(S % kTileRows == 0, "a block's 16 rows must belong to one field");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    v2f* tw = reinterpret_cast<v2f*>(smem_raw);
    v2f* buf = tw + TWN;

    const int tid = threadIdx.x;
    const int seq = tid / G, t = tid % G;
    v2f* my = buf + seq * PITCH;

    // PERSISTENT over 16-row tiles (round 3): a block keeps its twiddle table and, before it transforms a tile, requests the
    // RAW inputs of its next one (amplitude / phase or re / im: E interior values + two edges per lane) into registers.
    // Ablation (HOLO_ASM_ABLATE, 64 x 256^2): of this kernel's 26 us, 14.5 vanish with the input loads and 16 with the
    // transposing store although each moves its bytes in ~3 and ~10 us — one-tile blocks wait out the full load latency at
    // their start, two lockstep blocks per CU, two rounds.
    // element index i = j + r*P/R1 with j < P/R1: the replicate-pad regions (Angular_Spectrum_Method.py:12) are whole
    // r-slices.  The two edge values (columns 0 and S-1) are interior elements of the row's first / last lane where the
    // row is one butterfly per lane (broadcast instead of loaded).
    constexpr bool ONE_BFLY = (E == R1) && (G <= 64);
    constexpr int NRAW = E + 2;                      // raw slots: interior elements at their e, edges at E, E + 1
    auto load_raw = [&](int tile, v2f (&raw)[NRAW]) {
        const long row0 = (long)tile * kTileRows;
        const long f = row0 / S;
        const int y = (int)(row0 % S) + seq;
        const long base0 = f * stride0 + (long)y * S, base1 = f * stride1 + (long)y * S;
        auto fetch = [&](long col) {
            if (in_mode == HOLO_ASM_IN_COMPLEX) return reinterpret_cast<const v2f*>(in0)[base0 + col];
            return v2f{in0[base0 + col], in1[base1 + col]};
        };
        if (PAD && !ONE_BFLY) {
            raw[E] = fetch(0);
            raw[E + 1] = fetch(S - 1);
        }
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const int r = e % R1;
            const bool left = PAD && (r + 1) * 4 <= R1;
            const bool right = PAD && r * 4 >= 3 * R1;
            if (!left && !right) {
                const int i = fft_elem_index<P, G, R1>(t, e);
                raw[e] = ASM_ABL(32) ? v2f{(float)tid, (float)e} : fetch(PAD ? i - S / 2 : i);
            }
        }
    };
    // Forward_model.py:20-22: phase*phase_normalize, then amplitude*exp(1j*phase) in complex64
    auto pack = [&](v2f r) {
        if (in_mode == HOLO_ASM_IN_COMPLEX || ASM_ABL(4)) return r;
        float sn, cs;
        sincos_f32(r.y * phase_scale, &sn, &cs);
        return v2f{r.x * cs, r.x * sn};
    };
    const int n_tiles = (int)gridDim.x - tab_blocks >= 0 ? n_row_tiles : 0;
    int tile = (int)blockIdx.x - tab_blocks;
    const int stride_t = (int)gridDim.x - tab_blocks;
    v2f raw[NRAW];
    if (tile < n_tiles) load_raw(tile, raw);
    fft_fill_twiddles<P, -1>(tw, tid, THREADS);   // behind the first tile's loads
    __syncthreads();  // twiddle table
    for (; tile < n_tiles; tile += stride_t) {
        v2f v[E];
        {
            v2f edge0 = {0.f, 0.f}, edge1 = edge0;
            if (PAD && !ONE_BFLY) {
                edge0 = pack(raw[E]);
                edge1 = pack(raw[E + 1]);
            }
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const int r = e % R1;
                const bool left = PAD && (r + 1) * 4 <= R1;
                const bool right = PAD && r * 4 >= 3 * R1;
                if (!left && !right) v[e] = pack(raw[e]);
            }
            if (PAD && ONE_BFLY) {
                const v2f lo = v[R1 / 4], hi = v[3 * R1 / 4 - 1];
                edge0 = v2f{__shfl(lo.x, 0, G), __shfl(lo.y, 0, G)};
                edge1 = v2f{__shfl(hi.x, G - 1, G), __shfl(hi.y, G - 1, G)};
            }
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const int r = e % R1;
                if (PAD && (r + 1) * 4 <= R1) v[e] = edge0;
                if (PAD && r * 4 >= 3 * R1) v[e] = edge1;
            }
        }
        if (tile + stride_t < n_tiles) load_raw(tile + stride_t, raw);   // in flight during this tile's transform and store
        const long row0 = (long)tile * kTileRows;
        const long f = row0 / S;
        const int y0 = (int)(row0 % S);
        if (!ASM_ABL(8)) fft_run<P, G, -1>(v, my, tw, t);
        if (ASM_ABL(16)) {
            if (tid == 0) tbuf[((long)f * P) * S + y0] = v[0] + v[1] + v[2] + v[3] + v[4] + v[5] + v[6] + v[7];
            continue;
        }
        __syncthreads();  // every row is done with its FFT buffer: the tile aliases them
#pragma unroll
        for (int e = 0; e < E; ++e) buf[fft_elem_index<P, G, RL>(t, e) * kTilePitch + seq] = v[e];
        __syncthreads();
        // T^T[f][kx][y0 .. y0+15]: one 128-B run per kx, 16 B per thread
        v2f* dst = tbuf + ((long)f * P) * S + y0;
        for (int q = tid; q < P * (kTileRows / 2); q += THREADS) {
            const int kx = q / (kTileRows / 2), j = q % (kTileRows / 2);
            const v2f a = buf[kx * kTilePitch + 2 * j], b = buf[kx * kTilePitch + 2 * j + 1];
            *reinterpret_cast<float4*>(dst + (long)kx * S + 2 * j) = make_float4(a.x, a.y, b.x, b.y);
        }
        __syncthreads();  // the tile is read out: the next transform writes the FFT buffers it aliases
    }
}

template <int P>
struct ColTCfg {
    static constexpr int G = FftGeom<P>::G, E = FftGeom<P>::E;
    // columns are independent (no staging, no block barrier after the twiddle fill): small blocks, many per CU
    static constexpr int THREADS = P * G < 256 ? P * G : 256;
    static constexpr int CW = THREADS / G;
    static constexpr int PITCH = fft_padded_len(P) + 1;
    static constexpr int TWN = tw_entries_fwd<P>() + (fft_plan_symmetric<P>() ? 0 : tw_entries_inv<P>());
    static constexpr size_t LDS = (size_t)(TWN + CW * PITCH) * sizeof(float2);
    static constexpr int BPC_LDS = (int)((160 * 1024) / LDS);
    static constexpr int BPC = BPC_LDS * THREADS > 2048 ? 2048 / THREADS : BPC_LDS;
    static constexpr int WPS0 = (BPC * THREADS) / 256;
    static constexpr int WAVES_PER_SIMD = WPS0 < 1 ? 1 : (WPS0 > 8 ? 8 : WPS0);
    static_assert(P % CW == 0, "column groups must tile the transform side");
};

// K2: one column per G lanes, global -> registers -> global
// SWEEP (n_dist > 1; the reference idiom repeats a field against a list of distances, utils/Data_loader.py:59-62): a block owns a
// FIELD's columns, loads and forward-transforms them ONCE, keeps the spectrum in registers and runs  H_d . IFFT . store  per
// distance — instead of n_dist loads and forward transforms of the same column.  Same instruction sequence per output: bit-identical
// to the one-distance form.
template <int P, bool PAD, bool SWEEP>
__global__ __launch_bounds__(ColTCfg<P>::THREADS, SWEEP ? (ColTCfg<P>::WAVES_PER_SIMD > 4 ? 4 : ColTCfg<P>::WAVES_PER_SIMD) : ColTCfg<P>::WAVES_PER_SIMD) void asm_cols_t_kernel(
    const v2f* __restrict__ tbuf, v2f* __restrict__ ubuf, const double* __restrict__ phase_const,
    int n_dist, double inv_ppx, double lam2, double lamb, int ablate, const v2f* __restrict__ htab, int n_tiles) {
    using C = ColTCfg<P>;
    constexpr int S = PAD ? P / 2 : P;
    constexpr int CW = C::CW, G = C::G, E = C::E, PITCH = C::PITCH, THREADS = C::THREADS;
    constexpr int R1 = fft_radix<P, -1>(0);
    constexpr int RL = fft_radix<P, -1>(FftPlan<P>::n - 1);  // forward's last = inverse's first radix
    static_assert(fft_radix<P, +1>(0) == RL && fft_radix<P, +1>(FftPlan<P>::n - 1) == R1, "plan symmetry");
    constexpr bool SYM = fft_plan_symmetric<P>();
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    v2f* twf = reinterpret_cast<v2f*>(smem_raw);
    v2f* twi = SYM ? twf : twf + tw_entries_fwd<P>();
    v2f* cols = twf + C::TWN;

    const int tid = threadIdx.x;
    const int col = tid / G, t = tid % G;
    v2f* my = cols + col * PITCH;
    fft_fill_twiddles<P, -1>(twf, tid, THREADS);
    if (!SYM) fft_fill_twiddles<P, +1>(twi, tid, THREADS);
    __syncthreads();  // twiddle tables: the only block barrier
    // PERSISTENT over tiles (round 4): tile = (output field | field of a sweep, column group).  A block fills its twiddle tables
    // once (two sincospif per thread: ~15 % of a one-tile block's vector instructions) and walks tiles with the grid's stride; every
    // wave is on its own after the barrier above (wave-local exchanges, per-wave table slot).
    constexpr int NG = P / CW;
    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const int fy = tile / NG;           // one-distance form: output field (= field * n_dist + dist); SWEEP: field
    const int f = SWEEP ? fy : fy / n_dist;
    const int kx = (tile % NG) * CW + col;
    // the column's S values are contiguous; rows above/below the field are copies of its first/last row spectrum
    const v2f* src = tbuf + ((long)f * P + kx) * S;
    v2f v[E];
    {
        v2f edge0 = {0.f, 0.f}, edge1 = edge0;
        if (PAD) {
            edge0 = src[0];
            edge1 = src[S - 1];
        }
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const int r = e % R1;
            const bool left = PAD && (r + 1) * 4 <= R1;
            const bool right = PAD && r * 4 >= 3 * R1;
            if (left) v[e] = edge0;
            else if (right) v[e] = edge1;
            else {
                const int i = fft_elem_index<P, G, R1>(t, e);
                v[e] = src[PAD ? i - S / 2 : i];
            }
        }
    }
    if (!(ablate & 2)) fft_run<P, G, -1>(v, my, twf, t);   // spectrum in registers at ky = fft_elem_index<P,G,RL>(t,e)

    v2f spec[SWEEP ? E : 1];
    if constexpr (SWEEP) {
#pragma unroll
        for (int e = 0; e < E; ++e) spec[e] = v[e];
    }
    const int d_count = SWEEP ? n_dist : 1;
    for (int d = 0; d < d_count; ++d) {
        const int fo = SWEEP ? f * n_dist + d : fy;
        // every wave finds the table slot itself (a wave is 64 lanes here: THREADS >= 64 wherever a table exists)
        int uid = (THREADS >= 64 && htab != nullptr) ? htab_uid_wave(phase_const, fo, tid & 63) : -1;
        uid = __builtin_amdgcn_readfirstlane(uid);
        if (SWEEP && d > 0) {
            fft_sync<G>();    // the inverse transform of the previous distance is done with the column's LDS region
#pragma unroll
            for (int e = 0; e < E; ++e) v[e] = spec[e];
        }
        if (uid >= 0) {
            // transfer function from this distance's table: the column's |ky| values are contiguous
            const v2f* __restrict__ hcol = htab + (size_t)uid * htab_elems<P>() + (size_t)(kx <= P / 2 ? kx : P - kx) * (P / 2 + 1);
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const int ky = fft_elem_index<P, G, RL>(t, e);
                v[e] = cmul(v[e], hcol[ky <= P / 2 ? ky : P - ky]);
            }
        } else {
            // no table slot (more than kHSlots distinct distances in the call): one value per (column, |ky|), shared through
            // LDS by rows ky and P-ky.  The column's LDS region is private to its G lanes: wave-local ordering, no block barrier.
            fft_sync<G>();
            {
                // squared signed frequencies (k/(P*px))^2: numpy divides by S*px, here k * (1/(P*px)) — 1e-16 relative apart
                const double c_over_l = phase_const[fo] / lamb;
                const double fx = (double)(kx <= P / 2 ? kx : P - kx) * inv_ppx;
                const double fx2 = fx * fx;
                for (int m = t; m <= P / 2; m += G) {
                    const double fy = (double)m * inv_ppx;
                    my[m] = (ablate & 1) ? v2f{1.f, 0.f} : transfer_value(fx2 + fy * fy, lam2, c_over_l);
                }
            }
            fft_sync<G>();
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const int ky = fft_elem_index<P, G, RL>(t, e);
                v[e] = cmul(v[e], my[ky <= P / 2 ? ky : P - ky]);
            }
        }
        if (!(ablate & 2)) fft_run<P, G, +1>(v, my, twi, t);

        // keep the S centre rows: contiguous runs of G values per register slot
        v2f* dst = ubuf + ((long)fo * P + kx) * S;
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const int r = e % R1;
            if (PAD && ((r + 1) * 4 <= R1 || r * 4 >= 3 * R1)) continue;
            const int i = fft_elem_index<P, G, R1>(t, e);
            dst[PAD ? i - S / 2 : i] = v[e];
        }
    }
    fft_sync<G>();   // this column's LDS region is reused by the next tile's transforms
    }
}

// K3: 16 rows of one output field: transposed load, row IFFT, crop, epilogue
template <int P, bool PAD, int OUT_MODE>
__global__ __launch_bounds__(RowTCfg<P>::THREADS, RowTCfg<P>::waves_per_simd(tw_entries_inv<P>())) void asm_rows_inv_t_kernel(
    const v2f* __restrict__ ubuf, float* __restrict__ out0, float* __restrict__ out1,
    long ostride0, long ostride1, float scale, int n_tiles) {
    using C = RowTCfg<P>;
    constexpr int out_mode = OUT_MODE;
    constexpr int S = PAD ? P / 2 : P;
    constexpr int G = C::G, E = C::E, PITCH = C::PITCH, THREADS = C::THREADS;
    constexpr int R1 = fft_radix<P, +1>(0);
    constexpr int RL = fft_radix<P, +1>(FftPlan<P>::n - 1);
    constexpr int TWN = tw_entries_inv<P>();
    constexpr int LD = P * (kTileRows / 2) / THREADS;   // 16-byte pieces of a tile per thread (= E/2)
    static_assert(LD * THREADS == P * (kTileRows / 2), "the tile divides evenly over the block");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    v2f* tw = reinterpret_cast<v2f*>(smem_raw);
    v2f* buf = tw + TWN;

    const int tid = threadIdx.x;
    const int seq = tid / G, t = tid % G;
    v2f* my = buf + seq * PITCH;
    fft_fill_twiddles<P, +1>(tw, tid, THREADS);
    // The grid may be persistent (P >= 1024: one block per CU): the NEXT tile's 16-byte pieces are then requested into
    // registers before the current tile's transform and epilogue, so the memory pipe stays busy across the block's
    // barrier-separated phases.  With one tile per block (P <= 512) the loop runs once.
    auto load_tile = [&](int tile, float4 (&pre)[LD]) {
        const long row0 = (long)tile * kTileRows;
        const v2f* src = ubuf + (row0 / S) * P * S + (row0 % S);
#if HOLO_ASM_ABL_CONTIG
        src = ubuf + (row0 / S) * P * S + ((row0 % S) / kTileRows) * (long)(P * kTileRows);
#endif
#pragma unroll
        for (int l = 0; l < LD; ++l) {
            const int q = tid + l * THREADS;
#if HOLO_ASM_ABL_CONTIG
            pre[l] = *reinterpret_cast<const float4*>(src + (long)q * 2);
#else
            if (ASM_ABL(64)) pre[l] = make_float4((float)q, 1.f, 2.f, 3.f);
            else pre[l] = *reinterpret_cast<const float4*>(src + (long)(q / (kTileRows / 2)) * S + 2 * (q % (kTileRows / 2)));
#endif
        }
    };
    float4 pre[LD];
    int tile = blockIdx.x;
    if (tile < n_tiles) load_tile(tile, pre);
    for (; tile < n_tiles; tile += gridDim.x) {
        // U^T[fo][kx][y0 .. y0+15] -> tile[kx][row]
#pragma unroll
        for (int l = 0; l < LD; ++l) {
            const int q = tid + l * THREADS;
            const int kx = q / (kTileRows / 2), j = q % (kTileRows / 2);
            buf[kx * kTilePitch + 2 * j] = v2f{pre[l].x, pre[l].y};
            buf[kx * kTilePitch + 2 * j + 1] = v2f{pre[l].z, pre[l].w};
        }
        __syncthreads();   // (also covers the twiddle table on the first tile)
        v2f v[E];
#pragma unroll
        for (int e = 0; e < E; ++e) v[e] = buf[fft_elem_index<P, G, R1>(t, e) * kTilePitch + seq];
        __syncthreads();  // the tile is consumed: its space becomes the rows' FFT buffers
        if (tile + (int)gridDim.x < n_tiles) load_tile(tile + gridDim.x, pre);
        if (!ASM_ABL(128)) fft_run<P, G, +1>(v, my, tw, t);
        {
            const long row0 = (long)tile * kTileRows;
            const long fo = row0 / S;
            const int y = (int)(row0 % S) + seq;
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const int r = e % RL;
                // centre crop (Angular_Spectrum_Method.py:34): with padding only the middle r-slices survive
                if (PAD && ((r + 1) * 4 <= RL || r * 4 >= 3 * RL)) continue;
                const int i = fft_elem_index<P, G, RL>(t, e);
                const int x = PAD ? i - S / 2 : i;
                const v2f o = v[e] * scale;
                const long pix = (long)y * S + x;
                if (ASM_ABL(512)) {
                    if (o.x == 12345.678f) out0[pix] = o.y;
                } else if (ASM_ABL(256)) {
                    out0[fo * ostride0 + pix] = o.x;
                    out1[fo * ostride1 + pix] = o.y;
                } else if (out_mode == HOLO_ASM_OUT_AMP_PHASE) {
                    out0[fo * ostride0 + pix] = amp_f32(o);
                    out1[fo * ostride1 + pix] = atan2_poly(o.y, o.x);
                } else if (out_mode == HOLO_ASM_OUT_INTENSITY) {
                    const float a = amp_f32(o);  // Forward_model.py:39: pow(abs(.),2)
                    out0[fo * ostride0 + pix] = a * a;
                } else {
                    reinterpret_cast<v2f*>(out0)[fo * ostride0 + pix] = o;
                }
            }
        }
        __syncthreads();   // every row is done with its FFT buffer before the next tile is written over them
    }
}

// Phase constants c of theta = c * G for float32 distances, with the reference's rounding chain in ONE launch
// (Forward_model.py:18 + Angular_Spectrum_Method.py:29):
//   c = double( fl32( fl32(2 pi) * fl32( fl32( fl32(d + add) * mul ) * fl32(1e-3) ) ) )
// — every step a separately rounded fp32 operation, as the torch element-wise ops they replace (five small aten launches in
// front of every propagation).  out[i] takes d[(i % period) * stride]: period = batch for a per-sample distance, the number
// of distances for a sweep.
__global__ __launch_bounds__(256) void phase_constants_kernel(const float* __restrict__ d, long stride, int period, float add,
                                                             float mul, double* __restrict__ out, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float t = __fadd_rn(d[(long)(i % period) * stride], add);
    t = __fmul_rn(t, mul);
    t = __fmul_rn(t, 1e-3f);
    t = __fmul_rn(t, 6.28318530717958647692f);     // fl32(2 pi)
    out[i] = (double)t;
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
    float2 *twf, *twi;
    double* f2;
    float2 *tbuf, *ubuf;
    float2* htab;   // transfer-function tables (transposed-layout path), or nullptr
    hipStream_t stream;
    int ablate;   // -DHOLO_DEBUG builds only (HOLO_ASM_ABLATE): 1 = skip H evaluation, 2 = skip the FFT passes; 0 in the shipped build
};

template <int P, bool PAD>
static int launch_asm(const AsmArgs& a) {
    using C = ColCfg<P>;
    constexpr int S = PAD ? P / 2 : P;
    constexpr int G = FftGeom<P>::G;
    constexpr int SEQ = kRowThreads / G;
    const size_t row_lds_f = (size_t)(tw_entries_fwd<P>() + SEQ * (fft_padded_len(P) + 1)) * sizeof(float2);
    const size_t row_lds_i = (size_t)(tw_entries_inv<P>() + SEQ * (fft_padded_len(P) + 1)) * sizeof(float2);
    const size_t col_lds = C::LDS;
    static DeviceOnce attr_once;  // per device, published after the attributes are set
    attr_once.run([&] {
        if (col_lds > 48 * 1024)
        {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&asm_cols_kernel<P, PAD>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)col_lds);
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&asm_cols_persistent_kernel<P, PAD, true>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)col_lds);
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&asm_cols_persistent_kernel<P, PAD, false>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)col_lds);
        }
        if (row_lds_f > 48 * 1024) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&asm_rows_fwd_kernel<P, PAD, HOLO_ASM_IN_AMP_PHASE>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)row_lds_f);
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&asm_rows_fwd_kernel<P, PAD, HOLO_ASM_IN_COMPLEX>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)row_lds_f);
        }
        if (row_lds_i > 48 * 1024) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&asm_rows_inv_kernel<P, PAD, HOLO_ASM_OUT_AMP_PHASE>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)row_lds_i);
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&asm_rows_inv_kernel<P, PAD, HOLO_ASM_OUT_INTENSITY>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)row_lds_i);
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&asm_rows_inv_kernel<P, PAD, HOLO_ASM_OUT_COMPLEX>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)row_lds_i);
        }
    });
    const long rows_in = (long)a.n_fields * S;
    if (a.in_mode == HOLO_ASM_IN_COMPLEX)
        hipLaunchKernelGGL((asm_rows_fwd_kernel<P, PAD, HOLO_ASM_IN_COMPLEX>), dim3((unsigned)((rows_in + SEQ - 1) / SEQ)), dim3(kRowThreads),
                           row_lds_f, a.stream, a.in0, a.in1, a.s0, a.s1, a.phase_scale, reinterpret_cast<v2f*>(a.tbuf), rows_in);
    else
        hipLaunchKernelGGL((asm_rows_fwd_kernel<P, PAD, HOLO_ASM_IN_AMP_PHASE>), dim3((unsigned)((rows_in + SEQ - 1) / SEQ)), dim3(kRowThreads),
                           row_lds_f, a.stream, a.in0, a.in1, a.s0, a.s1, a.phase_scale, reinterpret_cast<v2f*>(a.tbuf), rows_in);
    {
        const long n_tiles = (long)(P / C::CW) * a.n_fields * a.n_dist;
        if (C::PERSIST && a.n_dist > 1) {   // a sweep: tiles are (field, column group), the distances looped inside on the shared spectrum
            const long f_tiles = (long)(P / C::CW) * a.n_fields;
            const unsigned grid = (unsigned)(f_tiles < 256 ? f_tiles : 256);
            hipLaunchKernelGGL((asm_cols_persistent_kernel<P, PAD, true>), dim3(grid), dim3(C::THREADS), col_lds, a.stream,
                               reinterpret_cast<const v2f*>(a.tbuf), reinterpret_cast<v2f*>(a.ubuf), a.phase_const,
                               a.n_dist, (int)f_tiles, 1.0 / a.ppx, a.lam2, a.lamb, a.ablate);
        } else if (C::PERSIST) {   // one block per CU, grid-stride over the tiles
            const unsigned grid = (unsigned)(n_tiles < 256 ? n_tiles : 256);
            hipLaunchKernelGGL((asm_cols_persistent_kernel<P, PAD, false>), dim3(grid), dim3(C::THREADS), col_lds, a.stream,
                               reinterpret_cast<const v2f*>(a.tbuf), reinterpret_cast<v2f*>(a.ubuf), a.phase_const,
                               a.n_dist, (int)n_tiles, 1.0 / a.ppx, a.lam2, a.lamb, a.ablate);
        } else {
            hipLaunchKernelGGL((asm_cols_kernel<P, PAD>), dim3(P / C::CW, a.n_fields * a.n_dist), dim3(C::THREADS), col_lds,
                               a.stream, reinterpret_cast<const v2f*>(a.tbuf), reinterpret_cast<v2f*>(a.ubuf), a.phase_const,
                               a.n_dist, 1.0 / a.ppx, a.lam2, a.lamb, a.ablate);
        }
    }
    const long rows_out = rows_in * a.n_dist;
    const float scale = 1.0f / ((float)P * (float)P);  // ifft2 "backward" normalisation; exact (power of two)
#define HOLO_LAUNCH_INV(mode_)                                                                                              \
    hipLaunchKernelGGL((asm_rows_inv_kernel<P, PAD, mode_>), dim3((unsigned)((rows_out + SEQ - 1) / SEQ)), dim3(kRowThreads), \
                       row_lds_i, a.stream, reinterpret_cast<const v2f*>(a.ubuf), a.out0, a.out1, a.os0, a.os1, rows_out, scale)
    if (a.out_mode == HOLO_ASM_OUT_AMP_PHASE) HOLO_LAUNCH_INV(HOLO_ASM_OUT_AMP_PHASE);
    else if (a.out_mode == HOLO_ASM_OUT_INTENSITY) HOLO_LAUNCH_INV(HOLO_ASM_OUT_INTENSITY);
    else HOLO_LAUNCH_INV(HOLO_ASM_OUT_COMPLEX);
#undef HOLO_LAUNCH_INV
    return check_launch("holo_asm_propagate");
}

// transposed intermediates (32 <= P <= 1024, S % 16 == 0)
template <int P, bool PAD>
static int launch_asm_t(const AsmArgs& a) {
    using R = RowTCfg<P>;
    using C = ColTCfg<P>;
    constexpr int S = PAD ? P / 2 : P;
    const size_t row_lds_f = R::lds(tw_entries_fwd<P>()), row_lds_i = R::lds(tw_entries_inv<P>());
    static DeviceOnce attr_once;  // per device, published after the attributes are set
    attr_once.run([&] {
        if (C::LDS > 48 * 1024)
        {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&asm_cols_t_kernel<P, PAD, false>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)C::LDS);
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&asm_cols_t_kernel<P, PAD, true>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)C::LDS);
        }
        if (row_lds_f > 48 * 1024) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&asm_rows_fwd_t_kernel<P, PAD, HOLO_ASM_IN_AMP_PHASE>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)row_lds_f);
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&asm_rows_fwd_t_kernel<P, PAD, HOLO_ASM_IN_COMPLEX>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)row_lds_f);
        }
        if (row_lds_i > 48 * 1024) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&asm_rows_inv_t_kernel<P, PAD, HOLO_ASM_OUT_AMP_PHASE>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)row_lds_i);
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&asm_rows_inv_t_kernel<P, PAD, HOLO_ASM_OUT_INTENSITY>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)row_lds_i);
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&asm_rows_inv_t_kernel<P, PAD, HOLO_ASM_OUT_COMPLEX>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)row_lds_i);
        }
    });
    const long blocks_in = (long)a.n_fields * S / kTileRows;
    // transfer-function tables: extra blocks of the row kernel (blocks of the smallest grids have fewer than 64 threads: no table)
    const int n_out = a.n_fields * a.n_dist;
    const bool tables = R::THREADS >= 64 && C::THREADS >= 64 && a.htab != nullptr;
    const int slots = n_out < kHSlots ? n_out : kHSlots;
    const long tab_blocks = tables ? (long)((htab_elems<P>() + R::THREADS - 1) / R::THREADS) : 0;
    (void)slots;
    // persistent row blocks: as many as are resident at once (LDS / wave limit), each walking tiles with that stride
    const long fwd_slots = (long)device_cus() * R::bpc(tw_entries_fwd<P>());
    const long fwd_grid = (HOLO_ASM_PERSIST_FWD && blocks_in > fwd_slots) ? fwd_slots : blocks_in;
#define HOLO_LAUNCH_FWD(mode_)                                                                                              \
    hipLaunchKernelGGL((asm_rows_fwd_t_kernel<P, PAD, mode_>), dim3((unsigned)(fwd_grid + tab_blocks)), dim3(R::THREADS), row_lds_f, \
                       a.stream, a.in0, a.in1, a.s0, a.s1, a.phase_scale, reinterpret_cast<v2f*>(a.tbuf),                   \
                       (int)tab_blocks, reinterpret_cast<v2f*>(a.htab), a.phase_const, n_out, 1.0 / a.ppx, a.lam2, a.lamb,   \
                       (int)blocks_in)
    if (a.in_mode == HOLO_ASM_IN_COMPLEX) HOLO_LAUNCH_FWD(HOLO_ASM_IN_COMPLEX);
    else HOLO_LAUNCH_FWD(HOLO_ASM_IN_AMP_PHASE);
#undef HOLO_LAUNCH_FWD
    // persistent column blocks: as many as are resident at once (LDS, the register-limited waves per SIMD of the variant)
    auto col_grid = [&](long tiles, int waves_cap) {
        const int wps = C::WAVES_PER_SIMD < waves_cap ? C::WAVES_PER_SIMD : waves_cap;
        long per_cu = (long)wps * 256 / C::THREADS;
        if (per_cu > C::BPC) per_cu = C::BPC;
        if (per_cu < 1) per_cu = 1;
        const long slots = (long)device_cus() * per_cu;
        return (unsigned)(tiles < slots ? tiles : slots);
    };
    if (a.n_dist > 1)    // a sweep: tiles are (field, column group), the distances looped inside on the shared spectrum
        hipLaunchKernelGGL((asm_cols_t_kernel<P, PAD, true>), dim3(col_grid((long)(P / C::CW) * a.n_fields, 4)), dim3(C::THREADS), C::LDS, a.stream,
                           reinterpret_cast<const v2f*>(a.tbuf), reinterpret_cast<v2f*>(a.ubuf), a.phase_const, a.n_dist,
                           1.0 / a.ppx, a.lam2, a.lamb, a.ablate, tables ? reinterpret_cast<const v2f*>(a.htab) : nullptr,
                           (int)((long)(P / C::CW) * a.n_fields));
    else
        hipLaunchKernelGGL((asm_cols_t_kernel<P, PAD, false>), dim3(col_grid((long)(P / C::CW) * n_out, 8)), dim3(C::THREADS), C::LDS, a.stream,
                           reinterpret_cast<const v2f*>(a.tbuf), reinterpret_cast<v2f*>(a.ubuf), a.phase_const, a.n_dist,
                           1.0 / a.ppx, a.lam2, a.lamb, a.ablate, tables ? reinterpret_cast<const v2f*>(a.htab) : nullptr,
                           (int)((long)(P / C::CW) * n_out));
    const float scale = 1.0f / ((float)P * (float)P);  // ifft2 "backward" normalisation; exact (power of two)
    const long tiles_out = blocks_in * a.n_dist;
    const long inv_slots = (long)device_cus() * R::bpc(tw_entries_inv<P>());
    // measured (64 fields): P = 1024: 35.5 -> 30.4 us with the persistent grid, P = 512: 23.0 -> 24.3 (one tile per block there)
    const long inv_grid = (HOLO_ASM_PERSIST_ROWS && P >= 1024 && tiles_out > inv_slots) ? inv_slots : tiles_out;
#define HOLO_LAUNCH_INV(mode_)                                                                                              \
    hipLaunchKernelGGL((asm_rows_inv_t_kernel<P, PAD, mode_>), dim3((unsigned)inv_grid), dim3(R::THREADS), row_lds_i, a.stream, \
                       reinterpret_cast<const v2f*>(a.ubuf), a.out0, a.out1, a.os0, a.os1, scale, (int)tiles_out)
    if (a.out_mode == HOLO_ASM_OUT_AMP_PHASE) HOLO_LAUNCH_INV(HOLO_ASM_OUT_AMP_PHASE);
    else if (a.out_mode == HOLO_ASM_OUT_INTENSITY) HOLO_LAUNCH_INV(HOLO_ASM_OUT_INTENSITY);
    else HOLO_LAUNCH_INV(HOLO_ASM_OUT_COMPLEX);
#undef HOLO_LAUNCH_INV
    return check_launch("holo_asm_propagate");
}

template <int P>
static int launch_asm_pad(const AsmArgs& a, bool pad) {
    if constexpr (P >= 32 && P <= 1024) {
        return pad ? launch_asm_t<P, true>(a) : launch_asm_t<P, false>(a);
    } else if constexpr (P == 16) {
        return pad ? launch_asm<P, true>(a) : launch_asm_t<P, false>(a);   // S = 8 rows cannot fill a 16-row tile
    } else {
        return pad ? launch_asm<P, true>(a) : launch_asm<P, false>(a);
    }
}

// ---- generic sizes: direct DFT ------------------------------------------------------------------------------
// Any FFT side 2 <= P <= 4096 that is not a power of two in [16, 2048] (e.g. the 40x40 output of a 36x36
// hologram, P = 80): the same three-stage decomposition with O(P) direct sums per output instead of the
// Stockham passes.  A correctness path, not a fast one (O(P^3) per field); twiddles W_P^m from a float64-
// rounded table indexed by (c*k) mod P, so the accuracy matches the FFT path.
__global__ void asm_generic_tables_kernel(float2* tw, double* f2, int p, double ppx) {
    const int m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= p) return;
    double s, c;
    sincospi(-2.0 * (double)m / (double)p, &s, &c);
    tw[m] = make_float2((float)c, (float)s);
    // fftshift puts bin k at index (k + p/2) mod p and G is sampled at (index - p/2)/(p*px): signed bin
    const int ks = m < p - p / 2 ? m : m - p;
    const double f = (double)ks / ppx;
    f2[m] = f * f;
}

// one block per distinct row: T[row][k] = sum_c x[c] W^{ck}
__global__ __launch_bounds__(256) void asm_generic_rows_fwd_kernel(
    const float* __restrict__ in0, const float* __restrict__ in1, int in_mode, long stride0, long stride1,
    float phase_scale, v2f* __restrict__ tbuf, const v2f* __restrict__ tw, int p, int s, int pad) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    v2f* x = reinterpret_cast<v2f*>(smem_raw);
    const long row = blockIdx.x;
    const long f = row / s;
    const int y = (int)(row % s);
    for (int c = threadIdx.x; c < p; c += 256) {
        int col = c - pad;
        col = col < 0 ? 0 : (col > s - 1 ? s - 1 : col);
        const long pix = (long)y * s + col;
        x[c] = load_field(in0, in1, in_mode, f * stride0 + pix, f * stride1 + pix, phase_scale);
    }
    __syncthreads();
    for (int k = threadIdx.x; k < p; k += 256) {
        v2f acc = {0.f, 0.f};
        int m = 0;
        for (int c = 0; c < p; ++c) {
            acc += cmul(x[c], tw[m]);
            m += k;
            if (m >= p) m -= p;
        }
        tbuf[row * p + k] = acc;
    }
}

// one block per (output field, column kx): column DFT of the replicate-padded column, H, inverse DFT of the kept rows
__global__ __launch_bounds__(256) void asm_generic_cols_kernel(
    const v2f* __restrict__ tbuf, v2f* __restrict__ ubuf, const v2f* __restrict__ tw, const double* __restrict__ f2,
    const double* __restrict__ phase_const, int n_dist, int p, int s, int pad, double lam2, double lamb) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    v2f* x = reinterpret_cast<v2f*>(smem_raw);   // p inputs, then p spectrum values
    v2f* spec = x + p;
    const int kx = blockIdx.x, fo = blockIdx.y, f = fo / n_dist;
    for (int r = threadIdx.x; r < p; r += 256) {
        int y = r - pad;
        y = y < 0 ? 0 : (y > s - 1 ? s - 1 : y);
        x[r] = tbuf[((long)f * s + y) * p + kx];
    }
    __syncthreads();
    const double c_over_l = phase_const[fo] / lamb;
    const double fx2 = f2[kx];
    for (int ky = threadIdx.x; ky < p; ky += 256) {
        v2f acc = {0.f, 0.f};
        int m = 0;
        for (int r = 0; r < p; ++r) {
            acc += cmul(x[r], tw[m]);
            m += ky;
            if (m >= p) m -= p;
        }
        spec[ky] = cmul(acc, transfer_value(fx2 + f2[ky], lam2, c_over_l));
    }
    __syncthreads();
    for (int y = threadIdx.x; y < s; y += 256) {
        const int r = y + pad;
        v2f acc = {0.f, 0.f};
        int m = 0;
        for (int ky = 0; ky < p; ++ky) {
            acc += cmul_conj(spec[ky], tw[m]);
            m += r;
            if (m >= p) m -= p;
        }
        ubuf[((long)fo * s + y) * p + kx] = acc;
    }
}

__global__ __launch_bounds__(256) void asm_generic_rows_inv_kernel(
    const v2f* __restrict__ ubuf, float* __restrict__ out0, float* __restrict__ out1, int out_mode, long ostride0,
    long ostride1, const v2f* __restrict__ tw, int p, int s, int pad, float scale) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    v2f* x = reinterpret_cast<v2f*>(smem_raw);
    const long row = blockIdx.x;
    const long fo = row / s;
    const int y = (int)(row % s);
    for (int k = threadIdx.x; k < p; k += 256) x[k] = ubuf[row * p + k];
    __syncthreads();
    for (int xo = threadIdx.x; xo < s; xo += 256) {
        const int c = xo + pad;
        v2f acc = {0.f, 0.f};
        int m = 0;
        for (int k = 0; k < p; ++k) {
            acc += cmul_conj(x[k], tw[m]);
            m += c;
            if (m >= p) m -= p;
        }
        const v2f o = acc * scale;
        const long pix = (long)y * s + xo;
        if (out_mode == HOLO_ASM_OUT_AMP_PHASE) {
            out0[fo * ostride0 + pix] = sqrtf(o.x * o.x + o.y * o.y);
            out1[fo * ostride1 + pix] = atan2f(o.y, o.x);
        } else if (out_mode == HOLO_ASM_OUT_INTENSITY) {
            const float a = sqrtf(o.x * o.x + o.y * o.y);
            out0[fo * ostride0 + pix] = a * a;
        } else {
            reinterpret_cast<v2f*>(out0)[fo * ostride0 + pix] = o;
        }
    }
}

static int launch_asm_generic(const AsmArgs& a, int p, int s, int pad) {
    hipLaunchKernelGGL(asm_generic_tables_kernel, dim3(ceil_div(p, 256)), dim3(256), 0, a.stream, a.twf, a.f2, p, a.ppx);
    const v2f* tw = reinterpret_cast<const v2f*>(a.twf);
    const long rows_in = (long)a.n_fields * s;
    hipLaunchKernelGGL(asm_generic_rows_fwd_kernel, dim3((unsigned)rows_in), dim3(256), (size_t)p * sizeof(float2), a.stream,
                       a.in0, a.in1, a.in_mode, a.s0, a.s1, a.phase_scale, reinterpret_cast<v2f*>(a.tbuf), tw, p, s, pad);
    hipLaunchKernelGGL(asm_generic_cols_kernel, dim3(p, a.n_fields * a.n_dist), dim3(256), (size_t)2 * p * sizeof(float2),
                       a.stream, reinterpret_cast<const v2f*>(a.tbuf), reinterpret_cast<v2f*>(a.ubuf), tw, a.f2, a.phase_const,
                       a.n_dist, p, s, pad, a.lam2, a.lamb);
    const float scale = (float)(1.0 / ((double)p * (double)p));
    hipLaunchKernelGGL(asm_generic_rows_inv_kernel, dim3((unsigned)(rows_in * a.n_dist)), dim3(256), (size_t)p * sizeof(float2),
                       a.stream, reinterpret_cast<const v2f*>(a.ubuf), a.out0, a.out1, a.out_mode, a.os0, a.os1, tw, p, s, pad,
                       scale);
    return check_launch("holo_asm_propagate (generic size)");
}

constexpr size_t kTableBytes = 2 * kTwSlots * sizeof(float2) + kTwSlots * sizeof(double);

}  // namespace holo

extern "C" {

// Workspace layout.  Head: tables of the generic-size path; then T (n_fields slabs), U (n_fields * n_dist slabs) and,
// where the path has them (power-of-two transform side 32..1024), the transfer-function tables.
struct AsmLayout {
    size_t t_off, u_off, htab_off, total;
};
static AsmLayout asm_layout(int n_fields, int n_dist, int s, int zero_padding) {
    AsmLayout l;
    const size_t p = zero_padding ? 2 * (size_t)s : (size_t)s;
    const size_t slab = (size_t)s * p * sizeof(float2);
    const bool pow2 = (p & (p - 1)) == 0;
    const bool transposed = pow2 && p >= 32 && p <= 1024;
    const size_t n_out = (size_t)n_fields * (size_t)n_dist;
    const size_t slots = n_out < (size_t)holo::kHSlots ? n_out : (size_t)holo::kHSlots;
    const size_t htab = transposed ? slots * (p / 2 + 1) * (p / 2 + 1) * sizeof(float2) : 0;
    l.t_off = holo::kTableBytes;
    l.u_off = l.t_off + slab * (size_t)n_fields;
    l.htab_off = l.u_off + slab * n_out;
    l.total = l.htab_off + htab;
    return l;
}

int holo_asm_phase_constants_f32(const float* d, long d_stride, int period, float add, float mul, double* out, int n, void* stream) {
    using namespace holo;
    HOLO_REQUIRE(d && out, HOLO_ERR_ARG, "holo_asm_phase_constants_f32: null pointer");
    HOLO_REQUIRE(n > 0 && period > 0 && d_stride >= 0, HOLO_ERR_ARG, "holo_asm_phase_constants_f32: n=%d period=%d stride=%ld", n, period,
                 d_stride);
    hipLaunchKernelGGL(phase_constants_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, as_stream(stream), d, d_stride, period, add,
                       mul, out, n);
    return check_launch("holo_asm_phase_constants_f32");
}

size_t holo_asm_workspace_bytes(int n_fields, int n_dist, int s, int zero_padding) {
    if (n_fields <= 0 || n_dist <= 0 || s <= 0) return 0;
    return asm_layout(n_fields, n_dist, s, zero_padding).total;
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
    // Angular_Spectrum_Method.py:12-14: pad S//2 per side but sample 2S frequencies: odd S cannot be padded there either
    HOLO_REQUIRE(!zero_padding || s % 2 == 0, HOLO_ERR_UNSUPPORTED,
                 "holo_asm_propagate: zero_padding needs an even field side (got %d), as in the reference", s);
    const int p = zero_padding ? 2 * s : s;
    HOLO_REQUIRE(p >= 2 && p <= 4096, HOLO_ERR_UNSUPPORTED, "holo_asm_propagate: transform side %d outside [2,4096]", p);
    const bool fast = (p & (p - 1)) == 0 && p >= 16 && p <= 2048;
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
    a.ppx = (double)s * pixel_size * (zero_padding ? 2.0 : 1.0);  // P*px; scaling by 2 is exact
    a.lam2 = wavelength * wavelength;
    a.lamb = wavelength;
    a.phase_scale = phase_scale;
    a.twf = static_cast<float2*>(workspace);
    a.twi = a.twf + kTwSlots;
    a.f2 = reinterpret_cast<double*>(a.twi + kTwSlots);
    const AsmLayout lay = asm_layout(n_fields, n_dist, s, zero_padding);
    unsigned char* ws = static_cast<unsigned char*>(workspace);
    a.tbuf = reinterpret_cast<float2*>(ws + lay.t_off);
    a.ubuf = reinterpret_cast<float2*>(ws + lay.u_off);
    a.htab = lay.total > lay.htab_off ? reinterpret_cast<float2*>(ws + lay.htab_off) : nullptr;
    a.stream = as_stream(stream);
    a.ablate = 0;
#ifdef HOLO_DEBUG   // timing experiments only (wrong results): never read in the shipped build
    {
        const char* env = getenv("HOLO_ASM_ABLATE");
        const int bits = env ? atoi(env) : 0;
        a.ablate = bits & 3;
        static int last = -1;
        if (bits != last) {
            (void)hipMemcpyToSymbol(HIP_SYMBOL(holo::g_asm_abl), &bits, sizeof(int));
            last = bits;
        }
    }
#endif
    const bool pad = zero_padding != 0;
    if (!fast) return launch_asm_generic(a, p, s, pad ? s / 2 : 0);
    switch (p) {
        case 16: return launch_asm_pad<16>(a, pad);
        case 32: return launch_asm_pad<32>(a, pad);
        case 64: return launch_asm_pad<64>(a, pad);
        case 128: return launch_asm_pad<128>(a, pad);
        case 256: return launch_asm_pad<256>(a, pad);
        case 512: return launch_asm_pad<512>(a, pad);
        case 1024: return launch_asm_pad<1024>(a, pad);
        case 2048: return launch_asm_pad<2048>(a, pad);
    }
    return HOLO_ERR_UNSUPPORTED;
}

}  // extern "C"
