// Backward pass of the training objective (Net.forward, net.py:199-226; losses net.py:186-197), fp32, NHWC.
//
// The reference trains the decoder (and the distance head) through torch.autograd over aten ops.  Here every backward op of the
// path is a HIP kernel behind the C-ABI, built so that the heavy products reuse the forward machinery:
//
//   * data gradient of ReflectionPad2d(1) + Conv2d(3x3):  dXp = full correlation of dA with the flipped, transposed weights on the
//     PADDED domain, then the adjoint of the reflection folds the pad ring back (row -1 onto row 1, row H onto row H-2).  The
//     correlation is the FORWARD convolution kernel (Winograd / implicit GEMM, conv_winograd.hip / conv_mfma.hip) run on dA embedded
//     in a zero border of TWO pixels: the forward kernel's own reflection then only ever mirrors zeros.  relu_pool_bwd_pad_kernel
//     builds that embedded tensor (ReLU mask and max-pool routing applied on the way, the bias gradient summed in the same pass) —
//     for a batch as ONE mosaic image whose neighbours share their zero rows and columns, so that the convolution rounds the
//     batch, not every image, up to whole tiles; fold_reflect_kernel does the fold.
//   * weight gradient: dW[m][n][tap] = sum over pixels of P[pixel][m] * Q[pixel + tap][n] — a GEMM whose K runs over PIXELS on
//     v_mfma_f32_32x32x2_f32.  pix_gemm9_lds_kernel (whole 32-pixel strips, channels % 64): both operands staged through LDS by
//     LDS-DMA, a ring of four Q rows serving the three vertical taps; pix_gemm_kernel (other widths, the one-tap case): operands
//     straight from global memory.  Partial tiles go to a workspace and are folded 32 to 1 in a fixed order (deterministic).
//   * ConvTranspose2d(k2,s2): its data gradient is a 1x1 convolution of the space-to-depth gradient (B,H,W,4*Cout) — the transposed
//     convolution's own one-tap GEMM with a plain store (holo_conv1x1_f32, conv_mfma.hip); its weight gradient is pix_gemm with
//     one tap.
//   * the skinny layers (3 -> 64 first, 64 -> 2 last), statistics (pixel-split for large maps), the distance head and its
//     training-mode Dropout: small VALU kernels.
//
// Everything is exact fp32 arithmetic; results differ from torch.autograd by summation order only (tests/test_hip_backward.py).
#include <algorithm>
#include <cstdlib>

#include "common.h"

namespace holo {

typedef float floatx16 __attribute__((ext_vector_type(16)));

// ---- MaxPool2d(2,2,ceil_mode=True), unfused (training keeps the pre-pool activation for the backward routing) ----------------
__global__ __launch_bounds__(256) void maxpool2x2_kernel(const float* __restrict__ x, float* __restrict__ y, int h, int w, int c4,
                                                         int ho, int wo, long total) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int cq = (int)(i % c4);
    long r = i / c4;
    const int X = (int)(r % wo);
    r /= wo;
    const int Y = (int)(r % ho);
    const long n = r / ho;
    const float4* xi = reinterpret_cast<const float4*>(x) + n * h * (long)w * c4;
    float4 m = xi[((long)(2 * Y) * w + 2 * X) * c4 + cq];
#pragma unroll
    for (int k = 1; k < 4; ++k) {
        const int yy = 2 * Y + (k >> 1), xx = 2 * X + (k & 1);
        if (yy < h && xx < w) {
            const float4 v = xi[((long)yy * w + xx) * c4 + cq];
            m.x = fmaxf(m.x, v.x); m.y = fmaxf(m.y, v.y); m.z = fmaxf(m.z, v.z); m.w = fmaxf(m.w, v.w);
        }
    }
    reinterpret_cast<float4*>(y)[i] = m;
}

// ---- dA = route(dy) * mask, embedded in a zero border -------------------------------------------------------------------------
// out: (B, h + 2*border, w + 2*border, c), zero outside the interior.  Interior pixel (y, x):
//   g = pool ? (y_act[y,x] is the FIRST maximum of its 2x2 window in scan order ? dy[y/2, x/2] : 0) : dy[y, x]
//   out = relu ? (y_act > 0 ? g : 0) : g                      (torch: max_pool2d_backward routes to the stored argmax, the first
//                                                              strict maximum; threshold_backward masks with the output > 0)
// MOSAIC.  pitch = rows from one image's block to the next: h + 2*border (separate blocks), or less — down to h + border — for the TALL
// layout, where consecutive images share their zero rows (image n's lower border rows are image n + 1's upper ones): the whole batch
// is then ONE image of (B - 1) * pitch + h + 2*border rows for the convolution that follows, which no longer rounds every image up
// to whole tiles separately.  gx > 1 puts gx images side by side as well, cpitch = w + border columns apart (shared zero columns):
// image n sits at (n / gx, n % gx), cells past the batch stay zero; the 16-pixel tile columns then round the mosaic's width.
__global__ __launch_bounds__(256) void relu_pool_bwd_pad_kernel(const float* __restrict__ dy, const float* __restrict__ y_act,
                                                                const float* __restrict__ y_pool, float* __restrict__ out, int batch,
                                                                int h, int w, int c4, int relu, int border, int pitch, int gx,
                                                                int cpitch, int wp, long total, double* __restrict__ bias_part) {
    // grid-stride over the float4 elements; the stride is a multiple of c4 (c4 divides 256), so a thread stays on one channel quad
    // and can keep that quad's running sum (the bias gradient = sum of the masked gradient over pixels) in float64
    __shared__ double red[256][4];
    double bsum[4] = {0, 0, 0, 0};
    const int img_rows = (batch + gx - 1) / gx;         // rows of images in the mosaic (gx side by side; gx = 1: a column of images)
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int cq = (int)(i % c4);                   // = threadIdx.x % c4 on every trip when c4 divides 256 (the bias sums rely on it)
        long r = i / c4;
        const int xp = (int)(r % wp);                   // wp = pixel columns of the whole output
        r /= wp;                                        // row of the whole output
        int nr = (int)(r / pitch);
        if (nr > img_rows - 1) nr = img_rows - 1;
        int nc = xp / cpitch;
        if (nc > gx - 1) nc = gx - 1;
        const long n = (long)nr * gx + nc;
        const int y = (int)(r - (long)nr * pitch) - border, x = xp - nc * cpitch - border;
        float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
        if (n < batch && y >= 0 && y < h && x >= 0 && x < w) {
            const float4* ya = reinterpret_cast<const float4*>(y_act) + n * h * (long)w * c4;
            const float4 a = relu || y_pool ? ya[((long)y * w + x) * c4 + cq] : g;
            if (y_pool) {
                const int ho = (h + 1) >> 1, wo = (w + 1) >> 1;
                const long po = (n * ho + (y >> 1)) * (long)wo + (x >> 1);
                const float4 m = reinterpret_cast<const float4*>(y_pool)[po * c4 + cq];
                const float4 d = reinterpret_cast<const float4*>(dy)[po * c4 + cq];
                // earlier members of the window (scan order) that already hold the maximum take the gradient instead
                bool e0 = false, e1 = false, e2 = false, e3 = false;
                const int k = ((y & 1) << 1) | (x & 1);
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    if (q < k) {
                        const int yy = (y & ~1) + (q >> 1), xx = (x & ~1) + (q & 1);     // always inside the image for q < k
                        const float4 v = ya[((long)yy * w + xx) * c4 + cq];
                        e0 |= v.x == m.x; e1 |= v.y == m.y; e2 |= v.z == m.z; e3 |= v.w == m.w;
                    }
                }
                g.x = (a.x == m.x && !e0) ? d.x : 0.f;
                g.y = (a.y == m.y && !e1) ? d.y : 0.f;
                g.z = (a.z == m.z && !e2) ? d.z : 0.f;
                g.w = (a.w == m.w && !e3) ? d.w : 0.f;
            } else {
                g = reinterpret_cast<const float4*>(dy)[((n * h + y) * (long)w + x) * c4 + cq];
            }
            if (relu) {
                g.x = a.x > 0.f ? g.x : 0.f; g.y = a.y > 0.f ? g.y : 0.f; g.z = a.z > 0.f ? g.z : 0.f; g.w = a.w > 0.f ? g.w : 0.f;
            }
        }
        reinterpret_cast<float4*>(out)[i] = g;
        bsum[0] += g.x; bsum[1] += g.y; bsum[2] += g.z; bsum[3] += g.w;
    }
    if (bias_part) {              // block partial: the threads of a quad in thread order
#pragma unroll
        for (int j = 0; j < 4; ++j) red[threadIdx.x][j] = bsum[j];
        __syncthreads();
        if (threadIdx.x < c4) {
            double t[4] = {0, 0, 0, 0};
            for (int k = threadIdx.x; k < 256; k += c4)
#pragma unroll
                for (int j = 0; j < 4; ++j) t[j] += red[k][j];
#pragma unroll
            for (int j = 0; j < 4; ++j) bias_part[(long)blockIdx.x * 4 * c4 + 4 * threadIdx.x + j] = t[j];
        }
    }
}

// ---- adjoint of ReflectionPad2d(1): fold the pad ring of dXp (padded coordinates -1 .. h at index + border) onto the interior ----
__global__ __launch_bounds__(256) void fold_reflect_kernel(const float* __restrict__ dxp, float* __restrict__ dx, int h, int w, int c4,
                                                           int border, int pitch, int gx, int cpitch, int wp, long total) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int cq = (int)(i % c4);
    long r = i / c4;
    const int x = (int)(r % w);
    r /= w;
    const int y = (int)(r % h);
    const long n = r / h;
    const float4* src = reinterpret_cast<const float4*>(dxp) + ((n / gx) * pitch * (long)wp + (n % gx) * cpitch) * c4;   // wp: the buffer's columns
    // padded rows that reflect onto row y: y itself, -1 if y == 1, h if y == h - 2
    int ys[3], xs[3], ny = 0, nx = 0;
    ys[ny++] = y;
    if (y == 1) ys[ny++] = -1;
    if (y == h - 2) ys[ny++] = h;
    xs[nx++] = x;
    if (x == 1) xs[nx++] = -1;
    if (x == w - 2) xs[nx++] = w;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int a = 0; a < ny; ++a)
        for (int b = 0; b < nx; ++b) {
            const float4 v = src[((long)(ys[a] + border) * wp + xs[b] + border) * c4 + cq];
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
    reinterpret_cast<float4*>(dx)[i] = s;
}

// ---- ConvTranspose2d(k2,s2) backward helper: G[n,Y,X,q*c + ch] = dy[n,2Y+dy,2X+dx,ch] * (y > 0), q = dy*2+dx ----------------
__global__ __launch_bounds__(256) void space_to_depth2_relu_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                                                   float* __restrict__ g, int h, int w, int c4, int relu, long total) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int cq = (int)(i % c4);
    long r = i / c4;
    const int q = (int)(r & 3);
    r >>= 2;
    const int X = (int)(r % w);
    r /= w;
    const int Y = (int)(r % h);
    const long n = r / h;
    const long src = ((n * (2 * h) + 2 * Y + (q >> 1)) * (long)(2 * w) + 2 * X + (q & 1)) * c4 + cq;
    float4 v = reinterpret_cast<const float4*>(dy)[src];
    if (relu) {
        const float4 a = reinterpret_cast<const float4*>(y)[src];
        v.x = a.x > 0.f ? v.x : 0.f; v.y = a.y > 0.f ? v.y : 0.f; v.z = a.z > 0.f ? v.z : 0.f; v.w = a.w > 0.f ? v.w : 0.f;
    }
    reinterpret_cast<float4*>(g)[i] = v;     // i = ((n*h + Y)*w + X)*4*c4 + q*c4 + cq
}

// ---- partial rows folded 32 to 1 (fixed order) until at most 64 remain: the finishing kernels below then walk a short list -----
template <typename T>
__global__ __launch_bounds__(256) void fold_rows_kernel(const T* __restrict__ in, T* __restrict__ out, long elems, int rows, int per) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= elems) return;
    const int r0 = blockIdx.y * per;
    const int r1 = r0 + per < rows ? r0 + per : rows;
    T s = 0;
    for (int r = r0; r < r1; ++r) s += in[(long)r * elems + i];
    out[(long)blockIdx.y * elems + i] = s;
}

// ---- sum over pixels per channel (bias gradients): x (rows, c) -> out[c]; deterministic two-stage ------------------------------
__global__ __launch_bounds__(256) void channel_sum_partial_kernel(const float* __restrict__ x, double* __restrict__ part, long rows,
                                                                  int c, int parts) {
    // grid (c / 64, parts): 16 lanes x float4 = 64 channels of a row, 16 row groups per block
    __shared__ double s_sum[16][64];
    const int tid = threadIdx.x, cq = tid & 15, pg = tid >> 4;
    const int c0 = blockIdx.x * 64;
    const long per = (rows + parts - 1) / parts;
    const long r0 = (long)blockIdx.y * per, r1 = r0 + per < rows ? r0 + per : rows;
    double s[4] = {0, 0, 0, 0};
    for (long p = r0 + pg; p < r1; p += 16) {
        const float4 v = *reinterpret_cast<const float4*>(x + p * c + c0 + 4 * cq);
        s[0] += v.x; s[1] += v.y; s[2] += v.z; s[3] += v.w;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) s_sum[pg][4 * cq + j] = s[j];
    __syncthreads();
    if (tid < 64) {
        double t = 0;
        for (int g = 0; g < 16; ++g) t += s_sum[g][tid];
        part[(long)blockIdx.y * c + c0 + tid] = t;
    }
}
__global__ __launch_bounds__(256) void channel_sum_final_kernel(const double* __restrict__ part, float* __restrict__ out, int c, int parts) {
    const int ch = blockIdx.x * 256 + threadIdx.x;
    if (ch >= c) return;
    double t = 0;
    for (int p = 0; p < parts; ++p) t += part[(long)p * c + ch];
    out[ch] = (float)t;
}

// ---- pixel-K GEMM: out[m][n][tap] = sum over pixels P[pixel][m] * Q[reflect(pixel + tap)][n] -----------------------------------
// P: (B, h + 2*pb, w + 2*pb, mc) (the embedded gradient; interior read), Q: (B, h, w, nc).  Block = 4 waves on one 32 x 32 (m, n)
// tile; blockIdx.z and the wave index split the image rows; every (split, wave) writes its own partial tile, pix_gemm_reduce adds
// them in order.  v_mfma_f32_32x32x2_f32: A lane (m = lane & 31, k = lane >> 5), B lane (k, n): k = two neighbouring pixels.
// NT = 32-wide n tiles per wave (one tap: 4, so that one P value feeds four MFMAs).  The (row, pixel pair) steps of a wave form
// one flat sequence; the operands of step s + 2 are in flight while step s multiplies (three register sets in rotation).
template <int TAPS, int NT>
struct PixFrag {
    float a;
    bool ok;          // false: the P value counts as zero (odd width's last pair, steps past the end); applied at the multiply
    float b[TAPS * NT];
};
template <int TAPS, int NT>
__global__ __launch_bounds__(256, 2) void pix_gemm_kernel(const float* __restrict__ p, int pb, const float* __restrict__ q,
                                                          float* __restrict__ part, int batch, int h, int w, int mc, int nc,
                                                          int rows_per_wave, int ppitch, int gx, int cpitch, int pwidth) {
    constexpr int NB = TAPS * NT;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 31, hs = lane >> 5;
    const int m0 = blockIdx.x * 32, n0 = blockIdx.y * 32 * NT;
    const long slot = (long)blockIdx.z * 4 + wave;                    // partial index
    const long row0 = slot * rows_per_wave;                           // image rows (n * h + y) of this wave
    const long rows_total = (long)batch * h;
    long row1 = row0 + rows_per_wave;
    if (row1 > rows_total) row1 = rows_total;
    floatx16 acc[NB];
#pragma unroll
    for (int t = 0; t < NB; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[t][i] = 0.0f;
    const int hp = ppitch, wp = pwidth;              // image n's block of P: row (n / gx) * ppitch, column (n % gx) * cpitch of a buffer pwidth wide
    const int xsteps = (w + 1) >> 1;
    const int steps = row1 > row0 ? (int)(row1 - row0) * xsteps : 0;
    if (steps > 0) {
        // load cursor (wave-uniform): image cn, row cy, pixel pair cx.  The loop body is branch-free so that the operands of two
        // later steps stay in flight behind counted waits: past the last step the cursor stays put and the P value reads as zero.
        int cn = (int)(row0 / h), cy = (int)(row0 - (long)cn * h), cx = 0;
        auto issue = [&](PixFrag<TAPS, NT>& f, int idx) {
            const int xx = cx + hs;
            const bool ok = xx < w && idx < steps;
            const int xc = xx < w ? xx : w - 1;
            const float av = p[(((long)(cn / gx) * hp + cy + pb) * wp + (cn % gx) * cpitch + pb + xc) * mc + m0 + r];
            f.a = av;
            f.ok = ok;
            if constexpr (TAPS == 9) {
                long qoff[3];
                int rx[3];
#pragma unroll
                for (int d = 0; d < 3; ++d) {
                    qoff[d] = ((long)cn * h + reflect1(cy + d - 1, h)) * w;
                    rx[d] = reflect1(xc + d - 1, w);
                }
#pragma unroll
                for (int t = 0; t < 9; ++t)
#pragma unroll
                    for (int j = 0; j < NT; ++j) f.b[t * NT + j] = q[(qoff[t / 3] + rx[t % 3]) * nc + n0 + r + 32 * j];
            } else {
                const long qoff = (((long)cn * h + cy) * w + xc) * nc + n0 + r;
#pragma unroll
                for (int j = 0; j < NT; ++j) f.b[j] = q[qoff + 32 * j];
            }
            const bool adv = idx + 1 < steps;
            const int nx = cx + 2;
            const bool wrap_x = nx >= w;
            const int ny = cy + (wrap_x ? 1 : 0);
            const bool wrap_y = ny >= h;
            cx = adv ? (wrap_x ? 0 : nx) : cx;
            cy = adv ? (wrap_y ? 0 : ny) : cy;
            cn = adv ? cn + (wrap_y ? 1 : 0) : cn;
        };
        auto multiply = [&](const PixFrag<TAPS, NT>& f) {
            const float a = f.ok ? f.a : 0.0f;
#pragma unroll
            for (int t = 0; t < NB; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, f.b[t], acc[t], 0, 0, 0);
        };
        PixFrag<TAPS, NT> f0, f1, f2;
        issue(f0, 0);
        issue(f1, 1);
        for (int s = 0; s < steps; s += 3) {         // the fences keep the compiler from sinking a set's loads below its use
            issue(f2, s + 2);
            __builtin_amdgcn_sched_barrier(0);
            multiply(f0);
            __builtin_amdgcn_sched_barrier(0);
            issue(f0, s + 3);
            __builtin_amdgcn_sched_barrier(0);
            multiply(f1);
            __builtin_amdgcn_sched_barrier(0);
            issue(f1, s + 4);
            __builtin_amdgcn_sched_barrier(0);
            multiply(f2);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    // D[m][n]: m = (i & 3) + 8 * (i >> 2) + 4 * hs, n = r.  Partial layout [slot][m][n][tap] over the FULL (mc, nc) matrix.
    float* __restrict__ dst = part + slot * (long)mc * nc * TAPS;
#pragma unroll
    for (int t = 0; t < TAPS; ++t)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int m = (i & 3) + 8 * (i >> 2) + 4 * hs;
                dst[((long)(m0 + m) * nc + n0 + 32 * j + r) * TAPS + t] = acc[t * NT + j][i];
            }
}
// ---- the 9-tap case with both operands staged through LDS (w % 32 == 0, channels % 64 == 0) ------------------------------------
// Block = 4 waves on a 64 x 64 (m, n) tile, wave (wm, wn) on 32 x 32 with its nine tap accumulators.  A block walks down the
// rows [y0, y1) of one 32-pixel column strip of one image: per row it holds the strip of P (32 pixels x 64 channels, two buffers)
// and a ring of four Q rows (the strip plus its reflected neighbours, 36 pixels x 64 channels), all filled by LDS-DMA one row
// ahead of the multiplies, so every global byte is read by wide coalesced pieces and each Q row is fetched once for its three
// vertical taps.  LDS image: [pixel][64 channels] linear with the two 32-channel halves of odd pixels exchanged (applied to the
// per-lane SOURCE address of the DMA), so that the MFMA operand read - lanes 0..31 pixel X, lanes 32..63 pixel X + 1, 32 channels
// each - hits 64 distinct banks.  One partial tile per block, tap-major ([unit][tap][m][n]); pix_gemm_reduce adds them in order.
constexpr int kPgChunk = 32;
constexpr int kPgHalo = 36;
constexpr int kPgPBytes = kPgChunk * 64 * 4;
constexpr int kPgQBytes = kPgHalo * 64 * 4;
constexpr int kPgLdsBytes = 2 * kPgPBytes + 4 * kPgQBytes;

__global__ __launch_bounds__(256, 2) void pix_gemm9_lds_kernel(const float* __restrict__ p, int pb, const float* __restrict__ q,
                                                               float* __restrict__ part, int h, int w, int mc, int nc,
                                                               int rows_per_unit, int groups, int ppitch, int gx, int cpitch, int pwidth) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[kPgLdsBytes];
    const float* sf = reinterpret_cast<const float*>(smem);
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 31, hs = lane >> 5;
    const int wm = wave & 1, wn = wave >> 1;
    const int m0 = blockIdx.x * 64, n0 = blockIdx.y * 64;
    const int chunks = w / kPgChunk;
    int u = blockIdx.z;
    const int g = u % groups;
    u /= groups;
    const int c = u % chunks, n = u / chunks;
    const int y0 = g * rows_per_unit;
    int y1 = y0 + rows_per_unit;
    if (y1 > h) y1 = h;
    const int cx0 = c * kPgChunk;
    const int hp = ppitch, wp = pwidth;              // image n's block of P: row (n / gx) * ppitch, column (n % gx) * cpitch of a buffer pwidth wide
    const unsigned lds_base = (unsigned)(unsigned long long)(const __attribute__((address_space(3))) void*)smem;

#define PG_GLDS(sbase_, voff_, ldst_)                                                                      \
    {                                                                                                      \
        unsigned m0_keep;                                                                                  \
        const unsigned lds_u = __builtin_amdgcn_readfirstlane(lds_base + (unsigned)(ldst_));               \
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\t" \
                     "s_mov_b32 m0, %0"                                                                     \
                     : "=&s"(m0_keep)                                                                      \
                     : "v"(voff_), "s"(sbase_), "s"(lds_u)                                                 \
                     : "memory");                                                                          \
    }
    // a DMA piece = 4 pixels x 64 channels (1 KiB): lane -> pixel lane >> 4 of the piece, 4 channels from (lane & 15) * 4
    const int lpx = lane >> 4, lch = (lane & 15) * 4;
    auto fill_p = [&](int y, int buf) {               // 8 pieces, two per wave
        const float* rowp = p + (((long)(n / gx) * hp + y + pb) * wp + (n % gx) * cpitch + pb + cx0) * mc + m0;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int j = wave + 4 * k;
            const unsigned off = (unsigned)(((4 * j + lpx) * mc + (lch ^ (32 * (lpx & 1)))) * 4);
            PG_GLDS(rowp, off, buf * kPgPBytes + j * 1024);
        }
    };
    auto fill_q = [&](int ry, int slot) {             // 9 pieces: wave 0 takes three
        const float* rowq = q + ((long)n * h + ry) * w * nc + n0;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int j = wave + 4 * k;
            if (j < 9) {
                const int x = reflect1(cx0 - 1 + 4 * j + lpx, w);
                const unsigned off = (unsigned)((x * nc + (lch ^ (32 * (lpx & 1)))) * 4);
                PG_GLDS(rowq, off, 2 * kPgPBytes + slot * kPgQBytes + j * 1024);
            }
        }
    };

    floatx16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[t][i] = 0.0f;
    // operand read positions (float index): pixel hs of a pair, channel half swapped for odd pixels
    const int pa_lane = hs * 64 + ((wm * 32 + r) ^ (32 * hs));
    int qb_lane[2];
    qb_lane[0] = hs * 64 + ((wn * 32 + r) ^ (32 * hs)) + 2 * kPgPBytes / 4;              // even tap column: pixel parity = hs
    qb_lane[1] = hs * 64 + ((wn * 32 + r) ^ (32 * (hs ^ 1))) + 2 * kPgPBytes / 4;        // odd tap column

    fill_p(y0, y0 & 1);
    if (y0 > 0) fill_q(y0 - 1, (y0 - 1) & 3);
    fill_q(y0, y0 & 3);
    if (y0 + 1 < h) fill_q(y0 + 1, (y0 + 1) & 3);
    for (int y = y0; y < y1; ++y) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                              // row y landed for every wave; every wave is done reading row y - 1
        if (y + 1 < y1) {
            fill_p(y + 1, (y + 1) & 1);
            if (y + 2 < h) fill_q(y + 2, (y + 2) & 3);
        }
        const int pa = pa_lane + (y & 1) * (kPgPBytes / 4);
        int qrow[3];
#pragma unroll
        for (int d = 0; d < 3; ++d) qrow[d] = (reflect1(y + d - 1, h) & 3) * (kPgQBytes / 4);
#pragma unroll
        for (int s2 = 0; s2 < kPgChunk / 2; ++s2) {
            const float a = sf[pa + s2 * 128];
            float b[9];
#pragma unroll
            for (int t = 0; t < 9; ++t) b[t] = sf[qrow[t / 3] + qb_lane[(t % 3) & 1] + (2 * s2 + t % 3) * 64];
#pragma unroll
            for (int t = 0; t < 9; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b[t], acc[t], 0, 0, 0);
        }
    }
#undef PG_GLDS
    // D[m][n]: m = (i & 3) + 8 * (i >> 2) + 4 * hs, n = r
    float* __restrict__ dst = part + (long)blockIdx.z * mc * nc * 9;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int m = (i & 3) + 8 * (i >> 2) + 4 * hs;
            dst[((long)t * mc + m0 + wm * 32 + m) * nc + n0 + wn * 32 + r] = acc[t][i];
        }
}

// out[m][n][tap] = sum over the partial tiles, in slot order.  tap_major: partials laid out [slot][tap][m][n], else [slot][m][n][tap]
__global__ __launch_bounds__(256) void pix_gemm_reduce_kernel(const float* __restrict__ part, float* __restrict__ out, long elems, int slots,
                                                              int taps, int tap_major) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= elems) return;
    float s = 0.0f;
    for (int k = 0; k < slots; ++k) s += part[(long)k * elems + i];
    if (tap_major) {
        const long mn = elems / taps;
        out[(i % mn) * taps + i / mn] = s;
    } else {
        out[i] = s;
    }
}

// ---- last layer (ReflPad + Conv2d(64, 2, 3x3), no activation; NHWC in, NCHW planes out), backward ------------------------------
// data gradient on the padded domain: dxp[i][j][ci] = sum_tap sum_co dy[co][i - dy][j - dx] * w[co][ci][tap]   (then fold_reflect)
__global__ __launch_bounds__(256) void conv_last_dgrad_kernel(const float* __restrict__ dy, const float* __restrict__ w,
                                                              float* __restrict__ dxp, int h, int wd, long total) {
    // dxp: (B, h + 2, wd + 2, 64), border 1; thread = (padded pixel, channel quad)
    __shared__ float ws[2 * 64 * 9];
    for (int i = threadIdx.x; i < 2 * 64 * 9; i += 256) ws[i] = w[i];
    __syncthreads();
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int cq = (int)(i & 15);
    long r = i >> 4;
    const int wp = wd + 2, hp = h + 2;
    const int jp = (int)(r % wp);
    r /= wp;
    const int ip = (int)(r % hp);
    const long n = r / hp;
    const float* __restrict__ d0 = dy + n * 2 * h * (long)wd;
    const float* __restrict__ d1 = d0 + h * (long)wd;
    float s[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const int y = ip - 1 - (t / 3 - 1), x = jp - 1 - (t % 3 - 1);     // output pixel whose tap t reads padded pixel (ip-1, jp-1)
        if (y >= 0 && y < h && x >= 0 && x < wd) {
            const float g0 = d0[(long)y * wd + x], g1 = d1[(long)y * wd + x];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int ci = 4 * cq + k;
                s[k] += g0 * ws[(0 * 64 + ci) * 9 + t] + g1 * ws[(1 * 64 + ci) * 9 + t];
            }
        }
    }
    reinterpret_cast<float4*>(dxp)[i] = make_float4(s[0], s[1], s[2], s[3]);
}
// weight / bias gradient partials: block b takes a range of image rows; thread = (tap row, input channel) with the three taps of
// that row x both output channels in registers, the row of x sliding through a three-pixel window (one load per pixel).
// Partial layout per block: [tap][co][ci] (1152) + 2 bias sums.
__global__ __launch_bounds__(192) void conv_last_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                              float* __restrict__ part, int batch, int h, int wd, int rows_per_block) {
    const long rows_total = (long)batch * h;
    const long row0 = (long)blockIdx.x * rows_per_block;
    long row1 = row0 + rows_per_block;
    if (row1 > rows_total) row1 = rows_total;
    const int tr = threadIdx.x >> 6, ci = threadIdx.x & 63;
    float acc[3][2] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};
    float bacc[2] = {0.f, 0.f};
    for (long row = row0; row < row1; ++row) {
        const long n = row / h;
        const int y = (int)(row - n * h);
        const float* __restrict__ d0 = dy + (n * 2 * h + y) * (long)wd;
        const float* __restrict__ d1 = d0 + h * (long)wd;
        const float* __restrict__ xr = x + (n * h + reflect1(y + tr - 1, h)) * (long)wd * 64 + ci;
        float xm = xr[64], x0 = xr[0];                   // reflect(-1) = 1
#pragma unroll 4
        for (int xq = 0; xq < wd; ++xq) {
            const float xp = xr[(long)reflect1(xq + 1, wd) * 64];
            const float g0 = d0[xq], g1 = d1[xq];
            acc[0][0] += g0 * xm; acc[0][1] += g1 * xm;
            acc[1][0] += g0 * x0; acc[1][1] += g1 * x0;
            acc[2][0] += g0 * xp; acc[2][1] += g1 * xp;
            bacc[0] += g0; bacc[1] += g1;
            xm = x0;
            x0 = xp;
        }
    }
    float* __restrict__ dst = part + (long)blockIdx.x * 1154;
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
        for (int co = 0; co < 2; ++co) dst[(tr * 3 + k) * 128 + co * 64 + ci] = acc[k][co];
    if (threadIdx.x < 2) dst[1152 + threadIdx.x] = bacc[threadIdx.x];
}
__global__ __launch_bounds__(256) void conv_last_wgrad_final_kernel(const float* __restrict__ part, float* __restrict__ dw,
                                                                    float* __restrict__ db, int blocks) {
    const int o = blockIdx.x * 256 + threadIdx.x;
    if (o >= 1154) return;
    double s = 0;
    for (int b = 0; b < blocks; ++b) s += part[(long)b * 1154 + o];
    if (o >= 1152) { db[o - 1152] = (float)s; return; }
    const int t = o >> 7, co = (o >> 6) & 1, ci = o & 63;
    dw[(co * 64 + ci) * 9 + t] = (float)s;             // OIHW (2, 64, 3, 3)
}

// ---- first layer (Conv2d(1,3,1x1) -> ReflPad -> Conv2d(3,64,3x3) -> ReLU; frozen encoder): gradient of its INPUT plane --------
// da: (B, h, w, 64) masked gradient of the ReLU'd output.  d3p[i][j][c] = sum_tap sum_co da[i - dy][j - dx][co] * w1[co][c][tap] on
// the padded domain (border 1), folded by the reflection's adjoint, then din = sum_c w0[c] * d3.
__global__ __launch_bounds__(256) void conv_first_dgrad_pad_kernel(const float* __restrict__ da, const float* __restrict__ w1,
                                                                   float* __restrict__ d3p, int h, int wd, long total) {
    __shared__ float ws[64 * 27];
    for (int i = threadIdx.x; i < 64 * 27; i += 256) ws[i] = w1[i];
    __syncthreads();
    const long i = (long)blockIdx.x * 256 + threadIdx.x;     // padded pixel
    if (i >= total) return;
    const int wp = wd + 2, hp = h + 2;
    long r = i;
    const int jp = (int)(r % wp);
    r /= wp;
    const int ip = (int)(r % hp);
    const long n = r / hp;
    float s[3] = {0.f, 0.f, 0.f};
    for (int t = 0; t < 9; ++t) {
        const int y = ip - 1 - (t / 3 - 1), x = jp - 1 - (t % 3 - 1);
        if (y >= 0 && y < h && x >= 0 && x < wd) {
            const float4* g = reinterpret_cast<const float4*>(da + ((n * h + y) * (long)wd + x) * 64);
#pragma unroll 4
            for (int cq = 0; cq < 16; ++cq) {
                const float4 v = g[cq];
                const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int co = 4 * cq + k;
#pragma unroll
                    for (int c = 0; c < 3; ++c) s[c] += vv[k] * ws[co * 27 + c * 9 + t];
                }
            }
        }
    }
    float* o = d3p + i * 4;
    o[0] = s[0]; o[1] = s[1]; o[2] = s[2]; o[3] = 0.f;       // (B, h+2, w+2, 4): channel 3 is padding
}
__global__ __launch_bounds__(256) void conv_first_dgrad_fold_kernel(const float* __restrict__ d3p, const float* __restrict__ w0,
                                                                    float* __restrict__ din, int h, int w, long total) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    long r = i;
    const int x = (int)(r % w);
    r /= w;
    const int y = (int)(r % h);
    const long n = r / h;
    const int hp = h + 2, wp = w + 2;
    const float4* src = reinterpret_cast<const float4*>(d3p) + n * hp * (long)wp;
    int ys[3], xs[3], ny = 0, nx = 0;
    ys[ny++] = y;
    if (y == 1) ys[ny++] = -1;
    if (y == h - 2) ys[ny++] = h;
    xs[nx++] = x;
    if (x == 1) xs[nx++] = -1;
    if (x == w - 2) xs[nx++] = w;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f;
    for (int a = 0; a < ny; ++a)
        for (int b = 0; b < nx; ++b) {
            const float4 v = src[(long)(ys[a] + 1) * wp + xs[b] + 1];
            s0 += v.x; s1 += v.y; s2 += v.z;
        }
    din[i] = w0[0] * s0 + w0[1] * s1 + w0[2] * s2;
}

// ---- calc_mean_std (function.py:4-12) of LARGE maps: the pixels split over blocks ------------------------------------------------
// holo_channel_mean_std_* runs one block per (64 channels, image): right for the 32 x 32 map of the forward path, 16 blocks for a
// 64-channel 256 x 256 map of the training losses (net.py:191-197).  Here grid (C/64, parts, B): float64 partial sums of 1024
// pixels per block, added in part order by the finishing kernel (same formulas as adain.hip).
constexpr int kWidePixels = 1024;
__global__ __launch_bounds__(256) void mean_std_wide_partial_kernel(const float* __restrict__ x, double* __restrict__ part, int hw, int c, int parts) {
    __shared__ double s_sum[16][64];
    __shared__ double s_sq[16][64];
    const int tid = threadIdx.x;
    const int cq = tid & 15, pg = tid >> 4;
    const int c0 = blockIdx.x * 64, pt = blockIdx.y, b = blockIdx.z;
    const float* __restrict__ xb = x + (size_t)b * hw * c + c0 + 4 * cq;
    const int p0 = pt * kWidePixels;
    const int p1 = p0 + kWidePixels < hw ? p0 + kWidePixels : hw;
    double s[4] = {0, 0, 0, 0}, q[4] = {0, 0, 0, 0};
    for (int p = p0 + pg; p < p1; p += 16) {
        const float4 v = *reinterpret_cast<const float4*>(xb + (size_t)p * c);
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
        double* __restrict__ dst = part + (((size_t)b * parts + pt) * 2) * c + c0 + tid;
        dst[0] = ts;
        dst[c] = tq;
    }
}
__global__ __launch_bounds__(256) void mean_std_wide_final_kernel(const double* __restrict__ part, float* __restrict__ mean,
                                                                  float* __restrict__ stdv, int hw, int c, int parts, float eps, long total) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const long b = i / c;
    const int ch = (int)(i - b * c);
    double ts = 0, tq = 0;
    for (int pt = 0; pt < parts; ++pt) {
        const double* __restrict__ src = part + (((size_t)b * parts + pt) * 2) * c + ch;
        ts += src[0];
        tq += src[c];
    }
    const double n = (double)hw;
    const double mu = ts / n;
    double var = (tq - ts * mu) / (n - 1.0);            // unbiased (function.py:9)
    var = var > 0 ? var : 0;
    mean[i] = (float)mu;
    stdv[i] = sqrtf((float)var + eps);
}

// ---- calc_mean_std backward (function.py:4-12): mean = sum x / HW, std = sqrt(var_unbiased + eps) ----------------------------
// dx[p][c] = dmean[c] / HW + dstd[c] * (x[p][c] - mean[c]) / ((HW - 1) * std[c])
__global__ __launch_bounds__(256) void mean_std_bwd_kernel(const float* __restrict__ x, const float* __restrict__ mean,
                                                           const float* __restrict__ stdv, const float* __restrict__ dmean,
                                                           const float* __restrict__ dstd, float* __restrict__ dx, int hw, int c4,
                                                           long total) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int cq = (int)(i % c4);
    const long n = i / ((long)hw * c4);
    const long sc = n * c4 + cq;
    const float4 v = reinterpret_cast<const float4*>(x)[i];
    const float4 m = reinterpret_cast<const float4*>(mean)[sc], s = reinterpret_cast<const float4*>(stdv)[sc];
    const float4 dm = reinterpret_cast<const float4*>(dmean)[sc], ds = reinterpret_cast<const float4*>(dstd)[sc];
    const float ihw = 1.0f / (float)hw, ih1 = 1.0f / (float)(hw - 1);
    float4 o;
    o.x = dm.x * ihw + ds.x * (v.x - m.x) * ih1 / s.x;
    o.y = dm.y * ihw + ds.y * (v.y - m.y) * ih1 / s.y;
    o.z = dm.z * ihw + ds.z * (v.z - m.z) * ih1 / s.z;
    o.w = dm.w * ihw + ds.w * (v.w - m.w) * ih1 / s.w;
    reinterpret_cast<float4*>(dx)[i] = o;
}

// ---- distance head backward (net.py:288-308): Linear, per-row norm + ReLU, sigmoid -------------------------------------------
// Linear y = x W^T + b with x (B, fin), W (fout, fin): dx = dy W; dW = dy^T x; db = sum_b dy.
// dx: block = 64 input features x 16 waves that split the output features; a lane reads its column of W once for up to 8 rows of
// the batch (one thread per (row, feature) walked all of W once per row: 0.09-0.24 ms a call at 16 rows); the 16 partial sums of
// a (row, feature) are added in wave order through LDS.
constexpr int kLinRows = 8;
__global__ __launch_bounds__(1024) void linear_bwd_dx_kernel(const float* __restrict__ dy, const float* __restrict__ w,
                                                             float* __restrict__ dx, int b, int fin, int fout) {
    __shared__ float red[16][kLinRows][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int k = blockIdx.x * 64 + lane;
    const int r0 = blockIdx.y * kLinRows;
    const int per = (fout + 15) / 16;
    const int o0 = wave * per, o1 = o0 + per < fout ? o0 + per : fout;
    float acc[kLinRows];
#pragma unroll
    for (int r = 0; r < kLinRows; ++r) acc[r] = 0.f;
    if (k < fin) {
        for (int o = o0; o < o1; ++o) {
            const float wv = w[(long)o * fin + k];
#pragma unroll
            for (int r = 0; r < kLinRows; ++r) {
                const int row = r0 + r < b ? r0 + r : b - 1;
                acc[r] = fmaf(dy[(long)row * fout + o], wv, acc[r]);
            }
        }
    }
#pragma unroll
    for (int r = 0; r < kLinRows; ++r) red[wave][r][lane] = acc[r];
    __syncthreads();
    for (int i = threadIdx.x; i < kLinRows * 64; i += 1024) {
        const int r = i >> 6, l = i & 63;
        float s = 0.f;
#pragma unroll
        for (int wv = 0; wv < 16; ++wv) s += red[wv][r][l];
        const int kk = blockIdx.x * 64 + l;
        if (r0 + r < b && kk < fin) dx[(long)(r0 + r) * fin + kk] = s;
    }
}
__global__ __launch_bounds__(256) void linear_bwd_dw_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                            float* __restrict__ dw, float* __restrict__ db, int b, int fin, int fout) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;   // (output feature, input feature)
    if (i >= (long)fout * fin) return;
    const int o = (int)(i / fin), k = (int)(i % fin);
    float s = 0.f;
    for (int row = 0; row < b; ++row) s += dy[row * fout + o] * x[row * fin + k];
    dw[i] = s;
    if (k == 0) {
        float t = 0.f;
        for (int row = 0; row < b; ++row) t += dy[row * fout + o];
        db[o] = t;
    }
}
// y = relu((z - mean_row) / sqrt(var_row_biased + eps)): one block per row
__global__ __launch_bounds__(256) void rownorm_relu_bwd_kernel(const float* __restrict__ z, const float* __restrict__ dy,
                                                               float* __restrict__ dz, int f, float eps) {
    __shared__ double red[256];
    const int row = blockIdx.x, tid = threadIdx.x;
    const float* zr = z + (long)row * f;
    const float* dr = dy + (long)row * f;
    double s = 0;
    for (int k = tid; k < f; k += 256) s += zr[k];
    red[tid] = s;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) { if (tid < st) red[tid] += red[tid + st]; __syncthreads(); }
    const double mean = red[0] / f;
    __syncthreads();
    s = 0;
    for (int k = tid; k < f; k += 256) { const double d = zr[k] - mean; s += d * d; }
    red[tid] = s;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) { if (tid < st) red[tid] += red[tid + st]; __syncthreads(); }
    const double inv = 1.0 / sqrt(red[0] / f + (double)eps);
    __syncthreads();
    // g = dy masked by the ReLU; dz = inv * (g - mean(g) - xhat * mean(g * xhat))
    double sg = 0, sgx = 0;
    for (int k = tid; k < f; k += 256) {
        const double xh = (zr[k] - mean) * inv;
        const double g = xh > 0 ? dr[k] : 0.0;
        sg += g; sgx += g * xh;
    }
    red[tid] = sg;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) { if (tid < st) red[tid] += red[tid + st]; __syncthreads(); }
    const double mg = red[0] / f;
    __syncthreads();
    red[tid] = sgx;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) { if (tid < st) red[tid] += red[tid + st]; __syncthreads(); }
    const double mgx = red[0] / f;
    for (int k = tid; k < f; k += 256) {
        const double xh = (zr[k] - mean) * inv;
        const double g = xh > 0 ? dr[k] : 0.0;
        dz[(long)row * f + k] = (float)(inv * (g - mg - xh * mgx));
    }
}
__global__ __launch_bounds__(256) void sigmoid_bwd_kernel(const float* __restrict__ y, const float* __restrict__ dy, float* __restrict__ dz, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) dz[i] = dy[i] * y[i] * (1.0f - y[i]);
}

// nn.Dropout in training mode (net.py:298-304: on each Linear's output, before the row norm): y = x * keep * scale, keep in {0, 1};
// the same kernel carries the gradient back
__global__ __launch_bounds__(256) void mask_scale_kernel(const float* __restrict__ x, const float* __restrict__ keep, float scale,
                                                         float* __restrict__ y, long n) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) y[i] = x[i] * keep[i] * scale;
}

static inline unsigned blocks_for(long total) { return (unsigned)((total + 255) / 256); }

// the mosaic of embedded images (relu_pool_bwd_pad_kernel): rows between image blocks, images per mosaic row, columns between
// them, and the buffer's size in pixels
struct Mosaic {
    int pitch, gx, cpitch, width;
    long rows;
};
static inline bool mosaic_of(int batch, int h, int w, int border, int pitch_rows, int per_row, Mosaic& m) {
    m.pitch = pitch_rows > 0 ? pitch_rows : h + 2 * border;
    m.gx = per_row > 1 ? per_row : 1;
    m.cpitch = m.gx > 1 ? w + border : w + 2 * border;
    m.width = (m.gx - 1) * m.cpitch + w + 2 * border;
    m.rows = (long)((batch + m.gx - 1) / m.gx - 1) * m.pitch + h + 2 * border;
    return m.pitch >= h + border && m.pitch <= h + 2 * border && (m.gx == 1 || border >= 1);
}

constexpr int kFoldKeep = 64, kFoldBy = 32;
constexpr int kPadBlocks = 2048;     // blocks of the mask / pad kernel when it also sums the bias gradient
// rows of scratch behind `rows` partial rows that fold_rows() may use
static inline long fold_scratch_rows(long rows) {
    long extra = 0;
    while (rows > kFoldKeep) {
        rows = (rows + kFoldBy - 1) / kFoldBy;
        extra += rows;
    }
    return extra;
}
// folds part[rows][elems] (scratch directly behind it) until rows <= kFoldKeep; returns where the remaining rows are
template <typename T>
static const T* fold_rows(T* part, long elems, int& rows, hipStream_t stream) {
    T* in = part;
    T* out = part + (long)rows * elems;
    while (rows > kFoldKeep) {
        const int chunks = (rows + kFoldBy - 1) / kFoldBy;
        hipLaunchKernelGGL((fold_rows_kernel<T>), dim3(blocks_for(elems), chunks), dim3(256), 0, stream, in, out, elems, rows, kFoldBy);
        in = out;
        out += (long)chunks * elems;
        rows = chunks;
    }
    return in;
}

// partial layout of pix_gemm: slots = ceil(rows / rows_per_wave) rounded up to a multiple of 4 (4 waves per block)
struct PixGemmPlan {
    int rows_per_wave, splits;
    long slots;
};
static inline int pix_gemm_nt(int nc, int taps) { return taps == 1 && nc % 128 == 0 ? 4 : 1; }
// HOLO_PIX_GEMM_DIRECT=1: the operands-from-global kernel for every shape (A/B timing, tests of both kernels on one shape)
static inline bool pix_gemm_force_direct() {
    const char* e = getenv("HOLO_PIX_GEMM_DIRECT");
    return e && e[0] == '1';
}
static inline bool pix_gemm_lds(int w, int mc, int nc, int taps) { return taps == 9 && w % kPgChunk == 0 && mc % 64 == 0 && nc % 64 == 0; }
// the LDS-staged kernel's units: (image, 32-pixel strip, row group); about two blocks per CU over the grid
struct PixGemmLdsPlan {
    int rows_per_unit, groups;
    long units;
};
static PixGemmLdsPlan pix_gemm_lds_plan(int batch, int h, int w, int mc, int nc) {
    const long tiles = (long)(mc / 64) * (nc / 64);
    const long strips = (long)batch * (w / kPgChunk);
    const long target = 2L * device_cus();          // one resident round of blocks (two per CU): 512 measured ahead of 1024 and 2048
    long groups = (target + tiles * strips - 1) / (tiles * strips);
    if (groups > h / 2) groups = h / 2;
    if (groups < 1) groups = 1;
    PixGemmLdsPlan pl;
    pl.rows_per_unit = (int)((h + groups - 1) / groups);
    pl.groups = (h + pl.rows_per_unit - 1) / pl.rows_per_unit;
    pl.units = strips * pl.groups;
    return pl;
}
static PixGemmPlan pix_gemm_plan(int batch, int h, int mc, int nc, int nt) {
    const long rows = (long)batch * h;
    const long tiles = (long)(mc / 32) * (nc / (32 * nt));
    // aim at ~8 blocks per CU over the whole grid, at least 2 image rows per wave
    long splits = (2048 + tiles - 1) / tiles;
    if (splits < 1) splits = 1;
    long rpw = (rows + splits * 4 - 1) / (splits * 4);
    if (rpw < 2) rpw = 2;
    splits = (rows + rpw * 4 - 1) / (rpw * 4);
    PixGemmPlan p;
    p.rows_per_wave = (int)rpw;
    p.splits = (int)splits;
    p.slots = splits * 4;
    return p;
}

}  // namespace holo

extern "C" {

using namespace holo;

int holo_maxpool2x2_f32(const float* x_nhwc, float* y_nhwc, int batch, int h, int w, int c, void* stream) {
    HOLO_REQUIRE(x_nhwc && y_nhwc, HOLO_ERR_ARG, "holo_maxpool2x2_f32: null pointer");
    HOLO_REQUIRE(batch > 0 && h > 0 && w > 0 && c > 0 && c % 4 == 0, HOLO_ERR_ARG, "holo_maxpool2x2_f32: batch=%d h=%d w=%d c=%d (c %% 4)", batch, h, w, c);
    const int ho = (h + 1) / 2, wo = (w + 1) / 2;
    const long total = (long)batch * ho * wo * (c / 4);
    hipLaunchKernelGGL(maxpool2x2_kernel, dim3(blocks_for(total)), dim3(256), 0, as_stream(stream), x_nhwc, y_nhwc, h, w, c / 4, ho, wo, total);
    return check_launch("holo_maxpool2x2_f32");
}

size_t holo_relu_pool_bwd_pad_workspace_bytes(int c) {
    return c > 0 ? (size_t)(holo::kPadBlocks + holo::fold_scratch_rows(holo::kPadBlocks)) * c * sizeof(double) : 0;
}
int holo_relu_pool_bwd_pad_f32(const float* dy, const float* y_act, const float* y_pool, float* out_padded, int batch, int h, int w,
                               int c, int relu, int border, int pitch_rows, int per_row, float* dbias, void* workspace,
                               size_t workspace_bytes, void* stream) {
    HOLO_REQUIRE(dy && out_padded && (y_act || (!relu && !y_pool)), HOLO_ERR_ARG, "holo_relu_pool_bwd_pad_f32: null pointer");
    HOLO_REQUIRE(batch > 0 && h > 0 && w > 0 && c > 0 && c % 4 == 0 && border >= 0 && border <= 2, HOLO_ERR_ARG,
                 "holo_relu_pool_bwd_pad_f32: batch=%d h=%d w=%d c=%d border=%d", batch, h, w, c, border);
    Mosaic mo;
    HOLO_REQUIRE(mosaic_of(batch, h, w, border, pitch_rows, per_row, mo), HOLO_ERR_ARG,
                 "holo_relu_pool_bwd_pad_f32: pitch_rows=%d outside [h + border, h + 2*border] = [%d, %d] (per_row=%d needs border >= 1)", pitch_rows,
                 h + border, h + 2 * border, per_row);
    HOLO_REQUIRE(256 % (c / 4) == 0 || !dbias, HOLO_ERR_ARG, "holo_relu_pool_bwd_pad_f32: the bias gradient needs c in {4,...,1024} with 1024 %% c == 0 (c=%d)", c);
    HOLO_REQUIRE(!dbias || (workspace && workspace_bytes >= holo_relu_pool_bwd_pad_workspace_bytes(c)), HOLO_ERR_WORKSPACE,
                 "holo_relu_pool_bwd_pad_f32: the bias gradient needs a workspace of %zu bytes", holo_relu_pool_bwd_pad_workspace_bytes(c));
    const long total = mo.rows * mo.width * (c / 4);
    const unsigned blocks = blocks_for(total) < (unsigned)kPadBlocks || !dbias ? blocks_for(total) : (unsigned)kPadBlocks;
    hipLaunchKernelGGL(relu_pool_bwd_pad_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), dy, y_act, y_pool, out_padded, batch,
                       h, w, c / 4, relu, border, mo.pitch, mo.gx, mo.cpitch, mo.width, total, dbias ? static_cast<double*>(workspace) : nullptr);
    if (dbias) {
        int left = (int)blocks;
        const double* folded = fold_rows(static_cast<double*>(workspace), c, left, as_stream(stream));
        hipLaunchKernelGGL(channel_sum_final_kernel, dim3((c + 255) / 256), dim3(256), 0, as_stream(stream), folded, dbias, c, left);
    }
    return check_launch("holo_relu_pool_bwd_pad_f32");
}

int holo_fold_reflect_f32(const float* dxp, float* dx, int batch, int h, int w, int c, int border, int pitch_rows, int per_row,
                          void* stream) {
    HOLO_REQUIRE(dxp && dx, HOLO_ERR_ARG, "holo_fold_reflect_f32: null pointer");
    HOLO_REQUIRE(batch > 0 && h >= 2 && w >= 2 && c > 0 && c % 4 == 0 && border >= 1 && border <= 2, HOLO_ERR_ARG,
                 "holo_fold_reflect_f32: batch=%d h=%d w=%d c=%d border=%d (reflection padding needs h,w >= 2)", batch, h, w, c, border);
    Mosaic mo;
    HOLO_REQUIRE(mosaic_of(batch, h, w, border, pitch_rows, per_row, mo), HOLO_ERR_ARG,
                 "holo_fold_reflect_f32: pitch_rows=%d outside [h + border, h + 2*border] = [%d, %d]", pitch_rows, h + border, h + 2 * border);
    const long total = (long)batch * h * w * (c / 4);
    hipLaunchKernelGGL(fold_reflect_kernel, dim3(blocks_for(total)), dim3(256), 0, as_stream(stream), dxp, dx, h, w, c / 4, border, mo.pitch, mo.gx, mo.cpitch, mo.width, total);
    return check_launch("holo_fold_reflect_f32");
}

int holo_space_to_depth2_relu_f32(const float* dy, const float* y, float* g, int batch, int h, int w, int c, int relu, void* stream) {
    HOLO_REQUIRE(dy && g && (y || !relu), HOLO_ERR_ARG, "holo_space_to_depth2_relu_f32: null pointer");
    HOLO_REQUIRE(batch > 0 && h > 0 && w > 0 && c > 0 && c % 4 == 0, HOLO_ERR_ARG, "holo_space_to_depth2_relu_f32: batch=%d h=%d w=%d c=%d", batch, h, w, c);
    const long total = (long)batch * h * w * 4 * (c / 4);
    hipLaunchKernelGGL(space_to_depth2_relu_kernel, dim3(blocks_for(total)), dim3(256), 0, as_stream(stream), dy, y, g, h, w, c / 4, relu, total);
    return check_launch("holo_space_to_depth2_relu_f32");
}

size_t holo_channel_sum_workspace_bytes(long rows, int c) {
    if (rows <= 0 || c <= 0) return 0;
    const int parts = rows < 4096 ? 1 : (rows < (1L << 20) ? 64 : 512);
    return (size_t)(parts + fold_scratch_rows(parts)) * c * sizeof(double);
}
int holo_channel_sum_f32(const float* x, float* out, long rows, int c, void* workspace, size_t workspace_bytes, void* stream) {
    HOLO_REQUIRE(x && out && workspace, HOLO_ERR_ARG, "holo_channel_sum_f32: null pointer");
    HOLO_REQUIRE(rows > 0 && c > 0 && c % 64 == 0, HOLO_ERR_ARG, "holo_channel_sum_f32: rows=%ld c=%d (c %% 64)", rows, c);
    HOLO_REQUIRE(workspace_bytes >= holo_channel_sum_workspace_bytes(rows, c), HOLO_ERR_WORKSPACE, "holo_channel_sum_f32: workspace too small");
    const int parts = rows < 4096 ? 1 : (rows < (1L << 20) ? 64 : 512);
    hipLaunchKernelGGL(channel_sum_partial_kernel, dim3(c / 64, parts), dim3(256), 0, as_stream(stream), x, static_cast<double*>(workspace), rows, c, parts);
    int left = parts;
    const double* folded = fold_rows(static_cast<double*>(workspace), c, left, as_stream(stream));
    hipLaunchKernelGGL(channel_sum_final_kernel, dim3((c + 255) / 256), dim3(256), 0, as_stream(stream), folded, out, c, left);
    return check_launch("holo_channel_sum_f32");
}

size_t holo_pix_gemm_workspace_bytes(int batch, int h, int w, int mc, int nc, int taps) {
    if (batch <= 0 || h <= 0 || w <= 0 || mc <= 0 || nc <= 0 || (taps != 1 && taps != 9)) return 0;
    const PixGemmPlan p = pix_gemm_plan(batch, h, mc, nc, pix_gemm_nt(nc, taps));
    long slots = p.slots;                              // room for either kernel's partial tiles
    if (pix_gemm_lds(w, mc, nc, taps)) slots = std::max(slots, pix_gemm_lds_plan(batch, h, w, mc, nc).units);
    return (size_t)(slots + fold_scratch_rows(slots)) * mc * nc * taps * sizeof(float);
}
/* out[m][n][tap] = sum over the B*h*w pixels of P[pixel][m] * Q[reflect(pixel + tap)][n]; taps = 9 (3x3, tap = ky*3+kx, reflected
 * neighbour) or 1 (the pixel itself).  P: (B, h + 2*p_border, w + 2*p_border, mc) NHWC read at its interior, Q: (B, h, w, nc). */
int holo_pix_gemm_f32(const float* p, int p_border, int p_pitch_rows, int p_per_row, const float* q, float* out, int batch, int h, int w,
                      int mc, int nc, int taps, void* workspace, size_t workspace_bytes, void* stream) {
    HOLO_REQUIRE(p && q && out && workspace, HOLO_ERR_ARG, "holo_pix_gemm_f32: null pointer");
    HOLO_REQUIRE(batch > 0 && h >= 2 && w >= 2 && mc > 0 && nc > 0 && mc % 32 == 0 && nc % 32 == 0 && (taps == 1 || taps == 9) && p_border >= 0,
                 HOLO_ERR_ARG, "holo_pix_gemm_f32: batch=%d h=%d w=%d mc=%d nc=%d taps=%d (channels %% 32, taps 1 | 9)", batch, h, w, mc, nc, taps);
    HOLO_REQUIRE(workspace_bytes >= holo_pix_gemm_workspace_bytes(batch, h, w, mc, nc, taps), HOLO_ERR_WORKSPACE, "holo_pix_gemm_f32: workspace too small");
    Mosaic mo;
    HOLO_REQUIRE(mosaic_of(batch, h, w, p_border, p_pitch_rows, p_per_row, mo), HOLO_ERR_ARG,
                 "holo_pix_gemm_f32: p_pitch_rows=%d outside [h + border, h + 2*border] = [%d, %d]", p_pitch_rows, h + p_border, h + 2 * p_border);
    float* part = static_cast<float*>(workspace);
    const long elems = (long)mc * nc * taps;
    if (pix_gemm_lds(w, mc, nc, taps) && !pix_gemm_force_direct()) {
        const PixGemmLdsPlan lp = pix_gemm_lds_plan(batch, h, w, mc, nc);
        HOLO_REQUIRE(lp.units <= 65535, HOLO_ERR_ARG, "holo_pix_gemm_f32: %ld work units (grid limit 65535)", lp.units);
        hipLaunchKernelGGL(pix_gemm9_lds_kernel, dim3(mc / 64, nc / 64, (unsigned)lp.units), dim3(256), 0, as_stream(stream), p, p_border, q, part,
                           h, w, mc, nc, lp.rows_per_unit, lp.groups, mo.pitch, mo.gx, mo.cpitch, mo.width);
        int left = (int)lp.units;
        const float* folded = fold_rows(part, elems, left, as_stream(stream));
        hipLaunchKernelGGL(pix_gemm_reduce_kernel, dim3(blocks_for(elems)), dim3(256), 0, as_stream(stream), folded, out, elems, left, taps, 1);
        return check_launch("holo_pix_gemm_f32");
    }
    const int nt = pix_gemm_nt(nc, taps);
    const PixGemmPlan pl = pix_gemm_plan(batch, h, mc, nc, nt);
    const dim3 grid(mc / 32, nc / (32 * nt), pl.splits);
    if (taps == 9)
        hipLaunchKernelGGL((pix_gemm_kernel<9, 1>), grid, dim3(256), 0, as_stream(stream), p, p_border, q, part, batch, h, w, mc, nc, pl.rows_per_wave, mo.pitch, mo.gx, mo.cpitch, mo.width);
    else if (nt == 4)
        hipLaunchKernelGGL((pix_gemm_kernel<1, 4>), grid, dim3(256), 0, as_stream(stream), p, p_border, q, part, batch, h, w, mc, nc, pl.rows_per_wave, mo.pitch, mo.gx, mo.cpitch, mo.width);
    else
        hipLaunchKernelGGL((pix_gemm_kernel<1, 1>), grid, dim3(256), 0, as_stream(stream), p, p_border, q, part, batch, h, w, mc, nc, pl.rows_per_wave, mo.pitch, mo.gx, mo.cpitch, mo.width);
    int left = (int)pl.slots;
    const float* folded = fold_rows(part, elems, left, as_stream(stream));
    hipLaunchKernelGGL(pix_gemm_reduce_kernel, dim3(blocks_for(elems)), dim3(256), 0, as_stream(stream), folded, out, elems, left, taps, 0);
    return check_launch("holo_pix_gemm_f32");
}

size_t holo_conv_last_bwd_workspace_bytes(int batch, int h, int w) {
    if (batch <= 0 || h <= 0 || w <= 0) return 0;
    const long rows = (long)batch * h;
    const long blocks = rows < 8192 ? rows : 8192;
    return (size_t)(blocks + fold_scratch_rows(blocks)) * 1154 * sizeof(float);
}
/* decoder[36:38] backward: dy (B,2,h,w) planes; x (B,h,w,64) the layer's input; dxp (B,h+2,w+2,64) the data gradient on the padded
 * domain (fold with holo_fold_reflect_f32, border 1), dw (2,64,3,3), db (2).  dxp / dw+db may be NULL to skip either part. */
int holo_conv_last_bwd_f32(const float* dy, const float* x_nhwc, const float* w_oihw, float* dxp, float* dw, float* db, int batch, int h,
                           int w, void* workspace, size_t workspace_bytes, void* stream) {
    HOLO_REQUIRE(dy && w_oihw, HOLO_ERR_ARG, "holo_conv_last_bwd_f32: null pointer");
    HOLO_REQUIRE(batch > 0 && h >= 2 && w >= 2, HOLO_ERR_ARG, "holo_conv_last_bwd_f32: batch=%d h=%d w=%d", batch, h, w);
    if (dxp) {
        const long total = (long)batch * (h + 2) * (w + 2) * 16;
        hipLaunchKernelGGL(conv_last_dgrad_kernel, dim3(blocks_for(total)), dim3(256), 0, as_stream(stream), dy, w_oihw, dxp, h, w, total);
    }
    if (dw) {
        HOLO_REQUIRE(db && x_nhwc && workspace, HOLO_ERR_ARG, "holo_conv_last_bwd_f32: weight gradient needs x, db and a workspace");
        HOLO_REQUIRE(workspace_bytes >= holo_conv_last_bwd_workspace_bytes(batch, h, w), HOLO_ERR_WORKSPACE, "holo_conv_last_bwd_f32: workspace too small");
        const long rows = (long)batch * h;
        const int blocks = (int)(rows < 8192 ? rows : 8192);
        const int rpb = (int)((rows + blocks - 1) / blocks);
        const int nb = (int)((rows + rpb - 1) / rpb);
        hipLaunchKernelGGL(conv_last_wgrad_kernel, dim3(nb), dim3(192), 0, as_stream(stream), x_nhwc, dy, static_cast<float*>(workspace), batch, h, w, rpb);
        int left = nb;
        const float* folded = fold_rows(static_cast<float*>(workspace), 1154, left, as_stream(stream));
        hipLaunchKernelGGL(conv_last_wgrad_final_kernel, dim3(5), dim3(256), 0, as_stream(stream), folded, dw, db, left);
    }
    return check_launch("holo_conv_last_bwd_f32");
}

/* vgg[0:4] backward, data gradient only (the encoder is frozen, net.py:167-170): da (B,h,w,64) = gradient of the ReLU'd output
 * already masked; din (B,1,h,w); workspace (B,h+2,w+2,4) floats */
int holo_conv_first_dgrad_f32(const float* da, const float* w0, const float* w1, float* din, int batch, int h, int w, void* workspace,
                              size_t workspace_bytes, void* stream) {
    HOLO_REQUIRE(da && w0 && w1 && din && workspace, HOLO_ERR_ARG, "holo_conv_first_dgrad_f32: null pointer");
    HOLO_REQUIRE(batch > 0 && h >= 2 && w >= 2, HOLO_ERR_ARG, "holo_conv_first_dgrad_f32: batch=%d h=%d w=%d", batch, h, w);
    HOLO_REQUIRE(workspace_bytes >= (size_t)batch * (h + 2) * (w + 2) * 4 * sizeof(float), HOLO_ERR_WORKSPACE, "holo_conv_first_dgrad_f32: workspace too small");
    const long tp = (long)batch * (h + 2) * (w + 2);
    hipLaunchKernelGGL(conv_first_dgrad_pad_kernel, dim3(blocks_for(tp)), dim3(256), 0, as_stream(stream), da, w1, static_cast<float*>(workspace), h, w, tp);
    const long t = (long)batch * h * w;
    hipLaunchKernelGGL(conv_first_dgrad_fold_kernel, dim3(blocks_for(t)), dim3(256), 0, as_stream(stream), static_cast<const float*>(workspace), w0, din, h, w, t);
    return check_launch("holo_conv_first_dgrad_f32");
}

size_t holo_channel_mean_std_wide_workspace_bytes(int batch, int hw, int c) {
    if (batch <= 0 || hw <= 0 || c <= 0) return 0;
    const size_t parts = (size_t)(hw + holo::kWidePixels - 1) / holo::kWidePixels;
    return (size_t)batch * parts * 2 * c * sizeof(double);
}
/* calc_mean_std (function.py:4-12) of an NHWC fp32 map with the pixels split over blocks (large maps at small batch: the training
 * losses, net.py:191-197); same formulas as holo_channel_mean_std_f32, partial sums added in a fixed order */
int holo_channel_mean_std_wide_f32(const float* x_nhwc, float* mean, float* std, int batch, int hw, int c, float eps, void* workspace,
                                   size_t workspace_bytes, void* stream) {
    HOLO_REQUIRE(x_nhwc && mean && std && workspace, HOLO_ERR_ARG, "holo_channel_mean_std_wide_f32: null pointer");
    HOLO_REQUIRE(batch > 0 && batch <= 65535 && hw >= 2 && c > 0 && c % 64 == 0, HOLO_ERR_ARG,
                 "holo_channel_mean_std_wide_f32: batch=%d hw=%d c=%d (c must be a multiple of 64, batch <= 65535)", batch, hw, c);
    HOLO_REQUIRE(workspace_bytes >= holo_channel_mean_std_wide_workspace_bytes(batch, hw, c), HOLO_ERR_WORKSPACE,
                 "holo_channel_mean_std_wide_f32: workspace too small");
    const int parts = (hw + kWidePixels - 1) / kWidePixels;
    HOLO_REQUIRE(parts <= 65535, HOLO_ERR_ARG, "holo_channel_mean_std_wide_f32: hw=%d too large", hw);
    hipLaunchKernelGGL(mean_std_wide_partial_kernel, dim3(c / 64, parts, batch), dim3(256), 0, as_stream(stream), x_nhwc,
                       static_cast<double*>(workspace), hw, c, parts);
    const long total = (long)batch * c;
    hipLaunchKernelGGL(mean_std_wide_final_kernel, dim3(blocks_for(total)), dim3(256), 0, as_stream(stream),
                       static_cast<const double*>(workspace), mean, std, hw, c, parts, eps, total);
    return check_launch("holo_channel_mean_std_wide_f32");
}

int holo_mean_std_bwd_f32(const float* x_nhwc, const float* mean, const float* stdv, const float* dmean, const float* dstd, float* dx,
                          int batch, int hw, int c, void* stream) {
    HOLO_REQUIRE(x_nhwc && mean && stdv && dmean && dstd && dx, HOLO_ERR_ARG, "holo_mean_std_bwd_f32: null pointer");
    HOLO_REQUIRE(batch > 0 && hw > 1 && c > 0 && c % 4 == 0, HOLO_ERR_ARG, "holo_mean_std_bwd_f32: batch=%d hw=%d c=%d", batch, hw, c);
    const long total = (long)batch * hw * (c / 4);
    hipLaunchKernelGGL(mean_std_bwd_kernel, dim3(blocks_for(total)), dim3(256), 0, as_stream(stream), x_nhwc, mean, stdv, dmean, dstd, dx, hw, c / 4, total);
    return check_launch("holo_mean_std_bwd_f32");
}

/* Linear backward: dx (B,fin) (or NULL), dw (fout,fin) + db (fout) (or NULL) */
int holo_linear_bwd_f32(const float* dy, const float* x, const float* w, float* dx, float* dw, float* db, int batch, int fin, int fout, void* stream) {
    HOLO_REQUIRE(dy && (dx || dw), HOLO_ERR_ARG, "holo_linear_bwd_f32: null pointer");
    HOLO_REQUIRE(batch > 0 && fin > 0 && fout > 0, HOLO_ERR_ARG, "holo_linear_bwd_f32: batch=%d fin=%d fout=%d", batch, fin, fout);
    if (dx) {
        HOLO_REQUIRE(w, HOLO_ERR_ARG, "holo_linear_bwd_f32: dx needs w");
        hipLaunchKernelGGL(linear_bwd_dx_kernel, dim3((fin + 63) / 64, (batch + kLinRows - 1) / kLinRows), dim3(1024), 0, as_stream(stream), dy, w,
                           dx, batch, fin, fout);
    }
    if (dw) {
        HOLO_REQUIRE(x && db, HOLO_ERR_ARG, "holo_linear_bwd_f32: dw needs x and db");
        hipLaunchKernelGGL(linear_bwd_dw_kernel, dim3(blocks_for((long)fout * fin)), dim3(256), 0, as_stream(stream), dy, x, dw, db, batch, fin, fout);
    }
    return check_launch("holo_linear_bwd_f32");
}
int holo_rownorm_relu_bwd_f32(const float* z, const float* dy, float* dz, int batch, int f, float eps, void* stream) {
    HOLO_REQUIRE(z && dy && dz, HOLO_ERR_ARG, "holo_rownorm_relu_bwd_f32: null pointer");
    HOLO_REQUIRE(batch > 0 && f > 0, HOLO_ERR_ARG, "holo_rownorm_relu_bwd_f32: batch=%d f=%d", batch, f);
    hipLaunchKernelGGL(rownorm_relu_bwd_kernel, dim3(batch), dim3(256), 0, as_stream(stream), z, dy, dz, f, eps);
    return check_launch("holo_rownorm_relu_bwd_f32");
}
int holo_sigmoid_bwd_f32(const float* y, const float* dy, float* dz, int n, void* stream) {
    HOLO_REQUIRE(y && dy && dz && n > 0, HOLO_ERR_ARG, "holo_sigmoid_bwd_f32: bad argument");
    hipLaunchKernelGGL(sigmoid_bwd_kernel, dim3(blocks_for(n)), dim3(256), 0, as_stream(stream), y, dy, dz, n);
    return check_launch("holo_sigmoid_bwd_f32");
}

/* y = x * keep * scale elementwise: nn.Dropout's training-mode forward (keep = the drawn 0/1 mask, scale = 1/(1-p)) and backward */
int holo_mask_scale_f32(const float* x, const float* keep, float scale, float* y, long n, void* stream) {
    HOLO_REQUIRE(x && keep && y && n > 0, HOLO_ERR_ARG, "holo_mask_scale_f32: bad argument");
    hipLaunchKernelGGL(mask_scale_kernel, dim3(blocks_for(n)), dim3(256), 0, as_stream(stream), x, keep, scale, y, n);
    return check_launch("holo_mask_scale_f32");
}

}  // extern "C"
