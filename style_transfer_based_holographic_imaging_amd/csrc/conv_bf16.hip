// bf16 3x3 convolutions on v_mfma_f32_16x16x32_bf16 (BASELINE configs[2]: "bf16 convs / fp32 FFT").
//
// Replaces  ReflectionPad2d(1) -> Conv2d(3x3) -> ReLU [-> MaxPool2d(2,2,ceil_mode=True)]  of net.py:78-107 / net.py:34-71 with
// bf16 activations and weights and fp32 accumulation, like conv_mfma_kernel<bf16> (conv_mfma.hip), which it supersedes for the
// layers it covers; same NHWC activations, same packed weights [Cin/32][tap][Cout][32], same results (every accumulator sums its
// 9*Cin products in the same order: chunk-major, tap, then k; the 16x16x32 instruction adds 32 products per step where the
// 32x32x16 one adds 16 — fp32 rounding of partial sums only).
//
// What is different, and why (profiles/r04_bf16_tile_sweep.txt):
//   * 8-wave blocks of 512 pixels x 128 output channels: one staged weight tile (8 KB per 32-channel chunk and tap) now feeds 512
//     pixels instead of 128.  The old tile moved 36 B/clk/CU from L2 into LDS at full matrix rate and ran at 45 %; this one 12.6.
//   * v_mfma_f32_16x16x32_bf16: the chip holds a higher clock on it than on the 32x32x16 shape (MI355X_MICROARCH.md, DVFS give-back
//     item 7).  A lane's 16-byte LDS read is a whole K = 32 fragment of one row, so a stage (tap) is 4 pixel + WTN weight reads.
//   * the product is computed transposed (D[channel][pixel]): a lane ends up with 4 consecutive channels of ONE pixel per tile;
//     v_permlane16_swap joins the quads of neighbouring 16-lane rows into 16-byte runs, so a store instruction writes 64-byte
//     runs of the NHWC pixel.
//
// LDS images (64-byte rows = 32 channels, 16-byte units XOR-swizzled so that every ds_read_b128 lane group covers 16 different
// 16-byte slots — found by enumeration, scratch/swz_search.py):
//   halo tile   [HR x 18 pixels][64 B]   unit' = unit ^ (((hx >> 1) & 1) << 1)      (needs HW = 2 mod 4)
//   weight tile [TN channels][64 B]      unit' = unit ^ g[(co >> 2) & 3], g = {0,2,3,1}
// Both are filled by LDS-DMA (global_load_lds_dwordx4: lane-linear destination, swizzle applied to the per-lane SOURCE address).
#include "common.h"

#ifndef HOLO_B16_ABLATE
#define HOLO_B16_ABLATE 0      // timing experiments only (wrong results): 1 = no staging loads in the main loop, 2 = no waits / barriers, 4 = no fragment reads, 8 = no stores
#endif

namespace holo {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 b16_t;
typedef __bf16 b16x8 __attribute__((ext_vector_type(8)));

constexpr int kB16KC = 32;     // channels per stage
constexpr int kB16HW = 18;     // halo width: 16 pixel columns + 2

// WM x WN waves; a wave owns 64 pixels (4 rows x 16 columns = 2 x 2 tiles of 2 rows x 8 columns) x (16 * WTN) output channels.
// A stage is one KERNEL ROW of a 32-channel chunk (3 taps): the weight tiles of its three taps are staged together, so a
// block meets at a barrier once per 3 * 4 * WTN MFMAs per wave (1536 cycles at WTN = 8) instead of once per tap.
template <int WM, int WN, int WTN>
struct B16Cfg {
    static constexpr int PR = 4 * WM;
    static constexpr int TN = 16 * WTN * WN;
    static constexpr int HR = PR + 2;
    static constexpr int THREADS = 64 * WM * WN;
    static constexpr int A_UNITS = HR * kB16HW * 4;
    static constexpr int A_ROUNDS = (A_UNITS + THREADS - 1) / THREADS;
    // the LDS images are padded to whole block-wide rounds: EVERY wave issues A_ROUNDS (+ BS_ROUNDS) loads, so one counted
    // s_waitcnt vmcnt(N) is right for all of them (lanes past the image load a valid dummy unit into the padding)
    static constexpr int A_BYTES = A_ROUNDS * THREADS * 16;
    static constexpr int BT_UNITS = TN * 4;                 // one tap's weight tile
    static constexpr int BT_BYTES = BT_UNITS * 16;
    static constexpr int BS_UNITS = 3 * BT_UNITS;           // one stage: three taps
    static constexpr int BS_ROUNDS = (BS_UNITS + THREADS - 1) / THREADS;
    static constexpr int BS_BYTES = BS_ROUNDS * THREADS * 16;
    static constexpr int DMA_STEP = WTN >= 8 ? 2 : 1;       // MFMA groups between two staging loads of a wave
    static constexpr int NBUF_B = 3;
    static constexpr int LDS_BYTES = 2 * A_BYTES + NBUF_B * BS_BYTES;
    static_assert(BT_UNITS % 64 == 0, "weight tiles are whole 1-KiB pieces");
    static_assert(LDS_BYTES <= 160 * 1024, "LDS");
};

__device__ __forceinline__ int b16_swz_a(int hx) { return ((hx >> 1) & 1) << 1; }
__device__ __forceinline__ int b16_swz_b(int co) { return (0x78 >> (((co >> 2) & 3) * 2)) & 3; }   // {0,2,3,1} packed two bits each: 0b01'11'10'00

// One tile = 4*WM rows x 16 columns of one image x TN output channels.  Tile ids are decoded nb-fastest (the channel blocks of a
// patch are neighbours and meet in one L2), then tx, ty, image.
struct B16Tile {
    int img, y0, x0, n0;
};
__device__ __forceinline__ B16Tile b16_decode(int t, int n_blk, int tiles_x, int tiles_y, int pr, int tn) {
    B16Tile o;
    const int nb = t % n_blk;
    t /= n_blk;
    const int tx = t % tiles_x;
    t /= tiles_x;
    const int ty = t % tiles_y;
    o.img = t / tiles_y;
    o.y0 = ty * pr;
    o.x0 = tx * 16;
    o.n0 = nb * tn;
    return o;
}

// PERSISTENT blocks (one per CU: the LDS images take 150 KB): a block walks tiles L, L + grid, ... and its staging runs as ONE
// stream across tile boundaries — the first halo tile and the first two weight stages of the next tile are requested during the
// last stages of the current one, exactly as inside a tile.  (As one block per tile the kernel paid ~19 k cycles per block —
// launch, plan, the first loads' latency, drain — on 74 k cycles of matrix work: profiles/r04_bf16_conv_pmc.json.)
template <int WM, int WN, int WTN>
__global__ __launch_bounds__(64 * WM * WN, (WM * WN) / 4) void conv3x3_b16_kernel(
    const b16_t* __restrict__ x, const b16_t* __restrict__ wp, const float* __restrict__ bias, b16_t* __restrict__ y,
    int H, int W, int Cin, int Cout, int tiles_x, int tiles_y, int total_tiles, int relu, int pool) {
    using C = B16Cfg<WM, WN, WTN>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* lds_a = smem;                   // [2][A_BYTES]
    unsigned char* lds_b = smem + 2 * C::A_BYTES;  // [3][BS_BYTES]

    const int tid = threadIdx.x;
    const int n_blk = Cout / C::TN;
    // XCD-aware order (blocks are dealt round-robin over the 8 XCDs): the blocks of one XCD take a contiguous range of tile ids
    // in every round, so the channel blocks of a patch and neighbouring patches meet in one L2.  Speed only.
    int first = blockIdx.x;
    if ((gridDim.x & 7) == 0) first = (first & 7) * (gridDim.x >> 3) + (first >> 3);
    const int nchunks = Cin / kB16KC;
    const int nstages = 3 * nchunks;
    const size_t img_elems = (size_t)H * W * Cin;

    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    // waves w and w + 4 share a SIMD (a workgroup's waves are dealt to the SIMDs cyclically): the upper half issues its staging
    // loads one MFMA group later than the lower half, so that a SIMD's two waves are never both busy issuing them
    const int wave_hi = __builtin_amdgcn_readfirstlane(wave >= (WM * WN) / 2 ? 1 : 0);
    // halo staging plan of a tile: unit u = tid + r * THREADS of the LDS image <- 16-byte unit ((u & 3) ^ swizzle) of halo pixel u >> 2
    // (reflection resolved here); byte offsets within the image (< 2^32: entry check)
    int a_q[C::A_ROUNDS];      // per round: hy * 32 + hx and the source unit, packed (tile independent)
#pragma unroll
    for (int r = 0; r < C::A_ROUNDS; ++r) {
        const int u = tid + r * C::THREADS;
        const int q = u < C::A_UNITS ? u >> 2 : 0;            // padding lanes: a valid dummy source
        const int hy = q / kB16HW, hx = q % kB16HW;
        a_q[r] = (hy << 16) | (hx << 8) | ((u & 3) ^ b16_swz_a(hx));
    }
#define B16_PLAN(dst_, tile_)                                                                              \
    _Pragma("unroll") for (int r = 0; r < C::A_ROUNDS; ++r) {                                              \
        const int ry = reflect1((tile_).y0 - 1 + (a_q[r] >> 16), H);                                       \
        const int rx = reflect1((tile_).x0 - 1 + ((a_q[r] >> 8) & 255), W);                                \
        dst_[r] = (unsigned)((ry * W + rx) * Cin + 8 * (a_q[r] & 255)) * 2u;                               \
    }
    // weight tiles of one stage: unit u = tap t, output channel co, 16-byte unit; the taps' tiles are Cout * 64 bytes apart in the
    // packed weights [chunk][tap][Cout][32] and contiguous in LDS
    int b_src[C::BS_ROUNDS];
#pragma unroll
    for (int r = 0; r < C::BS_ROUNDS; ++r) {
        const int u = tid + r * C::THREADS < C::BS_UNITS ? tid + r * C::THREADS : 0;     // padding lanes: a valid dummy source
        const int t = u / C::BT_UNITS, v = u % C::BT_UNITS;
        const int co = v >> 2, c = (v & 3) ^ b16_swz_b(co);
        b_src[r] = t * Cout * (kB16KC * 2) + (co * kB16KC + 8 * c) * 2;
    }
    const int wave_base = __builtin_amdgcn_readfirstlane((tid & ~63) * 16);
    const unsigned lds_base = (unsigned)(unsigned long long)(const __attribute__((address_space(3))) void*)smem;
// LDS-DMA from inline asm (through the builtin hipcc drains vmcnt(0) before the next ds_read: conv_mfma.hip); wave-uniform
// 64-bit base in SGPRs + 32-bit per-lane byte offset; M0 (the LDS base of the DMA) is saved and restored
#define B16_GLDS(sbase_, voff_, ldst_)                                                                      \
    {                                                                                                      \
        unsigned m0_keep;                                                                                  \
        const unsigned lds_u = __builtin_amdgcn_readfirstlane(lds_base + (unsigned)(ldst_));               \
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\t" \
                     "s_mov_b32 m0, %0"                                                                     \
                     : "=&s"(m0_keep)                                                                      \
                     : "v"(voff_), "s"(sbase_), "s"(lds_u)                                                 \
                     : "memory");                                                                          \
    }
#define B16_LOAD_A1(r_, src_, xchunk_, buf_) B16_GLDS((xchunk_), src_[r_], (buf_) * C::A_BYTES + (r_) * C::THREADS * 16 + wave_base)
#define B16_LOAD_B1(r_, wsrc_, buf_) B16_GLDS((wsrc_), b_src[r_], 2 * C::A_BYTES + (buf_) * C::BS_BYTES + (r_) * C::THREADS * 16 + wave_base)
#define B16_LOAD_A(src_, xchunk_, buf_) _Pragma("unroll") for (int r = 0; r < C::A_ROUNDS; ++r) { B16_LOAD_A1(r, src_, xchunk_, buf_) }
#define B16_LOAD_B(wsrc_, buf_) _Pragma("unroll") for (int r = 0; r < C::BS_ROUNDS; ++r) { B16_LOAD_B1(r, wsrc_, buf_) }
#define B16_WAIT_N(n_) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n_) : "memory")

    // ---- per-lane operand coordinates -----------------------------------------------------------------------
    // tile pixel p = lane & 15 -> (x, y) = ((p & 1) | ((p >> 2) << 1), (p >> 1) & 1): 2 rows x 8 columns, and a 2 x 2 pooling
    // window is the 4 lanes 4g .. 4g+3.  k-unit u = lane >> 4.  Pixel tile mt: columns (mt & 1) * 8, rows (mt >> 1) * 2.
    const int p = lane & 15, ku = lane >> 4;
    const int lpx = (p & 1) | ((p >> 2) << 1);
    const int lpy = wm * 4 + ((p >> 1) & 1);
    const int b_off0 = (wn * (16 * WTN) + p) * 64 + ((ku ^ b16_swz_b(p)) << 4);    // + nt * 1024 (16 rows further: same swizzle)
    // halo reads: pixel (lpx + 8*(mt&1) + dx, lpy + 2*(mt>>1) + dy); the swizzle bit follows bit 1 of the column, which the
    // tile offset 8 leaves alone: three per-lane bases (one per dx), everything else is an immediate offset
    int a_lane[3];
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) a_lane[dx] = (lpy * kB16HW + lpx) * 64 + ((ku ^ b16_swz_a(lpx + dx)) << 4);

    if (first >= total_tiles) return;
    const size_t w_stage = (size_t)3 * Cout * kB16KC;
    B16Tile cur = b16_decode(first, n_blk, tiles_x, tiles_y, C::PR, C::TN);
    unsigned a_src[C::A_ROUNDS];
    B16_PLAN(a_src, cur)
    {
        const b16_t* w0 = wp + (size_t)cur.n0 * kB16KC;
        B16_LOAD_A(a_src, x + cur.img * img_elems, 0)
        B16_LOAD_B(w0, 0)
        B16_LOAD_B(w0 + w_stage, 1)           // Cin >= 32: at least three stages exist
    }
    B16_WAIT_N(0);
    __syncthreads();

    int apar = 0;
    for (int t = first; t < total_tiles; t += gridDim.x) {
        const bool more_tiles = t + (int)gridDim.x < total_tiles;
        const B16Tile nxt = b16_decode(more_tiles ? t + (int)gridDim.x : t, n_blk, tiles_x, tiles_y, C::PR, C::TN);
        const b16_t* __restrict__ ximg = x + cur.img * img_elems;
        const b16_t* __restrict__ xnxt = x + nxt.img * img_elems;
        const b16_t* __restrict__ wcur = wp + (size_t)cur.n0 * kB16KC;
        const b16_t* __restrict__ wnxt = wp + (size_t)nxt.n0 * kB16KC;

        f32x4 acc[4][WTN];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int nt = 0; nt < WTN; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};

        // ---- stages: chunk-major, three stages (kernel rows) per chunk, fully unrolled inside a stage.  Weight tiles are triple
        // buffered: stage s issues the loads of stage s + 2 (of the next tile past the end) at its start and waits for them at
        // its end (they have the whole stage to land), so after the barrier that ends stage s the data of stages s + 1 and s + 2
        // are visible.  The halo tile of the next chunk (of the next tile's first chunk) is requested at the start of a chunk's
        // second stage.  Inside a stage the fragment reads run two MFMA groups ahead of their use (weight fragments in a ring of
        // three, the next tap's pixel fragments in the other half of a double buffer).
        int stage = 0;
        for (int chunk = 0; chunk < nchunks; ++chunk) {
            const bool more_chunks = chunk + 1 < nchunks;
#pragma unroll
            for (int dy = 0; dy < 3; ++dy, ++stage) {
                const unsigned char* abuf = lds_a + apar * C::A_BYTES;
                const unsigned char* bbuf = lds_b + dy * C::BS_BYTES;          // stage % 3 == dy
                b16x8 afr[2][4], bfr[3];
#if HOLO_B16_ABLATE & 4      // timing only: no fragment reads (every MFMA takes the same two registers)
#define B16_READ_A(t_, mt_) frag_const
#define B16_READ_B(t_, nt_) frag_const
                b16x8 frag_const = *reinterpret_cast<const b16x8*>(abuf + a_lane[0]);
                asm volatile("" : "+v"(frag_const));
#else
#define B16_READ_A(t_, mt_) *reinterpret_cast<const b16x8*>(abuf + a_lane[t_] + ((((mt_) >> 1) * 2 + dy) * kB16HW + ((mt_) & 1) * 8 + (t_)) * 64)
#define B16_READ_B(t_, nt_) *reinterpret_cast<const b16x8*>(bbuf + (t_) * C::BT_BYTES + b_off0 + (nt_) * 1024)
#endif
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) afr[0][mt] = B16_READ_A(0, mt);
                bfr[0] = B16_READ_B(0, 0);
                bfr[1] = B16_READ_B(0, 1);
                // this stage's staging loads: the weight tiles of stage + 2 (of the next tile past the end), then — in a chunk's
                // FIRST stage — the halo tile of the next chunk (of the next tile's first chunk).  They are issued one at a time
                // between the MFMA groups below.  The weight loads are waited for at the end of this stage; the halo loads, which
                // come from HBM, stay in flight for another stage (counted wait: they are younger than the weight loads).
                const b16_t* wdma = stage + 2 < nstages ? wcur + (size_t)(stage + 2) * w_stage
                                                        : wnxt + (size_t)(stage + 2 - nstages) * w_stage;
                const bool dma_b = !(HOLO_B16_ABLATE & 1) && (stage + 2 < nstages || more_tiles);
                const bool dma_a = !(HOLO_B16_ABLATE & 1) && dy == 0 && (more_chunks || more_tiles);
                const b16_t* xdma = more_chunks ? ximg + (chunk + 1) * kB16KC : xnxt;
                unsigned a_use[C::A_ROUNDS];
                if (dy == 0) {
                    if (more_chunks) {
#pragma unroll
                        for (int r = 0; r < C::A_ROUNDS; ++r) a_use[r] = a_src[r];
                    } else {   // the next tile's plan, made where it is needed (not kept across the tile: registers)
                        B16_PLAN(a_use, nxt)
                    }
                }
#pragma unroll
                for (int g = 0; g < 3 * WTN; ++g) {
                    const int tt = g / WTN, nt = g % WTN;
                    // staging load k goes behind group 1 + k * DMA_STEP (+ 1 in the upper waves)
#pragma unroll
                    for (int k = 0; k < C::BS_ROUNDS + (dy == 0 ? C::A_ROUNDS : 0); ++k) {
                        if (g == 1 + k * C::DMA_STEP || g == 2 + k * C::DMA_STEP) {
                            if (wave_hi == (g - 1 - k * C::DMA_STEP)) {
                                if (k < C::BS_ROUNDS) {
                                    if (dma_b) { B16_LOAD_B1(k, wdma, (dy + 2) % 3) }
                                } else {
                                    if (dma_a) { B16_LOAD_A1(k - C::BS_ROUNDS, a_use, xdma, apar ^ 1) }
                                }
                            }
                        }
                    }
                    if (g + 2 < 3 * WTN) bfr[(g + 2) % 3] = B16_READ_B((g + 2) / WTN, (g + 2) % WTN);
                    if (tt < 2) {
#pragma unroll
                        for (int mt = 0; mt < 4; ++mt)
                            if (nt == (mt * WTN) / 4) afr[(tt + 1) & 1][mt] = B16_READ_A(tt + 1, mt);
                    }
                    // D[m = channel][n = pixel] += W[channel][k] * X[pixel][k]
#pragma unroll
                    for (int mt = 0; mt < 4; ++mt)
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[g % 3], afr[tt & 1][mt], acc[mt][nt], 0, 0, 0);
                }
#undef B16_READ_A
#undef B16_READ_B
#if !(HOLO_B16_ABLATE & 2)
                if (stage + 1 < nstages || more_tiles) {
                    // this wave's weight loads of the stage have landed; a chunk's first stage leaves its halo loads in flight
                    if (dy == 0 && dma_a) { B16_WAIT_N(C::A_ROUNDS); }
                    else { B16_WAIT_N(0); }
                    __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): this stage's fragment reads are done
                    __builtin_amdgcn_s_barrier();         // raw barrier
                    asm volatile("" ::: "memory");
                }
#endif
            }
            apar ^= 1;
        }

        // ---- epilogue: bias, ReLU, optional 2x2 max-pool, bf16, 16-byte NHWC stores -------------------------------------------
        // tile (mt, nt): lane (p, h = lane >> 4) holds channels nt*16 + 4h .. 4h+3 of its pixel = 8 bytes of bf16.  For a pair of
        // channel tiles (nt, nt+1), v_permlane16_swap (lanes 16-31 of the first operand <-> lanes 0-15 of the second, same for the
        // upper half) leaves rows h = 0 / 2 with channels 0-7 / 8-15 of tile nt+1 and rows 1 / 3 with those of tile nt.
        const int Ho = (H + 1) >> 1, Wo = (W + 1) >> 1;
        const int cwave = cur.n0 + wn * (16 * WTN);
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            const int X = cur.x0 + lpx + (mt & 1) * 8, Y = cur.y0 + lpy + (mt >> 1) * 2;
            const bool inside = X < W && Y < H;
            bool writer = inside;
            size_t pix;
            b16_t* __restrict__ yimg;
            if (!pool) {
                yimg = y + (size_t)cur.img * H * W * Cout;
                pix = (size_t)Y * W + X;
            } else {
                yimg = y + (size_t)cur.img * Ho * Wo * Cout;
                writer = inside && (p & 3) == 0;          // the window's top-left pixel stores
                pix = (size_t)(Y >> 1) * Wo + (X >> 1);
            }
#pragma unroll
            for (int nt = 0; nt < WTN; nt += 2) {
                unsigned pk[2][2];
#pragma unroll
                for (int h2 = 0; h2 < 2; ++h2) {
                    const float4 b4 = *reinterpret_cast<const float4*>(bias + cwave + (nt + h2) * 16 + 4 * ku);
                    const float bq[4] = {b4.x, b4.y, b4.z, b4.w};
                    float v[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        float s = acc[mt][nt + h2][i] + bq[i];
                        s = relu ? fmaxf(s, 0.0f) : s;
                        if (pool) {
                            // MaxPool2d(2,2,ceil_mode=True): window = lanes 4g..4g+3; members beyond the image edge count as 0
                            // (pooling follows a ReLU)
                            s = inside ? s : 0.0f;
                            s = fmaxf(s, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, s), 0xB1, 0xF, 0xF, true)));
                            s = fmaxf(s, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, s), 0x4E, 0xF, 0xF, true)));
                        }
                        v[i] = s;
                    }
                    typedef __bf16 b16x2 __attribute__((ext_vector_type(2)));
                    const b16x2 lo = {(b16_t)v[0], (b16_t)v[1]}, hi = {(b16_t)v[2], (b16_t)v[3]};
                    pk[h2][0] = __builtin_bit_cast(unsigned, lo);
                    pk[h2][1] = __builtin_bit_cast(unsigned, hi);
                }
                // first operand = tile nt+1's quad, second = tile nt's
                const auto s0 = __builtin_amdgcn_permlane16_swap(pk[1][0], pk[0][0], false, false);
                const auto s1 = __builtin_amdgcn_permlane16_swap(pk[1][1], pk[0][1], false, false);
                const uint4 run = make_uint4(s0[0], s1[0], s0[1], s1[1]);
                const int ch = cwave + nt * 16 + ((ku & 1) ? 0 : 16) + (ku >> 1) * 8;
                if (writer && (!(HOLO_B16_ABLATE & 8) || run.x == 0x12345678u)) *reinterpret_cast<uint4*>(yimg + pix * Cout + ch) = run;
            }
        }
        cur = nxt;
        B16_PLAN(a_src, cur)
    }
}

#undef B16_GLDS
#undef B16_LOAD_A
#undef B16_LOAD_B
#undef B16_LOAD_A1
#undef B16_LOAD_B1
#undef B16_WAIT_N
#undef B16_PLAN

// ReLU, optional 2x2 max-pool, bf16 and the 16-byte NHWC stores of one wave's 64 pixels x (16 * WTN) channels (the accumulators
// already hold the bias: they start from it): tile (mt, nt): lane (p, h = lane >> 4) holds channels nt*16 + 4h .. 4h+3 of its pixel;
// v_permlane16_swap joins the quads of a pair of channel tiles into 16-byte runs.
template <int WTN>
__device__ __forceinline__ void b16_store_tile(f32x4 (&acc)[4][WTN], const B16Tile& cur,
                                               b16_t* __restrict__ y, int H, int W, int Cout, int relu, int pool, int p, int ku,
                                               int lpx, int lpy, int cwave) {
    const int Ho = (H + 1) >> 1, Wo = (W + 1) >> 1;
    const float fl = relu ? 0.0f : -__builtin_inff();
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
        const int X = cur.x0 + lpx + (mt & 1) * 8, Y = cur.y0 + lpy + (mt >> 1) * 2;
        const bool inside = X < W && Y < H;
        bool writer = inside;
        size_t pix;
        b16_t* __restrict__ yimg;
        if (!pool) {
            yimg = y + (size_t)cur.img * H * W * Cout;
            pix = (size_t)Y * W + X;
        } else {
            yimg = y + (size_t)cur.img * Ho * Wo * Cout;
            writer = inside && (p & 3) == 0;          // the window's top-left pixel stores
            pix = (size_t)(Y >> 1) * Wo + (X >> 1);
        }
#pragma unroll
        for (int nt = 0; nt < WTN; nt += 2) {
            unsigned pk[2][2];
#pragma unroll
            for (int h2 = 0; h2 < 2; ++h2) {
                // the bias is already in the accumulators (they start from it); ReLU as a max against 0 | -inf (no branch)
                float v[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = fmaxf(acc[mt][nt + h2][i], fl);
                if (pool) {   // MaxPool2d(2,2,ceil_mode=True): window = lanes 4g..4g+3; members beyond the image edge count as 0
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        float s = inside ? v[i] : 0.0f;
                        s = fmaxf(s, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, s), 0xB1, 0xF, 0xF, true)));
                        s = fmaxf(s, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, s), 0x4E, 0xF, 0xF, true)));
                        v[i] = s;
                    }
                }
                typedef __bf16 b16x2 __attribute__((ext_vector_type(2)));
                const b16x2 lo = {(b16_t)v[0], (b16_t)v[1]}, hi = {(b16_t)v[2], (b16_t)v[3]};
                pk[h2][0] = __builtin_bit_cast(unsigned, lo);
                pk[h2][1] = __builtin_bit_cast(unsigned, hi);
            }
            const auto s0 = __builtin_amdgcn_permlane16_swap(pk[1][0], pk[0][0], false, false);
            const auto s1 = __builtin_amdgcn_permlane16_swap(pk[1][1], pk[0][1], false, false);
            const uint4 run = make_uint4(s0[0], s1[0], s0[1], s1[1]);
            const int ch = cwave + nt * 16 + ((ku & 1) ? 0 : 16) + (ku >> 1) * 8;
            if (writer) *reinterpret_cast<uint4*>(yimg + pix * Cout + ch) = run;
        }
    }
}

// ================================================================================================================
// Weights-resident kernel for the 64 -> 64 layers (net.py:80 enc_2's first conv, net.py:66,69 the decoder's full-resolution pair):
// they are 20 % of the network's FLOP and its most memory-bound convolutions (128 B in + 128 B out per pixel for 73.7 kFLOP).
// All 2 x 9 weight tiles (72 KB) stay in LDS for the life of a persistent block, so the only staging left is the halo — one
// 32-channel chunk (39 KB) per 144 MFMAs of each wave — and the block meets at a barrier twice per 512-pixel tile instead of six
// times.  The two halo buffers hold chunk 0 and chunk 1: phase 0 of tile i (reads buffer 0) stages chunk 1 of tile i into buffer 1,
// which phase 1 of tile i - 1 has just left; phase 1 of tile i (reads buffer 1) stages chunk 0 of tile i + 1 into buffer 0.
// ================================================================================================================
struct B16W {
    static constexpr int THREADS = 512, WTN = 4;
    static constexpr int PR = 32, HR = PR + 2;
    static constexpr int A_UNITS = HR * kB16HW * 4;
    static constexpr int A_ROUNDS = (A_UNITS + THREADS - 1) / THREADS;     // 5
    static constexpr int A_BYTES = A_ROUNDS * THREADS * 16;                // padded to whole rounds (uniform load counts)
    static constexpr int W_TILE = 64 * 64;                                 // one (chunk, tap): 64 channels x 64 bytes
    static constexpr int W_BYTES = 18 * W_TILE;
    static constexpr int W_ROUNDS = W_BYTES / 16 / THREADS;                // 9
    static constexpr int LDS_BYTES = W_BYTES + 2 * A_BYTES;
    static_assert(LDS_BYTES <= 160 * 1024, "LDS");
};

__global__ __launch_bounds__(512, 2) void conv3x3_b16_wres_kernel(
    const b16_t* __restrict__ x, const b16_t* __restrict__ wp, const float* __restrict__ bias, b16_t* __restrict__ y,
    int H, int W, int tiles_x, int tiles_y, int total_tiles, int relu, int pool) {
    using C = B16W;
    constexpr int Cin = 64, Cout = 64, WTN = C::WTN;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* lds_w = smem;                   // [2 chunks][9 taps][64 co][64 B]
    unsigned char* lds_a = smem + C::W_BYTES;      // [2 chunks][A_BYTES]
    const int tid = threadIdx.x;
    int first = blockIdx.x;
    if ((gridDim.x & 7) == 0) first = (first & 7) * (gridDim.x >> 3) + (first >> 3);
    if (first >= total_tiles) return;
    const size_t img_elems = (size_t)H * W * Cin;
    const int lane = tid & 63, wave = tid >> 6;
    const int wave_hi = __builtin_amdgcn_readfirstlane(wave >= 4 ? 1 : 0);
    const int wave_base = __builtin_amdgcn_readfirstlane((tid & ~63) * 16);
    const unsigned lds_base = (unsigned)(unsigned long long)(const __attribute__((address_space(3))) void*)smem;
#define B16W_GLDS(sbase_, voff_, ldst_)                                                                     \
    {                                                                                                      \
        unsigned m0_keep;                                                                                  \
        const unsigned lds_u = __builtin_amdgcn_readfirstlane(lds_base + (unsigned)(ldst_));               \
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\t" \
                     "s_mov_b32 m0, %0"                                                                     \
                     : "=&s"(m0_keep)                                                                      \
                     : "v"(voff_), "s"(sbase_), "s"(lds_u)                                                 \
                     : "memory");                                                                          \
    }
#define B16W_PLAN(dst_, tile_)                                                                             \
    _Pragma("unroll") for (int r = 0; r < C::A_ROUNDS; ++r) {                                              \
        const int u = tid + r * C::THREADS;                                                                \
        const int q = u < C::A_UNITS ? u >> 2 : 0;            /* padding lanes: a valid dummy source */    \
        const int hy = q / kB16HW, hx = q % kB16HW;                                                        \
        const int ry = reflect1((tile_).y0 - 1 + hy, H);                                                   \
        const int rx = reflect1((tile_).x0 - 1 + hx, W);                                                   \
        dst_[r] = (unsigned)((ry * W + rx) * Cin + 8 * ((u & 3) ^ b16_swz_a(hx))) * 2u;                    \
    }
#define B16W_LOAD_A1(r_, xchunk_, buf_) B16W_GLDS((xchunk_), a_pl[r_], C::W_BYTES + (buf_) * C::A_BYTES + (r_) * C::THREADS * 16 + wave_base)

    // ---- the weights, once: packed [chunk][tap][co][32] = 18 tiles of 4 KB, contiguous; unit swizzle as in the staged kernel ----
#pragma unroll
    for (int r = 0; r < C::W_ROUNDS; ++r) {
        const int u = tid + r * C::THREADS;
        const int tile = u >> 8, v = u & 255;                 // 256 units per (chunk, tap) tile
        const int co = v >> 2, c = (v & 3) ^ b16_swz_b(co);
        const int src = tile * C::W_TILE + (co * kB16KC + 8 * c) * 2;
        B16W_GLDS(wp, src, r * C::THREADS * 16 + wave_base)
    }
    const int p = lane & 15, ku = lane >> 4;
    const int lpx = (p & 1) | ((p >> 2) << 1);
    const int lpy = wave * 4 + ((p >> 1) & 1);
    const int b_off0 = p * 64 + ((ku ^ b16_swz_b(p)) << 4);
    int a_lane[3];
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) a_lane[dx] = (lpy * kB16HW + lpx) * 64 + ((ku ^ b16_swz_a(lpx + dx)) << 4);

    f32x4 biasq[WTN];
#pragma unroll
    for (int nt = 0; nt < WTN; ++nt) {
        const float4 b4 = *reinterpret_cast<const float4*>(bias + nt * 16 + 4 * ku);
        biasq[nt] = f32x4{b4.x, b4.y, b4.z, b4.w};
    }
    B16Tile cur = b16_decode(first, 1, tiles_x, tiles_y, C::PR, 64);
    unsigned a_pl[C::A_ROUNDS];            // halo staging plan of the tile whose chunks are staged next
    B16W_PLAN(a_pl, cur)
    {
        const b16_t* x0p = x + cur.img * img_elems;
#pragma unroll
        for (int r = 0; r < C::A_ROUNDS; ++r) { B16W_LOAD_A1(r, x0p, 0) }
#pragma unroll
        for (int r = 0; r < C::A_ROUNDS; ++r) { B16W_LOAD_A1(r, x0p + kB16KC, 1) }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    for (int t = first; t < total_tiles; t += gridDim.x) {
        const bool more_tiles = t + (int)gridDim.x < total_tiles;
        const B16Tile nxt = b16_decode(more_tiles ? t + (int)gridDim.x : t, 1, tiles_x, tiles_y, C::PR, 64);
        const b16_t* __restrict__ xcur = x + cur.img * img_elems;
        const b16_t* __restrict__ xnxt = x + nxt.img * img_elems;
        f32x4 acc[4][WTN];      // start from the bias (channels nt*16 + 4*ku .. +3 of this lane): the epilogue has nothing to add
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int nt = 0; nt < WTN; ++nt) acc[mt][nt] = biasq[nt];

#pragma unroll
        for (int chunk = 0; chunk < 2; ++chunk) {
            // chunk phase: 9 taps x WTN channel tiles = 36 groups of 4 MFMAs; weight fragments two groups ahead, pixel fragments one
            // tap ahead; the phase's 5 halo loads go one at a time behind groups 2, 6, 10, ... (+ 1 in the upper waves)
            const unsigned char* abuf = lds_a + chunk * C::A_BYTES;
            const unsigned char* wbuf = lds_w + chunk * 9 * C::W_TILE;
            if (chunk == 1 && more_tiles) { B16W_PLAN(a_pl, nxt) }     // phase 0 staged with the current tile's plan; from here on the next tile's
            const bool dma = !(HOLO_B16_ABLATE & 1) && (chunk == 0 ? t != first : more_tiles);
            const b16_t* xdma = chunk == 0 ? xcur + kB16KC : xnxt;      // phase 0: this tile's chunk 1 -> buffer 1; phase 1: next tile's chunk 0 -> buffer 0
            b16x8 afr[2][4], bfr[3];
#define B16W_READ_A(tap_, mt_) *reinterpret_cast<const b16x8*>(abuf + a_lane[(tap_) % 3] + ((((mt_) >> 1) * 2 + (tap_) / 3) * kB16HW + ((mt_) & 1) * 8 + (tap_) % 3) * 64)
#define B16W_READ_B(tap_, nt_) *reinterpret_cast<const b16x8*>(wbuf + (tap_) * C::W_TILE + b_off0 + (nt_) * 1024)
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) afr[0][mt] = B16W_READ_A(0, mt);
            bfr[0] = B16W_READ_B(0, 0);
            bfr[1] = B16W_READ_B(0, 1);
#pragma unroll
            for (int g = 0; g < 9 * WTN; ++g) {
                const int tap = g / WTN, nt = g % WTN;
#pragma unroll
                for (int k = 0; k < C::A_ROUNDS; ++k) {
                    if (g == 2 + 4 * k || g == 3 + 4 * k) {
                        if (wave_hi == (g - 2 - 4 * k)) {
                            if (dma) { B16W_LOAD_A1(k, xdma, chunk ^ 1) }
                        }
                    }
                }
                if (g + 2 < 9 * WTN) bfr[(g + 2) % 3] = B16W_READ_B((g + 2) / WTN, (g + 2) % WTN);
                if (tap < 8) {
#pragma unroll
                    for (int mt = 0; mt < 4; ++mt)
                        if (nt == mt) afr[(tap + 1) & 1][mt] = B16W_READ_A(tap + 1, mt);
                }
#pragma unroll
                for (int mt = 0; mt < 4; ++mt)
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[g % 3], afr[tap & 1][mt], acc[mt][nt], 0, 0, 0);
            }
#undef B16W_READ_A
#undef B16W_READ_B
            if (!(HOLO_B16_ABLATE & 2) && (chunk == 0 || more_tiles)) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // this phase's halo loads (and the previous tile's stores) are done
                __builtin_amdgcn_s_waitcnt(0xc07f);                   // lgkmcnt(0): this phase's fragment reads are done
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
            }
        }
        b16_store_tile<WTN>(acc, cur, y, H, W, Cout, relu, pool, p, ku, lpx, lpy, 0);
        cur = nxt;
    }
}
#undef B16W_GLDS
#undef B16W_PLAN
#undef B16W_LOAD_A1

static int launch_conv_b16_wres(const b16_t* x, const b16_t* wp, const float* bias, b16_t* y, int batch, int h, int w, int relu, int pool,
                                hipStream_t stream) {
    using C = B16W;
    const int tiles_x = ceil_div(w, 16), tiles_y = ceil_div(h, C::PR);
    const long tiles = (long)batch * tiles_x * tiles_y;
    HOLO_REQUIRE(tiles > 0 && tiles < 2147483647L, HOLO_ERR_UNSUPPORTED, "conv: %ld tiles", tiles);
    static DeviceOnce attr_once;
    attr_once.run([] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_b16_wres_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  C::LDS_BYTES);
    });
    const long cus = device_cus();
    const unsigned grid = (unsigned)(tiles < cus ? tiles : cus);
    hipLaunchKernelGGL(conv3x3_b16_wres_kernel, dim3(grid), dim3(C::THREADS), C::LDS_BYTES, stream, x, wp, bias, y, h, w, tiles_x, tiles_y,
                       (int)tiles, relu, pool);
    return check_launch("holo_conv3x3_bf16");
}

template <int WM, int WN, int WTN>
static int launch_conv_b16(const b16_t* x, const b16_t* wp, const float* bias, b16_t* y, int batch, int h, int w, int cin, int cout,
                           int relu, int pool, hipStream_t stream) {
    using C = B16Cfg<WM, WN, WTN>;
    const int tiles_x = ceil_div(w, 16), tiles_y = ceil_div(h, C::PR);
    const long tiles = (long)batch * tiles_x * tiles_y * (cout / C::TN);
    HOLO_REQUIRE(tiles > 0 && tiles < 2147483647L, HOLO_ERR_UNSUPPORTED, "conv: %ld tiles", tiles);
    static DeviceOnce attr_once;
    attr_once.run([] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_b16_kernel<WM, WN, WTN>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES);
    });
    const long cus = device_cus();      // one block per CU (LDS); every block reaches the end of its tile list: no block waits for another
    const unsigned grid = (unsigned)(tiles < cus ? tiles : cus);
    hipLaunchKernelGGL((conv3x3_b16_kernel<WM, WN, WTN>), dim3(grid), dim3(C::THREADS), C::LDS_BYTES, stream, x, wp, bias, y,
                       h, w, cin, cout, tiles_x, tiles_y, (int)tiles, relu, pool);
    return check_launch("holo_conv3x3_bf16");
}

// ================================================================================================================
// Last layer, bf16 input: ReflectionPad2d(1) -> Conv2d(64, 2, 3x3), no activation (net.py:72-73), NHWC bf16 in, NCHW fp32
// amplitude / phase planes out.  As a VALU kernel (conv_direct.hip: 1152 FMAs, 144 LDS reads and the bf16 -> fp32 conversion of
// every halo element per pixel) it ran at 2.1 TB/s of its input read.  Here it is 18 v_mfma_f32_16x16x32_bf16 per 16 pixels:
//   D[pixel][n] += X[pixel][k] * W'[k][n],  n = 0..5 of 16 columns: the two output channels' fp32 weights split EXACTLY into
//   three bf16 pieces (hi, mid, lo: 8 + 8 + 8 mantissa bits), so the products are the fp32-weight products and the result keeps
//   the fp32-weight accuracy of the VALU kernel; columns 6..15 are zero (the matrix pipe has 16 x the VALU's rate to burn).
// The weight fragments live in REGISTERS (18 x 4 VGPRs, built once per persistent block from the OIHW tensor); the bf16 halo
// (18 x 18 pixels x 128 B) goes to LDS by LDS-DMA with no conversion; two blocks per CU overlap each other's load phases.
// LDS image: pixel rows of 128 B, unit' = unit ^ (hx & 6): conflict-free ds_read_b128 for 16 consecutive pixels (scratch/swz_search.py).
// ================================================================================================================
struct B16L {
    static constexpr int THREADS = 256, T = 16, HW = 18;
    static constexpr int A_UNITS = HW * HW * 8;
    static constexpr int A_ROUNDS = (A_UNITS + THREADS - 1) / THREADS;     // 11
    static constexpr int A_BYTES = A_ROUNDS * THREADS * 16;
};

__global__ __launch_bounds__(256, 2) void conv_last_b16_kernel(const b16_t* __restrict__ x, const float* __restrict__ w,
                                                               const float* __restrict__ bias, float* __restrict__ y, int H, int W,
                                                               int tiles_x, int tiles_y, int total_tiles) {
    using C = B16L;
    __shared__ __attribute__((aligned(16))) unsigned char halo[C::A_BYTES];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = lane & 15, ku = lane >> 4;
    // ---- weight fragments: lane (n, ku) holds W'[k = 32 kc + 8 ku + j][n] of every tap; n = 2 * piece + channel ----
    b16x8 wf[9][2];
    {
        const int co = n & 1, piece = n >> 1;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap)
#pragma unroll
            for (int kc = 0; kc < 2; ++kc)
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    float v = 0.0f;
                    if (n < 6) {
                        const float f = w[((size_t)co * 64 + 32 * kc + 8 * ku + j) * 9 + tap];     // OIHW (2, 64, 3, 3)
                        const float hi = (float)(b16_t)f;
                        const float mid = (float)(b16_t)(f - hi);
                        const float lo = (float)(b16_t)((f - hi) - mid);
                        v = piece == 0 ? hi : (piece == 1 ? mid : lo);
                    }
                    wf[tap][kc][j] = (b16_t)v;
                }
    }
    const float bsel = bias[n & 1];
    const int wave_base = __builtin_amdgcn_readfirstlane((tid & ~63) * 16);
    const unsigned lds_base = (unsigned)(unsigned long long)(const __attribute__((address_space(3))) void*)halo;
    // per-lane read base: pixel (x = n, row = 4 * wave) of the halo, unit ku; + tap / tile-row offsets as immediates; the swizzle
    // follows bits 1..2 of the halo column n + dx
    int a_lane[3];
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) a_lane[dx] = ((4 * wave) * C::HW + n + dx) * 128 + ((ku ^ ((n + dx) & 6)) << 4);

    for (int t = blockIdx.x; t < total_tiles; t += gridDim.x) {
        int r = t;
        const int tx = r % tiles_x;
        r /= tiles_x;
        const int ty = r % tiles_y;
        const int img = r / tiles_y;
        const int x0 = tx * C::T, y0 = ty * C::T;
        const b16_t* __restrict__ xi = x + (size_t)img * H * W * 64;
        // halo -> LDS (every lane issues A_ROUNDS loads; lanes past the image fetch a valid dummy unit into the padding)
#pragma unroll
        for (int rr = 0; rr < C::A_ROUNDS; ++rr) {
            const int u = tid + rr * C::THREADS;
            const int q = u < C::A_UNITS ? u >> 3 : 0;
            const int hy = q / C::HW, hx = q % C::HW;
            const int ry = reflect1(y0 - 1 + hy, H), rx = reflect1(x0 - 1 + hx, W);
            const unsigned src = (unsigned)((ry * W + rx) * 64 + 8 * ((u & 7) ^ (hx & 6))) * 2u;
            unsigned m0_keep;
            const unsigned lds_u = __builtin_amdgcn_readfirstlane(lds_base + (unsigned)(rr * C::THREADS * 16 + wave_base));
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                         : "=&s"(m0_keep)
                         : "v"(src), "s"(xi), "s"(lds_u)
                         : "memory");
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        f32x4 acc[4];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) acc[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
#pragma unroll
            for (int kc = 0; kc < 2; ++kc) {
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) {
                    const b16x8 a = *reinterpret_cast<const b16x8*>(halo + (a_lane[tap % 3] ^ (kc << 6)) + (mt + tap / 3) * C::HW * 128);
                    acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, wf[tap][kc], acc[mt], 0, 0, 0);
                }
            }
        }
        __syncthreads();     // the halo is consumed: the next tile's loads may overwrite it
        // D[pixel = 4 ku + i][n]: channel c = n & 1 is the sum of columns c, c + 2, c + 4 (hi, mid, lo pieces)
        float* __restrict__ yi = y + (size_t)img * 2 * H * W;
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            float v[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float a0 = acc[mt][i];
                const float a2 = __shfl_down(a0, 2, 16), a4 = __shfl_down(a0, 4, 16);
                v[i] = (a0 + a2) + a4 + bsel;
            }
            const int Y = y0 + 4 * wave + mt, X = x0 + 4 * ku;
            if (n < 2 && Y < H) {
                float* dst = yi + (size_t)n * H * W + (size_t)Y * W + X;
                if (X + 3 < W && (W & 3) == 0) *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]);
                else {
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        if (X + i < W) dst[i] = v[i];
                }
            }
        }
    }
}

int conv_last_b16_try(const void* x, const float* w, const float* bias, float* y, int batch, int h, int w_px, hipStream_t stream) {
    const int tiles_x = ceil_div(w_px, B16L::T), tiles_y = ceil_div(h, B16L::T);
    const long tiles = (long)batch * tiles_x * tiles_y;
    // chosen per image and for the open batch bucket only (common.h kOpenBatchBucket): images of >= 64 tiles (128 x 128) in batches
    // >= 13; anything smaller keeps the one-tile-per-block VALU kernel
    if (batch < kOpenBatchBucket || (long)tiles_x * tiles_y < 64 || tiles >= 2147483647L || (long)h * w_px * 64 >= (1L << 30)) return -1;
    const long slots = (long)device_cus() * 2;
    const unsigned grid = (unsigned)(tiles < slots ? tiles : slots);
    hipLaunchKernelGGL(conv_last_b16_kernel, dim3(grid), dim3(256), 0, stream, static_cast<const b16_t*>(x), w, bias, y, h, w_px, tiles_x,
                       tiles_y, (int)tiles);
    return check_launch("holo_conv_last_bf16");
}

// ================================================================================================================
// First layer, bf16 output: vgg[0:4]  Conv2d(1,3,1x1) -> ReflectionPad2d(1) -> Conv2d(3,64,3x3) -> ReLU (net.py:77-80), one fp32
// plane in, NHWC bf16 out.  On the fp32 MFMA (conv_direct.hip) the layer's 14 k-steps cost 3.6 k matrix-pipe cycles per 256
// pixels, as much as its 32 KB output write; the two did not overlap (0.84 ms at 64 x 512^2 for 0.43 ms of writes).  Here K = 27
// (padded to 32) is ONE step of v_mfma_f32_16x16x32_bf16 with both operands split into two bf16 pieces (a = hi + lo: 16 mantissa
// bits): hi*hi + hi*lo + lo*hi, 48 MFMAs of 16 cycles per wave and 256 pixels.  The products carry 2^-16, far below the bf16
// rounding of the output (2^-9): the result is the fp32 kernel's up to rare one-ulp differences at rounding boundaries.
//   D[co][pixel] = sum_k W1[co][k] * A[pixel][k],   A[pixel][9c + tap] = w0[c] * in(pixel + tap) + b0[c]   (vgg[0] on the fly)
// Same wave geometry and epilogue as the 3x3 kernels (4 rows x 16 columns per wave, b16_store_tile).
// ================================================================================================================
__global__ __launch_bounds__(256, 3) void conv_first_b16_kernel(const float* __restrict__ x, const float* __restrict__ w0,
                                                                const float* __restrict__ b0, const float* __restrict__ w1,
                                                                const float* __restrict__ b1, b16_t* __restrict__ y, int H, int W,
                                                                int tiles_x, int tiles_y, int total_tiles, int sqrt_in) {
    constexpr int HW = 18;
    __shared__ float halo[HW * HW];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int p = lane & 15, ku = lane >> 4;
    const int lpx = (p & 1) | ((p >> 2) << 1);
    const int lpy = wave * 4 + ((p >> 1) & 1);
    // weights: lane (co = 16 nt + p, ku) holds W1[co][k = 8 ku + j] split into hi / lo
    b16x8 whi[4], wlo[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = 8 * ku + j;
            const float f = k < 27 ? w1[(16 * nt + p) * 27 + k] : 0.0f;          // OIHW (64,3,3,3) = [co][27]
            const b16_t hi = (b16_t)f;
            whi[nt][j] = hi;
            wlo[nt][j] = (b16_t)(f - (float)hi);
        }
    // this lane's 8 columns of A: halo offset of the tap and the 1x1 pre-conv's scale / offset of the channel
    int koff[8];
    float kw[8], kb[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int k = 8 * ku + j;
        const int kk = k < 27 ? k : 0;
        const int c = kk / 9, tap = kk % 9;
        koff[j] = (tap / 3) * HW + tap % 3;
        kw[j] = k < 27 ? w0[c] : 0.0f;          // the padded columns multiply zero weights: make them exact zeros all the same
        kb[j] = k < 27 ? b0[c] : 0.0f;
    }
    f32x4 biasq[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
        const float4 b4 = *reinterpret_cast<const float4*>(b1 + nt * 16 + 4 * ku);
        biasq[nt] = f32x4{b4.x, b4.y, b4.z, b4.w};
    }
    for (int t = blockIdx.x; t < total_tiles; t += gridDim.x) {
        B16Tile cur = b16_decode(t, 1, tiles_x, tiles_y, 16, 64);
        const float* __restrict__ xi = x + (size_t)cur.img * H * W;
        __syncthreads();      // the previous tile's halo is consumed
        for (int i = tid; i < HW * HW; i += 256) {
            const int hy = i / HW, hx = i % HW;
            const float v = xi[reflect1(cur.y0 - 1 + hy, H) * W + reflect1(cur.x0 - 1 + hx, W)];
            halo[i] = sqrt_in ? sqrtf(v) : v;     // intensity in (holo_conv_first_sqrt_*); sqrtf: correctly rounded
        }
        __syncthreads();
        f32x4 acc[4][4];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            const int hbase = (lpy + (mt >> 1) * 2) * HW + lpx + (mt & 1) * 8;
            b16x8 ahi, alo;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float a = halo[hbase + koff[j]] * kw[j] + kb[j];
                const b16_t hi = (b16_t)a;
                ahi[j] = hi;
                alo[j] = (b16_t)(a - (float)hi);
            }
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                f32x4 c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wlo[nt], ahi, biasq[nt], 0, 0, 0);   // small terms first
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(whi[nt], alo, c, 0, 0, 0);
                acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(whi[nt], ahi, c, 0, 0, 0);
            }
        }
        b16_store_tile<4>(acc, cur, y, H, W, 64, 1, 0, p, ku, lpx, lpy, 0);
    }
}

int conv_first_b16_try(const float* x, const float* w0, const float* b0, const float* w1, const float* b1, void* y, int batch, int h,
                       int w_px, int sqrt_in, hipStream_t stream) {
    const int tiles_x = ceil_div(w_px, 16), tiles_y = ceil_div(h, 16);
    const long tiles = (long)batch * tiles_x * tiles_y;
    // per image, open batch bucket only (common.h kOpenBatchBucket): images of >= 64 tiles (128 x 128)
    if (batch < kOpenBatchBucket || (long)tiles_x * tiles_y < 64 || tiles >= 2147483647L) return -1;
    const long slots = (long)device_cus() * 3;
    const unsigned grid = (unsigned)(tiles < slots ? tiles : slots);
    hipLaunchKernelGGL(conv_first_b16_kernel, dim3(grid), dim3(256), 0, stream, x, w0, b0, w1, b1, static_cast<b16_t*>(y), h, w_px, tiles_x,
                       tiles_y, (int)tiles, sqrt_in);
    return check_launch("holo_conv_first_bf16");
}

// ================================================================================================================
// ConvTranspose2d(64, 64, k=2, s=2) + ReLU in bf16 (net.py:64, the decoder's last up-sampling: half of the transposed
// convolutions' time): a one-tap GEMM, out[2Y+dy][2X+dx][co] = sum_ci in[Y][X][ci] * W[ci][co][dy][dx], that moves 128 B in and
// 512 B out per input pixel for 32 kFLOP — pure streaming.  The generic kernel (conv_mfma.hip) gave a block two K stages between
// its prologue and epilogue and reached 3.2 TB/s.  Here the packed weights [2 chunks][4 * 64 columns][32] (32 KB) are staged once
// per persistent block and a tile is 256 input pixels (32 KB by LDS-DMA, no halo): each wave keeps its 64 pixels' fragments in
// registers and runs the four quadrants (dy, dx) as 4 x 32 MFMAs, storing 16-byte runs of the (2H x 2W) NHWC output after each.
// ================================================================================================================
struct B16T {
    static constexpr int THREADS = 256;
    static constexpr int W_BYTES = 2 * 256 * 64;           // [chunk][n = q * 64 + co][32 ch]
    static constexpr int A_BYTES = 2 * 256 * 64;           // [chunk][256 pixels][32 ch]
    static constexpr int LDS_BYTES = W_BYTES + A_BYTES;
};

__global__ __launch_bounds__(256, 2) void convt2x2_b16_c64_kernel(const b16_t* __restrict__ x, const b16_t* __restrict__ wp,
                                                                   const float* __restrict__ bias, b16_t* __restrict__ y, int H, int W,
                                                                   int tiles_x, int tiles_y, int total_tiles, int relu) {
    using C = B16T;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* lds_w = smem;
    unsigned char* lds_a = smem + C::W_BYTES;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int p = lane & 15, ku = lane >> 4;
    const int lpx = (p & 1) | ((p >> 2) << 1);
    const int lpy = wave * 4 + ((p >> 1) & 1);
    const int wave_base = __builtin_amdgcn_readfirstlane((tid & ~63) * 16);
    const unsigned lds_base = (unsigned)(unsigned long long)(const __attribute__((address_space(3))) void*)smem;
#define B16T_GLDS(sbase_, voff_, ldst_)                                                                     \
    {                                                                                                      \
        unsigned m0_keep;                                                                                  \
        const unsigned lds_u = __builtin_amdgcn_readfirstlane(lds_base + (unsigned)(ldst_));               \
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\t" \
                     "s_mov_b32 m0, %0"                                                                     \
                     : "=&s"(m0_keep)                                                                      \
                     : "v"(voff_), "s"(sbase_), "s"(lds_u)                                                 \
                     : "memory");                                                                          \
    }
    // weights once: 2 chunks x 256 rows x 4 units = 2048 units; row swizzle as in the 3x3 kernels
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const int u = tid + r * C::THREADS;
        const int row = u >> 2, c = (u & 3) ^ b16_swz_b(row & 255);
        const int src = (row * kB16KC + 8 * c) * 2;
        B16T_GLDS(wp, src, r * C::THREADS * 16 + wave_base)
    }
    f32x4 biasq[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
        const float4 b4 = *reinterpret_cast<const float4*>(bias + nt * 16 + 4 * ku);
        biasq[nt] = f32x4{b4.x, b4.y, b4.z, b4.w};
    }
    const int b_off0 = p * 64 + ((ku ^ b16_swz_b(p)) << 4);
    // pixel fragments: tile pixel q = ty * 16 + tx at LDS row q (64 B per chunk); the row pitch is 16 pixels = 4 bank rows exactly,
    // so the unit swizzle takes the tile row's parity besides bit 1 of the column: conflict-free ds_read_b128 (scratch/swz_search.py)
    auto a_swz = [](int tx, int ty) { return (((tx >> 1) ^ ty) & 1) << 1; };
    const float fl = relu ? 0.0f : -__builtin_inff();
    for (int t = blockIdx.x; t < total_tiles; t += gridDim.x) {
        const B16Tile cur = b16_decode(t, 1, tiles_x, tiles_y, 16, 64);
        const b16_t* __restrict__ xi = x + (size_t)cur.img * H * W * 64;
        __syncthreads();      // the previous tile's pixels are consumed
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const int u = tid + r * C::THREADS;                  // chunk = u >> 10, pixel = (u >> 2) & 255, unit = u & 3
            const int chunk = u >> 10, q = (u >> 2) & 255;
            const int ty = q >> 4, tx = q & 15;
            const int Y = min(cur.y0 + ty, H - 1), X = min(cur.x0 + tx, W - 1);       // overhang: clamped (masked at the store)
            const int c = (u & 3) ^ a_swz(tx, ty);
            const unsigned src = (unsigned)((Y * W + X) * 64 + chunk * 32 + 8 * c) * 2u;
            B16T_GLDS(xi, src, C::W_BYTES + r * C::THREADS * 16 + wave_base)
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        b16x8 af[2][4];
#pragma unroll
        for (int kc = 0; kc < 2; ++kc)
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                const int tx = lpx + (mt & 1) * 8, ty = lpy + (mt >> 1) * 2;
                af[kc][mt] = *reinterpret_cast<const b16x8*>(lds_a + kc * (256 * 64) + (ty * 16 + tx) * 64 + ((ku ^ a_swz(tx, ty)) << 4));
            }
        b16_t* __restrict__ yimg = y + (size_t)cur.img * (2 * H) * (2 * W) * 64;
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) {
            f32x4 acc[4][4];
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                const b16x8 b0 = *reinterpret_cast<const b16x8*>(lds_w + (qd * 64 + nt * 16) * 64 + b_off0);
                const b16x8 b1 = *reinterpret_cast<const b16x8*>(lds_w + 256 * 64 + (qd * 64 + nt * 16) * 64 + b_off0);
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) {
                    const f32x4 c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b0, af[0][mt], biasq[nt], 0, 0, 0);
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b1, af[1][mt], c, 0, 0, 0);
                }
            }
            // output pixel (2Y + dy, 2X + dx), channels nt*16 + 4*ku ..: pairs of channel tiles joined into 16-byte runs
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                const int X = cur.x0 + lpx + (mt & 1) * 8, Y = cur.y0 + lpy + (mt >> 1) * 2;
                const bool inside = X < W && Y < H;
                const size_t pix = (size_t)(2 * Y + (qd >> 1)) * (2 * W) + (2 * X + (qd & 1));
#pragma unroll
                for (int nt = 0; nt < 4; nt += 2) {
                    unsigned pk[2][2];
#pragma unroll
                    for (int h2 = 0; h2 < 2; ++h2) {
                        typedef __bf16 b16x2 __attribute__((ext_vector_type(2)));
                        const f32x4 a = acc[mt][nt + h2];
                        const b16x2 lo = {(b16_t)fmaxf(a[0], fl), (b16_t)fmaxf(a[1], fl)}, hi = {(b16_t)fmaxf(a[2], fl), (b16_t)fmaxf(a[3], fl)};
                        pk[h2][0] = __builtin_bit_cast(unsigned, lo);
                        pk[h2][1] = __builtin_bit_cast(unsigned, hi);
                    }
                    const auto s0 = __builtin_amdgcn_permlane16_swap(pk[1][0], pk[0][0], false, false);
                    const auto s1 = __builtin_amdgcn_permlane16_swap(pk[1][1], pk[0][1], false, false);
                    const int ch = nt * 16 + ((ku & 1) ? 0 : 16) + (ku >> 1) * 8;
                    if (inside) *reinterpret_cast<uint4*>(yimg + pix * 64 + ch) = make_uint4(s0[0], s1[0], s0[1], s1[1]);
                }
            }
        }
    }
#undef B16T_GLDS
}

int convt2x2_b16_try(const void* x, const void* wp, const float* bias, void* y, int batch, int h, int w, int cin, int cout, int relu,
                     hipStream_t stream) {
    if (cin != 64 || cout != 64) return -1;
    const int tiles_x = ceil_div(w, 16), tiles_y = ceil_div(h, 16);
    const long tiles = (long)batch * tiles_x * tiles_y;
    // per image, open batch bucket only (common.h kOpenBatchBucket): input images of >= 16 tiles (64 x 64)
    if (batch < kOpenBatchBucket || (long)tiles_x * tiles_y < 16 || tiles >= 2147483647L) return -1;
    static DeviceOnce attr_once;
    attr_once.run([] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&convt2x2_b16_c64_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  B16T::LDS_BYTES);
    });
    const long slots = (long)device_cus() * 2;
    const unsigned grid = (unsigned)(tiles < slots ? tiles : slots);
    hipLaunchKernelGGL(convt2x2_b16_c64_kernel, dim3(grid), dim3(256), B16T::LDS_BYTES, stream, static_cast<const b16_t*>(x),
                       static_cast<const b16_t*>(wp), bias, static_cast<b16_t*>(y), h, w, tiles_x, tiles_y, (int)tiles, relu);
    return check_launch("holo_convt2x2_bf16");
}

// Called by conv3x3_entry<bf16> (conv_mfma.hip) for the layers this kernel covers; returns -1 when it does not take the layer.
int conv3x3_b16_try(const void* x, const void* wp, const float* bias, void* y, int batch, int h, int w, int cin, int cout, int relu,
                    int pool, int variant, hipStream_t stream) {
    const b16_t* xb = static_cast<const b16_t*>(x);
    const b16_t* wb = static_cast<const b16_t*>(wp);
    b16_t* yb = static_cast<b16_t*>(y);
    if (cin % 32 != 0) return -1;
    if (variant == 3) return (cin == 64 && cout == 64) ? launch_conv_b16_wres(xb, wb, bias, yb, batch, h, w, relu, pool, stream) : -1;
    if (cout % 128 == 0) return launch_conv_b16<8, 1, 8>(xb, wb, bias, yb, batch, h, w, cin, cout, relu, pool, stream);   // 512 px x 128 co
    if (cout % 64 == 0) return launch_conv_b16<8, 1, 4>(xb, wb, bias, yb, batch, h, w, cin, cout, relu, pool, stream);    // 512 px x 64 co
    return -1;
}

}  // namespace holo
