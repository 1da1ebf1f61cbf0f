// fp32 implicit-GEMM convolutions on the gfx950 matrix cores (v_mfma_f32_32x32x2_f32).
//
// Replaces, per layer, the aten chain  ReflectionPad2d(1) -> Conv2d(3x3) -> ReLU [-> MaxPool2d(2,2,
// ceil_mode=True)]  of net.py:78-107 / net.py:34-71, and  ConvTranspose2d(k=2,s=2) -> ReLU  of
// net.py:40,55,64.  The f32-input MFMA is an exact fp32 fmaf chain (no TF32 on gfx950), so results
// differ from the CPU reference only by summation order.
//
// GEMM view:  out[pixel][co] = sum_{tap,ci} in[pixel + off(tap)][ci] * W[tap][ci][co]
//   M = output pixels of one image patch (PR rows x 16 columns), N = output channels, K = 9*Cin.
//   A block of 4 waves owns an (PR x 16) patch x TN channels; a wave owns 64 pixels x 64 channels
//   as 2x2 MFMA tiles of 32x32 (64 accumulator VGPRs).
//
// Data movement
//   * activations are NHWC: a halo pixel's 16-channel chunk is 64 contiguous bytes (4 x 16-B units)
//   * per 16-channel K chunk the (PR+2) x 18 halo is staged ONCE in LDS (reflection resolved by the
//     loader) and reused by all 9 taps; weights are staged per (chunk, tap) as [co][16]
//   * LDS rows are 64 B per pixel / per output channel; the 16-B unit index is XOR-swizzled with
//     bits 2..3 of the row so that the ds_read_b128 of 32 rows spreads over the banks
//   * a lane's ds_read_b128 returns 4 consecutive channels: lanes 0-31 read unit 2j, lanes 32-63 unit
//     2j+1, which is exactly the k = lane>>5 split of the 32x32x2 MFMA (A and B use the same split,
//     so no data permutation is needed — only the order of the k-sum changes)
//   * register-prefetched double buffering: global loads of stage s+1 are issued before the 32 MFMAs
//     of stage s and written to the other LDS buffer after them; one barrier per stage
//
// MFMA tile -> patch mapping (chosen so a 2x2 pooling window lies in 4 consecutive accumulator regs):
//   tile row m (0..31):  x = (m&1) | ((m>>2)<<1),  y = (m>>1)&1          (2 rows x 16 columns)
//   C/D layout: m = (reg&3) + 8*(reg>>2) + 4*(lane>>5), n = lane&31
#include <cstdlib>
#include "common.h"

#ifndef HOLO_CONV_MINWAVES
#define HOLO_CONV_MINWAVES 2   // minimum waves per SIMD the register allocator must allow
#endif
#ifndef HOLO_CONV_PRIVATE_B
#define HOLO_CONV_PRIVATE_B 0   // 0: block-shared weight tile, one barrier per tap (default: measured faster);
                                // 1: every wave DMAs its own weight rows, barrier only per chunk (-5 % fp32, -19 % bf16)
#endif
#ifndef HOLO_CONV_ABLATE
#define HOLO_CONV_ABLATE 0   // timing experiments only (results are garbage): 1 = no staging loads in the main loop, 2 = no stage waits/barriers
#endif
#ifndef HOLO_CONV_DMA_J
#define HOLO_CONV_DMA_J -1    // fragment step of a stage behind which the staging loads are issued (-1: the last one)
#endif
#ifndef HOLO_CONV_TRIPLE_F32
#define HOLO_CONV_TRIPLE_F32 1
#endif
#ifndef HOLO_CONV_TRIPLE_B
#define HOLO_CONV_TRIPLE_B 1   // 0: two weight buffers, queue drained at every stage (the first scheme)
#endif
#ifndef HOLO_CONV_XCD_SWIZZLE
#define HOLO_CONV_XCD_SWIZZLE 1   // +0.3 % fp32, +1 % bf16 (measured); cuts halo re-reads from HBM
#endif
#ifndef HOLO_CONV_STAMPS
#define HOLO_CONV_STAMPS 0
#endif
#ifndef HOLO_CONV_SETPRIO
#define HOLO_CONV_SETPRIO 0
#endif

namespace holo {

#if HOLO_CONV_STAMPS
// profiling build only (-DHOLO_CONV_STAMPS=1, scripts/conv_stage_stamps.py): cycle stamps of one wave's stage phases
__device__ unsigned long long holo_dbg[8];
#define HOLO_STAMP(var_) const unsigned long long var_ = __builtin_readcyclecounter()
#endif

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16_t;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 halfx8 __attribute__((ext_vector_type(8)));
// split16 ("f32s" storage) is defined in common.h: planar per 16-channel group, [16 x hi | 16 x lo'] = 64 bytes, so a lane's
// 16-byte fragment read is 8 consecutive channels of one half — one operand of v_mfma_f32_32x32x16_f16

constexpr int kKC = 16;  // smallest K chunk (channels); weights/activations need Cin % 16 == 0
constexpr int kPW = 16;  // patch width in pixels

// KC = channels per stage: 16 (one 64-B LDS row per pixel / output channel) or 32 (128-B rows, twice the MFMAs
// between two barriers; used where the deeper buffers still leave two blocks per CU)
// T = float (exact fp32 MFMA, v_mfma_f32_32x32x2_f32) or bf16_t (v_mfma_f32_32x32x16_bf16, fp32 accumulate; BASELINE
// config 3 "bf16 convs").  With 64-byte rows (16 fp32 / 32 bf16 channels) the staging plan, the swizzle and the
// fragment reads are byte-for-byte the same for both: a lane's 16-byte read is 4 fp32 k-steps or one bf16 MFMA.
template <int WM, int WN, bool CONVT, int KC, typename T, int WTM = 2, int WTN = WTM>
struct ConvCfg {
    static constexpr int ES = (int)sizeof(T);
    static constexpr int EPU = 16 / ES;        // elements per 16-byte unit
    static constexpr int UPR = KC * ES / 16;   // 16-byte units per LDS row
    static constexpr int ROWB = KC * ES;       // bytes per LDS row
    static_assert(ROWB == 64 || ROWB == 128, "LDS rows are 64 or 128 bytes");
    // WTM x WTN = MFMA tiles per wave along pixels / output channels: 2x2 (wave tile 64 px x 64 co), 1x1 (32 x 32,
    // for grids that would otherwise underfill the chip: 4x the blocks per layer) or 4x2 (128 px x 64 co: the bf16
    // MFMA is 16x faster than the fp32 one, so the kernel is bound by L2->LDS staging and LDS fragment reads; the
    // taller wave tile halves the staged and re-read bytes per MFMA)
    static constexpr int PR = WM * 2 * WTM;  // patch rows
    static constexpr int TN = WN * 32 * WTN;
    static constexpr int HALO = CONVT ? 0 : 1;
    static constexpr int HR = PR + 2 * HALO;
    static constexpr int HW = kPW + 2 * HALO;
    static constexpr int TAPS = CONVT ? 1 : 9;
    static constexpr int THREADS = WM * WN * 64;
    static constexpr int A_UNITS = HR * HW * UPR;  // 16-byte units per chunk
    static constexpr int A_ROUNDS = (A_UNITS + THREADS - 1) / THREADS;
    static constexpr int B_UNITS = TN * UPR;
    static constexpr int B_ROUNDS = (B_UNITS + THREADS - 1) / THREADS;
    static constexpr int A_BYTES = HR * HW * ROWB;
    static constexpr int B_BYTES = TN * ROWB;
    // wave-private weight tiles: each wave stages the 32*WTN weight rows it consumes itself (two buffers)
    static constexpr int BW_ROWS = 32 * WTN;
    static constexpr int BW_BYTES = BW_ROWS * ROWB;
    static constexpr int BP_ROUNDS = BW_ROWS * UPR / 64;
    // Weight tiles are triple buffered wherever the stage pattern allows it (9 taps: buffer = tap % 3 at compile time; every wave
    // issues the same number of weight loads): stage s issues the loads of stage s+2 and the wait at its end leaves exactly those
    // in flight (counted vmcnt; loads complete in order), followed by a RAW s_barrier behind an lgkmcnt-only wait —
    // __syncthreads() would drain the LDS-DMA queue (its fence waits for vmcnt(0)).  +8 % bf16, +5 % f32s, +0.6 % fp32.
    // A weight tile smaller than one block-wide round (64 output channels under an 8-wave block) is loaded by the first
    // B_TAIL_WAVES waves only; the counted waits then use the wave's own number of weight loads per stage.
    static constexpr int B_TAIL_WAVES = (B_UNITS % THREADS) / 64;
    static_assert(B_UNITS % 64 == 0, "weight tiles are whole 1-KiB pieces");
    static constexpr int NBUF_B = (!CONVT && !HOLO_CONV_PRIVATE_B && HOLO_CONV_TRIPLE_B && TAPS % 3 == 0 && (HOLO_CONV_TRIPLE_F32 || !__is_same(T, float))) ? 3 : 2;
    static constexpr int A_MIN_ROUNDS = A_UNITS % THREADS == 0 ? A_ROUNDS : A_ROUNDS - 1;
    static constexpr int LDS_BYTES = 2 * A_BYTES + (HOLO_CONV_PRIVATE_B ? WM * WN * 2 * BW_BYTES : NBUF_B * B_BYTES);
    // A is re-staged every chunk; for CONVT every stage is a new chunk
};

// byte offset of 16-byte unit `unit` of LDS row `row`: the unit index is XOR-swizzled with row bits so that the
// ds_read_b128 of 16 consecutive rows at one unit covers all 64 banks (64-B rows: 4 rows per 256-B bank row,
// swizzle by row>>2; 128-B rows: 2 rows per bank row, swizzle by row>>1)
template <int ROWB>
__device__ __forceinline__ int swz_unit(int row, int unit) {
    return ROWB == 64 ? (unit ^ ((row >> 2) & 3)) : (unit ^ ((row >> 1) & 7));
}
template <int ROWB>
__device__ __forceinline__ int swz(int row, int unit) { return row * ROWB + (swz_unit<ROWB>(row, unit) << 4); }

// Halo (A operand) tile, 64-B rows.  ds_read_b128 is served in four groups of 16 NON-contiguous lanes ({0-3,12-15,20-27},
// {4-11,16-19,28-31} and the same + 32) over 64 banks, i.e. 16 slots of 16 B; a row holds 4 slots, the quad chosen by the
// row index mod 4.  With the tile -> pixel mapping below a group reads 8 pixel columns on each of 2 pixel rows, and the
// row-index swizzle above put two of them on one slot for every tap (2 LDS cycles per group: the constant 0.333
// SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE of the first profiles).  Keyed on the pixel coordinates instead — bit 2 of the
// column and the parity of the row — the 16 lanes of a group fall on 16 different slots for all nine taps.
template <int ROWB>
__device__ __forceinline__ int swz_a_unit(int hx, int hy, int q, int unit) {
    return ROWB == 64 ? (unit ^ (((hx >> 2) & 1) | ((hy & 1) << 1))) : swz_unit<ROWB>(q, unit);
}

template <int WM, int WN, bool CONVT, int KC, typename T, int WTM, int WTN>
__global__ __launch_bounds__(WM * WN * 64, ((WTM * WTN > 4 && KC * sizeof(T) > 64) || (__is_same(T, split16) && WTM * WTN >= 4) ? 1 : HOLO_CONV_MINWAVES)) void conv_mfma_kernel(
    const T* __restrict__ x, const T* __restrict__ wp, const float* __restrict__ bias,
    T* __restrict__ y, int H, int W, int Cin, int Cout, int N, int tiles_x, int tiles_y, int relu, int pool) {
    using C = ConvCfg<WM, WN, CONVT, KC, T, WTM, WTN>;
    constexpr bool SPLIT = __is_same(T, split16);
    constexpr bool F32 = sizeof(T) == 4 && !SPLIT;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* lds_a = smem;                   // [2][A_BYTES]
    unsigned char* lds_b = smem + 2 * C::A_BYTES;  // [2][B_BYTES]

    const int tid = threadIdx.x;
    const int n_blk = N / C::TN;
    int bid = blockIdx.x;
#if HOLO_CONV_XCD_SWIZZLE
    // Blocks are dealt round-robin over the 8 XCDs (private L2s).  Give each XCD a contiguous range of tiles so that the
    // neighbouring patches of an image, which share halo rows, and the N tiles of one patch, which share the whole halo tile,
    // meet in the same L2.  A bijection whenever the grid is a multiple of 8; speed only.
    if ((gridDim.x & 7) == 0) bid = (bid & 7) * (gridDim.x >> 3) + (bid >> 3);
#endif
    const int nb = bid % n_blk;
    bid /= n_blk;
    const int tx = bid % tiles_x;
    bid /= tiles_x;
    const int ty = bid % tiles_y;
    const int img = bid / tiles_y;
    const int y0 = ty * C::PR, x0 = tx * kPW;
    const int n0 = nb * C::TN;
    const T* __restrict__ ximg = x + (size_t)img * H * W * Cin;
    const int nchunks = Cin / KC;

    // ---- staging plans (fixed per block) -------------------------------------------------------
    // 16-byte unit u = tid + r*THREADS of the LDS image holds unit (u&3) ^ swizzle of halo pixel / output
    // channel u>>2.  With LDS-DMA the destination is lane-linear (wave base + lane*16), so the XOR swizzle
    // is applied to the per-lane SOURCE address; the ds_read side applies the same XOR (swz()).
    unsigned a_src[C::A_ROUNDS];
#pragma unroll
    for (int r = 0; r < C::A_ROUNDS; ++r) {
        const int u = tid + r * C::THREADS;
        const int q = u / C::UPR;
        const int hy = q / C::HW, hx = q % C::HW;
        const int c = swz_a_unit<C::ROWB>(hx, hy, q, u % C::UPR);
        const int ry = reflect1(y0 - C::HALO + hy, H);
        const int rx = reflect1(x0 - C::HALO + hx, W);
        a_src[r] = (unsigned)((ry * W + rx) * Cin + C::EPU * c) * (unsigned)C::ES;   // byte offset within the image (< 2^32: entry check)
    }
    const int lane = tid & 63, wave = tid >> 6;
    const int wave_s = __builtin_amdgcn_readfirstlane(wave);   // wave-uniform copy for scalar branches
    const int wm = wave / WN, wn = wave % WN;
#if HOLO_CONV_PRIVATE_B
    // Experiment (off by default): weight tile WAVE-PRIVATE — unit u = lane + 64*r of this wave's region holds
    // (swizzled) unit u % UPR of the wave's weight row u / UPR.  Only the issuing wave reads it, so its own
    // s_waitcnt vmcnt(0) is all the synchronisation the weights need and the block barrier remains only where the
    // SHARED halo tile swaps (once per chunk instead of once per tap; the per-tap barrier held ~10 % of the
    // wave-cycles in the PMC profile).  Price: waves that share output channels fetch the same rows twice and each
    // lane issues twice the LDS-DMA instructions — measured 126.8 vs 133.8 TFLOP/s (fp32), 695 vs 860 (bf16).
    int b_src[C::BP_ROUNDS];
#pragma unroll
    for (int r = 0; r < C::BP_ROUNDS; ++r) {
        const int u = lane + r * 64;
        const int row = u / C::UPR, c = swz_unit<C::ROWB>(row, u % C::UPR);
        b_src[r] = (wn * C::BW_ROWS + row) * KC + C::EPU * c;
    }
    const int b_region = 2 * C::A_BYTES + wave * 2 * C::BW_BYTES;   // byte offset of this wave's two weight buffers
#else
    int b_src[C::B_ROUNDS];
#pragma unroll
    for (int r = 0; r < C::B_ROUNDS; ++r) {
        const int u = tid + r * C::THREADS;
        const int co = u / C::UPR, c = swz_unit<C::ROWB>(co, u % C::UPR);
        b_src[r] = (co * KC + C::EPU * c) * C::ES;   // byte offset within a [N][KC] weight tile
    }
#endif
    // global_load_lds_dwordx4: no staging VGPRs, no ds_write; completion = s_waitcnt vmcnt(0)
    // byte offset of this wave's first unit within a block-wide round: wave-uniform, kept in an SGPR so that LDS
    // destinations are pure scalar arithmetic
    const int wave_base = __builtin_amdgcn_readfirstlane((tid & ~63) * 16);
    const unsigned lds_base = (unsigned)(unsigned long long)(const __attribute__((address_space(3))) void*)smem;
// Issued from inline asm on purpose: through the builtin, hipcc orders every later ds_read behind the
// pending LDS write and drains vmcnt(0) before the first fragment read of the stage (seen in the ISA),
// which serialises the DMA with the MFMAs.  The asm form is invisible to that pass; completion is our
// own s_waitcnt vmcnt(0) at the end of the stage.  M0 (LDS base of the DMA) is saved/restored.
#define HOLO_GLDS16(gsrc_, ldst_)                                                                          \
    {                                                                                                     \
        unsigned m0_keep;                                                                                 \
        const unsigned lds_u = __builtin_amdgcn_readfirstlane(lds_base + (unsigned)(ldst_));              \
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\t" \
                     "s_mov_b32 m0, %0"                                                                    \
                     : "=&s"(m0_keep)                                                                     \
                     : "v"(gsrc_), "s"(lds_u)                                                             \
                     : "memory");                                                                         \
    }
// same with a wave-uniform 64-bit base in SGPRs and a 32-bit per-lane byte offset: no per-load VALU address arithmetic
#define HOLO_GLDS16_S(sbase_, voff_, ldst_)                                                                \
    {                                                                                                     \
        unsigned m0_keep;                                                                                 \
        const unsigned lds_u = __builtin_amdgcn_readfirstlane(lds_base + (unsigned)(ldst_));              \
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\t" \
                     "s_mov_b32 m0, %0"                                                                    \
                     : "=&s"(m0_keep)                                                                     \
                     : "v"(voff_), "s"(sbase_), "s"(lds_u)                                                \
                     : "memory");                                                                         \
    }
#define HOLO_LOAD_A(chunk_, buf_)                                                                         \
    _Pragma("unroll") for (int r = 0; r < C::A_ROUNDS; ++r) {                                             \
        if (C::A_UNITS % C::THREADS == 0 || tid + r * C::THREADS < C::A_UNITS)                            \
            HOLO_GLDS16_S(ximg + (chunk_) * KC, a_src[r],                                      \
                          (buf_) * C::A_BYTES + r * C::THREADS * 16 + wave_base)                         \
    }
#if HOLO_CONV_PRIVATE_B
#define HOLO_LOAD_B(wsrc_, buf_)                                                                          \
    {                                                                                                     \
        _Pragma("unroll") for (int r = 0; r < C::BP_ROUNDS; ++r)                                          \
            HOLO_GLDS16((wsrc_) + b_src[r], b_region + (buf_) * C::BW_BYTES + r * 1024)                   \
    }
#else
#define HOLO_LOAD_B(wsrc_, buf_)                                                                          \
    {                                                                                                     \
        _Pragma("unroll") for (int r = 0; r < C::B_ROUNDS; ++r) {                                         \
            if (C::B_UNITS % C::THREADS == 0 || tid + r * C::THREADS < C::B_UNITS)                        \
                HOLO_GLDS16_S((wsrc_), b_src[r], 2 * C::A_BYTES + (buf_) * C::B_BYTES + r * C::THREADS * 16 + wave_base) \
        }                                                                                                 \
    }
#endif
#define HOLO_STAGE_WAIT() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#define HOLO_STAGE_WAIT_N(n_) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n_) : "memory")
// leave this wave's weight loads of ONE stage (+ extra_ halo loads) in flight: a wave beyond the tail of a partial round has one fewer
#define HOLO_STAGE_WAIT_B(extra_)                                                                          \
    {                                                                                                     \
        if (C::B_TAIL_WAVES == 0 || wave_s < C::B_TAIL_WAVES) { HOLO_STAGE_WAIT_N(C::B_ROUNDS + (extra_)); } \
        else { HOLO_STAGE_WAIT_N(C::B_ROUNDS - 1 + (extra_)); }                                           \
    }

    // ---- per-lane MFMA operand coordinates -----------------------------------------------------------
    const int hsel = lane >> 5, li = lane & 31;
    const int lpx = (li & 1) | ((li >> 2) << 1);
    const int lpy = wm * (2 * WTM) + ((li >> 1) & 1);  // + 2*mt
    int b_off[WTN];
#pragma unroll
    for (int nt = 0; nt < WTN; ++nt)
        b_off[nt] = swz<C::ROWB>((HOLO_CONV_PRIVATE_B ? 0 : wn * (32 * WTN)) + nt * 32 + li, hsel);

    // acc: the MFMA chain of the current 16-channel chunk (9 taps x 16 channels = 144 terms);
    // tot: running sum of the finished chunks.  One fp32 fmaf chain over all 9*Cin terms (up to 4608)
    // was measured ~8x noisier against float64 than the CPU reference's blocked sums; folding per chunk
    // shortens every chain to 144 + Cin/16 terms.  (The transposed conv has K = Cin <= 256: no fold.)
    constexpr bool FOLD = !CONVT && (F32 || SPLIT);   // bf16 inputs dominate the error there: plain fp32 accumulation
    // transposed product (lane = pixel, registers = channels) wherever the epilogue's stores matter: all bf16 layers and
    // the fp32 transposed conv (4x the output pixels of its input, K = Cin only)
    constexpr bool TRANSPOSED = !F32 || CONVT;   // (split operands: transposed as well, fp32 quads out)
    floatx16 acc[WTM][WTN], tot[WTM][WTN];
    floatx16 accx[SPLIT ? WTM : 1][SPLIT ? WTN : 1];   // split operands: hi*lo' + lo'*hi, scaled by 2^-11 at the fold
#pragma unroll
    for (int mt = 0; mt < WTM; ++mt)
#pragma unroll
        for (int nt = 0; nt < WTN; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                acc[mt][nt][r] = 0.0f;   // live only for the transposed conv (no fold); dead otherwise
                tot[mt][nt][r] = 0.0f;
                if constexpr (SPLIT) accx[mt][nt][r] = 0.0f;
            }

    // ---- main loop ------------------------------------------------------------------------------------------
    // Stages run chunk-major with the TAPS stages of a chunk unrolled at compile time: tap offsets are
    // constants, the packed-weight pointer simply advances by one [N][16] tile per stage (layout
    // [chunk][tap][N][16]) and the two LDS buffer parities are toggled, so a stage carries no div/mod or
    // 64-bit multiply.  With the per-chunk fold the first MFMA of a chunk takes C = 0 (an inline constant):
    // the accumulators are never zeroed.
    const T* __restrict__ wsrc = wp + (size_t)n0 * KC;
    const size_t w_stage = (size_t)N * KC;
    constexpr int NB = C::NBUF_B;
    HOLO_LOAD_A(0, 0)
    HOLO_LOAD_B(wsrc, 0)
    if constexpr (NB == 3) {
        wsrc += w_stage;
        HOLO_LOAD_B(wsrc, 1)
    }
    HOLO_STAGE_WAIT();
    __syncthreads();

    float4 ah[WTM], bh[WTN];   // split operands: the hi fragments of the current 16-channel group
    int apar = 0, bpar = 0;
    for (int chunk = 0; chunk < nchunks; ++chunk) {
        const bool more_chunks = chunk + 1 < nchunks;
#pragma unroll
        for (int tap = 0; tap < C::TAPS; ++tap) {
            constexpr int kLast = C::TAPS - 1;
            const bool has_next = tap != kLast || more_chunks;

#if HOLO_CONV_STAMPS
            HOLO_STAMP(t_start);
            unsigned long long t_dma0 = 0, t_dma1 = 0;
#endif
            const unsigned char* abuf = lds_a + apar * C::A_BYTES;
#if HOLO_CONV_PRIVATE_B
            const unsigned char* bbuf = smem + b_region + bpar * C::BW_BYTES;
#else
            const unsigned char* bbuf = lds_b + (NB == 3 ? tap % 3 : bpar) * C::B_BYTES;
#endif
            const int dy = CONVT ? 0 : tap / 3, dx = CONVT ? 0 : tap % 3;   // constants after unrolling
            int a_off[WTM];
#pragma unroll
            for (int mt = 0; mt < WTM; ++mt) {
                const int hy = lpy + 2 * mt + dy, hx = lpx + dx;
                const int q = hy * C::HW + hx;
                a_off[mt] = q * C::ROWB + (swz_a_unit<C::ROWB>(hx, hy, q, hsel) << 4);
            }
            // lanes 0-31 read unit 2j, lanes 32-63 unit 2j+1 (the k = lane>>5 split of the 32x32x2 MFMA); the
            // swizzle is an XOR on the unit index, so unit 2j+h sits at (offset of unit h) ^ (j << 5)
#pragma unroll
            for (int j = 0; j < C::UPR / 2; ++j) {
                float4 av[WTM], bv[WTN];
#pragma unroll
                for (int mt = 0; mt < WTM; ++mt) av[mt] = *reinterpret_cast<const float4*>(abuf + (a_off[mt] ^ (j << 5)));
#pragma unroll
                for (int nt = 0; nt < WTN; ++nt) bv[nt] = *reinterpret_cast<const float4*>(bbuf + (b_off[nt] ^ (j << 5)));
#if !(HOLO_CONV_ABLATE & 1)
#if HOLO_CONV_STAMPS
                if (NB == 3 && j == (HOLO_CONV_DMA_J < 0 ? C::UPR / 2 - 1 : HOLO_CONV_DMA_J)) t_dma0 = __builtin_readcyclecounter();
#endif
                if (NB == 3 && j == (HOLO_CONV_DMA_J < 0 ? C::UPR / 2 - 1 : HOLO_CONV_DMA_J)) {
                    if (tap == kLast - 1 && more_chunks) { HOLO_LOAD_A(chunk + 1, apar ^ 1) }
                    if (tap + 2 <= kLast || more_chunks) {
                        wsrc += w_stage;
                        HOLO_LOAD_B(wsrc, (tap + 2) % 3)
                    }
                }
#if HOLO_CONV_STAMPS
                if (NB == 3 && j == (HOLO_CONV_DMA_J < 0 ? C::UPR / 2 - 1 : HOLO_CONV_DMA_J)) t_dma1 = __builtin_readcyclecounter();
#endif
                if (NB != 3 && j == 0) {
                    // the staging loads of the next stage are issued behind the first fragment reads, so their issue
                    // slots fall into the LDS latency the first MFMA waits for anyway (ablation: loads in front of the
                    // reads cost 5.6 % of the kernel, the per-stage barrier nothing)
                    if (has_next) {
                        wsrc += w_stage;
                        HOLO_LOAD_B(wsrc, bpar ^ 1)
                    }
                    if (tap == kLast && more_chunks) { HOLO_LOAD_A(chunk + 1, apar ^ 1) }
                }
#endif
#if HOLO_CONV_SETPRIO
                __builtin_amdgcn_s_setprio(1);
#endif
                if constexpr (SPLIT) {
                    // j-th pair of 16-byte units = half a 16-channel group: even j -> hi units, odd j -> lo' units of group j/2
                    // (unit 4g + h = hi, 4g + 2 + h = lo').  The MFMAs run when both halves of a group are in registers.
                    static_assert(!SPLIT || (C::UPR % 4 == 0), "split rows hold whole 16-channel groups");
                    if ((j & 1) == 0) {
#pragma unroll
                        for (int mt = 0; mt < WTM; ++mt) ah[mt] = av[mt];
#pragma unroll
                        for (int nt = 0; nt < WTN; ++nt) bh[nt] = bv[nt];
                    } else {
#pragma unroll
                        for (int mt = 0; mt < WTM; ++mt)
#pragma unroll
                            for (int nt = 0; nt < WTN; ++nt) {
                                const halfx8 a_hi = __builtin_bit_cast(halfx8, ah[mt]), a_lo = __builtin_bit_cast(halfx8, av[mt]);
                                const halfx8 b_hi = __builtin_bit_cast(halfx8, bh[nt]), b_lo = __builtin_bit_cast(halfx8, bv[nt]);
                                const floatx16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                                const bool first = FOLD && tap == 0 && j == 1;
                                acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b_hi, a_hi, first ? zero : acc[mt][nt], 0, 0, 0);
                                // the cross terms are never folded: their rounding errors are already scaled by 2^-11
                                accx[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b_hi, a_lo, accx[mt][nt], 0, 0, 0);
                                accx[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b_lo, a_hi, accx[mt][nt], 0, 0, 0);
                            }
                    }
                } else if constexpr (F32) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
#pragma unroll
                        for (int mt = 0; mt < WTM; ++mt) {
                            const float a = k == 0 ? av[mt].x : (k == 1 ? av[mt].y : (k == 2 ? av[mt].z : av[mt].w));
#pragma unroll
                            for (int nt = 0; nt < WTN; ++nt) {
                                const float b = k == 0 ? bv[nt].x : (k == 1 ? bv[nt].y : (k == 2 ? bv[nt].z : bv[nt].w));
                                if (FOLD && tap == 0 && j == 0 && k == 0) {
                                    const floatx16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, zero, 0, 0, 0);
                                } else if (TRANSPOSED) {   // same products in the same k order: bit-identical, D[channel][pixel]
                                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(b, a, acc[mt][nt], 0, 0, 0);
                                } else {
                                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[mt][nt], 0, 0, 0);
                                }
                            }
                        }
                    }
                } else {
                    // lane (r = lane&31, h = lane>>5) holds A[r][k = 8h + e], B[k = 8h + e][r]: the 8 bf16 of unit 2j+h
#pragma unroll
                    for (int mt = 0; mt < WTM; ++mt)
#pragma unroll
                        for (int nt = 0; nt < WTN; ++nt)
                            // operands swapped: D[m = channel][n = pixel], so a lane ends up with 16 channels of ONE pixel and
                            // the epilogue can store 16-byte pieces of an NHWC pixel (see the bf16 epilogue)
                            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, bv[nt]),
                                                                                  __builtin_bit_cast(bf16x8, av[mt]),
                                                                                  acc[mt][nt], 0, 0, 0);
                }
#if HOLO_CONV_SETPRIO
                __builtin_amdgcn_s_setprio(0);
#endif
            }
            if (FOLD && tap == kLast) {
#pragma unroll
                for (int mt = 0; mt < WTM; ++mt)
#pragma unroll
                    for (int nt = 0; nt < WTN; ++nt) {
                        tot[mt][nt] += acc[mt][nt];
                    }
            }
#if HOLO_CONV_STAMPS
            HOLO_STAMP(t_mfma_end);
#endif
#if !(HOLO_CONV_ABLATE & 2)
            if constexpr (NB == 3) {
                if (has_next) {
                    if (tap + 2 <= kLast) { HOLO_STAGE_WAIT_B(0) }
                    else if (!more_chunks) { HOLO_STAGE_WAIT(); }
                    else if (tap == kLast - 1) { HOLO_STAGE_WAIT_B(C::A_MIN_ROUNDS) }
                    else { HOLO_STAGE_WAIT_B(0) }
                    __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0) only: the fragment reads of this stage are done
                    __builtin_amdgcn_s_barrier();          // a raw barrier: __syncthreads() would drain the in-flight DMAs
                    asm volatile("" ::: "memory");         // and no LDS access may be moved across it by the compiler
                }
            } else {
                HOLO_STAGE_WAIT();                                   // this wave's DMAs (next weights, its share of the next halo)
                if (!HOLO_CONV_PRIVATE_B || tap == kLast) __syncthreads();   // shared data: everybody's DMAs have landed
            }
#endif
#if HOLO_CONV_STAMPS
            {
                HOLO_STAMP(t_end);
                if (blockIdx.x == 1000 && tid == 0) {
                    holo_dbg[0] += t_end - t_start;          // whole stage
                    holo_dbg[1] += t_dma1 - t_dma0;          // staging-load issue
                    holo_dbg[2] += t_end - t_mfma_end;       // wait + barrier
                    holo_dbg[3] += 1;
                }
            }
#endif
            bpar ^= 1;
        }
        apar ^= 1;
    }

    // ---- bf16 epilogue ---------------------------------------------------------------------------------------------
    // The bf16 product is computed transposed: lane = pixel (tile column li), registers = channels
    // (r&3) + 8*(r>>2) + 4*hsel of the 32-channel tile.  Each quad of registers is 4 consecutive channels = 8 bytes of
    // bf16; lanes l and l+32 hold neighbouring quads of the same pixel, so one v_permlane32_swap per dword turns two
    // quads into one 16-byte run per lane: 2 x global_store_dwordx4 per tile instead of 16 two-byte stores (without
    // its stores the bf16 kernel ran 14 % faster, the transposed conv 2x: the epilogue was store-issue bound).
    if constexpr (TRANSPOSED) {
        const int px_x = x0 + lpx;
        const int Ho = (H + 1) >> 1, Wo = (W + 1) >> 1;
#pragma unroll
        for (int mt = 0; mt < WTM; ++mt) {
            const int X = px_x, Y = y0 + wm * (2 * WTM) + 2 * mt + ((li >> 1) & 1);
            const bool inside = X < W && Y < H;
#pragma unroll
            for (int nt = 0; nt < WTN; ++nt) {
                const int cbase = n0 + wn * (32 * WTN) + nt * 32;          // first GEMM column of this tile (wave-uniform)
                // pool == 2 on the transposed-conv kernel = a PLAIN 1x1 convolution (holo_conv1x1_f32): the same one-tap GEMM, its
                // N columns stored as the channels of the input's own pixel instead of the four output quadrants
                const bool plain = CONVT && pool == 2;
                const int co_base = CONVT && !plain ? cbase % Cout : cbase;
                const int qd = CONVT && !plain ? cbase / Cout : 0;          // (dy,dx) quadrant of a transposed conv
                float v[16];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 b4 = *reinterpret_cast<const float4*>(bias + co_base + 8 * q + 4 * hsel);
                    const float bq[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        float t0;
                        if constexpr (FOLD && SPLIT) t0 = tot[mt][nt][4 * q + i] + accx[mt][nt][4 * q + i] * 0.00048828125f;   // 2^-11
                        else if constexpr (FOLD) t0 = tot[mt][nt][4 * q + i];
                        else if constexpr (SPLIT) t0 = acc[mt][nt][4 * q + i] + accx[mt][nt][4 * q + i] * 0.00048828125f;
                        else t0 = acc[mt][nt][4 * q + i];
                        const float t = t0 + bq[i];
                        v[4 * q + i] = relu ? fmaxf(t, 0.0f) : t;
                    }
                }
                bool writer = inside;
                size_t pix;                                                  // output pixel index (in pixels)
                T* __restrict__ yimg;
                int cstride = Cout;                                          // channels per output pixel
                if (CONVT && plain) {
                    cstride = N;
                    yimg = y + (size_t)img * H * W * N;
                    pix = (size_t)Y * W + X;
                } else if (CONVT) {
                    yimg = y + (size_t)img * (2 * H) * (2 * W) * Cout;
                    pix = (size_t)(2 * Y + (qd >> 1)) * (2 * W) + (2 * X + (qd & 1));
                } else if (!pool) {
                    yimg = y + (size_t)img * H * W * Cout;
                    pix = (size_t)Y * W + X;
                } else {
                    // MaxPool2d(2,2,ceil_mode=True): the window is the 4 lanes 4g..4g+3 (x bit0 = lane bit0, y bit0 = lane
                    // bit1).  Members beyond the image edge count as 0 (pooling follows a ReLU: values are >= 0).
                    yimg = y + (size_t)img * Ho * Wo * Cout;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        float t = inside ? v[r] : 0.0f;
                        t = fmaxf(t, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, t), 0xB1, 0xF, 0xF, true)));  // quad_perm [1,0,3,2]
                        t = fmaxf(t, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, t), 0x4E, 0xF, 0xF, true)));  // quad_perm [2,3,0,1]
                        v[r] = t;
                    }
                    writer = inside && (li & 3) == 0;                       // the window's top-left pixel stores
                    pix = (size_t)(Y >> 1) * Wo + (X >> 1);
                }
                if constexpr (SPLIT) {
                    // split output: channel c of the 32-tile lives in 16-channel group c/16 as hi at half (c%16), lo' at half
                    // 16 + (c%16) of the group's 64 bytes.  Quad q = channels 8q + 4*hsel .. +3: after the half-wave swap the
                    // lower lanes hold channels 16k..16k+7 and the upper lanes 16k+8..16k+15 of group k — one 16-byte run of
                    // hi and one of lo' each.
                    unsigned hi2[8], lo2[8];
#pragma unroll
                    for (int d = 0; d < 8; ++d) {
                        typedef _Float16 halfx2 __attribute__((ext_vector_type(2)));
                        const float f0 = fminf(fmaxf(v[2 * d], -65504.0f), 65504.0f), f1 = fminf(fmaxf(v[2 * d + 1], -65504.0f), 65504.0f);
                        const halfx2 h2 = {(_Float16)f0, (_Float16)f1};
                        const halfx2 l2 = {(_Float16)((f0 - (float)h2[0]) * 2048.0f), (_Float16)((f1 - (float)h2[1]) * 2048.0f)};
                        hi2[d] = __builtin_bit_cast(unsigned, h2);
                        lo2[d] = __builtin_bit_cast(unsigned, l2);
                    }
                    unsigned char* pbase = reinterpret_cast<unsigned char*>(yimg) + (pix * Cout + co_base) * 4;
#pragma unroll
                    for (int k = 0; k < 2; ++k) {
                        auto h0 = __builtin_amdgcn_permlane32_swap(hi2[4 * k], hi2[4 * k + 2], false, false);
                        auto h1 = __builtin_amdgcn_permlane32_swap(hi2[4 * k + 1], hi2[4 * k + 3], false, false);
                        auto l0 = __builtin_amdgcn_permlane32_swap(lo2[4 * k], lo2[4 * k + 2], false, false);
                        auto l1 = __builtin_amdgcn_permlane32_swap(lo2[4 * k + 1], lo2[4 * k + 3], false, false);
                        if (writer) {
                            *reinterpret_cast<uint4*>(pbase + 64 * k + 16 * hsel) = make_uint4(h0[0], h1[0], h0[1], h1[1]);
                            *reinterpret_cast<uint4*>(pbase + 64 * k + 32 + 16 * hsel) = make_uint4(l0[0], l1[0], l0[1], l1[1]);
                        }
                    }
                    continue;
                }
                if constexpr (F32) {
                    // fp32: a quad of registers is already a 16-byte run of the pixel
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        if (writer)
                            *reinterpret_cast<float4*>(reinterpret_cast<float*>(yimg) + pix * cstride + co_base + 8 * q + 4 * hsel) =
                                make_float4(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
                    continue;
                }
                unsigned pk[8];                                              // quad q -> dwords 2q, 2q+1 (bf16 pairs)
#pragma unroll
                for (int d = 0; d < 8; ++d) {
                    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
                    const bf16x2 two = {(bf16_t)v[2 * d], (bf16_t)v[2 * d + 1]};
                    pk[d] = __builtin_bit_cast(unsigned, two);
                }
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    // quads 2k (dwords 4k, 4k+1) and 2k+1 (dwords 4k+2, 4k+3): swap the upper half-wave's first quad with
                    // the lower half-wave's second one
                    auto s0 = __builtin_amdgcn_permlane32_swap(pk[4 * k], pk[4 * k + 2], false, false);
                    auto s1 = __builtin_amdgcn_permlane32_swap(pk[4 * k + 1], pk[4 * k + 3], false, false);
                    const uint4 chunk = make_uint4(s0[0], s1[0], s0[1], s1[1]);
                    // lower half-wave: channels 16k .. 16k+7, upper half-wave: 16k+8 .. 16k+15 of this pixel
                    if (writer)
                        *reinterpret_cast<uint4*>(yimg + pix * Cout + co_base + 16 * k + 8 * hsel) = chunk;
                }
            }
        }
        return;
    }

    // ---- epilogue: bias, ReLU, optional 2x2 max-pool, NHWC store ------------------------------------------
#pragma unroll
    for (int nt = 0; nt < WTN; ++nt) {
        const int n = n0 + wn * (32 * WTN) + nt * 32 + li;  // GEMM column
        const int co = CONVT ? (n % Cout) : n;
        const float bv = bias[co];
#pragma unroll
        for (int mt = 0; mt < WTM; ++mt) {
            float v[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float t = (FOLD ? tot[mt][nt][r] : acc[mt][nt][r]) + bv;
                v[r] = relu ? fmaxf(t, 0.0f) : t;
            }
            if (CONVT) {
                // n = q*Cout + co with q = dy*2+dx: output pixel (2Y+dy, 2X+dx)
                const int qd = n / Cout;
                const int ody = qd >> 1, odx = qd & 1;
                T* __restrict__ yimg = y + (size_t)img * (2 * H) * (2 * W) * Cout;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = (r & 3) + 8 * (r >> 2) + 4 * hsel;
                    const int X = x0 + ((m & 1) | ((m >> 2) << 1));
                    const int Y = y0 + wm * (2 * WTM) + mt * 2 + ((m >> 1) & 1);
                    if (X < W && Y < H) yimg[((size_t)(2 * Y + ody) * (2 * W) + (2 * X + odx)) * Cout + co] = (T)v[r];
                }
            } else if (!pool) {
                T* __restrict__ yimg = y + (size_t)img * H * W * Cout;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = (r & 3) + 8 * (r >> 2) + 4 * hsel;
                    const int X = x0 + ((m & 1) | ((m >> 2) << 1));
                    const int Y = y0 + wm * (2 * WTM) + mt * 2 + ((m >> 1) & 1);
                    if (X < W && Y < H) yimg[((size_t)Y * W + X) * Cout + co] = (T)v[r];
                }
            } else {
                // MaxPool2d(2,2,ceil_mode=True): window = regs 4g..4g+3 (x bit0 = reg bit0, y bit0 = reg bit1);
                // window members beyond the image edge are left out (post-ReLU values are >= 0)
                const int Ho = (H + 1) >> 1, Wo = (W + 1) >> 1;
                T* __restrict__ yimg = y + (size_t)img * Ho * Wo * Cout;
                const int Y = y0 + wm * (2 * WTM) + mt * 2;  // even
                const int PY = Y >> 1;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int X = x0 + ((hsel | (g << 1)) << 1);  // even
                    const int PX = X >> 1;
                    float mx = v[4 * g];
                    if (X + 1 < W) mx = fmaxf(mx, v[4 * g + 1]);
                    if (Y + 1 < H) {
                        mx = fmaxf(mx, v[4 * g + 2]);
                        if (X + 1 < W) mx = fmaxf(mx, v[4 * g + 3]);
                    }
                    if (X < W && Y < H) yimg[((size_t)PY * Wo + PX) * Cout + co] = (T)mx;
                }
            }
        }
    }
}

#undef HOLO_LOAD_A
#undef HOLO_LOAD_B
#undef HOLO_GLDS16
#undef HOLO_GLDS16_S
#undef HOLO_STAGE_WAIT
#undef HOLO_STAGE_WAIT_N
#undef HOLO_STAGE_WAIT_B

template <int WM, int WN, bool CONVT, int KC, typename T, int WTM = 2, int WTN = WTM>
static int launch_conv(const T* x, const T* wp, const float* bias, T* y, int batch, int h, int w, int cin,
                       int cout, int relu, int pool, hipStream_t stream) {
    using C = ConvCfg<WM, WN, CONVT, KC, T, WTM, WTN>;
    const int n = CONVT ? 4 * cout : cout;
    const int tiles_x = ceil_div(w, kPW), tiles_y = ceil_div(h, C::PR);
    const long blocks = (long)batch * tiles_x * tiles_y * (n / C::TN);
    HOLO_REQUIRE(blocks > 0 && blocks < 2147483647L, HOLO_ERR_UNSUPPORTED, "conv: grid of %ld blocks", blocks);
    static DeviceOnce attr_once;   // per device, published after the attribute is set
    if (C::LDS_BYTES > 48 * 1024)
        attr_once.run([] {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_mfma_kernel<WM, WN, CONVT, KC, T, WTM, WTN>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES);
        });
    hipLaunchKernelGGL((conv_mfma_kernel<WM, WN, CONVT, KC, T, WTM, WTN>), dim3((unsigned)blocks), dim3(C::THREADS), C::LDS_BYTES,
                       stream, x, wp, bias, y, h, w, cin, cout, n, tiles_x, tiles_y, relu, pool);
    return check_launch(CONVT ? "holo_convt2x2" : "holo_conv3x3");
}

// ---- weight packing --------------------------------------------------------------------------------------------
// packed weight element i of a [.. ][KC] row -> storage: plain for float / bf16; for split16 the row is KC/16 groups of
// [16 x hi | 16 x lo'] halves
template <typename T>
__device__ __forceinline__ void store_packed(T* out, long i, int kc, float w) {
    if constexpr (__is_same(T, split16)) {
        const long row = i / kc;
        const int k = (int)(i % kc);
        _Float16* h = reinterpret_cast<_Float16*>(out) + row * kc * 2 + (k / 16) * 32 + (k % 16);
        const float f = fminf(fmaxf(w, -65504.0f), 65504.0f);
        const _Float16 hi = (_Float16)f;
        h[0] = hi;
        h[16] = (_Float16)((f - (float)hi) * 2048.0f);
    } else {
        out[i] = (T)w;
    }
}

template <typename T>
__global__ void pack_conv3x3_kernel(const float* __restrict__ w, T* __restrict__ out, int cout, int cin, int kc) {
    const long total = (long)9 * cin * cout;
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int k = (int)(i % kc);
    long r = i / kc;
    const int co = (int)(r % cout);
    r /= cout;
    const int tap = (int)(r % 9);
    const int chunk = (int)(r / 9);
    const int ci = chunk * kc + k;
    store_packed<T>(out, i, kc, w[((long)co * cin + ci) * 9 + tap]);  // OIHW, tap = ky*3+kx; packed [chunk][tap][co][kc]
}

template <typename T>
__global__ void pack_convt2x2_kernel(const float* __restrict__ w, T* __restrict__ out, int cin, int cout, int kc) {
    const long total = (long)4 * cin * cout;
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int k = (int)(i % kc);
    long r = i / kc;
    const int n = (int)(r % (4 * cout));
    const int chunk = (int)(r / (4 * cout));
    const int q = n / cout, co = n % cout;
    const int ci = chunk * kc + k;
    store_packed<T>(out, i, kc, w[((long)ci * cout + co) * 4 + q]);  // (Cin,Cout,2,2), q = dy*2+dx
}

// K-chunk per stage for a layer: the packing and the launch must agree.  fp32: 32-channel stages (half the
// barriers, 78 KB of LDS, 256 VGPRs) measured within 0.3 % of 16-channel stages on MI355X, so the leaner 16 is
// used (HOLO_CONV_KC32=1 at build time selects 32 for the 128-wide tile).  bf16: 32 channels = the same 64-byte rows.
#ifndef HOLO_CONV_KC32
#define HOLO_CONV_KC32 0
#endif
#ifndef HOLO_CONV_BF16_KC64
#define HOLO_CONV_BF16_KC64 0   // 1: bf16, 128-wide tile with 64-channel stages (128-B LDS rows, twice the MFMAs per barrier).  Was +2 % with
                                // the first staging code; with scalar-base staging loads 32-channel stages (3 blocks per CU) are 1.4 % ahead
#endif
template <typename T>
static int conv_kc(int cin, int cout, bool convt) {
    if (__is_same(T, split16)) return 32;   // 128-byte rows: two 16-channel groups per stage
    if (sizeof(T) == 2) return (HOLO_CONV_BF16_KC64 && !convt && cout % 128 == 0 && cin % 64 == 0) ? 64 : 32;
    return (HOLO_CONV_KC32 && !convt && cout % 128 == 0 && cin % 32 == 0) ? 32 : 16;
}
template <typename T> constexpr int min_cin() { return (sizeof(T) == 2 || __is_same(T, split16)) ? 32 : 16; }
template <typename T> constexpr const char* dtype_name() { return __is_same(T, split16) ? "f32s" : (sizeof(T) == 2 ? "bf16" : "f32"); }

template <typename T>
static int pack3x3_entry(const float* w_oihw, T* w_packed, int cout, int cin, void* stream) {
    HOLO_REQUIRE(w_oihw && w_packed, HOLO_ERR_ARG, "holo_pack_conv3x3_weight_%s: null pointer", dtype_name<T>());
    HOLO_REQUIRE(cin > 0 && cout > 0 && cin % min_cin<T>() == 0, HOLO_ERR_ARG,
                 "holo_pack_conv3x3_weight_%s: Cin=%d must be a positive multiple of %d (Cout=%d)", dtype_name<T>(), cin,
                 min_cin<T>(), cout);
    const long total = (long)9 * cin * cout;
    hipLaunchKernelGGL((pack_conv3x3_kernel<T>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, as_stream(stream),
                       w_oihw, w_packed, cout, cin, conv_kc<T>(cin, cout, false));
    return check_launch("holo_pack_conv3x3_weight");
}

template <typename T>
static int packt2x2_entry(const float* w_iohw, T* w_packed, int cin, int cout, void* stream) {
    HOLO_REQUIRE(w_iohw && w_packed, HOLO_ERR_ARG, "holo_pack_convt2x2_weight_%s: null pointer", dtype_name<T>());
    HOLO_REQUIRE(cin > 0 && cout > 0 && cin % min_cin<T>() == 0, HOLO_ERR_ARG,
                 "holo_pack_convt2x2_weight_%s: Cin=%d must be a positive multiple of %d (Cout=%d)", dtype_name<T>(), cin,
                 min_cin<T>(), cout);
    const long total = (long)4 * cin * cout;
    hipLaunchKernelGGL((pack_convt2x2_kernel<T>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, as_stream(stream),
                       w_iohw, w_packed, cin, cout, conv_kc<T>(cin, cout, true));
    return check_launch("holo_pack_convt2x2_weight");
}

// conv_bf16.hip: the 8-wave v_mfma_f32_16x16x32_bf16 kernel; -1 when it does not take the layer
int conv3x3_b16_try(const void* x, const void* wp, const float* bias, void* y, int batch, int h, int w, int cin, int cout, int relu,
                    int pool, int variant, hipStream_t stream);

template <typename T>
static int conv3x3_entry(const T* x_nhwc, const T* w_packed, const float* bias, T* y_nhwc, int batch, int h, int w,
                         int cin, int cout, int relu, int pool, void* stream) {
    HOLO_REQUIRE(x_nhwc && w_packed && bias && y_nhwc, HOLO_ERR_ARG, "holo_conv3x3_%s: null pointer", dtype_name<T>());
    HOLO_REQUIRE(batch > 0 && h >= 2 && w >= 2, HOLO_ERR_ARG,
                 "holo_conv3x3_%s: batch=%d h=%d w=%d (reflection padding needs h,w >= 2)", dtype_name<T>(), batch, h, w);
    HOLO_REQUIRE(cin > 0 && cin % min_cin<T>() == 0 && cout > 0 && cout % 64 == 0, HOLO_ERR_UNSUPPORTED,
                 "holo_conv3x3_%s: Cin=%d must be a multiple of %d and Cout=%d of 64", dtype_name<T>(), cin, min_cin<T>(), cout);
    HOLO_REQUIRE(!pool || relu, HOLO_ERR_UNSUPPORTED, "holo_conv3x3_%s: fused pooling requires relu", dtype_name<T>());
    HOLO_REQUIRE((long)h * w * (cin > cout ? cin : cout) < (1L << 30), HOLO_ERR_UNSUPPORTED,
                 "holo_conv3x3_%s: one image must stay below 2^30 elements (32-bit byte offsets)", dtype_name<T>());
    hipStream_t s = as_stream(stream);
    constexpr int KCD = (sizeof(T) == 2 || __is_same(T, split16)) ? 32 : 16;   // default K chunk of the dtype
    // holo_debug_set_conv_tile (tests, A/B runs): 1 = standard tile, 2 = small tile, 3 = tall bf16 wave tile; 0 = automatic
    const int force = conv_tile_override();
    // Small problems (e.g. the bundled 5 x 128x128 batches): with the 128-px x 128-co (or 256 x 64) block tile a
    // layer has only 40-320 blocks for 256 CUs.  Below ~2 blocks per CU switch to the 64 px x 64 co tile: 4x
    // the blocks, each a quarter of the work.  Weight packing does not depend on the tile.
    {
        const long big = (long)batch * ceil_div(w, kPW) * (cout % 128 == 0 ? ceil_div(h, 8) * (cout / 128) : ceil_div(h, 16) * (cout / 64));
        if ((force == 2 || (force == 0 && big < 512)) && conv_kc<T>(cin, cout, false) == KCD)
            return launch_conv<2, 2, false, KCD, T, 1>(x_nhwc, w_packed, bias, y_nhwc, batch, h, w, cin, cout, relu, pool, s);
    }
    if constexpr (sizeof(T) == 2) {
        // Large layers with Cin >= 128 and Cout % 128 == 0: the 512 px x 128 co blocks of conv_bf16.hip, from 1.5 blocks per CU up (measured
        // +5-9 % on the 256- and 512-channel layers, level at 128 -> 128, behind the 4-wave tile at Cin = 64: profiles/r04_bf16_layer_ab.txt)
        // (holo_debug_set_conv_tile 8 forces them, also for Cout = 64; 1 keeps the older kernel)
        // The v_mfma_f32_16x16x32_bf16 kernels of conv_bf16.hip (another summation grouping: chosen per image, open batch bucket only,
        // common.h kOpenBatchBucket).  64 -> 64 layers of images >= 128 x 128: the weights-resident persistent kernel
        // (holo_debug_set_conv_tile 9 forces it); Cin >= 128, Cout % 128 == 0 with >= 30 tiles per image: the staged 512 px x 128 co
        // kernel (8 forces it, also for Cout = 64; measured +5-9 % on the 256- and 512-channel layers, level at 128 -> 128, behind the
        // 4-wave tile at Cin = 64: profiles/r04_bf16_layer_ab.txt); 1 keeps the older kernel
        const long img_tiles = (long)ceil_div(w, 16) * ceil_div(h, 32);
        if (conv_kc<T>(cin, cout, false) == KCD && cin == 64 && cout == 64 &&
            (force == 9 || (force == 0 && batch >= kOpenBatchBucket && img_tiles >= 32))) {
            const int r = conv3x3_b16_try(x_nhwc, w_packed, bias, y_nhwc, batch, h, w, cin, cout, relu, pool, 3, s);
            if (r >= 0) return r;
        }
        if (conv_kc<T>(cin, cout, false) == KCD &&
            (force == 8 || (force == 0 && batch >= kOpenBatchBucket && cout % 128 == 0 && cin >= 128 && img_tiles * (cout / 128) >= 30))) {
            const int r = conv3x3_b16_try(x_nhwc, w_packed, bias, y_nhwc, batch, h, w, cin, cout, relu, pool, 1, s);
            if (r >= 0) return r;
        }
        // 8-wave blocks (holo_debug_set_conv_tile 4..7): more pixels (and output channels) per staged weight tile
        if (force >= 4 && force <= 7 && conv_kc<T>(cin, cout, false) == KCD) {
            if (cout % 128 != 0) return launch_conv<8, 1, false, KCD, T, 2, 2>(x_nhwc, w_packed, bias, y_nhwc, batch, h, w, cin, cout, relu, pool, s);   // 512 px x 64 co
            if (force == 4) return launch_conv<4, 2, false, KCD, T, 2, 2>(x_nhwc, w_packed, bias, y_nhwc, batch, h, w, cin, cout, relu, pool, s);     // 256 px x 128 co
            if (force == 5 || (force == 6 && cout % 256 != 0))
                return launch_conv<8, 1, false, KCD, T, 2, 4>(x_nhwc, w_packed, bias, y_nhwc, batch, h, w, cin, cout, relu, pool, s);                 // 512 px x 128 co
            if (force == 6) return launch_conv<4, 2, false, KCD, T, 2, 4>(x_nhwc, w_packed, bias, y_nhwc, batch, h, w, cin, cout, relu, pool, s);     // 256 px x 256 co
            return launch_conv<4, 1, false, KCD, T, 2, 4>(x_nhwc, w_packed, bias, y_nhwc, batch, h, w, cin, cout, relu, pool, s);                     // 4 waves: 256 px x 128 co
        }
        // Experiment, off unless holo_debug_set_conv_tile(3): wave tile 128 px x 64 co (4x2 MFMA tiles; 256 px x 128 co blocks,
        // or 512 px x 64 co for the 64-wide layers).  It halves the staged and re-read LDS bytes per MFMA, yet measured
        // SLOWER on MI355X (batch 64, 256x256: 746 TFLOP/s with 64-channel stages / one block per CU, 833 with
        // 32-channel stages / two blocks per CU, against 898 for the 64 x 64 wave tile): results are bit-identical.
        const bool wide = cout % 128 == 0;
        const bool narrow = cout % 128 != 0 && conv_kc<T>(cin, cout, false) == 32;
        if ((wide || narrow) && force == 3) {
            if (wide && conv_kc<T>(cin, cout, false) == 64)
                return launch_conv<2, 2, false, 64, T, 4, 2>(x_nhwc, w_packed, bias, y_nhwc, batch, h, w, cin, cout, relu, pool, s);
            if (wide) return launch_conv<2, 2, false, 32, T, 4, 2>(x_nhwc, w_packed, bias, y_nhwc, batch, h, w, cin, cout, relu, pool, s);
            return launch_conv<4, 1, false, 32, T, 4, 2>(x_nhwc, w_packed, bias, y_nhwc, batch, h, w, cin, cout, relu, pool, s);
        }
    }
    if constexpr (__is_same(T, split16)) {
        // split operands carry three accumulator sets (main, cross terms, fold): with the 64 px x 64 co wave tile that is 384
        // VGPRs and one wave per SIMD; the 64 px x 32 co wave tile (222 VGPRs, two waves per SIMD, block 128 px x 64 co) is
        // 8 % faster (12.3 vs 13.5 ms of conv per step at batch 64 x 256^2).  holo_debug_set_conv_tile(1) selects the wide tile.
        if (force != 1) return launch_conv<2, 2, false, KCD, T, 2, 1>(x_nhwc, w_packed, bias, y_nhwc, batch, h, w, cin, cout, relu, pool, s);
    }
    if (cout % 128 == 0) {
        if constexpr (!__is_same(T, split16)) {
            if (conv_kc<T>(cin, cout, false) == 2 * KCD)   // the deeper-stage variant of this dtype
                return launch_conv<2, 2, false, 2 * KCD, T>(x_nhwc, w_packed, bias, y_nhwc, batch, h, w, cin, cout, relu, pool, s);
        }
        return launch_conv<2, 2, false, KCD, T>(x_nhwc, w_packed, bias, y_nhwc, batch, h, w, cin, cout, relu, pool, s);
    }
    return launch_conv<4, 1, false, KCD, T>(x_nhwc, w_packed, bias, y_nhwc, batch, h, w, cin, cout, relu, pool, s);
}

// conv_bf16.hip: the weights-resident streaming kernel for the 64 -> 64 transposed convolution; -1 when it does not take the layer
int convt2x2_b16_try(const void* x, const void* wp, const float* bias, void* y, int batch, int h, int w, int cin, int cout, int relu,
                     hipStream_t stream);

template <typename T>
static int convt2x2_entry(const T* x_nhwc, const T* w_packed, const float* bias, T* y_nhwc, int batch, int h, int w,
                          int cin, int cout, int relu, void* stream) {
    HOLO_REQUIRE(x_nhwc && w_packed && bias && y_nhwc, HOLO_ERR_ARG, "holo_convt2x2_%s: null pointer", dtype_name<T>());
    HOLO_REQUIRE(batch > 0 && h >= 1 && w >= 1, HOLO_ERR_ARG, "holo_convt2x2_%s: batch=%d h=%d w=%d", dtype_name<T>(), batch, h, w);
    HOLO_REQUIRE(cin > 0 && cin % min_cin<T>() == 0 && cout > 0 && cout % 64 == 0, HOLO_ERR_UNSUPPORTED,
                 "holo_convt2x2_%s: Cin=%d must be a multiple of %d and Cout=%d of 64", dtype_name<T>(), cin, min_cin<T>(), cout);
    HOLO_REQUIRE((long)h * w * 4 * (cin > cout ? cin : cout) < (1L << 30), HOLO_ERR_UNSUPPORTED,
                 "holo_convt2x2_%s: one image must stay below 2^30 elements (32-bit byte offsets)", dtype_name<T>());
    // a 128-wide N tile would straddle two (dy,dx) quadrants when Cout % 128 != 0: use the 64-wide tile then
    hipStream_t s = as_stream(stream);
    constexpr int KCD = (sizeof(T) == 2 || __is_same(T, split16)) ? 32 : 16;
    if constexpr (sizeof(T) == 2) {
        if (conv_tile_override() != 1) {     // holo_debug_set_conv_tile(1) keeps the generic kernel (A/B, tests)
            const int r = convt2x2_b16_try(x_nhwc, w_packed, bias, y_nhwc, batch, h, w, cin, cout, relu, s);
            if (r >= 0) return r;
        }
    }
    if constexpr (__is_same(T, split16))   // two accumulator sets: the 64 px x 32 co wave tile keeps two waves per SIMD
        return launch_conv<2, 2, true, KCD, T, 2, 1>(x_nhwc, w_packed, bias, y_nhwc, batch, h, w, cin, cout, relu, 0, s);
    if (cout % 128 == 0) return launch_conv<2, 2, true, KCD, T>(x_nhwc, w_packed, bias, y_nhwc, batch, h, w, cin, cout, relu, 0, s);
    return launch_conv<4, 1, true, KCD, T>(x_nhwc, w_packed, bias, y_nhwc, batch, h, w, cin, cout, relu, 0, s);
}

}  // namespace holo

extern "C" {

/* y (B,h,w,cout) = x (B,h,w,cin) . W (+ bias, ReLU): a plain 1x1 convolution on the transposed convolution's one-tap MFMA GEMM
 * (the data gradient of ConvTranspose2d(k2,s2) over the space-to-depth gradient).  w_packed = holo_pack_convt2x2_weight_f32 of the
 * (cin, cout/4, 2, 2) tensor T with T[k][c][dy][dx] = W[k][(2dy+dx)*cout/4 + c]. */
int holo_conv1x1_f32(const float* x_nhwc, const float* w_packed, const float* bias, float* y_nhwc, int batch, int h, int w, int cin,
                     int cout, int relu, void* stream) {
    using namespace holo;
    HOLO_REQUIRE(x_nhwc && w_packed && bias && y_nhwc, HOLO_ERR_ARG, "holo_conv1x1_f32: null pointer");
    HOLO_REQUIRE(batch > 0 && h >= 1 && w >= 1, HOLO_ERR_ARG, "holo_conv1x1_f32: batch=%d h=%d w=%d", batch, h, w);
    HOLO_REQUIRE(cin > 0 && cin % 16 == 0 && cout > 0 && cout % 64 == 0, HOLO_ERR_UNSUPPORTED,
                 "holo_conv1x1_f32: Cin=%d must be a multiple of 16 and Cout=%d of 64", cin, cout);
    HOLO_REQUIRE((long)h * w * (cin > cout ? cin : cout) < (1L << 30), HOLO_ERR_UNSUPPORTED,
                 "holo_conv1x1_f32: one image must stay below 2^30 elements (32-bit byte offsets)");
    hipStream_t s = as_stream(stream);
    if (cout % 128 == 0) return launch_conv<2, 2, true, 16, float>(x_nhwc, w_packed, bias, y_nhwc, batch, h, w, cin, cout / 4, relu, 2, s);
    return launch_conv<4, 1, true, 16, float>(x_nhwc, w_packed, bias, y_nhwc, batch, h, w, cin, cout / 4, relu, 2, s);
}

size_t holo_conv3x3_packed_elems(int cout, int cin) { return (size_t)9 * (size_t)cin * (size_t)cout; }
size_t holo_convt2x2_packed_elems(int cin, int cout) { return (size_t)4 * (size_t)cin * (size_t)cout; }

int holo_pack_conv3x3_weight_f32(const float* w_oihw, float* w_packed, int cout, int cin, void* stream) {
    return holo::pack3x3_entry<float>(w_oihw, w_packed, cout, cin, stream);
}
int holo_pack_convt2x2_weight_f32(const float* w_iohw, float* w_packed, int cin, int cout, void* stream) {
    return holo::packt2x2_entry<float>(w_iohw, w_packed, cin, cout, stream);
}
int holo_conv3x3_f32(const float* x_nhwc, const float* w_packed, const float* bias, float* y_nhwc, int batch, int h,
                     int w, int cin, int cout, int relu, int pool, void* stream) {
    return holo::conv3x3_entry<float>(x_nhwc, w_packed, bias, y_nhwc, batch, h, w, cin, cout, relu, pool, stream);
}
int holo_convt2x2_f32(const float* x_nhwc, const float* w_packed, const float* bias, float* y_nhwc, int batch, int h,
                      int w, int cin, int cout, int relu, void* stream) {
    return holo::convt2x2_entry<float>(x_nhwc, w_packed, bias, y_nhwc, batch, h, w, cin, cout, relu, stream);
}

// bf16 activations / weights (uint16_t storage), fp32 bias and accumulation
int holo_pack_conv3x3_weight_bf16(const float* w_oihw, uint16_t* w_packed, int cout, int cin, void* stream) {
    return holo::pack3x3_entry<holo::bf16_t>(w_oihw, reinterpret_cast<holo::bf16_t*>(w_packed), cout, cin, stream);
}
int holo_pack_convt2x2_weight_bf16(const float* w_iohw, uint16_t* w_packed, int cin, int cout, void* stream) {
    return holo::packt2x2_entry<holo::bf16_t>(w_iohw, reinterpret_cast<holo::bf16_t*>(w_packed), cin, cout, stream);
}
int holo_conv3x3_bf16(const uint16_t* x_nhwc, const uint16_t* w_packed, const float* bias, uint16_t* y_nhwc, int batch,
                      int h, int w, int cin, int cout, int relu, int pool, void* stream) {
    using holo::bf16_t;
    return holo::conv3x3_entry<bf16_t>(reinterpret_cast<const bf16_t*>(x_nhwc), reinterpret_cast<const bf16_t*>(w_packed), bias,
                                       reinterpret_cast<bf16_t*>(y_nhwc), batch, h, w, cin, cout, relu, pool, stream);
}
int holo_convt2x2_bf16(const uint16_t* x_nhwc, const uint16_t* w_packed, const float* bias, uint16_t* y_nhwc, int batch,
                       int h, int w, int cin, int cout, int relu, void* stream) {
    using holo::bf16_t;
    return holo::convt2x2_entry<bf16_t>(reinterpret_cast<const bf16_t*>(x_nhwc), reinterpret_cast<const bf16_t*>(w_packed), bias,
                                        reinterpret_cast<bf16_t*>(y_nhwc), batch, h, w, cin, cout, relu, stream);
}

// split-fp16 activations / weights ("f32s": fp32 values as hi + lo' * 2^-11 halves, planar per 16 channels; 4 bytes per
// element, passed as void*), fp32 bias and accumulation
int holo_pack_conv3x3_weight_f32s(const float* w_oihw, void* w_packed, int cout, int cin, void* stream) {
    return holo::pack3x3_entry<holo::split16>(w_oihw, static_cast<holo::split16*>(w_packed), cout, cin, stream);
}
int holo_pack_convt2x2_weight_f32s(const float* w_iohw, void* w_packed, int cin, int cout, void* stream) {
    return holo::packt2x2_entry<holo::split16>(w_iohw, static_cast<holo::split16*>(w_packed), cin, cout, stream);
}
int holo_conv3x3_f32s(const void* x_nhwc, const void* w_packed, const float* bias, void* y_nhwc, int batch, int h, int w,
                      int cin, int cout, int relu, int pool, void* stream) {
    using holo::split16;
    return holo::conv3x3_entry<split16>(static_cast<const split16*>(x_nhwc), static_cast<const split16*>(w_packed), bias,
                                        static_cast<split16*>(y_nhwc), batch, h, w, cin, cout, relu, pool, stream);
}
int holo_convt2x2_f32s(const void* x_nhwc, const void* w_packed, const float* bias, void* y_nhwc, int batch, int h, int w,
                       int cin, int cout, int relu, void* stream) {
    using holo::split16;
    return holo::convt2x2_entry<split16>(static_cast<const split16*>(x_nhwc), static_cast<const split16*>(w_packed), bias,
                                         static_cast<split16*>(y_nhwc), batch, h, w, cin, cout, relu, stream);
}

#if HOLO_CONV_STAMPS
int holo_debug_read(unsigned long long* out8) { return (int)hipMemcpyFromSymbol(out8, HIP_SYMBOL(holo::holo_dbg), 64); }
int holo_debug_reset() {
    unsigned long long z[8] = {0};
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(holo::holo_dbg), z, 64);
}
#endif

}  // extern "C"
