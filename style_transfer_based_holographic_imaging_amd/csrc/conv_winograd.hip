// Winograd F(2x2, 3x3) convolution in exact fp32 on the gfx950 matrix cores (v_mfma_f32_32x32x2_f32).
//
// Same layers as conv_mfma.hip — ReflectionPad2d(1) -> Conv2d(3x3) -> [ReLU] [-> MaxPool2d(2,2,ceil_mode=True)] of
// net.py:78-107 / net.py:34-71 — with 2.25x fewer multiplications: an output tile of 2x2 pixels is
//     Y = A^T [ sum_c (G g_c G^T) .* (B^T d_c B) ] A        (Lavin & Gray 2016; d = 4x4 input patch, g = 3x3 filter)
// i.e. 16 independent GEMMs (one per position of the 4x4 transformed patch) over K = Cin instead of one GEMM over
// K = 9*Cin.  The fp32 MFMA runs at the fp32 vector rate (157 TFLOP/s) and the direct kernel already sits at 90 % of
// it, so the only way to more fp32 throughput is fewer MFMAs.  All arithmetic is fp32 (transformed weights are
// computed in float64 and rounded once); measured against a float64 evaluation of the whole network it is as close as
// the direct kernel (each accumulator sums Cin products instead of 9*Cin; tests/test_hip_winograd.py).
//
// Block = 4 waves = 128 Winograd tiles (32 x 16 output pixels) x 32 output channels; wave w owns tile rows 4w..4w+3
// (32 tiles = the 32 rows of its MFMA tiles) and all 32 channels: 16 accumulator tiles = 256 VGPRs, one wave per SIMD.
// The four waves share the weight tile and differ in pixels: per MFMA that is the least staging traffic
// (8.8 B/clk/CU; sharing the pixels and splitting the channels would need 17).
//
// Per 16-channel chunk:
//   * the raw (34 x 18) input halo is staged once in LDS (LDS-DMA, reflection resolved by the loader), pixel slots
//     ordered so that the tiles of a ds_read_b128 lane group fall on different banks (even/odd columns in separate
//     runs, 16-byte unit XOR (row>>1)&3);
//   * the transformed weights of the chunk, [16 positions][32 co][16 ch] = 32 KB, are staged once;
//   * a lane computes B^T d B for its tile in registers (4 channels at a time: 16 ds_read_b128 + 32 float4 adds) for the
//     NEXT half-chunk while the MFMAs of the current one run: 64 MFMAs = 4096 cycles per half-chunk and wave.
// Two barriers per chunk; the DMA of the next chunk's halo / weights is spread over the first MFMA gaps of a step.
//
// Epilogue: A^T M A per lane (the 16 positions of one (tile, channel) sit in the same register of the 16
// accumulators), bias, ReLU, optional 2x2 max-pool (= exactly one Winograd tile), NHWC stores of 16 bytes per lane.
#include "common.h"

#ifndef WINO_ABL
#define WINO_ABL 0   // timing experiments only (-DWINO_ABL=n side builds; wrong results): 1 no epilogue stores, 2 no re-zeroing, 4 plan not rebuilt, 8 one tile per block
#endif

#ifndef WINO_STAMPS
#define WINO_STAMPS 0   // profiling build only (scripts/wino_stamps.py): cycle stamps of block 0 / thread 0 per phase
#endif

namespace holo {

#if WINO_STAMPS
__device__ unsigned long long wino_dbg[16];
#define WSTAMP(i_)                                                        \
    {                                                                     \
        const unsigned long long now_ = __builtin_readcyclecounter();     \
        if (blockIdx.x == 0 && tid == 0) wino_dbg[i_] += now_ - t_last;   \
        t_last = now_;                                                    \
    }
#else
#define WSTAMP(i_)
#endif

typedef float floatx16 __attribute__((ext_vector_type(16)));

namespace wino {
struct TrueC { static constexpr bool value = true; };
struct FalseC { static constexpr bool value = false; };
constexpr int kThreads = 256;
constexpr int kTW = 8, kTH = 16;                 // tiles per block: 8 across, 16 down (4 per wave)
constexpr int kPW = 2 * kTW, kPH = 2 * kTH;      // 16 x 32 output pixels
constexpr int kHW = kPW + 2, kHH = kPH + 2;      // 18 x 34 halo
constexpr int kSlots = kHW * kHH;                // 612 pixel slots of 64 B
constexpr int kAUnits = kSlots * 4;              // 16-byte units per halo tile
constexpr int kARounds = (kAUnits + kThreads - 1) / kThreads;   // 10 (the last one partial)
constexpr int kABytes = ((kSlots * 64 + 1023) / 1024) * 1024;   // 39936, rounded up to whole 1-KiB DMA pieces
constexpr int kTN = 32;                          // output channels per block
constexpr int kBBytes = 16 * kTN * 64;           // 32 KB: [pos][co][16 ch]
constexpr int kBRounds = kBBytes / 16 / kThreads;   // 8
constexpr int kPlanBytes = kARounds * kThreads * 4;    // per-thread halo source offsets, kept in LDS (see make_plan)
constexpr int kLdsBytes = 2 * kABytes + 2 * kBBytes + kPlanBytes;   // 155648
static_assert(kLdsBytes <= 160 * 1024, "LDS budget");

// a - b on register pairs / quads as packed instructions (VOP3P neg_lo / neg_hi negate the second source)
typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));
__device__ __forceinline__ v2f sub2(v2f a, v2f b) {
    v2f d;
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
__device__ __forceinline__ v4f sub4(v4f a, v4f b) {
    const v2f lo = sub2(__builtin_shufflevector(a, a, 0, 1), __builtin_shufflevector(b, b, 0, 1));
    const v2f hi = sub2(__builtin_shufflevector(a, a, 2, 3), __builtin_shufflevector(b, b, 2, 3));
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3);
}
// a + b on quads as two packed instructions.  hipcc lowers the float4 `b + c` of the patch transform to four v_add_f32; beside
// v_mfma_f32_32x32x2_f32 — which runs on the vector ALU's own FMA lanes, so nothing issued to the vector ALU hides behind it
// (scripts/mfma_f32_valu_overlap.hip: 64 + 8 N cycles per MFMA with N v_add_f32 after it, 64 + 5.5 N with v_pk_add_f32) — an
// unpacked instruction costs 8 cycles for one value, a packed one 5.5 for two.  (There is no packed fp32 max on gfx950.)
__device__ __forceinline__ v4f add4(v4f a, v4f b) {
    v2f lo, hi;
    asm("v_pk_add_f32 %0, %1, %2" : "=v"(lo) : "v"(__builtin_shufflevector(a, a, 0, 1)), "v"(__builtin_shufflevector(b, b, 0, 1)));
    asm("v_pk_add_f32 %0, %1, %2" : "=v"(hi) : "v"(__builtin_shufflevector(a, a, 2, 3)), "v"(__builtin_shufflevector(b, b, 2, 3)));
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3);
}
template <typename V16>
__device__ __forceinline__ V16 sub16(V16 a, V16 b) {
    V16 d;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const v2f x = {a[2 * k], a[2 * k + 1]}, y = {b[2 * k], b[2 * k + 1]};
        const v2f r = sub2(x, y);
        d[2 * k] = r[0];
        d[2 * k + 1] = r[1];
    }
    return d;
}

// halo pixel (hy, hx) -> 64-byte slot: even and odd columns of a row in separate runs, so that the 8 tiles of a tile
// row (2 pixels apart) sit in consecutive slots
__device__ __forceinline__ int slot_of(int hy, int hx) { return hy * kHW + (hx & 1) * (kHW / 2) + (hx >> 1); }
}  // namespace wino

// LDS-DMA of one 16-byte unit per lane: wave-uniform 64-bit base in SGPRs + 32-bit per-lane byte offset, scalar LDS
// destination in M0 (conv_mfma.hip explains why this is inline asm: through the builtin hipcc drains vmcnt(0) before
// every later ds_read).
#define WINO_GLDS16_S(sbase_, voff_, ldst_)                                                                \
    {                                                                                                     \
        unsigned m0_keep;                                                                                 \
        const unsigned lds_u = __builtin_amdgcn_readfirstlane(lds_base + (unsigned)(ldst_));              \
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\t" \
                     "s_mov_b32 m0, %0"                                                                    \
                     : "=&s"(m0_keep)                                                                     \
                     : "v"(voff_), "s"(sbase_), "s"(lds_u)                                                \
                     : "memory");                                                                         \
    }

// ksplit > 1 (split-K, for launches that would leave most CUs idle): a tile's channel chunks are dealt to `ksplit` work
// items, item s summing chunks [s*n/ksplit .. (s+1)*n/ksplit) (parts differ by at most one chunk: ksplit need not divide
// n, so that items * ksplit can be matched to the CU count); item s writes its RAW partial outputs (no bias, no activation) to
// y + s * part_stride, and holo_splitk_reduce_f32 adds the partials in the fixed order s = 0, 1, ... with bias, ReLU and
// pooling.  The split index is one more dimension of the persistent tile id: the loop body is unchanged.
__global__ __launch_bounds__(wino::kThreads, 1) void conv_wino_f32_kernel(
    const float* __restrict__ x, const float* __restrict__ wp, const float* __restrict__ bias, float* __restrict__ y,
    int H, int W, int Cin, int Cout, int tiles_x, int tiles_y, int total, int relu, int pool, int ksplit, long part_stride) {
    using namespace wino;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* lds_a = smem;                  // [2][kABytes]
    unsigned char* lds_b = smem + 2 * kABytes;    // [2][kBBytes]

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
#if WINO_STAMPS
    unsigned long long t_last = 0;
#endif
    const int n_blk = Cout / kTN;
    const int nchunks_all = Cin / 16;            // chunks of the layer (stride of the packed weights)

    // ---- tiles ----------------------------------------------------------------------------------------------------
    // The grid is PERSISTENT: block b works on tiles b, b + gridDim.x, ... (one block per CU: 512 registers per lane).  A
    // tile's fixed costs — the latency of its first halo/weight loads, the staging plan, zeroing 256 accumulators, a block
    // launch — were ~10 us against 14 us of MFMA work for the 64-channel layers; here the next tile's first chunk is
    // requested during the current tile's last chunk and the first MFMA of a tile takes C = 0.
    // XCD-aware order (blocks are dealt round-robin over the 8 XCDs): XCD c gets a contiguous range of tile ids, so that
    // the channel blocks of one patch (same halo) and neighbouring patches meet in one L2 at about the same time.
    // (Images before positions — a block's consecutive tiles at one position, so that the staging plan is kept — ran 0.3 % faster
    // at batch 64 but moved 1.32 instead of 1.10 GB per launch: the halo columns shared by neighbouring patches fell out of L2.)
    // Bijective when the tile count is a multiple of 8; speed only.
    struct Tile { int nb, img, y0, x0, ks, c0, nc; };   // c0, nc: first chunk and number of chunks of this work item
    auto decode = [&](int vb) {
        int bid = vb;
        if ((total & 7) == 0) bid = (vb & 7) * (total >> 3) + (vb >> 3);
        Tile t;
        t.ks = bid % ksplit;                     // fastest: the work items of one tile share its halo in one L2
        bid /= ksplit;
        t.c0 = t.ks * nchunks_all / ksplit;
        t.nc = (t.ks + 1) * nchunks_all / ksplit - t.c0;
        t.nb = bid % n_blk;
        bid /= n_blk;
        t.x0 = (bid % tiles_x) * kPW;
        bid /= tiles_x;
        t.y0 = (bid % tiles_y) * kPH;
        t.img = bid / tiles_y;
        return t;
    };

    // ---- staging plans ------------------------------------------------------------------------------------------
    // LDS unit U = tid + 256 r of the halo image = unit u' of slot U>>2; it holds logical unit u' ^ ((hy>>1)&3) of that pixel
    // The plan lives in LDS, not in registers: with 256 accumulators + two patches there is no room for 10 more VGPRs, and a
    // spilled offset is reloaded with scratch_load + s_waitcnt vmcnt(0) right in front of its DMA — which also waits for
    // every LDS-DMA in flight and serialised the whole staging pipeline (135 instead of 265 TFLOP/s when that happened).
    // Each thread reads back only what it wrote itself: no barrier needed.
    unsigned* plan = reinterpret_cast<unsigned*>(smem + 2 * kABytes + 2 * kBBytes) + tid;
    auto make_plan = [&](const Tile& t) {
        // The thread index is made opaque here on purpose: otherwise hipcc hoists the tile-independent half of the plan (30
        // per-thread values) out of the persistent loop, spills it, and reloads it per tile with 30 serialised
        // scratch_load + s_waitcnt vmcnt(0) pairs: ~24 us per tile, more than a 64-channel tile's MFMA time.  Recomputing
        // costs ~350 VALU instructions per tile.
        int tid_o = tid;
        asm volatile("" : "+v"(tid_o));
#pragma unroll
        for (int r = 0; r < kARounds; ++r) {
            int U = tid_o + r * kThreads;
            U = U < kAUnits ? U : kAUnits - 1;        // the partial round is masked at issue; keep the address valid anyway
            const int s = U >> 2, up = U & 3;
            const int hy = s / kHW, rem = s % kHW;
            const int hx = rem < kHW / 2 ? 2 * rem : 2 * (rem - kHW / 2) + 1;
            const int u = up ^ ((hy >> 1) & 3);
            const int ry = reflect1(t.y0 - 1 + hy, H), rx = reflect1(t.x0 - 1 + hx, W);
            plan[r * kThreads] = (unsigned)((ry * W + rx) * Cin + 4 * u) * 4u;    // byte offset within the image (< 2^32: entry check)
        }
    };
    // (Shifting the plan by a constant between two tiles whose halos need no reflection — 10 read-add-write triples instead of ~350
    // instructions, two thirds of the tiles of a 256 x 256 layer — measured 0.2 % SLOWER at batch 64 x 256^2 (18.14 against 18.10 ms of
    // Winograd time per step, three alternating runs): not kept.)
    // weight tile of a chunk: [pos][co][16 ch], contiguous 32 KB in the packed tensor; LDS unit U = row*4 + u' holds
    // logical unit u' ^ ((co>>2)&3) of row (pos, co)
    // Round r covers rows (tid>>2) + 64 r: the key ((row>>2)&3) does not depend on r, so one per-lane offset serves all
    // rounds and round r adds 4096 bytes to the wave-uniform base.
    const unsigned b_src0 = (unsigned)((tid >> 2) * 64 + (((tid & 3) ^ ((tid >> 4) & 3)) << 4));
    const int wave_base = __builtin_amdgcn_readfirstlane((tid & ~63) * 16);
    const unsigned lds_base = (unsigned)(unsigned long long)(const __attribute__((address_space(3))) void*)smem;

#define WINO_LOAD_A_ROUND(abase_, buf_, r_, off_)                                                               \
    if ((r_) < kARounds - 1 || tid + (r_) * kThreads < kAUnits)                                                 \
        WINO_GLDS16_S(abase_, off_, (buf_) * kABytes + (r_) * kThreads * 16 + wave_base)
#define WINO_LOAD_B_ROUND(bbase_, buf_, r_)                                                                     \
    WINO_GLDS16_S((bbase_) + (r_) * (kThreads * 16 / 4), b_src0, 2 * kABytes + (buf_) * kBBytes + (r_) * kThreads * 16 + wave_base)

    // ---- per-lane operand coordinates ---------------------------------------------------------------------------
    const int hsel = lane >> 5, li = lane & 31;
    const int hsel_ = hsel, li_ = li, wave_ = wave;        // (the epilogue shadows these with opaque copies)
    const int ltx = li & 7, lty = li >> 3;                 // tile within the wave's 4 x 8 tiles
    const int hy0 = 2 * (4 * wave + lty);                  // halo row of the patch's first row (halo origin = y0-1)
    // patch pixel (i, j): slot = (hy0+i)*18 + (j&1)*9 + ltx + (j>>1); unit key = ((hy0+i)>>1)&3 = (lty + (i>>1)) & 3
    const int a_base = (hy0 * kHW + ltx) * 64;
    const int key0 = lty & 3, key1 = (lty + 1) & 3;
    const int b_base = li * 64;                            // row (pos, co = li): + pos*2048; unit key (li>>2)&3
    const int bkey = (li >> 2) & 3;

    floatx16 acc[16];

    // ---- software-pipelined step -----------------------------------------------------------------------------------------
    // One step = 16 positions x 4 MFMAs (64 MFMAs, 4096 cycles) on the transformed patch V of one channel quad per lane half.
    // Everything else is cut into 16 slices that ride in the shadow of the previous slice's MFMAs:
    //   * the weight fragment of position p+1 is read from LDS before the MFMAs of position p are issued;
    //   * B^T d B of the NEXT quarter's patch: slices 0-3 transform patch column p (rows of B^T) from d values read one slice
    //     earlier, slices 4-7 transform row p-4 (columns);
    //   * LDS-DMA rounds of the next chunk's halo or weights, two per slice from the first slice on.
    // LDS read latency (~100+ cycles) would otherwise be exposed once per position: the first version of this kernel read
    // each fragment right before its use and reached 58 % of the matrix pipe.  (With channel pairs instead of quads — half
    // the patch registers, 32 MFMAs per step — a slice's 128 MFMA cycles no longer cover that latency: 150 instead of 260
    // TFLOP/s.)
    // kCW = channels per lane half and step: 4 (two steps of 64 MFMAs per chunk, ds_read_b128) or 2 (four steps of 32, ds_read_b64)
    constexpr int kCW = 4;
    constexpr int kSteps = 8 / kCW;                  // steps per chunk
    typedef float vecw __attribute__((ext_vector_type(kCW)));
    // step s of a chunk: lane half h takes channels 4*u + (kCW == 2 ? 2*(s&1) : 0) + {0..kCW-1} of 16-byte unit u = 2*(s*kCW/4) + h
    auto unit_of = [&](int st) { return 2 * ((st * kCW) / 4) + hsel; };
    auto sub_of = [&](int st) { return kCW == 2 ? (st & 1) * 8 : 0; };
    auto read_b = [&](const unsigned char* bbuf, int st, int p) {
        return *reinterpret_cast<const vecw*>(bbuf + p * (kTN * 64) + b_base + ((unit_of(st) ^ bkey) << 4) + sub_of(st));
    };
    auto read_d = [&](const unsigned char* abuf, int st, int j, vecw (&d)[4]) {
        const int u = unit_of(st);
#pragma unroll
        for (int i = 0; i < 4; ++i)
            d[i] = *reinterpret_cast<const vecw*>(abuf + a_base + (i * kHW + (j & 1) * (kHW / 2) + (j >> 1)) * 64 +
                                                 ((u ^ (i < 2 ? key0 : key1)) << 4) + sub_of(st));
    };
    // rows of B^T: [1,0,-1,0], [0,1,1,0], [0,-1,1,0], [0,1,0,-1]
    // The differences are PACKED subtractions written as v_pk_add_f32 with the neg modifiers on the second source: hipcc has
    // no packed subtract and lowers `a - c` on a float4 to four v_sub_f32 (576 of them in the round-2 kernel, 12 of the 14
    // VALU instructions of every bt4).  Measured: 18.5 -> 18.3 ms of Winograd time per step (+1.5 % holograms/s).
    auto bt4 = [&](vecw a, vecw b, vecw c, vecw e, vecw& o0, vecw& o1, vecw& o2, vecw& o3) {
        o0 = wino::sub4(a, c);
        o1 = wino::add4(b, c);
        o2 = wino::sub4(c, b);
        o3 = wino::sub4(b, e);
    };
    auto stage_sync = [&]() {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's DMAs have landed
        __syncthreads();                                     // everybody's have; everybody is done reading the old buffers
    };
    // MFMAs of step st on V; transforms the patch of (tbuf, tst) into Vn;  dma (runtime, block-uniform): 0 none, 1 halo rounds
    // from `gsrc` into halo buffer nbuf, 2 weight rounds from `gsrc` into weight buffer nbuf
    auto step = [&](auto do_t_c, const unsigned char* bbuf, int st, const vecw (&V)[16], const unsigned char* tbuf, int tst,
                    vecw (&Vn)[16], int dma, const float* gsrc, int nbuf) {
        constexpr bool do_t = decltype(do_t_c)::value;   // false: no transform (a tile's last step: the next tile's first patch
                                                         // is computed after the epilogue, so that it is not live across it)
        vecw bv = read_b(bbuf, st, 0);
        vecw dcur[4], t[4][4];
        if constexpr (do_t) read_d(tbuf, tst, 0, dcur);
        unsigned off0 = plan[0], off1 = plan[kThreads];   // halo source offsets of the rounds issued in the next slice
#pragma unroll
        for (int p = 0; p < 16; ++p) {
            vecw bn = bv;
            if (p < 15) bn = read_b(bbuf, st, p + 1);
            // One LDS-DMA instruction costs 60-185 issue cycles and an fp32 MFMA covers 64: the two rounds of a slice go
            // behind its first and its third MFMA, never back to back (two in front of the slice's MFMAs left ~140 cycles
            // per slice uncovered: 17 % of a step).
            static_assert(kCW == 4, "DMA placement below assumes four MFMAs per position");
            acc[p] = __builtin_amdgcn_mfma_f32_32x32x2f32(bv[0], V[p][0], acc[p], 0, 0, 0);
            if (dma == 1 && 2 * p < kARounds) { WINO_LOAD_A_ROUND(gsrc, nbuf, 2 * p, off0) }
            if (dma == 2 && 2 * p < kBRounds) { WINO_LOAD_B_ROUND(gsrc, nbuf, 2 * p) }
            acc[p] = __builtin_amdgcn_mfma_f32_32x32x2f32(bv[1], V[p][1], acc[p], 0, 0, 0);
            acc[p] = __builtin_amdgcn_mfma_f32_32x32x2f32(bv[2], V[p][2], acc[p], 0, 0, 0);
            if (dma == 1 && 2 * p + 1 < kARounds) { WINO_LOAD_A_ROUND(gsrc, nbuf, 2 * p + 1, off1) }
            if (dma == 2 && 2 * p + 1 < kBRounds) { WINO_LOAD_B_ROUND(gsrc, nbuf, 2 * p + 1) }
            acc[p] = __builtin_amdgcn_mfma_f32_32x32x2f32(bv[3], V[p][3], acc[p], 0, 0, 0);
            if (2 * p + 2 < kARounds) off0 = plan[(2 * p + 2) * kThreads];
            if (2 * p + 3 < kARounds) off1 = plan[(2 * p + 3) * kThreads];
            if constexpr (do_t) {
                if (p < 4) {
                    bt4(dcur[0], dcur[1], dcur[2], dcur[3], t[0][p], t[1][p], t[2][p], t[3][p]);
                    if (p < 3) read_d(tbuf, tst, p + 1, dcur);   // consumed one slice later: no second register set
                }
                if (p >= 4 && p < 8) {
                    const int i = p - 4;
                    bt4(t[i][0], t[i][1], t[i][2], t[i][3], Vn[4 * i + 0], Vn[4 * i + 1], Vn[4 * i + 2], Vn[4 * i + 3]);
                }
            }
            bv = bn;
            __builtin_amdgcn_sched_barrier(0);   // keep the slices in this order
        }
    };
    // the whole transform at once (a tile's first patch)
    auto transform_now = [&](const unsigned char* abuf, vecw (&V)[16]) {
        vecw t[4][4], d[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            read_d(abuf, 0, j, d);
            bt4(d[0], d[1], d[2], d[3], t[0][j], t[1][j], t[2][j], t[3][j]);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) bt4(t[i][0], t[i][1], t[i][2], t[i][3], V[4 * i + 0], V[4 * i + 1], V[4 * i + 2], V[4 * i + 3]);
    };

    // ---- epilogue of one tile: Y = A^T M A, bias, ReLU, optional pool, NHWC store ----------------------------------
    // lane: tile li of the wave's 32 (tile row li>>3, column li&7); register r: channel (r&3) + 8*(r>>2) + 4*hsel of the block's 32
    auto epilogue = [&](const Tile& tl) {
        // lane coordinates made opaque: nothing of the epilogue's addressing may be hoisted out of the persistent loop (hoisted
        // values are live across the MFMA loop, where every VGPR is taken, and come back as scratch reloads in front of the
        // LDS-DMAs: s_waitcnt vmcnt(0) then serialises the staging pipeline)
        int li = li_, hsel = hsel_, wave = wave_;
        asm volatile("" : "+v"(li), "+v"(hsel), "+v"(wave));
        const int co0 = tl.nb * kTN + 4 * hsel;      // first channel of this lane's quad 0 (quad q: + 8q)
        const int Ho = (H + 1) >> 1, Wo = (W + 1) >> 1;
        float* __restrict__ yimg = y + (size_t)tl.ks * part_stride + (size_t)tl.img * (pool ? (size_t)Ho * Wo : (size_t)H * W) * Cout;
        const bool full = tl.y0 + kPH <= H && tl.x0 + kPW <= W;
        // A^T M A on whole accumulator tiles, ONE TILE AT A TIME.  The accumulators live in AGPRs and VALU instructions cannot
        // read those: a tile used by the VALU is copied to 16 VGPRs first.  Walking register by register over all 16 positions
        // (the natural loop) needs every tile in VGPRs at once — 256 registers on top of the next patch that is already live —
        // and spilled ~600 registers inside the persistent loop.  Row i of the 4x4 positions: u0 = M_i0 + M_i1 + M_i2,
        // u1 = M_i1 - M_i2 - M_i3 (columns of A), then o[0][.] += u (i = 0,1,2) and o[1][.] += (+,-,-) u (i = 1,2,3).
        // The reads are explicit v_accvgpr_read with an AGPR-constrained operand: a plain VALU use would constrain the whole
        // accumulator tile to the VGPR class for its entire live range, and the register allocator then shuffles tiles between
        // the two files inside the MFMA loop.  The last MFMAs of the tile were issued before the step's closing barrier; the
        // s_nops cover the 16-pass MFMA -> AGPR read hazard, which hipcc does not pad for an asm statement.
        // the bias quads are requested first: their latency hides behind the accumulator reads and the transform
        const float act_lo = relu ? 0.0f : -__builtin_inff();
        float4 bq[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) bq[q] = bias ? *reinterpret_cast<const float4*>(bias + co0 + 8 * q) : make_float4(0.f, 0.f, 0.f, 0.f);   // split-K partials carry no bias
        asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");
        WSTAMP(9)    // addressing, bias request, hazard nops
        auto rd_tile = [&](const floatx16& t) {
            floatx16 v;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float e;
                asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(e) : "a"(t[r]));
                v[r] = e;
            }
            return v;
        };
        floatx16 o00, o01, o10, o11;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            floatx16 u0 = rd_tile(acc[4 * i + 0]);
            floatx16 u1 = rd_tile(acc[4 * i + 1]);
            u0 += u1;
            {
                const floatx16 m2 = rd_tile(acc[4 * i + 2]);
                u0 += m2;
                u1 = sub16(u1, m2);
            }
            u1 = sub16(u1, rd_tile(acc[4 * i + 3]));
            if (i == 0) {
                o00 = u0;
                o01 = u1;
            } else if (i == 1) {
                o00 += u0;
                o01 += u1;
                o10 = u0;
                o11 = u1;
            } else if (i == 2) {
                o00 += u0;
                o01 += u1;
                o10 = sub16(o10, u0);
                o11 = sub16(o11, u1);
            } else {
                o10 = sub16(o10, u0);
                o11 = sub16(o11, u1);
            }
        }
        WSTAMP(10)   // accumulator reads + A^T M A
        // The product is computed TRANSPOSED (weights as the A operand): lane = tile li, register r = channel
        // (r&3) + 8*(r>>2) + 4*hsel of the block's 32, so a quad of registers is 4 consecutive channels = one 16-byte piece of an
        // NHWC pixel: 16 global_store_dwordx4 per lane instead of 64 global_store_dword (the epilogue is store-issue bound).
        const int Y = tl.y0 + 2 * (4 * wave + (li >> 3)), X = tl.x0 + 2 * (li & 7);
        const bool in00 = Y < H && X < W, in01 = Y < H && X + 1 < W, in10 = Y + 1 < H && X < W, in11 = Y + 1 < H && X + 1 < W;
        float* __restrict__ q00 = yimg + ((size_t)Y * W + X) * Cout + co0;
        float* __restrict__ qp = yimg + ((size_t)(Y >> 1) * Wo + (X >> 1)) * Cout + co0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 b4 = bq[q];
            float4 o[2][2];
            o[0][0] = make_float4(o00[4 * q] + b4.x, o00[4 * q + 1] + b4.y, o00[4 * q + 2] + b4.z, o00[4 * q + 3] + b4.w);
            o[0][1] = make_float4(o01[4 * q] + b4.x, o01[4 * q + 1] + b4.y, o01[4 * q + 2] + b4.z, o01[4 * q + 3] + b4.w);
            o[1][0] = make_float4(o10[4 * q] + b4.x, o10[4 * q + 1] + b4.y, o10[4 * q + 2] + b4.z, o10[4 * q + 3] + b4.w);
            o[1][1] = make_float4(o11[4 * q] + b4.x, o11[4 * q + 1] + b4.y, o11[4 * q + 2] + b4.z, o11[4 * q + 3] + b4.w);
            // ReLU without a branch: max(x, 0) or max(x, -inf).  As `if (relu)` the two paths left their results in different
            // registers and every quad paid 24 moves to join them (ISA), on top of the branch
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b)
                    o[a][b] = make_float4(fmaxf(o[a][b].x, act_lo), fmaxf(o[a][b].y, act_lo), fmaxf(o[a][b].z, act_lo), fmaxf(o[a][b].w, act_lo));
            if (!pool) {
                // (hoisting the block-uniform `full` into its own branch with four plain stores measured 2 % slower: 18.43 against 18.05 ms)
                float* __restrict__ d = q00 + 8 * q;
                if (full || in00) *reinterpret_cast<float4*>(d) = o[0][0];
                if (full || in01) *reinterpret_cast<float4*>(d + Cout) = o[0][1];
                if (full || in10) *reinterpret_cast<float4*>(d + (size_t)W * Cout) = o[1][0];
                if (full || in11) *reinterpret_cast<float4*>(d + (size_t)W * Cout + Cout) = o[1][1];
            } else if (full || in00) {
                // MaxPool2d(2,2,ceil_mode=True): the window is this tile; members beyond the image edge are left out (post-ReLU
                // values are >= 0, so an absent member is a 0)
                auto mx4 = [](float4 a, float4 b) { return make_float4(fmaxf(a.x, b.x), fmaxf(a.y, b.y), fmaxf(a.z, b.z), fmaxf(a.w, b.w)); };
                const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
                float4 m = o[0][0];
                m = mx4(m, (full || in01) ? o[0][1] : z);
                m = mx4(m, (full || in10) ? o[1][0] : z);
                m = mx4(m, (full || in11) ? o[1][1] : z);
                *reinterpret_cast<float4*>(qp + 8 * q) = m;
            }
        }
    };

    // ---- prologue: the block's first tile -----------------------------------------------------------------------
    int vb = blockIdx.x;
    Tile cur = decode(vb);
    make_plan(cur);
    const float* __restrict__ ximg = x + (size_t)cur.img * H * W * Cin + (size_t)cur.c0 * 16;
    const float* __restrict__ wblk = wp + ((size_t)cur.nb * nchunks_all + (size_t)cur.c0) * (kBBytes / 4);   // [nb][chunk][pos][co][16]
#pragma unroll
    for (int r = 0; r < kARounds; ++r) { WINO_LOAD_A_ROUND(ximg, 0, r, plan[r * kThreads]) }
#pragma unroll
    for (int r = 0; r < kBRounds; ++r) { WINO_LOAD_B_ROUND(wblk, 0, r) }
    stage_sync();
    vecw Va[16], Vb[16];
    transform_now(lds_a, Va);

    // ---- main loop over (tile, chunk); one chunk = kSteps steps -----------------------------------------------------------
    // first step(s): MFMAs on Va; Vb = next step's patch; the next chunk's halo, then its weights, are requested (the halo
    //     buffer was released by the barrier before the previous chunk's last step, the weight buffer by the one that ended it)
    // barrier: the next halo has landed
    // last step: MFMAs on the last patch; Va = first patch of the NEXT chunk
    // barrier: the next weights have landed; this chunk's buffers are free.
    // The "next chunk" of a tile's last chunk is chunk 0 of the block's next tile.  One flat loop over (tile, chunk).
    // Accumulators are written (zeroed) and read (epilogue) through AGPR-constrained asm operands only, so that nothing but
    // the MFMAs and those statements touches them: any VALU definition or use lets the register allocator move whole tiles to
    // VGPRs, and inside the persistent loop it then kept all 16 there and copied them through one AGPR tile around every
    // position's MFMAs (seen in the ISA: 16 v_accvgpr_write per 2 MFMAs, ~200 spilled registers).
    auto zero_acc = [&]() {
#pragma unroll
        for (int p = 0; p < 16; ++p)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float z;
                asm volatile("v_accvgpr_write_b32 %0, 0" : "=a"(z));
                acc[p][r] = z;
            }
        asm volatile("s_nop 1");   // v_accvgpr_write -> MFMA operand wait states (hipcc pads nothing around asm statements)
    };
    zero_acc();
    int par = 0, c = 0;
#if WINO_STAMPS
    t_last = __builtin_readcyclecounter();
#endif
    for (;;) {
        const bool is_last = c + 1 == cur.nc;
        bool has_more = true;
        int vb_next = vb;
        Tile nxt = cur;
        const float* a_next = ximg + (c + 1) * 16;
        const float* b_next = wblk + (size_t)(c + 1) * (kBBytes / 4);
        if (is_last) {
            vb_next = vb + (int)gridDim.x;
            has_more = vb_next < total;
#if WINO_ABL & 8
            has_more = false;
#endif
            if (has_more) {
                nxt = decode(vb_next);
#if !(WINO_ABL & 4)
                if (nxt.y0 != cur.y0 || nxt.x0 != cur.x0) make_plan(nxt);   // one-tile images (the 16 x 16 maps of a 128^2 hologram): the plan stays
#endif      // the current tile's plan issued its last round in q0 of the previous chunk
                a_next = x + (size_t)nxt.img * H * W * Cin + (size_t)nxt.c0 * 16;
                b_next = wp + ((size_t)nxt.nb * nchunks_all + (size_t)nxt.c0) * (kBBytes / 4);
            }
        }
        const unsigned char* abuf = lds_a + par * kABytes;
        const unsigned char* bbuf = lds_b + par * kBBytes;
        WSTAMP(4)   // loop top: decode + plan of the next tile on a tile's last chunk
        static_assert(kSteps == 2, "the loop below is written for two steps per chunk");
        step(TrueC{}, bbuf, 0, Va, abuf, 1, Vb, has_more ? 1 : 0, a_next, par ^ 1);
        WSTAMP(0)
        stage_sync();
        WSTAMP(1)
        // ONE instance of step 1 for every chunk (a second, transform-free instance for a tile's last chunk made the register
        // allocator shuffle accumulators again): on a tile's last chunk the patch it computes is simply dropped — the next
        // tile's first patch is recomputed behind the epilogue, so that no patch is live across it (64 VGPRs the epilogue's
        // own four output tiles need: with them live the epilogue spilled and reloaded ~140 registers per tile, each reload
        // followed by s_waitcnt vmcnt(0): 21 us per tile).
        step(TrueC{}, bbuf, 1, Vb, lds_a + (par ^ 1) * kABytes, 0, Va, has_more ? 2 : 0, b_next, par ^ 1);
        WSTAMP(2)
        stage_sync();
        WSTAMP(3)
        par ^= 1;
        if (!is_last) {
            ++c;
        } else {
            epilogue(cur);
            WSTAMP(5)
            if (!has_more) break;
#if !(WINO_ABL & 2)
            zero_acc();
#endif
            cur = nxt;
            vb = vb_next;
            ximg = a_next;
            wblk = b_next;
            c = 0;
            transform_now(lds_a + par * kABytes, Va);   // the next tile's first halo landed before this chunk's first barrier
            WSTAMP(6)
        }
#if WINO_STAMPS
        if (blockIdx.x == 0 && tid == 0) wino_dbg[is_last ? 8 : 7] += 1;
#endif
    }
}

#undef WINO_LOAD_A_ROUND
#undef WINO_LOAD_B_ROUND
#undef WINO_GLDS16_S

// (Cout,Cin,3,3) -> [Cout/32][Cin/16][pos = 4*xi + nu][32 co][16 ch], U = G g G^T in float64, rounded once
__global__ void pack_wino_kernel(const float* __restrict__ w, float* __restrict__ out, int cout, int cin) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;   // one (co, ci) pair per thread
    if (idx >= (long)cout * cin) return;
    const int co = (int)(idx / cin), ci = (int)(idx % cin);
    double g[3][3];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b) g[a][b] = (double)w[((size_t)co * cin + ci) * 9 + a * 3 + b];
    // G = [[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]]
    double t[4][3];
#pragma unroll
    for (int b = 0; b < 3; ++b) {
        t[0][b] = g[0][b];
        t[1][b] = 0.5 * (g[0][b] + g[1][b] + g[2][b]);
        t[2][b] = 0.5 * (g[0][b] - g[1][b] + g[2][b]);
        t[3][b] = g[2][b];
    }
    const int nchunks = cin / 16;
    float* base = out + (((size_t)(co / 32) * nchunks + ci / 16) * 16 * 32 + (co % 32)) * 16 + (ci % 16);
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const double u[4] = {t[a][0], 0.5 * (t[a][0] + t[a][1] + t[a][2]), 0.5 * (t[a][0] - t[a][1] + t[a][2]), t[a][2]};
#pragma unroll
        for (int b = 0; b < 4; ++b) base[(size_t)(4 * a + b) * 32 * 16] = (float)u[b];
    }
}

// split-K reduction: y = act(sum_s partial[s] + bias), partials added in the fixed order s = 0, 1, ... (bit-identical for a
// pixel whatever the batch), then the optional MaxPool2d(2,2,ceil_mode=True) over the activated values
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ part, int ksplit, long part_stride,
                                                           const float* __restrict__ bias, float* __restrict__ y, int batch,
                                                           int H, int W, int C, int relu, int pool) {
    const int Ho = pool ? (H + 1) >> 1 : H, Wo = pool ? (W + 1) >> 1 : W;
    const int c4n = C >> 2;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long)batch * Ho * Wo * c4n) return;
    const int c4 = (int)(idx % c4n);
    long r = idx / c4n;
    const int xo = (int)(r % Wo);
    r /= Wo;
    const int yo = (int)(r % Ho);
    const int img = (int)(r / Ho);
    const float4 b4 = *reinterpret_cast<const float4*>(bias + 4 * c4);
    auto pixel = [&](int yy, int xx) {
        const float* p = part + (((size_t)img * H + yy) * W + xx) * C + 4 * c4;
        float4 a = *reinterpret_cast<const float4*>(p);
        for (int s2 = 1; s2 < ksplit; ++s2) {
            const float4 v = *reinterpret_cast<const float4*>(p + (size_t)s2 * part_stride);
            a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
        }
        a.x += b4.x; a.y += b4.y; a.z += b4.z; a.w += b4.w;
        if (relu) a = make_float4(fmaxf(a.x, 0.f), fmaxf(a.y, 0.f), fmaxf(a.z, 0.f), fmaxf(a.w, 0.f));
        return a;
    };
    float4 o;
    if (!pool) {
        o = pixel(yo, xo);
    } else {
        o = pixel(2 * yo, 2 * xo);
        auto mx = [&](float4 v) { o = make_float4(fmaxf(o.x, v.x), fmaxf(o.y, v.y), fmaxf(o.z, v.z), fmaxf(o.w, v.w)); };
        if (2 * xo + 1 < W) mx(pixel(2 * yo, 2 * xo + 1));
        if (2 * yo + 1 < H) {
            mx(pixel(2 * yo + 1, 2 * xo));
            if (2 * xo + 1 < W) mx(pixel(2 * yo + 1, 2 * xo + 1));
        }
    }
    *reinterpret_cast<float4*>(y + (((size_t)img * Ho + yo) * Wo + xo) * C + 4 * c4) = o;
}

}  // namespace holo

#if WINO_STAMPS
extern "C" void holo_wino_debug_read(unsigned long long* out) { (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(holo::wino_dbg), sizeof(unsigned long long) * 16); }
extern "C" void holo_wino_debug_reset(void) {
    unsigned long long z[16] = {0};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(holo::wino_dbg), z, sizeof(z));
}
#endif

extern "C" {

size_t holo_conv3x3_wino_packed_elems(int cout, int cin) {
    return cout > 0 && cin > 0 ? (size_t)16 * cout * cin : 0;
}

int holo_pack_conv3x3_wino_weight_f32(const float* w_oihw, float* w_packed, int cout, int cin, void* stream) {
    using namespace holo;
    HOLO_REQUIRE(w_oihw && w_packed, HOLO_ERR_ARG, "holo_pack_conv3x3_wino_weight_f32: null pointer");
    HOLO_REQUIRE(cin > 0 && cin % 16 == 0 && cout > 0 && cout % 32 == 0, HOLO_ERR_UNSUPPORTED,
                 "holo_pack_conv3x3_wino_weight_f32: Cin=%d must be a multiple of 16 and Cout=%d of 32", cin, cout);
    const long total = (long)cout * cin;
    hipLaunchKernelGGL(pack_wino_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, as_stream(stream), w_oihw,
                       w_packed, cout, cin);
    return check_launch("holo_pack_conv3x3_wino_weight_f32");
}

int holo_conv3x3_wino_f32(const float* x_nhwc, const float* w_packed, const float* bias, float* y_nhwc, int batch, int h,
                          int w, int cin, int cout, int relu, int pool, void* stream) {
    using namespace holo;
    HOLO_REQUIRE(x_nhwc && w_packed && bias && y_nhwc, HOLO_ERR_ARG, "holo_conv3x3_wino_f32: null pointer");
    HOLO_REQUIRE(batch > 0 && h >= 2 && w >= 2, HOLO_ERR_ARG,
                 "holo_conv3x3_wino_f32: batch=%d h=%d w=%d (reflection padding needs h,w >= 2)", batch, h, w);
    HOLO_REQUIRE(cin >= 32 && cin % 16 == 0 && cout > 0 && cout % 32 == 0, HOLO_ERR_UNSUPPORTED,
                 "holo_conv3x3_wino_f32: Cin=%d must be a multiple of 16 (at least 32) and Cout=%d a multiple of 32", cin, cout);
    HOLO_REQUIRE(!pool || relu, HOLO_ERR_UNSUPPORTED, "holo_conv3x3_wino_f32: fused pooling requires relu");
    HOLO_REQUIRE((long)h * w * (cin > cout ? cin : cout) < (1L << 30), HOLO_ERR_UNSUPPORTED,
                 "holo_conv3x3_wino_f32: one image must stay below 2^30 elements (32-bit byte offsets)");
    const int tiles_x = ceil_div(w, wino::kPW), tiles_y = ceil_div(h, wino::kPH);
    const long blocks = (long)batch * tiles_x * tiles_y * (cout / wino::kTN);
    HOLO_REQUIRE(blocks > 0 && blocks < 2147483647L, HOLO_ERR_UNSUPPORTED, "holo_conv3x3_wino_f32: grid of %ld blocks", blocks);
    static DeviceOnce attr_once;   // per device, published after the attribute is set
    attr_once.run([] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wino_f32_kernel),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, wino::kLdsBytes);
    });
    // persistent grid: one block per CU (512 registers per lane, 142 KB of LDS), tiles dealt round-robin.  Every block
    // runs to completion on its own (no inter-block waits), so a grid larger than the resident set would also be correct.
    const int cus = device_cus();
#if WINO_ABL & 8
    const long grid = blocks;
#else
    const long grid = blocks < cus ? blocks : cus;
#endif
    hipLaunchKernelGGL(conv_wino_f32_kernel, dim3((unsigned)grid), dim3(wino::kThreads), wino::kLdsBytes, as_stream(stream),
                       x_nhwc, w_packed, bias, y_nhwc, h, w, cin, cout, tiles_x, tiles_y, (int)blocks, relu, pool, 1, 0L);
    return check_launch("holo_conv3x3_wino_f32");
}

size_t holo_conv3x3_wino_splitk_workspace_bytes(int batch, int h, int w, int cout, int ksplit) {
    if (batch <= 0 || h <= 0 || w <= 0 || cout <= 0 || ksplit <= 0) return 0;
    return (size_t)ksplit * batch * h * w * cout * sizeof(float);
}

int holo_conv3x3_wino_splitk_f32(const float* x_nhwc, const float* w_packed, const float* bias, float* y_nhwc, int batch, int h,
                                 int w, int cin, int cout, int relu, int pool, int ksplit, float* workspace, size_t workspace_bytes,
                                 void* stream) {
    using namespace holo;
    HOLO_REQUIRE(x_nhwc && w_packed && bias && y_nhwc && workspace, HOLO_ERR_ARG, "holo_conv3x3_wino_splitk_f32: null pointer");
    HOLO_REQUIRE(batch > 0 && h >= 2 && w >= 2, HOLO_ERR_ARG,
                 "holo_conv3x3_wino_splitk_f32: batch=%d h=%d w=%d (reflection padding needs h,w >= 2)", batch, h, w);
    HOLO_REQUIRE(cin >= 32 && cin % 16 == 0 && cout > 0 && cout % 32 == 0, HOLO_ERR_UNSUPPORTED,
                 "holo_conv3x3_wino_splitk_f32: Cin=%d must be a multiple of 16 (at least 32) and Cout=%d a multiple of 32", cin, cout);
    HOLO_REQUIRE(ksplit >= 2 && ksplit <= 8 && (cin / 16) / ksplit >= 2, HOLO_ERR_UNSUPPORTED,
                 "holo_conv3x3_wino_splitk_f32: ksplit=%d must be in [2,8] and leave every part at least 2 of the %d channel chunks",
                 ksplit, cin / 16);
    HOLO_REQUIRE(!pool || relu, HOLO_ERR_UNSUPPORTED, "holo_conv3x3_wino_splitk_f32: fused pooling requires relu");
    HOLO_REQUIRE((long)h * w * (cin > cout ? cin : cout) < (1L << 30), HOLO_ERR_UNSUPPORTED,
                 "holo_conv3x3_wino_splitk_f32: one image must stay below 2^30 elements (32-bit byte offsets)");
    HOLO_REQUIRE(workspace_bytes >= holo_conv3x3_wino_splitk_workspace_bytes(batch, h, w, cout, ksplit), HOLO_ERR_WORKSPACE,
                 "holo_conv3x3_wino_splitk_f32: workspace %zu < required %zu", workspace_bytes,
                 holo_conv3x3_wino_splitk_workspace_bytes(batch, h, w, cout, ksplit));
    const int tiles_x = ceil_div(w, wino::kPW), tiles_y = ceil_div(h, wino::kPH);
    const long blocks = (long)batch * tiles_x * tiles_y * (cout / wino::kTN) * ksplit;
    HOLO_REQUIRE(blocks > 0 && blocks < 2147483647L, HOLO_ERR_UNSUPPORTED, "holo_conv3x3_wino_splitk_f32: grid of %ld blocks", blocks);
    static DeviceOnce attr_once;   // per device, published after the attribute is set
    attr_once.run([] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wino_f32_kernel),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, wino::kLdsBytes);
    });
    const int cus = device_cus();
    const long grid = blocks < cus ? blocks : cus;
    const long part = (long)batch * h * w * cout;
    hipLaunchKernelGGL(conv_wino_f32_kernel, dim3((unsigned)grid), dim3(wino::kThreads), wino::kLdsBytes, as_stream(stream),
                       x_nhwc, w_packed, static_cast<const float*>(nullptr), workspace, h, w, cin, cout, tiles_x, tiles_y, (int)blocks,
                       0, 0, ksplit, part);
    const int ho = pool ? (h + 1) / 2 : h, wo = pool ? (w + 1) / 2 : w;
    const long n4 = (long)batch * ho * wo * (cout / 4);
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, as_stream(stream), workspace, ksplit,
                       part, bias, y_nhwc, batch, h, w, cout, relu, pool);
    return check_launch("holo_conv3x3_wino_splitk_f32");
}

}  // extern "C"
