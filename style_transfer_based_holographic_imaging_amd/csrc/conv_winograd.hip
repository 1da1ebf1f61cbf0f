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
// accumulators), bias, ReLU, optional 2x2 max-pool (= exactly one Winograd tile), NHWC stores of 128-B runs.
#include "common.h"

namespace holo {

typedef float floatx16 __attribute__((ext_vector_type(16)));

namespace wino {
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
constexpr int kLdsBytes = 2 * kABytes + 2 * kBBytes;   // 145408
static_assert(kLdsBytes <= 160 * 1024, "LDS budget");

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

__global__ __launch_bounds__(wino::kThreads, 1) void conv_wino_f32_kernel(
    const float* __restrict__ x, const float* __restrict__ wp, const float* __restrict__ bias, float* __restrict__ y,
    int H, int W, int Cin, int Cout, int tiles_x, int tiles_y, int relu, int pool) {
    using namespace wino;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* lds_a = smem;                  // [2][kABytes]
    unsigned char* lds_b = smem + 2 * kABytes;    // [2][kBBytes]

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int n_blk = Cout / kTN;
    int bid = blockIdx.x;
    // XCD-aware order (blocks are dealt round-robin over the 8 XCDs): a contiguous range of blocks per XCD, so that the
    // channel blocks of one patch (same halo) and neighbouring patches meet in one L2.  Bijective when the grid is a
    // multiple of 8; speed only.
    if ((gridDim.x & 7) == 0) bid = (bid & 7) * (gridDim.x >> 3) + (bid >> 3);
    const int nb = bid % n_blk;
    bid /= n_blk;
    const int txb = bid % tiles_x;
    bid /= tiles_x;
    const int tyb = bid % tiles_y;
    const int img = bid / tiles_y;
    const int y0 = tyb * kPH, x0 = txb * kPW;
    const int n0 = nb * kTN;
    const float* __restrict__ ximg = x + (size_t)img * H * W * Cin;
    const int nchunks = Cin / 16;

    // ---- staging plans ------------------------------------------------------------------------------------------
    // LDS unit U = tid + 256 r of the halo image = unit u' of slot U>>2; it holds logical unit u' ^ ((hy>>1)&3) of that pixel
    unsigned a_src[kARounds];
#pragma unroll
    for (int r = 0; r < kARounds; ++r) {
        int U = tid + r * kThreads;
        U = U < kAUnits ? U : kAUnits - 1;        // the partial round is masked at issue; keep the address valid anyway
        const int s = U >> 2, up = U & 3;
        const int hy = s / kHW, rem = s % kHW;
        const int hx = rem < kHW / 2 ? 2 * rem : 2 * (rem - kHW / 2) + 1;
        const int u = up ^ ((hy >> 1) & 3);
        const int ry = reflect1(y0 - 1 + hy, H), rx = reflect1(x0 - 1 + hx, W);
        a_src[r] = (unsigned)((ry * W + rx) * Cin + 4 * u) * 4u;    // byte offset within the image (< 2^32: entry check)
    }
    // weight tile of a chunk: [pos][co][16 ch], contiguous 32 KB in the packed tensor; LDS unit U = row*4 + u' holds
    // logical unit u' ^ ((co>>2)&3) of row (pos, co)
    unsigned b_src[kBRounds];
#pragma unroll
    for (int r = 0; r < kBRounds; ++r) {
        const int U = tid + r * kThreads;
        const int row = U >> 2, up = U & 3;
        const int u = up ^ ((row >> 2) & 3);      // row = pos*32 + co: (row>>2)&3 == (co>>2)&3
        b_src[r] = (unsigned)(row * 64 + u * 16);
    }
    const int wave_base = __builtin_amdgcn_readfirstlane((tid & ~63) * 16);
    const unsigned lds_base = (unsigned)(unsigned long long)(const __attribute__((address_space(3))) void*)smem;
    const float* __restrict__ wblk = wp + (size_t)nb * nchunks * (kBBytes / 4);   // [nb][chunk][pos][co][16]

#define WINO_LOAD_A_ROUND(chunk_, buf_, r_)                                                                     \
    if ((r_) < kARounds - 1 || tid + (r_) * kThreads < kAUnits)                                                 \
        WINO_GLDS16_S(ximg + (chunk_) * 16, a_src[r_], (buf_) * kABytes + (r_) * kThreads * 16 + wave_base)
#define WINO_LOAD_B_ROUND(chunk_, buf_, r_)                                                                     \
    WINO_GLDS16_S(wblk + (size_t)(chunk_) * (kBBytes / 4), b_src[r_], 2 * kABytes + (buf_) * kBBytes + (r_) * kThreads * 16 + wave_base)

    // ---- per-lane operand coordinates ---------------------------------------------------------------------------
    const int hsel = lane >> 5, li = lane & 31;
    const int ltx = li & 7, lty = li >> 3;                 // tile within the wave's 4 x 8 tiles
    const int hy0 = 2 * (4 * wave + lty);                  // halo row of the patch's first row (halo origin = y0-1)
    // patch pixel (i, j): slot = (hy0+i)*18 + (j&1)*9 + ltx + (j>>1); unit key = ((hy0+i)>>1)&3 = (lty + (i>>1)) & 3
    const int a_base = (hy0 * kHW + ltx) * 64;
    const int key0 = lty & 3, key1 = (lty + 1) & 3;
    const int b_base = li * 64;                            // row (pos, co = li): + pos*2048; unit key (li>>2)&3
    const int bkey = (li >> 2) & 3;

    floatx16 acc[16];
#pragma unroll
    for (int p = 0; p < 16; ++p)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[p][r] = 0.0f;

    // B^T d B of this lane's tile for the channel quad (2*jj + hsel) of the chunk in halo buffer `abuf_`
    auto transform_in = [&](const unsigned char* abuf, int jj, float4 (&V)[16]) {
        const int u = 2 * jj + hsel;
        float4 t[4][4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float4 d[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int off = a_base + (i * kHW + (j & 1) * (kHW / 2) + (j >> 1)) * 64 + ((u ^ (i < 2 ? key0 : key1)) << 4);
                d[i] = *reinterpret_cast<const float4*>(abuf + off);
            }
            // rows of B^T: [1,0,-1,0], [0,1,1,0], [0,-1,1,0], [0,1,0,-1]
            t[0][j] = make_float4(d[0].x - d[2].x, d[0].y - d[2].y, d[0].z - d[2].z, d[0].w - d[2].w);
            t[1][j] = make_float4(d[1].x + d[2].x, d[1].y + d[2].y, d[1].z + d[2].z, d[1].w + d[2].w);
            t[2][j] = make_float4(d[2].x - d[1].x, d[2].y - d[1].y, d[2].z - d[1].z, d[2].w - d[1].w);
            t[3][j] = make_float4(d[1].x - d[3].x, d[1].y - d[3].y, d[1].z - d[3].z, d[1].w - d[3].w);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float4 a = t[i][0], b = t[i][1], c = t[i][2], e = t[i][3];
            V[4 * i + 0] = make_float4(a.x - c.x, a.y - c.y, a.z - c.z, a.w - c.w);
            V[4 * i + 1] = make_float4(b.x + c.x, b.y + c.y, b.z + c.z, b.w + c.w);
            V[4 * i + 2] = make_float4(c.x - b.x, c.y - b.y, c.z - b.z, c.w - b.w);
            V[4 * i + 3] = make_float4(b.x - e.x, b.y - e.y, b.z - e.z, b.w - e.w);
        }
    };
    // 64 MFMAs of one half-chunk: acc[pos] += V[pos] (4 channels) x weights(pos, same 4 channels)
    auto mfma_pos = [&](const unsigned char* bbuf, int jj, int p, const float4 (&V)[16]) {
        const float4 bv = *reinterpret_cast<const float4*>(bbuf + p * (kTN * 64) + b_base + (((2 * jj + hsel) ^ bkey) << 4));
        acc[p] = __builtin_amdgcn_mfma_f32_32x32x2f32(V[p].x, bv.x, acc[p], 0, 0, 0);
        acc[p] = __builtin_amdgcn_mfma_f32_32x32x2f32(V[p].y, bv.y, acc[p], 0, 0, 0);
        acc[p] = __builtin_amdgcn_mfma_f32_32x32x2f32(V[p].z, bv.z, acc[p], 0, 0, 0);
        acc[p] = __builtin_amdgcn_mfma_f32_32x32x2f32(V[p].w, bv.w, acc[p], 0, 0, 0);
    };
    auto stage_sync = [&]() {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's DMAs have landed
        __syncthreads();                                     // everybody's have; everybody is done reading the old buffers
    };

    // ---- prologue -----------------------------------------------------------------------------------------------
#pragma unroll
    for (int r = 0; r < kARounds; ++r) { WINO_LOAD_A_ROUND(0, 0, r) }
#pragma unroll
    for (int r = 0; r < kBRounds; ++r) { WINO_LOAD_B_ROUND(0, 0, r) }
    stage_sync();
    float4 V0[16], V1[16];
    transform_in(lds_a, 0, V0);

    // ---- main loop: one chunk = two steps (channel quads 0-1, then 2-3, of each lane half) -----------------------
    int par = 0;
    for (int chunk = 0; chunk < nchunks; ++chunk) {
        const bool more = chunk + 1 < nchunks;
        const unsigned char* abuf = lds_a + par * kABytes;
        const unsigned char* bbuf = lds_b + par * kBBytes;
        // step 0: MFMAs on V0; meanwhile V1 = second half of this chunk; the next chunk's halo is requested
        // (its buffer was released by the barrier that ended step 0 of the previous chunk)
        transform_in(abuf, 1, V1);
#pragma unroll
        for (int p = 0; p < 16; ++p) {
            mfma_pos(bbuf, 0, p, V0);
            if (more && p < kARounds) { WINO_LOAD_A_ROUND(chunk + 1, par ^ 1, p) }
        }
        stage_sync();
        // step 1: MFMAs on V1; meanwhile V0 = first half of the next chunk (its halo has just landed); the next
        // chunk's weights are requested (their buffer was released by the barrier that ended the previous chunk)
        if (more) transform_in(lds_a + (par ^ 1) * kABytes, 0, V0);
#pragma unroll
        for (int p = 0; p < 16; ++p) {
            mfma_pos(bbuf, 1, p, V1);
            if (more && p < kBRounds) { WINO_LOAD_B_ROUND(chunk + 1, par ^ 1, p) }
        }
        stage_sync();
        par ^= 1;
    }

    // ---- epilogue: Y = A^T M A, bias, ReLU, optional pool, NHWC store ---------------------------------------------
    // lane: channel n0 + li; register r: tile m = (r&3) + 8*(r>>2) + 4*hsel of the wave's 32 (tile row m>>3, column m&7)
    const int co = n0 + li;
    const float bv = bias[co];
    const int Ho = (H + 1) >> 1, Wo = (W + 1) >> 1;
    float* __restrict__ yimg = y + (size_t)img * (pool ? (size_t)Ho * Wo : (size_t)H * W) * Cout;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int m = (r & 3) + 8 * (r >> 2) + 4 * hsel;
        const int Y = y0 + 2 * (4 * wave + (m >> 3)), X = x0 + 2 * (m & 7);
        // A^T = [[1,1,1,0],[0,1,-1,-1]]: rows first
        float s[2][4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            s[0][j] = acc[0 + j][r] + acc[4 + j][r] + acc[8 + j][r];
            s[1][j] = acc[4 + j][r] - acc[8 + j][r] - acc[12 + j][r];
        }
        float o[2][2];
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            o[a][0] = s[a][0] + s[a][1] + s[a][2] + bv;
            o[a][1] = s[a][1] - s[a][2] - s[a][3] + bv;
            if (relu) {
                o[a][0] = fmaxf(o[a][0], 0.0f);
                o[a][1] = fmaxf(o[a][1], 0.0f);
            }
        }
        if (!pool) {
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b)
                    if (Y + a < H && X + b < W) yimg[((size_t)(Y + a) * W + (X + b)) * Cout + co] = o[a][b];
        } else if (Y < H && X < W) {
            // MaxPool2d(2,2,ceil_mode=True): the window is this tile; members beyond the image edge are left out
            float mx = o[0][0];
            if (X + 1 < W) mx = fmaxf(mx, o[0][1]);
            if (Y + 1 < H) {
                mx = fmaxf(mx, o[1][0]);
                if (X + 1 < W) mx = fmaxf(mx, o[1][1]);
            }
            yimg[((size_t)(Y >> 1) * Wo + (X >> 1)) * Cout + co] = mx;
        }
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

}  // namespace holo

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
    HOLO_REQUIRE(cin > 0 && cin % 16 == 0 && cout > 0 && cout % 32 == 0, HOLO_ERR_UNSUPPORTED,
                 "holo_conv3x3_wino_f32: Cin=%d must be a multiple of 16 and Cout=%d of 32", cin, cout);
    HOLO_REQUIRE(!pool || relu, HOLO_ERR_UNSUPPORTED, "holo_conv3x3_wino_f32: fused pooling requires relu");
    HOLO_REQUIRE((long)h * w * (cin > cout ? cin : cout) < (1L << 30), HOLO_ERR_UNSUPPORTED,
                 "holo_conv3x3_wino_f32: one image must stay below 2^30 elements (32-bit byte offsets)");
    const int tiles_x = ceil_div(w, wino::kPW), tiles_y = ceil_div(h, wino::kPH);
    const long blocks = (long)batch * tiles_x * tiles_y * (cout / wino::kTN);
    HOLO_REQUIRE(blocks > 0 && blocks < 2147483647L, HOLO_ERR_UNSUPPORTED, "holo_conv3x3_wino_f32: grid of %ld blocks", blocks);
    static unsigned long long attr_done = 0;   // per device
    if (first_call_on_device(attr_done))
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wino_f32_kernel),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, wino::kLdsBytes);
    hipLaunchKernelGGL(conv_wino_f32_kernel, dim3((unsigned)blocks), dim3(wino::kThreads), wino::kLdsBytes, as_stream(stream),
                       x_nhwc, w_packed, bias, y_nhwc, h, w, cin, cout, tiles_x, tiles_y, relu, pool);
    return check_launch("holo_conv3x3_wino_f32");
}

}  // extern "C"
