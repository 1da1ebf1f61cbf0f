// 2-D phase unwrapping on the device (the step after the hot path: utils/functions.py:44-59, called with
// unwrap=True from test_field_retrieval_mnist.py:126 through utils/Forward_model.py:30-31).
//
// The reference loops over images on the host and calls scikit-image's unwrap_phase, i.e. Herraez et al.
// (Appl. Opt. 41(35), 2002): pixels get a reliability (squared wrapped second differences), every neighbour
// pair is an edge weighted by the two reliabilities, and edges are visited in ascending weight, joining pixel
// groups and shifting the smaller group by the integer jump of the edge.  That visit order is Kruskal's
// algorithm, so the joining edges are the minimum spanning tree of the pixel grid and the relative offset of
// two pixels is the sum of jumps along the tree path.  A sort + sequential union is the wrong shape for a GPU;
// the same tree is built here with Boruvka rounds (every group picks its lightest outgoing edge at once):
//
//   unwrap_reliability_kernel   rel(x) in fp64 from the fp32 phase                         (thread / pixel)
//   unwrap_weights_kernel       fp32 weights of the right and down edge of every pixel     (thread / pixel)
//   unwrap_boruvka_kernel       one persistent 1024-thread block per image:
//        node[x] = (group root, offset of x relative to that root) packed in one 64-bit word, so that any
//        single load sees a consistent pair and in-place pointer jumping needs no double buffering;
//        round:  edge sweep   — 64-bit atomicMin of (round tag | weight bits | edge id) into both groups' slots
//                hook sweep   — each root follows its chosen edge; a mutual pair keeps the lower root
//                jump sweeps  — node[x] <- (parent(parent), off + off(parent)) until flat
//        the round tag (31 - round) in the top bits makes newer keys smaller than any stale one: no resets;
//        afterwards offsets are re-based so the most frequent one is 0 and value + 2 pi k is stored.
//
// Total order of edges = (float32(rel1 + rel2), edge id), edge id = 2*pixel + (0 right | 1 down); border
// pixels take the constant 9999999 (scikit-image adds rand() there).  oracle/unwrap_oracle.py restates the
// sequential algorithm with the same order; results agree bit for bit up to the global multiple of 2 pi.
#include "common.h"

namespace holo {

typedef unsigned long long u64;

constexpr double kPi = 3.141592653589793;
constexpr double kTwoPi = 6.283185307179586;
constexpr double kBorderReliability = 9999999.0;
constexpr int kUnwrapThreads = 1024;
constexpr int kEidBits = 28;
constexpr int kTagShift = 59;
constexpr int kHistBins = 1024;
constexpr int kMaxRounds = 30;
constexpr int kMaxJumps = 64;

__device__ __forceinline__ u64 ld64(const u64* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st64(u64* p, u64 v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ u64 pack_node(unsigned parent, int off) { return ((u64)parent << 32) | (unsigned)off; }
__device__ __forceinline__ unsigned node_parent(u64 n) { return (unsigned)(n >> 32); }
__device__ __forceinline__ int node_off(u64 n) { return (int)(unsigned)n; }

__device__ __forceinline__ double wrap_pi(double d) { return d > kPi ? d - kTwoPi : (d < -kPi ? d + kTwoPi : d); }

// individually rounded products and sums, in the oracle's order (no fused multiply-add)
__device__ __forceinline__ double second_diff_sq(double a, double c, double b) {
    const double s = __dsub_rn(wrap_pi(__dsub_rn(a, c)), wrap_pi(__dsub_rn(c, b)));
    return __dmul_rn(s, s);
}

__global__ __launch_bounds__(256) void unwrap_reliability_kernel(const float* __restrict__ phase, long in_stride,
                                                                 double* __restrict__ rel, int H, int W) {
    const int n = H * W;
    const int x = blockIdx.x * 256 + threadIdx.x;
    if (x >= n) return;
    const int img = blockIdx.y;
    const float* __restrict__ p = phase + (size_t)img * in_stride;
    const int yy = x / W, xx = x % W;
    double r = kBorderReliability;
    if (yy > 0 && yy < H - 1 && xx > 0 && xx < W - 1) {
        const double c = p[x];
        const double hh = second_diff_sq(p[x - 1], c, p[x + 1]);
        const double vv = second_diff_sq(p[x - W], c, p[x + W]);
        const double d1 = second_diff_sq(p[x - W - 1], c, p[x + W + 1]);
        const double d2 = second_diff_sq(p[x - W + 1], c, p[x + W - 1]);
        r = __dadd_rn(__dadd_rn(__dadd_rn(hh, vv), d1), d2);
    }
    rel[(size_t)img * n + x] = r;
}

__global__ __launch_bounds__(256) void unwrap_weights_kernel(const double* __restrict__ rel, float* __restrict__ wgt,
                                                             int H, int W) {
    const int n = H * W;
    const int x = blockIdx.x * 256 + threadIdx.x;
    if (x >= n) return;
    const int img = blockIdx.y;
    const double* __restrict__ r = rel + (size_t)img * n;
    const int yy = x / W, xx = x % W;
    const double rx = r[x];
    float2 w = make_float2(0.0f, 0.0f);
    if (xx + 1 < W) w.x = (float)__dadd_rn(rx, r[x + 1]);
    if (yy + 1 < H) w.y = (float)__dadd_rn(rx, r[x + W]);
    reinterpret_cast<float2*>(wgt)[(size_t)img * n + x] = w;
}

__device__ __forceinline__ void offer_edge(u64* best, unsigned root, u64 key) {
    if (key < ld64(best + root)) atomicMin(best + root, key);
}

__global__ __launch_bounds__(kUnwrapThreads) void unwrap_boruvka_kernel(const float* __restrict__ phase, long in_stride,
                                                                        float* __restrict__ out, long out_stride,
                                                                        u64* __restrict__ node_all,
                                                                        u64* __restrict__ best_all,
                                                                        const float* __restrict__ wgt_all, int H, int W) {
    __shared__ int base_offset;
    __shared__ int hist[kHistBins];
    const int tid = threadIdx.x;
    const int n = H * W;
    const int img = blockIdx.x;
    u64* __restrict__ node = node_all + (size_t)img * n;
    u64* __restrict__ best = best_all + (size_t)img * n;
    const float2* __restrict__ wgt = reinterpret_cast<const float2*>(wgt_all) + (size_t)img * n;
    const float* __restrict__ ph = phase + (size_t)img * in_stride;

    for (int x = tid; x < n; x += kUnwrapThreads) {
        st64(node + x, pack_node((unsigned)x, 0));
        st64(best + x, ~0ull);
    }
    __syncthreads();

    // a grid of n pixels needs at most ceil(log2 n) <= 24 rounds; both loops are capped so the block always exits
    for (int round = 0; round < kMaxRounds; ++round) {
        const u64 tag = (u64)(kMaxRounds - round) << kTagShift;  // 1..30: never the 31 of an untouched slot
        // ---- edge sweep: every group learns its lightest outgoing edge
        int any = 0;
        for (int x = tid; x < n; x += kUnwrapThreads) {
            const int yy = x / W, xx = x - yy * W;
            const unsigned rx = node_parent(ld64(node + x));
            const float2 w = wgt[x];
            if (xx + 1 < W) {
                const unsigned ry = node_parent(ld64(node + x + 1));
                if (ry != rx) {
                    const u64 key = tag | ((u64)__float_as_uint(w.x) << kEidBits) | (u64)(2 * x);
                    offer_edge(best, rx, key);
                    offer_edge(best, ry, key);
                    any = 1;
                }
            }
            if (yy + 1 < H) {
                const unsigned ry = node_parent(ld64(node + x + W));
                if (ry != rx) {
                    const u64 key = tag | ((u64)__float_as_uint(w.y) << kEidBits) | (u64)(2 * x + 1);
                    offer_edge(best, rx, key);
                    offer_edge(best, ry, key);
                    any = 1;
                }
            }
        }
        if (!__syncthreads_or(any)) break;  // one group left
        // ---- hook sweep: roots follow their edge (writes touch root entries only)
        for (int x = tid; x < n; x += kUnwrapThreads) {
            const u64 nx = ld64(node + x);
            if (node_parent(nx) != (unsigned)x) continue;
            const u64 b = ld64(best + x);
            if ((b >> kTagShift) != (tag >> kTagShift)) continue;
            const unsigned eid = (unsigned)(b & ((1ull << kEidBits) - 1));
            const int u = (int)(eid >> 1);
            const int v = u + ((eid & 1) ? W : 1);
            const u64 nu = ld64(node + u), nv = ld64(node + v);
            const bool mine_is_u = node_parent(nu) == (unsigned)x;
            const u64 na = mine_is_u ? nu : nv;   // endpoint inside this group
            const u64 nb = mine_is_u ? nv : nu;   // endpoint in the other group
            const unsigned other = node_parent(nb);
            if (other == (unsigned)x) continue;   // the partner of a mutual pair has already hooked onto this root
            if (ld64(best + other) == b && (unsigned)x < other) continue;  // mutual pair: the lower root stays
            // offsets k with unwrapped = value + 2 pi k:  k[a] - k[b] = +1 if value[b] - value[a] > pi, -1 if < -pi
            const double va = ph[mine_is_u ? u : v], vb = ph[mine_is_u ? v : u];
            const double d = vb - va;
            const int jump = d > kPi ? 1 : (d < -kPi ? -1 : 0);
            st64(node + x, pack_node(other, node_off(nb) - node_off(na) + jump));
        }
        __syncthreads();
        // ---- pointer jumping with offsets until every pixel points at its root
        for (int it = 0; it < kMaxJumps; ++it) {
            int changed = 0;
            for (int x = tid; x < n; x += kUnwrapThreads) {
                const u64 nx = ld64(node + x);
                const unsigned p = node_parent(nx);
                if (p == (unsigned)x) continue;
                const u64 np = ld64(node + p);
                const unsigned pp = node_parent(np);
                if (pp == p) continue;
                st64(node + x, pack_node(pp, node_off(nx) + node_off(np)));
                changed = 1;
            }
            if (!__syncthreads_or(changed)) break;
        }
    }

    // ---- re-base: the most frequent offset becomes 0 (lowest offset wins a tie)
    for (int i = tid; i < kHistBins; i += kUnwrapThreads) hist[i] = 0;
    __syncthreads();
    for (int x = tid; x < n; x += kUnwrapThreads) {
        const int k = node_off(ld64(node + x)) + kHistBins / 2;
        if (k >= 0 && k < kHistBins) atomicAdd(&hist[k], 1);
    }
    __syncthreads();
    if (tid == 0) {
        int arg = kHistBins / 2, top = -1;
        for (int i = 0; i < kHistBins; ++i)
            if (hist[i] > top) {
                top = hist[i];
                arg = i;
            }
        base_offset = arg - kHistBins / 2;
    }
    __syncthreads();
    const int base = base_offset;
    float* __restrict__ o = out + (size_t)img * out_stride;
    for (int x = tid; x < n; x += kUnwrapThreads) {
        const int k = node_off(ld64(node + x)) - base;
        o[x] = (float)((double)ph[x] + kTwoPi * (double)k);
    }
}

}  // namespace holo

extern "C" {

size_t holo_unwrap2d_workspace_bytes(int batch, int h, int w) {
    if (batch <= 0 || h <= 0 || w <= 0) return 0;
    return (size_t)batch * h * w * 24;  // node + edge slots (reliabilities first) + two fp32 weights per pixel
}

int holo_unwrap2d_f32(const float* phase, long in_stride, float* out, long out_stride, int batch, int h, int w,
                      void* workspace, size_t workspace_bytes, void* stream) {
    using namespace holo;
    HOLO_REQUIRE(phase && out && workspace, HOLO_ERR_ARG, "holo_unwrap2d: null pointer");
    HOLO_REQUIRE(batch > 0 && batch <= 65535 && h > 0 && w > 0, HOLO_ERR_ARG, "holo_unwrap2d: batch=%d h=%d w=%d", batch, h,
                 w);
    HOLO_REQUIRE((long)h * w <= (1L << 24), HOLO_ERR_UNSUPPORTED, "holo_unwrap2d: %d x %d exceeds 2^24 pixels", h, w);
    const long n = (long)h * w;
    HOLO_REQUIRE(in_stride >= n && out_stride >= n, HOLO_ERR_ARG, "holo_unwrap2d: strides %ld/%ld shorter than an image (%ld)",
                 in_stride, out_stride, n);
    HOLO_REQUIRE((reinterpret_cast<uintptr_t>(workspace) & 7) == 0, HOLO_ERR_ARG, "holo_unwrap2d: workspace not 8-byte aligned");
    HOLO_REQUIRE(workspace_bytes >= holo_unwrap2d_workspace_bytes(batch, h, w), HOLO_ERR_WORKSPACE,
                 "holo_unwrap2d: workspace %zu < %zu bytes", workspace_bytes, holo_unwrap2d_workspace_bytes(batch, h, w));
    u64* node = static_cast<u64*>(workspace);
    u64* best = node + (size_t)batch * n;
    float* wgt = reinterpret_cast<float*>(best + (size_t)batch * n);
    double* rel = reinterpret_cast<double*>(best);  // reliabilities live in the edge slots until the weights exist
    const dim3 grid((unsigned)((n + 255) / 256), (unsigned)batch);
    hipStream_t s = as_stream(stream);
    hipLaunchKernelGGL(unwrap_reliability_kernel, grid, dim3(256), 0, s, phase, in_stride, rel, h, w);
    hipLaunchKernelGGL(unwrap_weights_kernel, grid, dim3(256), 0, s, rel, wgt, h, w);
    hipLaunchKernelGGL(unwrap_boruvka_kernel, dim3((unsigned)batch), dim3(kUnwrapThreads), 0, s, phase, in_stride, out,
                       out_stride, node, best, wgt, h, w);
    return check_launch("holo_unwrap2d");
}

}  // extern "C"
