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
//        round:  edge sweep   — every pixel takes the lightest of its edges that leave its group; a wave folds
//                               runs of lanes with the same root (segmented min-scan) and the last lane of a run
//                               issues one 64-bit atomic min of (round tag | weight bits | edge id) on the
//                               group's slot; round 0 (every pixel alone) needs no atomics at all
//                hook sweep   — each root follows its chosen edge; a mutual pair keeps the lower root
//                jump sweeps  — node[x] <- (parent(parent), off + off(parent)) until flat
//        the round tag (30 - round) in the top bits makes newer keys smaller than any stale one: no resets;
//        afterwards offsets are re-based so the most frequent one is 0 and value + 2 pi k is stored.
//        The sweeps are bound by one CU's L2 bandwidth/latency: every stage issues all its loads for several
//        pixels unconditionally (clamped indices) before the first use.
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

// One image is owned by one work-group, so every shared access is a relaxed atomic at WORK-GROUP scope (agent scope
// would force each load past the XCD's L2 to stay coherent with the other seven, for nothing).
__device__ __forceinline__ u64 ld64(const u64* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void st64(u64* p, u64 v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
#ifndef HOLO_UNWRAP_SLOT_SCOPE
#define HOLO_UNWRAP_SLOT_SCOPE __HIP_MEMORY_SCOPE_WORKGROUP
#endif
__device__ __forceinline__ u64 ld_slot(const u64* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, HOLO_UNWRAP_SLOT_SCOPE); }
__device__ __forceinline__ void min64(u64* p, u64 v) {
    __hip_atomic_fetch_min(p, v, __ATOMIC_RELAXED, HOLO_UNWRAP_SLOT_SCOPE);
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

// The sweeps below are bound by L2 latency (every load is a dependent, coherent access), so each thread keeps
// several independent pixels in flight: all loads of a stage are issued before the first result is used.
constexpr int kBatchEdge = 4, kBatchHook = 4, kBatchJump = 8;  // sized to stay inside 128 VGPRs (1024 threads)

__device__ __forceinline__ u64 edge_key(u64 tag, float weight, unsigned eid) {
    return tag | ((u64)__float_as_uint(weight) << kEidBits) | (u64)eid;
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
    constexpr u64 kNoKey = ~0ull;

    // a grid of n pixels needs at most ceil(log2 n) <= 24 rounds; both loops are capped so the block always exits
#ifdef HOLO_DEBUG   // profiling build only (scripts/unwrap_stats.py): rounds, sweeps and clock ticks per phase
    int n_rounds = 0, n_jumps = 0;
    long long t_edge = 0, t_hook = 0, t_jump = 0, t0 = wall_clock64(), t1;
#define HOLO_UNWRAP_STAMP(acc_) do { t1 = wall_clock64(); acc_ += t1 - t0; t0 = t1; } while (0)
#define HOLO_UNWRAP_COUNT(var_) ++var_
#else
#define HOLO_UNWRAP_STAMP(acc_) do { } while (0)
#define HOLO_UNWRAP_COUNT(var_) do { } while (0)
#endif
    for (int round = 0; round < kMaxRounds; ++round) {
        const u64 tag = (u64)(kMaxRounds - round) << kTagShift;  // 1..30: never the 31 of an untouched slot
        int any = 0;
        if (round == 0) {
            // ---- every pixel is its own group: its lightest incident edge needs no atomics
            for (int base = tid; base < n; base += kUnwrapThreads * kBatchEdge) {
                float2 w[kBatchEdge];
                float wl[kBatchEdge], wu[kBatchEdge];
#pragma unroll
                for (int i = 0; i < kBatchEdge; ++i) {
                    // loads are unconditional on clamped indices: a predicated load becomes a branch + wait
                    const int x = min(base + i * kUnwrapThreads, n - 1);
                    w[i] = wgt[x];
                    wl[i] = wgt[max(x - 1, 0)].x;
                    wu[i] = wgt[max(x - W, 0)].y;
                }
#pragma unroll
                for (int i = 0; i < kBatchEdge; ++i) {
                    const int x = base + i * kUnwrapThreads;
                    if (x >= n) continue;
                    const int xx = x % W;
                    u64 k = kNoKey;
                    if (xx + 1 < W) k = min(k, edge_key(tag, w[i].x, 2u * x));
                    if (x + W < n) k = min(k, edge_key(tag, w[i].y, 2u * x + 1));
                    if (xx > 0) k = min(k, edge_key(tag, wl[i], 2u * (x - 1)));
                    if (x >= W) k = min(k, edge_key(tag, wu[i], 2u * (x - W) + 1));
                    st64(node + x, pack_node((unsigned)x, 0));
                    st64(best + x, k);
                    any |= k != kNoKey;
                }
            }
        } else {
            // ---- edge sweep: every pixel offers the lightest of its (up to four) edges that leave its group.  Lanes
            // are consecutive pixels of a row and groups are contiguous blobs, so a wave first folds runs of lanes
            // with the same root (segmented min-scan) and only the last lane of a run touches the group's slot.
            const int lane = tid & 63;
            for (int b0 = 0; b0 < n; b0 += kUnwrapThreads * kBatchEdge) {  // wave-uniform trip count (shuffles below)
                const int base = b0 + tid;
                unsigned rx[kBatchEdge];
                u64 key[kBatchEdge], nx[kBatchEdge], nr[kBatchEdge], nd[kBatchEdge], nl[kBatchEdge], nu[kBatchEdge];
                float2 w[kBatchEdge];
                float wl[kBatchEdge], wu[kBatchEdge];
                // unconditional loads on clamped indices, all issued before the first use (a missing neighbour reads
                // the pixel itself: same root, no edge)
#pragma unroll
                for (int i = 0; i < kBatchEdge; ++i) {
                    const int xc = min(base + i * kUnwrapThreads, n - 1), xx = xc % W;
                    const int il = xx > 0 ? xc - 1 : xc, iu = xc >= W ? xc - W : xc;
                    nx[i] = ld64(node + xc);
                    nr[i] = ld64(node + (xx + 1 < W ? xc + 1 : xc));
                    nd[i] = ld64(node + (xc + W < n ? xc + W : xc));
                    nl[i] = ld64(node + il);
                    nu[i] = ld64(node + iu);
                    w[i] = wgt[xc];
                    wl[i] = wgt[il].x;
                    wu[i] = wgt[iu].y;
                }
#pragma unroll
                for (int i = 0; i < kBatchEdge; ++i) {
                    const int x = base + i * kUnwrapThreads;
                    const bool ok = x < n;
                    const int xc = min(x, n - 1);
                    const unsigned r = node_parent(nx[i]);
                    u64 k = kNoKey;
                    if (node_parent(nr[i]) != r) k = min(k, edge_key(tag, w[i].x, 2u * xc));
                    if (node_parent(nd[i]) != r) k = min(k, edge_key(tag, w[i].y, 2u * xc + 1));
                    if (node_parent(nl[i]) != r) k = min(k, edge_key(tag, wl[i], 2u * (xc - 1)));
                    if (node_parent(nu[i]) != r) k = min(k, edge_key(tag, wu[i], 2u * (xc - W) + 1));
                    rx[i] = ok ? r : 0xffffffffu - (unsigned)lane;  // padding lanes: runs of their own, no key
                    key[i] = ok ? k : kNoKey;
                }
                u64 cur[kBatchEdge];
                bool tail[kBatchEdge];
#pragma unroll
                for (int i = 0; i < kBatchEdge; ++i) {
                    u64 k = key[i];
                    const unsigned r = rx[i];
                    // shuffles are executed by every lane (no short-circuit: a skipped lane would read as zero)
                    const unsigned r_below = __shfl_up(r, 1), r_above = __shfl_down(r, 1);
                    int sealed = (lane == 0) | (r_below != r);  // a run starts here: nothing further down joins
#pragma unroll
                    for (int d = 1; d < 64; d <<= 1) {
                        const u64 k_dn = __shfl_up(k, d);
                        const int s_dn = __shfl_up(sealed, d);
                        if (lane >= d && !sealed) {
                            k = min(k, k_dn);
                            sealed = s_dn;
                        }
                    }
                    tail[i] = ((lane == 63) | (r_above != r)) & (k != kNoKey);
                    key[i] = k;
                    // current slot contents: an offer that cannot win skips the atomic
                    cur[i] = ld_slot(best + (tail[i] ? r : 0u));
                    any |= tail[i];
                }
#pragma unroll
                for (int i = 0; i < kBatchEdge; ++i)
                    if (tail[i] && key[i] < cur[i]) min64(best + rx[i], key[i]);
            }
        }
        if (!__syncthreads_or(any)) break;  // one group left
        HOLO_UNWRAP_COUNT(n_rounds);
        HOLO_UNWRAP_STAMP(t_edge);
        // ---- hook sweep: roots follow their edge (writes touch root entries only)
        for (int base = tid; base < n; base += kUnwrapThreads * kBatchHook) {
            u64 b[kBatchHook], nu[kBatchHook], nv[kBatchHook];
            int eu[kBatchHook], ev[kBatchHook];
            bool live[kBatchHook];
#pragma unroll
            for (int i = 0; i < kBatchHook; ++i) {
                const int x = base + i * kUnwrapThreads;
                const int xc = min(x, n - 1);
                const u64 nx = ld64(node + xc);
                b[i] = ld_slot(best + xc);
                live[i] = x < n && node_parent(nx) == (unsigned)x;
            }
#pragma unroll
            for (int i = 0; i < kBatchHook; ++i) {
                live[i] = live[i] && (b[i] >> kTagShift) == (tag >> kTagShift);
                const unsigned eid = (unsigned)(b[i] & ((1ull << kEidBits) - 1));
                eu[i] = live[i] ? (int)(eid >> 1) : 0;  // dead lanes read pixel 0 and discard it
                ev[i] = live[i] ? eu[i] + ((eid & 1) ? W : 1) : 0;
                nu[i] = ld64(node + eu[i]);
                nv[i] = ld64(node + ev[i]);
            }
            u64 bo[kBatchHook];
            float pu[kBatchHook], pv[kBatchHook];
#pragma unroll
            for (int i = 0; i < kBatchHook; ++i) {
                const unsigned x = base + i * kUnwrapThreads;
                const bool mine_is_u = node_parent(nu[i]) == x;
                const unsigned other = node_parent(mine_is_u ? nv[i] : nu[i]);
                live[i] = live[i] && other != x;  // else: the partner of a mutual pair has already hooked onto this root
                bo[i] = ld_slot(best + (live[i] ? other : 0u));
                pu[i] = ph[eu[i]];
                pv[i] = ph[ev[i]];
            }
#pragma unroll
            for (int i = 0; i < kBatchHook; ++i) {
                const unsigned x = base + i * kUnwrapThreads;
                if (!live[i]) continue;
                const bool mine_is_u = node_parent(nu[i]) == x;
                const u64 na = mine_is_u ? nu[i] : nv[i];  // endpoint inside this group
                const u64 nb = mine_is_u ? nv[i] : nu[i];  // endpoint in the other group
                const unsigned other = node_parent(nb);
                if (bo[i] == b[i] && x < other) continue;  // mutual pair: the lower root stays
                // offsets k with unwrapped = value + 2 pi k:  k[a] - k[b] = +1 if value[b] - value[a] > pi, -1 if < -pi
                const double d = mine_is_u ? (double)pv[i] - (double)pu[i] : (double)pu[i] - (double)pv[i];
                const int jump = d > kPi ? 1 : (d < -kPi ? -1 : 0);
                st64(node + x, pack_node(other, node_off(nb) - node_off(na) + jump));
            }
        }
        __syncthreads();
        HOLO_UNWRAP_STAMP(t_hook);
        // ---- pointer jumping with offsets until every pixel points at its root
        for (int it = 0; it < kMaxJumps; ++it) {
            int changed = 0;
            for (int base = tid; base < n; base += kUnwrapThreads * kBatchJump) {
                u64 nx[kBatchJump], np[kBatchJump];
#pragma unroll
                for (int i = 0; i < kBatchJump; ++i) {
                    const int x = base + i * kUnwrapThreads;
                    nx[i] = ld64(node + min(x, n - 1));
                    if (x >= n) nx[i] = pack_node((unsigned)x, 0);
                }
#pragma unroll
                for (int i = 0; i < kBatchJump; ++i) {
                    const unsigned x = base + i * kUnwrapThreads;
                    const unsigned p = node_parent(nx[i]);
                    np[i] = ld64(node + (x < (unsigned)n ? p : 0u));
                    if (p == x) np[i] = nx[i];
                }
#pragma unroll
                for (int i = 0; i < kBatchJump; ++i) {
                    const unsigned p = node_parent(nx[i]), pp = node_parent(np[i]);
                    if (pp == p) continue;  // covers roots (np = nx) and pixels already pointing at a root
                    st64(node + base + i * kUnwrapThreads, pack_node(pp, node_off(nx[i]) + node_off(np[i])));
                    changed = 1;
                }
            }
            HOLO_UNWRAP_COUNT(n_jumps);
            if (!__syncthreads_or(changed)) break;
        }
        HOLO_UNWRAP_STAMP(t_jump);
    }
#ifdef HOLO_DEBUG
    // rounds and pointer-jumping sweeps it took (the edge slots are dead from here on)
    if (tid == 0 && n >= 5) {
        best[0] = (u64)n_rounds;
        best[1] = (u64)n_jumps;
        best[2] = (u64)t_edge; best[3] = (u64)t_hook; best[4] = (u64)t_jump;
    }
#endif
#undef HOLO_UNWRAP_STAMP
#undef HOLO_UNWRAP_COUNT

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
