// Micro-benchmark (hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_bench scripts/mfma_rate_bench.hip && /tmp/mfma_bench): what rate a bare
// v_mfma_f32_32x32x16_f16 loop reaches on this chip, alone and with one conflict-free ds_read_b128 per MFMA.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef _Float16 halfx8 __attribute__((ext_vector_type(8)));

template <int NACC, int MODE>   // MODE 0: MFMA only; 1: + ds_read_b128 of every operand (conflict-free, per j 2 A + 2 B reads)
__global__ __launch_bounds__(256) void k(float* out, int iters, int waves_dummy) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    floatx16 acc[NACC];
    for (int a = 0; a < NACC; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    float4* l4 = reinterpret_cast<float4*>(smem);
    for (int i = tid; i < 4096; i += 256) l4[i] = make_float4(1e-3f * i, 1.f, 0.5f, 0.25f);
    __syncthreads();
    float4 a0 = l4[lane], a1 = l4[lane + 64], b0 = l4[lane + 128], b1 = l4[lane + 192];
    const unsigned base = (tid >> 6) * 8192;   // per-wave region
    for (int it = 0; it < iters; ++it) {
        if (MODE == 1) {
            const unsigned off = base + ((it & 7) * 1024) + lane * 16;   // contiguous 1 KB per read: conflict-free
            a0 = *reinterpret_cast<const float4*>(smem + off);
            a1 = *reinterpret_cast<const float4*>(smem + ((off + 256 * 16) & 65535));
            b0 = *reinterpret_cast<const float4*>(smem + ((off + 512 * 16) & 65535));
            b1 = *reinterpret_cast<const float4*>(smem + ((off + 768 * 16) & 65535));
        }
        const halfx8 A0 = __builtin_bit_cast(halfx8, a0), A1 = __builtin_bit_cast(halfx8, a1);
        const halfx8 B0 = __builtin_bit_cast(halfx8, b0), B1 = __builtin_bit_cast(halfx8, b1);
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A0, B0, acc[0], 0, 0, 0);
        acc[1 % NACC] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A0, B1, acc[1 % NACC], 0, 0, 0);
        acc[2 % NACC] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A1, B0, acc[2 % NACC], 0, 0, 0);
        acc[3 % NACC] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A1, B1, acc[3 % NACC], 0, 0, 0);
    }
    float s = 0;
    for (int a = 0; a < NACC; ++a) for (int r = 0; r < 16; ++r) s += acc[a][r];
    out[blockIdx.x * 256 + tid] = s;
}

template <int NACC, int MODE>
void run(const char* name, int blocks_per_cu, int lds_bytes) {
    float* out; hipMalloc(&out, 256 * 16 * 256 * sizeof(float));
    const int iters = 20000, blocks = 256 * blocks_per_cu;
    hipFuncSetAttribute(reinterpret_cast<const void*>(&k<NACC, MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<NACC, MODE>), dim3(blocks), dim3(256), lds_bytes, 0, out, 100, 0);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<NACC, MODE>), dim3(blocks), dim3(256), lds_bytes, 0, out, iters, 0);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)blocks * 4 /*waves*/ * iters * 4 /*mfma*/ * 32768.0;
    printf("%-40s blocks/CU %d  LDS %6d B: %8.1f TFLOP/s (%.1f%% of 2500)\n", name, blocks_per_cu, lds_bytes, flops / ms / 1e9, flops / ms / 1e9 / 25.0);
    hipFree(out);
}

int main() {
    run<4, 0>("MFMA only, 4 acc", 1, 65536);
    run<4, 0>("MFMA only, 4 acc", 2, 65536);
    run<4, 1>("MFMA + 4 ds_read_b128 per 4 MFMA", 1, 65536);
    run<4, 1>("MFMA + 4 ds_read_b128 per 4 MFMA", 2, 65536);
    run<4, 1>("MFMA + 4 ds_read_b128 per 4 MFMA", 3, 49152 > 65536 ? 65536 : 49152 + 16384);
    return 0;
}
