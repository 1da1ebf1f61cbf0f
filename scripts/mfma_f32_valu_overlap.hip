// Micro-benchmark: do vector-ALU instructions issued between v_mfma_f32_32x32x2_f32 (the exact-fp32 matrix instruction, 64 cycles)
// run in its shadow, or do they take issue cycles away from it?  One wave per SIMD (256 threads, one block per CU), a loop of
// 4 independent MFMAs with N filler instructions after each; prints cycles per MFMA (s_memtime) and TFLOP/s.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/ovl scripts/mfma_f32_valu_overlap.hip && /tmp/ovl
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float float2v __attribute__((ext_vector_type(2)));

// KIND 0: v_add_f32; 1: v_pk_add_f32; 2: v_mov_b32; 3: ds_read_b128 (conflict-free, never waited for inside the loop); 4: s_nop 0
template <int N, int KIND>
__global__ __launch_bounds__(256) void k(float* out, unsigned long long* cyc, int iters) {
    __shared__ __attribute__((aligned(16))) float lds[4096];
    const int tid = threadIdx.x;
    for (int i = tid; i < 4096; i += 256) lds[i] = 1e-3f * i;
    __syncthreads();
    floatx16 acc[4];
    for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    float a0 = lds[tid], b0 = lds[tid + 256];
    float f[16];
    float2v p[16];
    float4 q[4];
    for (int i = 0; i < 16; ++i) { f[i] = lds[tid + i]; p[i] = float2v{lds[tid + 2 * i], lds[tid + 2 * i + 1]}; }
    const float4* l4 = reinterpret_cast<const float4*>(lds) + (tid & 63);
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(acc[m]) : "v"(a0), "v"(b0));
#pragma unroll
            for (int n = 0; n < N; ++n) {
                const int r = (m * N + n) & 15;
                if (KIND == 0) asm volatile("v_add_f32 %0, %1, %2" : "=v"(f[r]) : "v"(a0), "v"(b0));
                if (KIND == 1) asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(p[r]) : "v"(p[(r + 5) & 15]), "v"(p[(r + 9) & 15]));
                if (KIND == 2) asm volatile("v_mov_b32 %0, %1" : "=v"(f[r]) : "v"(a0));
                if (KIND == 3) asm volatile("ds_read_b128 %0, %1" : "=v"(q[r & 3]) : "v"((unsigned)(unsigned long long)(const void*)(l4) & 0xffffu));
                if (KIND == 4) asm volatile("s_nop 0");
            }
        }
        if (KIND == 3) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) s += acc[a][r];
    for (int i = 0; i < 16; ++i) s += f[i] + p[i].x + p[i].y;
    for (int i = 0; i < 4; ++i) s += q[i].x;
    out[blockIdx.x * 256 + tid] = s;
    if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int N, int KIND>
void run(const char* name) {
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 256 * 256 * sizeof(float)); hipMalloc(&cyc, 256 * sizeof(unsigned long long));
    const int iters = 20000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<N, KIND>), dim3(256), dim3(256), 0, 0, out, cyc, 100);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<N, KIND>), dim3(256), dim3(256), 0, 0, out, cyc, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[256]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double c = 0; for (int i = 0; i < 256; ++i) c += (double)h[i]; c /= 256;
    const double flops = 256.0 * 4 * iters * 4 * 4096.0;
    printf("%-14s N=%2d: %7.1f cycles per MFMA, %6.1f TFLOP/s\n", name, N, c / (iters * 4.0), flops / ms / 1e9);
    hipFree(out); hipFree(cyc);
}

int main() {
    run<0, 0>("bare");
    run<1, 0>("v_add_f32"); run<2, 0>("v_add_f32"); run<4, 0>("v_add_f32"); run<8, 0>("v_add_f32"); run<12, 0>("v_add_f32"); run<16, 0>("v_add_f32");
    run<1, 1>("v_pk_add_f32"); run<2, 1>("v_pk_add_f32"); run<4, 1>("v_pk_add_f32"); run<8, 1>("v_pk_add_f32"); run<16, 1>("v_pk_add_f32");
    run<2, 2>("v_mov_b32"); run<8, 2>("v_mov_b32"); run<16, 2>("v_mov_b32");
    run<1, 3>("ds_read_b128"); run<2, 3>("ds_read_b128"); run<4, 3>("ds_read_b128");
    run<4, 4>("s_nop 0"); run<12, 4>("s_nop 0");
    return 0;
}
