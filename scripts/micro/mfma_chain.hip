// Micro-benchmark: v_mfma_f32_32x32x2_f32 throughput for a DEPENDENT chain on one accumulator vs independent
// accumulators, at 1 / 2 / 4 waves per SIMD.   hipcc --offload-arch=gfx950 -O3 mfma_chain.hip -o mfma_chain
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC>
__global__ void __launch_bounds__(256) k(float *out, int iters, float a0, float b0)
{
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float a = a0 + threadIdx.x, b = b0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < 32; ++s) acc[s % NACC] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[s % NACC], 0, 0, 0);
    }
    float s = 0;
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NACC> void run(int blocks_per_cu, float *d)
{
    const int iters = 2000, grid = 256 * blocks_per_cu;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<NACC>, dim3(grid), dim3(256), 0, 0, d, 10, 1.f, 2.f);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<NACC>, dim3(grid), dim3(256), 0, 0, d, iters, 1.f, 2.f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double mfma_per_simd = (double)iters * 32 * blocks_per_cu;     // one wave of each block per SIMD
    printf("NACC=%d waves/SIMD=%d: %.3f ms, %.1f ns per MFMA per SIMD (64 cycles @2.4GHz = 26.7 ns), %.1f TF\n", NACC, blocks_per_cu, ms,
           ms * 1e6 / mfma_per_simd, mfma_per_simd * 1024 * 4096 / (ms * 1e-3) / 1e12);
}
int main()
{
    float *d; hipMalloc(&d, 256 * 256 * 8 * 4);
    for (int w : {1, 2, 4}) { run<1>(w, d); run<2>(w, d); run<4>(w, d); }
    return 0;
}
