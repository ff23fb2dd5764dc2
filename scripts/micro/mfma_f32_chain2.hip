// fp32 MFMA (v_mfma_f32_32x32x2_f32) dependent chain on ONE accumulator: cycles per MFMA at 1 and 2 waves per SIMD,
// operands constant (mode 0) or changing per MFMA like the kNN kernel's (32 different A registers; mode 1).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int MODE>
__global__ void __launch_bounds__(256, 2) k(float *out, long long *cyc, int iters, float a0)
{
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    float a[32], b[32];
    for (int s = 0; s < 32; ++s) { a[s] = a0 + s + threadIdx.x; b[s] = 0.5f + s; }
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < 32; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(MODE ? a[s] : a[0], MODE ? b[s] : b[0], acc, 0, 0, 0);
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    float res = 0.f;
    for (int r = 0; r < 16; ++r) res += acc[r];
    out[blockIdx.x * 256 + threadIdx.x] = res;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int MODE> void run(int blocks, float *d, long long *dc)
{
    const int iters = 1000;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, dc, 10, 1.f);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, dc, iters, 1.f);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    long long h[512]; (void)hipMemcpy(h, dc, sizeof(long long) * blocks, hipMemcpyDeviceToHost);
    printf("mode %d, %d blocks (%d wave(s)/SIMD): %.1f ticks per MFMA per wave, wall %.3f ms -> %.1f TFLOP/s\n", MODE, blocks, blocks / 256,
           (double)h[blocks / 2] / (iters * 32.0), ms, (double)blocks * 4 * iters * 32 * 4096.0 / (ms * 1e-3) / 1e12);
}
int main()
{
    float *d; long long *dc; (void)hipMalloc(&d, 512 * 256 * 4); (void)hipMalloc(&dc, 512 * 8);
    run<0>(256, d, dc); run<0>(512, d, dc); run<1>(256, d, dc); run<1>(512, d, dc);
    return 0;
}
