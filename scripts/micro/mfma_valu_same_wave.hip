// Micro-benchmark: inside ONE wave, do independent v_fma_f32 issued between MFMAs hide under them?
// Per iteration: 8 x (1 MFMA + NV independent v_fma).  One wave per SIMD (256-thread blocks, 1 block per CU).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
template <int BF16, int NV>
__global__ void __launch_bounds__(256) k(float *out, int iters, float a0)
{
    f32x16 acc[2];
    for (int i = 0; i < 2; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float a = a0 + threadIdx.x, b = 2.f;
    bf16x8 ab, bb;
    for (int e = 0; e < 8; ++e) { ab[e] = (__bf16)(a + e); bb[e] = (__bf16)(b + e); }
    float v[16];
    for (int i = 0; i < 16; ++i) v[i] = a0 + i + threadIdx.x;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            if (BF16) acc[s & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, bb, acc[s & 1], 0, 0, 0);
            else acc[s & 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[s & 1], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < NV; ++i) v[i & 15] = __builtin_fmaf(v[i & 15], 1.0001f, 0.5f);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float res = 0.f;
    for (int i = 0; i < 2; ++i) for (int r = 0; r < 16; ++r) res += acc[i][r];
    for (int i = 0; i < 16; ++i) res += v[i];
    out[blockIdx.x * 256 + threadIdx.x] = res;
}
template <int BF16, int NV> void run(float *d)
{
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<BF16, NV>), dim3(256), dim3(256), 0, 0, d, 10, 1.f);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<BF16, NV>), dim3(256), dim3(256), 0, 0, d, 4000, 1.f);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%s MFMA + %2d v_fma: %.1f cycles per (MFMA + VALU group)\n", BF16 ? "bf16 32x32x16" : "f32  32x32x2 ", NV, ms * 2.4e6 / (4000.0 * 8));
}
int main()
{
    float *d; (void)hipMalloc(&d, 256 * 256 * 4);
    run<0, 0>(d); run<0, 4>(d); run<0, 8>(d); run<0, 16>(d); run<0, 32>(d);
    run<1, 0>(d); run<1, 4>(d); run<1, 8>(d); run<1, 16>(d); run<1, 32>(d);
    return 0;
}
