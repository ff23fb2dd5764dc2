// Micro-benchmark: do an MFMA wave and a VALU wave that share a SIMD overlap?
// Block = 512 threads = 8 waves = 2 per SIMD: waves 0-3 run dependent f32 MFMAs, waves 4-7 run v_fma_f32.
// Times: MFMA only, VALU only, both.   hipcc --offload-arch=gfx950 -O3 mfma_valu_overlap.hip -o mfma_valu_overlap
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
template <int BF16>
__global__ void __launch_bounds__(512) k(float *out, int iters, int do_mfma, int do_valu, float a0)
{
    const int w = threadIdx.x >> 6;
    float res = 0.f;
    if (w < 4) {
        if (do_mfma) {
            f32x16 acc; for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            float a = a0 + threadIdx.x, b = 2.f;
            bf16x8 ab, bb;
            for (int e = 0; e < 8; ++e) { ab[e] = (__bf16)(a + e); bb[e] = (__bf16)(b + e); }
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int s = 0; s < 32; ++s) {
                    if (BF16) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, bb, acc, 0, 0, 0);
                    else acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
                }
            }
            for (int r = 0; r < 16; ++r) res += acc[r];
        }
    } else if (do_valu) {
        float v[8];
        for (int i = 0; i < 8; ++i) v[i] = a0 + i + threadIdx.x;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int s = 0; s < 64; ++s)
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = __builtin_fmaf(v[i], 1.0001f, 0.5f);
        }
        for (int i = 0; i < 8; ++i) res += v[i];
    }
    out[blockIdx.x * 512 + threadIdx.x] = res;
}
template <int BF16>
float run(float *d, int m, int v)
{
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<BF16>, dim3(256), dim3(512), 0, 0, d, 10, m, v, 1.f);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<BF16>, dim3(256), dim3(512), 0, 0, d, 1000, m, v, 1.f);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    return ms;
}
int main()
{
    float *d; (void)hipMalloc(&d, 256 * 512 * 4);
    {
        const float tm = run<1>(d, 1, 0), tv = run<1>(d, 0, 1), tb = run<1>(d, 1, 1);
        printf("bf16 32x32x16: MFMA only %.3f ms (%.1f cycles per MFMA), VALU only %.3f ms, both %.3f ms (sum %.3f, max %.3f)\n", tm,
               tm * 2.4e6 / 32000, tv, tb, tm + tv, tm > tv ? tm : tv);
    }
    const float tm = run<0>(d, 1, 0), tv = run<0>(d, 0, 1), tb = run<0>(d, 1, 1);
    printf("MFMA only %.3f ms (32 MFMA x 1000 iters: %.0f cycles/iter @2.4GHz)\n", tm, tm * 2.4e6 / 1000);
    printf("VALU only %.3f ms (512 v_fma x 1000 iters: %.0f cycles/iter -> %.2f cycles per wave64 VALU op)\n", tv, tv * 2.4e6 / 1000, tv * 2.4e6 / 1000 / 512);
    printf("both      %.3f ms  (sum %.3f, max %.3f)\n", tb, tm + tv, tm > tv ? tm : tv);
    return 0;
}
