// ds_read_b64_tr_b16 semantics check (gfx950): per 16-lane group, lane 4q+p supplies the address of row q, columns 4p..4p+3;
// lane i of the group receives column i of the 4 rows (element q = row q).   hipcc --offload-arch=gfx950 -O3 tr_read_check.hip -o tr_read_check
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ void k(short *out)
{
    __shared__ __attribute__((aligned(16))) short img[64 * 40];                 // [row][40]: value 100 row + col
    for (int i = threadIdx.x; i < 64 * 40; i += 64) img[i] = (short)(100 * (i / 40) + i % 40);
    __syncthreads();
    const int lane = threadIdx.x, g = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;
    const short *src = img + (4 * g + q) * 40 + 4 * p;                          // group g: rows 4g..4g+3, columns 0..15
    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)src);
    for (int e = 0; e < 4; ++e) out[lane * 4 + e] = v[e];
}
int main()
{
    short *d, h[256];
    hipMalloc(&d, sizeof(h));
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int lane = 0; lane < 64; ++lane)
        for (int e = 0; e < 4; ++e) {
            const int want = 100 * (4 * (lane >> 4) + e) + (lane & 15);
            if (h[lane * 4 + e] != want) { if (bad < 8) printf("lane %d e %d: got %d want %d\n", lane, e, h[lane * 4 + e], want); ++bad; }
        }
    printf("tr_read_check: %s (%d mismatches)\n", bad ? "FAIL" : "OK", bad);
    return bad != 0;
}
