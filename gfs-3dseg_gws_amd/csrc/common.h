// Shared helpers for the gfx950 kernels (wave64, CDNA4).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/gfs3d.h"

namespace gfs3d {

constexpr int WAVE = 64;

void set_error(const char *fmt, ...);

// measurement hooks (api.hip): no-ops unless gfs3d_profile_enable(1)
int prof_begin(int kernel_id, hipStream_t st);
void prof_end(int token, hipStream_t st);
struct ProfScope {
    int tok; hipStream_t st;
    ProfScope(int id, hipStream_t s) : tok(prof_begin(id, s)), st(s) {}
    ~ProfScope() { prof_end(tok, st); }
};

#define GFS3D_REQUIRE(cond, ...)                                                        \
    do {                                                                                \
        if (!(cond)) {                                                                  \
            ::gfs3d::set_error(__VA_ARGS__);                                            \
            return GFS3D_ERR_INVALID;                                                   \
        }                                                                               \
    } while (0)

#define GFS3D_LAUNCH_CHECK(name)                                                        \
    do {                                                                                \
        hipError_t e__ = hipGetLastError();                                             \
        if (e__ != hipSuccess) {                                                        \
            ::gfs3d::set_error("%s: launch failed: %s", name, hipGetErrorString(e__));  \
            return GFS3D_ERR_HIP;                                                       \
        }                                                                               \
    } while (0)

__device__ __forceinline__ float lrelu(float v, float slope) { return v > 0.f ? v : v * slope; }

// wave64 all-lanes sum via DPP-free xor shuffles
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ int wave_max(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o, 64));
    return v;
}

static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// pointwise.hip: out(M, Nc) = a(M, K) . w(Nc, K)^T on MFMA, point-major; the Nc columns are written as segments of
// width o_segw to out[0], out[1], ...  (K % 32 == 0, Nc % 64 == 0, o_segw % 64 == 0, at most 4 segments)
int pw_plain_gemm(const float *a, int lda, const float *w, int M, int K, int Nc, float *const *out, int o_segw, hipStream_t st,
                  const char *who);

}  // namespace gfs3d
