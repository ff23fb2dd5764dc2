// Shared helpers for the gfx950 kernels (wave64, CDNA4).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
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


// ---- fp32-accurate products on the bf16 matrix cores ("bf16x6", see pointwise.hip) ----
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
// x == h + m + l exactly, each piece a bf16 (stored in the high half of a dword): truncation splits of 8 + 8 + 8 bits
__device__ __forceinline__ void split3(float x, unsigned &h, unsigned &m, unsigned &l)
{
    h = __builtin_bit_cast(unsigned, x) & 0xffff0000u;
    const float r1 = x - __builtin_bit_cast(float, h);                  // exact: the low 16 significand bits
    m = __builtin_bit_cast(unsigned, r1) & 0xffff0000u;
    l = __builtin_bit_cast(unsigned, r1 - __builtin_bit_cast(float, m)); // exact, <= 8 significant bits: low half is 0
}
// two fp32 bit patterns whose low halves are dropped -> one dword of two bf16 (a first)
__device__ __forceinline__ unsigned pack_hi16(unsigned a, unsigned b) { return __builtin_amdgcn_perm(b, a, 0x07060302u); }
// piece pairs of a.b in ascending weight: (lo hi) (hi lo) (mid mid) (mid hi) (hi mid) (hi hi); pieces 0 = hi, 1 = mid, 2 = lo
#define GFS3D_X6_PA {2, 0, 1, 1, 0, 0}
#define GFS3D_X6_PB {0, 2, 1, 0, 1, 0}

static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// Dynamic LDS above 64 KB needs hipFuncAttributeMaxDynamicSharedMemorySize, per kernel AND per device.
// `seen` is the call site's static table (one per kernel instantiation): the largest size granted so far on each device.
constexpr int GFS3D_MAX_DEVICES = 32;
struct LdsGrant { int bytes[GFS3D_MAX_DEVICES] = {}; };
static inline int ensure_dynamic_lds(const void *kernel, size_t bytes, LdsGrant &seen, const char *who)
{
    if (bytes <= 64 * 1024) return GFS3D_OK;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) dev = -1;
    const bool track = dev >= 0 && dev < GFS3D_MAX_DEVICES;
    if (track && seen.bytes[dev] >= (int)bytes) return GFS3D_OK;
    const hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) {
        set_error("%s: hipFuncSetAttribute(%zu B of LDS): %s", who, bytes, hipGetErrorString(e));
        return GFS3D_ERR_HIP;
    }
    if (track) seen.bytes[dev] = (int)bytes;
    return GFS3D_OK;
}
#define GFS3D_ENSURE_LDS(kernel, bytes, who)                                                             \
    do {                                                                                                 \
        static ::gfs3d::LdsGrant grant__;                                                                \
        if (int rc__ = ::gfs3d::ensure_dynamic_lds(reinterpret_cast<const void *>(kernel), bytes, grant__, who)) return rc__; \
    } while (0)

// pointwise.hip: out(M, Nc) = a(M, K) . w(Nc, K)^T on MFMA, point-major; the Nc columns are written as segments of
// width o_segw to out[0], out[1], ...  (K % 32 == 0, Nc % 64 == 0, o_segw % 64 == 0, at most 4 segments)
// ea / eb (Nc floats each, or both NULL): out = fma(acc, ea[col], eb[col])
int pw_plain_gemm(const float *a, int lda, const float *w, int M, int K, int Nc, float *const *out, int o_segw,
                  const float *ea, const float *eb, hipStream_t st, const char *who);

}  // namespace gfs3d
