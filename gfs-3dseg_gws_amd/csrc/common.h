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
extern bool g_bn_tail;            // gfs3d_set_bn_tail: BatchNorm finalisation in the producing kernel's last workgroup (pointwise.hip: bn_tail)
extern bool g_deterministic;      // gfs3d_set_deterministic: ordered reductions instead of fp64 atomics (api.hip)

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


// Cross-lane sums as v_add_f32 with DPP operands instead of __shfl_xor (which compiles to ds_bpermute_b32: a trip through the LDS
// crossbar, ~100 cycles of latency each - four DEPENDENT ones per value made kpp_gain_kernel and the per-target sums of
// ec_bwd_segsum_kernel run at that latency).  Every lane of the group receives the group's sum, bit for bit the xor butterfly's
// value (after each step the exchanged lanes hold the same partial, and a + b == b + a).
//   quad_perm [1,0,3,2] = xor 1, quad_perm [2,3,0,1] = xor 2, row_half_mirror = lane 7 - i of each 8 (after the quad steps: xor 4),
//   row_mirror = lane 15 - i of each 16 (after those: xor 8)
#define GFS3D_DPP_ADD(v, ctrl) ((v) + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, (v)), (ctrl), 0xf, 0xf, true)))
__device__ __forceinline__ float group4_sum(float v) { v = GFS3D_DPP_ADD(v, 0xB1); v = GFS3D_DPP_ADD(v, 0x4E); return v; }
__device__ __forceinline__ float group8_sum(float v) { v = group4_sum(v); v = GFS3D_DPP_ADD(v, 0x141); return v; }
__device__ __forceinline__ float row16_sum(float v) { v = group8_sum(v); v = GFS3D_DPP_ADD(v, 0x140); return v; }
// value of lane (lane ^ ST) for ST in {1, 2, 4, 8} as DPP moves (no LDS crossbar), ds_bpermute otherwise:
//   xor 1 / xor 2 = quad_perm, xor 8 = row_ror:8 (a rotation by half a row is its own inverse), xor 4 = row_half_mirror (lane 7 - i
//   of each 8) followed by quad_perm [3,2,1,0]: i = 4 a + b -> 7 - i = 4 (1 - a) + (3 - b) -> 4 (1 - a) + b
#define GFS3D_DPP_MOV(x, ctrl) __builtin_amdgcn_update_dpp(0, (x), (ctrl), 0xf, 0xf, true)
__device__ __forceinline__ int lane_xor_i(int x, int st)
{
    switch (st) {
    case 1: return GFS3D_DPP_MOV(x, 0xB1);
    case 2: return GFS3D_DPP_MOV(x, 0x4E);
    case 4: return GFS3D_DPP_MOV(GFS3D_DPP_MOV(x, 0x141), 0x1B);
    case 8: return GFS3D_DPP_MOV(x, 0x128);
    default: return __shfl_xor(x, st, 64);
    }
}
__device__ __forceinline__ float lane_xor_f(float x, int st) { return __builtin_bit_cast(float, lane_xor_i(__builtin_bit_cast(int, x), st)); }

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

// ---- arithmetic of the GEMM-shaped kernels, selectable per call (gfs3d.h: GFS3D_COMPUTE_*) ----
//   PREC_X6   fp32 accuracy: exact 3-way bf16 split of both operands, six piece products per fp32 product (default)
//   PREC_BF16 ONE bf16 product per fp32 product: operands rounded to nearest-even bf16 (8 significand bits), fp32 accumulate
//   PREC_F16  the same with IEEE half operands (11 significand bits; forward passes only - gradients underflow in half)
// All three use the same LDS row images ([piece 0 | piece 1 | piece 2 | pad]); the reduced forms fill and read piece 0 only.
//   PREC_X3   (internal: screens with an error bound, never a result) operands as hi + mid round-to-nearest bf16 pairs, the three
//             products hi hi + hi mid + mid hi: ~2^-15 |x||y| per product dropped (the screened kNN's form, knn.hip)
enum { PREC_X6 = GFS3D_COMPUTE_F32, PREC_BF16 = GFS3D_COMPUTE_BF16, PREC_F16 = GFS3D_COMPUTE_F16, PREC_X3 = 3 };
typedef float mm_f32x16 __attribute__((ext_vector_type(16)));
typedef float mm_f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 mm_bf16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 mm_f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
template <int PREC> struct Mx {
    static constexpr int NP = PREC == PREC_X6 ? 3 : (PREC == PREC_X3 ? 2 : 1);        // operand pieces
    static constexpr int NTERM = PREC == PREC_X6 ? 6 : (PREC == PREC_X3 ? 3 : 1);     // piece products per fp32 product
};
// two fp32 values -> one dword of two 16-bit operands (a in the low half), rounded to nearest-even
template <int PREC> __device__ __forceinline__ unsigned mm_pack2(float a, float b)
{
    static_assert(PREC == PREC_BF16 || PREC == PREC_F16 || PREC == PREC_X3, "mm_pack2: reduced forms only");
    const mm_f32x2 v = {a, b};
    if (PREC == PREC_BF16 || PREC == PREC_X3) return __builtin_bit_cast(unsigned, __builtin_convertvector(v, mm_bf16x2));
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, mm_f16x2));
}
// one fp32 value -> its 16-bit operand (the weight / prototype images are written one element at a time)
template <int PREC> __device__ __forceinline__ unsigned short mm_cvt1(float a) { return (unsigned short)(mm_pack2<PREC>(a, 0.f) & 0xffffu); }
// two fp32 values -> their hi and mid bf16 pieces (both roundings to nearest-even; a - hi is exact), each pair in one dword
__device__ __forceinline__ void mm_split2_rne(float a, float b, unsigned &hi, unsigned &mid)
{
    hi = mm_pack2<PREC_BF16>(a, b);
    const float ra = a - __builtin_bit_cast(float, hi << 16), rb = b - __builtin_bit_cast(float, hi & 0xffff0000u);
    mid = mm_pack2<PREC_BF16>(ra, rb);
}
// eight consecutive K values of one operand row -> NP dwordx4 pieces ([0] hi, [1] mid, [2] lo for PREC_X6)
template <int PREC> __device__ __forceinline__ void mm_pieces8(const float *v8, uint4 (&out)[Mx<PREC>::NP])
{
    if constexpr (PREC == PREC_X6) {
        unsigned ph[8], pm[8], pl[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) split3(v8[e], ph[e], pm[e], pl[e]);
        out[0] = make_uint4(pack_hi16(ph[0], ph[1]), pack_hi16(ph[2], ph[3]), pack_hi16(ph[4], ph[5]), pack_hi16(ph[6], ph[7]));
        out[1] = make_uint4(pack_hi16(pm[0], pm[1]), pack_hi16(pm[2], pm[3]), pack_hi16(pm[4], pm[5]), pack_hi16(pm[6], pm[7]));
        out[2] = make_uint4(pack_hi16(pl[0], pl[1]), pack_hi16(pl[2], pl[3]), pack_hi16(pl[4], pl[5]), pack_hi16(pl[6], pl[7]));
    } else if constexpr (PREC == PREC_X3) {
        unsigned hi[4], mid[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) mm_split2_rne(v8[2 * e], v8[2 * e + 1], hi[e], mid[e]);
        out[0] = make_uint4(hi[0], hi[1], hi[2], hi[3]);
        out[1] = make_uint4(mid[0], mid[1], mid[2], mid[3]);
    } else {
        out[0] = make_uint4(mm_pack2<PREC>(v8[0], v8[1]), mm_pack2<PREC>(v8[2], v8[3]), mm_pack2<PREC>(v8[4], v8[5]), mm_pack2<PREC>(v8[6], v8[7]));
    }
}
// four consecutive K values -> NP dwordx2 pieces (the staging stores of 32-channel chunk images)
template <int PREC> __device__ __forceinline__ void mm_pieces4(const float *v4, uint2 (&out)[Mx<PREC>::NP])
{
    if constexpr (PREC == PREC_X6) {
        unsigned ph[4], pm[4], pl[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) split3(v4[e], ph[e], pm[e], pl[e]);
        out[0] = make_uint2(pack_hi16(ph[0], ph[1]), pack_hi16(ph[2], ph[3]));
        out[1] = make_uint2(pack_hi16(pm[0], pm[1]), pack_hi16(pm[2], pm[3]));
        out[2] = make_uint2(pack_hi16(pl[0], pl[1]), pack_hi16(pl[2], pl[3]));
    } else if constexpr (PREC == PREC_X3) {
        unsigned hi[2], mid[2];
#pragma unroll
        for (int e = 0; e < 2; ++e) mm_split2_rne(v4[2 * e], v4[2 * e + 1], hi[e], mid[e]);
        out[0] = make_uint2(hi[0], hi[1]);
        out[1] = make_uint2(mid[0], mid[1]);
    } else {
        out[0] = make_uint2(mm_pack2<PREC>(v4[0], v4[1]), mm_pack2<PREC>(v4[2], v4[3]));
    }
}
// one element of a 16-bit operand image: plane p of the row at `o`, planes `plane` elements apart
template <int PREC> __device__ __forceinline__ void mm_store1(float w, unsigned short *o, int plane)
{
    if constexpr (PREC == PREC_X6) {
        unsigned ph, pm, pl;
        split3(w, ph, pm, pl);
        o[0] = (unsigned short)(ph >> 16);
        o[plane] = (unsigned short)(pm >> 16);
        o[2 * plane] = (unsigned short)(pl >> 16);
    } else {
        o[0] = mm_cvt1<PREC>(w);
    }
}
template <int PREC> __device__ __forceinline__ mm_f32x16 mm_mfma(const uint4 &a, const uint4 &b, mm_f32x16 c)
{
    if constexpr (PREC == PREC_F16)
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    else
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
// acc += a . b over one K-step of 16: all piece products, smallest first
template <int PREC> __device__ __forceinline__ void mm_mac(const uint4 (&a)[Mx<PREC>::NP], const uint4 (&b)[Mx<PREC>::NP], mm_f32x16 &acc)
{
    if constexpr (PREC == PREC_X6) {
        constexpr int PA[6] = GFS3D_X6_PA, PB[6] = GFS3D_X6_PB;
#pragma unroll
        for (int term = 0; term < 6; ++term) acc = mm_mfma<PREC>(a[PA[term]], b[PB[term]], acc);
    } else if constexpr (PREC == PREC_X3) {
        acc = mm_mfma<PREC>(a[1], b[0], acc);
        acc = mm_mfma<PREC>(a[0], b[1], acc);
        acc = mm_mfma<PREC>(a[0], b[0], acc);
    } else {
        acc = mm_mfma<PREC>(a[0], b[0], acc);
    }
}
// the reduced form a call asks for, per pass: half operands cannot hold gradients (~1e-8 under a mean loss)
static inline int prec_of(int compute, bool backward) { return compute == PREC_F16 && backward ? PREC_BF16 : compute; }

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
// w_pq != 0 (Nc = 128): w is an EdgeConv first convolution (64, 2 K) = [Wa | Wb], the 128 weight rows are [Wa; Wb - Wa], formed
// while they are staged.  zero / nzero: doubles the first workgroup clears (the caller's BatchNorm accumulators).
int pw_plain_gemm(const float *a, int lda, const float *w, int M, int K, int Nc, float *const *out, int o_segw,
                  const float *ea, const float *eb, hipStream_t st, const char *who, int w_pq = 0, double *zero = nullptr, int nzero = 0);

}  // namespace gfs3d
