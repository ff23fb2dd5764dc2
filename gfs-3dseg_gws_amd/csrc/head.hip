// K6: the prototype head of the CAPL training step, fused  (reference model/capl.py:194-242 with generate_fake_proto
// :364-411, get_pred :290-322, post_refine_proto_v2 :245-287 and the two CrossEntropyLoss calls of train.py:317).
//
//   ori      = unit(main_proto), rows of the fake-novel classes replaced by the masked mean of the unit point features of the
//              support half of the batch                                                   (generate_fake_proto)
//   L1       = 10 cos(F, [bg; ori])                               loss_1 = CE(L1, y)          (get_pred, 2-D prototypes)
//   resp     = softmax over the POINTS of 10 cos(F, [bg; main])   pooled = resp F^T           (post_refine_proto_v2)
//   w        = relu(cos(pooled, main))                            ref0   = w pooled + (1 - w) main
//   refined  = ref0 + ori (base classes) | ori (the others)
//   L2       = 10 cos(F, [bg; refined_b])                         loss_2 = CE(L2, y), pred = argmax L2   (3-D prototypes)
//   loss     = (loss_1 + loss_2) / 2
//
// The torch form of this chain is ~200 small launches per step (1.2 of the 8.3 ms of BASELINE configs[3]); here it is
// four passes over the (B, 128, N) feature tensor (two forward, two backward) and a handful of per-class kernels:
//   hd_fwd1   per point: cosines against [bg; main], e = exp(L3 - 10); per workgroup the point contractions
//             sum_n e[k][n] F[c][n] (soft pooling) and sum_n [y_n = fake_f] F[c][n] / |F_n| (fake-novel sums) as ONE fp32 MFMA
//             product (rows = classes + fake classes, k = points)                             -> partial tiles
//   hd_mid1   fake-novel sums over the support half                                           -> fsum  (all-reduced under DP)
//   hd_mid2   per batch item: pooled, w, ref0, ori, refined -> unit prototype tables q1 (shared), q2 (per item)
//   hd_fwd2   per point: L1, L2, both cross-entropies, arg-max                                 -> loss partials, pred
//   hd_loss   fixed-order sum of the loss partials
//   hd_bwd1   per point: softmax - one-hot of both heads; the prototype-gradient contractions sum_n g[k][n] Fhat[c][n]
//   hd_bwdm1  per batch item: back through refined / ref0 / w / pooled                        -> dpooled, partial d main
//   hd_bwdm2  per class: gradient of ori, of the fake-novel sums (dfsum, all-reduced under DP), of main through unit()
//   hd_bwd2   per point: d F from all four paths (both heads, the soft pooling, the fake-novel means), written once;
//             contraction sum_n dL3[k][n] Fhat[c][n]
//   hd_bwdfin per class: gradient through the normalisation of [bg; main], final d main / d bg
// Layout of the point passes (round 5): every product runs on v_mfma_f32_16x16x4_f32 (exact fp32 products) - see "The four point
// passes on the fp32 matrix cores" below: a wave owns 32 or 64 points, the class tables live in LDS as A operands, the features
// travel in 32-channel slabs, the per-point arithmetic runs in place on the accumulator registers.  (The logits pass of the testing
// path, hd_fwd2_eval_kernel, keeps the lane = point form of rounds 3-4: hd_dots.)
// No atomics anywhere: every cross-workgroup sum is a partial tile added in a fixed order (bit-reproducible).
#include "common.h"

namespace gfs3d {
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int HD_C = 128;            // feature channels (main_dim of capl.py:46)
constexpr int HD_PTS = 256;          // points per workgroup (4 waves, lane = point)
constexpr int HD_FLD = HD_C + 1;     // row stride of the fsum / dfsum tables: [fake class][128 sums | count]
constexpr float HD_EPS = 1e-12f;
// per-item state of the refinement kept for the backward: hs[(b KT + k)][8] = {s, w, cos, |pooled|, |refined|, -, -, -}
constexpr int HD_HS = 8;
constexpr float HD_SCALE = 10.0f;    // get_pred's `pred * 10`
constexpr float HD_SHIFT = 10.0f;    // exp(L - 10): the logits are 10 cos in [-10, 10], so no running maximum is needed

struct HeadFake { int n, half; int cls[GFS3D_HEAD_MAX_FAKE]; };       // fake-novel labels; support half = batch items >= half

__device__ __forceinline__ int hd_acc_row(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }
__device__ __forceinline__ f32x4 hd_ldg4(const float *p) { return *reinterpret_cast<const f32x4 *>(p); }
__device__ __forceinline__ float hd_wsum(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ int hd_fake_index(const HeadFake &fk, long long y)
{
    int r = -1;
    for (int i = 0; i < fk.n; ++i) r = (y == (long long)fk.cls[i]) ? i : r;
    return r;
}

// rows [0, rows) of a (rows, 128) matrix into an LDS table of KT rows (the rest zero); 256 threads, ends with a barrier
template <int KT>
__device__ __forceinline__ void hd_load_table(float *dst, const float *src, int rows, int tid)
{
    for (int e = tid; e < KT * (HD_C / 4); e += 256) {
        const int row = e >> 5, c4 = e & 31;
        const f32x4 v = row < rows ? hd_ldg4(src + (size_t)row * HD_C + 4 * c4) : f32x4{0.f, 0.f, 0.f, 0.f};
        *reinterpret_cast<f32x4 *>(dst + row * HD_C + 4 * c4) = v;
    }
    __syncthreads();
}
// [bg; main] (K1 = classes + 1 rows; bg == NULL: main alone, K1 = classes), every row divided by max(|row|, eps)
// (F.normalize); 256 threads, ends with a barrier
template <int KT>
__device__ __forceinline__ void hd_load_bgmain_unit(float *dst, const float *bg, const float *main_proto, int K1, int tid)
{
    for (int e = tid; e < KT * (HD_C / 4); e += 256) {
        const int row = e >> 5, c4 = e & 31;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (bg == nullptr) { if (row < K1) v = hd_ldg4(main_proto + (size_t)row * HD_C + 4 * c4); }      // (the testing path: no bg row)
        else if (row == 0) v = hd_ldg4(bg + 4 * c4);
        else if (row < K1) v = hd_ldg4(main_proto + (size_t)(row - 1) * HD_C + 4 * c4);
        *reinterpret_cast<f32x4 *>(dst + row * HD_C + 4 * c4) = v;
    }
    __syncthreads();
    const int lane = tid & 63, w = tid >> 6;
    for (int row = w; row < K1; row += 4) {
        const float a = dst[row * HD_C + lane], b = dst[row * HD_C + 64 + lane];
        const float inv = 1.0f / fmaxf(sqrtf(hd_wsum(fmaf(a, a, b * b))), HD_EPS);
        dst[row * HD_C + lane] = a * inv;
        dst[row * HD_C + 64 + lane] = b * inv;
    }
    __syncthreads();
}

// lane = point: |f|^2 and the dot products of the point's 128 channels with NT tables of KT rows in LDS.  The column is
// streamed SIXTEEN channels at a time, the next sixteen requested before the current ones are used: 16 loads in flight per
// lane cover an HBM round trip with the ~4 KT NT 16-byte broadcast reads and 16 KT NT fma of a group (with four channels in
// flight the pass ran at the memory latency: 32 round trips per wave, 35 us for the two-table forward pass).
constexpr int HD_G = 16;
template <int KT, int NT>
__device__ __forceinline__ void hd_dots(const float *col, size_t N, const float *const (&tab)[NT], float (&d)[NT][KT], float &sq)
{
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int k = 0; k < KT; ++k) d[t][k] = 0.f;
    sq = 0.f;
    float a[HD_G];
#pragma unroll
    for (int e = 0; e < HD_G; ++e) a[e] = col[(size_t)e * N];
#pragma unroll 1
    for (int cg = 0; cg < HD_C / HD_G; ++cg) {
        float v[HD_G];
#pragma unroll
        for (int e = 0; e < HD_G; ++e) v[e] = a[e];
        if (cg + 1 < HD_C / HD_G) {
            const float *nx = col + (size_t)(HD_G * cg + HD_G) * N;
#pragma unroll
            for (int e = 0; e < HD_G; ++e) a[e] = nx[(size_t)e * N];
        }
        // The 16 channels x NT tables x KT rows of a group as a flat sequence of units of eight rows x four channels, software
        // pipelined by hand: the eight 16-byte table reads of unit u + 1 are issued, then the 32 fma of unit u, then an empty
        // asm statement that names the eight accumulators (and clobbers memory) pins that order.  Left to itself the compiler
        // places every table read of the group in front of the first fma (hundreds of registers); sched_barrier does not
        // help - instruction selection has already put the chain-free fma behind the last read when it sees the fences.
        constexpr int KG = 8, GPT = KT / KG, NU = (HD_G / 4) * NT * GPT;
        f32x4 qb[2][KG];
        auto rd = [&](int u, f32x4 (&q)[KG]) {
            const int q4 = u / (NT * GPT), t = (u / GPT) % NT, kg = u % GPT;
#pragma unroll
            for (int r = 0; r < KG; ++r) q[r] = *reinterpret_cast<const f32x4 *>(tab[t] + (kg * KG + r) * HD_C + HD_G * cg + 4 * q4);
        };
        rd(0, qb[0]);
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            const int q4 = u / (NT * GPT), t = (u / GPT) % NT, kg = u % GPT;
            if (u + 1 < NU) rd(u + 1, qb[(u + 1) & 1]);
            const float v0 = v[4 * q4], v1 = v[4 * q4 + 1], v2 = v[4 * q4 + 2], v3 = v[4 * q4 + 3];
            if (t == 0 && kg == 0) sq = fmaf(v3, v3, fmaf(v2, v2, fmaf(v1, v1, fmaf(v0, v0, sq))));
            float(&dk)[KT] = d[t];
#pragma unroll
            for (int r = 0; r < KG; ++r) {
                const f32x4 q = qb[u & 1][r];
                dk[kg * KG + r] = fmaf(v3, q[3], fmaf(v2, q[2], fmaf(v1, q[1], fmaf(v0, q[0], dk[kg * KG + r]))));
            }
            asm volatile("" : "+v"(dk[kg * KG]), "+v"(dk[kg * KG + 1]), "+v"(dk[kg * KG + 2]), "+v"(dk[kg * KG + 3]), "+v"(dk[kg * KG + 4]),
                         "+v"(dk[kg * KG + 5]), "+v"(dk[kg * KG + 6]), "+v"(dk[kg * KG + 7])
                         :
                         : "memory");
        }
    }
}




// =====================================================================================================================
// The four point passes on the fp32 matrix cores (round 5).  The lane = point form of rounds 3-4 (hd_dots, still used by the
// testing path's logits pass) spent its time on ~1000 LDS broadcast reads and 4096 fma per lane and pass with ONE wave per SIMD
// (65 536 points are 1024 waves) and 16 four-byte loads in flight per lane; here every product of the passes is a
// v_mfma_f32_16x16x4_f32 (exact fp32 products, fp32 accumulation):
//   dots         D[class][point]   = sum_c T[class][c] F[c][point]            (A = table, B = features, k = channel)
//   contractions D[row][channel]   = sum_p A[row][p] F[channel][p]            (A = per-point coefficients, B = features, k = point)
//   d F          D[channel][point] = sum_slot T[slot][channel] coef[slot][p]  (A = tables, B = coefficients, k = class slot)
// A wave owns 16 NTL points (NTL = 2 or 4 point tiles); lane = 16 g + j.  Point tile t has column j <-> point NTL j + t, so ONE
// access per lane covers all tiles; a 16x16 result tile leaves rows 4 g + r (r = 0..3) of column j in lane (g, j).  The table
// rows are fed permuted (A row i <-> class 4 (i & 3) + (i >> 2)), so lane (g, j) ends up with the classes k = 16 rt + 4 r + g of
// its points: exactly the B operand of K-step (rt, r) of the d F product - the per-point arithmetic (soft-max, cross-entropy,
// gradient coefficients) runs on those registers, IN PLACE, with two cross-lane steps (lanes 16 and 32 apart) per class
// reduction, and nothing goes back through LDS except the coefficient rows of the point contractions.
// The features travel in slabs of 32 channels per wave (16-byte global loads, all of a wave's loads issued up front and the tile
// kept in registers for the second sweep of a pass), rows 16 NTL + 4 floats apart in LDS: conflict-free for the reads along the
// points (B of the dots, epilogue of d F) and for the 16-byte reads along the channels' points (B of the contractions).
// NTL = 2 wherever two workgroups then fit a CU (LDS): the phases of a pass (loads, dot MFMAs, per-point arithmetic, d F MFMAs,
// stores) do not overlap inside a wave - measured, they add - so a second wave per SIMD is what overlaps them.
// =====================================================================================================================
#ifndef HD_ABL
#define HD_ABL 0                                 // ablation builds only (scripts/build_ablation.sh): 1 no dot MFMAs, 2 no feature loads, 8 no contraction, 16 no d F MFMAs, 32 no d F stores
#endif
constexpr int HD_LDT = HD_C + 4;                 // table row stride (floats): 16-byte reads of 16 rows hit 16 different slots
constexpr int HD_SLAB_C = 32;                    // channels per slab
constexpr int HD_NSLAB = HD_C / HD_SLAB_C;
template <int NTL> struct HdT {
    static_assert(NTL == 2 || NTL == 4, "two or four point tiles per wave");
    typedef float V __attribute__((ext_vector_type(NTL)));      // one value per point tile of a lane
    static constexpr int PW = 16 * NTL;          // points per wave
    static constexpr int PTS = 4 * PW;           // points per workgroup
    static constexpr int LD = PW + 4;            // slab / coefficient-tile row stride (floats)
    static constexpr int SLAB = HD_SLAB_C * LD;  // floats of one wave's slab
    static constexpr int LPR = PW / 4;           // lanes per channel row of a 16-byte fetch
    static constexpr int RPI = 64 / LPR;         // channel rows per fetch instruction
    static constexpr int NF = HD_SLAB_C / RPI;   // fetches per slab and lane
};

template <int KT, int NTL> constexpr size_t hd_lds_fwd1()
{
    return sizeof(float) * ((size_t)KT * HD_LDT + 4 * HdT<NTL>::SLAB + 4 * 32 * HdT<NTL>::LD + 4 * 32 * 33 + 128);
}
template <int KT, int NTL> constexpr size_t hd_lds_fwd2() { return sizeof(float) * ((size_t)2 * KT * HD_LDT + 4 * HdT<NTL>::SLAB + 16); }
template <int KT, int NTL> constexpr size_t hd_lds_bwd1()
{
    return sizeof(float) * ((size_t)2 * KT * HD_LDT + 4 * HdT<NTL>::SLAB + 4 * 2 * KT * HdT<NTL>::LD + 4 * 2 * KT * 33);
}
template <int KT, int NTL> constexpr size_t hd_lds_bwd2()
{
    return sizeof(float) * ((size_t)(4 * KT + GFS3D_HEAD_MAX_FAKE) * HD_LDT + 4 * HdT<NTL>::SLAB + 4 * 16 * ((KT + 15) / 16) * (HdT<NTL>::LD + 33));
}
// workgroups per CU the LDS allows (the register budget of __launch_bounds__ follows it)
constexpr int hd_occ(size_t lds) { return 2 * lds <= 160 * 1024 ? 2 : 1; }

__device__ __forceinline__ float hd_gsum(float v) { v += __shfl_xor(v, 16, 64); v += __shfl_xor(v, 32, 64); return v; }   // over g: the 4 lanes of a column
__device__ __forceinline__ float hd_gmax(float v) { v = fmaxf(v, __shfl_xor(v, 16, 64)); v = fmaxf(v, __shfl_xor(v, 32, 64)); return v; }
__device__ __forceinline__ f32x4 hd_mfma16(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

// rows [0, rows) of a (rows, 128) matrix -> LDS table of KT rows, HD_LDT floats apart (the rest zero); ends with a barrier
template <int KT>
__device__ __forceinline__ void hd_mm_load_table(float *dst, const float *src, int rows, int tid)
{
    for (int e = tid; e < KT * (HD_C / 4); e += 256) {
        const int row = e >> 5, c4 = e & 31;
        const f32x4 v = (src != nullptr && row < rows) ? hd_ldg4(src + (size_t)row * HD_C + 4 * c4) : f32x4{0.f, 0.f, 0.f, 0.f};
        *reinterpret_cast<f32x4 *>(dst + row * HD_LDT + 4 * c4) = v;
    }
    __syncthreads();
}
// unit([bg; main]) (bg == NULL: unit(main)) in the same layout; ends with a barrier
template <int KT>
__device__ __forceinline__ void hd_mm_load_bgmain_unit(float *dst, const float *bg, const float *main_proto, int K1, int tid)
{
    for (int e = tid; e < KT * (HD_C / 4); e += 256) {
        const int row = e >> 5, c4 = e & 31;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (bg == nullptr) { if (row < K1) v = hd_ldg4(main_proto + (size_t)row * HD_C + 4 * c4); }
        else if (row == 0) v = hd_ldg4(bg + 4 * c4);
        else if (row < K1) v = hd_ldg4(main_proto + (size_t)(row - 1) * HD_C + 4 * c4);
        *reinterpret_cast<f32x4 *>(dst + row * HD_LDT + 4 * c4) = v;
    }
    __syncthreads();
    const int lane = tid & 63, w = tid >> 6;
    for (int row = w; row < K1; row += 4) {
        const float a = dst[row * HD_LDT + lane], b = dst[row * HD_LDT + 64 + lane];
        const float inv = 1.0f / fmaxf(sqrtf(hd_wsum(fmaf(a, a, b * b))), HD_EPS);
        dst[row * HD_LDT + lane] = a * inv;
        dst[row * HD_LDT + 64 + lane] = b * inv;
    }
    __syncthreads();
}

// slab s of the wave's points: fetch i of lane (rr = lane / LPR, q = lane % LPR) holds channel 32 s + RPI i + rr, points
// n0 + 4 q .. + 3.  VEC: N % 4 == 0 and 16-byte aligned rows; a group past N reads the last four points (its coefficients are zero).
template <int NTL, bool VEC>
__device__ __forceinline__ void hd_slab_fetch(const float *fb, int N, int n0, int s, int lane, f32x4 (&r)[HdT<NTL>::NF])
{
    typedef HdT<NTL> T;
    const int q = lane % T::LPR, rr = lane / T::LPR;
#pragma unroll
    for (int i = 0; i < T::NF; ++i) {
        const float *row = fb + (size_t)(HD_SLAB_C * s + T::RPI * i + rr) * N;
        if (HD_ABL & 2) r[i] = f32x4{0.5f, 0.25f, 1.f, 2.f};
        else if (VEC) r[i] = hd_ldg4(row + min(n0 + 4 * q, N - 4));
        else {
#pragma unroll
            for (int e = 0; e < 4; ++e) r[i][e] = row[min(n0 + 4 * q + e, N - 1)];
        }
    }
}
template <int NTL>
__device__ __forceinline__ void hd_slab_put(float *slab, int lane, const f32x4 (&r)[HdT<NTL>::NF])
{
    typedef HdT<NTL> T;
    const int q = lane % T::LPR, rr = lane / T::LPR;
#pragma unroll
    for (int i = 0; i < T::NF; ++i) *reinterpret_cast<f32x4 *>(slab + (T::RPI * i + rr) * T::LD + 4 * q) = r[i];
}

// acc[tb][rt][t] += T_tb[class][c] F[c][point] over the 32 channels of the slab; sq[t] += this lane's share of |F|^2
// (K-step (u, e), k = g <-> channel 16 u + 4 g + e of the slab; table rows past KT - 1 repeat the last row: masked by k < K1)
template <int KT, int NT, int NTL>
__device__ __forceinline__ void hd_mm_dots(const float *slab, const float *const (&tab)[NT], int cbase, int g, int j,
                                           f32x4 (&acc)[NT][(KT + 15) / 16][NTL], float (&sq)[NTL])
{
    typedef HdT<NTL> T;
    constexpr int RT = (KT + 15) / 16;
    const int prow = 4 * (j & 3) + (j >> 2);
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        f32x4 a[NT][RT];
#pragma unroll
        for (int tb = 0; tb < NT; ++tb)
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
                a[tb][rt] = *reinterpret_cast<const f32x4 *>(tab[tb] + min(16 * rt + prow, KT - 1) * HD_LDT + cbase + 16 * u + 4 * g);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const typename T::V bv = *reinterpret_cast<const typename T::V *>(slab + (16 * u + 4 * g + e) * T::LD + NTL * j);
#pragma unroll
            for (int t = 0; t < NTL; ++t) sq[t] = fmaf(bv[t], bv[t], sq[t]);
#pragma unroll
            for (int tb = 0; tb < NT; ++tb)
#pragma unroll
                for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                    for (int t = 0; t < NTL; ++t)
                        if (!(HD_ABL & 1)) acc[tb][rt][t] = hd_mfma16(a[tb][rt][e], bv[t], acc[tb][rt][t]);
        }
    }
}

// The point contraction of one slab: D[row][channel] = sum over the wave's points of ct[row][p] slab[channel][p] for the
// 16 RTC rows of the wave's coefficient tile, the four waves' tiles added in wave order and stored:
// part[rowmap(row)][32 s + c].  Ends with a barrier (the slab and `red` are free again).
template <int RTC, int NTL, typename RowMap>
__device__ __forceinline__ void hd_mm_contract(const float *slab, const float *ct, float *red /* [4][16 RTC][33] */, int s, int tid,
                                               float *part, RowMap rowmap)
{
    typedef HdT<NTL> T;
    const int lane = tid & 63, w = tid >> 6, g = lane >> 4, j = lane & 15;
    f32x4 acc[RTC][2];
#pragma unroll
    for (int rt = 0; rt < RTC; ++rt)
#pragma unroll
        for (int m = 0; m < 2; ++m) acc[rt][m] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < NTL; ++u) {                  // K-step (u, e), k = g <-> point 16 u + 4 g + e of the wave
        f32x4 a[RTC], bm[2];
#pragma unroll
        for (int rt = 0; rt < RTC; ++rt) a[rt] = *reinterpret_cast<const f32x4 *>(ct + (16 * rt + j) * T::LD + 16 * u + 4 * g);
#pragma unroll
        for (int m = 0; m < 2; ++m) bm[m] = *reinterpret_cast<const f32x4 *>(slab + (16 * m + j) * T::LD + 16 * u + 4 * g);
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int rt = 0; rt < RTC; ++rt)
#pragma unroll
                for (int m = 0; m < 2; ++m)
                    if (!(HD_ABL & 8)) acc[rt][m] = hd_mfma16(a[rt][e], bm[m][e], acc[rt][m]);
    }
#pragma unroll
    for (int rt = 0; rt < RTC; ++rt)
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r) red[(w * 16 * RTC + 16 * rt + 4 * g + r) * 33 + 16 * m + j] = acc[rt][m][r];
    __syncthreads();
    for (int e = tid; e < 16 * RTC * 32; e += 256) {
        const int row = e >> 5, c = e & 31;
        const float v = ((red[row * 33 + c] + red[(16 * RTC + row) * 33 + c]) + red[(32 * RTC + row) * 33 + c]) + red[(48 * RTC + row) * 33 + c];
        part[(size_t)rowmap(row) * HD_C + 32 * s + c] = v;
    }
    __syncthreads();
}

// one value per point tile -> the lane's part of a coefficient row
template <int NTL, int RT>
__device__ __forceinline__ typename HdT<NTL>::V hd_row_of(const f32x4 (&a)[RT][NTL], int rt, int r)
{
    typename HdT<NTL>::V v;
#pragma unroll
    for (int t = 0; t < NTL; ++t) v[t] = a[rt][t][r];
    return v;
}

// ------------------------------------------------------------------------------------------------ forward, pass 1
// part1[(b nchunk + chunk)][32][128]: rows k < K1: sum_n e[k][n] F[c][n], e = exp(10 cos(F_n, unit([bg; main])[k]) - 10) (soft pooling);
//                                     rows K1 + f: sum_n [support half, y_n = fake class f] F[c][n] / (|F_n| + eps)
// sums1[(b nchunk + chunk)][32]:      rows k < K1: sum_n e[k][n];  rows K1 + f: the number of such points
template <int KT, int NTL, bool VEC>
__global__ void __launch_bounds__(256, hd_occ(hd_lds_fwd1<KT, NTL>()))
hd_fwd1_mm_kernel(const float *__restrict__ feat, long long bs, int N, const long long *__restrict__ label, const float *__restrict__ main_proto,
                  const float *__restrict__ bg, int K1, const HeadFake fk, float *__restrict__ part1, float *__restrict__ sums1)
{
    typedef HdT<NTL> T;
    typedef typename T::V V;
    constexpr int RT = (KT + 15) / 16;
    extern __shared__ __attribute__((aligned(16))) float hd_smem[];
    float *q3 = hd_smem;                                 // [KT][132]
    float *slabs = q3 + KT * HD_LDT;                     // [4][32][LD]
    float *cts = slabs + 4 * T::SLAB;                    // [4][32][LD]   rows k < K1: e;  rows K1 + f: the fake-novel weights
    float *red = cts + 4 * 32 * T::LD;                   // [4][32][33]
    float *wsums = red + 4 * 32 * 33;                    // [4][32]
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, g = lane >> 4, j = lane & 15;
    const int b = blockIdx.y, chunk = blockIdx.x, nchunk = gridDim.x;
    const int n0 = chunk * T::PTS + w * T::PW;
    const float *fb = feat + (size_t)b * bs;
    float *slab = slabs + w * T::SLAB, *ct = cts + w * 32 * T::LD;
    f32x4 rg[HD_NSLAB][T::NF];                           // the wave's whole 128-channel tile: every load in flight at once, kept for both sweeps
#pragma unroll
    for (int s = 0; s < HD_NSLAB; ++s) hd_slab_fetch<NTL, VEC>(fb, N, n0, s, lane, rg[s]);
    for (int e = lane; e < 32 * T::LD; e += 64) ct[e] = 0.f;
    if (lane < 32) wsums[w * 32 + lane] = 0.f;
    hd_mm_load_bgmain_unit<KT>(q3, bg, main_proto, K1, tid);
    f32x4 acc[1][RT][NTL];
    float sq[NTL];
#pragma unroll
    for (int t = 0; t < NTL; ++t) {
        sq[t] = 0.f;
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) acc[0][rt][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const float *const tabs[1] = {q3};
#pragma unroll
    for (int s = 0; s < HD_NSLAB; ++s) {
        hd_slab_put<NTL>(slab, lane, rg[s]);
        __builtin_amdgcn_wave_barrier();         // (the slab is the wave's own: LDS operations of one wave complete in order)
        hd_mm_dots<KT, 1, NTL>(slab, tabs, HD_SLAB_C * s, g, j, acc, sq);
        __builtin_amdgcn_wave_barrier();
    }
    float nrm[NTL], rn[NTL];
    bool valid[NTL];
#pragma unroll
    for (int t = 0; t < NTL; ++t) {
        nrm[t] = sqrtf(hd_gsum(sq[t]));
        rn[t] = 1.0f / fmaxf(nrm[t], HD_EPS);
        valid[t] = n0 + NTL * j + t < N;
    }
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int k = 16 * rt + 4 * r + g;
            V ev;
            float se = 0.f;
#pragma unroll
            for (int t = 0; t < NTL; ++t) {
                ev[t] = (valid[t] && k < K1) ? __expf(HD_SCALE * acc[0][rt][t][r] * rn[t] - HD_SHIFT) : 0.f;
                se += ev[t];
            }
            if (k < K1) *reinterpret_cast<V *>(ct + k * T::LD + NTL * j) = ev;
            se = row16_sum(se);
            if (j == 0 && k < K1) wsums[w * 32 + k] = se;
        }
    if (fk.n > 0 && b >= fk.half) {                      // (workgroup-uniform)
        int fi[NTL];
        float rn2[NTL];
#pragma unroll
        for (int t = 0; t < NTL; ++t) {
            const long long y = label[(size_t)b * N + min(n0 + NTL * j + t, N - 1)];
            fi[t] = valid[t] ? hd_fake_index(fk, y) : -1;
            rn2[t] = 1.0f / (nrm[t] + HD_EPS);           // generate_fake_proto divides by (|f| + 1e-12)
        }
        for (int f = g; f < fk.n; f += 4) {
            V v;
            float cnt = 0.f;
#pragma unroll
            for (int t = 0; t < NTL; ++t) { v[t] = fi[t] == f ? rn2[t] : 0.f; cnt += fi[t] == f ? 1.f : 0.f; }
            *reinterpret_cast<V *>(ct + (K1 + f) * T::LD + NTL * j) = v;
            cnt = row16_sum(cnt);
            if (j == 0) wsums[w * 32 + K1 + f] = cnt;
        }
    }
    float *part = part1 + (size_t)(b * nchunk + chunk) * 32 * HD_C;
#pragma unroll
    for (int s = 0; s < HD_NSLAB; ++s) {
        hd_slab_put<NTL>(slab, lane, rg[s]);
        __builtin_amdgcn_wave_barrier();
        hd_mm_contract<2, NTL>(slab, ct, red, s, tid, part, [](int row) { return row; });
    }
    if (tid < 32) sums1[(size_t)(b * nchunk + chunk) * 32 + tid] = ((wsums[tid] + wsums[32 + tid]) + wsums[64 + tid]) + wsums[96 + tid];
}

// The per-point soft-max of one head on the class registers of lane (g, j): L[rt][r] (class k = 16 rt + 4 r + g, rows >= K1
// excluded) -> probabilities in place; returns the log-sum-exp.  Same arithmetic as hd_softmax, the class reductions across
// the four lanes of the column.
template <int RT>
__device__ __forceinline__ float hd_mm_softmax(float (&L)[RT][4], int K1, int g)
{
    float mx = -__builtin_inff();
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int r = 0; r < 4; ++r) mx = (16 * rt + 4 * r + g) < K1 ? fmaxf(mx, L[rt][r]) : mx;
    mx = hd_gmax(mx);
    float s = 0.f;
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int r = 0; r < 4; ++r) { L[rt][r] = (16 * rt + 4 * r + g) < K1 ? expf(L[rt][r] - mx) : 0.f; s += L[rt][r]; }
    s = hd_gsum(s);
    const float inv = 1.0f / s;
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int r = 0; r < 4; ++r) L[rt][r] *= inv;
    return mx + logf(s);
}

// ------------------------------------------------------------------------------------------------ forward, pass 2
// loss partials part2[(b nchunk + chunk)][4] = {sum nll_1, sum nll_2, counted points, -}; pred (B, N) int64
template <int KT, int NTL, bool VEC>
__global__ void __launch_bounds__(256, hd_occ(hd_lds_fwd2<KT, NTL>()))
hd_fwd2_mm_kernel(const float *__restrict__ feat, long long bs, int N, const long long *__restrict__ label, long long ignore_index,
                  const float *__restrict__ q1g, const float *__restrict__ q2g, int K1, float *__restrict__ part2, long long *__restrict__ pred)
{
    typedef HdT<NTL> T;
    constexpr int RT = (KT + 15) / 16;
    extern __shared__ __attribute__((aligned(16))) float hd_smem[];
    float *q1 = hd_smem, *q2 = q1 + KT * HD_LDT;
    float *slabs = q2 + KT * HD_LDT;                     // [4][32][LD]
    float *wred = slabs + 4 * T::SLAB;                   // [4][4]
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, g = lane >> 4, j = lane & 15;
    const int b = blockIdx.y, chunk = blockIdx.x, nchunk = gridDim.x;
    const int n0 = chunk * T::PTS + w * T::PW;
    const float *fb = feat + (size_t)b * bs;
    float *slab = slabs + w * T::SLAB;
    f32x4 rg[HD_NSLAB][T::NF];
#pragma unroll
    for (int s = 0; s < HD_NSLAB; ++s) hd_slab_fetch<NTL, VEC>(fb, N, n0, s, lane, rg[s]);
    hd_mm_load_table<KT>(q1, q1g, K1, tid);
    hd_mm_load_table<KT>(q2, q2g + (size_t)b * KT * HD_C, K1, tid);
    f32x4 acc[2][RT][NTL];
    float sq[NTL];
#pragma unroll
    for (int t = 0; t < NTL; ++t) {
        sq[t] = 0.f;
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) acc[h][rt][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const float *const tabs[2] = {q1, q2};
#pragma unroll
    for (int s = 0; s < HD_NSLAB; ++s) {
        hd_slab_put<NTL>(slab, lane, rg[s]);
        __builtin_amdgcn_wave_barrier();
        hd_mm_dots<KT, 2, NTL>(slab, tabs, HD_SLAB_C * s, g, j, acc, sq);
        __builtin_amdgcn_wave_barrier();
    }
    float s1 = 0.f, s2 = 0.f, sc = 0.f;
#pragma unroll
    for (int t = 0; t < NTL; ++t) {
        const int n = n0 + NTL * j + t;
        const bool valid = n < N;
        const float rn = HD_SCALE / fmaxf(sqrtf(hd_gsum(sq[t])), HD_EPS);
        const long long y = label[(size_t)b * N + min(n, N - 1)];
        const bool ok = valid && y != ignore_index && y >= 0 && y < (long long)K1;
        float L1[RT][4], L2[RT][4];
        float pick1 = 0.f, pick2 = 0.f, best = -__builtin_inff();
        int bk = 0x7fffffff;
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int k = 16 * rt + 4 * r + g;
                L1[rt][r] = acc[0][rt][t][r] * rn;
                L2[rt][r] = acc[1][rt][t][r] * rn;
                pick1 = (long long)k == y ? L1[rt][r] : pick1;
                pick2 = (long long)k == y ? L2[rt][r] : pick2;
                if (k < K1 && L2[rt][r] > best) { best = L2[rt][r]; bk = k; }       // (ascending k within the lane)
            }
        pick1 = hd_gsum(pick1);                          // (one lane of the column holds class y, the others add 0)
        pick2 = hd_gsum(pick2);
#pragma unroll
        for (int o = 16; o <= 32; o <<= 1) {             // arg-max over the column's four lanes: larger value, then smaller class
            const float ov = __shfl_xor(best, o, 64);
            const int ok2 = __shfl_xor(bk, o, 64);
            if (ov > best || (ov == best && ok2 < bk)) { best = ov; bk = ok2; }
        }
        bk = bk == 0x7fffffff ? 0 : bk;                  // (every logit NaN: class 0, like the scan of the lane = point form)
        const float lse1 = hd_mm_softmax<RT>(L1, K1, g), lse2 = hd_mm_softmax<RT>(L2, K1, g);
        if (valid && g == 0) pred[(size_t)b * N + n] = (long long)bk;
        if (ok && g == 0) { s1 += lse1 - pick1; s2 += lse2 - pick2; sc += 1.f; }
    }
    s1 = hd_wsum(s1); s2 = hd_wsum(s2); sc = hd_wsum(sc);
    if (lane == 0) { wred[w * 4 + 0] = s1; wred[w * 4 + 1] = s2; wred[w * 4 + 2] = sc; }
    __syncthreads();
    if (tid < 3) part2[(size_t)(b * nchunk + chunk) * 4 + tid] = ((wred[tid] + wred[4 + tid]) + wred[8 + tid]) + wred[12 + tid];
}

// ------------------------------------------------------------------------------------------------ backward, pass 1
// g_t[k][n] = d loss / d (cos_t[k][n]) = 10 . 0.5 . dloss / count . (softmax_t - onehot); part3 rows sum_n g_t[k][n] Fhat[c][n]
// coefficient rows of the wave tile: head 1 at k, head 2 at KT + k (compact); part3 rows: head 1 at k, head 2 at R2 + k
template <int KT, int NTL, bool VEC>
__global__ void __launch_bounds__(256, hd_occ(hd_lds_bwd1<KT, NTL>()))
hd_bwd1_mm_kernel(const float *__restrict__ feat, long long bs, int N, const long long *__restrict__ label, long long ignore_index,
                  const float *__restrict__ q1g, const float *__restrict__ q2g, int K1, const float *__restrict__ loss,
                  const float *__restrict__ dloss, float *__restrict__ part3)
{
    typedef HdT<NTL> T;
    typedef typename T::V V;
    constexpr int RT = (KT + 15) / 16, R2 = 2 * KT <= 32 ? KT : 32, NRW = 2 * KT, RTC = NRW / 16;
    static_assert(NRW % 16 == 0, "coefficient rows in whole tiles");
    extern __shared__ __attribute__((aligned(16))) float hd_smem[];
    float *q1 = hd_smem, *q2 = q1 + KT * HD_LDT;
    float *slabs = q2 + KT * HD_LDT;                     // [4][32][LD]
    float *cts = slabs + 4 * T::SLAB;                    // [4][NRW][LD]
    float *red = cts + 4 * NRW * T::LD;                  // [4][NRW][33]
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, g = lane >> 4, j = lane & 15;
    const int b = blockIdx.y, chunk = blockIdx.x, nchunk = gridDim.x;
    const int n0 = chunk * T::PTS + w * T::PW;
    const float *fb = feat + (size_t)b * bs;
    float *slab = slabs + w * T::SLAB, *ct = cts + w * NRW * T::LD;
    f32x4 rg[HD_NSLAB][T::NF];
#pragma unroll
    for (int s = 0; s < HD_NSLAB; ++s) hd_slab_fetch<NTL, VEC>(fb, N, n0, s, lane, rg[s]);
    for (int e = lane; e < NRW * T::LD; e += 64) ct[e] = 0.f;
    hd_mm_load_table<KT>(q1, q1g, K1, tid);
    hd_mm_load_table<KT>(q2, q2g + (size_t)b * KT * HD_C, K1, tid);
    f32x4 acc[2][RT][NTL];
    float sq[NTL];
#pragma unroll
    for (int t = 0; t < NTL; ++t) {
        sq[t] = 0.f;
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) acc[h][rt][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const float *const tabs[2] = {q1, q2};
#pragma unroll
    for (int s = 0; s < HD_NSLAB; ++s) {
        hd_slab_put<NTL>(slab, lane, rg[s]);
        __builtin_amdgcn_wave_barrier();
        hd_mm_dots<KT, 2, NTL>(slab, tabs, HD_SLAB_C * s, g, j, acc, sq);
        __builtin_amdgcn_wave_barrier();
    }
    const float cbase = HD_SCALE * 0.5f * dloss[0] / loss[1];
#pragma unroll
    for (int t = 0; t < NTL; ++t) {
        const int n = n0 + NTL * j + t;
        const bool valid = n < N;
        const float rinv = 1.0f / fmaxf(sqrtf(hd_gsum(sq[t])), HD_EPS);
        const long long y = label[(size_t)b * N + min(n, N - 1)];
        const bool ok = valid && y != ignore_index && y >= 0 && y < (long long)K1;
        const float coef = ok ? cbase : 0.f;
        float L1[RT][4], L2[RT][4];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int r = 0; r < 4; ++r) { L1[rt][r] = acc[0][rt][t][r] * (HD_SCALE * rinv); L2[rt][r] = acc[1][rt][t][r] * (HD_SCALE * rinv); }
        hd_mm_softmax<RT>(L1, K1, g);
        hd_mm_softmax<RT>(L2, K1, g);
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float oh = (long long)(16 * rt + 4 * r + g) == y ? 1.f : 0.f;
                acc[0][rt][t][r] = coef * (L1[rt][r] - oh) * rinv;       // (in place; Fhat = F / |F|: the 1 / |F| rides in the A operand)
                acc[1][rt][t][r] = coef * (L2[rt][r] - oh) * rinv;
            }
    }
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int k = 16 * rt + 4 * r + g;
            if (k < K1) {
                *reinterpret_cast<V *>(ct + k * T::LD + NTL * j) = hd_row_of<NTL, RT>(acc[0], rt, r);
                *reinterpret_cast<V *>(ct + (KT + k) * T::LD + NTL * j) = hd_row_of<NTL, RT>(acc[1], rt, r);
            }
        }
    float *part = part3 + (size_t)(b * nchunk + chunk) * 64 * HD_C;
#pragma unroll
    for (int s = 0; s < HD_NSLAB; ++s) {
        hd_slab_put<NTL>(slab, lane, rg[s]);
        __builtin_amdgcn_wave_barrier();
        hd_mm_contract<RTC, NTL>(slab, ct, red, s, tid, part, [](int row) { return row < KT ? row : R2 + (row - KT); });
    }
}

// ------------------------------------------------------------------------------------------------ backward, pass 2
// d F from all four paths, written once; part4 rows k: sum_n gl3[k][n] Fhat[c][n]  (gl3 = the gradient of cos(F, [bg; main]))
template <int KT, int NTL, bool VEC>
__global__ void __launch_bounds__(256, hd_occ(hd_lds_bwd2<KT, NTL>()))
hd_bwd2_mm_kernel(const float *__restrict__ feat, long long bs, int N, const long long *__restrict__ label, long long ignore_index,
                  const float *__restrict__ q1g, const float *__restrict__ q2g, const float *__restrict__ main_proto, const float *__restrict__ bg,
                  int K1, const HeadFake fk, const float *__restrict__ loss, const float *__restrict__ dloss, const float *__restrict__ hs,
                  const float *__restrict__ dpoolg, const float *__restrict__ t3g, const float *__restrict__ dfsum,
                  float *__restrict__ dfeat, float *__restrict__ part4)
{
    typedef HdT<NTL> T;
    typedef typename T::V V;
    constexpr int RT = (KT + 15) / 16, FT = GFS3D_HEAD_MAX_FAKE;
    static_assert(FT == 16, "the fake-novel rows are one 16-row tile");
    constexpr bool HOLD = RT == 1 || NTL == 2;           // two row tiles of four point tiles: 288 coefficient / accumulator registers - one slab ahead, each sweep fetches the tile
    constexpr int NB = HOLD ? HD_NSLAB : 1;
    extern __shared__ __attribute__((aligned(16))) float hd_smem[];
    float *q1 = hd_smem, *q2 = q1 + KT * HD_LDT, *q3 = q2 + KT * HD_LDT, *dpl = q3 + KT * HD_LDT;     // four [KT][132] tables
    float *dfs = dpl + KT * HD_LDT;                      // [16][132]  d fsum rows (support half only)
    float *slabs = dfs + FT * HD_LDT;                    // [4][32][LD]
    float *cts = slabs + 4 * T::SLAB;                    // [4][16 RT][LD]
    float *red = cts + 4 * 16 * RT * T::LD;              // [4][16 RT][33]
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, g = lane >> 4, j = lane & 15;
    const int b = blockIdx.y, chunk = blockIdx.x, nchunk = gridDim.x;
    const int n0 = chunk * T::PTS + w * T::PW;
    const float *fb = feat + (size_t)b * bs;
    float *slab = slabs + w * T::SLAB, *ct = cts + w * 16 * RT * T::LD;
    const bool support = fk.n > 0 && b >= fk.half;       // (workgroup-uniform)
    f32x4 rg[NB][T::NF];                                 // HOLD: the wave's whole 128-channel tile, every load in flight at once, kept for both sweeps
#pragma unroll
    for (int s = 0; s < NB; ++s) hd_slab_fetch<NTL, VEC>(fb, N, n0, s, lane, rg[s]);
    for (int e = lane; e < 16 * RT * T::LD; e += 64) ct[e] = 0.f;
    for (int e = tid; e < FT * (HD_C / 4); e += 256) {   // d fsum rows: [fake class][128 | count] -> [f][132]
        const int row = e >> 5, c4 = e & 31;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (support && row < fk.n) {
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = dfsum[row * HD_FLD + 4 * c4 + i];
        }
        *reinterpret_cast<f32x4 *>(dfs + row * HD_LDT + 4 * c4) = v;
    }
    hd_mm_load_table<KT>(q1, q1g, K1, tid);
    hd_mm_load_table<KT>(q2, q2g + (size_t)b * KT * HD_C, K1, tid);
    hd_mm_load_table<KT>(dpl, dpoolg + (size_t)b * KT * HD_C, K1, tid);
    hd_mm_load_bgmain_unit<KT>(q3, bg, main_proto, K1, tid);
    f32x4 acc[4][RT][NTL], accf[1][1][NTL];
    float sq[NTL], sqf[NTL];
#pragma unroll
    for (int t = 0; t < NTL; ++t) {
        sq[t] = 0.f; sqf[t] = 0.f;
        accf[0][0][t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int tb = 0; tb < 4; ++tb)
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) acc[tb][rt][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const float *const tabs[4] = {q1, q2, q3, dpl};
    const float *const tabf[1] = {dfs};
#pragma unroll 1
    for (int s2 = 0; s2 < HD_NSLAB / NB; ++s2)
#pragma unroll
    for (int h = 0; h < NB; ++h) {
        const int s = s2 * NB + h;
        hd_slab_put<NTL>(slab, lane, rg[h]);
        __builtin_amdgcn_wave_barrier();         // (the slab is the wave's own: LDS operations of one wave complete in order)
        if (!HOLD && s + NB < HD_NSLAB) hd_slab_fetch<NTL, VEC>(fb, N, n0, s + NB, lane, rg[h]);
        hd_mm_dots<KT, 4, NTL>(slab, tabs, HD_SLAB_C * s, g, j, acc, sq);
        if (support) hd_mm_dots<16, 1, NTL>(slab, tabf, HD_SLAB_C * s, g, j, accf, sqf);       // <F_n, d fsum[f]>
        __builtin_amdgcn_wave_barrier();
    }
    if (!HOLD) hd_slab_fetch<NTL, VEC>(fb, N, n0, 0, lane, rg[0]);      // the second sweep's first slab: in flight during the per-point arithmetic
    // per point: the coefficient rows of d F (the B operands of its K-steps) take the place of the dot products they are made of -
    // acc[table][rt][t][r] -, and the scalar of the -F term
    const float cbase = HD_SCALE * 0.5f * dloss[0] / loss[1];
    float scl[NTL];
#pragma unroll
    for (int t = 0; t < NTL; ++t) {
        const int n = n0 + NTL * j + t;
        const bool valid = n < N;
        const float nrm = sqrtf(hd_gsum(sq[t]));
        const float rinv = 1.0f / fmaxf(nrm, HD_EPS);
        const long long y = label[(size_t)b * N + min(n, N - 1)];
        const bool ok = valid && y != ignore_index && y >= 0 && y < (long long)K1;
        const float coef = ok ? cbase : 0.f;
        float ca[RT][4], cb[RT][4], L1[RT][4], L2[RT][4];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                ca[rt][r] = acc[0][rt][t][r] * rinv; cb[rt][r] = acc[1][rt][t][r] * rinv;
                L1[rt][r] = ca[rt][r] * HD_SCALE; L2[rt][r] = cb[rt][r] * HD_SCALE;
            }
        hd_mm_softmax<RT>(L1, K1, g);
        hd_mm_softmax<RT>(L2, K1, g);
        float tsum = 0.f;                                // <Fhat, dFhat> = sum over the three cosine paths of g[k] cos[k]
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int k = 16 * rt + 4 * r + g;
                const float oh = (long long)k == y ? 1.f : 0.f;
                const float g1 = coef * (L1[rt][r] - oh), g2 = coef * (L2[rt][r] - oh);
                tsum = fmaf(g1, ca[rt][r], fmaf(g2, cb[rt][r], tsum));
                const bool in = k >= 1 && k < K1;
                const float skv = in ? 1.0f / hs[((size_t)b * KT + k) * HD_HS] : 0.f;
                const float tkv = k < K1 ? t3g[(size_t)b * KT + k] : 0.f;
                const float c3 = acc[2][rt][t][r] * rinv;
                const float resp = (valid && in) ? __expf(HD_SCALE * c3 - HD_SHIFT) * skv : 0.f;      // softmax over the points
                const float gl = HD_SCALE * resp * (acc[3][rt][t][r] - tkv);                         // acc[3] = <dpool[k], F_n> (raw features)
                tsum = fmaf(gl, c3, tsum);
                acc[0][rt][t][r] = g1 * rinv;
                acc[1][rt][t][r] = g2 * rinv;
                acc[2][rt][t][r] = gl * rinv;
                acc[3][rt][t][r] = resp;
            }
        tsum = hd_gsum(tsum);
        // the fake-novel mean path: this point contributes F / (|F| + eps) to fsum[fi]
        float fa1 = 0.f, fa2 = 0.f;
        int fi = -1;
        if (support) {                                   // (uniform)
            fi = valid ? hd_fake_index(fk, y) : -1;
            float tf = 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) tf = (4 * r + g) == fi ? accf[0][0][t][r] : tf;
            tf = hd_gsum(tf);
            const float re = nrm + HD_EPS;
            fa1 = fi >= 0 ? 1.0f / re : 0.f;
            fa2 = fi >= 0 ? tf / (fmaxf(nrm, 1e-30f) * re * re) : 0.f;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) accf[0][0][t][r] = (4 * r + g) == fi ? fa1 : 0.f;
        scl[t] = fmaf(tsum * rinv, rinv, fa2);           // dF = (dFhat - Fhat <Fhat, dFhat>) / |F| + ..., Fhat = F / |F|
    }
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int k = 16 * rt + 4 * r + g;
            if (k < K1) *reinterpret_cast<V *>(ct + k * T::LD + NTL * j) = hd_row_of<NTL, RT>(acc[2], rt, r);
        }
    float *part = part4 + (size_t)(b * nchunk + chunk) * 32 * HD_C;
    float *dfb = dfeat + (size_t)b * HD_C * N;
#pragma unroll 1
    for (int s2 = 0; s2 < HD_NSLAB / NB; ++s2)
#pragma unroll
    for (int h = 0; h < NB; ++h) {
        const int s = s2 * NB + h;
        hd_slab_put<NTL>(slab, lane, rg[h]);
        __builtin_amdgcn_wave_barrier();
        if (!HOLD && s + NB < HD_NSLAB) hd_slab_fetch<NTL, VEC>(fb, N, n0, s + NB, lane, rg[h]);
        // d F of the slab's channels: tile m (0, 1), row i = 4 g + r of column j <-> channel 32 s + 2 i + m, point NTL j + t
        f32x4 da[2][NTL];
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int t = 0; t < NTL; ++t) da[m][t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int tb = 0; tb < 4; ++tb)
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {            // K-step: k = g <-> class 16 rt + 4 r + g of table tb
                    const float *trow = tabs[tb] + min(16 * rt + 4 * r + g, KT - 1) * HD_LDT + HD_SLAB_C * s + 2 * j;
                    const float a0 = trow[0], a1 = trow[1];
#pragma unroll
                    for (int t = 0; t < NTL; ++t) {
                        if (HD_ABL & 16) continue;
                        da[0][t] = hd_mfma16(a0, acc[tb][rt][t][r], da[0][t]);
                        da[1][t] = hd_mfma16(a1, acc[tb][rt][t][r], da[1][t]);
                    }
                }
        if (support) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float *trow = dfs + (4 * r + g) * HD_LDT + HD_SLAB_C * s + 2 * j;
                const float a0 = trow[0], a1 = trow[1];
#pragma unroll
                for (int t = 0; t < NTL; ++t) {
                    da[0][t] = hd_mfma16(a0, accf[0][0][t][r], da[0][t]);
                    da[1][t] = hd_mfma16(a1, accf[0][0][t][r], da[1][t]);
                }
            }
        }
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int cl = 2 * (4 * g + r) + m;
                const V v = *reinterpret_cast<const V *>(slab + cl * T::LD + NTL * j);
                V o;
#pragma unroll
                for (int t = 0; t < NTL; ++t) o[t] = fmaf(-v[t], scl[t], da[m][t][r]);
                float *dst = dfb + (size_t)(HD_SLAB_C * s + cl) * N + n0 + NTL * j;
                if (HD_ABL & 32) { if (o[0] == 1.2345f) dst[0] = o[1]; }
                else if (VEC) { if (n0 + NTL * j < N) *reinterpret_cast<V *>(dst) = o; }
                else {
#pragma unroll
                    for (int t = 0; t < NTL; ++t)
                        if (n0 + NTL * j + t < N) dst[t] = o[t];
                }
            }
        hd_mm_contract<RT, NTL>(slab, ct, red, s, tid, part, [](int row) { return row; });
    }
}


// Fixed-order sum of `count` partial rows, `stride` floats apart, by a block of 1024 threads: thread (g, c) adds the rows
// p = g, g + 8, ... of channel c (independent loads, eight in flight), the eight group sums are then added in group order.
// Returns the total in the threads of group 0 (c = tid & 127); ends with a barrier.
__device__ __forceinline__ float hd_sum_rows(const float *base, size_t stride, int count, float (*sh)[HD_C], int tid)
{
    const int c = tid & 127, g = tid >> 7;
    float s = 0.f;
#pragma unroll 8
    for (int p = g; p < count; p += 8) s += base[(size_t)p * stride + c];
    __syncthreads();                                     // (sh may still be read from a previous call)
    sh[g][c] = s;
    __syncthreads();
    float t = 0.f;
    if (g == 0)
#pragma unroll
        for (int q = 0; q < 8; ++q) t += sh[q][c];
    return t;
}

// fsum[f][c] = sum over the support half's partial tiles (fixed order); fsum[f][128] = the point count.  grid = n_fake
__global__ void __launch_bounds__(1024)
hd_mid1_kernel(const float *__restrict__ part1, const float *__restrict__ sums1, int B, int nchunk, int K1, int half, float *__restrict__ fsum)
{
    __shared__ float sh[8][HD_C];
    const int f = blockIdx.x, tid = threadIdx.x;
    const int count = (B - half) * nchunk;
    const float s = hd_sum_rows(part1 + ((size_t)half * nchunk * 32 + K1 + f) * HD_C, (size_t)32 * HD_C, count, sh, tid);
    if (tid < HD_C) fsum[f * HD_FLD + tid] = s;
    // the point count: the partial counts are small integers in fp32, any summation order gives the same (exact) result
    float cnt = 0.f;
    for (int p = tid; p < count; p += 1024) cnt += sums1[((size_t)half * nchunk + p) * 32 + K1 + f];
    cnt = hd_wsum(cnt);
    __syncthreads();
    if ((tid & 63) == 0) sh[0][tid >> 6] = cnt;
    __syncthreads();
    if (tid == 0) {
        float t = 0.f;
        for (int q = 0; q < 16; ++q) t += sh[0][q];
        fsum[f * HD_FLD + HD_C] = t;
    }
}

// grid = (B, ceil(K1 / 4)), 4 waves = 4 rows k; a lane owns channels lane and lane + 64
template <int KT>
__global__ void __launch_bounds__(256)
hd_mid2_kernel(const float *__restrict__ part1, const float *__restrict__ sums1, int nchunk, const float *__restrict__ main_proto,
               const float *__restrict__ bg, int K1, int base_num, const HeadFake fk, const float *__restrict__ fsum,
               float *__restrict__ q1, float *__restrict__ q2, float *__restrict__ ori_s, float *__restrict__ onrm,
               float *__restrict__ pooled_s, float *__restrict__ hs)
{
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, b = blockIdx.x;
    {
        const int k = blockIdx.y * 4 + w;                // one wave per row
        if (k >= K1) return;
        float *q2r = q2 + ((size_t)b * KT + k) * HD_C;
        if (k == 0) {
            const float v0 = bg[lane], v1 = bg[64 + lane];
            const float inv = 1.0f / fmaxf(sqrtf(hd_wsum(fmaf(v0, v0, v1 * v1))), HD_EPS);
            q2r[lane] = v0 * inv; q2r[64 + lane] = v1 * inv;
            if (b == 0) { q1[lane] = v0 * inv; q1[64 + lane] = v1 * inv; }
            return;
        }
        const int j = k - 1;
        const float P0 = main_proto[(size_t)j * HD_C + lane], P1 = main_proto[(size_t)j * HD_C + 64 + lane];
        float u0 = 0.f, u1 = 0.f, s = 0.f;
#pragma unroll 8
        for (int ch = 0; ch < nchunk; ++ch) {
            const size_t p = (size_t)b * nchunk + ch;
            u0 += part1[(p * 32 + k) * HD_C + lane];
            u1 += part1[(p * 32 + k) * HD_C + 64 + lane];
            s += sums1[p * 32 + k];
        }
        const float is = 1.0f / s;
        const float p0 = u0 * is, p1 = u1 * is;                          // pooled = softmax-over-points . F^T
        const float Pr = sqrtf(hd_wsum(fmaf(P0, P0, P1 * P1)));
        const float pn = fmaxf(sqrtf(hd_wsum(fmaf(p0, p0, p1 * p1))), HD_EPS), Pn = fmaxf(Pr, HD_EPS);
        const float cosw = hd_wsum(fmaf(p0 / pn, P0 / Pn, (p1 / pn) * (P1 / Pn)));
        const float wv = cosw > 0.f ? cosw : 0.f;
        const float r0 = fmaf(wv, p0, (1.0f - wv) * P0), r1 = fmaf(wv, p1, (1.0f - wv) * P1);
        // ori row: unit(main) or, for a fake-novel class, the masked mean of the unit features
        const int fi = hd_fake_index(fk, (long long)(j + 1));
        float o0, o1;
        if (fi >= 0) {
            const float den = fsum[fi * HD_FLD + HD_C] + HD_EPS;
            o0 = fsum[fi * HD_FLD + lane] / den; o1 = fsum[fi * HD_FLD + 64 + lane] / den;
        } else {
            o0 = P0 / (Pr + HD_EPS); o1 = P1 / (Pr + HD_EPS);
        }
        const float f0 = j < base_num ? r0 + o0 : o0, f1 = j < base_num ? r1 + o1 : o1;
        const float fn = fmaxf(sqrtf(hd_wsum(fmaf(f0, f0, f1 * f1))), HD_EPS);
        q2r[lane] = f0 / fn; q2r[64 + lane] = f1 / fn;
        float *ps = pooled_s + ((size_t)b * KT + k) * HD_C;
        ps[lane] = p0; ps[64 + lane] = p1;
        if (lane == 0) {
            float *hr = hs + ((size_t)b * KT + k) * HD_HS;
            hr[0] = s; hr[1] = wv; hr[2] = cosw; hr[3] = pn; hr[4] = fn;
        }
        if (b == 0) {
            const float on = fmaxf(sqrtf(hd_wsum(fmaf(o0, o0, o1 * o1))), HD_EPS);
            q1[(size_t)k * HD_C + lane] = o0 / on; q1[(size_t)k * HD_C + 64 + lane] = o1 / on;
            ori_s[(size_t)j * HD_C + lane] = o0; ori_s[(size_t)j * HD_C + 64 + lane] = o1;
            if (lane == 0) onrm[j] = on;
        }
    }
}


// loss[0] = (mean nll_1 + mean nll_2) / 2 over the counted points, loss[1] = their number   (one wave)
__global__ void __launch_bounds__(64)
hd_loss_kernel(const float *__restrict__ part2, int nparts, float *__restrict__ loss)
{
    double s1 = 0.0, s2 = 0.0, sc = 0.0;
    for (int p = threadIdx.x; p < nparts; p += 64) { s1 += part2[(size_t)p * 4]; s2 += part2[(size_t)p * 4 + 1]; sc += part2[(size_t)p * 4 + 2]; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { s1 += __shfl_xor(s1, o, 64); s2 += __shfl_xor(s2, o, 64); sc += __shfl_xor(sc, o, 64); }
    if (threadIdx.x == 0) { loss[0] = (float)(0.5 * (s1 + s2) / sc); loss[1] = (float)sc; }
}


// grid = (B, ceil(K1 / 4)); wave per row k.  Back through q2 = unit([bg; refined_b]) and the refinement, per batch item:
//   dbg_part[b], dori_part[b][j], dP_part[b][j], dpool[b][k] (the gradient of pooled, row 0 zero), t3[b][k] = <dpool, pooled>
template <int KT>
__global__ void __launch_bounds__(256)
hd_bwdm1_kernel(const float *__restrict__ part3, int nchunk, const float *__restrict__ main_proto, const float *__restrict__ bg, int K1,
                int base_num, const float *__restrict__ q2, const float *__restrict__ pooled_s, const float *__restrict__ hs,
                float *__restrict__ dbg_part, float *__restrict__ dori_part, float *__restrict__ dP_part, float *__restrict__ dpool,
                float *__restrict__ t3)
{
    constexpr int R2 = 2 * KT <= 32 ? KT : 32;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, b = blockIdx.x;
    {
        const int k = blockIdx.y * 4 + w;                // one wave per row
        if (k >= K1) return;
        float g0 = 0.f, g1 = 0.f;                        // d loss / d q2[b][k]
#pragma unroll 8
        for (int ch = 0; ch < nchunk; ++ch) {
            const float *pr = part3 + (((size_t)b * nchunk + ch) * 64 + R2 + k) * HD_C;
            g0 += pr[lane]; g1 += pr[64 + lane];
        }
        float *dpr = dpool + ((size_t)b * KT + k) * HD_C;
        if (k == 0) {
            const float v0 = bg[lane], v1 = bg[64 + lane];
            const float nr = fmaxf(sqrtf(hd_wsum(fmaf(v0, v0, v1 * v1))), HD_EPS);
            const float q0 = v0 / nr, q1v = v1 / nr;
            const float dt = hd_wsum(fmaf(q0, g0, q1v * g1));
            dbg_part[(size_t)b * HD_C + lane] = (g0 - q0 * dt) / nr;
            dbg_part[(size_t)b * HD_C + 64 + lane] = (g1 - q1v * dt) / nr;
            dpr[lane] = 0.f; dpr[64 + lane] = 0.f;
            if (lane == 0) t3[(size_t)b * KT] = 0.f;
            return;
        }
        const int j = k - 1;
        const float *hr = hs + ((size_t)b * KT + k) * HD_HS;
        const float wv = hr[1], cosw = hr[2], pn = hr[3], fn = hr[4];
        const float *qr = q2 + ((size_t)b * KT + k) * HD_C;
        const float q0 = qr[lane], q1v = qr[64 + lane];
        const float dt = hd_wsum(fmaf(q0, g0, q1v * g1));
        const float dR0 = (g0 - q0 * dt) / fn, dR1 = (g1 - q1v * dt) / fn;            // gradient of refined[b][j]
        dori_part[((size_t)b * 32 + j) * HD_C + lane] = dR0;
        dori_part[((size_t)b * 32 + j) * HD_C + 64 + lane] = dR1;
        const float r0 = j < base_num ? dR0 : 0.f, r1 = j < base_num ? dR1 : 0.f;     // gradient of ref0[b][j]
        const float P0 = main_proto[(size_t)j * HD_C + lane], P1 = main_proto[(size_t)j * HD_C + 64 + lane];
        const float p0 = pooled_s[((size_t)b * KT + k) * HD_C + lane], p1 = pooled_s[((size_t)b * KT + k) * HD_C + 64 + lane];
        const float Pn = fmaxf(sqrtf(hd_wsum(fmaf(P0, P0, P1 * P1))), HD_EPS);
        const float dw = hd_wsum(fmaf(r0, p0 - P0, r1 * (p1 - P1)));
        const float dc = cosw > 0.f ? dw : 0.f;
        const float ph0 = p0 / pn, ph1 = p1 / pn, Ph0 = P0 / Pn, Ph1 = P1 / Pn;
        const float a = hd_wsum(fmaf(ph0, Ph0, ph1 * Ph1));                            // <phat, Phat>
        // d pooled = w dref0 + dc (Phat - phat <phat, Phat>) / |pooled|;  d main += (1 - w) dref0 + dc (phat - Phat <Phat, phat>) / |main|
        const float dp0 = fmaf(wv, r0, dc * (Ph0 - ph0 * a) / pn), dp1 = fmaf(wv, r1, dc * (Ph1 - ph1 * a) / pn);
        dP_part[((size_t)b * 32 + j) * HD_C + lane] = fmaf(1.0f - wv, r0, dc * (ph0 - Ph0 * a) / Pn);
        dP_part[((size_t)b * 32 + j) * HD_C + 64 + lane] = fmaf(1.0f - wv, r1, dc * (ph1 - Ph1 * a) / Pn);
        dpr[lane] = dp0; dpr[64 + lane] = dp1;
        const float tt = hd_wsum(fmaf(dp0, p0, dp1 * p1));
        if (lane == 0) t3[(size_t)b * KT + k] = tt;
    }
}

// grid = K1 (row k of the shared table q1 = unit([bg; ori])), 1024 threads: the sum of ALL partial tiles in a fixed order, then
//   k = 0: dbg_a = the bg gradient through q1 + sum_b dbg_part;   k >= 1: d ori[j] -> dfsum (fake class) or d main through unit()
__global__ void __launch_bounds__(1024)
hd_bwdm2_kernel(const float *__restrict__ part3, int B, int nchunk, const float *__restrict__ main_proto, const float *__restrict__ bg,
                const HeadFake fk, const float *__restrict__ fsum, const float *__restrict__ ori_s, const float *__restrict__ onrm,
                const float *__restrict__ dbg_part, const float *__restrict__ dori_part, const float *__restrict__ dP_part,
                float *__restrict__ dbg_a, float *__restrict__ dmain_a, float *__restrict__ dfsum)
{
    __shared__ float shr[8][HD_C];
    __shared__ float sh[3][HD_C];
    const int tid = threadIdx.x, k = blockIdx.x;
    const float s = hd_sum_rows(part3 + (size_t)k * HD_C, (size_t)64 * HD_C, B * nchunk, shr, tid);
    const float so = k == 0 ? hd_sum_rows(dbg_part, (size_t)HD_C, B, shr, tid)
                            : hd_sum_rows(dori_part + (size_t)(k - 1) * HD_C, (size_t)32 * HD_C, B, shr, tid);
    const float sp = k == 0 ? 0.f : hd_sum_rows(dP_part + (size_t)(k - 1) * HD_C, (size_t)32 * HD_C, B, shr, tid);
    if (tid < HD_C) { sh[0][tid] = s; sh[1][tid] = so; sh[2][tid] = sp; }
    __syncthreads();
    if (tid >= 64) return;                               // wave 0 finishes the row: channels lane and lane + 64
    const int lane = tid;
    const float g0 = sh[0][lane], g1 = sh[0][64 + lane], o0 = sh[1][lane], o1 = sh[1][64 + lane];
    if (k == 0) {
        const float v0 = bg[lane], v1 = bg[64 + lane];
        const float nr = fmaxf(sqrtf(hd_wsum(fmaf(v0, v0, v1 * v1))), HD_EPS);
        const float q0 = v0 / nr, q1v = v1 / nr;
        const float dt = hd_wsum(fmaf(q0, g0, q1v * g1));
        dbg_a[lane] = (g0 - q0 * dt) / nr + o0;
        dbg_a[64 + lane] = (g1 - q1v * dt) / nr + o1;
        return;
    }
    const int j = k - 1;
    const float on = onrm[j];
    const float q0 = ori_s[(size_t)j * HD_C + lane] / on, q1v = ori_s[(size_t)j * HD_C + 64 + lane] / on;
    const float dt = hd_wsum(fmaf(q0, g0, q1v * g1));
    const float dO0 = (g0 - q0 * dt) / on + o0, dO1 = (g1 - q1v * dt) / on + o1;          // gradient of ori[j]
    const int fi = hd_fake_index(fk, (long long)(j + 1));
    float m0 = sh[2][lane], m1 = sh[2][64 + lane];
    if (fi >= 0) {                                       // ori[j] = fsum[fi] / (count + eps)
        const float den = fsum[fi * HD_FLD + HD_C] + HD_EPS;
        dfsum[fi * HD_FLD + lane] = dO0 / den;
        dfsum[fi * HD_FLD + 64 + lane] = dO1 / den;
        if (lane == 0) dfsum[fi * HD_FLD + HD_C] = 0.f;
    } else {                                             // ori[j] = P / (|P| + eps)
        const float P0 = main_proto[(size_t)j * HD_C + lane], P1 = main_proto[(size_t)j * HD_C + 64 + lane];
        const float r = sqrtf(hd_wsum(fmaf(P0, P0, P1 * P1))), re = r + HD_EPS;
        const float pd = hd_wsum(fmaf(P0, dO0, P1 * dO1)) / (fmaxf(r, 1e-30f) * re * re);
        m0 += dO0 / re - P0 * pd;
        m1 += dO1 / re - P1 * pd;
    }
    dmain_a[(size_t)j * HD_C + lane] = m0;
    dmain_a[(size_t)j * HD_C + 64 + lane] = m1;
}


// grid = K1: d unit([bg; main])[k] summed over all partial tiles, back through the normalisation, added to the other paths
__global__ void __launch_bounds__(1024)
hd_bwdfin_kernel(const float *__restrict__ part4, int nparts, const float *__restrict__ main_proto, const float *__restrict__ bg,
                 const float *__restrict__ dbg_a, const float *__restrict__ dmain_a, float *__restrict__ dbg, float *__restrict__ dmain)
{
    __shared__ float shr[8][HD_C];
    __shared__ float sh[HD_C];
    const int tid = threadIdx.x, k = blockIdx.x;
    const float s = hd_sum_rows(part4 + (size_t)k * HD_C, (size_t)32 * HD_C, nparts, shr, tid);
    if (tid < HD_C) sh[tid] = s;
    __syncthreads();
    if (tid >= 64) return;
    const int lane = tid;
    const float *src = k == 0 ? bg : main_proto + (size_t)(k - 1) * HD_C;
    const float v0 = src[lane], v1 = src[64 + lane], g0 = sh[lane], g1 = sh[64 + lane];
    const float nr = fmaxf(sqrtf(hd_wsum(fmaf(v0, v0, v1 * v1))), HD_EPS);
    const float q0 = v0 / nr, q1v = v1 / nr;
    const float dt = hd_wsum(fmaf(q0, g0, q1v * g1));
    const float r0 = (g0 - q0 * dt) / nr, r1 = (g1 - q1v * dt) / nr;
    if (k == 0) { dbg[lane] = dbg_a[lane] + r0; dbg[64 + lane] = dbg_a[64 + lane] + r1; }
    else {
        dmain[(size_t)(k - 1) * HD_C + lane] = dmain_a[(size_t)(k - 1) * HD_C + lane] + r0;
        dmain[(size_t)(k - 1) * HD_C + 64 + lane] = dmain_a[(size_t)(k - 1) * HD_C + 64 + lane] + r1;
    }
}

// ------------------------------------------------------------------------------------------------ the testing path
// capl.forward(eval_model=True) after getFeatures (capl.py:170-192): post_refine_proto_v2 WITHOUT the bg row, the refined
// prototypes of the base classes + gened_proto, get_pred against them, the geometric-word re-weighting (get_gp_weight
// :92-142 as a table look-up by word index) and its two accuracy figures.  hd_fwd1 (bg = NULL, no fake classes) does the
// soft pooling; grid = (B, ceil(classes / 4)), one wave per class:
template <int KT>
__global__ void __launch_bounds__(256)
hd_mid2_eval_kernel(const float *__restrict__ part1, const float *__restrict__ sums1, int nchunk, const float *__restrict__ main_proto,
                    const float *__restrict__ gened, int K1, int base_num, float *__restrict__ q2)
{
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, b = blockIdx.x;
    const int k = blockIdx.y * 4 + w;
    if (k >= K1) return;
    const float P0 = main_proto[(size_t)k * HD_C + lane], P1 = main_proto[(size_t)k * HD_C + 64 + lane];
    float u0 = 0.f, u1 = 0.f, s = 0.f;
#pragma unroll 8
    for (int ch = 0; ch < nchunk; ++ch) {
        const size_t p = (size_t)b * nchunk + ch;
        u0 += part1[(p * 32 + k) * HD_C + lane];
        u1 += part1[(p * 32 + k) * HD_C + 64 + lane];
        s += sums1[p * 32 + k];
    }
    const float is = 1.0f / s;
    const float p0 = u0 * is, p1 = u1 * is;
    const float pn = fmaxf(sqrtf(hd_wsum(fmaf(p0, p0, p1 * p1))), HD_EPS), Pn = fmaxf(sqrtf(hd_wsum(fmaf(P0, P0, P1 * P1))), HD_EPS);
    const float cosw = hd_wsum(fmaf(p0 / pn, P0 / Pn, (p1 / pn) * (P1 / Pn)));
    const float wv = cosw > 0.f ? cosw : 0.f;
    const float g0 = gened[(size_t)k * HD_C + lane], g1 = gened[(size_t)k * HD_C + 64 + lane];
    const float f0 = k < base_num ? fmaf(wv, p0, (1.0f - wv) * P0) + g0 : g0, f1 = k < base_num ? fmaf(wv, p1, (1.0f - wv) * P1) + g1 : g1;
    const float fn = fmaxf(sqrtf(hd_wsum(fmaf(f0, f0, f1 * f1))), HD_EPS);
    float *q2r = q2 + ((size_t)b * KT + k) * HD_C;
    q2r[lane] = f0 / fn; q2r[64 + lane] = f1 / fn;
}

// out[b][k][n] = 10 cos(F_n, refined[b][k]) * (coding[k][word_n] == 1 ? th : 1); part2 = {sum hit, sum hit over novel points,
// novel points, -} with hit = coding[y_n][word_n]
template <int KT>
__global__ void __launch_bounds__(256, 1)
hd_fwd2_eval_kernel(const float *__restrict__ feat, long long bs, int N, const float *__restrict__ q2g, int K1, const int *__restrict__ word,
                    const float *__restrict__ coding, int Kw, float th, const long long *__restrict__ label, int base_num,
                    float *__restrict__ out, float *__restrict__ part2)
{
    extern __shared__ __attribute__((aligned(16))) float hd_smem[];
    float *q2 = hd_smem;                                 // [KT][128]
    float *wred = q2 + KT * HD_C;                        // [4][4]
    float *cod = wred + 16;                              // [K1][Kw]
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int b = blockIdx.y, chunk = blockIdx.x, nchunk = gridDim.x;
    const int n = chunk * HD_PTS + w * 64 + lane;
    const bool valid = n < N;
    const float *col = feat + (size_t)b * bs + min(n, N - 1);
    for (int e = tid; e < K1 * Kw; e += 256) cod[e] = coding[e];
    hd_load_table<KT>(q2, q2g + (size_t)b * KT * HD_C, K1, tid);
    float d[1][KT], sq;
    const float *const tabs[1] = {q2};
    hd_dots<KT, 1>(col, (size_t)N, tabs, d, sq);
    const float rn = HD_SCALE / fmaxf(sqrtf(sq), HD_EPS);
    const int wd = min(max(word[(size_t)b * N + min(n, N - 1)], 0), Kw - 1);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(out + (size_t)b * K1 * N, 0, K1 * N * 4, 0x00020000);
    const int voff = valid ? n * 4 : 0x7fffffff;
#pragma unroll
    for (int k = 0; k < KT; ++k)
        if (k < K1) {
            const float wt = cod[k * Kw + wd] == 1.0f ? th : 1.0f;
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, d[0][k] * rn * wt), rs, voff, k * N * 4, 0);
        }
    float hit = 0.f, nov = 0.f;
    if (label != nullptr) {
        const long long y = label[(size_t)b * N + min(n, N - 1)];
        const bool ok = valid && y >= 0 && y < (long long)K1;
        hit = ok ? cod[(int)(ok ? y : 0) * Kw + wd] : 0.f;
        nov = (ok && y > (long long)(base_num - 1)) ? 1.f : 0.f;
    }
    const float s0 = hd_wsum(hit), s1 = hd_wsum(nov > 0.f ? hit : 0.f), s2 = hd_wsum(nov);
    if (lane == 0) { wred[w * 4 + 0] = s0; wred[w * 4 + 1] = s1; wred[w * 4 + 2] = s2; }
    __syncthreads();
    if (tid < 3) part2[(size_t)(b * nchunk + chunk) * 4 + tid] = ((wred[tid] + wred[4 + tid]) + wred[8 + tid]) + wred[12 + tid];
}

// stats[0] = mean hit over all points, stats[1] = mean hit over the novel-class points (0 if there are none)
__global__ void __launch_bounds__(64)
hd_evalstat_kernel(const float *__restrict__ part2, int nparts, double npoints, float *__restrict__ stats)
{
    double s0 = 0.0, s1 = 0.0, s2 = 0.0;
    for (int p = threadIdx.x; p < nparts; p += 64) { s0 += part2[(size_t)p * 4]; s1 += part2[(size_t)p * 4 + 1]; s2 += part2[(size_t)p * 4 + 2]; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { s0 += __shfl_xor(s0, o, 64); s1 += __shfl_xor(s1, o, 64); s2 += __shfl_xor(s2, o, 64); }
    if (threadIdx.x == 0) { stats[0] = (float)(s0 / npoints); stats[1] = s2 > 0.0 ? (float)(s1 / s2) : 0.f; }
}

// bit c of mask[0] set <=> some label equals c (0 <= c < 64); integer atomics: order-independent
__global__ void __launch_bounds__(256)
hd_presence_kernel(const long long *__restrict__ label, long long n, unsigned long long *__restrict__ mask)
{
    unsigned long long m = 0ull;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const long long y = label[i];
        if (y >= 0 && y < 64) m |= 1ull << y;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m |= __shfl_xor(m, o, 64);
    if ((threadIdx.x & 63) == 0 && m) atomicOr(mask, m);
}

// ------------------------------------------------------------------------------------------------ host side
struct HeadPlan {
    int KT, nchunk;
    size_t off_part, off_sums1, off_part2, off_q1, off_q2, off_ori, off_onrm, off_pooled, off_hs, off_dpool, off_t3, off_dbgp, off_dorip,
        off_dPp, off_dbga, off_dmaina, total;
};
HeadPlan hd_plan(int B, int N, int classes)
{
    HeadPlan p{};
    const int K1 = classes + 1;
    p.KT = K1 <= 16 ? 16 : 24;
    p.nchunk = (N + 127) / 128;                         // sized for the finest chunking of the passes (128-point workgroups)
    size_t o = 0;
    auto take = [&](size_t floats) { const size_t at = o; o += align_up(floats * sizeof(float), 256); return at; };
    p.off_part = take((size_t)B * p.nchunk * 64 * HD_C);           // part1 (32 rows) / part3 (64 rows) / part4 (32 rows), one after the other
    p.off_sums1 = take((size_t)B * p.nchunk * 32);
    p.off_part2 = take((size_t)B * p.nchunk * 4);
    p.off_q1 = take((size_t)p.KT * HD_C);
    p.off_q2 = take((size_t)B * p.KT * HD_C);
    p.off_ori = take((size_t)32 * HD_C);
    p.off_onrm = take(32);
    p.off_pooled = take((size_t)B * p.KT * HD_C);
    p.off_hs = take((size_t)B * p.KT * HD_HS);
    p.off_dpool = take((size_t)B * p.KT * HD_C);
    p.off_t3 = take((size_t)B * p.KT);
    p.off_dbgp = take((size_t)B * HD_C);
    p.off_dorip = take((size_t)B * 32 * HD_C);
    p.off_dPp = take((size_t)B * 32 * HD_C);
    p.off_dbga = take(HD_C);
    p.off_dmaina = take((size_t)32 * HD_C);
    p.total = o;
    return p;
}

int hd_check(const gfs3d_capl_head_args *a, const char *who, HeadFake &fk)
{
    GFS3D_REQUIRE(a != nullptr, "%s: args is NULL", who);
    GFS3D_REQUIRE(a->B >= 1 && a->B <= 65535 && a->N >= 1 && a->N < (1 << 22), "%s: bad shape B=%d N=%d", who, a->B, a->N);
    GFS3D_REQUIRE(a->C == HD_C, "%s: %d feature channels unsupported (the prototype head is built for main_dim = %d)", who, a->C, HD_C);
    GFS3D_REQUIRE(a->classes >= 1 && a->classes + 1 <= 24, "%s: %d classes unsupported (at most 23 + background)", who, a->classes);
    GFS3D_REQUIRE(a->base_num >= 0 && a->base_num <= a->classes, "%s: base_num %d outside [0, classes]", who, a->base_num);
    GFS3D_REQUIRE(a->n_fake >= 0 && a->n_fake <= GFS3D_HEAD_MAX_FAKE && a->classes + 1 + a->n_fake <= 32,
                  "%s: %d fake-novel classes unsupported (at most %d, and classes + 1 + fakes <= 32)", who, a->n_fake, GFS3D_HEAD_MAX_FAKE);
    GFS3D_REQUIRE(a->half >= 0 && a->half <= a->B, "%s: support half %d outside [0, B]", who, a->half);
    GFS3D_REQUIRE(a->feat_bstride >= (long long)a->C * a->N, "%s: feat_bstride smaller than C * N", who);
    GFS3D_REQUIRE(a->feat && a->label && a->main_proto && a->bg_proto && a->fsum && a->workspace && a->loss, "%s: NULL buffer", who);
    GFS3D_REQUIRE(a->workspace_bytes >= gfs3d_capl_head_workspace_bytes(a->B, a->N, a->classes), "%s: workspace too small", who);
    fk.n = a->n_fake; fk.half = a->half;
    for (int i = 0; i < GFS3D_HEAD_MAX_FAKE; ++i) fk.cls[i] = i < a->n_fake ? a->fake[i] : -1;
    for (int i = 0; i < a->n_fake; ++i)
        GFS3D_REQUIRE(a->fake[i] >= 1 && a->fake[i] <= a->classes, "%s: fake-novel label %d outside [1, classes]", who, a->fake[i]);
    return GFS3D_OK;
}

// 16-byte accesses along the points: whole groups of four inside every channel row
static bool hd_vec_ok(const void *p, long long bstride, int N)
{
    return N % 4 == 0 && bstride % 4 == 0 && (reinterpret_cast<uintptr_t>(p) & 15) == 0;
}
// Point tiles per wave of each matrix-core pass: 2 (128-point workgroups, two per CU) wherever the LDS of two workgroups fits a CU
template <int KT> struct HdCfg {
    static constexpr int F1 = 2, F2 = 2, B1 = KT <= 16 ? 2 : 4, B2 = KT <= 16 ? 2 : 4;     // (24-row tables: one workgroup per CU either way; measured)
};
static int hd_chunks(int N, int pts) { return (N + pts - 1) / pts; }
static_assert(2 * hd_lds_fwd1<16, 2>() <= 160 * 1024 && 2 * hd_lds_bwd1<16, 2>() <= 160 * 1024 && 2 * hd_lds_bwd2<16, 2>() <= 160 * 1024 &&
              2 * hd_lds_fwd1<24, 2>() <= 160 * 1024 && hd_lds_bwd1<24, 4>() <= 160 * 1024 && hd_lds_bwd2<24, 4>() <= 160 * 1024,
              "LDS of the matrix-core passes");

// forward pass 1 (also the soft pooling of the testing path); *nchunk_out = the chunk count of part1 / sums1
template <int KT>
int hd_launch_fwd1(const float *feat, long long bs, int N, int B, const long long *label, const float *main_proto, const float *bg, int K1,
                   const HeadFake &fk, float *part1, float *sums1, hipStream_t st, const char *who, int *nchunk_out)
{
    constexpr int NTL = HdCfg<KT>::F1;
    const dim3 grid(hd_chunks(N, HdT<NTL>::PTS), B);
    constexpr size_t lds = hd_lds_fwd1<KT, NTL>();
    if (hd_vec_ok(feat, bs, N)) {
        GFS3D_ENSURE_LDS((hd_fwd1_mm_kernel<KT, NTL, true>), lds, who);
        hipLaunchKernelGGL((hd_fwd1_mm_kernel<KT, NTL, true>), grid, dim3(256), lds, st, feat, bs, N, label, main_proto, bg, K1, fk, part1, sums1);
    } else {
        GFS3D_ENSURE_LDS((hd_fwd1_mm_kernel<KT, NTL, false>), lds, who);
        hipLaunchKernelGGL((hd_fwd1_mm_kernel<KT, NTL, false>), grid, dim3(256), lds, st, feat, bs, N, label, main_proto, bg, K1, fk, part1, sums1);
    }
    *nchunk_out = grid.x;
    GFS3D_LAUNCH_CHECK("hd_fwd1_kernel");
    return GFS3D_OK;
}
// the chunk count hd_launch_fwd1 used (phase 1 of the forward runs in a later call)
template <int KT> int hd_fwd1_chunks(int N) { return hd_chunks(N, HdT<HdCfg<KT>::F1>::PTS); }
template <int KT> int hd_bwd1_chunks(int N) { return hd_chunks(N, HdT<HdCfg<KT>::B1>::PTS); }

template <int KT>
int hd_forward(const gfs3d_capl_head_args *a, int phase, const HeadFake &fk, const char *who)
{
    const HeadPlan p = hd_plan(a->B, a->N, a->classes);
    char *ws = static_cast<char *>(a->workspace);
    auto F = [&](size_t off) { return reinterpret_cast<float *>(ws + off); };
    hipStream_t st = static_cast<hipStream_t>(a->stream);
    const int K1 = a->classes + 1;
    const long long *label = reinterpret_cast<const long long *>(a->label);
    if (phase == 0) {
        int nch = 0;
        if (int rc = hd_launch_fwd1<KT>(a->feat, a->feat_bstride, a->N, a->B, label, a->main_proto, a->bg_proto, K1, fk, F(p.off_part), F(p.off_sums1), st, who, &nch))
            return rc;
        if (fk.n > 0) {
            hipLaunchKernelGGL(hd_mid1_kernel, dim3(fk.n), dim3(1024), 0, st, F(p.off_part), F(p.off_sums1), a->B, nch, K1, fk.half, a->fsum);
            GFS3D_LAUNCH_CHECK("hd_mid1_kernel");
        }
        return GFS3D_OK;
    }
    GFS3D_REQUIRE(a->pred != nullptr, "%s: NULL pred", who);
    hipLaunchKernelGGL((hd_mid2_kernel<KT>), dim3(a->B, (K1 + 3) / 4), dim3(256), 0, st, F(p.off_part), F(p.off_sums1), hd_fwd1_chunks<KT>(a->N), a->main_proto,
                       a->bg_proto, K1, a->base_num, fk, a->fsum, F(p.off_q1), F(p.off_q2), F(p.off_ori), F(p.off_onrm), F(p.off_pooled), F(p.off_hs));
    GFS3D_LAUNCH_CHECK("hd_mid2_kernel");
    int nch2;
    {
        constexpr int NTL = HdCfg<KT>::F2;
        constexpr size_t lds = hd_lds_fwd2<KT, NTL>();
        const dim3 grid(hd_chunks(a->N, HdT<NTL>::PTS), a->B);
        nch2 = grid.x;
        if (hd_vec_ok(a->feat, a->feat_bstride, a->N)) {
            GFS3D_ENSURE_LDS((hd_fwd2_mm_kernel<KT, NTL, true>), lds, who);
            hipLaunchKernelGGL((hd_fwd2_mm_kernel<KT, NTL, true>), grid, dim3(256), lds, st, a->feat, a->feat_bstride, a->N, label,
                               (long long)a->ignore_index, F(p.off_q1), F(p.off_q2), K1, F(p.off_part2), reinterpret_cast<long long *>(a->pred));
        } else {
            GFS3D_ENSURE_LDS((hd_fwd2_mm_kernel<KT, NTL, false>), lds, who);
            hipLaunchKernelGGL((hd_fwd2_mm_kernel<KT, NTL, false>), grid, dim3(256), lds, st, a->feat, a->feat_bstride, a->N, label,
                               (long long)a->ignore_index, F(p.off_q1), F(p.off_q2), K1, F(p.off_part2), reinterpret_cast<long long *>(a->pred));
        }
    }
    GFS3D_LAUNCH_CHECK("hd_fwd2_kernel");
    hipLaunchKernelGGL(hd_loss_kernel, dim3(1), dim3(64), 0, st, F(p.off_part2), a->B * nch2, a->loss);
    GFS3D_LAUNCH_CHECK("hd_loss_kernel");
    return GFS3D_OK;
}

template <int KT>
int hd_backward(const gfs3d_capl_head_args *a, int phase, const HeadFake &fk, const char *who)
{
    const HeadPlan p = hd_plan(a->B, a->N, a->classes);
    char *ws = static_cast<char *>(a->workspace);
    auto F = [&](size_t off) { return reinterpret_cast<float *>(ws + off); };
    hipStream_t st = static_cast<hipStream_t>(a->stream);
    const int K1 = a->classes + 1;
    const long long *label = reinterpret_cast<const long long *>(a->label);
    GFS3D_REQUIRE(a->dloss && a->dfsum && a->dfeat && a->dmain && a->dbg, "%s: NULL gradient buffer", who);
    if (phase == 0) {
        const int nch = hd_bwd1_chunks<KT>(a->N);
        const dim3 grid(nch, a->B);
        {
            constexpr int NTL = HdCfg<KT>::B1;
            constexpr size_t lds = hd_lds_bwd1<KT, NTL>();
            if (hd_vec_ok(a->feat, a->feat_bstride, a->N)) {
                GFS3D_ENSURE_LDS((hd_bwd1_mm_kernel<KT, NTL, true>), lds, who);
                hipLaunchKernelGGL((hd_bwd1_mm_kernel<KT, NTL, true>), grid, dim3(256), lds, st, a->feat, a->feat_bstride, a->N, label,
                                   (long long)a->ignore_index, F(p.off_q1), F(p.off_q2), K1, a->loss, a->dloss, F(p.off_part));
            } else {
                GFS3D_ENSURE_LDS((hd_bwd1_mm_kernel<KT, NTL, false>), lds, who);
                hipLaunchKernelGGL((hd_bwd1_mm_kernel<KT, NTL, false>), grid, dim3(256), lds, st, a->feat, a->feat_bstride, a->N, label,
                                   (long long)a->ignore_index, F(p.off_q1), F(p.off_q2), K1, a->loss, a->dloss, F(p.off_part));
            }
        }
        GFS3D_LAUNCH_CHECK("hd_bwd1_kernel");
        hipLaunchKernelGGL((hd_bwdm1_kernel<KT>), dim3(a->B, (K1 + 3) / 4), dim3(256), 0, st, F(p.off_part), nch, a->main_proto, a->bg_proto, K1, a->base_num,
                           F(p.off_q2), F(p.off_pooled), F(p.off_hs), F(p.off_dbgp), F(p.off_dorip), F(p.off_dPp), F(p.off_dpool), F(p.off_t3));
        GFS3D_LAUNCH_CHECK("hd_bwdm1_kernel");
        hipLaunchKernelGGL(hd_bwdm2_kernel, dim3(K1), dim3(1024), 0, st, F(p.off_part), a->B, nch, a->main_proto, a->bg_proto, fk, a->fsum,
                           F(p.off_ori), F(p.off_onrm), F(p.off_dbgp), F(p.off_dorip), F(p.off_dPp), F(p.off_dbga), F(p.off_dmaina), a->dfsum);
        GFS3D_LAUNCH_CHECK("hd_bwdm2_kernel");
        return GFS3D_OK;
    }
    int nch;
    {
        constexpr int NTL = HdCfg<KT>::B2;
        constexpr size_t lds = hd_lds_bwd2<KT, NTL>();
        const dim3 grid(hd_chunks(a->N, HdT<NTL>::PTS), a->B);
        nch = grid.x;
        if (hd_vec_ok(a->feat, a->feat_bstride, a->N) && hd_vec_ok(a->dfeat, 0, a->N)) {
            GFS3D_ENSURE_LDS((hd_bwd2_mm_kernel<KT, NTL, true>), lds, who);
            hipLaunchKernelGGL((hd_bwd2_mm_kernel<KT, NTL, true>), grid, dim3(256), lds, st, a->feat, a->feat_bstride, a->N, label,
                               (long long)a->ignore_index, F(p.off_q1), F(p.off_q2), a->main_proto, a->bg_proto, K1, fk, a->loss, a->dloss, F(p.off_hs),
                               F(p.off_dpool), F(p.off_t3), a->dfsum, a->dfeat, F(p.off_part));
        } else {
            GFS3D_ENSURE_LDS((hd_bwd2_mm_kernel<KT, NTL, false>), lds, who);
            hipLaunchKernelGGL((hd_bwd2_mm_kernel<KT, NTL, false>), grid, dim3(256), lds, st, a->feat, a->feat_bstride, a->N, label,
                               (long long)a->ignore_index, F(p.off_q1), F(p.off_q2), a->main_proto, a->bg_proto, K1, fk, a->loss, a->dloss, F(p.off_hs),
                               F(p.off_dpool), F(p.off_t3), a->dfsum, a->dfeat, F(p.off_part));
        }
    }
    GFS3D_LAUNCH_CHECK("hd_bwd2_kernel");
    hipLaunchKernelGGL(hd_bwdfin_kernel, dim3(K1), dim3(1024), 0, st, F(p.off_part), a->B * nch, a->main_proto, a->bg_proto, F(p.off_dbga),
                       F(p.off_dmaina), a->dbg, a->dmain);
    GFS3D_LAUNCH_CHECK("hd_bwdfin_kernel");
    return GFS3D_OK;
}

}  // namespace
}  // namespace gfs3d

using namespace gfs3d;

extern "C" size_t gfs3d_capl_head_workspace_bytes(int B, int N, int classes)
{
    if (B < 1 || N < 1 || classes < 1 || classes + 1 > 24) return 0;
    return hd_plan(B, N, classes).total;
}

extern "C" int gfs3d_capl_head_forward_phase(const gfs3d_capl_head_args *a, int phase)
{
    const char *who = "gfs3d_capl_head_forward_phase";
    HeadFake fk;
    if (int rc = hd_check(a, who, fk)) return rc;
    GFS3D_REQUIRE(phase == 0 || phase == 1, "%s: phase %d outside [0, 1]", who, phase);
    ProfScope prof(3, static_cast<hipStream_t>(a->stream));
    return a->classes + 1 <= 16 ? hd_forward<16>(a, phase, fk, who) : hd_forward<24>(a, phase, fk, who);
}

extern "C" int gfs3d_capl_head_backward_phase(const gfs3d_capl_head_args *a, int phase)
{
    const char *who = "gfs3d_capl_head_backward_phase";
    HeadFake fk;
    if (int rc = hd_check(a, who, fk)) return rc;
    GFS3D_REQUIRE(phase == 0 || phase == 1, "%s: phase %d outside [0, 1]", who, phase);
    ProfScope prof(3, static_cast<hipStream_t>(a->stream));
    return a->classes + 1 <= 16 ? hd_backward<16>(a, phase, fk, who) : hd_backward<24>(a, phase, fk, who);
}

extern "C" int gfs3d_label_presence(const int64_t *label, long long n, uint64_t *mask, void *stream)
{
    const char *who = "gfs3d_label_presence";
    GFS3D_REQUIRE(label && mask && n >= 1, "%s: bad argument", who);
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (hipMemsetAsync(mask, 0, sizeof(uint64_t), st) != hipSuccess) {
        set_error("%s: hipMemsetAsync failed", who);
        return GFS3D_ERR_HIP;
    }
    const int blocks = (int)((n + 256 * 2 - 1) / (256 * 2));
    hipLaunchKernelGGL(hd_presence_kernel, dim3(blocks < 1 ? 1 : (blocks > 256 ? 256 : blocks)), dim3(256), 0, st,
                       reinterpret_cast<const long long *>(label), n, reinterpret_cast<unsigned long long *>(mask));
    GFS3D_LAUNCH_CHECK("hd_presence_kernel");
    return GFS3D_OK;
}

template <int KT>
static int hd_eval(const gfs3d_capl_eval_args *a, const char *who)
{
    const HeadPlan p = hd_plan(a->B, a->N, a->classes);
    char *ws = static_cast<char *>(a->workspace);
    auto F = [&](size_t off) { return reinterpret_cast<float *>(ws + off); };
    hipStream_t st = static_cast<hipStream_t>(a->stream);
    const int K1 = a->classes;                           // no bg row on the testing path
    HeadFake fk{};
    fk.n = 0; fk.half = a->B;
    int nch1 = 0;
    if (int rc = hd_launch_fwd1<KT>(a->feat, a->feat_bstride, a->N, a->B, nullptr, a->main_proto, nullptr, K1, fk, F(p.off_part), F(p.off_sums1), st, who, &nch1))
        return rc;
    hipLaunchKernelGGL((hd_mid2_eval_kernel<KT>), dim3(a->B, (K1 + 3) / 4), dim3(256), 0, st, F(p.off_part), F(p.off_sums1), nch1,
                       a->main_proto, a->gened_proto, K1, a->base_num, F(p.off_q2));
    GFS3D_LAUNCH_CHECK("hd_mid2_eval_kernel");
    const dim3 grid(hd_chunks(a->N, HD_PTS), a->B);       // (the lane = point logits pass: 256-point workgroups)
    const size_t lds = sizeof(float) * (KT * HD_C + 16 + (size_t)K1 * a->Kw);
    GFS3D_ENSURE_LDS((hd_fwd2_eval_kernel<KT>), lds, who);
    hipLaunchKernelGGL((hd_fwd2_eval_kernel<KT>), grid, dim3(256), lds, st, a->feat, a->feat_bstride, a->N, F(p.off_q2), K1, a->word, a->coding,
                       a->Kw, a->th, reinterpret_cast<const long long *>(a->label), a->base_num, a->logits, F(p.off_part2));
    GFS3D_LAUNCH_CHECK("hd_fwd2_eval_kernel");
    hipLaunchKernelGGL(hd_evalstat_kernel, dim3(1), dim3(64), 0, st, F(p.off_part2), a->B * (int)grid.x, (double)a->B * a->N, a->stats);
    GFS3D_LAUNCH_CHECK("hd_evalstat_kernel");
    return GFS3D_OK;
}

extern "C" int gfs3d_capl_head_eval(const gfs3d_capl_eval_args *a)
{
    const char *who = "gfs3d_capl_head_eval";
    GFS3D_REQUIRE(a != nullptr, "%s: args is NULL", who);
    GFS3D_REQUIRE(a->B >= 1 && a->B <= 65535 && a->N >= 1 && a->N < (1 << 22), "%s: bad shape B=%d N=%d", who, a->B, a->N);
    GFS3D_REQUIRE(a->C == HD_C, "%s: %d feature channels unsupported (the prototype head is built for main_dim = %d)", who, a->C, HD_C);
    GFS3D_REQUIRE(a->classes >= 1 && a->classes <= 23, "%s: %d classes unsupported (at most 23)", who, a->classes);
    GFS3D_REQUIRE(a->base_num >= 0 && a->base_num <= a->classes, "%s: base_num %d outside [0, classes]", who, a->base_num);
    GFS3D_REQUIRE(a->Kw >= 1 && (size_t)a->classes * a->Kw <= 16384, "%s: coding of %d x %d words does not fit the kernel's table", who,
                  a->classes, a->Kw);
    GFS3D_REQUIRE(a->feat_bstride >= (long long)a->C * a->N, "%s: feat_bstride smaller than C * N", who);
    GFS3D_REQUIRE(a->feat && a->main_proto && a->gened_proto && a->word && a->coding && a->logits && a->stats && a->workspace,
                  "%s: NULL buffer", who);
    GFS3D_REQUIRE(a->workspace_bytes >= gfs3d_capl_head_workspace_bytes(a->B, a->N, a->classes), "%s: workspace too small", who);
    ProfScope prof(3, static_cast<hipStream_t>(a->stream));
    return a->classes <= 16 ? hd_eval<16>(a, who) : hd_eval<24>(a, who);
}
