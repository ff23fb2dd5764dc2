// K3: point x prototype cosine matching on the gfx950 matrix cores, and the
// k-means assignment step of get_basis.
//
// cos[b,w,n] = <g_w, f_n> / (max(|g_w|,eps) max(|f_n|,eps))   (F.normalize, eps 1e-12)
//   * geometric words: softmax(10*cos) over words + arg-max one-hot (model/capl.py:344-353)
//   * get_pred: 10*cos logits against (cls,c) or per-sample (b,cls,c) prototypes
//     (model/capl.py:290-322)
// This is a true dense (B*N) x C . C x K contraction, so it runs on MFMA:
// v_mfma_f32_32x32x2_f32 (fp32 in, fp32 accumulate - an exact fp32 fma chain, the
// only MFMA that meets the 1e-4 fp32 contract without splitting).
//
// Tiling: workgroup = 4 waves = 128 points; a wave owns 32 points (MFMA columns)
// and ALL K prototype rows (NT row-tiles of 32), so the softmax / arg-max over
// prototypes is an in-lane reduction over accumulator registers plus one
// cross-half exchange - no LDS round trip for the scores.  Channels are
// processed in chunks of 64: the prototype chunk [NT*32][64] is staged in LDS
// (row stride 68 floats => conflict-free ds_read_b128 A-fragments), the point
// chunk is read straight from the channel-first input (128-B coalesced rows).
// Norms are accumulated on the fly and applied to the fp32 accumulators.
#include <type_traits>

#include "common.h"

namespace gfs3d {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int PR_CH = 64;     // channels per chunk
constexpr int PR_LD = 68;     // LDS row stride (floats)
constexpr float NORM_EPS = 1e-12f;

enum { MODE_LOGITS = 0, MODE_SOFTMAX = 1 };

// row (prototype) index inside a 32x32 accumulator tile for register r of lane-half h
__device__ __forceinline__ int acc_row(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

// Scores -> outputs, in the C/D layout both forward kernels share (lane = one point, register (t, r) <-> prototype row
// 32 t + acc_row(r, h)): cosine scaling, then either the logits or softmax over the words + arg-max one-hot.
// Stores go through a buffer descriptor on this batch element's (K, N) plane: the row part of the offset
// ((32 t + acc_row(r, 0)) N, wave-uniform) rides in the scalar offset, the lane part ((4 h N + n) 4 bytes) is ONE VGPR
// for all 16 NT stores, and a lane whose point is past N gets an offset the range check drops (no exec masking).
// Only the row tile that straddles K needs a per-register test; tiles past K are skipped wave-uniformly.
template <int NT, int MODE>
__device__ __forceinline__ void proto_epilogue(f32x16 (&acc)[NT], const float *ginv, float finv, float scale, int b, int n, int N, int K, int h,
                                               float *__restrict__ out, int32_t *__restrict__ assign, float *__restrict__ onehot)
{
    const int plane = (int)min((size_t)K * N * 4, (size_t)0x7fffffff);
    const int voff = (n < N) ? (4 * h * N + n) * 4 : 0x7fffffff;
    const float fs = finv * scale;
    auto put = [&](const __amdgpu_buffer_rsrc_t &rs, int t, int r, float v) {
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rs, voff, (t * 32 + acc_row(r, 0)) * N * 4, 0);
    };
    if (MODE == MODE_LOGITS) {
        const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc(out + (size_t)b * K * N, 0, plane, 0x00020000);
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            if (t * 32 >= K) break;
            const bool full = t * 32 + 32 <= K;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int w = t * 32 + acc_row(r, h);
                if (full || w < K) put(rs_out, t, r, acc[t][r] * ginv[w] * fs);
            }
        }
        return;
    }
    float mx = -__builtin_inff();
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int w = t * 32 + acc_row(r, h);
            const float v = (w < K) ? acc[t][r] * ginv[w] * fs : -__builtin_inff();
            acc[t][r] = v;
            mx = fmaxf(mx, v);
        }
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    float sum = 0.f;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float e = __expf(acc[t][r] - mx);     // exp(-inf) = 0 for padded rows; |rel err| ~1e-6 at the 1e-4 contract
            acc[t][r] = e;
            sum += e;
        }
    sum += __shfl_xor(sum, 32, 64);
    const float rsum = 1.0f / sum;
    // arg-max of the probabilities, ties -> lower word: a lane meets its rows in increasing order, so "first maximum"
    // is the strict '>' (padded rows have p = 0 and the maximum has p >= 1/K_w > 0)
    float best = -1.f;
    int bw = 0x7fffffff;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float p = acc[t][r] * rsum;
            acc[t][r] = p;
            if (p > best) { best = p; bw = t * 32 + acc_row(r, h); }
        }
    {   // combine the two lane halves: larger p wins, ties -> lower word index
        const float ob = __shfl_xor(best, 32, 64);
        const int ow = __shfl_xor(bw, 32, 64);
        if (ob > best || (ob == best && ow < bw)) { best = ob; bw = ow; }
    }
    if (assign && h == 0 && n < N) assign[(size_t)b * N + n] = bw;
    const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc(out ? out + (size_t)b * K * N : out, 0, out ? plane : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_oh = __builtin_amdgcn_make_buffer_rsrc(onehot ? onehot + (size_t)b * K * N : onehot, 0, onehot ? plane : 0, 0x00020000);
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        if (t * 32 >= K) break;
        const bool full = t * 32 + 32 <= K;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int w = t * 32 + acc_row(r, h);
            if (full || w < K) {
                put(rs_out, t, r, acc[t][r]);                       // (a NULL output has a zero-sized plane: dropped)
                put(rs_oh, t, r, (w == bw) ? 1.f : 0.f);
            }
        }
    }
}

template <int NT, int MODE>
__global__ void __launch_bounds__(256)
proto_mfma_kernel(const float *__restrict__ feat, long long feat_bstride, int C, int N,
                  const float *__restrict__ proto, int P, int K, float scale,
                  float *__restrict__ out,        // LOGITS: (B,K,N) ; SOFTMAX: soft (B,K,N) or NULL
                  int32_t *__restrict__ assign,   // SOFTMAX only, may be NULL
                  float *__restrict__ onehot)     // SOFTMAX only, may be NULL
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *gs = smem;                          // [NT*32][PR_LD]
    float *ginv = gs + NT * 32 * PR_LD;        // [NT*32] 1/max(|g|,eps)
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int j = lane & 31, h = lane >> 5;
    const int b = blockIdx.y;
    const int n0 = blockIdx.x * 128 + wv * 32;
    const int n = n0 + j;
    const int nc = min(n, N - 1);
    const float *fb = feat + (size_t)b * feat_bstride;
    const float *pb = proto + (P > 1 ? (size_t)b * K * C : 0);

    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    float fsq = 0.f;        // partial |f_n|^2 (this lane-half's channels)
    float gsq = 0.f;        // thread tid < NT*32 accumulates |g_tid|^2

    for (int c0 = 0; c0 < C; c0 += PR_CH) {
        __syncthreads();
        // stage prototype chunk: gs[w][cc] = proto[w][c0+cc]  (zero padded)
        for (int e = tid; e < NT * 32 * PR_CH; e += 256) {
            const int w = e >> 6, cc = e & 63;
            const float v = (w < K && c0 + cc < C) ? pb[(size_t)w * C + c0 + cc] : 0.f;
            gs[w * PR_LD + cc] = v;
        }
        // this lane's point chunk: channels c0 + 8t + 4h + s
        float f[32];
#pragma unroll
        for (int t = 0; t < 8; ++t)
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const int c = c0 + 8 * t + 4 * h + s;
                const float v = (c < C) ? fb[(size_t)c * N + nc] : 0.f;
                f[4 * t + s] = v;
                fsq = fmaf(v, v, fsq);
            }
        __syncthreads();
        if (tid < NT * 32) {
            const float *row = gs + tid * PR_LD;
#pragma unroll 8
            for (int cc = 0; cc < PR_CH; ++cc) gsq = fmaf(row[cc], row[cc], gsq);
        }
#pragma unroll
        for (int t = 0; t < 8; ++t) {
#pragma unroll
            for (int wt = 0; wt < NT; ++wt) {
                const float4 a = *reinterpret_cast<const float4 *>(gs + (wt * 32 + j) * PR_LD + 8 * t + 4 * h);
                acc[wt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, f[4 * t + 0], acc[wt], 0, 0, 0);
                acc[wt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, f[4 * t + 1], acc[wt], 0, 0, 0);
                acc[wt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, f[4 * t + 2], acc[wt], 0, 0, 0);
                acc[wt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, f[4 * t + 3], acc[wt], 0, 0, 0);
            }
        }
    }
    __syncthreads();
    if (tid < NT * 32) ginv[tid] = 1.0f / fmaxf(sqrtf(gsq), NORM_EPS);
    __syncthreads();
    fsq += __shfl_xor(fsq, 32, 64);
    const float finv = 1.0f / fmaxf(sqrtf(fsq), NORM_EPS);

    proto_epilogue<NT, MODE>(acc, ginv, finv, scale, b, n, N, K, h, out, assign, onehot);
}

// ---------------------------------------------------------------------------
// The same contraction on the bf16 matrix cores at fp32 accuracy (bf16x6: exact 3-way bf16 split of both operands,
// six piece products per fp32 product, see pointwise.hip) for K > 32 prototypes and C % 32 == 0 - the geometric-word
// case (C = 192, K_w = 200), which the fp32 MFMA form ran at 15 % of its matrix rate behind unpipelined staging.
//   workgroup = 8 waves = 256 points; a wave owns 32 points and all NT row tiles (same C/D layout, same epilogue);
//   channels in chunks of 32: the prototype chunk is split while it is staged (registers -> [row][hi|mid|lo|pad]
//   = 208-B rows, odd 16-B slot stride => conflict-free ds_read_b128) into one of TWO LDS images, the loads of chunk
//   c+1 (prototype float4s and this lane's 16 point channels) are issued before the MFMAs of chunk c: one barrier
//   per chunk, nothing waits on global memory in steady state.
// ---------------------------------------------------------------------------
constexpr int PX_ROW = 208;              // bytes per prototype row and 32-channel chunk
constexpr int PX_THREADS = 512;

// acc[t][r] = <g_w, f_n> for this lane's point n and the prototype rows w = 32 t + acc_row(r, h); fsq = this lane
// half's share of |f_n|^2; gsq_out[w] = |g_w|^2 (LDS, valid after the trailing barrier).
//   ROWMAJOR = false: pt = &feat[b][0][n], channel c at pt[c * cstride]  (channel-first input, 128-B coalesced rows)
//   ROWMAJOR = true : pt = &pts[m0][0] for the wave's first point, rows of `cstride` floats, `nrows` >= 1 of the 32
//                     valid: the wave reads its 32 x 32 chunk as 128-B row pieces (8 lanes per row) and turns it
//                     through a wave-private LDS tile `wtile` [32][XS_LD] so that lane (j, h) gets point j's channels
constexpr int XS_LD = 36;            // floats per row of that tile (144 B: 16-B aligned, 9 slots = odd)
//                     (`ridx` != NULL: row r of the wave is point ridx[r] of the matrix at `pt` - the exact pass over a list)
template <int NT, bool ROWMAJOR, int PREC = PREC_X6>
__device__ __forceinline__ void x6_scores(const float *__restrict__ pt, size_t cstride, int nrows, float *wtile,
                                          const float *__restrict__ pb, int C, int K,
                                          unsigned char *img, float *gsq_out, f32x16 (&acc)[NT], float &fsq,
                                          const int *__restrict__ ridx = nullptr)
{
    constexpr int NROW = NT * 32;
    constexpr int NU = (NROW * 8 + PX_THREADS - 1) / PX_THREADS;     // float4 units of a prototype chunk per thread (NT / 2, rounded up)
    constexpr bool RAGGED = (NROW * 8) % PX_THREADS != 0;            // odd NT: the last unit covers rows past the image for half the threads
    constexpr int NP = Mx<PREC>::NP;
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    const int tid = threadIdx.x, lane = tid & 63;
    const int j = lane & 31, h = lane >> 5;
    const int c4 = tid & 7, row0 = tid >> 3;             // staging: rows row0 + 64 u, channels 4 c4 .. +3 of the chunk
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    fsq = 0.f;
    float gsq[NU];
#pragma unroll
    for (int u = 0; u < NU; ++u) gsq[u] = 0.f;

    f32x4 pr[NU];
    float fr[16];
    auto load = [&](int c0) {
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            const int w = row0 + 64 * u;
            pr[u] = (w < K) ? *reinterpret_cast<const f32x4 *>(pb + (size_t)w * C + c0 + 4 * c4) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
        if (ROWMAJOR) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {              // rows (lane >> 3) + 8 i, 16 B at column 4 (lane & 7)
                const int row = min((lane >> 3) + 8 * i, nrows - 1);
                const size_t prow = ridx ? (size_t)ridx[row] : (size_t)row;
                const f32x4 v = *reinterpret_cast<const f32x4 *>(pt + prow * cstride + c0 + 4 * (lane & 7));
#pragma unroll
                for (int e = 0; e < 4; ++e) fr[4 * i + e] = v[e];
            }
        } else {
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int e = 0; e < 8; ++e) fr[8 * s + e] = pt[(size_t)(c0 + 16 * s + 8 * h + e) * cstride];
        }
    };
    auto stage = [&](unsigned char *dst_img) {
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            float v4[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                v4[e] = pr[u][e];
                gsq[u] = fmaf(pr[u][e], pr[u][e], gsq[u]);
            }
            uint2 pc[NP];
            mm_pieces4<PREC>(v4, pc);
            unsigned char *dst = dst_img + (row0 + 64 * u) * PX_ROW + c4 * 8;
            if (!RAGGED || row0 + 64 * u < NROW) {
#pragma unroll
                for (int q = 0; q < NP; ++q) *reinterpret_cast<uint2 *>(dst + 64 * q) = pc[q];
            }
        }
    };
    uint4 bp[2][NP];                                     // this lane's point: pieces of 8 channels per K-step
    auto split_point = [&]() {
        if (ROWMAJOR) {                                // row pieces -> tile -> this lane's 16 channels of point j
#pragma unroll
            for (int i = 0; i < 4; ++i)
                *reinterpret_cast<f32x4 *>(wtile + ((lane >> 3) + 8 * i) * XS_LD + 4 * (lane & 7)) =
                    f32x4{fr[4 * i], fr[4 * i + 1], fr[4 * i + 2], fr[4 * i + 3]};
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const f32x4 v = *reinterpret_cast<const f32x4 *>(wtile + j * XS_LD + 16 * s + 8 * h + 4 * q);
#pragma unroll
                    for (int e = 0; e < 4; ++e) fr[8 * s + 4 * q + e] = v[e];
                }
            __builtin_amdgcn_wave_barrier();
        }
#pragma unroll
        for (int s = 0; s < 2; ++s) {
#pragma unroll
            for (int e = 0; e < 8; ++e) fsq = fmaf(fr[8 * s + e], fr[8 * s + e], fsq);
            mm_pieces8<PREC>(fr + 8 * s, bp[s]);
        }
    };
    load(0);
    stage(img);
    split_point();
    __syncthreads();

    const int nchunk = C / 32;
    for (int c = 0; c < nchunk; ++c) {
        const unsigned char *cur = img + (c & 1) * NROW * PX_ROW;
        if (c + 1 < nchunk) load(32 * (c + 1));
        const unsigned char *pa = cur + j * PX_ROW + h * 16;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
#pragma unroll
            for (int wt = 0; wt < NT; ++wt) {
                uint4 af[NP];
#pragma unroll
                for (int pl = 0; pl < NP; ++pl)
                    af[pl] = *reinterpret_cast<const uint4 *>(pa + wt * 32 * PX_ROW + pl * 64 + s * 32);
                mm_mac<PREC>(af, bp[s], acc[wt]);
            }
        }
        if (c + 1 < nchunk) {
            stage(img + ((c + 1) & 1) * NROW * PX_ROW);   // last read of that image: chunk c-1, a barrier ago
            split_point();
        }
        __syncthreads();
    }
    // |g_w|^2: the 8 threads of a row are 8 consecutive lanes
#pragma unroll
    for (int u = 0; u < NU; ++u) {
        float v = gsq[u];
        v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64);
        if (c4 == 0 && (!RAGGED || row0 + 64 * u < NROW)) gsq_out[row0 + 64 * u] = v;
    }
    __syncthreads();
}

template <int NT, int MODE, int PREC>
__global__ void __launch_bounds__(PX_THREADS)
proto_x6_kernel(const float *__restrict__ feat, long long feat_bstride, int C, int N,
                const float *__restrict__ proto, int P, int K, float scale,
                float *__restrict__ out, int32_t *__restrict__ assign, float *__restrict__ onehot)
{
    constexpr int NROW = NT * 32;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    unsigned char *img = reinterpret_cast<unsigned char *>(smem);            // [2][NROW][PX_ROW]
    float *ginv = reinterpret_cast<float *>(img + 2 * NROW * PX_ROW);         // [NROW] |g|^2, then 1/max(|g|,eps)
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int j = lane & 31, h = lane >> 5;
    const int b = blockIdx.y;
    const int n = blockIdx.x * 256 + wv * 32 + j;
    f32x16 acc[NT];
    float fsq;
    x6_scores<NT, false, PREC>(feat + (size_t)b * feat_bstride + min(n, N - 1), (size_t)N, 0, nullptr,
                               proto + (P > 1 ? (size_t)b * K * C : 0), C, K, img, ginv, acc, fsq);
    if (tid < NROW) ginv[tid] = 1.0f / fmaxf(sqrtf(ginv[tid]), NORM_EPS);
    __syncthreads();
    fsq += __shfl_xor(fsq, 32, 64);
    const float finv = 1.0f / fmaxf(sqrtf(fsq), NORM_EPS);
    proto_epilogue<NT, MODE>(acc, ginv, finv, scale, b, n, N, K, h, out, assign, onehot);
}

template <int NT, int MODE>
static int launch_proto(const float *feat, long long bs, int B, int C, int N, const float *proto, int P, int K,
                        float scale, float *out, int32_t *assign, float *onehot, hipStream_t st, const char *who)
{
    const size_t lds = sizeof(float) * (NT * 32 * PR_LD + NT * 32);
    GFS3D_ENSURE_LDS((proto_mfma_kernel<NT, MODE>), lds, who);
    dim3 grid((N + 127) / 128, B);
    ProfScope prof(3, st);
    hipLaunchKernelGGL((proto_mfma_kernel<NT, MODE>), grid, dim3(256), lds, st, feat, bs, C, N, proto, P, K, scale,
                       out, assign, onehot);
    GFS3D_LAUNCH_CHECK(who);
    return GFS3D_OK;
}

template <int NT, int MODE, int PREC>
static int launch_proto_x6p(const float *feat, long long bs, int B, int C, int N, const float *proto, int P, int K,
                            float scale, float *out, int32_t *assign, float *onehot, hipStream_t st, const char *who)
{
    const size_t lds = (size_t)2 * NT * 32 * PX_ROW + sizeof(float) * NT * 32;
    GFS3D_ENSURE_LDS((proto_x6_kernel<NT, MODE, PREC>), lds, who);
    dim3 grid((N + 255) / 256, B);
    ProfScope prof(3, st);
    hipLaunchKernelGGL((proto_x6_kernel<NT, MODE, PREC>), grid, dim3(PX_THREADS), lds, st, feat, bs, C, N, proto, P, K, scale,
                       out, assign, onehot);
    GFS3D_LAUNCH_CHECK(who);
    return GFS3D_OK;
}
template <int NT, int MODE>
static int launch_proto_x6(const float *feat, long long bs, int B, int C, int N, const float *proto, int P, int K,
                           float scale, float *out, int32_t *assign, float *onehot, hipStream_t st, const char *who, int prec)
{
    if (prec == PREC_BF16) return launch_proto_x6p<NT, MODE, PREC_BF16>(feat, bs, B, C, N, proto, P, K, scale, out, assign, onehot, st, who);
    if (prec == PREC_F16) return launch_proto_x6p<NT, MODE, PREC_F16>(feat, bs, B, C, N, proto, P, K, scale, out, assign, onehot, st, who);
    return launch_proto_x6p<NT, MODE, PREC_X6>(feat, bs, B, C, N, proto, P, K, scale, out, assign, onehot, st, who);
}

template <int MODE>
static int dispatch_proto(const float *feat, long long bs, int B, int C, int N, const float *proto, int P, int K,
                          float scale, float *out, int32_t *assign, float *onehot, hipStream_t st, const char *who,
                          int prec = PREC_X6)
{
    const int nt = (K + 31) / 32;
    static const bool x6 = [] { const char *e = getenv("GFS3D_PROTO_X6"); return !(e && e[0] == '0'); }();     // tuning / A-B only
    if (x6 && nt >= 2 && C % 32 == 0 && (reinterpret_cast<uintptr_t>(proto) & 15) == 0) {
        if (nt <= 2) return launch_proto_x6<2, MODE>(feat, bs, B, C, N, proto, P, K, scale, out, assign, onehot, st, who, prec);
        if (nt <= 4) return launch_proto_x6<4, MODE>(feat, bs, B, C, N, proto, P, K, scale, out, assign, onehot, st, who, prec);
        // (K_w = 180 -> 6 row tiles, K_w = 200 -> 7: an eighth / a quarter less matrix, LDS and epilogue work than 8)
        if (nt <= 6) return launch_proto_x6<6, MODE>(feat, bs, B, C, N, proto, P, K, scale, out, assign, onehot, st, who, prec);
        if (nt <= 7) return launch_proto_x6<7, MODE>(feat, bs, B, C, N, proto, P, K, scale, out, assign, onehot, st, who, prec);
        return launch_proto_x6<8, MODE>(feat, bs, B, C, N, proto, P, K, scale, out, assign, onehot, st, who, prec);
    }
    if (nt <= 1) return launch_proto<1, MODE>(feat, bs, B, C, N, proto, P, K, scale, out, assign, onehot, st, who);
    if (nt <= 2) return launch_proto<2, MODE>(feat, bs, B, C, N, proto, P, K, scale, out, assign, onehot, st, who);
    if (nt <= 4) return launch_proto<4, MODE>(feat, bs, B, C, N, proto, P, K, scale, out, assign, onehot, st, who);
    return launch_proto<8, MODE>(feat, bs, B, C, N, proto, P, K, scale, out, assign, onehot, st, who);
}

// ---------------------------------------------------------------------------
// Backward of the cosine heads with respect to the point features (the autograd of capl.py:344-353 and of
// get_pred, capl.py:290-322):
//   d[w][n]   = scale * dout[w][n]                                      (logits)
//             = scale * soft[w][n] (dsoft[w][n] - sum_w' soft dsoft)     (softmax over the words)
//   dfhat[c]  = sum_w ghat[w][c] d[w]        - a (C x K) . (K x points) contraction on MFMA: A = the prototype
//               chunk from LDS (rows = channels), B = d straight from registers (the lane owns ONE point, register
//               (t, r) <-> prototype row 32 t + acc_row(r, h): the C/D layout of the forward kernel)
//   dfeat[c]  = (dfhat[c] - fhat[c] <fhat, dfhat>) / max(|f|, eps)       in-lane over the accumulator registers
// One workgroup = 128 points; a wave owns 32 points, ALL prototypes (NT row tiles) and ALL channels (CT tiles of 32).
// ---------------------------------------------------------------------------
template <int NT, int CT, int MODE>
__global__ void __launch_bounds__(256)
proto_bwd_feat_kernel(const float *__restrict__ feat, long long feat_bstride, int C, int N,
                      const float *__restrict__ proto, int P, int K, float scale,
                      const float *__restrict__ soft,        // SOFTMAX: (B,K,N) forward result
                      const float *__restrict__ dup,         // (B,K,N) gradient of soft / of the logits
                      float *__restrict__ dfeat)             // (B,C,N)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *gs = smem;                          // [NT*32][PR_LD]  prototype chunk (64 channels)
    float *ginv = gs + NT * 32 * PR_LD;        // [NT*32]
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int j = lane & 31, h = lane >> 5;
    const int b = blockIdx.y;
    const int n = blockIdx.x * 128 + wv * 32 + j;
    const int nc = min(n, N - 1);
    const float *fb = feat + (size_t)b * feat_bstride;
    const float *pb = proto + (P > 1 ? (size_t)b * K * C : 0);

    if (tid < NT * 32) {                        // 1 / max(|g_w|, eps) over all channels
        float gsq = 0.f;
        if (tid < K)
            for (int c = 0; c < C; ++c) { const float v = pb[(size_t)tid * C + c]; gsq = fmaf(v, v, gsq); }
        ginv[tid] = (tid < K) ? 1.0f / fmaxf(sqrtf(gsq), NORM_EPS) : 0.f;
    }
    __syncthreads();

    // d[t][r] for prototype row w = 32 t + acc_row(r, h), already multiplied by scale / |g_w|
    float dot = 0.f;
    if (MODE == MODE_SOFTMAX) {
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int w = t * 32 + acc_row(r, h);
                if (w < K) {
                    const size_t o = ((size_t)b * K + w) * N + nc;
                    dot = fmaf(soft[o], dup[o], dot);
                }
            }
        dot += __shfl_xor(dot, 32, 64);
    }
    float d[NT][16];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int w = t * 32 + acc_row(r, h);
            float v = 0.f;
            if (w < K) {
                const size_t o = ((size_t)b * K + w) * N + nc;
                v = (MODE == MODE_SOFTMAX) ? soft[o] * (dup[o] - dot) : dup[o];
                v *= scale * ginv[w];
            }
            d[t][r] = v;
        }

    f32x16 G[CT];                               // dfhat, rows = channel 32 ct + acc_row(r, h), column = point
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int r = 0; r < 16; ++r) G[ct][r] = 0.f;
#pragma unroll
    for (int cc = 0; cc < CT / 2; ++cc) {       // 64-channel chunks
        const int c0 = cc * PR_CH;
        __syncthreads();
        for (int e = tid; e < NT * 32 * PR_CH; e += 256) {
            const int w = e >> 6, c = e & 63;
            gs[w * PR_LD + c] = (w < K && c0 + c < C) ? pb[(size_t)w * C + c0 + c] : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float *row = gs + (t * 32 + acc_row(r, h)) * PR_LD + j;     // A[i = channel j][k = h] = g[w][c0 + j]
                G[2 * cc] = __builtin_amdgcn_mfma_f32_32x32x2f32(row[0], d[t][r], G[2 * cc], 0, 0, 0);
                G[2 * cc + 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(row[32], d[t][r], G[2 * cc + 1], 0, 0, 0);
            }
    }
    // normalisation backward, in-lane: |f|^2 and <f, dfhat> over this lane's channels, then the other lane half's
    float fsq = 0.f, fd = 0.f;
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int c = ct * 32 + acc_row(r, h);
            const float fv = (c < C) ? fb[(size_t)c * N + nc] : 0.f;
            fsq = fmaf(fv, fv, fsq);
            fd = fmaf(fv, G[ct][r], fd);
        }
    fsq += __shfl_xor(fsq, 32, 64);
    fd += __shfl_xor(fd, 32, 64);
    const float finv = 1.0f / fmaxf(sqrtf(fsq), NORM_EPS);
    const float coef = fd * finv * finv;         // <fhat, dfhat> / |f|
    if (n < N) {
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int c = ct * 32 + acc_row(r, h);
                if (c < C) {
                    const float fv = fb[(size_t)c * N + n];
                    dfeat[((size_t)b * C + c) * N + n] = (G[ct][r] - fv * coef) * finv;
                }
            }
    }
}

template <int NT, int CT, int MODE>
static int launch_proto_bwd(const float *feat, long long bs, int B, int C, int N, const float *proto, int P, int K, float scale,
                            const float *soft, const float *dup, float *dfeat, hipStream_t st, const char *who)
{
    const size_t lds = sizeof(float) * (NT * 32 * PR_LD + NT * 32);
    GFS3D_ENSURE_LDS((proto_bwd_feat_kernel<NT, CT, MODE>), lds, who);
    dim3 grid((N + 127) / 128, B);
    ProfScope prof(3, st);
    hipLaunchKernelGGL((proto_bwd_feat_kernel<NT, CT, MODE>), grid, dim3(256), lds, st, feat, bs, C, N, proto, P, K, scale, soft, dup, dfeat);
    GFS3D_LAUNCH_CHECK(who);
    return GFS3D_OK;
}

// ---------------------------------------------------------------------------
// The same backward on the bf16 matrix cores (bf16x6) for K > 32 prototypes, C % 32 == 0, C <= 192 - the geometric
// words.  dfhat[c][n] = sum_w ghat[w][c] d[w][n] contracts over the WORDS, so the roles of proto_x6_kernel swap:
//   output rows = channels (CT tiles), K-steps = 16 words, one chunk = one 32-word row tile t;
//   B operand = d[t] of the lane's point, straight from the registers of the C/D layout: K-index 8 h + e of step s is
//               word 32 t + 16 s + (e & 3) + 8 (e >> 2) + 4 h  (register r = 8 s + e of the lane half);
//   A operand = the prototype tile TRANSPOSED while it is staged: a thread reads 16 B along the channels of one word,
//               splits, and writes one bf16 per channel row and piece at that word's permuted K position
//               (image [channel][hi | mid | lo | pad], 208-B rows, two images, one barrier per chunk).
// d[t] needs the softmax dot over ALL words first: a pre-pass over soft/dup (coalesced; the main loop reads the
// tile's 32 values again, one chunk ahead).  1 / |g_w| comes from a block-wide pre-pass over the prototypes.
// ---------------------------------------------------------------------------
__device__ __forceinline__ int px_kpos(int u) { return 8 * ((u >> 2) & 1) + (u & 3) + 4 * (u >> 3); }   // word u of a 16-group -> K index

template <int NT, int CT, int MODE, int PREC>
__global__ void __launch_bounds__(PX_THREADS)
proto_bwd_x6_kernel(const float *__restrict__ feat, long long feat_bstride, int C, int N,
                    const float *__restrict__ proto, int P, int K, float scale,
                    const float *__restrict__ soft, const float *__restrict__ dup, float *__restrict__ dfeat)
{
    constexpr int NROW = CT * 32;                        // image rows = channels
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    extern __shared__ __attribute__((aligned(16))) float smem[];
    unsigned char *img = reinterpret_cast<unsigned char *>(smem);            // [2][NROW][PX_ROW]
    float *ginv = reinterpret_cast<float *>(img + 2 * NROW * PX_ROW);         // [NT*32]
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int j = lane & 31, h = lane >> 5;
    const int b = blockIdx.y;
    const int n = blockIdx.x * 256 + wv * 32 + j;
    const int nc = min(n, N - 1);
    const float *fb = feat + (size_t)b * feat_bstride;
    const float *pb = proto + (P > 1 ? (size_t)b * K * C : 0);
    const float *sb = soft ? soft + (size_t)b * K * N + nc : nullptr;
    const float *ub = dup + (size_t)b * K * N + nc;

    {   // 1 / max(|g_w|, eps): two threads per word, float4 loads
        const int w = tid >> 1, half = tid & 1;
        float gsq = 0.f;
        if (w < K)
            for (int c = half * (C / 2); c < (half + 1) * (C / 2); c += 4) {
                const f32x4 v = *reinterpret_cast<const f32x4 *>(pb + (size_t)w * C + c);
                gsq = fmaf(v[0], v[0], fmaf(v[1], v[1], fmaf(v[2], v[2], fmaf(v[3], v[3], gsq))));
            }
        gsq += __shfl_xor(gsq, 1, 64);
        if (half == 0 && w < NT * 32) ginv[w] = (w < K) ? 1.0f / fmaxf(sqrtf(gsq), NORM_EPS) : 0.f;
    }
    float dot = 0.f;
#if !(defined(PBX_ABL) && (PBX_ABL & 1))      // ablation builds only (wrong results, timing of the rest)
    if (MODE == MODE_SOFTMAX) {
#pragma unroll 4
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int w = t * 32 + acc_row(r, h);
                if (w < K) dot = fmaf(sb[(size_t)w * N], ub[(size_t)w * N], dot);
            }
        dot += __shfl_xor(dot, 32, 64);
    }
#endif
    __syncthreads();

    // staging of tile t: units of (word, 4 channels); unit q = tid + 512 u covers word q / (C/4), channels 4 (q % (C/4))
    constexpr int NU = (32 * CT * 8 + PX_THREADS - 1) / PX_THREADS;          // 32 words x C/4 units per tile
    const int upw = C >> 2;                              // units per word
    f32x4 pr[NU];
    float sv[16], uv[16];
    auto load = [&](int t) {
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            const int q = tid + PX_THREADS * u, wl = q / upw, cq = q - wl * upw, w = t * 32 + wl;
            pr[u] = (wl < 32 && w < K) ? *reinterpret_cast<const f32x4 *>(pb + (size_t)w * C + 4 * cq) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int w = t * 32 + acc_row(r, h);
            const bool ok = w < K;
#if defined(PBX_ABL) && (PBX_ABL & 4)
            uv[r] = 0.25f * (float)r; sv[r] = 0.5f;
#else
            uv[r] = ok ? ub[(size_t)w * N] : 0.f;
            sv[r] = (MODE == MODE_SOFTMAX && ok) ? sb[(size_t)w * N] : 0.f;
#endif
        }
    };
    auto stage = [&](unsigned char *dst_img) {
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            const int q = tid + PX_THREADS * u, wl = q / upw, cq = q - wl * upw;
            if (wl < 32) {
                unsigned short *dst = reinterpret_cast<unsigned short *>(dst_img + (4 * cq) * PX_ROW) + 16 * (wl >> 4) + px_kpos(wl & 15);
#pragma unroll
                for (int e = 0; e < 4; ++e) mm_store1<PREC>(pr[u][e], dst + e * (PX_ROW / 2), 32);
            }
        }
    };
    constexpr int NP = Mx<PREC>::NP;
    uint4 bp[2][NP];                                     // pieces of d[t]: K-step s <-> registers 8 s .. 8 s + 7
    auto split_d = [&](int t) {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            float v8[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int r = 8 * s + e, w = t * 32 + acc_row(r, h);
                const float v = (MODE == MODE_SOFTMAX) ? sv[r] * (uv[r] - dot) : uv[r];
                v8[e] = v * (scale * ginv[min(w, NT * 32 - 1)]);
            }
            mm_pieces8<PREC>(v8, bp[s]);
        }
    };

    f32x16 G[CT];                               // dfhat, rows = channel 32 ct + acc_row(r, h), column = point
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int r = 0; r < 16; ++r) G[ct][r] = 0.f;
    const int ntile = (K + 31) / 32;
    load(0);
    stage(img);
    split_d(0);
    __syncthreads();
    for (int t = 0; t < ntile; ++t) {
        const unsigned char *cur = img + (t & 1) * NROW * PX_ROW;
        if (t + 1 < ntile) load(t + 1);
        const unsigned char *pa = cur + j * PX_ROW + h * 16;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
                uint4 af[NP];
#pragma unroll
                for (int pl = 0; pl < NP; ++pl)
                    af[pl] = *reinterpret_cast<const uint4 *>(pa + ct * 32 * PX_ROW + pl * 64 + s * 32);
                mm_mac<PREC>(af, bp[s], G[ct]);
            }
        }
        if (t + 1 < ntile) {
            stage(img + ((t + 1) & 1) * NROW * PX_ROW);
            split_d(t + 1);
        }
        __syncthreads();
    }
    // normalisation backward, in-lane: |f|^2 and <f, dfhat> over this lane's channels, then the other lane half's
    const unsigned voff = (unsigned)(4 * h * N + nc) * 4u;                   // + (32 ct + acc_row(r, 0)) N 4 bytes
    const __amdgpu_buffer_rsrc_t rs_f = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(fb), 0, (int)min((size_t)C * N * 4, (size_t)0x7fffffff), 0x00020000);
    float fsq = 0.f, fd = 0.f;
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
#if defined(PBX_ABL) && (PBX_ABL & 2)
            const float fv = 1.0f + 0.01f * r;
#else
            const float fv = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_f, voff, (ct * 32 + acc_row(r, 0)) * N * 4, 0));
#endif
            fsq = fmaf(fv, fv, fsq);
            fd = fmaf(fv, G[ct][r], fd);
        }
    fsq += __shfl_xor(fsq, 32, 64);
    fd += __shfl_xor(fd, 32, 64);
    const float finv = 1.0f / fmaxf(sqrtf(fsq), NORM_EPS);
    const float coef = fd * finv * finv;         // <fhat, dfhat> / |f|
    const __amdgpu_buffer_rsrc_t rs_o = __builtin_amdgcn_make_buffer_rsrc(dfeat + (size_t)b * C * N, 0, (int)min((size_t)C * N * 4, (size_t)0x7fffffff), 0x00020000);
    const int ovoff = (n < N) ? (int)voff : 0x7fffffff;                     // a lane past N: dropped by the range check
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int so = (ct * 32 + acc_row(r, 0)) * N * 4;
#if defined(PBX_ABL) && (PBX_ABL & 2)
            const float fv = 1.0f + 0.01f * r;
#else
            const float fv = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_f, voff, so, 0));
#endif
#if defined(PBX_ABL) && (PBX_ABL & 8)
            if (G[ct][r] == 12345.678f)
#endif
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, (G[ct][r] - fv * coef) * finv), rs_o, ovoff, so, 0);
        }
}

template <int NT, int CT, int MODE, int PREC>
static int launch_proto_bwd_x6p(const float *feat, long long bs, int B, int C, int N, const float *proto, int P, int K, float scale,
                                const float *soft, const float *dup, float *dfeat, hipStream_t st, const char *who)
{
    const size_t lds = (size_t)2 * CT * 32 * PX_ROW + sizeof(float) * NT * 32;
    GFS3D_ENSURE_LDS((proto_bwd_x6_kernel<NT, CT, MODE, PREC>), lds, who);
    dim3 grid((N + 255) / 256, B);
    ProfScope prof(3, st);
    hipLaunchKernelGGL((proto_bwd_x6_kernel<NT, CT, MODE, PREC>), grid, dim3(PX_THREADS), lds, st, feat, bs, C, N, proto, P, K, scale, soft, dup, dfeat);
    GFS3D_LAUNCH_CHECK(who);
    return GFS3D_OK;
}
template <int NT, int CT, int MODE>
static int launch_proto_bwd_x6(const float *feat, long long bs, int B, int C, int N, const float *proto, int P, int K, float scale,
                               const float *soft, const float *dup, float *dfeat, hipStream_t st, const char *who, int prec)
{
    if (prec != PREC_X6)       // gradients: bf16 operands in both reduced modes
        return launch_proto_bwd_x6p<NT, CT, MODE, PREC_BF16>(feat, bs, B, C, N, proto, P, K, scale, soft, dup, dfeat, st, who);
    return launch_proto_bwd_x6p<NT, CT, MODE, PREC_X6>(feat, bs, B, C, N, proto, P, K, scale, soft, dup, dfeat, st, who);
}

// ---------------------------------------------------------------------------
// The same kernel fed from a PER-CALL IMAGE of the words (shared prototypes, a caller workspace): proto_bwd_x6_kernel has every
// workgroup transpose and split the words itself - 2-byte LDS stores, 72 per thread and tile - at ONE workgroup per CU, so its
// memory phases (softmax-dot pre-pass, tile loads, feature loads, stores) run one after the other with nothing beside them
// (DESIGN: 20 + 26 + 46 + 27 us around a 65 us core).  Here `proto_words_image_kernel` writes, once per call, the transposed
// piece image [tile][channel][hi | mid | lo][32 words at their K positions] (192-B rows) and 1 / |g_w|; a workgroup of FOUR
// waves (128 points, 81 KB of LDS: two per CU, their phases overlap) copies a tile's rows with 16-byte moves.
// ---------------------------------------------------------------------------
typedef __attribute__((address_space(3))) char px_lds_char;
// 16 B per lane, global -> LDS without a VGPR destination (see ks_dma16_off in knn.hip): lane l lands at lds_base + 16 l
__device__ __forceinline__ void px_dma16_off(const void *gbase_wave_uniform, unsigned voff, const void *lds_dst_wave_uniform)
{
    const unsigned base = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(px_lds_char *)lds_dst_wave_uniform);
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(base), "v"(voff), "s"(gbase_wave_uniform) : "memory");
}
constexpr int PXI_THREADS = 256;
constexpr int PXI_GROW = 192;            // bytes per image row in global memory (three 64-B piece planes)

template <int PREC>
__global__ void __launch_bounds__(256)
proto_words_image_kernel(const float *__restrict__ proto, int K, int C, unsigned short *__restrict__ img, float *__restrict__ ginv)
{
    const int ntile = (K + 31) / 32;
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e < ntile * 32 * C) {
        const int w = e / C, c = e - w * C;
        const int t = w >> 5, wl = w & 31;
        const float v = w < K ? proto[(size_t)w * C + c] : 0.f;
        unsigned short *dst = img + ((size_t)t * C + c) * (PXI_GROW / 2) + 16 * (wl >> 4) + px_kpos(wl & 15);
        if constexpr (PREC == PREC_X6) mm_store1<PREC_X6>(v, dst, 32);
        else { dst[0] = mm_cvt1<PREC>(v); dst[32] = 0; dst[64] = 0; }
    }
    if (blockIdx.x == 0)
        for (int w = threadIdx.x; w < ntile * 32; w += 256) {
            float gsq = 0.f;
            if (w < K)
                for (int c = 0; c < C; ++c) gsq = fmaf(proto[(size_t)w * C + c], proto[(size_t)w * C + c], gsq);
            ginv[w] = w < K ? 1.0f / fmaxf(sqrtf(gsq), NORM_EPS) : 0.f;
        }
}

template <int NT, int CT, int MODE, int PREC>
__global__ void __launch_bounds__(PXI_THREADS, 2)
proto_bwd_x6i_kernel(const float *__restrict__ feat, long long feat_bstride, int C, int N, int K, float scale,
                     const unsigned char *__restrict__ gimg, const float *__restrict__ gginv,
                     const float *__restrict__ soft, const float *__restrict__ dup, float *__restrict__ dfeat)
{
    constexpr int NROW = CT * 32;                        // image rows = channels
    constexpr int NP = Mx<PREC>::NP, PARTS = 4 * NP;     // 16-byte parts of an image row that this mode reads
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    extern __shared__ __attribute__((aligned(16))) float smem[];
    unsigned char *img = reinterpret_cast<unsigned char *>(smem);            // [2][NROW][PX_ROW]
    float *ginv = reinterpret_cast<float *>(img + 2 * NROW * PX_ROW);         // [NT*32]
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int j = lane & 31, h = lane >> 5;
    const int b = blockIdx.y;
    const int n = blockIdx.x * (PXI_THREADS / 2) + wv * 32 + j;
    const int nc = min(n, N - 1);
    const float *fb = feat + (size_t)b * feat_bstride;
    const float *sb = soft ? soft + (size_t)b * K * N + nc : nullptr;
    const float *ub = dup + (size_t)b * K * N + nc;
    // the workspace holds ceil(K / 32) * 32 inverse norms (proto_bwd_image_bytes); the tiles past them are never used: zero
    for (int w = tid; w < NT * 32; w += PXI_THREADS) ginv[w] = w < ((K + 31) / 32) * 32 ? gginv[w] : 0.f;

    float dot = 0.f;
    if (MODE == MODE_SOFTMAX) {
#pragma unroll 4
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int w = t * 32 + acc_row(r, h);
                if (w < K) dot = fmaf(sb[(size_t)w * N], ub[(size_t)w * N], dot);
            }
        dot += __shfl_xor(dot, 32, 64);
    }

    // a tile's LDS image ([NROW][PX_ROW] bytes) arrives by LDS-DMA (global_load_lds: no VGPR destination - the kernel sits at its
    // register budget): 1-KB pieces, lane l of piece i fills image bytes 1024 i + 16 l .. + 15 = (row, 16-byte part) and reads that
    // part of the 192-byte global row (the pad part of a 208-byte LDS row re-reads part 0)
    constexpr int NPIECE = (NROW * PX_ROW + 1023) / 1024, ND = (NPIECE + PXI_THREADS / 64 - 1) / (PXI_THREADS / 64);
    static_assert((NROW * PX_ROW) % 1024 == 0, "the tile image is a whole number of DMA pieces");
    unsigned soff[ND];
#pragma unroll
    for (int u = 0; u < ND; ++u) {
        const int d = ((wv * ND + u) * 64 + lane) * 16, row = d / PX_ROW, within = d - row * PX_ROW;
        soff[u] = (unsigned)(row * PXI_GROW + (within < PXI_GROW ? within : 0));
    }
    const int wvu = __builtin_amdgcn_readfirstlane(wv);
    auto dma = [&](int t, unsigned char *dst_img) {
        const unsigned char *src = gimg + (size_t)t * C * PXI_GROW;
#pragma unroll
        for (int u = 0; u < ND; ++u)
            if (wvu * ND + u < NPIECE) px_dma16_off(src, soff[u], dst_img + (wvu * ND + u) * 1024);
    };
    float sv[16], uv[16];
    auto load = [&](int t) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int w = t * 32 + acc_row(r, h);
            const bool ok = w < K;
            uv[r] = ok ? ub[(size_t)w * N] : 0.f;
            sv[r] = (MODE == MODE_SOFTMAX && ok) ? sb[(size_t)w * N] : 0.f;
        }
    };
    uint4 bp[2][NP];                                     // pieces of d[t]: K-step s <-> registers 8 s .. 8 s + 7
    auto split_d = [&](int t) {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            float v8[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int r = 8 * s + e, w = t * 32 + acc_row(r, h);
                const float v = (MODE == MODE_SOFTMAX) ? sv[r] * (uv[r] - dot) : uv[r];
                v8[e] = v * (scale * ginv[min(w, NT * 32 - 1)]);
            }
            mm_pieces8<PREC>(v8, bp[s]);
        }
    };

    f32x16 G[CT];                               // dfhat, rows = channel 32 ct + acc_row(r, h), column = point
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int r = 0; r < 16; ++r) G[ct][r] = 0.f;
    const int ntile = (K + 31) / 32;
    dma(0, img);
    load(0);
    __syncthreads();                                     // ginv
    split_d(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's pieces of tile 0 have landed ...
    __syncthreads();                                     // ... and everyone's
    for (int t = 0; t < ntile; ++t) {
        const unsigned char *cur = img + (t & 1) * NROW * PX_ROW;
        if (t + 1 < ntile) {
            dma(t + 1, img + ((t + 1) & 1) * NROW * PX_ROW);       // (the other image was last read before the previous barrier)
            load(t + 1);
        }
        const unsigned char *pa = cur + j * PX_ROW + h * 16;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
                uint4 af[NP];
#pragma unroll
                for (int pl = 0; pl < NP; ++pl)
                    af[pl] = *reinterpret_cast<const uint4 *>(pa + ct * 32 * PX_ROW + pl * 64 + s * 32);
                mm_mac<PREC>(af, bp[s], G[ct]);
            }
        }
        if (t + 1 < ntile) split_d(t + 1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    // normalisation backward, in-lane: |f|^2 and <f, dfhat> over this lane's channels, then the other lane half's
    const unsigned voff = (unsigned)(4 * h * N + nc) * 4u;                   // + (32 ct + acc_row(r, 0)) N 4 bytes
    const __amdgpu_buffer_rsrc_t rs_f = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(fb), 0, (int)min((size_t)C * N * 4, (size_t)0x7fffffff), 0x00020000);
    float fsq = 0.f, fd = 0.f;
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float fv = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_f, voff, (ct * 32 + acc_row(r, 0)) * N * 4, 0));
            fsq = fmaf(fv, fv, fsq);
            fd = fmaf(fv, G[ct][r], fd);
        }
    fsq += __shfl_xor(fsq, 32, 64);
    fd += __shfl_xor(fd, 32, 64);
    const float finv = 1.0f / fmaxf(sqrtf(fsq), NORM_EPS);
    const float coef = fd * finv * finv;         // <fhat, dfhat> / |f|
    const __amdgpu_buffer_rsrc_t rs_o = __builtin_amdgcn_make_buffer_rsrc(dfeat + (size_t)b * C * N, 0, (int)min((size_t)C * N * 4, (size_t)0x7fffffff), 0x00020000);
    const int ovoff = (n < N) ? (int)voff : 0x7fffffff;                     // a lane past N: dropped by the range check
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int so = (ct * 32 + acc_row(r, 0)) * N * 4;
            const float fv = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_f, voff, so, 0));
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, (G[ct][r] - fv * coef) * finv), rs_o, ovoff, so, 0);
        }
}

static size_t proto_bwd_image_bytes(int C, int K)
{
    const int ntile = (K + 31) / 32;
    return align_up((size_t)ntile * C * PXI_GROW, 256) + align_up(sizeof(float) * ntile * 32, 256);
}

template <int NT, int CT, int MODE, int PREC>
static int launch_proto_bwd_x6i(const float *feat, long long bs, int B, int C, int N, const float *proto, int K, float scale,
                                const float *soft, const float *dup, float *dfeat, void *ws, hipStream_t st, const char *who)
{
    unsigned char *gimg = static_cast<unsigned char *>(ws);
    const int ntile = (K + 31) / 32;
    float *gginv = reinterpret_cast<float *>(gimg + align_up((size_t)ntile * C * PXI_GROW, 256));
    ProfScope prof(3, st);
    hipLaunchKernelGGL(proto_words_image_kernel<PREC>, dim3((ntile * 32 * C + 255) / 256), dim3(256), 0, st, proto, K, C,
                       reinterpret_cast<unsigned short *>(gimg), gginv);
    GFS3D_LAUNCH_CHECK("proto_words_image_kernel");
    const size_t lds = (size_t)2 * CT * 32 * PX_ROW + sizeof(float) * NT * 32;
    GFS3D_ENSURE_LDS((proto_bwd_x6i_kernel<NT, CT, MODE, PREC>), lds, who);
    dim3 grid((N + PXI_THREADS / 2 - 1) / (PXI_THREADS / 2), B);
    hipLaunchKernelGGL((proto_bwd_x6i_kernel<NT, CT, MODE, PREC>), grid, dim3(PXI_THREADS), lds, st, feat, bs, C, N, K, scale, gimg, gginv, soft, dup,
                       dfeat);
    GFS3D_LAUNCH_CHECK(who);
    return GFS3D_OK;
}

// shapes the image form covers: shared words (P == 1), more than 32 of them, C in {64, 128, 192}
template <int MODE>
static int dispatch_proto_bwd_image(const float *feat, long long bs, int B, int C, int N, const float *proto, int K, float scale,
                                    const float *soft, const float *dup, float *dfeat, void *ws, hipStream_t st, const char *who, int prec)
{
    const int nt = (K + 31) / 32;
#define GFS3D_PXI(NT_, CT_)                                                                                                                  \
    return prec != PREC_X6 ? launch_proto_bwd_x6i<NT_, CT_, MODE, PREC_BF16>(feat, bs, B, C, N, proto, K, scale, soft, dup, dfeat, ws, st, who) \
                           : launch_proto_bwd_x6i<NT_, CT_, MODE, PREC_X6>(feat, bs, B, C, N, proto, K, scale, soft, dup, dfeat, ws, st, who)
    if (C == 64) { if (nt <= 2) GFS3D_PXI(2, 2); if (nt <= 4) GFS3D_PXI(4, 2); GFS3D_PXI(8, 2); }
    if (C == 128) { if (nt <= 2) GFS3D_PXI(2, 4); if (nt <= 4) GFS3D_PXI(4, 4); GFS3D_PXI(8, 4); }
    if (nt <= 2) GFS3D_PXI(2, 6); if (nt <= 4) GFS3D_PXI(4, 6); GFS3D_PXI(8, 6);
#undef GFS3D_PXI
}

template <int MODE>
static int dispatch_proto_bwd(const float *feat, long long bs, int B, int C, int N, const float *proto, int P, int K, float scale,
                              const float *soft, const float *dup, float *dfeat, hipStream_t st, const char *who, int prec = PREC_X6)
{
    const int nt = (K + 31) / 32, ct = (C + 63) / 64 * 2;
    static const bool x6 = [] { const char *e = getenv("GFS3D_PROTO_X6"); return !(e && e[0] == '0'); }();     // tuning / A-B only
    if (x6 && nt >= 2 && C % 64 == 0 && C <= 192 && (reinterpret_cast<uintptr_t>(proto) & 15) == 0) {
#define GFS3D_PX(NT_, CT_) return launch_proto_bwd_x6<NT_, CT_, MODE>(feat, bs, B, C, N, proto, P, K, scale, soft, dup, dfeat, st, who, prec)
        if (C == 64) { if (nt <= 2) GFS3D_PX(2, 2); if (nt <= 4) GFS3D_PX(4, 2); GFS3D_PX(8, 2); }
        if (C == 128) { if (nt <= 2) GFS3D_PX(2, 4); if (nt <= 4) GFS3D_PX(4, 4); GFS3D_PX(8, 4); }
        if (nt <= 2) GFS3D_PX(2, 6); if (nt <= 4) GFS3D_PX(4, 6); GFS3D_PX(8, 6);
#undef GFS3D_PX
    }
#define GFS3D_PB(NT_, CT_) return launch_proto_bwd<NT_, CT_, MODE>(feat, bs, B, C, N, proto, P, K, scale, soft, dup, dfeat, st, who)
    if (ct <= 2) { if (nt <= 1) GFS3D_PB(1, 2); if (nt <= 2) GFS3D_PB(2, 2); if (nt <= 4) GFS3D_PB(4, 2); GFS3D_PB(8, 2); }
    if (ct <= 4) { if (nt <= 1) GFS3D_PB(1, 4); if (nt <= 2) GFS3D_PB(2, 4); if (nt <= 4) GFS3D_PB(4, 4); GFS3D_PB(8, 4); }
    if (nt <= 1) GFS3D_PB(1, 6); if (nt <= 2) GFS3D_PB(2, 6); if (nt <= 4) GFS3D_PB(4, 6); GFS3D_PB(8, 6);
#undef GFS3D_PB
}

// ---------------------------------------------------------------------------
// Backward of get_pred with respect to the PROTOTYPES (the autograd of capl.py:308-322 for `proto`):
//   dphat[k][c] = scale * sum_n dout[k][n] fhat[c][n]      fhat = f / max(|f|, eps), per point
//   dproto[k]   = (dphat[k] - phat[k] <phat[k], dphat[k]>) / max(|proto[k]|, eps)
// a (K x points) . (points x C) contraction with K <= 32, C <= 192: v_mfma_f32_32x32x2_f32, A = dout rows (k), B = fhat
// (channel columns), the points are the K dimension.  A block walks PB_STAGES stages of 64 points of one batch item
// (both operands staged through LDS, the point norms taken from the staged columns) and writes ONE partial (32 x C);
// proto_bwd_proto_reduce_kernel adds the partials of a prototype set in a fixed order (bit-reproducible, no atomics)
// and applies the normalisation backward.
// ---------------------------------------------------------------------------
constexpr int PB_NP = 64, PB_LD = 68, PB_STAGES = 2;
typedef float pb_f32x4 __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(256)
proto_bwd_proto_kernel(const float *__restrict__ feat, long long feat_bstride, int C, int N, int K, float scale,
                       const float *__restrict__ dout, float *__restrict__ part)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *dt = smem;                          // [32][PB_LD]  scale * dout rows (k >= K: zero)
    float *ft = dt + 32 * PB_LD;               // [CT * 32][PB_LD] feature rows, normalised in place
    const int CT = (C + 31) / 32;
    float *finv = ft + CT * 32 * PB_LD;        // [64]
    float *psq = finv + 64;                    // [4][64] partial column sums of squares
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int i = lane & 31, h = lane >> 5;
    const int b = blockIdx.y, chunk = blockIdx.x;
    const float *fb = feat + (size_t)b * feat_bstride;
    const float *db = dout + (size_t)b * K * N;
    f32x16 acc[2];
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[u][r] = 0.f;
    for (int st = 0; st < PB_STAGES; ++st) {
        const int n0 = (chunk * PB_STAGES + st) * PB_NP;
        if (n0 >= N) break;
        __syncthreads();
        // staging: 16-byte loads where the rows allow it (N % 4 == 0: every row start is 16-byte aligned), all of a thread's
        // loads in flight before the first LDS store; the point norms from 4 partial column sums per point
        const bool vec = (N & 3) == 0 && n0 + PB_NP <= N;
        if (vec) {
            const int q = tid & 15, r0 = tid >> 4;                 // float4 column q of rows r0, r0 + 16, ...
            pb_f32x4 dv[2], fv[12];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int k = r0 + 16 * u;
                dv[u] = k < K ? *reinterpret_cast<const pb_f32x4 *>(db + (size_t)k * N + n0 + 4 * q) : pb_f32x4{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int u = 0; u < 12; ++u) {
                const int c = r0 + 16 * u;
                fv[u] = (u < 2 * CT && c < C) ? *reinterpret_cast<const pb_f32x4 *>(fb + (size_t)c * N + n0 + 4 * q) : pb_f32x4{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) *reinterpret_cast<pb_f32x4 *>(dt + (r0 + 16 * u) * PB_LD + 4 * q) = dv[u] * scale;
#pragma unroll
            for (int u = 0; u < 12; ++u)
                if (u < 2 * CT) *reinterpret_cast<pb_f32x4 *>(ft + (r0 + 16 * u) * PB_LD + 4 * q) = fv[u];
        } else {
            for (int e = tid; e < 32 * PB_NP; e += 256) {
                const int k = e >> 6, n = e & 63;
                dt[k * PB_LD + n] = (k < K && n0 + n < N) ? scale * db[(size_t)k * N + n0 + n] : 0.f;
            }
            for (int e = tid; e < CT * 32 * PB_NP; e += 256) {
                const int c = e >> 6, n = e & 63;
                ft[c * PB_LD + n] = (c < C && n0 + n < N) ? fb[(size_t)c * N + n0 + n] : 0.f;
            }
        }
        __syncthreads();
        {
            const int n = tid & 63, part = tid >> 6;               // channels part, part + 4, ...
            float sq = 0.f;
            for (int c = part; c < C; c += 4) sq = fmaf(ft[c * PB_LD + n], ft[c * PB_LD + n], sq);
            psq[part * 64 + n] = sq;
        }
        __syncthreads();
        if (tid < PB_NP) finv[tid] = 1.0f / fmaxf(sqrtf((psq[tid] + psq[64 + tid]) + (psq[128 + tid] + psq[192 + tid])), NORM_EPS);
        __syncthreads();
        for (int e = tid; e < CT * 32 * PB_NP; e += 256) {
            const int c = e >> 6, n = e & 63;
            ft[c * PB_LD + n] *= finv[n];
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int ct = w + 4 * u;
            if (ct < CT) {
                const float *arow = dt + i * PB_LD, *brow = ft + (ct * 32 + i) * PB_LD;
#pragma unroll
                for (int q = 0; q < PB_NP / 4; ++q) {
                    const float4 a4 = *reinterpret_cast<const float4 *>(arow + 4 * q);
                    const float4 b4 = *reinterpret_cast<const float4 *>(brow + 4 * q);
                    acc[u] = __builtin_amdgcn_mfma_f32_32x32x2f32(h ? a4.y : a4.x, h ? b4.y : b4.x, acc[u], 0, 0, 0);
                    acc[u] = __builtin_amdgcn_mfma_f32_32x32x2f32(h ? a4.w : a4.z, h ? b4.w : b4.z, acc[u], 0, 0, 0);
                }
            }
        }
    }
    // C/D layout: column = channel ct * 32 + i, rows = prototypes acc_row(r, h)
    float *out = part + ((size_t)b * gridDim.x + chunk) * 32 * CT * 32;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int ct = w + 4 * u;
        if (ct < CT)
#pragma unroll
            for (int r = 0; r < 16; ++r) out[(size_t)acc_row(r, h) * CT * 32 + ct * 32 + i] = acc[u][r];
    }
}
// one block per (prototype set p, prototype k): fixed-order sum of the partials (G thread groups take every G-th partial,
// the groups are added in order), then the normalisation backward
__global__ void __launch_bounds__(1024)
proto_bwd_proto_reduce_kernel(const float *__restrict__ part, int nchunk, int B, int P, int K, int C, const float *__restrict__ proto,
                              float *__restrict__ dproto)
{
    __shared__ float red[1024];
    const int k = blockIdx.x, p = blockIdx.y;
    const int CT = (C + 31) / 32, LDC = CT * 32;
    const int G = 1024 / LDC, g = threadIdx.x / LDC, c = threadIdx.x - g * LDC;
    const int b0 = P > 1 ? p : 0, nb = P > 1 ? 1 : B;
    const int total = nb * nchunk;                     // partial (b0 + q / nchunk, q % nchunk), q = 0 .. total - 1
    float s = 0.f;
    if (g < G)
        for (int q = g; q < total; q += G) s += part[(((size_t)(b0 + q / nchunk) * nchunk + q % nchunk) * 32 + k) * LDC + c];
    red[threadIdx.x] = s;
    __syncthreads();
    float tot = 0.f;
    if ((int)threadIdx.x < LDC)
        for (int gg = 0; gg < G; ++gg) tot += red[gg * LDC + threadIdx.x];
    __syncthreads();
    const bool own = (int)threadIdx.x < C;             // thread c < C owns channel c from here on
    const float pv = own ? proto[((size_t)p * K + k) * C + threadIdx.x] : 0.f;
    auto block_sum = [&](float v) {
        red[threadIdx.x] = v;
        __syncthreads();
        for (int o = 512; o > 0; o >>= 1) {
            if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
            __syncthreads();
        }
        const float t = red[0];
        __syncthreads();
        return t;
    };
    const float pn = fmaxf(sqrtf(block_sum(pv * pv)), NORM_EPS);
    const float ph = pv / pn;
    const float dotv = block_sum(own ? ph * tot : 0.f);
    if (own) dproto[((size_t)p * K + k) * C + threadIdx.x] = (tot - ph * dotv) / pn;
}

// ---------------------------------------------------------------------------
// k-means assignment (Lloyd step), first version: 64 points per workgroup in LDS,
// centroids streamed through LDS in tiles of 64; thread (p, q) scores point p
// against centroids q, q+4, ... of the tile.  sums/counts accumulate with float
// atomics (256-B contiguous rows).
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
kmeans_assign_kernel(const float *__restrict__ pts, long long M, int C, const float *__restrict__ cent, int K,
                     int32_t *__restrict__ label, float *__restrict__ sums, float *__restrict__ counts,
                     double *__restrict__ inertia)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int LD = C + 1;
    float *ps = smem;                 // [64][LD]
    float *cs = ps + 64 * LD;         // [64][LD]
    float *cn = cs + 64 * LD;         // [64] |c|^2
    float *bd = cn + 64;              // [4][64] best distance per (q, p)
    int *bi = reinterpret_cast<int *>(bd + 256);   // [4][64]
    const int tid = threadIdx.x, p = tid & 63, q = tid >> 6;
    const long long m0 = (long long)blockIdx.x * 64;
    for (int e = tid; e < 64 * C; e += 256) {
        const int pp = e / C, c = e - pp * C;
        ps[pp * LD + c] = (m0 + pp < M) ? pts[(size_t)(m0 + pp) * C + c] : 0.f;
    }
    float best = __builtin_inff();
    int bidx = 0;
    for (int k0 = 0; k0 < K; k0 += 64) {
        __syncthreads();
        for (int e = tid; e < 64 * C; e += 256) {
            const int kk = e / C, c = e - kk * C;
            cs[kk * LD + c] = (k0 + kk < K) ? cent[(size_t)(k0 + kk) * C + c] : 0.f;
        }
        __syncthreads();
        if (tid < 64) {
            float s = 0.f;
            for (int c = 0; c < C; ++c) s = fmaf(cs[tid * LD + c], cs[tid * LD + c], s);
            cn[tid] = s;
        }
        __syncthreads();
        for (int kk = q; kk < 64 && k0 + kk < K; kk += 4) {
            float dot = 0.f;
            for (int c = 0; c < C; ++c) dot = fmaf(ps[p * LD + c], cs[kk * LD + c], dot);
            const float d = cn[kk] - 2.f * dot;          // + |p|^2 (constant per point)
            if (d < best) { best = d; bidx = k0 + kk; }
        }
    }
    bd[q * 64 + p] = best;
    bi[q * 64 + p] = bidx;
    __syncthreads();
    if (q == 0) {
        for (int qq = 1; qq < 4; ++qq) {
            const float d = bd[qq * 64 + p];
            const int i = bi[qq * 64 + p];
            if (d < best || (d == best && i < bidx)) { best = d; bidx = i; }
        }
        bi[p] = bidx;
        float pn = 0.f;
        for (int c = 0; c < C; ++c) pn = fmaf(ps[p * LD + c], ps[p * LD + c], pn);
        bd[p] = best + pn;
        if (m0 + p < M) {
            label[m0 + p] = bidx;
            atomicAdd(&counts[bidx], 1.0f);
        }
    }
    __syncthreads();
    // accumulate centroid sums: wave q handles points q, q+4, ...; lanes over channels
    for (int pp = q; pp < 64 && m0 + pp < M; pp += 4) {
        const int li = bi[pp];
        for (int c = p; c < C; c += 64) atomicAdd(&sums[(size_t)li * C + c], ps[pp * LD + c]);
    }
    if (inertia && tid < 64) {
        double v = (m0 + tid < M) ? (double)fmaxf(bd[tid], 0.f) : 0.0;
        v = wave_sum(v);
        if (tid == 0) atomicAdd(inertia, v);
    }
}


// ---------------------------------------------------------------------------
// k-means on the matrix pipe (K <= 256 centroids).
//   assign:  D[k][m] = <c_k, p_m>  exactly like the prototype kernel (A = centroid rows from the
//            padded LDS chunk, B = points; points are ROW-major here, so a lane reads 16-B pieces of
//            its own row); epilogue: arg-min_k (|c_k|^2 - 2 D), label, count, inertia.
//   sums:    S[k][c] = sum_m [label_m == k] p_m[c]  as a one-hot GEMM: A[k][m] is built on the fly
//            from the labels (LDS), B[m][c] are 128-B coalesced row pieces; every workgroup owns one
//            32-channel column tile and 4096 points, 2 centroid row-tiles per wave; partial tiles are
//            added to the global sums with 128-B float-atomic rows.
// ---------------------------------------------------------------------------
template <int NT>
__global__ void __launch_bounds__(256)
kmeans_assign_mfma_kernel(const float *__restrict__ pts, long long M, int C, const float *__restrict__ cent, int K,
                          int32_t *__restrict__ label, float *__restrict__ counts, double *__restrict__ inertia)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *gs = smem;                          // [NT*32][PR_LD]
    float *cn = gs + NT * 32 * PR_LD;          // [NT*32] |c_k|^2
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int j = lane & 31, h = lane >> 5;
    const long long m = (long long)blockIdx.x * 128 + wv * 32 + j;
    const long long mc = m < M ? m : M - 1;
    const float *prow = pts + (size_t)mc * C;

    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    float psq = 0.f, csq = 0.f;
    for (int c0 = 0; c0 < C; c0 += PR_CH) {
        __syncthreads();
        for (int e = tid; e < NT * 32 * PR_CH; e += 256) {
            const int w = e >> 6, cc = e & 63;
            gs[w * PR_LD + cc] = (w < K && c0 + cc < C) ? cent[(size_t)w * C + c0 + cc] : 0.f;
        }
        float f[32];
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const int c = c0 + 8 * t + 4 * h;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float v = (c + u < C) ? prow[c + u] : 0.f;
                f[4 * t + u] = v;
                psq = fmaf(v, v, psq);
            }
        }
        __syncthreads();
        if (tid < NT * 32) {
            const float *row = gs + tid * PR_LD;
#pragma unroll 8
            for (int cc = 0; cc < PR_CH; ++cc) csq = fmaf(row[cc], row[cc], csq);
        }
#pragma unroll
        for (int t = 0; t < 8; ++t) {
#pragma unroll
            for (int wt = 0; wt < NT; ++wt) {
                const float4 a = *reinterpret_cast<const float4 *>(gs + (wt * 32 + j) * PR_LD + 8 * t + 4 * h);
                acc[wt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, f[4 * t + 0], acc[wt], 0, 0, 0);
                acc[wt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, f[4 * t + 1], acc[wt], 0, 0, 0);
                acc[wt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, f[4 * t + 2], acc[wt], 0, 0, 0);
                acc[wt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, f[4 * t + 3], acc[wt], 0, 0, 0);
            }
        }
    }
    __syncthreads();
    if (tid < NT * 32) cn[tid] = csq;
    __syncthreads();
    psq += __shfl_xor(psq, 32, 64);
    float best = __builtin_inff();
    int bk = 0x7fffffff;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int w = t * 32 + acc_row(r, h);
            const float d = cn[w] - 2.f * acc[t][r];            // + |p|^2, constant per point
            if (w < K && (d < best || (d == best && w < bk))) { best = d; bk = w; }
        }
    {
        const float ob = __shfl_xor(best, 32, 64);
        const int ok = __shfl_xor(bk, 32, 64);
        if (ob < best || (ob == best && ok < bk)) { best = ob; bk = ok; }
    }
    double contrib = 0.0;
    if (h == 0 && m < M) {
        label[m] = bk;
        atomicAdd(&counts[bk], 1.0f);
        contrib = (double)fmaxf(best + psq, 0.f);
    }
    if (inertia) {
        contrib = wave_sum(contrib);
        if (lane == 0) atomicAdd(inertia, contrib);
    }
}

// The assignment step on the bf16x6 main loop of proto_x6_kernel (C % 32 == 0): 8 waves = 256 points per workgroup.
//   PREC_X6, LIST = false   every point scored at fp32 accuracy (the form gfs3d_kmeans_assign_* ships)
//   PREC_X3, LIST = false   SCREEN (gfs3d_kmeans_assign_fast_f32): hi + mid bf16 operands, three products - half the matrix work
//                           and split arithmetic - with the error bound E of the screened kNN (knn.hip); a point whose two
//                           best scores are more than 2 E apart has its exact arg-min decided, the others go to `flist`
//   PREC_X6, LIST = true    the exact pass over `flist` (grid sized for M, blocks past the list's end leave at once)
// so the labels - and with them the sums and counts - are those of the fp32-accurate form; only the inertia of the
// screened points is the approximate score (within E of the exact one).
struct KmFast { int *flist; int *fcount; float erel, eabs; };
template <int NT, int PREC, bool LIST>
__global__ void __launch_bounds__(PX_THREADS)
kmeans_assign_x6_kernel(const float *__restrict__ pts, long long M, int C, const float *__restrict__ cent, int K,
                        int32_t *__restrict__ label, float *__restrict__ counts, double *__restrict__ inertia, const KmFast fa)
{
    constexpr int NROW = NT * 32;
    constexpr bool SCREEN = PREC == PREC_X3;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    unsigned char *img = reinterpret_cast<unsigned char *>(smem);
    float *cn = reinterpret_cast<float *>(img + 2 * NROW * PX_ROW);           // [NROW] |c_k|^2
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int j = lane & 31, h = lane >> 5;
    long long total = M;
    if (LIST) total = (long long)*fa.fcount;
    // (the exact pass runs a FIXED grid of a few hundred blocks that stride over the list: a grid sized for M would launch
    // thousands of blocks that leave at once - 54 us for an empty list)
    auto tile = [&](const long long blk) {
    const long long m0 = blk * 256 + wv * 32;
    float *wtile = cn + NROW + wv * 32 * XS_LD;                               // wave-private [32][XS_LD]
    int *hist = reinterpret_cast<int *>(cn + NROW + 8 * 32 * XS_LD);          // [NROW] members of this workgroup per centroid
    if (tid < NROW) hist[tid] = 0;                                            // (ordered by the barriers of the main loop)
    f32x16 acc[NT];
    float psq;
    const long long mb = m0 < total ? m0 : total - 1;                          // (a wave past the end recomputes the last point, stores nothing)
    const int nrows = (int)min((long long)32, total - mb);
    if (LIST) x6_scores<NT, true, PREC>(pts, (size_t)C, nrows, wtile, cent, C, K, img, cn, acc, psq, fa.flist + mb);
    else      x6_scores<NT, true, PREC>(pts + (size_t)mb * C, (size_t)C, nrows, wtile, cent, C, K, img, cn, acc, psq);
    psq += __shfl_xor(psq, 32, 64);
    const bool live = h == 0 && m0 + j < total;
    const long long m = LIST ? (long long)fa.flist[min(m0 + j, total - 1)] : m0 + j;
    float best = __builtin_inff(), second = __builtin_inff();
    int bk = 0x7fffffff;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int w = t * 32 + acc_row(r, h);               // increasing within a lane: '<' keeps the lowest index of a tie
            if constexpr (SCREEN) {                                  // (branch-free: as `if`s this became 128 guarded blocks with spills)
                const float d0 = cn[w] - 2.f * acc[t][r];                // + |p|^2, constant per point (cn holds all NT 32 rows)
                const float d = (w < K) ? d0 : __builtin_inff();         // (select AFTER the read: a guarded read became a branch per row)
                const bool lt = d < best;
                second = fminf(second, lt ? best : d);
                bk = lt ? w : bk;
                best = lt ? d : best;
            } else {
                const float d = cn[w] - 2.f * acc[t][r];
                if (w < K && d < best) { best = d; bk = w; }
            }
        }
    {
        const float ob = __shfl_xor(best, 32, 64);
        const int ok = __shfl_xor(bk, 32, 64);
        if (SCREEN) second = fminf(fminf(second, __shfl_xor(second, 32, 64)), fmaxf(best, ob));
        if (ob < best || (ob == best && ok < bk)) { best = ob; bk = ok; }
    }
    bool decided = true;
    if (SCREEN) {
        float cmx = 0.f;                                  // max_k |c_k|^2
        for (int w = lane; w < NROW; w += 64) cmx = fmaxf(cmx, w < K ? cn[w] : 0.f);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) cmx = fmaxf(cmx, __shfl_xor(cmx, o, 64));
        const float E = fa.erel * sqrtf(psq * cmx) + fa.eabs * (psq + cmx);
        decided = second - best > 2.0f * E;               // (K == 1: second = +inf)
        if (live && !decided) fa.flist[atomicAdd(fa.fcount, 1)] = (int)m;      // (order of the list: irrelevant, integer atomics)
    }
    double contrib = 0.0;
    if (live && decided) {
        label[m] = bk;
        atomicAdd(&hist[bk], 1);
        contrib = (double)fmaxf(best + psq, 0.f);
    }
    // counts: one global float add per workgroup and non-empty centroid (the K counters share 7 cache lines: per-point
    // adds from 300 k points queue up on them at the memory side)
    __syncthreads();
    if (tid < K && hist[tid] != 0) atomicAdd(&counts[tid], (float)hist[tid]);
    if (inertia) {      // ONE fp64 atomic per workgroup: same-address fp64 atomics serialise at ~14 ns each
        contrib = wave_sum(contrib);
        __syncthreads();                                 // every wave is done with cn
        double *red = reinterpret_cast<double *>(cn);
        if (lane == 0) red[wv] = contrib;
        __syncthreads();
        if (tid == 0) {
            double tot = 0.0;
#pragma unroll
            for (int w = 0; w < PX_THREADS / 64; ++w) tot += red[w];
            atomicAdd(inertia, tot);
        }
    }
    };   // tile
    if constexpr (!LIST) tile((long long)blockIdx.x);
    else
        for (long long blk = blockIdx.x; blk * 256 < total; blk += gridDim.x) {
            if (blk != (long long)blockIdx.x) __syncthreads();                  // (the previous tile's LDS reads are done)
            tile(blk);
        }
}

// ---------------------------------------------------------------------------------------------------------
// The SCREEN as a persistent, software-pipelined kernel (round 6).  kmeans_assign_x6_kernel<NT, PREC_X3, false> ran one 256-point
// tile per workgroup at one workgroup per CU: every tile paid the latency of its first loads with nothing in flight, re-read and
// re-split the centroid chunks (half of the vector work of its main loop), requested point rows ONE 32-channel step (~1 us of
// MFMAs) ahead of their use - 32 KB in flight per CU against the ~64 KB that 8 TB/s x 2 us need - and ran the 112-score epilogue
// of its eight waves with no load outstanding: 717 us for 1.61 GB, 2.3 TB/s, neither roof.  Here
//   * km_cent_image_kernel writes, once per call, the centroids' operand image [chunk][NROW rows][hi 64 B | mid 64 B | pad 16 B]
//     (144-B rows: 9 sixteen-byte slots, odd => conflict-free ds_read_b128) and |c_k|^2; a step's image chunk travels
//     global (L2) -> LDS by LDS-DMA (global_load_lds: no registers, no vector instructions);
//   * a workgroup walks tiles blockIdx.x, + gridDim.x, ...: the chunk steps of consecutive tiles form ONE pipeline - at step g the
//     image of step g + 1 and the point rows of step g + 2 are requested (the DMA first: loads return in order, so the wait for the
//     image leaves exactly the newest four row loads outstanding), and the epilogue of a tile runs under the first loads of the next;
//   * per-centroid counts stay in LDS for all of a workgroup's tiles (K global adds per workgroup, not per tile), the inertia
//     in a register; undecided points are appended to the list with one atomic per wave.
// Same scores, same error bound, same decisions as the one-tile form (|c_k|^2 is summed in a different order: inside eabs).
// ---------------------------------------------------------------------------------------------------------
constexpr int KI_ROWB = 144;            // hi | mid | pad
static inline int ki_rows(int K) { return (K + 15) / 16 * 16; }
template <int PREC>
__global__ void __launch_bounds__(256)
km_cent_image_kernel(const float *__restrict__ cent, int K, int C, int nrow, unsigned char *__restrict__ gimg, float *__restrict__ gcn)
{
    static_assert(Mx<PREC>::NP * 64 + 16 == KI_ROWB, "image rows: the pieces of 32 channels + one pad slot");
    constexpr int NP = Mx<PREC>::NP;
    const int nchunk = C / 32;
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e < nchunk * nrow * 8) {
        const int c4 = e & 7, w = (e >> 3) % nrow, ch = (e >> 3) / nrow;
        float v4[4] = {0.f, 0.f, 0.f, 0.f};
        if (w < K) {
            const float4 v = *reinterpret_cast<const float4 *>(cent + (size_t)w * C + ch * 32 + 4 * c4);
            v4[0] = v.x; v4[1] = v.y; v4[2] = v.z; v4[3] = v.w;
        }
        uint2 pc[NP];
        mm_pieces4<PREC>(v4, pc);
        unsigned char *row = gimg + ((size_t)ch * nrow + w) * KI_ROWB;
#pragma unroll
        for (int q = 0; q < NP; ++q) *reinterpret_cast<uint2 *>(row + 64 * q + 8 * c4) = pc[q];
        if (c4 < 2) *reinterpret_cast<uint2 *>(row + 64 * NP + 8 * c4) = make_uint2(0u, 0u);
    }
    // |c_w|^2: one wave per row (the blocks after the image's; a lane's channels in ascending order, then the wave's sum)
    const int nimg = (nchunk * nrow * 8 + 255) / 256;
    if ((int)blockIdx.x >= nimg) {
        const int w = ((int)blockIdx.x - nimg) * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
        if (w < nrow) {
            float sq = 0.f;
            if (w < K)
                for (int c = lane; c < C; c += 64) sq = fmaf(cent[(size_t)w * C + c], cent[(size_t)w * C + c], sq);
            sq = wave_sum(sq);
            if (lane == 0) gcn[w] = sq;
        }
    }
}

template <int NT>
__global__ void __launch_bounds__(PX_THREADS)
kmeans_screen_img_kernel(const float *__restrict__ pts, long long M, int C, int K, const unsigned char *__restrict__ gimg,
                         const float *__restrict__ gcn, int32_t *__restrict__ label, float *__restrict__ counts,
                         double *__restrict__ inertia, const KmFast fa)
{
    constexpr int PREC = PREC_X3, NP = Mx<PREC>::NP, NROW = NT * 32, IMGB = NROW * KI_ROWB;
    constexpr int NWAVE = PX_THREADS / 64, NPIECE = (IMGB + 1023) / 1024, ND = (NPIECE + NWAVE - 1) / NWAVE;
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    extern __shared__ __attribute__((aligned(16))) float smem[];
    unsigned char *img = reinterpret_cast<unsigned char *>(smem);            // [2][IMGB]
    float *cn = reinterpret_cast<float *>(img + 2 * IMGB);                    // [NROW] |c_k|^2
    int *hist = reinterpret_cast<int *>(cn + NROW);                           // [NROW] members of this workgroup per centroid
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int j = lane & 31, h = lane >> 5;
    float *wtile = reinterpret_cast<float *>(hist + NROW) + wv * 32 * XS_LD;  // wave-private [32][XS_LD]
    const long long ntile = (M + 255) / 256;
    const int nchunk = C / 32;                                                // even (the launcher checks)
    // (rows past K: +inf, their image rows are zero - no per-row `w < K` test in the epilogue, which the compiler would hoist out of the
    // tile loop as 16 NT wave masks)
    for (int w = tid; w < NROW; w += PX_THREADS) { cn[w] = w < K ? gcn[w] : __builtin_inff(); hist[w] = 0; }

    // image chunk -> LDS: 1-KB pieces, lane l of piece i moves image bytes 1024 i + 16 l .. + 15 (global layout == LDS layout)
    unsigned doff[ND];
#pragma unroll
    for (int u = 0; u < ND; ++u) doff[u] = (unsigned)(((wv * ND + u) * 64 + lane) * 16);
    const int wvu = __builtin_amdgcn_readfirstlane(wv);
    auto dma = [&](int chunk) {
        const unsigned char *src = gimg + (size_t)chunk * IMGB;
        unsigned char *dst = img + (chunk & 1) * IMGB;
#pragma unroll
        for (int u = 0; u < ND; ++u)
            if (doff[u] < (unsigned)IMGB) px_dma16_off(src, doff[u], dst + (wvu * ND + u) * 1024);
    };
    // the wave's 32 rows x 32 channels of a step as 128-B row pieces: rows (lane >> 3) + 8 i, 16 B at column 4 (lane & 7)
    auto loadp = [&](const float *base, int nrows, int c0, float (&f)[16]) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = min((lane >> 3) + 8 * i, nrows - 1);
            const f32x4 v = *reinterpret_cast<const f32x4 *>(base + (size_t)row * C + c0 + 4 * (lane & 7));
#pragma unroll
            for (int e = 0; e < 4; ++e) f[4 * i + e] = v[e];
        }
    };
    auto wave_rows = [&](long long blk, const float *&base, int &nrows) {
        const long long m0 = blk * 256 + wv * 32, mb = m0 < M ? m0 : M - 1;     // (a wave past the end re-reads the last point, stores nothing)
        nrows = (int)min((long long)32, M - mb);
        base = pts + (size_t)mb * C;
    };
    uint4 bp[2][NP];                                     // this lane's point: pieces of 8 channels per K-step
    float fsq = 0.f;
    auto split_point = [&](float (&f)[16]) {             // row pieces -> tile -> this lane's 16 channels of point j -> pieces
#pragma unroll
        for (int i = 0; i < 4; ++i)
            *reinterpret_cast<f32x4 *>(wtile + ((lane >> 3) + 8 * i) * XS_LD + 4 * (lane & 7)) = f32x4{f[4 * i], f[4 * i + 1], f[4 * i + 2], f[4 * i + 3]};
        __builtin_amdgcn_wave_barrier();
        float g[16];
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const f32x4 v = *reinterpret_cast<const f32x4 *>(wtile + j * XS_LD + 16 * s + 8 * h + 4 * q);
#pragma unroll
                for (int e = 0; e < 4; ++e) g[8 * s + 4 * q + e] = v[e];
            }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int s = 0; s < 2; ++s) {
#pragma unroll
            for (int e = 0; e < 8; ++e) fsq = fmaf(g[8 * s + e], g[8 * s + e], fsq);
            mm_pieces8<PREC>(g + 8 * s, bp[s]);
        }
    };

    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    double contrib = 0.0;
    float cmx = 0.f;

    auto epilogue = [&](const long long blk) {
        const long long m0 = blk * 256 + wv * 32;
        const float psq = fsq + __shfl_xor(fsq, 32, 64);
        const bool live = h == 0 && m0 + j < M;
        const long long m = m0 + j;
        float best = __builtin_inff(), second = __builtin_inff();
        int bk = 0x7fffffff;
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int w = t * 32 + acc_row(r, h);               // increasing within a lane: '<' keeps the lowest index of a tie
                const float d = cn[w] - 2.f * acc[t][r];             // + |p|^2, constant per point; +inf for w >= K
                const bool lt = d < best;
                second = fminf(second, lt ? best : d);
                bk = lt ? w : bk;
                best = lt ? d : best;
            }
        {
            const float ob = __shfl_xor(best, 32, 64);
            const int ok = __shfl_xor(bk, 32, 64);
            second = fminf(fminf(second, __shfl_xor(second, 32, 64)), fmaxf(best, ob));
            if (ob < best || (ob == best && ok < bk)) { best = ob; bk = ok; }
        }
        const float E = fa.erel * sqrtf(psq * cmx) + fa.eabs * (psq + cmx);
        const bool decided = second - best > 2.0f * E;               // (K == 1: second = +inf)
        const bool undecided = live && !decided;
        const unsigned long long um = __ballot(undecided);
        if (um) {                                                     // one list atomic per wave (order of the list: irrelevant)
            const int first = __builtin_ctzll(um);
            int base = 0;
            if (lane == first) base = atomicAdd(fa.fcount, __builtin_popcountll(um));
            base = __shfl(base, first, 64);
            if (undecided) fa.flist[base + __builtin_popcountll(um & ((1ull << lane) - 1ull))] = (int)m;
        }
        if (live && decided) {
            label[m] = bk;
            atomicAdd(&hist[bk], 1);
            contrib += (double)fmaxf(best + psq, 0.f);
        }
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
        fsq = 0.f;
    };

    float fr[2][16];
    long long blk = blockIdx.x;
    const float *cb, *nb;
    int cr, nr;
    wave_rows(blk, cb, cr);
    dma(0);
    loadp(cb, cr, 0, fr[0]);
    loadp(cb, cr, 32, fr[1]);
    split_point(fr[0]);
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");     // image chunk 0 has landed (this wave's pieces); outstanding: the rows of chunk 1
    __syncthreads();                                     // ... everyone's; cn, hist
    for (int w = lane; w < NROW; w += 64) cmx = fmaxf(cmx, w < K ? cn[w] : 0.f);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) cmx = fmaxf(cmx, __shfl_xor(cmx, o, 64));

    auto step = [&](int c, auto parity) {
        constexpr int P = decltype(parity)::value;
        const bool last = P == 1 && c + 1 == nchunk;     // (an even number of chunks: a tile ends on an odd step)
        dma(last ? 0 : c + 1);                           // into image P ^ 1: last read a barrier ago
        const int c2 = c + 2;
        if (c2 < nchunk) loadp(cb, cr, 32 * c2, fr[P]);  // fr[P]: split into bp during the previous step
        else             loadp(nb, nr, 32 * (c2 - nchunk), fr[P]);
        const unsigned char *pa = img + P * IMGB + j * KI_ROWB + h * 16;
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int wt = 0; wt < NT; ++wt) {
                uint4 af[NP];
#pragma unroll
                for (int pl = 0; pl < NP; ++pl) af[pl] = *reinterpret_cast<const uint4 *>(pa + wt * 32 * KI_ROWB + pl * 64 + s * 32);
                mm_mac<PREC>(af, bp[s], acc[wt]);
            }
        if (last) {
            epilogue(blk);
            cb = nb; cr = nr;
        }
        split_point(fr[P ^ 1]);
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        __syncthreads();
    };
    for (; blk < ntile; blk += gridDim.x) {
        const long long nblk = blk + gridDim.x < ntile ? blk + gridDim.x : ntile - 1;   // (past the end: the last tile again, never scored)
        wave_rows(nblk, nb, nr);
        for (int c = 0; c < nchunk; c += 2) {
            step(c, std::integral_constant<int, 0>{});
            step(c + 1, std::integral_constant<int, 1>{});
        }
    }
    // counts: one global float add per workgroup and non-empty centroid; inertia: one fp64 atomic per workgroup
    if (tid < K && hist[tid] != 0) atomicAdd(&counts[tid], (float)hist[tid]);
    if (inertia) {
        contrib = wave_sum(contrib);
        double *red = reinterpret_cast<double *>(cn);    // (every wave is past its last read of cn: the barrier of the last step)
        if (lane == 0) red[wv] = contrib;
        __syncthreads();
        if (tid == 0) {
            double tot = 0.0;
#pragma unroll
            for (int w = 0; w < NWAVE; ++w) tot += red[w];
            atomicAdd(inertia, tot);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// The screen with the WHOLE operand image resident in LDS (C = 192, K <= 208: 6 chunks x 208 rows x 128 B = 156 KB): no staging
// step, hence no workgroup barrier after the first - the eight waves of a workgroup (one per CU, persistent) walk their own 32-point
// tiles independently.  The pipelined form above keeps its waves in lock step (a barrier per 32-channel step): the two waves of a SIMD
// issue their MFMAs together and their vector work (operand split, epilogue) together, so the matrix pipe idles under every vector
// phase - its time did not change when the points came from the memory-side cache (300 k points: 20.6 us per tile and workgroup;
// 2.1 M from HBM: 17.8).  Here a wave's vector phases fall under the other wave's MFMAs.
//   * image rows are 128 B (hi | mid, no pad); the 16-byte units of a row are XOR-swizzled with (row >> 1) & 7 instead, which keeps
//     ds_read_b128 conflict-free for 16 consecutive rows; the lanes of the last row tile whose rows lie past the image read its last
//     (zero) row, and cn = +inf masks them;
//   * a lane loads ITS point's channels directly (row j, 32 B per K-step half): no transposition tile in LDS; rows two steps ahead;
//   * the compiler counts the outstanding loads itself (no LDS-DMA inside the loop).
// ---------------------------------------------------------------------------------------------------------
constexpr int KR_ROWB = 128;
static inline size_t kr_lds_bytes(int NT, int C, int K) { return (size_t)(C / 32) * ki_rows(K) * KR_ROWB + sizeof(float) * 2 * NT * 32; }
__global__ void __launch_bounds__(256)
km_cent_image_swz_kernel(const float *__restrict__ cent, int K, int C, int imgr, int nrow, unsigned char *__restrict__ gimg, float *__restrict__ gcn)
{
    const int nchunk = C / 32;
    const int e = blockIdx.x * 256 + threadIdx.x;
    const int nimg = (nchunk * imgr * 8 + 255) / 256;
    if (e < nchunk * imgr * 8) {
        const int c4 = e & 7, w = (e >> 3) % imgr, ch = (e >> 3) / imgr;
        float v4[4] = {0.f, 0.f, 0.f, 0.f};
        if (w < K) {
            const float4 v = *reinterpret_cast<const float4 *>(cent + (size_t)w * C + ch * 32 + 4 * c4);
            v4[0] = v.x; v4[1] = v.y; v4[2] = v.z; v4[3] = v.w;
        }
        uint2 pc[2];
        mm_pieces4<PREC_X3>(v4, pc);
        const int key = (w >> 1) & 7;
        unsigned char *row = gimg + ((size_t)ch * imgr + w) * KR_ROWB + 8 * (c4 & 1);
#pragma unroll
        for (int q = 0; q < 2; ++q) *reinterpret_cast<uint2 *>(row + 16 * ((4 * q + (c4 >> 1)) ^ key)) = pc[q];
    }
    if ((int)blockIdx.x >= nimg) {                      // |c_w|^2: one wave per row
        const int w = ((int)blockIdx.x - nimg) * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
        if (w < nrow) {
            float sq = 0.f;
            if (w < K)
                for (int c = lane; c < C; c += 64) sq = fmaf(cent[(size_t)w * C + c], cent[(size_t)w * C + c], sq);
            sq = wave_sum(sq);
            if (lane == 0) gcn[w] = sq;
        }
    }
}

template <int NT>
__global__ void __launch_bounds__(PX_THREADS)
kmeans_screen_res_kernel(const float *__restrict__ pts, long long M, int C, int K, const unsigned char *__restrict__ gimg,
                         const float *__restrict__ gcn, int32_t *__restrict__ label, float *__restrict__ counts,
                         double *__restrict__ inertia, const KmFast fa)
{
    constexpr int PREC = PREC_X3, NP = Mx<PREC>::NP, NROW = NT * 32, NWAVE = PX_THREADS / 64;
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int nchunk = C / 32;                                                // even (the launcher checks)
    const int imgr = (K + 15) / 16 * 16, chunkb = imgr * KR_ROWB, imgb = nchunk * chunkb;     // (a multiple of 1024)
    unsigned char *img = reinterpret_cast<unsigned char *>(smem);            // [nchunk][imgr][128]
    float *cn = reinterpret_cast<float *>(img + imgb);                        // [NROW] -|c_k|^2 / 2 (-inf past K): the accumulators START there
    int *hist = reinterpret_cast<int *>(cn + NROW);                           // [NROW] members of this workgroup per centroid
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int j = lane & 31, h = lane >> 5;
    for (int w = tid; w < NROW; w += PX_THREADS) { cn[w] = w < K ? -0.5f * gcn[w] : -__builtin_inff(); hist[w] = 0; }
    {   // the image, once: 1-KB pieces by LDS-DMA (global layout == LDS layout), wave v moves pieces v, v + 8, ...
        const int wvu = __builtin_amdgcn_readfirstlane(wv);
        for (int p = wvu; p < imgb / 1024; p += NWAVE) px_dma16_off(gimg, (unsigned)(p * 1024 + lane * 16), img + p * 1024);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    float cmx = 0.f;
    for (int w = lane; w < NROW; w += 64) cmx = fmaxf(cmx, w < K ? gcn[w] : 0.f);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) cmx = fmaxf(cmx, __shfl_xor(cmx, o, 64));

    // A-operand byte offsets of this lane inside a chunk: row tile wt at + 4096 wt; [pl][s] = piece, K-step
    unsigned aoff[NP][2], aoff_last[NP][2];
    {
        const int jl = min((NT - 1) * 32 + j, imgr - 1);
#pragma unroll
        for (int pl = 0; pl < NP; ++pl)
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const int u = 4 * pl + 2 * s + h;
                aoff[pl][s] = (unsigned)(j * KR_ROWB + 16 * (u ^ ((j >> 1) & 7)));
                aoff_last[pl][s] = (unsigned)(jl * KR_ROWB + 16 * (u ^ ((jl >> 1) & 7)));
            }
    }
    const long long nwt = (M + 31) / 32;                 // 32-point wave tiles
    const long long gw = (long long)blockIdx.x * NWAVE + wv, GW = (long long)gridDim.x * NWAVE;
    // lane (j, h) reads row j of the wave tile: channels 16 s + 8 h .. + 7 of a chunk = 32 bytes at 64 s + 32 h
    auto rows_of = [&](long long t, const float *&base, unsigned &vo) {
        const long long m0 = t * 32;
        const int nrows = (int)min((long long)32, M - m0);
        base = pts + (size_t)m0 * C;
        vo = (unsigned)(min(j, nrows - 1) * C + 8 * h) * 4u;
    };
    auto loadp = [&](const float *base, unsigned vo, int c, float (&f)[16]) {
        const char *sb = reinterpret_cast<const char *>(base + 32 * c) + vo;
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const f32x4 v = *reinterpret_cast<const f32x4 *>(sb + 64 * s + 16 * q);
#pragma unroll
                for (int e = 0; e < 4; ++e) f[8 * s + 4 * q + e] = v[e];
            }
    };
    uint4 bp[2][NP];
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    f32x2 fsq2 = {0.f, 0.f};                             // |p|^2 of this lane's channels, even / odd positions (v_pk_fma_f32)
    auto split_point = [&](const float (&f)[16]) {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
#pragma unroll
            for (int e = 0; e < 8; e += 2) {
                const f32x2 v = {f[8 * s + e], f[8 * s + e + 1]};
                fsq2 = __builtin_elementwise_fma(v, v, fsq2);
            }
            mm_pieces8<PREC>(f + 8 * s, bp[s]);
        }
    };
    // score of (point, centroid w) = <c_w, p> - |c_w|^2 / 2 = -d / 2: the accumulators start at -|c_w|^2 / 2 (one ds_read_b128 per
    // four rows, no vector instruction), so the epilogue compares them as they are - no zeroing pass, no fma per score
    f32x16 acc[NT];
    auto acc_init = [&]() {
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][r] = cn[t * 32 + acc_row(r, h)];
    };
    acc_init();
    double contrib = 0.0;

    auto epilogue = [&](const long long t) {
        const long long m = t * 32 + j;
        const float fsq = fsq2[0] + fsq2[1];
        const float psq = fsq + __shfl_xor(fsq, 32, 64);
        const bool live = h == 0 && m < M;
        // four instructions per score s (larger = nearer): second = med3(best, second, s) (best >= second always: the median is
        // max(second, min(best, s))), the row index as a literal select (+ 4 h at the end), best = max(best, s)
        float best = -__builtin_inff(), second = -__builtin_inff();
        int bk = 0x7ffffff0;
#pragma unroll
        for (int tt = 0; tt < NT; ++tt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float sc = acc[tt][r];
                second = __builtin_amdgcn_fmed3f(best, second, sc);
                const bool gt = sc > best;                         // (one compare, two selects: fmaxf adds a canonicalising v_max per call)
                bk = gt ? tt * 32 + acc_row(r, 0) : bk;            // increasing within a lane: '>' keeps the lowest index of a tie
                best = gt ? sc : best;
            }
        bk += 4 * h;
        {
            const float ob = __shfl_xor(best, 32, 64);
            const int ok = __shfl_xor(bk, 32, 64);
            second = fmaxf(fmaxf(second, __shfl_xor(second, 32, 64)), fminf(best, ob));
            if (ob > best || (ob == best && ok < bk)) { best = ob; bk = ok; }
        }
        best *= -2.f;                                    // d = |c|^2 - 2 <c, p>  (+ |p|^2, constant per point)
        second *= -2.f;
        const float E = fa.erel * sqrtf(psq * cmx) + fa.eabs * (psq + cmx);
        const bool decided = second - best > 2.0f * E;               // (K == 1: second = +inf)
        const bool undecided = live && !decided;
        const unsigned long long um = __ballot(undecided);
        if (um) {                                                     // one list atomic per wave (order of the list: irrelevant)
            const int first = __builtin_ctzll(um);
            int base = 0;
            if (lane == first) base = atomicAdd(fa.fcount, __builtin_popcountll(um));
            base = __shfl(base, first, 64);
            if (undecided) fa.flist[base + __builtin_popcountll(um & ((1ull << lane) - 1ull))] = (int)m;
        }
        if (live && decided) {
            label[m] = bk;
            atomicAdd(&hist[bk], 1);
            contrib += (double)fmaxf(best + psq, 0.f);
        }
        acc_init();
        fsq2 = f32x2{0.f, 0.f};
    };

    if (gw < nwt) {
        float fr[2][16];
        long long t = gw;
        const float *cb, *nb;
        unsigned vo, nvo;
        rows_of(t, cb, vo);
        loadp(cb, vo, 0, fr[0]);
        loadp(cb, vo, 1, fr[1]);
        split_point(fr[0]);
        auto step = [&](int c, auto parity) {
            constexpr int P = decltype(parity)::value;
            const bool last = P == 1 && c + 1 == nchunk;     // (an even number of chunks: a tile ends on an odd step)
            const int c2 = c + 2;
            if (c2 < nchunk) loadp(cb, vo, c2, fr[P]);      // fr[P]: split into bp during the previous step
            else             loadp(nb, nvo, c2 - nchunk, fr[P]);
            const unsigned char *pc = img + c * chunkb;
            // the A fragments of group g + 1 are requested before the MFMAs of group g are issued: a wave alone on its SIMD (the other
            // one in its epilogue) keeps the matrix pipe fed
            uint4 af[2][NP];
            auto fetch = [&](int g, uint4 (&a)[NP]) {
                const int s = g / NT, wt = g % NT;
#pragma unroll
                for (int pl = 0; pl < NP; ++pl)
                    a[pl] = *reinterpret_cast<const uint4 *>(pc + (wt == NT - 1 ? aoff_last[pl][s] : aoff[pl][s] + wt * 32 * KR_ROWB));
            };
            fetch(0, af[0]);
#pragma unroll
            for (int g = 0; g < 2 * NT; ++g) {
                if (g + 1 < 2 * NT) fetch(g + 1, af[(g + 1) & 1]);
                mm_mac<PREC>(af[g & 1], bp[g / NT], acc[g % NT]);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (last) {
                epilogue(t);
                cb = nb;
                vo = nvo;
            }
            split_point(fr[P ^ 1]);
        };
        for (; t < nwt; t += GW) {
            const long long tn = t + GW < nwt ? t + GW : nwt - 1;       // (past the end: the last tile again, never scored)
            rows_of(tn, nb, nvo);
            for (int c = 0; c < nchunk; c += 2) {
                step(c, std::integral_constant<int, 0>{});
                step(c + 1, std::integral_constant<int, 1>{});
            }
        }
    }
    __syncthreads();                                     // every wave's LDS counts
    // counts: one global float add per workgroup and non-empty centroid; inertia: one fp64 atomic per workgroup
    if (tid < K && hist[tid] != 0) atomicAdd(&counts[tid], (float)hist[tid]);
    if (inertia) {
        contrib = wave_sum(contrib);
        double *red = reinterpret_cast<double *>(cn);    // (every wave is past its last read of cn)
        if (lane == 0) red[wv] = contrib;
        __syncthreads();
        if (tid == 0) {
            double tot = 0.0;
#pragma unroll
            for (int w = 0; w < NWAVE; ++w) tot += red[w];
            atomicAdd(inertia, tot);
        }
    }
}

constexpr int KM_CHUNK = 2048;     // points per workgroup of the sums kernel
// S[k][c] = sum_m [label_m == k] p_m[c] as a one-hot GEMM on the bf16 matrix cores: the one-hot operand is exact in
// bf16, the point values go in as their three bf16 pieces (hi + mid + lo == the fp32 value), so the three MFMAs per
// row tile and 16 points add exactly the fp32 values into the fp32 accumulator - 96 matrix cycles where the fp32
// MFMA form took 512.  A workgroup owns one 32-channel column tile and KM_CHUNK points; wave w owns the centroid row
// tiles w and w + 4 and walks all the points (lane (j, h): channel j of points 8 h .. 8 h + 7 of the step).
__global__ void __launch_bounds__(256)
kmeans_sums_mfma_kernel(const float *__restrict__ pts, long long M, int C, int K, const int32_t *__restrict__ label,
                        float *__restrict__ sums)
{
    __shared__ int slab[KM_CHUNK];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int j = lane & 31, h = lane >> 5;
    const long long m0 = (long long)blockIdx.x * KM_CHUNK;
    const int c = blockIdx.y * 32 + j;                     // this lane's output channel
    const int cc = min(c, C - 1);
    const int npts = (int)(M - m0 < KM_CHUNK ? M - m0 : KM_CHUNK);
    for (int e = tid; e < KM_CHUNK; e += 256) slab[e] = (e < npts) ? label[m0 + e] : -1;
    __syncthreads();
    const int k0 = wv * 32 + j, k1 = (wv + 4) * 32 + j;    // centroid rows of this lane in the wave's two row tiles
    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
    const bool two = (wv + 4) * 32 < K;                    // second row tile in range (wave-uniform)
    const float *col = pts + (size_t)m0 * C + cc;
#pragma unroll 2
    for (int p0 = 0; p0 < npts; p0 += 16) {
        float v[8];
        int lb[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int i = p0 + 8 * h + e;
            v[e] = col[(size_t)min(i, npts - 1) * C];       // (a point past the chunk has label -1: its row of A is zero)
            lb[e] = slab[i];
        }
        unsigned ph[8], pm[8], pl[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) split3(v[e], ph[e], pm[e], pl[e]);
        const uint4 bh = make_uint4(pack_hi16(ph[0], ph[1]), pack_hi16(ph[2], ph[3]), pack_hi16(ph[4], ph[5]), pack_hi16(ph[6], ph[7]));
        const uint4 bm = make_uint4(pack_hi16(pm[0], pm[1]), pack_hi16(pm[2], pm[3]), pack_hi16(pm[4], pm[5]), pack_hi16(pm[6], pm[7]));
        const uint4 bl = make_uint4(pack_hi16(pl[0], pl[1]), pack_hi16(pl[2], pl[3]), pack_hi16(pl[4], pl[5]), pack_hi16(pl[6], pl[7]));
        auto onehot = [&](int k) {                          // bf16 1.0 = 0x3F80, two points per dword
            unsigned d[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) d[q] = (lb[2 * q] == k ? 0x3F80u : 0u) | (lb[2 * q + 1] == k ? 0x3F800000u : 0u);
            return make_uint4(d[0], d[1], d[2], d[3]);
        };
        const bf16x8 a0 = __builtin_bit_cast(bf16x8, onehot(k0));
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, __builtin_bit_cast(bf16x8, bl), acc0, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, __builtin_bit_cast(bf16x8, bm), acc0, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, __builtin_bit_cast(bf16x8, bh), acc0, 0, 0, 0);
        if (two) {
            const bf16x8 a1 = __builtin_bit_cast(bf16x8, onehot(k1));
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, __builtin_bit_cast(bf16x8, bl), acc1, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, __builtin_bit_cast(bf16x8, bm), acc1, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, __builtin_bit_cast(bf16x8, bh), acc1, 0, 0, 0);
        }
    }
    if (c < C) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int ka = wv * 32 + acc_row(r, h), kb = (wv + 4) * 32 + acc_row(r, h);
            if (ka < K && acc0[r] != 0.f) atomicAdd(&sums[(size_t)ka * C + c], acc0[r]);
            if (two && kb < K && acc1[r] != 0.f) atomicAdd(&sums[(size_t)kb * C + c], acc1[r]);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// Centroid sums as a SEGMENT SUM over the points grouped by label (workspace form of the entry point): a stable counting
// sort of the point indices by label (per-chunk histograms -> prefix over the chunks -> in-order placement), then every
// label's rows are summed in that fixed order, 256 at a time, and the partial rows are added in chunk order.  The points
// are read ONCE as whole 4 C-byte rows; no float atomics, the sums are bit-reproducible.  (The one-hot GEMM above spends
// K/32 row tiles of MFMAs on every point although it belongs to one centroid.)
// ---------------------------------------------------------------------------------------------------------
constexpr int KS_SORT_CHUNK = 1024;       // points per (one-wave) block of the histogram / placement kernels
constexpr int KS_SEG = 256;               // rows per partial sum
struct KmWs { size_t off_flist, off_fcnt, off_perm, off_bh, off_tot, off_off, off_cb, off_part, off_img, off_cn, total; int nblk, nchmax; };
static KmWs km_ws(long long M, int C, int K)
{
    KmWs w{};
    size_t o = 0;
    auto take = [&](size_t bytes) { size_t r = o; o = align_up(o + bytes, 256); return r; };
    w.nblk = (int)((M + KS_SORT_CHUNK - 1) / KS_SORT_CHUNK);
    w.nchmax = (int)((M + KS_SEG - 1) / KS_SEG) + K;
    w.off_flist = take(sizeof(int) * (size_t)M);                 // fast form: points the screen could not decide
    w.off_fcnt = take(256);
    w.off_perm = take(sizeof(int) * (size_t)M);
    w.off_bh = take(sizeof(int) * (size_t)w.nblk * K);          // per-chunk histograms, then their exclusive prefix over the chunks
    w.off_tot = take(sizeof(int) * (size_t)K);
    w.off_off = take(sizeof(int) * (size_t)(K + 1));
    w.off_cb = take(sizeof(int) * (size_t)(K + 1));
    w.off_part = take(sizeof(float) * (size_t)w.nchmax * C);
    w.off_img = take((size_t)((C + 31) / 32) * 256 * KI_ROWB);   // the screen's operand image of the centroids (K <= 256 rows per chunk)
    w.off_cn = take(sizeof(float) * 256);
    w.total = o;
    return w;
}

__global__ void __launch_bounds__(64)
km_hist_kernel(const int32_t *__restrict__ label, long long M, int K, int *__restrict__ bh)
{
    extern __shared__ int km_h[];
    for (int k = threadIdx.x; k < K; k += 64) km_h[k] = 0;
    __builtin_amdgcn_wave_barrier();
    const long long m0 = (long long)blockIdx.x * KS_SORT_CHUNK;
    for (int e = threadIdx.x; e < KS_SORT_CHUNK; e += 64)
        if (m0 + e < M) atomicAdd(&km_h[label[m0 + e]], 1);
    __builtin_amdgcn_wave_barrier();
    for (int k = threadIdx.x; k < K; k += 64) bh[(size_t)blockIdx.x * K + k] = km_h[k];
}

// block k: exclusive prefix of bh[.][k] over the chunks (in place) and the label's total
__global__ void __launch_bounds__(256)
km_scan_kernel(int *__restrict__ bh, int nblk, int K, int *__restrict__ total)
{
    __shared__ int part[256];
    const int k = blockIdx.x, t = threadIdx.x;
    const int per = (nblk + 255) / 256, b0 = t * per, b1 = min(nblk, b0 + per);
    int s = 0;
    for (int b = b0; b < b1; ++b) s += bh[(size_t)b * K + k];
    part[t] = s;
    __syncthreads();
    if (t == 0) {
        int run = 0;
        for (int q = 0; q < 256; ++q) { const int v = part[q]; part[q] = run; run += v; }
        total[k] = run;
    }
    __syncthreads();
    int run = part[t];
    for (int b = b0; b < b1; ++b) { const int v = bh[(size_t)b * K + k]; bh[(size_t)b * K + k] = run; run += v; }
}

// label offsets and the table of 256-row partial-sum chunks (one block)
__global__ void __launch_bounds__(64)
km_offsets_kernel(const int *__restrict__ total, int K, int *__restrict__ off, int *__restrict__ cb)
{
    if (threadIdx.x != 0) return;
    int o = 0, c = 0;
    for (int k = 0; k < K; ++k) {
        off[k] = o; cb[k] = c;
        o += total[k];
        c += (total[k] + KS_SEG - 1) / KS_SEG;
    }
    off[K] = o; cb[K] = c;
}

// stable placement: one wave walks its chunk 64 points at a time.  A step sorts the 64 keys (label, lane) with a bitonic network
// across the lanes; in sorted order the points of one label are consecutive and in lane order, so a point's place is its
// label's running position + its distance from the label's first lane (a prefix-max scan), and the last lane of each run
// advances the label's counter - no atomics, no per-label loop (the first form iterated over the distinct labels of a step:
// ~55 of them for 200 clusters, 122 us at 2.1 M points).
__global__ void __launch_bounds__(64)
km_place_kernel(const int32_t *__restrict__ label, long long M, int K, const int *__restrict__ bh, const int *__restrict__ off,
                int *__restrict__ perm)
{
    extern __shared__ int km_c[];                                   // running position per label
    const int lane = threadIdx.x;
    for (int k = lane; k < K; k += 64) km_c[k] = off[k] + bh[(size_t)blockIdx.x * K + k];
    __builtin_amdgcn_wave_barrier();
    const long long m0 = (long long)blockIdx.x * KS_SORT_CHUNK;
    for (int e0 = 0; e0 < KS_SORT_CHUNK; e0 += 64) {
        const long long m = m0 + e0 + lane;
        unsigned key = m < M ? ((unsigned)label[m] << 6) | (unsigned)lane : 0xffffffffu;       // invalid points sort last
#pragma unroll
        for (int kk = 2; kk <= 64; kk <<= 1)
#pragma unroll
            for (int jj = kk >> 1; jj > 0; jj >>= 1) {
                const unsigned other = (unsigned)__shfl_xor((int)key, jj, 64);
                const bool up = (lane & kk) == 0, lower = (lane & jj) == 0;
                key = (lower == up) ? min(key, other) : max(key, other);
            }
        const bool valid = key != 0xffffffffu;
        const int lb = (int)(key >> 6), src = (int)(key & 63u);
        const unsigned prev = (unsigned)__shfl_up((int)key, 1, 64);
        const bool head = lane == 0 || (prev >> 6) != (key >> 6);
        int start = head ? lane : 0;                                // first lane of this lane's run: prefix maximum of the heads
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int o = __shfl_up(start, d, 64);
            if (lane >= d) start = max(start, o);
        }
        const unsigned next = (unsigned)__shfl_down((int)key, 1, 64);
        const bool tail = lane == 63 || (next >> 6) != (key >> 6);
        const int base = valid ? km_c[lb] : 0;
        __builtin_amdgcn_wave_barrier();
        const int pos = base + (lane - start);
        if (valid) {
            perm[pos] = (int)(m0 + e0 + src);
            if (tail) km_c[lb] = pos + 1;
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// partial sums: block c of the chunk table = up to KS_SEG consecutive rows of one label, summed in order
__global__ void __launch_bounds__(256)
km_segsum_kernel(const float *__restrict__ pts, int C, int K, const int *__restrict__ perm, const int *__restrict__ off,
                 const int *__restrict__ cb, float *__restrict__ part)
{
    extern __shared__ float km_red[];                               // [4][C]
    const int c = blockIdx.x;
    if (c >= cb[K]) return;
    int lo = 0, hi = K;                                             // the label whose chunk range holds c
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (cb[mid] <= c) lo = mid; else hi = mid; }
    const int k = lo;
    const int p0 = off[k] + (c - cb[k]) * KS_SEG, p1 = min(off[k + 1], p0 + KS_SEG);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int q0 = p0 + wv * (KS_SEG / 4), q1 = min(p1, q0 + KS_SEG / 4);
    for (int ch0 = 0; ch0 < C; ch0 += 256) {                        // 4 channels per lane and pass
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
        const int c0 = ch0 + lane, c1 = c0 + 64, c2 = c0 + 128, c3 = c0 + 192;
        int q = q0;
        for (; q + 4 <= q1; q += 4) {                               // four rows in flight, added in row order
            const float *r0 = pts + (size_t)perm[q] * C, *r1 = pts + (size_t)perm[q + 1] * C;
            const float *r2 = pts + (size_t)perm[q + 2] * C, *r3 = pts + (size_t)perm[q + 3] * C;
            float v[4][4];
            const float *rr[4] = {r0, r1, r2, r3};
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                v[u][0] = c0 < C ? rr[u][c0] : 0.f; v[u][1] = c1 < C ? rr[u][c1] : 0.f;
                v[u][2] = c2 < C ? rr[u][c2] : 0.f; v[u][3] = c3 < C ? rr[u][c3] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) { a0 += v[u][0]; a1 += v[u][1]; a2 += v[u][2]; a3 += v[u][3]; }
        }
        for (; q < q1; ++q) {
            const float *r = pts + (size_t)perm[q] * C;
            if (c0 < C) a0 += r[c0];
            if (c1 < C) a1 += r[c1];
            if (c2 < C) a2 += r[c2];
            if (c3 < C) a3 += r[c3];
        }
        if (c0 < C) km_red[wv * C + c0] = a0;
        if (c1 < C) km_red[wv * C + c1] = a1;
        if (c2 < C) km_red[wv * C + c2] = a2;
        if (c3 < C) km_red[wv * C + c3] = a3;
    }
    __syncthreads();
    for (int ch = threadIdx.x; ch < C; ch += 256)
        part[(size_t)c * C + ch] = ((km_red[ch] + km_red[C + ch]) + km_red[2 * C + ch]) + km_red[3 * C + ch];
}

// sums[k] = its partial rows added in chunk order (an empty label: zeros)
__global__ void __launch_bounds__(256)
km_final_kernel(const float *__restrict__ part, int C, const int *__restrict__ cb, float *__restrict__ sums)
{
    // four row slots (waves) take the partial rows j0 + slot, + 4, ... with four loads in flight each; the slots are added in
    // slot order - a fixed order, like everything above
    extern __shared__ float km_fin[];                               // [4][C]
    const int k = blockIdx.x, j0 = cb[k], j1 = cb[k + 1];
    const int lane = threadIdx.x & 63, slot = threadIdx.x >> 6;
    for (int ch = lane; ch < C; ch += 64) {
        float s = 0.f;
        int j = j0 + slot;
        for (; j + 12 < j1; j += 16) {
            const float v0 = part[(size_t)j * C + ch], v1 = part[(size_t)(j + 4) * C + ch];
            const float v2 = part[(size_t)(j + 8) * C + ch], v3 = part[(size_t)(j + 12) * C + ch];
            s = (((s + v0) + v1) + v2) + v3;
        }
        for (; j < j1; j += 4) s += part[(size_t)j * C + ch];
        km_fin[slot * C + ch] = s;
    }
    __syncthreads();
    for (int ch = threadIdx.x; ch < C; ch += 256)
        sums[(size_t)k * C + ch] = ((km_fin[ch] + km_fin[C + ch]) + km_fin[2 * C + ch]) + km_fin[3 * C + ch];
}

static int launch_km_segsums(const float *pts, long long M, int C, int K, const int32_t *label, float *sums, char *ws, hipStream_t st)
{
    const KmWs w = km_ws(M, C, K);
    int *perm = reinterpret_cast<int *>(ws + w.off_perm), *bh = reinterpret_cast<int *>(ws + w.off_bh);
    int *tot = reinterpret_cast<int *>(ws + w.off_tot), *off = reinterpret_cast<int *>(ws + w.off_off), *cb = reinterpret_cast<int *>(ws + w.off_cb);
    float *part = reinterpret_cast<float *>(ws + w.off_part);
    hipLaunchKernelGGL(km_hist_kernel, dim3(w.nblk), dim3(64), sizeof(int) * K, st, label, M, K, bh);
    GFS3D_LAUNCH_CHECK("km_hist_kernel");
    hipLaunchKernelGGL(km_scan_kernel, dim3(K), dim3(256), 0, st, bh, w.nblk, K, tot);
    GFS3D_LAUNCH_CHECK("km_scan_kernel");
    hipLaunchKernelGGL(km_offsets_kernel, dim3(1), dim3(64), 0, st, tot, K, off, cb);
    GFS3D_LAUNCH_CHECK("km_offsets_kernel");
    hipLaunchKernelGGL(km_place_kernel, dim3(w.nblk), dim3(64), sizeof(int) * K, st, label, M, K, bh, off, perm);
    GFS3D_LAUNCH_CHECK("km_place_kernel");
    hipLaunchKernelGGL(km_segsum_kernel, dim3(w.nchmax), dim3(256), sizeof(float) * 4 * C, st, pts, C, K, perm, off, cb, part);
    GFS3D_LAUNCH_CHECK("km_segsum_kernel");
    hipLaunchKernelGGL(km_final_kernel, dim3(K), dim3(256), sizeof(float) * 4 * C, st, part, C, cb, sums);
    GFS3D_LAUNCH_CHECK("km_final_kernel");
    return GFS3D_OK;
}

// ---------------------------------------------------------------------------------------------------------
// Lloyd iterations after the first few move only a small share of the points: the centroid sums then follow the labels by a
// DELTA - every point whose label changed leaves its old centroid's sum and joins the new one's - instead of a second pass
// over all 4 C M bytes.  The running sums are 64-bit FIXED-POINT integers (value * 2^shift, the shift chosen by the caller
// from max |x| and M so that no sum can overflow): integer atomic adds are associative, so the result does not depend on the
// order in which the workgroups arrive - bit-reproducible like the sorted segment sums, at a resolution (2^-shift) far below
// fp32 rounding.  km_fix_from_f32 turns a full pass's fp32 sums into that representation, km_fix_to_f32 back.
// ---------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
km_delta_fix_kernel(const float *__restrict__ pts, long long M, int C, const int32_t *__restrict__ lab_old,
                    const int32_t *__restrict__ lab_new, double scale, long long *__restrict__ sums_fix,
                    long long *__restrict__ counts_fix, unsigned long long *__restrict__ changed)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const long long m0 = ((long long)blockIdx.x * 4 + wv) * 64;
    if (m0 >= M) return;
    const long long m = m0 + lane;
    const int lo = m < M ? lab_old[m] : 0, ln = m < M ? lab_new[m] : 0;
    unsigned long long mask = __ballot(m < M && lo != ln);
    if (lane == 0 && mask) atomicAdd(changed, (unsigned long long)__builtin_popcountll(mask));
    while (mask) {                                            // wave-uniform: one moved point per trip, lanes along its channels
        const int src = (int)__builtin_ctzll(mask);
        mask &= mask - 1;
        const int ko = __builtin_amdgcn_readlane(lo, src), kn = __builtin_amdgcn_readlane(ln, src);
        const float *row = pts + (size_t)(m0 + src) * C;
        for (int c = lane; c < C; c += 64) {
            const long long v = __double2ll_rn((double)row[c] * scale);
            atomicAdd(reinterpret_cast<unsigned long long *>(sums_fix + (size_t)kn * C + c), (unsigned long long)v);
            atomicAdd(reinterpret_cast<unsigned long long *>(sums_fix + (size_t)ko * C + c), (unsigned long long)(-v));
        }
        if (lane == 0) {
            atomicAdd(reinterpret_cast<unsigned long long *>(counts_fix + kn), 1ull);
            atomicAdd(reinterpret_cast<unsigned long long *>(counts_fix + ko), (unsigned long long)(-1ll));
        }
    }
}
__global__ void km_fix_from_f32_kernel(const float *__restrict__ sums, const float *__restrict__ counts, int K, int C, double scale,
                                       long long *__restrict__ sums_fix, long long *__restrict__ counts_fix)
{
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e < K * C) sums_fix[e] = __double2ll_rn((double)sums[e] * scale);
    if (e < K) counts_fix[e] = (long long)counts[e];
}
__global__ void km_fix_to_f32_kernel(const long long *__restrict__ sums_fix, const long long *__restrict__ counts_fix, int K, int C,
                                     double inv_scale, float *__restrict__ sums, float *__restrict__ counts)
{
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e < K * C) sums[e] = (float)((double)sums_fix[e] * inv_scale);
    if (e < K) counts[e] = (float)counts_fix[e];
}

// The tail of a Lloyd iteration on the running sums, ONE workgroup (K C <= a few 10^4 elements): the fp32 sums / counts, the new
// centres (member means; an empty cluster keeps its centre) and the iteration's three scalars - what get_basis._lloyd otherwise builds
// from ~15 element-wise launches of 5 us each: state = {sum of squared centre shifts (fp32 squares, fp64 sum in a fixed order),
// labels changed, empty clusters}; `changed` is reset to zero once read.
__global__ void __launch_bounds__(1024)
km_fix_step_kernel(const long long *__restrict__ sums_fix, const long long *__restrict__ counts_fix, int K, int C, double inv_scale,
                   const float *__restrict__ cent_old, unsigned long long *__restrict__ changed, float *__restrict__ sums,
                   float *__restrict__ counts, float *__restrict__ cent_new, float *__restrict__ state)
{
    __shared__ double red[1024];
    __shared__ int redn[1024];
    const int tid = threadIdx.x;
    double acc = 0.0;
    int nempty = 0;
    for (int e = tid; e < K * C; e += 1024) {
        const int k = e / C;
        const float s = (float)((double)sums_fix[e] * inv_scale);            // (km_fix_to_f32_kernel's conversion)
        const float cnt = (float)counts_fix[k];
        const float old = cent_old[e];
        const float nv = cnt > 0.f ? s / fmaxf(cnt, 1.f) : old;
        sums[e] = s;
        cent_new[e] = nv;
        const float d = nv - old;
        acc += (double)(d * d);
    }
    for (int k = tid; k < K; k += 1024) {
        const float cnt = (float)counts_fix[k];
        counts[k] = cnt;
        nempty += cnt == 0.f ? 1 : 0;
    }
    red[tid] = acc;
    redn[tid] = nempty;
    __syncthreads();
    for (int o = 512; o > 0; o >>= 1) {
        if (tid < o) { red[tid] += red[tid + o]; redn[tid] += redn[tid + o]; }
        __syncthreads();
    }
    if (tid == 0) {
        state[0] = (float)red[0];
        state[1] = (float)changed[0];
        state[2] = (float)redn[0];
        changed[0] = 0ull;                               // read and reset: the next gfs3d_kmeans_delta_fix counts from zero
    }
}

// relative part of the screen's error bound for C channels (knn.hip derives the terms): dropped piece products 2^-15 (1 + 2^-6) per
// dot, the screen's 3 C fp32 accumulations at 2^-23, the exact form's 6 C at 2^-24 - doubled for d = |c|^2 - 2 <c, p>
static inline float km_screen_rel(int C) { return 2.0f * (3.1e-5f + (float)C * (3.0f * 1.1920929e-7f + 6.0f * 5.9604645e-8f)) * 1.05f; }

template <int NT>
static int launch_kmeans(const float *pts, long long M, int C, const float *cent, int K, int32_t *label, float *sums,
                         float *counts, double *inertia, hipStream_t st, char *ws = nullptr, bool fast = false)
{
    ProfScope prof(5, st);
    bool x6 = false;
    if constexpr (NT >= 2) {
        if (C % 32 == 0 && ((reinterpret_cast<uintptr_t>(pts) | reinterpret_cast<uintptr_t>(cent)) & 15) == 0) {
            const size_t lds = (size_t)2 * NT * 32 * PX_ROW + sizeof(float) * (2 * NT * 32 + 8 * 32 * XS_LD);
            const dim3 grid((unsigned)((M + 255) / 256));
            KmFast fa{nullptr, nullptr, 0.f, 0.f};
            if (fast && ws) {
                const KmWs w = km_ws(M, C, K);
                fa = KmFast{reinterpret_cast<int *>(ws + w.off_flist), reinterpret_cast<int *>(ws + w.off_fcnt), km_screen_rel(C), 4e-6f};
                if (hipMemsetAsync(fa.fcount, 0, sizeof(int), st) != hipSuccess) {
                    set_error("gfs3d_kmeans_assign_fast_f32: hipMemsetAsync failed");
                    return GFS3D_ERR_HIP;
                }
                static const bool img_off = [] { const char *e = getenv("GFS3D_KM_IMG"); return e && atoi(e) == 0; }();       // A-B only
                static const int img_mode = [] { const char *e = getenv("GFS3D_KM_IMG"); return e ? atoi(e) : 2; }();            // A-B only
                unsigned char *gimg = reinterpret_cast<unsigned char *>(ws + w.off_img);
                float *gcn = reinterpret_cast<float *>(ws + w.off_cn);
                if (C % 64 == 0 && img_mode == 2 && kr_lds_bytes(NT, C, K) <= 160 * 1024 - 1024) {
                    // the whole operand image resident in LDS, independent waves (C = 192: K <= 208)
                    const int imgr = ki_rows(K);
                    hipLaunchKernelGGL(km_cent_image_swz_kernel, dim3(((C / 32) * imgr * 8 + 255) / 256 + (NT * 32 + 3) / 4), dim3(256), 0, st, cent, K, C,
                                       imgr, NT * 32, gimg, gcn);
                    GFS3D_LAUNCH_CHECK("km_cent_image_swz_kernel");
                    const size_t rlds = kr_lds_bytes(NT, C, K);
                    GFS3D_ENSURE_LDS((kmeans_screen_res_kernel<NT>), rlds, "gfs3d_kmeans_assign_fast_f32");
                    hipLaunchKernelGGL((kmeans_screen_res_kernel<NT>), dim3(min(grid.x, 256u)), dim3(PX_THREADS), rlds, st, pts, M, C, K, gimg, gcn, label,
                                       counts, inertia, fa);
                    GFS3D_LAUNCH_CHECK("kmeans_screen_res_kernel");
                } else if (C % 64 == 0 && !img_off) {     // persistent, pipelined screen on a per-call operand image (an even number of chunks)
                    hipLaunchKernelGGL(km_cent_image_kernel<PREC_X3>, dim3(((C / 32) * NT * 32 * 8 + 255) / 256 + (NT * 32 + 3) / 4), dim3(256), 0, st, cent, K, C, NT * 32, gimg, gcn);
                    GFS3D_LAUNCH_CHECK("km_cent_image_kernel");
                    const size_t ilds = (size_t)2 * NT * 32 * KI_ROWB + sizeof(float) * (2 * NT * 32 + 8 * 32 * XS_LD);
                    GFS3D_ENSURE_LDS((kmeans_screen_img_kernel<NT>), ilds, "gfs3d_kmeans_assign_fast_f32");
                    hipLaunchKernelGGL((kmeans_screen_img_kernel<NT>), dim3(min(grid.x, 256u)), dim3(PX_THREADS), ilds, st, pts, M, C, K, gimg, gcn, label,
                                       counts, inertia, fa);
                    GFS3D_LAUNCH_CHECK("kmeans_screen_img_kernel");
                } else {
                    GFS3D_ENSURE_LDS((kmeans_assign_x6_kernel<NT, PREC_X3, false>), lds, "gfs3d_kmeans_assign_fast_f32");
                    hipLaunchKernelGGL((kmeans_assign_x6_kernel<NT, PREC_X3, false>), grid, dim3(PX_THREADS), lds, st, pts, M, C, cent, K, label, counts,
                                       inertia, fa);
                    GFS3D_LAUNCH_CHECK("kmeans_assign_x6_kernel (screen)");
                }
                GFS3D_ENSURE_LDS((kmeans_assign_x6_kernel<NT, PREC_X6, true>), lds, "gfs3d_kmeans_assign_fast_f32");
                hipLaunchKernelGGL((kmeans_assign_x6_kernel<NT, PREC_X6, true>), dim3(min(grid.x, 512u)), dim3(PX_THREADS), lds, st, pts, M, C, cent, K,
                                   label, counts, inertia, fa);
                GFS3D_LAUNCH_CHECK("kmeans_assign_x6_kernel (exact pass over the list)");
            } else {
                GFS3D_ENSURE_LDS((kmeans_assign_x6_kernel<NT, PREC_X6, false>), lds, "gfs3d_kmeans_assign_f32");
                hipLaunchKernelGGL((kmeans_assign_x6_kernel<NT, PREC_X6, false>), grid, dim3(PX_THREADS), lds, st, pts, M, C, cent, K, label, counts,
                                   inertia, fa);
                GFS3D_LAUNCH_CHECK("kmeans_assign_x6_kernel");
            }
            x6 = true;
        }
    }
    if (!x6) {
        const size_t lds = sizeof(float) * (NT * 32 * PR_LD + NT * 32);
        GFS3D_ENSURE_LDS(kmeans_assign_mfma_kernel<NT>, lds, "gfs3d_kmeans_assign_f32");
        hipLaunchKernelGGL((kmeans_assign_mfma_kernel<NT>), dim3((unsigned)((M + 127) / 128)), dim3(256), lds, st,
                           pts, M, C, cent, K, label, counts, inertia);
        GFS3D_LAUNCH_CHECK("kmeans_assign_mfma_kernel");
    }
    if (!sums) return GFS3D_OK;                                                // labels / counts / inertia only (gfs3d_kmeans_delta_fix follows)
    if (ws) return launch_km_segsums(pts, M, C, K, label, sums, ws, st);      // sorted segment sums: no atomics, fixed order
    hipLaunchKernelGGL(kmeans_sums_mfma_kernel, dim3((unsigned)((M + KM_CHUNK - 1) / KM_CHUNK), (C + 31) / 32), dim3(256), 0, st,
                       pts, M, C, K, label, sums);
    GFS3D_LAUNCH_CHECK("kmeans_sums_mfma_kernel");
    return GFS3D_OK;
}

}  // namespace gfs3d

extern "C" int gfs3d_word_assign_f32(const float *feat, long long feat_bstride, int B, int C, int N,
                                     const float *words, int Kw, float scale,
                                     float *soft, int32_t *assign, float *onehot, void *stream)
{
    return gfs3d_word_assign_mixed(feat, feat_bstride, B, C, N, words, Kw, scale, soft, assign, onehot, GFS3D_COMPUTE_F32, stream);
}

extern "C" int gfs3d_word_assign_mixed(const float *feat, long long feat_bstride, int B, int C, int N,
                                       const float *words, int Kw, float scale,
                                       float *soft, int32_t *assign, float *onehot, int compute, void *stream)
{
    using namespace gfs3d;
    GFS3D_REQUIRE(compute == GFS3D_COMPUTE_F32 || compute == GFS3D_COMPUTE_BF16 || compute == GFS3D_COMPUTE_F16,
                  "gfs3d_word_assign_mixed: compute must be one of GFS3D_COMPUTE_*");
    GFS3D_REQUIRE(feat && words, "gfs3d_word_assign_f32: NULL input");
    GFS3D_REQUIRE(soft || assign || onehot, "gfs3d_word_assign_f32: no output requested");
    GFS3D_REQUIRE(B >= 1 && B <= 65535 && C >= 1 && N >= 1, "gfs3d_word_assign_f32: bad shape B=%d C=%d N=%d", B, C, N);
    GFS3D_REQUIRE((long long)N * 256 * 4 < 0x7fffffffll && (long long)N * C * 4 < 0x7fffffffll, "gfs3d_word_assign_f32: N=%d too large for 32-bit plane offsets", N);
    GFS3D_REQUIRE(Kw >= 1 && Kw <= 256, "gfs3d_word_assign_f32: need 1 <= Kw <= 256, got %d", Kw);
    GFS3D_REQUIRE(feat_bstride >= (long long)C * N, "gfs3d_word_assign_f32: bad feat_bstride");
    return dispatch_proto<MODE_SOFTMAX>(feat, feat_bstride, B, C, N, words, 1, Kw, scale, soft, assign, onehot,
                                        static_cast<hipStream_t>(stream), "gfs3d_word_assign_f32", prec_of(compute, false));
}

// ---------------------------------------------------------------------------
// Softmax backward of the word assignment in front of the fused word-gradient kernel:  d[b][w][n] = soft (dsoft - <soft, dsoft>_w).
// `dsoft` arrives as the first Kw columns of the packed fusion gradient (point-major rows of `ld` elements, fp32 or bf16: the
// adjoint of the pack of capl.py:357-359); `soft` and `d` are channel-first (B, Kw, N).  With `d` in hand the word gradient
// is a plain logits backward (proto_bwd_x6i_kernel<MODE_LOGITS>): soft and its gradient are read ONCE - the softmax form of that
// kernel read both twice (its dot pre-pass and its main loop: 391 MB per launch for 160 MB of operands, round 5) and the
// unpack kernel wrote the channel-first dsoft in between.
// One workgroup: 64 points x all words.  The 64 packed rows go through an LDS tile (coalesced along the rows); then lane =
// point, wave v owns words v, v + 4, ...: soft is read once into registers (coalesced along the points).
// ---------------------------------------------------------------------------
constexpr int WSB_MAXW = 64;             // words per wave: Kw <= 256
template <bool BF16>
__global__ void __launch_bounds__(256)
word_soft_bwd_rows_kernel(const void *__restrict__ dpacked, int ld, const float *__restrict__ soft, int Kw, int N, float *__restrict__ d)
{
    extern __shared__ float wsb_tile[];                    // [64][Kw | 1] dsoft of the tile's points, then [4][64] partial dots
    const int TLD = Kw | 1;
    float *red = wsb_tile + 64 * TLD;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int b = blockIdx.y, n0 = blockIdx.x * 64;
    const int np = min(64, N - n0);
    // the tile's rows in 16-byte pieces (8 bf16 / 4 fp32 columns; a row starts 16-byte aligned: ld % 8 == 0 is required)
    constexpr int VW = BF16 ? 8 : 4;
    const int nch = (Kw + VW - 1) / VW;
    for (int e = tid; e < 64 * nch; e += 256) {
        const int p = e / nch, c = e - p * nch;
        float v[VW];
#pragma unroll
        for (int i = 0; i < VW; ++i) v[i] = 0.f;
        if (p < np) {
            const size_t o = ((size_t)b * N + n0 + p) * ld + VW * c;
            if constexpr (BF16) {
                const uint4 r = *reinterpret_cast<const uint4 *>(static_cast<const unsigned short *>(dpacked) + o);
                const unsigned rr[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    v[2 * i] = __builtin_bit_cast(float, rr[i] << 16);
                    v[2 * i + 1] = __builtin_bit_cast(float, rr[i] & 0xffff0000u);
                }
            } else {
                const float4 r = *reinterpret_cast<const float4 *>(static_cast<const float *>(dpacked) + o);
                v[0] = r.x; v[1] = r.y; v[2] = r.z; v[3] = r.w;
            }
        }
#pragma unroll
        for (int i = 0; i < VW; ++i)
            if (VW * c + i < Kw) wsb_tile[p * TLD + VW * c + i] = v[i];
    }
    __syncthreads();
    const int n = n0 + min(lane, np - 1);
    const float *sb = soft + (size_t)b * Kw * N + n;
    float sv[WSB_MAXW];
    float dot = 0.f;
#pragma unroll
    for (int q = 0; q < WSB_MAXW; ++q) {
        const int w = wv + 4 * q;
        sv[q] = w < Kw ? sb[(size_t)w * N] : 0.f;
    }
#pragma unroll
    for (int q = 0; q < WSB_MAXW; ++q) {
        const int w = wv + 4 * q;
        if (w < Kw) dot = fmaf(sv[q], wsb_tile[lane * TLD + w], dot);
    }
    red[wv * 64 + lane] = dot;
    __syncthreads();
    dot = ((red[lane] + red[64 + lane]) + red[128 + lane]) + red[192 + lane];           // fixed order: bit-reproducible
    if (lane < np) {
        float *db = d + (size_t)b * Kw * N + n0 + lane;
#pragma unroll
        for (int q = 0; q < WSB_MAXW; ++q) {
            const int w = wv + 4 * q;
            if (w < Kw) db[(size_t)w * N] = sv[q] * (wsb_tile[lane * TLD + w] - dot);
        }
    }
}

extern "C" int gfs3d_word_soft_backward_rows(const void *dpacked, int dtype, int ld, const float *soft, int B, int Kw, int N, float *d,
                                             void *stream)
{
    using namespace gfs3d;
    const char *who = "gfs3d_word_soft_backward_rows";
    GFS3D_REQUIRE(dpacked && soft && d, "%s: NULL pointer", who);
    GFS3D_REQUIRE(dtype == 0 || dtype == GFS3D_COMPUTE_BF16, "%s: dtype=%d unsupported (0 fp32, GFS3D_COMPUTE_BF16)", who, dtype);
    GFS3D_REQUIRE(B >= 1 && B <= 65535 && N >= 1 && Kw >= 1 && Kw <= 4 * WSB_MAXW && ld >= (Kw + 7) / 8 * 8 && ld % 8 == 0,
                  "%s: bad shape B=%d Kw=%d N=%d ld=%d (ld a multiple of 8, >= Kw rounded up to 8)", who, B, Kw, N, ld);
    GFS3D_REQUIRE(((size_t)dpacked & 15) == 0, "%s: dpacked must be 16-byte aligned", who);
    const size_t lds = sizeof(float) * ((size_t)64 * (Kw | 1) + 256);
    const dim3 grid((N + 63) / 64, B);
    hipStream_t st = static_cast<hipStream_t>(stream);
    ProfScope prof(3, st);
    if (dtype) {
        GFS3D_ENSURE_LDS((word_soft_bwd_rows_kernel<true>), lds, who);
        hipLaunchKernelGGL((word_soft_bwd_rows_kernel<true>), grid, dim3(256), lds, st, dpacked, ld, soft, Kw, N, d);
    } else {
        GFS3D_ENSURE_LDS((word_soft_bwd_rows_kernel<false>), lds, who);
        hipLaunchKernelGGL((word_soft_bwd_rows_kernel<false>), grid, dim3(256), lds, st, dpacked, ld, soft, Kw, N, d);
    }
    GFS3D_LAUNCH_CHECK(who);
    return GFS3D_OK;
}

extern "C" int gfs3d_proto_cosine_f32(const float *feat, long long feat_bstride, int B, int C, int N,
                                      const float *proto, int P, int K, float scale, float *out, void *stream)
{
    using namespace gfs3d;
    GFS3D_REQUIRE(feat && proto && out, "gfs3d_proto_cosine_f32: NULL pointer");
    GFS3D_REQUIRE(B >= 1 && B <= 65535 && C >= 1 && N >= 1, "gfs3d_proto_cosine_f32: bad shape B=%d C=%d N=%d", B, C, N);
    GFS3D_REQUIRE((long long)N * 256 * 4 < 0x7fffffffll && (long long)N * C * 4 < 0x7fffffffll, "gfs3d_proto_cosine_f32: N=%d too large for 32-bit plane offsets", N);
    GFS3D_REQUIRE(K >= 1 && K <= 256, "gfs3d_proto_cosine_f32: need 1 <= K <= 256, got %d", K);
    GFS3D_REQUIRE(P == 1 || P == B, "gfs3d_proto_cosine_f32: P must be 1 or B (got %d)", P);
    GFS3D_REQUIRE(feat_bstride >= (long long)C * N, "gfs3d_proto_cosine_f32: bad feat_bstride");
    return dispatch_proto<MODE_LOGITS>(feat, feat_bstride, B, C, N, proto, P, K, scale, out, nullptr, nullptr,
                                       static_cast<hipStream_t>(stream), "gfs3d_proto_cosine_f32");
}

extern "C" size_t gfs3d_kmeans_workspace_bytes(long long M, int C, int K)
{
    if (M < 1 || M >= (1ll << 31) || C < 1 || C > 1024 || K < 1 || K > 256) return 0;
    return gfs3d::km_ws(M, C, K).total;
}

extern "C" int gfs3d_kmeans_assign_f32(const float *pts, long long M, int C, const float *cent, int K,
                                       int32_t *label, float *sums, float *counts, double *inertia, void *stream)
{
    return gfs3d_kmeans_assign_ws_f32(pts, M, C, cent, K, label, sums, counts, inertia, nullptr, 0, stream);
}

static int km_assign(const float *pts, long long M, int C, const float *cent, int K, int32_t *label, float *sums, float *counts,
                     double *inertia, void *workspace, size_t workspace_bytes, void *stream, bool fast);

extern "C" int gfs3d_kmeans_assign_ws_f32(const float *pts, long long M, int C, const float *cent, int K,
                                          int32_t *label, float *sums, float *counts, double *inertia,
                                          void *workspace, size_t workspace_bytes, void *stream)
{
    return km_assign(pts, M, C, cent, K, label, sums, counts, inertia, workspace, workspace_bytes, stream, false);
}

extern "C" int gfs3d_kmeans_assign_fast_f32(const float *pts, long long M, int C, const float *cent, int K,
                                            int32_t *label, float *sums, float *counts, double *inertia,
                                            void *workspace, size_t workspace_bytes, void *stream)
{
    GFS3D_REQUIRE(workspace != nullptr, "gfs3d_kmeans_assign_fast_f32: the screened form needs the workspace");
    return km_assign(pts, M, C, cent, K, label, sums, counts, inertia, workspace, workspace_bytes, stream, true);
}

static int km_assign(const float *pts, long long M, int C, const float *cent, int K, int32_t *label, float *sums, float *counts,
                     double *inertia, void *workspace, size_t workspace_bytes, void *stream, bool fast)
{
    using namespace gfs3d;
    GFS3D_REQUIRE(pts && cent && label && counts && (sums || (fast && workspace)), "gfs3d_kmeans_assign_f32: NULL pointer");
    GFS3D_REQUIRE(M >= 1 && C >= 1 && K >= 1 && (K <= 256 || C <= 256), "gfs3d_kmeans_assign_f32: bad shape M=%lld C=%d K=%d", M, C, K);
    if (K <= 256 && (M + 127) / 128 < (1ll << 31)) {           // matrix-pipe path
        hipStream_t st = static_cast<hipStream_t>(stream);
        const int nt = (K + 31) / 32;
        // with a workspace (and M < 2^31, C <= 1024) the centroid sums are sorted segment sums, else float-atomic tiles
        char *ws = nullptr;
        if (workspace && M < (1ll << 31) && C <= 1024) {
            if (workspace_bytes < km_ws(M, C, K).total) {
                set_error("gfs3d_kmeans_assign_ws_f32: workspace too small (%zu < %zu)", workspace_bytes, km_ws(M, C, K).total);
                return GFS3D_ERR_WORKSPACE;
            }
            ws = static_cast<char *>(workspace);
        }
        if (nt <= 1) return launch_kmeans<1>(pts, M, C, cent, K, label, sums, counts, inertia, st, ws, fast);
        if (nt <= 2) return launch_kmeans<2>(pts, M, C, cent, K, label, sums, counts, inertia, st, ws, fast);
        if (nt <= 4) return launch_kmeans<4>(pts, M, C, cent, K, label, sums, counts, inertia, st, ws, fast);
        if (nt <= 6) return launch_kmeans<6>(pts, M, C, cent, K, label, sums, counts, inertia, st, ws, fast);
        if (nt <= 7) return launch_kmeans<7>(pts, M, C, cent, K, label, sums, counts, inertia, st, ws, fast);      // 200 centroids
        return launch_kmeans<8>(pts, M, C, cent, K, label, sums, counts, inertia, st, ws, fast);
    }
    const size_t lds = sizeof(float) * (2 * 64 * (size_t)(C + 1) + 64 + 512);
    GFS3D_ENSURE_LDS(kmeans_assign_kernel, lds, "gfs3d_kmeans_assign_f32");
    const long long nblk = (M + 63) / 64;
    GFS3D_REQUIRE(nblk < (1ll << 31), "gfs3d_kmeans_assign_f32: M too large");
    ProfScope prof(5, static_cast<hipStream_t>(stream));
    hipLaunchKernelGGL(kmeans_assign_kernel, dim3((unsigned)nblk), dim3(256), lds, static_cast<hipStream_t>(stream),
                       pts, M, C, cent, K, label, sums, counts, inertia);
    GFS3D_LAUNCH_CHECK("gfs3d_kmeans_assign_f32");
    return GFS3D_OK;
}

extern "C" int gfs3d_proto_backward_feat_f32(const float *feat, long long feat_bstride, int B, int C, int N,
                                             const float *proto, int P, int K, float scale,
                                             const float *soft, const float *dup, float *dfeat, void *stream)
{
    return gfs3d_proto_backward_feat_mixed(feat, feat_bstride, B, C, N, proto, P, K, scale, soft, dup, dfeat, GFS3D_COMPUTE_F32, stream);
}

extern "C" int gfs3d_proto_backward_feat_mixed(const float *feat, long long feat_bstride, int B, int C, int N,
                                               const float *proto, int P, int K, float scale,
                                               const float *soft, const float *dup, float *dfeat, int compute, void *stream)
{
    using namespace gfs3d;
    const char *who = "gfs3d_proto_backward_feat_f32";
    GFS3D_REQUIRE(compute == GFS3D_COMPUTE_F32 || compute == GFS3D_COMPUTE_BF16 || compute == GFS3D_COMPUTE_F16,
                  "%s: compute must be one of GFS3D_COMPUTE_*", who);
    const int prec = prec_of(compute, true);
    GFS3D_REQUIRE(feat && proto && dup && dfeat, "%s: NULL pointer", who);
    GFS3D_REQUIRE(B >= 1 && B <= 65535 && C >= 1 && C <= 192 && N >= 1, "%s: bad shape B=%d C=%d N=%d (C <= 192)", who, B, C, N);
    GFS3D_REQUIRE(K >= 1 && K <= 256, "%s: need 1 <= K <= 256, got %d", who, K);
    GFS3D_REQUIRE(P == 1 || P == B, "%s: P must be 1 or B (got %d)", who, P);
    GFS3D_REQUIRE(feat_bstride >= (long long)C * N, "%s: bad feat_bstride", who);
    GFS3D_REQUIRE((long long)N * 256 * 4 < 0x7fffffffll, "%s: N=%d too large for 32-bit plane offsets", who, N);
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (soft) return dispatch_proto_bwd<MODE_SOFTMAX>(feat, feat_bstride, B, C, N, proto, P, K, scale, soft, dup, dfeat, st, who, prec);
    return dispatch_proto_bwd<MODE_LOGITS>(feat, feat_bstride, B, C, N, proto, P, K, scale, nullptr, dup, dfeat, st, who, prec);
}

extern "C" size_t gfs3d_proto_backward_feat_workspace_bytes(int C, int K)
{
    if (C < 1 || K < 1 || K > 256) return 0;
    return gfs3d::proto_bwd_image_bytes(C, K);
}

extern "C" int gfs3d_proto_backward_feat_ws(const float *feat, long long feat_bstride, int B, int C, int N,
                                            const float *proto, int P, int K, float scale,
                                            const float *soft, const float *dup, float *dfeat, int compute,
                                            void *workspace, size_t workspace_bytes, void *stream)
{
    using namespace gfs3d;
    const char *who = "gfs3d_proto_backward_feat_ws";
    static const bool image_on = [] { const char *e = getenv("GFS3D_PROTO_IMAGE"); return !(e && e[0] == '0'); }();     // A-B only
    const bool image = image_on && workspace && P == 1 && K > 32 && K <= 256 && (C == 64 || C == 128 || C == 192) && B >= 1 && B <= 65535 && N >= 1 &&
                       feat && proto && dup && dfeat && feat_bstride >= (long long)C * N && (long long)N * 256 * 4 < 0x7fffffffll &&
                       (compute == GFS3D_COMPUTE_F32 || compute == GFS3D_COMPUTE_BF16 || compute == GFS3D_COMPUTE_F16);
    if (!image)        // other shapes (get_pred-sized heads, per-sample prototypes) and argument errors: the form without a workspace
        return gfs3d_proto_backward_feat_mixed(feat, feat_bstride, B, C, N, proto, P, K, scale, soft, dup, dfeat, compute, stream);
    GFS3D_REQUIRE(workspace_bytes >= proto_bwd_image_bytes(C, K), "%s: workspace too small", who);
    const int prec = prec_of(compute, true);
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (soft) return dispatch_proto_bwd_image<MODE_SOFTMAX>(feat, feat_bstride, B, C, N, proto, K, scale, soft, dup, dfeat, workspace, st, who, prec);
    return dispatch_proto_bwd_image<MODE_LOGITS>(feat, feat_bstride, B, C, N, proto, K, scale, nullptr, dup, dfeat, workspace, st, who, prec);
}

extern "C" size_t gfs3d_proto_backward_proto_workspace_bytes(int B, int C, int N)
{
    if (B < 1 || C < 1 || N < 1) return 0;
    const int nchunk = (N + gfs3d::PB_NP * gfs3d::PB_STAGES - 1) / (gfs3d::PB_NP * gfs3d::PB_STAGES);
    return sizeof(float) * (size_t)B * nchunk * 32 * ((C + 31) / 32) * 32;
}

extern "C" int gfs3d_proto_backward_proto_f32(const float *feat, long long feat_bstride, int B, int C, int N,
                                              const float *proto, int P, int K, float scale, const float *dout, float *dproto,
                                              void *workspace, size_t workspace_bytes, void *stream)
{
    using namespace gfs3d;
    const char *who = "gfs3d_proto_backward_proto_f32";
    GFS3D_REQUIRE(feat && proto && dout && dproto && workspace, "%s: NULL pointer", who);
    GFS3D_REQUIRE(B >= 1 && B <= 65535 && C >= 1 && C <= 192 && N >= 1, "%s: bad shape B=%d C=%d N=%d (C <= 192)", who, B, C, N);
    GFS3D_REQUIRE(K >= 1 && K <= 32, "%s: need 1 <= K <= 32 prototypes, got %d", who, K);
    GFS3D_REQUIRE(P == 1 || P == B, "%s: P must be 1 or B (got %d)", who, P);
    GFS3D_REQUIRE(feat_bstride >= (long long)C * N, "%s: bad feat_bstride", who);
    if (workspace_bytes < gfs3d_proto_backward_proto_workspace_bytes(B, C, N)) {
        set_error("%s: workspace too small", who);
        return GFS3D_ERR_WORKSPACE;
    }
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int CT = (C + 31) / 32;
    const int nchunk = (N + PB_NP * PB_STAGES - 1) / (PB_NP * PB_STAGES);
    const size_t lds = sizeof(float) * ((size_t)(32 + CT * 32) * PB_LD + 64 + 256);
    GFS3D_ENSURE_LDS(proto_bwd_proto_kernel, lds, who);
    float *part = static_cast<float *>(workspace);
    ProfScope prof(3, st);
    hipLaunchKernelGGL(proto_bwd_proto_kernel, dim3(nchunk, B), dim3(256), lds, st, feat, feat_bstride, C, N, K, scale, dout, part);
    GFS3D_LAUNCH_CHECK("proto_bwd_proto_kernel");
    hipLaunchKernelGGL(proto_bwd_proto_reduce_kernel, dim3(K, P), dim3(1024), 0, st, part, nchunk, B, P, K, C, proto, dproto);
    GFS3D_LAUNCH_CHECK("proto_bwd_proto_reduce_kernel");
    return GFS3D_OK;
}

// ---- Lloyd-loop delta form of the centroid sums (fixed-point, order-independent) ----
extern "C" int gfs3d_kmeans_fix_from_f32(const float *sums, const float *counts, int K, int C, int shift, long long *sums_fix,
                                         long long *counts_fix, void *stream)
{
    using namespace gfs3d;
    GFS3D_REQUIRE(sums && counts && sums_fix && counts_fix, "gfs3d_kmeans_fix_from_f32: NULL pointer");
    GFS3D_REQUIRE(K >= 1 && C >= 1 && shift >= 0 && shift <= 60, "gfs3d_kmeans_fix_from_f32: bad argument (K=%d C=%d shift=%d)", K, C, shift);
    hipLaunchKernelGGL(km_fix_from_f32_kernel, dim3((unsigned)((K * C + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), sums, counts,
                       K, C, ldexp(1.0, shift), sums_fix, counts_fix);
    GFS3D_LAUNCH_CHECK("km_fix_from_f32_kernel");
    return GFS3D_OK;
}

extern "C" int gfs3d_kmeans_fix_to_f32(const long long *sums_fix, const long long *counts_fix, int K, int C, int shift, float *sums,
                                       float *counts, void *stream)
{
    using namespace gfs3d;
    GFS3D_REQUIRE(sums && counts && sums_fix && counts_fix, "gfs3d_kmeans_fix_to_f32: NULL pointer");
    GFS3D_REQUIRE(K >= 1 && C >= 1 && shift >= 0 && shift <= 60, "gfs3d_kmeans_fix_to_f32: bad argument (K=%d C=%d shift=%d)", K, C, shift);
    hipLaunchKernelGGL(km_fix_to_f32_kernel, dim3((unsigned)((K * C + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), sums_fix,
                       counts_fix, K, C, ldexp(1.0, -shift), sums, counts);
    GFS3D_LAUNCH_CHECK("km_fix_to_f32_kernel");
    return GFS3D_OK;
}

extern "C" int gfs3d_kmeans_fix_step(const long long *sums_fix, const long long *counts_fix, int K, int C, int shift, const float *cent_old,
                                     unsigned long long *changed, float *sums, float *counts, float *cent_new, float *state, void *stream)
{
    using namespace gfs3d;
    GFS3D_REQUIRE(sums_fix && counts_fix && cent_old && changed && sums && counts && cent_new && state, "gfs3d_kmeans_fix_step: NULL pointer");
    GFS3D_REQUIRE(K >= 1 && C >= 1 && (long long)K * C <= (1 << 22) && shift >= 0 && shift <= 60,
                  "gfs3d_kmeans_fix_step: bad argument (K=%d C=%d shift=%d)", K, C, shift);
    hipLaunchKernelGGL(km_fix_step_kernel, dim3(1), dim3(1024), 0, static_cast<hipStream_t>(stream), sums_fix, counts_fix, K, C,
                       ldexp(1.0, -shift), cent_old, changed, sums, counts, cent_new, state);
    GFS3D_LAUNCH_CHECK("km_fix_step_kernel");
    return GFS3D_OK;
}

extern "C" int gfs3d_kmeans_delta_fix(const float *pts, long long M, int C, int K, const int32_t *label_old, const int32_t *label_new,
                                      int shift, long long *sums_fix, long long *counts_fix, unsigned long long *changed, void *stream)
{
    using namespace gfs3d;
    GFS3D_REQUIRE(pts && label_old && label_new && sums_fix && counts_fix && changed, "gfs3d_kmeans_delta_fix: NULL pointer");
    GFS3D_REQUIRE(M >= 1 && M < (1ll << 38) && C >= 1 && K >= 1 && shift >= 0 && shift <= 60,
                  "gfs3d_kmeans_delta_fix: bad argument (M=%lld C=%d K=%d shift=%d)", M, C, K, shift);
    hipLaunchKernelGGL(km_delta_fix_kernel, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), pts, M, C,
                       label_old, label_new, ldexp(1.0, shift), sums_fix, counts_fix, changed);
    GFS3D_LAUNCH_CHECK("km_delta_fix_kernel");
    return GFS3D_OK;
}
