// K1: fused pairwise-distance + top-k (kNN graph) for gfx950.
//
// Replaces knn(x, k) of the reference (model/dgcnn.py:17-23), which materialises
// three (B,N,N) fp32 tensors.  Here nothing of size N^2 ever exists.
//
// Work decomposition (wave64):
//   workgroup = 256 threads = 4 waves, owns 64 query points of one block b;
//   lane l of every wave owns query q0+l; wave w scans candidate quarter w.
//   Candidates stream through a per-wave LDS tile [C][16] (coalesced 64 B rows
//   from the channel-first input); every lane reads the same candidate value
//   (LDS broadcast) and accumulates 16 dot products as fp32 fma chains in
//   channel order - bit-identical to oracle/knn_oracle.c.
//   Selection: a lane keeps its sorted top-K in registers; candidates that beat
//   the lane's current K-th value go to a lane-private LDS queue and are merged
//   into the register list only when some lane's queue is nearly full (the
//   per-candidate path is ~6 VALU ops; a sorted insert is ~5*K).
//   The 4 partial lists are merged by wave 0 in candidate-quarter order, which
//   keeps "equal distance -> lower index first".
#include "common.h"

namespace gfs3d {

constexpr int KNN_TJ = 16;       // candidates per LDS tile
constexpr int KNN_QMIN = 24;     // queue slots per lane (>= TJ + 8)

template <int K>
struct TopList {
    float v[K];
    int i[K];
    __device__ __forceinline__ void init() {
#pragma unroll
        for (int t = 0; t < K; ++t) { v[t] = -__builtin_inff(); i[t] = -1; }
    }
    // stable insert: goes after every entry >= val
    __device__ __forceinline__ void insert(float val, int id) {
#pragma unroll
        for (int t = K - 1; t >= 1; --t) {
            const bool sh = val > v[t - 1];
            const bool here = val > v[t];
            i[t] = sh ? i[t - 1] : (here ? id : i[t]);
            v[t] = sh ? v[t - 1] : (here ? val : v[t]);
        }
        const bool h0 = val > v[0];
        i[0] = h0 ? id : i[0];
        v[0] = h0 ? val : v[0];
    }
};

template <int K>
__global__ void __launch_bounds__(256)
knn_topk_kernel(const float *__restrict__ x, int C, int N, int k,
                int64_t *__restrict__ idx64, int32_t *__restrict__ idx32)
{
    constexpr int QS = (K > KNN_QMIN) ? K : KNN_QMIN;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = tid >> 6;
    const int b = blockIdx.y;
    const int q0 = blockIdx.x * 64;
    const int qi = min(q0 + lane, N - 1);

    // per-wave LDS carve: tile[C][16] | sq[16] | qv[QS][64] | qid[QS][64]
    const int per_wave = C * KNN_TJ + KNN_TJ + 2 * QS * 64;
    float *tile = smem + (size_t)w * per_wave;
    float *sqt = tile + C * KNN_TJ;
    float *qv = sqt + KNN_TJ;
    int *qid = reinterpret_cast<int *>(qv + QS * 64);

    const float *xb = x + (size_t)b * C * N;

    // |x_i|^2 of the lane's own query: sequential sum of rounded squares (oracle order)
    float sqi = 0.f;
    for (int c = 0; c < C; ++c) {
        const float v = xb[(size_t)c * N + qi];
        sqi = __fadd_rn(sqi, __fmul_rn(v, v));
    }

    int nq = (N + 3) / 4;
    nq = (nq + KNN_TJ - 1) / KNN_TJ * KNN_TJ;
    const int jbeg = min(w * nq, N);
    const int jend = min(jbeg + nq, N);

    TopList<K> best;
    best.init();
    float thr = -__builtin_inff();
    int cnt = 0;

    for (int j0 = jbeg; j0 < jend; j0 += KNN_TJ) {
        // stage tile: tile[c][jj] = x[b][c][j0+jj]
        for (int e = lane; e < C * KNN_TJ; e += 64) {
            const int c = e >> 4, jj = e & 15;
            const int j = j0 + jj;
            tile[e] = (j < jend) ? xb[(size_t)c * N + j] : 0.f;
        }
        __builtin_amdgcn_wave_barrier();
        {
            const int jj = lane & 15;
            float s = 0.f;
            for (int c = 0; c < C; ++c) {
                const float v = tile[c * KNN_TJ + jj];
                s = __fadd_rn(s, __fmul_rn(v, v));
            }
            if (lane < KNN_TJ) sqt[jj] = s;
        }
        __builtin_amdgcn_wave_barrier();

        float acc[KNN_TJ];
#pragma unroll
        for (int jj = 0; jj < KNN_TJ; ++jj) acc[jj] = 0.f;
        const float *xq = xb + qi;
#pragma unroll 4
        for (int c = 0; c < C; ++c) {
            const float qc = xq[(size_t)c * N];
            const float4 *t4 = reinterpret_cast<const float4 *>(tile + c * KNN_TJ);
            const float4 a0 = t4[0], a1 = t4[1], a2 = t4[2], a3 = t4[3];
            acc[0] = fmaf(qc, a0.x, acc[0]);   acc[1] = fmaf(qc, a0.y, acc[1]);
            acc[2] = fmaf(qc, a0.z, acc[2]);   acc[3] = fmaf(qc, a0.w, acc[3]);
            acc[4] = fmaf(qc, a1.x, acc[4]);   acc[5] = fmaf(qc, a1.y, acc[5]);
            acc[6] = fmaf(qc, a1.z, acc[6]);   acc[7] = fmaf(qc, a1.w, acc[7]);
            acc[8] = fmaf(qc, a2.x, acc[8]);   acc[9] = fmaf(qc, a2.y, acc[9]);
            acc[10] = fmaf(qc, a2.z, acc[10]); acc[11] = fmaf(qc, a2.w, acc[11]);
            acc[12] = fmaf(qc, a3.x, acc[12]); acc[13] = fmaf(qc, a3.y, acc[13]);
            acc[14] = fmaf(qc, a3.z, acc[14]); acc[15] = fmaf(qc, a3.w, acc[15]);
        }
#pragma unroll
        for (int jj = 0; jj < KNN_TJ; ++jj) {
            const float pd = __fsub_rn(__fsub_rn(2.0f * acc[jj], sqt[jj]), sqi);
            const int j = j0 + jj;
            if (pd > thr && j < jend) {
                qv[cnt * 64 + lane] = pd;
                qid[cnt * 64 + lane] = j;
                ++cnt;
            }
        }
        if (__any(cnt > QS - KNN_TJ)) {
            const int mx = wave_max(cnt);
            for (int s = 0; s < mx; ++s) {
                if (s < cnt) best.insert(qv[s * 64 + lane], qid[s * 64 + lane]);
            }
            cnt = 0;
            thr = best.v[K - 1];
        }
        __builtin_amdgcn_wave_barrier();
    }
    {   // drain the queue
        const int mx = wave_max(cnt);
        for (int s = 0; s < mx; ++s) {
            if (s < cnt) best.insert(qv[s * 64 + lane], qid[s * 64 + lane]);
        }
    }
    // publish partial lists (reuse the queue region; QS >= K)
    if (w > 0) {
#pragma unroll
        for (int t = 0; t < K; ++t) {
            qv[t * 64 + lane] = best.v[t];
            qid[t * 64 + lane] = best.i[t];
        }
    }
    __syncthreads();
    int *outl = reinterpret_cast<int *>(smem) + C * KNN_TJ + KNN_TJ;   // wave 0's queue region
    if (w == 0) {
        for (int ww = 1; ww < 4; ++ww) {
            const float *pv = smem + (size_t)ww * per_wave + C * KNN_TJ + KNN_TJ;
            const int *pi = reinterpret_cast<const int *>(pv + QS * 64);
            for (int t = 0; t < K; ++t) best.insert(pv[t * 64 + lane], pi[t * 64 + lane]);
        }
#pragma unroll
        for (int t = 0; t < K; ++t) outl[t * 64 + lane] = best.i[t];
    }
    __syncthreads();
    const int nvalid = min(64, N - q0);
    const size_t obase = ((size_t)b * N + q0) * k;
    for (int e = tid; e < nvalid * k; e += 256) {
        const int q = e / k, t = e - q * k;
        const int v = outl[t * 64 + q];
        if (idx64) idx64[obase + e] = (int64_t)v;
        if (idx32) idx32[obase + e] = v;
    }
}


// ---------------------------------------------------------------------------
// MFMA variant (C <= 2*CS): the dot products run on the matrix pipe.
//   v_mfma_f32_32x32x2_f32, A = 32 candidates (rows), B = 32 queries (columns), two channels
//   per step in ascending order: the accumulator is bit-for-bit the c-ordered fp32 fma chain
//   of the oracle (checked by the bit-exact tests), and the VALU is left to the selection.
// A wave owns 32 queries and scans ALL candidates (no cross-wave merge); lane (j, h) sees, for
// query j, the 16 candidate rows (r&3)+8(r>>2)+4h of every 32-candidate tile, in increasing
// index order, and keeps its own top-K; the h=1 list is merged into the h=0 list at the end
// with the full (value desc, index asc) comparator.  Operands are read straight from the
// channel-first input (one 128-B row per half-wave and channel), next tile prefetched.
// ---------------------------------------------------------------------------
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int K>
struct TopListOrd : TopList<K> {
    // insert keeping (value desc, index asc) order - needed when lists from interleaved
    // candidate subsets are merged
    __device__ __forceinline__ void insert_ordered(float val, int id) {
        auto before = [](float a, int ai, float b, int bi) { return a > b || (a == b && ai < bi); };
#pragma unroll
        for (int t = K - 1; t >= 1; --t) {
            const bool sh = before(val, id, this->v[t - 1], this->i[t - 1]);
            const bool here = before(val, id, this->v[t], this->i[t]);
            this->i[t] = sh ? this->i[t - 1] : (here ? id : this->i[t]);
            this->v[t] = sh ? this->v[t - 1] : (here ? val : this->v[t]);
        }
        const bool h0 = before(val, id, this->v[0], this->i[0]);
        this->i[0] = h0 ? id : this->i[0];
        this->v[0] = h0 ? val : this->v[0];
    }
};

// |x|^2 of the point whose even channels sit in the h=0 lane and odd channels in the h=1 lane:
// sequential c = 0,1,2,... sum of rounded squares (oracle order), identical in both lanes.
template <int CS>
__device__ __forceinline__ float sqnorm_split(const float (&v)[CS], int h)
{
    float sq = 0.f;
#pragma unroll
    for (int s = 0; s < CS; ++s) {
        const float other = __shfl_xor(v[s], 32, 64);
        const float even = h ? other : v[s];
        const float odd = h ? v[s] : other;
        sq = __fadd_rn(sq, __fmul_rn(even, even));
        sq = __fadd_rn(sq, __fmul_rn(odd, odd));      // zero-padded channel adds +0 exactly
    }
    return sq;
}

// |x_j|^2 for every point, once per call, in the oracle's order (sequential c, rounded squares)
__global__ void __launch_bounds__(256)
knn_sqnorm_kernel(const float *__restrict__ x, int C, int N, float *__restrict__ sq)
{
    const int n = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;
    if (n >= N) return;
    const float *xb = x + (size_t)b * C * N + n;
    float s = 0.f;
    for (int c = 0; c < C; ++c) {
        const float v = xb[(size_t)c * N];
        s = __fadd_rn(s, __fmul_rn(v, v));
    }
    sq[(size_t)b * N + n] = s;
}

template <int CS, int K>
__global__ void __launch_bounds__(256, 2)
knn_mfma_kernel(const float *__restrict__ x, const float *__restrict__ sqg, int C, int N, int k,
                int64_t *__restrict__ idx64, int32_t *__restrict__ idx32)
{
    constexpr int QS = (K > KNN_QMIN) ? K : KNN_QMIN;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int j = lane & 31, h = lane >> 5;
    const int b = blockIdx.y;
    const int q0 = blockIdx.x * 128 + w * 32;
    if (q0 >= N) return;                               // whole wave out of range (wave-uniform)
    const int per_wave = 2 * QS * 64;
    float *qv = smem + (size_t)w * per_wave;
    int *qid = reinterpret_cast<int *>(qv + QS * 64);
    const float *xb = x + (size_t)b * C * N;

    float qb[CS];
    {
        const int qi = min(q0 + j, N - 1);
#pragma unroll
        for (int s = 0; s < CS; ++s) {
            const int c = 2 * s + h;
            qb[s] = (c < C) ? xb[(size_t)c * N + qi] : 0.f;
        }
    }
    const float sqq = sqg[(size_t)b * N + min(q0 + j, N - 1)];
    const float *sqb = sqg + (size_t)b * N;

    TopListOrd<K> best;
    best.init();
    float thr = -__builtin_inff();
    int cnt = 0;

    float a[CS];
    auto load_tile = [&](float (&dst)[CS], int j0) {
        const int ci = min(j0 + j, N - 1);
#pragma unroll
        for (int s = 0; s < CS; ++s) {
            const int c = 2 * s + h;
            dst[s] = (c < C) ? xb[(size_t)c * N + ci] : 0.f;
        }
    };
    // norms of the 16 candidate rows this lane sees in a tile: rows 4h + 8g + {0..3}, g = 0..3
    float sqc[16];
    auto load_sq = [&](int j0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) sqc[r] = sqb[min(j0 + (r & 3) + 8 * (r >> 2) + 4 * h, N - 1)];
    };
    load_tile(a, 0);
    for (int j0 = 0; j0 < N; j0 += 32) {
        load_sq(j0);
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int s = 0; s < CS; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], qb[s], acc, 0, 0, 0);
        // the MFMAs have consumed a[]: refill it with the next tile while the selection below runs
        if (j0 + 32 < N) load_tile(a, j0 + 32);
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
            const int cand = j0 + row;
            const float pd = __fsub_rn(__fsub_rn(2.0f * acc[r], sqc[r]), sqq);
            if (pd > thr && cand < N) {
                qv[cnt * 64 + lane] = pd;
                qid[cnt * 64 + lane] = cand;
                ++cnt;
            }
        }
        if (__any(cnt > QS - 16)) {
            const int mx = wave_max(cnt);
            for (int s = 0; s < mx; ++s)
                if (s < cnt) best.insert(qv[s * 64 + lane], qid[s * 64 + lane]);
            cnt = 0;
            thr = best.v[K - 1];
        }
        __builtin_amdgcn_wave_barrier();
    }
    {
        const int mx = wave_max(cnt);
        for (int s = 0; s < mx; ++s)
            if (s < cnt) best.insert(qv[s * 64 + lane], qid[s * 64 + lane]);
    }
    // merge the h=1 half-list into h=0 (interleaved candidate subsets -> full comparator)
#pragma unroll
    for (int t = 0; t < K; ++t) {
        const float ov = __shfl(best.v[t], lane + 32, 64);
        const int oi = __shfl(best.i[t], lane + 32, 64);
        if (h == 0) best.insert_ordered(ov, oi);
    }
    __builtin_amdgcn_wave_barrier();
    int *outl = qid;                                   // [K][32]
    if (h == 0) {
#pragma unroll
        for (int t = 0; t < K; ++t) outl[t * 32 + j] = best.i[t];
    }
    __builtin_amdgcn_wave_barrier();
    const int nvalid = min(32, N - q0);
    const size_t obase = ((size_t)b * N + q0) * k;
    for (int e = lane; e < nvalid * k; e += 64) {
        const int q = e / k, t = e - q * k;
        const int v = outl[t * 32 + q];
        if (idx64) idx64[obase + e] = (int64_t)v;
        if (idx32) idx32[obase + e] = v;
    }
}

template <int CS, int K>
static int launch_knn_mfma(const float *x, float *sq, int B, int C, int N, int k, int64_t *idx64, int32_t *idx32, hipStream_t st)
{
    constexpr int QS = (K > KNN_QMIN) ? K : KNN_QMIN;
    const size_t lds = sizeof(float) * 4 * (size_t)(2 * QS * 64);
    dim3 grid((N + 127) / 128, B);
    ProfScope prof(0, st);
    hipLaunchKernelGGL(knn_sqnorm_kernel, dim3((N + 255) / 256, B), dim3(256), 0, st, x, C, N, sq);
    GFS3D_LAUNCH_CHECK("knn_sqnorm_kernel");
    hipLaunchKernelGGL((knn_mfma_kernel<CS, K>), grid, dim3(256), lds, st, x, sq, C, N, k, idx64, idx32);
    GFS3D_LAUNCH_CHECK("gfs3d_knn_topk_f32(mfma)");
    return GFS3D_OK;
}

template <int K>
static int launch_knn(const float *x, int B, int C, int N, int k, int64_t *idx64, int32_t *idx32, hipStream_t st)
{
    constexpr int QS = (K > KNN_QMIN) ? K : KNN_QMIN;
    const size_t lds = sizeof(float) * 4 * (size_t)(C * KNN_TJ + KNN_TJ + 2 * QS * 64);
    if (lds > 160 * 1024) {
        set_error("gfs3d_knn_topk_f32: C=%d needs %zu B of LDS (> 160 KiB)", C, lds);
        return GFS3D_ERR_INVALID;
    }
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(knn_topk_kernel<K>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) {
            set_error("gfs3d_knn_topk_f32: hipFuncSetAttribute: %s", hipGetErrorString(e));
            return GFS3D_ERR_HIP;
        }
    }
    dim3 grid((N + 63) / 64, B);
    ProfScope prof(0, st);
    hipLaunchKernelGGL(knn_topk_kernel<K>, grid, dim3(256), lds, st, x, C, N, k, idx64, idx32);
    GFS3D_LAUNCH_CHECK("gfs3d_knn_topk_f32");
    return GFS3D_OK;
}

}  // namespace gfs3d

extern "C" size_t gfs3d_knn_workspace_bytes(int B, int C, int N, int k)
{
    (void)C; (void)k;
    return (B < 1 || N < 1) ? 0 : sizeof(float) * (size_t)B * N;
}

extern "C" int gfs3d_knn_topk_f32(const float *x, int B, int C, int N, int k,
                                  int64_t *idx64, int32_t *idx32, void *workspace, size_t workspace_bytes, void *stream)
{
    using namespace gfs3d;
    GFS3D_REQUIRE(x != nullptr, "gfs3d_knn_topk_f32: x is NULL");
    GFS3D_REQUIRE(idx64 != nullptr || idx32 != nullptr, "gfs3d_knn_topk_f32: both outputs NULL");
    GFS3D_REQUIRE(B >= 0 && C >= 1 && N >= 1, "gfs3d_knn_topk_f32: bad shape B=%d C=%d N=%d", B, C, N);
    GFS3D_REQUIRE(k >= 1 && k <= N && k <= 64, "gfs3d_knn_topk_f32: need 1 <= k <= min(N,64), got k=%d N=%d", k, N);
    GFS3D_REQUIRE(B <= 65535, "gfs3d_knn_topk_f32: B=%d exceeds grid.y", B);
    if (B == 0) return GFS3D_OK;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (C <= 64 && k <= 32) {          // MFMA path; needs B*N floats of workspace for the point norms
        GFS3D_REQUIRE(workspace != nullptr, "gfs3d_knn_topk_f32: NULL workspace");
        if (workspace_bytes < sizeof(float) * (size_t)B * N) {
            set_error("gfs3d_knn_topk_f32: workspace %zu < required %zu", workspace_bytes, sizeof(float) * (size_t)B * N);
            return GFS3D_ERR_WORKSPACE;
        }
        float *sq = static_cast<float *>(workspace);
        if (k <= 20 && C <= 10) return launch_knn_mfma<5, 20>(x, sq, B, C, N, k, idx64, idx32, st);
        if (k <= 20) return launch_knn_mfma<32, 20>(x, sq, B, C, N, k, idx64, idx32, st);
        return launch_knn_mfma<32, 32>(x, sq, B, C, N, k, idx64, idx32, st);
    }
    // generic shapes: VALU kernel (any C that fits LDS, k <= 64)
    if (k <= 20) return launch_knn<20>(x, B, C, N, k, idx64, idx32, st);
    if (k <= 32) return launch_knn<32>(x, B, C, N, k, idx64, idx32, st);
    return launch_knn<64>(x, B, C, N, k, idx64, idx32, st);
}
