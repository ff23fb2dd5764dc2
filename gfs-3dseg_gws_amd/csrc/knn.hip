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
#include <stdlib.h>

#include "common.h"

namespace gfs3d {

constexpr int KNN_TJ = 16;       // candidates per LDS tile
constexpr int KNN_QMIN = 32;     // queue slots per lane (>= TJ + 16)
// bound pass: sample every 2nd tile when the distance itself is cheap (C <= 10: 5 MFMAs per tile); at C = 64 the
// extra 32 MFMAs per sampled tile cost what the saved inserts gain (fp32 MFMA and VALU share the SIMD's issue time)
constexpr int KNN_TWO_PASS_SMALL_C = 2, KNN_TWO_PASS_LARGE_C = 0;
constexpr int KNN_TWO_PASS_MIN_N = 512;
// (a lane's queue is merged into its list, and the threshold refreshed, once any lane holds more than QS - 16 entries;
// flushing at 8 instead of 16 was measured: the loop set-up costs what the fresher threshold saves)   // below this the bound pass costs more than the inserts it saves

template <int K>
struct TopList {
    float v[K];
    int i[K];
    __device__ __forceinline__ void init() {
#pragma unroll
        for (int t = 0; t < K; ++t) { v[t] = -__builtin_inff(); i[t] = -1; }
    }
    // stable insert: goes after every entry >= val.  Per slot: one compare, one v_med3 for the value (the list is
    // sorted, so slot t becomes clamp(val, v[t], v[t-1])) and two selects for the index.
    __device__ __forceinline__ void insert(float val, int id) {
        bool here = val > v[K - 1];
#pragma unroll
        for (int t = K - 1; t >= 1; --t) {
            const bool sh = val > v[t - 1];
            i[t] = sh ? i[t - 1] : (here ? id : i[t]);
            v[t] = __builtin_amdgcn_fmed3f(v[t - 1], val, v[t]);
            here = sh;
        }
        i[0] = here ? id : i[0];
        v[0] = fmaxf(v[0], val);
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

typedef float f32x4 __attribute__((ext_vector_type(4)));

// largest float strictly below a finite v (-inf stays -inf)
__device__ __forceinline__ float next_below(float v)
{
    if (v == -__builtin_inff()) return v;
    if (v == 0.f) return -__builtin_bit_cast(float, 1u);
    const unsigned u = __builtin_bit_cast(unsigned, v);
    return __builtin_bit_cast(float, v > 0.f ? u - 1u : u + 1u);
}

// queue slots per lane: with the bound pass (small C) few candidates queue up, and 24 slots (flush above 8) leave room
// for a third workgroup per CU
template <int CS, int K>
constexpr int knn_mfma_queue() { return (CS <= 8 && K <= 24) ? 24 : ((K > KNN_QMIN) ? K : KNN_QMIN); }

template <int CS, int K>
__global__ void __launch_bounds__(256, 2)
knn_mfma_kernel(const float *__restrict__ x, const float *__restrict__ sqg, int B, int C, int N, int k, int two_pass,
                int64_t *__restrict__ idx64, int32_t *__restrict__ idx32)
{
    constexpr int QS = knn_mfma_queue<CS, K>();
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int j = lane & 31, h = lane >> 5;
    const int nq = (N + 127) / 128;
    const int b = blockIdx.x / nq, qblk = blockIdx.x % nq;
    const int q0 = qblk * 128 + w * 32;
    if (q0 >= N) return;                               // whole wave out of range (wave-uniform)
    // candidate queue of this wave: [QS][64] entries of (distance bits, candidate id)
    uint2 *queue = reinterpret_cast<uint2 *>(smem) + (size_t)w * QS * 64;
    const float *xb = x + (size_t)b * C * N;
    const float *sqb = sqg + (size_t)b * N;
    // Buffer descriptors (wave-uniform base + per-lane 32-bit offset + scalar offset): the CS loads of a tile need
    // no per-load address arithmetic.  Lane half h reads channel 2s+h; where 2s+1 >= C (odd C, or C < 2 CS) both
    // halves read the valid row min(2s, C-1) instead and the QUERY operand of the missing channel is zero.
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(xb), 0, C * N * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_sq =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(sqb), 0, (int)min((size_t)(B - b) * N * 4 + 16, (size_t)0x7fffffff), 0x00020000);
    auto chan_off = [&](int s) { return min(2 * s, C - 1) * N * 4; };   // byte offset of channel 2s (scalar)

    float qb[CS];
    {
        const int voff_lo = min(q0 + j, N - 1) * 4, voff = voff_lo + h * N * 4;
#pragma unroll
        for (int s = 0; s < CS; ++s) {
            const float v = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_x, (2 * s + 1 < C) ? voff : voff_lo, chan_off(s), 0));
            qb[s] = (2 * s + h < C) ? v : 0.f;
        }
    }
    const float sqq = sqb[min(q0 + j, N - 1)];

    float a[CS];
    auto load_tile = [&](float (&dst)[CS], int j0) {
        const int voff_lo = min(j0 + j, N - 1) * 4, voff = voff_lo + h * N * 4;
#pragma unroll
        for (int s = 0; s < CS; ++s)
            dst[s] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_x, (2 * s + 1 < C) ? voff : voff_lo, chan_off(s), 0));
    };
    // norms of the 16 candidate rows this lane sees in a tile: rows 4h + 8g + {0..3}, g = 0..3 (one 16-byte load per g;
    // rows past N are never used)
    float sqc[16];
    auto load_sq = [&](int j0) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            // (bit-cast the whole vector: element-wise casts of the builtin's result are narrowed to one dword by hipcc 7.2)
            const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_sq, (j0 + 8 * g + 4 * h) * 4, 0, 0));
            sqc[4 * g + 0] = v.x; sqc[4 * g + 1] = v.y; sqc[4 * g + 2] = v.z; sqc[4 * g + 3] = v.w;
        }
    };

    // ---- pass 1 (optional): a lower bound of the k-th best distance from a SAMPLE of the candidate tiles.  The 16
    // candidates a lane sees in one tile are a bucket; the K largest bucket maxima are K distinct candidates, so
    // the K-th largest maximum (over both lane halves of a query) is <= the K-th best value.  One max per element
    // plus a K-long v_med3 chain per sampled tile - no indices, no divergence.  Pass 2 then queues only the few
    // candidates above the bound instead of ~K ln(N / K) per lane, and its sorted inserts (60 VALU each, executed
    // for the slowest lane of the wave) stop dominating the instruction count.
    float thr = -__builtin_inff();
    if (two_pass > 0) {
        const int step = 32 * two_pass;                  // every two_pass-th tile is sampled
        float bm[K];
#pragma unroll
        for (int t = 0; t < K; ++t) bm[t] = -__builtin_inff();
        auto bm_insert = [&](float m) {
#pragma unroll
            for (int t = K - 1; t >= 1; --t) bm[t] = __builtin_amdgcn_fmed3f(bm[t - 1], m, bm[t]);
            bm[0] = fmaxf(bm[0], m);
        };
        load_tile(a, 0);
        for (int j0 = 0; j0 < N; j0 += step) {
            load_sq(j0);
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
            for (int s = 0; s < CS; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], qb[s], acc, 0, 0, 0);
            if (j0 + step < N) load_tile(a, j0 + step);
            // max_r round(t_r - |q|^2) == round(max_r t_r - |q|^2): rounding is monotone, so one subtraction per tile
            float mt = -__builtin_inff();
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int cand = j0 + (r & 3) + 8 * (r >> 2) + 4 * h;
                const float t = __fsub_rn(2.0f * acc[r], sqc[r]);
                mt = fmaxf(mt, cand < N ? t : -__builtin_inff());
            }
            bm_insert(__fsub_rn(mt, sqq));
        }
        float other[K];
#pragma unroll
        for (int t = 0; t < K; ++t) other[t] = __shfl_xor(bm[t], 32, 64);
#pragma unroll
        for (int t = 0; t < K; ++t) bm_insert(other[t]);
        thr = next_below(bm[K - 1]);       // the strict '>' below must keep candidates equal to the bound
    }

    // ---- pass 2: exact selection among the candidates above the bound (next tile in flight during the selection;
    // a second tile of lookahead was measured: no gain, 32 more registers)
    TopListOrd<K> best;
    best.init();
    int cnt = 0;
    uint2 *qslot = queue + lane;                        // next free entry of this lane
    auto process = [&](float (&cur)[CS], int j0) {
        load_sq(j0);
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int s = 0; s < CS; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(cur[s], qb[s], acc, 0, 0, 0);
        // the MFMAs have consumed cur[]: refill it with the next tile while the selection below runs
        if (j0 + 32 < N) load_tile(cur, j0 + 32);
        __builtin_amdgcn_wave_barrier();
        const int cbase = j0 + 4 * h;
        if (j0 + 32 <= N) {                              // a ragged last tile takes the checked path
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float pd = __fsub_rn(__fsub_rn(2.0f * acc[r], sqc[r]), sqq);
                if (pd > thr) {
                    *qslot = make_uint2(__builtin_bit_cast(unsigned, pd), (unsigned)(cbase + (r & 3) + 8 * (r >> 2)));
                    qslot += 64;
                    ++cnt;
                }
            }
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int cand = cbase + (r & 3) + 8 * (r >> 2);
                const float pd = __fsub_rn(__fsub_rn(2.0f * acc[r], sqc[r]), sqq);
                if (pd > thr && cand < N) {
                    *qslot = make_uint2(__builtin_bit_cast(unsigned, pd), (unsigned)cand);
                    qslot += 64;
                    ++cnt;
                }
            }
        }
        if (__any(cnt > QS - 16)) {
            const int mx = wave_max(cnt);
            for (int s = 0; s < mx; ++s)
                if (s < cnt) {
                    const uint2 e = queue[s * 64 + lane];
                    best.insert(__builtin_bit_cast(float, e.x), (int)e.y);
                }
            cnt = 0;
            qslot = queue + lane;
            thr = fmaxf(thr, best.v[K - 1]);
            // The two lane halves of a query keep separate lists over interleaved candidates, but the final answer is the
            // top-K of their union: each half has K/2 values >= its own (K/2)-th best, so the smaller of the two (K/2)-th
            // bests is reached by K candidates of the union - a bound close to the union's K-th best, about twice as tight
            // as a half's own K-th.  (next_below: a tie with it may still win on the index across the halves.)
            const float mid = best.v[K / 2 - 1];
            thr = fmaxf(thr, next_below(fminf(mid, __shfl_xor(mid, 32, 64))));
        }
        __builtin_amdgcn_wave_barrier();
    };
    load_tile(a, 0);
    for (int j0 = 0; j0 < N; j0 += 32) process(a, j0);
    {
        const int mx = wave_max(cnt);
        for (int s = 0; s < mx; ++s)
            if (s < cnt) {
                const uint2 e = queue[s * 64 + lane];
                best.insert(__builtin_bit_cast(float, e.x), (int)e.y);
            }
    }
    // merge the h=1 half-list into h=0 (interleaved candidate subsets -> full comparator)
#pragma unroll
    for (int t = 0; t < K; ++t) {
        const float ov = __shfl(best.v[t], lane + 32, 64);
        const int oi = __shfl(best.i[t], lane + 32, 64);
        if (h == 0) best.insert_ordered(ov, oi);
    }
    __builtin_amdgcn_wave_barrier();
    int *outl = reinterpret_cast<int *>(queue);        // [K][32]
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

// ---------------------------------------------------------------------------
// Screened kNN for wide C (16 <= C <= 64, k <= 32): the exact fp32 chain is evaluated ONLY for the few candidates that can
// reach the result.  The fp32 MFMA chain of knn_mfma_kernel sits at its roof (32 MFMAs of 64 cycles per 32 x 32 tile, and
// the chip lowers its clock to ~1.4-1.6 GHz under that load); 98 % of those exact distances are far from the k-th best.
//   S  knn_screen_kernel: ALL distances once more cheaply - operands as hi + mid bf16 pairs (16 significant bits; one bf16
//      alone leaves an error of 2^-8 |x||y|, larger than the neighbour gaps of real feature maps whose norms dwarf their
//      distances), v_mfma_f32_32x32x16_bf16: 12 MFMAs of 32 cycles per tile instead of 32 of 64 - with a RIGOROUS bound E
//      of the approximation (KNN_SCREEN_*).  Sweep 0 takes, per query,
//      the K-th largest of the per-(lane half, tile) bucket maxima: K distinct candidates reach it approximately, hence
//      exactly within E, so the exact K-th best is >= tau - E and every member of the exact top-K has an approximate
//      value >= tau - 2 E.  Sweep 1 writes the ids of those candidates (about K + 5 per query) to `cand`.
//   E  knn_exact_kernel: for every query the bit-exact distances of its candidates (sequential fp32 fma chain over the
//      channels of the point-major rows, pd = fl(fl(2 dot - |xj|^2) - |xi|^2): oracle/knn_oracle.c) and a bitonic sort by
//      (distance desc, index asc) across 32 lanes; a query whose candidate list overflowed (ties, tiny N) is scanned
//      exhaustively in chunks of 32 with a running merged top-32.  Two queries per wave.
// Bit-exactness does not depend on the screen being tight - only on E being a true bound; a loose E costs candidates.
// ---------------------------------------------------------------------------
// The bound E (re-derived in round 3: the round-2 text took bf16's unit roundoff as 2^-9; it is 2^-8 - 8 significand bits,
// round to nearest).  Operands enter as hi + mid: hi = RNE_bf16(x), mid = RNE_bf16(x - hi) (x - hi is exact in fp32), so
// x = hi + mid + lo with |mid| <= 2^-8 |x| (1 + 2^-8) and |lo| <= 2^-17 |x| (a residual of exactly half an ulp is a bf16 number).
// Per channel the three products hi hi + hi mid + mid hi miss  mid_x mid_y + lo_x y + x lo_y - lo_x lo_y:
//     <= (2^-16 + 2^-17 + 2^-17)(1 + 2^-7) |x||y| < 2^-15 (1 + 2^-6) |x||y|,
// summed over the channels with Cauchy-Schwarz: <= 3.1e-5 |x_i||x_j| per dot product.  On top of that
//   * the screen's own accumulation: 3 C <= 192 piece products (each exact in fp32) added in fp32 inside the matrix core, whose
//     internal order and rounding mode are not documented - priced at one 2^-23 (twice round-to-nearest) per term:
//     192 * 2^-23 * (1 + 2^-7) |x_i||x_j| < 2.4e-5 |x_i||x_j|;
//   * the oracle's fma chain over C <= 64 channels: 64 * 2^-24 |x_i||x_j| < 0.4e-5 |x_i||x_j|.
// Per dot 5.9e-5, per distance (2 dot) 1.18e-4 |x_i||x_j| <= 1.18e-4 sqrt(|x_i|^2 max_j |x_j|^2); the norms are the oracle's own
// fp32 values (relative error 64 * 2^-24, covered by the margin to 1.25e-4).  A larger E only costs candidates: at the
// norms of BatchNorm-ed features (|x|^2 ~ 64) 2 E = 0.03 against neighbour distances of ~10, a fraction of a candidate per query.
constexpr float KNN_SCREEN_REL = 1.25e-4f;     // > 2 (2^-15 (1 + 2^-6) + 192 2^-23 + 64 2^-24) = 1.18e-4, see above
constexpr float KNN_SCREEN_ABS = 4e-6f;        // the roundings of 2 dot - |xj|^2 - |xi|^2 in both forms, relative to |xi|^2 + max |xj|^2
constexpr int KS_QS = 32;                      // candidate ids a lane (= query half) may store
constexpr int KS_SLACK_ROWS = GFS3D_KNN_IMG_SLACK_ROWS;   // zeroed image rows behind the last cloud (>= rows of a screen stage)
constexpr int KS_CAP = 64;                     // candidate ids kept per query
template <int V> struct ks_int { static constexpr int value = V; };

// two fp32 values -> their hi and mid bf16 pieces (both roundings to nearest-even), each pair packed in one dword (a low)
__device__ __forceinline__ void split2_rne(float a, float b, unsigned &hi, unsigned &mid)
{
    hi = mm_pack2<PREC_BF16>(a, b);
    const float ra = a - __builtin_bit_cast(float, hi << 16), rb = b - __builtin_bit_cast(float, hi & 0xffff0000u);   // exact
    mid = mm_pack2<PREC_BF16>(ra, rb);
}

// Everything the screened path derives from a channel-first x (B, C, N), in one pass over a 64-point tile:
//   xpm (B, N, RW) point-major rows (RW = 64 or 16), channels >= C zero          - the exact kernel's rows
//   img rows [hi RW x bf16 | mid RW x bf16] + the zeroed slack behind the last    - the screen's operands (as knn_img_kernel)
//   sq (or NULL) |x_j|^2 in the oracle's order (as knn_sqnorm_kernel)
__global__ void __launch_bounds__(256)
knn_prep_kernel(const float *__restrict__ x, int B, int C, int N, int RW, float *__restrict__ xpm, unsigned *__restrict__ img,
                float *__restrict__ sq)
{
    __shared__ float tile[64][65];
    const int tid = threadIdx.x, b = blockIdx.y, n0 = blockIdx.x * 64;
    const int p = tid & 63, wv = tid >> 6;
    for (int c = wv; c < RW; c += 4) tile[p][c] = (c < C && n0 + p < N) ? x[((size_t)b * C + c) * N + n0 + p] : 0.f;
    __syncthreads();
    if (p < RW)
        for (int q = wv; q < 64; q += 4)
            if (n0 + q < N) xpm[((size_t)b * N + n0 + q) * RW + p] = tile[q][p];
    const int half = RW >> 1;
    if (p < half)
        for (int q = wv; q < 64; q += 4)
            if (n0 + q < N) {
                unsigned hh, mm;
                split2_rne(tile[q][2 * p], tile[q][2 * p + 1], hh, mm);
                unsigned *row = img + ((size_t)b * N + n0 + q) * RW;
                row[p] = hh;
                row[half + p] = mm;
            }
    if (sq && wv == 0 && n0 + p < N) {
        float s = 0.f;
        for (int c = 0; c < C; ++c) s = __fadd_rn(s, __fmul_rn(tile[p][c], tile[p][c]));
        sq[(size_t)b * N + n0 + p] = s;
    }
    if (blockIdx.x == 0 && blockIdx.y == 0)
        for (int e = tid; e < KS_SLACK_ROWS * RW; e += 256) img[(size_t)B * N * RW + e] = 0u;
}

// ---------------------------------------------------------------------------
// knn_screen_kernel.  The candidate operands come from a PRE-SPLIT image - one row per point, [hi RW x bf16 | mid RW x bf16]
// (knn_img_kernel, once per call) - and travel global -> LDS by DMA (global_load_lds: no VGPRs, no VALU; a wave-uniform base
// + a per-lane offset computed once) through a ring of KS_RING stages, one barrier per stage.
//   * a stage holds KS_NT MFMA tiles per wave (NKB = 4: 64 rows = 16 KB; NKB = 1: 128 rows = 8 KB);
//   * the LDS image is lane-linear (DMA lands lane l at base + 16 l), rows of 64 NKB bytes without padding: the XOR swizzle
//     that makes the ds_read_b128 operand reads conflict-free is applied to the SOURCE slot of each lane and again on the read;
//   * stage norms arrive by the dword form of the same DMA;
//   * inside a wave the MFMA chains of stage s are issued together with the selection work on the scores of stage s - 1;
//   * sweep 1 writes candidate ids straight to `cand` (lane half 0 from the front of the query's KS_CAP slots, half 1 from
//     the back; ccnt = total | c0 << 16), so the only LDS is the ring: 50 KB (two 4-wave blocks per CU).
// QW = waves (32 queries each) per block: 8 when the grid then still covers the chip (halves the L2 -> LDS traffic).
// History (B = 32, N = 2048, C = 64): the round-2 form converted fp32 channel-first columns inside the loop (~40 VALU and 17
// VGPRs per tile of staging), had a barrier per 32 candidates and an LDS candidate queue of 34 KB per block: 147 us, 28 % of
// the matrix pipe at 2.4 GHz.  This form: 115 us; in-kernel stamps give 74 k cycles for sweep 0 at 1.78 GHz (66 % of the
// MFMA cycles) and 97 k for sweep 1 at 1.95 GHz (51 %) - the issue port carries ~6 VALU per MFMA on top of the MFMAs.
// ---------------------------------------------------------------------------
typedef __attribute__((address_space(3))) char ks_lds_char;
// 16 B (4 B) per lane, global -> LDS without a VGPR destination: lane l lands at lds_base + 16 l (4 l); wave-uniform base
// (SGPR pair) + per-lane 32-bit byte offset, so a piece costs no address arithmetic.  Inline asm: see lds_dma16 in edgeconv.hip
__device__ __forceinline__ void ks_dma16_off(const void *gbase_wave_uniform, unsigned voff, const void *lds_dst_wave_uniform)
{
    const unsigned base = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(ks_lds_char *)lds_dst_wave_uniform);
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(base), "v"(voff), "s"(gbase_wave_uniform) : "memory");
}
__device__ __forceinline__ void ks_dma4_off(const void *gbase_wave_uniform, unsigned voff, const void *lds_dst_wave_uniform)
{
    const unsigned base = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(ks_lds_char *)lds_dst_wave_uniform);
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2" ::"s"(base), "v"(voff), "s"(gbase_wave_uniform) : "memory");
}
template <int V> __device__ __forceinline__ void ks_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(V) : "memory"); }

constexpr int KS_RING = 3;
constexpr int ks_nt(int nkb) { return nkb == 4 ? 2 : 4; }                 // MFMA tiles per wave and stage
constexpr size_t ks_lds(int nkb) { return (size_t)KS_RING * (32 * ks_nt(nkb) * 64 * nkb + 32 * ks_nt(nkb) * 4) + 16; }

// xpm (B N, RW) fp32 point-major rows -> image rows [hi RW x bf16 | mid RW x bf16]; thread = (row, channel pair)
__global__ void __launch_bounds__(256)
knn_img_kernel(const float *__restrict__ xpm, long long rows, int RW, unsigned *__restrict__ img)
{
    const int half = RW >> 1;
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= (rows + KS_SLACK_ROWS) * half) return;
    const long long row = t / half;
    const int pi = (int)(t - row * half);
    float2 v = make_float2(0.f, 0.f);                     // the slack rows behind the last cloud (a ragged last stage reads them)
    if (row < rows) v = *reinterpret_cast<const float2 *>(xpm + row * RW + 2 * pi);
    unsigned hh, mm;
    split2_rne(v.x, v.y, hh, mm);
    img[row * RW + pi] = hh;
    img[row * RW + half + pi] = mm;
}

template <int K, int NKB, int QW>
__global__ void __launch_bounds__(64 * QW, QW == 4 ? 2 : 1)
knn_screen_kernel(const unsigned char *__restrict__ img, const float *__restrict__ sqg, int B, int N,
                   int *__restrict__ cand, int *__restrict__ ccnt)
{
    constexpr int RB = 64 * NKB, SPR = 4 * NKB, RPB = 256 / RB;          // row bytes, 16-B slots per row, rows per 256-B bank row
    constexpr int KS_NT = ks_nt(NKB), SR = 32 * KS_NT, SB = SR * RB;     // tiles / rows / bytes per stage
    constexpr int NPIECE = SB / 1024, D = NPIECE / QW, NQ = SR / QW;      // 1-KB DMA pieces per stage, per wave; norms per wave
    static_assert(NPIECE % QW == 0 && D >= 1, "stage pieces must divide over the waves");
    extern __shared__ __attribute__((aligned(16))) unsigned char ks_smem[];
    unsigned char *ring = ks_smem;                                              // [KS_RING][SR][RB] swizzled rows
    float *tsq = reinterpret_cast<float *>(ring + KS_RING * SB);               // [KS_RING][SR] their norms
    float *red = tsq + KS_RING * SR;                                           // [QW] (QW <= 8 -> 16 floats reserved below)
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int j = lane & 31, h = lane >> 5;
    // blocks b and b + 8 share an XCD (round-robin dispatch): the blocks of one point cloud stay on one XCD (speed only)
    const int nq = (N + 32 * QW - 1) / (32 * QW);
    const int slot = blockIdx.x >> 3;
    const int b = (slot / nq) * 8 + (blockIdx.x & 7), qblk = slot % nq;
    if (b >= B) return;
    const int q0 = qblk * 32 * QW + w * 32;
    const unsigned char *imgb = img + (size_t)b * N * RB;
    const float *sqb = sqg + (size_t)b * N;

    float mx = 0.f;                                        // largest norm of the cloud (the bound E needs max_j |x_j|)
    for (int n = tid; n < N; n += 64 * QW) mx = fmaxf(mx, sqb[n]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    if (lane == 0) red[w] = mx;
    // query operand: K-block kb, element e <-> channel 16 kb + 8 h + e of query q0 + j
    uint4 qph[NKB], qpm[NKB];
    {
        const unsigned char *qrow = imgb + (size_t)min(q0 + j, N - 1) * RB + 16 * h;
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) {
            qph[kb] = *reinterpret_cast<const uint4 *>(qrow + 32 * kb);
            qpm[kb] = *reinterpret_cast<const uint4 *>(qrow + 32 * NKB + 32 * kb);
        }
    }
    const float sqq = sqb[min(q0 + j, N - 1)];
    __syncthreads();
    float sqmax = red[0];
#pragma unroll
    for (int q = 1; q < QW; ++q) sqmax = fmaxf(sqmax, red[q]);
    const float E = KNN_SCREEN_REL * sqrtf(sqq * sqmax) + KNN_SCREEN_ABS * (sqq + sqmax);

    const int ns = (N + SR - 1) / SR;
    // STRIDED stages (round 5): local row r of stage st is point r ns + st - a stage's rows are spread over the whole cloud,
    // ns apart.  The threshold of sweep 0 is the K-th largest BUCKET maximum (a bucket = the 16 rows a lane sees of one tile) and
    // is tight only if the K nearest points fall into about K different buckets.  With stages of CONSECUTIVE points that held for
    // randomly ordered clouds only: in a spatially sorted cloud the neighbours share a few buckets, the candidate lists overflowed
    // and knn_exact_kernel fell back to its exhaustive scan (45 -> 775 us per launch, measured with Morton-ordered blocks).  Spread
    // rows make the bucket statistics independent of the order the points come in; for random clouds nothing changes.
    // DMA of stage st into ring slot `buf`: piece i = w D + u, lane l fills LDS slot g = 64 i + l = (row r, physical slot p),
    // which holds the logical slot p ^ f(r) of that row; rows past N read the slack / the next cloud (their norms are patched to
    // +inf): r ns + st < SR ns < N + SR, inside the zeroed stage of slack the image keeps behind the last cloud
    // ... and the local rows of a stage are ROTATED by 4 st: local row r of stage st is point ((r - 4 st) mod SR) ns + st, so
    // that consecutive points (stages st, st + 1, ...) alternate between the two lane halves of a query (rows 4 h + {0..3} + 8 g of a
    // tile).  Without the rotation the ~25 candidates of a query in a sorted cloud - a run of consecutive points - all fell to ONE
    // half and overflowed its KS_QS slots in one query of a few hundred, each of which then costs a 64-chunk exhaustive scan.
    static_assert((SR & (SR - 1)) == 0, "stage rows: a power of two");
    auto point_of = [&](int r, int st) { return ((r - 4 * st) & (SR - 1)) * ns + st; };
    int srow[D], sslot[D];
#pragma unroll
    for (int u = 0; u < D; ++u) {
        const int g = (w * D + u) * 64 + lane;
        srow[u] = g / SPR;
        sslot[u] = 16 * ((g % SPR) ^ ((srow[u] / RPB) & (SPR - 1)));
    }
    const int wu = __builtin_amdgcn_readfirstlane(w);
    auto issue = [&](int st, int buf) {
        unsigned char *dst = ring + buf * SB + wu * D * 1024;
#pragma unroll
        for (int u = 0; u < D; ++u) ks_dma16_off(imgb, (unsigned)(point_of(srow[u], st) * RB + sslot[u]), dst + u * 1024);
        if (lane < NQ) ks_dma4_off(sqb, (unsigned)(4 * min(point_of(wu * NQ + lane, st), N - 1)), tsq + buf * SR + wu * NQ);
    };
    // Software pipeline inside a wave: the MFMA chains of stage st are issued together with the selection work on the scores
    // of stage st - 1 (held in tv), so a wave fills its own MFMA gaps; tv of stage st is formed when its chains have finished.
    // stage_mfma: all KS_NT tiles of ring slot buf, acc = <x_j, x_q> for the 16 candidate rows this lane sees of each tile (C/D
    // layout); every operand read of a K-block issued first, the tiles' chains interleaved.
    const int fsw = (j / RPB) & (SPR - 1);                 // f(row) of this lane's row 32 sub + j
    auto stage_mfma = [&](int buf, f32x16 (&acc)[KS_NT]) {
#pragma unroll
        for (int sub = 0; sub < KS_NT; ++sub)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[sub][r] = 0.f;
        const unsigned char *rowb = ring + buf * SB + j * RB;
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) {                  // smallest products first
            uint4 ah[KS_NT], am[KS_NT];
#pragma unroll
            for (int sub = 0; sub < KS_NT; ++sub) {
                ah[sub] = *reinterpret_cast<const uint4 *>(rowb + 32 * sub * RB + (((2 * kb + h) ^ fsw) << 4));
                am[sub] = *reinterpret_cast<const uint4 *>(rowb + 32 * sub * RB + (((2 * NKB + 2 * kb + h) ^ fsw) << 4));
            }
#pragma unroll
            for (int sub = 0; sub < KS_NT; ++sub) acc[sub] = mm_mfma<PREC_BF16>(am[sub], qph[kb], acc[sub]);
#pragma unroll
            for (int sub = 0; sub < KS_NT; ++sub) acc[sub] = mm_mfma<PREC_BF16>(ah[sub], qpm[kb], acc[sub]);
#pragma unroll
            for (int sub = 0; sub < KS_NT; ++sub) acc[sub] = mm_mfma<PREC_BF16>(ah[sub], qph[kb], acc[sub]);
        }
    };
    // t_r = 2 <x_j, x_q> - |x_j|^2 (approximately)
    auto stage_tv = [&](int buf, const f32x16 (&acc)[KS_NT], float (&tv)[KS_NT][16]) {
#pragma unroll
        for (int sub = 0; sub < KS_NT; ++sub) {
            const float *sq = tsq + buf * SR + 32 * sub + 4 * h;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 s4 = *reinterpret_cast<const float4 *>(sq + 8 * g);
                tv[sub][4 * g + 0] = __builtin_fmaf(2.0f, acc[sub][4 * g + 0], -s4.x);
                tv[sub][4 * g + 1] = __builtin_fmaf(2.0f, acc[sub][4 * g + 1], -s4.y);
                tv[sub][4 * g + 2] = __builtin_fmaf(2.0f, acc[sub][4 * g + 2], -s4.z);
                tv[sub][4 * g + 3] = __builtin_fmaf(2.0f, acc[sub][4 * g + 3], -s4.w);
            }
        }
    };

    float bm[K];                                          // sweep 0: the K largest bucket maxima of this lane (query half)
#pragma unroll
    for (int t = 0; t < K; ++t) bm[t] = -__builtin_inff();
    auto bm_insert = [&](float m) {
#pragma unroll
        for (int t = K - 1; t >= 1; --t) bm[t] = __builtin_amdgcn_fmed3f(bm[t - 1], m, bm[t]);
        bm[0] = fmaxf(bm[0], m);
    };
    float thr = -__builtin_inff(), thr2 = -__builtin_inff();
    int cnt = 0;
    const int qme = q0 + j;
    int *myc = cand + ((size_t)b * N + min(qme, N - 1)) * KS_CAP + (h ? KS_CAP - 1 : 0);
    const int cstep = h ? -1 : 1;
    const bool qvalid = qme < N;
    // sweep 0, scores of one stage: bucket maxima (rows past N carry a norm of +inf: score -inf, no change)
    auto select0 = [&](const float (&tv)[KS_NT][16]) {
#pragma unroll
        for (int sub = 0; sub < KS_NT; ++sub) {
            float mt = fmaxf(fmaxf(tv[sub][0], tv[sub][1]), tv[sub][2]);
#pragma unroll
            for (int r = 3; r < 15; r += 2) mt = fmaxf(fmaxf(mt, tv[sub][r]), tv[sub][r + 1]);       // v_max3_f32
            mt = fmaxf(mt, tv[sub][15]);
            bm_insert(__fsub_rn(mt, sqq));
        }
    };
    // sweep 1: candidates are rare (~12 of a lane's 1024 elements): one compare + one add-with-carry per element builds a
    // 16-bit mask per tile (element r at bit 15 - r), memory is touched only for its set bits.  The compare is on tv against
    // thr2 = thr + |xq|^2 lowered by the roundings of both forms: a superset of {pd >= thr}.
    auto mask1 = [&](const float (&tv)[KS_NT][16], unsigned (&bits)[KS_NT]) {
#pragma unroll
        for (int sub = 0; sub < KS_NT; ++sub) {
            bits[sub] = 0;
#pragma unroll
            for (int r = 0; r < 16; ++r) bits[sub] = bits[sub] + bits[sub] + (tv[sub][r] >= thr2 ? 1u : 0u);
        }
    };
    // two tiles' masks in one word (tile sp at bits 16 ..): leading zeros lz = 0..15: tile sp, row lz; 16..31: tile sp + 1, row lz - 16
    auto extract1 = [&](int st, const unsigned (&bits)[KS_NT]) {
#pragma unroll
        for (int sp = 0; sp < KS_NT; sp += 2) {
            const int lbase = 32 * sp + 4 * h;                  // local row of the stage -> point_of(local row, st)
            unsigned bt = qvalid ? ((bits[sp] << 16) | bits[sp + 1]) : 0u;
            while (__any(bt != 0)) {
                if (bt != 0) {
                    const int lz = __builtin_clz(bt);
                    bt &= ~(0x80000000u >> lz);
                    const int r = lz & 15;
                    if (cnt < KS_QS) myc[cstep * cnt] = point_of(lbase + 2 * (lz & 16) + (r & 3) + 8 * (r >> 2), st);
                    ++cnt;
                }
            }
        }
    };
    auto run_sweep = [&](auto sweep_c) {
        constexpr int sweep = decltype(sweep_c)::value;
        __syncthreads();                                   // (the previous sweep's last stage has been read by every wave)
        for (int st = 0; st < KS_RING - 1 && st < ns; ++st) issue(st, st);
        int buf = 0;
        float tv[KS_NT][16];                               // scores of the previous stage (none before the first: -inf selects nothing)
#pragma unroll
        for (int sub = 0; sub < KS_NT; ++sub)
#pragma unroll
            for (int r = 0; r < 16; ++r) tv[sub][r] = -__builtin_inff();
        for (int st = 0; st < ns; ++st) {
            // stage st has landed once at most the later stages' operations of this wave are outstanding (vmcnt retires in order;
            // the candidate stores of sweep 1 only make the wait conservative)
            if (st + 1 < ns) ks_wait_vm<(KS_RING - 2) * (D + 1)>(); else ks_wait_vm<0>();
            __syncthreads();                               // ... for every wave, and ring slot (st - 1) % RING is free
            if ((SR - 1) * ns + st >= N) {                 // a stage whose last rows lie past N: they score -inf (block-uniform test)
                if (tid < SR && point_of(tid, st) >= N) tsq[buf * SR + tid] = __builtin_inff();
                __syncthreads();
            }
            if (st + KS_RING - 1 < ns) issue(st + KS_RING - 1, buf == 0 ? KS_RING - 1 : buf - 1);
            f32x16 acc[KS_NT];
            stage_mfma(buf, acc);
            if (sweep == 0) {
                select0(tv);
            } else {
                unsigned bits[KS_NT];
                mask1(tv, bits);
                extract1(st - 1, bits);
            }
            stage_tv(buf, acc, tv);
            buf = buf + 1 == KS_RING ? 0 : buf + 1;
        }
        if (sweep == 0) {
            select0(tv);
            float other[K];
#pragma unroll
            for (int t = 0; t < K; ++t) other[t] = __shfl_xor(bm[t], 32, 64);
#pragma unroll
            for (int t = 0; t < K; ++t) bm_insert(other[t]);
            // K candidates have an approximate value >= bm[K-1]; all members of the exact top-K are >= bm[K-1] - 2 E approximately
            const float lo = bm[K - 1] - 2.0f * E * 1.0001f;
            thr = (bm[K - 1] == -__builtin_inff()) ? -__builtin_inff() : next_below(next_below(lo));
            // pd = fl(tv - |xq|^2) >= thr  implies  tv >= thr + |xq|^2 - (roundings of the subtraction and of this sum)
            thr2 = (thr == -__builtin_inff()) ? thr : (thr + sqq) - 4.8e-7f * (fabsf(thr) + fabsf(sqq)) - 1e-37f;
        } else {
            unsigned bits[KS_NT];
            mask1(tv, bits);
            extract1(ns - 1, bits);
        }
    };
    run_sweep(ks_int<0>{});
    run_sweep(ks_int<1>{});
    // ---- publish the counts; a half that overflowed its KS_QS slots (ties, tiny N) -> exhaustive scan in the exact kernel
    const int c0 = __shfl(cnt, j, 64), c1 = __shfl(cnt, j + 32, 64);
    // (no bound - fewer than K buckets in a tiny cloud - also means: rows past N of a ragged tile passed the -inf threshold)
    const bool bad = c0 > KS_QS || c1 > KS_QS || thr == -__builtin_inff();
    if (qvalid && h == 0) ccnt[(size_t)b * N + qme] = bad ? -1 : ((c0 + c1) | (c0 << 16));
}

template <int K, int RW>
__global__ void __launch_bounds__(256)
knn_exact_kernel(const float *__restrict__ xpm, const float *__restrict__ sqg, const int *__restrict__ cand,
                 const int *__restrict__ ccnt, int B, int N, int k, int64_t *__restrict__ idx64, int32_t *__restrict__ idx32)
{
    typedef float f32x4_ __attribute__((ext_vector_type(4)));
    __shared__ __attribute__((aligned(16))) float qrow[4][2][RW];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int hq = lane >> 5, l = lane & 31;
    // (same XCD-aware block -> point cloud map as the screen kernel: the gathered rows of a cloud stay in one L2)
    const int bpc = (N + 7) / 8;                               // blocks per cloud: 8 queries each
    const int slot = blockIdx.x >> 3;
    const int bb = (slot / bpc) * 8 + (blockIdx.x & 7);
    const int qi_raw = (slot % bpc) * 8 + w * 2 + hq;
    const bool qok = bb < B && qi_raw < N;
    const int b = min(bb, B - 1);
    const size_t gq = (size_t)b * N + min(qi_raw, N - 1);
    const float *sqb = sqg + (size_t)b * N;
    if (l < RW / 4) *reinterpret_cast<f32x4_ *>(&qrow[w][hq][4 * l]) = *reinterpret_cast<const f32x4_ *>(xpm + gq * RW + 4 * l);
    const float sqq = sqg[gq];
    const int n = ccnt[gq];                                  // total | (ids stored from the front of the KS_CAP slots) << 16; the rest
    const int cfront = n >> 16;                              // were stored from the back.  -1: exhaustive
    const int count = n >= 0 ? (n & 0xffff) : N;
    const int nch_mine = (count + 31) / 32;
    const int nch = max(nch_mine, __shfl_xor(nch_mine, 32, 64));
    __builtin_amdgcn_wave_barrier();
    // (bitwise, not short-circuit: the lane masks are combined by scalar and / or - as `||` / `&&` every comparison of the sorting
    // network became a divergent branch, 87 of them per chunk)
    auto before = [](float a, int ai, float bb, int bi) { return (a > bb) | ((a == bb) & (ai < bi)); };
    float tv = -__builtin_inff();
    int ti = 0x7fffffff;                                      // sorted best-first across the 32 lanes of the half
    for (int c = 0; c < nch; ++c) {
        const int pos = c * 32 + l;
        const bool valid = pos < count;
        const int id = valid ? (n >= 0 ? cand[gq * KS_CAP + (pos < cfront ? pos : KS_CAP - 1 - (pos - cfront))] : pos) : 0;
        const float *row = xpm + ((size_t)b * N + id) * RW;
        f32x4_ xr[RW / 4];
#pragma unroll
        for (int u = 0; u < RW / 4; ++u) xr[u] = *reinterpret_cast<const f32x4_ *>(row + 4 * u);
        float dot = 0.f;                                     // channels past C are zero in both rows: fma(0, 0, dot) == dot
#pragma unroll
        for (int u = 0; u < RW / 4; ++u) {
            const f32x4_ qv = *reinterpret_cast<const f32x4_ *>(&qrow[w][hq][4 * u]);
            dot = __builtin_fmaf(qv[0], xr[u][0], dot);
            dot = __builtin_fmaf(qv[1], xr[u][1], dot);
            dot = __builtin_fmaf(qv[2], xr[u][2], dot);
            dot = __builtin_fmaf(qv[3], xr[u][3], dot);
        }
        float v = valid ? __fsub_rn(__fsub_rn(2.0f * dot, sqb[id]), sqq) : -__builtin_inff();
        int vi = valid ? id : 0x7fffffff;
        // bitonic sort of the chunk, best first
#pragma unroll
        for (int sz = 2; sz <= 32; sz <<= 1) {
#pragma unroll
            for (int st = sz >> 1; st > 0; st >>= 1) {
                const float ov = lane_xor_f(v, st);            // partners 1, 2, 4, 8 lanes away by DPP moves, 16 through the crossbar
                const int oi = lane_xor_i(vi, st);
                const bool want_better = (((l & st) == 0) == ((l & sz) == 0));
                const bool take = want_better ? before(ov, oi, v, vi) : before(v, vi, ov, oi);
                v = take ? ov : v;
                vi = take ? oi : vi;
            }
        }
        // merge with the running list: max(top[i], chunk[31 - i]) is bitonic and holds the best 32 of the union
        {
            const int src = (lane & 32) | (31 - l);
            const float rv = __shfl(v, src, 64);
            const int ri = __shfl(vi, src, 64);
            const bool take = before(rv, ri, tv, ti);
            tv = take ? rv : tv;
            ti = take ? ri : ti;
#pragma unroll
            for (int st = 16; st > 0; st >>= 1) {
                const float ov = lane_xor_f(tv, st);
                const int oi = lane_xor_i(ti, st);
                const bool want_better = (l & st) == 0;
                const bool tk = want_better ? before(ov, oi, tv, ti) : before(tv, ti, ov, oi);
                tv = tk ? ov : tv;
                ti = tk ? oi : ti;
            }
        }
    }
    if (qok && l < k) {
        if (idx64) idx64[gq * k + l] = (int64_t)ti;
        if (idx32) idx32[gq * k + l] = ti;
    }
}

struct KnnWs { size_t off_sq, off_pm, off_cand, off_cnt, off_img, total; };
static KnnWs knn_ws(int B, int C, int N, int k)
{
    (void)k;
    KnnWs w{};
    size_t o = 0;
    auto take = [&](size_t bytes) { size_t r = o; o = align_up(o + bytes, 256); return r; };
    w.off_sq = take(sizeof(float) * (size_t)B * N + 64);                 // norms + slack for 16-byte reads at the end
    const bool screened = C <= 64;
    w.off_pm = take(screened ? sizeof(float) * (size_t)B * N * 64 : 0);
    w.off_cand = take(screened ? sizeof(int) * (size_t)B * N * KS_CAP : 0);
    w.off_cnt = take(screened ? sizeof(int) * (size_t)B * N : 0);
    w.off_img = take(screened ? ((size_t)B * N + KS_SLACK_ROWS) * (C <= 16 ? 64 : 256) : 0);              // bf16 hi | mid rows of the screen
    w.total = o;
    return w;
}

template <int K, int NKB>
static int launch_knn_screened(const float *x, const float *xpm_in, const void *img_in, float *sq, bool have_norms, int B, int C, int N, int k,
                               int64_t *idx64, int32_t *idx32, char *ws, const KnnWs &lay, hipStream_t st)
{
    const char *who = "gfs3d_knn_topk_f32(screened)";
    ProfScope prof(0, st);
    constexpr int RW = 16 * NKB;
    int *cand = reinterpret_cast<int *>(ws + lay.off_cand), *ccnt = reinterpret_cast<int *>(ws + lay.off_cnt);
    static const int qw_env = [] { const char *e = getenv("GFS3D_KNN_SCREEN_QW"); return e ? atoi(e) : 0; }();        // A-B only
    const unsigned char *img = reinterpret_cast<unsigned char *>(ws + lay.off_img);
    const float *xpm = (xpm_in && C == 64) ? xpm_in : nullptr;      // the caller's copy is used as it is for 64 channels only
    if (!xpm) {                                             // point-major rows (zero-padded to RW channels), image and norms in one pass
        float *t = reinterpret_cast<float *>(ws + lay.off_pm);
        hipLaunchKernelGGL(knn_prep_kernel, dim3((N + 63) / 64, B), dim3(256), 0, st, x, B, C, N, RW, t,
                           reinterpret_cast<unsigned *>(ws + lay.off_img), have_norms ? nullptr : sq);
        GFS3D_LAUNCH_CHECK("knn_prep_kernel");
        xpm = t;
    } else {
        if (!have_norms) {
            hipLaunchKernelGGL(knn_sqnorm_kernel, dim3((N + 255) / 256, B), dim3(256), 0, st, x, C, N, sq);
            GFS3D_LAUNCH_CHECK("knn_sqnorm_kernel");
        }
        if (img_in) img = static_cast<const unsigned char *>(img_in);      // the producer of x_pm wrote the operand rows as well
        else {
            const long long rows = (long long)B * N;
            hipLaunchKernelGGL(knn_img_kernel, dim3((unsigned)(((rows + KS_SLACK_ROWS) * (RW / 2) + 255) / 256)), dim3(256), 0, st, xpm, rows, RW,
                               reinterpret_cast<unsigned *>(ws + lay.off_img));
            GFS3D_LAUNCH_CHECK("knn_img_kernel");
        }
    }
    const size_t lds = ks_lds(NKB);
    const int groups = (B + 7) / 8;
    const bool wide = qw_env ? qw_env == 8 : (long long)((N + 255) / 256) * groups * 8 >= 224;      // 256-query blocks still cover the chip
    if (wide) {
        GFS3D_ENSURE_LDS((knn_screen_kernel<K, NKB, 8>), lds, who);
        hipLaunchKernelGGL((knn_screen_kernel<K, NKB, 8>), dim3(((N + 255) / 256) * groups * 8), dim3(512), lds, st, img, sq, B, N, cand, ccnt);
    } else {
        GFS3D_ENSURE_LDS((knn_screen_kernel<K, NKB, 4>), lds, who);
        hipLaunchKernelGGL((knn_screen_kernel<K, NKB, 4>), dim3(((N + 127) / 128) * groups * 8), dim3(256), lds, st, img, sq, B, N, cand, ccnt);
    }
    GFS3D_LAUNCH_CHECK("knn_screen_kernel");
    hipLaunchKernelGGL((knn_exact_kernel<K, RW>), dim3((unsigned)(((N + 7) / 8) * groups * 8)), dim3(256), 0, st, xpm, sq, cand, ccnt, B, N, k, idx64, idx32);
    GFS3D_LAUNCH_CHECK("knn_exact_kernel");
    return GFS3D_OK;
}

template <int CS, int K>
static int launch_knn_mfma(const float *x, float *sq, bool have_norms, int B, int C, int N, int k, int64_t *idx64, int32_t *idx32,
                           hipStream_t st)
{
    constexpr int QS = knn_mfma_queue<CS, K>();
    const size_t lds = sizeof(float) * 4 * (size_t)(2 * QS * 64);
    dim3 grid(((N + 127) / 128) * B);
    static const int two_pass_env = [] { const char *e = getenv("GFS3D_KNN_TWO_PASS"); return e ? atoi(e) : -1; }();   // tuning only
    int two_pass = two_pass_env >= 0 ? two_pass_env : (CS <= 8 ? KNN_TWO_PASS_SMALL_C : KNN_TWO_PASS_LARGE_C);
    // the sample must hold at least K buckets (2 per sampled tile and query), else there is no bound to be had
    while (two_pass > 0 && 2 * ((N / 32 + two_pass - 1) / two_pass) < 2 * K) --two_pass;
    if (N < KNN_TWO_PASS_MIN_N) two_pass = 0;
    ProfScope prof(0, st);
    if (!have_norms) {
        hipLaunchKernelGGL(knn_sqnorm_kernel, dim3((N + 255) / 256, B), dim3(256), 0, st, x, C, N, sq);
        GFS3D_LAUNCH_CHECK("knn_sqnorm_kernel");
    }
    hipLaunchKernelGGL((knn_mfma_kernel<CS, K>), grid, dim3(256), lds, st, x, sq, B, C, N, k, two_pass, idx64, idx32);
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
    GFS3D_ENSURE_LDS(knn_topk_kernel<K>, lds, "gfs3d_knn_topk_f32");
    dim3 grid((N + 63) / 64, B);
    ProfScope prof(0, st);
    hipLaunchKernelGGL(knn_topk_kernel<K>, grid, dim3(256), lds, st, x, C, N, k, idx64, idx32);
    GFS3D_LAUNCH_CHECK("gfs3d_knn_topk_f32");
    return GFS3D_OK;
}

}  // namespace gfs3d

extern "C" size_t gfs3d_knn_workspace_bytes(int B, int C, int N, int k)
{
    return (B < 1 || N < 1 || C < 1) ? 0 : gfs3d::knn_ws(B, C, N, k).total;
}

static int knn_entry(const char *who, const float *x, const float *x_pm, const void *x_img, const float *norms, int B, int C, int N, int k,
                     int64_t *idx64, int32_t *idx32, void *workspace, size_t workspace_bytes, void *stream)
{
    using namespace gfs3d;
    GFS3D_REQUIRE(x != nullptr, "%s: x is NULL", who);
    GFS3D_REQUIRE(idx64 != nullptr || idx32 != nullptr, "%s: both outputs NULL", who);
    GFS3D_REQUIRE(B >= 0 && C >= 1 && N >= 1, "%s: bad shape B=%d C=%d N=%d", who, B, C, N);
    GFS3D_REQUIRE(k >= 1 && k <= N && k <= 64, "%s: need 1 <= k <= min(N,64), got k=%d N=%d", who, k, N);
    GFS3D_REQUIRE(B <= 65535, "%s: B=%d exceeds grid.y", who, B);
    if (B == 0) return GFS3D_OK;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (C <= 64 && k <= 32) {          // matrix paths; the point norms come from the caller or are computed into the workspace
        GFS3D_REQUIRE((size_t)C * N * 4 < 0x7fffffffull, "%s: C*N too large for 32-bit buffer offsets", who);
        const KnnWs lay = knn_ws(B, C, N, k);
        static const bool screen_on = [] { const char *e = getenv("GFS3D_KNN_SCREEN"); return !(e && e[0] == '0'); }();     // A-B only
        const bool screened = screen_on && N >= 8 && workspace != nullptr && workspace_bytes >= lay.total &&
                              (long long)B * N * KS_CAP < (1ll << 31);
        float *sq = const_cast<float *>(norms);
        if (!norms) {
            GFS3D_REQUIRE(workspace != nullptr, "%s: NULL workspace", who);
            if (workspace_bytes < lay.off_pm) {              // at least the norms
                set_error("%s: workspace %zu < required %zu", who, workspace_bytes, lay.total);
                return GFS3D_ERR_WORKSPACE;
            }
            sq = reinterpret_cast<float *>(static_cast<char *>(workspace) + lay.off_sq);
        }
        if (screened) {
            char *ws = static_cast<char *>(workspace);
            if (C <= 16) {
                if (k <= 20) return launch_knn_screened<20, 1>(x, x_pm, x_img, sq, norms != nullptr, B, C, N, k, idx64, idx32, ws, lay, st);
                return launch_knn_screened<32, 1>(x, x_pm, x_img, sq, norms != nullptr, B, C, N, k, idx64, idx32, ws, lay, st);
            }
            if (k <= 20) return launch_knn_screened<20, 4>(x, x_pm, x_img, sq, norms != nullptr, B, C, N, k, idx64, idx32, ws, lay, st);
            return launch_knn_screened<32, 4>(x, x_pm, x_img, sq, norms != nullptr, B, C, N, k, idx64, idx32, ws, lay, st);
        }
        if (k <= 20 && C <= 10) return launch_knn_mfma<5, 20>(x, sq, norms != nullptr, B, C, N, k, idx64, idx32, st);
        if (k <= 20) return launch_knn_mfma<32, 20>(x, sq, norms != nullptr, B, C, N, k, idx64, idx32, st);
        return launch_knn_mfma<32, 32>(x, sq, norms != nullptr, B, C, N, k, idx64, idx32, st);
    }
    // generic shapes: VALU kernel (any C that fits LDS, k <= 64); computes its own norms
    if (k <= 20) return launch_knn<20>(x, B, C, N, k, idx64, idx32, st);
    if (k <= 32) return launch_knn<32>(x, B, C, N, k, idx64, idx32, st);
    return launch_knn<64>(x, B, C, N, k, idx64, idx32, st);
}

extern "C" int gfs3d_knn_topk_f32(const float *x, int B, int C, int N, int k,
                                  int64_t *idx64, int32_t *idx32, void *workspace, size_t workspace_bytes, void *stream)
{
    return knn_entry("gfs3d_knn_topk_f32", x, nullptr, nullptr, nullptr, B, C, N, k, idx64, idx32, workspace, workspace_bytes, stream);
}

extern "C" int gfs3d_knn_topk_norms_f32(const float *x, const float *sqnorm, int B, int C, int N, int k,
                                        int64_t *idx64, int32_t *idx32, void *stream)
{
    using namespace gfs3d;
    GFS3D_REQUIRE(sqnorm != nullptr, "gfs3d_knn_topk_norms_f32: sqnorm is NULL");
    return knn_entry("gfs3d_knn_topk_norms_f32", x, nullptr, nullptr, sqnorm, B, C, N, k, idx64, idx32, nullptr, 0, stream);
}

extern "C" int gfs3d_knn_topk_pm_f32(const float *x, const float *x_pm, const float *sqnorm, int B, int C, int N, int k,
                                     int64_t *idx64, int32_t *idx32, void *workspace, size_t workspace_bytes, void *stream)
{
    return knn_entry("gfs3d_knn_topk_pm_f32", x, x_pm, nullptr, sqnorm, B, C, N, k, idx64, idx32, workspace, workspace_bytes, stream);
}

extern "C" int gfs3d_knn_topk_img_f32(const float *x, const float *x_pm, const void *x_img, const float *sqnorm, int B, int C, int N,
                                      int k, int64_t *idx64, int32_t *idx32, void *workspace, size_t workspace_bytes, void *stream)
{
    return knn_entry("gfs3d_knn_topk_img_f32", x, x_pm, x_img, sqnorm, B, C, N, k, idx64, idx32, workspace, workspace_bytes, stream);
}
