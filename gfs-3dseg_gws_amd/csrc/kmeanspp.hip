// k-means++ seeding on the device (configs[2]: 200 centroids over up to 2.1 M x 192 fp32 points).
// Replaces the seeding half of sklearn's KMeans(init='k-means++') in get_basis.py:210 of the reference (a third-party
// dependency, not vendored; the D^2 sampling of Arthur & Vassilvitskii is restated here: the next centre is point i with
// probability d2[i] / sum d2, d2[i] = squared distance to the nearest centre so far).
//
// The K - 1 picks are sequential by nature; each costs one pass over the points.  Everything stays on the device and nothing
// waits for the host: the uniform variates are drawn on the host up front (numpy RandomState - the same stream the host form
// consumed one at a time) and uploaded once.
//   kpp_update_kernel  d2[i] = min(d2[i], |p_i - c_t|^2), nearest[i], and the fp64 sum of d2 per block of 1024 points.  A point
//                      whose nearest centre a satisfies |c_t - c_a|^2 >= 4 d2[i] cannot move (triangle inequality): its 768-byte
//                      row is not read at all - late picks read a small part of the array.
//   kpp_pick_kernel    one workgroup: cdf over the block sums -> block; cdf over that block's d2 -> the first point whose
//                      cumulative mass exceeds u * total (numpy's searchsorted(cdf, u, side='right')); copies its row to
//                      centers[t + 1] and leaves |c_new - c_a|^2 for the next update's pruning test.
// Fixed summation orders everywhere: the picks are a pure function of (points, first, u).
#include "common.h"

namespace gfs3d {
namespace {

constexpr int KPP_BLOCK = 1024;          // points per workgroup of the update kernel = granularity of the two-level cdf
constexpr int KPP_PICK_THREADS = 1024;
constexpr float KPP_PRUNE_MARGIN = 1.0f + 1e-3f;       // skip only when the bound holds with room for the roundings of both sides

typedef float kf4 __attribute__((ext_vector_type(4)));

// 16 lanes per row (float4 each per piece), 4 rows per wave step; the 64 rows of a wave chunk are filtered first and only the
// rows that can move are visited.
template <int C4MAX>       // float4 pieces of a row handled per lane: ceil(C / 64)
__global__ void __launch_bounds__(256)
kpp_update_kernel(const float *__restrict__ pts, long long M, int C, const float *__restrict__ centers, int t,
                  const float *__restrict__ cc, int prune, float *__restrict__ d2, int *__restrict__ nearest, double *__restrict__ bsum,
                  const unsigned char *__restrict__ cmask, const int *__restrict__ cbest)
{
    __shared__ double wsum[4];
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int q = lane & 15, grp = lane >> 4;
    const int c4n = C >> 2;
    kf4 cen[C4MAX];
#pragma unroll
    for (int s = 0; s < C4MAX; ++s) {
        const int pc = q + 16 * s;
        cen[s] = pc < c4n ? *reinterpret_cast<const kf4 *>(centers + (size_t)t * C + 4 * pc) : kf4{0.f, 0.f, 0.f, 0.f};
    }
    const long long base = (long long)blockIdx.x * KPP_BLOCK + wv * 256;
    double acc = 0.0;
    for (int ch = 0; ch < 4; ++ch) {
        const long long row0 = base + 64 * ch;
        if (row0 >= M) break;
        const long long mine = row0 + lane;
        const bool have = mine < M;
        float myd = 0.f;
        int mya = 0;
        bool need = have;
        if (t > 0 && have) {
            myd = d2[mine];
            mya = nearest[mine];
            // |p - c_t| >= |c_t - c_a| - |p - c_a| >= 2 r - r = r when |c_t - c_a|^2 >= 4 r^2
            if (cmask) need = (cmask[mine] >> *cbest) & 1;         // greedy form: the gain pass already knows which rows move
            else if (prune && cc[mya] >= 4.f * myd * KPP_PRUNE_MARGIN) need = false;
        }
        unsigned long long mask = __ballot(need);
        float newd = 0.f;
        while (mask) {                                             // wave-uniform: up to 4 rows per trip
            int b[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                b[g] = mask ? (int)__builtin_ctzll(mask) : -1;
                mask &= mask - 1;
            }
            const int r = grp == 0 ? b[0] : grp == 1 ? b[1] : grp == 2 ? b[2] : b[3];
            float s_ = 0.f;
            if (r >= 0) {
                const float *row = pts + (size_t)(row0 + r) * C;
                kf4 v[C4MAX];
#pragma unroll
                for (int s = 0; s < C4MAX; ++s) {
                    const int pc = q + 16 * s;
                    v[s] = pc < c4n ? *reinterpret_cast<const kf4 *>(row + 4 * pc) : cen[s];
                }
#pragma unroll
                for (int s = 0; s < C4MAX; ++s) {
                    const kf4 d = v[s] - cen[s];
                    s_ = fmaf(d.x, d.x, s_);
                    s_ = fmaf(d.y, d.y, s_);
                    s_ = fmaf(d.z, d.z, s_);
                    s_ = fmaf(d.w, d.w, s_);
                }
            }
            s_ = row16_sum(s_);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float dg = __shfl(s_, 16 * g, 64);
                if (lane == b[g]) newd = dg;
            }
        }
        if (have) {
            if (t == 0) { myd = newd; mya = 0; d2[mine] = myd; nearest[mine] = 0; }
            else if (need && newd < myd) { myd = newd; mya = t; d2[mine] = myd; nearest[mine] = t; }
            acc += (double)myd;
        }
    }
    acc = wave_sum(acc);
    if (lane == 0) wsum[wv] = acc;
    __syncthreads();
    if (threadIdx.x == 0) bsum[blockIdx.x] = ((wsum[0] + wsum[1]) + wsum[2]) + wsum[3];
}

// inclusive scan of one double per thread over the workgroup: shuffle scan inside each wave, the 16 wave totals through LDS (a
// fixed tree: two barriers).  buf[KPP_PICK_THREADS - 1 .. ] is not used: the total is returned through `total`.
__device__ __forceinline__ double block_scan_incl(double v, double *buf, int tid, int n, double &total)
{
    (void)n;
    const int lane = tid & 63, wv = tid >> 6;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const double up = __shfl_up(v, o, 64);
        if (lane >= o) v += up;
    }
    __syncthreads();                                    // (buf may still be read from a previous scan)
    if (lane == 63) buf[wv] = v;
    __syncthreads();
    double before = 0.0, tot = 0.0;
#pragma unroll
    for (int w = 0; w < KPP_PICK_THREADS / 64; ++w) {
        const double x = buf[w];
        if (w < wv) before += x;
        tot += x;
    }
    total = tot;
    return v + before;
}

__global__ void __launch_bounds__(KPP_PICK_THREADS)
kpp_pick_kernel(const float *__restrict__ pts, long long M, int C, int t_new, long long first, const double *__restrict__ u,
                const float *__restrict__ d2, const double *__restrict__ bsum, int nblk, float *__restrict__ centers,
                long long *__restrict__ picks, float *__restrict__ cc)
{
    __shared__ double buf[KPP_PICK_THREADS];
    __shared__ long long s_pick;
    __shared__ int s_blk, s_win, s_last;
    __shared__ double s_before, s_target;
    const int tid = threadIdx.x;
    if (t_new == 0) {
        if (tid == 0) s_pick = first;
        __syncthreads();
    } else {
        // level 1: thread j owns `per` consecutive blocks
        const int per = (nblk + KPP_PICK_THREADS - 1) / KPP_PICK_THREADS;
        double mine = 0.0;
        for (int e = 0; e < per; ++e) {
            const int b = tid * per + e;
            if (b < nblk) mine += bsum[b];
        }
        double total;
        const double incl = block_scan_incl(mine, buf, tid, KPP_PICK_THREADS, total);
        const double target = u[t_new - 1] * total;
        if (tid == 0) { s_blk = -1; s_target = target; s_win = KPP_PICK_THREADS; }
        __syncthreads();
        const double before = incl - mine;
        if (total > 0.0 && incl > target && before <= target) atomicMin(&s_win, tid);   // (rounding of incl - mine: the lowest match wins)
        __syncthreads();
        if (tid == s_win) {
            double run = before;
            int pickb = -1;
            for (int e = 0; e < per; ++e) {
                const int b = tid * per + e;
                if (b >= nblk) break;
                const double nxt = run + bsum[b];
                if (nxt > target) { pickb = b; break; }
                run = nxt;
            }
            if (pickb < 0) {                                        // rounding at the range's end: its last non-empty block
                run = before;
                for (int e = 0; e < per; ++e) {
                    const int b = tid * per + e;
                    if (b >= nblk) break;
                    if (bsum[b] > 0.0) { pickb = b; s_before = run; }
                    run += bsum[b];
                }
            } else {
                s_before = run;
            }
            s_blk = pickb;
        }
        __syncthreads();
        if (s_blk < 0 && total > 0.0) {
            // u * total rounded up to the total: no range matched.  The host form's searchsorted is clipped to the last index WITH
            // mass: the last block with a positive sum, then (target past its end) its last point with d2 > 0 - never a point
            // that already is a centre (M - 1 could be one, giving a duplicate centre and an empty cluster)
            int lastb = -1;
            for (int e = 0; e < per; ++e) {
                const int b = tid * per + e;
                if (b < nblk && bsum[b] > 0.0) lastb = b;
            }
            if (lastb >= 0) atomicMax(&s_blk, lastb);
            if (tid == 0) { s_before = 0.0; s_target = 1.0e300; }
            __syncthreads();
        }
        if (s_blk < 0) {                                            // no mass left: every point coincides with a centre
            if (tid == 0) {
                long long i = (long long)(u[t_new - 1] * (double)M);
                s_pick = i < 0 ? 0 : (i >= M ? M - 1 : i);
            }
            __syncthreads();
        } else {
            // level 2: the block's 1024 values, one per thread
            const long long i = (long long)s_blk * KPP_BLOCK + tid;
            const double v = i < M ? (double)d2[i] : 0.0;
            double btot;
            const double in2 = s_before + block_scan_incl(v, buf, tid, KPP_PICK_THREADS, btot);
            if (tid == 0) { s_win = KPP_PICK_THREADS; s_last = -1; }
            __syncthreads();
            if (v > 0.0 && in2 > s_target && in2 - v <= s_target) atomicMin(&s_win, tid);
            if (v > 0.0) atomicMax(&s_last, tid);
            __syncthreads();
            // (no match: rounding between the two levels put the target past the block's end - its last point with mass)
            if (tid == 0) s_pick = (long long)s_blk * KPP_BLOCK + (s_win < KPP_PICK_THREADS ? s_win : (s_last < 0 ? 0 : s_last));
            __syncthreads();
        }
    }
    const long long pk = s_pick;
    if (tid == 0) picks[t_new] = pk;
    // the new centre's row, and its squared distance to every earlier centre (the next update's pruning table)
    float *crow = centers + (size_t)t_new * C;
    for (int c = tid; c < C; c += KPP_PICK_THREADS) crow[c] = pts[(size_t)pk * C + c];
    // |c_new - c_a|^2 for every earlier centre: a wave per centre, lanes along the channels (coalesced rows, one shuffle tree)
    {
        const int lane = tid & 63, wv = tid >> 6;
        const float *nr = pts + (size_t)pk * C;
        for (int a = wv; a <= t_new; a += KPP_PICK_THREADS / 64) {
            float s_ = 0.f;
            if (a < t_new) {
                const float *ar = centers + (size_t)a * C;
                for (int c = lane; c < C; c += 64) {
                    const float d = ar[c] - nr[c];
                    s_ = fmaf(d, d, s_);
                }
            }
            s_ = wave_sum(s_);
            if (lane == 0) cc[a] = s_;
        }
    }
}


// ---------------------------------------------------------------------------------------------------------------------------
// GREEDY k-means++ : what sklearn's KMeans(init='k-means++') actually runs (sklearn/cluster/_kmeans.py `_kmeans_plusplus`,
// scikit-learn 1.7.2; called by get_basis.py:210 of the reference).  Per pick: T = 2 + int(log K) candidates are drawn by D^2
// sampling from T variates (`searchsorted(cumsum(d2), u * potential)`, side 'left'), each candidate's potential
// sum_i min(d2[i], |p_i - x_cand|^2) is evaluated and the candidate with the lowest potential becomes the centre.
//   kpp_cand_kernel    one workgroup: the two-level cdf search for the T variates; candidate rows; |x_cand - c_a|^2 for every
//                      earlier centre (per candidate, and the minimum over the candidates: the gain pass's pruning table)
//   kpp_gain_kernel    one pass over the points: gain_j = sum_i max(0, d2[i] - |p_i - x_cand_j|^2) in fp64 per block of 1024
//                      points (potential_j = potential - gain_j).  A point whose nearest centre a has
//                      min_j |x_cand_j - c_a|^2 >= 4 d2[i] cannot gain from any candidate: its row is not read.
//   kpp_choose_kernel  block gains added in block order -> the first candidate with the largest gain (numpy's argmin of the
//                      potentials); its row, index and centre distances become centre t.
//   kpp_update_kernel  (above) then folds the new centre into d2 / nearest / block sums, reading ONLY the rows the winner moves:
//                      the gain pass leaves one byte per point (bit j = candidate j comes closer than the nearest centre).
// ---------------------------------------------------------------------------------------------------------------------------
constexpr int KPP_TMAX = 8;

template <int CS>           // 16-byte pieces per lane of a 16-lane group in the centre-distance loop: ceil(C / 64)
__global__ void __launch_bounds__(KPP_PICK_THREADS)
kpp_cand_kernel(const float *__restrict__ pts, long long M, int C, int K, int t_new, int T, const double *__restrict__ u,
                const float *__restrict__ d2, const double *__restrict__ bsum, int nblk, const float *__restrict__ centers,
                long long *__restrict__ cand, float *__restrict__ candrows, float *__restrict__ cct, float *__restrict__ ccmin)
{
    __shared__ double buf[KPP_PICK_THREADS];
    __shared__ long long s_cand[KPP_TMAX];
    __shared__ int s_win[KPP_TMAX], s_blk[KPP_TMAX];
    __shared__ double s_before[KPP_TMAX];
    const int tid = threadIdx.x;
    // level 1, all T variates at once: thread i owns `per` consecutive blocks; the first thread whose inclusive mass reaches a
    // target walks its blocks for it
    const int per = (nblk + KPP_PICK_THREADS - 1) / KPP_PICK_THREADS;
    double mine = 0.0;
    for (int e = 0; e < per; ++e) {
        const int b = tid * per + e;
        if (b < nblk) mine += bsum[b];
    }
    double total;
    const double incl = block_scan_incl(mine, buf, tid, KPP_PICK_THREADS, total);
    if (tid < KPP_TMAX) { s_win[tid] = KPP_PICK_THREADS; s_blk[tid] = -1; s_before[tid] = 0.0; }
    __syncthreads();
    for (int j = 0; j < T; ++j) {
        const double target = u[j] * total;
        if (target > 0.0 && incl >= target) atomicMin(&s_win[j], tid);
    }
    __syncthreads();
    for (int j = 0; j < T; ++j) {
        if (tid != s_win[j]) continue;
        const double target = u[j] * total;
        double run = incl - mine;
        int pickb = -1;
        for (int e = 0; e < per; ++e) {
            const int b = tid * per + e;
            if (b >= nblk) break;
            const double nxt = run + bsum[b];
            if (nxt >= target) { pickb = b; break; }
            run = nxt;
        }
        if (pickb < 0) {                                        // rounding at the range's end: its last block with mass
            run = incl - mine;
            for (int e = 0; e < per; ++e) {
                const int b = tid * per + e;
                if (b >= nblk) break;
                if (bsum[b] > 0.0) { pickb = b; s_before[j] = run; }
                run += bsum[b];
            }
        } else {
            s_before[j] = run;
        }
        s_blk[j] = pickb;
    }
    __syncthreads();
    // level 2: wave j searches its block's 1024 values - lane l owns values 16 l .. 16 l + 15 (a running sum), the lane totals are
    // scanned across the wave, the first lane holding a crossing names the point
    {
        const int lane = tid & 63, wv = tid >> 6;
        if (wv < T) {
            const int jq = wv;
            if (s_blk[jq] < 0) {
                // target <= 0 (u == 0 or no mass left: every point coincides with a centre): searchsorted(cs, 0) = 0, as sklearn
                if (lane == 0) s_cand[jq] = 0;
            } else {
                const double target = u[jq] * total;
                const long long i0 = (long long)s_blk[jq] * KPP_BLOCK + 16 * lane;
                float v[16];
#pragma unroll
                for (int e = 0; e < 16; ++e) v[e] = (i0 + e < M) ? d2[i0 + e] : 0.f;
                double tot = 0.0;
#pragma unroll
                for (int e = 0; e < 16; ++e) tot += (double)v[e];
                double inc = tot;                                   // inclusive scan of the lane totals
#pragma unroll
                for (int o = 1; o < 64; o <<= 1) {
                    const double up = __shfl_up(inc, o, 64);
                    if (lane >= o) inc += up;
                }
                double run = s_before[jq] + (inc - tot);
                int hit = -1, lastmass = -1;
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    run += (double)v[e];
                    if (v[e] > 0.f) {
                        lastmass = 16 * lane + e;
                        if (hit < 0 && run >= target) hit = 16 * lane + e;
                    }
                }
                const unsigned long long any = __ballot(hit >= 0);
                int pos;
                if (any) pos = __shfl(hit, (int)__builtin_ctzll(any), 64);
                else pos = max(wave_max(lastmass), 0);              // rounding put the target past the block's end: its last point with mass
                if (lane == 0) s_cand[jq] = (long long)s_blk[jq] * KPP_BLOCK + pos;
            }
        }
    }
    __syncthreads();
    if (tid < T) cand[tid] = s_cand[tid];
    for (int e = tid; e < T * C; e += KPP_PICK_THREADS) {
        const int jj = e / C, c = e - jj * C;
        candrows[e] = pts[(size_t)s_cand[jj] * C + c];
    }
    // |x_cand_j - c_a|^2 for every (candidate j, earlier centre a) pair: a 16-lane group per pair (64 pairs per step of the
    // workgroup), 16-byte pieces q + 16 s of both rows, one 4-step shuffle tree; then min_j per centre
    __shared__ float s_cct[KPP_TMAX][256];
    __syncthreads();                                     // (s_cand is complete; the rows themselves are read from `pts`)
    {
        const int lane = tid & 63, q = lane & 15;
        const int gidx = tid >> 4, ngrp = KPP_PICK_THREADS / 16;
        const int c4n = C >> 2;
        const int npair = T * t_new;
        for (int pr = gidx; pr < ((npair + ngrp - 1) / ngrp) * ngrp; pr += ngrp) {
            const bool live = pr < npair;
            const int jj = live ? pr / t_new : 0, a = live ? pr - jj * t_new : 0;
            const float *xr = pts + (size_t)s_cand[jj] * C, *ar = centers + (size_t)a * C;
            float s_ = 0.f;
#pragma unroll
            for (int sgm = 0; sgm < CS; ++sgm) {
                const int pc = q + 16 * sgm;
                if (pc < c4n) {
                    const kf4 d = *reinterpret_cast<const kf4 *>(ar + 4 * pc) - *reinterpret_cast<const kf4 *>(xr + 4 * pc);
                    s_ = fmaf(d.x, d.x, s_);
                    s_ = fmaf(d.y, d.y, s_);
                    s_ = fmaf(d.z, d.z, s_);
                    s_ = fmaf(d.w, d.w, s_);
                }
            }
            s_ = row16_sum(s_);
            if (live && q == 0) {
                cct[jj * K + a] = s_;
                if (a < 256) s_cct[jj][a] = s_;
            }
        }
    }
    __syncthreads();
    for (int a = tid; a < t_new; a += KPP_PICK_THREADS) {
        float mn = 3.0e38f;
        for (int jj = 0; jj < T; ++jj) mn = fminf(mn, a < 256 ? s_cct[jj][a] : cct[jj * K + a]);
        ccmin[a] = mn;
    }
}

template <int C4MAX>
__global__ void __launch_bounds__(256)
kpp_gain_kernel(const float *__restrict__ pts, long long M, int C, int T, const float *__restrict__ candrows,
                const float *__restrict__ ccmin, int prune, const float *__restrict__ d2, const int *__restrict__ nearest,
                double *__restrict__ gsum, unsigned char *__restrict__ cmask)
{
    __shared__ double wsum[4][KPP_TMAX];
    // the candidate rows are staged through LDS once; the compiler keeps this lane's pieces of them in registers across the trips
    // (measured: re-reading them from LDS every trip - 7 waves per SIMD instead of 3 - made the pass LDS-bound: 367 -> 605 us)
    __shared__ kf4 scand[KPP_TMAX][C4MAX * 16];
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int q = lane & 15, grp = lane >> 4;
    const int c4n = C >> 2;
    for (int e = threadIdx.x; e < KPP_TMAX * C4MAX * 16; e += 256) {
        const int j = e / (C4MAX * 16), pc = e - j * (C4MAX * 16);
        scand[j][pc] = (j < T && pc < c4n) ? *reinterpret_cast<const kf4 *>(candrows + (size_t)j * C + 4 * pc) : kf4{0.f, 0.f, 0.f, 0.f};
    }
    __syncthreads();
    double acc[KPP_TMAX];
#pragma unroll
    for (int j = 0; j < KPP_TMAX; ++j) acc[j] = 0.0;
    const long long base = (long long)blockIdx.x * KPP_BLOCK + wv * 256;
    for (int ch = 0; ch < 4; ++ch) {
        const long long row0 = base + 64 * ch;
        if (row0 >= M) break;
        const long long mine = row0 + lane;
        const bool have = mine < M;
        float myd = 0.f;
        bool need = false;
        if (have) {
            myd = d2[mine];
            need = myd > 0.f;
            // |p - x| >= |x - c_a| - |p - c_a| >= r for every candidate x when min_j |x_j - c_a|^2 >= 4 r^2: no candidate gains here
            if (need && prune && ccmin[nearest[mine]] >= 4.f * myd * KPP_PRUNE_MARGIN) need = false;
        }
        unsigned long long mask = __ballot(need);
        unsigned mymask = 0u;                                      // this lane's row: bit j = candidate j comes closer than d2
        // wave-uniform trips of up to 4 rows (16 lanes per row); the rows of trip n + 1 are requested before trip n is computed
        int b[4], bn[4];
        kf4 v[C4MAX], vn[C4MAX];
        auto take = [&](int (&bb)[4]) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                bb[g] = mask ? (int)__builtin_ctzll(mask) : -1;
                mask &= mask - 1;
            }
        };
        auto fetch = [&](const int (&bb)[4], kf4 (&vv)[C4MAX]) {
            const int r = grp == 0 ? bb[0] : grp == 1 ? bb[1] : grp == 2 ? bb[2] : bb[3];
            const float *row = pts + (size_t)(row0 + max(r, 0)) * C;
#pragma unroll
            for (int s = 0; s < C4MAX; ++s) {
                const int pc = q + 16 * s;
                vv[s] = (r >= 0 && pc < c4n) ? *reinterpret_cast<const kf4 *>(row + 4 * pc) : kf4{0.f, 0.f, 0.f, 0.f};
            }
        };
        bool more = mask != 0;
        if (more) { take(b); fetch(b, v); }
        while (more) {
            const bool next = mask != 0;
            if (next) { take(bn); fetch(bn, vn); }
            const int r = grp == 0 ? b[0] : grp == 1 ? b[1] : grp == 2 ? b[2] : b[3];
            const float rd = __shfl(myd, r < 0 ? 0 : r, 64);      // the row's current d2
            unsigned gm = 0u;
#pragma unroll
            for (int j = 0; j < KPP_TMAX; ++j) {
                float s_ = 0.f;
#pragma unroll
                for (int s = 0; s < C4MAX; ++s) {
                    const kf4 d = v[s] - scand[j][q + 16 * s];
                    s_ = fmaf(d.x, d.x, s_);
                    s_ = fmaf(d.y, d.y, s_);
                    s_ = fmaf(d.z, d.z, s_);
                    s_ = fmaf(d.w, d.w, s_);
                }
                s_ = row16_sum(s_);
                if (r >= 0 && j < T && s_ < rd) {
                    if (q == 0) acc[j] += (double)(rd - s_);
                    gm |= 1u << j;
                }
            }
#pragma unroll
            for (int g = 0; g < 4; ++g) {                          // the row's candidate bits go to the lane that owns the row
                const unsigned mg = (unsigned)__shfl((int)gm, 16 * g, 64);
                if (lane == b[g]) mymask = mg;
            }
            more = next;
#pragma unroll
            for (int g = 0; g < 4; ++g) b[g] = bn[g];
#pragma unroll
            for (int s = 0; s < C4MAX; ++s) v[s] = vn[s];
        }
        // which candidates MOVE this point: the winner's update then reads only the rows that change (one byte per point
        // written here instead of a second full pass over the 4 C M bytes of points)
        if (have) cmask[mine] = (unsigned char)mymask;
    }
#pragma unroll
    for (int j = 0; j < KPP_TMAX; ++j) {
        const double w = wave_sum(acc[j]);
        if (lane == 0) wsum[wv][j] = w;
    }
    __syncthreads();
    if (threadIdx.x < KPP_TMAX)
        gsum[(size_t)blockIdx.x * KPP_TMAX + threadIdx.x] = ((wsum[0][threadIdx.x] + wsum[1][threadIdx.x]) + wsum[2][threadIdx.x]) + wsum[3][threadIdx.x];
}

__global__ void __launch_bounds__(256)
kpp_choose_kernel(int C, int K, int t_new, int T, const double *__restrict__ gsum, int nblk, const long long *__restrict__ cand,
                  const float *__restrict__ candrows, const float *__restrict__ cct, float *__restrict__ centers,
                  long long *__restrict__ picks, float *__restrict__ cc, int *__restrict__ best_out)
{
    __shared__ double part[256][KPP_TMAX + 1];
    __shared__ int s_best;
    const int tid = threadIdx.x;
    double a[KPP_TMAX];
#pragma unroll
    for (int j = 0; j < KPP_TMAX; ++j) a[j] = 0.0;
    const int per = (nblk + 255) / 256;                      // consecutive blocks per thread, then a fixed tree: one summation order
    for (int e = 0; e < per; ++e) {
        const int b = tid * per + e;
        if (b < nblk) {
#pragma unroll
            for (int j = 0; j < KPP_TMAX; ++j) a[j] += gsum[(size_t)b * KPP_TMAX + j];
        }
    }
#pragma unroll
    for (int j = 0; j < KPP_TMAX; ++j) part[tid][j] = a[j];
    __syncthreads();
    for (int h = 128; h > 0; h >>= 1) {
        if (tid < h) {
#pragma unroll
            for (int j = 0; j < KPP_TMAX; ++j) part[tid][j] += part[tid + h][j];
        }
        __syncthreads();
    }
    if (tid == 0) {
        int best = 0;
        for (int j = 1; j < T; ++j)
            if (part[0][j] > part[0][best]) best = j;            // the first of equal potentials, as numpy's argmin
        s_best = best;
        picks[t_new] = cand[best];
        *best_out = best;
    }
    __syncthreads();
    const int best = s_best;
    for (int c = tid; c < C; c += 256) centers[(size_t)t_new * C + c] = candrows[(size_t)best * C + c];
    for (int a2 = tid; a2 <= t_new; a2 += 256) cc[a2] = a2 < t_new ? cct[best * K + a2] : 0.f;
}

}  // namespace
}  // namespace gfs3d

using namespace gfs3d;

extern "C" size_t gfs3d_kmeanspp_workspace_bytes(long long M, int K)
{
    if (M < 1 || K < 1) return 0;
    const size_t nblk = (size_t)((M + KPP_BLOCK - 1) / KPP_BLOCK);
    return align_up(sizeof(float) * (size_t)M, 256) + align_up(sizeof(int) * (size_t)M, 256) + align_up(sizeof(double) * nblk, 256) +
           align_up(sizeof(float) * (size_t)K, 256);
}

extern "C" int gfs3d_kmeanspp_seed_f32(const float *pts, long long M, int C, int K, long long first, const double *u, float *centers,
                                       long long *picks, void *workspace, size_t workspace_bytes, int prune, void *stream)
{
    const char *who = "gfs3d_kmeanspp_seed_f32";
    GFS3D_REQUIRE(pts && u && centers && picks, "%s: NULL argument", who);
    GFS3D_REQUIRE(M >= 1 && K >= 1 && K <= M, "%s: need 1 <= K <= M (M=%lld K=%d)", who, M, K);
    GFS3D_REQUIRE(C >= 4 && C % 4 == 0 && C <= 1024, "%s: C=%d unsupported (a multiple of 4, <= 1024)", who, C);
    GFS3D_REQUIRE(first >= 0 && first < M, "%s: first=%lld out of range", who, first);
    GFS3D_REQUIRE((M + KPP_BLOCK - 1) / KPP_BLOCK < (1ll << 30), "%s: M too large", who);
    GFS3D_REQUIRE(workspace && workspace_bytes >= gfs3d_kmeanspp_workspace_bytes(M, K), "%s: workspace too small", who);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int nblk = (int)((M + KPP_BLOCK - 1) / KPP_BLOCK);
    char *ws = static_cast<char *>(workspace);
    float *d2 = reinterpret_cast<float *>(ws);
    ws += align_up(sizeof(float) * (size_t)M, 256);
    int *nearest = reinterpret_cast<int *>(ws);
    ws += align_up(sizeof(int) * (size_t)M, 256);
    double *bsum = reinterpret_cast<double *>(ws);
    ws += align_up(sizeof(double) * (size_t)nblk, 256);
    float *cc = reinterpret_cast<float *>(ws);
    const int c4 = (C / 4 + 15) / 16;                          // float4 pieces per lane
    for (int t = 0; t < K; ++t) {
        hipLaunchKernelGGL(kpp_pick_kernel, dim3(1), dim3(KPP_PICK_THREADS), 0, st, pts, M, C, t, first, u, d2, bsum, nblk, centers, picks, cc);
        GFS3D_LAUNCH_CHECK("kpp_pick_kernel");
        if (t == K - 1) break;
#define KPP_LAUNCH(N_)                                                                                                          \
    hipLaunchKernelGGL(kpp_update_kernel<N_>, dim3(nblk), dim3(256), 0, st, pts, M, C, centers, t, cc, prune, d2, nearest, bsum, \
                       (const unsigned char *)nullptr, (const int *)nullptr)
        if (c4 <= 1) KPP_LAUNCH(1);
        else if (c4 <= 2) KPP_LAUNCH(2);
        else if (c4 <= 3) KPP_LAUNCH(3);
        else if (c4 <= 4) KPP_LAUNCH(4);
        else if (c4 <= 8) KPP_LAUNCH(8);
        else KPP_LAUNCH(16);
#undef KPP_LAUNCH
        GFS3D_LAUNCH_CHECK("kpp_update_kernel");
    }
    return GFS3D_OK;
}

// workspace: the plain form's (d2 | nearest | block sums | cc) + candidate ids, rows, centre-distance tables, block gains
extern "C" size_t gfs3d_kmeanspp_greedy_workspace_bytes(long long M, int C, int K)
{
    if (M < 1 || K < 1 || C < 1) return 0;
    const size_t nblk = (size_t)((M + KPP_BLOCK - 1) / KPP_BLOCK);
    return gfs3d_kmeanspp_workspace_bytes(M, K) + align_up(sizeof(long long) * KPP_TMAX, 256) +
           align_up(sizeof(float) * (size_t)KPP_TMAX * C, 256) + align_up(sizeof(float) * (size_t)KPP_TMAX * K, 256) +
           align_up(sizeof(float) * (size_t)K, 256) + align_up(sizeof(double) * nblk * KPP_TMAX, 256) + 256 +
           align_up((size_t)M, 256);
}

extern "C" int gfs3d_kmeanspp_greedy_f32(const float *pts, long long M, int C, int K, long long first, const double *u, int n_trials,
                                         float *centers, long long *picks, void *workspace, size_t workspace_bytes, int prune,
                                         void *stream)
{
    const char *who = "gfs3d_kmeanspp_greedy_f32";
    GFS3D_REQUIRE(pts && u && centers && picks, "%s: NULL argument", who);
    GFS3D_REQUIRE(M >= 1 && K >= 1 && K <= M, "%s: need 1 <= K <= M (M=%lld K=%d)", who, M, K);
    GFS3D_REQUIRE(C >= 4 && C % 4 == 0 && C <= 256, "%s: C=%d unsupported (a multiple of 4, <= 256)", who, C);
    GFS3D_REQUIRE(n_trials >= 1 && n_trials <= KPP_TMAX, "%s: n_trials=%d unsupported (1..%d)", who, n_trials, KPP_TMAX);
    GFS3D_REQUIRE(first >= 0 && first < M, "%s: first=%lld out of range", who, first);
    GFS3D_REQUIRE((M + KPP_BLOCK - 1) / KPP_BLOCK < (1ll << 30), "%s: M too large", who);
    GFS3D_REQUIRE(workspace && workspace_bytes >= gfs3d_kmeanspp_greedy_workspace_bytes(M, C, K), "%s: workspace too small", who);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int nblk = (int)((M + KPP_BLOCK - 1) / KPP_BLOCK);
    char *ws = static_cast<char *>(workspace);
    float *d2 = reinterpret_cast<float *>(ws);
    ws += align_up(sizeof(float) * (size_t)M, 256);
    int *nearest = reinterpret_cast<int *>(ws);
    ws += align_up(sizeof(int) * (size_t)M, 256);
    double *bsum = reinterpret_cast<double *>(ws);
    ws += align_up(sizeof(double) * (size_t)nblk, 256);
    float *cc = reinterpret_cast<float *>(ws);
    ws += align_up(sizeof(float) * (size_t)K, 256);
    long long *cand = reinterpret_cast<long long *>(ws);
    ws += align_up(sizeof(long long) * KPP_TMAX, 256);
    float *candrows = reinterpret_cast<float *>(ws);
    ws += align_up(sizeof(float) * (size_t)KPP_TMAX * C, 256);
    float *cct = reinterpret_cast<float *>(ws);
    ws += align_up(sizeof(float) * (size_t)KPP_TMAX * K, 256);
    float *ccmin = reinterpret_cast<float *>(ws);
    ws += align_up(sizeof(float) * (size_t)K, 256);
    double *gsum = reinterpret_cast<double *>(ws);
    ws += align_up(sizeof(double) * (size_t)nblk * KPP_TMAX, 256);
    int *best = reinterpret_cast<int *>(ws);
    ws += 256;
    unsigned char *cmask = reinterpret_cast<unsigned char *>(ws);     // [M]: bit j = candidate j moves the point
    const int c4 = (C / 4 + 15) / 16;                          // float4 pieces per lane of the row kernels (<= 4)
    const int cs = (C + 63) / 64;                              // channels per lane of the centre-distance loop (<= 4)
#define KPP_UPDATE(N_, t_)                                                                                                      \
    hipLaunchKernelGGL(kpp_update_kernel<N_>, dim3(nblk), dim3(256), 0, st, pts, M, C, centers, t_, cc, prune, d2, nearest, bsum, \
                       (t_) > 0 ? cmask : (const unsigned char *)nullptr, best)
#define KPP_GAIN(N_)                                                                                                            \
    hipLaunchKernelGGL(kpp_gain_kernel<N_>, dim3(nblk), dim3(256), 0, st, pts, M, C, n_trials, candrows, ccmin, prune, d2, nearest, gsum, cmask)
#define KPP_CAND(N_, t_)                                                                                                        \
    hipLaunchKernelGGL(kpp_cand_kernel<N_>, dim3(1), dim3(KPP_PICK_THREADS), 0, st, pts, M, C, K, t_, n_trials,                  \
                       u + (size_t)(t_ - 1) * n_trials, d2, bsum, nblk, centers, cand, candrows, cct, ccmin)
    // centre 0: the given index (its row through the plain pick kernel), then the first distances
    hipLaunchKernelGGL(kpp_pick_kernel, dim3(1), dim3(KPP_PICK_THREADS), 0, st, pts, M, C, 0, first, u, d2, bsum, nblk, centers, picks, cc);
    GFS3D_LAUNCH_CHECK("kpp_pick_kernel");
    for (int t = 0; t < K; ++t) {
        if (t > 0) {
            if (cs <= 1) KPP_CAND(1, t);
            else if (cs <= 2) KPP_CAND(2, t);
            else if (cs <= 3) KPP_CAND(3, t);
            else KPP_CAND(4, t);
            GFS3D_LAUNCH_CHECK("kpp_cand_kernel");
            if (c4 <= 1) KPP_GAIN(1);
            else if (c4 <= 2) KPP_GAIN(2);
            else if (c4 <= 3) KPP_GAIN(3);
            else KPP_GAIN(4);
            GFS3D_LAUNCH_CHECK("kpp_gain_kernel");
            hipLaunchKernelGGL(kpp_choose_kernel, dim3(1), dim3(256), 0, st, C, K, t, n_trials, gsum, nblk, cand, candrows, cct, centers, picks, cc, best);
            GFS3D_LAUNCH_CHECK("kpp_choose_kernel");
        }
        if (t == K - 1) break;
        if (c4 <= 1) KPP_UPDATE(1, t);
        else if (c4 <= 2) KPP_UPDATE(2, t);
        else if (c4 <= 3) KPP_UPDATE(3, t);
        else KPP_UPDATE(4, t);
        GFS3D_LAUNCH_CHECK("kpp_update_kernel");
    }
#undef KPP_UPDATE
#undef KPP_GAIN
#undef KPP_CAND
    return GFS3D_OK;
}
