// K2: EdgeConv block (gather + 2x[1x1 conv, BatchNorm2d, LeakyReLU] + max over k),
// forward and backward, for gfx950.  Replaces get_edge_feature + conv2d block +
// max (model/dgcnn.py:26-61,116-118) without ever materialising (B,2C,N,k) or
// any (B,64,N,k) tensor.
//
// Algebra used (all exact rewrites in real arithmetic; fp32 rounding differs from
// the reference by ~1e-7 relative, inside the 1e-4 contract):
//   conv1 on cat(xj - xi, xi):  W1 [xj-xi; xi] = Wa xj + (Wb-Wa) xi = P[j] + Q[i]
//       -> two N-wide projections instead of an (N*k)-wide conv (k x fewer flops).
//   max_j LReLU(BN2(h2_j)) = LReLU(BN2(max_j h2_j))  if scale2 >= 0, min_j otherwise
//       -> the heavy kernel only tracks max/min/arg of the pre-BN conv2 output, so
//          BN2 batch statistics and the output come out of ONE pass over the edges.
//   backward of conv2+BN2 (training): dh2 = A*route + Bc + Cc*h2 (A,Bc,Cc per
//       channel), hence  da1 = M a1 + v0 + sparse(route)  with M = W2^T diag(Cc) W2
//       and  dW2 = sparse + Bc (x) sum(a1) + diag(Cc) W2 S,  S = sum_e a1 a1^T.
//       The backward needs no h2 recomputation and no stored (B,64,N,k) tensor.
//   backward of BN1 needs only per-point sums: G_Q[n] = sum_j dy1[n,j],
//       G_P[m] = sum over in-edges of dy1, SP, SQ, indeg (see ec_bwd_pq_kernel).
//
// Layouts: API tensors are channel-first (B,C,N) like the reference; everything
// internal is point-major (B,N,64) so one point's 64 channels are one 256-B row
// (one coalesced wave64 load / one 256-B atomic row).
#include <map>
#include <mutex>
#include <utility>

#include "common.h"

namespace gfs3d {

constexpr int EC = GFS3D_EC_WIDTH;   // 64
typedef float f32x16 __attribute__((ext_vector_type(16)));

// stats rows
enum { ST_MEAN1 = 0, ST_RSTD1, ST_SCALE1, ST_SHIFT1, ST_MEAN2, ST_RSTD2, ST_SCALE2, ST_SHIFT2 };

// Where a block's [2][64] fp64 partial sums go: float64 atomics on the 128 accumulators (fast; the order of the adds can
// move the double sum in its last bit), or - gfs3d_set_deterministic(1) - row `blk` of a partial table that
// bn_reduce_rows_kernel / bn_reduce_final_kernel then add in a fixed order.
struct BnAcc {
    double *acc;        // [2][64] accumulators (atomics mode: zeroed by the launcher)
    double *rows;       // ordered mode: [nblocks][2][64], else NULL
};
__device__ __forceinline__ void bn_acc_add(const BnAcc &a, int blk, int which, int c, double v)
{
    if (a.rows) a.rows[(size_t)blk * 2 * GFS3D_EC_WIDTH + which * GFS3D_EC_WIDTH + c] = v;
    else atomicAdd(&a.acc[which * GFS3D_EC_WIDTH + c], v);
}
// ordered mode, level 1: block g adds rows g R .. g R + R - 1 (thread = (row slot, element): coalesced 1-KB rows)
constexpr int BNR_GROUPS = 16;
__global__ void __launch_bounds__(1024)
bn_reduce_rows_kernel(const double *__restrict__ rows, int nrows, double *__restrict__ mid)
{
    __shared__ double red[8][128];
    const int e = threadIdx.x & 127, slot = threadIdx.x >> 7;
    const int per = (nrows + BNR_GROUPS - 1) / BNR_GROUPS;
    const int r0 = blockIdx.x * per, r1 = min(nrows, r0 + per);
    double s = 0.0;
    for (int r = r0 + slot; r < r1; r += 8) s += rows[(size_t)r * 128 + e];
    red[slot][e] = s;
    __syncthreads();
    if (slot == 0) {
        double t = 0.0;
#pragma unroll
        for (int q = 0; q < 8; ++q) t += red[q][e];
        mid[blockIdx.x * 128 + e] = t;
    }
}
__global__ void bn_reduce_final_kernel(const double *__restrict__ mid, double *__restrict__ acc)
{
    const int e = threadIdx.x;
    double t = 0.0;
#pragma unroll
    for (int g = 0; g < BNR_GROUPS; ++g) t += mid[g * 128 + e];
    acc[e] = t;
}

// XCD-aware dealing of work units (speed only: no result depends on which workgroup takes a unit).  Workgroups go
// round-robin to the 8 XCDs (blockIdx % 8, MI355X_MICROARCH.md) and each XCD has its own 4 MB L2.  The gather kernels read a
// cloud's table of 64-channel rows (P', Q, dy1: N x 256 B = 512 KB at N = 2048) k times per point; dealt plainly every XCD
// touches every cloud (16.8 MB at B = 32) and the row gathers miss its L2.  Dealt by cloud - the workgroups of XCD x take the
// clouds c with c % 8 == x - one L2 holds B/8 tables.  `upc` = units (super-tiles / points / chunks) per cloud; used when
// B % 8 == 0, the grid is a multiple of 8 and no unit straddles two clouds, the plain round-robin otherwise.
struct UnitDeal {
    int first, stride, count, upc, x;
    bool xcd;
    __device__ __forceinline__ UnitDeal(int nunits, int upc_, int B, bool whole, int workers, int worker)
    {
#ifdef EC_PLAIN_DEAL
        xcd = false;
#else
        xcd = whole && (B & 7) == 0 && (gridDim.x & 7) == 0;
#endif
        upc = upc_;
        if (xcd) {
            x = blockIdx.x & 7;
            first = (blockIdx.x >> 3) * workers + worker;
            stride = (gridDim.x >> 3) * workers;
            count = (B >> 3) * upc;
        } else {
            x = 0;
            first = blockIdx.x * workers + worker;
            stride = gridDim.x * workers;
            count = nunits;
        }
    }
    __device__ __forceinline__ int unit(int u) const
    {
        if (!xcd) return u;
        const int c = u / upc;
        return (x + 8 * c) * upc + (u - c * upc);
    }
};

// ---------------------------------------------------------------------------
// F2: BN1 batch statistics of h1 = P[idx] + Q over all edges, and SP (pure gathers).
// wave per point, lane = channel.  partial: [gridDim.x][2][64] doubles.
// ---------------------------------------------------------------------------
constexpr int ST1_FLIGHT = 20;      // row gathers in flight per lane (k = 20: a point's whole neighbourhood; 10: 52.7 -> 45.8 us)
constexpr int ST1_GRID_DIV = 4;     // super-tile groups per workgroup (2, i.e. twice the workgroups: 45.8 -> 59.9 us)
constexpr int ST1_WAVES = 16;       // every block ends in 128 same-address fp64 atomics (~14 ns each, serialised): few big blocks
__global__ void __launch_bounds__(64 * ST1_WAVES)
ec_stats1_kernel(const float *__restrict__ P, const float *__restrict__ Q, const int32_t *__restrict__ idx,
                 int BN, int N, int k, float *__restrict__ SP, const BnAcc partial)
{
    __shared__ double red[ST1_WAVES][2][EC];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    double s = 0.0, ss = 0.0;
    const UnitDeal deal(BN, N, BN / N, true, ST1_WAVES, wv);
    for (int w = deal.first; w < deal.count; w += deal.stride) {
        const int pt = deal.unit(w);
        const int b = pt / N;
        const float q = Q[(size_t)pt * EC + lane];
        const int myidx = (lane < k) ? idx[(size_t)pt * k + lane] : 0;
        float sp = 0.f, s1 = 0.f, s2 = 0.f;
        for (int j0 = 0; j0 < k; j0 += ST1_FLIGHT) {      // ST1_FLIGHT independent row gathers in flight
            float pv[ST1_FLIGHT];
#pragma unroll
            for (int u = 0; u < ST1_FLIGHT; ++u) {
                // (a wave-uniform source lane: v_readlane into a scalar, not a trip through the LDS crossbar as __shfl compiles to)
                const int m = __builtin_amdgcn_readlane(myidx, min(j0 + u, k - 1));
                pv[u] = P[((size_t)b * N + m) * EC + lane];
            }
#pragma unroll
            for (int u = 0; u < ST1_FLIGHT; ++u) {
                if (j0 + u < k) {
                    const float h = pv[u] + q;
                    sp += pv[u];
                    s1 += h;
                    s2 = fmaf(h, h, s2);
                }
            }
        }
        SP[(size_t)pt * EC + lane] = sp;
        s += (double)s1;
        ss += (double)s2;
    }
    red[wv][0][lane] = s;
    red[wv][1][lane] = ss;
    __syncthreads();
    if (threadIdx.x < 128) {
        const int which = threadIdx.x >> 6;
        double t = 0.0;
#pragma unroll
        for (int w = 0; w < ST1_WAVES; ++w) t += red[w][which][lane];
        bn_acc_add(partial, blockIdx.x, which, lane, t);
    }
}

// ---------------------------------------------------------------------------
// Reverse kNN graph, built in the backward pass: SQ[m] = sum over in-edges (n -> m) of Q[n] and
// the in-degree are what BN1's backward needs per target point.  One pass claims a slot per
// in-edge in a fixed-capacity table (integer atomics only; in-degrees above the capacity - hubs
// in high-dimensional feature space - go to an overflow list), then the float sums are plain
// gathers of 256-B rows.  (A float-atomic scatter of Q costs 0.37 ms per layer, this ~0.1 ms.)
// ---------------------------------------------------------------------------
constexpr int REV_CAP = 64;
__global__ void __launch_bounds__(256)
rev_slot_kernel(const int32_t *__restrict__ idx, size_t nedges, int N, int k, int *__restrict__ cnt,
                int *__restrict__ slot, int *__restrict__ ovf_cnt, int2 *__restrict__ ovf)
{
    const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= nedges) return;
    const size_t pt = e / k;
    const size_t b = pt / N;
    const size_t tm = b * N + idx[e];
    const int pos = atomicAdd(&cnt[tm], 1);
    if (pos < REV_CAP) {
        slot[tm * REV_CAP + pos] = (int)pt;
    } else {
        const int o = atomicAdd(ovf_cnt, 1);
        ovf[o] = make_int2((int)tm, (int)pt);
    }
}
// The same table built per point-cloud block in LDS (N <= 8192): a workgroup owns RB_TARGETS targets of one block, scans
// that block's N k edges and appends the sources of its targets to a [RB_TARGETS][64] u16 table with LDS atomics; the
// table then goes out as whole coalesced rows (the global form above pays a returning device atomic and a scattered
// 4-byte store per edge: 51 -> 13 us per layer at B = 32 x 2048).
constexpr int RB_TARGETS = 1024;
constexpr size_t RB_LDS = sizeof(int) * RB_TARGETS + sizeof(unsigned short) * RB_TARGETS * REV_CAP;
__global__ void __launch_bounds__(1024)
rev_block_kernel(const int32_t *__restrict__ idx, int N, int k, int *__restrict__ cnt, int *__restrict__ slot,
                 int *__restrict__ ovf_cnt, int2 *__restrict__ ovf)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char rb_smem[];
    int *lcnt = reinterpret_cast<int *>(rb_smem);
    unsigned short *tab = reinterpret_cast<unsigned short *>(lcnt + RB_TARGETS);
    const int tid = threadIdx.x, b = blockIdx.y, m0 = blockIdx.x * RB_TARGETS;
    const int rows = min(RB_TARGETS, N - m0);
    const int base = b * N;
    for (int m = tid; m < RB_TARGETS; m += 1024) lcnt[m] = 0;
    __syncthreads();
    const int32_t *ib = idx + (size_t)base * k;
    const int ne = N * k;
    for (int e = tid; e < ne; e += 1024) {
        const unsigned m = (unsigned)(ib[e] - m0);
        if (m < (unsigned)rows) {
            const int n = e / k;
            const int pos = atomicAdd(&lcnt[m], 1);
            if (pos < REV_CAP) {
                tab[m * REV_CAP + pos] = (unsigned short)n;
            } else {
                const int o = atomicAdd(ovf_cnt, 1);
                ovf[o] = make_int2(base + m0 + (int)m, base + n);
            }
        }
    }
    __syncthreads();
    for (int q = tid; q < rows * REV_CAP; q += 1024) {      // entries past a row's count are never read
        const int m = q / REV_CAP, pos = q - m * REV_CAP;
        slot[(size_t)(base + m0 + m) * REV_CAP + pos] = base + (int)tab[q];
    }
    for (int m = tid; m < rows; m += 1024) cnt[base + m0 + m] = lcnt[m];
}
// wave per target point, lane = channel
__global__ void __launch_bounds__(256)
rev_gather_kernel(const float *__restrict__ Q, const int *__restrict__ cnt, const int *__restrict__ slot,
                  int BN, float *__restrict__ SQ, float *__restrict__ indeg)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int tm = blockIdx.x * 4 + wv; tm < BN; tm += gridDim.x * 4) {
        const int d = cnt[tm];
        const int dd = min(d, REV_CAP);
        const int mine = (lane < dd) ? slot[(size_t)tm * REV_CAP + lane] : 0;
        float acc = 0.f;
        int s = 0;
        for (; s + 8 <= dd; s += 8) {                      // 8 independent row gathers in flight
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = Q[(size_t)__shfl(mine, s + u, 64) * EC + lane];
            acc += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
        }
        for (; s + 4 <= dd; s += 4) {
            const float v0 = Q[(size_t)__shfl(mine, s + 0, 64) * EC + lane];
            const float v1 = Q[(size_t)__shfl(mine, s + 1, 64) * EC + lane];
            const float v2 = Q[(size_t)__shfl(mine, s + 2, 64) * EC + lane];
            const float v3 = Q[(size_t)__shfl(mine, s + 3, 64) * EC + lane];
            acc += (v0 + v1) + (v2 + v3);
        }
        for (; s < dd; ++s) acc += Q[(size_t)__shfl(mine, s, 64) * EC + lane];
        SQ[(size_t)tm * EC + lane] = acc;
        if (lane == 0) indeg[tm] = (float)d;
    }
}
// in-edges beyond REV_CAP (rare): wave per overflow entry, float atomics
__global__ void __launch_bounds__(256)
rev_overflow_kernel(const float *__restrict__ Q, const int *__restrict__ ovf_cnt, const int2 *__restrict__ ovf,
                    float *__restrict__ SQ)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int n = *ovf_cnt;
    for (int o = blockIdx.x * 4 + wv; o < n; o += gridDim.x * 4) {
        const int2 en = ovf[o];
        atomicAdd(&SQ[(size_t)en.x * EC + lane], Q[(size_t)en.y * EC + lane]);
    }
}

// ---------------------------------------------------------------------------
// The reverse graph in CSR form (N k < 65536 edges per point-cloud block and the tables below within LDS): for every
// target m its in-edges in ASCENDING edge order (edge e = n k + j), so that everything summed over the in-edges of a
// point - SQ here, G_P in the backward - has a FIXED order: bit-reproducible, and no float atomics.
//   rowptr[b N + m]      first row of target m in the global CSR order (rows b N k .. (b+1) N k - 1 belong to block b);
//                        rowptr[B N] = B N k
//   csr_src[row]         global source point (b N + n) of that in-edge
//   pos[(b N + n) k + j] the CSR row of edge (n, j)   - where the backward stores that edge's dy1 row
// A counting sort by (target, source) without a sort.  A cloud's sources are cut into R ranges, one workgroup each (R = 8
// at N = 2048: 8 x B workgroups instead of the B of a one-workgroup-per-cloud form, which left 7/8 of the chip idle for
// ~120 us per layer), a workgroup's 16 waves own 16 consecutive sub-ranges of `span` sources.
//   csr_count_kernel   per (range r, target m) in-edge counts -> rangecnt[b][r][m] (u16), LDS atomics (order-free);
//   csr_place_kernel
//     A  per (wave w, target m) counts of the workgroup's own range, LDS atomics on packed u16 pairs;
//     B  a thread per target: in-edges from the ranges before r (rangecnt) + exclusive prefix over the waves, and the
//        total in-degree; block-wide scan -> ptr (every workgroup of the cloud computes the same ptr);
//     C  wave w walks ITS sources in ascending order, one source (k distinct targets = k lanes, no conflict) per step:
//        row = ptr[m] + cursor[w][m]++ - the wave is the only writer of its cursors, its LDS operations stay in program
//        order, so the ranks follow the source order exactly.  A hub's in-edges are spread over the steps like any others.
// ---------------------------------------------------------------------------
constexpr int CSR_THREADS = 1024, CSR_WAVES = CSR_THREADS / 64, CSR_MAX_RANGES = 8;
static inline size_t csr_lds_bytes(int N, int k) { (void)k; return sizeof(unsigned) * (CSR_WAVES / 2) * (size_t)N + sizeof(int) * (size_t)N + sizeof(int) * 64; }
static inline int csr_ranges(int N) { const int r = (N + 255) / 256; return r < 1 ? 1 : (r > CSR_MAX_RANGES ? CSR_MAX_RANGES : r); }
__host__ __device__ static inline int csr_span(int N, int R) { return (N + R * CSR_WAVES - 1) / (R * CSR_WAVES); }     // sources per wave

__global__ void __launch_bounds__(CSR_THREADS)
csr_count_kernel(const int32_t *__restrict__ idx, int N, int k, unsigned short *__restrict__ rangecnt)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char csr_smem[];
    unsigned *cnt = reinterpret_cast<unsigned *>(csr_smem);                // [N]
    const int tid = threadIdx.x, r = blockIdx.x, R = gridDim.x, b = blockIdx.y;
    const int span = csr_span(N, R);
    const int n_lo = min(N, r * CSR_WAVES * span), n_hi = min(N, n_lo + CSR_WAVES * span);
    const int32_t *ib = idx + (size_t)b * N * k;
    for (int m = tid; m < N; m += CSR_THREADS) cnt[m] = 0u;
    __syncthreads();
    const int e_hi = n_hi * k;
    for (int e0 = n_lo * k + tid; e0 < e_hi; e0 += 8 * CSR_THREADS) {     // 8 loads in flight per thread
        int tg[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int e = e0 + q * CSR_THREADS;
            tg[q] = e < e_hi ? min(max(ib[e], 0), N - 1) : -1;      // (ids outside [0, N) are a caller error: clamped, so that the LDS tables stay intact)
        }
#pragma unroll
        for (int q = 0; q < 8; ++q)
            if (tg[q] >= 0) atomicAdd(&cnt[tg[q]], 1u);
    }
    __syncthreads();
    unsigned short *out = rangecnt + ((size_t)b * R + r) * N;
    for (int m = tid; m < N; m += CSR_THREADS) out[m] = (unsigned short)cnt[m];
}

__global__ void __launch_bounds__(CSR_THREADS)
csr_place_kernel(const int32_t *__restrict__ idx, const unsigned short *__restrict__ rangecnt, int B, int N, int k,
                 int *__restrict__ rowptr, int *__restrict__ csr_src, int *__restrict__ pos)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char csr_smem[];
    unsigned *cw = reinterpret_cast<unsigned *>(csr_smem);                 // [8][N]: waves 2 q (low half) and 2 q + 1 (high half)
    int *ptr = reinterpret_cast<int *>(cw + (CSR_WAVES / 2) * (size_t)N);  // [N] in-degree, then exclusive scan
    int *wsum = ptr + N;                                                   // [64] scan scratch
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, r = blockIdx.x, R = gridDim.x, b = blockIdx.y;
    const int ne = N * k;
    const int span = csr_span(N, R);
    const int r_lo = min(N, r * CSR_WAVES * span), r_hi = min(N, r_lo + CSR_WAVES * span);      // this workgroup's sources
    const int32_t *ib = idx + (size_t)b * ne;
    for (int q = tid; q < (CSR_WAVES / 2) * N; q += CSR_THREADS) cw[q] = 0u;
    __syncthreads();
    // A: the edges of the own range, 8 loads in flight per thread
    {
        const int e_hi = r_hi * k;
        for (int e0 = r_lo * k + tid; e0 < e_hi; e0 += 8 * CSR_THREADS) {
            int tg[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int e = e0 + q * CSR_THREADS;
                tg[q] = e < e_hi ? min(max(ib[e], 0), N - 1) : -1;
            }
#pragma unroll
            for (int q = 0; q < 8; ++q)
                if (tg[q] >= 0) {
                    const int w = ((e0 + q * CSR_THREADS) / k - r_lo) / span;
                    atomicAdd(&cw[(size_t)(w >> 1) * N + tg[q]], (w & 1) ? 0x10000u : 1u);
                }
        }
    }
    __syncthreads();
    // B: per target, the in-edges from earlier ranges + the exclusive prefix over the waves; thread t owns the `per`
    // consecutive targets t per ..
    const int per = (N + CSR_THREADS - 1) / CSR_THREADS;
    const unsigned short *rc = rangecnt + (size_t)b * R * N;
    int loc = 0;
    for (int q = 0; q < per; ++q) {
        const int m = tid * per + q;
        if (m < N) {
            unsigned run = 0, total = 0;
            for (int r2 = 0; r2 < R; ++r2) {
                const unsigned c = rc[(size_t)r2 * N + m];
                run += r2 < r ? c : 0u;
                total += c;
            }
#pragma unroll
            for (int w2 = 0; w2 < CSR_WAVES / 2; ++w2) {
                const unsigned v = cw[(size_t)w2 * N + m];
                const unsigned lo = v & 0xffffu, hi = v >> 16;
                cw[(size_t)w2 * N + m] = run | ((run + lo) << 16);
                run += lo + hi;
            }
            ptr[m] = (int)total;
            loc += (int)total;
        }
    }
    int inc = loc;                                          // inclusive scan inside the wave
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int v = __shfl_up(inc, o, 64); if (lane >= o) inc += v; }
    if (lane == 63) wsum[wv] = inc;
    __syncthreads();
    if (wv == 0) {
        int v = lane < CSR_WAVES ? wsum[lane] : 0, w = v;
#pragma unroll
        for (int o = 1; o < CSR_WAVES; o <<= 1) { const int u = __shfl_up(w, o, 64); if (lane >= o) w += u; }
        if (lane < CSR_WAVES) wsum[lane] = w - v;           // exclusive wave offsets
    }
    __syncthreads();
    int run = wsum[wv] + inc - loc;
    for (int q = 0; q < per; ++q) {
        const int m = tid * per + q;
        if (m < N) {
            const int c = ptr[m];
            ptr[m] = run;
            if (r == 0) rowptr[(size_t)b * N + m] = b * ne + run;
            run += c;
        }
    }
    if (b == B - 1 && r == 0 && tid == 0) rowptr[(size_t)B * N] = B * ne;
    __syncthreads();
    // C: ordered placement
    unsigned *cur = cw + (size_t)(wv >> 1) * N;             // this wave's cursors: the low (even wave) or high half of word m
    const unsigned one = (wv & 1) ? 0x10000u : 1u;
    const int sh = (wv & 1) ? 16 : 0;
    const int n_lo = min(r_hi, r_lo + wv * span), n_hi = min(r_hi, n_lo + span);
    for (int n0 = n_lo; n0 < n_hi; n0 += 8) {               // the neighbour lists of 8 sources in flight
        int tg[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) tg[u] = (lane < k && n0 + u < n_hi) ? min(max(ib[(size_t)(n0 + u) * k + lane], 0), N - 1) : -1;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (tg[u] >= 0) {
                // (an atomic, so that a caller's list with a repeated neighbour still gets distinct rows; one wave's LDS
                // operations on an address are served in issue order, which is the source order)
                const int c = (int)((atomicAdd(&cur[tg[u]], one) >> sh) & 0xffffu);
                const int row = b * ne + ptr[tg[u]] + c;
                csr_src[row] = b * N + n0 + u;
                pos[(size_t)b * ne + (size_t)(n0 + u) * k + lane] = row;
            }
        }
    }
}
// SQ[m] = sum over the in-edges of m, in CSR (= ascending source) order, of Q[source]; in-degree.
// lane = (row slot sub = lane >> 4, channel quad c4 = lane & 15): a wave instruction gathers 4 rows of 256 B as 16-byte
// pieces, 8 of them in flight; a wave owns 8 consecutive targets (one load brings their row pointers).
__global__ void __launch_bounds__(256)
csr_gather_kernel(const float *__restrict__ Q, const int *__restrict__ rowptr, const int *__restrict__ csr_src,
                  int BN, int N, float *__restrict__ SQ, float *__restrict__ indeg)
{
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int sub = lane >> 4, c4 = lane & 15;
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    constexpr int TPW = 8;
    const int nchunk = (BN + TPW - 1) / TPW;
    const UnitDeal deal(nchunk, N / TPW, BN / N, N % TPW == 0, 4, wv);
    for (int w = deal.first; w < deal.count; w += deal.stride) {
        const int t0 = deal.unit(w) * TPW;
        const int myptr = rowptr[min(t0 + min(lane, TPW), BN)];
#pragma unroll 2
        for (int q = 0; q < TPW; ++q) {
            const int tm = t0 + q;
            if (tm >= BN) break;
            const int p0 = __shfl(myptr, q, 64), p1 = __shfl(myptr, q + 1, 64);
            f32x4 acc = zero;
            for (int p = p0 + sub; p < p1; p += 32) {
                int src[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) src[u] = (p + 4 * u < p1) ? csr_src[p + 4 * u] : -1;
                f32x4 v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = src[u] >= 0 ? *reinterpret_cast<const f32x4 *>(Q + (size_t)src[u] * EC + 4 * c4) : zero;
                acc += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float t = acc[e];
                t += __shfl_xor(t, 16, 64);
                t += __shfl_xor(t, 32, 64);
                acc[e] = t;
            }
            if (sub == 0) *reinterpret_cast<f32x4 *>(SQ + (size_t)tm * EC + 4 * c4) = acc;
            if (lane == 0) indeg[tm] = (float)(p1 - p0);
        }
    }
}

// ---------------------------------------------------------------------------
// F3: BN statistics of channel c -> (mean, rstd, scale, shift); `writer` also stores them as stats rows
// row0..row0+3 and updates the running statistics (momentum, unbiased variance) when training.
// Called from the prologue of the kernel that consumes the constants (every block recomputes its 64 channels
// from the 128 fp64 sums - a separate 5 us launch per BatchNorm otherwise).
// ---------------------------------------------------------------------------
struct BnFin {
    const double *acc;            // [2][64] sums (training) or NULL (constants are already in `stats`)
    double count;
    const float *gamma, *beta;
    float *rm, *rv;
    float eps, momentum;
    int training;
};
__device__ __forceinline__ void bn_constants(const BnFin &f, int c, bool writer, float *__restrict__ stats, int row0,
                                             float &scale, float &shift)
{
    if (!f.acc && f.training) {                          // already finalised (backward pass)
        scale = stats[(row0 + 2) * EC + c];
        shift = stats[(row0 + 3) * EC + c];
        return;
    }
    float mean, var;
    if (f.training) {
        const double mu = f.acc[c] / f.count;
        double v = f.acc[EC + c] / f.count - mu * mu;
        if (v < 0.0) v = 0.0;
        mean = (float)mu;
        var = (float)v;
        if (writer && f.rm) {
            const double unbiased = (f.count > 1.0) ? v * f.count / (f.count - 1.0) : v;
            f.rm[c] = (1.f - f.momentum) * f.rm[c] + f.momentum * mean;
            f.rv[c] = (1.f - f.momentum) * f.rv[c] + f.momentum * (float)unbiased;
        }
    } else {
        mean = f.rm[c];
        var = f.rv[c];
    }
    const float rstd = 1.0f / sqrtf(var + f.eps);
    scale = f.gamma[c] * rstd;
    shift = f.beta[c] - mean * scale;
    if (writer) {
        stats[(row0 + 0) * EC + c] = mean;
        stats[(row0 + 1) * EC + c] = rstd;
        stats[(row0 + 2) * EC + c] = scale;
        stats[(row0 + 3) * EC + c] = shift;
    }
}

// ---------------------------------------------------------------------------
// F1: P = Wa x, Q = (Wb - Wa) x            grid: (ceil(N/64), B), block 256
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
ec_pq_kernel(const float *__restrict__ x, const float *__restrict__ w1, int C, int N,
             float *__restrict__ P, float *__restrict__ Q, const BnFin fin, float *__restrict__ stats, int fold,
             double *__restrict__ zero_acc)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    // (training: the 4 x 64 BatchNorm accumulators of the passes that follow on this stream start at zero - a memset launch less)
    if (zero_acc && blockIdx.x == 0 && blockIdx.y == 0) zero_acc[threadIdx.x] = 0.0;
    float *wa = smem;                 // [C][64]
    float *wd = wa + C * EC;          // [C][64]
    float *xs = wd + C * EC;          // [C][64 points]
    const int tid = threadIdx.x, b = blockIdx.y, n0 = blockIdx.x * 64;
    for (int e = tid; e < C * EC; e += 256) {
        const int c = e >> 6, co = e & 63;
        const float a = w1[(size_t)co * 2 * C + c];
        const float bb = w1[(size_t)co * 2 * C + C + c];
        wa[e] = a;
        wd[e] = bb - a;
    }
    const float *xb = x + (size_t)b * C * N;
    for (int e = tid; e < C * 64; e += 256) {
        const int c = e >> 6, p = e & 63;
        xs[e] = (n0 + p < N) ? xb[(size_t)c * N + n0 + p] : 0.f;
    }
    __syncthreads();
    const int co = tid & 63, wv = tid >> 6;
    float sc = 1.f, sh = 0.f;                          // eval mode: the BatchNorm1 fold of ec_prescale_kernel happens here
    if (fold) bn_constants(fin, co, blockIdx.x == 0 && blockIdx.y == 0 && wv == 0, stats, (int)ST_MEAN1, sc, sh);
    for (int p = wv; p < 64; p += 4) {
        if (n0 + p >= N) break;
        float ap = 0.f, aq = 0.f;
        for (int c = 0; c < C; ++c) {
            const float xv = xs[c * 64 + p];
            ap = fmaf(wa[c * EC + co], xv, ap);
            aq = fmaf(wd[c * EC + co], xv, aq);
        }
        const size_t o = ((size_t)b * N + n0 + p) * EC + co;
        P[o] = fold ? ap * sc : ap;
        Q[o] = fold ? fmaf(aq, sc, sh) : aq;
    }
}

// F1 on MFMA (C % 32 == 0, point-major x available): rows 0..63 of wpq = Wa, rows 64..127 = Wb - Wa, so that
// [P | Q](M, 128) = x_pm(M, C) . wpq^T is one plain GEMM (pointwise.hip).  In eval mode the BatchNorm1 fold of
// ec_prescale_kernel rides in that GEMM's epilogue instead (P' = P scale1, Q' = Q scale1 + shift1: `eab` = 128 column
// scales, then 128 column shifts) and the prescale pass disappears.
__global__ void ec_wpq_kernel(const float *__restrict__ w1, int C, float *__restrict__ wpq, const BnFin fin, float *__restrict__ stats,
                              float *__restrict__ eab, double *__restrict__ zero_acc)
{
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (zero_acc && blockIdx.x == 0) zero_acc[threadIdx.x] = 0.0;      // (as in ec_pq_kernel: 256 threads, 4 x 64 accumulators)
    if (eab && e < EC) {
        float sc, sh;
        bn_constants(fin, e, true, stats, (int)ST_MEAN1, sc, sh);
        eab[e] = sc; eab[EC + e] = sc;
        eab[2 * EC + e] = 0.f; eab[3 * EC + e] = sh;
    }
    if (e >= EC * C) return;
    const int co = e / C, c = e - co * C;
    const float a = w1[(size_t)co * 2 * C + c];
    wpq[e] = a;
    wpq[EC * C + e] = w1[(size_t)co * 2 * C + C + c] - a;
}

// ---------------------------------------------------------------------------
// F3b: fold BN1 into the projections: P' = P*scale1, Q' = Q*scale1 + shift1, so that the
// heavy kernels form a1 = LReLU(P'[m] + Q'[n]) with one add per element and no per-channel
// constants in registers.
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
ec_prescale_kernel(const float *__restrict__ P, const float *__restrict__ Q, const BnFin fin, float *__restrict__ stats,
                   size_t total, float *__restrict__ Pp, float *__restrict__ Qp)
{
    // few, long-lived workgroups (every one pays the fp64 finalisation of the 64 channels once), 16 bytes per lane and step:
    // one workgroup per 256 elements spent most of its time in that prologue (32 -> 14 us at 32 x 2048 points)
    typedef float f32x4_ __attribute__((ext_vector_type(4)));
    __shared__ float sc[EC], sh[EC];
    if (threadIdx.x < EC) bn_constants(fin, threadIdx.x, blockIdx.x == 0, stats, (int)ST_MEAN1, sc[threadIdx.x], sh[threadIdx.x]);
    __syncthreads();
    const int c = (threadIdx.x * 4) & 63;                 // the stride 256 x 4 x gridDim keeps a thread on its 4 channels
    const f32x4_ s4 = {sc[c], sc[c + 1], sc[c + 2], sc[c + 3]}, h4 = {sh[c], sh[c + 1], sh[c + 2], sh[c + 3]};
    const size_t nq = total >> 2;                         // total = B N 64
    for (size_t q = (size_t)blockIdx.x * 256 + threadIdx.x; q < nq; q += (size_t)gridDim.x * 256) {
        const f32x4_ pv = reinterpret_cast<const f32x4_ *>(P)[q], qv = reinterpret_cast<const f32x4_ *>(Q)[q];
        f32x4_ po, qo;
#pragma unroll
        for (int e = 0; e < 4; ++e) { po[e] = pv[e] * s4[e]; qo[e] = fmaf(qv[e], s4[e], h4[e]); }
        reinterpret_cast<f32x4_ *>(Pp)[q] = po;
        reinterpret_cast<f32x4_ *>(Qp)[q] = qo;
    }
}

// ---------------------------------------------------------------------------
// F4: the heavy forward pass on the bf16 matrix cores at fp32 accuracy (bf16x6: every fp32 operand is the exact
// sum of three bf16 pieces, six piece products per fp32 product - pointwise.hip has the derivation).
//
//   H2[edge][c_out] = sum_c a1[edge][c] * W2[c_out][c]      A = a1 rows, B = W2^T
//
// A wave walks "super-tiles" of 8 points.  Row i of the 32-row MFMA tile t is the edge
//   point p = 4*((i>>2)&1) + (i>>3),  neighbour slot j = 4*t + (i&3)
// so that in the C/D layout (lane = column c_out, register r <-> row (r&3)+8(r>>2)+4h) lane
// half h owns points 4h..4h+3 and register group q = r>>2 is ONE point: the max / min / arg
// over the k neighbours and the BN2 statistics are in-lane reductions over accumulator
// registers - no LDS, no shuffles.  k <= 32 (8 tiles); slots j >= k are masked.
// A-operand: lane (i, h) gathers the 128-B half row P'[m][32h..32h+31] of ITS edge (8 x
// dwordx4 from the L2-resident projection), adds Q'[n], applies LeakyReLU and splits the 32 values into
// bf16 pieces (4 VALU each + packing).  B-operand: W2 pre-split in LDS, row = [hi 64 x bf16 | mid | lo | pad]
// = 400 B = 25 sixteen-byte slots (odd: conflict-free ds_read_b128).  K-block kb of v_mfma_f32_32x32x16_bf16
// holds channels 8kb..8kb+7 (lane half 0) and 32+8kb.. (lane half 1) for both operands.
// 48 MFMAs of 32 cycles per 32-edge tile instead of 64 of 64 cycles.
// ---------------------------------------------------------------------------
// PREC (common.h): PREC_X6 as described; PREC_BF16 / PREC_F16 = one rounded 16-bit product per fp32 product
// (8 MFMAs per 32-edge tile, one v_cvt per two operand values instead of the 3-way split).
template <bool STATS, int PREC>
__global__ void __launch_bounds__(256, STATS ? 2 : 3)      // the eval form fits three workgroups per CU (168 VGPRs): 1.06 -> 0.99 ms eval forward;
                                                           // forcing the training form there spills and was measured slower (0.251 -> 0.273 ms per step)
ec_main_fwd_kernel(const float *__restrict__ Pp, const float *__restrict__ Qp, const int32_t *__restrict__ idx,
                   const float *__restrict__ w2, const float *__restrict__ gamma2,
                   int BN, int N, int k, float slope,
                   float *__restrict__ hsel, uint8_t *__restrict__ jsel,
                   const BnAcc partial,
                   const BnFin fin1, float *__restrict__ stats, float *__restrict__ Pp_out, float *__restrict__ Qp_out)
{
    // FOLD (training form, Pp_out != NULL): `Pp` / `Qp` are the RAW projections P, Q and BatchNorm1 is applied here -
    //   Q'[n] = scale1 Q[n] + shift1 once per point, z = fma(scale1, P[m], Q'[n]) per edge (the fma replaces the add: no extra
    //   vector instruction per edge, 32 more registers for this lane's scales) - and the folded rows P'[n], Q'[n] of the
    //   super-tile's own 8 points are stored for the backward.  What ec_prescale_kernel did in a 92 MB pass of its own now rides
    //   inside a kernel the matrix pipe bounds; the fp64 finalisation of the batch statistics happens in this prologue.
    constexpr bool FOLD = STATS;
    constexpr int WROW = 3 * 2 * EC + 16;               // bytes per W2 row image
    __shared__ __attribute__((aligned(16))) unsigned char sW3[EC * WROW];
    __shared__ double red[4][2][2][EC];                 // [wave][half][s|ss][64]
    __shared__ __attribute__((aligned(16))) float sc1[EC], sh1[EC];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int i = lane & 31, h = lane >> 5;
    if (FOLD && Pp_out && tid < EC) bn_constants(fin1, tid, blockIdx.x == 0, stats, (int)ST_MEAN1, sc1[tid], sh1[tid]);
    // max_j LReLU(BN2(h_j)) is reached at max_j h_j where scale2 >= 0 and at min_j h_j elsewhere, and sign(scale2) =
    // sign(gamma2) whatever the batch statistics turn out to be: the rows of W2 with a negative gamma go into the image
    // NEGATED, the kernel tracks one maximum (of +-h, exactly the negation) and undoes the sign where it stores.
    for (int e = tid; e < EC * EC; e += 256)
        mm_store1<PREC>(gamma2[e >> 6] < 0.f ? -w2[e] : w2[e], reinterpret_cast<unsigned short *>(sW3 + (e >> 6) * WROW) + (e & 63), EC);
    __syncthreads();

    const int pA = 4 * ((i >> 2) & 1) + (i >> 3);       // point (0..7) of this lane's A rows
    const int jjA = i & 3;
    const int T = (k + 3) >> 2;
    double sd0 = 0.0, ssd0 = 0.0, sd1 = 0.0, ssd1 = 0.0;
    const unsigned char *wrow0 = sW3 + i * WROW + 64 * h;            // + piece * 128 + kb * 16
    const unsigned char *wrow1 = sW3 + (32 + i) * WROW + 64 * h;
    const float sg0 = gamma2[i] < 0.f ? -1.f : 1.f, sg1 = gamma2[32 + i] < 0.f ? -1.f : 1.f;      // C/D layout: lane = channel
    const bool fold = FOLD && Pp_out != nullptr;          // (kernel-uniform)
    float sA[FOLD ? 32 : 1];                              // scale1 of this lane's 32 channels (fold only)
    if constexpr (FOLD) {                                 // (sc1 / sh1 were written before the barrier that closes the W2 staging)
#pragma unroll
        for (int c = 0; c < 32; ++c) sA[c] = fold ? sc1[32 * h + c] : 1.f;
    }

    const int nst = (BN + 7) >> 3;
    const UnitDeal deal(nst, N >> 3, BN / N, (N & 7) == 0, 4, wv);
    for (int w = deal.first; w < deal.count; w += deal.stride) {
        const int st = deal.unit(w);
        const int n0 = st * 8;
        const int ptA = min(n0 + pA, BN - 1);
        const int bA = ptA / N;
        const int32_t *myidx = idx + (size_t)ptA * k;
        const float *prow_base = Pp + (size_t)bA * N * EC + 32 * h;
        float qA[32];
        {
            const float4 *q4 = reinterpret_cast<const float4 *>(Qp + (size_t)ptA * EC + 32 * h);
#pragma unroll
            for (int v = 0; v < 8; ++v) {
                const float4 t4 = q4[v];
                qA[4 * v + 0] = t4.x; qA[4 * v + 1] = t4.y; qA[4 * v + 2] = t4.z; qA[4 * v + 3] = t4.w;
            }
        }
        if constexpr (FOLD) {
            if (fold) {
                // Q' of this lane's point, kept in registers and (one lane of the four that hold the same half row) stored for
                // the backward; another of the four folds and stores the point's own P half row
#pragma unroll
                for (int v = 0; v < 8; ++v) {
                    const float4 t4 = *reinterpret_cast<const float4 *>(sh1 + 32 * h + 4 * v);
                    qA[4 * v + 0] = fmaf(qA[4 * v + 0], sA[4 * v + 0], t4.x);
                    qA[4 * v + 1] = fmaf(qA[4 * v + 1], sA[4 * v + 1], t4.y);
                    qA[4 * v + 2] = fmaf(qA[4 * v + 2], sA[4 * v + 2], t4.z);
                    qA[4 * v + 3] = fmaf(qA[4 * v + 3], sA[4 * v + 3], t4.w);
                }
                if (n0 + pA < BN) {
                    if (jjA == 0) {
                        float4 *qo = reinterpret_cast<float4 *>(Qp_out + (size_t)ptA * EC + 32 * h);
#pragma unroll
                        for (int v = 0; v < 8; ++v) qo[v] = make_float4(qA[4 * v + 0], qA[4 * v + 1], qA[4 * v + 2], qA[4 * v + 3]);
                    } else if (jjA == 1) {
                        const float4 *pi = reinterpret_cast<const float4 *>(Pp + (size_t)ptA * EC + 32 * h);
                        float4 *po = reinterpret_cast<float4 *>(Pp_out + (size_t)ptA * EC + 32 * h);
#pragma unroll
                        for (int v = 0; v < 8; ++v) {
                            const float4 t4 = pi[v];
                            po[v] = make_float4(t4.x * sA[4 * v + 0], t4.y * sA[4 * v + 1], t4.z * sA[4 * v + 2], t4.w * sA[4 * v + 3]);
                        }
                    }
                }
            }
        }
        float mx0[4], mx1[4];
        int jx0[4], jx1[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            mx0[q] = mx1[q] = -__builtin_inff();
            jx0[q] = jx1[q] = 0;
        }
        float s0 = 0.f, ss0 = 0.f, s1 = 0.f, ss1 = 0.f;
        const bool full = (k & 3) == 0 && n0 + 8 <= BN;        // wave-uniform
        // software pipeline: rows of tile t+1 and the neighbour id of tile t+2 are in flight while
        // tile t runs on the matrix pipe (no dependent idx -> row load chain inside the loop)
        float4 pf[8];
        int m_next = myidx[min(4 + jjA, k - 1)];
        {
            const float4 *p4 = reinterpret_cast<const float4 *>(prow_base + (size_t)myidx[min(jjA, k - 1)] * EC);
#pragma unroll
            for (int v = 0; v < 8; ++v) pf[v] = p4[v];
        }
        for (int t = 0; t < T; ++t) {
            const float4 *p4n = reinterpret_cast<const float4 *>(prow_base + (size_t)m_next * EC);
            f32x16 acc0, acc1;
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) {
                // a1 for channels 8kb..8kb+7 of this lane's half row, then refill those two pieces with the next
                // tile's row (harmless re-read of the last row after the last tile): the loads overlap the MFMAs
                float av[8];
                {
                    const float4 x0 = pf[2 * kb], x1 = pf[2 * kb + 1];
                    float z;
                    if constexpr (FOLD) {                                            // (scale 1 when the rows come in folded)
                        z = fmaf(sA[8 * kb + 0], x0.x, qA[8 * kb + 0]); av[0] = fmaxf(z, z * slope);          // LeakyReLU, 0 < slope < 1
                        z = fmaf(sA[8 * kb + 1], x0.y, qA[8 * kb + 1]); av[1] = fmaxf(z, z * slope);
                        z = fmaf(sA[8 * kb + 2], x0.z, qA[8 * kb + 2]); av[2] = fmaxf(z, z * slope);
                        z = fmaf(sA[8 * kb + 3], x0.w, qA[8 * kb + 3]); av[3] = fmaxf(z, z * slope);
                        z = fmaf(sA[8 * kb + 4], x1.x, qA[8 * kb + 4]); av[4] = fmaxf(z, z * slope);
                        z = fmaf(sA[8 * kb + 5], x1.y, qA[8 * kb + 5]); av[5] = fmaxf(z, z * slope);
                        z = fmaf(sA[8 * kb + 6], x1.z, qA[8 * kb + 6]); av[6] = fmaxf(z, z * slope);
                        z = fmaf(sA[8 * kb + 7], x1.w, qA[8 * kb + 7]); av[7] = fmaxf(z, z * slope);
                    } else {
                    z = x0.x + qA[8 * kb + 0]; av[0] = fmaxf(z, z * slope);
                    z = x0.y + qA[8 * kb + 1]; av[1] = fmaxf(z, z * slope);
                    z = x0.z + qA[8 * kb + 2]; av[2] = fmaxf(z, z * slope);
                    z = x0.w + qA[8 * kb + 3]; av[3] = fmaxf(z, z * slope);
                    z = x1.x + qA[8 * kb + 4]; av[4] = fmaxf(z, z * slope);
                    z = x1.y + qA[8 * kb + 5]; av[5] = fmaxf(z, z * slope);
                    z = x1.z + qA[8 * kb + 6]; av[6] = fmaxf(z, z * slope);
                    z = x1.w + qA[8 * kb + 7]; av[7] = fmaxf(z, z * slope);
                    }
                }
                pf[2 * kb] = p4n[2 * kb];
                pf[2 * kb + 1] = p4n[2 * kb + 1];
                constexpr int NP = Mx<PREC>::NP;
                uint4 ap[NP];                            // 16-bit operand pieces, two per dword
                mm_pieces8<PREC>(av, ap);
                uint4 b0[NP], b1[NP];
#pragma unroll
                for (int pc = 0; pc < NP; ++pc) {
                    b0[pc] = *reinterpret_cast<const uint4 *>(wrow0 + pc * 128 + kb * 16);
                    b1[pc] = *reinterpret_cast<const uint4 *>(wrow1 + pc * 128 + kb * 16);
                }
                if constexpr (PREC == PREC_X6) {         // interleaved: the two accumulator chains alternate
                    constexpr int PA[6] = GFS3D_X6_PA, PB[6] = GFS3D_X6_PB;
#pragma unroll
                    for (int term = 0; term < 6; ++term) {
                        acc0 = mm_mfma<PREC>(ap[PA[term]], b0[PB[term]], acc0);
                        acc1 = mm_mfma<PREC>(ap[PA[term]], b1[PB[term]], acc1);
                    }
                } else {
                    mm_mac<PREC>(ap, b0, acc0);
                    mm_mac<PREC>(ap, b1, acc1);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            m_next = myidx[min(4 * (t + 2) + jjA, k - 1)];
            if (full) {
                // every row of the tile is a real edge (k a multiple of 4, all 8 points inside the batch - all tiles but those
                // of the last super-tile when k = 20): no masks; the eval form needs no winner slot either (nothing is
                // differentiated there), so its running maximum is one v_max per element
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int q = r >> 2;
                    const float v0 = acc0[r], v1 = acc1[r];
                    if (STATS) {
                        const int j = 4 * t + (r & 3);
                        const bool gx0 = v0 > mx0[q], gx1 = v1 > mx1[q];
                        mx0[q] = gx0 ? v0 : mx0[q]; jx0[q] = gx0 ? j : jx0[q];
                        mx1[q] = gx1 ? v1 : mx1[q]; jx1[q] = gx1 ? j : jx1[q];
                        s0 += v0; ss0 = fmaf(v0, v0, ss0);
                        s1 += v1; ss1 = fmaf(v1, v1, ss1);
                    } else {
                        mx0[q] = fmaxf(mx0[q], v0);
                        mx1[q] = fmaxf(mx1[q], v1);
                    }
                }
            } else {
                // masked rows look like -inf / +inf / 0 to the reductions
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int q = r >> 2;
                    const int j = 4 * t + (r & 3);
                    const bool ok = (j < k) && (n0 + 4 * h + q < BN);
                    const float v0 = acc0[r], v1 = acc1[r];
                    const bool gx0 = ok && v0 > mx0[q];
                    const bool gx1 = ok && v1 > mx1[q];
                    mx0[q] = gx0 ? v0 : mx0[q]; jx0[q] = gx0 ? j : jx0[q];
                    mx1[q] = gx1 ? v1 : mx1[q]; jx1[q] = gx1 ? j : jx1[q];
                    if (STATS) {
                        const float u0 = ok ? v0 : 0.f, u1 = ok ? v1 : 0.f;
                        s0 += u0; ss0 = fmaf(u0, u0, ss0);
                        s1 += u1; ss1 = fmaf(u1, u1, ss1);
                    }
                }
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int pt = n0 + 4 * h + q;
            if (pt < BN) {
                const size_t o = (size_t)pt * EC + i;
                hsel[o] = sg0 * mx0[q]; hsel[o + 32] = sg1 * mx1[q];          // the true pre-BN2 value
                jsel[o] = (uint8_t)jx0[q]; jsel[o + 32] = (uint8_t)jx1[q];
            }
        }
        sd0 += (double)s0; ssd0 += (double)ss0;
        sd1 += (double)s1; ssd1 += (double)ss1;
    }
    if (STATS) {
        red[wv][h][0][i] = sg0 * sd0;  red[wv][h][0][i + 32] = sg1 * sd1;       // sums of the true h (squares need no sign)
        red[wv][h][1][i] = ssd0; red[wv][h][1][i + 32] = ssd1;
        __syncthreads();
        if (tid < 128) {
            const int which = tid >> 6, c = tid & 63;
            double tsum = 0.0;
#pragma unroll
            for (int w = 0; w < 4; ++w) tsum += red[w][0][which][c] + red[w][1][which][c];
            bn_acc_add(partial, blockIdx.x, which, c, tsum);
        }
    }
}

// ---------------------------------------------------------------------------
// F5: epilogue: pick max or min by sign(scale2), BN2 + LReLU, write y in both
// layouts, compact (hsel, jsel).  block = 256 threads = tile of 64 points.
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
ec_epilogue_kernel(const float *__restrict__ hwin, const uint8_t *__restrict__ jwin,
                   const BnFin fin, float *__restrict__ stats, int N, float slope,
                   float *__restrict__ y_cf, long long y_bstride, float *__restrict__ y_pm,
                   float *__restrict__ hsel, uint8_t *__restrict__ jsel, float *__restrict__ y_sq, unsigned *__restrict__ y_img, int B)
{
    __shared__ float tile[64][65];
    __shared__ float scs[EC], shs[EC];
    const int tid = threadIdx.x, b = blockIdx.y, n0 = blockIdx.x * 64;
    const int co = tid & 63, wv = tid >> 6;
    if (tid < EC) bn_constants(fin, tid, blockIdx.x == 0 && blockIdx.y == 0, stats, (int)ST_MEAN2, scs[tid], shs[tid]);
    __syncthreads();
    const float sc = scs[co], sh = shs[co];
    for (int p = wv; p < 64; p += 4) {
        const int n = n0 + p;
        float y = 0.f;
        if (n < N) {
            const size_t o = ((size_t)b * N + n) * EC + co;
            const float h = hwin[o];                   // the pre-BN2 value whose image is the maximum (ec_main_fwd_kernel)
            const uint8_t j = jwin[o];
            y = lrelu(fmaf(h, sc, sh), slope);
            if (hsel) { hsel[o] = h; jsel[o] = j; }
            if (y_pm) y_pm[o] = y;
        }
        tile[p][co] = y;
    }
    __syncthreads();
    if (y_sq && tid < 64 && n0 + tid < N) {            // |y_n|^2 for the next layer's kNN, in its kernel's order (knn.hip: knn_sqnorm_kernel)
        float sq = 0.f;
#pragma unroll 8
        for (int c = 0; c < EC; ++c) sq = __fadd_rn(sq, __fmul_rn(tile[tid][c], tile[tid][c]));
        y_sq[(size_t)b * N + n0 + tid] = sq;
    }
    if (y_cf) {
        const int p = tid & 63;
        for (int c = wv; c < EC; c += 4) {
            if (n0 + p < N) y_cf[(size_t)b * y_bstride + (size_t)c * N + n0 + p] = tile[p][c];
        }
    }
    if (y_img) {       // the next block's screened kNN reads these rows ([hi 64 x bf16 | mid 64 x bf16], knn.hip: knn_img_kernel)
        if (co < 32)
            for (int q = wv; q < 64; q += 4)
                if (n0 + q < N) {
                    unsigned hh, mm;
                    mm_split2_rne(tile[q][2 * co], tile[q][2 * co + 1], hh, mm);
                    unsigned *row = y_img + ((size_t)b * N + n0 + q) * EC;
                    row[co] = hh;
                    row[32 + co] = mm;
                }
        if (blockIdx.x == 0 && blockIdx.y == 0)
            for (int e = tid; e < GFS3D_KNN_IMG_SLACK_ROWS * EC; e += 256) y_img[(size_t)B * N * EC + e] = 0u;
    }
}

// ===========================================================================
// backward
// ===========================================================================
constexpr int PREP_TILES = 4;
// B1: g = dy * lrelu'(y) in point-major layout; partial sums of g and g*xhat2.
__global__ void __launch_bounds__(256)
ec_bwd_prep_kernel(const float *__restrict__ dy_cf, long long dy_bstride, const float *__restrict__ dy_pm,
                   const float *__restrict__ y_pm,
                   const float *__restrict__ hsel, const float *__restrict__ stats, int N, float slope,
                   float *__restrict__ g_pm, const BnAcc partial)
{
    __shared__ float tile[64][65];
    __shared__ double red[4][2][EC];
    const int tid = threadIdx.x, b = blockIdx.y;
    const int co = tid & 63, wv = tid >> 6;
    const float mean2 = stats[ST_MEAN2 * EC + co], rstd2 = stats[ST_RSTD2 * EC + co];
    double s = 0.0, sx = 0.0;
    // a block walks PREP_TILES tiles of 64 points (its 128 fp64 atomics at the end are same-address traffic)
    for (int tt = 0; tt < PREP_TILES; ++tt) {
    const int n0 = (blockIdx.x * PREP_TILES + tt) * 64;
    if (n0 >= N) break;
    __syncthreads();
    // all 16 loads of a thread are issued before the first use (the loops are fully unrolled): the kernel is pure
    // streaming and lives on memory-level parallelism
    float ry[16], rh[16], rd[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const int n = min(n0 + wv + 4 * j, N - 1);
        const size_t o = ((size_t)b * N + n) * EC + co;
        ry[j] = y_pm[o];
        rh[j] = hsel[o];
        rd[j] = dy_pm ? dy_pm[o] : 0.f;                                     // both consumers of the block output
    }
    {
        const int p = tid & 63;
        float rt[16];
#pragma unroll
        for (int j = 0; j < 16; ++j)
            rt[j] = (dy_cf && n0 + p < N) ? dy_cf[(size_t)b * dy_bstride + (size_t)(wv + 4 * j) * N + n0 + p] : 0.f;
#pragma unroll
        for (int j = 0; j < 16; ++j) tile[p][wv + 4 * j] = rt[j];
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const int p = wv + 4 * j, n = n0 + p;
        if (n < N) {
            const size_t o = ((size_t)b * N + n) * EC + co;
            const float g = (tile[p][co] + rd[j]) * (ry[j] > 0.f ? 1.f : slope);
            g_pm[o] = g;
            s += (double)g;
            sx += (double)(g * ((rh[j] - mean2) * rstd2));
        }
    }
    }
    red[wv][0][co] = s;
    red[wv][1][co] = sx;
    __syncthreads();
    if (tid < 128) {
        const int which = tid >> 6;
        const double t = red[0][which][co] + red[1][which][co] + red[2][which][co] + red[3][which][co];
        bn_acc_add(partial, blockIdx.y * gridDim.x + blockIdx.x, which, co, t);
    }
}

// coefficient block (floats): A[64] Bc[64] Cc[64]
enum { CF_A = 0, CF_BC = 64, CF_CC = 128, CF_TOTAL = 192 };

// B2: finalize dgamma2 / dbeta2 and the per-channel constants of the BN2 backward:
//     dh2 = A*route + Bc + Cc*h2   with  A = gamma2*rstd2 (= scale2),
//     Bc = -A (m1 - mean2 rstd2 m2),  Cc = -A rstd2 m2,  m1 = mean(dy2), m2 = mean(dy2 xhat2).
struct CoefIn {
    const double *red, *red_local;    // [2][64] sums over the whole (possibly multi-rank) batch / over this rank
    double count;
    const float *stats;
    float *dbn2_w, *dbn2_b;
};
// per-channel constants of channel c into sC[192]; `writer` also stores this rank's dgamma2 / dbeta2
__device__ __forceinline__ void bwd_coef(const CoefIn &f, int c, bool writer, float *sC)
{
    const double s = f.red[c], sx = f.red[EC + c];
    if (writer) {
        f.dbn2_b[c] = (float)f.red_local[c];
        f.dbn2_w[c] = (float)f.red_local[EC + c];
    }
    const double m1 = s / f.count, m2 = sx / f.count;
    const double A = f.stats[ST_SCALE2 * EC + c];
    const double mean2 = f.stats[ST_MEAN2 * EC + c], rstd2 = f.stats[ST_RSTD2 * EC + c];
    sC[CF_A + c] = (float)A;
    sC[CF_BC + c] = (float)(-A * (m1 - mean2 * rstd2 * m2));
    sC[CF_CC + c] = (float)(-A * rstd2 * m2);
}

// ---------------------------------------------------------------------------
// B3: heavy backward pass on the bf16 matrix cores at fp32 accuracy (bf16x6 as in ec_main_fwd_kernel; direct form,
// three matrix products per 32-edge tile, 3 x 48 MFMAs of 32 cycles; same super-tile row mapping as the forward):
//   (1) H2  = A1 W2^T                       A-layout a1 (lane = edge) x W2 from LDS
//   (2) dW2 += dH2^T A1                     both operands in C/D layout (lane = channel,
//                                           register = edge): the accumulator of (1), turned
//                                           into dH2 in place, IS the A operand
//   (3) dA1 = dH2 W2                        dH2 transposed to A-layout through LDS
// dH2 = Cc*H2 + Bc + [slot == winner] A*g is elementwise in the C/D layout (the routed term is
// one compare per register); a1 in C/D layout comes from the same LDS transpose buffer (no
// second gather); dy1 = dA1 * LReLU'(z1) (sign of a1); G_Q in-lane, G_P as 2x128-B atomic rows.
// LDS: bf16 piece images of W2 (rows co, for (1)) and of W2^T (rows ci, for (3)), 25.6 KB each, shared by the 4 waves
// of the block; per wave one [32][68] transpose tile, the neighbour ids of all tiles of the super-tile [8][32] and
// the Q' rows of its 8 points [8][64].
// Memory pipeline: the kernel sits at the VGPR cap, so the gathered P' half rows of tile t+1 arrive by LDS-DMA
// (global_load_lds, per-lane source row, no VGPR destination) into the transpose tile's bytes, which are idle
// from the last dH2 read of tile t to the a1 write of tile t+1.  The DMA is issued BEFORE tile t's 32 G_P atomic
// instructions: vmcnt retires in issue order, so `s_waitcnt vmcnt(32)` at the top of tile t+1 covers the gather
// without waiting for the atomics' round trips (plain loads at the tile top would wait for every older atomic).
// Occupancy: 256 VGPRs + 96 AGPRs (64 of them the dW2 accumulators, live across all tiles), so ONE wave per SIMD.
// Measured alternative: 8 waves per block (256 registers each, no spill in the tile loop) with dW2 summed per tile
// through ds_add_f32 into a block-wide [64][64] image - 0.37 -> 1.02 ms per launch, the LDS float adds are that slow.
// ---------------------------------------------------------------------------
typedef __attribute__((address_space(3))) char lds_char;
typedef short s16x4_ __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) s16x4_ lds_s16x4;
// 16 B per lane, global -> LDS without a VGPR destination: lane l lands at lds_base + 16 l.  Issued as inline asm
// on purpose: the compiler's own bookkeeping for the LDS-DMA builtin puts an `s_waitcnt vmcnt(0)` in front of the
// next LDS read of ANY address (it cannot tell the regions apart), which would expose the whole gather latency
// right after it is issued.  The waits are the explicit vmcnt(N) at the top of each tile.  (M0 is a reserved
// register: the compiler re-materialises it before every use of its own.)
__device__ __forceinline__ void lds_dma16(const void *gsrc, const void *lds_dst_wave_uniform)
{
    const unsigned base = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds_char *)lds_dst_wave_uniform);
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(base), "v"(gsrc) : "memory");
}
constexpr int BWD_DIMG = 3 * EC * 64;                                       // bytes: bf16 piece planes of the dH2 tile, [plane][co][32 edges], swizzled
constexpr int BWD_WAVE_LDS = 32 * (EC + 4) + 8 * 32 + 8 * EC + 8 * 32 + BWD_DIMG / 4;      // floats per wave: transpose tile | ids | Q' tile | CSR rows | dH2 pieces
constexpr int BWD_PART = EC * EC;
constexpr int TLD = EC + 4;
__device__ __forceinline__ int drow(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

#ifndef GFS3D_ABL          // ablation builds only (scripts/build_ablation.sh): 1 no G_P atomics, 2 gather one fixed row, 4 no dW2 product, 8 no a1 split for the dW2 product, 16 no dy1 row stores, 32 / 64 no MFMAs in GEMM 3 / GEMM 1
#define GFS3D_ABL 0
#endif
constexpr int BWD_WAVES = 4;
constexpr int BWD_WROW = 3 * 2 * EC + 16;                          // bytes per row of a W2 piece image
constexpr size_t BWD_LDS_BYTES = 2 * EC * BWD_WROW + sizeof(float) * (BWD_WAVES * BWD_WAVE_LDS + 192);

// CSR = true: the dy1 row of every edge is STORED to row pos[edge] of `dyb` (the edge's place in its target's CSR
// segment; masked rows go to a per-wave scratch row behind the B N k real ones) and ec_bwd_segsum_kernel sums the
// segments: plain stores at ~5x the float-atomic rate, a fixed summation order, no memset of G_P.  CSR = false: float
// atomics into G_P (shapes beyond csr_place_kernel's LDS tables).
// Geometry of ec_main_bwd_kernel per arithmetic.  The fp32-accurate form: 3 piece planes everywhere, 25 KB of LDS and ~350 registers per
// wave -> 4 waves per workgroup, one per SIMD.  The reduced forms keep ONE plane (W2 rows 144 B, dH2 pieces 4 KB): 16.7 KB per wave and
// < 256 registers -> EIGHT waves per workgroup, two independent tile streams per SIMD (152 KB).  Round 6: with a sixth of the matrix work
// the wave-pair form (ec_main_bwd_pair_kernel, two COUPLED waves per SIMD, two workgroup barriers per tile) and this kernel at one wave
// per SIMD ran the c5 step in the same time (A/B 7.00 vs 7.03 ms) - neither overlaps a wave's own gather / LDS round trips with anything.
template <int PREC> struct BwdCfg {
    static constexpr int NP = Mx<PREC>::NP;
    static constexpr int WAVES = PREC == PREC_X6 ? 4 : 8;
    static constexpr int WROW = NP * 2 * EC + 16;                     // bytes per row of a W2 piece image
    static constexpr int DIMG = NP * EC * 64;                         // bytes: piece planes of the dH2 tile
    static constexpr int WAVE_LDS = 32 * (EC + 4) + 8 * 32 + 8 * EC + 8 * 32 + DIMG / 4;      // floats per wave
    static constexpr size_t LDS_BYTES = 2 * EC * WROW + sizeof(float) * (WAVES * WAVE_LDS + 192);
};
static_assert(BwdCfg<PREC_X6>::WROW == BWD_WROW && BwdCfg<PREC_X6>::DIMG == BWD_DIMG && BwdCfg<PREC_X6>::WAVE_LDS == BWD_WAVE_LDS &&
              BwdCfg<PREC_X6>::LDS_BYTES == BWD_LDS_BYTES, "the fp32-accurate geometry is the one the pair kernel shares");
constexpr int BWD_MAX_WAVES = 8;                                      // trash rows behind dyb per workgroup (edgeconv plan: off_dyb)

template <int PREC, bool CSR>
__global__ void __launch_bounds__(64 * BwdCfg<PREC>::WAVES)
ec_main_bwd_kernel(const float *__restrict__ Pp, const float *__restrict__ Qp, const int32_t *__restrict__ idx,
                   const float *__restrict__ w2, const CoefIn cin,
                   const float *__restrict__ g_pm, const uint8_t *__restrict__ jsel,
                   int BN, int N, int k, float slope,
                   float *__restrict__ GP, float *__restrict__ GQ, float *__restrict__ partial,
                   const int *__restrict__ pos, float *__restrict__ dyb)
{
    typedef BwdCfg<PREC> Cfg;
    constexpr int BWD_WAVES = Cfg::WAVES, BWD_WROW = Cfg::WROW, BWD_WAVE_LDS = Cfg::WAVE_LDS;      // (shadow the fp32-accurate globals)
    extern __shared__ __attribute__((aligned(16))) float smem[];
    unsigned char *sW3 = reinterpret_cast<unsigned char *>(smem);        // [co][hi | mid | lo][ci] bf16 pieces of W2
    unsigned char *sW3T = sW3 + EC * BWD_WROW;                           // [ci][hi | mid | lo][co] pieces of W2^T
    float *wbase = reinterpret_cast<float *>(sW3T + EC * BWD_WROW);
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int i = lane & 31, h = lane >> 5;
    float *tr = wbase + (size_t)wv * BWD_WAVE_LDS;              // [32][TLD] transpose tile == [8][64 lanes][4] DMA image
    int *mall = reinterpret_cast<int *>(tr + 32 * TLD);         // [8][32] neighbour id of each row of each tile
    float *qt = reinterpret_cast<float *>(mall + 8 * 32);       // [8][64] Q' rows of the super-tile's points
    int *pall = reinterpret_cast<int *>(qt + 8 * EC);           // [8][32] CSR row of each row of each tile (CSR mode)
    // dH2 as 16-bit operand PIECES, [plane][co][edge column], 64-byte rows.  Written from the C/D layout (a lane's 8 edges of
    // one K-block are one 16-byte slot: column 16 kb + 8 h + e for register 8 kb + e), read back TRANSPOSED with
    // ds_read_b64_tr_b16 as the A operand of (3): the split of (2) is reused, no fp32 transpose and no second split.
    // Slot s of row co sits at physical slot s ^ ((co >> 1) & 3): conflict-free b128 writes (32 banks) and transposed reads.
    unsigned char *dimg = reinterpret_cast<unsigned char *>(pall + 8 * 32);
    const int trash_row = BN * k + (int)(blockIdx.x * BWD_MAX_WAVES + wv);     // where masked rows are stored (never read)

    for (int e = tid; e < EC * EC; e += 64 * BWD_WAVES) {
        const int co = e >> 6, ci = e & 63;
        mm_store1<PREC>(w2[e], reinterpret_cast<unsigned short *>(sW3 + co * BWD_WROW) + ci, EC);
        mm_store1<PREC>(w2[e], reinterpret_cast<unsigned short *>(sW3T + ci * BWD_WROW) + co, EC);
    }
    float *sC = wbase + BWD_WAVES * BWD_WAVE_LDS;       // [192] BN2-backward constants (re-read per tile: the kernel is at the VGPR cap)
    if (tid < EC) bwd_coef(cin, tid, blockIdx.x == 0, sC);       // (a separate 5 us launch otherwise)
    const unsigned char *wr0 = sW3 + i * BWD_WROW + 64 * h, *wr1 = sW3 + (32 + i) * BWD_WROW + 64 * h;        // + piece * 128 + kb * 16
    const unsigned char *wt0 = sW3T + i * BWD_WROW + 64 * h, *wt1 = sW3T + (32 + i) * BWD_WROW + 64 * h;
    f32x16 dW00, dW01, dW10, dW11;
#pragma unroll
    for (int r = 0; r < 16; ++r) { dW00[r] = 0.f; dW01[r] = 0.f; dW10[r] = 0.f; dW11[r] = 0.f; }
    const int pA = 4 * ((i >> 2) & 1) + (i >> 3);
    const int jjA = i & 3;
    const int T = (k + 3) >> 2;
    __syncthreads();

    const int nst = (BN + 7) >> 3;
    const UnitDeal deal(nst, N >> 3, BN / N, (N & 7) == 0, BWD_WAVES, wv);
    for (int w = deal.first; w < deal.count; w += deal.stride) {
        const int n0 = deal.unit(w) * 8;
        const int ptA = min(n0 + pA, BN - 1);
        const int bA = ptA / N;
        float Ag0[4], Ag1[4], gq0[4], gq1[4];
        unsigned js0 = 0, js1 = 0;                      // winning slot of the 4 points, one byte each
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int pt = n0 + 4 * h + q;
            const bool ok = pt < BN;
            const size_t o = (size_t)min(pt, BN - 1) * EC + i;
            Ag0[q] = ok ? sC[CF_A + i] * g_pm[o] : 0.f;
            Ag1[q] = ok ? sC[CF_A + 32 + i] * g_pm[o + 32] : 0.f;
            js0 |= (ok ? (unsigned)jsel[o] : 255u) << (8 * q);
            js1 |= (ok ? (unsigned)jsel[o + 32] : 255u) << (8 * q);
            gq0[q] = 0.f; gq1[q] = 0.f;
        }
        // neighbour ids of every tile (lane half h takes tiles h, h+2, ...), Q' rows by DMA, first P' gather by DMA
        // (global row index b N + m: the gather and the G_P atomics both address rows of (B N, 64) arrays)
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) {
            const int t = 2 * tt + h;
            if (t < T) {
                const int jc = 4 * t + jjA;
                mall[t * 32 + i] = (GFS3D_ABL & 2) ? bA * N : bA * N + idx[(size_t)ptA * k + min(jc, k - 1)];
                if (CSR) pall[t * 32 + i] = (jc < k && n0 + pA < BN) ? pos[(size_t)ptA * k + jc] : trash_row;
            }
        }
        __builtin_amdgcn_wave_barrier();
        auto gather = [&](int t) {                     // lane (edge row i, half h): 8 x 16 B of P'[m][32h..32h+31]
            const float *src = Pp + (size_t)mall[t * 32 + i] * EC + 32 * h;
#pragma unroll
            for (int v = 0; v < 8; ++v)
                lds_dma16(src + 4 * v, tr + v * 256);
        };
        asm volatile("" ::: "memory");
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int pt = min(n0 + (lane >> 4) + 4 * c, BN - 1);
            lds_dma16(Qp + (size_t)pt * EC + 4 * (lane & 15), qt + c * 256);
        }
        gather(0);
        asm volatile("" ::: "memory");
        for (int t = 0; t < T; ++t) {
            // tile 0: everything older (ids, Q', P' gather); later tiles: all but the previous tile's 32 atomics
            // (the previous tile issued, AFTER this tile's gather: 32 stores / atomics in the fp32-accurate and the atomic forms, 16 in
            // the reduced CSR form - one dword of two bf16 per row - so vmcnt(32) there would let the gather itself stay in flight)
            if (t == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else if (GFS3D_ABL & 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else if constexpr (CSR && PREC != PREC_X6) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
            const int *mrow = CSR ? pall + t * 32 : mall + t * 32;
            float a[32];
            {
                const float4 *p4 = reinterpret_cast<const float4 *>(tr) + lane;              // DMA image [v][lane]
                const float4 *q4 = reinterpret_cast<const float4 *>(qt + pA * EC + 32 * h);
#pragma unroll
                for (int v = 0; v < 8; ++v) {
                    const float4 pv = p4[v * 64], qv = q4[v];
                    float z;
                    z = pv.x + qv.x; a[4 * v + 0] = fmaxf(z, z * slope);
                    z = pv.y + qv.y; a[4 * v + 1] = fmaxf(z, z * slope);
                    z = pv.z + qv.z; a[4 * v + 2] = fmaxf(z, z * slope);
                    z = pv.w + qv.w; a[4 * v + 3] = fmaxf(z, z * slope);
                }
            }
            constexpr int NP = Mx<PREC>::NP;
            // 16-bit operand pieces of 8 consecutive values, two per dword (PREC_X6: [0] hi, [1] mid, [2] lo)
            auto pieces = [&](const float *v8, uint4 (&out)[NP]) { mm_pieces8<PREC>(v8, out); };
            // both accumulator chains of a K-step, alternating
            auto mac2 = [&](const uint4 (&a_)[NP], const uint4 (&b0_)[NP], const uint4 (&b1_)[NP], f32x16 &c0, f32x16 &c1) {
                if constexpr (PREC == PREC_X6) {
                    constexpr int PA[6] = GFS3D_X6_PA, PB[6] = GFS3D_X6_PB;
#pragma unroll
                    for (int term = 0; term < 6; ++term) {
                        c0 = mm_mfma<PREC>(a_[PA[term]], b0_[PB[term]], c0);
                        c1 = mm_mfma<PREC>(a_[PA[term]], b1_[PB[term]], c1);
                    }
                } else {
                    c0 = mm_mfma<PREC>(a_[0], b0_[0], c0);
                    c1 = mm_mfma<PREC>(a_[0], b1_[0], c1);
                }
            };
            // (1) H2 = A1 W2^T
            f32x16 acc0, acc1;
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) {
                uint4 ap[NP], b0[NP], b1[NP];
#pragma unroll
                for (int pc = 0; pc < NP; ++pc) {              // issue the weight reads first: the split below hides them
                    b0[pc] = *reinterpret_cast<const uint4 *>(wr0 + pc * 128 + kb * 16);
                    b1[pc] = *reinterpret_cast<const uint4 *>(wr1 + pc * 128 + kb * 16);
                }
                __builtin_amdgcn_sched_barrier(0);
                pieces(a + 8 * kb, ap);
                if (!(GFS3D_ABL & 64)) mac2(ap, b0, b1, acc0, acc1);
                else asm volatile("" :: "v"(ap[0].x), "v"(b0[0].x), "v"(b1[0].x));
                __builtin_amdgcn_sched_barrier(0);
            }
            // a1: A-layout -> C/D layout through the transpose tile
#pragma unroll
            for (int v = 0; v < 8; ++v)
                *reinterpret_cast<float4 *>(tr + i * TLD + 32 * h + 4 * v) =
                    make_float4(a[4 * v + 0], a[4 * v + 1], a[4 * v + 2], a[4 * v + 3]);
            __builtin_amdgcn_wave_barrier();
            float aD0[16], aD1[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                aD0[r] = tr[drow(r, h) * TLD + i];
                aD1[r] = tr[drow(r, h) * TLD + 32 + i];
            }
            __builtin_amdgcn_wave_barrier();
            // dH2 in place (C/D layout): Cc*h2 + Bc + routed term; masked rows -> 0
            const float Bc0 = sC[CF_BC + i], Bc1 = sC[CF_BC + 32 + i], Cc0 = sC[CF_CC + i], Cc1 = sC[CF_CC + 32 + i];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int q = r >> 2;
                const int j = 4 * t + (r & 3);
                const bool ok = (j < k) && (n0 + 4 * h + q < BN);
                const float d0 = fmaf(Cc0, acc0[r], Bc0) + ((int)((js0 >> (8 * q)) & 255u) == j ? Ag0[q] : 0.f);
                const float d1 = fmaf(Cc1, acc1[r], Bc1) + ((int)((js1 >> (8 * q)) & 255u) == j ? Ag1[q] : 0.f);
                acc0[r] = ok ? d0 : 0.f;
                acc1[r] = ok ? d1 : 0.f;
            }
            // a1 is needed once more, for LReLU'(z1): keep its signs as two bit masks instead of 32 registers
            unsigned pos0 = 0u, pos1 = 0u;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                pos0 |= (aD0[r] > 0.f ? 1u : 0u) << r;
                pos1 |= (aD1[r] > 0.f ? 1u : 0u) << r;
            }
            // (2) dW2[co][ci] += sum_edges dH2[e][co] a1[e][ci]   (register r <-> the same edge in both operands)
            // (K-block kb = registers 8kb..8kb+7 of both lane halves: the same 16 edges in both operands)
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                float v8[8];
                uint4 d[NP], x0[NP], x1[NP];
                if (!(GFS3D_ABL & 8)) {
                    pieces(aD0 + 8 * kb, x0);
                    pieces(aD1 + 8 * kb, x1);
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) v8[e] = acc0[8 * kb + e];
                pieces(v8, d);
                if (GFS3D_ABL & 8) {
#pragma unroll
                    for (int pc = 0; pc < NP; ++pc) { x0[pc] = d[pc]; x1[pc] = d[(pc + 1) % NP]; }
                }
#pragma unroll
                for (int pc = 0; pc < NP; ++pc)
                    *reinterpret_cast<uint4 *>(dimg + pc * (EC * 64) + i * 64 + 16 * ((2 * kb + h) ^ ((i >> 1) & 3))) = d[pc];
                if (!(GFS3D_ABL & 4)) mac2(d, x0, x1, dW00, dW01);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int e = 0; e < 8; ++e) v8[e] = acc1[8 * kb + e];
                pieces(v8, d);
#pragma unroll
                for (int pc = 0; pc < NP; ++pc)
                    *reinterpret_cast<uint4 *>(dimg + pc * (EC * 64) + (32 + i) * 64 + 16 * ((2 * kb + h) ^ ((i >> 1) & 3))) = d[pc];
                if (!(GFS3D_ABL & 4)) mac2(d, x0, x1, dW10, dW11);
                __builtin_amdgcn_sched_barrier(0);
            }
            // (dH2 reaches the A-layout of (3) as pieces, through dimg.)  The transpose tile is idle until the next tile's a1
            // write: start the next gather into it, ahead of this tile's stores in issue order
            __builtin_amdgcn_wave_barrier();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (t + 1 < T) gather(t + 1);
            asm volatile("" ::: "memory");
            // (3) dA1 = dH2 W2 :  B[k = co = 32h+s][j = ci]
            f32x16 e0, e1;
#pragma unroll
            for (int r = 0; r < 16; ++r) { e0[r] = 0.f; e1[r] = 0.f; }
            // A fragment of K-block kb = pieces of dH2[edge i][co = 32h + 8kb + e]: two transposed reads of 4 co rows each.  In
            // its 16-lane group, lane 4q + pq supplies the address of row q, edge columns of chunk pq (edges 16 (g & 1) + 4 pq ..
            // + 3, stored at column 16 (g & 1) + 8 (pq & 1) + 4 (pq >> 1)), and receives its own edge's column.
            const int tq = (lane >> 2) & 3, tp = lane & 3, tslot = 2 * ((lane >> 4) & 1) + (tp & 1);
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) {                    // K = co: lane half h holds co 32h + 8kb ..
                uint4 ap[NP], b0[NP], b1[NP];
#pragma unroll
                for (int pc = 0; pc < NP; ++pc) {
                    b0[pc] = *reinterpret_cast<const uint4 *>(wt0 + pc * 128 + kb * 16);
                    b1[pc] = *reinterpret_cast<const uint4 *>(wt1 + pc * 128 + kb * 16);
                }
#pragma unroll
                for (int pc = 0; pc < NP; ++pc) {
                    unsigned w4[4];
#pragma unroll
                    for (int hf = 0; hf < 2; ++hf) {
                        const int co = 32 * h + 8 * kb + 4 * hf + tq;
                        const unsigned char *src = dimg + pc * (EC * 64) + co * 64 + 16 * (tslot ^ ((co >> 1) & 3)) + 8 * (tp >> 1);
                        const s16x4_ v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4 *)src);
                        const uint2 u = __builtin_bit_cast(uint2, v);
                        w4[2 * hf] = u.x; w4[2 * hf + 1] = u.y;
                    }
                    ap[pc] = make_uint4(w4[0], w4[1], w4[2], w4[3]);
                }
                if (!(GFS3D_ABL & 32)) mac2(ap, b0, b1, e0, e1);
                else asm volatile("" :: "v"(ap[0].x), "v"(b0[0].x), "v"(b1[0].x));
                __builtin_amdgcn_sched_barrier(0);
            }
            // dy1 = dA1 * LReLU'(z1); G_Q (in-lane), G_P (atomic 128-B half rows)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int q = r >> 2;
                const int j = 4 * t + (r & 3);
                const bool ok = (j < k) && (n0 + 4 * h + q < BN);
                // exactly 32 atomic instructions per tile (the vmcnt(32) above counts on it): masked rows add 0 to a
                // valid row (their neighbour slot and point index are clamped)
                const float dy0 = ok ? e0[r] * ((pos0 >> r) & 1u ? 1.f : slope) : 0.f;
                const float dy1 = ok ? e1[r] * ((pos1 >> r) & 1u ? 1.f : slope) : 0.f;
                gq0[q] += dy0;
                gq1[q] += dy1;
                const unsigned row = (unsigned)mrow[drow(r, h)] * EC + i;
                if (CSR) {                                  // (a masked row's values are 0 and its row is the wave's scratch row)
                    if constexpr (PREC == PREC_X6) {
                        if (!(GFS3D_ABL & 16)) {
                            dyb[row] = dy0;
                            dyb[row + 32] = dy1;
                        } else asm volatile("" :: "v"(dy0), "v"(dy1));
                    } else {
                        // reduced modes: the row as 32 dwords of two bf16 (channels i | 32 + i) - half the bytes of the one
                        // tensor of size N k 64 the backward writes and ec_bwd_segsum_kernel reads back (the kernel is bound by
                        // exactly that traffic once the matrix work is a single product)
                        __builtin_nontemporal_store(mm_pack2<PREC_BF16>(dy0, dy1), &reinterpret_cast<unsigned *>(dyb)[(size_t)mrow[drow(r, h)] * 32 + i]);
                    }
                } else if (!(GFS3D_ABL & 1)) {
                    atomicAdd(&GP[row], dy0);
                    atomicAdd(&GP[row + 32], dy1);
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int pt = n0 + 4 * h + q;
            if (pt < BN) {
                GQ[(size_t)pt * EC + i] = gq0[q];
                GQ[(size_t)pt * EC + 32 + i] = gq1[q];
            }
        }
    }
    // block-level sum of the waves' dW2 accumulators in a FIXED order (one [64][64] image per wave: 64 / 128 KB of the
    // LDS this block owns), then one partial per block
    __syncthreads();
    float *sD = smem + (size_t)wv * EC * EC;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = drow(r, h);                     // C/D layout: row = co (within half), col = ci = i
        sD[row * EC + i] = dW00[r];
        sD[row * EC + 32 + i] = dW01[r];
        sD[(32 + row) * EC + i] = dW10[r];
        sD[(32 + row) * EC + 32 + i] = dW11[r];
    }
    __syncthreads();
    float *out = partial + (size_t)blockIdx.x * BWD_PART;
    static_assert((size_t)BWD_WAVES * EC * EC * sizeof(float) <= Cfg::LDS_BYTES, "the dW2 images reuse the workgroup's LDS");
    for (int e = tid; e < EC * EC; e += 64 * BWD_WAVES) {
        float sum = smem[e];
#pragma unroll
        for (int w = 1; w < BWD_WAVES; ++w) sum += smem[w * EC * EC + e];
        out[e] = sum;
    }
}

// B3': the same backward (fp32-accurate mode, CSR form) as a PAIR of waves per SIMD.  In ec_main_bwd_kernel a wave's matrix
// work (3 x 48 MFMAs per tile = 4.6 k cycles), its ~1200 vector instructions and its LDS / gather round trips are strictly serial
// (one wave per SIMD: 352 registers) - ablations: the MFMAs are 32 % of the time and overlap with nothing.  Here wave p (FRONT)
// and wave p + 4 (BACK) of an 8-wave workgroup share SIMD p and split a tile:
//   FRONT  gather P' rows (registers, one tile ahead), a1 = LReLU(P'[m] + Q'[n]), split, (1) H2 = A1 W2^T, dH2 in place;
//          hands over, through LDS: the a1 pieces as they stand in the A layout (`aimg`: rows = edges, 128 B per plane),
//          the dH2 pieces (`dimg`, as before);
//   BACK   (2) dW2 += dH2^T A1: A = dimg rows as written, B = aimg read TRANSPOSED (ds_read_b64_tr_b16: lane = channel, 8
//          edges) - no fp32 transpose of a1 and no second split;  (3) dA1 = dH2 W2: A = dimg transposed (as before), B = the
//          ONE W2 image read transposed (no W2^T image);  dy1 (the sign of a1 from the hi pieces it already holds), G_Q,
//          the row stores.
// The hand-off is single-buffered: FRONT works one tile ahead and keeps its a1 pieces in registers; two workgroup barriers per
// tile (X: BACK has read tile g - 1, FRONT may write tile g; Y: tile g is in LDS).  Both roles fit 256 registers.
// LDS: W2 image 25.6 KB + per pair (aimg 12 KB + dimg 12 KB + Q' tile 2 KB + CSR rows 1 KB) = 134 KB.
// the hand-off barriers order LDS traffic only: no wait for the global loads in flight (the next tile's P' rows) or the dy1 stores
__device__ __forceinline__ void bp_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
constexpr int BP_PAIRS = 4;
constexpr int BP_AIMG = 3 * 32 * 128;                                       // bytes: a1 piece planes [plane][edge][64 ci], 16-B slots swizzled
constexpr int BP_PAIR_LDS = BP_AIMG + BWD_DIMG + 8 * EC * 4 + 8 * 32 * 4;      // aimg | dimg | Q' tile | CSR rows
constexpr size_t BP_LDS_BYTES = EC * BWD_WROW + (size_t)BP_PAIRS * BP_PAIR_LDS + 192 * 4;
static_assert(BP_LDS_BYTES >= (size_t)BP_PAIRS * EC * EC * 4, "the dW2 reduction images reuse the workgroup's LDS");

template <int PREC>
__global__ void __launch_bounds__(128 * BP_PAIRS)
ec_main_bwd_pair_kernel(const float *__restrict__ Pp, const float *__restrict__ Qp, const int32_t *__restrict__ idx,
                        const float *__restrict__ w2, const CoefIn cin,
                        const float *__restrict__ g_pm, const uint8_t *__restrict__ jsel,
                        int BN, int N, int k, float slope,
                        float *__restrict__ GQ, float *__restrict__ partial,
                        const int *__restrict__ pos, float *__restrict__ dyb)
{
    constexpr int NP = Mx<PREC>::NP;               // operand piece planes: 3 (fp32-accurate) or 1 (reduced modes: bf16 operands)
    extern __shared__ __attribute__((aligned(16))) float smem[];
    unsigned char *sW3 = reinterpret_cast<unsigned char *>(smem);        // [co][hi | mid | lo][ci] pieces of W2
    // the wave index as a SCALAR (readfirstlane): role, pair and the walk derived from it are wave-uniform to the compiler, so the
    // role bodies are entered by s_cbranch_scc and a wave can only ever execute the barriers of its own role
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pair = wv & 3;
    const bool front = wv < BP_PAIRS;
    static_assert(BP_PAIRS <= BWD_MAX_WAVES, "the trash rows behind dyb are sized by BWD_MAX_WAVES per workgroup (edgeconv plan: off_dyb)");
    const int i = lane & 31, h = lane >> 5;
    unsigned char *pbase = sW3 + EC * BWD_WROW + (size_t)pair * BP_PAIR_LDS;
    unsigned char *aimg = pbase;
    unsigned char *dimg = aimg + BP_AIMG;
    float *qt = reinterpret_cast<float *>(dimg + BWD_DIMG);             // [8][64] Q' rows of FRONT's super-tile
    int *pall = reinterpret_cast<int *>(qt + 8 * EC);                   // [8][32] CSR row of each row of each tile of BACK's super-tile
    float *sC = reinterpret_cast<float *>(sW3 + EC * BWD_WROW + (size_t)BP_PAIRS * BP_PAIR_LDS);
    const int trash_row = BN * k + (int)(blockIdx.x * BWD_MAX_WAVES + pair);  // where masked rows are stored (never read)

    for (int e = tid; e < EC * EC; e += 128 * BP_PAIRS)
        mm_store1<PREC>(w2[e], reinterpret_cast<unsigned short *>(sW3 + (e >> 6) * BWD_WROW) + (e & 63), EC);
    if (tid < EC) bwd_coef(cin, tid, blockIdx.x == 0, sC);
    __syncthreads();

    const int pA = 4 * ((i >> 2) & 1) + (i >> 3), jjA = i & 3;
    const int T = (k + 3) >> 2;
    const int nst = (BN + 7) >> 3;
    const UnitDeal deal(nst, N >> 3, BN / N, (N & 7) == 0, BP_PAIRS, pair);
    const int first0 = deal.first - pair;                                 // pair 0's first unit: the longest walk of the workgroup
    const int nsup = deal.count > first0 ? (deal.count - first0 + deal.stride - 1) / deal.stride : 0;
    const int G = nsup * T;                                               // tile steps every pair runs (uniform barrier count)

    auto mac2 = [&](const uint4 (&a_)[NP], const uint4 (&b0_)[NP], const uint4 (&b1_)[NP], f32x16 &c0, f32x16 &c1) {
        if constexpr (PREC == PREC_X6) {
            constexpr int PA[6] = GFS3D_X6_PA, PB[6] = GFS3D_X6_PB;
#pragma unroll
            for (int term = 0; term < 6; ++term) {
                c0 = mm_mfma<PREC>(a_[PA[term]], b0_[PB[term]], c0);
                c1 = mm_mfma<PREC>(a_[PA[term]], b1_[PB[term]], c1);
            }
        } else {
            c0 = mm_mfma<PREC>(a_[0], b0_[0], c0);
            c1 = mm_mfma<PREC>(a_[0], b1_[0], c1);
        }
    };

    if (front) {
        // ------------------------------------------------------------------ FRONT
        const unsigned char *wr0 = sW3 + i * BWD_WROW + 64 * h, *wr1 = sW3 + (32 + i) * BWD_WROW + 64 * h;     // + piece * 128 + kb * 16
        int u = deal.first, t = 0;                 // my unit (super-tile) and tile of step g
        int n0 = 0, ptA = 0;
        float Ag0[4], Ag1[4];
        unsigned js0 = 0, js1 = 0;
        float4 pn[8];                              // P' half row of the NEXT tile's edge (in flight)
        auto row_of = [&](int uu, int tt) -> const float * {       // P' half row of edge (point pA, slot 4 tt + jjA) of unit uu
            const int nn0 = deal.unit(uu) * 8;
            const int pt = min(nn0 + pA, BN - 1);
            const int m = idx[(size_t)pt * k + min(4 * tt + jjA, k - 1)];
            return Pp + ((size_t)(pt / N) * N + m) * EC + 32 * h;
        };
        auto fetch = [&](int uu, int tt) {
            const float4 *src = reinterpret_cast<const float4 *>(row_of(uu, tt));
#pragma unroll
            for (int v = 0; v < 8; ++v) pn[v] = src[v];
        };
        if (u < deal.count) fetch(u, 0);
#ifdef BP_TIMING
        long long tk[4] = {0, 0, 0, 0};
#endif
        for (int g = 0; g <= G; ++g) {
#ifdef BP_TIMING
            const long long s0 = __builtin_amdgcn_s_memtime();
#endif
            const bool work = g < G && u < deal.count;
            uint4 ap[4][NP];
            f32x16 acc0, acc1;
            bool okrow[16];
            if (work) {
                if (t == 0) {                      // a new super-tile: its routed-gradient terms and Q' rows
                    n0 = deal.unit(u) * 8;
                    ptA = min(n0 + pA, BN - 1);
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int pt = n0 + 4 * h + q;
                        const bool ok = pt < BN;
                        const size_t o = (size_t)min(pt, BN - 1) * EC + i;
                        Ag0[q] = ok ? sC[CF_A + i] * g_pm[o] : 0.f;
                        Ag1[q] = ok ? sC[CF_A + 32 + i] * g_pm[o + 32] : 0.f;
                        if (q == 0) { js0 = 0; js1 = 0; }
                        js0 |= (ok ? (unsigned)jsel[o] : 255u) << (8 * q);
                        js1 |= (ok ? (unsigned)jsel[o + 32] : 255u) << (8 * q);
                    }
#pragma unroll
                    for (int c = 0; c < 2; ++c) {
                        const int pt = min(n0 + (lane >> 4) + 4 * c, BN - 1);
                        *reinterpret_cast<float4 *>(qt + ((lane >> 4) + 4 * c) * EC + 4 * (lane & 15)) =
                            *reinterpret_cast<const float4 *>(Qp + (size_t)pt * EC + 4 * (lane & 15));
                    }
                    __builtin_amdgcn_wave_barrier();
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                }
                // a1 of this tile from the rows requested one step ago; request the next tile's
                float a[32];
                {
                    const float4 *q4 = reinterpret_cast<const float4 *>(qt + pA * EC + 32 * h);
#pragma unroll
                    for (int v = 0; v < 8; ++v) {
                        const float4 pv = pn[v], qv = q4[v];
                        float z;
                        z = pv.x + qv.x; a[4 * v + 0] = fmaxf(z, z * slope);
                        z = pv.y + qv.y; a[4 * v + 1] = fmaxf(z, z * slope);
                        z = pv.z + qv.z; a[4 * v + 2] = fmaxf(z, z * slope);
                        z = pv.w + qv.w; a[4 * v + 3] = fmaxf(z, z * slope);
                    }
                }
                {
                    int un = u, tn = t + 1;
                    if (tn == T) { tn = 0; un = u + deal.stride; }
                    if (un < deal.count) fetch(un, tn);
                }
                // (1) H2 = A1 W2^T
#pragma unroll
                for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
#pragma unroll
                for (int kb = 0; kb < 4; ++kb) {
                    uint4 b0[NP], b1[NP];
#pragma unroll
                    for (int pc = 0; pc < NP; ++pc) {
                        b0[pc] = *reinterpret_cast<const uint4 *>(wr0 + pc * 128 + kb * 16);
                        b1[pc] = *reinterpret_cast<const uint4 *>(wr1 + pc * 128 + kb * 16);
                    }
                    mm_pieces8<PREC>(a + 8 * kb, ap[kb]);
                    mac2(ap[kb], b0, b1, acc0, acc1);
                }
                // dH2 in place (C/D layout): Cc*h2 + Bc + routed term; masked rows -> 0
                const float Bc0 = sC[CF_BC + i], Bc1 = sC[CF_BC + 32 + i], Cc0 = sC[CF_CC + i], Cc1 = sC[CF_CC + 32 + i];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int q = r >> 2;
                    const int j = 4 * t + (r & 3);
                    const bool ok = (j < k) && (n0 + 4 * h + q < BN);
                    okrow[r] = ok;
                    const float d0 = fmaf(Cc0, acc0[r], Bc0) + ((int)((js0 >> (8 * q)) & 255u) == j ? Ag0[q] : 0.f);
                    const float d1 = fmaf(Cc1, acc1[r], Bc1) + ((int)((js1 >> (8 * q)) & 255u) == j ? Ag1[q] : 0.f);
                    acc0[r] = ok ? d0 : 0.f;
                    acc1[r] = ok ? d1 : 0.f;
                }
            }
#ifdef BP_TIMING
            asm volatile("" : "+v"(acc0[0]), "+v"(acc1[15]));
            const long long s1 = __builtin_amdgcn_s_memtime();
#endif
            bp_barrier();                          // X: BACK has read the hand-off of step g - 1
#ifdef BP_TIMING
            const long long s2 = __builtin_amdgcn_s_memtime();
#endif
            if (work) {
#pragma unroll
                for (int kb = 0; kb < 4; ++kb)
#pragma unroll
                    for (int pc = 0; pc < NP; ++pc)
                        *reinterpret_cast<uint4 *>(aimg + pc * (32 * 128) + i * 128 + 16 * ((4 * h + kb) ^ (i & 7))) = ap[kb][pc];
#pragma unroll
                for (int kb = 0; kb < 2; ++kb) {
                    float v8[8];
                    uint4 d[NP];
#pragma unroll
                    for (int e = 0; e < 8; ++e) v8[e] = acc0[8 * kb + e];
                    mm_pieces8<PREC>(v8, d);
#pragma unroll
                    for (int pc = 0; pc < NP; ++pc)
                        *reinterpret_cast<uint4 *>(dimg + pc * (EC * 64) + i * 64 + 16 * ((2 * kb + h) ^ ((i >> 1) & 3))) = d[pc];
#pragma unroll
                    for (int e = 0; e < 8; ++e) v8[e] = acc1[8 * kb + e];
                    mm_pieces8<PREC>(v8, d);
#pragma unroll
                    for (int pc = 0; pc < NP; ++pc)
                        *reinterpret_cast<uint4 *>(dimg + pc * (EC * 64) + (32 + i) * 64 + 16 * ((2 * kb + h) ^ ((i >> 1) & 3))) = d[pc];
                }
                if (++t == T) { t = 0; u += deal.stride; }
            }
#ifdef BP_TIMING
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            const long long s3 = __builtin_amdgcn_s_memtime();
#endif
            bp_barrier();                          // Y: the hand-off of step g is in LDS
#ifdef BP_TIMING
            const long long s4 = __builtin_amdgcn_s_memtime();
            tk[0] += s1 - s0; tk[1] += s2 - s1; tk[2] += s3 - s2; tk[3] += s4 - s3;
#endif
        }
#ifdef BP_TIMING
        if (blockIdx.x == 9 && lane == 0 && pair == 1) printf("bp FRONT steps %d: compute %lld waitX %lld write %lld waitY %lld\n", G + 1, tk[0], tk[1], tk[2], tk[3]);
#endif
    } else {
        // ------------------------------------------------------------------ BACK
        f32x16 dW00, dW01, dW10, dW11;
#pragma unroll
        for (int r = 0; r < 16; ++r) { dW00[r] = 0.f; dW01[r] = 0.f; dW10[r] = 0.f; dW11[r] = 0.f; }
        int u = deal.first, t = 0, n0 = 0;
        float gq0[4], gq1[4];
        const int g4 = lane >> 4, tq = (lane >> 2) & 3, tp = lane & 3;
        // transposed-read lane roles (a 16-lane group transposes 4 rows x 16 columns of 2 bytes: lane 4 q + pq supplies row q,
        // columns 4 pq .. 4 pq + 3, lane c receives column c of the 4 rows)
        const int tslot = 2 * (g4 & 1) + (tp & 1);                       // dimg: see ec_main_bwd_kernel
#ifdef BP_TIMING
        long long tk[4] = {0, 0, 0, 0};
#endif
        for (int g = 0; g <= G; ++g) {
#ifdef BP_TIMING
            const long long s0 = __builtin_amdgcn_s_memtime();
#endif
            const bool work = g >= 1 && u < deal.count;                   // (u, t): the tile of step g - 1
            f32x16 e0, e1;
            unsigned pos0 = 0u, pos1 = 0u;
            if (work) {
                if (t == 0) {                      // a new super-tile: CSR rows of its edges, G_Q sums
                    n0 = deal.unit(u) * 8;
                    const int ptA = min(n0 + pA, BN - 1);
#pragma unroll
                    for (int tt = 0; tt < 4; ++tt) {
                        const int tl = 2 * tt + h;
                        if (tl < T) {
                            const int jc = 4 * tl + jjA;
                            pall[tl * 32 + i] = (jc < k && n0 + pA < BN) ? pos[(size_t)ptA * k + jc] : trash_row;
                        }
                    }
#pragma unroll
                    for (int q = 0; q < 4; ++q) { gq0[q] = 0.f; gq1[q] = 0.f; }
                }
                // (2) dW2[co][ci] += sum_edges dH2[e][co] a1[e][ci]: K-block kb, K index (h, e) <-> edge 16 kb + 8 (e >> 2) + 4 h + (e & 3)
#pragma unroll
                for (int kb = 0; kb < 2; ++kb) {
                    uint4 dA0[NP], dA1[NP], x0[NP], x1[NP];
#pragma unroll
                    for (int pc = 0; pc < NP; ++pc) {
                        dA0[pc] = *reinterpret_cast<const uint4 *>(dimg + pc * (EC * 64) + i * 64 + 16 * ((2 * kb + h) ^ ((i >> 1) & 3)));
                        dA1[pc] = *reinterpret_cast<const uint4 *>(dimg + pc * (EC * 64) + (32 + i) * 64 + 16 * ((2 * kb + h) ^ ((i >> 1) & 3)));
                        unsigned w0[4], w1[4];
#pragma unroll
                        for (int hf = 0; hf < 2; ++hf) {
                            const int erow = 16 * kb + 8 * hf + 4 * h + tq;              // edge row this lane addresses
#pragma unroll
                            for (int ct = 0; ct < 2; ++ct) {
                                const int c0 = 32 * ct + 16 * (g4 & 1) + 4 * tp;         // its 4 channels
                                const unsigned char *src = aimg + pc * (32 * 128) + erow * 128 + 16 * ((c0 >> 3) ^ (erow & 7)) + 2 * (c0 & 7);
                                const s16x4_ v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4 *)src);
                                const uint2 uu = __builtin_bit_cast(uint2, v);
                                if (ct == 0) { w0[2 * hf] = uu.x; w0[2 * hf + 1] = uu.y; }
                                else { w1[2 * hf] = uu.x; w1[2 * hf + 1] = uu.y; }
                            }
                        }
                        x0[pc] = make_uint4(w0[0], w0[1], w0[2], w0[3]);
                        x1[pc] = make_uint4(w1[0], w1[1], w1[2], w1[3]);
                        if (pc == 0) {
                            // LReLU'(z1) needs a1 > 0 for the edges of this K-block - the hi pieces just read ARE a1 in this lane's
                            // C/D row order (register 8 kb + e): positive <=> the bf16 is a positive non-zero number (a1 > 0 rounds to
                            // +0 only below 4.6e-41).  No sign masks through LDS, no second pass over a1.
#pragma unroll
                            for (int wq = 0; wq < 4; ++wq) {
                                const int r = 8 * kb + 2 * wq;
                                pos0 |= ((int)(w0[wq] << 16) > 0 ? 1u : 0u) << r;
                                pos0 |= ((int)w0[wq] >= 0x10000 ? 1u : 0u) << (r + 1);
                                pos1 |= ((int)(w1[wq] << 16) > 0 ? 1u : 0u) << r;
                                pos1 |= ((int)w1[wq] >= 0x10000 ? 1u : 0u) << (r + 1);
                            }
                        }
                    }
                    mac2(dA0, x0, x1, dW00, dW01);
                    mac2(dA1, x0, x1, dW10, dW11);
                }
                // (3) dA1 = dH2 W2 :  K = co, lane half h holds co 32 h + 8 kb ..
#pragma unroll
                for (int r = 0; r < 16; ++r) { e0[r] = 0.f; e1[r] = 0.f; }
#pragma unroll
                for (int kb = 0; kb < 4; ++kb) {
                    uint4 ap[NP], b0[NP], b1[NP];
#pragma unroll
                    for (int pc = 0; pc < NP; ++pc) {
                        unsigned w4[4], wb0[4], wb1[4];
#pragma unroll
                        for (int hf = 0; hf < 2; ++hf) {
                            const int co = 32 * h + 8 * kb + 4 * hf + tq;
                            {
                                const unsigned char *src = dimg + pc * (EC * 64) + co * 64 + 16 * (tslot ^ ((co >> 1) & 3)) + 8 * (tp >> 1);
                                const s16x4_ v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4 *)src);
                                const uint2 uu = __builtin_bit_cast(uint2, v);
                                w4[2 * hf] = uu.x; w4[2 * hf + 1] = uu.y;
                            }
#pragma unroll
                            for (int ct = 0; ct < 2; ++ct) {
                                const unsigned char *src = sW3 + co * BWD_WROW + pc * 128 + 2 * (32 * ct + 16 * (g4 & 1) + 4 * tp);
                                const s16x4_ v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4 *)src);
                                const uint2 uu = __builtin_bit_cast(uint2, v);
                                if (ct == 0) { wb0[2 * hf] = uu.x; wb0[2 * hf + 1] = uu.y; }
                                else { wb1[2 * hf] = uu.x; wb1[2 * hf + 1] = uu.y; }
                            }
                        }
                        ap[pc] = make_uint4(w4[0], w4[1], w4[2], w4[3]);
                        b0[pc] = make_uint4(wb0[0], wb0[1], wb0[2], wb0[3]);
                        b1[pc] = make_uint4(wb1[0], wb1[1], wb1[2], wb1[3]);
                    }
                    mac2(ap, b0, b1, e0, e1);
                }
            }
#ifdef BP_TIMING
            asm volatile("" : "+v"(e0[0]), "+v"(e1[15]), "+v"(pos0));
            const long long s1 = __builtin_amdgcn_s_memtime();
#endif
            bp_barrier();                          // X
#ifdef BP_TIMING
            const long long s2 = __builtin_amdgcn_s_memtime();
#endif
            if (work) {
                const int *mrow = pall + t * 32;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int q = r >> 2;
                    const int j = 4 * t + (r & 3);
                    const bool ok = (j < k) && (n0 + 4 * h + q < BN);
                    const float dy0 = ok ? e0[r] * ((pos0 >> r) & 1u ? 1.f : slope) : 0.f;
                    const float dy1 = ok ? e1[r] * ((pos1 >> r) & 1u ? 1.f : slope) : 0.f;
                    gq0[q] += dy0;
                    gq1[q] += dy1;
                    if constexpr (PREC == PREC_X6) {
                        const unsigned row = (unsigned)mrow[drow(r, h)] * EC + i;  // (a masked row's values are 0 and its row is the pair's scratch row)
                        // streamed: written once, read once by the next kernel - no reason to keep the lines in this XCD's L2, whose
                        // capacity the P' gathers of the front wave live on (228 -> 223 us per launch)
                        __builtin_nontemporal_store(dy0, &dyb[row]);
                        __builtin_nontemporal_store(dy1, &dyb[row + 32]);
                    } else                                  // reduced modes: the row as 32 dwords of two bf16 (channels i | 32 + i)
                        __builtin_nontemporal_store(mm_pack2<PREC_BF16>(dy0, dy1), &reinterpret_cast<unsigned *>(dyb)[(size_t)mrow[drow(r, h)] * 32 + i]);
                }
                if (++t == T) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int pt = n0 + 4 * h + q;
                        if (pt < BN) {
                            GQ[(size_t)pt * EC + i] = gq0[q];
                            GQ[(size_t)pt * EC + 32 + i] = gq1[q];
                        }
                    }
                    t = 0;
                    u += deal.stride;
                }
            }
#ifdef BP_TIMING
            const long long s3 = __builtin_amdgcn_s_memtime();
#endif
            bp_barrier();                          // Y
#ifdef BP_TIMING
            const long long s4 = __builtin_amdgcn_s_memtime();
            tk[0] += s1 - s0; tk[1] += s2 - s1; tk[2] += s3 - s2; tk[3] += s4 - s3;
#endif
        }
#ifdef BP_TIMING
        if (blockIdx.x == 9 && lane == 0 && pair == 1) printf("bp BACK  steps %d: gemms %lld waitX %lld dy1 %lld waitY %lld\n", G + 1, tk[0], tk[1], tk[2], tk[3]);
#endif
        // the four BACK waves' dW2 accumulators, summed in a FIXED order: one [64][64] image per wave, then one partial per block
        __syncthreads();
        float *sD = smem + (size_t)pair * EC * EC;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = drow(r, h);
            sD[row * EC + i] = dW00[r];
            sD[row * EC + 32 + i] = dW01[r];
            sD[(32 + row) * EC + i] = dW10[r];
            sD[(32 + row) * EC + 32 + i] = dW11[r];
        }
    }
    if (front) __syncthreads();                    // (pairs with BACK's barrier before its image stores)
    __syncthreads();
    float *out = partial + (size_t)blockIdx.x * BWD_PART;
    for (int e = tid; e < EC * EC; e += 128 * BP_PAIRS)
        out[e] = ((smem[e] + smem[EC * EC + e]) + smem[2 * EC * EC + e]) + smem[3 * EC * EC + e];
}

// B4 (CSR mode): G_P[m] = sum of the dy1 rows of m's in-edges - one contiguous CSR segment of `dyb`, read as a stream
// and summed in a fixed order - fused with the BN1-backward reductions of ec_bwd_bn1_reduce_kernel (which it replaces):
//     sum_n GQ  and  sum_e dy1 xhat1 = sum_m P''[m] GP[m] + sum_n Q''[n] GQ[n]
// wave per target, lane = channel; partial[2][64] doubles.
// HALF = true (reduced modes): rows of 32 dwords holding two bf16 each (channels i | 32 + i), summed in fp32
template <bool HALF>
__global__ void __launch_bounds__(256)
ec_bwd_segsum_kernel(const float *__restrict__ dyb, const int *__restrict__ rowptr,
                     const float *__restrict__ P, const float *__restrict__ Q, const float *__restrict__ GQ,
                     const float *__restrict__ stats, int BN, float *__restrict__ GP, const BnAcc partial, int rev)
{
    if constexpr (HALF) {
        // The two HALVES of a wave take two targets at once: lane = (target slot hw = lane >> 5, row slot sub = (lane >> 3) & 3, dword
        // quad c4 = lane & 7); one wave instruction reads 4 rows of 128 B per half, 8 of them in flight = 8 KB per wave as in the
        // fp32-row form (one target per wave with 4 KB in flight ran at half that form's rate for half its bytes: 161 us per launch
        // at B = 64).  A lane owns channels 4 c4 .. 4 c4 + 3 (low halves of the dwords) and 32 + 4 c4 .. (high halves).
        typedef float f32x4 __attribute__((ext_vector_type(4)));
        __shared__ double red[4][2][2][EC];
        const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
        const int hw = lane >> 5, sub = (lane >> 3) & 3, c4 = lane & 7;
        const unsigned *rows = reinterpret_cast<const unsigned *>(dyb);
        f32x4 mean1[2], rstd1[2];
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
            mean1[hf] = *reinterpret_cast<const f32x4 *>(stats + ST_MEAN1 * EC + 32 * hf + 4 * c4);
            rstd1[hf] = *reinterpret_cast<const f32x4 *>(stats + ST_RSTD1 * EC + 32 * hf + 4 * c4);
        }
        double s[8], sx[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { s[e] = 0.0; sx[e] = 0.0; }
        constexpr int SEG_TPW = 8;
        const int nchunk = (BN + SEG_TPW - 1) / SEG_TPW;
        for (int c_ = blockIdx.x * 4 + wv; c_ < nchunk; c_ += gridDim.x * 4) {
            const int ch = rev ? nchunk - 1 - c_ : c_;
            const int t0 = ch * SEG_TPW;
            const int myptr = rowptr[min(t0 + min(lane, SEG_TPW), BN)];
            for (int q = 0; q < SEG_TPW; q += 2) {
                const int tq = q + hw, tm = t0 + tq;
                const bool live = tm < BN;
                const int p0 = __shfl(myptr, tq, 64), p1 = live ? __shfl(myptr, tq + 1, 64) : p0;
                float acc[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[e] = 0.f;
                for (int p = p0 + sub; __any(p < p1); p += 32) {     // 32 rows of 128 B per half in flight per step
                    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
                    u32x4 v[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u)
                        v[u] = (p + 4 * u < p1) ? __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(rows + (size_t)(p + 4 * u) * 32 + 4 * c4)) : u32x4{0u, 0u, 0u, 0u};
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const unsigned w4[4] = {v[u][0], v[u][1], v[u][2], v[u][3]};
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            acc[e] += __builtin_bit_cast(float, w4[e] << 16);
                            acc[4 + e] += __builtin_bit_cast(float, w4[e] & 0xffff0000u);
                        }
                    }
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    float t = acc[e];
                    t += __shfl_xor(t, 8, 64);
                    t += __shfl_xor(t, 16, 64);
                    acc[e] = t;
                }
                if (sub == 0 && live) {
#pragma unroll
                    for (int hf = 0; hf < 2; ++hf) {
                        const size_t o = (size_t)tm * EC + 32 * hf + 4 * c4;
                        const f32x4 gq = *reinterpret_cast<const f32x4 *>(GQ + o), pv = *reinterpret_cast<const f32x4 *>(P + o),
                                    qv = *reinterpret_cast<const f32x4 *>(Q + o);
                        *reinterpret_cast<f32x4 *>(GP + o) = f32x4{acc[4 * hf], acc[4 * hf + 1], acc[4 * hf + 2], acc[4 * hf + 3]};
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            s[4 * hf + e] += (double)gq[e];
                            sx[4 * hf + e] += (double)(pv[e] * rstd1[hf][e] * acc[4 * hf + e]) + (double)((qv[e] - mean1[hf][e]) * rstd1[hf][e] * gq[e]);
                        }
                    }
                }
            }
        }
        if (sub == 0) {
#pragma unroll
            for (int hf = 0; hf < 2; ++hf)
#pragma unroll
                for (int e = 0; e < 4; ++e) { red[wv][hw][0][32 * hf + 4 * c4 + e] = s[4 * hf + e]; red[wv][hw][1][32 * hf + 4 * c4 + e] = sx[4 * hf + e]; }
        }
        __syncthreads();
        if (threadIdx.x < 128) {
            const int which = threadIdx.x >> 6;
            double t = 0.0;
#pragma unroll
            for (int w = 0; w < 4; ++w) t += red[w][0][which][lane] + red[w][1][which][lane];
            bn_acc_add(partial, blockIdx.x, which, lane, t);
        }
        return;
    } else {
    // lane = (row slot sub = lane >> 4, channel quad c4 = lane & 15): one wave instruction reads 4 rows of 256 B as 16-byte
    // pieces; 4 of them in flight = 16 rows per step.  The four row slots are added in a fixed tree at the end.
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    __shared__ double red[4][2][EC];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int sub = lane >> 4, c4 = lane & 15;
    const f32x4 mean1 = *reinterpret_cast<const f32x4 *>(stats + ST_MEAN1 * EC + 4 * c4);
    const f32x4 rstd1 = *reinterpret_cast<const f32x4 *>(stats + ST_RSTD1 * EC + 4 * c4);
    double s[4] = {0.0, 0.0, 0.0, 0.0}, sx[4] = {0.0, 0.0, 0.0, 0.0};
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    // a wave owns SEG_TPW consecutive targets = one contiguous run of CSR rows; their row pointers come in with one load
    constexpr int SEG_TPW = 8;
    const int nchunk = (BN + SEG_TPW - 1) / SEG_TPW;
    // rev: the clouds written LAST by the heavy kernel are read FIRST (their rows are the ones still in the memory-side cache)
    for (int c_ = blockIdx.x * 4 + wv; c_ < nchunk; c_ += gridDim.x * 4) {
        const int ch = rev ? nchunk - 1 - c_ : c_;
        const int t0 = ch * SEG_TPW;
        const int myptr = rowptr[min(t0 + min(lane, SEG_TPW), BN)];
#pragma unroll 2
        for (int q = 0; q < SEG_TPW; ++q) {
            const int tm = t0 + q;
            if (tm >= BN) break;
            const int p0 = __shfl(myptr, q, 64), p1 = __shfl(myptr, q + 1, 64);
            const size_t o = (size_t)tm * EC + 4 * c4;
            const f32x4 gq = *reinterpret_cast<const f32x4 *>(GQ + o), pv = *reinterpret_cast<const f32x4 *>(P + o),
                        qv = *reinterpret_cast<const f32x4 *>(Q + o);
            f32x4 acc = zero;
            for (int p = p0 + sub; p < p1; p += 32) {          // 32 rows of 256 B in flight per wave and step
                f32x4 v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    v[u] = (p + 4 * u < p1) ? __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(dyb + (size_t)(p + 4 * u) * EC + 4 * c4)) : zero;
                acc += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float t = acc[e];
                t += __shfl_xor(t, 16, 64);
                t += __shfl_xor(t, 32, 64);
                acc[e] = t;
            }
            if (sub == 0) {
                *reinterpret_cast<f32x4 *>(GP + o) = acc;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    s[e] += (double)gq[e];
                    sx[e] += (double)(pv[e] * rstd1[e] * acc[e]) + (double)((qv[e] - mean1[e]) * rstd1[e] * gq[e]);
                }
            }
        }
    }
    if (sub == 0) {
#pragma unroll
        for (int e = 0; e < 4; ++e) { red[wv][0][4 * c4 + e] = s[e]; red[wv][1][4 * c4 + e] = sx[e]; }
    }
    __syncthreads();
    if (threadIdx.x < 128) {
        const int which = threadIdx.x >> 6;
        bn_acc_add(partial, blockIdx.x, which, lane,
                   red[0][which][lane] + red[1][which][lane] + red[2][which][lane] + red[3][which][lane]);
    }
    }   // fp32 rows
}

// B5: BN1 backward reductions: sum_n GQ (= sum_e dy1) and
//     sum_e dy1*xhat1 = sum_m P''[m] GP[m] + sum_n Q''[n] GQ[n].   partial [grid][2][64] doubles
__global__ void __launch_bounds__(256)
ec_bwd_bn1_reduce_kernel(const float *__restrict__ P, const float *__restrict__ Q,
                         const float *__restrict__ GP, const float *__restrict__ GQ,
                         const float *__restrict__ stats, int BN, const BnAcc partial)
{
    __shared__ double red[4][2][EC];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const float mean1 = stats[ST_MEAN1 * EC + lane], rstd1 = stats[ST_RSTD1 * EC + lane];
    double s = 0.0, sx = 0.0;
    const int stride = gridDim.x * 4;
    for (int pt0 = blockIdx.x * 4 + wv; pt0 < BN; pt0 += 4 * stride) {      // 16 row loads in flight per wave
        float gq[4], gp[4], pv[4], qv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const size_t o = (size_t)min(pt0 + u * stride, BN - 1) * EC + lane;
            gq[u] = GQ[o]; gp[u] = GP[o]; pv[u] = P[o]; qv[u] = Q[o];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (pt0 + u * stride < BN) {
                s += (double)gq[u];
                sx += (double)(pv[u] * rstd1 * gp[u]) + (double)((qv[u] - mean1) * rstd1 * gq[u]);
            }
    }
    red[wv][0][lane] = s;
    red[wv][1][lane] = sx;
    __syncthreads();
    if (threadIdx.x < 128) {
        const int which = threadIdx.x >> 6;
        bn_acc_add(partial, blockIdx.x, which, lane,
                   red[0][which][lane] + red[1][which][lane] + red[2][which][lane] + red[3][which][lane]);
    }
}
struct Bn1Fin {
    const double *red, *red_local;    // [2][64] sums over the whole (possibly multi-rank) batch / over this rank
    double count;
    float *dbn1_w, *dbn1_b;
};

// B6: dP, dQ per point (BN1 backward, elementwise), then on the matrix pipe
//        dx[c][p]  = sum_o Wa[o][c] dP[p][o] + Wd[o][c] dQ[p][o]          (K = 128: dP half | dQ half)
//        dWa[o][c] += sum_p dP[p][o] x[c][p],  dWd[o][c] += sum_p dQ[p][o] x[c][p]   (K = 64 points)
// A workgroup walks tiles of 64 points (grid-stride) and keeps its share of the weight-gradient
// tiles in MFMA accumulators across tiles: one partial per workgroup.  MFMA tile tasks are dealt
// round-robin to the 4 waves.  LDS rows are padded to 68 floats (conflict-free b128 / b32 reads).
// partial: [gridDim.x][2][64][CP]  (CP = C rounded up to 32)
constexpr int PQ_LD = EC + 4;
constexpr int PQ_WAVES = 8, PQ_THREADS = 64 * PQ_WAVES;      // one workgroup per CU (86 KB of LDS): 8 waves = 2 per SIMD
constexpr int PQ_NX = 128 * 64 / PQ_THREADS, PQ_NP = 64 / PQ_WAVES, PQ_WT = 16 / PQ_WAVES;
__global__ void __launch_bounds__(PQ_THREADS)
ec_bwd_pq_kernel(const float *__restrict__ x, const float *__restrict__ w1,
                 const float *__restrict__ P, const float *__restrict__ Q,
                 const float *__restrict__ SP, const float *__restrict__ SQ, const float *__restrict__ indeg,
                 const float *__restrict__ GP, const float *__restrict__ GQ,
                 const float *__restrict__ stats, const Bn1Fin fin,
                 int C, int N, int k, int tiles_per_b, int ntiles, float *__restrict__ dx, float *__restrict__ partial)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int CT = (C + 31) >> 5, CP = CT * 32;
    const int XLD = CP + 1;
    float *dPt = smem;                          // [64 pts][PQ_LD]
    float *dQt = dPt + 64 * PQ_LD;              // [64 pts][PQ_LD]
    float *wa = dQt + 64 * PQ_LD;               // [CP][PQ_LD]   wa[c][o] = Wa[o][c]
    float *wd = wa + CP * PQ_LD;                // [CP][PQ_LD]
    float *xT = wd + CP * PQ_LD;                // [64 pts][XLD] xT[p][c]
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int i = lane & 31, h = lane >> 5;
    const int co = tid & 63;
    for (int e = tid; e < CP * EC; e += PQ_THREADS) {
        const int c = e >> 6, o = e & 63;
        const float a = (c < C) ? w1[(size_t)o * 2 * C + c] : 0.f;
        const float bb = (c < C) ? w1[(size_t)o * 2 * C + C + c] : 0.f;
        wa[c * PQ_LD + o] = a;
        wd[c * PQ_LD + o] = bb - a;
    }
    const float A1 = stats[ST_SCALE1 * EC + co], mean1 = stats[ST_MEAN1 * EC + co], rstd1 = stats[ST_RSTD1 * EC + co];
    // BN1-backward means of dy1 and dy1 * xhat1 (block 0 also stores this rank's dgamma1 / dbeta1)
    const float mu_dy = (float)(fin.red[co] / fin.count), mu_dyx = (float)(fin.red[EC + co] / fin.count);
    if (blockIdx.x == 0 && tid < EC) {
        fin.dbn1_b[tid] = (float)fin.red_local[tid];
        fin.dbn1_w[tid] = (float)fin.red_local[EC + tid];
    }
    const float kf = (float)k;

    // weight-gradient tasks: (which in {dP,dQ}) x (o tile in {0,1}) x (c tile < CT); wave w owns tasks w, w+8, ...
    const int n_wtasks = 4 * CT;
    f32x16 wacc[PQ_WT];                          // CT <= 4  ->  at most 2 tasks per wave
#pragma unroll
    for (int u = 0; u < PQ_WT; ++u)
#pragma unroll
        for (int r = 0; r < 16; ++r) wacc[u][r] = 0.f;

    // Software pipeline over the tiles (the LDS footprint allows one workgroup = one wave per SIMD per CU, so nothing
    // else hides the global latency): the inputs of tile t+1 are loaded into registers before the MFMA phase of tile t
    // and turned into the LDS images of tile t+1 after it.
    float rx[PQ_NX];                             // x tile elements e = tid + PQ_THREADS j  (c = e >> 6, p = e & 63), CP <= 128
    float rP[PQ_NP], rQ[PQ_NP], rGP[PQ_NP], rGQ[PQ_NP], rSP[PQ_NP], rSQ[PQ_NP], rdeg[PQ_NP];      // points p = wv + PQ_WAVES j, channel co
    auto fetch = [&](int tile) {
        const int b = tile / tiles_per_b, n0 = (tile - b * tiles_per_b) * 64;
        const float *xb = x + (size_t)b * C * N;
#pragma unroll
        for (int j = 0; j < PQ_NX; ++j) {
            const int e = tid + PQ_THREADS * j, c = e >> 6, p = e & 63;
            rx[j] = (e < CP * 64 && c < C && n0 + p < N) ? xb[(size_t)c * N + n0 + p] : 0.f;
        }
#pragma unroll
        for (int j = 0; j < PQ_NP; ++j) {
            const int n = n0 + wv + PQ_WAVES * j;
            const bool ok = n < N;
            const size_t o = ((size_t)b * N + (ok ? n : 0)) * EC + co;
            rdeg[j] = ok ? indeg[(size_t)b * N + n] : 0.f;
            rP[j] = P[o]; rQ[j] = Q[o]; rGP[j] = GP[o]; rGQ[j] = GQ[o]; rSP[j] = SP[o]; rSQ[j] = SQ[o];
        }
    };
    auto stage = [&](int tile) {
        const int b = tile / tiles_per_b, n0 = (tile - b * tiles_per_b) * 64;
#pragma unroll
        for (int j = 0; j < PQ_NX; ++j) {
            const int e = tid + PQ_THREADS * j, c = e >> 6, p = e & 63;
            if (e < CP * 64) xT[p * XLD + c] = rx[j];
        }
#pragma unroll
        for (int j = 0; j < PQ_NP; ++j) {
            const int p = wv + PQ_WAVES * j;
            const bool ok = n0 + p < N;
            const float dq = A1 * (rGQ[j] - kf * mu_dy - mu_dyx * rstd1 * (rSP[j] + kf * (rQ[j] - mean1)));
            const float dp = A1 * (rGP[j] - rdeg[j] * mu_dy - mu_dyx * rstd1 * (rdeg[j] * (rP[j] - mean1) + rSQ[j]));
            dPt[p * PQ_LD + co] = ok ? dp : 0.f;
            dQt[p * PQ_LD + co] = ok ? dq : 0.f;
        }
    };
    if ((int)blockIdx.x < ntiles) {
        fetch(blockIdx.x);
        stage(blockIdx.x);
    }
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int b = tile / tiles_per_b, n0 = (tile - b * tiles_per_b) * 64;
        const int next = tile + gridDim.x;
        __syncthreads();                         // the LDS images of this tile are complete
        if (next < ntiles) fetch(next);          // in flight during the MFMA phase below
        // ---- dx tiles: (c tile rt < CT) x (point tile ct < 2); K = 128 (lane half h = 0: dP part, 1: dQ part)
        for (int task = wv; task < 2 * CT; task += PQ_WAVES) {
            const int rt = task >> 1, ct = task & 1;
            const float *arow = (h ? wd : wa) + (rt * 32 + i) * PQ_LD;
            const float *brow = (h ? dQt : dPt) + (ct * 32 + i) * PQ_LD;
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const float4 a4 = *reinterpret_cast<const float4 *>(arow + 4 * v);
                const float4 b4 = *reinterpret_cast<const float4 *>(brow + 4 * v);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.x, b4.x, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.y, b4.y, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.z, b4.z, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.w, b4.w, acc, 0, 0, 0);
            }
            // C/D layout: column = point ct*32+i, row = channel rt*32 + drow(r,h)
            const int n = n0 + ct * 32 + i;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int c = rt * 32 + drow(r, h);
                if (c < C && n < N) dx[(size_t)b * C * N + (size_t)c * N + n] = acc[r];
            }
        }
        // ---- weight-gradient tiles, accumulated across point tiles
#pragma unroll
        for (int u = 0; u < PQ_WT; ++u) {
            const int task = wv + PQ_WAVES * u;
            if (task < n_wtasks) {
                const int which = task / (2 * CT), rem = task - which * 2 * CT;
                const int ot = rem / CT, cc = rem - ot * CT;
                const float *src = which ? dQt : dPt;
#pragma unroll 8
                for (int s = 0; s < 32; ++s) {
                    const float av = src[(2 * s + h) * PQ_LD + ot * 32 + i];      // A[i = o][k = point]
                    const float bv = xT[(2 * s + h) * XLD + cc * 32 + i];         // B[k = point][j = c]
                    wacc[u] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, wacc[u], 0, 0, 0);
                }
            }
        }
        __syncthreads();                     // every wave has finished reading this tile's LDS images
        if (next < ntiles) stage(next);
    }
    // one partial per workgroup: [2][64][CP]
    float *out = partial + (size_t)blockIdx.x * 2 * EC * CP;
#pragma unroll
    for (int u = 0; u < PQ_WT; ++u) {
        const int task = wv + PQ_WAVES * u;
        if (task < n_wtasks) {
            const int which = task / (2 * CT), rem = task - which * 2 * CT;
            const int ot = rem / CT, cc = rem - ot * CT;
#pragma unroll
            for (int r = 0; r < 16; ++r)
                out[((size_t)which * EC + ot * 32 + drow(r, h)) * CP + cc * 32 + i] = wacc[u][r];
        }
    }
}
// B7: dW1[:, :C] = dWa - dWd ; dW1[:, C:] = dWd, straight from the workgroup partials [nparts][2][64][CP] (summed in double).
// block = 64 outputs x 16 part-groups.  The blocks past the dW1 outputs sum the heavy backward's dW2 partials
// [nparts2][64 x 64] the same way (one launch for both weight gradients; the partials sit in the workspace since phase 1).
__global__ void __launch_bounds__(1024)
ec_bwd_dw_kernel(const float *__restrict__ partial, int nparts, int C, int CP, float *__restrict__ dw1,
                 const float *__restrict__ partial2, int nparts2, float *__restrict__ dw2)
{
    __shared__ double red[16][2][64];
    const int el = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const int nb1 = (EC * C + 63) / 64;
    if ((int)blockIdx.x >= nb1) {
        const int e2 = ((int)blockIdx.x - nb1) * 64 + el;     // < 64 x 64 by the grid
        double s2 = 0.0;
        for (int p = grp; p < nparts2; p += 16) s2 += (double)partial2[(size_t)p * (EC * EC) + e2];
        red[grp][0][el] = s2;
        __syncthreads();
        if (grp == 0) {
            double t = 0.0;
#pragma unroll
            for (int g = 0; g < 16; ++g) t += red[g][0][el];
            dw2[e2] = (float)t;
        }
        return;
    }
    const int e = blockIdx.x * 64 + el;                      // (o, c) pair
    const int o = e / C, c = e - o * C;
    double sa = 0.0, sd = 0.0;
    if (e < EC * C) {
        const size_t width = 2 * (size_t)EC * CP;
        for (int p = grp; p < nparts; p += 16) {
            sa += (double)partial[p * width + (size_t)o * CP + c];
            sd += (double)partial[p * width + (size_t)(EC + o) * CP + c];
        }
    }
    red[grp][0][el] = sa;
    red[grp][1][el] = sd;
    __syncthreads();
    if (grp == 0 && e < EC * C) {
        double a = 0.0, d = 0.0;
#pragma unroll
        for (int g = 0; g < 16; ++g) { a += red[g][0][el]; d += red[g][1][el]; }
        dw1[(size_t)o * 2 * C + c] = (float)(a - d);
        dw1[(size_t)o * 2 * C + C + c] = (float)d;
    }
}

// ---------------------------------------------------------------------------
// host side: workspace plan + launch sequences
// ---------------------------------------------------------------------------
struct Plan {
    int g_heavy;        // blocks of the grid-stride heavy kernels
    int g_pq;           // workgroups of ec_bwd_pq_kernel (one weight-gradient partial each)
    int g_mfma;         // blocks of the forward MFMA kernel (4 waves each, a wave walks super-tiles of 8 points)
    int g_bwd;          // blocks of the heavy backward kernel (4 waves each: one per SIMD, one block per CU)
    int tiles;          // ceil(N/64) * B
    size_t off_part, off_hmax, off_jmax, off_Pp, off_Qp;   // forward (Pp/Qp also backward)
    size_t off_dred;                                                     // reduced partials (doubles)
    size_t off_acc;                                                      // 4 x 64 fp64 atomic accumulators (BN sums)
    size_t off_g, off_GP, off_GQ, off_coef, off_bpart, off_mu, off_wpart;   // backward
    size_t off_cnt, off_slot, off_ovf;                                       // reverse graph (cnt[BN] | ovf_cnt)
    size_t off_csrsrc, off_rangecnt, off_dyb;                                              // CSR mode: sources in CSR order; per-edge dy1 rows
    bool csr_ok;                                                             // the shape fits csr_place_kernel's LDS tables
    size_t part_rows;                                                        // rows of the ordered-reduction table at off_part
    size_t total;
};

static Plan make_plan(int B, int C, int N, int k)
{
    Plan p;
    const size_t BN = (size_t)B * N;
    p.tiles = ((N + 63) / 64) * B;
    int g = (int)((BN + 7) / 8);
    if (g > 1024) g = 1024;
    if (g < 1) g = 1;
    p.g_heavy = g;
    p.g_pq = p.tiles < 256 ? p.tiles : 256;
    {
        int gm = (int)((BN + 31) / 32);          // one super-tile per wave at least
        if (gm > 512) gm = 512;                  // 2 resident workgroups per CU
        if (gm < 1) gm = 1;
        p.g_mfma = gm;
        p.g_bwd = (gm + 1) / 2;          // one block per CU (LDS)
    }
    int maxparts = p.tiles > g ? p.tiles : g;
    if (p.g_mfma > maxparts) maxparts = p.g_mfma;
    size_t o = 0;
    auto take = [&](size_t bytes) { size_t r = o; o = align_up(o + bytes, 256); return r; };
    {   // ordered BN reductions: one [2][64] row per producer block (the largest grid is the segment-sum kernel's) + level-1 sums
        size_t rows = (size_t)maxparts > 2048 ? (size_t)maxparts : 2048;
        const size_t prep = (size_t)(((N + 63) / 64 + 3) / 4) * B;
        if (prep > rows) rows = prep;
        p.off_part = take(sizeof(double) * 2 * EC * (rows + BNR_GROUPS));
        p.part_rows = rows;
    }
    {
        size_t w = BWD_PART;
        const size_t CPp = (size_t)((C + 31) / 32) * 32;
        if ((size_t)2 * EC * CPp > w) w = (size_t)2 * EC * CPp;
        p.off_dred = take(sizeof(double) * w);
    }
    p.off_acc = take(sizeof(double) * 8 * EC);
    p.off_hmax = take(sizeof(float) * BN * EC);
    p.off_jmax = take(BN * EC);
    p.off_Pp = take(sizeof(float) * BN * EC);
    p.off_Qp = take(sizeof(float) * BN * EC);
    p.off_g = take(sizeof(float) * BN * EC);
    p.off_GP = take(sizeof(float) * BN * EC);
    p.off_GQ = take(sizeof(float) * BN * EC);
    p.off_coef = take(sizeof(float) * CF_TOTAL);
    p.off_bpart = take(sizeof(float) * (size_t)BWD_PART * p.g_mfma);
    p.off_mu = take(sizeof(float) * 2 * EC);
    p.off_cnt = take(sizeof(int) * (BN + 4));
    p.off_slot = take(sizeof(int) * BN * (size_t)REV_CAP);
    p.off_ovf = take(sizeof(int2) * BN * (size_t)k);
    p.off_wpart = take(sizeof(float) * 2 * EC * (size_t)(((C + 31) / 32) * 32) * p.g_pq);
    // N k < 65536 (strictly): the packed u16 prefixes / cursors of csr_place_kernel hold up to the largest in-degree, which a
    // caller's idx with repeated neighbours can push to N k (every edge of the cloud into one target)
    p.csr_ok = (size_t)N * k < 65536 && csr_lds_bytes(N, k) <= 160 * 1024 && (BN * k + 4096) < (size_t)(1u << 24);
    p.off_csrsrc = take(p.csr_ok ? sizeof(int) * BN * k : 0);
    p.off_rangecnt = take(p.csr_ok ? sizeof(unsigned short) * (size_t)csr_ranges(N) * BN : 0);
    p.off_dyb = take(p.csr_ok ? sizeof(float) * EC * (BN * k + (size_t)p.g_bwd * BWD_MAX_WAVES) : 0);
    p.total = o;
    return p;
}

static int check_args(const gfs3d_edgeconv_args *a, const char *who)
{
    GFS3D_REQUIRE(a != nullptr, "%s: args is NULL", who);
    GFS3D_REQUIRE(a->B >= 1 && a->N >= 1 && a->C >= 1 && a->C <= 128, "%s: bad shape B=%d C=%d N=%d (need C<=128)", who, a->B, a->C, a->N);
    GFS3D_REQUIRE(a->k >= 1 && a->k <= GFS3D_MAX_K && a->k <= a->N, "%s: need 1 <= k <= min(N,%d), got %d", who, GFS3D_MAX_K, a->k);
    GFS3D_REQUIRE((long long)a->B * a->N < (1ll << 31) / EC, "%s: B*N too large", who);
    GFS3D_REQUIRE(a->B <= 65535, "%s: B exceeds grid.y", who);
    GFS3D_REQUIRE(a->x && a->idx && a->w1 && a->bn1_w && a->bn1_b && a->w2 && a->bn2_w && a->bn2_b, "%s: NULL input", who);
    GFS3D_REQUIRE(a->P && a->Q && a->stats, "%s: NULL P/Q/stats buffer", who);
    GFS3D_REQUIRE(a->workspace != nullptr, "%s: NULL workspace", who);
    GFS3D_REQUIRE(a->compute == GFS3D_COMPUTE_F32 || a->compute == GFS3D_COMPUTE_BF16 || a->compute == GFS3D_COMPUTE_F16,
                  "%s: compute must be one of GFS3D_COMPUTE_*", who);
    const Plan p = make_plan(a->B, a->C, a->N, a->k);
    if (a->workspace_bytes < p.total) {
        set_error("%s: workspace %zu < required %zu", who, a->workspace_bytes, p.total);
        return GFS3D_ERR_WORKSPACE;
    }
    return GFS3D_OK;
}

}  // namespace gfs3d

extern "C" size_t gfs3d_edgeconv_workspace_bytes(int B, int C, int N, int k)
{
    if (B < 1 || C < 1 || N < 1 || k < 1) return 0;
    return gfs3d::make_plan(B, C, N, k).total;
}

namespace gfs3d {
// after a producer launched with `nblocks` blocks: in ordered mode add its partial rows into acc[0:128] in a fixed order
static int bn_finish(const BnAcc &ba, int nblocks, hipStream_t st, double *mid)
{
    if (!ba.rows) return GFS3D_OK;
    hipLaunchKernelGGL(bn_reduce_rows_kernel, dim3(BNR_GROUPS), dim3(1024), 0, st, ba.rows, nblocks, mid);
    GFS3D_LAUNCH_CHECK("bn_reduce_rows_kernel");
    hipLaunchKernelGGL(bn_reduce_final_kernel, dim3(1), dim3(128), 0, st, mid, ba.acc);
    GFS3D_LAUNCH_CHECK("bn_reduce_final_kernel");
    return GFS3D_OK;
}
// accumulator block (fp64): [0:128] stage-A sums, [128:256] stage-B sums, [256:384] / [384:512] this
// rank's copies of A / B taken before the caller all-reduces A / B across ranks (synchronised BN)
constexpr int ACC_DOUBLES = 8 * EC;

// The reverse-graph kernels (atomic- and latency-bound, few registers, no LDS) only need idx and Q,
// so the training forward builds SQ/indeg for the backward on a side stream, forked once P and Q
// exist and joined at the end of phase 2.  Measured at B=32: +2 % step throughput forked here,
// nothing forked before ec_main_fwd_kernel, and beside ec_main_bwd_kernel the contention on the
// atomic path costs that kernel half of what the overlap hides.
struct SideLane {
    hipStream_t stream = nullptr;
    hipEvent_t fork = nullptr, fork2 = nullptr, join = nullptr;      // fork: idx is there (graph build); fork2: Q is there (gather)
    bool ok = false;
    bool pending = false;      // forked work the caller's stream has not joined yet
};

// One lane per (device, caller stream): two host threads driving different streams never share a fork / join pair.
static std::mutex g_lane_mutex;
static SideLane *side_lane(hipStream_t caller)
{
    static std::map<std::pair<int, hipStream_t>, SideLane> lanes;
    static const bool enabled = [] { const char *e = getenv("GFS3D_EC_OVERLAP"); return !(e && e[0] == '0'); }();
    if (!enabled) return nullptr;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    std::lock_guard<std::mutex> lk(g_lane_mutex);
    auto it = lanes.find({dev, caller});
    if (it == lanes.end()) {
        SideLane l;
        l.ok = hipStreamCreateWithFlags(&l.stream, hipStreamNonBlocking) == hipSuccess &&
               hipEventCreateWithFlags(&l.fork, hipEventDisableTiming) == hipSuccess &&
               hipEventCreateWithFlags(&l.fork2, hipEventDisableTiming) == hipSuccess &&
               hipEventCreateWithFlags(&l.join, hipEventDisableTiming) == hipSuccess;
        it = lanes.emplace(std::make_pair(dev, caller), l).first;
    }
    return it->second.ok ? &it->second : nullptr;
}

// Make `st` wait for whatever its lane still has in flight (end of the forward, and every error return after the fork:
// the side stream writes SQ / indeg / the reverse-graph scratch, which the caller may free or reuse once we return).
static int side_join(hipStream_t st, const char *who)
{
    SideLane *lane = side_lane(st);
    if (!lane || !lane->pending) return GFS3D_OK;
    lane->pending = false;
    if (hipEventRecord(lane->join, lane->stream) != hipSuccess || hipStreamWaitEvent(st, lane->join, 0) != hipSuccess) {
        set_error("%s: join of the side stream failed", who);
        return GFS3D_ERR_HIP;
    }
    return GFS3D_OK;
}

// part 0: the graph itself (needs idx only); part 1: SQ / indeg gathered over it (needs Q)
static int reverse_graph(const gfs3d_edgeconv_args *a, const Plan &p, hipStream_t rs, const char *who, int part)
{
    char *ws = static_cast<char *>(a->workspace);
    const int BN = a->B * a->N;
    if (p.csr_ok && a->csr_rowptr && a->csr_pos) {           // CSR form: fixed in-edge order, no float atomics anywhere
        int *csr_src = reinterpret_cast<int *>(ws + p.off_csrsrc);
        unsigned short *rangecnt = reinterpret_cast<unsigned short *>(ws + p.off_rangecnt);
        const size_t lds = csr_lds_bytes(a->N, a->k);
        const dim3 grid((unsigned)csr_ranges(a->N), (unsigned)a->B);
        if (part == 0) {
        GFS3D_ENSURE_LDS(csr_place_kernel, lds, who);
        hipLaunchKernelGGL(csr_count_kernel, grid, dim3(CSR_THREADS), sizeof(unsigned) * (size_t)a->N, rs, a->idx, a->N, a->k, rangecnt);
        GFS3D_LAUNCH_CHECK("csr_count_kernel");
        hipLaunchKernelGGL(csr_place_kernel, grid, dim3(CSR_THREADS), lds, rs, a->idx, rangecnt, a->B, a->N, a->k, a->csr_rowptr, csr_src, a->csr_pos);
        GFS3D_LAUNCH_CHECK("csr_place_kernel");
        return GFS3D_OK;
        }
        // (512 long-lived workgroups beside the main stream's kernels instead of 2048: c2 step -0.7 %, A/B on one box)
        // ... at B = 64 the lane (count + place + gather) outlasts the main stream's forward of the layer and the join waits for it:
        // there 2048 workgroups (c5 step 8.13 -> 7.95 ms; at B = 32 no difference beyond noise)
        static const int cg_env = [] { const char *e = getenv("GFS3D_EC_CSRGATHER_GRID"); return e ? atoi(e) : 0; }();       // A-B only
        const int cg = cg_env > 0 ? cg_env : (BN > 98304 ? 2048 : 512);
        hipLaunchKernelGGL(csr_gather_kernel, dim3((unsigned)min((BN + 31) / 32, cg)), dim3(256), 0, rs, a->Q, a->csr_rowptr, csr_src, BN, a->N, a->SQ, a->indeg);
        GFS3D_LAUNCH_CHECK("csr_gather_kernel");
        return GFS3D_OK;
    }
    int *cnt = reinterpret_cast<int *>(ws + p.off_cnt), *ovf_cnt = cnt + BN;
    int *slot = reinterpret_cast<int *>(ws + p.off_slot);
    int2 *ovf = reinterpret_cast<int2 *>(ws + p.off_ovf);
    const size_t nedges = (size_t)BN * a->k;
    if (part == 1) {
        hipLaunchKernelGGL(rev_gather_kernel, dim3(p.g_heavy), dim3(256), 0, rs, a->Q, cnt, slot, BN, a->SQ, a->indeg);
        GFS3D_LAUNCH_CHECK("rev_gather_kernel");
        hipLaunchKernelGGL(rev_overflow_kernel, dim3(64), dim3(256), 0, rs, a->Q, ovf_cnt, ovf, a->SQ);
        GFS3D_LAUNCH_CHECK("rev_overflow_kernel");
        return GFS3D_OK;
    }
    if (a->N <= 8192) {                    // per-block table in LDS; only the overflow counter needs zeroing
        GFS3D_ENSURE_LDS(rev_block_kernel, RB_LDS, who);
        if (hipMemsetAsync(ovf_cnt, 0, sizeof(int) * 4, rs) != hipSuccess) {
            set_error("%s: hipMemsetAsync failed", who);
            return GFS3D_ERR_HIP;
        }
        hipLaunchKernelGGL(rev_block_kernel, dim3((a->N + RB_TARGETS - 1) / RB_TARGETS, a->B), dim3(1024), RB_LDS, rs,
                           a->idx, a->N, a->k, cnt, slot, ovf_cnt, ovf);
        GFS3D_LAUNCH_CHECK("rev_block_kernel");
    } else {
        if (hipMemsetAsync(cnt, 0, sizeof(int) * ((size_t)BN + 4), rs) != hipSuccess) {
            set_error("%s: hipMemsetAsync failed", who);
            return GFS3D_ERR_HIP;
        }
        hipLaunchKernelGGL(rev_slot_kernel, dim3((unsigned)((nedges + 255) / 256)), dim3(256), 0, rs, a->idx, nedges, a->N, a->k,
                           cnt, slot, ovf_cnt, ovf);
        GFS3D_LAUNCH_CHECK("rev_slot_kernel");
    }
    return GFS3D_OK;
}

static int forward_phase(const gfs3d_edgeconv_args *a, int phase, double *acc_user, double stat_count, hipStream_t st)
{
    const char *who = "gfs3d_edgeconv_forward_f32";
    const Plan p = make_plan(a->B, a->C, a->N, a->k);
    char *ws = static_cast<char *>(a->workspace);
    const int BN = a->B * a->N;
    const double count = stat_count > 0.0 ? stat_count : (double)BN * a->k;
    // winning pre-BN2 value / neighbour slot per (point, channel): training writes them where the backward reads them (the
    // epilogue used to copy them there: 21 MB of stores per layer)
    float *hmax = a->training ? a->hsel : reinterpret_cast<float *>(ws + p.off_hmax);
    uint8_t *jmax = a->training ? a->jsel : reinterpret_cast<uint8_t *>(ws + p.off_jmax);
    double *acc = acc_user ? acc_user : reinterpret_cast<double *>(ws + p.off_acc);
    double *acc1 = acc, *acc2 = acc + 2 * EC;
    double *rows = g_deterministic ? reinterpret_cast<double *>(ws + p.off_part) : nullptr;      // ordered BN reductions
    double *mid = reinterpret_cast<double *>(ws + p.off_part) + 2 * EC * p.part_rows;
    float *Pp = a->Pp ? a->Pp : reinterpret_cast<float *>(ws + p.off_Pp), *Qp = a->Qp ? a->Qp : reinterpret_cast<float *>(ws + p.off_Qp);
    dim3 tgrid((a->N + 63) / 64, a->B);
    // eval mode: BatchNorm1 (running statistics) is folded into the projection kernels, P / Q are never written and
    // there is no prescale pass
    const bool fold = !a->training;
    if (phase == 0) {
        double *zero_acc = (a->training && !rows) ? acc : nullptr;          // atomics mode: the projection kernel clears the accumulators
        // SQ / indeg and the reverse graph are the backward's: built on a side stream.  The graph needs idx only, so it is forked
        // BEFORE the projections (round 5: the lane used to start after them and outlasted the main stream's forward of the layer by
        // ~12 us, which the join at the end of phase 2 waited for); the gather over it waits for Q (second event).
        SideLane *lane = a->training ? side_lane(st) : nullptr;
        hipStream_t rs = st;
        if (lane) {
            if (hipEventRecord(lane->fork, st) != hipSuccess || hipStreamWaitEvent(lane->stream, lane->fork, 0) != hipSuccess) {
                set_error("%s: fork onto the side stream failed", who);
                return GFS3D_ERR_HIP;
            }
            rs = lane->stream;
            lane->pending = true;                      // joined in phase 2, or by side_join() on any error return
#ifndef GFS3D_EC_LATE_FORK                             // (ablation builds: the graph build after the projections, as in rounds 1-4)
            if (int rc = reverse_graph(a, p, rs, who, 0)) return rc;
#endif
        }
        if (a->x_pm && a->C % 32 == 0) {
            float *wpq = reinterpret_cast<float *>(ws + p.off_wpart);        // (128, C); the backward's scratch, free here
            float *eab = fold ? reinterpret_cast<float *>(ws + p.off_bpart) : nullptr;       // (4, 64); likewise
            float *outs[2] = {fold ? Pp : a->P, fold ? Qp : a->Q};
            if (fold) {                                   // eval: [Wa; Wb - Wa] and the folded BatchNorm1 constants from a small kernel
                const BnFin fin1{nullptr, count, a->bn1_w, a->bn1_b, a->bn1_rm, a->bn1_rv, a->eps, a->momentum, 0};
                hipLaunchKernelGGL(ec_wpq_kernel, dim3((EC * a->C + 255) / 256), dim3(256), 0, st, a->w1, a->C, wpq, fin1, a->stats, eab, zero_acc);
                GFS3D_LAUNCH_CHECK("ec_wpq_kernel");
                if (int rc = pw_plain_gemm(a->x_pm, a->C, wpq, BN, a->C, 2 * EC, outs, EC, eab, eab + 2 * EC, st, who)) return rc;
            } else {                                      // training: the GEMM forms the weight rows itself and clears the accumulators
                if (int rc = pw_plain_gemm(a->x_pm, a->C, a->w1, BN, a->C, 2 * EC, outs, EC, nullptr, nullptr, st, who, 1, zero_acc,
                                           zero_acc ? 4 * EC : 0)) return rc;
            }
        } else {
            const size_t lds = sizeof(float) * (size_t)a->C * 64 * 3;
            GFS3D_ENSURE_LDS(ec_pq_kernel, lds, who);
            const BnFin fin1{nullptr, count, a->bn1_w, a->bn1_b, a->bn1_rm, a->bn1_rv, a->eps, a->momentum, 0};
            hipLaunchKernelGGL(ec_pq_kernel, tgrid, dim3(256), lds, st, a->x, a->w1, a->C, a->N, fold ? Pp : a->P, fold ? Qp : a->Q,
                               fin1, a->stats, fold ? 1 : 0, zero_acc);
            GFS3D_LAUNCH_CHECK("ec_pq_kernel");
        }
        if (a->training) {
            if (lane) {                                // Q is written: the gather may start
                if (hipEventRecord(lane->fork2, st) != hipSuccess || hipStreamWaitEvent(lane->stream, lane->fork2, 0) != hipSuccess) {
                    set_error("%s: fork onto the side stream failed", who);
                    return GFS3D_ERR_HIP;
                }
            } else if (int rc = reverse_graph(a, p, rs, who, 0)) return rc;
#ifdef GFS3D_EC_LATE_FORK
            if (lane)
                if (int rc = reverse_graph(a, p, rs, who, 0)) return rc;
#endif
            if (int rc = reverse_graph(a, p, rs, who, 1)) return rc;
            {
                ProfScope prof(4, st);
                hipLaunchKernelGGL(ec_stats1_kernel, dim3((p.g_heavy + ST1_GRID_DIV - 1) / ST1_GRID_DIV), dim3(64 * ST1_WAVES), 0, st, a->P, a->Q, a->idx, BN, a->N, a->k,
                                   a->SP, BnAcc{acc1, rows});
                GFS3D_LAUNCH_CHECK("ec_stats1_kernel");
            }
            if (int rc = bn_finish(BnAcc{acc1, rows}, (p.g_heavy + ST1_GRID_DIV - 1) / ST1_GRID_DIV, st, mid)) return rc;
        }
    } else if (phase == 1) {
        const BnFin fin1{acc1, count, a->bn1_w, a->bn1_b, a->bn1_rm, a->bn1_rv, a->eps, a->momentum, a->training};
        const size_t total = (size_t)BN * EC;
        // training: BatchNorm1 is folded into the projections INSIDE the heavy kernel (it reads the raw P / Q, finalises the
        // batch statistics in its prologue and leaves P' / Q' behind for the backward); GFS3D_EC_PRESCALE=1 keeps the separate
        // pass of rounds 1-4 (A-B only)
        static const int prescale_env = [] { const char *e = getenv("GFS3D_EC_PRESCALE"); return e ? atoi(e) : 0; }();
        const bool in_kernel = !fold && !prescale_env;
        if (!fold && !in_kernel) {
            hipLaunchKernelGGL(ec_prescale_kernel, dim3((unsigned)min((size_t)2048, (total / 4 + 255) / 256)), dim3(256), 0, st, a->P, a->Q, fin1, a->stats, total, Pp, Qp);
            GFS3D_LAUNCH_CHECK("ec_prescale_kernel");
        }
        ProfScope prof(1, st);
        const dim3 g_fwd(min(p.g_mfma + p.g_mfma / 2, (BN + 31) / 32));       // eval: 3 resident workgroups per CU
        const int prec = prec_of(a->compute, false);
        const float *Pin = in_kernel ? a->P : Pp, *Qin = in_kernel ? a->Q : Qp;
        float *Pout = in_kernel ? Pp : nullptr, *Qout = in_kernel ? Qp : nullptr;
#define GFS3D_FWD_LAUNCH(STATS_, PREC_, GRID_)                                                                                \
        hipLaunchKernelGGL((ec_main_fwd_kernel<STATS_, PREC_>), GRID_, dim3(256), 0, st, Pin, Qin, a->idx, a->w2, a->bn2_w,   \
                           BN, a->N, a->k, a->slope, hmax, jmax, BnAcc{acc2, rows}, fin1, a->stats, Pout, Qout)
        if (a->training) {
            if (prec == PREC_BF16) GFS3D_FWD_LAUNCH(true, PREC_BF16, dim3(p.g_mfma));
            else if (prec == PREC_F16) GFS3D_FWD_LAUNCH(true, PREC_F16, dim3(p.g_mfma));
            else GFS3D_FWD_LAUNCH(true, PREC_X6, dim3(p.g_mfma));
        } else {
            if (prec == PREC_BF16) GFS3D_FWD_LAUNCH(false, PREC_BF16, g_fwd);
            else if (prec == PREC_F16) GFS3D_FWD_LAUNCH(false, PREC_F16, g_fwd);
            else GFS3D_FWD_LAUNCH(false, PREC_X6, g_fwd);
        }
#undef GFS3D_FWD_LAUNCH
        GFS3D_LAUNCH_CHECK("ec_main_fwd_kernel");
        if (a->training)
            if (int rc = bn_finish(BnAcc{acc2, rows}, p.g_mfma, st, mid)) return rc;
    } else {
        const BnFin fin2{acc2, count, a->bn2_w, a->bn2_b, a->bn2_rm, a->bn2_rv, a->eps, a->momentum, a->training};
        hipLaunchKernelGGL(ec_epilogue_kernel, tgrid, dim3(256), 0, st, hmax, jmax, fin2, a->stats, a->N, a->slope,
                           a->y_cf, a->y_cf_bstride, a->y_pm, a->training ? nullptr : a->hsel, a->training ? nullptr : a->jsel, a->y_sqnorm,
                           static_cast<unsigned *>(a->y_knn_img), a->B);
        GFS3D_LAUNCH_CHECK("ec_epilogue_kernel");
        if (a->training)
            if (int rc = side_join(st, who)) return rc;
    }
    return GFS3D_OK;
}

static int backward_phase(const gfs3d_edgeconv_bwd_args *ba, int phase, double *acc_user, double stat_count, hipStream_t st)
{
    const char *who = "gfs3d_edgeconv_backward_f32";
    const gfs3d_edgeconv_args *a = &ba->f;
    const Plan p = make_plan(a->B, a->C, a->N, a->k);
    char *ws = static_cast<char *>(a->workspace);
    const int BN = a->B * a->N;
    const double count = stat_count > 0.0 ? stat_count : (double)BN * a->k;
    float *g = reinterpret_cast<float *>(ws + p.off_g);
    float *GP = reinterpret_cast<float *>(ws + p.off_GP), *GQ = reinterpret_cast<float *>(ws + p.off_GQ);
    float *bpart = reinterpret_cast<float *>(ws + p.off_bpart);
    float *wpart = reinterpret_cast<float *>(ws + p.off_wpart);
    double *acc = acc_user ? acc_user : reinterpret_cast<double *>(ws + p.off_acc);
    // monolithic call: this rank's sums ARE the global sums, no local copies needed
    double *acc1 = acc, *acc2 = acc + 2 * EC, *loc1 = acc_user ? acc + 4 * EC : acc1, *loc2 = acc_user ? acc + 6 * EC : acc2;
    double *rows = g_deterministic ? reinterpret_cast<double *>(ws + p.off_part) : nullptr;      // ordered BN reductions
    double *mid = reinterpret_cast<double *>(ws + p.off_part) + 2 * EC * p.part_rows;
    dim3 tgrid((a->N + 63) / 64, a->B);
    if (phase == 0) {
        if (!rows && hipMemsetAsync(acc, 0, sizeof(double) * 4 * EC, st) != hipSuccess) {
            set_error("%s: hipMemsetAsync failed", who);
            return GFS3D_ERR_HIP;
        }
        const dim3 pgrid(((a->N + 63) / 64 + PREP_TILES - 1) / PREP_TILES, a->B);
        hipLaunchKernelGGL(ec_bwd_prep_kernel, pgrid, dim3(256), 0, st, ba->dy_cf, ba->dy_bstride, ba->dy_pm, a->y_pm, a->hsel, a->stats,
                           a->N, a->slope, g, BnAcc{acc1, rows});
        GFS3D_LAUNCH_CHECK("ec_bwd_prep_kernel");
        if (int rc = bn_finish(BnAcc{acc1, rows}, (int)(pgrid.x * pgrid.y), st, mid)) return rc;
        if (acc_user && hipMemcpyAsync(loc1, acc1, sizeof(double) * 2 * EC, hipMemcpyDeviceToDevice, st) != hipSuccess) {
            set_error("%s: hipMemcpyAsync failed", who);
            return GFS3D_ERR_HIP;
        }
    } else if (phase == 1) {
        const CoefIn cin{acc1, loc1, count, a->stats, ba->dbn2_w, ba->dbn2_b};
        const bool csr = p.csr_ok && a->csr_rowptr && a->csr_pos;
        float *dyb = reinterpret_cast<float *>(ws + p.off_dyb);
        if (!csr && hipMemsetAsync(GP, 0, sizeof(float) * (size_t)BN * EC, st) != hipSuccess) {
            set_error("%s: hipMemsetAsync failed", who);
            return GFS3D_ERR_HIP;
        }
        float *Pp = a->Pp ? a->Pp : reinterpret_cast<float *>(ws + p.off_Pp), *Qp = a->Qp ? a->Qp : reinterpret_cast<float *>(ws + p.off_Qp);
        {
            if (!(a->Pp && a->Qp)) {                  // the forward did not leave the folded projections behind
                const size_t total = (size_t)BN * EC;
                const BnFin done{nullptr, 0.0, nullptr, nullptr, nullptr, nullptr, a->eps, a->momentum, 1};
                hipLaunchKernelGGL(ec_prescale_kernel, dim3((unsigned)min((size_t)2048, (total / 4 + 255) / 256)), dim3(256), 0, st, a->P, a->Q, done, a->stats, total, Pp, Qp);
                GFS3D_LAUNCH_CHECK("ec_prescale_kernel");
            }
            const bool reduced = prec_of(a->compute, true) != PREC_X6;           // gradients: bf16 operands in both reduced modes
#define GFS3D_BWD_LAUNCH(PREC_, CSR_)                                                                                              \
            do {                                                                                                                   \
                GFS3D_ENSURE_LDS((ec_main_bwd_kernel<PREC_, CSR_>), BwdCfg<PREC_>::LDS_BYTES, who);                                \
                ProfScope prof(2, st);                                                                                             \
                hipLaunchKernelGGL((ec_main_bwd_kernel<PREC_, CSR_>), dim3(p.g_bwd), dim3(64 * BwdCfg<PREC_>::WAVES), BwdCfg<PREC_>::LDS_BYTES, st, Pp, Qp, a->idx, \
                                   a->w2, cin, g, a->jsel, BN, a->N, a->k, a->slope, GP, GQ, bpart, a->csr_pos, dyb);              \
            } while (0)
            // GFS3D_EC_BWD_PAIR (A-B only): 0 = never the wave-pair form, 2 = the wave-pair form in the reduced modes too.  Default: pairs for the
            // fp32-accurate arithmetic, ec_main_bwd_kernel with eight independent waves per workgroup for the reduced ones (round 6)
            static const int pair_env = [] { const char *e = getenv("GFS3D_EC_BWD_PAIR"); return e ? atoi(e) : 1; }();
            if (csr && (reduced ? pair_env == 2 : pair_env != 0)) {       // CSR form as wave pairs (ec_main_bwd_pair_kernel)
                ProfScope prof(2, st);
                if (reduced) {
                    GFS3D_ENSURE_LDS(ec_main_bwd_pair_kernel<PREC_BF16>, BP_LDS_BYTES, who);
                    hipLaunchKernelGGL(ec_main_bwd_pair_kernel<PREC_BF16>, dim3(p.g_bwd), dim3(128 * BP_PAIRS), BP_LDS_BYTES, st, Pp, Qp, a->idx, a->w2,
                                       cin, g, a->jsel, BN, a->N, a->k, a->slope, GQ, bpart, a->csr_pos, dyb);
                } else {
                    GFS3D_ENSURE_LDS(ec_main_bwd_pair_kernel<PREC_X6>, BP_LDS_BYTES, who);
                    hipLaunchKernelGGL(ec_main_bwd_pair_kernel<PREC_X6>, dim3(p.g_bwd), dim3(128 * BP_PAIRS), BP_LDS_BYTES, st, Pp, Qp, a->idx, a->w2,
                                       cin, g, a->jsel, BN, a->N, a->k, a->slope, GQ, bpart, a->csr_pos, dyb);
                }
            } else if (reduced) { if (csr) GFS3D_BWD_LAUNCH(PREC_BF16, true); else GFS3D_BWD_LAUNCH(PREC_BF16, false); }
            else { if (csr) GFS3D_BWD_LAUNCH(PREC_X6, true); else GFS3D_BWD_LAUNCH(PREC_X6, false); }
#undef GFS3D_BWD_LAUNCH
            GFS3D_LAUNCH_CHECK("ec_main_bwd_kernel");
        }
        if (csr) {
            // every workgroup ends in 128 same-address fp64 atomics (the BN1-backward sums): few, long-lived workgroups
            // (2048 -> 512 at B = 32: 93 -> 85 us per launch; 256: slower again)
            // (B = 64: 2048 workgroups, c5 step 8.03 -> 7.97 ms)
            static const int sg_env = [] { const char *e = getenv("GFS3D_EC_SEGSUM_GRID"); return e ? atoi(e) : 0; }();       // A-B only
            const int sg = min((BN + 31) / 32, sg_env > 0 ? sg_env : (BN > 98304 ? 2048 : 512));
            static const int seg_rev = [] { const char *e = getenv("GFS3D_EC_SEGSUM_REV"); return e ? atoi(e) : 1; }();       // A-B only
            if (prec_of(a->compute, true) != PREC_X6)
                hipLaunchKernelGGL(ec_bwd_segsum_kernel<true>, dim3((unsigned)sg), dim3(256), 0, st, dyb, a->csr_rowptr, a->P, a->Q, GQ,
                                   a->stats, BN, GP, BnAcc{acc2, rows}, seg_rev);
            else
                hipLaunchKernelGGL(ec_bwd_segsum_kernel<false>, dim3((unsigned)sg), dim3(256), 0, st, dyb, a->csr_rowptr, a->P, a->Q, GQ,
                                   a->stats, BN, GP, BnAcc{acc2, rows}, seg_rev);
            GFS3D_LAUNCH_CHECK("ec_bwd_segsum_kernel");
            if (int rc = bn_finish(BnAcc{acc2, rows}, sg, st, mid)) return rc;
        } else {
            // every block ends in 128 same-address fp64 atomics: few, long-running blocks
            const int rg = p.g_heavy < 256 ? p.g_heavy : 256;
            hipLaunchKernelGGL(ec_bwd_bn1_reduce_kernel, dim3(rg), dim3(256), 0, st, a->P, a->Q, GP, GQ, a->stats, BN, BnAcc{acc2, rows});
            GFS3D_LAUNCH_CHECK("ec_bwd_bn1_reduce_kernel");
            if (int rc = bn_finish(BnAcc{acc2, rows}, rg, st, mid)) return rc;
        }
        if (acc_user && hipMemcpyAsync(loc2, acc2, sizeof(double) * 2 * EC, hipMemcpyDeviceToDevice, st) != hipSuccess) {
            set_error("%s: hipMemcpyAsync failed", who);
            return GFS3D_ERR_HIP;
        }
    } else {
        const Bn1Fin fin{acc2, loc2, count, ba->dbn1_w, ba->dbn1_b};
        const int CP = ((a->C + 31) / 32) * 32;
        const size_t lds = sizeof(float) * (2 * 64 * PQ_LD + 2 * (size_t)CP * PQ_LD + 64 * (size_t)(CP + 1));
        GFS3D_ENSURE_LDS(ec_bwd_pq_kernel, lds, who);
        const int tiles_per_b = (a->N + 63) / 64;
        hipLaunchKernelGGL(ec_bwd_pq_kernel, dim3(p.g_pq), dim3(PQ_THREADS), lds, st, a->x, a->w1, a->P, a->Q, a->SP, a->SQ, a->indeg,
                           GP, GQ, a->stats, fin, a->C, a->N, a->k, tiles_per_b, p.tiles, ba->dx, wpart);
        GFS3D_LAUNCH_CHECK("ec_bwd_pq_kernel");
        static_assert(BWD_PART == EC * EC, "dW2 partials are 64 x 64 per workgroup");
        hipLaunchKernelGGL(ec_bwd_dw_kernel, dim3((EC * a->C + 63) / 64 + BWD_PART / 64), dim3(1024), 0, st, wpart, p.g_pq, a->C, CP, ba->dw1,
                           bpart, p.g_bwd, ba->dw2);
        GFS3D_LAUNCH_CHECK("ec_bwd_dw_kernel");
    }
    return GFS3D_OK;
}

static int check_fwd(const gfs3d_edgeconv_args *a, const char *who)
{
    int rc = check_args(a, who);
    if (rc) return rc;
    GFS3D_REQUIRE(a->y_cf || a->y_pm, "%s: no output buffer", who);
    GFS3D_REQUIRE((a->Pp == nullptr) == (a->Qp == nullptr), "%s: Pp and Qp go together", who);
    GFS3D_REQUIRE(a->training || (a->bn1_rm && a->bn1_rv && a->bn2_rm && a->bn2_rv), "%s: eval mode needs running statistics", who);
    if (a->training)
        GFS3D_REQUIRE(a->SP && a->SQ && a->indeg && a->hsel && a->jsel && a->y_pm,
                      "%s: training mode needs SP/SQ/indeg/hsel/jsel/y_pm buffers", who);
    GFS3D_REQUIRE(!a->y_cf || a->y_cf_bstride >= (long long)EC * a->N, "%s: bad y_cf_bstride", who);
    return GFS3D_OK;
}

static int check_bwd(const gfs3d_edgeconv_bwd_args *ba, const char *who)
{
    GFS3D_REQUIRE(ba != nullptr, "%s: args is NULL", who);
    const gfs3d_edgeconv_args *a = &ba->f;
    int rc = check_args(a, who);
    if (rc) return rc;
    GFS3D_REQUIRE(a->training, "%s: only the training-mode (batch statistics) backward exists", who);
    GFS3D_REQUIRE(a->SP && a->SQ && a->indeg && a->hsel && a->jsel && a->y_pm, "%s: NULL saved buffer", who);
    GFS3D_REQUIRE((ba->dy_cf || ba->dy_pm) && ba->dx && ba->dw1 && ba->dbn1_w && ba->dbn1_b && ba->dw2 && ba->dbn2_w && ba->dbn2_b,
                  "%s: NULL gradient buffer", who);
    GFS3D_REQUIRE(ba->dy_bstride >= (long long)EC * a->N, "%s: bad dy_bstride", who);
    return GFS3D_OK;
}
}  // namespace gfs3d

extern "C" int gfs3d_set_deterministic(int on)
{
    gfs3d::g_deterministic = on != 0;
    return GFS3D_OK;
}

extern "C" int gfs3d_edgeconv_forward_f32(const gfs3d_edgeconv_args *a, void *stream)
{
    using namespace gfs3d;
    int rc = check_fwd(a, "gfs3d_edgeconv_forward_f32");
    for (int ph = 0; ph < 3 && !rc; ++ph) rc = forward_phase(a, ph, nullptr, 0.0, static_cast<hipStream_t>(stream));
    if (rc) (void)side_join(static_cast<hipStream_t>(stream), "gfs3d_edgeconv_forward_f32");      // never return with forked work un-joined
    return rc;
}

extern "C" int gfs3d_edgeconv_forward_phase_f32(const gfs3d_edgeconv_args *a, int phase, double *acc, double stat_count, void *stream)
{
    using namespace gfs3d;
    int rc = check_fwd(a, "gfs3d_edgeconv_forward_phase_f32");
    if (rc) return rc;
    GFS3D_REQUIRE(phase >= 0 && phase <= 2, "gfs3d_edgeconv_forward_phase_f32: phase must be 0, 1 or 2");
    rc = forward_phase(a, phase, acc, stat_count, static_cast<hipStream_t>(stream));
    if (rc) (void)side_join(static_cast<hipStream_t>(stream), "gfs3d_edgeconv_forward_phase_f32");
    return rc;
}

extern "C" int gfs3d_edgeconv_backward_f32(const gfs3d_edgeconv_bwd_args *ba, void *stream)
{
    using namespace gfs3d;
    int rc = check_bwd(ba, "gfs3d_edgeconv_backward_f32");
    for (int ph = 0; ph < 3 && !rc; ++ph) rc = backward_phase(ba, ph, nullptr, 0.0, static_cast<hipStream_t>(stream));
    return rc;
}

extern "C" int gfs3d_edgeconv_backward_phase_f32(const gfs3d_edgeconv_bwd_args *ba, int phase, double *acc, double stat_count, void *stream)
{
    using namespace gfs3d;
    int rc = check_bwd(ba, "gfs3d_edgeconv_backward_phase_f32");
    if (rc) return rc;
    GFS3D_REQUIRE(phase >= 0 && phase <= 2, "gfs3d_edgeconv_backward_phase_f32: phase must be 0, 1 or 2");
    return backward_phase(ba, phase, acc, stat_count, static_cast<hipStream_t>(stream));
}
