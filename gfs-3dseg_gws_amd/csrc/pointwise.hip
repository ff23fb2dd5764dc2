// K4: point-wise MLP stack  [Conv1d 1x1 -> BatchNorm1d -> LeakyReLU] x L  on the fp32 MFMA pipe.
//
// Replaces `self.conv(torch.cat(edgeconv_outputs, dim=1))` of DGCNN.forward (reference
// model/dgcnn.py:121-122, the conv1d class dgcnn.py:64-80).  The EdgeConv kernels already keep a
// point-major copy of every block output, so the stack runs entirely in point-major layout
// ((B*N, C) rows): the concatenation is a list of K-segments, no transposes, and every BatchNorm /
// LeakyReLU pass is folded into the prologue or epilogue of the neighbouring GEMM:
//
//   forward (train)   H_l = act(bn_{l-1}(H_{l-1})) W_l^T   + per-channel sum / sum^2  (fp64 atomics)
//                     y   = act(bn_L(H_L)) written channel-first (B, C_L, N)
//   backward          G_L = dy^T * act'                     + sum g, sum g h
//                     dW_l = dH_l^T act(bn(H_{l-1}))        (split-K over the points, fp32 atomics)
//                     G_{l-1} = (dH_l W_l) * act'           + sums;   dH_l = a g + b + c h  (BN backward)
//
// fp32 in / fp32 accumulate (v_mfma_f32_32x32x2_f32); 128 x {128,64} block tiles, K-step 32, operands
// staged through padded LDS and read with ds_read_b128 (K permuted identically for both operands).
#include "common.h"

namespace gfs3d {
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// ---- storage format of the stack's INTERNAL tensors in HBM (gfs3d.h: gfs3d_pointwise_args.store16) --------------------------
// fp32-accurate mode: everything fp32.  Reduced modes with store16: the pre-BatchNorm activations h[l] are kept in the forward
// operand format of the mode (ST_F16 under GFS3D_COMPUTE_F16, ST_BF16 under GFS3D_COMPUTE_BF16), the gradients g[l] in the
// workspace as bf16 (half cannot hold the gradients of a mean loss).  Accumulators, BatchNorm sums and constants stay fp32 /
// fp64; the statistics are taken from the fp32 accumulators BEFORE the rounding (the rounding error is zero-mean).
enum { ST_F32 = 0, ST_BF16 = 1, ST_F16 = 2 };
template <int ST> struct Raw4 { typedef f32x4 type; };      // four consecutive stored elements, as loaded
template <> struct Raw4<ST_BF16> { typedef uint2 type; };
template <> struct Raw4<ST_F16> { typedef uint2 type; };
template <int ST> __device__ __forceinline__ typename Raw4<ST>::type st_ld4(const void *base, size_t idx)
{
    if constexpr (ST == ST_F32) return *reinterpret_cast<const f32x4 *>(static_cast<const float *>(base) + idx);
    else return *reinterpret_cast<const uint2 *>(static_cast<const unsigned short *>(base) + idx);
}
template <int ST> __device__ __forceinline__ typename Raw4<ST>::type st_zero4()
{
    if constexpr (ST == ST_F32) { const f32x4 z = {0.f, 0.f, 0.f, 0.f}; return z; }
    else return make_uint2(0u, 0u);
}
template <int ST> __device__ __forceinline__ f32x4 st_cvt4(const typename Raw4<ST>::type &r)
{
    if constexpr (ST == ST_F32) return r;
    else if constexpr (ST == ST_BF16) {
        const f32x4 v = {__builtin_bit_cast(float, r.x << 16), __builtin_bit_cast(float, r.x & 0xffff0000u),
                         __builtin_bit_cast(float, r.y << 16), __builtin_bit_cast(float, r.y & 0xffff0000u)};
        return v;
    } else {
        const mm_f32x2 a = __builtin_convertvector(__builtin_bit_cast(mm_f16x2, r.x), mm_f32x2);
        const mm_f32x2 b = __builtin_convertvector(__builtin_bit_cast(mm_f16x2, r.y), mm_f32x2);
        const f32x4 v = {a[0], a[1], b[0], b[1]};
        return v;
    }
}
// half saturates instead of overflowing to inf (a pre-BatchNorm activation is unbounded in principle)
__device__ __forceinline__ float st_sat16(float v) { return __builtin_fminf(__builtin_fmaxf(v, -65504.f), 65504.f); }
template <int ST> __device__ __forceinline__ typename Raw4<ST>::type st_pack4(const f32x4 &v)
{
    if constexpr (ST == ST_F32) return v;
    else if constexpr (ST == ST_BF16) return make_uint2(mm_pack2<PREC_BF16>(v[0], v[1]), mm_pack2<PREC_BF16>(v[2], v[3]));
    else return make_uint2(mm_pack2<PREC_F16>(st_sat16(v[0]), st_sat16(v[1])), mm_pack2<PREC_F16>(st_sat16(v[2]), st_sat16(v[3])));
}
template <int ST> __device__ __forceinline__ void st_st4(void *base, size_t idx, const f32x4 &v)
{
    if constexpr (ST == ST_F32) *reinterpret_cast<f32x4 *>(static_cast<float *>(base) + idx) = v;
    else *reinterpret_cast<uint2 *>(static_cast<unsigned short *>(base) + idx) = st_pack4<ST>(v);
}
template <int ST> __device__ __forceinline__ float st_ld1(const void *base, size_t idx)
{
    if constexpr (ST == ST_F32) return static_cast<const float *>(base)[idx];
    else if constexpr (ST == ST_BF16) return __builtin_bit_cast(float, (unsigned)static_cast<const unsigned short *>(base)[idx] << 16);
    else return (float)static_cast<const _Float16 *>(base)[idx];
}
template <int ST> __device__ __forceinline__ void st_st1(void *base, size_t idx, float v)
{
    if constexpr (ST == ST_F32) static_cast<float *>(base)[idx] = v;
    else if constexpr (ST == ST_BF16) static_cast<unsigned short *>(base)[idx] = mm_cvt1<PREC_BF16>(v);
    else static_cast<unsigned short *>(base)[idx] = mm_cvt1<PREC_F16>(st_sat16(v));
}
// the value a consumer reads back: v rounded to the storage format (the BatchNorm-backward sums pair g with the stored h)
template <int ST> __device__ __forceinline__ f32x4 st_round4(const f32x4 &v) { return st_cvt4<ST>(st_pack4<ST>(v)); }
__host__ __device__ constexpr size_t st_bytes(int st) { return st == ST_F32 ? 4 : 2; }

constexpr int PW_BM = 128, PW_BK = 32, PW_LD = 36;      // LD 36: conflict-free ds_read_b128 of 32-float rows
enum { PRO_NONE = 0, PRO_BNACT = 1, PRO_BWD = 2 };
enum { EPI_RAW_STATS = 0, EPI_AFFINE_ACT = 1, EPI_BWD_ACT = 2, EPI_PLAIN = 3 };
enum { PST_SCALE = 0, PST_SHIFT = 1, PST_MEAN = 2, PST_RSTD = 3 };

// batch statistics -> (scale, shift, mean, rstd) and the running-statistics update of nn.BatchNorm1d, channel c
// (s1, s2 = sum h, sum h^2; h is stored without the convolution's bias b: BatchNorm(h + b) with batch statistics does not depend
// on b, only the running mean does)
__device__ __forceinline__ void pw_bn_finalize_one(int c, int C, double s1, double s2, double count, const float *gamma, const float *beta,
                                                   float *rm, float *rv, float eps, float momentum, const float *bias, float *stats)
{
    const double bc = bias ? (double)bias[c] : 0.0;
    const double mean = s1 / count;
    double var = s2 / count - mean * mean;
    if (var < 0.0) var = 0.0;
    if (rm) rm[c] = (float)((1.0 - momentum) * rm[c] + momentum * (mean + bc));
    if (rv) rv[c] = (float)((1.0 - momentum) * rv[c] + momentum * var * (count > 1.0 ? count / (count - 1.0) : 1.0));
    const float rstd = (float)(1.0 / sqrt(var + (double)eps));
    const float sc = gamma[c] * rstd;
    stats[PST_SCALE * C + c] = sc;
    stats[PST_SHIFT * C + c] = beta[c] - (float)mean * sc;
    stats[PST_MEAN * C + c] = (float)mean;
    stats[PST_RSTD * C + c] = rstd;
}
// BatchNorm backward coefficients of channel c: dh = a g + b + c h; parameter gradients.  (S1, Sgh) = (sum g, sum g h) over the
// whole batch -> coefficients; (L1, Lgh) = this rank's sums -> its parameter gradients
__device__ __forceinline__ void pw_bwd_coef_one(int c, int C, double S1, double Sgh, double L1, double Lgh, double count, const float *stats,
                                                float *coef, float *dgamma, float *dbeta)
{
    const double a = stats[PST_SCALE * C + c], mean = stats[PST_MEAN * C + c], rstd = stats[PST_RSTD * C + c];
    const double S2 = rstd * (Sgh - mean * S1);                            // sum g xhat
    const double cc = -a * rstd * S2 / count;
    coef[c] = (float)a;
    coef[C + c] = (float)(-a * S1 / count - cc * mean);
    coef[2 * C + c] = (float)cc;
    dgamma[c] = (float)(rstd * (Lgh - mean * L1));
    dbeta[c] = (float)L1;
}
// The LAST workgroup to add its column sums does what pw_bn_finalize_kernel / pw_bwd_coef_kernel would do in a launch of their
// own (~5 us each on an otherwise idle stream): a ticket counter (cleared with the accumulators), taken once the workgroup's adds
// have been performed, the sums read back through device-scope atomic loads.  Monolithic, atomics-mode calls only (one rank's sums are the
// batch's sums; the ordered mode and the phased entry points of synchronised BatchNorm keep the separate kernels).
struct BnTail {
    unsigned *counter;          // NULL: no tail
    int kind;                   // 1: forward finalisation, 2: backward coefficients
    int C;
    double count;
    const double *acc;          // [2][C] the sums the producing kernel adds to
    const float *gamma, *beta, *bias;      // kind 1
    float *rm, *rv;
    float eps, momentum;
    float *stats;               // kind 1: out (4, C); kind 2: in
    float *coef, *dgamma, *dbeta;          // kind 2
};
__device__ __forceinline__ void bn_tail(const BnTail &tl, unsigned nblocks, int t, int nthreads, unsigned *flag)
{
    if (!tl.counter) return;
    // Only ATOMICS cross workgroups here (device-scope adds, performed at the coherence point; the last workgroup reads them back
    // with device-scope atomic loads), so "this thread's adds have been performed" is all the ticket must wait for: vmcnt(0).
    // A release fence (__threadfence) would also write back the XCD's whole L2 - the tile of h this workgroup has just stored -
    // once per workgroup: measured 3.35 -> 4.10 ms per c2 step.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                                    // ... for every thread of the workgroup, before the ticket is taken
    if (t == 0) *flag = atomicAdd(tl.counter, 1u);
    __syncthreads();
    if (*flag != nblocks - 1u) return;
    for (int c = t; c < tl.C; c += nthreads) {
        const double s1 = __hip_atomic_load(&tl.acc[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const double s2 = __hip_atomic_load(&tl.acc[tl.C + c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (tl.kind == 1) pw_bn_finalize_one(c, tl.C, s1, s2, tl.count, tl.gamma, tl.beta, tl.rm, tl.rv, tl.eps, tl.momentum, tl.bias, tl.stats);
        else pw_bwd_coef_one(c, tl.C, s1, s2, s1, s2, tl.count, tl.stats, tl.coef, tl.dgamma, tl.dbeta);
    }
}

struct RowGemm {
    const void *a[4];         // A operand: K-segments of width a_segw, rows of stride lda   (M, K); element type: template AT
    const void *a2[4];        // PRO_BWD: second tensor (h) of the same shape; element type: template HT
    int a_segw, lda;
    const float *pa, *pb, *pc;  // prologue coefficients per K channel
    const float *w;           // (Nc, K) row-major, K contiguous
    int w_pq;                 // plain kernel only, != 0: w is an EdgeConv first convolution (64, 2 K) = [Wa | Wb]; the GEMM's 128 rows are [Wa; Wb - Wa]
    double *zero;             // plain kernel only: block 0 clears nzero doubles (BatchNorm accumulators of the caller's next passes)
    int nzero;
    const void *w3;           // bf16x3 image of w for the split kernel: [Nc][K/32][3 planes][32] bf16
    int M, K, Nc;
    float slope;
    void *out[4];             // point-major output, column segments of width o_segw (row stride o_segw); element type: template OT
    int o_segw;
    int at, ht, ot;           // ST_* of a / (a2, eh) / out: picks the instantiation (launch_rowgemm6)
    void *a16_out;            // reduced forward on an fp32 A: the operands it forms anyway, written once ((M, K) 16-bit, by the column-block-0
                              // workgroups) - the backward's weight gradient reads them instead of the fp32 rows
    float *out_cf;            // OUT_CF: (B, Nc, npts) channel-first ...
    float *out_cfs[4];        // ... or, cf_segw > 0: Nc / cf_segw tensors (B, cf_segw, npts)
    int cf_segw;
    int npts;
    const float *ea, *eb;     // epilogue scale / shift per output channel
    const void *eh;           // EPI_BWD_ACT: pre-BN activations of the produced layer (M, Nc); element type: template HT
    double *acc;              // (2, Nc) column sums (fp64 atomics) ...
    double *acc_rows;         // ... or, deterministic mode: [row block][2][Nc] partial sums, added in order by pw_colsum_reduce_kernel
    int prec;                 // PREC_* of the bf16-pipe kernel (common.h)
    BnTail tail;              // EPI_RAW_STATS / EPI_BWD_ACT in atomics mode: what the last workgroup finalises (counter NULL: nothing)
};

__device__ __forceinline__ f32x4 ldg4(const float *p) { return *reinterpret_cast<const f32x4 *>(p); }
// workgroup barrier that orders LDS traffic only: global stores / loads in flight are NOT waited for (__syncthreads() would: on
// CDNA4 vmcnt counts stores too, so the second epilogue pass waited for the first pass's stores to reach memory)
__device__ __forceinline__ void pw_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// Shared epilogue of the row GEMMs: acc[TM][TN] are 32x32 MFMA result tiles (column on the lane, rows in the 16
// registers); wave (wm, wn) owns rows wm*WM.. and columns wn*64.. of the block tile at (row0, col0).
template <int BN, int EPI, bool OUT_CF, int TM, int TN, int HT = ST_F32, int OT = ST_F32>
__device__ __forceinline__ void rowgemm_epilogue(const RowGemm &g, f32x16 (&acc)[TM][TN], float *smem, int row0, int col0, int wm,
                                                 int wn, int wv, int t, int li, int lh)
{
    constexpr int WM = TM * 32;
    if (OUT_CF) {                               // D rows = channels, lanes = points
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int pt = row0 + wm * WM + i * 32 + li;
            if (pt >= g.M) continue;
            const int b = pt / g.npts, n = pt - b * g.npts;
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int ch = col0 + wn * 64 + j * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    if (EPI == EPI_PLAIN) {
                        const int sg = ch / g.cf_segw, cc = ch - sg * g.cf_segw;
                        g.out_cfs[sg][((size_t)b * g.cf_segw + cc) * g.npts + n] = acc[i][j][r];
                    } else {
                        const float z = fmaf(acc[i][j][r], g.ea[ch], g.eb[ch]);
                        g.out_cf[((size_t)b * g.Nc + ch) * g.npts + n] = lrelu(z, g.slope);
                    }
                }
        }
        return;
    }
    if (BN == 128) {
        // Through LDS, so that the global side is 16-byte wide and coalesced over whole rows (the direct form issues 64
        // scalar loads + 64 scalar stores per lane): two passes of 64 tile rows; in pass ps the two waves with wm == ps
        // drop their 64 x 64 accumulators into a [64][132] image, then thread t handles columns 4 (t & 31) .. +3 of
        // rows (t >> 5) + 8 jj.
        constexpr int ELD = 132;
        const int c4 = (t & 31) * 4, rr = t >> 5;
        const int gcol = col0 + c4;
        const int seg = gcol / g.o_segw, ocol = gcol - seg * g.o_segw;
        void *const op = g.out[seg];
        f32x4 ea = {0.f, 0.f, 0.f, 0.f}, eb = ea;
        double ts1[4] = {0.0, 0.0, 0.0, 0.0}, ts2[4] = {0.0, 0.0, 0.0, 0.0};   // fp64: sum x^2 - mean^2 cancels when |mean| >> std
        if (EPI == EPI_AFFINE_ACT || EPI == EPI_BWD_ACT) { ea = ldg4(g.ea + gcol); eb = ldg4(g.eb + gcol); }
        // EPI_BWD_ACT with 16-bit activations: BOTH passes' rows of `eh` are requested here, in front of the LDS round trips of the
        // accumulators (32 registers) - issued inside a pass they were a serial trip to HBM per pass with nothing beside them
        constexpr bool EH_AHEAD = EPI == EPI_BWD_ACT && HT != ST_F32;
        typename Raw4<HT>::type hva[EH_AHEAD ? 2 : 1][8];
        if constexpr (EH_AHEAD) {
#pragma unroll
            for (int ps = 0; ps < 2; ++ps)
#pragma unroll
                for (int jj = 0; jj < 8; ++jj) {
                    const int row = row0 + ps * 64 + rr + 8 * jj;
                    hva[ps][jj] = row < g.M ? st_ld4<HT>(g.eh, (size_t)row * g.Nc + gcol) : st_zero4<HT>();
                }
        }
#pragma unroll
        for (int ps = 0; ps < 2; ++ps) {
            pw_lds_barrier();
            if (wm == ps) {
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
#pragma unroll
                        for (int r = 0; r < 16; ++r)
                            smem[(i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh) * ELD + wn * 64 + j * 32 + li] = acc[i][j][r];
            }
            pw_lds_barrier();
            float p1[4] = {0.f, 0.f, 0.f, 0.f}, p2[4] = {0.f, 0.f, 0.f, 0.f};     // fp32 over the 8 rows of a pass, fp64 beyond
            typename Raw4<HT>::type hv[8];
            if (EPI == EPI_BWD_ACT) {
#pragma unroll
                for (int jj = 0; jj < 8; ++jj) {
                    const int row = row0 + ps * 64 + rr + 8 * jj;
                    if constexpr (EH_AHEAD) hv[jj] = hva[ps][jj];
                    else hv[jj] = row < g.M ? st_ld4<HT>(g.eh, (size_t)row * g.Nc + gcol) : st_zero4<HT>();
                }
            }
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) {
                const int row = row0 + ps * 64 + rr + 8 * jj;
                if (row >= g.M) continue;
                f32x4 v = *reinterpret_cast<const f32x4 *>(&smem[(rr + 8 * jj) * ELD + c4]);
                f32x4 hh4 = ea;
                if (EPI == EPI_BWD_ACT) hh4 = st_cvt4<HT>(hv[jj]);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (EPI == EPI_RAW_STATS) { p1[e] += v[e]; p2[e] = fmaf(v[e], v[e], p2[e]); }
                    if (EPI == EPI_AFFINE_ACT) v[e] = lrelu(fmaf(v[e], ea[e], eb[e]), g.slope);
                    if (EPI == EPI_BWD_ACT) {
                        const float hh = hh4[e];
                        v[e] = fmaf(hh, ea[e], eb[e]) > 0.f ? v[e] : v[e] * g.slope;
                        p1[e] += v[e]; p2[e] = fmaf(v[e], hh, p2[e]);
                    }
                }
                st_st4<OT>(op, (size_t)row * g.o_segw + ocol, v);
            }
            if (EPI == EPI_RAW_STATS || EPI == EPI_BWD_ACT) {
#pragma unroll
                for (int e = 0; e < 4; ++e) { ts1[e] += (double)p1[e]; ts2[e] += (double)p2[e]; }
            }
        }
        if (EPI == EPI_RAW_STATS || EPI == EPI_BWD_ACT) {
            __syncthreads();
            double *red = reinterpret_cast<double *>(smem);      // [8 thread rows][2][128]
#pragma unroll
            for (int e = 0; e < 4; ++e) { red[(rr * 2 + 0) * 128 + c4 + e] = ts1[e]; red[(rr * 2 + 1) * 128 + c4 + e] = ts2[e]; }
            __syncthreads();
            const int which = t >> 7, col = t & 127;
            double sum = 0.0;
#pragma unroll
            for (int q = 0; q < 8; ++q) sum += red[(q * 2 + which) * 128 + col];
            if (g.acc_rows) g.acc_rows[((size_t)(row0 / PW_BM) * 2 + which) * g.Nc + col0 + col] = sum;
            else atomicAdd(&g.acc[which * g.Nc + col0 + col], sum);
            bn_tail(g.tail, (unsigned)((g.Nc / BN) * ((g.M + PW_BM - 1) / PW_BM)), t, 256, reinterpret_cast<unsigned *>(red + 8 * 2 * 128));
        }
        return;
    }
    double s1[TN], s2[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int col = col0 + wn * 64 + j * 32 + li;
        const int seg = col / g.o_segw, ocol = col - seg * g.o_segw;
        void *const op = g.out[seg];
        float ea = 0.f, eb = 0.f;
        if (EPI == EPI_AFFINE_ACT || EPI == EPI_BWD_ACT) { ea = g.ea[col]; eb = g.eb[col]; }
        s1[j] = 0.0; s2[j] = 0.0;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = row0 + wm * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (row >= g.M) continue;
                float v = acc[i][j][r];
                if (EPI == EPI_RAW_STATS) { s1[j] += (double)v; s2[j] += (double)v * (double)v; }
                if (EPI == EPI_AFFINE_ACT) v = lrelu(fmaf(v, ea, eb), g.slope);
                if (EPI == EPI_BWD_ACT) {
                    const float h = st_ld1<HT>(g.eh, (size_t)row * g.Nc + col);
                    v = fmaf(h, ea, eb) > 0.f ? v : v * g.slope;
                    s1[j] += (double)v; s2[j] += (double)v * (double)h;
                }
                st_st1<OT>(op, (size_t)row * g.o_segw + ocol, v);
            }
    }
    if (EPI == EPI_RAW_STATS || EPI == EPI_BWD_ACT) {
        double *red = reinterpret_cast<double *>(smem);      // [4 waves][TN][2][32]  (the staging buffer is idle: last loop barrier passed)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            s1[j] += __shfl_xor(s1[j], 32, 64);
            s2[j] += __shfl_xor(s2[j], 32, 64);
            if (lh == 0) { red[((wv * TN + j) * 2 + 0) * 32 + li] = s1[j]; red[((wv * TN + j) * 2 + 1) * 32 + li] = s2[j]; }
        }
        __syncthreads();
        if (t < 2 * BN) {
            const int which = t / BN, col = t - which * BN;
            const int cwn = col >> 6, j = (col >> 5) & 1, i = col & 31;
            double s = 0.0;
            if (BN == 128) {
#pragma unroll
                for (int m = 0; m < 2; ++m) s += red[(((m * 2 + cwn) * TN + j) * 2 + which) * 32 + i];
            } else {
#pragma unroll
                for (int m = 0; m < 4; ++m) s += red[((m * TN + j) * 2 + which) * 32 + i];
            }
            if (g.acc_rows) g.acc_rows[((size_t)(row0 / PW_BM) * 2 + which) * g.Nc + col0 + col] = s;
            else atomicAdd(&g.acc[which * g.Nc + col0 + col], s);
        }
        bn_tail(g.tail, (unsigned)((g.Nc / BN) * ((g.M + PW_BM - 1) / PW_BM)), t, 256, reinterpret_cast<unsigned *>(red + 4 * TN * 2 * 32));
    }
}

// Out(M, Nc) = T(A)(M, K) . W(Nc, K)^T          grid: 8 * ncb * ceil(nrb / 8) blocks of 256 threads
template <int BN, int PRO, int EPI, bool OUT_CF>
__global__ void __launch_bounds__(256, 2) pw_rowgemm_kernel(const RowGemm g)
{
    extern __shared__ f32x4 pw_smem4[];
    float *smem = reinterpret_cast<float *>(pw_smem4);
    constexpr int TM = (BN == 128) ? 2 : 1, TN = 2, WM = TM * 32, NB = BN / 32;
    float *sA = smem;                               // [2][128][36]
    float *sB = smem + 2 * PW_BM * PW_LD;           // [2][BN][36]
    const int ncb = g.Nc / BN, nrb = (g.M + PW_BM - 1) / PW_BM;
    // blocks i, i+8, i+16.. share an XCD (and its L2): give them the column blocks of one row block
    const int id = blockIdx.x, xcd = id & 7, slot = id >> 3;
    const int rb = (slot / ncb) * 8 + xcd, cb = slot % ncb;
    if (rb >= nrb) return;
    const int row0 = rb * PW_BM, col0 = cb * BN;
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6, li = lane & 31, lh = lane >> 5;
    const int wm = (BN == 128) ? (wv >> 1) : wv, wn = (BN == 128) ? (wv & 1) : 0;
    const int c4 = t & 7, r0 = t >> 3;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    f32x4 ra[4], ra2[4], rw[NB];
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    auto load = [&](int kt) {
        const int kcol = kt * PW_BK + 4 * c4;
        const int seg = kcol / g.a_segw, koff = kcol - seg * g.a_segw;
        const float *ap = static_cast<const float *>(g.a[seg]) + koff;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int row = row0 + r0 + 32 * p;
            ra[p] = row < g.M ? ldg4(ap + (size_t)row * g.lda) : zero4;
        }
        if (PRO == PRO_BWD) {
            const float *ap2 = static_cast<const float *>(g.a2[seg]) + koff;
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const int row = row0 + r0 + 32 * p;
                ra2[p] = row < g.M ? ldg4(ap2 + (size_t)row * g.lda) : zero4;
            }
        }
#pragma unroll
        for (int p = 0; p < NB; ++p) {
            const int n = col0 + r0 + 32 * p;
            if (g.w_pq) {
                const float *wr = g.w + (size_t)(n & 63) * 2 * g.K + kcol;
                const f32x4 wa = ldg4(wr);
                rw[p] = n < 64 ? wa : ldg4(wr + g.K) - wa;
            } else
                rw[p] = ldg4(g.w + (size_t)n * g.K + kcol);
        }
    };
    if (g.zero && blockIdx.x == 0)
        for (int z = t; z < g.nzero; z += 256) g.zero[z] = 0.0;
    auto store = [&](int kt, int buf) {
        const int kcol = kt * PW_BK + 4 * c4;
        f32x4 ca = zero4, cb4 = zero4, cc = zero4;
        if (PRO != PRO_NONE) { ca = ldg4(g.pa + kcol); cb4 = ldg4(g.pb + kcol); }
        if (PRO == PRO_BWD) cc = ldg4(g.pc + kcol);
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            f32x4 v = ra[p];
            if (PRO == PRO_BNACT) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = lrelu(fmaf(v[e], ca[e], cb4[e]), g.slope);
            } else if (PRO == PRO_BWD) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = fmaf(ra2[p][e], cc[e], fmaf(v[e], ca[e], cb4[e]));
            }
            if (PRO != PRO_NONE && row0 + r0 + 32 * p >= g.M) v = zero4;
            *reinterpret_cast<f32x4 *>(&sA[(buf * PW_BM + r0 + 32 * p) * PW_LD + 4 * c4]) = v;
        }
#pragma unroll
        for (int p = 0; p < NB; ++p)
            *reinterpret_cast<f32x4 *>(&sB[(buf * BN + r0 + 32 * p) * PW_LD + 4 * c4]) = rw[p];
    };

    const int KT = g.K / PW_BK;
    load(0);
    store(0, 0);
    __syncthreads();
    for (int kt = 0; kt < KT; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < KT) load(kt + 1);
        const float *pa = &sA[(buf * PW_BM + wm * WM + li) * PW_LD + 4 * lh];
        const float *pb = &sB[(buf * BN + wn * 64 + li) * PW_LD + 4 * lh];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            f32x4 af[TM], bf[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const f32x4 *>(pa + i * 32 * PW_LD + 8 * q);
#pragma unroll
            for (int j = 0; j < TN; ++j) bf[j] = *reinterpret_cast<const f32x4 *>(pb + j * 32 * PW_LD + 8 * q);
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        if (OUT_CF) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(bf[j][e], af[i][e], acc[i][j], 0, 0, 0);
                        else        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][e], bf[j][e], acc[i][j], 0, 0, 0);
                    }
        }
        if (kt + 1 < KT) store(kt + 1, buf ^ 1);
        __syncthreads();
    }

    rowgemm_epilogue<BN, EPI, OUT_CF, TM, TN>(g, acc, smem, row0, col0, wm, wn, wv, t, li, lh);
}

// ---------------------------------------------------------------------------------------------------------
// The same GEMM on the bf16 matrix cores at fp32 accuracy ("bf16x6").  An fp32 number is EXACTLY the sum of three
// bf16 numbers (truncation splits: 8 + 8 + 8 significand bits), so a . b = sum over the six piece products of
// weight >= 2^-24 (hi hi, hi mid, mid hi, hi lo, mid mid, lo hi; each product is exact in the fp32 accumulator
// input) - what is dropped is below fp32 rounding.  Six v_mfma_f32_32x32x16_bf16 (32 cycles, K = 16) replace
// eight v_mfma_f32_32x32x2_f32 (64 cycles, K = 2 each): 2.7x less matrix time, and on gfx950 the fp32 MFMA runs
// at the vector rate on issue time it shares with the VALU (scripts/micro/), so this is the only way to a GEMM
// faster than the vector peak.  A is split once per block while it is staged into LDS (4 VALU per element + the
// packing), W comes pre-split from pw_split_w_kernel.
// LDS row (one K-step of 32 for one tile row): [hi 32 x bf16 | mid | lo | 16 B pad] = 208 B = 13 sixteen-byte slots
// (odd, so the 16-lane groups of ds_read_b128 hit 16 different slots).
// ---------------------------------------------------------------------------------------------------------
constexpr int PW6_ROW = 208;
#ifndef PW_RG6_OCC
#define PW_RG6_OCC 2                    // workgroups per CU the row GEMM is compiled for (3 fits the LDS exactly and 168 VGPRs with 4-20
                                        // spilled: measured SLOWER, 117 -> 123 / 133 -> 144 / 190 -> 251 us, c2 step +0.07 ms: DESIGN_HISTORY)
#endif

// w3[n][kt][plane][32] = bf16 pieces of w(n, 32 kt + j);  w element (n, k) is read at w[n * sn + k * sk]
template <int PREC>
__global__ void pw_split_w_kernel(const float *__restrict__ w, int Nc, int K, long long sn, long long sk, unsigned short *__restrict__ w3)
{
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= Nc * K) return;
    const int n = e / K, k = e - n * K;
    mm_store1<PREC>(w[(size_t)n * sn + (size_t)k * sk], w3 + ((size_t)n * (K / 32) + k / 32) * 96 + (k & 31), 32);   // reduced forms: plane 0 only
}

// PREC_BF16 / PREC_F16: one rounded 16-bit product per fp32 product - plane 0 of the same images, one MFMA per K-step of 16.
// AT / HT / OT: ST_* of the A operand, of the pre-BN activations read beside it (PRO_BWD's second tensor, EPI_BWD_ACT's `eh`)
// and of the output.  A 16-bit A that is already in the operand format of PREC (a forward GEMM on the packed fusion input)
// goes to LDS as loaded - no conversion.
template <int BN, int PRO, int EPI, bool OUT_CF, int PREC, int AT = ST_F32, int HT = ST_F32, int OT = ST_F32>
__global__ void __launch_bounds__(256, PW_RG6_OCC) pw_rowgemm6_kernel(const RowGemm g)
{
    extern __shared__ f32x4 pw_smem4[];
    char *smem = reinterpret_cast<char *>(pw_smem4);
    constexpr int NP = Mx<PREC>::NP, PARTS = 4 * NP;          // 16-byte parts of a weight row per 32-wide K-chunk
    // K-chunks of 32 per pipeline step.  A reduced mode fills one of the three 64-byte piece planes of an LDS row; its MFMA work
    // per chunk is a sixth of the exact form's, too little to hide a trip to HBM (measured: 1.2 TB/s and 0.04 of the matrix pipe
    // at one chunk per step).  Two chunks per step - the second in the row's MID plane - put twice the bytes in flight per barrier
    // in the same LDS (layer widths are multiples of 64).
    constexpr int KM = (PREC == PREC_X6) ? 1 : 2;
    constexpr int TM = (BN == 128) ? 2 : 1, TN = 2, WM = TM * 32, NB = (BN * PARTS + 255) / 256;
    char *sA = smem;                                // [128][208 B]
    char *sB = smem + PW_BM * PW6_ROW;              // [BN][208 B]
    const int ncb = g.Nc / BN, nrb = (g.M + PW_BM - 1) / PW_BM;
    const int id = blockIdx.x, xcd = id & 7, slot = id >> 3;
    const int rb = (slot / ncb) * 8 + xcd, cb = slot % ncb;
    if (rb >= nrb) return;
    const int row0 = rb * PW_BM, col0 = cb * BN;
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6, li = lane & 31, lh = lane >> 5;
    const int wm = (BN == 128) ? (wv >> 1) : wv, wn = (BN == 128) ? (wv & 1) : 0;
    const int c4 = t & 7, r0 = t >> 3;
    const int KT = g.K / (PW_BK * KM), KT32 = g.K / PW_BK;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // 16-bit A with two chunks per step: ONE 16-byte load per row covers 8 channels - thread c4 reads chunk c4 >> 2, channels
    // 8 (c4 & 3) .. + 7 of it - so that a row's 64 channels of the step are one 128-byte segment of one instruction (two 8-byte
    // loads per row left 64-byte segments: the row GEMMs of the reduced modes ran at 1.2 TB/s)
    constexpr bool WIDE = KM == 2 && AT != ST_F32 && (PRO != PRO_BWD || HT != ST_F32);
    typename Raw4<AT>::type ra[KM][4];
    typename Raw4<HT>::type ra2[KM][4];
    uint4 rwa[WIDE ? 4 : 1], rwa2[(WIDE && PRO == PRO_BWD) ? 4 : 1];
    f32x4 rw[KM][NB];
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    constexpr bool DIRECT = PRO == PRO_NONE && ((AT == ST_F16 && PREC == PREC_F16) || (AT == ST_BF16 && PREC == PREC_BF16));
    auto load = [&](int kt) {
        if constexpr (WIDE) {
            const int kcol = kt * KM * PW_BK + 8 * c4;
            const int seg = kcol / g.a_segw, koff = kcol - seg * g.a_segw;
            const unsigned short *ap = static_cast<const unsigned short *>(g.a[seg]);
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const int row = row0 + r0 + 32 * p;
                rwa[p] = row < g.M ? *reinterpret_cast<const uint4 *>(ap + (size_t)row * g.lda + koff) : make_uint4(0u, 0u, 0u, 0u);
            }
            if constexpr (PRO == PRO_BWD) {
                const unsigned short *ap2 = static_cast<const unsigned short *>(g.a2[seg]);
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    const int row = row0 + r0 + 32 * p;
                    rwa2[p] = row < g.M ? *reinterpret_cast<const uint4 *>(ap2 + (size_t)row * g.lda + koff) : make_uint4(0u, 0u, 0u, 0u);
                }
            }
        }
#pragma unroll
        for (int kc = 0; kc < KM; ++kc) {
            const int kcol = (kt * KM + kc) * PW_BK + 4 * c4;
            const int seg = kcol / g.a_segw, koff = kcol - seg * g.a_segw;
            const void *ap = g.a[seg];
            if constexpr (!WIDE) {
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const int row = row0 + r0 + 32 * p;
                ra[kc][p] = row < g.M ? st_ld4<AT>(ap, (size_t)row * g.lda + koff) : st_zero4<AT>();
            }
            }
            if (PRO == PRO_BWD && !WIDE) {
                const void *ap2 = g.a2[seg];
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    const int row = row0 + r0 + 32 * p;
                    ra2[kc][p] = row < g.M ? st_ld4<HT>(ap2, (size_t)row * g.lda + koff) : st_zero4<HT>();
                }
            }
            const char *w3 = static_cast<const char *>(g.w3);
#pragma unroll
            for (int p = 0; p < NB; ++p) {
                const int q = t + 256 * p, row = q / PARTS, part = q - row * PARTS;
                if (BN * PARTS % 256 == 0 || q < BN * PARTS)
                    rw[kc][p] = ldg4(reinterpret_cast<const float *>(w3 + ((size_t)(col0 + row) * KT32 + kt * KM + kc) * 192 + part * 16));
            }
        }
    };
    auto store = [&](int kt) {
        if constexpr (WIDE) {
            // this thread's 8 channels: K columns kt 64 + 8 c4 .. + 7 = chunk c4 >> 2, 16-byte slot c4 & 3 of its plane
            const int kcol = kt * KM * PW_BK + 8 * c4;
            f32x4 ca[2], cb4[2], cc[2];
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
                ca[hf] = cb4[hf] = cc[hf] = zero4;
                if (PRO != PRO_NONE) { ca[hf] = ldg4(g.pa + kcol + 4 * hf); cb4[hf] = ldg4(g.pb + kcol + 4 * hf); }
                if (PRO == PRO_BWD) cc[hf] = ldg4(g.pc + kcol + 4 * hf);
            }
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                uint4 outw;
                if constexpr (DIRECT) {
                    outw = rwa[p];
                } else {
                    unsigned o4[4];
#pragma unroll
                    for (int hf = 0; hf < 2; ++hf) {
                        f32x4 v = st_cvt4<AT>(hf ? make_uint2(rwa[p].z, rwa[p].w) : make_uint2(rwa[p].x, rwa[p].y));
                        if constexpr (PRO == PRO_BNACT) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] = lrelu(fmaf(v[e], ca[hf][e], cb4[hf][e]), g.slope);
                        } else if constexpr (PRO == PRO_BWD) {
                            const f32x4 h4 = st_cvt4<HT>(hf ? make_uint2(rwa2[p].z, rwa2[p].w) : make_uint2(rwa2[p].x, rwa2[p].y));
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] = fmaf(h4[e], cc[hf][e], fmaf(v[e], ca[hf][e], cb4[hf][e]));
                        }
                        if (PRO != PRO_NONE && row0 + r0 + 32 * p >= g.M) v = zero4;
                        o4[2 * hf] = mm_pack2<PREC>(v[0], v[1]);
                        o4[2 * hf + 1] = mm_pack2<PREC>(v[2], v[3]);
                    }
                    outw = make_uint4(o4[0], o4[1], o4[2], o4[3]);
                }
                *reinterpret_cast<uint4 *>(sA + (r0 + 32 * p) * PW6_ROW + 64 * (c4 >> 2) + 16 * (c4 & 3)) = outw;
            }
        }
#pragma unroll
        for (int kc = 0; kc < KM; ++kc) {
            const int kcol = (kt * KM + kc) * PW_BK + 4 * c4;
            f32x4 ca = zero4, cb4 = zero4, cc = zero4;
            if (PRO != PRO_NONE) { ca = ldg4(g.pa + kcol); cb4 = ldg4(g.pb + kcol); }
            if (PRO == PRO_BWD) cc = ldg4(g.pc + kcol);
            if constexpr (!WIDE) {
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                uint2 pc[NP];
                if constexpr (DIRECT) {
                    pc[0] = ra[kc][p];                 // already operands of this PREC (rows past M were loaded as zeros)
                } else {
                    f32x4 v = st_cvt4<AT>(ra[kc][p]);
                    if (PRO == PRO_BNACT) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = lrelu(fmaf(v[e], ca[e], cb4[e]), g.slope);
                    } else if (PRO == PRO_BWD) {
                        const f32x4 h4 = st_cvt4<HT>(ra2[kc][p]);
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = fmaf(h4[e], cc[e], fmaf(v[e], ca[e], cb4[e]));
                    }
                    if (PRO != PRO_NONE && row0 + r0 + 32 * p >= g.M) v = zero4;
                    const float v4[4] = {v[0], v[1], v[2], v[3]};
#if defined(PW_ABL) && (PW_ABL & 1)       // ablation builds only: no operand split (wrong results, timing of the rest)
#pragma unroll
                    for (int q = 0; q < NP; ++q) pc[q] = make_uint2(__builtin_bit_cast(unsigned, v4[0]) ^ __builtin_bit_cast(unsigned, v4[1]), __builtin_bit_cast(unsigned, v4[2]));
#else
                    mm_pieces4<PREC>(v4, pc);
#endif
                    if constexpr (PRO == PRO_NONE && AT == ST_F32 && PREC != PREC_X6) {
                        if (g.a16_out && cb == 0 && row0 + r0 + 32 * p < g.M)
                            *reinterpret_cast<uint2 *>(static_cast<unsigned short *>(g.a16_out) + (size_t)(row0 + r0 + 32 * p) * g.K + kcol) = pc[0];
                    }
                }
                char *dst = sA + (r0 + 32 * p) * PW6_ROW + c4 * 8 + 64 * kc;
#pragma unroll
                for (int q = 0; q < NP; ++q) *reinterpret_cast<uint2 *>(dst + 64 * q) = pc[q];
            }
            }
#pragma unroll
            for (int p = 0; p < NB; ++p) {
                const int q = t + 256 * p, row = q / PARTS, part = q - row * PARTS;
                if (BN * PARTS % 256 == 0 || q < BN * PARTS)
                    *reinterpret_cast<f32x4 *>(sB + row * PW6_ROW + part * 16 + 64 * kc) = rw[kc][p];
            }
        }
    };

    const char *pa = sA + (wm * WM + li) * PW6_ROW + lh * 16;
    const char *pb = sB + (wn * 64 + li) * PW6_ROW + lh * 16;
    load(0);
    for (int kt = 0; kt < KT; ++kt) {
        store(kt);
        __syncthreads();
        if (kt + 1 < KT) load(kt + 1);
#pragma unroll
        for (int kc = 0; kc < KM; ++kc)
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            uint4 af[TM][NP], bf[TN][NP];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int pl = 0; pl < NP; ++pl)
                    af[i][pl] = *reinterpret_cast<const uint4 *>(pa + i * 32 * PW6_ROW + (pl + kc) * 64 + kb * 32);
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int pl = 0; pl < NP; ++pl)
                    bf[j][pl] = *reinterpret_cast<const uint4 *>(pb + j * 32 * PW6_ROW + (pl + kc) * 64 + kb * 32);
            // smallest pieces first
            constexpr int PA[6] = GFS3D_X6_PA, PB[6] = GFS3D_X6_PB;
#pragma unroll
            for (int term = 0; term < Mx<PREC>::NTERM; ++term)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        const int ia = PREC == PREC_X6 ? PA[term] : 0, ib = PREC == PREC_X6 ? PB[term] : 0;
                        if (OUT_CF) acc[i][j] = mm_mfma<PREC>(bf[j][ib], af[i][ia], acc[i][j]);
                        else        acc[i][j] = mm_mfma<PREC>(af[i][ia], bf[j][ib], acc[i][j]);
                    }
        }
        __syncthreads();
    }
    rowgemm_epilogue<BN, EPI, OUT_CF, TM, TN, HT, OT>(g, acc, reinterpret_cast<float *>(smem), row0, col0, wm, wn, wv, t, li, lh);
}

struct WGrad {
    const void *u, *u2;       // (M, Co): PRO_BWD combine a u + b + c u2 with per-Co coefficients, or plain u; element types: templates GT, HT
    int ldu;
    const float *ua, *ub, *uc;
    const void *v[4];         // (M, Ci) in column segments of width v_segw, row stride ldv; element type: template VT
    int gt, ht, vt;           // ST_* of u / u2 / v: picks the instantiation (launch_wgrad_t)
    int v_segw, ldv;
    const float *va, *vb;     // PRO_BNACT per-Ci scale / shift
    float slope;
    int M, Co, Ci, rows_per_split, nsplit;
    float *dw;                // (Co, Ci)
    float *part;              // [nsplit][Co][Ci] partial products of the point splits, added in split order by pw_wgrad_reduce_kernel
    int prec;                 // PREC_X6 or PREC_BF16 (gradients: no half operands)
};

// dW(Co, Ci) += sum_m T(U)[m, co] T(V)[m, ci] on the bf16 matrix cores (bf16x6, see pw_rowgemm6_kernel).  The reduction
// runs over the POINTS, so both operands are transposed while they are staged: a thread owns 4 channels x 2
// consecutive points, splits the 8 values and writes, per channel and piece, one dword (2 bf16) into that channel's
// LDS row - rows are channels, 32 points per K-step; lanes of a 16-lane group walk along a row (conflict-free).
template <int BMU, int BNV, int VPRO, int PREC, int GT = ST_F32, int HT = ST_F32, int VT = ST_F32>
__global__ void __launch_bounds__(256, 2) pw_wgrad6_kernel(const WGrad g)
{
    constexpr int NP = Mx<PREC>::NP;
    constexpr int KM = (PREC == PREC_X6) ? 1 : 2;       // 32-point chunks per pipeline step (see pw_rowgemm6_kernel): chunk kc in piece plane kc
    extern __shared__ f32x4 pw_smem4[];
    char *smem = reinterpret_cast<char *>(pw_smem4);
    constexpr int TM = BMU / 64, TN = BNV / 64, NU = BMU / 64, NV = BNV / 64;      // (quad, pair) items per thread
    char *sU = smem;                          // [BMU][208 B]
    char *sV = smem + BMU * PW6_ROW;          // [BNV][208 B]
    const int nci = g.Ci / BNV, ntile = (g.Co / BMU) * nci;
    const int id = blockIdx.x, xcd = id & 7, slot = id >> 3;
    const int split = (slot / ntile) * 8 + xcd, tile = slot % ntile;
    if (split >= g.nsplit) return;
    const int co0 = (tile / nci) * BMU, ci0 = (tile % nci) * BNV;
    const int m0 = split * g.rows_per_split, m1 = min(g.M, m0 + g.rows_per_split);
    if (m0 >= m1) return;
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6, li = lane & 31, lh = lane >> 5;
    const int wm = wv >> 1, wn = wv & 1;
    const int pair = t & 15, quad0 = t >> 4;                 // item p: channel quad quad0 + 16 p, points 2 pair, 2 pair + 1
    const void *up = g.u, *up2 = g.u2;
    const void *vp[NV];                                      // a V tile may span several 64-channel input segments
    int voff[NV];
#pragma unroll
    for (int p = 0; p < NV; ++p) {
        const int ch = ci0 + 4 * (quad0 + 16 * p), vseg = ch / g.v_segw;
        vp[p] = g.v[vseg];
        voff[p] = ch - vseg * g.v_segw;
    }
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    f32x4 ua[NU], ub[NU], uc[NU], va[NV], vb[NV];
#pragma unroll
    for (int p = 0; p < NU; ++p) {
        ua[p] = ub[p] = uc[p] = zero4;
        if (up2) {
            const int c = co0 + 4 * (quad0 + 16 * p);
            ua[p] = ldg4(g.ua + c); ub[p] = ldg4(g.ub + c); uc[p] = ldg4(g.uc + c);
        }
    }
#pragma unroll
    for (int p = 0; p < NV; ++p) {
        va[p] = vb[p] = zero4;
        if (VPRO == PRO_BNACT) { const int c = ci0 + 4 * (quad0 + 16 * p); va[p] = ldg4(g.va + c); vb[p] = ldg4(g.vb + c); }
    }

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    typename Raw4<GT>::type rU[KM][NU][2];
    typename Raw4<HT>::type rU2[KM][NU][2];
    typename Raw4<VT>::type rV[KM][NV][2];
    auto load = [&](int kt) {
#pragma unroll
        for (int kc = 0; kc < KM; ++kc) {
            const int m = m0 + (kt * KM + kc) * 32 + 2 * pair;
#pragma unroll
            for (int p = 0; p < NU; ++p)
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const size_t o = (size_t)(m + e) * g.ldu + co0 + 4 * (quad0 + 16 * p);
                    rU[kc][p][e] = m + e < m1 ? st_ld4<GT>(up, o) : st_zero4<GT>();
                    if (up2) rU2[kc][p][e] = m + e < m1 ? st_ld4<HT>(up2, o) : st_zero4<HT>();
                }
#pragma unroll
            for (int p = 0; p < NV; ++p)
#pragma unroll
                for (int e = 0; e < 2; ++e)
                    rV[kc][p][e] = m + e < m1 ? st_ld4<VT>(vp[p], (size_t)(m + e) * g.ldv + voff[p]) : st_zero4<VT>();
        }
    };
    // one item: 4 channels x 2 points -> per channel and piece one dword at [channel row][piece][pair]
    auto put = [&](char *base, int quad, const f32x4 &x0, const f32x4 &x1) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            char *dst = base + (4 * quad + c) * PW6_ROW + pair * 4;
            if constexpr (PREC == PREC_X6) {
                unsigned h0, m0_, l0, h1, m1_, l1;
                split3(x0[c], h0, m0_, l0);
                split3(x1[c], h1, m1_, l1);
                *reinterpret_cast<unsigned *>(dst) = pack_hi16(h0, h1);
                *reinterpret_cast<unsigned *>(dst + 64) = pack_hi16(m0_, m1_);
                *reinterpret_cast<unsigned *>(dst + 128) = pack_hi16(l0, l1);
            } else {
                *reinterpret_cast<unsigned *>(dst) = mm_pack2<PREC>(x0[c], x1[c]);
            }
        }
    };
    auto store = [&](int kt) {
#pragma unroll
      for (int kc = 0; kc < KM; ++kc) {
        const int m = m0 + (kt * KM + kc) * 32 + 2 * pair;
#pragma unroll
        for (int p = 0; p < NU; ++p) {
            f32x4 x[2];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                x[e] = st_cvt4<GT>(rU[kc][p][e]);
                if (up2) {
                    const f32x4 h4 = st_cvt4<HT>(rU2[kc][p][e]);
#pragma unroll
                    for (int c = 0; c < 4; ++c) x[e][c] = fmaf(h4[c], uc[p][c], fmaf(x[e][c], ua[p][c], ub[p][c]));
                    if (m + e >= m1) x[e] = zero4;
                }
            }
            put(sU + 64 * kc, quad0 + 16 * p, x[0], x[1]);
        }
#pragma unroll
        for (int p = 0; p < NV; ++p) {
            f32x4 x[2];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                x[e] = st_cvt4<VT>(rV[kc][p][e]);
                if (VPRO == PRO_BNACT) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) x[e][c] = lrelu(fmaf(x[e][c], va[p][c], vb[p][c]), g.slope);
                    if (m + e >= m1) x[e] = zero4;
                }
            }
            put(sV + 64 * kc, quad0 + 16 * p, x[0], x[1]);
        }
      }
    };

    const int KT = (m1 - m0 + 32 * KM - 1) / (32 * KM);
    const char *pu = sU + (wm * (BMU / 2) + li) * PW6_ROW + lh * 16;
    const char *pv = sV + (wn * (BNV / 2) + li) * PW6_ROW + lh * 16;
    load(0);
    for (int kt = 0; kt < KT; ++kt) {
        store(kt);
        __syncthreads();
        if (kt + 1 < KT) load(kt + 1);
#pragma unroll
        for (int kc = 0; kc < KM; ++kc)
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            uint4 af[TM][NP], bf[TN][NP];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int pl = 0; pl < NP; ++pl)
                    af[i][pl] = *reinterpret_cast<const uint4 *>(pu + i * 32 * PW6_ROW + (pl + kc) * 64 + kb * 32);
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int pl = 0; pl < NP; ++pl)
                    bf[j][pl] = *reinterpret_cast<const uint4 *>(pv + j * 32 * PW6_ROW + (pl + kc) * 64 + kb * 32);
            constexpr int PA[6] = GFS3D_X6_PA, PB[6] = GFS3D_X6_PB;
#pragma unroll
            for (int term = 0; term < Mx<PREC>::NTERM; ++term)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = mm_mfma<PREC>(af[i][PREC == PREC_X6 ? PA[term] : 0], bf[j][PREC == PREC_X6 ? PB[term] : 0], acc[i][j]);
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = co0 + wm * (BMU / 2) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                const int ci = ci0 + wn * (BNV / 2) + j * 32 + li;
                g.part[((size_t)split * g.Co + co) * g.Ci + ci] = acc[i][j][r];
            }
}

// ---------------------------------------------------------------------------------------------------------
// The weight gradient of the REDUCED modes (one bf16 product per fp32 product).  pw_wgrad6_kernel transposes both operands on
// their way INTO LDS (rows = channels, 4-byte stores of point pairs, 16 per thread and step) from loads that touch 32 bytes of
// every row they visit: with a sixth of the exact form's matrix work per byte that staging is the whole kernel (round 6: 213 us
// for 34 GF and 268 MB - 1.2 TB/s, 0.04 of the matrix pipe).  Here the tiles stay in their NATURAL layout - image rows = points,
// 16-byte loads along the channels (a point's tile slice is one contiguous segment), 16-byte LDS stores - and the transposition
// happens in the operand fetch: ds_read_b64_tr_b16 delivers, for rows = channels, the 8 points of a K-step half in MFMA
// operand order (the read K5 and the EdgeConv backward use).  Both operands are fetched that way, so their K order agrees.
// 64 points per pipeline step; bf16 operands (gradients: no half).  GT / HT / VT: ST_* of u (g) / u2 (h) / v.
// ---------------------------------------------------------------------------------------------------------
typedef short pw_s16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) pw_s16x4 pw_lds_s16x4;
#ifndef PW_WG16_PTS
#define PW_WG16_PTS 64
#endif
#ifndef PW_WG16_OCC
#define PW_WG16_OCC 2
#endif
constexpr int WG16_PTS = PW_WG16_PTS;
__host__ __device__ constexpr int wg16_prow(int w) { return 2 * w + 16; }          // bytes per point row of a W-channel image
// A / B operand of K-step `s` (16 points) for channel rows 32 dt + (lane & 31): lanes 4 q + p4 of a 16-lane group address row q,
// channels 4 p4 .. + 3 of the group's 16; lane c receives its channel's 4 rows (element q = row q).  EXEC all ones.
template <int PROW>
__device__ __forceinline__ uint4 wg16_frag(const unsigned char *img, int lane, int dt, int s)
{
    const int h = lane >> 5, g1 = (lane >> 4) & 1, q = (lane >> 2) & 3, p4 = lane & 3;
    const unsigned char *base = img + (16 * s + 4 * h + q) * PROW + 2 * (32 * dt + 16 * g1 + 4 * p4);
    const pw_s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((pw_lds_s16x4 *)base);
    const pw_s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((pw_lds_s16x4 *)(base + 8 * PROW));
    const uint2 a = __builtin_bit_cast(uint2, lo), b = __builtin_bit_cast(uint2, hi);
    return make_uint4(a.x, a.y, b.x, b.y);
}
// eight consecutive stored channels of one point -> two f32x4
template <int ST> struct Raw8 { typedef uint4 type; };
template <> struct Raw8<ST_F32> { struct type { f32x4 a, b; }; };
template <int ST> __device__ __forceinline__ typename Raw8<ST>::type st_ld8(const void *base, size_t idx, bool ok)
{
    typename Raw8<ST>::type r;
    if constexpr (ST == ST_F32) {
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
        r.a = ok ? *reinterpret_cast<const f32x4 *>(static_cast<const float *>(base) + idx) : z;
        r.b = ok ? *reinterpret_cast<const f32x4 *>(static_cast<const float *>(base) + idx + 4) : z;
    } else
        r = ok ? *reinterpret_cast<const uint4 *>(static_cast<const unsigned short *>(base) + idx) : make_uint4(0u, 0u, 0u, 0u);
    return r;
}
template <int ST> __device__ __forceinline__ void st_cvt8(const typename Raw8<ST>::type &r, f32x4 &lo, f32x4 &hi)
{
    if constexpr (ST == ST_F32) { lo = r.a; hi = r.b; }
    else { lo = st_cvt4<ST>(make_uint2(r.x, r.y)); hi = st_cvt4<ST>(make_uint2(r.z, r.w)); }
}

template <int BMU, int BNV, int VPRO, int GT, int HT, int VT>
__global__ void __launch_bounds__(256, PW_WG16_OCC) pw_wgrad16_kernel(const WGrad g)
{
    extern __shared__ f32x4 pw_smem4[];
    unsigned char *smem = reinterpret_cast<unsigned char *>(pw_smem4);
    constexpr int TM = BMU / 64, TN = BNV / 64;
    constexpr int UROW = wg16_prow(BMU), VROW = wg16_prow(BNV);
    constexpr int UO = BMU / 8, VO = BNV / 8;                        // channel octets per point
    constexpr int NUI = WG16_PTS * UO / 256, NVI = WG16_PTS * VO / 256;      // (point, octet) items per thread and step
    static_assert(WG16_PTS * UO % 256 == 0 && WG16_PTS * VO % 256 == 0, "whole items per thread");
    static_assert(VPRO == PRO_NONE || 256 % VO == 0, "a thread's V octet must not change between its items when V has coefficients");
    static_assert(256 % UO == 0, "a thread's U octet must not change between its items");
    unsigned char *sU = smem;                                        // [64 points][UROW]
    unsigned char *sV = smem + WG16_PTS * UROW;                      // [64 points][VROW]
    const int nci = g.Ci / BNV, ntile = (g.Co / BMU) * nci;
    const int id = blockIdx.x, xcd = id & 7, slot = id >> 3;
    const int split = (slot / ntile) * 8 + xcd, tile = slot % ntile;
    if (split >= g.nsplit) return;
    const int co0 = (tile / nci) * BMU, ci0 = (tile % nci) * BNV;
    const int m0 = split * g.rows_per_split, m1 = min(g.M, m0 + g.rows_per_split);
    if (m0 >= m1) return;
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6, li = lane & 31, lh = lane >> 5;
    const int wm = wv >> 1, wn = wv & 1;
    const bool comb = g.u2 != nullptr;                               // BatchNorm-backward combine a u + b + c u2 (per Co channel)
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    const int uoct = t % UO;
    f32x4 ua[2], ub[2], uc[2], va[2], vb[2];
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
        ua[hf] = ub[hf] = uc[hf] = va[hf] = vb[hf] = zero4;
        if (comb) { const int c = co0 + 8 * uoct + 4 * hf; ua[hf] = ldg4(g.ua + c); ub[hf] = ldg4(g.ub + c); uc[hf] = ldg4(g.uc + c); }
        if (VPRO == PRO_BNACT) { const int c = ci0 + 8 * (t % VO) + 4 * hf; va[hf] = ldg4(g.va + c); vb[hf] = ldg4(g.vb + c); }
    }

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    typename Raw8<GT>::type rU[NUI];
    typename Raw8<HT>::type rU2[NUI];
    typename Raw8<VT>::type rV[NVI];
    auto load = [&](int kt) {
        const int mb = m0 + kt * WG16_PTS;
#pragma unroll
        for (int p = 0; p < NUI; ++p) {
            const int it = t + 256 * p, pt = it / UO, oc = it - pt * UO;
            const bool ok = mb + pt < m1;
            const size_t o = (size_t)(mb + pt) * g.ldu + co0 + 8 * oc;
            rU[p] = st_ld8<GT>(g.u, o, ok);
            if (comb) rU2[p] = st_ld8<HT>(g.u2, o, ok);
        }
#pragma unroll
        for (int p = 0; p < NVI; ++p) {
            const int it = t + 256 * p, pt = it / VO, oc = it - pt * VO;
            const bool ok = mb + pt < m1;
            const int ch = ci0 + 8 * oc, vseg = ch / g.v_segw;
            rV[p] = st_ld8<VT>(g.v[vseg], (size_t)(mb + pt) * g.ldv + (ch - vseg * g.v_segw), ok);
        }
    };
    auto store = [&](int kt) {
        const int mb = m0 + kt * WG16_PTS;
#pragma unroll
        for (int p = 0; p < NUI; ++p) {
            const int it = t + 256 * p, pt = it / UO, oc = it - pt * UO;
            uint4 o;
            if constexpr (GT == ST_BF16) {
                if (!comb) { o = rU[p]; *reinterpret_cast<uint4 *>(sU + pt * UROW + 16 * oc) = o; continue; }
            }
            f32x4 x[2];
            st_cvt8<GT>(rU[p], x[0], x[1]);
            if (comb) {
                f32x4 h[2];
                st_cvt8<HT>(rU2[p], h[0], h[1]);
#pragma unroll
                for (int hf = 0; hf < 2; ++hf)
#pragma unroll
                    for (int c = 0; c < 4; ++c) x[hf][c] = fmaf(h[hf][c], uc[hf][c], fmaf(x[hf][c], ua[hf][c], ub[hf][c]));
                if (mb + pt >= m1) { x[0] = zero4; x[1] = zero4; }
            }
            o = make_uint4(mm_pack2<PREC_BF16>(x[0][0], x[0][1]), mm_pack2<PREC_BF16>(x[0][2], x[0][3]),
                           mm_pack2<PREC_BF16>(x[1][0], x[1][1]), mm_pack2<PREC_BF16>(x[1][2], x[1][3]));
            *reinterpret_cast<uint4 *>(sU + pt * UROW + 16 * oc) = o;
        }
#pragma unroll
        for (int p = 0; p < NVI; ++p) {
            const int it = t + 256 * p, pt = it / VO, oc = it - pt * VO;
            uint4 o;
            if constexpr (VT == ST_BF16 && VPRO == PRO_NONE) {
                o = rV[p];
            } else {
                f32x4 x[2];
                st_cvt8<VT>(rV[p], x[0], x[1]);
                if (VPRO == PRO_BNACT) {
#pragma unroll
                    for (int hf = 0; hf < 2; ++hf)
#pragma unroll
                        for (int c = 0; c < 4; ++c) x[hf][c] = lrelu(fmaf(x[hf][c], va[hf][c], vb[hf][c]), g.slope);
                    if (mb + pt >= m1) { x[0] = zero4; x[1] = zero4; }
                }
                o = make_uint4(mm_pack2<PREC_BF16>(x[0][0], x[0][1]), mm_pack2<PREC_BF16>(x[0][2], x[0][3]),
                               mm_pack2<PREC_BF16>(x[1][0], x[1][1]), mm_pack2<PREC_BF16>(x[1][2], x[1][3]));
            }
            *reinterpret_cast<uint4 *>(sV + pt * VROW + 16 * oc) = o;
        }
    };

    const int KT = (m1 - m0 + WG16_PTS - 1) / WG16_PTS;
    load(0);
    for (int kt = 0; kt < KT; ++kt) {
        store(kt);
        __syncthreads();
        if (kt + 1 < KT) load(kt + 1);
#pragma unroll
        for (int s = 0; s < WG16_PTS / 16; ++s) {
            uint4 af[TM], bf[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) af[i] = wg16_frag<UROW>(sU, lane, wm * TM + i, s);
#pragma unroll
            for (int j = 0; j < TN; ++j) bf[j] = wg16_frag<VROW>(sV, lane, wn * TN + j, s);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = mm_mfma<PREC_BF16>(af[i], bf[j], acc[i][j]);
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = co0 + wm * (BMU / 2) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                const int ci = ci0 + wn * (BNV / 2) + j * 32 + li;
                g.part[((size_t)split * g.Co + co) * g.Ci + ci] = acc[i][j][r];
            }
}

// y_cf(b, c, n) = act(scale h(b n, c) + shift): 64 points x 64 channels per block, grid (ceil(N/64), C/64, B)
template <int HT>
__global__ void __launch_bounds__(256)
pw_out_kernel(const void *__restrict__ h, const float *__restrict__ stats, int N, int C, float slope, float *__restrict__ y_cf,
              float *__restrict__ y_pm)
{
    __shared__ float tile[64][65];
    const int tid = threadIdx.x, n0 = blockIdx.x * 64, c0 = blockIdx.y * 64, b = blockIdx.z;
    const int c = tid & 63, wv = tid >> 6;
    const float sc = stats ? stats[PST_SCALE * C + c0 + c] : 1.f, sh = stats ? stats[PST_SHIFT * C + c0 + c] : 0.f;   // NULL: plain transpose
    for (int p = wv; p < 64; p += 4) {
        const int n = n0 + p;
        const float v = n < N ? lrelu(fmaf(st_ld1<HT>(h, ((size_t)b * N + n) * C + c0 + c), sc, sh), slope) : 0.f;
        tile[p][c] = v;
        if (y_pm && n < N) y_pm[((size_t)b * N + n) * C + c0 + c] = v;
    }
    if (!y_cf) return;
    __syncthreads();
    const int p = tid & 63;
    if (n0 + p < N)
        for (int cc = wv; cc < 64; cc += 4) y_cf[((size_t)b * C + c0 + cc) * N + n0 + p] = tile[p][cc];
}

// g(b n, c) = dy_cf(b, c, n) * act'(scale h + shift);  acc[0][c] += sum g, acc[1][c] += sum g h
// grid (ceil(N / 256), C / 64, B): a block walks 4 tiles of 64 points (loads batched ahead of their use; one set of
// same-address fp64 atomics per block)
constexpr int PW_IN_TILES = 4;
template <int HT, int GT>
__global__ void __launch_bounds__(256)
pw_bwd_in_kernel(const float *__restrict__ dy_cf, const float *__restrict__ dy_pm, const void *__restrict__ h,
                 const float *__restrict__ stats, int N, int C, float slope, void *__restrict__ gout, double *__restrict__ acc,
                 double *__restrict__ acc_rows, const BnTail tail)
{
    __shared__ float tile[64][65];
    __shared__ float red[4][2][64];
    __shared__ unsigned tail_flag;
    const int tid = threadIdx.x, c0 = blockIdx.y * 64, b = blockIdx.z;
    const int c = tid & 63, wv = tid >> 6;
    const float sc = stats[PST_SCALE * C + c0 + c], sh = stats[PST_SHIFT * C + c0 + c];
    float s = 0.f, sx = 0.f;
    for (int tt = 0; tt < PW_IN_TILES; ++tt) {
        const int n0 = (blockIdx.x * PW_IN_TILES + tt) * 64;
        if (n0 >= N) break;
        __syncthreads();
        float rh[16], rt[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) rh[j] = st_ld1<HT>(h, ((size_t)b * N + min(n0 + wv + 4 * j, N - 1)) * C + c0 + c);
        {
            const int p = tid & 63;
#pragma unroll
            for (int j = 0; j < 16; ++j) rt[j] = (dy_cf && n0 + p < N) ? dy_cf[((size_t)b * C + c0 + wv + 4 * j) * N + n0 + p] : 0.f;
#pragma unroll
            for (int j = 0; j < 16; ++j) tile[p][wv + 4 * j] = rt[j];
        }
        __syncthreads();
        float rp[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) rp[j] = (dy_pm && n0 + wv + 4 * j < N) ? dy_pm[((size_t)b * N + n0 + wv + 4 * j) * C + c0 + c] : 0.f;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int p = wv + 4 * j, n = n0 + p;
            if (n < N) {
                const float hv = rh[j];
                const float gin = tile[p][c] + rp[j];
                const float gv = fmaf(hv, sc, sh) > 0.f ? gin : gin * slope;
                st_st1<GT>(gout, ((size_t)b * N + n) * C + c0 + c, gv);
                s += gv;
                sx = fmaf(gv, hv, sx);
            }
        }
    }
    red[wv][0][c] = s;
    red[wv][1][c] = sx;
    __syncthreads();
    if (tid < 128) {
        const int which = tid >> 6;
        const float tsum = red[0][which][c] + red[1][which][c] + red[2][which][c] + red[3][which][c];
        if (acc_rows) acc_rows[((size_t)(blockIdx.z * gridDim.x + blockIdx.x) * 2 + which) * C + c0 + c] = (double)tsum;
        else atomicAdd(&acc[which * C + c0 + c], (double)tsum);
    }
    bn_tail(tail, gridDim.x * gridDim.y * gridDim.z, tid, 256, &tail_flag);
}

// deterministic mode: acc[e] = sum over the partial rows in index order, e < 2 C
__global__ void __launch_bounds__(256)
pw_colsum_reduce_kernel(const double *__restrict__ rows, int nrows, int C2, double *__restrict__ acc)
{
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= C2) return;
    double s = 0.0;
    for (int r = 0; r < nrows; ++r) s += rows[(size_t)r * C2 + e];
    acc[e] = s;
}

// dW[e] = sum over the point splits in a FIXED order (the weight gradient is reproducible bit for bit): the splits in four
// consecutive quarters, each summed in split order by its own wave, then ((q0 + q1) + q2) + q3.  (One thread per element walking all
// 64 - 128 splits was a chain of dependent trips to memory on 128 workgroups: 10 - 20 us per launch, six launches per CAPL step.)
__global__ void __launch_bounds__(256)
pw_wgrad_reduce_kernel(const float *__restrict__ part, int nsplit, int n, float *__restrict__ dw)
{
    __shared__ f32x4 quarter[3][64];
    const int eg = threadIdx.x & 63, sg = threadIdx.x >> 6;
    const int e = 4 * (blockIdx.x * 64 + eg);
    const int per = (nsplit + 3) / 4, q0 = sg * per, q1 = min(nsplit, q0 + per);
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    if (e < n) {
        int q = q0;
        for (; q + 4 <= q1; q += 4) {                         // four loads in flight, added in order
            const f32x4 a = ldg4(part + (size_t)q * n + e), b = ldg4(part + (size_t)(q + 1) * n + e);
            const f32x4 c = ldg4(part + (size_t)(q + 2) * n + e), d = ldg4(part + (size_t)(q + 3) * n + e);
            s = (((s + a) + b) + c) + d;
        }
        for (; q < q1; ++q) s = s + ldg4(part + (size_t)q * n + e);
    }
    if (sg) quarter[sg - 1][eg] = s;
    __syncthreads();
    if (sg == 0 && e < n) *reinterpret_cast<f32x4 *>(dw + e) = ((s + quarter[0][eg]) + quarter[1][eg]) + quarter[2][eg];
}

// the eval-mode constants of channel c of a C-wide layer: scale (h - (rm - b)) + beta
__device__ __forceinline__ void pw_bn_eval_one(int c, int C, const float *gamma, const float *beta, const float *rm, const float *rv, const float *bias,
                                               float eps, float *stats)
{
    const double bc = bias ? (double)bias[c] : 0.0;
    const double mean = (double)rm[c] - bc, var = rv[c];
    const float rstd = (float)(1.0 / sqrt(var + (double)eps));
    const float sc = gamma[c] * rstd;
    stats[PST_SCALE * C + c] = sc;
    stats[PST_SHIFT * C + c] = beta[c] - (float)mean * sc;
    stats[PST_MEAN * C + c] = (float)mean;
    stats[PST_RSTD * C + c] = rstd;
}
// batch statistics -> (scale, shift, mean, rstd) and the running-statistics update of nn.BatchNorm1d
__global__ void pw_bn_finalize_kernel(const double *__restrict__ acc, int C, double count, const float *__restrict__ gamma,
                                      const float *__restrict__ beta, float *__restrict__ rm, float *__restrict__ rv,
                                      float eps, float momentum, int training, float *__restrict__ stats,
                                      const float *__restrict__ bias)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    if (training) {
        pw_bn_finalize_one(c, C, acc[c], acc[C + c], count, gamma, beta, rm, rv, eps, momentum, bias, stats);
        return;
    }
    pw_bn_eval_one(c, C, gamma, beta, rm, rv, bias, eps, stats);      // running statistics
}

// BatchNorm backward coefficients: dh = a g + b + c h; parameter gradients
// acc: sums over the whole (possibly multi-rank) batch -> coefficients; loc: this rank's sums -> its parameter gradients
__global__ void pw_bwd_coef_kernel(const double *__restrict__ acc, const double *__restrict__ loc, int C, double count,
                                   const float *__restrict__ stats, float *__restrict__ coef, float *__restrict__ dgamma,
                                   float *__restrict__ dbeta)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    pw_bwd_coef_one(c, C, acc[c], acc[C + c], loc[c], loc[C + c], count, stats, coef, dgamma, dbeta);
}

// ------------------------------------------------------------------ host side
template <int BN, int PRO, int EPI, bool OUT_CF, int PREC, int AT, int HT, int OT>
int launch_rg6_t(const RowGemm &rg, hipStream_t st)
{
    const size_t lds = (size_t)(PW_BM + BN) * PW6_ROW;
    const int ncb = rg.Nc / BN, nrb = (rg.M + PW_BM - 1) / PW_BM;
    const unsigned grid = 8u * ncb * ((nrb + 7) / 8);
    hipLaunchKernelGGL((pw_rowgemm6_kernel<BN, PRO, EPI, OUT_CF, PREC, AT, HT, OT>), dim3(grid), dim3(256), lds, st, rg);
    GFS3D_LAUNCH_CHECK("pw_rowgemm6_kernel");
    return GFS3D_OK;
}

// The instantiation for (arithmetic, storage formats).  fp32 storage: every PREC.  16-bit storage (reduced modes, training):
//   forward  (EPI_RAW_STATS)   A = fp32 input (PRO_NONE only) or 16-bit in the mode's format, out = h in the mode's format
//   backward (PRO_BWD)         A = g (bf16), second tensor / eh = h (the mode's format), out = g (bf16) or dx (fp32 | bf16)
template <int BN, int PRO, int EPI, bool OUT_CF>
int launch_rowgemm6(const RowGemm &rg, hipStream_t st, const char *who)
{
    constexpr bool bwd = PRO == PRO_BWD || EPI == EPI_BWD_ACT;           // gradient GEMMs never run on half operands
    const int prec = (rg.prec == PREC_F16 && bwd) ? PREC_BF16 : rg.prec;
    if (rg.at == ST_F32 && rg.ht == ST_F32 && rg.ot == ST_F32) {
        if (prec == PREC_BF16) return launch_rg6_t<BN, PRO, EPI, OUT_CF, PREC_BF16, ST_F32, ST_F32, ST_F32>(rg, st);
        if constexpr (!bwd) { if (prec == PREC_F16) return launch_rg6_t<BN, PRO, EPI, OUT_CF, PREC_F16, ST_F32, ST_F32, ST_F32>(rg, st); }
        return launch_rg6_t<BN, PRO, EPI, OUT_CF, PREC_X6, ST_F32, ST_F32, ST_F32>(rg, st);
    }
    if constexpr (EPI == EPI_RAW_STATS && !OUT_CF) {
        if (prec == PREC_BF16 && rg.ot == ST_BF16) {
            if constexpr (PRO == PRO_NONE) { if (rg.at == ST_F32) return launch_rg6_t<BN, PRO, EPI, OUT_CF, PREC_BF16, ST_F32, ST_F32, ST_BF16>(rg, st); }
            if (rg.at == ST_BF16) return launch_rg6_t<BN, PRO, EPI, OUT_CF, PREC_BF16, ST_BF16, ST_F32, ST_BF16>(rg, st);
        }
        if (prec == PREC_F16 && rg.ot == ST_F16) {
            if constexpr (PRO == PRO_NONE) { if (rg.at == ST_F32) return launch_rg6_t<BN, PRO, EPI, OUT_CF, PREC_F16, ST_F32, ST_F32, ST_F16>(rg, st); }
            if (rg.at == ST_F16) return launch_rg6_t<BN, PRO, EPI, OUT_CF, PREC_F16, ST_F16, ST_F32, ST_F16>(rg, st);
        }
    }
    if constexpr (PRO == PRO_BWD && !OUT_CF) {
        if (prec == PREC_BF16 && rg.at == ST_BF16) {
            if constexpr (EPI == EPI_BWD_ACT) {
                if (rg.ot == ST_BF16 && rg.ht == ST_BF16) return launch_rg6_t<BN, PRO, EPI, OUT_CF, PREC_BF16, ST_BF16, ST_BF16, ST_BF16>(rg, st);
                if (rg.ot == ST_BF16 && rg.ht == ST_F16) return launch_rg6_t<BN, PRO, EPI, OUT_CF, PREC_BF16, ST_BF16, ST_F16, ST_BF16>(rg, st);
            }
            if constexpr (EPI == EPI_PLAIN) {
                if (rg.ot == ST_F32 && rg.ht == ST_BF16) return launch_rg6_t<BN, PRO, EPI, OUT_CF, PREC_BF16, ST_BF16, ST_BF16, ST_F32>(rg, st);
                if (rg.ot == ST_F32 && rg.ht == ST_F16) return launch_rg6_t<BN, PRO, EPI, OUT_CF, PREC_BF16, ST_BF16, ST_F16, ST_F32>(rg, st);
                if (rg.ot == ST_BF16 && rg.ht == ST_BF16) return launch_rg6_t<BN, PRO, EPI, OUT_CF, PREC_BF16, ST_BF16, ST_BF16, ST_BF16>(rg, st);
                if (rg.ot == ST_BF16 && rg.ht == ST_F16) return launch_rg6_t<BN, PRO, EPI, OUT_CF, PREC_BF16, ST_BF16, ST_F16, ST_BF16>(rg, st);
            }
        }
    }
    set_error("%s: no row GEMM for storage formats a=%d h=%d out=%d with arithmetic %d", who, rg.at, rg.ht, rg.ot, prec);
    return GFS3D_ERR_INVALID;
}

template <int BN, int PRO, int EPI, bool OUT_CF>
int launch_rowgemm(const RowGemm &rg, hipStream_t st, const char *who)
{
    if (rg.w3) return launch_rowgemm6<BN, PRO, EPI, OUT_CF>(rg, st, who);
    const size_t lds = sizeof(float) * 2 * (PW_BM + BN) * PW_LD;
    GFS3D_ENSURE_LDS((pw_rowgemm_kernel<BN, PRO, EPI, OUT_CF>), lds, who);
    const int ncb = rg.Nc / BN, nrb = (rg.M + PW_BM - 1) / PW_BM;
    const unsigned grid = 8u * ncb * ((nrb + 7) / 8);
    hipLaunchKernelGGL((pw_rowgemm_kernel<BN, PRO, EPI, OUT_CF>), dim3(grid), dim3(256), lds, st, rg);
    GFS3D_LAUNCH_CHECK("pw_rowgemm_kernel");
    return GFS3D_OK;
}

template <int PRO, int EPI, bool OUT_CF>
int rowgemm(const RowGemm &rg, hipStream_t st, const char *who)
{
#ifdef PW_ABL
    static const int force_bn = getenv("GFS3D_PW_BN") ? atoi(getenv("GFS3D_PW_BN")) : 0;      // experiments: GFS3D_PW_BN=64
#else
    constexpr int force_bn = 0;
#endif
    if (rg.Nc % 128 == 0 && force_bn != 64) return launch_rowgemm<128, PRO, EPI, OUT_CF>(rg, st, who);
    return launch_rowgemm<64, PRO, EPI, OUT_CF>(rg, st, who);
}

// split-K geometry of a weight-gradient GEMM: tile (bmu x bnv), point splits of `rows` rows (a multiple of 32)
struct WgGeo { int bmu, bnv, nsplit, rows; };
WgGeo wg_geo(int M, int Co, int Ci, bool allow192)
{
    WgGeo g{};
    g.bmu = Co % 128 == 0 ? 128 : 64;
    g.bnv = (allow192 && Ci == 192) ? 192 : (Ci % 128 == 0 ? 128 : 64);
    const int ntile = (Co / g.bmu) * (Ci / g.bnv);
    // ONE round of resident blocks: the 128 x 128 (and wider) tiles run two blocks per CU (their 200-odd registers, not the LDS,
    // decide), so 512 blocks; 768 - sized for the three blocks the LDS would admit - left half the card idle through a second round
    // (round 6: 752 blocks = 1.47 rounds in the c2 step).  Smaller tiles fit 3 - 4 per CU.
    const int per_cu = g.bmu * g.bnv >= 128 * 128 ? 2 : (g.bmu * g.bnv >= 128 * 64 ? 3 : 4);
    int nsplit = (256 * per_cu + ntile - 1) / ntile;
    int rows = (M + nsplit - 1) / nsplit;
    rows = (rows + 31) / 32 * 32;
    if (rows < 256) rows = 256;
    g.rows = rows;
    g.nsplit = (M + rows - 1) / rows;
    return g;
}
size_t wg_part_bytes(int M, int Co, int Ci, bool allow192)
{
    return sizeof(float) * (size_t)wg_geo(M, Co, Ci, allow192).nsplit * Co * Ci;
}

template <int BMU, int BNV, int VPRO, int GT, int HT, int VT>
int launch_wg16_t(const WGrad &wg, unsigned grid, hipStream_t st, const char *who)
{
    const size_t lds = (size_t)WG16_PTS * (wg16_prow(BMU) + wg16_prow(BNV));
    GFS3D_ENSURE_LDS((pw_wgrad16_kernel<BMU, BNV, VPRO, GT, HT, VT>), lds, who);
    hipLaunchKernelGGL((pw_wgrad16_kernel<BMU, BNV, VPRO, GT, HT, VT>), dim3(grid), dim3(256), lds, st, wg);
    GFS3D_LAUNCH_CHECK("pw_wgrad16_kernel");
    return GFS3D_OK;
}

// fp32-accurate mode: pw_wgrad6_kernel (exact bf16 split).  Reduced modes: pw_wgrad16_kernel; storage formats: all fp32, or
// (16-bit storage) u = g (bf16), u2 = h (the mode's format), v = the previous layer's h (VPRO == PRO_BNACT: the same format) or the
// stack's input (fp32 or the mode's format)
template <int BMU, int BNV, int VPRO>
int launch_wgrad_t(WGrad wg, const WgGeo &geo, hipStream_t st, const char *who)
{
    const size_t lds = (size_t)(BMU + BNV) * PW6_ROW;
    const int ntile = (wg.Co / BMU) * (wg.Ci / BNV);
    wg.rows_per_split = geo.rows;
    wg.nsplit = geo.nsplit;
    const unsigned grid = 8u * ntile * ((geo.nsplit + 7) / 8);
    int rc = -1;                                     // -1: no instantiation for these formats
    if (wg.prec == PREC_X6) {
        if (wg.gt == ST_F32 && wg.ht == ST_F32 && wg.vt == ST_F32) {
            GFS3D_ENSURE_LDS((pw_wgrad6_kernel<BMU, BNV, VPRO, PREC_X6>), lds, who);
            hipLaunchKernelGGL((pw_wgrad6_kernel<BMU, BNV, VPRO, PREC_X6>), dim3(grid), dim3(256), lds, st, wg);
            GFS3D_LAUNCH_CHECK("pw_wgrad6_kernel");
            rc = GFS3D_OK;
        }
    } else if (wg.gt == ST_F32 && wg.ht == ST_F32 && wg.vt == ST_F32) {
        rc = launch_wg16_t<BMU, BNV, VPRO, ST_F32, ST_F32, ST_F32>(wg, grid, st, who);
    } else if (wg.gt == ST_BF16 && (wg.ht == ST_BF16 || wg.ht == ST_F16)) {
        if (wg.vt == wg.ht) {
            if (wg.ht == ST_BF16) rc = launch_wg16_t<BMU, BNV, VPRO, ST_BF16, ST_BF16, ST_BF16>(wg, grid, st, who);
            else rc = launch_wg16_t<BMU, BNV, VPRO, ST_BF16, ST_F16, ST_F16>(wg, grid, st, who);
        } else if (wg.vt == ST_F32) {
            if constexpr (VPRO == PRO_NONE) {
                if (wg.ht == ST_BF16) rc = launch_wg16_t<BMU, BNV, VPRO, ST_BF16, ST_BF16, ST_F32>(wg, grid, st, who);
                else rc = launch_wg16_t<BMU, BNV, VPRO, ST_BF16, ST_F16, ST_F32>(wg, grid, st, who);
            }
        }
    }
    if (rc == -1) {
        set_error("%s: no weight-gradient GEMM for storage formats g=%d h=%d v=%d", who, wg.gt, wg.ht, wg.vt);
        return GFS3D_ERR_INVALID;
    }
    if (rc) return rc;
    const int n = wg.Co * wg.Ci;
    hipLaunchKernelGGL(pw_wgrad_reduce_kernel, dim3((n / 4 + 63) / 64), dim3(256), 0, st, wg.part, geo.nsplit, n, wg.dw);
    GFS3D_LAUNCH_CHECK("pw_wgrad_reduce_kernel");
    return GFS3D_OK;
}

// dW = T(U)^T T(V) with the tile shape wg_geo() picks; VPRO = the prologue of V
template <int VPRO>
int wgrad(const WGrad &wg, bool allow192, hipStream_t st, const char *who)
{
    const WgGeo geo = wg_geo(wg.M, wg.Co, wg.Ci, allow192);
    if (geo.bmu == 128) {
        if constexpr (VPRO == PRO_NONE) { if (geo.bnv == 192) return launch_wgrad_t<128, 192, VPRO>(wg, geo, st, who); }
        if (geo.bnv == 128) return launch_wgrad_t<128, 128, VPRO>(wg, geo, st, who);
        return launch_wgrad_t<128, 64, VPRO>(wg, geo, st, who);
    }
    if constexpr (VPRO == PRO_NONE) { if (geo.bnv == 192) return launch_wgrad_t<64, 192, VPRO>(wg, geo, st, who); }
    if (geo.bnv == 128) return launch_wgrad_t<64, 128, VPRO>(wg, geo, st, who);
    return launch_wgrad_t<64, 64, VPRO>(wg, geo, st, who);
}

// All weight images of one stack pass in ONE launch (a launch per layer and a memset cost ~5 us each on an idle stream, as much
// as the whole work): up to PW_MAX_SPLIT matrices, entry i covers elements [first[i], first[i + 1]); the same launch clears
// the BatchNorm accumulators of the pass.
constexpr int PW_MAX_SPLIT = 8;
struct SplitJob {
    const float *w[PW_MAX_SPLIT];
    unsigned short *w3[PW_MAX_SPLIT];
    int Nc[PW_MAX_SPLIT], K[PW_MAX_SPLIT];
    long long sn[PW_MAX_SPLIT], sk[PW_MAX_SPLIT];
    int first[PW_MAX_SPLIT + 1];
    int n;
    double *zero;
    int nzero;
    // eval forward: the BatchNorm constants of every layer from the running statistics (what pw_bn_finalize_kernel(training = 0) writes),
    // in the same launch - they depend on nothing the pass computes.  bn_n layers, entry l: C = Nc[l].
    int bn_n;
    const float *bn_gamma[PW_MAX_SPLIT], *bn_beta[PW_MAX_SPLIT], *bn_rm[PW_MAX_SPLIT], *bn_rv[PW_MAX_SPLIT], *bn_bias[PW_MAX_SPLIT];
    float *bn_stats[PW_MAX_SPLIT];
    float bn_eps;
};
template <int PREC>
__global__ void pw_split_all_kernel(const SplitJob job)
{
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    for (int z = e; z < job.nzero; z += gridDim.x * blockDim.x) job.zero[z] = 0.0;
    if (job.bn_n > 0 && blockIdx.x == 0)                 // (at most 8 x 1024 channels: one workgroup)
        for (int l = 0; l < job.bn_n; ++l)
            for (int c = threadIdx.x; c < job.Nc[l]; c += blockDim.x)
                pw_bn_eval_one(c, job.Nc[l], job.bn_gamma[l], job.bn_beta[l], job.bn_rm[l], job.bn_rv[l], job.bn_bias[l], job.bn_eps, job.bn_stats[l]);
    if (e >= job.first[job.n]) return;
    int i = 0;
#pragma unroll
    for (int q = 1; q < PW_MAX_SPLIT; ++q) i += (q < job.n && e >= job.first[q]) ? 1 : 0;
    const int r = e - job.first[i], K = job.K[i];
    const int n = r / K, k = r - n * K;
    mm_store1<PREC>(job.w[i][(size_t)n * job.sn[i] + (size_t)k * job.sk[i]], job.w3[i] + ((size_t)n * (K / 32) + k / 32) * 96 + (k & 31), 32);
}
static int split_all(const SplitJob &job, hipStream_t st, int prec)
{
    const int total = job.first[job.n] > job.nzero / 8 ? job.first[job.n] : job.nzero / 8;
    const dim3 grid((total + 255) / 256 > 0 ? (total + 255) / 256 : 1);
    if (prec == PREC_BF16) hipLaunchKernelGGL(pw_split_all_kernel<PREC_BF16>, grid, dim3(256), 0, st, job);
    else if (prec == PREC_F16) hipLaunchKernelGGL(pw_split_all_kernel<PREC_F16>, grid, dim3(256), 0, st, job);
    else hipLaunchKernelGGL(pw_split_all_kernel<PREC_X6>, grid, dim3(256), 0, st, job);
    GFS3D_LAUNCH_CHECK("pw_split_all_kernel");
    return GFS3D_OK;
}

int split_weights(const float *w, int Nc, int K, long long sn, long long sk, void *w3, hipStream_t st, int prec)
{
    const dim3 grid((Nc * K + 255) / 256);
    unsigned short *o = static_cast<unsigned short *>(w3);
    if (prec == PREC_BF16) hipLaunchKernelGGL(pw_split_w_kernel<PREC_BF16>, grid, dim3(256), 0, st, w, Nc, K, sn, sk, o);
    else if (prec == PREC_F16) hipLaunchKernelGGL(pw_split_w_kernel<PREC_F16>, grid, dim3(256), 0, st, w, Nc, K, sn, sk, o);
    else hipLaunchKernelGGL(pw_split_w_kernel<PREC_X6>, grid, dim3(256), 0, st, w, Nc, K, sn, sk, o);
    GFS3D_LAUNCH_CHECK("pw_split_w_kernel");
    return GFS3D_OK;
}

struct PwPlan {
    int L, M, in_width, maxw;
    size_t off_acc, off_coef, off_w3[GFS3D_PW_MAX_LAYERS], off_g[GFS3D_PW_MAX_LAYERS], off_part, off_rows, total;
    int nrows;                // partial rows of the ordered column sums (deterministic mode)
};

inline int pw_segw(int seg_width) { return seg_width > 0 ? seg_width : GFS3D_PW_SEG_WIDTH; }
// storage formats of a call: h[l] / the workspace gradients / the input segments
inline int pw_ht(const gfs3d_pointwise_args *a)
{
    if (!a->store16 || !a->training || a->compute == GFS3D_COMPUTE_F32) return ST_F32;
    return a->compute == GFS3D_COMPUTE_F16 ? ST_F16 : ST_BF16;
}
inline int pw_gt(const gfs3d_pointwise_args *a) { return pw_ht(a) == ST_F32 ? ST_F32 : ST_BF16; }
inline int pw_xt(const gfs3d_pointwise_args *a) { return a->x_type ? pw_ht(a) : ST_F32; }
inline float pw_slope(const gfs3d_pointwise_args *a, int l) { return a->per_layer_slope ? a->slope_l[l] : a->slope; }

PwPlan pw_plan(int B, int N, int n_seg, int seg_width, int n_layers, const int *width)
{
    PwPlan p{};
    p.L = n_layers;
    p.M = B * N;
    p.in_width = pw_segw(seg_width) * n_seg;
    p.maxw = 0;
    for (int l = 0; l < n_layers; ++l) p.maxw = width[l] > p.maxw ? width[l] : p.maxw;
    size_t o = 0;
    p.off_acc = o;  o = align_up(o + sizeof(double) * (2 * p.maxw + 1) * n_layers, 256);      // [layer][2][maxw], then one 8-byte ticket slot per layer (bn_tail)
    p.off_coef = o; o = align_up(o + sizeof(float) * 3 * p.maxw, 256);
    for (int l = 0; l < n_layers; ++l) {
        const int ci = l ? width[l - 1] : p.in_width;
        p.off_w3[l] = o; o = align_up(o + 6 * (size_t)ci * width[l], 256);      // bf16x3 image of W_l (or of W_l^T)
    }
    for (int l = 0; l < n_layers; ++l) { p.off_g[l] = o; o = align_up(o + sizeof(float) * (size_t)p.M * width[l], 256); }
    size_t part = 0;                                   // split-K partial tiles of the largest weight gradient
    for (int l = 0; l < n_layers; ++l) {
        const size_t b = wg_part_bytes(p.M, width[l], l ? width[l - 1] : p.in_width, l == 0);
        part = b > part ? b : part;
    }
    p.off_part = o; o = align_up(o + part, 256);
    const int nrb = (p.M + PW_BM - 1) / PW_BM, nin = B * (((N + 63) / 64 + PW_IN_TILES - 1) / PW_IN_TILES);
    p.nrows = nrb > nin ? nrb : nin;
    p.off_rows = o; o = align_up(o + sizeof(double) * (size_t)p.nrows * 2 * p.maxw, 256);
    p.total = o;
    return p;
}

int pw_check(const gfs3d_pointwise_args *a, const char *who, bool need_y)
{
    GFS3D_REQUIRE(a, "%s: NULL args", who);
    GFS3D_REQUIRE(a->B >= 1 && a->N >= 1, "%s: bad shape B=%d N=%d", who, a->B, a->N);
    GFS3D_REQUIRE((long long)a->B * a->N < (1ll << 31), "%s: B*N too large", who);
    GFS3D_REQUIRE(a->n_seg >= 1 && a->n_seg <= GFS3D_PW_MAX_SEGS, "%s: n_seg=%d unsupported (1..%d)", who, a->n_seg, GFS3D_PW_MAX_SEGS);
    GFS3D_REQUIRE(a->n_layers >= 1 && a->n_layers <= GFS3D_PW_MAX_LAYERS, "%s: n_layers=%d unsupported (1..%d)", who, a->n_layers,
                  GFS3D_PW_MAX_LAYERS);
    GFS3D_REQUIRE(a->seg_width >= 0 && pw_segw(a->seg_width) % 64 == 0 && pw_segw(a->seg_width) * a->n_seg <= 1024,
                  "%s: seg_width=%d unsupported (multiple of 64, seg_width * n_seg <= 1024)", who, a->seg_width);
    for (int s = 0; s < a->n_seg; ++s) GFS3D_REQUIRE(a->x_seg[s], "%s: NULL x_seg[%d]", who, s);
    for (int l = 0; l < a->n_layers; ++l) {
        GFS3D_REQUIRE(a->width[l] >= 64 && a->width[l] % 64 == 0 && a->width[l] <= 1024,
                      "%s: layer width %d unsupported (multiple of 64, <= 1024)", who, a->width[l]);
        GFS3D_REQUIRE(a->w[l] && a->bn_w[l] && a->bn_b[l] && a->stats[l], "%s: NULL parameter of layer %d", who, l);
        GFS3D_REQUIRE(a->training || (a->bn_rm[l] && a->bn_rv[l]), "%s: eval mode needs running statistics (layer %d)", who, l);
        GFS3D_REQUIRE(a->h[l] || (!a->training && l == a->n_layers - 1), "%s: NULL h[%d]", who, l);
    }
    GFS3D_REQUIRE(a->y_cf || a->y_pm || !need_y, "%s: NULL y_cf and y_pm", who);
    GFS3D_REQUIRE(a->workspace, "%s: NULL workspace", who);
    GFS3D_REQUIRE(a->compute == GFS3D_COMPUTE_F32 || a->compute == GFS3D_COMPUTE_BF16 || a->compute == GFS3D_COMPUTE_F16,
                  "%s: compute must be one of GFS3D_COMPUTE_*", who);
    GFS3D_REQUIRE(a->x_type == 0 || (a->x_type == a->compute && pw_ht(a) != ST_F32),
                  "%s: x_type=%d needs store16, training mode and x_type == compute (a reduced mode)", who, a->x_type);
    return GFS3D_OK;
}

void fill_a(RowGemm &rg, const gfs3d_pointwise_args *a, int l, const void *prev)
{
    if (l == 0) {
        for (int s = 0; s < a->n_seg; ++s) rg.a[s] = a->x_seg[s];
        rg.a_segw = pw_segw(a->seg_width);
        rg.lda = pw_segw(a->seg_width);
        rg.K = pw_segw(a->seg_width) * a->n_seg;
        rg.at = pw_xt(a);
    } else {
        rg.at = pw_ht(a);
        rg.a[0] = prev;
        rg.a_segw = a->width[l - 1];
        rg.lda = a->width[l - 1];
        rg.K = a->width[l - 1];
    }
}

}  // namespace

int pw_plain_gemm(const float *a, int lda, const float *w, int M, int K, int Nc, float *const *out, int o_segw,
                  const float *ea, const float *eb, hipStream_t st, const char *who, int w_pq, double *zero, int nzero)
{
    if (K % PW_BK || Nc % 64 || o_segw % 64 || Nc / o_segw > 4 || Nc % o_segw) {
        set_error("%s: pw_plain_gemm: unsupported shape K=%d Nc=%d segw=%d", who, K, Nc, o_segw);
        return GFS3D_ERR_INVALID;
    }
    RowGemm rg{};
    rg.a[0] = a; rg.a_segw = K; rg.lda = lda; rg.K = K;
    rg.w = w; rg.M = M; rg.Nc = Nc;
    rg.w_pq = w_pq; rg.zero = zero; rg.nzero = nzero;
    for (int s = 0; s < Nc / o_segw; ++s) rg.out[s] = out[s];
    rg.o_segw = o_segw;
    if (ea && eb) {                                  // affine epilogue, identity activation
        rg.ea = ea; rg.eb = eb; rg.slope = 1.f;
        return rowgemm<PRO_NONE, EPI_AFFINE_ACT, false>(rg, st, who);
    }
    return rowgemm<PRO_NONE, EPI_PLAIN, false>(rg, st, who);
}
}  // namespace gfs3d

using namespace gfs3d;

extern "C" size_t gfs3d_pointwise_workspace_bytes(int B, int N, int n_seg, int n_layers, const int *width)
{
    if (B < 1 || N < 1 || n_seg < 1 || n_seg > GFS3D_PW_MAX_SEGS || n_layers < 1 || n_layers > GFS3D_PW_MAX_LAYERS || !width) return 0;
    return pw_plan(B, N, n_seg, 0, n_layers, width).total;
}

extern "C" size_t gfs3d_pointwise_workspace_bytes_ex(int B, int N, int n_seg, int seg_width, int n_layers, const int *width)
{
    if (B < 1 || N < 1 || n_seg < 1 || n_seg > GFS3D_PW_MAX_SEGS || n_layers < 1 || n_layers > GFS3D_PW_MAX_LAYERS || !width) return 0;
    if (seg_width < 0 || pw_segw(seg_width) % 64 || pw_segw(seg_width) * n_seg > 1024) return 0;
    return pw_plan(B, N, n_seg, seg_width, n_layers, width).total;
}

// Forward in phases 0 .. L (L = n_layers); training mode: phase p < L runs the GEMM of layer p (after finalising the
// BatchNorm of layer p - 1 from the sums in acc row p - 1) and leaves the column sums of layer p in acc row p; phase L
// finalises the last BatchNorm and writes the channel-first result.  acc: [L][2][maxw] doubles - the caller's buffer under
// synchronised BatchNorm (it all-reduces row p between phases p and p + 1), else workspace.
static int pw_forward_phase(const gfs3d_pointwise_args *a, int phase, double *acc_user, double stat_count, hipStream_t st, const char *who)
{
    const PwPlan p = pw_plan(a->B, a->N, a->n_seg, a->seg_width, a->n_layers, a->width);
    char *ws = static_cast<char *>(a->workspace);
    double *acc = acc_user ? acc_user : reinterpret_cast<double *>(ws + p.off_acc);
    const int L = a->n_layers, M = p.M;
    const double count = stat_count > 0.0 ? stat_count : (double)M;
    // phase 0: the weight images of every layer and (training) cleared accumulators, one launch
    const bool split_once = L <= PW_MAX_SPLIT;
    if (phase == 0) {
        if (split_once) {
            SplitJob job{};
            job.n = L;
            for (int l = 0; l < L; ++l) {
                const int K = l ? a->width[l - 1] : p.in_width;
                job.w[l] = a->w[l]; job.w3[l] = reinterpret_cast<unsigned short *>(ws + p.off_w3[l]);
                job.Nc[l] = a->width[l]; job.K[l] = K; job.sn[l] = K; job.sk[l] = 1;
                job.first[l + 1] = job.first[l] + a->width[l] * K;
            }
            job.zero = a->training ? acc : nullptr;
            job.nzero = a->training ? (acc_user ? 2 * p.maxw * L : (2 * p.maxw + 1) * L) : 0;      // (+ the ticket slots of bn_tail)
            if (!a->training) {                          // eval: every layer's BatchNorm constants in the same launch
                job.bn_n = L;
                job.bn_eps = a->eps;
                for (int l = 0; l < L; ++l) {
                    job.bn_gamma[l] = a->bn_w[l]; job.bn_beta[l] = a->bn_b[l]; job.bn_rm[l] = a->bn_rm[l]; job.bn_rv[l] = a->bn_rv[l];
                    job.bn_bias[l] = a->bias[l]; job.bn_stats[l] = a->stats[l];
                }
            }
            if (int rc = split_all(job, st, prec_of(a->compute, false))) return rc;
        } else if (a->training && hipMemsetAsync(acc, 0, sizeof(double) * 2 * p.maxw * L, st) != hipSuccess) {
            set_error("%s: hipMemsetAsync failed", who);
            return GFS3D_ERR_HIP;
        }
    }
    // monolithic call in atomics mode: the producing GEMM's last workgroup finalises its BatchNorm (bn_tail), no launch for it
    const bool use_tail = g_bn_tail && a->training && !acc_user && !g_deterministic && split_once;
    unsigned *tickets = reinterpret_cast<unsigned *>(acc + (size_t)2 * p.maxw * L);          // slot 2 l: layer l
    if (a->training && phase > 0 && !use_tail) {         // BatchNorm of layer phase - 1: batch statistics are complete (and global)
        const int l = phase - 1, C = a->width[l];
        hipLaunchKernelGGL(pw_bn_finalize_kernel, dim3((C + 63) / 64), dim3(64), 0, st, acc + (size_t)2 * p.maxw * l, C, count, a->bn_w[l], a->bn_b[l],
                           a->bn_rm[l], a->bn_rv[l], a->eps, a->momentum, 1, a->stats[l], a->bias[l]);
        GFS3D_LAUNCH_CHECK("pw_bn_finalize_kernel");
    }
    if (phase < L) {
        const int l = phase, C = a->width[l];
        double *acc_l = acc + (size_t)2 * p.maxw * l;
        RowGemm rg{};
        fill_a(rg, a, l, l ? a->h[l - 1] : nullptr);
        rg.w = a->w[l];
        rg.w3 = ws + p.off_w3[l];
        rg.prec = prec_of(a->compute, false);
        if (!split_once)
            if (int rc = split_weights(a->w[l], C, rg.K, rg.K, 1, ws + p.off_w3[l], st, rg.prec)) return rc;
        rg.M = M;
        rg.Nc = C;
        rg.slope = a->training ? (l ? pw_slope(a, l - 1) : 1.f) : pw_slope(a, l);    // training: the prologue's activation; eval: the epilogue's
        rg.out[0] = a->h[l];
        rg.o_segw = C;
        rg.ot = a->training ? pw_ht(a) : ST_F32;
        if (l == 0 && a->training && a->x16 && pw_ht(a) != ST_F32 && !a->x_type) rg.a16_out = a->x16;
        rg.acc = acc_l;
        rg.acc_rows = (a->training && g_deterministic) ? reinterpret_cast<double *>(ws + p.off_rows) : nullptr;
        if (use_tail) {
            BnTail &tl = rg.tail;
            tl.counter = tickets + 2 * l; tl.kind = 1; tl.C = C; tl.count = count; tl.acc = acc_l;
            tl.gamma = a->bn_w[l]; tl.beta = a->bn_b[l]; tl.bias = a->bias[l]; tl.rm = a->bn_rm[l]; tl.rv = a->bn_rv[l];
            tl.eps = a->eps; tl.momentum = a->momentum; tl.stats = a->stats[l];
        }
        int rc;
        if (a->training) {
            if (l == 0) rc = rowgemm<PRO_NONE, EPI_RAW_STATS, false>(rg, st, who);
            else {
                rg.pa = a->stats[l - 1] + PST_SCALE * a->width[l - 1];
                rg.pb = a->stats[l - 1] + PST_SHIFT * a->width[l - 1];
                rc = rowgemm<PRO_BNACT, EPI_RAW_STATS, false>(rg, st, who);
            }
            if (rc) return rc;
            if (rg.acc_rows) {
                hipLaunchKernelGGL(pw_colsum_reduce_kernel, dim3((2 * C + 255) / 256), dim3(256), 0, st, rg.acc_rows, (M + PW_BM - 1) / PW_BM, 2 * C, acc_l);
                GFS3D_LAUNCH_CHECK("pw_colsum_reduce_kernel");
            }
        } else {
            if (!split_once) {                           // (more layers than one split launch takes: the constants layer by layer)
                hipLaunchKernelGGL(pw_bn_finalize_kernel, dim3((C + 63) / 64), dim3(64), 0, st, acc, C, (double)M, a->bn_w[l], a->bn_b[l],
                                   a->bn_rm[l], a->bn_rv[l], a->eps, a->momentum, 0, a->stats[l], a->bias[l]);
                GFS3D_LAUNCH_CHECK("pw_bn_finalize_kernel");
            }
            rg.ea = a->stats[l] + PST_SCALE * C;
            rg.eb = a->stats[l] + PST_SHIFT * C;
            if (l == L - 1 && a->y_pm) {                 // point-major result (+ a transposed copy if both are wanted)
                rg.out[0] = a->y_pm;
                rc = rowgemm<PRO_NONE, EPI_AFFINE_ACT, false>(rg, st, who);
                if (!rc && a->y_cf) {
                    hipLaunchKernelGGL(pw_out_kernel<ST_F32>, dim3((a->N + 63) / 64, C / 64, a->B), dim3(256), 0, st, (const void *)a->y_pm, (const float *)nullptr,
                                       a->N, C, 1.f, a->y_cf, (float *)nullptr);
                    GFS3D_LAUNCH_CHECK("pw_out_kernel");
                }
            } else if (l == L - 1) {
                rg.out_cf = a->y_cf;
                rg.npts = a->N;
                rc = rowgemm<PRO_NONE, EPI_AFFINE_ACT, true>(rg, st, who);
            } else
                rc = rowgemm<PRO_NONE, EPI_AFFINE_ACT, false>(rg, st, who);
            if (rc) return rc;
        }
    } else if (a->training) {
        const int C = a->width[L - 1];
        const dim3 og((a->N + 63) / 64, C / 64, a->B);
        const void *hl = a->h[L - 1];
        const int ht = pw_ht(a);
        if (ht == ST_F16) hipLaunchKernelGGL(pw_out_kernel<ST_F16>, og, dim3(256), 0, st, hl, a->stats[L - 1], a->N, C, pw_slope(a, L - 1), a->y_cf, a->y_pm);
        else if (ht == ST_BF16) hipLaunchKernelGGL(pw_out_kernel<ST_BF16>, og, dim3(256), 0, st, hl, a->stats[L - 1], a->N, C, pw_slope(a, L - 1), a->y_cf, a->y_pm);
        else hipLaunchKernelGGL(pw_out_kernel<ST_F32>, og, dim3(256), 0, st, hl, a->stats[L - 1], a->N, C, pw_slope(a, L - 1), a->y_cf, a->y_pm);
        GFS3D_LAUNCH_CHECK("pw_out_kernel");
    }
    return GFS3D_OK;
}

static int pw_fwd_common(const gfs3d_pointwise_args *a, const char *who)
{
    if (int rc = pw_check(a, who, true)) return rc;
    const PwPlan p = pw_plan(a->B, a->N, a->n_seg, a->seg_width, a->n_layers, a->width);
    if (a->workspace_bytes < p.total) {
        set_error("%s: workspace too small (%zu < %zu)", who, a->workspace_bytes, p.total);
        return GFS3D_ERR_WORKSPACE;
    }
    return GFS3D_OK;
}

extern "C" int gfs3d_pointwise_forward_f32(const gfs3d_pointwise_args *a, void *stream)
{
    const char *who = "gfs3d_pointwise_forward_f32";
    if (int rc = pw_fwd_common(a, who)) return rc;
    hipStream_t st = static_cast<hipStream_t>(stream);
    ProfScope prof(6, st);
    for (int ph = 0; ph <= a->n_layers; ++ph)
        if (int rc = pw_forward_phase(a, ph, nullptr, 0.0, st, who)) return rc;
    return GFS3D_OK;
}

extern "C" int gfs3d_pointwise_forward_phase_f32(const gfs3d_pointwise_args *a, int phase, double *acc, double stat_count, void *stream)
{
    const char *who = "gfs3d_pointwise_forward_phase_f32";
    if (int rc = pw_fwd_common(a, who)) return rc;
    GFS3D_REQUIRE(phase >= 0 && phase <= a->n_layers, "%s: phase must be 0 .. n_layers", who);
    GFS3D_REQUIRE(acc != nullptr, "%s: NULL acc", who);
    hipStream_t st = static_cast<hipStream_t>(stream);
    ProfScope prof(6, st);
    return pw_forward_phase(a, phase, acc, stat_count, st, who);
}

// Backward in phases 0 .. L: phase 0 turns dy into the gradient of the last pre-activation and its BatchNorm-backward sums
// (acc row L - 1); phase q >= 1 handles layer l = L - q: BatchNorm-backward coefficients from acc row l (global sums;
// the parameter gradients use this rank's own sums, kept in loc row l), dW_l, and the gradient of layer l - 1 with ITS sums
// (acc row l - 1) - or dx for l = 0.  acc / loc: [L][2][maxw] doubles each.
static int pw_backward_phase(const gfs3d_pointwise_bwd_args *ba, int phase, double *acc_user, double stat_count, hipStream_t st, const char *who)
{
    const gfs3d_pointwise_args *a = &ba->f;
    const PwPlan p = pw_plan(a->B, a->N, a->n_seg, a->seg_width, a->n_layers, a->width);
    char *ws = static_cast<char *>(a->workspace);
    const int L = a->n_layers, M = p.M;
    double *acc = acc_user ? acc_user : reinterpret_cast<double *>(ws + p.off_acc);
    double *loc = acc_user ? acc_user + (size_t)2 * p.maxw * L : acc;          // monolithic: this rank's sums are the global sums
    const bool split_once = L <= PW_MAX_SPLIT;
    // monolithic call in atomics mode: the kernel that completes the sums of layer l computes that layer's coefficients and
    // BatchNorm parameter gradients in its last workgroup (bn_tail), no pw_bwd_coef_kernel launch
    const bool use_tail = g_bn_tail && !acc_user && !g_deterministic && split_once;
    unsigned *tickets = reinterpret_cast<unsigned *>(acc + (size_t)2 * p.maxw * L);
    float *coef = reinterpret_cast<float *>(ws + p.off_coef);
    const double count_ = stat_count > 0.0 ? stat_count : (double)M;
    auto coef_tail = [&](int l) {                     // sums of layer l -> coef, dgamma_l, dbeta_l
        BnTail tl{};
        if (!use_tail) return tl;
        tl.counter = tickets + 2 * l; tl.kind = 2; tl.C = a->width[l]; tl.count = count_; tl.acc = acc + (size_t)2 * p.maxw * l;
        tl.stats = a->stats[l]; tl.coef = coef; tl.dgamma = ba->dbn_w[l]; tl.dbeta = ba->dbn_b[l];
        return tl;
    };
    const double count = stat_count > 0.0 ? stat_count : (double)M;
    auto keep_local = [&](int l) -> int {
        if (!acc_user) return GFS3D_OK;
        if (hipMemcpyAsync(loc + (size_t)2 * p.maxw * l, acc + (size_t)2 * p.maxw * l, sizeof(double) * 2 * p.maxw, hipMemcpyDeviceToDevice, st) != hipSuccess) {
            set_error("%s: hipMemcpyAsync failed", who);
            return GFS3D_ERR_HIP;
        }
        return GFS3D_OK;
    };
    if (phase == 0) {
        const int C = a->width[L - 1];
        if (split_once) {                                  // the transposed weight images of every layer + cleared accumulators, one launch
            SplitJob job{};
            job.n = L;
            for (int l = 0; l < L; ++l) {
                const int Ci = l ? a->width[l - 1] : p.in_width;
                job.w[l] = a->w[l]; job.w3[l] = reinterpret_cast<unsigned short *>(ws + p.off_w3[l]);
                job.Nc[l] = Ci; job.K[l] = a->width[l]; job.sn[l] = 1; job.sk[l] = Ci;          // pieces of W_l^T (Ci, C)
                job.first[l + 1] = job.first[l] + a->width[l] * Ci;
            }
            job.zero = acc;
            job.nzero = acc_user ? 2 * p.maxw * L : (2 * p.maxw + 1) * L;                       // (+ the ticket slots of bn_tail)
            if (int rc = split_all(job, st, prec_of(a->compute, true))) return rc;
        } else if (hipMemsetAsync(acc, 0, sizeof(double) * 2 * p.maxw * L, st) != hipSuccess) { set_error("%s: hipMemsetAsync failed", who); return GFS3D_ERR_HIP; }
        double *rows = g_deterministic ? reinterpret_cast<double *>(ws + p.off_rows) : nullptr;
        const int gx = ((a->N + 63) / 64 + PW_IN_TILES - 1) / PW_IN_TILES;
        {
            const dim3 ig(gx, C / 64, a->B);
            const void *hl = a->h[L - 1];
            void *gl_ = ws + p.off_g[L - 1];
            double *accl = acc + (size_t)2 * p.maxw * (L - 1);
            const int ht = pw_ht(a);
            if (ht == ST_F16)
                hipLaunchKernelGGL((pw_bwd_in_kernel<ST_F16, ST_BF16>), ig, dim3(256), 0, st, ba->dy_cf, ba->dy_pm, hl, a->stats[L - 1], a->N, C,
                                   pw_slope(a, L - 1), gl_, accl, rows, coef_tail(L - 1));
            else if (ht == ST_BF16)
                hipLaunchKernelGGL((pw_bwd_in_kernel<ST_BF16, ST_BF16>), ig, dim3(256), 0, st, ba->dy_cf, ba->dy_pm, hl, a->stats[L - 1], a->N, C,
                                   pw_slope(a, L - 1), gl_, accl, rows, coef_tail(L - 1));
            else
                hipLaunchKernelGGL((pw_bwd_in_kernel<ST_F32, ST_F32>), ig, dim3(256), 0, st, ba->dy_cf, ba->dy_pm, hl, a->stats[L - 1], a->N, C,
                                   pw_slope(a, L - 1), gl_, accl, rows, coef_tail(L - 1));
        }
        GFS3D_LAUNCH_CHECK("pw_bwd_in_kernel");
        if (rows) {
            hipLaunchKernelGGL(pw_colsum_reduce_kernel, dim3((2 * C + 255) / 256), dim3(256), 0, st, rows, gx * a->B, 2 * C, acc + (size_t)2 * p.maxw * (L - 1));
            GFS3D_LAUNCH_CHECK("pw_colsum_reduce_kernel");
        }
        return keep_local(L - 1);
    }
    const int l = L - phase;
    const int C = a->width[l], Ci = l ? a->width[l - 1] : p.in_width;
    void *gl = ws + p.off_g[l];
    void *w3t = ws + p.off_w3[l];
    const int ht = pw_ht(a), gt = pw_gt(a);
    if (!use_tail) {
        hipLaunchKernelGGL(pw_bwd_coef_kernel, dim3((C + 63) / 64), dim3(64), 0, st, acc + (size_t)2 * p.maxw * l, loc + (size_t)2 * p.maxw * l, C, count,
                           a->stats[l], coef, ba->dbn_w[l], ba->dbn_b[l]);
        GFS3D_LAUNCH_CHECK("pw_bwd_coef_kernel");
    }
    // dW_l: per-split partial tiles, added in split order (no atomics, no memset)
    WGrad wg{};
    wg.part = reinterpret_cast<float *>(ws + p.off_part);
    wg.u = gl; wg.u2 = a->h[l]; wg.ldu = C;
    wg.ua = coef; wg.ub = coef + C; wg.uc = coef + 2 * C;
    wg.slope = l ? pw_slope(a, l - 1) : 1.f;
    wg.M = M; wg.Co = C; wg.Ci = Ci; wg.dw = ba->dw[l];
    wg.prec = prec_of(a->compute, true);
    wg.gt = gt; wg.ht = ht; wg.vt = l ? ht : pw_xt(a);
    int rc;
    if (l == 0 && a->x16 && ht != ST_F32 && !a->x_type) {        // the forward's 16-bit copy of the fp32 input (one matrix, all segments)
        wg.v[0] = a->x16; wg.v_segw = p.in_width; wg.ldv = p.in_width; wg.vt = ht;
        rc = wgrad<PRO_NONE>(wg, true, st, who);
    } else if (l == 0) {
        for (int s = 0; s < a->n_seg; ++s) wg.v[s] = a->x_seg[s];
        wg.v_segw = pw_segw(a->seg_width); wg.ldv = pw_segw(a->seg_width);
        // all of a 192-channel input in one tile: dH (two tensors + the BN-backward combine) is then read once
        rc = wgrad<PRO_NONE>(wg, true, st, who);
    } else {
        wg.v[0] = a->h[l - 1]; wg.v_segw = Ci; wg.ldv = Ci;
        wg.va = a->stats[l - 1] + PST_SCALE * Ci; wg.vb = a->stats[l - 1] + PST_SHIFT * Ci;
        rc = wgrad<PRO_BNACT>(wg, false, st, who);
    }
    if (rc) return rc;
    // dA_{l-1} = dH_l W_l
    if (!split_once)
        if (int rc2 = split_weights(a->w[l], Ci, C, 1, Ci, w3t, st, prec_of(a->compute, true))) return rc2;      // pieces of W_l^T (Ci, C)
    RowGemm rg{};
    rg.a[0] = gl; rg.a2[0] = a->h[l]; rg.a_segw = C; rg.lda = C; rg.K = C;
    rg.pa = coef; rg.pb = coef + C; rg.pc = coef + 2 * C;
    rg.w3 = w3t;
    rg.prec = prec_of(a->compute, true);
    rg.M = M; rg.Nc = Ci; rg.slope = l ? pw_slope(a, l - 1) : 1.f;
    rg.at = gt; rg.ht = ht;
    if (l == 0) {
        for (int s = 0; s < a->n_seg; ++s) rg.out[s] = ba->dx_seg[s];
        rg.o_segw = pw_segw(a->seg_width);
        rg.ot = ba->dx_type ? ST_BF16 : ST_F32;
        // the 128-wide tile where the input width allows it: the gradient rows (two tensors) are fetched once per COLUMN block, and its
        // epilogue stores whole rows 16 / 8 bytes per thread (the 64-wide tile: 4 / 2)
        if (Ci % 128 == 0) return launch_rowgemm<128, PRO_BWD, EPI_PLAIN, false>(rg, st, who);
        return launch_rowgemm<64, PRO_BWD, EPI_PLAIN, false>(rg, st, who);
    }
    rg.out[0] = ws + p.off_g[l - 1];
    rg.ot = gt;
    rg.o_segw = Ci;
    rg.ea = a->stats[l - 1] + PST_SCALE * Ci;
    rg.eb = a->stats[l - 1] + PST_SHIFT * Ci;
    rg.eh = a->h[l - 1];
    rg.acc = acc + (size_t)2 * p.maxw * (l - 1);
    rg.acc_rows = g_deterministic ? reinterpret_cast<double *>(ws + p.off_rows) : nullptr;
    rg.tail = coef_tail(l - 1);
    if (int rc3 = rowgemm<PRO_BWD, EPI_BWD_ACT, false>(rg, st, who)) return rc3;
    if (rg.acc_rows) {
        hipLaunchKernelGGL(pw_colsum_reduce_kernel, dim3((2 * Ci + 255) / 256), dim3(256), 0, st, rg.acc_rows, (M + PW_BM - 1) / PW_BM, 2 * Ci, rg.acc);
        GFS3D_LAUNCH_CHECK("pw_colsum_reduce_kernel");
    }
    return keep_local(l - 1);
}

static int pw_bwd_common(const gfs3d_pointwise_bwd_args *ba, const char *who)
{
    GFS3D_REQUIRE(ba, "%s: NULL args", who);
    const gfs3d_pointwise_args *a = &ba->f;
    if (int rc = pw_check(a, who, false)) return rc;
    GFS3D_REQUIRE(a->training, "%s: backward is implemented for training mode (batch statistics)", who);
    GFS3D_REQUIRE(ba->dy_cf || ba->dy_pm, "%s: NULL dy_cf and dy_pm", who);
    GFS3D_REQUIRE(ba->dx_type == 0 || (ba->dx_type == 1 && pw_ht(a) != ST_F32), "%s: dx_type=%d needs store16 in a reduced mode", who, ba->dx_type);
    for (int s = 0; s < a->n_seg; ++s) GFS3D_REQUIRE(ba->dx_seg[s], "%s: NULL dx_seg[%d]", who, s);
    for (int l = 0; l < a->n_layers; ++l)
        GFS3D_REQUIRE(ba->dw[l] && ba->dbn_w[l] && ba->dbn_b[l], "%s: NULL gradient buffer of layer %d", who, l);
    const PwPlan p = pw_plan(a->B, a->N, a->n_seg, a->seg_width, a->n_layers, a->width);
    if (a->workspace_bytes < p.total) {
        set_error("%s: workspace too small (%zu < %zu)", who, a->workspace_bytes, p.total);
        return GFS3D_ERR_WORKSPACE;
    }
    return GFS3D_OK;
}

extern "C" int gfs3d_pointwise_backward_f32(const gfs3d_pointwise_bwd_args *ba, void *stream)
{
    const char *who = "gfs3d_pointwise_backward_f32";
    if (int rc = pw_bwd_common(ba, who)) return rc;
    hipStream_t st = static_cast<hipStream_t>(stream);
    ProfScope prof(7, st);
    for (int ph = 0; ph <= ba->f.n_layers; ++ph)
        if (int rc = pw_backward_phase(ba, ph, nullptr, 0.0, st, who)) return rc;
    return GFS3D_OK;
}

extern "C" int gfs3d_pointwise_backward_phase_f32(const gfs3d_pointwise_bwd_args *ba, int phase, double *acc, double stat_count, void *stream)
{
    const char *who = "gfs3d_pointwise_backward_phase_f32";
    if (int rc = pw_bwd_common(ba, who)) return rc;
    GFS3D_REQUIRE(phase >= 0 && phase <= ba->f.n_layers, "%s: phase must be 0 .. n_layers", who);
    GFS3D_REQUIRE(acc != nullptr, "%s: NULL acc", who);
    hipStream_t st = static_cast<hipStream_t>(stream);
    ProfScope prof(7, st);
    return pw_backward_phase(ba, phase, acc, stat_count, st, who);
}

// ---------------------------------------------------------------------------------------------------------
// Channel concatenation into one point-major matrix (and its adjoint): 64 points x 64 columns per block through an LDS
// tile, so that both the channel-first side (lanes along the points) and the point-major side (lanes along the columns)
// are coalesced.  grid (ceil(N / 64), column tiles, B); the column tiles enumerate (source, 64-channel slice) pairs and
// then the zero padding up to dst_ld.
// ---------------------------------------------------------------------------------------------------------
namespace gfs3d {
namespace {
struct PackPlan {
    int n_src, tiles[GFS3D_PACK_MAX_SRC + 1], off[GFS3D_PACK_MAX_SRC + 1], total_tiles, used;
};

template <bool UNPACK, int DT>
__global__ void __launch_bounds__(256)
pw_pack_kernel(const gfs3d_pack_args a, const PackPlan pl)
{
    __shared__ float tile[64][65];
    const int tid = threadIdx.x, n0 = blockIdx.x * 64, b = blockIdx.z, N = a.N;
    int s = 0, ty = blockIdx.y;
    while (s < pl.n_src && ty >= pl.tiles[s]) { ty -= pl.tiles[s]; ++s; }
    const int lane = tid & 63, wv = tid >> 6;
    if (s == pl.n_src) {                                   // padding columns
        if (UNPACK) return;
        const int col = pl.used + ty * 64 + lane;
        if (col < a.dst_ld)
            for (int p = wv; p < 64; p += 4)
                if (n0 + p < N) st_st1<DT>(a.dst, ((size_t)b * N + n0 + p) * a.dst_ld + col, 0.f);
        return;
    }
    const int W = a.width[s], c0 = ty * 64, cw = min(64, W - c0), dcol = pl.off[s] + c0;
    float *src = a.src[s];
    if (UNPACK && !src) return;                            // (adjoint only: a source whose gradient the caller forms elsewhere)
    if (!UNPACK) {
        if (a.channel_first[s]) {
            for (int cc = wv; cc < cw; cc += 4) tile[lane][cc] = n0 + lane < N ? src[((size_t)b * W + c0 + cc) * N + n0 + lane] : 0.f;
        } else {
            for (int p = wv; p < 64; p += 4) tile[p][lane] = (lane < cw && n0 + p < N) ? src[((size_t)b * N + n0 + p) * W + c0 + lane] : 0.f;
        }
        __syncthreads();
        if (lane < cw)
            for (int p = wv; p < 64; p += 4)
                if (n0 + p < N) st_st1<DT>(a.dst, ((size_t)b * N + n0 + p) * a.dst_ld + dcol + lane, tile[p][lane]);
    } else {
        for (int p = wv; p < 64; p += 4)
            tile[p][lane] = (lane < cw && n0 + p < N) ? st_ld1<DT>(a.dst, ((size_t)b * N + n0 + p) * a.dst_ld + dcol + lane) : 0.f;
        __syncthreads();
        if (a.channel_first[s]) {
            if (n0 + lane < N)
                for (int cc = wv; cc < cw; cc += 4) src[((size_t)b * W + c0 + cc) * N + n0 + lane] = tile[lane][cc];
        } else if (lane < cw) {
            for (int p = wv; p < 64; p += 4)
                if (n0 + p < N) src[((size_t)b * N + n0 + p) * W + c0 + lane] = tile[p][lane];
        }
    }
}

int pack_launch(const gfs3d_pack_args *a, bool unpack, hipStream_t st, const char *who)
{
    GFS3D_REQUIRE(a, "%s: NULL args", who);
    GFS3D_REQUIRE(a->B >= 1 && a->N >= 1 && (long long)a->B * a->N < (1ll << 31), "%s: bad shape B=%d N=%d", who, a->B, a->N);
    GFS3D_REQUIRE(a->n_src >= 1 && a->n_src <= GFS3D_PACK_MAX_SRC, "%s: n_src=%d unsupported (1..%d)", who, a->n_src, GFS3D_PACK_MAX_SRC);
    GFS3D_REQUIRE(a->dst, "%s: NULL dst", who);
    PackPlan pl{};
    pl.n_src = a->n_src;
    int off = 0, tiles = 0;
    for (int s = 0; s < a->n_src; ++s) {
        GFS3D_REQUIRE((a->src[s] || unpack) && a->width[s] >= 1, "%s: NULL or empty source %d", who, s);
        pl.off[s] = off;
        pl.tiles[s] = (a->width[s] + 63) / 64;
        off += a->width[s];
        tiles += pl.tiles[s];
    }
    GFS3D_REQUIRE(a->dst_ld >= off, "%s: dst_ld=%d < total width %d", who, a->dst_ld, off);
    pl.used = off;
    pl.tiles[a->n_src] = (a->dst_ld - off + 63) / 64;
    pl.total_tiles = tiles + (unpack ? 0 : pl.tiles[a->n_src]);
    GFS3D_REQUIRE(pl.total_tiles <= 65535 && a->B <= 65535, "%s: too many tiles", who);
    const dim3 grid((a->N + 63) / 64, pl.total_tiles, a->B);
    GFS3D_REQUIRE(a->dst_type == 0 || a->dst_type == GFS3D_COMPUTE_BF16 || (a->dst_type == GFS3D_COMPUTE_F16 && !unpack),
                  "%s: dst_type=%d unsupported (0 fp32, GFS3D_COMPUTE_BF16, GFS3D_COMPUTE_F16 for packing only)", who, a->dst_type);
    if (unpack) {
        if (a->dst_type == GFS3D_COMPUTE_BF16) hipLaunchKernelGGL((pw_pack_kernel<true, ST_BF16>), grid, dim3(256), 0, st, *a, pl);
        else hipLaunchKernelGGL((pw_pack_kernel<true, ST_F32>), grid, dim3(256), 0, st, *a, pl);
    } else {
        if (a->dst_type == GFS3D_COMPUTE_BF16) hipLaunchKernelGGL((pw_pack_kernel<false, ST_BF16>), grid, dim3(256), 0, st, *a, pl);
        else if (a->dst_type == GFS3D_COMPUTE_F16) hipLaunchKernelGGL((pw_pack_kernel<false, ST_F16>), grid, dim3(256), 0, st, *a, pl);
        else hipLaunchKernelGGL((pw_pack_kernel<false, ST_F32>), grid, dim3(256), 0, st, *a, pl);
    }
    GFS3D_LAUNCH_CHECK("pw_pack_kernel");
    return GFS3D_OK;
}

struct LinPlan { size_t off_w3, off_dy, off_part, total; };
LinPlan lin_plan(int B, int N, int K, int Nc)
{
    LinPlan p{};
    size_t o = 0;
    p.off_w3 = o; o = align_up(o + 6 * (size_t)K * Nc, 256);
    p.off_dy = o; o = align_up(o + sizeof(float) * (size_t)B * N * Nc, 256);
    p.off_part = o; o = align_up(o + wg_part_bytes(B * N, Nc, K, true), 256);
    p.total = o;
    return p;
}

int lin_check(const float *x, int B, int N, int K, const float *w, int Nc, const void *ws, size_t wsb, int compute, const char *who)
{
    GFS3D_REQUIRE(x && w && ws, "%s: NULL pointer", who);
    GFS3D_REQUIRE(B >= 1 && N >= 1 && (long long)B * N < (1ll << 31), "%s: bad shape B=%d N=%d", who, B, N);
    GFS3D_REQUIRE(K >= 64 && K % 64 == 0 && K <= 1024, "%s: K=%d unsupported (multiple of 64, <= 1024)", who, K);
    GFS3D_REQUIRE(Nc >= 64 && Nc % 64 == 0 && Nc <= 256, "%s: Nc=%d unsupported (64, 128, 192, 256)", who, Nc);
    GFS3D_REQUIRE(compute == GFS3D_COMPUTE_F32 || compute == GFS3D_COMPUTE_BF16 || compute == GFS3D_COMPUTE_F16,
                  "%s: compute must be one of GFS3D_COMPUTE_*", who);
    if (wsb < lin_plan(B, N, K, Nc).total) {
        set_error("%s: workspace too small (%zu < %zu)", who, wsb, lin_plan(B, N, K, Nc).total);
        return GFS3D_ERR_WORKSPACE;
    }
    return GFS3D_OK;
}
}  // namespace
}  // namespace gfs3d

extern "C" int gfs3d_pack_rows_f32(const gfs3d_pack_args *a, void *stream)
{
    return pack_launch(a, false, static_cast<hipStream_t>(stream), "gfs3d_pack_rows_f32");
}

extern "C" int gfs3d_unpack_rows_f32(const gfs3d_pack_args *a, void *stream)
{
    return pack_launch(a, true, static_cast<hipStream_t>(stream), "gfs3d_unpack_rows_f32");
}

extern "C" size_t gfs3d_linear_workspace_bytes(int B, int N, int K, int Nc)
{
    if (B < 1 || N < 1 || K < 64 || K % 64 || K > 1024 || Nc < 64 || Nc % 64 || Nc > 256) return 0;
    return lin_plan(B, N, K, Nc).total;
}

extern "C" int gfs3d_linear_forward_f32(const float *x_pm, int B, int N, int K, const float *w, int Nc, float *const *y_cf, int compute,
                                        void *workspace, size_t workspace_bytes, void *stream)
{
    const char *who = "gfs3d_linear_forward_f32";
    if (int rc = lin_check(x_pm, B, N, K, w, Nc, workspace, workspace_bytes, compute, who)) return rc;
    GFS3D_REQUIRE(y_cf, "%s: NULL y_cf", who);
    for (int s = 0; s < Nc / 64; ++s) GFS3D_REQUIRE(y_cf[s], "%s: NULL y_cf[%d]", who, s);
    hipStream_t st = static_cast<hipStream_t>(stream);
    ProfScope prof(6, st);
    const LinPlan p = lin_plan(B, N, K, Nc);
    char *ws = static_cast<char *>(workspace);
    RowGemm rg{};
    rg.a[0] = x_pm; rg.a_segw = K; rg.lda = K; rg.K = K;
    rg.w = w; rg.w3 = ws + p.off_w3; rg.prec = prec_of(compute, false);
    if (int rc = split_weights(w, Nc, K, K, 1, ws + p.off_w3, st, rg.prec)) return rc;
    rg.M = B * N; rg.Nc = Nc; rg.slope = 1.f;
    for (int s = 0; s < Nc / 64; ++s) rg.out_cfs[s] = y_cf[s];
    rg.cf_segw = 64; rg.npts = N;
    return rowgemm<PRO_NONE, EPI_PLAIN, true>(rg, st, who);
}

extern "C" int gfs3d_linear_backward_f32(const float *x_pm, int B, int N, int K, const float *w, int Nc, const float *const *dy_cf,
                                         float *dx_pm, float *dw, int compute, void *workspace, size_t workspace_bytes, void *stream)
{
    const char *who = "gfs3d_linear_backward_f32";
    if (int rc = lin_check(x_pm, B, N, K, w, Nc, workspace, workspace_bytes, compute, who)) return rc;
    GFS3D_REQUIRE(dy_cf && dx_pm && dw, "%s: NULL pointer", who);
    hipStream_t st = static_cast<hipStream_t>(stream);
    ProfScope prof(7, st);
    const LinPlan p = lin_plan(B, N, K, Nc);
    char *ws = static_cast<char *>(workspace);
    float *dy = reinterpret_cast<float *>(ws + p.off_dy);
    gfs3d_pack_args pa{};
    pa.B = B; pa.N = N; pa.n_src = Nc / 64; pa.dst = dy; pa.dst_ld = Nc;
    for (int s = 0; s < Nc / 64; ++s) {
        GFS3D_REQUIRE(dy_cf[s], "%s: NULL dy_cf[%d]", who, s);
        pa.src[s] = const_cast<float *>(dy_cf[s]); pa.width[s] = 64; pa.channel_first[s] = 1;
    }
    if (int rc = pack_launch(&pa, false, st, who)) return rc;
    const int prec = prec_of(compute, true);
    // dW = dY^T X
    WGrad wg{};
    wg.part = reinterpret_cast<float *>(ws + p.off_part);
    wg.u = dy; wg.ldu = Nc;
    wg.v[0] = x_pm; wg.v_segw = K; wg.ldv = K;
    wg.slope = 1.f; wg.M = B * N; wg.Co = Nc; wg.Ci = K; wg.dw = dw; wg.prec = prec;
    if (int rc = wgrad<PRO_NONE>(wg, true, st, who)) return rc;
    // dX = dY W
    if (int rc = split_weights(w, K, Nc, 1, K, ws + p.off_w3, st, prec)) return rc;       // pieces of W^T (K, Nc)
    RowGemm rg{};
    rg.a[0] = dy; rg.a_segw = Nc; rg.lda = Nc; rg.K = Nc;
    rg.w3 = ws + p.off_w3; rg.prec = prec;
    rg.M = B * N; rg.Nc = K; rg.slope = 1.f;
    rg.out[0] = dx_pm; rg.o_segw = K;
    return rowgemm<PRO_NONE, EPI_PLAIN, false>(rg, st, who);
}
