// K5: fused self-attention over the points of a block  (SelfAttention.forward, reference model/attention.py:39-48)
//
//     attn = dropout(softmax_k(q^T k / sqrt(D)))        (B, N, N)   never materialised here
//     y    = attn v^T                                   -> (B, D, N) channel-first, D = 64
//
// The reference materialises the (B, N, N) score matrix three times (512 MiB each at B = 32, N = 2048).  Here a
// wave owns 32 queries and streams over key tiles of 32 (flash-attention recurrence, fp32 results):
//   S^T tile = K_tile Q^T with the QUERY on the lane (C/D layout: column = query, registers = 16 keys per lane
//   half) - the same orientation as the kNN kernel - so the running max / sum of a query, its dropout decisions
//   and the rescaling of its output accumulators are in-lane work plus one cross-half shuffle per tile;
//   O^T += V^T P consumes the probabilities straight from the accumulator registers (registers 8 s .. 8 s + 7 of
//   both lane halves = the K index of MFMA step s), V^T from the staged tile through the transposed LDS read.
// Arithmetic: bf16x6 on v_mfma_f32_32x32x16_bf16 (exact 3-way bf16 split of both operands, six piece products per
// fp32 product - common.h): 48 MFMAs of 32 cycles per tile in the forward where the fp32 form (v_mfma_f32_32x32x2_f32)
// took 64 of 64 cycles.
// Dropout: keep(b, q, k) is a counter-based hash of (seed, b, q N + k), so the backward regenerates it; tests can
// pass an explicit (B, N, N) byte mask instead.  The backward recomputes S and P from q, k and the saved
// log-sum-exp (one kernel per side: dQ with the query on the lane, dK / dV with the key on the lane), no atomics.
#include "common.h"

namespace gfs3d {
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int AT_D = 64;

__device__ __forceinline__ int at_drow(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }
__device__ __forceinline__ unsigned at_hash(unsigned x)
{
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
__device__ __forceinline__ unsigned at_seed(unsigned long long seed, int b)
{
    return at_hash((unsigned)seed + 0x9e3779b9u * (unsigned)(b + 1)) ^ (unsigned)(seed >> 32);
}
// keep decision of attention weight (q, k) of batch item b
template <bool HAS_MASK>
__device__ __forceinline__ bool at_keep(const uint8_t *mask, unsigned sb, unsigned thresh, int b, int q, int k, int N)
{
    if (HAS_MASK) return mask[((size_t)b * N + q) * N + k] != 0;
    return at_hash((unsigned)(q * N + k) ^ sb) >= thresh;
}

struct AttnArgs {
    const float *q, *k, *v;       // (B, 64, N)
    const float *out, *lse;       // forward results (backward inputs)
    const float *dout, *Dq;       // (B, 64, N); (B, N)
    int B, N;
    float scale, inv_keep;
    unsigned thresh;              // drop if hash < thresh  (0: no dropout)
    unsigned long long seed;
    const uint8_t *mask;          // optional (B, N, N) keep mask
};

// ---------------------------------------------------------------------------------------------------------
// Operand images.  Every GEMM of the three kernels runs on v_mfma_f32_32x32x16_bf16 with both operands split exactly into
// three bf16 pieces (six piece products = fp32 accuracy, common.h); a staged (32 positions x 64 channels) tile is split ONCE
// per block while it is written to LDS as piece planes [plane][position][64 x bf16 | 16 B pad], and serves
//   * as an A operand with rows = positions, k = channels   (ds_read_b128: lane (row j, half h) reads channels 16 kb + 8 h ..)
//   * as an A operand with rows = channels, k = positions   (ds_read_b64_tr_b16, the hardware transpose read: no second image)
// The B operand is always a register tile: the per-wave constant rows (q, dout / k, v: split once per wave) or a result of
// the previous MFMA in its C/D layout (probabilities, dS) - k-step s takes registers 8 s .. 8 s + 7 of both lane halves, i.e.
// positions at_drow(8 s + e, h), and the transposed read delivers the A operand in exactly that k order.
// ---------------------------------------------------------------------------------------------------------
constexpr int AT_PROW = 2 * AT_D + 16;             // bytes per position row of one piece plane
constexpr int AT_PLANE = 32 * AT_PROW;
typedef short at_s16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) at_s16x4 at_lds_s16x4;

// thread (pos = t & 31, cg = t >> 5) stages channels 8 cg .. 8 cg + 7 of position p0 + pos (clamped: a tile past N repeats
// the last position, whose scores are masked)
__device__ __forceinline__ void at_load8(const float *base, int N, int p0, int t, float (&r)[8])
{
    const int pos = min(p0 + (t & 31), N - 1), c0 = 8 * (t >> 5);
#pragma unroll
    for (int e = 0; e < 8; ++e) r[e] = base[(size_t)(c0 + e) * N + pos];
}
template <int PREC>
__device__ __forceinline__ void at_put(unsigned char *img, int t, const float (&r)[8], float s)
{
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = r[e] * s;
    uint4 pc[Mx<PREC>::NP];
    mm_pieces8<PREC>(v, pc);
#pragma unroll
    for (int p = 0; p < Mx<PREC>::NP; ++p) *reinterpret_cast<uint4 *>(img + p * AT_PLANE + (t & 31) * AT_PROW + 16 * (t >> 5)) = pc[p];
}
// Backward of a HALF forward (compute = f16): the gradient GEMMs run on bf16 operands (gradients underflow in half), but S
// must be recomputed from the operands the FORWARD used - exp(S_bf16 - lse_f16) is not a normalised row, and the rows of dS
// would no longer sum to zero (a systematic gradient error).  The images of such a pass carry one more plane: the
// f16 rounding of the same tile, read only by the S product.  SP = precision of the S operands (= PREC in every other mode).
template <int PREC, int SP> struct AtImg {
    static constexpr bool DUAL = SP != PREC;
    static constexpr int NPI = Mx<PREC>::NP + (DUAL ? 1 : 0);     // planes per image
    static constexpr int SOFF = DUAL ? Mx<PREC>::NP * AT_PLANE : 0;  // byte offset of the S operand's plane(s)
};
template <int PREC, int SP>
__device__ __forceinline__ void at_put2(unsigned char *img, int t, const float (&r)[8], float s)
{
    at_put<PREC>(img, t, r, s);
    if constexpr (AtImg<PREC, SP>::DUAL) at_put<SP>(img + AtImg<PREC, SP>::SOFF, t, r, s);
}
// rows = positions: lane (row j, half h), K-block kb -> channels 16 kb + 8 h + e
template <int PREC>
__device__ __forceinline__ void at_rowfrag(const unsigned char *img, int j, int h, int kb, uint4 (&f)[Mx<PREC>::NP])
{
#pragma unroll
    for (int p = 0; p < Mx<PREC>::NP; ++p) f[p] = *reinterpret_cast<const uint4 *>(img + p * AT_PLANE + j * AT_PROW + 32 * kb + 16 * h);
}
// rows = channels 32 dt + (lane & 31), k-step s: positions at_drow(8 s + e, h) = 16 s + 8 (e >> 2) + 4 h + (e & 3).  Two
// transposed reads of 4 position rows each: in its 16-lane group, lane 4 q + p supplies the address of row q, channels
// 16 (group & 1) + 4 p .. + 3, and receives the 4 rows of its own channel.  EXEC must be all ones here.
template <int PREC>
__device__ __forceinline__ void at_trfrag(const unsigned char *img, int lane, int dt, int s, uint4 (&f)[Mx<PREC>::NP])
{
    const int h = lane >> 5, g1 = (lane >> 4) & 1, q = (lane >> 2) & 3, p4 = lane & 3;
    const unsigned char *base = img + (16 * s + 4 * h + q) * AT_PROW + 2 * (32 * dt + 16 * g1 + 4 * p4);
#pragma unroll
    for (int p = 0; p < Mx<PREC>::NP; ++p) {
        const at_s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((at_lds_s16x4 *)(base + p * AT_PLANE));
        const at_s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((at_lds_s16x4 *)(base + p * AT_PLANE + 8 * AT_PROW));
        const uint2 a = __builtin_bit_cast(uint2, lo), b = __builtin_bit_cast(uint2, hi);
        f[p] = make_uint4(a.x, a.y, b.x, b.y);
    }
}
// the per-wave constant operand: lane (column j, half h) holds channels 16 kb + 8 h + e of its position, times s
template <int PREC>
__device__ __forceinline__ void at_colfrags(const float *base, int N, int pos, int h, float s, uint4 (&f)[4][Mx<PREC>::NP])
{
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = base[(size_t)(16 * kb + 8 * h + e) * N + pos] * s;
        mm_pieces8<PREC>(v, f[kb]);
    }
}
template <int PREC>
__device__ __forceinline__ void at_regfrag(const float (&v)[16], int s, uint4 (&f)[Mx<PREC>::NP])
{
    float v8[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v8[e] = v[8 * s + e];
    mm_pieces8<PREC>(v8, f);
}

// ------------------------------------------------------------------ forward
// DROP: 0 = no dropout (eval), 1 = counter-based hash, 2 = explicit keep mask - a template parameter: as a run-time flag the
// compiler kept the hash arithmetic (~180 of 530 VALU per tile) in the loop of every mode
template <int DROP, int PREC>
__global__ void __launch_bounds__(256, 2) attn_fwd_kernel(const AttnArgs g, float *__restrict__ out, float *__restrict__ lse)
{
    constexpr int NP = Mx<PREC>::NP, IMG = NP * AT_PLANE;
    extern __shared__ __attribute__((aligned(16))) unsigned char at_smem[];
    unsigned char *Kimg = at_smem, *Vimg = at_smem + 2 * IMG;          // [2 buffers] each
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6, j = lane & 31, h = lane >> 5;
    const int b = blockIdx.y, N = g.N, q0 = blockIdx.x * 128 + wv * 32;
    const float *qb = g.q + (size_t)b * AT_D * N, *kb_ = g.k + (size_t)b * AT_D * N, *vb = g.v + (size_t)b * AT_D * N;
    const int qi = min(q0 + j, N - 1);
    const unsigned sb = at_seed(g.seed, b);
    constexpr bool drop = DROP != 0, HAS_MASK = DROP == 2;
    uint4 qp[4][NP];
    at_colfrags<PREC>(qb, N, qi, h, g.scale, qp);
    float m = -__builtin_inff(), l = 0.f;
    f32x16 O0, O1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { O0[r] = 0.f; O1[r] = 0.f; }

    float rk[8], rv[8];
    at_load8(kb_, N, 0, t, rk);
    at_load8(vb, N, 0, t, rv);
    at_put<PREC>(Kimg, t, rk, 1.f);
    at_put<PREC>(Vimg, t, rv, 1.f);
    __syncthreads();
    const int nt = (N + 31) / 32;
    for (int tt = 0; tt < nt; ++tt) {
        const int cur = tt & 1, j0 = tt * 32;
        const unsigned char *Kc = Kimg + cur * IMG, *Vc = Vimg + cur * IMG;
        if (tt + 1 < nt) { at_load8(kb_, N, j0 + 32, t, rk); at_load8(vb, N, j0 + 32, t, rv); }
        f32x16 acc;                                        // S^T tile: rows = keys, column = the lane's query
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
            uint4 ak[NP];
            at_rowfrag<PREC>(Kc, j, h, kb, ak);
            mm_mac<PREC>(ak, qp[kb], acc);
        }
        float tmax = -__builtin_inff();
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            if (j0 + at_drow(r, h) >= N) acc[r] = -__builtin_inff();
            tmax = fmaxf(tmax, acc[r]);
        }
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
        const float m_new = fmaxf(m, tmax);              // finite: every tile holds at least one valid key
        const float sc = __expf(m - m_new);
        float psum = 0.f, pd[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float p = __expf(acc[r] - m_new);
            psum += p;
            pd[r] = p;
        }
        psum += __shfl_xor(psum, 32, 64);
        l = fmaf(l, sc, psum);
        m = m_new;
#pragma unroll
        for (int r = 0; r < 16; ++r) { O0[r] *= sc; O1[r] *= sc; }
        if constexpr (drop) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = min(j0 + at_drow(r, h), N - 1);
                pd[r] = at_keep<HAS_MASK>(g.mask, sb, g.thresh, b, qi, key, N) ? pd[r] * g.inv_keep : 0.f;
            }
        }
        // O^T += V^T P: rows = output channels (two tiles of 32), k = keys, B = the probabilities as they sit in the registers
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            uint4 pp[NP], av[NP];
            at_regfrag<PREC>(pd, s, pp);
            at_trfrag<PREC>(Vc, lane, 0, s, av);
            mm_mac<PREC>(av, pp, O0);
            at_trfrag<PREC>(Vc, lane, 1, s, av);
            mm_mac<PREC>(av, pp, O1);
        }
        if (tt + 1 < nt) { at_put<PREC>(Kimg + (cur ^ 1) * IMG, t, rk, 1.f); at_put<PREC>(Vimg + (cur ^ 1) * IMG, t, rv, 1.f); }
        __syncthreads();
    }
    if (q0 + j < N) {
        const float inv_l = 1.f / l;
        float *ob = out + (size_t)b * AT_D * N + q0 + j;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            ob[(size_t)at_drow(r, h) * N] = O0[r] * inv_l;
            ob[(size_t)(32 + at_drow(r, h)) * N] = O1[r] * inv_l;
        }
        if (h == 0) lse[(size_t)b * N + q0 + j] = m + __logf(l);
    }
}

// D[b][q] = sum_d dout[b][d][q] out[b][d][q]
__global__ void __launch_bounds__(256) attn_bwd_D_kernel(const float *__restrict__ dout, const float *__restrict__ out, int N,
                                                          float *__restrict__ Dq)
{
    const int q = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;
    if (q >= N) return;
    const float *a = dout + (size_t)b * AT_D * N + q, *o = out + (size_t)b * AT_D * N + q;
    float s = 0.f;
#pragma unroll 8
    for (int d = 0; d < AT_D; ++d) s = fmaf(a[(size_t)d * N], o[(size_t)d * N], s);
    Dq[(size_t)b * N + q] = s;
}

// ------------------------------------------------------------------ backward, query side: dQ
template <int DROP, int PREC, int SP>
__global__ void __launch_bounds__(256, 2) attn_bwd_dq_kernel(const AttnArgs g, float *__restrict__ dq)
{
    constexpr int NP = Mx<PREC>::NP, NPS = Mx<SP>::NP, IMG = AtImg<PREC, SP>::NPI * AT_PLANE, SOFF = AtImg<PREC, SP>::SOFF;
    extern __shared__ __attribute__((aligned(16))) unsigned char at_smem[];
    unsigned char *Kimg = at_smem, *Vimg = at_smem + 2 * IMG;
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6, j = lane & 31, h = lane >> 5;
    const int b = blockIdx.y, N = g.N, q0 = blockIdx.x * 128 + wv * 32;
    const size_t boff = (size_t)b * AT_D * N;
    const float *kb_ = g.k + boff, *vb = g.v + boff;
    const int qi = min(q0 + j, N - 1);
    const unsigned sb = at_seed(g.seed, b);
    constexpr bool drop = DROP != 0, HAS_MASK = DROP == 2;
    uint4 qp[4][NPS], gp[4][NP];
    at_colfrags<SP>(g.q + boff, N, qi, h, g.scale, qp);
    at_colfrags<PREC>(g.dout + boff, N, qi, h, 1.f, gp);
    const float Lq = g.lse[(size_t)b * N + qi], Dv = g.Dq[(size_t)b * N + qi];
    f32x16 G0, G1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { G0[r] = 0.f; G1[r] = 0.f; }

    float rk[8], rv[8];
    at_load8(kb_, N, 0, t, rk);
    at_load8(vb, N, 0, t, rv);
    at_put2<PREC, SP>(Kimg, t, rk, 1.f);
    at_put<PREC>(Vimg, t, rv, 1.f);
    __syncthreads();
    const int nt = (N + 31) / 32;
    for (int tt = 0; tt < nt; ++tt) {
        const int cur = tt & 1, j0 = tt * 32;
        const unsigned char *Kc = Kimg + cur * IMG, *Vc = Vimg + cur * IMG;
        if (tt + 1 < nt) { at_load8(kb_, N, j0 + 32, t, rk); at_load8(vb, N, j0 + 32, t, rv); }
        f32x16 acc, dp;                                    // S^T and dP^T tiles: rows = keys, column = the lane's query
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
            uint4 ak[NPS], av[NP];
            at_rowfrag<SP>(Kc + SOFF, j, h, kb, ak);
            at_rowfrag<PREC>(Vc, j, h, kb, av);
            mm_mac<SP>(ak, qp[kb], acc);
            mm_mac<PREC>(av, gp[kb], dp);
        }
        float ds[16];                                      // element-wise part, branch-free (one uniform branch for the dropout)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float pe = __expf(acc[r] - Lq);            // (a key past N is a clamped copy of the last one: finite)
            ds[r] = j0 + at_drow(r, h) < N ? pe : 0.f;
        }
        if constexpr (drop) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const bool keep = at_keep<HAS_MASK>(g.mask, sb, g.thresh, b, qi, min(j0 + at_drow(r, h), N - 1), N);
                ds[r] *= (keep ? dp[r] * g.inv_keep : 0.f) - Dv;
            }
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) ds[r] *= dp[r] - Dv;
        }
        // dQ^T += K^T dS: rows = channels, k = keys
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            uint4 dsp[NP], ak[NP];
            at_regfrag<PREC>(ds, s, dsp);
            at_trfrag<PREC>(Kc, lane, 0, s, ak);
            mm_mac<PREC>(ak, dsp, G0);
            at_trfrag<PREC>(Kc, lane, 1, s, ak);
            mm_mac<PREC>(ak, dsp, G1);
        }
        if (tt + 1 < nt) { at_put2<PREC, SP>(Kimg + (cur ^ 1) * IMG, t, rk, 1.f); at_put<PREC>(Vimg + (cur ^ 1) * IMG, t, rv, 1.f); }
        __syncthreads();
    }
    if (q0 + j < N) {
        float *ob = dq + boff + q0 + j;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            ob[(size_t)at_drow(r, h) * N] = G0[r] * g.scale;
            ob[(size_t)(32 + at_drow(r, h)) * N] = G1[r] * g.scale;
        }
    }
}

// ------------------------------------------------------------------ backward, key side: dK, dV
template <int DROP, int PREC, int SP>
__global__ void __launch_bounds__(256, 2) attn_bwd_dkv_kernel(const AttnArgs g, float *__restrict__ dk, float *__restrict__ dv)
{
    constexpr int NP = Mx<PREC>::NP, NPS = Mx<SP>::NP, IMG = AtImg<PREC, SP>::NPI * AT_PLANE, SOFF = AtImg<PREC, SP>::SOFF;
    extern __shared__ __attribute__((aligned(16))) unsigned char at_smem[];
    unsigned char *Qimg = at_smem, *Gimg = at_smem + 2 * IMG;                  // queries (scaled) and dout, [2 buffers] each
    float (*Ls)[32] = reinterpret_cast<float (*)[32]>(at_smem + 4 * IMG);       // [2][32]
    float (*Ds)[32] = Ls + 2;
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6, j = lane & 31, h = lane >> 5;
    const int b = blockIdx.y, N = g.N, k0 = blockIdx.x * 128 + wv * 32;
    const size_t boff = (size_t)b * AT_D * N;
    const float *qb = g.q + boff, *gb = g.dout + boff;
    const int ki = min(k0 + j, N - 1);
    const bool kvalid = k0 + j < N;
    const unsigned sb = at_seed(g.seed, b);
    constexpr bool drop = DROP != 0, HAS_MASK = DROP == 2;
    uint4 kp[4][NPS], vp[4][NP];
    at_colfrags<SP>(g.k + boff, N, ki, h, 1.f, kp);
    at_colfrags<PREC>(g.v + boff, N, ki, h, 1.f, vp);
    f32x16 K0, K1, V0, V1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { K0[r] = 0.f; K1[r] = 0.f; V0[r] = 0.f; V1[r] = 0.f; }

    float rq[8], rg[8];
    auto put = [&](int buf, int p0) {
        at_put2<PREC, SP>(Qimg + buf * IMG, t, rq, g.scale);
        at_put<PREC>(Gimg + buf * IMG, t, rg, 1.f);
        if (t < 32) { const int qq = min(p0 + t, N - 1); Ls[buf][t] = g.lse[(size_t)b * N + qq]; Ds[buf][t] = g.Dq[(size_t)b * N + qq]; }
    };
    at_load8(qb, N, 0, t, rq);
    at_load8(gb, N, 0, t, rg);
    put(0, 0);
    __syncthreads();
    const int nt = (N + 31) / 32;
    for (int tt = 0; tt < nt; ++tt) {
        const int cur = tt & 1, p0 = tt * 32;
        const unsigned char *Qc = Qimg + cur * IMG, *Gc = Gimg + cur * IMG;
        if (tt + 1 < nt) { at_load8(qb, N, p0 + 32, t, rq); at_load8(gb, N, p0 + 32, t, rg); }
        f32x16 acc, dp;                                    // S and dP tiles: rows = queries, column = the lane's key
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
            uint4 aq[NPS], ag[NP];
            at_rowfrag<SP>(Qc + SOFF, j, h, kb, aq);
            at_rowfrag<PREC>(Gc, j, h, kb, ag);
            mm_mac<SP>(aq, kp[kb], acc);
            mm_mac<PREC>(ag, vp[kb], dp);
        }
        // element-wise part, branch-free (ONE uniform branch for the dropout decisions: as a per-element `if` it became 16
        // basic blocks with a wait each)
        float pd[16], ds[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const bool ok = p0 + at_drow(r, h) < N && kvalid;
            const float pe = __expf(acc[r] - Ls[cur][at_drow(r, h)]);     // (a query past N is a clamped copy of the last one: finite)
            pd[r] = ok ? pe : 0.f;
        }
        if constexpr (drop) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const bool keep = at_keep<HAS_MASK>(g.mask, sb, g.thresh, b, min(p0 + at_drow(r, h), N - 1), ki, N);
                ds[r] = pd[r] * ((keep ? dp[r] * g.inv_keep : 0.f) - Ds[cur][at_drow(r, h)]);
                pd[r] = keep ? pd[r] * g.inv_keep : 0.f;
            }
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) ds[r] = pd[r] * (dp[r] - Ds[cur][at_drow(r, h)]);
        }
        // dV^T += dout^T P and dK^T += Q^T dS: rows = channels, k = queries
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            uint4 bp[NP], af[NP];
            at_regfrag<PREC>(pd, s, bp);
            at_trfrag<PREC>(Gc, lane, 0, s, af);
            mm_mac<PREC>(af, bp, V0);
            at_trfrag<PREC>(Gc, lane, 1, s, af);
            mm_mac<PREC>(af, bp, V1);
            at_regfrag<PREC>(ds, s, bp);
            at_trfrag<PREC>(Qc, lane, 0, s, af);
            mm_mac<PREC>(af, bp, K0);
            at_trfrag<PREC>(Qc, lane, 1, s, af);
            mm_mac<PREC>(af, bp, K1);
        }
        if (tt + 1 < nt) put(cur ^ 1, p0 + 32);
        __syncthreads();
    }
    if (kvalid) {
        float *ok = dk + boff + k0 + j, *ov = dv + boff + k0 + j;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            ok[(size_t)at_drow(r, h) * N] = K0[r];
            ok[(size_t)(32 + at_drow(r, h)) * N] = K1[r];
            ov[(size_t)at_drow(r, h) * N] = V0[r];
            ov[(size_t)(32 + at_drow(r, h)) * N] = V1[r];
        }
    }
}

__global__ void __launch_bounds__(256) attn_mask_kernel(int B, int N, unsigned thresh, unsigned long long seed, uint8_t *__restrict__ mask)
{
    const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x, per = (size_t)N * N;
    if (e >= (size_t)B * per) return;
    const int b = (int)(e / per);
    const unsigned qk = (unsigned)(e - (size_t)b * per);
    mask[e] = at_hash(qk ^ at_seed(seed, b)) >= thresh ? 1 : 0;
}


int at_check(const char *who, const float *q, const float *k, const float *v, int B, int D, int N, float p_drop)
{
    GFS3D_REQUIRE(q && k && v, "%s: NULL input", who);
    GFS3D_REQUIRE(D == AT_D, "%s: head dimension %d unsupported (64)", who, D);
    GFS3D_REQUIRE(B >= 1 && B <= 65535 && N >= 1, "%s: bad shape B=%d N=%d", who, B, N);
    GFS3D_REQUIRE((long long)N * N < (1ll << 32), "%s: N=%d too large for the dropout counter", who, N);
    GFS3D_REQUIRE(p_drop >= 0.f && p_drop < 1.f, "%s: dropout probability %g outside [0, 1)", who, (double)p_drop);
    return GFS3D_OK;
}

AttnArgs at_args(const float *q, const float *k, const float *v, int B, int N, float scale, float p_drop, unsigned long long seed,
                 const uint8_t *mask)
{
    AttnArgs a{};
    a.q = q; a.k = k; a.v = v; a.B = B; a.N = N; a.scale = scale;
    a.inv_keep = (p_drop > 0.f) ? 1.f / (1.f - p_drop) : 1.f;
    a.thresh = mask ? 0u : (unsigned)((double)p_drop * 4294967296.0);
    a.seed = seed;
    a.mask = mask;
    return a;
}

}  // namespace
}  // namespace gfs3d

using namespace gfs3d;

template <int PREC>
static int at_forward(const AttnArgs &a, bool has_mask, dim3 grid, float *out, float *lse, hipStream_t st)
{
    const size_t lds = 4 * Mx<PREC>::NP * AT_PLANE;
    if (has_mask) hipLaunchKernelGGL((attn_fwd_kernel<2, PREC>), grid, dim3(256), lds, st, a, out, lse);
    else if (a.thresh != 0u) hipLaunchKernelGGL((attn_fwd_kernel<1, PREC>), grid, dim3(256), lds, st, a, out, lse);
    else hipLaunchKernelGGL((attn_fwd_kernel<0, PREC>), grid, dim3(256), lds, st, a, out, lse);
    GFS3D_LAUNCH_CHECK("attn_fwd_kernel");
    return GFS3D_OK;
}

template <int PREC, int SP>
static int at_backward(const AttnArgs &a, bool has_mask, dim3 grid, float *dq, float *dk, float *dv, hipStream_t st)
{
    const size_t lds = 4 * AtImg<PREC, SP>::NPI * AT_PLANE;
    if (has_mask) {
        hipLaunchKernelGGL((attn_bwd_dq_kernel<2, PREC, SP>), grid, dim3(256), lds, st, a, dq);
        hipLaunchKernelGGL((attn_bwd_dkv_kernel<2, PREC, SP>), grid, dim3(256), lds + sizeof(float) * 4 * 32, st, a, dk, dv);
    } else if (a.thresh != 0u) {
        hipLaunchKernelGGL((attn_bwd_dq_kernel<1, PREC, SP>), grid, dim3(256), lds, st, a, dq);
        hipLaunchKernelGGL((attn_bwd_dkv_kernel<1, PREC, SP>), grid, dim3(256), lds + sizeof(float) * 4 * 32, st, a, dk, dv);
    } else {
        hipLaunchKernelGGL((attn_bwd_dq_kernel<0, PREC, SP>), grid, dim3(256), lds, st, a, dq);
        hipLaunchKernelGGL((attn_bwd_dkv_kernel<0, PREC, SP>), grid, dim3(256), lds + sizeof(float) * 4 * 32, st, a, dk, dv);
    }
    GFS3D_LAUNCH_CHECK("attn_bwd kernels");
    return GFS3D_OK;
}

extern "C" int gfs3d_attention_forward_mixed(const float *q, const float *k, const float *v, int B, int D, int N, float scale, float p_drop,
                                             unsigned long long seed, const uint8_t *keep_mask, float *out, float *lse, int compute,
                                             void *stream)
{
    const char *who = "gfs3d_attention_forward_f32";
    if (int rc = at_check(who, q, k, v, B, D, N, p_drop)) return rc;
    GFS3D_REQUIRE(out && lse, "%s: NULL output", who);
    GFS3D_REQUIRE(compute == GFS3D_COMPUTE_F32 || compute == GFS3D_COMPUTE_BF16 || compute == GFS3D_COMPUTE_F16,
                  "%s: compute must be one of GFS3D_COMPUTE_*", who);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const AttnArgs a = at_args(q, k, v, B, N, scale, p_drop, seed, keep_mask);
    dim3 grid((N + 127) / 128, B);
    ProfScope prof(8, st);
    if (compute == GFS3D_COMPUTE_BF16) return at_forward<PREC_BF16>(a, keep_mask != nullptr, grid, out, lse, st);
    if (compute == GFS3D_COMPUTE_F16) return at_forward<PREC_F16>(a, keep_mask != nullptr, grid, out, lse, st);
    return at_forward<PREC_X6>(a, keep_mask != nullptr, grid, out, lse, st);
}

extern "C" int gfs3d_attention_forward_f32(const float *q, const float *k, const float *v, int B, int D, int N, float scale, float p_drop,
                                           unsigned long long seed, const uint8_t *keep_mask, float *out, float *lse, void *stream)
{
    return gfs3d_attention_forward_mixed(q, k, v, B, D, N, scale, p_drop, seed, keep_mask, out, lse, GFS3D_COMPUTE_F32, stream);
}

extern "C" int gfs3d_attention_backward_mixed(const float *q, const float *k, const float *v, const float *out, const float *lse,
                                              const float *dout, int B, int D, int N, float scale, float p_drop, unsigned long long seed,
                                              const uint8_t *keep_mask, float *dq, float *dk, float *dv, float *workspace, int compute,
                                              void *stream)
{
    const char *who = "gfs3d_attention_backward_f32";
    if (int rc = at_check(who, q, k, v, B, D, N, p_drop)) return rc;
    GFS3D_REQUIRE(out && lse && dout && dq && dk && dv && workspace, "%s: NULL buffer", who);
    GFS3D_REQUIRE(compute == GFS3D_COMPUTE_F32 || compute == GFS3D_COMPUTE_BF16 || compute == GFS3D_COMPUTE_F16,
                  "%s: compute must be one of GFS3D_COMPUTE_*", who);
    hipStream_t st = static_cast<hipStream_t>(stream);
    AttnArgs a = at_args(q, k, v, B, N, scale, p_drop, seed, keep_mask);
    a.out = out; a.lse = lse; a.dout = dout; a.Dq = workspace;
    ProfScope prof(9, st);
    hipLaunchKernelGGL(attn_bwd_D_kernel, dim3((N + 255) / 256, B), dim3(256), 0, st, dout, out, N, workspace);
    GFS3D_LAUNCH_CHECK("attn_bwd_D_kernel");
    dim3 grid((N + 127) / 128, B);
    // gradients never run on half operands (prec_of); the backward of a half forward recomputes S from HALF operands all the same,
    // so that exp(S - lse) is the forward's normalised row (AtImg)
    if (compute == GFS3D_COMPUTE_F16) return at_backward<PREC_BF16, PREC_F16>(a, keep_mask != nullptr, grid, dq, dk, dv, st);
    if (compute == GFS3D_COMPUTE_BF16) return at_backward<PREC_BF16, PREC_BF16>(a, keep_mask != nullptr, grid, dq, dk, dv, st);
    return at_backward<PREC_X6, PREC_X6>(a, keep_mask != nullptr, grid, dq, dk, dv, st);
}

extern "C" int gfs3d_attention_backward_f32(const float *q, const float *k, const float *v, const float *out, const float *lse,
                                            const float *dout, int B, int D, int N, float scale, float p_drop, unsigned long long seed,
                                            const uint8_t *keep_mask, float *dq, float *dk, float *dv, float *workspace, void *stream)
{
    return gfs3d_attention_backward_mixed(q, k, v, out, lse, dout, B, D, N, scale, p_drop, seed, keep_mask, dq, dk, dv, workspace,
                                          GFS3D_COMPUTE_F32, stream);
}

extern "C" int gfs3d_attention_dropout_mask(int B, int N, float p_drop, unsigned long long seed, uint8_t *mask, void *stream)
{
    const char *who = "gfs3d_attention_dropout_mask";
    GFS3D_REQUIRE(mask && B >= 1 && N >= 1 && (long long)N * N < (1ll << 32), "%s: bad argument", who);
    GFS3D_REQUIRE(p_drop >= 0.f && p_drop < 1.f, "%s: dropout probability %g outside [0, 1)", who, (double)p_drop);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const size_t total = (size_t)B * N * N;
    hipLaunchKernelGGL(attn_mask_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, B, N, (unsigned)((double)p_drop * 4294967296.0), seed, mask);
    GFS3D_LAUNCH_CHECK("attn_mask_kernel");
    return GFS3D_OK;
}
