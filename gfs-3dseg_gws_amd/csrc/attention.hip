// K5: fused self-attention over the points of a block  (SelfAttention.forward, reference model/attention.py:39-48)
//
//     attn = dropout(softmax_k(q^T k / sqrt(D)))        (B, N, N)   never materialised here
//     y    = attn v^T                                   -> (B, D, N) channel-first, D = 64
//
// The reference materialises the (B, N, N) score matrix three times (512 MiB each at B = 32, N = 2048).  Here a
// wave owns 32 queries and streams over key tiles of 32 (flash-attention recurrence, fp32):
//   S^T tile = K_tile Q^T on v_mfma_f32_32x32x2_f32 with the QUERY on the lane (C/D layout: column = query,
//   registers = 16 keys per lane half) - the same orientation as the kNN kernel - so the running max / sum of a
//   query, its dropout decisions and the rescaling of its output accumulators are in-lane work plus one
//   cross-half shuffle per tile;   O^T += V^T P consumes the probabilities straight from the accumulator
//   registers (register r of both lane halves = K index of one MFMA step), V from an LDS tile.
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

// stage a (64 channels x 32 positions) tile of a channel-first tensor: thread (pos = t & 31, c0 = t >> 5) holds rows c0 + 8p
__device__ __forceinline__ void at_load(const float *base, int N, int p0, int t, float (&r)[8])
{
    const int pos = min(p0 + (t & 31), N - 1), c0 = t >> 5;
#pragma unroll
    for (int p = 0; p < 8; ++p) r[p] = base[(size_t)(c0 + 8 * p) * N + pos];
}
__device__ __forceinline__ void at_put_cf(float (*dst)[32], int t, const float (&r)[8], float s)      // [c][pos]
{
#pragma unroll
    for (int p = 0; p < 8; ++p) dst[(t >> 5) + 8 * p][t & 31] = r[p] * s;
}
__device__ __forceinline__ void at_put_pm(float (*dst)[AT_D + 1], int t, const float (&r)[8], float s)  // [pos][c]
{
#pragma unroll
    for (int p = 0; p < 8; ++p) dst[t & 31][(t >> 5) + 8 * p] = r[p] * s;
}

// ------------------------------------------------------------------ forward
template <bool HAS_MASK>
__global__ void __launch_bounds__(256, 2) attn_fwd_kernel(const AttnArgs g, float *__restrict__ out, float *__restrict__ lse)
{
    __shared__ float Ks[2][AT_D][32];          // [c][key]
    __shared__ float Vs[2][32][AT_D + 1];      // [key][d]
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6, j = lane & 31, h = lane >> 5;
    const int b = blockIdx.y, N = g.N, q0 = blockIdx.x * 128 + wv * 32;
    const float *qb = g.q + (size_t)b * AT_D * N, *kb = g.k + (size_t)b * AT_D * N, *vb = g.v + (size_t)b * AT_D * N;
    const int qi = min(q0 + j, N - 1);
    const unsigned sb = at_seed(g.seed, b);
    const bool drop = HAS_MASK || g.thresh != 0u;
    float qreg[32];
#pragma unroll
    for (int s = 0; s < 32; ++s) qreg[s] = qb[(size_t)(2 * s + h) * N + qi] * g.scale;
    float m = -__builtin_inff(), l = 0.f;
    f32x16 O0, O1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { O0[r] = 0.f; O1[r] = 0.f; }

    float rk[8], rv[8];
    at_load(kb, N, 0, t, rk);
    at_load(vb, N, 0, t, rv);
    at_put_cf(Ks[0], t, rk, 1.f);
    at_put_pm(Vs[0], t, rv, 1.f);
    __syncthreads();
    const int nt = (N + 31) / 32;
    for (int tt = 0; tt < nt; ++tt) {
        const int cur = tt & 1, j0 = tt * 32;
        if (tt + 1 < nt) { at_load(kb, N, j0 + 32, t, rk); at_load(vb, N, j0 + 32, t, rv); }
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int s = 0; s < 32; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(Ks[cur][2 * s + h][j], qreg[s], acc, 0, 0, 0);
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
        if (drop) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = min(j0 + at_drow(r, h), N - 1);
                pd[r] = at_keep<HAS_MASK>(g.mask, sb, g.thresh, b, qi, key, N) ? pd[r] * g.inv_keep : 0.f;
            }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = at_drow(r, h);
            O0 = __builtin_amdgcn_mfma_f32_32x32x2f32(Vs[cur][row][j], pd[r], O0, 0, 0, 0);
            O1 = __builtin_amdgcn_mfma_f32_32x32x2f32(Vs[cur][row][32 + j], pd[r], O1, 0, 0, 0);
        }
        if (tt + 1 < nt) { at_put_cf(Ks[cur ^ 1], t, rk, 1.f); at_put_pm(Vs[cur ^ 1], t, rv, 1.f); }
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
template <bool HAS_MASK>
__global__ void __launch_bounds__(256, 2) attn_bwd_dq_kernel(const AttnArgs g, float *__restrict__ dq)
{
    extern __shared__ float at_smem[];
    float (*Kc)[AT_D][32] = reinterpret_cast<float (*)[AT_D][32]>(at_smem);                              // [2][c][key]
    float (*Vc)[AT_D][32] = reinterpret_cast<float (*)[AT_D][32]>(at_smem + 2 * AT_D * 32);               // [2][c][key]
    float (*Km)[32][AT_D + 1] = reinterpret_cast<float (*)[32][AT_D + 1]>(at_smem + 4 * AT_D * 32);       // [2][key][d]
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6, j = lane & 31, h = lane >> 5;
    const int b = blockIdx.y, N = g.N, q0 = blockIdx.x * 128 + wv * 32;
    const size_t boff = (size_t)b * AT_D * N;
    const float *kb = g.k + boff, *vb = g.v + boff;
    const int qi = min(q0 + j, N - 1);
    const unsigned sb = at_seed(g.seed, b);
    const bool drop = HAS_MASK || g.thresh != 0u;
    float qreg[32], doreg[32];
#pragma unroll
    for (int s = 0; s < 32; ++s) {
        qreg[s] = g.q[boff + (size_t)(2 * s + h) * N + qi] * g.scale;
        doreg[s] = g.dout[boff + (size_t)(2 * s + h) * N + qi];
    }
    const float Lq = g.lse[(size_t)b * N + qi], Dv = g.Dq[(size_t)b * N + qi];
    f32x16 G0, G1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { G0[r] = 0.f; G1[r] = 0.f; }

    float rk[8], rv[8];
    at_load(kb, N, 0, t, rk);
    at_load(vb, N, 0, t, rv);
    at_put_cf(Kc[0], t, rk, 1.f); at_put_pm(Km[0], t, rk, 1.f); at_put_cf(Vc[0], t, rv, 1.f);
    __syncthreads();
    const int nt = (N + 31) / 32;
    for (int tt = 0; tt < nt; ++tt) {
        const int cur = tt & 1, j0 = tt * 32;
        if (tt + 1 < nt) { at_load(kb, N, j0 + 32, t, rk); at_load(vb, N, j0 + 32, t, rv); }
        f32x16 acc, dp;
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
        for (int s = 0; s < 32; ++s) {
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(Kc[cur][2 * s + h][j], qreg[s], acc, 0, 0, 0);
            dp = __builtin_amdgcn_mfma_f32_32x32x2f32(Vc[cur][2 * s + h][j], doreg[s], dp, 0, 0, 0);
        }
        float ds[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = j0 + at_drow(r, h);
            const float p = key < N ? __expf(acc[r] - Lq) : 0.f;
            float dpv = dp[r];
            if (drop) dpv = at_keep<HAS_MASK>(g.mask, sb, g.thresh, b, qi, min(key, N - 1), N) ? dpv * g.inv_keep : 0.f;
            ds[r] = p * (dpv - Dv);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = at_drow(r, h);
            G0 = __builtin_amdgcn_mfma_f32_32x32x2f32(Km[cur][row][j], ds[r], G0, 0, 0, 0);
            G1 = __builtin_amdgcn_mfma_f32_32x32x2f32(Km[cur][row][32 + j], ds[r], G1, 0, 0, 0);
        }
        if (tt + 1 < nt) { at_put_cf(Kc[cur ^ 1], t, rk, 1.f); at_put_pm(Km[cur ^ 1], t, rk, 1.f); at_put_cf(Vc[cur ^ 1], t, rv, 1.f); }
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
template <bool HAS_MASK>
__global__ void __launch_bounds__(256, 2) attn_bwd_dkv_kernel(const AttnArgs g, float *__restrict__ dk, float *__restrict__ dv)
{
    extern __shared__ float at_smem[];
    float (*Qc)[AT_D][32] = reinterpret_cast<float (*)[AT_D][32]>(at_smem);                               // [2][c][query]  (scaled)
    float (*Gc)[AT_D][32] = reinterpret_cast<float (*)[AT_D][32]>(at_smem + 2 * AT_D * 32);                // [2][c][query]  dout
    float (*Qm)[32][AT_D + 1] = reinterpret_cast<float (*)[32][AT_D + 1]>(at_smem + 4 * AT_D * 32);        // [2][query][d] (scaled)
    float (*Gm)[32][AT_D + 1] = reinterpret_cast<float (*)[32][AT_D + 1]>(at_smem + 4 * AT_D * 32 + 2 * 32 * (AT_D + 1));
    float (*Ls)[32] = reinterpret_cast<float (*)[32]>(at_smem + 4 * AT_D * 32 + 4 * 32 * (AT_D + 1));       // [2][32]
    float (*Ds)[32] = Ls + 2;
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6, j = lane & 31, h = lane >> 5;
    const int b = blockIdx.y, N = g.N, k0 = blockIdx.x * 128 + wv * 32;
    const size_t boff = (size_t)b * AT_D * N;
    const float *qb = g.q + boff, *gb = g.dout + boff;
    const int ki = min(k0 + j, N - 1);
    const bool kvalid = k0 + j < N;
    const unsigned sb = at_seed(g.seed, b);
    const bool drop = HAS_MASK || g.thresh != 0u;
    float kreg[32], vreg[32];
#pragma unroll
    for (int s = 0; s < 32; ++s) {
        kreg[s] = g.k[boff + (size_t)(2 * s + h) * N + ki];
        vreg[s] = g.v[boff + (size_t)(2 * s + h) * N + ki];
    }
    f32x16 K0, K1, V0, V1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { K0[r] = 0.f; K1[r] = 0.f; V0[r] = 0.f; V1[r] = 0.f; }

    float rq[8], rg[8];
    auto put = [&](int buf, int p0) {
        at_put_cf(Qc[buf], t, rq, g.scale); at_put_pm(Qm[buf], t, rq, g.scale);
        at_put_cf(Gc[buf], t, rg, 1.f); at_put_pm(Gm[buf], t, rg, 1.f);
        if (t < 32) { const int qq = min(p0 + t, N - 1); Ls[buf][t] = g.lse[(size_t)b * N + qq]; Ds[buf][t] = g.Dq[(size_t)b * N + qq]; }
    };
    at_load(qb, N, 0, t, rq);
    at_load(gb, N, 0, t, rg);
    put(0, 0);
    __syncthreads();
    const int nt = (N + 31) / 32;
    for (int tt = 0; tt < nt; ++tt) {
        const int cur = tt & 1, p0 = tt * 32;
        if (tt + 1 < nt) { at_load(qb, N, p0 + 32, t, rq); at_load(gb, N, p0 + 32, t, rg); }
        f32x16 acc, dp;
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
        for (int s = 0; s < 32; ++s) {
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(Qc[cur][2 * s + h][j], kreg[s], acc, 0, 0, 0);
            dp = __builtin_amdgcn_mfma_f32_32x32x2f32(Gc[cur][2 * s + h][j], vreg[s], dp, 0, 0, 0);
        }
        float pd[16], ds[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = at_drow(r, h), qq = p0 + row;
            const bool ok = qq < N && kvalid;
            const float p = ok ? __expf(acc[r] - Ls[cur][row]) : 0.f;
            float pk = p, dpv = dp[r];
            if (drop) {
                const bool keep = at_keep<HAS_MASK>(g.mask, sb, g.thresh, b, min(qq, N - 1), ki, N);
                pk = keep ? p * g.inv_keep : 0.f;
                dpv = keep ? dpv * g.inv_keep : 0.f;
            }
            pd[r] = pk;
            ds[r] = p * (dpv - Ds[cur][row]);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = at_drow(r, h);
            V0 = __builtin_amdgcn_mfma_f32_32x32x2f32(Gm[cur][row][j], pd[r], V0, 0, 0, 0);
            V1 = __builtin_amdgcn_mfma_f32_32x32x2f32(Gm[cur][row][32 + j], pd[r], V1, 0, 0, 0);
            K0 = __builtin_amdgcn_mfma_f32_32x32x2f32(Qm[cur][row][j], ds[r], K0, 0, 0, 0);
            K1 = __builtin_amdgcn_mfma_f32_32x32x2f32(Qm[cur][row][32 + j], ds[r], K1, 0, 0, 0);
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

constexpr size_t AT_DQ_LDS = sizeof(float) * (4 * AT_D * 32 + 2 * 32 * (AT_D + 1));
constexpr size_t AT_DKV_LDS = sizeof(float) * (4 * AT_D * 32 + 4 * 32 * (AT_D + 1) + 4 * 32);

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

extern "C" int gfs3d_attention_forward_f32(const float *q, const float *k, const float *v, int B, int D, int N, float scale, float p_drop,
                                           unsigned long long seed, const uint8_t *keep_mask, float *out, float *lse, void *stream)
{
    const char *who = "gfs3d_attention_forward_f32";
    if (int rc = at_check(who, q, k, v, B, D, N, p_drop)) return rc;
    GFS3D_REQUIRE(out && lse, "%s: NULL output", who);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const AttnArgs a = at_args(q, k, v, B, N, scale, p_drop, seed, keep_mask);
    dim3 grid((N + 127) / 128, B);
    ProfScope prof(8, st);
    if (keep_mask) hipLaunchKernelGGL(attn_fwd_kernel<true>, grid, dim3(256), 0, st, a, out, lse);
    else hipLaunchKernelGGL(attn_fwd_kernel<false>, grid, dim3(256), 0, st, a, out, lse);
    GFS3D_LAUNCH_CHECK("attn_fwd_kernel");
    return GFS3D_OK;
}

extern "C" int gfs3d_attention_backward_f32(const float *q, const float *k, const float *v, const float *out, const float *lse,
                                            const float *dout, int B, int D, int N, float scale, float p_drop, unsigned long long seed,
                                            const uint8_t *keep_mask, float *dq, float *dk, float *dv, float *workspace, void *stream)
{
    const char *who = "gfs3d_attention_backward_f32";
    if (int rc = at_check(who, q, k, v, B, D, N, p_drop)) return rc;
    GFS3D_REQUIRE(out && lse && dout && dq && dk && dv && workspace, "%s: NULL buffer", who);
    hipStream_t st = static_cast<hipStream_t>(stream);
    AttnArgs a = at_args(q, k, v, B, N, scale, p_drop, seed, keep_mask);
    a.out = out; a.lse = lse; a.dout = dout; a.Dq = workspace;
    GFS3D_ENSURE_LDS(attn_bwd_dkv_kernel<true>, AT_DKV_LDS, who);
    GFS3D_ENSURE_LDS(attn_bwd_dkv_kernel<false>, AT_DKV_LDS, who);
    ProfScope prof(9, st);
    hipLaunchKernelGGL(attn_bwd_D_kernel, dim3((N + 255) / 256, B), dim3(256), 0, st, dout, out, N, workspace);
    GFS3D_LAUNCH_CHECK("attn_bwd_D_kernel");
    dim3 grid((N + 127) / 128, B);
    if (keep_mask) {
        hipLaunchKernelGGL(attn_bwd_dq_kernel<true>, grid, dim3(256), AT_DQ_LDS, st, a, dq);
        hipLaunchKernelGGL(attn_bwd_dkv_kernel<true>, grid, dim3(256), AT_DKV_LDS, st, a, dk, dv);
    } else {
        hipLaunchKernelGGL(attn_bwd_dq_kernel<false>, grid, dim3(256), AT_DQ_LDS, st, a, dq);
        hipLaunchKernelGGL(attn_bwd_dkv_kernel<false>, grid, dim3(256), AT_DKV_LDS, st, a, dk, dv);
    }
    GFS3D_LAUNCH_CHECK("attn_bwd kernels");
    return GFS3D_OK;
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
