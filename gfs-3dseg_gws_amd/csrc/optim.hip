// Adam over a LIST of parameter tensors in one launch (the optimiser step of train.py:425-435: torch.optim.Adam with
// per-group learning rates).  torch's fused multi-tensor Adam spends 2 x 37 us per CAPL step on 57 small tensors (chunks of
// 65 536 elements: ~40 workgroups in flight); here every workgroup owns 1024 elements of one tensor - a few hundred
// workgroups, ~5 us - and the per-tensor constants (learning rate, weight decay) ride in the kernel arguments.
#include "common.h"

namespace gfs3d {
namespace {
constexpr int ADAM_CHUNK = 1024;      // elements per workgroup

__global__ void __launch_bounds__(256) adam_multi_kernel(const gfs3d_adam_args a)
{
    int ti = 0;
#pragma unroll 8
    for (int q = 1; q < GFS3D_ADAM_MAX_TENSORS; ++q) ti += (q < a.n_tensors && (int)blockIdx.x >= a.first_block[q]) ? 1 : 0;
    const long long n = a.numel[ti];
    const long long e0 = (long long)((int)blockIdx.x - a.first_block[ti]) * ADAM_CHUNK + 4 * threadIdx.x;
    if (e0 >= n) return;
    float *p = a.param[ti] + e0, *m = a.exp_avg[ti] + e0, *v = a.exp_avg_sq[ti] + e0;
    const float *g = a.grad[ti] + e0;
    const float lr = a.lr[ti], wd = a.weight_decay[ti];
    const float step_size = lr / a.bias_correction1;
    const float inv_bc2_sqrt = 1.0f / sqrtf(a.bias_correction2);
    const int cnt = (int)(n - e0 < 4 ? n - e0 : 4);
    const bool vec = cnt == 4 && (((size_t)p | (size_t)g | (size_t)m | (size_t)v) & 15) == 0;
    float pv[4], gv[4], mv[4], vv[4];
    if (vec) {
        const float4 P = *reinterpret_cast<const float4 *>(p), G = *reinterpret_cast<const float4 *>(g);
        const float4 Mm = *reinterpret_cast<const float4 *>(m), V = *reinterpret_cast<const float4 *>(v);
        pv[0] = P.x; pv[1] = P.y; pv[2] = P.z; pv[3] = P.w; gv[0] = G.x; gv[1] = G.y; gv[2] = G.z; gv[3] = G.w;
        mv[0] = Mm.x; mv[1] = Mm.y; mv[2] = Mm.z; mv[3] = Mm.w; vv[0] = V.x; vv[1] = V.y; vv[2] = V.z; vv[3] = V.w;
    } else {
        for (int i = 0; i < cnt; ++i) { pv[i] = p[i]; gv[i] = g[i]; mv[i] = m[i]; vv[i] = v[i]; }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (i >= cnt) break;
        const float gi = wd != 0.f ? fmaf(wd, pv[i], gv[i]) : gv[i];             // L2 weight decay (torch.optim.Adam's, not AdamW's)
        mv[i] = mv[i] + (gi - mv[i]) * a.one_minus_beta1;                        // exp_avg.lerp_(grad, 1 - beta1)
        vv[i] = fmaf(gi * gi, a.one_minus_beta2, vv[i] * a.beta2);                 // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, 1 - beta2)
        const float denom = sqrtf(vv[i]) * inv_bc2_sqrt + a.eps;
        pv[i] = pv[i] - step_size * (mv[i] / denom);
    }
    if (vec) {
        *reinterpret_cast<float4 *>(p) = make_float4(pv[0], pv[1], pv[2], pv[3]);
        *reinterpret_cast<float4 *>(m) = make_float4(mv[0], mv[1], mv[2], mv[3]);
        *reinterpret_cast<float4 *>(v) = make_float4(vv[0], vv[1], vv[2], vv[3]);
    } else {
        for (int i = 0; i < cnt; ++i) { p[i] = pv[i]; m[i] = mv[i]; v[i] = vv[i]; }
    }
}
}  // namespace
}  // namespace gfs3d

using namespace gfs3d;

extern "C" int gfs3d_adam_multi_f32(gfs3d_adam_args *a, void *stream)
{
    const char *who = "gfs3d_adam_multi_f32";
    GFS3D_REQUIRE(a, "%s: NULL args", who);
    GFS3D_REQUIRE(a->n_tensors >= 1 && a->n_tensors <= GFS3D_ADAM_MAX_TENSORS, "%s: n_tensors=%d (1..%d per call)", who, a->n_tensors,
                  GFS3D_ADAM_MAX_TENSORS);
    GFS3D_REQUIRE(a->bias_correction1 > 0.f && a->bias_correction2 > 0.f, "%s: bias corrections must be positive (step >= 1)", who);
    if (a->one_minus_beta1 == 0.f && a->beta1 != 1.f) a->one_minus_beta1 = (float)(1.0 - (double)a->beta1);      // (callers that leave them zero)
    if (a->one_minus_beta2 == 0.f && a->beta2 != 1.f) a->one_minus_beta2 = (float)(1.0 - (double)a->beta2);
    long long blocks = 0;
    for (int i = 0; i < a->n_tensors; ++i) {
        GFS3D_REQUIRE(a->param[i] && a->grad[i] && a->exp_avg[i] && a->exp_avg_sq[i] && a->numel[i] >= 1, "%s: NULL or empty tensor %d", who, i);
        a->first_block[i] = (int)blocks;
        blocks += (a->numel[i] + ADAM_CHUNK - 1) / ADAM_CHUNK;
        GFS3D_REQUIRE(blocks < (1ll << 30), "%s: too many elements", who);
    }
    hipLaunchKernelGGL(adam_multi_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), *a);
    GFS3D_LAUNCH_CHECK("adam_multi_kernel");
    return GFS3D_OK;
}
