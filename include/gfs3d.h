/* gfs3d.h - C ABI of libgfs3d_hip.so: the MI355X (gfx950) hot path of
 * Pixie8888/GFS-3DSeg_GWs (DGCNN kNN graph + EdgeConv + geometric-word /
 * prototype cosine matching).
 *
 * The reference has no FFI: its "interface" for this path is the set of Python
 * functions below.  Each entry point names the reference
 * code it replaces.  All pointers are DEVICE pointers unless stated; all
 * tensors are dense fp32 in the layout given; `stream` is a hipStream_t passed
 * as void* (NULL = default stream).  No entry point allocates, frees or
 * synchronises: callers hand in outputs and a workspace (graph-capture safe).
 * The training EdgeConv forward forks a few kernels onto a library-owned side
 * stream (one per device, created on first use) and joins it back into `stream`
 * with events before it returns work to the caller; GFS3D_EC_OVERLAP=0 keeps
 * everything on `stream`.
 *
 * Return value: 0 on success, non-zero error code otherwise;
 * gfs3d_last_error() gives a thread-local human-readable message.  Nothing
 * here ever falls back to a CPU path.
 */
#ifndef GFS3D_H
#define GFS3D_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GFS3D_OK 0
#define GFS3D_ERR_INVALID 1      /* bad shape / null pointer / unsupported size */
#define GFS3D_ERR_WORKSPACE 2    /* workspace too small */
#define GFS3D_ERR_HIP 3          /* a HIP runtime call or launch failed */

/* Arithmetic of the GEMM-shaped kernels (EdgeConv conv2, point-wise stack, word / prototype matching, k-means),
 * chosen per call.  Every tensor that crosses this ABI is fp32 in every mode (exceptions are explicit fields:
 * gfs3d_pointwise_args.store16 / x_type, gfs3d_pointwise_bwd_args.dx_type, gfs3d_pack_args.dst_type); what changes is the
 * matrix-core operand form - and, in the reduced modes, the storage of tensors that live INSIDE an operator between its
 * forward and its backward: the point-wise stacks' pre-BatchNorm activations (the mode's 16-bit format), their gradients
 * and the packed fusion input's gradient (bf16), the EdgeConv backward's per-edge dy1 rows (bf16 pairs).
 *   GFS3D_COMPUTE_F32   fp32 accuracy (default): every fp32 operand enters the bf16 matrix cores as the exact sum of
 *                       three bf16 pieces, six piece products per fp32 product - within 1e-4 of the fp32 reference
 *   GFS3D_COMPUTE_BF16  ONE bf16 product per fp32 product (operands rounded to nearest-even), fp32 accumulate:
 *                       BASELINE configs[4]'s reduced-precision mode; tolerance vs the fp32 oracle in DESIGN.md
 *   GFS3D_COMPUTE_F16   the same with IEEE half operands in the FORWARD passes (11 significand bits instead of 8);
 *                       backward passes use bf16 operands (gradients of a mean loss underflow in half)
 * kNN (bit-exact contract), BatchNorm statistics and all reductions stay fp32 / fp64 in every mode. */
#define GFS3D_COMPUTE_F32 0
#define GFS3D_COMPUTE_BF16 1
#define GFS3D_COMPUTE_F16 2

#define GFS3D_EC_WIDTH 64        /* EdgeConv block widths supported: [64, 64] - the reference's only configuration
                                    (train.py:776 `--edgeconv_widths [[64,64],[64,64],[64,64]]`, never overridden by its scripts).
                                    The heavy kernels are built around it: one point's 64 channels are one 256-B row (one
                                    coalesced load / store per wave), conv2 is exactly two 32x32 MFMA tiles per lane half, and the
                                    max over k, the BatchNorm sums and G_Q are in-lane reductions over the accumulator registers
                                    of that tile shape.  Other widths return GFS3D_ERR_INVALID; the Python mirror (model/dgcnn.py: DGCNN.edgeconv_features)
                                    runs them - and k > 32, and a backward pass in eval mode - through its general path: the
                                    graph from gfs3d_knn_topk_*, the block itself as device tensor ops. */
#define GFS3D_MAX_K 32           /* neighbours per point supported by the EdgeConv kernels */

int gfs3d_version(void);
const char *gfs3d_last_error(void);

/* Measurement hooks (bench.py): when enabled, the launchers bracket the heavy kernels
 * with hipEvents on the launch stream.  gfs3d_profile_read synchronises the recorded
 * events of one kernel id, returns the summed device time in ms and the number of
 * launches, and clears them.  Ids: 0 knn_topk, 1 ec_main_fwd, 2 ec_main_bwd,
 * 3 proto_mfma (word assign / get_pred), 4 ec_stats1, 5 kmeans_assign, 6 pointwise stack forward (all its
 * kernels), 7 pointwise stack backward, 8 attention forward, 9 attention backward. */
#define GFS3D_PROF_KERNELS 10
int gfs3d_profile_enable(int on);
int gfs3d_profile_read(int kernel_id, double *total_ms, long long *launches);

/* ------------------------------------------------------------------ K1 --
 * Fused pairwise-distance + top-k.  Replaces knn(x, k), model/dgcnn.py:17-23
 * (bmm + 3 elementwise (B,N,N) passes + topk).
 *   x      (B, C, N) channel-first (the reference layout)
 *   idx64  (B, N, k) int64, nearest first - what knn() returns; may be NULL
 *   idx32  (B, N, k) int32 copy consumed by the EdgeConv kernels; may be NULL
 * Numerics contract (oracle/knn_oracle.c): dot = fp32 fma chain over c,
 * pd = fl(fl(2 dot - |xj|^2) - |xi|^2), ties -> lower index first.
 * Requires 1 <= k <= min(N, 64), C >= 1.  workspace: gfs3d_knn_workspace_bytes() bytes of
 * device scratch (the per-point squared norms; for 16 <= C <= 64 also the candidate lists of the screened path). */
size_t gfs3d_knn_workspace_bytes(int B, int C, int N, int k);
int gfs3d_knn_topk_f32(const float *x, int B, int C, int N, int k,
                       int64_t *idx64, int32_t *idx32, void *workspace, size_t workspace_bytes, void *stream);

/* The same with the point norms supplied by the caller: sqnorm[b*N + n] = sum_c x[b][c][n]^2 in sequential-c order with
 * rounded squares (what gfs3d_edgeconv_forward_f32 leaves in y_sqnorm), B*N floats followed by >= 16 readable floats.
 * Shapes outside the matrix path (C > 64 or k > 32) ignore the norms and behave like gfs3d_knn_topk_f32 (which then
 * needs no workspace either). */
int gfs3d_knn_topk_norms_f32(const float *x, const float *sqnorm, int B, int C, int N, int k,
                             int64_t *idx64, int32_t *idx32, void *stream);

/* The general form: x_pm, if not NULL, is the point-major copy (B, N, 64) of x with C == 64 (what the EdgeConv entry points
 * leave in y_pm; ignored for other C) and sqnorm, if not NULL, the point norms as above.  With C <= 64, k <= 32, N >= 8
 * and the full workspace (gfs3d_knn_workspace_bytes) the graph is built by the SCREENED path: all distances approximately
 * on the bf16 matrix cores with a rigorous error bound (operands of <= 16 channels in one K-step, else four), the exact
 * fp32 chain only for the ~k + 10 candidates per point that can reach the result - the same indices bit for bit. */
int gfs3d_knn_topk_pm_f32(const float *x, const float *x_pm, const float *sqnorm, int B, int C, int N, int k,
                          int64_t *idx64, int32_t *idx32, void *workspace, size_t workspace_bytes, void *stream);

/* The same with the screen's operand image handed over as well: x_img = what gfs3d_edgeconv_forward_f32 leaves in y_knn_img
 * for the same features (used with C == 64 and x_pm given; ignored otherwise) - the call then launches the screen and the
 * exact pass only. */
#define GFS3D_KNN_IMG_SLACK_ROWS 128
int gfs3d_knn_topk_img_f32(const float *x, const float *x_pm, const void *x_img, const float *sqnorm, int B, int C, int N, int k,
                           int64_t *idx64, int32_t *idx32, void *workspace, size_t workspace_bytes, void *stream);

/* ------------------------------------------------------------------ K2 --
 * One EdgeConv block = get_edge_feature (dgcnn.py:26-42) + conv2d block
 * [1x1 conv -> BatchNorm2d -> LeakyReLU(0.2)] x2 (dgcnn.py:45-61) + max over
 * k (dgcnn.py:118), widths [64, 64], never materialising (B,2C,N,k). */
typedef struct gfs3d_edgeconv_args {
    int B, C, N, k;               /* C = input channels of this block */
    int training;                 /* 1: batch statistics over (B,N,k); 0: running statistics */
    float eps, momentum, slope;   /* 1e-5, 0.1, 0.2 in the reference */
    const float *x;               /* (B, C, N) channel-first */
    const int32_t *idx;           /* (B, N, k) from gfs3d_knn_topk_f32 */
    const float *w1;              /* (64, 2C)  edge_convs[i].layer[0].weight */
    const float *bn1_w, *bn1_b;   /* (64)      edge_convs[i].layer[1] */
    float *bn1_rm, *bn1_rv;       /* (64) running stats; updated in place when training */
    const float *w2;              /* (64, 64)  edge_convs[i].layer[3].weight */
    const float *bn2_w, *bn2_b;   /* (64)      edge_convs[i].layer[4] */
    float *bn2_rm, *bn2_rv;
    float *y_cf;                  /* out: (B, 64, N) channel-first; batch stride below */
    long long y_cf_bstride;       /* elements between consecutive b in y_cf (>= 64*N) */
    float *y_pm;                  /* out: (B, N, 64) point-major copy (input layout of the next block) */
    /* saved for backward (outputs of forward, inputs of backward) */
    float *P, *Q;                 /* (B, N, 64): Wa x_n and (Wb - Wa) x_n */
    float *SP;                    /* (B, N, 64): sum_j P[idx[n,j]] (written by forward) */
    float *SQ;                    /* (B, N, 64): sum over in-edges of Q (reverse kNN graph; written by the training forward) */
    float *indeg;                 /* (B, N): in-degree in the kNN graph, as float (written by the training forward) */
    float *stats;                 /* (8, 64): mean1 rstd1 scale1 shift1 mean2 rstd2 scale2 shift2 */
    float *hsel;                  /* (B, N, 64) pre-BN2 value that won the max */
    uint8_t *jsel;                /* (B, N, 64) neighbour slot j that won the max */
    void *workspace;
    size_t workspace_bytes;       /* >= gfs3d_edgeconv_workspace_bytes() */
    const float *x_pm;            /* optional (B*N, C) point-major copy of x (the previous block's y_pm): with C == 64 the
                                     two projections run as one MFMA GEMM; NULL = channel-first path */
    float *Pp, *Qp;               /* optional (B, N, 64) each: the BatchNorm-folded projections the heavy kernels read; forward
                                     writes them here and backward reuses them (NULL: scratch, recomputed in backward) */
    float *y_sqnorm;              /* optional out (B*N floats + 16 floats of slack): |y_n|^2 summed over the 64 channels in the kNN
                                     kernel's order (sequential c, rounded squares) - hand it to gfs3d_knn_topk_norms_f32 for
                                     the next layer's graph instead of a norm pass */
    int compute;                  /* GFS3D_COMPUTE_*: arithmetic of conv2 (heavy kernels) and of the point-major projection GEMM */
    int32_t *csr_rowptr;          /* optional, saved for backward: (B*N + 1) first CSR row of every target point's in-edges */
    int32_t *csr_pos;             /* optional, saved for backward: (B, N, k) CSR row of every edge.  Both given (and N k < 65536
                                     edges per block; idx values outside [0, N) are a caller error - the graph builder clamps them): the training forward builds the reverse kNN graph in CSR form and the
                                     backward stores per-edge rows + sums CSR segments instead of float atomics - every sum over
                                     the in-edges of a point then has a fixed order.  NULL: slot table + float atomics */
    void *y_knn_img;              /* optional out, (B*N + GFS3D_KNN_IMG_SLACK_ROWS) rows of 256 bytes: the operand rows of y for the
                                     next block's screened kNN ([hi 64 x bf16 | mid 64 x bf16] per point, the slack rows zeroed) -
                                     hand it to gfs3d_knn_topk_img_f32 together with y_pm and y_sqnorm instead of an image pass */
} gfs3d_edgeconv_args;

size_t gfs3d_edgeconv_workspace_bytes(int B, int C, int N, int k);

/* Reproducibility of the EdgeConv entry points.  With csr_rowptr / csr_pos given, every sum over the in-edges of a point
 * (SQ, the input-gradient scatter) and the conv2 weight gradient have a fixed order.  What remains are the 128-wide
 * BatchNorm reductions: by default float64 atomics (the order of the adds can change a double sum in its last bit);
 * gfs3d_set_deterministic(1) makes the producers write per-block rows that are added in a fixed order (two small extra
 * launches per reduction): forward and backward are then bit-reproducible from run to run.  Process-wide switch. */
int gfs3d_set_deterministic(int on);
int gfs3d_edgeconv_forward_f32(const gfs3d_edgeconv_args *a, void *stream);

/* Backward of the block above (autograd of dgcnn.py:116-118 in the reference).
 *   dy_cf  (B, 64, N) channel-first gradient of y, batch stride dy_bstride
 * outputs: dx (B,C,N); dw1 (64,2C); dbn1_w/b (64); dw2 (64,64); dbn2_w/b (64).
 * Uses (and overwrites) the workspace; P,Q,SP,SQ,indeg,stats,hsel,jsel,y_pm must
 * be what forward produced for the same inputs. */
typedef struct gfs3d_edgeconv_bwd_args {
    gfs3d_edgeconv_args f;
    const float *dy_cf;
    long long dy_bstride;
    float *dx;
    float *dw1, *dbn1_w, *dbn1_b, *dw2, *dbn2_w, *dbn2_b;
    const float *dy_pm;      /* optional (B*N, 64) point-major gradient w.r.t. y_pm, ADDED to dy_cf (either may be NULL) */
} gfs3d_edgeconv_bwd_args;

int gfs3d_edgeconv_backward_f32(const gfs3d_edgeconv_bwd_args *a, void *stream);

/* Phased variants for SYNCHRONISED BatchNorm under block-sharded data parallelism (the reference
 * is one device = whole-batch statistics).  Both passes split at their two global reductions:
 *   forward   phase 0: projections + BN1 sums -> acc[0:128]      phase 1: BN1 finalize, heavy pass, BN2
 *             sums -> acc[128:256]                                phase 2: BN2 finalize, epilogue
 *   backward  phase 0: BN2-backward sums -> acc[0:128]            phase 1: heavy pass ..., BN1-backward sums
 *             -> acc[128:256]                                     phase 2: BN1 backward, dx, dW1
 * Between phases the caller all-reduces (SUM) the 128 doubles just produced over the ranks; `acc`
 * is a caller-owned device buffer of 512 doubles (the library keeps each rank's own sums in
 * acc[256:512] for the parameter gradients), `stat_count` the GLOBAL number of (b,n,k) samples
 * (<= 0: this rank's).  The workspace must not be touched between the phases of one pass.
 * Phases 0,1,2 with acc = NULL, stat_count = 0 are exactly the monolithic calls above. */
int gfs3d_edgeconv_forward_phase_f32(const gfs3d_edgeconv_args *a, int phase, double *acc, double stat_count, void *stream);
int gfs3d_edgeconv_backward_phase_f32(const gfs3d_edgeconv_bwd_args *a, int phase, double *acc, double stat_count, void *stream);

/* ------------------------------------------------------------------ K4 --
 * Point-wise MLP stack [Conv1d 1x1 (no bias) -> BatchNorm1d -> LeakyReLU(slope)] x n_layers on the
 * concatenation of n_seg 64-channel feature maps.  Replaces `self.conv(torch.cat(edgeconv_outputs, 1))`,
 * model/dgcnn.py:121-122 with the conv1d stack of model/dgcnn.py:64-80 (2 x [cuDNN conv, BatchNorm,
 * LeakyReLU] + the cat in the reference).
 *   x_seg[s]  (B*N, 64) point-major: the y_pm buffers the EdgeConv entry points leave behind
 *   w[l]      (width[l], width[l-1]) = nn.Conv1d.weight[:, :, 0]; width[-1] = 64 * n_seg
 *   bn_*[l]   BatchNorm1d affine parameters and running statistics (updated in place when training)
 *   h[l]      (B*N, width[l]) pre-BatchNorm activations, kept for the backward (training); in eval mode
 *             the post-activation values (h[n_layers-1] may then be NULL)
 *   stats[l]  (4, width[l]) scale, shift, mean, 1/std actually applied
 *   y_cf      (B, width[n_layers-1], N) channel-first result (what the reference returns)
 * widths: multiples of 64, <= 1024.  Backward (training mode): dy_cf (B, width[last], N) ->
 * dx_seg[s] (B*N, 64) point-major, dw[l], dbn_w[l], dbn_b[l]; f.y_cf and the running statistics are not used. */
#define GFS3D_PW_MAX_LAYERS 4
#define GFS3D_PW_MAX_SEGS 4
#define GFS3D_PW_SEG_WIDTH 64
typedef struct gfs3d_pointwise_args {
    int B, N, n_seg, n_layers;
    int width[GFS3D_PW_MAX_LAYERS];
    int training;
    float eps, momentum, slope;
    const float *x_seg[GFS3D_PW_MAX_SEGS];
    const float *w[GFS3D_PW_MAX_LAYERS];
    const float *bn_w[GFS3D_PW_MAX_LAYERS];
    const float *bn_b[GFS3D_PW_MAX_LAYERS];
    float *bn_rm[GFS3D_PW_MAX_LAYERS];
    float *bn_rv[GFS3D_PW_MAX_LAYERS];
    float *h[GFS3D_PW_MAX_LAYERS];
    float *stats[GFS3D_PW_MAX_LAYERS];
    float *y_cf;
    void *workspace;
    size_t workspace_bytes;
    int compute;                  /* GFS3D_COMPUTE_*: arithmetic of the row GEMMs and weight-gradient GEMMs */
    /* ---- generalisations for the CAPL heads; all zero / NULL = the DGCNN stack described above ----
     * BaseLearner (model/dgcnn.py:130-152: Conv1d WITH bias + BatchNorm1d, ReLU between the layers, none after the last)
     * and the fusion layer (model/capl.py:64-65, 359: Conv1d(392 -> 128, bias) + BatchNorm1d + LeakyReLU(0.2)). */
    int seg_width;                /* width of every input segment (0 = 64); a multiple of 64, seg_width * n_seg <= 1024 */
    const float *bias[GFS3D_PW_MAX_LAYERS];   /* Conv1d bias of layer l or NULL.  In front of a BatchNorm it only moves the
                                     running mean (training) / the shift (eval); h[l] is stored WITHOUT it */
    int per_layer_slope;          /* != 0: layer l ends in LeakyReLU(slope_l[l]) (0 = ReLU, 1 = no activation) instead of `slope` */
    float slope_l[GFS3D_PW_MAX_LAYERS];
    float *y_pm;                  /* optional (B*N, width[last]) point-major copy of the result; y_cf may be NULL when it is given */
    /* ---- 16-bit storage of the stack's INTERNAL tensors (reduced modes; BASELINE configs[4]) ----
     * store16 != 0, training, compute != GFS3D_COMPUTE_F32: h[l] point at 2-byte elements in the forward operand format of the
     * mode - IEEE half (saturating at +-65504) under GFS3D_COMPUTE_F16, bf16 under GFS3D_COMPUTE_BF16 - and the gradients the
     * backward keeps in the workspace are bf16: half the HBM traffic of every kernel of the stack.  Accumulators, BatchNorm
     * sums / constants, parameters, parameter gradients and the tensors at the call's boundary (x_seg, y_*, dy_*, dx_seg) stay
     * fp32 - unless x_type / dx_type say otherwise.  Ignored (fp32 storage) in eval mode and under GFS3D_COMPUTE_F32. */
    int store16;
    void *x16;                    /* optional, store16 with fp32 x_seg: (B*N, n_seg * seg_width) 2-byte elements.  The forward's first GEMM
                                     writes the operands it forms from x_seg (the mode's 16-bit format) here; the backward's weight
                                     gradient of layer 0 reads them instead of the fp32 rows (which it would otherwise fetch once per
                                     128-channel output tile) */
    int x_type;                   /* 0: x_seg are fp32; GFS3D_COMPUTE_BF16 / _F16 (= the call's `compute`, store16 set): x_seg hold bf16 /
                                     half elements (the packed fusion input written by gfs3d_pack_rows_t) */
} gfs3d_pointwise_args;

typedef struct gfs3d_pointwise_bwd_args {
    gfs3d_pointwise_args f;
    const float *dy_cf;
    float *dx_seg[GFS3D_PW_MAX_SEGS];
    float *dw[GFS3D_PW_MAX_LAYERS];
    float *dbn_w[GFS3D_PW_MAX_LAYERS];
    float *dbn_b[GFS3D_PW_MAX_LAYERS];
    const float *dy_pm;           /* optional (B*N, width[last]) gradient w.r.t. y_pm, ADDED to dy_cf (either may be NULL) */
    int dx_type;                  /* 0: dx_seg are fp32; 1 (store16 set): bf16 elements (read back by gfs3d_unpack_rows_t) */
} gfs3d_pointwise_bwd_args;

size_t gfs3d_pointwise_workspace_bytes(int B, int N, int n_seg, int n_layers, const int *width);
/* the same for segments of `seg_width` channels (0 = 64) */
size_t gfs3d_pointwise_workspace_bytes_ex(int B, int N, int n_seg, int seg_width, int n_layers, const int *width);
int gfs3d_pointwise_forward_f32(const gfs3d_pointwise_args *a, void *stream);
int gfs3d_pointwise_backward_f32(const gfs3d_pointwise_bwd_args *a, void *stream);
/* In a monolithic (non-phased, non-deterministic) training call the LAST workgroup of the kernel that completes a layer's
 * BatchNorm column sums also finalises that layer (scale / shift / running statistics in the forward; coefficients, dgamma,
 * dbeta in the backward) instead of a tiny launch of its own.  gfs3d_set_bn_tail(0) - or GFS3D_PW_TAIL=0 in the
 * environment at load time - selects the separate finalisation kernels (same results up to the order of the fp64 atomic
 * sums): a field switch for bisecting, and what tests/test_gpu_stress.py flips.  Diagnostic contract used by that test:
 * after either call the first n_layers * 2 * maxw doubles of `workspace` (maxw = the largest width[l]) still hold the
 * column sums the finalisation was computed from: layer l's two rows of width[l] doubles at [2 maxw l] and [2 maxw l + width[l]]. */
int gfs3d_set_bn_tail(int on);

/* Phased variants for SYNCHRONISED BatchNorm (see the EdgeConv phases above).  L = n_layers, maxw = the largest layer
 * width, `acc` a caller-owned device buffer of 2 * L * 2 * maxw doubles: rows [l][2][maxw] of global sums, followed by
 * the library's copies of this rank's own sums (parameter gradients).  `stat_count`: the GLOBAL number of points.
 *   forward   phase p < L: GEMM of layer p -> column sums in row p (the caller all-reduces row p, 2 * maxw doubles,
 *             before phase p + 1); phase L: last BatchNorm + the channel-first output
 *   backward  phase 0: gradient of the last pre-activation -> sums in row L - 1; phase q = 1 .. L: layer l = L - q from
 *             the all-reduced row l: BatchNorm backward, dW_l, gradient of layer l - 1 -> sums in row l - 1 (dx for l = 0)
 * Phases 0 .. L in order are exactly the monolithic calls. */
int gfs3d_pointwise_forward_phase_f32(const gfs3d_pointwise_args *a, int phase, double *acc, double stat_count, void *stream);
int gfs3d_pointwise_backward_phase_f32(const gfs3d_pointwise_bwd_args *a, int phase, double *acc, double stat_count, void *stream);

/* Plain bias-free 1x1 convolution without BatchNorm: y = x W^T.  Replaces q_map / k_map / v_map of SelfAttention
 * (model/attention.py:39-41) as ONE GEMM on the row-concatenated weights (Nc = 3 * 64).
 *   x_pm (B*N, K) point-major, K a multiple of 64 (<= 1024);  w (Nc, K), Nc = 64 * n_out, n_out <= 4
 *   y_cf[s] (B, 64, N) channel-first output s (q, k, v: the layout gfs3d_attention_* reads)
 * backward: dy_cf[s] (B, 64, N) -> dx_pm (B*N, K), dw (Nc, K).  Workspace: gfs3d_linear_workspace_bytes(B, N, K, Nc). */
size_t gfs3d_linear_workspace_bytes(int B, int N, int K, int Nc);
int gfs3d_linear_forward_f32(const float *x_pm, int B, int N, int K, const float *w, int Nc, float *const *y_cf, int compute,
                             void *workspace, size_t workspace_bytes, void *stream);
int gfs3d_linear_backward_f32(const float *x_pm, int B, int N, int K, const float *w, int Nc, const float *const *dy_cf,
                              float *dx_pm, float *dw, int compute, void *workspace, size_t workspace_bytes, void *stream);

/* Concatenation along the channels into ONE point-major matrix, and its adjoint.  Replaces the torch.cat of
 * model/capl.py:357-359 (soft word assignment | edge feature | attention | base feature -> fusion) together with the
 * layout change the point-major GEMMs need.
 *   src[s]  width[s] channels, channel-first (B, width[s], N) if channel_first[s] else point-major (B*N, width[s])
 *   dst     (B*N, dst_ld) point-major; columns [sum width, dst_ld) are written as zeros (K padding of the GEMM)
 * gfs3d_unpack_rows_f32 is the adjoint: it scatters the columns of `dst` (read-only there) back into src[s] (written); a NULL
 * src[s] skips that source's columns (its gradient is formed elsewhere, e.g. gfs3d_word_soft_backward_rows). */
#define GFS3D_PACK_MAX_SRC 8
typedef struct gfs3d_pack_args {
    int B, N, n_src;
    float *src[GFS3D_PACK_MAX_SRC];
    int width[GFS3D_PACK_MAX_SRC];
    int channel_first[GFS3D_PACK_MAX_SRC];
    float *dst;
    int dst_ld;
    int dst_type;                 /* element type of `dst`: 0 fp32; GFS3D_COMPUTE_BF16 bf16; GFS3D_COMPUTE_F16 IEEE half (packing only) -
                                     the 16-bit fusion input / its bf16 gradient of a reduced-mode call (gfs3d_pointwise_args.x_type,
                                     gfs3d_pointwise_bwd_args.dx_type); `src` are always fp32 */
} gfs3d_pack_args;
int gfs3d_pack_rows_f32(const gfs3d_pack_args *a, void *stream);
int gfs3d_unpack_rows_f32(const gfs3d_pack_args *a, void *stream);

/* ------------------------------------------------------------------ K5 --
 * Fused self-attention over the points of a block.  Replaces SelfAttention.forward after the three 1x1 projections,
 * model/attention.py:39-48:  y = dropout(softmax_k(q^T k * scale)) v^T  without materialising the (B, N, N) scores.
 *   q, k, v (B, 64, N) channel-first; scale = 1 / sqrt(64) in the reference; out (B, 64, N); lse (B, N) = log-sum-exp
 *   of every query's scores (kept for the backward).
 *   Dropout (p_drop > 0): weight (q, k) of item b is kept iff hash(seed, b, q N + k) >= p_drop 2^32 and scaled by
 *   1 / (1 - p_drop); the backward regenerates the same decisions from (seed, p_drop).  keep_mask, if not NULL, is an
 *   explicit (B, N, N) byte mask used instead (tests; p_drop then only sets the scale).
 *   gfs3d_attention_dropout_mask writes the decisions of (seed, p_drop) as such a mask.
 * Backward: dout (B, 64, N) -> dq, dk, dv (B, 64, N); workspace: B*N floats. */
int gfs3d_attention_forward_f32(const float *q, const float *k, const float *v, int B, int D, int N, float scale, float p_drop,
                                unsigned long long seed, const uint8_t *keep_mask, float *out, float *lse, void *stream);
int gfs3d_attention_backward_f32(const float *q, const float *k, const float *v, const float *out, const float *lse,
                                 const float *dout, int B, int D, int N, float scale, float p_drop, unsigned long long seed,
                                 const uint8_t *keep_mask, float *dq, float *dk, float *dv, float *workspace, void *stream);
int gfs3d_attention_dropout_mask(int B, int N, float p_drop, unsigned long long seed, uint8_t *mask, void *stream);
/* The same with the operand precision of the matrix products chosen per call (GFS3D_COMPUTE_*; _f32 above = GFS3D_COMPUTE_F32:
 * exact 3-way bf16 split, fp32 accuracy).  Scores, softmax, dropout and all accumulators stay fp32; the backward of an F16
 * forward runs on bf16 operands. */
int gfs3d_attention_forward_mixed(const float *q, const float *k, const float *v, int B, int D, int N, float scale, float p_drop,
                                  unsigned long long seed, const uint8_t *keep_mask, float *out, float *lse, int compute, void *stream);
int gfs3d_attention_backward_mixed(const float *q, const float *k, const float *v, const float *out, const float *lse,
                                   const float *dout, int B, int D, int N, float scale, float p_drop, unsigned long long seed,
                                   const uint8_t *keep_mask, float *dq, float *dk, float *dv, float *workspace, int compute,
                                   void *stream);

/* ------------------------------------------------------------------ K3 --
 * Point x prototype cosine matching on MFMA (fp32-in / fp32-acc).
 *
 * gfs3d_word_assign_f32: geometric-word assignment, model/capl.py:344-353
 *   feat   (B, C, N) channel-first, batch stride feat_bstride
 *   words  (Kw, C)   geometric words (NOT normalised; normalised on load)
 *   soft   (B, Kw, N) softmax(scale * cos) over words   (may be NULL)
 *   assign (B, N) int32 arg-max word                    (may be NULL)
 *   onehot (B, Kw, N) float one-hot of assign           (may be NULL)
 * gfs3d_proto_cosine_f32: scale * cosine logits, model/capl.py:290-322 (get_pred)
 *   proto (P, K, C) with P == 1 (shared) or P == B (per-sample); out (B, K, N)
 * gfs3d_kmeans_assign_f32: Lloyd assignment + centroid partial sums for
 *   get_basis.py:210 (sklearn KMeans on CPU in the reference)
 *   pts (M, C) row-major, cent (K, C); label (M) int32; sums (K, C) and counts (K)
 *   are ACCUMULATED into (zero them first); inertia (1) double accumulated. */
int gfs3d_word_assign_f32(const float *feat, long long feat_bstride, int B, int C, int N,
                          const float *words, int Kw, float scale,
                          float *soft, int32_t *assign, float *onehot, void *stream);
int gfs3d_proto_cosine_f32(const float *feat, long long feat_bstride, int B, int C, int N,
                           const float *proto, int P, int K, float scale, float *out, void *stream);
/* Backward of the two cosine heads with respect to the point features: dup = gradient of `soft` (word
 * assignment; pass the forward's soft) or of the logits (get_pred; soft = NULL); dfeat (B, C, N) dense, C <= 192. */
int gfs3d_proto_backward_feat_f32(const float *feat, long long feat_bstride, int B, int C, int N,
                                  const float *proto, int P, int K, float scale,
                                  const float *soft, const float *dup, float *dfeat, void *stream);
/* Backward of gfs3d_proto_cosine_f32 with respect to the prototypes (get_pred, capl.py:308-322): dout (B, K, N) ->
 * dproto (P, K, C); K <= 32, C <= 192.  Partial sums per block + a fixed-order reduction (bit-reproducible). */
size_t gfs3d_proto_backward_proto_workspace_bytes(int B, int C, int N);
int gfs3d_proto_backward_proto_f32(const float *feat, long long feat_bstride, int B, int C, int N,
                                   const float *proto, int P, int K, float scale, const float *dout, float *dproto,
                                   void *workspace, size_t workspace_bytes, void *stream);
/* The prototype head of the CAPL TRAINING step, fused (reference model/capl.py:194-242: generate_fake_proto :364-411, three
 * get_pred calls :290-322, post_refine_proto_v2 :245-287, the two CrossEntropyLoss(ignore_index) calls of train.py:317 and
 * `x_pre_2.max(1)[1]`), forward and backward.  Replaces ~200 small tensor-library launches per step by four passes over the
 * feature tensor and a handful of per-class kernels (csrc/head.hip); bit-reproducible (no atomics).
 *   forward phase 0:  feat, label, main_proto, bg_proto, fake[] -> fsum (the fake-novel classes' masked feature sums | point
 *                     counts over the support half: items b >= half).  Under block-sharded data parallelism the caller
 *                     all-reduces (SUM) fsum between the phases (capl.py:378-408 averages over the whole support half).
 *   forward phase 1:  -> loss[0] (= 0.5 loss_1 + 0.5 loss_2), loss[1] (number of counted points), pred (B, N) int64
 *   backward phase 0: dloss -> dfsum (gradient of fsum: all-reduce SUM under data parallelism)
 *   backward phase 1: -> dfeat (B, C, N), dmain (classes, C), dbg (1, C)
 * C must be 128 (main_dim, capl.py:46), classes <= 23, n_fake <= GFS3D_HEAD_MAX_FAKE.  The workspace carries the state
 * from the forward to the backward and must not be reused in between.  Labels outside [0, classes] other than
 * ignore_index are not counted (the reference would raise). */
#define GFS3D_HEAD_MAX_FAKE 16
typedef struct gfs3d_capl_head_args {
    int B, C, N;
    const float *feat;            /* (B, C, N) point_feat, rows dense */
    long long feat_bstride;       /* floats between batch items */
    const int64_t *label;         /* (B, N): 0 = background, 1 .. classes */
    long long ignore_index;
    const float *main_proto;      /* (classes, C) */
    int classes;
    const float *bg_proto;        /* (1, C) */
    int base_num;                 /* classes [0, base_num) keep the refined prototype (capl.py:229-230) */
    int n_fake;
    int fake[GFS3D_HEAD_MAX_FAKE];/* HOST values: labels (1-based) of the fake-novel classes drawn for this step */
    int half;                     /* first batch item of the support half (B / 2) */
    float *fsum;                  /* (GFS3D_HEAD_MAX_FAKE, C + 1) */
    void *workspace;
    size_t workspace_bytes;       /* >= gfs3d_capl_head_workspace_bytes() */
    float *loss;                  /* 2 floats */
    int64_t *pred;                /* (B, N) */
    const float *dloss;           /* backward: device scalar, gradient of loss[0] */
    float *dfsum;                 /* (GFS3D_HEAD_MAX_FAKE, C + 1) */
    float *dfeat;
    float *dmain;
    float *dbg;
    void *stream;
} gfs3d_capl_head_args;
/* mask[0] = OR over the n labels y in [0, 64) of (1 << y): the set of classes present in a label tensor (the `unique(y)` of
 * generate_fake_proto, capl.py:378-386, without a device sort); labels outside [0, 64) are skipped. */
int gfs3d_label_presence(const int64_t *label, long long n, uint64_t *mask, void *stream);
size_t gfs3d_capl_head_workspace_bytes(int B, int N, int classes);      /* 0: unsupported shape */
int gfs3d_capl_head_forward_phase(const gfs3d_capl_head_args *args, int phase);
int gfs3d_capl_head_backward_phase(const gfs3d_capl_head_args *args, int phase);
/* The same head on the TESTING path (capl.forward(eval_model=True), capl.py:170-192, after getFeatures): prototype refinement
 * without the background row, refined base prototypes + gened_proto, get_pred against them, the geometric-word
 * re-weighting of get_gp_weight (capl.py:92-142: weight = th where coding[class][word of the point] == 1) and its two
 * accuracy figures.  logits (B, classes, N); stats[0] = mean coding[y][word] over all points, stats[1] = over the points of
 * novel classes (y >= base_num; 0 if none); label may be NULL (both figures 0).  classes <= 23, classes * Kw <= 16384. */
typedef struct gfs3d_capl_eval_args {
    int B, C, N;
    const float *feat;            /* (B, C, N) point_feat */
    long long feat_bstride;
    const float *main_proto;      /* (classes, C) */
    int classes;
    int base_num;
    const float *gened_proto;     /* (classes, C) */
    const int32_t *word;          /* (B, N) geometric-word index of every point */
    const float *coding;          /* (classes, Kw) multi-hot word coding of the classes */
    int Kw;
    float th;                     /* args.eval_weight */
    const int64_t *label;         /* (B, N) or NULL */
    float *logits;                /* out (B, classes, N) */
    float *stats;                 /* out: 2 floats */
    void *workspace;
    size_t workspace_bytes;       /* >= gfs3d_capl_head_workspace_bytes(B, N, classes) */
    void *stream;
} gfs3d_capl_eval_args;
int gfs3d_capl_head_eval(const gfs3d_capl_eval_args *args);
int gfs3d_kmeans_assign_f32(const float *pts, long long M, int C, const float *cent, int K,
                            int32_t *label, float *sums, float *counts, double *inertia, void *stream);
/* The same with a workspace (gfs3d_kmeans_workspace_bytes; 0 = unsupported shape): the centroid sums are then SEGMENT SUMS over
 * the points sorted by label (stable counting sort, fixed summation order: no float atomics, bit-reproducible; `sums` need
 * not be zeroed) - the points are read once as whole rows instead of going through a one-hot GEMM. */
size_t gfs3d_kmeans_workspace_bytes(long long M, int C, int K);
/* The Lloyd-loop form: the same labels, sums and counts at about half the matrix work.  A SCREEN scores every point with
 * hi + mid bf16 operands (three products instead of six) under a rigorous error bound E; a point whose two best scores
 * are more than 2 E apart has its arg-min decided, the few others are re-scored at fp32 accuracy from a list.  `inertia`
 * is summed from the screened scores (each within E of the exact one): use gfs3d_kmeans_assign_ws_f32 for a final, exact
 * inertia (get_basis.device_kmeans does).  Needs the workspace; C % 32 == 0 and 33 <= K <= 256 take the screened path,
 * other shapes run the exact form. */
int gfs3d_kmeans_assign_fast_f32(const float *pts, long long M, int C, const float *cent, int K,
                                 int32_t *label, float *sums, float *counts, double *inertia,
                                 void *workspace, size_t workspace_bytes, void *stream);
int gfs3d_kmeans_assign_ws_f32(const float *pts, long long M, int C, const float *cent, int K,
                               int32_t *label, float *sums, float *counts, double *inertia,
                               void *workspace, size_t workspace_bytes, void *stream);
/* Centroid sums that FOLLOW the labels (the Lloyd loop of get_basis.py:210 after its first iterations, when a few per cent of the
 * points move): gfs3d_kmeans_assign_fast_f32 with sums == NULL computes labels / counts / inertia only, and
 * gfs3d_kmeans_delta_fix moves every point whose label differs between `label_old` and `label_new` from its old centroid's sum
 * to the new one's, reading only those rows; `changed` (device, += ) counts them - what sklearn's strict-convergence test needs.
 * The running sums are 64-bit fixed-point integers (value * 2^shift; choose shift <= 62 - ceil(log2(M)) - ceil(log2(max |x|))
 * so that no sum overflows): integer atomics are associative, the sums do not depend on the order of arrival (bit-reproducible).
 * gfs3d_kmeans_fix_from_f32 / _to_f32 convert a full pass's fp32 sums and counts to and from that representation. */
int gfs3d_kmeans_fix_from_f32(const float *sums, const float *counts, int K, int C, int shift, long long *sums_fix,
                              long long *counts_fix, void *stream);
int gfs3d_kmeans_fix_to_f32(const long long *sums_fix, const long long *counts_fix, int K, int C, int shift, float *sums,
                            float *counts, void *stream);
int gfs3d_kmeans_delta_fix(const float *pts, long long M, int C, int K, const int32_t *label_old, const int32_t *label_new,
                           int shift, long long *sums_fix, long long *counts_fix, unsigned long long *changed, void *stream);
/* The tail of such an iteration in one launch (`_kmeans_single_lloyd`'s centre update and stopping quantities, get_basis.py:210):
 * sums / counts as fp32 (like gfs3d_kmeans_fix_to_f32), cent_new (K, C) = member means - an empty cluster keeps cent_old's row -
 * and state[3] = {sum over all elements of (cent_new - cent_old)^2, *changed, number of empty clusters} as floats: the caller's
 * single host read of the iteration; *changed is reset to 0 once read (the next gfs3d_kmeans_delta_fix counts from zero).
 * K C <= 2^22. */
int gfs3d_kmeans_fix_step(const long long *sums_fix, const long long *counts_fix, int K, int C, int shift, const float *cent_old,
                          unsigned long long *changed, float *sums, float *counts, float *cent_new, float *state, void *stream);
/* k-means++ seeding on the device: the `init='k-means++'` half of `KMeans(n_clusters=num_cnt, init='k-means++').fit(point_feat)`,
 * get_basis.py:210 of the reference (sklearn, not vendored; restated as the plain D^2 sampling of Arthur & Vassilvitskii 2007:
 * centre t + 1 is point i with probability d2[i] / sum d2, d2 = squared distance to the nearest of the centres so far).
 *   pts (M, C) row-major fp32, C % 4 == 0, C <= 1024;  first = index of the first centre (the caller's uniform draw);
 *   u  K - 1 uniform variates in [0, 1) on the DEVICE (drawn on the host, uploaded once): pick t + 1 is
 *      searchsorted(cumsum(d2), u[t] * sum(d2), side='right') - numpy's RandomState.choice(M, p = d2 / sum d2) on the same variate;
 *   centers (K, C) out: the picked rows;  picks (K) out: their indices.
 * 2 K - 1 launches on `stream`, no host synchronisation, fixed summation orders (the picks are a pure function of the
 * arguments).  prune != 0: a point whose nearest centre a has |c_new - c_a|^2 >= 4 d2 is not re-read (triangle
 * inequality: it cannot move) - same picks up to rounding-level ties, a fraction of the memory traffic. */
size_t gfs3d_kmeanspp_workspace_bytes(long long M, int K);
int gfs3d_kmeanspp_seed_f32(const float *pts, long long M, int C, int K, long long first, const double *u, float *centers,
                            long long *picks, void *workspace, size_t workspace_bytes, int prune, void *stream);
/* GREEDY k-means++: the seeding sklearn's `KMeans(init='k-means++')` really runs (get_basis.py:210 of the reference calls it
 * with the defaults; sklearn/cluster/_kmeans.py `_kmeans_plusplus`, scikit-learn 1.7.2): every pick draws
 * n_trials = 2 + int(log K) candidates by D^2 sampling and keeps the one that lowers the potential sum_i d2[i] most.
 *   pts (M, C) row-major fp32, C % 4 == 0, C <= 256;  first = index of the first centre (sklearn:
 *      `random_state.choice(n_samples, p=uniform)`, the caller's draw);
 *   u  (K - 1) x n_trials uniform variates on the DEVICE in sklearn's consumption order (`random_state.uniform(size=n_trials)`
 *      per pick): candidate j of pick t is searchsorted(cumsum(d2), u[t - 1][j] * sum(d2)) (side 'left', 0 when no mass is left);
 *   the winner is the first candidate with the smallest potential (numpy's argmin);  potentials are summed in fp64 in a fixed
 *   order (sklearn: a float32 BLAS dot), so the picks are a pure function of the arguments and equal sklearn's except where a
 *   variate lands within rounding of a cdf step or two candidates' potentials tie - sklearn's float32 potential is ~1e-7 of the
 *   potential away from an exact sum.  Measured against sklearn 1.7.2 (tests/golden/kmeanspp*.npz, basis_flow.npz): every pick
 *   equal on all fixtures up to 50 000 points (K = 200) and on 3 of 4 seeds at 300 000 points (K = 24); at configs[2] size (2.1e6
 *   points: the mean cdf step is 5e-7 of the potential) most sequences part somewhere - the same algorithm on the same
 *   variates, statistically equivalent, NOT seed-for-seed.
 *   centers (K, C) out;  picks (K) out.  4 K - 2 launches on `stream`, no host synchronisation.  prune as above (the gain
 *   pass skips a point when no candidate can come closer than its nearest centre). */
size_t gfs3d_kmeanspp_greedy_workspace_bytes(long long M, int C, int K);
int gfs3d_kmeanspp_greedy_f32(const float *pts, long long M, int C, int K, long long first, const double *u, int n_trials,
                              float *centers, long long *picks, void *workspace, size_t workspace_bytes, int prune, void *stream);
/* gfs3d_word_assign_f32 / gfs3d_proto_backward_feat_f32 with the matrix-core operand form chosen by the caller
 * (`compute` = GFS3D_COMPUTE_*; tensors stay fp32).  The reduced forms apply to the geometric-word shapes (more than 32
 * prototypes, C % 32 == 0); get_pred-sized calls (<= 32 prototypes) are HBM-bound on fp32 MFMAs and ignore it. */
int gfs3d_word_assign_mixed(const float *feat, long long feat_bstride, int B, int C, int N,
                            const float *words, int Kw, float scale,
                            float *soft, int32_t *assign, float *onehot, int compute, void *stream);
/* Softmax backward of the word assignment fed from the PACKED fusion gradient (model/capl.py:344-359: soft = softmax(10 cos) is
 * concatenated into the fusion layer's input):  d[b][w][n] = soft[b][w][n] (dsoft[b][n][w] - sum_w' soft[b][w'][n] dsoft[b][n][w']).
 *   dpacked  (B*N, ld) point-major rows whose first Kw columns are dsoft; dtype 0: fp32, GFS3D_COMPUTE_BF16: bf16 elements
 *            (gfs3d_pointwise_bwd_args.dx_type of a reduced-mode call)
 *   soft, d  (B, Kw, N) channel-first fp32, Kw <= 256
 * `d` is then the logits gradient of a plain cosine head: gfs3d_proto_backward_feat_ws(..., soft = NULL, dup = d) finishes the
 * word gradient reading it once (the softmax form of that kernel reads soft and dsoft twice each).  Fixed summation order. */
int gfs3d_word_soft_backward_rows(const void *dpacked, int dtype, int ld, const float *soft, int B, int Kw, int N, float *d, void *stream);

int gfs3d_proto_backward_feat_mixed(const float *feat, long long feat_bstride, int B, int C, int N,
                                    const float *proto, int P, int K, float scale,
                                    const float *soft, const float *dup, float *dfeat, int compute, void *stream);
/* The same with a caller workspace (gfs3d_proto_backward_feat_workspace_bytes(C, K); 0 = no such form): for SHARED words
 * (P == 1, 32 < K <= 256, C in {64, 128, 192}) the transposed bf16 piece image of the words and 1 / |g_w| are built ONCE per call
 * into the workspace and every workgroup copies tiles of it instead of transposing and splitting the words itself (smaller
 * workgroups, two per CU).  Other shapes run gfs3d_proto_backward_feat_mixed.  Same results up to fp32 summation order. */
size_t gfs3d_proto_backward_feat_workspace_bytes(int C, int K);
int gfs3d_proto_backward_feat_ws(const float *feat, long long feat_bstride, int B, int C, int N,
                                 const float *proto, int P, int K, float scale,
                                 const float *soft, const float *dup, float *dfeat, int compute,
                                 void *workspace, size_t workspace_bytes, void *stream);

/* ------------------------------------------------------------------ optimiser step --
 * torch.optim.Adam (train.py:425-435: parameter groups with their own learning rates, betas (0.9, 0.999), eps 1e-8, L2
 * weight_decay, no amsgrad) over up to GFS3D_ADAM_MAX_TENSORS parameter tensors in ONE launch:
 *     g += weight_decay p;  m += (g - m)(1 - beta1);  v = beta2 v + (1 - beta2) g g;
 *     p -= lr / bias_correction1 * m / (sqrt(v) / sqrt(bias_correction2) + eps)      bias_correction_i = 1 - beta_i^step
 * param / exp_avg / exp_avg_sq are updated in place; first_block is filled by the call (scratch). */
#define GFS3D_ADAM_MAX_TENSORS 64
typedef struct gfs3d_adam_args {
    int n_tensors;
    float *param[GFS3D_ADAM_MAX_TENSORS];
    const float *grad[GFS3D_ADAM_MAX_TENSORS];
    float *exp_avg[GFS3D_ADAM_MAX_TENSORS];
    float *exp_avg_sq[GFS3D_ADAM_MAX_TENSORS];
    long long numel[GFS3D_ADAM_MAX_TENSORS];
    float lr[GFS3D_ADAM_MAX_TENSORS];
    float weight_decay[GFS3D_ADAM_MAX_TENSORS];
    int first_block[GFS3D_ADAM_MAX_TENSORS];
    float beta1, beta2, eps, bias_correction1, bias_correction2;
    float one_minus_beta1, one_minus_beta2;   /* (float)(1.0 - (double)beta): 1.0f - 0.999f is 1.3e-5 off 0.001 */
} gfs3d_adam_args;
int gfs3d_adam_multi_f32(gfs3d_adam_args *a, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* GFS3D_H */
