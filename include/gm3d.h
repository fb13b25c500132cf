/*
 * gm3d.h -- C ABI of libgm3d_hip.so, the MI355X (gfx950) native back end of the
 * Point-MAE + GeoMask3D pretrain hot path.
 *
 * The reference's boundary for this path is Python-level torch ops in three
 * third-party packages; each entry point below names the
 * reference interface it sits beneath.  Conventions for every call:
 *   - plain device pointers + sizes, no torch types; the caller owns every buffer;
 *   - all tensors are dense row-major ("contiguous") in the stated shape;
 *   - work is enqueued on `stream` (a hipStream_t passed as void*; NULL = default
 *     stream) and the call returns without synchronising;
 *   - return value GM3D_OK or a negative GM3D_E* code, never throws, no global
 *     state, thread-compatible; argument/shape errors are detected on the host
 *     BEFORE any launch (a bad shape never reaches a kernel);
 *   - inputs are never written.
 */
#ifndef GM3D_H
#define GM3D_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void *gm3d_stream_t; /* hipStream_t */

enum {
    GM3D_OK = 0,
    GM3D_EINVAL = -1,       /* null pointer / non-positive size / inconsistent shape */
    GM3D_EUNSUPPORTED = -2, /* shape outside what the kernels are built for */
    GM3D_ELAUNCH = -3       /* hipLaunchKernel reported an error */
};

enum { GM3D_F32 = 0, GM3D_BF16 = 1 };

/* ABI version (bumped on any signature change) and error text. */
int gm3d_abi_version(void);
const char *gm3d_strerror(int code);

/* Farthest point sampling.
 * Replaces pointnet2_ops.pointnet2_utils.furthest_point_sample
 *   (Point-MAE_SA3D/models_mae_learn_loss.py:931, utils/miscc.py:18,
 *    engine_finetune.py:132, tools/runner_finetune.py:141).
 * xyz (B,N,3) f32 -> idx (B,npoint) int32; centers (B,npoint,3) f32 optional (NULL
 * to skip): the gather_operation of :932 fused into the same launch.
 * Rule: idx[0]=0, running min init 1e10, points with |p|^2<=1e-3 never selected,
 * argmax ties -> lowest index, d=((dx*dx+dy*dy)+dz*dz) fp32 without FMA.
 * Limits: 1 <= npoint, N <= 16384. */
int gm3d_fps(const float *xyz, int B, int N, int npoint, int32_t *idx, float *centers,
             gm3d_stream_t stream);

/* out[b,c,j] = feat[b,c,idx[b,j]].
 * Replaces pointnet2_utils.gather_operation forward (models_mae_learn_loss.py:932,
 * utils/miscc.py:19, engine_finetune.py:134).  feat (B,C,N) f32, idx (B,M) int32. */
int gm3d_gather_points(const float *feat, const int32_t *idx, int B, int C, int N, int M,
                       float *out, gm3d_stream_t stream);

/* Backward of gather_operation: grad_feat (B,C,N) is fully written (zero where no
 * index hits); duplicates accumulate in ascending j (deterministic). */
int gm3d_gather_points_grad(const float *grad_out, const int32_t *idx, int B, int C, int N, int M,
                            float *grad_feat, gm3d_stream_t stream);

/* Brute-force k-NN, ascending, ties -> lower reference index.
 * Replaces knn_cuda.KNN(k, transpose_mode=True)(ref, query)
 *   (models_mae_learn_loss.py:924,946; models/Point_MAE.py:55,68).
 * ref (B,N,3), query (B,G,3) f32 -> idx (B,G,k) int64, dist (B,G,k) f32 Euclidean
 * (sqrt applied) or NULL.  Limits: 1 <= k <= min(N,64), N <= 12288. */
int gm3d_knn(const float *ref, const float *query, int B, int N, int G, int k,
             float *dist, int64_t *idx, gm3d_stream_t stream);

/* Fused k-NN + neighbourhood gather + centre subtract = the body of Group.forward
 * after FPS (models_mae_learn_loss.py:946-957) in one launch.
 * Outputs: idx (B,G,k) int64 (may be NULL), neighborhood (B,G,k,3) = xyz[idx]-center,
 * neighborhood_org (B,G,k,3) = xyz[idx] (may be NULL). */
int gm3d_knn_group(const float *xyz, const float *center, int B, int N, int G, int k,
                   int64_t *idx, float *neighborhood, float *neighborhood_org,
                   gm3d_stream_t stream);

/* Chamfer nearest-neighbour squared-L2 distances, both directions, first minimum wins.
 * Replaces extensions.chamfer_dist forward (ChamferDistanceL2 at
 * models_mae_learn_loss.py:188,407; models/Point_MAE.py:392-394,426).
 * xyz1 (P,n,3), xyz2 (P,m,3) f32 -> dist1 (P,n), dist2 (P,m) f32, idx1 (P,n),
 * idx2 (P,m) int32.  n==m==32 takes the wave-per-patch path. */
int gm3d_chamfer_fwd(const float *xyz1, const float *xyz2, int P, int n, int m,
                     float *dist1, float *dist2, int32_t *idx1, int32_t *idx2,
                     gm3d_stream_t stream);

/* Backward of the above: gxyz1 (P,n,3), gxyz2 (P,m,3) fully written.
 * grad_dist1/2 may be NULL (treated as zeros).  n==m==32 is atomic-free and
 * deterministic; other shapes use float atomics like the upstream kernel. */
int gm3d_chamfer_bwd(const float *xyz1, const float *xyz2, const int32_t *idx1, const int32_t *idx2,
                     const float *grad_dist1, const float *grad_dist2, int P, int n, int m,
                     float *gxyz1, float *gxyz2, gm3d_stream_t stream);

/* Multi-head softmax attention core of timm-0.4.5 Attention.forward
 * (in-tree twin Point-MAE_SA3D/models/Point_MAE.py:113-125): for each (b,h)
 *   out[b,t,h,:] = softmax(scale * q k^T)[t,:] v
 * qkv is the (B,T,3,H,64) output of the qkv Linear, out is (B,T,H*64).
 * dtype GM3D_F32 (exact-f32 MFMA, parity mode) or GM3D_BF16 (bf16 MFMA, f32
 * softmax/accumulate).  lse (B,H,T) f32 = log-sum-exp of the scaled scores, saved
 * for backward (may be NULL for inference).  Limits: head_dim == 64, 1 <= T <= 128. */
int gm3d_attention_fwd(const void *qkv, void *out, float *lse, int B, int T, int H,
                       float scale, int dtype, gm3d_stream_t stream);

/* The qkv projection and the attention in one launch (bf16, T <= 64, C == 384 == H*64): replaces
 * `qkv = self.qkv(x)` + the attention above (Point-MAE_SA3D/models/Point_MAE.py:113-122) for one Block.
 * h (B*T, C) bf16 = the normalised tokens, wqkv (3*C, C) bf16 row-major (nn.Linear weight, no bias:
 * models_mae_learn_loss.py:907-911 qkv_bias=False).  out (B,T,C) bf16; lse (B,H,T) f32 or NULL;
 * qkv_out (B,T,3,H,64) bf16 or NULL (written for the backward; bit-identical to gm3d_gemm_tn_bf16_ring's
 * product, so out/lse are bit-identical to gm3d_attention_fwd on it).  GM3D_EUNSUPPORTED outside those limits. */
int gm3d_attention_qkv_fwd(const void *h, const void *wqkv, void *out, float *lse, void *qkv_out,
                           int B, int T, int H, int C, float scale, int dtype, gm3d_stream_t stream);

/* Backward: dqkv (B,T,3,H,64) fully written from dout (B,T,H*64), qkv, out, lse. */
int gm3d_attention_bwd(const void *qkv, const void *out, const void *dout, const float *lse,
                       void *dqkv, int B, int T, int H, float scale, int dtype,
                       gm3d_stream_t stream);

/* The pre-training loss on the masked patches in one pass (models_mae_learn_loss.py:384-412 forward_loss: target[mask] gather, fp32
 * cast, ChamferDistanceL2 per point (d1 + d2), mean over the 32 points -> `matrix`, mean of everything -> Chamfer_mean).
 * pred: the decoder head's output for the M masked tokens of each cloud, (B, M, 96) in `dtype` with batch stride pred_bstride
 * elements (a view of the (B, L, 96) output is fine); target (B, T, 32, 3) f32 = the neighbourhoods; ids (B, M) int64 = the masked
 * group ids (row stride ids_bstride).  Out: matrix (B, M) f32, idx1 / idx2 (B*M, 32) int32 (argmins, for the backward), mean_out (1)
 * f32.  Distances and argmins as gm3d_chamfer_fwd (bit-identical); the means are summed in a fixed order. */
int gm3d_patch_chamfer_loss_fwd(const void *pred, long long pred_bstride, const float *target, const long long *ids,
                                long long ids_bstride, int B, int T, int M, float *matrix, int32_t *idx1, int32_t *idx2,
                                float *mean_out, int dtype, gm3d_stream_t stream);
/* dpred (B, M, 96) contiguous in `dtype` = gmean[0] * d(mean_out)/d(pred). */
int gm3d_patch_chamfer_loss_bwd(const void *pred, long long pred_bstride, const float *target, const long long *ids,
                                long long ids_bstride, const int32_t *idx1, const int32_t *idx2, const float *gmean, int B, int T,
                                int M, void *dpred, int dtype, gm3d_stream_t stream);
/* ... as the gradient of the FULL prediction (B, lead + M, 96) whose last M patches entered the loss (outs['pix_pred'][:, -M:],
 * engine_pretrain.py:131): the first `lead` patches of every cloud receive zeros -- the slice's backward in the same launch. */
int gm3d_patch_chamfer_loss_bwd_full(const void *pred, long long pred_bstride, const float *target, const long long *ids,
                                     long long ids_bstride, const int32_t *idx1, const int32_t *idx2, const float *gmean, int B, int T,
                                     int M, int lead, void *dpred, int dtype, gm3d_stream_t stream);

/* Plain LayerNorm over (R, C) rows, 4 <= C <= 512, C % 4 == 0 (nn.LayerNorm of the 96 / 192 / 384-wide Point-M2AE levels):
 * h = (x - mean) * rstd * gamma + beta in `dtype`, mean / rstd (R) f32 saved for the backward.  Backward: dx in `dtype`;
 * partial (gm3d_ln_plain_partial_rows(R), 2, C) f32: [.][0] = per-block sums of dh * xhat (dgamma), [.][1] = of dh (dbeta), to be
 * finished by gm3d_colsum_finish over 2*C columns. */
int gm3d_ln_plain_partial_rows(int R);
/* measurement knob: workgroup cap (= rows of partial sums) of the LayerNorm backward kernels for 8192 <= R < 32768 (default 512) */
int gm3d_ln_set_grid_cap(int cap);
int gm3d_ln_plain_fwd(const void *x, const float *gamma, const float *beta, float eps, void *h, float *mean, float *rstd, int R,
                      int C, int dtype, gm3d_stream_t stream);
int gm3d_ln_plain_bwd(const void *dh, const void *x, const float *mean, const float *rstd, const float *gamma, void *dx,
                      float *partial, int R, int C, int dtype, gm3d_stream_t stream);
/* The same LayerNorm with the residual sum formed in the kernel (the pre-norm blocks of the hierarchical encoder / decoder, any
 * width 4 <= C <= 512, C % 4 == 0: Point-M2AE_SA3D/cfgs/config_Point_M2AE.yaml:57-99 dims 96 / 192 / 384):
 *   s = x + rowscale[row / rows_per_sample] * (y + ybias) + z   (y, ybias, rowscale, z optional; s_out required with y or z),
 *   h = LayerNorm(s as stored).   One pass instead of two or three elementwise launches + a LayerNorm.
 * gamma == NULL: the sum only (the tail of a block stack); gm3d_add_ln_bwd with dh == NULL is its backward (dx = gin, dy, and the
 * column sum of dy). */
int gm3d_add_ln_fwd(const void *x, const void *y, const float *ybias, const float *rowscale, int rows_per_sample, const void *z,
                    const float *gamma, const float *beta, float eps, void *s_out, void *h, float *mean, float *rstd, int R, int C,
                    int dtype, gm3d_stream_t stream);
/* Its backward: dx = gin + LayerNorm-backward(dh) (gin optional: the gradient arriving at s from the next residual sum), dy =
 * rowscale * dx (optional), partial (gm3d_ln_plain_partial_rows(R), nsum, C): [0] dgamma, [1] dbeta and, with nsum == 3, [2]
 * sum_rows dy = the gradient of ybias. */
int gm3d_add_ln_bwd(const void *dh, const void *gin, const void *x, const float *mean, const float *rstd, const float *gamma,
                    const float *rowscale, int rows_per_sample, void *dx, void *dy, float *partial, int nsum, int R, int C, int dtype,
                    gm3d_stream_t stream);
/* ... with a running sum over the sites that share an addend (the positional embedding re-added in front of every block of a
 * stack): acc (R,C) `dtype`, acc_mode 0 none / 1 acc = dx / 2 acc += dx.  dh == NULL (no LayerNorm behind the sum): x, mean, rstd,
 * gamma are not read and may be NULL. */
int gm3d_add_ln_bwd_acc(const void *dh, const void *gin, const void *x, const float *mean, const float *rstd, const float *gamma,
                        const float *rowscale, int rows_per_sample, void *dx, void *dy, float *partial, int nsum, void *acc,
                        int acc_mode, int R, int C, int dtype, gm3d_stream_t stream);

/* ---- Row-wise fused passes around the transformer-block GEMMs (gm3d_amd/csrc/rowops.hip) ------------
 * Together they restate timm-0.4.5 Block.forward (in-tree twin Point-MAE_SA3D/models/Point_MAE.py:128-146)
 * as driven by TransformerEncoder/Decoder.forward (models_mae_learn_loss.py:914-917,984-990).
 * The residual stream is fp32; `dtype` (GM3D_F32 / GM3D_BF16) is the type of the GEMM-side tensors.
 * C must be 384 (trans_dim) for the LayerNorm entry points. */

/* out_res[r,:] = res[r,:] + rowscale[r / rows_per_sample] * (y[r,:] + bias) + add[r,:]
 * h[r,:]       = LayerNorm(out_res[r,:]) * gamma + beta          (eps as given; mean/rstd saved per row)
 * res/out_res f32 (R,C); y, add, h `dtype` (R,C); bias, gamma, beta f32 (C); rowscale f32 (R/rows_per_sample)
 * = the DropPath keep/(1-p) factor per sample.  y, bias, rowscale, add, out_res, res may be NULL (at
 * least one of res/y/add must be given). */
int gm3d_residual_ln_fwd(const float *res, const void *y, const float *bias, const float *rowscale,
                         int rows_per_sample, const void *add, const float *gamma, const float *beta,
                         float eps, float *out_res, void *h, float *mean, float *rstd, int R, int C,
                         int dtype, gm3d_stream_t stream);

/* Backward of one such site.  dh (R,C) `dtype` = grad of h; gin (R,C) f32 = grad already flowing on the
 * residual stream (NULL = none); x = the saved out_res.  Writes dx (R,C) f32 = grad wrt out_res,
 * dy (R,C) `dtype` = rowscale * dx (NULL to skip), acc (R,C) f32 += dx (NULL to skip; positional grad), acc_out (R,C) `dtype` =
 * the updated acc in the GEMM-side type as well (NULL to skip; needs acc), and per-workgroup column partial sums
 * partial[gm3d_ln_partial_rows(R)][3][C] f32: [0] dgamma, [1] dbeta, [2] colsum(dy) (= gradient of `bias`). */
int gm3d_residual_ln_bwd(const void *dh, const float *gin, const float *x, const float *mean,
                         const float *rstd, const float *gamma, const float *rowscale, int rows_per_sample,
                         float *dx, void *dy, float *acc, void *acc_out, float *partial, int R, int C, int dtype,
                         gm3d_stream_t stream);
int gm3d_ln_partial_rows(int R);   /* rows of `partial` the call above writes */

/* out[c] (+)= sum_{r<nrows} partial[r*pitch + c], c < ncols (second stage of the column sums). */
int gm3d_colsum_finish(const float *partial, int nrows, int pitch, int ncols, float *out, int accumulate,
                       gm3d_stream_t stream);

/* Batched form: job j sums partial[j*job_stride + r*pitch + c] over r < nrows into out[j*out_stride + c]. */
int gm3d_colsum_finish_batched(const float *partial, int njobs, long long job_stride, int nrows, int pitch, int ncols,
                               float *out, int out_stride, gm3d_stream_t stream);

/* g = GELU(f + bias), exact erf form (nn.GELU, Mlp at models/Point_MAE.py:82-98).  C % 8 == 0. */
int gm3d_bias_gelu_fwd(const void *f, const float *bias, void *g, int R, int C, int dtype,
                       gm3d_stream_t stream);
/* df = dg * GELU'(f + bias); partial[gm3d_gelu_partial_rows(R)][C] f32 = column partial sums of df. */
int gm3d_bias_gelu_bwd(const void *dg, const void *f, const float *bias, void *df, float *partial, int R,
                       int C, int dtype, gm3d_stream_t stream);
int gm3d_gelu_partial_rows(int R);

/* ---- Mini-PointNet token embed: streaming passes between its three wide GEMMs (gm3d_amd/csrc/embed.hip) ----
 * Together with the GEMMs they restate Encoder.forward (Point-MAE_SA3D/models_mae_learn_loss.py:868-899) and its
 * backward.  Activations are (G groups, K points, C channels) row-major in `dtype`; C % 8 == 0, C <= 1024, K <= 255.
 * Kernels that reduce over rows write per-workgroup partial rows; gm3d_embed_partial_rows(kind, n, C) gives the
 * number of rows (kind 0: gm3d_moments3, n = rows; kind 1: (G,K,C) kernels, n = G; kind 2: (R,C) kernels, n = R; kind 3: gm3d_pn_layer1_bwd_stats, n = R),
 * and gm3d_colsum_finish adds them up. */
int gm3d_embed_partial_rows(int kind, int n, int C);

/* partial[row][0..2] = sum x_j, [3..8] = sum xx,xy,xz,yy,yz,zz (fp64) of the (R,3) f32 input: BatchNorm statistics of the
 * K=3 layer are analytic in these moments (first_conv.0/.1, :873-874). */
int gm3d_moments3(const float *x, int R, double *partial, gm3d_stream_t stream);

/* a1 (R,C1) = relu(x . wf^T + bf): Conv1d(3,128) + BatchNorm (folded into wf (C1,3), bf (C1)) + ReLU (:873-875). */
int gm3d_pn_layer1_fwd(const float *x, const float *wf, const float *bf, void *a1, int R, int C1, int dtype,
                       gm3d_stream_t stream);

/* out (G,C) = max over K of in (G,K,C) (+ bias); arg (G,C) uint8 = first maximising k (torch.max(dim)[0], :895,898). */
int gm3d_group_max_fwd(const void *in, const float *bias, void *out, uint8_t *arg, int G, int K, int C, int dtype,
                       gm3d_stream_t stream);
/* din (G,K,C) = dout (G,C) at k == arg, 0 elsewhere (dense; fully written). */
int gm3d_group_max_bwd(const void *dout, const uint8_t *arg, void *din, int G, int K, int C, int dtype,
                       gm3d_stream_t stream);

/* BatchNorm over y = y0 (G,K,C) + t (G,C)[group] (second_conv.0/.1 with the concat folded, :896-897,879-880):
 * partial[row][0][c] = sum y, [1][c] = sum y^2. */
int gm3d_bn_bcast_stats(const void *y0, const void *t, int G, int K, int C, float *partial, int dtype,
                        gm3d_stream_t stream);
/* a2 = act(y * scale + shift), act(h) = h > 0 ? h : slope*h  (slope 0: ReLU of the embed; 0.2: LeakyReLU of the
 * loss-predictor head increase_dim_2, :152-158). */
int gm3d_bn_bcast_apply_relu(const void *y0, const void *t, const float *scale, const float *shift, void *a2,
                             int G, int K, int C, float slope, int dtype, gm3d_stream_t stream);
/* backward pass 1: g = da2 * [y*scale+shift > 0]; partial[row][0][c] = sum g, [1][c] = sum g * (y-mean)*rstd. */
int gm3d_bn_bcast_bwd_stats(const void *da2, const void *y0, const void *t, const float *scale, const float *shift,
                            const float *mean, const float *rstd, int G, int K, int C, float *partial, float slope,
                            int dtype, gm3d_stream_t stream);
/* backward pass 2: dy (G,K,C) = scale * (g - s1/R - yhat * s2/R), dt (G,C) f32 = sum_k dy. */
int gm3d_bn_bcast_bwd_apply(const void *da2, const void *y0, const void *t, const float *scale, const float *shift,
                            const float *mean, const float *rstd, const float *s1, const float *s2, void *dy,
                            float *dt, int G, int K, int C, float slope, int dtype, gm3d_stream_t stream);

/* Tail of the loss-predictor head increase_dim_2 (models_mae_learn_loss.py:152-158) + `.mean(-1)` (:677): Conv1d(C -> nout) followed by
 * the mean over its nout outputs is one C-vector.  gm3d_head_fold: wv[c] = mean_o W1[o][c] (f32, and a copy in `dtype`), bm = mean(b1).
 * gm3d_head_rowdot: out[r] = round_dtype(a[r,:] . wv) + bm.  Backward: gm3d_head_fold_bwd: dW1[o][c] = dwv[c]/nout,
 * db1[o] = sum_r d[r]/nout;  gm3d_head_outer: da[r][c] = d[r] * wv[c] in `dtype`. */
int gm3d_head_fold(const float *W1, const float *b1, int nout, int C, float *wv, void *wv_t, float *bm, int dtype,
                   gm3d_stream_t stream);
int gm3d_head_rowdot(const void *a, const void *wv_t, const float *bm, int R, int C, float *out, int dtype, gm3d_stream_t stream);
int gm3d_head_fold_bwd(const float *dwv, const float *d, int R, int nout, int C, float *dW1, float *db1, gm3d_stream_t stream);
int gm3d_head_outer(const float *d, const float *wv, int R, int C, void *da, int dtype, gm3d_stream_t stream);

/* Training augmentation PointcloudScaleAndTranslate.__call__ (Point-MAE_SA3D/datasets/data_transforms.py:27-35, called at
 * engine_pretrain.py:78), in place on pc (B,N,3) f32: u (2,B,3) f32 uniform draws in [0,1) -> per-cloud scale = u[0]*span+lo (span = hi-lo),
 * shift = (u[1]*2-1)*t, p = p*scale + shift. */
int gm3d_scale_translate(float *pc, const float *u, float lo, float span, float t, int B, int N, gm3d_stream_t stream);

/* The same three passes when only a SUBSET of the groups goes on to the next layer -- the student's visible tokens: x_vis =
 * tokens[~mask] (models_mae_learn_loss.py:298) discards 39 of 64 groups per cloud after the last conv + max-pool, so that conv
 * (and its backward) only needs the visible groups' rows, while the batch statistics still cover every row.
 * gm3d_group_select_maps: ids (B,V) int64 (row pitch id_pitch) -> sel (B*V) = flat source group of every selected group and
 * inv (B*G) = position in sel, or -1.  _apply_relu_sel: a2 holds the G selected groups (compact), read from groups sel[g] of
 * y0 / t.  _bwd_stats_sel: da2 holds the G selected groups (the others' gradient is zero and contributes nothing to the sums).
 * _bwd_apply_sel: all G groups are written; da2 of group g is row block inv[g] of the compact buffer, zero when inv[g] < 0.
 * sel / inv NULL = the plain forms above. */
int gm3d_group_select_maps(const long long *ids, int id_pitch, int B, int V, int G, int *sel, int *inv, gm3d_stream_t stream);
int gm3d_bn_bcast_apply_relu_sel(const void *y0, const void *t, const float *scale, const float *shift, void *a2,
                                 const int *sel, int G, int K, int C, float slope, int dtype, gm3d_stream_t stream);
int gm3d_bn_bcast_bwd_stats_sel(const void *da2, const void *y0, const void *t, const float *scale, const float *shift,
                                const float *mean, const float *rstd, const int *sel, int G, int K, int C, float *partial,
                                float slope, int dtype, gm3d_stream_t stream);
int gm3d_bn_bcast_bwd_apply_sel(const void *da2, const void *y0, const void *t, const float *scale, const float *shift,
                                const float *mean, const float *rstd, const float *s1, const float *s2, void *dy, float *dt,
                                const int *inv, int G, int K, int C, float slope, int dtype, gm3d_stream_t stream);

/* df (G,K,C) += dfg (G,C) at k == arg (in place); partial[row][c] = column sums of the result. */
int gm3d_group_scatter_add(void *df, const void *dfg, const uint8_t *arg, int G, int K, int C, float *partial,
                           int dtype, gm3d_stream_t stream);

/* Backward reductions of Conv1d(3,C1)+BatchNorm+ReLU: with g1 = da1*[a1>0], hhat = (x.w1+b1-mean)*rstd,
 * xc = x - xmean (xmean (3) f32): partial[row][q][c] in fp64 (the weight gradient is a small difference of these
 * sums), q = 0: sum g1, 1: sum g1*hhat, 2..4: sum g1*xc_j; rows = gm3d_embed_partial_rows(3, R, C1); C1 <= 256. */
int gm3d_pn_layer1_bwd_stats(const void *da1, const void *a1, const float *x, const float *w1, const float *b1,
                             const float *mean, const float *rstd, const float *xmean, int R, int C1,
                             double *partial, int dtype, gm3d_stream_t stream);
/* fp64 second stage: out[c] = sum_r partial[r*pitch + c]. */
int gm3d_colsum_finish_f64(const double *partial, int nrows, int pitch, int ncols, double *out,
                           gm3d_stream_t stream);

/* partial[row][c] = column sums of a (R,C) matrix in `dtype`; rows = gm3d_embed_partial_rows(2, R, C); C % 8 == 0, 8 <= C <= 2048. */
int gm3d_colsum_partial(const void *m, int R, int C, float *partial, int dtype, gm3d_stream_t stream);
/* The same with a per-row weight roww (R) f32 (NULL = 1): partial sums of roww[r] * m[r][c].  Replaces the library GEMV a^T d in
 * the backward of the loss-prediction head (increase_dim_2 + mean(-1), models_mae_learn_loss.py:152-158,677): fixed summation
 * order, replay-safe. */
int gm3d_colsum_partial_w(const void *m, const float *roww, int R, int C, float *partial, int dtype, gm3d_stream_t stream);

/* out (R,C) = GELU(x (R,3) . w (C,3)^T + b): first layer + activation of pos_embed (models_mae_learn_loss.py:104-108). */
int gm3d_lin3_gelu_fwd(const float *x, const float *w, const float *b, void *out, int R, int C, int dtype,
                       gm3d_stream_t stream);
/* its backward reductions: dpre = dout * GELU'(pre); partial[row][q][c] fp64, q = 0: sum dpre, 1..3: sum dpre * x_j;
 * rows = gm3d_embed_partial_rows(3, R, C); C <= 512. */
int gm3d_lin3_gelu_bwd(const void *dout, const float *x, const float *w, const float *b, int R, int C,
                       double *partial, int dtype, gm3d_stream_t stream);
/* The finish of its partial sums in the layer's own layouts: db (C) f32 = the column sums of block 0, dW (C,3) f32 column k = block 1 + k
 * (partial (nrows, 4 C) f64, the buffer gm3d_lin3_gelu_bwd filled). */
int gm3d_lin3_finish(const double *partial, int nrows, int C, float *dW, float *db, gm3d_stream_t stream);

/* Pairwise ranking loss of forward_learning_loss(relative=True) (models_mae_learn_loss.py:795-805), per sample:
 * out (B,2) f32 = [sum of pair terms, number of ordered pairs]; dpred (B,M) = d(sum of pair terms)/d pred.  M <= 64. */
int gm3d_rank_loss(const float *pred, const float *target, int B, int M, float *out, float *dpred,
                   gm3d_stream_t stream);
/* The whole loss_learn of engine_pretrain.py:156-171 on the last M columns of a (B, ldp)-pitched prediction (pred points at the first
 * of them): gm3d_rank_loss plus tot (2) = column sums of out in a fixed order and loss (1) = tot[0] / tot[1] -- two launches, no
 * PyTorch reduction.  _bwd: dfull (B,L) = [zeros (L - M) | dpred * (g[0] / tot[1])], the gradient of the full (B,L) prediction. */
int gm3d_rank_loss_tail(const float *pred, int ldp, const float *target, int B, int M, float *out, float *dpred, float *tot, float *loss,
                        gm3d_stream_t stream);
int gm3d_rank_loss_tail_bwd(const float *dpred, const float *g, const float *tot, int B, int M, int L, float *dfull, gm3d_stream_t stream);
/* ModelEma.update on the integer buffers (BatchNorm num_batches_tracked, P/engine_pretrain.py:36-52 applies the same blend to every state_dict
 * entry): e[i] = (int64)(float(e[i]) * decay + w * float(m[i])), w = 1 - decay formed by the caller (in double, like the Python expression). */
int gm3d_ema_counters(long long *e, const long long *m, int n, float decay, float w, gm3d_stream_t stream);
/* DropPath factors out (S,B) = floor(keep[s] + u[s][b]) / keep[s] from one uniform draw u (timm DropPath, P/models/Point_MAE.py:9). */
int gm3d_drop_path_scales(const float *u, const float *keep, int S, int B, float *out, gm3d_stream_t stream);
/* dst (R,Np) = [src (R,N), row pitch lds | zeros]; N, Np, lds multiples of 8 (a narrow operand padded to a GEMM tile width). */
int gm3d_pad_cols(const void *src, long long lds, int R, int N, void *dst, int Np, int dtype, gm3d_stream_t stream);

/* ---- Optimizer step on flat buffers (gm3d_amd/csrc/optim.hip) ------------------------------------------------------
 * clip_grad_norm_(max_norm) + torch.optim.AdamW + timm ModelEma.update + bf16 shadows, replacing
 * NativeScalerWithGradNormCount.__call__ (Point-MAE_SA3D/util/misc.py:256-270), the AdamW of tools/builder.py:40-56 and
 * model_ema.update (engine_pretrain.py:212).  All buffers hold n fp32 elements (n % 4 == 0) in one common layout whose
 * first n_decay elements (n_decay % 4 == 0) take weight decay.  lr_dev, ema_w_dev (= 1 - decay) and step_dev (the step
 * count, incremented here) are single floats in device memory.  ema / shadow_p / shadow_e (bf16) may be NULL.
 * partial: gm3d_flat_partial_rows(n) floats of scratch; scal (4 floats) receives {clip coefficient, 1-beta1^t,
 * 1-beta2^t, gradient norm before clipping}.  max_norm <= 0 disables clipping. */
int gm3d_flat_partial_rows(long long n);
int gm3d_adamw_ema_flat_step(float *params, const float *grads, float *exp_avg, float *exp_avg_sq, float *ema,
                             void *shadow_p, void *shadow_e, long long n, long long n_decay, const float *lr_dev,
                             float weight_decay, float beta1, float beta2, float eps, const float *ema_w_dev,
                             float max_norm, float *step_dev, float *partial, float *scal, gm3d_stream_t stream);
/* The same step with layer-wise learning-rate decay (fine-tuning: Point-MAE_SA3D/util/lr_decay.py:15-62 builds one AdamW
 * group per layer with lr = base_lr * lr_scale): lr_scale holds one multiplier per element, in the flat layout (NULL = 1). */
int gm3d_adamw_ema_flat_step_lrd(float *params, const float *grads, float *exp_avg, float *exp_avg_sq, float *ema,
                                 void *shadow_p, void *shadow_e, long long n, long long n_decay, const float *lr_dev,
                                 const float *lr_scale, float weight_decay, float beta1, float beta2, float eps,
                                 const float *ema_w_dev, float max_norm, float *step_dev, float *partial, float *scal,
                                 gm3d_stream_t stream);

/* BatchNorm statistic finalisation in one launch (train: from sums = [sum(C), sumsq(C)] over `rows` rows, with
 * nn.BatchNorm1d's running-statistic update -- momentum, unbiased variance, num_batches_tracked += 1; eval: from the
 * running buffers): scale = gamma*rstd, shift = beta - mean*scale, plus mean / rstd for the backward. */
int gm3d_bn_finalize(const float *sums, double rows, const float *gamma, const float *beta, float eps, float momentum,
                     float *running_mean, float *running_var, long long *nbt, float *scale, float *shift,
                     float *mean_out, float *rstd_out, int C, int training, gm3d_stream_t stream);
/* Same for the K=3 first layer, whose statistics are analytic in the input moments mom9 (gm3d_moments3, summed):
 * emits the BatchNorm-folded conv wf (C,3) / bf (C), mean / rstd, and mcov = [mean(3), cov(3x3)] fp64 and
 * xmean (3) f32 for the backward. */
int gm3d_pn1_finalize(const double *mom9, double rows, const float *w, const float *b, const float *gamma,
                      const float *beta, float eps, float momentum, float *running_mean, float *running_var,
                      long long *nbt, float *wf, float *bf, float *mean_out, float *rstd_out, double *mcov,
                      float *xmean, int C, int training, gm3d_stream_t stream);
/* Layer-1 backward tail: q (5,C) fp64 = summed gm3d_pn_layer1_bwd_stats partials, mcov from gm3d_pn1_finalize ->
 * dw (C,3), dgamma (C), dbeta (C). */
int gm3d_pn1_bwd_finalize(const double *q, const double *mcov, const float *w, const float *gamma, const float *rstd,
                          float *dw, float *dgamma, float *dbeta, int C, gm3d_stream_t stream);
/* Teacher-guided mask (models_mae_learn_loss.py:744-784 generate_mask) and the visible / masked token id lists the
 * boolean-mask indexing of :298-299,649-650 produces, in one launch.  loss_pred, noise (B,L) f32; the len_loss tokens of
 * highest loss_pred are always masked, the others are ranked by noise ((value, index) order) and the first len_keep stay
 * visible.  mask (B,L) f32: 0 keep / 1 remove; vis_ids (B,len_keep), mask_ids (B,L-len_keep) int64 ascending.  L <= 64.
 * id_pitch = 0: the two id arrays are dense; id_pitch >= L: their row pitch in elements (vis_ids = order, mask_ids = order + len_keep
 * of one (B,id_pitch) buffer gives the [visible | masked] permutation gm3d_token_assemble_* consume). */
int gm3d_mask_select(const float *loss_pred, const float *noise, int B, int L, int len_keep, int len_loss, float *mask,
                     long long *vis_ids, long long *mask_ids, int id_pitch, gm3d_stream_t stream);
/* ... with the mask also as bytes (mask_bool (B,L) 0 / 1 or NULL): bool_masked_pos of engine_pretrain.py:100 without a conversion pass. */
int gm3d_mask_select_b(const float *loss_pred, const float *noise, int B, int L, int len_keep, int len_loss, float *mask,
                       unsigned char *mask_bool, long long *vis_ids, long long *mask_ids, int id_pitch, gm3d_stream_t stream);
/* C (M,N) bf16 = A (M,K) bf16 . W (N,K)^T bf16 (+ bias (N) f32), fp32 accumulation: nn.Linear / Conv1d(k=1) forward with W the
 * weight as stored, input gradient with W the transposed weight.  Row pitches lda / ldw / ldc in elements (multiples of 8).
 * Limits: N % 128 == 0, K % 64 == 0. */
int gm3d_gemm_tn_bf16(const void *A, const void *W, const float *bias, void *C, int M, int N, int K, int lda, int ldw,
                      int ldc, gm3d_stream_t stream);
/* fc1 + bias + exact-erf GELU in the GEMM epilogue (timm Mlp: fc1 -> GELU, Point_MAE.py:92-94):
 * F (M,N) bf16 = A . W^T (no bias; optional, NULL when no backward will follow), G (M,N) bf16 = GELU(F + bias).
 * Same operand rules and limits as gm3d_gemm_tn_bf16. */
int gm3d_gemm_tn_bf16_gelu(const void *A, const void *W, const float *bias, void *F, void *G, int M, int N, int K, int lda,
                           int ldw, int ldf, int ldg, gm3d_stream_t stream);
/* Conv1d(k=1) + max over the 32 points of each group in the GEMM epilogue (mini-PointNet, models_mae_learn_loss.py:891-897):
 * rows are (group, point) with 32 points per group (M % 32 == 0).  P (M/32,N) bf16 = max_k, arg (M/32,N) u8 = its first argmax.
 * bias_after_pool = 0: C (optional) = A.W^T + bias is written and pooled (conv2: the rows feed the next layer);
 * bias_after_pool = 1: the product is pooled and the bias added to the maximum (conv4: C may be NULL -- the (M,N) product is
 * never needed again, not even by the backward, which works from `arg`). */
int gm3d_gemm_tn_bf16_pool(const void *A, const void *W, const float *bias, void *C, void *P, uint8_t *arg, int M, int N, int K,
                           int lda, int ldw, int ldc, int ldp, int bias_after_pool, gm3d_stream_t stream);
/* fc2 input gradient + GELU backward in the GEMM epilogue (timm Mlp backward): dF (M,N) bf16 = (dO (M,K) . Wt (N,K)^T) *
 * GELU'(F + bias) with Wt the TRANSPOSED fc2 weight (N = hidden, K = model width), F the bf16 pre-activation saved by
 * gm3d_gemm_tn_bf16_gelu; colpart (gm3d_gemm_tile_rows(M), N) f32 receives per-row-tile column sums of dF (the fc1 bias
 * gradient after gm3d_colsum_finish). */
int gm3d_gemm_tn_bf16_gelu_bwd(const void *dO, const void *Wt, const void *F, const float *bias, void *dF, float *colpart, int M,
                               int N, int K, int lda, int ldw, int ldf, int lddf, gm3d_stream_t stream);
/* LayerNorm folded into the GEMMs around it (timm Block norm1 / norm2 + the residual adds, Point-MAE_SA3D/models/Point_MAE.py:128-146):
 *   gm3d_gemm_tn_bf16_res   proj / fc2 (N = 384) whose epilogue writes the fp32 residual stream
 *                           U = res + rowscale[row / rows_per_sample] * (bf16(A.W^T) + bias) + add   (rowscale, add may be NULL)
 *                           its bf16 copy U16, and stats (3, M, 2) f32 = per row and 128-column tile (mean, sum of squared
 *                           deviations) of the fp32 values;
 *   gm3d_gemm_tn_bf16_lna   qkv / fc1 (K = 384) that reads U16 and stats, normalises (gamma, beta, eps) while staging its A operand and
 *                           computes C = LN(U).W^T + bias (G == NULL) or F (optional) / G = GELU(..) like gm3d_gemm_tn_bf16_gelu;
 *                           H (M,384) bf16, mean / rstd (M) f32 (all optional): the normalised rows and the row statistics, for the
 *                           backward pass.
 * Together they replace GEMM -> gm3d_residual_ln_fwd -> GEMM in the bf16 (throughput) mode: the statistics are those of the fp32
 * stream; the value that is normalised is the stream rounded to bf16 (an error of at most one bf16 half-ulp of |u| / sigma per
 * element, i.e. of the size of the rounding of the normalised row itself).  The fp32 parity mode keeps the three-kernel form. */
int gm3d_gemm_tn_bf16_res(const void *A, const void *W, const float *bias, const float *res, const float *rowscale, int rows_per_sample,
                          const void *add, float *U, void *U16, float *stats, int M, int N, int K, int lda, int ldw, int bm,
                          gm3d_stream_t stream);
int gm3d_gemm_tn_bf16_lna(const void *U16, const float *stats, const float *gamma, const float *beta, float eps, const void *W,
                          const float *bias, void *C, void *G, void *H, float *mean, float *rstd, int M, int N, int K, int ldu, int ldw,
                          int ldc, int ldg, gm3d_stream_t stream);
int gm3d_gemm_tile_rows(int M);
/* The same product as gm3d_gemm_tn_bf16 (bit-identical results) through a four-stage LDS-DMA ring: for long K over few output
 * tiles (fc2, the fc1 / qkv input gradients: N = 384, K = 1152 .. 1536), where one workgroup per CU has to keep more loads in
 * flight than the register-prefetch kernel does.  bm = 64 or 128: tile height.  N need only be a multiple of 8 here (a ragged
 * last column tile: the 96-wide reconstruction head increase_dim_just_network_without_feature, models_mae_learn_loss.py:169-176). */
int gm3d_gemm_tn_bf16_ring(const void *A, const void *W, const float *bias, void *C, int M, int N, int K, int lda, int ldw, int ldc,
                           int bm, gm3d_stream_t stream);
/* Ring depth of gm3d_gemm_tn_bf16_ring: K-stages requested ahead of the one being multiplied (2 .. 4 for bm = 64, 2 .. 3 for
 * bm = 128; clamped).  Results do not depend on it.  Process-wide; set before capturing a graph. */
int gm3d_gemm_ring_set_depth(int bm, int depth);
/* The TALL products of the mini-PointNet (Encoder.first_conv.3 / second_conv.0 / second_conv.3 over the B*G*k = 262,144 point rows,
 * models_mae_learn_loss.py:876-882, and their input gradients; K <= 512) with the weights STATIONARY IN REGISTERS (csrc/gemm_ws.hip):
 * persistent workgroups, 32 output columns of W per wave for the whole K, A streamed once per column block through an LDS ring
 * filled by loader waves.  HBM-bound by design (A in + C out).  Instantiated for (K -> N) = 256 -> 512, 512 -> 256, 384 -> 512,
 * 256 -> 128, 128 -> 256, 512 -> 384 (gm3d_gemm_ws_supported); results bit-identical to gm3d_gemm_tn_bf16. */
int gm3d_gemm_tn_bf16_ws(const void *A, const void *W, const float *bias, void *C, int M, int N, int K, int lda, int ldw, int ldc,
                         gm3d_stream_t stream);
/* ... with the max-pool epilogue of gm3d_gemm_tn_bf16_pool (one 32-row tile = one group of 32 points): 128 -> 256 and 512 -> 384. */
int gm3d_gemm_tn_bf16_ws_pool(const void *A, const void *W, const float *bias, void *C, void *P, unsigned char *arg, int M, int N, int K,
                              int lda, int ldw, int ldc, int ldp, int bias_after_pool, gm3d_stream_t stream);
/* ... for groups of group_rows = 16 or 32 rows (Point-M2AE's level-0 groups hold 16 points, Point-M2AE_SA3D/cfgs/config_Point_M2AE.yaml:60:
 * a 32-row tile is then two groups): P / arg (M / group_rows, N), arg counts within its group.  Also 512 -> 96 (that level's token width).
 * The weight-stationary kernel also takes ragged K (multiples of 8: 96, 192, 288 against N = 96 / 192 / 288 / 384 / 512) for the
 * hierarchical model's tall products; gm3d_gemm_ws_supported lists every (N, K). */
int gm3d_gemm_tn_bf16_ws_poolg(const void *A, const void *W, const float *bias, void *C, void *P, unsigned char *arg, int M, int N, int K,
                               int lda, int ldw, int ldc, int ldp, int bias_after_pool, int group_rows, gm3d_stream_t stream);
int gm3d_gemm_ws_supported(int N, int K, int pool);
/* second_conv.0 (256 -> 512 on the local half of the split concat, models_mae_learn_loss.py:880) together with the BatchNorm1d(512)
 * + ReLU behind it (:881), T (M / 32, N) bf16 being the per-group term (global feature @ W_g^T + bias), M % 32 == 0:
 *   _bn_apply  eval-mode BatchNorm (the EMA teacher): C = act((bf16(A.W^T) + T[row / 32]) * scale + shift), act(h) = h > 0 ? h : slope h
 *              -- bit-identical to gm3d_gemm_tn_bf16_ws followed by gm3d_bn_bcast_apply_relu, the product never reaches HBM;
 *   _bn_stats  train mode: C = bf16(A.W^T) plus per-workgroup column sums of y = C + T[row / 32] and y^2 into partial
 *              (gm3d_gemm_ws_stats_rows(M, N, K) rows x 2 N f32, every row written): what gm3d_bn_bcast_stats would read C again for.
 *              Sum the rows in order (gm3d_colsum_finish) -> the input of gm3d_bn_finalize.  Deterministic. */
int gm3d_gemm_tn_bf16_ws_bn_apply(const void *A, const void *W, const void *T, const float *scale, const float *shift, float slope, void *C,
                                  int M, int N, int K, int lda, int ldw, int ldt, int ldc, gm3d_stream_t stream);
int gm3d_gemm_tn_bf16_ws_bn_stats(const void *A, const void *W, const void *T, void *C, float *partial, int M, int N, int K, int lda,
                                  int ldw, int ldt, int ldc, gm3d_stream_t stream);
/* ... for groups of group_rows = 16 or 32 rows (T (M / group_rows, N)): the hierarchical model's level-0 groups hold 16 points */
int gm3d_gemm_tn_bf16_ws_bn_apply_g(const void *A, const void *W, const void *T, const float *scale, const float *shift, float slope, void *C,
                                    int M, int N, int K, int lda, int ldw, int ldt, int ldc, int group_rows, gm3d_stream_t stream);
int gm3d_gemm_tn_bf16_ws_bn_stats_g(const void *A, const void *W, const void *T, void *C, float *partial, int M, int N, int K, int lda,
                                    int ldw, int ldt, int ldc, int group_rows, gm3d_stream_t stream);
int gm3d_gemm_ws_stats_rows(int M, int N, int K);
/* measurement knob: persistent workgroups per CU (1 or 2; two only where the LDS ring allows). Results do not depend on it. */
int gm3d_gemm_ws_set_occupancy(int wg_per_cu);
/* The ring kernel with 96-column tiles (N % 96 == 0): N = 384 -- attn.proj, mlp.fc2 and the input gradients of fc1 / qkv / proj of
 * the timm Block (Point-MAE_SA3D/models/Point_MAE.py:82-125) -- as four column tiles per row block instead of three: 200 / 256
 * workgroups instead of 150 / 192 for 256 CUs, each pulling fewer operand bytes from L2.  Bit-identical results.  bm = 64 or 128. */
int gm3d_gemm_tn_bf16_ring96(const void *A, const void *W, const float *bias, void *C, int M, int N, int K, int lda, int ldw, int ldc,
                             int bm, gm3d_stream_t stream);
/* The same products for SHORT K over many tiles (csrc/gemm_dma.hip: 64- or 128-row x 192-column tiles, LDS-DMA double buffer, two
 * workgroups per CU): qkv, fc1, proj, the proj / fc2 input gradients (K = 384).  N % 192 == 0, K % 64 == 0, 16-byte aligned
 * operands; results bit-identical to gm3d_gemm_tn_bf16 / gm3d_gemm_tn_bf16_ring.  bm = 64 or 128. */
int gm3d_gemm_tn_bf16_dma(const void *A, const void *W, const float *bias, void *C, int M, int N, int K, int lda, int ldw, int ldc,
                          int bm, gm3d_stream_t stream);
/* The same kernel with the tile width chosen by the caller: bn = 128, 192 or 256 columns (N % bn == 0).  The 256-column form is
 * for the mini-PointNet convolutions over the B*G*k = 262,144 point rows and their input gradients (Encoder.second_conv,
 * models_mae_learn_loss.py:878-883: 256 -> 512, 512 -> 384 and back), which stream A once from HBM; the 128-column form for
 * first_conv.3's input gradient (256 -> 128, :876) and pos_embed.2's (384 -> 128, :104-108).  Bit-identical to the other forms. */
int gm3d_gemm_tn_bf16_dmaw(const void *A, const void *W, const float *bias, void *C, int M, int N, int K, int lda, int ldw, int ldc,
                           int bm, int bn, gm3d_stream_t stream);
/* ... with the max-pool epilogue of gm3d_gemm_tn_bf16_pool (mini-PointNet Conv1d(k=1) + max over each group's 32 points,
 * models_mae_learn_loss.py:893,897): same roundings and decisions, bm x bn tiles with bn = 128 or 192. */
int gm3d_gemm_tn_bf16_dma_pool(const void *A, const void *W, const float *bias, void *C, void *P, unsigned char *arg, int M, int N,
                               int K, int lda, int ldw, int ldc, int ldp, int bias_after_pool, int bm, int bn, gm3d_stream_t stream);
/* ... with the fc1 epilogue of gm3d_gemm_tn_bf16_gelu: F (optional) = bf16(A.W^T), G = GELU(F + bias). */
int gm3d_gemm_tn_bf16_dma_gelu(const void *A, const void *W, const float *bias, void *F, void *G, int M, int N, int K, int lda,
                               int ldw, int ldf, int ldg, int bm, gm3d_stream_t stream);
/* ... with the fc2 input-gradient epilogue of gm3d_gemm_tn_bf16_gelu_bwd: dF = bf16(dO.Wt^T) * GELU'(F + bias) (bit-identical);
 * colpart (ceil(M / bm), N) f32 = per-row-tile column sums of the fp32 products. */
int gm3d_gemm_tn_bf16_dma_gelu_bwd(const void *dO, const void *Wt, const void *F, const float *bias, void *dF, float *colpart, int M,
                                   int N, int K, int lda, int ldw, int ldf, int lddf, int bm, gm3d_stream_t stream);
/* dst (batch, cols, rows) = transposes of `batch` row-major (rows, cols) bf16 matrices that start src_batch_stride elements apart
 * (the per-block weights of one kind inside the optimizer's flat bf16 shadow).  rows, cols multiples of 8 (64 x 64 tiles, ragged edges
 * guarded per 8-element chunk: the hierarchical model's 96 / 288-wide weights). */
int gm3d_transpose_bf16_batched(const void *src, void *dst, int batch, int rows, int cols, long long src_batch_stride,
                                gm3d_stream_t stream);
/* `count` (<= 8) of those in ONE launch (the four transposed weight shadows a block stack's backward reads). */
int gm3d_transpose_bf16_multi(int count, const void *const *src, void *const *dst, const int *batch, const int *rows, const int *cols,
                              const long long *src_batch_stride, gm3d_stream_t stream);
/* Token / positional-embedding assembly around the mask (models_mae_learn_loss.py:298-300,649-658) in one pass each way.
 * order (B,L) int64 = [visible ids | masked ids], a permutation of 0..L-1 per sample (gm3d_mask_select writes exactly this when its
 * two id outputs are the halves of one (B,L) buffer).  fwd: x_vis, pos_vis (B,V,C) and pos_full (B,L,C) gathered from tokens / pos
 * (B,L,C).  bwd: dtokens, dpos (B,L,C) from dx_vis, dpos_vis (B,V,C) and dpos_full (B,L,C) (any may be NULL = zero).
 * tokens and x_vis both NULL (fwd) / dtokens NULL (bwd): positions only -- the tokens were embedded for the visible groups alone. */
int gm3d_token_assemble_fwd(const void *tokens, const void *pos, const long long *order, int B, int L, int V, int C, void *x_vis,
                            void *pos_vis, void *pos_full, int dtype, gm3d_stream_t stream);
int gm3d_token_assemble_bwd(const void *dx_vis, const void *dpos_vis, const void *dpos_full, const long long *order, int B, int L,
                            int V, int C, void *dtokens, void *dpos, int dtype, gm3d_stream_t stream);
/* out (njobs, ncols) f32 = sum over the nrows (<= 64) rows of each job of partial (njobs, nrows, ncols): the S-way row-split
 * weight-gradient partial products.  ncols % 4 == 0. */
int gm3d_sum_few_rows(const float *partial, int njobs, int nrows, long long ncols, float *out, gm3d_stream_t stream);
/* Weight-gradient GEMM, batched: out[b] (N,K) f32 = dY[b]^T . X[b] with dY[b] (R,N) and X[b] (R,K) bf16, both row-major with row
 * pitches ldy / ldx (multiples of 8) and batch strides in elements; out row pitch ldo.  This is the .grad of every Linear /
 * Conv1d(k=1) weight of the path (P/models/Point_MAE.py:82-125, P/models_mae_learn_loss.py:873-882) -- the reference leaves it
 * to autograd (one cuBLAS NT GEMM per weight).  N, K multiples of 128, R a multiple of 32 * splits, 16-byte aligned operands.
 * splits > 1: the rows are cut into `splits` ranges and split s writes its partial product at out + b*stride_o + s*stride_split
 * (gm3d_sum_few_rows adds them in order); gm3d_gemm_nt_splits suggests a value for a shape. */
int gm3d_gemm_nt_bf16(const void *dY, const void *X, float *out, int batch, int R, int N, int K, int ldy, int ldx, int ldo,
                      long long stride_y, long long stride_x, long long stride_o, int splits, long long stride_split,
                      gm3d_stream_t stream);
int gm3d_gemm_nt_splits(int batch, int R, int N, int K);
/* Tile shape of gm3d_gemm_nt_bf16 where (N, K) is a multiple of (128, 384) -- every weight of the transformer blocks: 0 (default) =
 * 128 x 128 output tiles everywhere, 1 = 128 x 384 tiles (512-thread workgroups, 1.5 x the flops per byte staged from L2, fragment reads
 * software-pipelined against the MFMAs; measured SLOWER on MI355X, kept as a tested variant).  Results are equal for equal row splits;
 * gm3d_gemm_nt_splits follows the setting.  Process-wide measurement knob; set before capturing a graph. */
int gm3d_gemm_nt_set_big_tiles(int on);
/* measurement knob of gm3d_gemm_nt_bf16_multi: tile order inside a (block, row-split) group -- 0 (default): the K-tiles of one dY column
 * block adjacent; 1: the N-tiles of one X column block adjacent */
int gm3d_gemm_nt_set_order(int tn_fastest);
/* gm3d_gemm_nt_bf16 with the sum over the row splits INSIDE the launch (no gm3d_sum_few_rows pass): the workgroup that finishes a
 * tile's last slab adds that tile's slabs in slab order (bit-identical to the two-launch form, whatever the arrival order) and writes
 * out (batch, N, ldo; batch stride stride_o; ldo % 4 == 0).  part: (batch, splits, N, K) f32 scratch; counters: batch *
 * gm3d_gemm_nt_tiles(N, K) ints, zero on entry, zero again on exit, not shared with a launch that may run concurrently.  splits >= 2. */
int gm3d_gemm_nt_bf16_sum(const void *dY, const void *X, float *part, float *out, int *counters, int batch, int R, int N, int K, int ldy,
                          int ldx, int ldo, long long stride_y, long long stride_x, long long stride_o, int splits, gm3d_stream_t stream);
int gm3d_gemm_nt_tiles(int N, int K);
/* `count` (<= 16) problems of gm3d_gemm_nt_bf16 (128 x 128 tiles) as ONE launch + ONE slab-sum launch: the twelve batched weight
 * gradients of a step's three block stacks (timm Block qkv / proj / fc1 / fc2, P/models/Point_MAE.py:82-125) without twelve ramps and
 * partly filled last waves.  Per problem j: dY[j] (batch, R, N) bf16 rows of pitch ldy[j], X[j] (batch, R, K), out[j] (batch, N, K) f32
 * with batch stride stride_o[j], part[j] (batch, splits, N, K) f32 scratch when splits[j] > 1 (else ignored).  Bit-identical to the
 * separate launches.  The descriptors are passed by value (a captured launch keeps them). */
int gm3d_gemm_nt_bf16_multi(int count, const void *const *dY, const void *const *X, float *const *out, float *const *part, const int *batch,
                            const int *R, const int *N, const int *K, const int *ldy, const int *ldx, const long long *stride_y,
                            const long long *stride_x, const long long *stride_o, const int *splits, gm3d_stream_t stream);
/* Masked multi-head attention of the hierarchical (Point-M2AE) encoder blocks -- SURVEY.md 8f.4; the reference ships only the
 * hyper-parameters (Point-M2AE_SA3D/cfgs/config_Point_M2AE.yaml:57-99: dims 96/192/384, 6 heads, local_radius 0.32/0.64/1.28).
 * qkv (B,T,3,H,HD) as the qkv Linear emits it, HD in {16,32,64}, T <= 512; mask (B,T,ceil(T/32)) uint32 bitset, bit (j&31) of
 * word j>>5 of row i set = query i must not attend to key j (NULL: no mask); the mask must be symmetric.  out (B,T,H*HD), lse
 * (B,H,T) f32 (0 for a query with no allowed key, whose output and gradients are zero).  dtype GM3D_BF16: flash-style MFMA
 * kernels; GM3D_F32: exact fp32 kernels (parity mode).  bf16 keeps the head in LDS: the forward needs (2*32*ceil(T/32) + 128) *
 * (2 HD + 16) bytes, the backward 4*32*ceil(T/32) * (2 HD + 16) + 8*32*ceil(T/32), at most 160 KiB -- i.e. T <= 512 for HD <= 32
 * and, for HD = 64, T <= 480 forward / T <= 256 backward (GM3D_EUNSUPPORTED beyond).  With mask = NULL and HD = 64 these are also
 * the attention of the Point-MAE models for more than 128 tokens (cfgs/config_3.yaml: 256 groups). */
/* Deterministic backward of a row gather with repeated indices (the hierarchical model's member / 3-NN token gathers: PyTorch's
 * gather backward scatter-adds with colliding float atomics).  gm3d_gather_inverse: idx (B,J) int64 with values in [0,S) -> the
 * inverse lists in CSR form, off (B,S+1) int32 and list (B,J) int32, every list in ascending j (S <= 4096, J <= 16384);
 * gm3d_gather_rows_bwd: dx (B,S,C) = sum over each source's list of dy (B,J,C) rows, in list order (fp32 accumulation, C % 8 == 0). */
int gm3d_gather_inverse(const long long *idx, int B, int J, int S, int *off, int *list, gm3d_stream_t stream);
int gm3d_gather_rows_bwd(const void *dy, const int *off, const int *list, void *dx, int B, int J, int S, int C, int dtype,
                         gm3d_stream_t stream);
/* The bitset mask of one level of the hierarchical encoder: bits (B,G,ceil(G/32)), bit j of row i set iff token i or token j is
 * not visible (vis (B,G) bytes, NULL = all visible) or their centres (B,G,3) are >= radius apart (radius <= 0: no radius test). */
int gm3d_radius_mask_bits(const float *center, const unsigned char *vis, float radius, int B, int G, unsigned *bits,
                          gm3d_stream_t stream);
/* ... with the flags' sense selectable: flags_are_masked != 0 -> a non-zero byte means NOT visible (the multi-scale masks as they are) */
int gm3d_radius_mask_bits_m(const float *center, const unsigned char *flags, int flags_are_masked, float radius, int B, int G,
                            unsigned *bits, gm3d_stream_t stream);
/* Visible-first token order of one masked level (the student's pass: multi-scale masking at ratio 0.8,
 * Point-M2AE_SA3D/cfgs/config_Point_M2AE.yaml:57-99; gm3d_amd/point_m2ae.py).  masked (B,T) bytes (non-zero = masked), T <= 1024,
 * 1 <= Tc <= T (a static bound on the visible count, or T):  perm_c (B,Tc) int32 = token at compact slot j (visible tokens in
 * ascending order, then masked ones as filler); perm_v (B,Tc) = the same with -1 in filler slots; inv_v (B,T) = slot of token t
 * if visible and below Tc, else -1; inv_m (B,T) = t if token t is masked, else -1; vis_c (B,Tc) bytes = 1 in slots holding a
 * visible token; *overflow set to 1 if a cloud has
 * more than Tc visible tokens (the caller's bound was wrong; never cleared here). */
int gm3d_partition_visible(const unsigned char *masked, int B, int T, int Tc, int *perm_c, int *perm_v, int *inv_v, int *inv_m,
                           unsigned char *vis_c, int *overflow, gm3d_stream_t stream);
/* Multi-scale masking one level down (gm3d_amd/point_m2ae.py back_project): masked_f (B,Gf) bytes = 1 unless some VISIBLE group of the
 * coarser level (masked_c (B,Gc) bytes, 0 = visible) lists the finer group among its k members (member (B,Gc,k) int64).  Gf <= 8192. */
int gm3d_back_project(const unsigned char *masked_c, const long long *member, int B, int Gc, int k, int Gf, unsigned char *masked_f,
                      gm3d_stream_t stream);
/* Token propagation of the hierarchical decoder (3-NN inverse-squared-distance interpolation, PointNet++ style; configuration
 * Point-M2AE_SA3D/cfgs/config_Point_M2AE.yaml:88-99): out (B,N,C1+C2) = [fine (B,N,C1) | sum_j w[b][n][j] * coarse[b][idx[b][n][j]]],
 * coarse (B,S,C2), idx (B,N,3) int64, w (B,N,3) f32; C1, C2 % 8 == 0 (C1 = 0: no fine part).  Backward of the interpolated half:
 * gm3d_gather_inverse over idx viewed as (B,3N), then gm3d_gather_rows_bwd_w: dx (B,S,C) = sum over each source's list, in list order,
 * of w[b][j] * dy[b][j / refs_per_row][col0 : col0 + C] (dy rows of pitch ldy). */
int gm3d_interp3_fwd(const void *coarse, const long long *idx, const float *w, const void *fine, void *out, int B, int N, int S, int C1,
                     int C2, int dtype, gm3d_stream_t stream);
int gm3d_gather_rows_bwd_w(const void *dy, int ldy, int col0, int refs_per_row, const float *w, const int *off, const int *list, void *dx,
                           int B, int J, int S, int C, int dtype, gm3d_stream_t stream);
/* out[row] = flag[row] (xor invert) ? alt[row] : a[row] over `rows` rows of C elements (elem_bytes 2 / 4); a NULL / alt NULL: zeros;
 * alt_bcast: alt is one row (C) for every row (the decoder's mask token).  gm3d_amd/point_m2ae.py's per-token choices and their backward. */
int gm3d_where_rows(const unsigned char *flag, int invert, const void *a, const void *alt, int alt_bcast, void *out, long long rows, int C,
                    int elem_bytes, gm3d_stream_t stream);
/* out (B,J,C) = a[b][idx[b][j]] for int64 idx (B,J) with repeats (the forward of the member gathers; backward: gm3d_gather_rows_bwd). */
int gm3d_take_rows(const void *a, const long long *idx, void *out, int B, int S, int J, int C, int elem_bytes, gm3d_stream_t stream);
/* out (B,T,C) row by row: idx[b][t] >= 0 -> a[b][idx[b][t]] (a is (B,Ta,C)), else alt[b][t] (alt (B,T,C)) or zeros (alt NULL).
 * elem_bytes 2 or 4.  Gather into / scatter out of the compact order above, forward and backward (indices without repeats). */
int gm3d_select_rows(const void *a, const int *idx, const void *alt, void *out, int B, int Ta, int T, int C, int elem_bytes,
                     gm3d_stream_t stream);
/* measurement knob: 1 (default) = eight 32-row tiles per workgroup for HD <= 32 and T > 128, 0 = four everywhere */
int gm3d_attention_masked_set_wide(int on);
int gm3d_attention_masked_fwd(const void *qkv, const unsigned *mask, void *out, float *lse, int B, int T, int H, int HD, float scale,
                              int dtype, gm3d_stream_t stream);
int gm3d_attention_masked_bwd(const void *qkv, const unsigned *mask, const void *out, const void *dout, const float *lse, void *dqkv,
                              int B, int T, int H, int HD, float scale, int dtype, gm3d_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* GM3D_H */
