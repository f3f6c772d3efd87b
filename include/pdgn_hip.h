/*
 * pdgn_hip.h -- C ABI of libpdgn_hip.so: the MI355X (gfx950) drop-in for the native
 * side of PDGN's hot path.
 *
 * Every entry point takes plain device pointers + sizes + a hipStream_t (passed as
 * void*), launches asynchronously on that stream, never allocates, never synchronises,
 * and returns 0 on success, a hipError_t value (>0) on a launch failure, or
 * PDGN_ERR_INVALID (-1) when an argument is out of the supported range.  The pointops /
 * structural-loss / BatchNorm / gather entry points keep no state.  The dense contractions
 * (pdgn_gemm_*) have state, all of it listed at pdgn_gemm_set_mode below: process-wide
 * switches (arithmetic mode, forced tile, matrix instruction: set once from the
 * environment, changed only by tests / tools), a process-wide arena for the two-part
 * mode's own scans (pdgn_gemm_set_scale_slots), and THREAD-LOCAL hand-overs (operand
 * maxima, tail workspace) that belong to the calling thread's next contraction call.
 * One process per GPU, contraction calls from one thread at a time per hand-over.
 *
 * All tensors are contiguous, batch-major; float = fp32, idx = int32 -- exactly the
 * layouts of the reference launchers cited per function (paths relative to the
 * reference repository root).
 */
#ifndef PDGN_HIP_H
#define PDGN_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PDGN_ERR_INVALID (-1)
#define PDGN_KNN_MAX_NSAMPLE 200 /* lib/pointops/src/knnquery/knnquery_cuda_kernel.cu:21-22 */

typedef void *pdgn_stream_t; /* hipStream_t */

/* ABI version of this header (bumped on any signature change). */
int pdgn_abi_version(void);

/* ------------------------------------------------------------------ pointops
 * Replaces knnquery_cuda_launcher (lib/pointops/src/knnquery/knnquery_cuda_kernel.h:14,
 * kernel knnquery_cuda_kernel.cu:6-50).  xyz (b,n,3), new_xyz (b,m,3) ->
 * idx (b,m,nsample) int32, dist2 (b,m,nsample) f32 (dist2 may be NULL).
 * Ascending (squared distance, index); for n < nsample the tail is idx 0 / +inf.
 * nsample <= PDGN_KNN_MAX_NSAMPLE. */
int pdgn_knnquery(int b, int n, int m, int nsample, const float *xyz, const float *new_xyz,
                  int32_t *idx, float *dist2, pdgn_stream_t stream);

/* Replaces grouping_forward_cuda_launcher_fast (grouping/grouping_cuda_kernel.h:19,
 * kernel grouping_cuda_kernel.cu:60-74).  points (b,c,n), idx (b,m,nsample) ->
 * out (b,c,m,nsample). */
int pdgn_grouping_forward(int b, int c, int n, int m, int nsample, const float *points,
                          const int32_t *idx, float *out, pdgn_stream_t stream);

/* Replaces grouping_backward_cuda_launcher (grouping/grouping_cuda_kernel.h:17,
 * kernel :28-46).  grad_out (b,c,m,nsample), idx -> grad_points (b,c,n), ACCUMULATED
 * into the caller's buffer (the caller zero-fills it, pointops.py:146). */
int pdgn_grouping_backward(int b, int c, int n, int m, int nsample, const float *grad_out,
                           const int32_t *idx, float *grad_points, pdgn_stream_t stream);

/* Replaces nearestneighbor_cuda_launcher_fast (interpolation/interpolation_cuda_kernel.h:22,
 * kernel interpolation_cuda_kernel.cu:134-176).  unknown (b,n,3), known (b,m,3) ->
 * dist2 (b,n,3) squared distances, idx (b,n,3). */
int pdgn_nearestneighbor(int b, int n, int m, const float *unknown, const float *known,
                         float *dist2, int32_t *idx, pdgn_stream_t stream);

/* Replaces interpolation_forward_cuda_launcher_fast (interpolation_cuda_kernel.h:23,
 * kernel :181-195).  points (b,c,m), idx/weight (b,n,3) -> out (b,c,n). */
int pdgn_interpolation_forward(int b, int c, int m, int n, const float *points,
                               const int32_t *idx, const float *weight, float *out,
                               pdgn_stream_t stream);

/* Replaces interpolation_backward_cuda_launcher (interpolation_cuda_kernel.h:20,
 * kernel :90-114).  grad_out (b,c,n) -> grad_points (b,c,m), ACCUMULATED. */
int pdgn_interpolation_backward(int b, int c, int n, int m, const float *grad_out,
                                const int32_t *idx, const float *weight, float *grad_points,
                                pdgn_stream_t stream);

/* ------------------------------------------------------------------ structural losses
 * Replaces nndistance (evaluation/pytorch_structural_losses/src/nndistance.cuh:1,
 * nndistance.cu:2-128).  xyz (b,n,3), xyz2 (b,m,3) -> result/result_i (b,n),
 * result2/result2_i (b,m): min squared distance and its argmin (lowest index on ties). */
int pdgn_nndistance(int b, int n, const float *xyz, int m, const float *xyz2, float *result,
                    int32_t *result_i, float *result2, int32_t *result2_i, pdgn_stream_t stream);

/* Replaces nndistancegrad (nndistance.cuh:2, nndistance.cu:129-154).  Zero-fills
 * grad_xyz1 (b,n,3) / grad_xyz2 (b,m,3) itself (on `stream`), then accumulates. */
int pdgn_nndistance_grad(int b, int n, const float *xyz1, int m, const float *xyz2,
                         const float *grad_dist1, const int32_t *idx1, const float *grad_dist2,
                         const int32_t *idx2, float *grad_xyz1, float *grad_xyz2,
                         pdgn_stream_t stream);

/* Replaces approxmatch (src/approxmatch.cuh:6, approxmatch.cu:3-182,299-307).
 * xyz1 (b,n,3), xyz2 (b,m,3) -> match (b,m,n); temp (b, 2*(n+m)) is scratch. */
int pdgn_approxmatch(int b, int n, int m, const float *xyz1, const float *xyz2, float *match,
                     float *temp, pdgn_stream_t stream);

/* Replaces matchcost (approxmatch.cuh:7, approxmatch.cu:184-224,309-316) -> out (b). */
int pdgn_matchcost(int b, int n, int m, const float *xyz1, const float *xyz2,
                   const float *match, float *out, pdgn_stream_t stream);

/* Replaces matchcostgrad (approxmatch.cuh:8, approxmatch.cu:229-291,318-326) ->
 * grad1 (b,n,3), grad2 (b,m,3). */
int pdgn_matchcost_grad(int b, int n, int m, const float *xyz1, const float *xyz2,
                        const float *match, float *grad1, float *grad2, pdgn_stream_t stream);

/* Fused forward-only EMD cost for the eval path (evaluation_metrics.py:26-31 calls
 * ApproxMatch then MatchCost and drops `match`): same arithmetic as
 * pdgn_approxmatch + pdgn_matchcost but `match` is never written to HBM.
 * temp: pdgn_emd_cost_temp_floats(b, n, m) floats of scratch (per pair the reference's remain / ratio vectors and the two
 * clouds as x-sorted float4 rows: the sweeps skip the runs of a tile whose exponentials are exactly zero), out (b). */
long long pdgn_emd_cost_temp_floats(long long pairs, int n, int m);
int pdgn_emd_cost(int b, int n, int m, const float *xyz1, const float *xyz2, float *temp,
                  float *out, pdgn_stream_t stream);

/* ------------------------------------------------------------------ point-deconvolution stack
 * Feature-space kNN graph of models/PDGNet_v2.py:447-458 / :488-502 (get_edge_features[_xyz]):
 * x (b,f,n) channels-first -> idx (b,n,k) int32 = ranks 1..k of the row-wise ascending order of
 * (-2<x_i,x_j> + |x_i|^2) + |x_j|^2 (rank 0 dropped; ties by index).  sqnorm (b,n) is scratch.
 * f <= 256, k <= 31, n >= k+1. */
int pdgn_feature_knn(int b, int f, int n, int k, const float *x, float *sqnorm, int32_t *idx,
                     pdgn_stream_t stream);

/* Gather half of the re-associated edge convolutions (inte_conv_hk / conv2 / conv_fea / conv_xyz,
 * models/PDGNet_v2.py:559-625 applied to the edge tensors of :462-477, :505-525):
 *   out[b,n,p,c] = bias[b*bias_bstride + c] + Y[b,n,offc+c] + sum_{t<T} Y[b, idx[b,n,p+t], off + t*C + c]
 * Y (b,n,ldy) point-major, idx (b,n,k) with k >= T+P-1, out (b,n,P,C); bias may be NULL, is shared
 * (bias_bstride = 0) or per batch (bias_bstride = C: the contribution of input channels that are
 * constant over the points of a sample -- the broadcast global vector of :704-708 -- never enters
 * the per-point GEMM); offc < 0 drops the centre term. */
int pdgn_window_gather_sum(int b, int n, int k, int ldy, int T, int P, int C, int off, int offc,
                           const float *Y, const int32_t *idx, const float *bias, int bias_bstride,
                           float *out, pdgn_stream_t stream);
/* The same, also emitting the BatchNorm partial statistics of out viewed as (b*n*P, C) rows into `scratch`
 * (>= pdgn_bn_scratch_floats(b*n*P, C) floats; finish with pdgn_bn_stats_from_partials): the BatchNorm that follows
 * the gather-sum needs no statistics pass.  float4 path only (C, ldy, off, offc, bias_bstride multiples of 4). */
int pdgn_window_gather_sum_stats(int b, int n, int k, int ldy, int T, int P, int C, int off, int offc,
                                 const float *Y, const int32_t *idx, const float *bias, int bias_bstride,
                                 float *out, float *scratch, pdgn_stream_t stream);

/* Its adjoint: dY[b, idx[b,n,p+t], off+t*C+c] += dout[b,n,p,c] (atomic), and
 * dY[b,n,offc+c] = sum_p dout[b,n,p,c].  dY must be zero-filled by the caller. */
int pdgn_window_gather_sum_backward(int b, int n, int k, int ldy, int T, int P, int C, int off,
                                    int offc, const float *dout, const int32_t *idx, float *dY,
                                    pdgn_stream_t stream);

/* Transposed kNN graph (CSR over source points) of idx (b,n,k): rowptr (b,n+1), edges (b,n*k) with
 * edge records n*32 + s (k <= 31); scratch: 2*b*n ints. */
int pdgn_knn_graph_transpose(int b, int n, int k, const int32_t *idx, int32_t *rowptr, int32_t *edges,
                             int32_t *scratch, pdgn_stream_t stream);
/* Atomic-free form of pdgn_window_gather_sum_backward over the transposed graph: every element of
 * dY's column blocks [off, off+T*C) and [offc, offc+C) is WRITTEN exactly once (no zero-fill). */
int pdgn_window_gather_sum_backward_csr(int b, int n, int k, int ldy, int T, int P, int C, int off,
                                        int offc, const float *dout, const int32_t *rowptr,
                                        const int32_t *edges, float *dY, unsigned *max_out, int max_init, pdgn_stream_t stream);
/* (max_out, may be NULL: uint32[b n], the maximum of |dY| (bit pattern) over every row (b, j) of dY, collected over the calls that
 * fill one dY -- the first of them passes max_init = 1 and the array is zero-filled; atomic max, order-independent -- for
 * pdgn_gemm_set_operand_scales: the per-point product's input gradient takes dY as its first operand and scales it row by row.) */

/* Fused BatchNorm + activation over channels-last (rows x c) activations -- the
 * nn.BatchNorm2d + LeakyReLU/ReLU pairs of models/PDGNet_v2.py:537-545, 561-565, 603-625 in the
 * point-major layout.  act: 0 none, 1 ReLU, 2 LeakyReLU(0.01).  c % 4 == 0.
 *
 * pdgn_bn_scratch_floats(rows, c): number of floats of `scratch` the two reductions below need.
 * pdgn_bn_stats: batch statistics (training mode), two-stage reduction through `scratch`; stats out:
 * [scale | shift | mean | invstd] (4c floats), scale = gamma*invstd, shift = beta - mean*scale;
 * running_mean/var (may be NULL) are updated with `momentum` and the unbiased variance.
 * pdgn_bn_eval_stats: the same `stats` from the running statistics (eval mode).
 * pre_bias (may be NULL, c floats): the bias of the conv / linear layer that produced x when the caller left it
 * OUT of x (BatchNorm(x + b) == BatchNorm(x), so the GEMM needs no bias epilogue); it only enters the running mean
 * (training) and the effective mean (eval), which keeps nn.BatchNorm's buffers identical to the reference's. */
long long pdgn_bn_scratch_floats(long long rows, int c);
int pdgn_bn_stats(long long rows, int c, float eps, float momentum, const float *x, const float *gamma,
                  const float *beta, const float *pre_bias, float *running_mean, float *running_var,
                  float *scratch, float *stats, pdgn_stream_t stream);
int pdgn_bn_eval_stats(int c, float eps, const float *gamma, const float *beta, const float *pre_bias,
                       const float *running_mean, const float *running_var, float *stats,
                       pdgn_stream_t stream);
/* Statistics whose first stage was done by a producer: `scratch` holds the per-row-block partial sums of x in the
 * layout of pdgn_bn_stats (pdgn_window_gather_sum_stats writes them while producing x). */
int pdgn_bn_stats_from_partials(long long rows, int c, float eps, float momentum, const float *gamma,
                                const float *beta, const float *pre_bias, float *running_mean, float *running_var,
                                const float *scratch, float *stats, pdgn_stream_t stream);
/* The second stage over the BLOCK-SHIFTED partials pdgn_gemm_nt / pdgn_gemm_nn / pdgn_thin_nt write from their epilogue:
 * nparts rows of [3c] floats, row b = per-column  sum (x - pv_b) | sum (x - pv_b)^2 | pv_b  over rows b*block_rows .. of x,
 * pv_b = the block's own first row (no cancellation when |mean| >> std); combined per block in fp64. */
int pdgn_bn_stats_from_gemm_partials(long long rows, int c, long long nparts, int block_rows, float eps, float momentum,
                                     const float *gamma, const float *beta, const float *pre_bias,
                                     float *running_mean, float *running_var, const float *partials,
                                     float *stats, double *scratch, pdgn_stream_t stream);
/* scratch: pdgn_bn_blocks_scratch_doubles(c, nparts) fp64 elements (may be NULL, and is unused for short lists): long
 * lists are summed by several workgroups per channel group in a first launch and joined, in order, by a second. */
long long pdgn_bn_blocks_scratch_doubles(int c, long long nparts);
/* y = act(x*scale + shift) [* mul]   (mul may be NULL; same shape as x) */
int pdgn_bn_act_forward(long long rows, int c, int act, const float *x, const float *stats,
                        const float *mul, float *y, int interleave_n, pdgn_stream_t stream);
/* dz = dy [* mul] * act'(z);  bsums (2c floats out): [0:c] = sum dz (= dbeta), [c:2c] = sum dz*xhat (= dgamma);
 * dx = scale*(dz - mean(dz) - xhat*mean(dz*xhat)) if training else scale*dz;
 * dmul (may be NULL) = dy * act(z). */
int pdgn_bn_act_backward(long long rows, int c, int act, int training, const float *x,
                         const float *dy, const float *mul, const float *stats, float *scratch,
                         float *bsums, float *dx, float *dmul, int interleave_n, pdgn_stream_t stream);
/* interleave_n = N > 0 (c even... a multiple of 4, mul NULL, rows % N == 0): y / dy are stored INTERLEAVED -- x row b*N + n, channel
 * 2c + j  <->  y row b*2N + j*N + n, channel c ((rows * 2) x (c / 2)): the (B,2Fout,N,1) -> (B,Fout,2N) regrouping of a
 * deconvolution block's result (models/PDGNet_v2.py:645-647) in point-major form, without a separate permute copy. */

/* BatchNorm + activation + max-pool over the n rows of each sample (the tail of the PointNet-style
 * discriminators, models/PDGNet_v2.py:886-911): x (b*n, c) -> ymax / yarg (b, c); the activated tensor
 * is never written.  stats from pdgn_bn_stats / pdgn_bn_eval_stats over all b*n rows.
 * scratch: pdgn_bn_maxpool_scratch_floats(b, c) floats. */
long long pdgn_bn_maxpool_scratch_floats(int b, int c);
int pdgn_bn_act_maxpool(int b, int n, int c, int act, const float *x, const float *stats, float *scratch,
                        float *ymax, int32_t *yarg, pdgn_stream_t stream);
/* Training mode without statistics from the producer: batch statistics (stats (4c) out; running statistics updated as by pdgn_bn_stats)
 * AND the pooled output in one pass over x -- the pass keeps per (sample, split, channel) the largest and the smallest x and picks by
 * the sign of the channel's scale afterwards (act(scale x + shift) is monotone in x).  scratch: pdgn_bn_stats_maxpool_scratch_floats. */
long long pdgn_bn_stats_maxpool_scratch_floats(int b, int c);
int pdgn_bn_stats_act_maxpool(int b, int n, int c, int act, float eps, float momentum, const float *x, const float *gamma,
                              const float *beta, const float *pre_bias, float *running_mean, float *running_var, float *scratch,
                              float *stats, float *ymax, int32_t *yarg, pdgn_stream_t stream);
/* Its adjoint in one streaming pass: dx (b*n, c) from dout (b, c); bsums (2c) = [dbeta | dgamma];
 * scratch: b*c + 2c floats. */
int pdgn_bn_act_maxpool_backward(int b, int n, int c, int act, int training, const float *x,
                                 const float *dout, const int32_t *yarg, const float *stats,
                                 float *scratch, float *bsums, float *dx, pdgn_stream_t stream);
/* The same adjoint carried through the dense layer in front of it (PointDiscriminator_k.fc1's last Conv1d + BatchNorm1d + LeakyReLU +
 * MaxPool1d, models/PDGNet_v2.py:886-911; training statistics): with x = h W^T and dx = S - 1 ca^T - x diag(cb) (S non-zero at the
 * b*c arg-max entries only)
 *     dh = S W - 1 (ca^T W) - h (W^T diag(cb) W)        one (b*n, k, k) product, a k x k Gram matrix, a scatter of b*c rows of W
 *     dW = S^T h - ca (1^T h) - diag(cb) W (h^T h)      one k x k Gram matrix of h, one (c, k, k) product, a gather of b*c rows of h
 * -- the dense (b*n, c) gradient is never formed and neither (b*n, c, k) product is computed.
 * x (b*n, c) the layer's raw output, yarg / stats as saved by pdgn_bn_act_maxpool, dout (b, c), h (b*n, k) pitch ldh, W (c, k) pitch
 * ldw (16-byte aligned rows); k % 4 == 0, k <= 256, n <= 65535.  Outputs, each may be NULL: dh (b*n, k) contiguous; dW (c, k) pitch
 * lddw (needs k a power of two >= 16); bsums (2c) = [dbeta | dgamma].  scratch: pdgn_dense_bn_maxpool_backward_scratch floats. */
long long pdgn_dense_bn_maxpool_backward_scratch(int b, int c, int k);
int pdgn_dense_bn_maxpool_backward(int b, int n, int c, int k, int act, const float *x, const float *dout, const int32_t *yarg,
                                   const float *stats, const float *h, int ldh, const float *W, int ldw, float *scratch,
                                   float *dh, float *dW, int lddw, float *bsums, pdgn_stream_t stream);
/* The input gradient alone (frozen parameters: the generator's update through a discriminator, models/PDGNet_v2.py:330-352). */
long long pdgn_dense_bn_maxpool_input_grad_scratch(int b, int c, int k);
int pdgn_dense_bn_maxpool_input_grad(int b, int n, int c, int k, int act, const float *x, const float *dout,
                                     const int32_t *yarg, const float *stats, const float *h, int ldh,
                                     const float *W, int ldw, float *scratch, float *dh, pdgn_stream_t stream);

/* Column sums per group of rows: out[g, c] = sum over rows g*group_rows .. (g+1)*group_rows - 1 of x[r, c]; x (groups*group_rows, c)
 * pitch ldx (16-byte aligned rows), c a power of two in 16 .. 1024, out (groups, c) contiguous and ZERO on entry.  The bias
 * gradients torch's autograd takes with sum(dim=...) behind nn.Conv1d / nn.Conv2d layers that no BatchNorm follows
 * (models/PDGNet_v2.py:835-862 heads, :604-618 per-sample terms of the edge convolutions). */
int pdgn_group_colsum(long long groups, long long group_rows, int c, const float *x, long long ldx, float *out,
                      pdgn_stream_t stream);

/* Softmax over the k neighbour slots fused with the slot/channel interleave of
 * models/PDGNet_v2.py:634-641: h (m,k,c) -> w (m, k/2, 2c) with w[m,p,2c'+j] = softmax_s(h[m,:,c'])[s=(k/2)j+p].
 * k even, k <= 32. */
int pdgn_softmax_slots_permute(long long m, int k, int c, const float *h, float *w, pdgn_stream_t stream);
/* Same with the preceding BatchNorm + activation folded in (conv_all.4 + LeakyReLU, models/PDGNet_v2.py:623-625):
 * x (m,k,c) is the raw conv output, stats the [scale|shift|mean|invstd] row of pdgn_bn_stats / pdgn_bn_eval_stats,
 * act as in pdgn_bn_act_forward; the activated logits are never written.  c even. */
int pdgn_bn_softmax_slots_permute(long long m, int k, int c, int act, const float *x, const float *stats,
                                  float *w, pdgn_stream_t stream);
/* The whole bilateral weighting (models/PDGNet_v2.py:623-642) in one pass: w as above and
 * y = act_u(u*scale_u + shift_u) * w for u (m, k/2, 2c) the raw inte_conv_hk output (stats_u: its 4*(2c) statistics
 * row); w may be NULL (not needed when no backward pass follows). */
int pdgn_bn_softmax_slots_permute_mul(long long m, int k, int c, int act, const float *x, const float *stats,
                                      int act_u, const float *u, const float *stats_u, float *w, float *y,
                                      unsigned *max_out, const float *gamma_u, const float *beta_u, float bound_scale,
                                      unsigned *cmax_out, pdgn_stream_t stream);
/* (max_out, may be NULL: uint32[m], the maximum of |y| (bit pattern) over each point's k c outputs = the row maxima of y as the
 * (m, k c) first operand of conv2's dense half -- what pdgn_absmax_rows_cols would compute by a pass over y; zero-filled by the
 * call, atomic max.
 * cmax_out, may be NULL: uint32[k c], an upper BOUND of that operand's column maxima (its weight gradient scales it column by column):
 * |beta_u[j]| + |gamma_u[j]| * bound_scale for column (p, j) -- with bound_scale = sqrt(n - 1) for batch statistics over n = m k / 2
 * samples, since |xhat| <= sqrt(n - 1) and the softmax weights are <= 1; gamma_u / beta_u: the 2c BatchNorm parameters of u.) */
/* Adjoint of pdgn_bn_softmax_slots_permute_mul in two passes over (x, u, w, dy): BatchNorm_u backward, slot-softmax
 * backward and BatchNorm_x backward with dW / dh kept in registers.  scratch: pdgn_bilateral_scratch_floats(m,k,c)
 * floats; bsums_x (2c) = [sum dz_x | sum dz_x*xhat] (= dbeta, dgamma of BN_x), bsums_u (4c) likewise for BN_u;
 * training = 0: running-statistics BatchNorms (no batch terms in dx / du).  c % 4 == 0, k even, k <= 16 (all k slots of a
 * channel pair live in registers; wider neighbourhoods: pdgn_bn_softmax_slots_permute + pdgn_bn_act_backward). */
long long pdgn_bilateral_scratch_floats(long long m, int k, int c);
int pdgn_bilateral_weighting_backward(long long m, int k, int c, int act, int training, const float *x,
                                      const float *stats_x, const float *u, const float *stats_u, const float *w,
                                      const float *dy, float *scratch, float *bsums_x, float *bsums_u, float *dx,
                                      float *du, pdgn_stream_t stream);
/* dh[m,s,c'] = w_s (dw_s - sum_s' w_s' dw_s'), w / dw in the permuted layout. */
int pdgn_softmax_slots_permute_backward(long long m, int k, int c, const float *w, const float *dw,
                                        float *dh, pdgn_stream_t stream);

/* Dense contraction of a point-major layer on the matrix cores, fp32 in and out:
 *   C (m x n, row pitch ldc) = A (m x k, pitch lda) W (n x k, pitch ldw)^T (+ bias[n]) (+ addend (m x n, pitch ldadd)),
 * all row-major; n, k and every pitch are multiples of 4 floats, base pointers 16-byte aligned.
 * Arithmetic (pdgn_gemm_nt / _nn / _nt_ex / _tn_big): every fp32 operand value is split into three bf16 parts
 * (x = h + m + l to 2^-25 |x|) and a product is six bf16 MFMA products accumulated in fp32 (csrc/gemm_x3.hip) -- per product
 * an error of <= 2^-23 |a w|, measured against fp64 below that of the fp32 matrix instructions; no scaling, the fp32
 * exponent range is kept; an infinite operand value yields NaN (inf - inf in the split), where fp32 arithmetic may yield inf.
 * In the default mode the large products run on TWO scaled fp16 parts and three fp16 MFMA products instead (mode 2 below: same
 * operands and results, half the matrix-core work, against fp64 below both other forms); PDGN_GEMM=x3 keeps three parts everywhere.
 * PDGN_GEMM=fp32 in the environment at first use, or pdgn_gemm_set_mode(0), selects the fp32 matrix instructions instead
 * (csrc/gemm_nt.hip; 0.6-0.9x the rate).
 * (The reference's Conv2d/Conv1d/Linear forward at models/PDGNet_v2.py:559-625, 835-862, 886-1014 in
 * point-major form; with the transposed weight it is their input gradient dX = dY W.)  stat_part (may be
 * NULL): pdgn_gemm_nt_stat_rows(m, n, k) rows of [3n] floats = per-column sum (x - pv) | sum (x - pv)^2 | pv of blocks of
 * pdgn_gemm_nt_stat_block_rows(m, n, k) rows of C (pv: the block's first row), the partials
 * pdgn_bn_stats_from_gemm_partials turns into BatchNorm statistics.  Without
 * stat_part and with ldc == n the launch may add partial tiles with fp32 atomics (C is zero-filled by the
 * call itself where needed). */
/* Products with a per-sample operand of R <= 64 rows (csrc/skinny.hip; the 35-row layers of models/PDGNet_v2.py:704-707,
 * 825-828, 835-862 and the constant-channel contribution of DESIGN.md section 2), fp32 matrix instructions, one pass over the large
 * operand:
 *   pdgn_skinny_nt: C (R x N) = A (R x K) B (N x K)^T (+ bias[N])      K % 4 == 0, rows of A / B 16-byte aligned
 *   pdgn_skinny_nn: C (R x N) += A (R x K) B (K x N)                    C ZERO-FILLED by the caller (K slices add with fp32 atomics); N % 4 == 0
 *   pdgn_skinny_tn: C (N x K) = A (R x N)^T B (R x K)
 * every operand with its own row pitch (a column slice of a wider weight needs no copy). */
int pdgn_skinny_nt(int R, int N, int K, const float *A, int lda, const float *B, int ldb, const float *bias, float *C, int ldc,
                   pdgn_stream_t stream);
int pdgn_skinny_nn(int R, int N, int K, const float *A, int lda, const float *B, int ldb, float *C, int ldc, pdgn_stream_t stream);
/* pdgn_skinny_nt with an activation after the bias (act 0 none, 1 ReLU, 2 LeakyReLU(0.01)): a discriminator head's nn.Linear +
 * nn.LeakyReLU on the batch's pooled rows (models/PDGNet_v2.py:896-911) in one launch. */
int pdgn_skinny_nt_act(int R, int N, int K, const float *A, int lda, const float *B, int ldb, const float *bias, float *C, int ldc,
                       int act, pdgn_stream_t stream);
/* pdgn_skinny_nn with A[r][k] scaled by act'(P[r][k]) on load (act 1 = ReLU, 2 = LeakyReLU(0.01); P (R x K) the pre-activation a
 * Linear + activation saved): that layer's input gradient dx = (dy * act'(pre)) W in one launch when its parameters are frozen
 * (the discriminators' nn.Linear heads during the generator's update, models/PDGNet_v2.py:330-352, :896-911). */
int pdgn_skinny_nn_masked(int R, int N, int K, const float *A, int lda, const float *P, int ldp, int act, const float *B,
                          int ldb, float *C, int ldc, pdgn_stream_t stream);
int pdgn_skinny_tn(int R, int N, int K, const float *A, int lda, const float *B, int ldb, float *C, int ldc, pdgn_stream_t stream);
/* A weight matrix split ONCE into the three bf16 parts the contractions multiply (x = h + m + l, csrc/split.hip) instead of by
 * every workgroup's loader: src (rows x cols fp32, pitch ld_src) -> planes (3 x [rows][ld_planes] bf16, plane_stride elements
 * apart; may be NULL) and / or planes_t, the same for the TRANSPOSE (3 x [cols][ld_planes_t]; may be NULL).
 * pdgn_gemm_nt_ps = pdgn_gemm_nt_ex with such planes as the second operand (n x k, pitch ldw, wplane elements between planes;
 * ldw and wplane multiples of 8, the planes 16-byte aligned):
 * forward y = x W^T with W's planes, input gradient dX = dY W with the planes of W^T.  parts = 3 (these planes) or 2 (the two
 * fp16 planes of pdgn_split_f16x2, below).  Bit-identical to the unsplit call that runs on the same number of parts; matrix-core
 * modes only (PDGN_ERR_INVALID under pdgn_gemm_set_mode(0)).  No reference counterpart. */
int pdgn_split_bf16x3(int rows, int cols, const float *src, int ld_src, unsigned short *planes, int ld_planes,
                      long long plane_stride, unsigned short *planes_t, int ld_planes_t, long long plane_stride_t,
                      pdgn_stream_t stream);
int pdgn_gemm_nt_ps(long long m, int n, int k, const float *A, int lda, const unsigned short *Wplanes, int ldw, long long wplane,
                    int parts, const float *bias, const float *addend, int ldadd, float *C, int ldc, float *stat_part,
                    const float *row_bias, int ld_rb, int rows_per_group, int act, const float *gate, int ldgate,
                    pdgn_stream_t stream);
/* Stream-K tails without atomics (matrix-core mode).  A launch whose tiles do not fill the last round of workgroups finishes with a
 * tail that splits the leftover tiles' k range over the CUs.  With a workspace of pdgn_gemm_tail_workspace_floats(m, n, k, with_stats)
 * floats handed over by pdgn_gemm_set_tail_workspace -- to the NEXT pdgn_gemm_nt / _nn / _nt_ps call of the calling thread, which
 * consumes it -- the tail stores partial tiles there and a small kernel adds them into C (deterministic, no zero-fill); without one
 * it adds them with fp32 atomics.  The buffer must stay valid until that call's launches have run.  No reference counterpart. */
long long pdgn_gemm_tail_workspace_floats(long long m, int n, int k, int with_stats);
/* The same for pdgn_gemm_tn_big (dW = dY^T X: few output tiles, the reduction over the m rows split in k slices): with a workspace of
 * this many floats handed over the same way the slices' partial tiles are summed in a fixed order (no atomics, no zero-fill of dW,
 * bit-identical from run to run); 0: the call does not split. */
long long pdgn_gemm_tn_big_workspace_floats(long long m, int n, int k);
int pdgn_gemm_set_tail_workspace(float *ws, long long floats);
/* The kernel instance (host symbol; NULL for the 16x16x32 arm) and grid of the plain data-parallel launch of pdgn_gemm_nt_ps(m, n, k)
 * under the switches in force: for measurements that look that launch up in a recorded iteration.  Host-side only. */
int pdgn_gemm_nt_ps_launch_info(long long m, int n, int k, int parts, const void **sym, int *grid);
/* Host symbols of the small kernels a contraction call may launch around its matrix-core kernels: the reduce of a stream-K tail's
 * partial tiles, the scan of an operand's maxima (mode 2). */
int pdgn_gemm_aux_symbols(const void **reduce, const void **scan);
/* Process-wide switches of the dense contractions (read from PDGN_GEMM / PDGN_NT_CFG once, at first use).
 * pdgn_gemm_set_mode: 2 = matrix cores, two scaled fp16 parts where that pays and three bf16 parts elsewhere (default; below),
 * 1 = three bf16 parts everywhere, 0 = fp32 matrix instructions, < 0 = query; returns the previous mode.
 * pdgn_gemm_set_config: -1 = the launch model's pick (default), 0 .. 3 = force a tile configuration (measurement / tests),
 * < -1 = query; returns the previous value.
 * pdgn_gemm_set_shape: the bf16 matrix instruction of the matrix-core mode: 32 = v_mfma_f32_32x32x16_bf16 for every launch, 16 =
 * v_mfma_f32_16x16x32_bf16 for every launch (same tiles, same operands; results differ in the last bits), -1 = the built-in choice
 * per instance class, 0x1000 | mask = a per-class mask (bit 4 * tile + class; gemm_x3.hip), anything else = query; returns the
 * previous mask.  PDGN_X3_SHAPE / PDGN_X3_SHAPE16_MASK set the process default. */
int pdgn_gemm_set_mode(int mode);
/* Mode 2 (round 5; the default, PDGN_GEMM=x2): the contractions whose time is their matrix-core work run on the fp16 matrix
 * cores with TWO parts per value; the others stay on three bf16 parts (mode 1's arithmetic).  pdgn_gemm_two_part(m, n, k,
 * scan_bytes) says which a call (m, n, k) is when scan_bytes of its operands would still have to be scanned for their maxima: the
 * 256 x 128 tile, k >= 128, >= 20 GFLOP and at most 4.5 bytes to scan per kflop (measured: tools/x2_shapes.py).
 * Two parts (round 6: scaled PER ROW; one scale per operand until then): every ROW of each operand as the kernel sees it -- row m
 * of A, row n of W; for an operand given transposed the COLUMNS of the matrix in memory: pdgn_gemm_nn's Wt, both operands of
 * pdgn_gemm_tn_big -- is multiplied by a power of two 2^e_r, e_r = 14 - floor(log2 max |x| over that row) (so that nothing leaves
 * fp16's range), split as x 2^e = h + l (round-to-nearest fp16 of the value and of the exact remainder: |err| <= 2^-23 |x| for
 * values within 2^-16 of THEIR ROW's largest, an absolute 2^-39 of the row's maximum below; mode 1 needs no scale and keeps every
 * value's own 24 bits), a product is the three fp16 MFMA products al wh + ah wl + ah wh accumulated in fp32, and C[m, n] is
 * multiplied by 2^-e_A[m] 2^-e_W[n] (exact).  An output row therefore depends on its own input row's scale only: the input gradient
 * of a point with a small output gradient is as accurate, relative to that row, as any other's (tests/test_gpu_deconv.py::
 * test_two_part_rows_spanning_thirty_binades).  Per product |err| <= ~2^-21 |a w| in the worst case; in sums the fp32 accumulation
 * all modes share dominates, of which this form does half as much: measured against fp64 BELOW the other two modes (bench.py
 * gemm_accuracy, tools/x2_check.py: all three operand layouts, 1e-20 .. 1e15).  Half the matrix-core work of mode 1.
 * The maxima are arrays of bit patterns of |x| (uint32; unsigned order = magnitude order: producers combine them with integer /
 * atomic max in any order).  They come from a scan in front of the launch (x2_maxima_kernel: one pass, row and / or column
 * maxima) into an arena the caller provides once -- pdgn_gemm_set_scale_slots(device memory, bytes; handed out round-robin in 1-KB
 * units, 4 bytes per kernel-side row: it must hold the arrays of every launch that can be in flight; the library never
 * allocates); without one a two-part call that has to scan returns PDGN_ERR_INVALID -- or from the caller: pdgn_absmax_rows_cols
 * (rowmax[rows] and / or colmax[cols] of a matrix in one pass), or the kernel that WRITES the operand
 * (pdgn_bn_softmax_slots_permute_mul's and pdgn_window_gather_sum_backward_csr's max_out: row maxima), handed over for the operands of
 * the calling thread's NEXT contraction call (pdgn_gemm_set_operand_scales(max_a, max_w): the kernel-side rows of the first /
 * second operand as that entry point takes them -- _nt / _nt_ps: rows of A, rows of W; _nn: rows of A, columns of Wt; _tn_big:
 * columns of dY, columns of X; NULL = scanned by the call; consumed by that call; finite upper bounds serve as well): an
 * activation that feeds several products is scanned once, or never.
 * pdgn_split_f16x2 = pdgn_split_bf16x3 for a two-part product: two fp16 planes (h | l, every row scaled by its own power of two) and
 * the rows' maxima as uint32[rows] right behind them (element offset 2 * plane_stride: the buffer holds 2 * plane_stride + 2 * rows
 * elements; plane_stride >= rows * ld_planes, a multiple of 8); the transposed planes' rows are the matrix's columns.
 * pdgn_gemm_nt_ps takes either kind of planes and is told which (parts = 3 | 2).  No reference counterpart.
 * State: the mode / tile / instruction switches are process-wide; the arena is process-wide; the hand-overs (operand maxima, tail
 * workspace) are THREAD-LOCAL and belong to the calling thread's next contraction call only, whether it launches or is refused. */
int pdgn_gemm_set_scale_slots(void *slots, long long bytes);
int pdgn_absmax_rows_cols(long long rows, int cols, const float *src, int ld, unsigned *rowmax, unsigned *colmax,
                          pdgn_stream_t stream);
int pdgn_gemm_set_operand_scales(const unsigned *max_a, const unsigned *max_w);
int pdgn_gemm_two_part(long long m, int n, int k, long long scan_bytes);
/* The same question for a product against PRE-SPLIT planes (which kind to make: pdgn_split_f16x2 or _bf16x3): with nothing to
 * convert for the weight the two-part loop is ahead from ~2 GFLOP on, k >= 32; two-part planes run on the 256 x 128 tile whatever
 * pdgn_gemm_nt_config says for three parts (a launch with stat_part: only where that IS the pick, else PDGN_ERR_INVALID), and their
 * tail workspace is pdgn_gemm_nt_ps_workspace_floats(m, n, k, parts, with_stats). */
int pdgn_gemm_two_part_planes(long long m, int n, int k, long long scan_bytes);
/* Round 6: a product on two-part planes with a short reduction (k = 32, 64, 128), n >= 128, m >= 4096, m n >= 1.6e7 and neither
 * bias, addend nor epilogue extras runs on the ROW-PANEL kernel (csrc/gemm_rp.hip: a workgroup keeps 256 rows of A in registers as
 * matrix-instruction fragments and streams the weight's column tiles; same arithmetic as the tile kernel, bit for bit; no tail, no
 * workspace).  Its BatchNorm partials (stat_part) cover a 256-row panel each (the eight waves of a workgroup join their 32-row sums): pdgn_gemm_nt_ps_stat_rows / _stat_block_rows answer for the
 * call that will actually run (plain != 0: no bias / addend / extras). */
long long pdgn_gemm_nt_ps_stat_rows(long long m, int n, int k, int parts, int plain);
int pdgn_gemm_nt_ps_stat_block_rows(long long m, int n, int k, int parts, int plain);
/* 1 when that call runs on the row-panel kernel (the rule above under the switches in force), else 0: a caller skips the scan of
 * A's row maxima then -- the kernel takes them itself. */
int pdgn_gemm_nt_ps_row_panel(long long m, int n, int k, int parts, int plain);
long long pdgn_gemm_nt_ps_workspace_floats(long long m, int n, int k, int parts, int with_stats);
int pdgn_split_f16x2(int rows, int cols, const float *src, int ld_src, unsigned short *planes, int ld_planes,
                     long long plane_stride, unsigned short *planes_t, int ld_planes_t, long long plane_stride_t,
                     pdgn_stream_t stream);
int pdgn_gemm_set_config(int cfg);
int pdgn_gemm_set_shape(int shape);
int pdgn_gemm_nt(long long m, int n, int k, const float *A, int lda, const float *W, int ldw,
                 const float *bias, const float *addend, int ldadd, float *C, int ldc, float *stat_part,
                 pdgn_stream_t stream);
/* Dense layers with at most 4 channels on one side (the xyz-in / xyz-out layers: Conv1d(3, 64) of the discriminators
 * and conv_xyz, models/PDGNet_v2.py:559-566, 886-1014; Conv1d(64, 3) of the heads, :835-862) as streaming kernels, no padding:
 *   Y (m x n, pitch ldy) = X (m x k, pitch ldx) W'^T (+ bias[n]),   W'[j, kk] = W[j * wrs + kk * wcs]
 * with either k <= 4 and n % 4 == 0 (Y 16-byte aligned rows) or n <= 4 and k % 4 == 0 (X 16-byte aligned rows).  The strides
 * let one entry serve a weight and its transpose (forward: wrs = k, wcs = 1; input gradient dX = dY W: wrs = 1, wcs = C_in).
 * stat_part (k <= 4 form only, may be NULL): pdgn_thin_stat_rows(m) rows of [3n] floats (blocks of pdgn_thin_stat_block_rows()
 * rows), the BatchNorm partials of Y in the layout pdgn_bn_stats_from_gemm_partials reads.  Returns -3 for shapes outside
 * these two forms. */
long long pdgn_thin_stat_rows(long long m);
int pdgn_thin_stat_block_rows(void);
int pdgn_thin_nt(long long m, int n, int k, const float *X, int ldx, const float *W, int wrs, int wcs,
                 const float *bias, float *Y, int ldy, float *stat_part, pdgn_stream_t stream);
/* The same with a LeakyReLU-derivative factor (k <= 4 form): Y *= (gate > 0 ? 1 : 0.01), gate (m x n, pitch ldgate) a saved
 * activation -- Y is then the gradient wrt that activation's input (backward of the heads, models/PDGNet_v2.py:835-862). */
int pdgn_thin_nt_ex(long long m, int n, int k, const float *X, int ldx, const float *W, int wrs, int wcs,
                    const float *bias, float *Y, int ldy, float *stat_part, const float *gate, int ldgate, pdgn_stream_t stream);
/* Weight (+ bias) gradient of such a layer: O[i * osi + j * osj] += sum_r A[r, i] B[r, j] for A (m x ta <= 4, pitch lda),
 * B (m x wb, wb % 4 == 0, pitch ldb, 16-byte aligned rows); sum_a[ta] += column sums of A, sum_b[wb] += column sums of B
 * (either may be NULL).  O and the sums are accumulated with fp32 atomics: the caller passes them zero-filled. */
int pdgn_thin_tn(long long m, int ta, int wb, const float *A, int lda, const float *B, int ldb, float *O, int osi,
                 int osj, float *sum_a, float *sum_b, pdgn_stream_t stream);
/* The same product with the second operand given transposed, C = A Wt with Wt (k x n, row pitch ldw): the input
 * gradient dX = dY W of a dense layer straight from its (C_out x C_in) weight (models/PDGNet_v2.py conv / linear backward). */
int pdgn_gemm_nn(long long m, int n, int k, const float *A, int lda, const float *Wt, int ldw,
                 const float *bias, const float *addend, int ldadd, float *C, int ldc, float *stat_part,
                 pdgn_stream_t stream);
/* Weight gradient on the same kernel (both operands transposed, reduction over the m rows split over the workgroups):
 * dW (n x k) = dY (m x n, pitch ldy)^T X (m x k, pitch ldx); dW is zero-filled by the call.  Meant for outputs of at
 * least one 128 x 64 tile (pdgn_gemm_tn keeps the small ones). */
int pdgn_gemm_tn_big(long long m, int n, int k, const float *dY, int ldy, const float *X, int ldx, float *dW, int dw_is_zero,
                     pdgn_stream_t stream);
/* dw_is_zero != 0: the caller hands over an all-zero dW (e.g. a slice of one zero-filled arena per backward pass): the
 * split-K launch then adds into it without a zero-fill launch of its own. */
/* pdgn_gemm_nt / pdgn_gemm_nn (transposed_w != 0) with an extended epilogue, applied after bias / addend in this order:
 *   + row_bias[(row / rows_per_group) * ld_rb + col]   a bias per group of rows (the per-sample term of the generator's heads:
 *                                                      mlp1..4 on cat([g broadcast, x]), models/PDGNet_v2.py:835-862, 868-876)
 *   LeakyReLU(0.01) on the result (act = 2; 0 = none)
 *   * (gate[row, col] > 0 ? 1 : 0.01)                  the LeakyReLU derivative of a saved activation: C is then the gradient
 *                                                      wrt that layer's pre-activation
 * each replacing an elementwise pass over C.  Whole tiles only (no stream-K tail).  row_bias / gate may be NULL. */
int pdgn_gemm_nt_ex(long long m, int n, int k, const float *A, int lda, const float *W, int ldw, const float *bias,
                    const float *addend, int ldadd, float *C, int ldc, float *stat_part, const float *row_bias, int ld_rb,
                    int rows_per_group, int act, const float *gate, int ldgate, int transposed_w, pdgn_stream_t stream);
long long pdgn_gemm_nt_stat_rows(long long m, int n, int k);
int pdgn_gemm_nt_stat_block_rows(long long m, int n, int k);
/* Tile configuration the launch model picks (0: 256x128, 1: 128x128, 2: 160x64, 3: 128x64 workgroup tiles); host only. */
int pdgn_gemm_nt_config(long long m, int n, int k, int with_stats);

/* Weight gradient of a point-major dense layer on the fp32 matrix cores, reduction split over
 * workgroups:  dW (n x k) += dY (m x n)^T  X (m x k), all row-major, m >> n, k.
 * dW must be zero-filled by the caller; n % 4 == 0, k % 4 == 0. */
int pdgn_gemm_tn(long long m, int n, int k, const float *dY, const float *X, float *dW,
                 pdgn_stream_t stream);

/* ------------------------------------------------------------------ local-pair shape loss
 * Neighbourhood mean / covariance of models/PDGNet_v2.py:127-134 fused with the grouping of
 * :142-147: xyz (b,n,3), idx (b,m,k) (from pdgn_knnquery) -> mu (b,m,3), cov (b,m,9)
 * (cov = (1/k) sum_s (p_s - mu)(p_s - mu)^T). */
int pdgn_local_stats(int b, int n, int m, int k, const float *xyz, const int32_t *idx, float *mu,
                     float *cov, pdgn_stream_t stream);
/* Adjoint: dxyz (b,n,3) += scatter of dmu (b,m,3), dcov (b,m,9); dxyz zero-filled by the caller. */
int pdgn_local_stats_backward(int b, int n, int m, int k, const float *xyz, const int32_t *idx,
                              const float *dmu, const float *dcov, float *dxyz, pdgn_stream_t stream);

/* utils/chamfer_loss.py:13-38 without the (B,M,N) matrix: P[i,j] = (|x_i|^2 + |y_j|^2) - 2<x_i,y_j>
 * (Gram form, no clamp); x (b,m,d), y (b,n,d), d <= 16 -> minx/argx (b,m) = min/argmin over j,
 * miny/argy (b,n) = min/argmin over i (lowest index on ties). */
int pdgn_chamfer_gram(int b, int m, int n, int d, const float *x, const float *y, float *minx,
                      int32_t *argx, float *miny, int32_t *argy, pdgn_stream_t stream);
/* Gradient of sum(gminx*minx) + sum(gminy*miny) wrt x and y (zero-fills gx, gy itself). */
int pdgn_chamfer_gram_grad(int b, int m, int n, int d, const float *x, const float *y,
                           const float *gminx, const int32_t *argx, const float *gminy,
                           const int32_t *argy, float *gx, float *gy, pdgn_stream_t stream);
/* The same for the loss scale * (sum minx + sum miny) (utils/chamfer_loss.py:16-20) with upstream gradient g[0] (a device
 * scalar): every minimum's gradient is g[0] * scale, no expanded gradient tensor.  When gy == gx + b*m*d one fill zeroes both. */
int pdgn_chamfer_gram_grad_uniform(int b, int m, int n, int d, const float *x, const float *y, const float *g,
                                   float scale, const int32_t *argx, const int32_t *argy, float *gx, float *gy,
                                   pdgn_stream_t stream);
/* Scalar ends of the step's losses, one launch each way (csrc/loss_small.hip; fixed summation order):
 * pdgn_scaled_sum: out[0] = scale * sum_i x[i] (the Chamfer minima, chamfer_loss.py:16-20);
 * pdgn_mse_const: out[0] = scale * mean_i (x[i] - target)^2 -- nn.MSELoss against a constant, the adversarial terms
 * mse(D(x), 1) / mse(D(x), 0) of models/PDGNet_v2.py:186-190, 246-250 (scale: their 1/2);
 * pdgn_mse_const_backward: dx[i] = g[0] * scale * 2 (x[i] - target) / n. */
int pdgn_scaled_sum(long long n, const float *x, float scale, float *out, pdgn_stream_t stream);
int pdgn_mse_const(long long n, const float *x, float target, float scale, float *out, pdgn_stream_t stream);
int pdgn_mse_const_backward(long long n, const float *x, float target, float scale, const float *g, float *dx,
                            pdgn_stream_t stream);

/* All-pairs evaluation (evaluation/evaluation_metrics.py:85-121 expands every sample against every
 * reference batch): the same kernels over explicit pair lists, pair p = (cloud ia[p] of the first
 * tensor, cloud ib[p] of the second); outputs are indexed by p.  temp: npairs*2*(n+m) floats. */
int pdgn_emd_cost_indexed(int npairs, int n, int m, const float *xyz1, const int32_t *ia,
                          const float *xyz2, const int32_t *ib, float *temp, float *out,
                          pdgn_stream_t stream);
int pdgn_chamfer_gram_indexed(int npairs, int m, int n, int d, const float *x, const int32_t *ia,
                              const float *y, const int32_t *ib, float *minx, int32_t *argx,
                              float *miny, int32_t *argy, pdgn_stream_t stream);

/* Weight re-association of a point-deconvolution block and its adjoint, one launch each:
 * Wi = inte_conv_hk.0.weight (4F,2F,1,T), W2 = conv2.conv.weight (2Fo,2F,1,2k), Wf = conv_fea.0.weight (16,2F,1,1) or
 * NULL -> the per-point GEMM operand Wcat (Mw x F) split into its first fc columns (WcatC, NULL when fc == 0) and the
 * rest (WcatV), and conv2's dense operand Wb (2Fo, (k-T+1)*4F).  Mw = T*4F + 4F + k*2Fo + 2Fo (+ 32 with Wf).
 * Backward: any of the three upstream gradients may be NULL (= zero); dWf NULL for blocks without conv_fea. */
int pdgn_deconv_assemble(int F, int Fo, int k, int T, int fc, const float *Wi, const float *W2, const float *Wf,
                         float *WcatC, float *WcatV, float *Wb, pdgn_stream_t stream);
int pdgn_deconv_assemble_backward(int F, int Fo, int k, int T, int fc, const float *gWcatC, const float *gWcatV,
                                  const float *gWb, float *dWi, float *dW2, float *dWf, pdgn_stream_t stream);

/* Per-sample biases of a block's gather-sums under the constant-channel split: for spec i = (T_i, C_i, off_i, offc_i)
 *   bb[b, o_i + c] = bias_i[c] + Yc[b, offc_i + c] + sum_{t<T_i} Yc[b, off_i + t*C_i + c],   o_i = C_0 + .. + C_{i-1},
 * Yc (b, ldy); nspec <= 4; bias[i] may be NULL.  Adjoint: g (b, sum C_i) -> dYc (b, ldy), dbias[i] (C_i) or NULL. */
int pdgn_sample_bias(int b, int ldy, int nspec, const int *T, const int *C, const int *off, const int *offc,
                     const float *const *bias, const float *Yc, float *bb, pdgn_stream_t stream);
int pdgn_sample_bias_backward(int b, int ldy, int nspec, const int *T, const int *C, const int *off, const int *offc,
                              const float *g, float *dYc, float *const *dbias, pdgn_stream_t stream);

/* nn.Sequential(Linear, [BatchNorm1d,] [LeakyReLU | ReLU]) on r <= 64 rows in one launch (the generator's per-sample
 * global branches and first layer, models/PDGNet_v2.py:704-707, 825-828): x (r,k), W (n,k), y (r,n); k <= 1024.
 * bn_mode 0: no BatchNorm; 1: batch statistics (running_mean / running_var updated when given); 2: running statistics.
 * pre (r,n) and stat (2n: mean | invstd) are written for the adjoint (may be NULL).  Backward: dpre (r,n) = gradient
 * wrt the linear output (the caller forms dx = dpre W), dgamma, dbeta, dbias (n), dW (n,k); any of those four may be NULL. */
int pdgn_small_mlp_forward(int r, int k, int n, int act, int bn_mode, float eps, float momentum, const float *x,
                           const float *W, const float *bias, const float *gamma, const float *beta,
                           float *running_mean, float *running_var, float *y, float *pre, float *stat,
                           pdgn_stream_t stream);
int pdgn_small_mlp_backward(int r, int k, int n, int act, int bn_mode, const float *x, const float *dy, const float *pre,
                            const float *stat, const float *gamma, const float *beta, float *dpre, float *dgamma,
                            float *dbeta, float *dbias, float *dW, pdgn_stream_t stream);

/* ---- pointops entry points PDGN itself never calls, same argument meaning as the
 * reference launchers; index outputs int32, label statistics int32, caller allocates (and zero-fills where the
 * reference's Python does). */
/* ballquery_cuda_launcher_fast (ballquery/ballquery_cuda_kernel.h:10): first <= nsample points with d2 < r^2 in
 * index order; the first hit fills every slot; an empty ball leaves idx untouched (caller zero-fills, pointops.py:190). */
int pdgn_ballquery(int b, int n, int m, float radius, int nsample, const float *new_xyz, const float *xyz,
                   int32_t *idx, pdgn_stream_t stream);
/* furthestsampling_cuda_launcher (sampling/sampling_cuda_kernel.h:13): xyz (b,n,3), temp (b,n) pre-filled with 1e10
 * (pointops.py:24), idx (b,m); starts at point 0; exact ties go to the lowest index. */
int pdgn_furthestsampling(int b, int n, int m, const float *xyz, float *temp, int32_t *idx, pdgn_stream_t stream);
/* gathering_{forward,backward}_cuda_launcher (sampling_cuda_kernel.h:10-11), also featuregather
 * (featuredistribute_cuda_kernel.h:12-13): out[b,c,j] = points[b,c,idx[b,j]]; the adjoint accumulates (+=). */
int pdgn_gathering_forward(int b, int c, int n, int m, const float *points, const int32_t *idx, float *out,
                           pdgn_stream_t stream);
int pdgn_gathering_backward(int b, int c, int n, int m, const float *grad_out, const int32_t *idx,
                            float *grad_points, pdgn_stream_t stream);
/* grouping_int_forward_cuda_launcher_fast (grouping_int/grouping_int_cuda_kernel.h:10): int64 features. */
int pdgn_grouping_int_forward(int b, int c, int n, int m, int nsample, const long long *points, const int32_t *idx,
                              long long *out, pdgn_stream_t stream);
/* featuredistribute_cuda_launcher (featuredistribute_cuda_kernel.h:10): nearest max_xyz (b,n,3) point of every
 * xyz (b,m,3) point. */
int pdgn_featuredistribute(int b, int n, int m, const float *max_xyz, const float *xyz, int32_t *distribute_idx,
                           pdgn_stream_t stream);
/* labelstat_* launchers (labelstat/labelstat_cuda_kernel.h:10-17). */
int pdgn_labelstat_idx(int b, int n, int m, int nsample, int nclass, const int32_t *label_stat, const int32_t *idx,
                       int32_t *new_label_stat, pdgn_stream_t stream);
int pdgn_labelstat_ballrange(int b, int n, int m, float radius, int nclass, const float *new_xyz, const float *xyz,
                             const int32_t *label_stat, int32_t *new_label_stat, pdgn_stream_t stream);
int pdgn_labelstat_and_ballquery(int b, int n, int m, float radius, int nsample, int nclass, const float *new_xyz,
                                 const float *xyz, const int32_t *label_stat, int32_t *idx, int32_t *new_label_stat,
                                 pdgn_stream_t stream);

/* ------------------------------------------------------------------ max-pool over the points
 * nn.MaxPool2d((1, N)) at the head of every generator block (models/PDGNet_v2.py:675/:699, :716/:736, :754/:777,
 * :794/:810) on the point-major layout: x (b,n,c) -> out (b,c) and the point index of each maximum (lowest index on
 * ties), which is all the adjoint needs: grad_x[b,r,ch] = (r == arg[b,ch]) ? grad_out[b,ch] : 0 (c % 4 == 0).
 * scratch_val / scratch_arg hold pdgn_point_max_scratch(b,n,c) floats / int32 each. */
long long pdgn_point_max_scratch(int b, int n, int c);
int pdgn_point_max(int b, int n, int c, const float *x, float *scratch_val, int32_t *scratch_arg, float *out, int32_t *arg,
                   pdgn_stream_t stream);
int pdgn_point_max_backward(int b, int n, int c, const float *grad_out, const int32_t *arg, float *grad_x,
                            pdgn_stream_t stream);

/* ------------------------------------------------------------------ scheduling support
 * No reference counterpart (the reference runs on one CUDA stream).  One wavefront that occupies `stream` for
 * `microseconds` (<= 100000) of wall time: pdgn_amd/streams.py times pairs of these to learn which HIP streams share a
 * hardware queue -- streams on one queue serialise, and the stream-overlapped G+D schedule keeps the default stream's
 * queue to itself. */
int pdgn_spin(unsigned int microseconds, pdgn_stream_t stream);

/* ------------------------------------------------------------------ launch-list replay (csrc/replay.hip)
 * No reference counterpart (the reference issues its iteration op by op from Python, models/PDGNet_v2.py:171-256).
 * A captured hipGraph_t of the iteration is read back node by node and re-issued with plain launches on caller-chosen
 * streams: pdgn_replay_marker(id, stream) inside the capture tags a stream; pdgn_replay_build turns the graph into a plan
 * (kernel / memset / flat device-to-device memcpy / empty nodes only: anything else returns a negative code); chains =
 * runs of nodes that were captured on one stream (pdgn_replay_chains: marker id or -1, and length, per chain);
 * pdgn_replay_set_stream binds a chain to a hipStream_t; pdgn_replay_launch issues every node once, in capture order,
 * with event pairs for the dependencies that cross chains.  counts8 = nodes, kernels, memsets, memcpys, empty nodes,
 * chains, events, labelled chains.  The graph (and the memory its launches address) must outlive the plan. */
int pdgn_replay_marker(int id, pdgn_stream_t stream);
int pdgn_replay_build(void *hip_graph, void **plan_out);
int pdgn_replay_info(void *plan, int *counts8);
int pdgn_replay_chains(void *plan, int *labels, int *nodes_per_chain);
int pdgn_replay_set_stream(void *plan, int chain, pdgn_stream_t stream);
int pdgn_replay_launch(void *plan);
int pdgn_replay_launch_range(void *plan, int lo, int hi); /* nodes [lo, hi) of the list */
int pdgn_replay_position(void *plan, int chain, int nth); /* list position of a chain's n-th node, or -1 */
/* Markers with id >= 64 are HOST POINTS, not stream tags: places of the list at which the caller does what a capture cannot
 * record (the RCCL all-reduces of the data-parallel iteration).  Returns their number; ids / list positions / chains of the first
 * max_out in list order.  The caller issues [lo, pos + 1), makes its own call on that chain's stream, and goes on. */
int pdgn_replay_points(void *plan, int *ids, int *pos, int *chain, int max_out);
int pdgn_replay_joined(void *plan); /* 1: the chain of marker 0 ends behind every other chain's last node */
/* In-iteration timing: pdgn_replay_kernel_nodes finds the list positions of the nodes launched through a host symbol (a kernel
 * instance, e.g. from pdgn_gemm_nt_ps_launch_info) with a given grid.x (< 0: any); pdgn_replay_chain_neighbor steps along a node's
 * chain (the memset in front of a stream-K launch, the tail kernel behind it); pdgn_replay_time_spans brackets spans [first, last] of
 * one chain with two timing events on that chain's stream for the next `slots` passes; pdgn_replay_time_read waits for them and
 * returns ms_out[i * slots + j] for j < counts[i].  bench.py measures its roofline kernel this way, inside the timed steps. */
int pdgn_replay_kernel_nodes(void *plan, const void *sym, int gx, int *pos, int max_out);
int pdgn_replay_chain_neighbor(void *plan, int pos, int dir, int *kind_out, const void **sym_out);
int pdgn_replay_time_spans(void *plan, const int *first, const int *last, int n, int slots);
int pdgn_replay_time_read(void *plan, float *ms_out, int *counts);
int pdgn_replay_launch_timed(void *plan, double *us32); /* measurement: host microseconds per call kind / chain */
int pdgn_replay_probe_chain(void *plan, int chain, int stride, float *ms_out, int *pos_out, int max_out); /* measurement: device-time progress of one chain */
int pdgn_replay_destroy(void *plan);

/* ------------------------------------------------------------------ optimiser
 * One Adam step (lr, betas, eps; no weight decay / amsgrad / maximize -- the reference's five torch.optim.Adam, models/PDGNet_v2.py:
 * 121-125, 186-226, 256) of a whole list of fp32 tensors: replaces torch._fused_adam_ (five launches of 64 K-element chunks, 230 us
 * for the generator's 12.7 M parameters).  p, g, m, v: HOST arrays of ntensors device pointers (parameter, gradient, first and
 * second moment; 4-byte aligned, 16-byte aligned ones take the vector path), n: their element counts; the pointers travel in the
 * kernel arguments (72 tensors per launch, one workgroup per 4096 elements).  step (device): the count t >= 1 of THIS update as one
 * float.  torch's arithmetic, bit for bit: m <- beta1 m + (1 - beta1) g; v <- beta2 v + (1 - beta2) g g; p <- p - lr / (1 - beta1^t) * m /
 * (sqrt(v) / sqrt(1 - beta2^t) + eps), bias corrections and 1 - beta in fp64 (torch keeps lr and the betas as doubles). */
int pdgn_adam_multi(int ntensors, void *const *p, const void *const *g, void *const *m, void *const *v, const long long *n, double lr,
                    double beta1, double beta2, double eps, const float *step, pdgn_stream_t stream);
/* dst[i] (n[i] floats) <- src[i] for a list of fp32 tensors, the same way (128 tensors per launch): the pack of a network's fresh
 * gradients into the flat buffer of its one all-reduce (the gradient reduction of nn.DataParallel, models/PDGNet_v2.py:101-105);
 * replaces torch._foreach_copy_. */
int pdgn_copy_multi(int ntensors, void *const *dst, const void *const *src, const long long *n, pdgn_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* PDGN_HIP_H */
