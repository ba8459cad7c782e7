/*
 * fpsg_hip.h -- C ABI of libfpsg_hip.so, the MI355X (gfx950) hot-path library.
 *
 * The reference (voidstrike/FPSG) has no native code of its own; its hot ops live in
 * third-party CUDA packages reached through Python imports.  Each entry point below
 * names the reference call site it replaces (paths relative to /root/reference).
 *
 * Conventions:
 *   - every pointer is a DEVICE pointer to a contiguous, 4-byte aligned buffer owned by
 *     the caller; the library never allocates, frees or keeps device memory;
 *   - a call only enqueues work on `stream` and returns (asynchronous);
 *   - return value 0 = success; >0 = hipError_t from the launch; <0 = argument check
 *     (FPSG_E_*); fpsg_last_error() gives a thread-local message for the last failure;
 *   - no C++ exception crosses the boundary; no global state besides that message (tuning variants
 *     are explicit arguments, never process-wide settings).
 *   - index outputs are int32 (the Python mirror widens to int64 where the reference
 *     API exposes int64).
 */
#ifndef FPSG_HIP_H
#define FPSG_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* hipStream_t without dragging hip headers into C callers */
typedef void* fpsg_stream_t;

#define FPSG_ABI_VERSION 1

#define FPSG_E_NULL   (-1)  /* null pointer argument            */
#define FPSG_E_SHAPE  (-2)  /* non-positive / unsupported shape */
#define FPSG_E_ALIGN  (-3)  /* pointer not 4-byte aligned       */
#define FPSG_E_LIMIT  (-4)  /* size beyond a documented limit   */

int         fpsg_version(void);
const char* fpsg_last_error(void);

/* ---- K1: Chamfer sided distances ------------------------------------------------
 * Replaces kaolin.metrics.pointcloud.chamfer_distance (Kaolin 0.9.0, its CUDA
 * `sided_distance` forward), bound at src/models/few_shot.py:13,57 and called at
 * src/models/few_shot.py:110,117,167.
 *
 * xyz1 [B,N,3], xyz2 [B,M,3] fp32.  For every point of xyz1 the squared L2 distance to
 * its nearest point of xyz2 and that point's index (lowest index on ties), and vice
 * versa:  dist1,idx1 [B,N];  dist2,idx2 [B,M].
 * d(i,j) = fma(dz,dz, fma(dy,dy, dx*dx)),  dx = xyz2[j].x - xyz1[i].x  (fp32, RN).
 */
int fpsg_chamfer_fwd(const float* xyz1, const float* xyz2, int B, int N, int M,
                     float* dist1, int32_t* idx1, float* dist2, int32_t* idx2,
                     fpsg_stream_t stream);

/* The same with an explicit kernel variant (micro-benchmarks, tests): (queries per lane, waves per
 * workgroup) 0=(1,16) 1=(2,16) 2=(4,8) 3=(8,4) 4=(4,4) 5=(2,8) 6=(2,4); -1 = automatic.  Results do
 * not depend on it. */
int fpsg_chamfer_fwd_variant(const float* xyz1, const float* xyz2, int B, int N, int M,
                             float* dist1, int32_t* idx1, float* dist2, int32_t* idx2,
                             int cfg, fpsg_stream_t stream);

/* One-pass form of the same op (the default of the Python mirror for clouds of at most 4096 points):
 * every d(i,j) is evaluated once and serves both directions (d is bit-symmetric); 2-D tiles of
 * 64*R rows x W*cpw*16 candidates leave 32-bit partial keys in the caller's workspace (a tile's exact minimum with
 * its six lowest mantissa bits replaced by where inside the tile it was found), a second launch re-evaluates the
 * named range of every tile whose truncated minimum equals the smallest one and so recovers the exact distances and
 * first-minimum indices.  Bit-identical results to fpsg_chamfer_fwd.
 * ws: fpsg_chamfer_workspace_bytes(B,N,M,variant) bytes, 8-byte aligned.  variant: -1 automatic,
 * else (R==8 ? 100 : 0) + 10*W + cpw with R in {4,8}, W in {1,2,4}, cpw in 1..9.  The workspace size
 * is 0 when this form does not apply: N or M > 4096, or (variant -1) fewer than ~7 pairs of
 * 2048-point clouds, where fpsg_chamfer_fwd's small workgroups fill the chip better.
 * fpsg_chamfer_fwd_tiled_losses: the same op plus K1l's three loss sums (fpsg_chamfer_losses below: same values, bit
 * for bit) -- the second launch also leaves every 256-point block's distance sum in the workspace, a one-workgroup
 * third launch adds them up; no separate pass over dist1 / dist2.  B <= 4096. */
size_t fpsg_chamfer_workspace_bytes(int B, int N, int M, int variant);
int fpsg_chamfer_fwd_tiled(const float* xyz1, const float* xyz2, int B, int N, int M,
                           float* dist1, int32_t* idx1, float* dist2, int32_t* idx2,
                           void* ws, size_t ws_bytes, int variant, fpsg_stream_t stream);
int fpsg_chamfer_fwd_tiled_losses(const float* xyz1, const float* xyz2, int B, int N, int M,
                                  float* dist1, int32_t* idx1, float* dist2, int32_t* idx2,
                                  void* ws, size_t ws_bytes, int variant, int n_first, float w_first,
                                  float w_rest, float* out3, fpsg_stream_t stream);

/* Backward of the two sided distances w.r.t. both clouds (Kaolin's
 * sided_distance backward, reached through autograd from
 * src/trainNetwork.py:144 `ttl_loss.backward()`).
 * g1 [B,N], g2 [B,M] are the upstream gradients of dist1/dist2.
 * gxyz1 [B,N,3], gxyz2 [B,M,3] are fully overwritten.  Deterministic (Kaolin scatters with float
 * atomics, i.e. in no defined order).  Order fixed here, per output point i of cloud a:
 *   own term   ga_i = (2 g_a[i]) (a_i - b[idx_a[i]]);
 *   the sources j with idx_b[j] == i, in ascending j, are cut into blocks of 32; block k is summed
 *   from +0 by S_k = fma(2 g_b[j], a_i - b_j, S_k); then ga_i += S_0, += S_1, ... in block order.
 * fpsg_chamfer_bwd picks between the two kernels below (same results):
 *   _sorted: N, M <= 4096; one workgroup per (pair, side) inverts the argmin list by sorting
 *            (target, source) keys in LDS -- time independent of how many sources share a target;
 *   _scan  : any size; every 256-point tile scans the other side's argmin list. */
int fpsg_chamfer_bwd(const float* xyz1, const float* xyz2,
                     const int32_t* idx1, const int32_t* idx2,
                     const float* g1, const float* g2,
                     int B, int N, int M,
                     float* gxyz1, float* gxyz2, fpsg_stream_t stream);
int fpsg_chamfer_bwd_sorted(const float* xyz1, const float* xyz2,
                            const int32_t* idx1, const int32_t* idx2,
                            const float* g1, const float* g2,
                            int B, int N, int M,
                            float* gxyz1, float* gxyz2, fpsg_stream_t stream);
int fpsg_chamfer_bwd_scan(const float* xyz1, const float* xyz2,
                          const int32_t* idx1, const int32_t* idx2,
                          const float* g1, const float* g2,
                          int B, int N, int M,
                          float* gxyz1, float* gxyz2, fpsg_stream_t stream);

/* K1l: the episode's reconstruction losses from the distances of one fpsg_chamfer_fwd* call over B cloud pairs of which
 * the first n_first are the query pairs (src/models/few_shot.py:110-124: chamfer_distance(...).sum() per group, then
 * query_factor * q + support_factor * s).  out3 = { sum_{b < n_first} cd_b, sum_{b >= n_first} cd_b,
 * w_first * out3[0] + w_rest * out3[1] },  cd_b = mean_i dist1[b,i] + mean_j dist2[b,j].  B <= 4096.  Deterministic;
 * the summation order (also fpsg_chamfer_fwd_tiled_losses's and the oracle's oracle_chamfer_losses):
 *   a row (dist1[b,:] or dist2[b,:]) is cut into blocks of 256 consecutive values, values past the end count as +0;
 *   inside a block each of the four groups of 64 consecutive values is summed by the balanced binary tree over the
 *   position in the group (pairs at distance 1, then 2, 4, ... 32), block = ((T0 + T1) + T2) + T3;
 *   the row's sum adds its blocks in ascending order starting from +0;  cd_b = s1 * (1/N) + s2 * (1/M) with the
 *   reciprocals rounded to fp32 first (how PyTorch's GPU mean divides; equal to a division when N is a power of two);
 *   a group's sum (the pairs below n_first / the rest; a pair outside the group counts as +0): 64 partial sums
 *   p_l = cd_l + cd_(l+64) + ... in ascending order from +0, then the same balanced tree over l = 0..63;
 *   the total is w_first * q + w_rest * r, unfused. */
int fpsg_chamfer_losses(const float* dist1, const float* dist2, int B, int N, int M, int n_first,
                        float w_first, float w_rest, float* out3, fpsg_stream_t stream);
/* Its backward: g1 [B,N], g2 [B,M] (fully overwritten) = the gradients of the three values with respect to dist1 /
 * dist2 given the upstream gradients g_first, g_rest, g_total (device scalars; null = none), ready for fpsg_chamfer_bwd:
 *   g1[b,:] = g * (1/N), g2[b,:] = g * (1/M) (fp32 reciprocals, the bits of autograd's mean backward on the GPU),
 *   g = g_total * (b < n_first ? w_first : w_rest) + (b < n_first ? g_first : g_rest). */
int fpsg_chamfer_loss_grads(const float* g_first, const float* g_rest, const float* g_total, int B, int N, int M,
                            int n_first, float w_first, float w_rest, float* g1, float* g2, fpsg_stream_t stream);
/* fpsg_chamfer_loss_grads + fpsg_chamfer_bwd_sorted in one launch (N, M <= 4096): the per-pair constants are formed
 * inside the backward kernel, the two constant [B,N] / [B,M] arrays are never written or read.  Same bits. */
int fpsg_chamfer_bwd_losses(const float* xyz1, const float* xyz2, const int32_t* idx1, const int32_t* idx2,
                            const float* g_first, const float* g_rest, const float* g_total,
                            int B, int N, int M, int n_first, float w_first, float w_rest,
                            float* gxyz1, float* gxyz2, fpsg_stream_t stream);

/* ---- K3: kNN graph ----------------------------------------------------------------
 * Replaces `knn(x, k)` of src/dgcnn/model.py:13-20 (torch.matmul into a [B,N,N] matrix +
 * torch.topk).  x [B,C,N] fp32 channel-major (the reference's layout); idx [B,N,k] int32: for every point the k
 * points with the largest  pd_ij = (-|x_j|^2 + 2 x_i.x_j) - |x_i|^2, nearest first (self
 * included), equal values -> lower index first.  ws: caller scratch of fpsg_knn_workspace_floats(B,C,N) floats,
 * 16-byte aligned (squared norms + a zero-padded, point-major, k-interleaved copy of x whose 16-byte pieces are the
 * MFMA operands of four channel steps).
 * Two kernels, identical results (bit for bit the oracle's):
 *   - C <= 128 and k <= 24: the streaming kernel (knn_stream.hip) -- a workgroup owns 128 queries, the cloud's
 *     candidates pass once through LDS, a row keeps only the scores above a running lower bound of its k-th best;
 *   - otherwise (C <= 440, k <= 64): the score-tile kernel (knn.hip) -- 16 queries per workgroup, their scores
 *     against 2048 candidates at a time in LDS, longer clouds in chunks whose sorted top-k lists are merged.
 * fpsg_knn_ex: `layout` says how x is stored -- FPSG_KNN_CHANNEL_MAJOR [B,C,N] or FPSG_KNN_POINT_MAJOR [B,N,C]
 * (what the fused EdgeConv layers produce; only where the streaming kernel serves, FPSG_E_LIMIT otherwise);
 * `flags`: FPSG_KNN_FORCE_TILE selects the score-tile kernel, FPSG_KNN_FORCE_SLOW makes every wave of the streaming
 * kernel take its slow exact path (the one a row buffer that cannot be compacted falls back to) -- both for tests
 * and A/B timing.  fpsg_knn(x, ...) = fpsg_knn_ex(x, FPSG_KNN_CHANNEL_MAJOR, ..., 0).
 */
#define FPSG_KNN_CHANNEL_MAJOR 0
#define FPSG_KNN_POINT_MAJOR   1
#define FPSG_KNN_FORCE_TILE    1
#define FPSG_KNN_FORCE_SLOW    2
size_t fpsg_knn_workspace_floats(int B, int C, int N);
int fpsg_knn(const float* x, int B, int C, int N, int k, int32_t* idx, float* ws,
             fpsg_stream_t stream);
int fpsg_knn_ex(const float* x, int layout, int B, int C, int N, int k, int32_t* idx, float* ws,
                int flags, fpsg_stream_t stream);

/* ---- K4a: EdgeConv edge features (materialising form) ------------------------------
 * Replaces `get_graph_feature(x, k, idx)` of src/dgcnn/model.py:23-42.
 * out [B,2C,N,k]: out[b,c,n,j] = x[b,c,idx[b,n,j]] - x[b,c,n];  out[b,C+c,n,j] = x[b,c,n].
 */
int fpsg_edge_feature_fwd(const float* x, const int32_t* idx, int B, int C, int N, int k,
                          float* out, fpsg_stream_t stream);

/* Gradient of the above w.r.t. x (gx [B,C,N], fully overwritten).  The neighbour terms
 * are accumulated with fp32 atomics. */
int fpsg_edge_feature_bwd(const float* gout, const int32_t* idx, int B, int C, int N, int k,
                          float* gx, fpsg_stream_t stream);

/* ---- K2: EMD, approximate-assignment solver -------------------------------------
 * Stands where the reference calls neuralnet_pytorch.metrics.emd_loss through
 * emd_wrapper (src/models/utils.py:12-13, used at src/models/few_shot.py:168).
 * Auction-style soft assignment over 10 temperature levels (Fan/Su/Guibas approxmatch)
 * followed by the transport cost sum_kl match_kl * |xyz1_k - xyz2_l|; the N x M match
 * matrix is never stored.  cost [B].  gxyz1 [B,N,3] / gxyz2 [B,M,3] may each be NULL;
 * when given they receive d cost / d xyz with the assignment held constant (the
 * convention of the original matchcost gradient).  ws: caller scratch of
 * fpsg_emd_workspace_floats(B,N,M) floats, 16-byte aligned (per-point state and
 * coordinate-major copies of both clouds, rows padded to multiples of 4 points).
 * Deterministic (no float atomics).
 * fpsg_emd_approx_variant: `variant` -1 = automatic (what fpsg_emd_approx passes), else two bits that only act on
 * forward-only calls (both gradients NULL: the evaluation path): bit 0 = a level's assignment sweep and the next
 * level's row-normaliser sweep as ONE launch (same owners, same swept cloud; 21 sweeps per call instead of 30; the
 * assignment's exp(level d^2) is formed as the fourth power of the normaliser's exp(level/4 d^2)), bit 1 = four owner
 * points per wave instead of two.  Same values up to fp32 rounding of the exponentials (~4e-7 per weight).
 */
size_t fpsg_emd_workspace_floats(int B, int N, int M);
int fpsg_emd_approx(const float* xyz1, const float* xyz2, int B, int N, int M, float* cost,
                    float* gxyz1, float* gxyz2, float* ws, fpsg_stream_t stream);
int fpsg_emd_approx_variant(const float* xyz1, const float* xyz2, int B, int N, int M, float* cost,
                            float* gxyz1, float* gxyz2, float* ws, int variant, fpsg_stream_t stream);

/* ---- K4b: fused EdgeConv (gather + BatchNorm statistics + max over k) -----------------
 * Replaces the chain get_graph_feature -> Conv2d 1x1 -> BatchNorm2d -> LeakyReLU -> max_k of
 * src/dgcnn/model.py:23-42,53-56,63-76 without materialising [B,2C,N,k].  The caller first
 * forms PQ [B,N,2*Co] = x^T [W1 ; W2-W1]^T (one GEMM): y(n,j) = P[idx[n,j]] + Q[n].
 *   fwd : sgn [Co]: >= 0 take max_j, < 0 take min_j per channel (only the sign is used: pass BN's gamma);
 *         ysel [B,N,Co] selected extreme of y over j; jsel [B,N,Co] uint8 its slot j (first on
 *         ties); s1 [B,N,Co] = sum_j y (NULL to skip); part [fpsg_edgeconv_blocks(B,N,Co)][2][Co]
 *         per-workgroup sums of y and y^2 for the BatchNorm statistics (NULL to skip).
 *   bwd : dzs [B,N,Co] = dL/dz * scale at the selected edge; rev [B,N*k] edge ids n*k+j sorted
 *         by destination idx (ascending id inside a destination), off [B,N+1] offsets into rev;
 *         coef [3][Co] = (A, Bc, mu) BatchNorm statistic terms (s1 == NULL: eval mode, coef
 *         ignored).  dPQ [B,N,2*Co] is fully overwritten.  Deterministic (no float atomics).
 * Co must be 64, 128 or 256; float buffers 16-byte aligned.
 */
int fpsg_edgeconv_blocks(int B, int N, int Co);
int fpsg_edgeconv_fwd(const float* PQ, const int32_t* idx, const float* sgn, int B, int N, int k,
                      int Co, float* ysel, uint8_t* jsel, float* s1, float* part,
                      fpsg_stream_t stream);
int fpsg_edgeconv_bwd(const float* dzs, const uint8_t* jsel, const float* PQ, const float* s1,
                      const int32_t* rev, const int32_t* off, const float* coef, int B, int N, int k,
                      int Co, float* dPQ, fpsg_stream_t stream);

/* The in-edge lists fpsg_edgeconv_bwd gathers through, from the neighbour lists idx [B][N][k] (values in [0,N)):
 * rev [B][N*k] = the edges e = n*k + j grouped by destination idx[n][j], ascending e inside a group (the backward's
 * summation order); off [B][N+1] = first slot of every destination's group, off[N] = N*k.  Replaces what autograd's
 * index backward of get_graph_feature's gather (src/dgcnn/model.py:30-56) does with atomics.  One workgroup per cloud,
 * a stable counting sort in LDS; deterministic.  Limits: N*k <= 65535 and N small enough for the LDS --
 * fpsg_edgeconv_reverse_graph_fits(N,k) answers 1 when both hold (the Python mirror sorts with torch otherwise).
 * Entries of idx outside [0,N) are skipped (their slots of rev stay unwritten). */
int fpsg_edgeconv_reverse_graph_fits(int N, int k);
int fpsg_edgeconv_reverse_graph(const int32_t* idx, int B, int N, int k, int32_t* rev, int32_t* off,
                                fpsg_stream_t stream);
/* The elementwise halves around them, on point-major [rows = B*N, Co] tensors: fpsg_edgeconv_act = the BatchNorm
 * affine form + LeakyReLU of the selected neighbour sum, out = lrelu(fma(ysel, scale[c], shift[c])) (one pass instead
 * of addcmul + leaky_relu); fpsg_edgeconv_bwd_prep = the head of the backward: z re-derived with the same arithmetic,
 * dz = g * lrelu'(z), dzs = dz * scale[c] for fpsg_edgeconv_bwd, and part [fpsg_edgeconv_prep_blocks(rows)][2][Co] =
 * per-workgroup sums of dz and dz * ysel (-> dbeta, dgamma, BatchNorm coefficients): g and ysel read once, dzs written
 * once (six torch ops = twelve passes otherwise).  Co in {64, 128, 256}; 16-byte aligned.  Deterministic. */
int fpsg_edgeconv_prep_blocks(long rows);
/* The per-channel scalar work between them, one launch each: fpsg_edgeconv_stats_finalize turns the forward kernel's
 * partial sums (count = B*N*k edge activations) into chan [4][Co] = (gamma*rstd, beta - mean*gamma*rstd, mean, rstd) and
 * updates the running statistics as nn.BatchNorm2d does (training = 0: chan from the running statistics);
 * fpsg_edgeconv_bwd_finalize turns fpsg_edgeconv_bwd_prep's partial sums into dgamma, dbeta and coef [3][Co] of
 * fpsg_edgeconv_bwd (zeros in eval mode).  fp64 sums in block order. */
int fpsg_edgeconv_stats_finalize(const float* part, int blocks, const float* gamma, const float* beta,
                                 float* running_mean, float* running_var, float momentum, float eps, double count, int Co,
                                 int training, float* chan, fpsg_stream_t stream);
/* The same with the partial rows summed in two stages (round 4): slice sums over whole coalesced row pieces by ~256-512
 * workgroups into ws (fpsg_edgeconv_stats_ws_floats(blocks, Co) floats, 8-byte aligned: [Z][2 Co] doubles), then one wave per
 * channel.  Deterministic; equal to the one-launch form up to the order of its fp64 sums.  ws NULL, eval mode or fewer than
 * 256 rows: the one-launch form. */
size_t fpsg_edgeconv_stats_ws_floats(int blocks, int Co);
int fpsg_edgeconv_stats_finalize_ws(const float* part, int blocks, const float* gamma, const float* beta,
                                    float* running_mean, float* running_var, float momentum, float eps, double count,
                                    int Co, int training, float* chan, float* ws, fpsg_stream_t stream);
int fpsg_edgeconv_bwd_finalize(const float* part, int blocks, const float* chan, double count, int Co, int training,
                               float* dgamma, float* dbeta, float* coef, fpsg_stream_t stream);
int fpsg_edgeconv_act(const float* ysel, const float* scale, const float* shift, float slope, long rows, int Co,
                      float* out, fpsg_stream_t stream);
int fpsg_edgeconv_bwd_prep(const float* g, const float* ysel, const float* scale, const float* shift, float slope,
                           long rows, int Co, float* dzs, float* part, fpsg_stream_t stream);

/* ---- K2b: soft-min of the Sinkhorn loop ---------------------------------------------
 * The operator under neuralnet_pytorch.metrics.emd_loss(sinkhorn=True) = geomloss.SamplesLoss()
 * (the form src/models/utils.py:12-13 actually calls): with cost |x-y|^2 / 2,
 *   out[b,i] = -eps * log sum_j exp( h[b,j] - |x_i - y_j|^2 / (2 eps) )
 * x [B,N,3], y [B,M,3], h [B,M] (log-weights + dual potential / eps), out [B,N].
 * No [B,N,M] matrix is formed.  Deterministic.
 */
int fpsg_softmin(const float* x, const float* y, const float* h, int B, int N, int M, float eps,
                 float* out, fpsg_stream_t stream);

/* The whole divergence of emd_wrapper (src/models/utils.py:12-13): the symmetric, annealed, debiased
 * Sinkhorn loop of geomloss.SamplesLoss("sinkhorn", p=2) between uniform clouds x [B,N,3], y [B,M,3]:
 *   duals a_x, b_x [N], a_y, b_y [M] start as the soft-mins of the log-weights at eps[0]; for every
 *   eps of the schedule the four soft-mins (h = log-weight + dual * (1/eps)) are evaluated from the
 *   previous duals and averaged in, new = (old + softmin)/2; one last un-averaged evaluation at
 *   eps[n-1];  out[b] = mean_i (b_x - a_x) + mean_j (a_y - b_y).
 * ONE launch per schedule entry (the four soft-mins side by side), n_eps + 3 launches in all.
 * eps_host: the schedule as n_eps HOST floats (geomloss: diameter^2, then diameter^2 * scaling^2k
 * down to blur^2, then blur^2); ws: fpsg_sinkhorn_workspace_floats(B,N,M) device floats.  Deterministic. */
size_t fpsg_sinkhorn_workspace_floats(int B, int N, int M);
int fpsg_sinkhorn_divergence(const float* x, const float* y, int B, int N, int M,
                             const float* eps_host, int n_eps, float* out, float* ws,
                             fpsg_stream_t stream);

/* ---- K9: first layer of the decoder's patch MLPs ------------------------------------------
 * Replaces, per decode, the 16 x [conv1 (1539 -> 1539, 1x1) + BatchNorm1d + ReLU] of
 * PrimitiveNode.forward (src/models/point_cloud_net.py:76-80) applied to cat(x.repeat, patch points)
 * (src/models/point_cloud_net.py:105-110), in the split form  h[g,d,b,q] = hlat[g,d,b] + w[g,d,wofs:wofs+3] . pts[g,:,b,q]
 * with hlat = W[:, :L] x + bias computed by the caller (one GEMM per patch).  G patches, D channels, B clouds,
 * P points per patch (4 x a power of two, <= 256; B*P <= 8192).  w [G,D,ldw] is the stacked conv1 weight.
 *   fwd: out [G,D,B*P] = relu(BN(h)); statistics per (g,d) over the B*P values (training) or the given running
 *        statistics; chan [4][G*D] = (scale, shift, mean, rstd) for the backward; batch_mean /
 *        batch_var_unbiased [G*D] optional (training).  The pre-BatchNorm tensor is never stored.
 *   bwd: from dout [G,D,B*P]: dhlat [G,D,B] (gradient of the latent GEMM's output), the three point columns of
 *        dw (same layout / ldw / wofs as w; other columns untouched), dpts_part [G, fpsg_dec1_tiles(D), 3, B*P]
 *        (partial sums over channel tiles; the caller adds them up), dgamma, dbeta [G*D].
 * pts, out, dout 16-byte aligned.  Deterministic. */
int fpsg_dec1_tiles(int D);
int fpsg_dec1_fwd(const float* hlat, const float* w, int ldw, int wofs, const float* pts,
                  const float* gamma, const float* beta, const float* running_mean, const float* running_var,
                  int G, int D, int B, int P, int training, float eps, float* out, float* chan,
                  float* batch_mean, float* batch_var_unbiased, fpsg_stream_t stream);
int fpsg_dec1_bwd(const float* dout, const float* hlat, const float* w, int ldw, int wofs, const float* pts,
                  const float* chan, int G, int D, int B, int P, int training, float* dhlat, float* dw,
                  float* dpts_part, float* dgamma, float* dbeta, fpsg_stream_t stream);

/* The same with row strides: pts rows ld_pts apart, out / dout rows ld_out / ld_dout apart (multiples of 4, >= B*P),
 * hlat / dhlat rows ld_hlat apart (>= B) --
 * a call then owns a column range of tensors that hold several decodes side by side (an episode's query and support
 * decodes share every GEMM after this layer).  accumulate != 0: dw's point columns, dgamma and dbeta are added to the
 * buffers' contents (the second call of a pair) instead of overwriting them. */
int fpsg_dec1_fwd_ld(const float* hlat, int ld_hlat, const float* w, int ldw, int wofs, const float* pts, int ld_pts,
                     const float* gamma, const float* beta, const float* running_mean, const float* running_var,
                     int G, int D, int B, int P, int training, float eps, float* out, int ld_out, float* chan,
                     float* batch_mean, float* batch_var_unbiased, fpsg_stream_t stream);
int fpsg_dec1_bwd_ld(const float* dout, int ld_dout, const float* hlat, int ld_hlat, const float* w, int ldw, int wofs,
                     const float* pts, int ld_pts, const float* chan, int G, int D, int B, int P, int training,
                     int accumulate, float* dhlat, float* dw, float* dpts_part, float* dgamma, float* dbeta,
                     fpsg_stream_t stream);

/* ---- K5: BatchNorm fused with its activation (training and eval mode) -------------------
 * Replaces the BatchNorm{1,2}d + ReLU / LeakyReLU module pairs of the reference networks
 * (src/models/image_net.py:14 VGG16-BN trunk; src/pointnet/model.py:30-44,220-233;
 * src/models/point_cloud_net.py:52-54,76-79) on [N,C,L]-contiguous fp32 tensors; statistics
 * per channel over (N,L).  act: 0 none, 1 ReLU, 2 LeakyReLU(slope).
 *   fwd: y = act((x-mean)*rstd*gamma+beta); chan [4][C] receives (scale, shift, mean, rstd) for
 *        the backward; training != 0: batch statistics; running_mean / running_var [C] (optional)
 *        are then updated in place, running <- (1-momentum)*running + momentum*batch (unbiased
 *        variance), unless momentum < 0; batch_mean / batch_var_unbiased [C] are optional outputs
 *        for a caller with its own update rule.  training == 0: running_mean/var are the statistics.
 *        training == 2 (fpsg_bn_act_fwd, fpsg_bn_act_pool_fwd, fpsg_bn_act_max_fwd): evaluation mode with chan ALREADY
 *        holding the coefficients -- the caller got them once from fpsg_bn_stats(training = 0) for a block of calls in
 *        which the running statistics do not change (the evaluation loop, src/evaluate_Network.py:107-118); saves the
 *        coefficient launch of every layer and item.
 *   bwd: dx [N,C,L], dgamma [C], dbeta [C] from x, dy and chan (the activation mask is
 *        re-derived from x); coef [3][C] scratch.
 * pre_bias [C] (optional, NULL = none): the bias of the convolution in front of the BatchNorm
 * (the Conv2d/Conv1d + BatchNorm + ReLU triples of image_net.py:14, pointnet/model.py:30-44);
 * x is then the bias-free convolution output and the op is act(BN(x + pre_bias[c])), with
 * fl(x + pre_bias) formed in registers; bwd additionally returns dpre_bias[c] = sum over (n,l)
 * of dx (optional, NULL = skip) -- the bias gradient autograd would reduce from dx.
 * ws: fpsg_bn_workspace_floats(N,C,L) floats.  x, y, dy, dx 16-byte aligned.  Deterministic.
 * Environment FPSG_BN_FINALIZE_FOLD=1 (measurement switch, default off): fpsg_bn_act_fwd's finalize runs in the last-arriving
 * workgroup of the statistics kernel instead of in its own launch -- the same bits, measured 10-93 us SLOWER per call
 * (profiles/r05/bn_finalize_fold_rejected.txt).
 */
size_t fpsg_bn_workspace_floats(int N, int C, int L);
int fpsg_bn_act_fwd(const float* x, const float* pre_bias, const float* gamma, const float* beta,
                    float* running_mean, float* running_var, float momentum, int N, int C, int L, int training,
                    float eps, int act, float slope, float* y, float* chan, float* batch_mean,
                    float* batch_var_unbiased, float* ws, fpsg_stream_t stream);
/* Statistics half of fpsg_bn_act_fwd alone: chan [4][C] (scale, shift, mean, rstd) from the batch (training; running
 * statistics updated as there) or from the running statistics (eval); nothing is applied or written besides.
 * parts (optional, training): partial sums [C][n_parts][2] = (sum(x + pre_bias), sum((x + pre_bias)^2)) that the
 * producing convolution's epilogue accumulated (fpsg_wino_output_transform_stats, fpsg_wino_conv_fused_stats):
 * the pass over x is skipped and only the fp64 finalize runs (ws may then be NULL). */
int fpsg_bn_stats(const float* x, const float* pre_bias, const float* gamma, const float* beta,
                  float* running_mean, float* running_var, float momentum, int N, int C, int L, int training,
                  float eps, float* chan, float* batch_mean, float* batch_var_unbiased, float* ws,
                  const float* parts, int n_parts, fpsg_stream_t stream);
int fpsg_bn_act_bwd(const float* x, const float* pre_bias, const float* dy, const float* chan, int N, int C,
                    int L, int training, int act, float slope, float* dx, float* dgamma, float* dbeta,
                    float* dpre_bias, float* coef, float* ws, fpsg_stream_t stream);
/* The sums + coefficient half of fpsg_bn_act_bwd alone: dgamma, dbeta, coef [3,C] = (k1, k2, k3) of
 * dx = k1 dz + k2 (x + pre_bias) + k3 -- for a consumer that forms dx itself while reading x and dy
 * (fpsg_conv_first_dw_fold).  N * L above the small-tensor limit (16384) only; ws as fpsg_bn_act_bwd. */
int fpsg_bn_act_bwd_coef(const float* x, const float* pre_bias, const float* dy, const float* chan, int N, int C,
                         int L, int training, int act, float slope, float* dgamma, float* dbeta, float* coef,
                         float* ws, fpsg_stream_t stream);
/* Training-mode K5 over column segments of C rows that lie ld elements apart (x, y, dy, dx share the layout): segment i
 * = the elements seg_off[i] .. seg_off[i] + seg_len[i] - 1 of every row is one BatchNorm call (one reference
 * BatchNorm1d call per patch and decode, src/models/point_cloud_net.py:76-80): own statistics, one launch for all
 * segments.  nseg <= 4, seg_len <= 16384, ld and seg_off multiples of 4 (host arrays), segments disjoint.
 *   fwd: y = act(BN(x + pre_bias)); chan [nseg][4][C]; stats [nseg][2][C] (optional): batch mean, unbiased batch variance.
 *   bwd: dx; dgamma, dbeta, dpre_bias (optional) [nseg][C] -- the caller adds the segments (the affine parameters are
 *        shared).  Columns outside every segment are neither read nor written. */
int fpsg_bn_act_rows_fwd(const float* x, int ld, const int* seg_off, const int* seg_len, int nseg,
                         const float* pre_bias, const float* gamma, const float* beta, int C, float eps, int act,
                         float slope, float* y, float* chan, float* stats, fpsg_stream_t stream);
int fpsg_bn_act_rows_bwd(const float* x, int ld, const int* seg_off, const int* seg_len, int nseg,
                         const float* pre_bias, const float* dy, const float* chan, int C, int act, float slope,
                         float* dx, float* dgamma, float* dbeta, float* dpre_bias, fpsg_stream_t stream);
/* fpsg_bn_act_bwd with the two sums per channel delivered by the kernel that produced dy
 * (fpsg_wino_output_transform_bwd_stats): parts [C][n_parts][2].  ws may be NULL unless dpre_bias is wanted. */
int fpsg_bn_act_bwd_parts(const float* x, const float* pre_bias, const float* dy, const float* chan, int N,
                          int C, int L, int training, int act, float slope, float* dx, float* dgamma,
                          float* dbeta, float* dpre_bias, float* coef, float* ws, const float* parts,
                          int n_parts, fpsg_stream_t stream);

/* K5 followed by MaxPool2d(kernel 2, stride 2): the conv + BatchNorm + ReLU + max-pool groups
 * that end the five stages of the VGG16-BN trunk (src/models/image_net.py:14).  x [N,C,H,W]
 * (H, W even), y_pooled / dy_pooled [N,C,H/2,W/2].  The forward writes only the pooled
 * tensor; the backward re-derives each window's activations and arg-max from x (scan order
 * (h,w), first strictly greater or NaN wins, as torch's max_pool2d), so neither the
 * full-resolution activation, nor pooling indices, nor the scattered gradient exist in HBM.
 * Other arguments as fpsg_bn_act_fwd / _bwd; ws: fpsg_bn_pool_workspace_floats(N,C,H,W); parts / n_parts as
 * fpsg_bn_stats (statistics delivered by the producing convolution).
 */
size_t fpsg_bn_pool_workspace_floats(int N, int C, int H, int W);
int fpsg_bn_act_pool_fwd(const float* x, const float* pre_bias, const float* gamma, const float* beta,
                         float* running_mean, float* running_var, float momentum, int N, int C, int H, int W,
                         int training, float eps, int act, float slope, float* y_pooled, float* chan,
                         float* batch_mean, float* batch_var_unbiased, float* ws, const float* parts, int n_parts,
                         fpsg_stream_t stream);
int fpsg_bn_act_pool_bwd(const float* x, const float* pre_bias, const float* dy_pooled, const float* chan,
                         int N, int C, int H, int W, int training, int act, float slope, float* dx,
                         float* dgamma, float* dbeta, float* dpre_bias, float* coef, float* ws,
                         fpsg_stream_t stream);

/* K5 followed by the max over the row: max_l act(BN(x + pre_bias))[n,c,l] -> out [N,C], the
 * BatchNorm (+ReLU) + torch.max(x, 2) tails of PointNet's shared MLPs (src/pointnet/model.py:35-37,
 * 222-224).  act and BN are monotone, so only each row's extreme of x is normalised: the forward
 * is one read of x (no normalised tensor is written), idx [N,C] receives the selected position
 * (first arg-max of x where scale >= 0, first arg-min otherwise); the backward takes gout [N,C],
 * and is one read + one write (dx).  Other arguments as fpsg_bn_act_fwd / _bwd;
 * ws: fpsg_bn_max_workspace_floats(N,C,L).  fpsg_bn_max_dz_offset(N,C,L): where, in floats from the start of that
 * workspace, fpsg_bn_act_max_bwd_coef leaves dz [N,C] (the gradient at the selected positions, in front of the
 * BatchNorm) for fpsg_max_bwd_gather / _prep / _scatter -- the workspace layout is the library's, not the caller's.
 */
size_t fpsg_bn_max_workspace_floats(int N, int C, int L);
size_t fpsg_bn_max_dz_offset(int N, int C, int L);
int fpsg_bn_act_max_fwd(const float* x, const float* pre_bias, const float* gamma, const float* beta,
                        float* running_mean, float* running_var, float momentum, int N, int C, int L, int training,
                        float eps, int act, float slope, float* out, int32_t* idx, float* chan, float* batch_mean,
                        float* batch_var_unbiased, float* ws, fpsg_stream_t stream);
int fpsg_bn_act_max_bwd(const float* x, const float* pre_bias, const float* gout, const int32_t* idx,
                        const float* chan, int N, int C, int L, int training, int act, float slope, float* dx,
                        float* dgamma, float* dbeta, float* dpre_bias, float* coef, float* ws, fpsg_stream_t stream);
/* The coefficient pass of fpsg_bn_act_max_bwd alone (no dense dx): dgamma, dbeta, coef [3][C] = k1, k2, k3 of
 * dx'[n,c,l] = k1_c dz[n,c] [l = idx[n,c]] + k2_c (x[n,c,l] + pre_bias_c) + k3_c, and dz [N,C] left in ws at float offset
 * C*128 + N*C*ceil(L/4096)*4.  ws: fpsg_bn_max_workspace_floats(N,C,L) floats. */
int fpsg_bn_act_max_bwd_coef(const float* x, const float* pre_bias, const float* gout, const int32_t* idx,
                             const float* chan, int N, int C, int L, int training, int act, float slope,
                             float* dgamma, float* dbeta, float* coef, float* ws, fpsg_stream_t stream);

/* ---- K5m: backward of conv1x1 -> BatchNorm (+ReLU) -> max over the points without the dense gradient ----------
 * (PointNet's last shared layer, src/pointnet/model.py:35-37, 222-224).  With the coefficients of
 * fpsg_bn_act_max_bwd_coef the two GEMMs over the [B,C,L] gradient reduce to K x K algebra (library GEMMs on the
 * caller's side, fpsg_amd/fused_bn.py) plus these two passes.  a [B,K,L] the layer's input, W [C,K], idx [B,C] the
 * selected point of every row, dz [B,C].
 *   fpsg_max_bwd_gather : S[c,k] = sum_b dz[b,c] a[b,k,idx[b,c]]            (ascending b; L % 4 == 0, L <= 2048)
 *   fpsg_max_bwd_scatter: da[b,k,l] += v[k] + sum_{c: idx[b,c] = l} (k1[c] dz[b,c]) W[c,k]   (ascending c; K <= 128,
 *                         L < 65407; ws: fpsg_max_bwd_scatter_workspace_floats(B,C,L) floats of caller scratch)
 * Deterministic (no float atomics). */
int fpsg_max_bwd_gather(const float* a, const float* dz, const int32_t* idx, int B, int K, int C, int L, float* S,
                        float* spart /* [B,K] row sums of a, or NULL */, fpsg_stream_t stream);
/*   fpsg_max_bwd_prep   : Wk = diag(k2) W, u = k2 pb + k3, dpre_bias = k1 sum_b dz + (k2 mean + k3) B L, and (spart, s
 *                         given) s[k] = sum_b spart[b,k], spart [B,K] = the row sums of a                 (one launch)
 *   fpsg_max_bwd_dw     : dw = k1 S + k2 (WG + pb s^T) + k3 s^T; WG = NULL (eval mode): dw = k1 S */
int fpsg_max_bwd_prep(const float* W, const float* coef, const float* pre_bias, const float* mean, const float* dz,
                      const float* spart, int B, int K, int C, int L, float* Wk, float* u, float* dpre_bias, float* s,
                      fpsg_stream_t stream);
int fpsg_max_bwd_dw(const float* S, const float* WG, const float* coef, const float* pre_bias, const float* s, int B,
                    int K, int C, float* dw, fpsg_stream_t stream);
size_t fpsg_max_bwd_scatter_workspace_floats(int B, int C, int L);
int fpsg_max_bwd_scatter(float* da, const float* W, const float* k1, const float* dz, const int32_t* idx, const float* v,
                         int B, int K, int C, int L, float* ws, fpsg_stream_t stream);


/* ---- K6: Winograd F(m x m, 3x3) transforms (m = 2 or 4) for the deep 3x3 convolutions -------
 * Replaces, together with the caller's fp32 batched GEMM (hipBLASLt, MFMA), the library
 * convolution under the Conv2d(3x3, padding 1) layers of torchvision's vgg16_bn.features that
 * src/models/image_net.py:14 instantiates (forward :21-24, full backward: SURVEY.md F9).
 * With A = m + 2 and P = N*(H/m)*(W/m) tiles (tile p = (n*(H/m) + th)*(W/m) + tw; H, W multiples
 * of m):
 *   y  = conv(x, w):           U = filter(w, flip 0) [A*A,K,C];  V = input(x) [A*A,C,P];
 *                              M[xi] = U[xi] V[xi] [A*A,K,P];     y = output(M)
 *   dx = conv(dy, rot180 w^T): U' = filter(w, flip 1) [A*A,C,K]; V' = input(dy) [A*A,K,P];
 *                              M'[xi] = U'[xi] V'[xi] [A*A,C,P]; dx = output(M')
 *   dw:                        dM = grad_output(dy) [A*A,K,P];   dU[xi] = dM[xi] V[xi]^T [A*A,K,C];
 *                              dw = filter_grad(dU)
 * m = 2: 2.25x fewer multiplications than the direct form, fp32 error a few ulp; m = 4: 4x fewer,
 * error ~1e-5 of the output scale (transform constants up to 8 and 1/24).  All tensors fp32,
 * contiguous, caller-allocated; image tensors 16-byte aligned.  Deterministic.
 * ldp (round 5): the ROW STRIDE of the transform-domain tensors V / M / dM in floats -- [A*A][channels][ldp], tile p of a
 * row at offset p -- 0 (= P, dense) or any value >= P.  P is 29008 / 7252 / 1813 / 592 at the trunk's shapes: never a
 * whole number of 128-byte lines (1813 is odd), so dense rows make every row piece a GEMM tile reads or writes share
 * its first and last line with the neighbouring tile; with ldp = P rounded up to 32 floats the library's products run
 * 4-13 % faster (profiles/r05/wino_row_stride.txt).  The transforms that WRITE a transform-domain tensor fill the pad
 * columns P .. ldp-1 with zeros (the products may then simply run over ldp columns: a zero column adds nothing to the
 * weight gradient's reduction); the output transforms never read them.
 */
int fpsg_wino_input_transform(int m, const float* x, int N, int C, int H, int W, float* V, long ldp,
                              fpsg_stream_t stream);
int fpsg_wino_output_transform(int m, const float* M, int N, int K, int H, int W, float* y, long ldp,
                               fpsg_stream_t stream);
/* The same, also accumulating the statistics of the BatchNorm that follows the convolution
 * (nn.Conv2d -> nn.BatchNorm2d of image_net.py:14): parts [K][fpsg_wino_stats_parts(m,N,H,W)][2] =
 * (sum(y + bias[k]), sum((y + bias[k])^2)) per workgroup of 256 tiles (bias optional: the convolution's bias, which K5
 * adds inside the BatchNorm); fpsg_bn_stats / fpsg_bn_act_pool_fwd take them instead of reading y again.  Deterministic. */
int fpsg_wino_stats_parts(int m, int N, int H, int W);
int fpsg_wino_output_transform_stats(int m, const float* M, int N, int K, int H, int W, float* y, const float* bias,
                                     float* parts, long ldp, fpsg_stream_t stream);
/* The output transform of a DATA-GRADIENT convolution whose result is the gradient of relu(bn(xpre + pre_bias)):
 * besides y it delivers the two sums BatchNorm's backward starts from -- sum(dz) and sum(dz * (x - mean) * rstd) with
 * dz = y * [fma(x, scale, shift) > 0], x = xpre + pre_bias[k] -- per output channel and workgroup into
 * parts [K][fpsg_wino_stats_parts(m,N,H,W)][2] (deterministic), in the arithmetic of fpsg_bn_act_bwd's own pass.
 * xpre [N,K,H,W]: the pre-BatchNorm tensor; chan [4][K]: scale, shift, mean, rstd (fpsg_bn_stats / _act_fwd).
 * fpsg_bn_act_bwd_parts then skips its pass over (x, dy). */
int fpsg_wino_output_transform_bwd_stats(int m, const float* M, int N, int K, int H, int W, float* y,
                                         const float* xpre, const float* pre_bias, const float* chan,
                                         float* parts, long ldp, fpsg_stream_t stream);
int fpsg_wino_grad_output_transform(int m, const float* dy, int N, int K, int H, int W, float* dM, long ldp,
                                    fpsg_stream_t stream);
/* Both transforms of an output gradient dy [N,K,H,W] in ONE pass (round 4): V = B^T d B of its 6x6 (4x4) patches -- the
 * "input" transform of the data gradient's convolution -- and dM = A dy A^T of its tiles -- the weight gradient's -- the
 * tile being the patch's interior.  V, dM [A*A, K, P]; values bit-identical to fpsg_wino_input_transform(dy) and
 * fpsg_wino_grad_output_transform(dy); one read of dy instead of two. */
int fpsg_wino_grad_transforms(int m, const float* dy, int N, int K, int H, int W, float* V, float* dM, long ldp,
                              fpsg_stream_t stream);
int fpsg_wino_filter_transform(int m, const float* w, int K, int C, int flip_transpose, float* U,
                               fpsg_stream_t stream);
/* Every filter transform of an optimizer step in one launch (the weights change once per step): jobs [n_jobs][6]
 * int64 in DEVICE memory -- w pointer [K][C][3][3], U pointer (layout of fpsg_wino_filter_transform), K, C,
 * 2*m + flip_transpose, first workgroup of the job -- with the jobs' workgroups (256 filters each, ceil(K*C/256) per
 * job) numbered consecutively; total_blocks = their sum.  Same arithmetic per filter as the single form. */
int fpsg_wino_filter_transform_batch(const int64_t* jobs, int n_jobs, long total_blocks, fpsg_stream_t stream);
int fpsg_wino_filter_grad_transform(int m, const float* dU, int K, int C, float* dw, fpsg_stream_t stream);

/* ---- K10 (round 5, opt-in: FPSG_GEMM_SPLIT=1): batched fp32 GEMM on the bf16 matrix pipe, operands split three ways ----
 * Replaces the library fp32 GEMMs (torch.bmm -> rocBLAS / hipBLASLt, fp32 MFMA at 1/16 of the bf16 rate) behind the
 * Winograd-domain products of the trunk's 3x3 convolutions (torchvision vgg16_bn.features built at
 * src/models/image_net.py:14, run at src/models/image_net.py:21-24).
 *   transB = 0:  C[b] [M x N] = A[b] [M x K] . B[b] [K x N]     (forward / data gradient: U[xi] . V[xi])
 *   transB = 1:  C[b] [M x N] = A[b] [M x K] . B[b]^T, B [N x K] (weight gradient: dM[xi] . V[xi]^T; the reduction is
 *                split over workgroups, partial slabs in ws, added in a fixed order: deterministic)
 * Row-major, leading dimensions lda / ldb / ldc and batch strides sA / sB / sC in floats.  Every fp32 operand x enters as
 * bf16(x) + bf16(x - x1) + bf16(x - x1 - x2) (an exact split) and the six products of order <= 2^-18 are accumulated in
 * fp32 by v_mfma_f32_32x32x16_bf16: fp32-grade results (error against float64: profiles/r05/), not the library's bits.
 * variant: -1 automatic; else tile + 10 * splits, tile 0 = 256x256x16, 1 = 256x128x32, splits 0 = automatic.
 * ws: fpsg_gemm_split_workspace_floats(...) floats (0 when the reduction is not split; then ws may be NULL); a split
 * reduction needs a dense output (ldc == N, sC == M*N).  One batch entry of each matrix must stay below 2 GiB. */
size_t fpsg_gemm_split_workspace_floats(int batch, int M, int N, int K, int transB, int variant);
int fpsg_gemm_split(const float* A, const float* B, float* C, int batch, int M, int N, int K, int lda, int ldb, int ldc,
                    long sA, long sB, long sC, int transB, int variant, float* ws, size_t ws_floats,
                    fpsg_stream_t stream);

/* K10 with the A operand split once (the transformed filters: constant over an optimizer step).  fpsg_gemm_split_pack_a writes
 * A [batch][M][K] (row-major, lda, batch stride sA) as three bf16 planes in the layout of the GEMM's LDS image, zero-padded
 * (fpsg_gemm_split_packed_a_bytes bytes, 16-byte aligned; about 1.5x the fp32 matrix); fpsg_gemm_split_nn_packed computes
 * C[b] [M x N] = A[b] . B[b] [K x N] with A brought in by LDS-DMA and only B split on the way into LDS: the same values as
 * fpsg_gemm_split(transB = 0), bit for bit.  variant (the same for the three calls): -1 / 0 = 256 x 128 x 16 tiles, two
 * workgroups per CU; 1 = 256 x 256 x 16; 2 = 128 x 128 x 16. */
size_t fpsg_gemm_split_packed_a_bytes(int batch, int M, int K, int variant);
int fpsg_gemm_split_pack_a(const float* A, int batch, int M, int K, int lda, long sA, int variant, void* Ap,
                           fpsg_stream_t stream);
int fpsg_gemm_split_nn_packed(const void* Ap, const float* B, float* C, int batch, int M, int N, int K, int ldb, int ldc,
                              long sB, long sC, int variant, fpsg_stream_t stream);

/* The same product as fpsg_gemm_split_nn_packed (Ap packed with variant 0 or 1: 256-row tiles) as ONE persistent launch:
 * a workgroup per CU walks an equal share of the flattened (batch, row tile, column) space, the load / split / MFMA
 * pipeline runs across tile boundaries (no per-tile launch, prologue or drain; 1044 tiles on 256 CUs cost 4.08 rounds, not
 * 5).  Same values as fpsg_gemm_split_nn_packed bit for bit.  variant: -1 / 0 = 256 columns per tile, 1 = 128;
 * 6 / 12 = the form with specialised waves (4 waves only multiply, 4 only stage: LDS-DMA of both operands, the split of B
 * from LDS), 256 x 128 tiles, 12 with staggered first pieces; 13 / 14 = the same with 128 x 128 tiles and two workgroups
 * per CU (Ap packed with variant 2).  6 and 12-14 move B rows by 16-byte DMA: B, ldb and sB aligned to 4 floats
 * (FPSG_E_SHAPE otherwise).  2-5 and 7-11 are measurement builds (wrong results by design).  Measured: none of the
 * persistent forms beats the tiled kernel (profiles/r05/gemm_split_ablation.txt); they are kept as measured evidence and
 * are not on any product path.  B below 4 GiB in total, one C matrix and the packed A below 2 GiB. */
int fpsg_gemm_split_nn_persistent(const void* Ap, const float* B, float* C, int batch, int M, int N, int K, int ldb, int ldc,
                                  long sB, long sC, int variant, fpsg_stream_t stream);

/* ---- K11: batched fp32 GEMM on the fp32 matrix pipe, hand-written (round 5) -------------------------------------------
 * C[b] [M x N] = A[b] [M x K] . B[b] [K x N], row-major, leading dimensions / batch strides in floats: the product
 * torch.bmm hands to rocBLAS / hipBLASLt for the Winograd-domain GEMMs of torchvision vgg16_bn.features
 * (src/models/image_net.py:14,21-24) and the decoder's wide layers (src/models/point_cloud_net.py:66-79), with the same
 * arithmetic class (v_mfma_f32_32x32x2_f32: fp32 products, fp32 accumulation; the order of the sum differs from the
 * library's, so results agree to fp32 round-off, not bit for bit).  One persistent launch: a workgroup per CU walks an
 * equal share of the flattened (batch, row tile, column) space; four waves stage both operands by LDS-DMA, four multiply.
 * K, lda, ldb, sA, sB multiples of 4 floats, A and B 16-byte aligned (FPSG_E_ALIGN otherwise: the caller then takes the
 * library GEMM); A and B below 4 GiB each in total, one C matrix below 2 GiB.  variant: -1 / 0 default, 1 = staggered
 * first pieces.  Deterministic. */
int fpsg_gemm_f32_nn(const float* A, const float* B, float* C, int batch, int M, int N, int K, int lda, int ldb, int ldc,
                     long sA, long sB, long sC, int variant, fpsg_stream_t stream);

/* K6 in one kernel for 64 input channels (F(4x4,3x3); conv1_2 / conv2_1 of the trunk and their data
 * gradients): y [N,K,H,W] = conv(x [N,64,H,W], w) from U = fpsg_wino_filter_transform(4, w, ...)
 * [36,K,64]; input transform, the 36 MFMA products and the output transform stay on chip -- with 64
 * channels the separate GEMM is bound by the traffic of V and M (2 x 2.25 x the image tensor each
 * way).  C must be 64, K a multiple of 16, H and W multiples of 4; x, y, U 16-byte aligned; x below 4 GiB
 * (32-bit lane offsets; FPSG_E_LIMIT otherwise -- the caller then takes the three-kernel form).
 */
int fpsg_wino_conv_fused(const float* x, const float* U, int N, int C, int K, int H, int W, float* y,
                         fpsg_stream_t stream);

/* fpsg_wino_conv_fused / _act (chan = NULL: the plain form) also accumulating the statistics of the BatchNorm that
 * follows this convolution: parts [K][fpsg_wino_conv_fused_parts(N,K,H,W)][2] = (sum(y + out_bias[k]),
 * sum((y + out_bias[k])^2)) per workgroup (out_bias optional), taken by fpsg_bn_stats / fpsg_bn_act_pool_fwd instead of
 * reading y again.  Deterministic. */
int fpsg_wino_conv_fused_parts(int N, int K, int H, int W);
int fpsg_wino_conv_fused_stats(const float* x, const float* chan, const float* pre_bias, const float* U, int N, int C,
                               int K, int H, int W, float* y, const float* out_bias, float* parts, fpsg_stream_t stream);

/* K6w: the WEIGHT gradient of the same convolutions (64 input channels) in one pass: dU [36,K,64] = sum over tiles of
 * (A dY A^T)[k] x (B^T d B)[c], both transform-domain operands formed in registers from dy [N,K,H,W] and the layer's
 * input x [N,64,H,W] (chan != NULL: x is the PRE-BatchNorm tensor and relu(fma(x + pre_bias[c], chan[c], chan[C+c]))
 * is what the convolution saw, as in fpsg_wino_conv_fused_act); dw = fpsg_wino_filter_grad_transform(4, dU).  Replaces
 * fpsg_wino_input_transform + fpsg_wino_grad_output_transform + the batched GEMM with the tile-long reduction.
 * H a multiple of 4, W of 16 (the four tiles of an MFMA step lie in one tile row); x, dy 16-byte aligned and below
 * 2 GiB each; ws: fpsg_wino_dw_fused_workspace_floats(N,K,H,W) floats (per-range partials, summed in a fixed order in
 * fp64).  Deterministic.
 */
size_t fpsg_wino_dw_fused_workspace_floats(int N, int K, int H, int W);
int fpsg_wino_dw_fused(const float* x, const float* chan, const float* pre_bias, const float* dy, int N, int C, int K,
                       int H, int W, float* dU, float* ws, fpsg_stream_t stream);

/* The same two entry points reading a PRE-BatchNorm tensor: the values fed to the transform are
 * relu(fma(x + pre_bias[c], chan[c], chan[C + c])) -- K5's apply arithmetic with chan = (scale, shift, ..) from
 * fpsg_bn_stats / fpsg_bn_act_fwd -- so that the BatchNorm + ReLU apply pass between two convolutions of a VGG
 * stage (conv -> BN -> ReLU -> conv, src/models/image_net.py:14) is folded into the second convolution's
 * load (one read + one write of the activation tensor less per layer).  pre_bias may be NULL. */
int fpsg_wino_input_transform_act(int m, const float* x, const float* chan, const float* pre_bias, int N, int C,
                                  int H, int W, float* V, long ldp, fpsg_stream_t stream);
int fpsg_wino_conv_fused_act(const float* x, const float* chan, const float* pre_bias, const float* U, int N, int C,
                             int K, int H, int W, float* y, fpsg_stream_t stream);

/* ---- K8: weight gradient of the first VGG convolution (3 -> 64 channels, 3x3, padding 1) ------
 * vgg16_bn.features[0] of src/models/image_net.py:14 in the backward of the train step:
 * dw [64,3,3,3] = sum over (n,h,w) of dy[n,k,h,w] * x[n,c,h+a-1,w+b-1], x [N,3,H,W], dy [N,64,H,W]
 * (W a multiple of 4, dy 16-byte aligned).  dy is read once (LDS-staged tiles feeding fp32 MFMA),
 * the workgroups' partial sums are reduced in a fixed order: HBM-bound, deterministic.
 * ws: fpsg_conv_first_dw_workspace_floats(N,H,W) floats.
 */
size_t fpsg_conv_first_dw_workspace_floats(int N, int H, int W);
int fpsg_conv_first_dw(const float* x, const float* dy, int N, int C, int K, int H, int W, float* dw, float* ws,
                       fpsg_stream_t stream);
/* The same weight gradient when dy is the BatchNorm + ReLU backward of the layer behind this convolution
 * (conv1_1 -> bn -> relu -> conv1_2 of src/models/image_net.py:14 in loss.backward()): instead of dy the call takes
 * the convolution's own output y [N,64,H,W], the gradient ga [N,64,H,W] of the activation relu(bn(y + pre_bias)), K5's
 * chan [4,64] (scale, shift, ..) and coef [3,64] (fpsg_bn_act_bwd_coef), and forms
 *   dy = k1 dz + k2 (y + pre_bias) + k3,  dz = ga * [(y + pre_bias) scale + shift > 0]
 * while it stages a tile (fpsg_bn_act_bwd's dx pass, bit for bit): K5's dx pass and its 475 MB dy (37 images at
 * 224 x 224) never exist.  Same result as fpsg_bn_act_bwd + fpsg_conv_first_dw, bit for bit. */
int fpsg_conv_first_dw_fold(const float* x, const float* y, const float* ga, const float* chan, const float* coef,
                            const float* pre_bias, int N, int C, int K, int H, int W, float* dw, float* ws,
                            fpsg_stream_t stream);
/* K8f: the forward of the same layer, y [N,64,H,W] = conv(x [N,3,H,W], w [64,3,3,3]), zero padding, no bias
 * (nn.Conv2d(3, 64, 3, padding=1) of vgg16_bn.features[0]; K5 adds the bias inside the BatchNorm).  Bound by the
 * write of y; per output the 27 taps are added in (c, a, b) order by an fma chain from 0.  parts (optional):
 * [64][fpsg_conv_first_parts(N,H,W)][2] partial sums of y + bias[k] (bias optional) and its square for the
 * BatchNorm that follows (fpsg_bn_stats with parts).  W a multiple of 4; x, y 16-byte aligned.  Deterministic.
 */
int fpsg_conv_first_parts(int N, int H, int W);
int fpsg_conv_first_fwd(const float* x, const float* w, int N, int C, int K, int H, int W, float* y, const float* bias,
                        float* parts, fpsg_stream_t stream);

/* ---- K7: Adam step over flat buffers ---------------------------------------------------------
 * Replaces torch.optim.Adam(lr, betas=(.9,.999)).step() of the train loop (src/trainNetwork.py:
 * 118-123, 144) when parameters, gradients and the two moments each live in one flat fp32 buffer
 * of n elements (fpsg_amd/optim.py): one 4-read / 3-write stream.  step = 1 for the first update
 * (bias corrections 1 - beta^step); grad_scale multiplies the gradient first (1/E for the mean
 * over the E episodes of a step).  amsgrad, weight decay and maximize are not part of the reference
 * configuration and not provided.  Buffers 16-byte aligned.
 */
int fpsg_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, size_t n, float lr,
                   float beta1, float beta2, float eps, int step, float grad_scale, fpsg_stream_t stream);
/* The same step with the gradient left where autograd put it: segment s of the flat buffers is
 * [seg_off[s], seg_off[s+1]) (seg_off: nseg+1 int64 on the device, seg_off[0] = 0, seg_off[nseg] = n)
 * and its gradient is the contiguous fp32 tensor at grad_ptrs[s] (device array of nseg device
 * pointers; NULL = no gradient = zero).  Saves the gather of the gradients into a flat buffer when a
 * step is one episode on one rank (the reference's own loop, trainNetwork.py:140-148).  param, exp_avg and
 * exp_avg_sq 16-byte aligned (FPSG_E_ALIGN otherwise); the gradient tensors may start anywhere.
 */
int fpsg_adam_step_segments(float* param, const float* const* grad_ptrs, const long long* seg_off, int nseg,
                            float* exp_avg, float* exp_avg_sq, size_t n, float lr, float beta1, float beta2,
                            float eps, int step, float grad_scale, fpsg_stream_t stream);
/* The gradient accumulation of a step of several episodes (optimizer.zero_grad() once, loss.backward() per
 * episode, trainNetwork.py:140-148) on the flat gradient buffer: flat[i] += the segment table's gradient
 * (accumulate = 1; segments without gradient untouched) or flat[i] = it (accumulate = 0: first episode, zeros where a
 * segment has none).  Same table format as fpsg_adam_step_segments; one stream at HBM rate.  flat 16-byte aligned.
 */
int fpsg_flat_accumulate_segments(float* flat, const float* const* grad_ptrs, const long long* seg_off, int nseg,
                                  size_t n, int accumulate, fpsg_stream_t stream);
/* The same for ntab (<= 8) tables at once, grad_ptrs [ntab][nseg]: flat (+)= g_0 + g_1 + ... added in table order -- the
 * fp32 sums of ntab consecutive fpsg_flat_accumulate_segments calls bit for bit, with flat read and written once (the
 * episodes of a step keep their gradient tensors until the step's last backward). */
int fpsg_flat_accumulate_tables(float* flat, const float* const* grad_ptrs, const long long* seg_off, int nseg, int ntab,
                                size_t n, int accumulate, fpsg_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* FPSG_HIP_H */
