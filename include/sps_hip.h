/* sps_hip.h -- C ABI of libsps_hip.so: the MI355X (gfx950) native implementation of the
 * SPS per-scan sparse-convnet hot path.
 *
 * The reference (ibrahimhroob/SPS) has no native code of its own; the native boundary it
 * crosses on this path is MinkowskiEngine's pybind11 module (MinkowskiEngineBackend._C:
 * CoordinateMapManager, ConvolutionForwardGPU, ...), reached from the Python call sites
 * cited next to each entry point below.  This header is what a maintainer binds INSTEAD of
 * MinkowskiEngine for that path (ctypes stub: INTEGRATION.md).
 *
 * Conventions
 *   - every function returns SPS_OK (0) or a negative error code; the message for the last
 *     error on the calling thread is available from sps_last_error();
 *   - no exceptions and no torch types cross the boundary: plain pointers and sizes;
 *   - pointers named *_dev are DEVICE pointers owned by the caller (torch tensors'
 *     data_ptr()); *_host are host pointers; the library owns only what lives inside ctx
 *     (arena, hash tables, weights);
 *   - one ctx per (process, device); a ctx is not thread-safe; all work is ordered on the
 *     hipStream_t passed as `stream` (void*, NULL = the default stream) and functions do
 *     not synchronise with the host unless their comment says so.
 */
#ifndef SPS_HIP_H
#define SPS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SPS_OK 0
#define SPS_ERR_INVALID (-1) /* bad argument / state                              */
#define SPS_ERR_HIP (-2)     /* a HIP runtime call failed                        */
#define SPS_ERR_NOMEM (-3)   /* device allocation failed / compact arena overflow */
#define SPS_ERR_RANGE (-4)   /* a coordinate does not fit the 64-bit voxel key   */
#define SPS_ERR_NOWEIGHTS (-5)
#define SPS_ERR_ITEMCAP (-6) /* sps_radius_item: the caller's item buffer is too small   */

#define SPS_NUM_LEVELS 5 /* tensor strides 1,2,4,8,16 (minkunet.py:161-219) */

/* Voxel-key range (sps_amd/csrc/sps_hip.hip packs (b,x,y,z,t) into 64 bits):
 * x,y,z in [-131072, 131071] voxels, t in [-16, 15], b in [0, 30]. */
#define SPS_COORD_MIN (-131072)
#define SPS_COORD_MAX (131071)
#define SPS_T_MIN (-16)
#define SPS_T_MAX (15)
#define SPS_BATCH_MAX (30)
#define SPS_MAX_POINTS (1 << 23) /* rows per forward / submap call (32-bit byte offsets into [rows,96] f32) */

typedef struct sps_ctx sps_ctx;

const char *sps_last_error(void);
int sps_version(void);

/* ---- context ------------------------------------------------------------------------ */
/* Replaces the per-forward ME coordinate manager + the module's .cuda() placement
 * (reference scripts/predict.py:59, src/sps/datasets/util.py:40). */
int sps_ctx_create(int device, sps_ctx **out);
int sps_ctx_destroy(sps_ctx *ctx);
/* Pre-size the arena for clouds of up to max_points rows (optional: sps_forward grows it
 * on demand, which synchronises the device). */
int sps_reserve(sps_ctx *ctx, int64_t max_points);
/* Arena sizing.  By default every tensor stride ("level" l = 0..4, stride 2^l) can hold as many voxels as there are
 * points: no input can overflow, at ~6.9 KB of device memory per point.  LiDAR clouds thin out quickly with the stride
 * (V_l / V_0 ~ 0.39 / 0.14 / 0.05 / 0.015 at 0.1 m), so a streaming caller may give level l only frac[l] * max_points rows
 * (frac[0] is ignored: level 0 always holds every point; blocks get half the rows' capacity): ~2.2 KB per point with
 * {1, 0.6, 0.3, 0.15, 0.08}.  A forward whose cloud needs more is ABORTED on the device (its scores are NaN) and the next
 * synchronising call (sps_check, sps_metrics) returns SPS_ERR_NOMEM after switching the context back to full-size
 * arenas; the caller re-issues it.  frac = NULL restores the default.  Takes effect at the next reserve / forward
 * (re-allocation: synchronises).  sps_arena_bytes: device bytes the context's arena holds. */
int sps_ctx_set_level_fractions(sps_ctx *ctx, const float *frac);
int64_t sps_arena_bytes(sps_ctx *ctx);
/* Inference-only context (streaming callers: sps_amd.engine.ScanEngine).  The one-column-tile 3x3x3x3 layers of levels 0 and 1
 * run on the map's RULEBOOK (per 64-row supertile and offset the compacted (output row, input row) pairs; the counterpart of
 * ME's kernel map, in / out index lists per offset), which k_maps builds next to the output-stationary neighbour table.
 * on != 0: at those levels the rulebook takes the place AND the memory of the neighbour table, which is then neither written
 * nor kept (17 MB less to store per config-2 scan, 324 B per row less arena): sps_get_nbr fails there, sps_get_map_pairs
 * counts from the rulebook, and sps_train_forward switches the context back (re-allocation).  Takes effect at the next
 * reserve / forward (re-allocation: synchronises). */
int sps_ctx_set_inference_only(sps_ctx *ctx, int on);
/* How the context's forwards are scheduled -- a hint for the launch geometry, never for the results (bit-identical either way).
 * on == 0 (default): one forward after another (a plain `model(batch)` loop, the online filter: sps_node.callback handles one scan
 * at a time, sps_node.py:88-176): the coarse levels, which hold few tiles, run one column tile per wave -- the shortest chain
 * (serial forward 366 -> 359 us at config 2).  on != 0: forwards of several contexts are in flight beside each other
 * (sps_amd.engine.ScanEngine with more than one pipeline; no reference counterpart: predict.py:64-67 is one Trainer loop): two
 * column tiles per wave -- half the gathers for the same MFMAs, +1.3-1.8 % scans/s pipelined.  Takes effect at the next forward. */
int sps_ctx_set_pipelined(sps_ctx *ctx, int on);

/* ---- weights ------------------------------------------------------------------------
 * Replaces nn.Module.load_state_dict on CustomMinkUNet (reference scripts/predict.py:56-58,
 * src/sps/datasets/util.py:33-39).  The blob is the concatenation of the tensors listed by
 * sps_weights_tensor_info in index order; names are the reference state_dict keys without
 * the "model.MinkUNet." prefix, e.g. "block2.0.conv1.kernel" [81,8,16],
 * "block2.0.downsample.0.kernel" [8,16], "bn0.bn.running_var" [8], "final.bias" [1]. */
int sps_weights_num_tensors(void);
int sps_weights_tensor_info(int idx, char *name, int name_cap, int64_t *offset, int64_t *numel);
int64_t sps_weights_numel(void);
/* Copies the blob to the device and derives the folded BatchNorm scale/shift
 * (eval mode, eps = 1e-5).  Blocking copies into fresh allocations (= sps_weights_create + sps_ctx_set_weights). */
int sps_weights_load(sps_ctx *ctx, const float *blob_host, int64_t numel);
/* A device-resident weight set that several contexts of one device share (a pipelined loop runs one context per
 * stream: the module's .cuda() happens once, reference scripts/predict.py:59, not once per stream).
 * sps_weights_create uploads (blocking copies, no device-wide synchronise); sps_ctx_set_weights attaches it to a
 * context in O(1) without any synchronisation -- forwards issued afterwards use it, forwards already issued keep
 * reading the previous set, which stays alive until its last user lets go; sps_weights_destroy drops the caller's
 * reference.  out_channels as in sps_weights_load_head below. */
typedef struct sps_weights_handle sps_weights_handle;
int sps_weights_create(int device, const float *blob_host, int64_t numel, int out_channels, sps_weights_handle **out);
int sps_weights_destroy(sps_weights_handle *w);
int sps_ctx_set_weights(sps_ctx *ctx, sps_weights_handle *w);

/* ---- forward ------------------------------------------------------------------------
 * Replaces SPSModel.forward (reference src/sps/models/models.py:20-30): quantise by
 * [1,vs,vs,vs,1] in f32, floor, unique voxels + inverse map, CustomMinkUNet (33 sparse
 * convs + eval BN + ReLU + residual + concat), slice back to points, sigmoid.
 *   coords_dev : float32 rows (b,x,y,z,t,...) with row stride `ld` floats (ld >= 5)
 *   scores_dev : float32 [n]
 * Rows whose voxel does not fit the key range get score NaN and the call that next
 * synchronises (sps_metrics / sps_check) reports SPS_ERR_RANGE.
 * NOT capturable into a HIP graph: the single-pass ranking kernels tell the forwards of a context apart by a generation
 * number that is a KERNEL ARGUMENT (host counter); a replayed graph would carry a frozen generation and match the previous
 * replay's aggregates without waiting for them.  Every forward must be issued through these entry points. */
int sps_forward(sps_ctx *ctx, const float *coords_dev, int64_t ld, int64_t n, float voxel_size,
                float *scores_dev, void *stream);
/* Synchronises `stream` and returns SPS_ERR_RANGE if any forward since the last check
 * met an unrepresentable coordinate. */
int sps_check(sps_ctx *ctx, void *stream);

/* Forward + metric sums in one call = one scan of SPSNet.predict_step (reference models.py:84-105): `batch_dev` rows
 * are (b,x,y,z,t,label,...) as BacchusModule.collate_fn builds them (ld >= 6); scores_dev [n] as sps_forward,
 * out_dev [n_batches][8] doubles as sps_metrics_dev.  The sums are accumulated by the forward's last kernel while it
 * produces the scores (no separate fill / metrics launches, no second pass over the scores). */
int sps_forward_metrics(sps_ctx *ctx, const float *batch_dev, int64_t ld, int64_t n, float voxel_size, float eps,
                        int n_batches, float *scores_dev, double *out_dev, void *stream);

/* ---- baseline heads on the same backbone ------------------------------
 * The two baselines the reference ships run the SAME CustomMinkUNet14 wiring:
 *   4DMOS  : MOS4DNet.forward  (reference c_ws/src/mos4d/scripts/mos4d.py:11-32) --
 *            CustomMinkUNet(in_channels=1, out_channels=3), constant 0.5 feature, t = scan index of
 *            a 10-scan buffer, voxel 0.2 m, returns the raw logits of column 2;
 *   MapMOS : MapMOSNet.forward (reference c_ws/src/mapmos/scripts/mapmos.py:59-83) --
 *            out_channels=1, a per-point feature (1 + normalised index) whose per-voxel mean
 *            feeds conv0 (ME UNWEIGHTED_AVERAGE), t in {0,-1}, returns raw logits.
 * sps_head_* describe the weight blob of a k-channel `final` (spec as sps_weights_*, with
 * "final.kernel" [8,k] and "final.bias" [k]); sps_weights_load_head(…, k) loads it (k = 1 is
 * sps_weights_load).  sps_forward_head:
 *   feats_dev : float32 [n] per-point input feature, or NULL for the constant 0.5
 *   t_base    : integer subtracted from floor(t) before hashing (the network is shift-invariant
 *               along t; lets a long-running scan index fit the key's t range [-16,15])
 *   out_dev   : float32 [n, out_channels] with row stride `ldo` floats (>= out_channels)
 *   activation: 0 = raw logits, 1 = sigmoid */
int sps_head_num_tensors(int out_channels);
int sps_head_tensor_info(int out_channels, int idx, char *name, int name_cap, int64_t *offset, int64_t *numel);
int64_t sps_head_numel(int out_channels);
int sps_weights_load_head(sps_ctx *ctx, const float *blob_host, int64_t numel, int out_channels);
int sps_forward_head(sps_ctx *ctx, const float *coords_dev, int64_t ld, int64_t n, float voxel_size,
                     const float *feats_dev, float t_base, float *out_dev, int64_t ldo, int activation, void *stream);

/* ---- metrics ------------------------------------------------------------------------
 * Replaces the per-scan part of SPSNet.predict_step (reference models.py:84-105) +
 * util.calculate_metrics (util.py:285-299).  For every batch index b < n_batches it
 * accumulates over the rows with t == 1 (scan rows):
 *   out[b*8+0..7] = count, TP, FP, FN, TN, sum (s-g)^2, sum g, sum g^2
 * with pred = s < eps ? 0 : 1, gt = g < eps ? 0 : 1 (float32 compares), positive = 1.
 *   batch_dev : float32 rows (b,x,y,z,t,label) with row stride ld (ld >= 6)
 * Synchronises `stream` (copies 8*n_batches doubles to out_host). */
int sps_metrics(sps_ctx *ctx, const float *scores_dev, const float *batch_dev, int64_t ld, int64_t n,
                float eps, int n_batches, double *out_host, void *stream);

/* Same accumulators written to DEVICE memory out_dev[n_batches*8] (doubles) without synchronising:
 * lets a streaming loop keep per-scan metric rows on the device and gather them once per sequence
 * (RCCL all-gather in the multi-GPU predict loop). */
int sps_metrics_dev(sps_ctx *ctx, const float *scores_dev, const float *batch_dev, int64_t ld, int64_t n,
                    float eps, int n_batches, double *out_dev, void *stream);

/* ---- variant-B submap (online path) ---------------------------------------------------
 * Replaces util.to_coords_features + util.prune (reference util.py:67-114): voxel =
 * trunc(xyz / ds) in f32; the map's unique voxel set is kept in a device-resident hash
 * (the reference rebuilds it every callback, util.py:86-89). */
int sps_map_upload(sps_ctx *ctx, const float *map_xyz_dev, int64_t ld, int64_t m, float ds, void *stream);
/* Same, from int32 voxel indices [m,3] already produced by util.to_coords_features
 * (reference util.py:75: (xyz/ds).int()), row stride ld ints. */
int sps_map_upload_voxels(sps_ctx *ctx, const int32_t *map_ijk_dev, int64_t ld, int64_t m, void *stream);
/* out_xyz_dev must hold n rows of 3 floats.  Writes the voxel corners (ix*ds, f32) of
 * (unique scan voxels INTERSECT map voxels) in scan first-occurrence order.
 * Synchronises; *n_sub = rows written, *n_scan_vox = number of unique scan voxels.
 * The float form truncates xyz/ds itself with the ds given to sps_map_upload. */
int sps_submap_voxel(sps_ctx *ctx, const float *scan_xyz_dev, int64_t ld, int64_t n, float *out_xyz_dev,
                     int64_t *n_sub, int64_t *n_scan_vox, void *stream);
int sps_submap_voxel_ijk(sps_ctx *ctx, const int32_t *scan_ijk_dev, int64_t ld, int64_t n, float ds,
                         float *out_xyz_dev, int64_t *n_sub, int64_t *n_scan_vox, void *stream);

/* ---- streaming filter (online path, stream-ordered end to end) --------------------------------
 * The per-scan body of the reference's ROS node (c_ws/src/sps_filter/scripts/sps_node.py:88-176) without the
 * transport, as four stream-ordered calls that never synchronise with the host: the row counts they produce stay in
 * a caller-owned device array counts_dev (int32[4]) that the caller reads back ONCE, after the whole scan was issued.
 *
 * sps_transform_points: util.transform_point_cloud (reference util.py:187-194; sps_node.py:103, blt_dataset.py:69-70):
 *   p' = T [p;1] with perspective divide in float64 (fused multiply-add chain over k = 0..3, the order numpy's dgemm
 *   uses: bit-identical to the reference in float64), stored as float32 (out_f64 = 0; sps_node.py:107) or float64.
 *   xyz_dev rows are float32 (in_f64 = 0) or float64 (in_f64 = 1) with row stride ld; T_host is the row-major 4x4
 *   matrix ON THE HOST (passed by value to the kernel: no copy), NULL = identity.
 * sps_filter_prepare: sps_node.py:103-117 + the tensor assembly of util.infer (util.py:163-176).  Writes into
 *   batch_dev (float32 [2n, 5], caller-owned) the rows (0, x', y', z', 1) of the transformed scan followed by the rows
 *   (0, vx, vy, vz, 0) of the variant-B submap (voxel corners of scan voxels INTERSECT map voxels, scan
 *   first-occurrence order, as sps_submap_voxel), and counts_dev[0] = n_sub, [1] = n_scan_vox, [2] = n + n_sub.
 *   Needs sps_map_upload (float form).
 * sps_forward_n: sps_forward whose row count is read from DEVICE memory (*n_dev <= n_max; grids are sized for
 *   n_max); scores_dev [n_max], rows >= *n_dev are left untouched.
 * sps_compact_stable: the epsilon filter `scan[scores <= eps]` (sps_node.py:147-148): copies, in input order, the
 *   first `cols` floats of every row i < n of rows_dev (row stride ld) whose score is <= eps to out_dev [., cols];
 *   *count_dev = rows kept (NaN scores are dropped). */
int sps_transform_points(sps_ctx *ctx, const void *xyz_dev, int in_f64, int64_t ld, int64_t n, const double *T_host,
                         void *out_dev, int out_f64, int64_t ldo, void *stream);
int sps_filter_prepare(sps_ctx *ctx, const void *raw_xyz_dev, int in_f64, int64_t ld, int64_t n, const double *T_host,
                       float *batch_dev, int32_t *counts_dev, void *stream);
int sps_forward_n(sps_ctx *ctx, const float *coords_dev, int64_t ld, int64_t n_max, const int32_t *n_dev, float voxel_size,
                  float *scores_dev, void *stream);
int sps_compact_stable(sps_ctx *ctx, const float *scores_dev, const float *rows_dev, int64_t ld, int cols, int64_t n,
                       float eps, float *out_dev, int32_t *count_dev, void *stream);

/* ---- variant-A submap (offline path) ----------------------------------------------------
 * Replaces BacchusDataset.select_closest_points (reference src/sps/datasets/blt_dataset.py:258-271:
 * scipy cKDTree.query_ball_tree(map_tree, r = VOXEL_SIZE)): for every scan point the indices of the
 * map points within Euclidean distance r (closed ball, float64), one hit list per scan point,
 * concatenated in scan-point order with duplicates kept; inside a list the hits are ordered by
 * neighbour cell ((dx+1) + 3(dy+1) + 9(dz+1)), ascending map index inside a cell (scipy's in-list
 * order is the tree's traversal order, i.e. unspecified).
 * The map is binned by the caller into cells of size cell_size >= r: cell_keys (u64, see
 * sps_amd/datasets/blt_dataset.py), cell_start [n_cells+1], cell_pts [m] (map indices grouped by cell),
 * map_xyz float64 [m,3] compact.  The ctx keeps device copies.  Synchronises. */
int sps_radius_grid_upload(sps_ctx *ctx, const uint64_t *cell_keys_dev, const int32_t *cell_start_dev,
                           const int32_t *cell_pts_dev, const double *map_xyz_dev, int64_t n_cells, int64_t m,
                           double cell_size, double r, void *stream);
/* counts_dev[i*27 + c] = number of hits of scan point i in its neighbour cell c (float64 rows xyz...,
 * row stride ld); counts_dev holds 27*n ints. */
int sps_radius_count(sps_ctx *ctx, const double *scan_xyz_dev, int64_t ld, int64_t n, int32_t *counts_dev, void *stream);
/* Writes the hits of (point i, cell c) at out_idx_dev[offsets_dev[i*27 + c] ...] (offsets = exclusive prefix
 * sum of the 27*n counts, computed by the caller), so a point's list is contiguous. */
int sps_radius_fill(sps_ctx *ctx, const double *scan_xyz_dev, int64_t ld, int64_t n, const int64_t *offsets_dev,
                    int64_t *out_idx_dev, void *stream);

/* A second context of the same device uses the owner's radius grid without a copy (a view: the owner keeps and frees
 * the allocations and must outlive it).  The S pipelined contexts of one evaluation loop share ONE grid this way. */
int sps_radius_grid_attach(sps_ctx *ctx, sps_ctx *owner);
/* The whole offline item on the device, stream-ordered, no host synchronisation: replaces BacchusDataset.__getitem__
 * (reference src/sps/datasets/blt_dataset.py:209-244: scan rows + add_timestamp + select_closest_points + map rows) and
 * the batch column of BacchusModule.collate_fn (:173-182).  scan_dev rows are (x, y, z, label) in the scan's own dtype
 * (float64: in_f64 = 1, else float32, promoted to float64 for the radius test as cKDTree does), row stride ld >= 4.
 * Writes into rows_dev (float32 [row_cap, ldo], ldo >= 6), starting at row *row_off_dev (NULL = 0):
 *   n rows (b, x, y, z, 1, label)   then   m rows (b, mx, my, mz, 0, 1)
 * with b = batch_index and the m map points within r of some scan point (one list per scan point, duplicates kept, as
 * sps_radius_count / sps_radius_fill), and *n_rows_dev = *row_off_dev + n + m: chaining calls with
 * row_off_dev = n_rows_dev appends the items of a batch.  Rows beyond row_cap are dropped and the next synchronising
 * call (sps_check) returns SPS_ERR_ITEMCAP. */
int sps_radius_item(sps_ctx *ctx, const void *scan_dev, int in_f64, int64_t ld, int64_t n, float batch_index,
                    const int32_t *row_off_dev, float *rows_dev, int64_t ldo, int64_t row_cap, int32_t *n_rows_dev,
                    void *stream);
/* sps_forward_metrics whose row count is read from DEVICE memory (*n_dev <= n_max; grids are sized for n_max): the
 * consumer of sps_radius_item in the offline loop (reference scripts/predict.py:64-67 -> models.py:84-111). */
int sps_forward_metrics_n(sps_ctx *ctx, const float *batch_dev, int64_t ld, int64_t n_max, const int32_t *n_dev,
                          float voxel_size, float eps, int n_batches, float *scores_dev, double *out_dev, void *stream);

/* ---- training step ---------------------------------------------------------
 * Replaces the forward + loss.backward() of SPSNet.training_step / common_step (reference
 * src/sps/models/models.py:62-82) for the network part; the loss (nn.MSELoss on the scan rows) and the optimiser
 * (Adam + StepLR, models.py:154-160) stay with the caller.
 * sps_train_forward: params_dev is the flat parameter blob ON THE DEVICE in the layout of sps_weights_tensor_info
 *   (kernels [K][C_in][C_out], BN weight / bias / running_mean / running_var, final.bias); the running statistics are
 *   ignored: BatchNorm normalises with the batch statistics of the active rows (train mode, eps = 1e-5) and
 *   batch_stats_dev (optional, [3 * sum of BN widths] floats: per conv in layer order the batch mean, the BIASED
 *   and the UNBIASED batch variance of its BN) lets the caller update running_mean / running_var as nn.BatchNorm1d does.  scores_dev [n]
 *   = sigmoid(final(...)) sliced to the points, as sps_forward.  The activations the backward needs stay in ctx.
 * sps_train_backward: dscores_dev [n] = d(loss)/d(scores); scores_dev the forward's output; grad_dev receives
 *   d(loss)/d(parameter) in the blob layout (zeros in the running-statistics slots).  Deterministic (fixed-order
 *   reductions).  Stream-ordered, no host synchronisation. */
int sps_train_forward(sps_ctx *ctx, const float *params_dev, int64_t numel, const float *coords_dev, int64_t ld, int64_t n,
                      float voxel_size, float *scores_dev, float *batch_stats_dev, void *stream);
int sps_train_backward(sps_ctx *ctx, const float *dscores_dev, const float *scores_dev, float *grad_dev, int64_t numel,
                       void *stream);
/* The activations, kernel maps and parameter copy a backward reads live in ctx and belong to ONE forward: every forward
 * of the context (training or inference) bumps its generation.  sps_train_generation returns the generation of the
 * training forward whose activations are held (what torch keeps in the autograd node, sps_amd/models/models.py
 * ::_TrainForward); sps_train_backward_at is sps_train_backward that fails with SPS_ERR_INVALID when that forward's
 * activations have been overwritten by a later forward (two graphs alive on one context, an evaluation forward
 * issued mid-step) instead of returning gradients of the wrong forward.  sps_train_backward itself refuses a backward
 * after an intervening inference forward.
 * Train-mode BatchNorm needs more than one active row per level (nn.BatchNorm1d raises "Expected more than 1 value per
 * channel when training", reference resnet.py:100-107): a training forward whose coarsest level has a single voxel sets a
 * sticky flag and the next synchronising call (sps_check) returns SPS_ERR_INVALID. */
int sps_train_generation(sps_ctx *ctx, int64_t *generation);
int sps_train_backward_at(sps_ctx *ctx, int64_t generation, const float *dscores_dev, const float *scores_dev,
                          float *grad_dev, int64_t numel, void *stream);

/* The loss of common_step (reference src/sps/models/models.py:62-72): nn.MSELoss between the scores and the labels of the
 * rows whose time index is 1 (the scan; the reference selects them with np.where on the host), and the R2Score the same
 * step logs.  sps_scan_mse: scores_dev [n]; labels_dev / t_dev point at the label / time column of the batch rows (row
 * strides ld_labels / ld_t in floats); work_dev [4 * 257] doubles (scratch the backward reads: work[0] = the number of
 * selected rows); out_dev [2] floats = loss, R2.  sps_scan_mse_backward: dscores_dev [n] = gloss * d loss / d scores
 * (gloss_dev: one float on the device).  f64 sums in a fixed order; stream-ordered, no host synchronisation; no context
 * (the caller has made the device current). */
int sps_scan_mse(const float *scores_dev, const float *labels_dev, int64_t ld_labels, const float *t_dev, int64_t ld_t, int64_t n,
                 double *work_dev, float *out_dev, void *stream);
int sps_scan_mse_backward(const float *scores_dev, const float *labels_dev, int64_t ld_labels, const float *t_dev, int64_t ld_t,
                          int64_t n, const double *work_dev, const float *gloss_dev, float *dscores_dev, void *stream);

/* ---- per-stage timing (hipEvents on the caller's stream; for bench.py / DESIGN.md) ------
 * With profiling on, sps_forward records one event after every stage ("reset", "voxelize",
 * "pyramid", "maps", one per convolution by state_dict name, "slice_sigmoid").  After a forward,
 * sps_profile_count gives the number of stages and sps_profile_read(i) synchronises on stage i's
 * closing event and returns its name and duration in milliseconds.  Replaces the reference's
 * time.time() deltas (util.py:164,182; sps_node.py:164-176). */
int sps_profile_enable(sps_ctx *ctx, int on);
int sps_profile_count(sps_ctx *ctx);
int sps_profile_read(sps_ctx *ctx, int idx, char *name, int name_cap, float *ms);
/* The kernel (class) stage idx launched, e.g. "k_conv_px", "k_conv<3x3x3x3, levels 2-4>", "k_upconv", "k_maps": lets
 * bench.py aggregate the stages per kernel the way `rocprofv3 --stats` does. */
int sps_profile_kernel(sps_ctx *ctx, int idx, char *name, int name_cap);

/* ---- introspection (parity tests; all synchronise) ----------------------------------- */
/* Number of active voxels at each tensor stride of the last forward. */
int sps_level_counts(sps_ctx *ctx, int64_t counts_host[SPS_NUM_LEVELS]);
/* Integer coordinates [V,5] (b,x,y,z,t) of level `level` (0 = tensor stride 1). */
int sps_get_voxels(sps_ctx *ctx, int level, int32_t *coords_dev);
/* inverse map point -> ts1 voxel row, int64 [n] (-1 for out-of-range rows). */
int sps_get_inverse(sps_ctx *ctx, int64_t *inv_dev);
/* parent row (level+1) of each voxel of `level` (0..3), int32 [V_level]. */
int sps_get_parent(sps_ctx *ctx, int level, int32_t *parent_dev);
/* Kernel-map pair counts: which = 0..4 -> 3x3x3x3 map at level which; 5 -> 5x5x5x1 map at
 * level 0.  pairs_host[k] = number of (in,out) pairs of offset k (81 or 125 entries). */
int sps_get_map_pairs(sps_ctx *ctx, int which, int64_t *pairs_host);
/* Present-offset masks of the 16-row output tiles of a kernel map (which as above): uint32
 * [n_tiles][4].  which = 0..4 (3x3x3x3): one word per time slice, offset k = 27 * word + bit (word 3 unused);
 * which = 5: bit k of the 128-bit field.  A set bit = some row of the tile has a neighbour through offset k. */
int sps_get_tile_masks(sps_ctx *ctx, int which, uint32_t *masks_dev, int64_t *n_tiles);
/* The 3x3x3x3 neighbour table of level `which` (0..4), int32 [81][V] compact; entries of (tile, k)
 * pairs whose tile-mask bit is clear are unspecified (never written, never read by the convolution). */
int sps_get_nbr(sps_ctx *ctx, int which, int32_t *nbr_dev);
/* A kernel map of the last forward as ME holds it (the set of (input row, output row) pairs per kernel offset; reached from
 * /root/reference/src/sps/models/MinkowskiEngine/minkunet.py:162-217), exported as a dense int32 table [K][V_out] compact:
 * out[k * V_out + u] = input row paired with output row u through offset k, or -1.
 *   which = 0..4: 3x3x3x3 map of level which (K = 81, k = (dx+1) + 3(dy+1) + 9(dz+1) + 27(dt+1));
 *   which = 5:    5x5x5x1 map of level 0 (K = 125; materialised for this call -- conv0 derives the same presence bits from the
 *                 block tables without ever storing the map);
 *   which = 6..9: stride-2 map into coarse level which - 5 (K = 8, k = dx + 2dy + 4dz; rows = coarse voxels, entries = fine
 *                 rows; the transposed convolutions read the same table with the roles swapped).
 * source = 0: the output-stationary neighbour / child table; source = 1 (which = 0..4 at the pair-exact levels): decoded from
 * the RULEBOOK the pair-exact convolutions read; *n_entries (may be NULL) = number of pairs it holds (malformed or duplicate
 * entries fail the call). */
int sps_get_kernel_map(sps_ctx *ctx, int which, int source, int32_t *out_dev, int64_t *n_entries);
/* Per-voxel logits of the last forward, float32 [V_0]. */
int sps_get_logits(sps_ctx *ctx, float *logits_dev);
/* Named intermediate feature maps: "out_p1","block1".."block8"; copies [V,C] row-major
 * f32 (compact, ld = C) into out_dev; *rows,*cols describe it. */
int sps_get_feature(sps_ctx *ctx, const char *name, float *out_dev, int64_t *rows, int64_t *cols);

#ifdef __cplusplus
}
#endif
#endif /* SPS_HIP_H */
