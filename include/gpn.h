/*
 * gpn.h — C-ABI of libgpn_hip.so: the MI355X (gfx950) implementation of the GAPartNet
 * sparse-3D-conv perception hot path.
 *
 * Boundary contract:
 *   - extern "C", plain pointers and sizes, no torch types.  Every pointer is a DEVICE pointer
 *     unless the parameter name ends in `_host`.
 *   - the caller allocates every output and the workspace (query `*_ws_bytes`); kernels never allocate.
 *   - all work is enqueued asynchronously on `stream` (a hipStream_t passed as void*).
 *   - return value 0 = ok; otherwise an error code, message via gpn_last_error().  The library never
 *     calls exit() (the vendored reference lib does: ball_query_gpu.cu:62-66) so DDP ranks survive.
 *   - data-dependent sizes are written to a device counter; the caller sizes buffers by the stated
 *     upper bound and reads the counter when it needs the value on the host.
 *
 * Each entry point cites the reference interface it replaces (file:line relative to the
 * PKU-EPIC/GAPartNet tree).
 */
#ifndef GPN_H
#define GPN_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* gpn_stream_t; /* hipStream_t */

#define GPN_OK 0
#define GPN_ERR_ARG 1      /* bad argument (null pointer, unsupported size) */
#define GPN_ERR_WS 2       /* workspace too small */
#define GPN_ERR_HIP 3      /* a HIP runtime call / kernel launch failed */

const char* gpn_last_error(void);
int gpn_version(void);
/* number of exported compute entry points and their names (used by the symbol test) */
int gpn_num_entry_points(void);
const char* gpn_entry_point_name(int i);

/* ---- H: dense heads: y = x W^T + b on [N, cin] rows, cin % 4 == 0, cin, cout <= 64 (csrc/linear.hip) ----------------------
 * Replaces torch.nn.functional.linear of network/model.py:140, 159-160, 322, 337 (sem_seg_head, offset_head, score_head,
 * npcs_head) and its autograd: W [cout, cin] in torch.nn.Linear's layout, b [cout] or NULL.  One launch forward; backward
 * writes dx [N, cin], dW [cout, cin], db [cout] (each may be NULL = skipped), deterministic (fixed-order sums). */
int gpn_linear_supported(int cin, int cout);
int gpn_linear_fwd(const float* x, const float* W, const float* b, int64_t N, int cin, int cout, float* y, gpn_stream_t stream);
size_t gpn_linear_bwd_ws_bytes(int64_t N, int cin, int cout);
int gpn_linear_bwd(const float* x, const float* W, const float* dy, int64_t N, int cin, int cout, float* dx, float* dW, float* db,
                   void* ws, size_t ws_bytes, gpn_stream_t stream);

/* ---- optional in-library kernel timing (hipEvent pairs around launches; used by bench.py) ------- */
/* kernel ids for gpn_prof_get */
enum {
  GPN_K_SPCONV_FWD = 0, /* fused gather-MFMA-scatter conv (forward and dgrad launches) */
  GPN_K_SPCONV_WGRAD = 1,
  GPN_K_VOXELIZE = 2,
  GPN_K_RULEBOOK = 3,
  GPN_K_BALL_QUERY = 4,
  GPN_K_CCL = 5,
  GPN_K_BN = 6, /* BatchNorm passes (statistics where not taken by a conv epilogue, apply forward / backward) */
  GPN_K_LINEAR = 7, /* the dense heads (section H) */
  GPN_K_COUNT = 8
};
/* fixed cost of a (start event, launch, stop event) bracket, measured around an empty kernel on `stream` (median, us):
 * subtract it from hipEvent-measured launch durations before comparing them with a profiler's kernel durations. */
int gpn_prof_bracket_overhead_us(gpn_stream_t stream, double* overhead_us);
int gpn_prof_enable(int on);
int gpn_prof_reset(void);
/* synchronises the recorded events; returns launches, total milliseconds, algorithmic flops and bytes */
int gpn_prof_get(int kernel_id, int64_t* launches_host, double* ms_host, double* flops_host,
                 double* bytes_host);
/* the same measurements launch by launch, in launch order (first `cap` records; *count_host = how many there are).  tag of a
 * conv fwd / dgrad launch: bit 62 = two problems in the launch (paired pass), bits 48-53 taps K, 40-47 cin / 16, 32-39
 * cout / 16, 0-31 rows of the launch's output; 0 = untagged. */
int gpn_prof_get_launches(int kernel_id, int64_t cap, double* ms_host, int64_t* tag_host, int64_t* count_host);

/* ================================================================================================
 * V — voxelize.   replaces epic_ops.voxelize.voxelize (call sites dataset/gapartnet.py:188-195,
 * network/grouping_utils.py:93-101).
 *   points [M,3] f32, feats [M,C] f32, seg_offsets [S+1] i64 (CSR over points),
 *   seg_range_min/max [S,3] f32 (per segment; the reference passes one range per call — the
 *   host wrapper broadcasts it), voxel_size_host[3], grid_dims_host[3] = cells per axis used to
 *   linearise keys (points whose cell index >= grid_dims are dropped like out-of-range points).
 *   coord = floor((p - range_min) / voxel_size) evaluated in fp32 without contraction.
 * outputs (capacity M rows each): voxel_feats [V,C] (mean, summed in ascending point order),
 *   voxel_coords [V,3] i32, voxel_seg [V] i32, pc_voxel_id [M] i32 (-1 = dropped),
 *   num_voxels [1] i64.  Voxels are ordered by ascending (segment,x,y,z).
 * ================================================================================================ */
size_t gpn_voxelize_ws_bytes(int64_t M, int C);
int gpn_voxelize(const float* points, const float* feats, const int64_t* seg_offsets,
                 const float* seg_range_min, const float* seg_range_max, int64_t M, int C, int64_t S,
                 const float* voxel_size_host, const int32_t* grid_dims_host, float* voxel_feats,
                 int32_t* voxel_coords, int32_t* voxel_seg, int32_t* pc_voxel_id, int64_t* num_voxels,
                 void* ws, size_t ws_bytes, gpn_stream_t stream);

/* same, plus the CSR of points grouped by voxel: point_order [M] i32 (point ids sorted by voxel,
 * ascending point id inside a voxel; dropped points last) and voxel_point_start [M+1] i32 (first V+1
 * entries valid).  Either may be NULL.  Feeds gpn_scatter_rows_csr (the gather's transpose). */
int gpn_voxelize_ex(const float* points, const float* feats, const int64_t* seg_offsets,
                    const float* seg_range_min, const float* seg_range_max, int64_t M, int C, int64_t S,
                    const float* voxel_size_host, const int32_t* grid_dims_host, float* voxel_feats,
                    int32_t* voxel_coords, int32_t* voxel_seg, int32_t* pc_voxel_id, int64_t* num_voxels,
                    int32_t* point_order, int32_t* voxel_point_start, void* ws, size_t ws_bytes,
                    gpn_stream_t stream);

/* scene batches with the reference's per-scene conventions (dataset/gapartnet.py:179-205: range = [min - 1e-4, max + 1e-4]
 * of each scene, cell = floor((p - range_min) / voxel_size)) WITHOUT a host read before or between the launches: per-scene
 * range reduced on the device, keys packed with 10 bits per axis, and ONE read of `stats` [8 + n_levels] i64 afterwards:
 *   [0] #voxels, [1..3] largest cell index per axis (spatial extent = max(that + 1, 128)), [4] dropped points,
 *   [5] != 0: a cell index >= 1024 occurred (results incomplete: use gpn_voxelize_ex with the true extent),
 *   [8 + l] rows of stride-2 level l + 1 below the voxel set (= gpn_rulebook_level_counts), l < n_levels.
 * indices4 [M,4] i32 = (segment, x, y, z) per voxel; the other outputs as gpn_voxelize_ex; same order, same means. */
/* round 5: gpn_voxelize_scenes runs WITHOUT a sort (BASELINE.json "hash-table voxelization"): cell occupancy in a bitmap laid out in
 * (segment, x, y, z) order over the batch's grid (cells per axis reduced on the device), voxel id = popcount rank of the cell's bit,
 * points grouped by a counting placement + an ascending sort of every voxel's own few points.  Bit-identical outputs.  stats[5] = 2:
 * the batch's grid exceeds the bitmap (2^27 cells) - like stats[5] = 1 (a cell index >= 1024) the caller takes another path:
 * gpn_voxelize_scenes_sorted (the stable radix sort of 10-bit-per-axis packed keys; same contract, same workspace query). */
size_t gpn_voxelize_scenes_ws_bytes(int64_t M, int C, int64_t S, int n_levels);
int gpn_voxelize_scenes_sorted(const float* points, const float* feats, const int64_t* seg_offsets, int64_t M, int C, int64_t S,
                               const float* voxel_size_host, int n_levels, float* voxel_feats, int32_t* indices4,
                               int32_t* pc_voxel_id, int32_t* point_order, int32_t* voxel_point_start, int64_t* stats, void* ws,
                               size_t ws_bytes, gpn_stream_t stream);
int gpn_voxelize_scenes(const float* points, const float* feats, const int64_t* seg_offsets, int64_t M, int C, int64_t S,
                        const float* voxel_size_host, int n_levels, float* voxel_feats, int32_t* indices4,
                        int32_t* pc_voxel_id, int32_t* point_order, int32_t* voxel_point_start, int64_t* stats, void* ws,
                        size_t ws_bytes, gpn_stream_t stream);

/* ================================================================================================
 * K1/K2 — rulebooks.   replace the indice-pair construction inside spconv.pytorch.SubMConv3d /
 * SparseConv3d / SparseInverseConv3d (call sites network/backbone.py:19-36,74-90,149-152).
 *
 * A rulebook is a set of K pair lists, stored flat and ordered by (tap k, dst row):
 *   pair_src [P] i32, pair_dst [P] i32, tile_off [K, n_tiles+1] i32 where
 *   tile_off[k][t] = index of the first pair of tap k whose dst >= t*GPN_TILE_ROWS
 *   (so tile_off[k][0] is the start of list k and tile_off[k][n_tiles] its end).
 * Each dst row appears at most once per tap.  n_tiles = ceil(n_dst / GPN_TILE_ROWS).
 * ================================================================================================ */
#define GPN_TILE_ROWS 32

/* SubM k=3 pad=1: indices [N,4] i32 (batch,x,y,z); tap = (dx+1)*9+(dy+1)*3+(dz+1); src = row at
 * coord(dst)+delta.  pair arrays need capacity 27*N.  num_pairs [1] i64.
 * nbr (optional, capacity 27*N+1): the tap-major neighbour table nbr[k*N + dst] = src row or -1 that the fused
 * conv kernel (gpn_spconv_fwd) gathers from; the pair lists are its compaction and feed gpn_spconv_wgrad. */
size_t gpn_rulebook_subm3_ws_bytes(int64_t N);
int gpn_rulebook_subm3(const int32_t* indices, int64_t N, const int32_t* spatial_shape_host, int32_t* nbr,
                       int32_t* pair_src, int32_t* pair_dst, int32_t* tile_off, int64_t* num_pairs,
                       void* ws, size_t ws_bytes, gpn_stream_t stream);

/* tile order for the fused conv (optional, large levels): rows of every block of block_rows (power of two) consecutive
 * rows sorted by their neighbour mask (which of the K <= 32 taps exist), stable.  perm [ceil(n/16)*16 + 16] i32 (padding
 * = n-1), nbr_p [K*n + 1] i32 = nbr with its columns in that order. */
size_t gpn_rulebook_tile_order_ws_bytes(int64_t n);
int gpn_rulebook_tile_order(const int32_t* nbr, int K, int64_t n, int block_rows, int32_t* perm, int32_t* nbr_p,
                            void* ws, size_t ws_bytes, gpn_stream_t stream);
/* k=2 stride=2 down-conv.  out shape = floor(D/2) per axis; inputs mapping outside are dropped.
 * Produces the coarse index set (ascending linear key; capacity N rows), fine_to_coarse [N] i32
 * (-1 = dropped), tap [N] i32 = (x&1)*4+(y&1)*2+(z&1), num_out [1] i64.
 * The two pair-list views (dst = coarse for the down conv / dgrad of the inverse conv;
 * dst = fine for the inverse conv / dgrad of the down conv) are then built with
 * gpn_rulebook_down_lists once num_out is known on the host. */
size_t gpn_rulebook_down_ws_bytes(int64_t N);
int gpn_rulebook_down(const int32_t* indices, int64_t N, int64_t batch_size,
                      const int32_t* spatial_shape_host, int32_t* out_indices, int32_t* fine_to_coarse, int32_t* tap, int64_t* num_out,
                      void* ws, size_t ws_bytes, gpn_stream_t stream);
/* the K = 1 rulebook of a SubMConv3d(k=1) (network/backbone.py:19-20, the residual blocks' shortcut) / linear layer over n rows
 * (every row its own neighbour): rows [max(n,1)] (serves as
 * pair_src and pair_dst), tile_off [n_tiles + 1], nbr [n + 1], num_pairs [1] - one launch instead of seven torch ops */
int gpn_rulebook_identity(int64_t n, int32_t* rows, int32_t* tile_off, int32_t* nbr, int64_t* num_pairs, gpn_stream_t stream);
/* row counts of ALL coarse levels below a voxel set in one pass (what n_levels successive gpn_rulebook_down calls - the
 * SparseConv3d(k=2,s=2) of every UBlock, network/backbone.py:74-77 - would report in num_out): counts [n_levels] i64 on the device, so that a U-Net's rulebook pyramid costs ONE host read instead of
 * one per level; gpn_rulebook_down itself is unchanged (its num_out is then only a cross-check).  n_dev (optional, device):
 * the number of valid rows when indices is an upper-bound buffer. */
size_t gpn_rulebook_level_counts_ws_bytes(int64_t n_max, int n_levels);
int gpn_rulebook_level_counts(const int32_t* indices, int64_t n_max, const int64_t* n_dev, int64_t batch_size,
                              const int32_t* spatial_shape_host, int n_levels, int64_t* counts, void* ws, size_t ws_bytes,
                              gpn_stream_t stream);
size_t gpn_rulebook_down_lists_ws_bytes(int64_t N, int64_t n_out);
int gpn_rulebook_down_lists(const int32_t* fine_to_coarse, const int32_t* tap, int64_t N,
                            int64_t n_out,
                            int32_t* fwd_nbr /* [8*n_out+1] or NULL */, int32_t* fwd_src, int32_t* fwd_dst,
                            int32_t* fwd_tile_off, /* dst = coarse, cap N */
                            int32_t* bwd_nbr /* [8*N+1] or NULL */, int32_t* bwd_src, int32_t* bwd_dst,
                            int32_t* bwd_tile_off, /* dst = fine, cap N */
                            int64_t* num_pairs, void* ws, size_t ws_bytes, gpn_stream_t stream);

/* ================================================================================================
 * C — sparse convolution.  replaces the conv forward/backward inside spconv (network/backbone.py).
 * weights: canonical layout W [K, Cin, Cout] row-major fp32 (tap-major).
 * gpn_spconv_pack_weights writes the MFMA-fragment layout the kernels read:
 *   flags bit0 = transpose (use W_k^T: Cin/Cout swap), bit1 = reverse taps (k -> K-1-k); dgrad of
 *   a SubM conv uses both, dgrad of down/inverse convs uses transpose only.
 *   cin/cout here are the dims of the *packed* operator (after the optional transpose); both must be
 *   multiples of 16.  packed size = K*cin*cout floats.
 * gpn_spconv_fwd: out[dst] = sum_k in[nbr[k][dst]] @ W_k over the tap-major neighbour table of a rulebook
 *   (nbr [K, n_dst] i32, -1 = no neighbour; out is fully overwritten).  in [n_src, cin], out [n_dst, cout].
 * gpn_spconv_wgrad: dW[k] = sum_{pairs of k} in[src]^T (x) dout[dst]  -> dW [K, cin, cout] canonical.
 * ================================================================================================ */
#define GPN_PACK_TRANSPOSE 1
#define GPN_PACK_REVERSE 2
#define GPN_LAYOUT_OKI 4 /* weights stored as the spconv-2.x parameter [Cout][K][Cin] instead of canonical [K][Cin][Cout] */
int gpn_spconv_pack_weights(const float* W, int K, int cin_w, int cout_w, int flags, float* packed,
                            gpn_stream_t stream);
size_t gpn_spconv_fwd_ws_bytes(int K, int64_t n_dst, int cin, int cout);
int gpn_spconv_fwd(const float* in, const float* packed_w, const int32_t* nbr, int K, int64_t n_dst, int cin,
                   int cout, float* out, void* ws, size_t ws_bytes, gpn_stream_t stream);
/* pack + conv in one call (packed copy lives in the head of ws); cin_w/cout_w are the STORED weight's dims */
/* the same conv over a rulebook that carries a TILE ORDER (gpn_rulebook_tile_order): nbr_p = the neighbour table in
 * that order, perm = the destination row of every tile position.  Results are identical (a row's taps are summed in
 * the same order); tiles whose rows share their neighbour mask simply skip more taps.  nbr_p / perm may both be NULL. */
int gpn_spconv_fwd_ordered(const float* in, const float* packed_w, const int32_t* nbr, const int32_t* nbr_p,
                           const int32_t* perm, int K, int64_t n_dst, int cin, int cout, float* out, void* ws,
                           size_t ws_bytes, gpn_stream_t stream);
/* kernel selection knob: layers of >= min_tiles 16-row tiles run on the masked-tile kernel (csrc/spconv_tiles.hip), smaller
 * ones on the direct / lock-step kernels; results are identical.  min_tiles < 0 queries.  Returns the previous value. */
int64_t gpn_spconv_tiles_min_tiles(int64_t min_tiles);
/* layers with few (16-row tile, 16-column tile) units run the direct kernel in a tap-split form (2 or 4 waves per unit,
 * partial sums added in LDS in wave order; csrc/spconv_fwd.hip): thresholds in units, negative = unchanged, 0 = never. */
int gpn_spconv_direct_split(int64_t split4_below_units, int64_t split2_below_units);
/* kernel selection knob (round 6): k = 27 / 8 layers below the masked-tile kernel's size run on the masked tap-split kernel
 * (csrc/spconv_msplit.hip: a workgroup per row tile, its waves own tap ranges, dead taps skipped); mode 0 hands them back to the
 * direct kernel, mode < 0 leaves the setting.  force_nt (1 - 4, a divisor of the layer's column tiles) / force_sp (4 or 9) fix the
 * column tiles per workgroup / the waves per row tile for every layer instead of the built-in table (0 = table, < 0 = unchanged):
 * measurement and test knob.  Returns the previous mode. */
int gpn_spconv_msplit(int mode, int force_nt, int force_sp);
size_t gpn_spconv_fwd_w_ws_bytes(int K, int64_t n_dst, int cin, int cout);
int gpn_spconv_fwd_w(const float* in, const float* W, int K, int cin_w, int cout_w, int pack_flags, const int32_t* nbr,
                     int64_t n_dst, float* out, void* ws, size_t ws_bytes, gpn_stream_t stream);
size_t gpn_spconv_wgrad_ws_bytes(int K, int cin, int cout, int64_t n_dst);
int gpn_spconv_wgrad(const float* in, const float* dout, const int32_t* pair_src,
                     const int32_t* pair_dst, const int32_t* tile_off, int K, int64_t n_dst, int cin,
                     int cout, int flags /* GPN_LAYOUT_OKI: write dW as [Cout][K][Cin] */, float* dW, void* ws,
                     size_t ws_bytes, gpn_stream_t stream);

/* BN — BatchNorm1d over a feature matrix [N, C] fused with the residual add and ReLU that follow it in every block of
 * the reference network (network/backbone.py:40-49 relu(bn(conv(x)) [+ shortcut]); norm_fn = BatchNorm1d(eps=1e-4,
 * momentum=0.1), network/model.py:86).  C % 4 == 0.  res may be NULL; relu = 0/1.
 * train: batch statistics (biased variance) -> mean / invstd [C] outputs (saved for backward); running_mean/var (may be
 *   NULL) are updated in place with `momentum` and the unbiased variance, as torch.nn.BatchNorm1d does.
 * eval: caller passes mean = running_mean and invstd = 1/sqrt(running_var + eps).
 * bwd: y = forward output (ReLU mask), dy = its gradient -> dx, dres (NULL if no residual), dweight, dbias. */
size_t gpn_bn_ws_bytes(int64_t N, int C);
int gpn_bn_fwd_train(const float* x, const float* res, const float* weight, const float* bias, int64_t N, int C,
                     float eps, float momentum, int relu, float* y, float* mean, float* invstd, float* running_mean,
                     float* running_var, void* ws, size_t ws_bytes, gpn_stream_t stream);
int gpn_bn_fwd_eval(const float* x, const float* res, const float* weight, const float* bias, const float* mean,
                    const float* invstd, int64_t N, int C, int relu, float* y, gpn_stream_t stream);
int gpn_bn_bwd(const float* x, const float* y, const float* dy, const float* weight, const float* mean,
               const float* invstd, int64_t N, int C, int relu, int training, float* dx, float* dres, float* dweight,
               float* dbias, void* ws, size_t ws_bytes, gpn_stream_t stream);

/* G — row gather voxels->points and its deterministic transpose (model.py:153,359,394).
 * out[i] = idx[i] >= 0 ? table[idx[i]] : 0.   bwd: dtable[r] = sum_{i: idx[i]==r} dout[i] using the
 * CSR (order,starts) of points grouped by row (ascending point order inside a row). */
int gpn_gather_rows(const float* table, const int32_t* idx, int64_t n, int C, float* out,
                    gpn_stream_t stream);
int gpn_scatter_rows_csr(const float* dout, const int32_t* order, const int32_t* starts,
                         int64_t n_rows, int C, float* dtable, gpn_stream_t stream);

/* ================================================================================================
 * U — layer-program executor for the sparse residual U-Net (network/backbone.py:8-165: ResBlock.forward 40-49,
 * UBlock.forward 126-141, SparseUNet.forward 150-155).  The reference walks ~200 spconv / BatchNorm modules per
 * forward from Python; here the host side describes the same walk once as a flat op list and one call runs every
 * conv / BN / concat launch of the forward (or of the backward, in reverse, accumulating gradients), so the step is
 * not bound by per-layer interpreter overhead.  All arrays are host memory; every pointer inside them is device memory.
 *   slot     : an activation matrix [rows, channels] (data) and, for backward, its gradient buffer (grad)
 *   rulebook : neighbour table of the map and of its transpose (dgrad), plus the (tap,dst)-ordered pair lists (wgrad)
 *   conv     : weight in parameter layout [Cout][K][Cin] (GPN_LAYOUT_OKI) and where its gradient goes
 *   bn       : BatchNorm1d parameters / running stats / saved batch stats / gradients
 *   op       : CONV dst = conv(src0)        | BN dst = act(bn(src0) [+ src1])   | CONCAT dst = [src0 | src1]
 * forward: training == 1 uses batch statistics (saved into save_mean / save_invstd, running stats updated);
 *          training == 0 uses the running statistics (save_invstd receives 1/sqrt(var+eps));
 *          training == GPN_NET_INFERENCE (round 6): running statistics, and the caller promises that NO backward pass follows -
 *          every BatchNorm that directly follows a conv is then applied in that conv launch's epilogue (same arithmetic per
 *          element, bit-equal outputs; the conv's own output slot and save_invstd stay unwritten): a validation step of the
 *          full model has no BatchNorm launch left but the proposal networks' first one (network/backbone.py:40-49).
 * backward: slots[].grad of the final slot holds d(loss)/d(output); slots[].grad_state must be 0 everywhere except
 *          slots that already hold a gradient (1).  Gradients of multiply-consumed slots are summed in program order
 *          (reverse), so results are deterministic.  need_input_grad = 0 skips d/d(slot 0). */
typedef struct gpn_net_slot {
  float* data;
  float* grad;
  int64_t rows;
  int32_t channels;
  int32_t grad_state;
  /* Row counts that are still on the device (round 4: a pass issued without a host read of its sizes).  rows_dev != NULL:
   * `rows` is the BOUND the buffers (and the rulebook tables) are allocated for and *rows_dev the live row count, written by an
   * earlier launch on the same stream; every kernel of the pass reads it and walks its rows with a grid stride, so no result
   * depends on the host's knowledge.  rows_plan (> 0) is the host's ESTIMATE of that count (e.g. the previous step's): it only
   * sizes grids and picks kernel variants.  Tables of such a pass use the LIVE count as their leading dimension (nbr is
   * [K][*rows_dev], tile_off [K][ceil(*rows_dev / 32) + 1]) - exactly the layout of an exactly-sized pass, in larger buffers.
   * rows_dev == NULL: `rows` is the exact count (rows_plan ignored). */
  const int64_t* rows_dev;
  int64_t rows_plan;
} gpn_net_slot_t;
typedef struct gpn_net_rulebook {
  const int32_t* nbr;   /* [K][n_dst]  forward table */
  const int32_t* nbr_t; /* [K][n_src]  table of the transposed map */
  const int32_t* nbr_p;   /* optional tile order of the forward map (gpn_rulebook_tile_order): table ... */
  const int32_t* perm;    /* ... and destination rows; both NULL if absent */
  const int32_t* nbr_t_p; /* the same for the transposed map */
  const int32_t* perm_t;
  const int32_t* pair_src;
  const int32_t* pair_dst;
  const int32_t* tile_off;
  int64_t n_src;
  int64_t n_dst;
  int32_t K;
  int32_t reverse_taps; /* 1: SubM (transposed map = same table with taps reversed) */
} gpn_net_rulebook_t;
typedef struct gpn_net_conv {
  const float* W;
  float* dW;
  int32_t cin;
  int32_t cout;
} gpn_net_conv_t;
typedef struct gpn_net_bn {
  const float* weight;
  const float* bias;
  float* running_mean;
  float* running_var;
  float* save_mean;
  float* save_invstd;
  float* dweight;
  float* dbias;
  float eps;
  float momentum;
  int32_t C;
  int32_t reserved;
} gpn_net_bn_t;
#define GPN_NET_CONV 0
#define GPN_NET_BN 1
#define GPN_NET_CONCAT 2
#define GPN_NET_RELU 1 /* op.flags, BN only */
#define GPN_NET_INFERENCE 2 /* gpn_net_forward(_pair)'s `training`: eval mode without a backward pass to follow (see above) */
typedef struct gpn_net_op {
  int32_t kind;
  int32_t src0;
  int32_t src1; /* BN: residual slot or -1; CONCAT: right-hand input */
  int32_t dst;
  int32_t rulebook; /* CONV */
  int32_t param;    /* CONV: index into convs; BN: index into bns */
  int32_t flags;
  int32_t reserved;
} gpn_net_op_t;
/* training-mode BatchNorm sums (forward: sum x, sum x^2 of a conv's output; backward: sum g, sum g xhat of the gradient a dgrad
 * conv writes) are accumulated by the epilogue of the producing conv launch as order-independent 64-bit fixed-point integers
 * (csrc/bn_stats.h), so that a conv + BatchNorm pair costs 2 launches per direction instead of 3.  on = 0 restores the separate
 * statistics launches (results agree to ~1e-7 relative; both forms are deterministic).  on < 0 queries.  Returns the previous
 * setting. */
int gpn_net_bn_fusion(int on);
/* consecutive weight-gradient contractions of one shape (the convs of a level's residual blocks, the two networks of a paired
 * pass) that share a launch on the executor's second stream: 1 ... 4 layers (default 4, env GPN_WGRAD_GROUP; only layers with
 * fewer than GPN_WGRAD_GROUP_ROWS = 16384 rows are held back for it).  Same arithmetic per layer for every setting (bit-equal
 * gradients).  layers < 1 queries.  Returns the previous setting. */
int gpn_net_wgrad_group(int layers);
size_t gpn_net_ws_bytes(const gpn_net_op_t* ops, int n_ops, const gpn_net_slot_t* slots, int n_slots,
                        const gpn_net_rulebook_t* rulebooks, const gpn_net_conv_t* convs);
int gpn_net_forward(const gpn_net_op_t* ops, int n_ops, gpn_net_slot_t* slots, int n_slots,
                    const gpn_net_rulebook_t* rulebooks, int n_rulebooks, const gpn_net_conv_t* convs, int n_convs,
                    const gpn_net_bn_t* bns, int n_bns, int training, void* ws, size_t ws_bytes, gpn_stream_t stream);
int gpn_net_backward(const gpn_net_op_t* ops, int n_ops, gpn_net_slot_t* slots, int n_slots,
                     const gpn_net_rulebook_t* rulebooks, int n_rulebooks, const gpn_net_conv_t* convs, int n_convs,
                     const gpn_net_bn_t* bns, int n_bns, int training, int need_input_grad, void* ws, size_t ws_bytes,
                     gpn_stream_t stream);
/* Paired passes: TWO networks with the same program over the same rulebooks (the ScoreNet and NPCS-Net U-Nets of
 * network/model.py:116-118 read the same proposal grid), layer i of both computed by ONE launch per kernel (blockIdx.y picks
 * the network).  Each network's values are those of its own single pass.  Tables a / b: the two networks' slots, weights and
 * BatchNorms; ops and rulebooks are shared.  Workspace: 2 x gpn_net_ws_bytes. */
int gpn_net_forward_pair(const gpn_net_op_t* ops, int n_ops, gpn_net_slot_t* slots_a, gpn_net_slot_t* slots_b, int n_slots,
                         const gpn_net_rulebook_t* rulebooks, int n_rulebooks, const gpn_net_conv_t* convs_a,
                         const gpn_net_conv_t* convs_b, int n_convs, const gpn_net_bn_t* bns_a, const gpn_net_bn_t* bns_b,
                         int n_bns, int training, void* ws, size_t ws_bytes, gpn_stream_t stream);
int gpn_net_backward_pair(const gpn_net_op_t* ops, int n_ops, gpn_net_slot_t* slots_a, gpn_net_slot_t* slots_b, int n_slots,
                          const gpn_net_rulebook_t* rulebooks, int n_rulebooks, const gpn_net_conv_t* convs_a,
                          const gpn_net_conv_t* convs_b, int n_convs, const gpn_net_bn_t* bns_a, const gpn_net_bn_t* bns_b,
                          int n_bns, int training, int need_input_grad, void* ws, size_t ws_bytes, gpn_stream_t stream);

/* ================================================================================================
 * B — ball query.  replaces epic_ops.ball_query.ball_query (network/grouping_utils.py:119-128).
 * points [Np,3], query [Q,3], batch_indices [Q] i32, batch_offsets [S+1] i32 (CSR over points),
 * point_labels [Np] / query_labels [Q] i32 (both may be NULL = no label filter).
 * hit: same segment, equal label, d2 < radius^2 (strict; d2 = (dx*dx+dy*dy)+dz*dz, fp32, no fma),
 * first K hits in ascending point index.  indices [Q,K] i32 (-1 padded), count [Q] i32.
 * ================================================================================================ */
int gpn_ball_query(const float* points, const float* query, const int32_t* batch_indices,
                   const int32_t* batch_offsets, const int32_t* point_labels,
                   const int32_t* query_labels, int64_t Np, int64_t Q, int64_t S, float radius, int K,
                   int32_t* indices, int32_t* count, gpn_stream_t stream);
/* same contract and results, O(n k) instead of O(n^2): points binned into a uniform grid (one radix sort), one wave per
 * query over its 27 neighbour cells, hits ranked back into ascending point index; dense neighbourhoods fall back to the
 * index-order scan inside the kernel. */
/* Labels < 0 mark INACTIVE points / queries (grid form only): an inactive point is never a hit, an inactive query gets
 * count 0 - lets a caller run the query over a whole batch with the unwanted points masked instead of compacted away
 * (gpn_proposals_build).  K | GPN_BQ_NO_PAD: rows are not -1 filled beyond count[q] (saves a Q*K*4-byte fill). */
#define GPN_BQ_NO_PAD (1 << 30)
size_t gpn_ball_query_grid_ws_bytes(int64_t Np);
int gpn_ball_query_grid(const float* points, const float* query, const int32_t* batch_indices,
                        const int32_t* batch_offsets, const int32_t* point_labels, const int32_t* query_labels,
                        int64_t Np, int64_t Q, int64_t S, float radius, int K, int32_t* indices, int32_t* count, void* ws,
                        size_t ws_bytes, gpn_stream_t stream);

/* L — connected components.  replaces epic_ops.ccl.connected_components_labeling
 * (network/grouping_utils.py:135-137).  begin_end [2Q] i32 interleaved (begin,end) into edges [E];
 * edges are treated as undirected; labels [Q] i32 = minimum vertex index of the component.
 * compacted != 0 -> labels renumbered 0..n_comp-1 in order of that minimum (needs ws). */
size_t gpn_ccl_ws_bytes(int64_t Q);
int gpn_ccl(const int32_t* begin_end, const int32_t* edges, int64_t Q, int64_t E, int compacted,
            int32_t* labels, void* ws, size_t ws_bytes, gpn_stream_t stream);

/* R — segmented reductions.  replace epic_ops.reduce.segmented_reduce (grouping_utils.py:59-70) and
 * epic_ops.reduce.segmented_maxpool (model.py:360-362).  values [M,C], begin/end [P] i32.
 * mode 0=sum 1=min 2=max; summation in ascending row order. empty segment -> 0.
 * maxpool: ties -> lowest row; argmax [P,C] i32 (-1 for empty). bwd scatters dpooled to drows. */
int gpn_segmented_reduce(const float* values, const int32_t* begin, const int32_t* end, int64_t P,
                         int C, int mode, float* out, gpn_stream_t stream);
int gpn_segmented_maxpool_fwd(const float* values, const int32_t* begin, const int32_t* end, int64_t P,
                              int C, float* pooled, int32_t* argmax, gpn_stream_t stream);
int gpn_segmented_maxpool_bwd(const float* dpooled, const int32_t* argmax, int64_t P, int C, int64_t M,
                              float* dvalues, gpn_stream_t stream);

/* I — instance IoU.  replaces epic_ops.iou.batch_instance_seg_iou (model.py:373-378).
 * proposal_offsets [P+1] i32, instance_labels [M] i32, batch_indices [M] i32,
 * num_points_per_instance [B,I] i32 -> ious [P,I] f32. */
int gpn_instance_iou(const int32_t* proposal_offsets, const int32_t* instance_labels,
                     const int32_t* batch_indices, const int32_t* num_points_per_instance, int64_t P,
                     int64_t B, int I, float* ious, gpn_stream_t stream);

/* N — greedy NMS on a precomputed IoU matrix.  replaces epic_ops.nms.nms (grouping_utils.py:244).
 * ious [P,P] f32, order [P] i32 = proposal ids by descending score (ties -> lower id; the host
 * wrapper sorts).  keep [P] i32 receives kept ids in visiting order, num_keep [1] i32. */
size_t gpn_nms_ws_bytes(int64_t P);
int gpn_nms(const float* ious, const int32_t* order, int64_t P, float threshold, int32_t* keep,
            int32_t* num_keep, void* ws, size_t ws_bytes, gpn_stream_t stream);

/* ================================================================================================
 * F — PointNet++ family.  replace the pybind module pointnet2_cuda
 * (dataset/process_tools/utils/pointnet_lib/src/pointnet2_api.cpp:10-25); argument order and meaning
 * follow the vendored wrappers (ball_query.cpp:14-25, sampling.cpp, interpolate.cpp, group_points.cpp).
 * ================================================================================================ */
/* ball_query_gpu.cu:9-45 — first nsample (d2<r2) in index order; row prefilled with first hit;
 * rows without hits are left untouched. */
int gpn_pn2_ball_query(int b, int n, int m, float radius, int nsample, const float* new_xyz,
                       const float* xyz, int32_t* idx, gpn_stream_t stream);
/* group_points_gpu.cu:47-66 / :8-25 */
int gpn_pn2_group_points(int b, int c, int n, int npoints, int nsample, const float* points,
                         const int32_t* idx, float* out, gpn_stream_t stream);
int gpn_pn2_group_points_grad(int b, int c, int n, int npoints, int nsample, const float* grad_out,
                              const int32_t* idx, float* grad_points, gpn_stream_t stream);
/* sampling_gpu.cu:8-24 / :46-63 */
int gpn_pn2_gather_points(int b, int c, int n, int npoints, const float* points, const int32_t* idx,
                          float* out, gpn_stream_t stream);
int gpn_pn2_gather_points_grad(int b, int c, int n, int npoints, const float* grad_out,
                               const int32_t* idx, float* grad_points, gpn_stream_t stream);
/* sampling_gpu.cu:93-209 — start index 0; temp [b,n] must be pre-filled with 1e10 by the caller
 * (pointnet2_utils.py:27); ties resolved as the reference's block reduction does for its block size
 * opt_n_threads(n) (cuda_utils.h:10-14). */
int gpn_pn2_furthest_point_sampling(int b, int n, int m, const float* dataset, float* temp,
                                    int32_t* idxs, gpn_stream_t stream);
/* same samples; clouds of >= 65536 points are shared by several workgroups (the pre-processing call,
 * dataset/process_tools/convert_rendered_into_input.py:115 of the reference: N ~ 1e5..1e6 -> 20 000). ws may be NULL when
 * gpn_pn2_furthest_point_sampling_ws_bytes returns 0. */
size_t gpn_pn2_furthest_point_sampling_ws_bytes(int b, int n);
int gpn_pn2_furthest_point_sampling_ws(int b, int n, int m, const float* dataset, float* temp, int32_t* idxs,
                                       void* ws, size_t ws_bytes, gpn_stream_t stream);
/* interpolate_gpu.cu:81-124 / :9-57 / :149-169 / :192-214 */
int gpn_pn2_three_nn(int b, int n, int m, const float* unknown, const float* known, float* dist2,
                     int32_t* idx, gpn_stream_t stream);
int gpn_pn2_knn(int b, int n, int m, int k, const float* unknown, const float* known, float* dist2,
                int32_t* idx, gpn_stream_t stream);
int gpn_pn2_three_interpolate(int b, int c, int m, int n, const float* points, const int32_t* idx,
                              const float* weight, float* out, gpn_stream_t stream);
int gpn_pn2_three_interpolate_grad(int b, int c, int n, int m, const float* grad_out,
                                   const int32_t* idx, const float* weight, float* grad_points,
                                   gpn_stream_t stream);

/* ================================================================================================
 * P — per-point losses of the training step in one pass (network/model.py:177-226: loss_sem_seg = focal (gamma 2, rows with
 * label == ignore_index dropped, 0 if none is left) + dice on the 1e-6-smoothed one-hot target; loss_offset = L1 distance
 * and negative cosine on points with label > 0 and instance label >= 0; network/losses.py:35-64, 111-158).
 * logits [M,C] (C <= 32), labels [M] i64, offsets / gt_offsets [M,3], instance_labels [M] i32.
 * fwd -> losses [4] f32 = (focal, dice, offset distance, offset direction) and a 32-byte device block `stats` for bwd;
 * bwd: grad_losses [4] f32 (device) -> d_logits [M,C], d_offsets [M,3].  Fixed-order reductions (deterministic).
 * ================================================================================================ */
size_t gpn_point_losses_ws_bytes(int64_t M);
int gpn_point_losses_fwd(const float* logits, const int64_t* labels, const float* offsets, const float* gt_offsets,
                         const int32_t* instance_labels, int64_t M, int C, int64_t ignore_index, float* losses,
                         void* stats, void* ws, size_t ws_bytes, gpn_stream_t stream);
/* the same with the step's by-products of the logits: preds [M] i64 = the predicted class of every point (first maximum, as
 * torch.argmax), accu [2] f32 = (share of points with preds == labels, the same share among the points with labels > 0; both
 * as fp32 quotients of the exact counts - network/model.py:535-541) */
int gpn_point_losses_fwd_metrics(const float* logits, const int64_t* labels, const float* offsets, const float* gt_offsets,
                                 const int32_t* instance_labels, int64_t M, int C, int64_t ignore_index, float* losses,
                                 int64_t* preds, float* accu, void* stats_out, void* ws, size_t ws_bytes, gpn_stream_t stream);
int gpn_point_losses_bwd(const float* logits, const int64_t* labels, const float* offsets, const float* gt_offsets,
                         const int32_t* instance_labels, int64_t M, int C, int64_t ignore_index, const void* stats,
                         const float* grad_losses, float* d_logits, float* d_offsets, gpn_stream_t stream);

/* Proposal score loss in one launch: class selection (model.py:560-566 of the reference), get_gt_scores
 * (grouping_utils.py:144-156) on the row maxima of ious, binary_cross_entropy_with_logits (mean), plus sigmoid scores and
 * d loss / d logits.  logits [P, C1]; cls_i64 / cls_i32 [M]: class of every proposal point (exactly one non-NULL); offsets
 * [P+1] i32; ious [P, I] (gpn_instance_iou).  Outputs: loss [1], score_preds [P], d_logits [P, C1]. */
int gpn_score_loss(const float* logits, int C1, const int64_t* cls_i64, const int32_t* cls_i32, const int32_t* offsets,
                   const float* ious, int I, int64_t P, float fg_thresh, float bg_thresh, float* loss, float* score_preds,
                   float* d_logits, gpn_stream_t stream);

/* NPCS loss of all proposals in two launches (+ one backward): network/model.py:398-462 with compute_npcs_loss
 * (network/grouping_utils.py:14-43).  logits [M, n_cls3 = 3 (classes - 1)], gt_npcs [M,3], sem_preds [M] i32 (one class per
 * proposal point), sem_labels [M] i64, proposal_offsets [P+1] i32 / proposal_indices [M] i64 (CSR of the proposals),
 * sym_of_class [classes] i64 (device), mats [n_mats,3,3] f32 (device): candidate rotations of every symmetry type back to
 * back; type_first / type_count (<= 24) / type_group (0..2) [n_types <= 8] on the HOST.  loss [1] f32 (device);
 * scratch: (9 P + 4) * 4 bytes kept by the caller between forward and backward.  d_logits [M, n_cls3] fully written. */
int gpn_npcs_loss_fwd(const float* logits, int n_cls3, const float* gt_npcs, const int32_t* sem_preds,
                      const int64_t* sem_labels, const int32_t* proposal_offsets, int64_t P, const int64_t* sym_of_class,
                      const float* mats, const int32_t* type_first, const int32_t* type_count, const int32_t* type_group,
                      int n_types, float* loss, void* scratch, gpn_stream_t stream);
int gpn_npcs_loss_bwd(const float* logits, int n_cls3, const float* gt_npcs, const int32_t* sem_preds,
                      const int64_t* sem_labels, const int64_t* proposal_indices, int64_t M, int64_t P,
                      const int64_t* sym_of_class, const float* mats, const int32_t* type_first, const int32_t* type_count,
                      const int32_t* type_group, int n_types, const void* scratch, const float* grad_loss, float* d_logits,
                      gpn_stream_t stream);

/* ================================================================================================
 * PR — the proposal stage of a step in one call: GAPartNet.proposal_clustering_and_revoxelize (network/model.py:228-346)
 * with cluster_proposals (network/grouping_utils.py:108-140) and the geometry of segmented_voxelize (:47-104).
 * Inputs: points [N, point_stride] f32 (xyz = first three columns), offset_preds [N,3] f32, sem_preds [N] i64 (arg-max of
 * the semantic head), instance_labels [N] i32 or NULL, batch_indices [N] i32 (scene of every point, non-decreasing),
 * jitter [6] f32 on the DEVICE = the two uniform 3-vectors of grouping_utils.py:86-90.
 * A point is valid if sem_preds > 0 (and instance_labels >= 0 when given).  Both cluster sets (xyz, xyz + offset; radius,
 * K1 / K2 neighbours, label-aware) -> proposals of >= min_points points, ordered by their first point, members ascending,
 * set A's proposals before set B's -> per-proposal frame (centre, scale <= max_scale, jittered shift) -> fullscale^3 voxel
 * grid per proposal.  Nothing is read back: every output has its upper-bound capacity and counts [8] i64 (device) holds
 * {Q valid points, M proposal points, P proposals, V voxels, points dropped by the voxeliser, ...}.
 * Outputs (capacity): valid_mask [N] u8, valid_indices [N] i64; per proposal point [2N]: sorted_indices i64 (numbering
 * among the valid points, as the reference), point_indices i64 (row of `points`), proposal_indices i64, batch_indices_p
 * i32, pt_xyz_p [.,3] f32, sem_preds_p i32, instance_labels_p i32, pc_voxel_id i32, point_order i32 (+ voxel_point_start
 * [2N+1] i32: points grouped by voxel); member_slot [2N] i32 (row of the proposal-point list holding point i in set A
 * (entry i) / set B (entry N+i), -1 if none); per proposal [gpn_proposals_max_proposals + 1]: sizes i64,
 * proposal_offsets i32; voxel_coords4 [2N,4] i32 = (proposal, x, y, z), ordered.
 * gpn_proposals_voxel_mean(_bwd): voxel features = ordered mean of feats[point_indices] over each voxel's points, and its
 * transpose (each point sums its at most two memberships in fixed order) - the differentiable half of kernel V.
 * ================================================================================================ */
int64_t gpn_proposals_max_proposals(int64_t N, int min_points);
size_t gpn_proposals_build_ws_bytes(int64_t N, int64_t B, int K1, int K2, int min_points);
int gpn_proposals_build(const float* points, int point_stride, const float* offset_preds, const int64_t* sem_preds,
                        const int32_t* instance_labels, const int32_t* batch_indices, int64_t N, int64_t B, float radius,
                        int K1, int K2, int min_points, float fullscale, float max_scale, const float* jitter,
                        int64_t* counts, uint8_t* valid_mask, int64_t* valid_indices, int64_t* sorted_indices,
                        int64_t* point_indices, int64_t* proposal_indices, int32_t* batch_indices_p, float* pt_xyz_p,
                        int32_t* sem_preds_p, int32_t* instance_labels_p, int64_t* sizes, int32_t* proposal_offsets,
                        int32_t* member_slot, int32_t* voxel_coords4, int32_t* pc_voxel_id, int32_t* point_order,
                        int32_t* voxel_point_start, void* ws, size_t ws_bytes, gpn_stream_t stream);
/* the re-voxelisation step of gpn_proposals_build on its own: the points of the proposals (grouped by proposal, ascending
 * point order inside one; scaled [T2,3] = coordinates in the proposal's grid of (fullscale + 1)^3 unit cells, kept when
 * 0 <= c < fullscale on every axis) -> what gpn_voxelize_ex gives for them with the proposals as segments - unique cells in
 * (proposal, x, y, z) order, voxel of every point (-1 = outside its grid), points grouped by voxel in ascending point order
 * (point_order [T2], voxel_point_start [V + 1]; the dropped points behind all others) - WITHOUT a sort: one workgroup per
 * proposal keeps the grid's bitmap and per-cell counts in LDS.  counts = the stage's count block on the device: [1] = points
 * and [2] = proposals are read, [3] = voxels is written.  fullscale <= 30. */
size_t gpn_proposals_revoxelize_ws_bytes(int64_t P_ub);
int gpn_proposals_revoxelize(const float* scaled, const int32_t* proposal_offsets, int64_t* counts, int64_t T2, int64_t P_ub,
                             float fullscale, int32_t* voxel_coords3, int32_t* voxel_seg, int32_t* pc_voxel_id,
                             int32_t* point_order, int32_t* voxel_point_start, void* ws, size_t ws_bytes, gpn_stream_t stream);
int gpn_proposals_voxel_mean(const float* feats, const int64_t* point_indices, const int32_t* point_order,
                             const int32_t* voxel_point_start, int64_t V, int C, float* out, gpn_stream_t stream);
int gpn_proposals_voxel_mean_bwd(const float* dout, const int32_t* member_slot, const int32_t* pc_voxel_id,
                                 const int32_t* voxel_point_start, int64_t N, int C, float* dfeats, gpn_stream_t stream);

/* ================================================================================================
 * BP - the preparation of one batch for a sparse U-Net in ONE call (round 5): gpn_voxelize_scenes, its one host read, and the
 * rulebook pyramid of the backbone - per level the SubM k = 3 tables (+ tile order from tile_order_min_rows rows), the stride-2
 * map to the next level and its transpose (+ tile order), and, for the levels in the bit mask ident_levels, the k = 1 identity
 * map.  Replaces the loader's per-scene CPU voxelisation (dataset/gapartnet.py:179-205) and spconv's indice-pair construction
 * inside the forward pass (network/backbone.py:19-36, 74-90, 149-152) - the same launches as the separate entry points of
 * sections V / K1 / K2, issued from one native loop into ONE caller-allocated arena.  The call BLOCKS (a blocking-sync event on
 * `stream`) for the voxeliser's sizes; it may be called from any thread (`device` = HIP device ordinal, set for the calling
 * thread) - measured, a worker thread loses to calling it inline (csrc/prepare.hip).
 * desc_host [gpn_backbone_prepare_desc_words(n_levels)] i64, HOST memory; offsets are bytes from the arena base, -1 = absent:
 *   [0] voxels V  [1..3] spatial shape of level 0  [4] dropped points  [5] != 0: fall back to the separate calls (a cell index
 *   beyond the packed keys, an empty batch, an empty level)  [6] n_levels  [7] arena bytes used
 *   [8] voxel_feats [M,C] f32  [9] indices4 [M,4] i32  [10] pc_voxel_id [M] i32  [11] point_order [M] i32  [12] voxel_point_start [M+1]
 *   level l at 16 + 48 l: rows, shape[3], indices offset ([rows,4] i32), 3 spare, then 4 rulebooks x 10 words
 *   (SubM, down = dst coarse, its transpose = dst fine, identity): nbr, pair_src, pair_dst, tile_off, num_pairs, nbr_p, perm
 *   (offsets), n_src, n_dst, K.  Pair lists have the capacities of the separate entry points (27 n, n fine rows, n).
 * pinned_stats_host [8 + n_levels] i64: pinned host memory the statistics are copied to.
 * ================================================================================================ */
int gpn_backbone_prepare_desc_words(int n_levels);
size_t gpn_backbone_prepare_arena_bytes(int64_t M, int C, int64_t S, int n_levels);
int gpn_backbone_prepare(const float* points, const float* feats, const int64_t* seg_offsets, int64_t M, int C, int64_t S,
                         const float* voxel_size_host, int n_levels, uint32_t ident_levels, int64_t tile_order_min_rows,
                         int tile_order_block, int device, void* arena, size_t arena_bytes, int64_t* desc_host,
                         int64_t* pinned_stats_host, gpn_stream_t stream);

/* ================================================================================================
 * SP - the per-scene preparation of RAW scenes for a whole batch (round 5): what the reference's loader workers do per scene in
 * numpy (dataset/gapartnet.py:55-82) - compact_instance_labels (:134-142), apply_augmentations (:85-120; the random draws stay
 * on the host, 9 + C doubles per scene come in), generate_inst_info (:145-176) - in three launches.
 * Inputs: points [N, 3 + C] f32 (scenes back to back, seg_offsets [B + 1] i64 on the device), sem_labels [N] (sem_bytes = 2 / 4 / 8:
 * int16 / int32 / int64), instance_labels [N] i32 (negative = no instance), mats [B, 3, 3] f64 or NULL (xyz @ M in float64, rounded
 * once), shifts [B, C] f64 or NULL (colour + shift in float64, rounded once).
 * Outputs: points_out [N, 3 + C]; batch_indices [N] i32; instance_out [N] i32 (the non-negative ids of every scene renumbered
 * 0..K-1 ascending; negatives kept); regions [N, 9] f32 = mean | min | max xyz of the point's instance (zeros off-instance; the mean
 * from an order-independent fixed-point sum, within an ulp of numpy's); num_points_per_instance / instance_sem_labels
 * [B, gpn_scene_prepare_max_instances()] i32 (count, semantic label of the instance's first point; 0 / -1 past a scene's K);
 * num_instances_host [B] i64 = K per scene, written by the device into PINNED host memory (or NULL), -1 for a scene with more
 * distinct ids than the tables hold - *overflow (device) is then 1 and the outputs are not to be used. */
int gpn_scene_prepare_max_instances(void);
size_t gpn_scene_prepare_ws_bytes(int B);
int gpn_scene_prepare(const float* points, const void* sem_labels, int sem_bytes, const int32_t* instance_labels,
                      const int64_t* seg_offsets, int64_t N, int C, int B, const double* mats, const double* shifts,
                      float* points_out, int32_t* batch_indices, int32_t* instance_out, float* regions,
                      int32_t* num_points_per_instance, int32_t* instance_sem_labels, int64_t* num_instances_host,
                      int32_t* overflow, void* ws, size_t ws_bytes, gpn_stream_t stream);

/* ================================================================================================
 * PP - post-processing of a validation / test step's proposals in one call (round 5): filter_invalid_proposals +
 * apply_nms of the reference (network/grouping_utils.py:159-298, called from network/model.py:667-692, 807-857).
 * Inputs as gpn_proposals_build left them: score_preds [P] f32 (the sigmoid scores), sizes [P] i64, proposal_offsets [P+1] i32,
 * point_indices / proposal_indices [M] i64, member_slot [2N] i32 (N = points of the batch).  P may be a bound with the live count
 * in *p_dev (p_plan sizes grids only).  A proposal survives if score > score_threshold and size > min_points (both strict) and
 * greedy NMS by descending score (ties: lower id first) on the point-set IoU inter / ((|a| + |b|) - inter + 1e-8) (fp32) does
 * not suppress it (IoU > iou_threshold with a kept proposal of higher rank).
 * Outputs: kept_ids [P] i32 = ids of the kept proposals, ascending; new_offsets [P+1] i32 = their CSR offsets after compaction;
 * src_row [M] i64 = for every point of a kept proposal its row in the inputs; counts [3] i64 (device) = {kept proposals, their
 * points, != 0: a proposal shared points with more proposals than the kernel tables hold (results incomplete)}.  The caller
 * re-indexes whatever per-proposal / per-point fields it needs with kept_ids / src_row. */
size_t gpn_proposals_postprocess_ws_bytes(int64_t P);
int gpn_proposals_postprocess(const float* score_preds, const int64_t* sizes, const int32_t* proposal_offsets,
                              const int64_t* point_indices, const int64_t* proposal_indices, const int32_t* member_slot, int64_t N,
                              int64_t P, const int64_t* p_dev, int64_t p_plan, float score_threshold, int64_t min_points,
                              float iou_threshold, int32_t* kept_ids, int32_t* new_offsets, int64_t* src_row, int64_t* counts,
                              void* ws, size_t ws_bytes, gpn_stream_t stream);
/* The NMS kernel keeps one status byte per LIVE proposal in LDS up to n proposals (default and maximum 131072) and in the
 * workspace beyond - any bound P is accepted (round 5 rejected bounds above 131072: 32 scenes of 20k points).  n < 0 queries;
 * returns the previous value.  Tests lower it to run the workspace form on small inputs. */
int64_t gpn_proposals_postprocess_lds_proposals(int64_t n);

/* ================================================================================================
 * O — the optimizer step.  GAPartNet.configure_optimizers (network/model.py:1051-1055): torch.optim.Adam(lr) over every
 * parameter.  One launch for the whole model: table [n_tensors] on the DEVICE describes the fp32 tensors (built once;
 * pointers are stable from step to step), block_first [n_tensors] i32 = first workgroup of each tensor with
 * gpn_adam_blocks(numel) workgroups per tensor.  Update = torch's single-tensor Adam in fp32 (no amsgrad, no weight decay),
 * `step` = the 1-based step count shared by the tensors of the call.
 * ================================================================================================ */
typedef struct {
  void* param;
  const void* grad;
  void* exp_avg;
  void* exp_avg_sq;
  int64_t numel;
} gpn_adam_tensor_t;
/* n_segs fp32 segments dst[0..numel) = src[0..numel) (device memory) in one launch per 96 segments; segs_host is read during
 * the call.  (Gradients that autograd allocates afresh every step -> the persistent buffers an Adam table points at.) */
typedef struct {
  const void* src;
  void* dst;
  int64_t numel;
} gpn_copy_seg_t;
int gpn_copy_many(const gpn_copy_seg_t* segs_host, int n_segs, gpn_stream_t stream);
int gpn_adam_blocks(int64_t numel);
int gpn_adam_step(const gpn_adam_tensor_t* table_dev, const int32_t* block_first_dev, int n_tensors, int n_blocks, double lr,
                  double beta1, double beta2, double eps, int64_t step, gpn_stream_t stream);
/* the same step for tensors whose gradients exist only if a DEVICE counter is non-zero (round 5): the ScoreNet / NPCS-Net
 * parameters of a training step whose proposal count was never read on the host (section DEV).  The reference does not run
 * those networks for a batch without proposals (network/model.py:573-574: the gradients stay None and torch.optim.Adam skips
 * the tensors - value, moments and step count unchanged); the device-counted step runs them over zero rows and gets zero
 * gradients, which an ungated Adam step would still turn into a parameter update (momentum) and a moment decay.
 * *gate_dev == 0: nothing is touched and *skipped_dev += 1.  Otherwise the step number is step - *skipped_dev (>= 1).
 * gate_dev == skipped_dev == NULL: gpn_adam_step. */
int gpn_adam_step_gated(const gpn_adam_tensor_t* table_dev, const int32_t* block_first_dev, int n_tensors, int n_blocks,
                        double lr, double beta1, double beta2, double eps, int64_t step, const int64_t* gate_dev,
                        int64_t* skipped_dev, gpn_stream_t stream);

/* ================================================================================================
 * PF - pose fitting.  replaces the per-proposal numpy loop of gapartnet/misc/pose_fitting.py:4-147 (estimate_pose_from_npcs:
 * 5-point RANSAC over Umeyama similarity fits, Umeyama on the inliers, NPCS-aligned box; callers network/model.py:966-980,
 * structure/utils.py:172-188) for ALL proposals of a batch: two launches, float64 like the reference.
 * xyz / npcs [M,3] f64 (proposal p = rows offsets[p]:offsets[p+1], non-empty), picks [P,H,5] i64 = the reference's
 * np.random.randint(n, size=5) draws per iteration (drawn by the caller: gapartnet_amd.misc.pose_fitting_batched.draw_picks).
 * outputs: valid [P] u8, scale [P], rotation [P,3,3], translation [P,3], transform [P,4,4], bbox [P,8,3] (NaN where not
 * valid), inlier_mask [M] u8, best_iteration [P] i64, residual [P,H] (of every hypothesis, as the reference computes it).
 * ================================================================================================ */
size_t gpn_pose_fit_ws_bytes(int64_t P, int H);
int gpn_pose_fit(const double* xyz, const double* npcs, const int64_t* offsets, const int64_t* picks, int64_t P, int64_t M,
                 int H, double stop_thrsh, uint8_t* valid, double* scale, double* rotation, double* translation,
                 double* transform, double* bbox, uint8_t* inlier_mask, int64_t* best_iteration, double* residual, void* ws,
                 size_t ws_bytes, gpn_stream_t stream);

/* ================================================================================================
 * DEV - entry points whose extents are DEVICE COUNTERS (round 4: the proposal stage of a training step without a host read;
 * reference glue: network/model.py:228-346, 348-462 - there every stage boundary is a device->host read of a size).
 * Convention of every *_dev entry point: an extent argument (N, P, M, V ...) is the BOUND its buffers are allocated for;
 * `<x>_dev` points at an int64 on the device holding the live value (written by an earlier launch on the stream, e.g.
 * gpn_proposals_build's counts) and `<x>_plan` is the host's estimate of it (<= 0: none) that only sizes the grid.  Kernels read
 * the counter and walk their work with a grid stride, so results never depend on the estimate; tables are laid out for the LIVE
 * count (see gpn_net_slot_t).  Each computes exactly what its exactly-sized twin computes on the first *<x>_dev rows.
 * ================================================================================================ */
int gpn_rulebook_subm3_dev(const int32_t* indices, int64_t N, const int64_t* n_dev, int64_t n_plan,
                           const int32_t* spatial_shape_host, int32_t* nbr, int32_t* pair_src, int32_t* pair_dst,
                           int32_t* tile_off, int64_t* num_pairs, void* ws, size_t ws_bytes, gpn_stream_t stream);
size_t gpn_rulebook_down_dev_ws_bytes(int64_t N, int64_t batch_size, const int32_t* spatial_shape_host);
int gpn_rulebook_down_dev(const int32_t* indices, int64_t N, const int64_t* n_dev, int64_t n_plan, int64_t batch_size,
                          const int64_t* batch_dev, int64_t batch_plan, const int32_t* spatial_shape_host, int32_t* out_indices,
                          int32_t* fine_to_coarse, int32_t* tap, int64_t* num_out, void* ws, size_t ws_bytes, gpn_stream_t stream);
int gpn_rulebook_down_lists_dev(const int32_t* fine_to_coarse, const int32_t* tap, int64_t N, const int64_t* n_dev, int64_t n_plan,
                                int64_t n_out, const int64_t* n_out_dev, int64_t n_out_plan, int32_t* fwd_nbr, int32_t* fwd_src,
                                int32_t* fwd_dst, int32_t* fwd_tile_off, int32_t* bwd_nbr, int32_t* bwd_src, int32_t* bwd_dst,
                                int32_t* bwd_tile_off, int64_t* num_pairs, void* ws, size_t ws_bytes, gpn_stream_t stream);
int gpn_rulebook_identity_dev(int64_t n, const int64_t* n_dev, int64_t n_plan, int32_t* rows, int32_t* tile_off, int32_t* nbr,
                              int64_t* num_pairs, gpn_stream_t stream);
int gpn_gather_rows_dev(const float* table, const int32_t* idx, int64_t n, const int64_t* n_dev, int64_t n_plan, int C, float* out,
                        gpn_stream_t stream);
int gpn_scatter_rows_csr_dev(const float* dout, const int32_t* order, const int32_t* starts, int64_t n_rows, const int64_t* n_dev,
                             int64_t n_plan, int C, float* dtable, gpn_stream_t stream);
int gpn_proposals_voxel_mean_dev(const float* feats, const int64_t* point_indices, const int32_t* point_order,
                                 const int32_t* voxel_point_start, int64_t V, const int64_t* v_dev, int64_t v_plan, int C, float* out,
                                 gpn_stream_t stream);
/* sem_labels [N] i64 / gt_npcs [N,3] f32 (either may be NULL) at the proposal points: the reference's sem_labels[rows],
 * gt_npcs[rows] (model.py:556-571) for the first *m_dev of M rows */
int gpn_proposals_targets_dev(const int64_t* sem_labels, const float* gt_npcs, const int64_t* point_indices, int64_t M,
                              const int64_t* m_dev, int64_t m_plan, int64_t* sem_out, float* npcs_out, gpn_stream_t stream);
int gpn_linear_fwd_dev(const float* x, const float* W, const float* b, int64_t N, const int64_t* n_dev, int64_t n_plan, int cin,
                       int cout, float* y, gpn_stream_t stream);
int gpn_linear_bwd_dev(const float* x, const float* W, const float* dy, int64_t N, const int64_t* n_dev, int64_t n_plan, int cin,
                       int cout, float* dx, float* dW, float* db, void* ws, size_t ws_bytes, gpn_stream_t stream);
int gpn_segmented_maxpool_fwd_dev(const float* values, const int32_t* begin, const int32_t* end, int64_t P, const int64_t* p_dev,
                                  int64_t p_plan, int C, float* pooled, int32_t* argmax, gpn_stream_t stream);
int gpn_segmented_maxpool_bwd_dev(const float* dpooled, const int32_t* argmax, int64_t P, const int64_t* p_dev, int64_t p_plan, int C,
                                  int64_t M, const int64_t* m_dev, int64_t m_plan, float* dvalues, gpn_stream_t stream);
int gpn_instance_iou_dev(const int32_t* proposal_offsets, const int32_t* instance_labels, const int32_t* batch_indices,
                         const int32_t* num_points_per_instance, int64_t P, const int64_t* p_dev, int64_t p_plan, int64_t B, int I,
                         float* ious, gpn_stream_t stream);
int gpn_score_loss_dev(const float* logits, int C1, const int64_t* cls_i64, const int32_t* cls_i32, const int32_t* offsets,
                       const float* ious, int I, int64_t P, const int64_t* p_dev, float fg_thresh, float bg_thresh, float* loss,
                       float* score_preds, float* d_logits, gpn_stream_t stream);
int gpn_npcs_loss_fwd_dev(const float* logits, int n_cls3, const float* gt_npcs, const int32_t* sem_preds, const int64_t* sem_labels,
                          const int32_t* proposal_offsets, int64_t P, const int64_t* p_dev, int64_t p_plan,
                          const int64_t* sym_of_class, const float* mats, const int32_t* type_first, const int32_t* type_count,
                          const int32_t* type_group, int n_types, float* loss, void* scratch, gpn_stream_t stream);
int gpn_npcs_loss_bwd_dev(const float* logits, int n_cls3, const float* gt_npcs, const int32_t* sem_preds, const int64_t* sem_labels,
                          const int64_t* proposal_indices, int64_t M, const int64_t* m_dev, int64_t m_plan, int64_t P,
                          const int64_t* sym_of_class, const float* mats, const int32_t* type_first, const int32_t* type_count,
                          const int32_t* type_group, int n_types, const void* scratch, const float* grad_loss, float* d_logits,
                          gpn_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* GPN_H */
