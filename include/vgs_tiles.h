/* include/vgs_tiles.h -- C-ABI of the native tiled driver (libvgs_tiles.so): one process per GPU, spatial tiles, one
 * exchange of boundary records (BASELINE.json configs[4]).
 *
 * The reference is single-process (segmentationVGS, `test`:9-86); this is what a caller with a scene larger than one GPU
 * puts around the same stages.  The data path of include/vgs.h stays on each rank's GPU; the only data-path exchange is ONE
 * all-gather of boundary-voxel records per run (plus two small ones for the shared grid).  The collectives run over RCCL:
 * the caller hands in its ncclComm_t (as void*, so that this header needs no rccl.h).  The Python twin of this driver
 * (vgs-svgs-segmentation_amd/dist.py, torch.distributed) is kept as the test harness; both give the same labels.
 *
 * Protocol per run (csrc/tiles.cpp): shared octree grid (all-gather of the tiles' bounding boxes, the growth replayed on the
 * host, a GPU scan + broadcast only where a box leaves the step open) -> the four stages on tile + halo -> unique boundary
 * voxels (code, local root, owned voxels of that root) and the number of purely local segments leave the GPU -> ncclAllGather
 * -> the same union-find over (rank, root) on every rank, size filter on global sizes -> labels applied on the GPU.
 */
#ifndef VGS_TILES_H_
#define VGS_TILES_H_

#include "vgs.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct vgs_tiles vgs_tiles;

/* Communicator.  kind VGS_TILES_COMM_RCCL: `handle` is an ncclComm_t over `world` ranks (RCCL over xGMI inside a node), this
 * process is rank `rank` and uses HIP device p->device.  kind VGS_TILES_COMM_LOCAL: `handle` comes from
 * vgs_tiles_local_group_create -- `world` driver THREADS of one process meet in shared memory (what the tests use to run
 * several ranks on a single GPU, where RCCL refuses two ranks on one device). */
enum { VGS_TILES_COMM_RCCL = 0, VGS_TILES_COMM_LOCAL = 1, VGS_TILES_COMM_CALLBACKS = 2 };

/* kind VGS_TILES_COMM_CALLBACKS: `handle` points to this struct (copied by vgs_tiles_create) -- the caller's own transport over HOST
 * buffers (MPI, gloo, a test harness that runs two processes on one GPU).  Both return 0 on success.  all_gather: every rank sends
 * `bytes` bytes, recv holds world * bytes in rank order.  bcast: `bytes` bytes from rank `root` to everyone, in place. */
typedef struct vgs_tiles_callbacks {
  void* user;
  int (*all_gather)(void* user, const void* send, void* recv, uint64_t bytes);
  int (*bcast)(void* user, void* buf, uint64_t bytes, int root);
} vgs_tiles_callbacks;

/* Failures.  A rank that fails locally (a stage error, out of memory, an injected fault) still takes part in the next collective and
 * sends its status as a word of that payload, so that no peer is left waiting inside a collective: the failing rank returns its own
 * status and message, every other rank returns VGS_E_PEER naming it.  After either, the process should exit non-zero and let the
 * launcher end the job; a vgs_tiles handle is not usable after a failed run.  An error of the collective itself (RCCL, a callback)
 * is returned as VGS_E_HIP. */

vgs_status vgs_tiles_local_group_create(int world, void** group);
void vgs_tiles_local_group_destroy(void* group);
/* a rank thread that fails calls this so that the others leave their collectives with an error instead of waiting for it */
void vgs_tiles_local_group_abort(void* group);

/* layout: tiles_x x tiles_y tiles of side `pitch` centred on (center_x, center_y); rank k owns tile (k % tiles_x, k / tiles_x),
 * the outer tiles are open ended.  pitch <= 0: the largest x-extent over the ranks' clouds (agreed with one all-gather at the
 * first vgs_tiles_set_points). */
vgs_status vgs_tiles_create(const vgs_params* p, int comm_kind, void* comm_handle, int rank, int world, int tiles_x, int tiles_y,
                            double pitch, double center_x, double center_y, vgs_tiles** out);
void vgs_tiles_destroy(vgs_tiles* t);
const char* vgs_tiles_last_error_string(const vgs_tiles* t);

/* options.  VGS_TILES_OPT_STRICT_REGION (0 / 1, default 0): points a rank holds outside its own region may come back unlabelled
 * (their voxels are owned and cut by another rank).  0: vgs_tiles_set_points warns once on stderr; 1: it fails with VGS_E_ARG on that
 * rank and VGS_E_PEER on the others.  vgs_tiles_get_info reports the count either way. */
enum { VGS_TILES_OPT_STRICT_REGION = 1 };
vgs_status vgs_tiles_set_option(vgs_tiles* t, int32_t option, int64_t value);

/* this rank's points (host memory, stride_bytes 12 or 16).  The ranks exchange their border strips (2 * graph_size + voxel_size
 * wide, one all-gather: data loading, not part of a run) and every rank uploads its tile + halo. */
vgs_status vgs_tiles_set_points(vgs_tiles* t, const float* xyz_host, int64_t n, int32_t stride_bytes);
/* shared grid, the four stages, the boundary exchange, global labels */
vgs_status vgs_tiles_run(vgs_tiles* t);
/* host wall time of the last run's phases on this rank, milliseconds: shared grid (collectives included), the four stages,
 * boundary records off the GPU, the exchange (ONE all-gather), the boundary union-find, labels applied on the GPU, total */
enum { VGS_TILES_T_GRID = 0, VGS_TILES_T_STAGES = 1, VGS_TILES_T_RECORDS = 2, VGS_TILES_T_EXCHANGE = 3, VGS_TILES_T_MERGE = 4,
       VGS_TILES_T_LABELS = 5, VGS_TILES_T_TOTAL = 6, VGS_TILES_T_COUNT = 7 };
vgs_status vgs_tiles_get_times(vgs_tiles* t, double* ms, int32_t n /* <= VGS_TILES_T_COUNT */);
/* labels of this rank's own points (global segment ids, -1 = dropped), the number of segments kept over all ranks */
vgs_status vgs_tiles_get_point_labels(vgs_tiles* t, int32_t* labels /* n */, int64_t* kept_global);
/* points this rank holds outside its own region (they may come back unlabelled: load by region) */
vgs_status vgs_tiles_get_info(vgs_tiles* t, int64_t* n_outside, int64_t* n_local /* tile + halo */, int64_t* n_boundary_records);
/* the last run's boundary exchange as this rank saw it: payload bytes sent and received, and how many collectives carried them (1: every
 * rank's records fit the fixed-size all-gather of 8192 records; 3: a size word and one padded all-gather behind it) */
vgs_status vgs_tiles_get_exchange(vgs_tiles* t, int64_t* bytes_sent, int64_t* bytes_received, int32_t* collectives);
/* The boundary merge on its own (host arithmetic, no context, no GPU; for tests): rank r's records are entries rec_off[r] ..
 * rec_off[r+1] of code / root / cnt (vgs_get_boundary_roots), kept_local[r] its purely local segments.  Outputs: base[r] (labels of
 * rank r's local segments start there), per rank the unique local roots named by records (uroot, entries uoff[r] .. uoff[r+1]) and
 * their global labels (-1 = dropped by the `> voxels_min` filter on the GLOBAL size), the number of segments kept over all ranks.
 * uroot / ulabel hold at most rec_off[world] entries. */
vgs_status vgs_tiles_merge_boundary(int world, const int64_t* rec_off, const uint64_t* code, const int32_t* root, const int32_t* cnt,
                                    const int64_t* kept_local, int voxels_min, int64_t* base, int64_t* uoff, int32_t* uroot, int32_t* ulabel,
                                    int64_t* kept_total);
/* the rank's engine context (read-only use: counts, stage times) */
vgs_ctx* vgs_tiles_context(vgs_tiles* t);

#ifdef __cplusplus
}
#endif
#endif /* VGS_TILES_H_ */
