/* include/vgs.h -- C-ABI of the MI355X-native VGS / SVGS segmentation engine (libvgs_hip.so).
 *
 * Drop-in boundary for the hot path of Yusheng-Xu/VGS-SVGS-Segmentation.  The reference has no
 * FFI layer: its boundary is the public surface of two header-only class templates plus the
 * Task_File line indices.  Each entry point below names the reference member
 * function(s) it replaces (paths under the reference repo):
 *   VS: = voxel_segmentation.h   SS: = supervoxel_segmentation.h   T: = test   IOC: = point_clouds_IO.cpp
 * include/vgs_segmentation.hpp re-creates the two classes (same method names and call order) on top
 * of these functions; INTEGRATION.md shows the binding a maintainer of the reference would add.
 *
 * Conventions: plain pointers and sizes only; all outputs are caller-allocated (size query first);
 * every function returns a vgs_status; a context owns one HIP stream and is not re-entrant.
 * There is no CPU fallback: without a HIP device vgs_create fails with VGS_E_HIP.
 */
#ifndef VGS_H_
#define VGS_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
  VGS_OK = 0,
  VGS_E_ARG = 1,          /* bad argument */
  VGS_E_STATE = 2,        /* call-order contract violated (getVoxelNum before attributes ...) */
  VGS_E_HIP = 3,          /* HIP runtime error / no device */
  VGS_E_NOMEM = 4,
  VGS_E_UNSUPPORTED = 5,  /* configuration outside the built kernels' limits (message says which) */
  VGS_E_IO = 6,
  VGS_E_PEER = 7          /* tiled driver (vgs_tiles.h): another rank reported a failure; this rank stopped with it */
} vgs_status;

/* Parameter surface = Task_File_VGS.txt / Task_File_SVGS.txt (T:25-37, T:108-125). */
typedef struct {
  int32_t method;        /* 2 = VGS, 3 = SVGS (task line 24) */
  float voxel_size;      /* line 28 */
  float graph_size;      /* VGS line 30, SVGS line 32 */
  float sig_p, sig_n, sig_o, sig_e, sig_c, sig_w;
  float cut_thred;
  int32_t points_min, adjacency_min, voxels_min;
  float seed_size;       /* SVGS line 30 */
  float color_impt, spatial_impt, normal_impt; /* SVGS sig_a, sig_b, sig_c(2nd) lines 46,48,50 */
  int32_t q7_count_as_index; /* 1 = reproduce closestCheck reading the neighbour count as a voxel id (VS:2243) */
  int32_t device;        /* HIP device ordinal */
  int32_t vccs_mode;     /* svgs_supervoxels: 1 (the default of vgs_params_default_svgs since round 6) = pcl::SupervoxelClustering's own
                          * order: supervoxels take their turns one after the other in label order, 2-ring normals, seed rejection, the
                          * adjacency octree's own lattice, refineNormals, re-seeding by the nearest of all voxels;
                          * 0 = a faster synchronous variant (every voxel decides from the state at the start of a round): a DIFFERENT
                          * algorithm whose final segments are not within P2 of mode 1's (csrc/vccs.hip; unpinned against PCL either way) */
} vgs_params;

typedef struct vgs_ctx vgs_ctx;

/* counts returned by vgs_get_counts */
enum {
  VGS_N_POINTS = 0,      /* getCloudPointNum()  VS:94 */
  VGS_N_FINITE = 1,      /* points that entered the octree */
  VGS_N_VOXELS = 2,      /* getVoxelNum()       VS:104 */
  VGS_N_USED = 3,        /* voxels with > points_min points (VS:322) */
  VGS_N_ADJ = 4,         /* sum of adjacency list lengths over used voxels */
  VGS_N_CLUSTERS = 5,    /* getClusterNum()     VS:111 (all clusters, singletons included) */
  VGS_N_KEPT = 6,        /* clusters with > voxels_min voxels = getClusterIdx().size()  VS:969 */
  VGS_N_PAIRS = 7,       /* pair affinities evaluated by the local-graph kernel */
  VGS_N_DEPTH = 8,       /* octree depth */
  VGS_N_ISOLATED = 9,    /* closestCheck candidates */
  VGS_N_REATTACHED = 10, /* closestCheck successes */
  VGS_N_SUPERVOXELS = 11,
  VGS_N_COUNTS = 16
};

/* stage timers (milliseconds, HIP events on the context's stream) */
enum {
  VGS_T_VOXELIZE = 0, VGS_T_FEATURES = 1, VGS_T_ADJACENCY = 2, VGS_T_LOCALCUT = 3 /* incl. its hand-over kernels that end inside the merge stage */, VGS_T_MERGE = 4,
  VGS_T_LABELS = 5 /* cluster filter + voxel and point labels: the last part of VGS_T_MERGE, not added to the total again */, VGS_T_TOTAL = 6, VGS_T_LOCALCUT_KERNEL = 7, VGS_T_SUPERVOXEL = 8,
  VGS_T_LOCALCUT_BULK = 9, /* the one launch of the bulk-class local-cut kernel (k_localcut_wave<96,448,1>), HIP events on its stream */
  VGS_T_COUNT = 12
};

/* ---- parameters ------------------------------------------------------------------------ */
vgs_status vgs_params_default_vgs(vgs_params* p);   /* Task_File_VGS.txt values  */
vgs_status vgs_params_default_svgs(vgs_params* p);  /* Task_File_SVGS.txt values */
/* inputTaskTxtFile + the fixed line indices of segmentationVGS/SVGS (IOC:148-169, T:25-37, T:108-125);
 * strips CR; in_name/out_name (may be NULL) receive lines 15 and 21. */
vgs_status vgs_parse_task_file(const char* path, vgs_params* p, char* in_name, char* out_name, int name_cap);

/* ---- lifetime -------------------------------------------------------------------------- */
/* VoxelBasedSegmentation(res) / SuperVoxelBasedSegmentation(res) ctor + setVoxelSize (+ setSupervoxelSize,
 * setGraphSize) (VS:84,124  SS:85,143-164) */
vgs_status vgs_create(const vgs_params* p, vgs_ctx** out);
void vgs_destroy(vgs_ctx* ctx);
/* The reference passes parameters when a stage is called (setVoxelSize VS:124, findAllVoxelAdjacency(graph_size)
 * VS:223, segmentVoxelCloudWithGraphModel(cut, sigmas) VS:372).  Replaces the context's parameters; stages whose
 * inputs changed must be re-run (the context's stage state is rolled back accordingly). method/device are fixed. */
vgs_status vgs_set_params(vgs_ctx* ctx, const vgs_params* p);
const char* vgs_last_error_string(const vgs_ctx* ctx); /* ctx may be NULL: last create error */

/* ---- input: setInputCloud + getCloudPointNum (T:52-53, VS:94) ---------------------------- */
/* stride_bytes 12 (packed xyz) or 16 (pcl::PointXYZ).  Host variant copies once to HBM. */
vgs_status vgs_set_points(vgs_ctx* ctx, const float* xyz_host, int64_t n, int32_t stride_bytes);
/* device variant: no copy, the caller keeps the buffer alive and unchanged until results are read */
vgs_status vgs_set_points_device(vgs_ctx* ctx, const float* xyz_dev, int64_t n, int32_t stride_bytes);

/* A sequence of clouds (no reference counterpart: the reference loads one PCD, T:41-49).  vgs_stage_points starts the copy
 * of the NEXT cloud into the context's second input buffer on a copy stream and returns at once; the current cloud, its
 * stages and its results stay untouched.  vgs_commit_points makes the staged cloud the current one, like vgs_set_points
 * without the host-side wait (the stages' stream waits for the copy on the device).  The host buffer must stay unchanged
 * until the commit's first stage has run; pinned memory (vgs_host_alloc / vgs_host_register) makes the copy asynchronous
 * and about twice as fast, pageable memory works. */
vgs_status vgs_stage_points(vgs_ctx* ctx, const float* xyz_host, int64_t n, int32_t stride_bytes);
vgs_status vgs_commit_points(vgs_ctx* ctx);
/* pinned host memory for the two calls above and for vgs_get_point_labels_async (hipHostMalloc / hipHostRegister) */
vgs_status vgs_host_alloc(void** p, uint64_t bytes);
vgs_status vgs_host_free(void* p);
vgs_status vgs_host_register(void* p, uint64_t bytes);
vgs_status vgs_host_unregister(void* p);

/* ---- VGS stages, in the reference's call order (T:54-74) --------------------------------- */
vgs_status vgs_voxelize(vgs_ctx* ctx);   /* addPointsFromInputCloud + getBoundingBox/setBoundingBox + setVoxelCenters + getVoxelNum (T:54-62, VS:146-189) */
vgs_status vgs_features(vgs_ctx* ctx);   /* calcualteVoxelCloudAttributes (VS:290-369) */
vgs_status vgs_adjacency(vgs_ctx* ctx);  /* findAllVoxelAdjacency(graph_size) (VS:223-265) */
vgs_status vgs_segment(vgs_ctx* ctx);    /* segmentVoxelCloudWithGraphModel (VS:372-421) + drawColorMapofPointsinClusters' cluster filter (VS:963-1009) */
vgs_status vgs_run(vgs_ctx* ctx);        /* all stages for ctx's method (segmentationVGS T:51-76 / segmentationSVGS T:138-160) */

/* ---- SVGS ------------------------------------------------------------------------------- */
/* segmentSupervoxelCloudWithGraphModel from a caller-supplied supervoxel labelling (what
 * pcl::SupervoxelClustering::getLabeledCloud returns, SS:283): labels_dev/labels_host hold one int32 per
 * point, 0 = unassigned; max_label = getMaxLabel().  (SS:279-421) */
vgs_status svgs_set_supervoxel_labels(vgs_ctx* ctx, const int32_t* labels_host, int32_t max_label);
vgs_status svgs_supervoxels(vgs_ctx* ctx);  /* createSupervoxels: VCCS-style clustering on the GPU (SS:245-331) */
vgs_status svgs_segment(vgs_ctx* ctx);      /* attributes + neighbours + local cuts + merge (SS:362-421) */
/* getLabeledCloud / getMaxLabel (SS:283-284): one label per point (0 = unassigned) */
vgs_status svgs_get_supervoxel_labels(vgs_ctx* ctx, int32_t* labels, int32_t* max_label);

/* ---- results ---------------------------------------------------------------------------- */
vgs_status vgs_get_counts(vgs_ctx* ctx, int64_t* counts /* VGS_N_COUNTS */);
vgs_status vgs_get_stage_times(vgs_ctx* ctx, double* ms /* VGS_T_COUNT */);
/* Schedule diagnostics of the last local cut (no reference counterpart; tests use them to see that an input reached the
 * path it was built for).  out[0..7]: 0 rounds the lazy schedule gave up in, 1 voxels it handed over because a shell or
 * phase B overflowed its list, 2 handed over to the dense kernel (all causes), 3 sent on by the dense kernel to the
 * general kernel (a list of 2048 edges overflowed), 4 handed over by the classes above 128 neighbours, 5 voxels
 * outside every kernel's limits (result incomplete: vgs_segment reports it), 6 rows crossValidation put off, 7 voxels for which a dense
 * kernel took a phase in bands of descending weight (more edges than its list holds) */
vgs_status vgs_get_schedule_counters(vgs_ctx* ctx, int64_t* out /* 8 */);
/* The same with room to grow (n <= 16): 8 neighbourhoods above 2048 used voxels, cut by the extra-large instantiation of the general
 * kernel (the reference sizes its matrix to any n, VS:1815-1818; round 4 -- such a voxel used to end the run with VGS_E_UNSUPPORTED) */
/* round 5: 9 voxels cut by the pair-list kernel (csrc/localcut_pg.hpp), 10 entries of the pair lists built for them (csrc/pairlist.hpp),
 * 11 one-wavefront voxels handed over without a try on the strength of the scene's samples (LwParams::vote), 12 rows that found the
 * pair lists' pool exhausted */
vgs_status vgs_get_schedule_counters_ex(vgs_ctx* ctx, int64_t* out, int32_t n);
/* Screening table of the dense hand-over kernels for a parameter set (host arithmetic, no context, no GPU; for tests): a
 * pair of valid positions and normals whose squared centroid distance d2 is >= *d2_stop, or whose dot(n1, n2) lies in
 * [-1, ctab[min(63, int(d2 * *ctab_scale))]], weighs at most 1 - cut_thred and is not evaluated (csrc/localcut.hip). */
vgs_status vgs_screen_table(const vgs_params* params, float* d2_stop, float* ctab_scale, float* ctab /* 64 */);
vgs_status vgs_get_bbox(vgs_ctx* ctx, double* min3_max3);                 /* getBoundingBox (T:56) */
/* voxel table in leaf order: key 3*V, start V+1 (offsets into point_idx), point_idx N' ; any may be NULL */
vgs_status vgs_get_voxel_table(vgs_ctx* ctx, uint32_t* key, int32_t* start, int32_t* point_idx);
vgs_status vgs_get_voxel_centers(vgs_ctx* ctx, float* center3V);          /* getVoxelCenters (VS:191) */
vgs_status vgs_get_point_voxel(vgs_ctx* ctx, int32_t* voxel_of_point /* N, -1 = not in octree */);
/* per-node attributes (voxels for VGS, supervoxels for SVGS): centroid 3*V, normal 3*V, eigen 8*V, used V */
vgs_status vgs_get_attributes(vgs_ctx* ctx, float* centroid, float* normal, float* eigen8, uint8_t* used);
/* ragged lists, two-call protocol: pass idx == NULL to get offsets (n_nodes+1, int64) and the total first.
 * which: 0 adjacency (getOneVoxelAdjacency order, VS:268; every voxel, used or not, with every neighbour, as findAllVoxelAdjacency
 *        builds them VS:236-263 -- computed on request, the hot path keeps only what the cuts read), 1 connect lists after the local cut,
 *        2 after crossValidation (VS:2111), 3 after closestCheck (VS:2181) */
vgs_status vgs_get_lists(vgs_ctx* ctx, int32_t which, int64_t* offsets, int32_t* idx);
/* Element order of the connect lists (which >= 1) and of getClusterIdx.  VGS_ORDER_VOXEL_ID: members in the order of the
 * adjacency row / ascending voxel id -- what the hot path keeps (a flag per adjacency slot).  VGS_ORDER_REFERENCE: the
 * reference's own order: cutGraphSegmentation returns its vertex list in merge-history order (every merge appends the absorbed
 * segment's vertices, VS:1986-1998, 2003-2026), crossValidation filters it in place, closestCheck appends (VS:2293-2294).
 * Computed on request by replaying the scan inside every list that was found (csrc/cutorder.hip); ties as the oracle's lean
 * flavour (the reference's std::sort of both orientations of a pair leaves them unspecified). */
enum { VGS_ORDER_VOXEL_ID = 0, VGS_ORDER_REFERENCE = 1 };
vgs_status vgs_get_lists_ordered(vgs_ctx* ctx, int32_t which, int32_t order, int64_t* offsets, int32_t* idx);
/* voxels_adjacency_idx_[v][0] (VS:253): the number of neighbours of every node inside graph_size, itself included; 0 for a
 * node that has no list (unused voxels, see `which` above) */
vgs_status vgs_get_adjacency_counts(vgs_ctx* ctx, int32_t* n_all /* n_nodes */);
/* The affinity matrix buildAdjacencyGraph fills for one node (VS:1796-1910; SS twin): weights[a * n + b] =
 * distanceWeight(measuringDistance(ids[a], ids[b])), ids = the node's stored adjacency row (the used neighbours when unused
 * voxels are inert, every neighbour otherwise), ids[0] = the node itself.  Two-call protocol: ids == NULL returns n (0 for a
 * node without a local graph).  Diagnostics / parity tests: the hot path never materialises this matrix. */
vgs_status vgs_get_local_weights(vgs_ctx* ctx, int32_t node_id, int32_t* n, int32_t* ids, float* weights);
vgs_status vgs_get_node_labels(vgs_ctx* ctx, int32_t* component_root /* V: smallest node id of its cluster */,
                               int32_t* kept_label /* V: index into kept clusters or -1 */);
vgs_status vgs_get_point_labels(vgs_ctx* ctx, int32_t* labels /* N host; -1 = dropped */);
/* starts the copy of the labels to `labels` (N int32, ideally pinned) on a copy stream and returns; the next run of the
 * stages writes a second label buffer, so the copy overlaps it.  vgs_wait_point_labels blocks until `labels` is complete.
 * One copy in flight per context: a second call waits for the first. */
vgs_status vgs_get_point_labels_async(vgs_ctx* ctx, int32_t* labels);
vgs_status vgs_wait_point_labels(vgs_ctx* ctx);
vgs_status vgs_get_point_labels_device(vgs_ctx* ctx, const int32_t** labels_dev /* N, valid until next run */);
/* getClusterIdx (VS:117): offsets (kept+1, int64) and point indices grouped by cluster (cluster order =
 * ascending smallest voxel id, as the reference; inside a cluster ascending voxel id then point index) */
vgs_status vgs_get_clusters(vgs_ctx* ctx, int64_t* offsets, int32_t* point_idx);
/* The same with a choice of the order inside a cluster.  VGS_ORDER_REFERENCE is the reference's own: nodes in the pre-order
 * of recursionSearch over the final connect lists with the seed appended LAST (VS:2032-2053, 2064-2080; SS:2079-2103), the
 * points of each node in ascending index (VS:981-999; SS:2109-2126) -- element for element what getClusterIdx() holds.
 * (Output formatting: the walk runs on the host over the downloaded lists.) */
vgs_status vgs_get_clusters_ordered(vgs_ctx* ctx, int32_t order, int64_t* offsets, int32_t* point_idx);
/* getClusterIdx left in HBM (round 5; csrc/clusters.hip): the lists of vgs_get_clusters (default order) as device pointers --
 * offsets_dev[kept + 1] (int64), point_idx_dev[offsets[kept]] (int32) -- made by one stable sort of the leaf order by label; valid
 * until the next run of the stages.  vgs_get_clusters[_ordered] with the default order copies exactly these to the host. */
vgs_status vgs_get_clusters_device(vgs_ctx* ctx, const int64_t** offsets_dev, const int32_t** point_idx_dev);

/* ---- multi-GPU support (spatial tiles) -------------------------------------- */
/* The reference is single-process; these entry points are what a tiled driver needs around the same stages.
 * One context per GPU holds one tile plus a halo of raw points (2*graph_size + voxel_size wide).
 * Shared grid: the octree growth state (PCL OctreePointCloud box) is chained rank to rank. */
typedef struct { double min[3]; uint64_t shift[3]; int32_t depth; int32_t defined; } vgs_grid_state;
vgs_status vgs_grid_state_init(vgs_grid_state* g);
/* advance g over this context's points in index order (what inserting them after all earlier ranks' points does
 * to the octree box); call on rank r after receiving g from rank r-1 */
vgs_status vgs_grid_advance(vgs_ctx* ctx, vgs_grid_state* g);
/* shortcut of the chain: bounding box (min x, y, z, max x, y, z) and number of the finite points of this context's cloud,
 * and the replay of the growth over a cloud of which only that box is known (host arithmetic, no context).  The replay
 * advances g while the growth step is the same for every point outside the box and sets *need_scan when it is not (or
 * when g is still undefined): then vgs_grid_advance on the owner of the cloud continues from the state reached. */
vgs_status vgs_points_bbox(vgs_ctx* ctx, float* bbox6, int64_t* n_finite);
vgs_status vgs_grid_advance_bbox(vgs_grid_state* g, double voxel_size, const float* bbox6, int32_t* need_scan);
/* pin the final grid before vgs_voxelize: every rank bins with the state left by the last rank */
vgs_status vgs_set_grid(vgs_ctx* ctx, const vgs_grid_state* g);
/* the same from a caller that has replayed the growth over THIS context's cloud itself (vgs_grid_advance on it, or
 * vgs_grid_advance_bbox over its bounding box, as the tiled driver does for every rank): the voxelize stage then takes the grid
 * as final without scanning the points for one outside it (0.2 ms of a 10 M-point tile's step).  A point outside such a grid
 * is the caller's error and is binned with a wrapped key; vgs_set_grid keeps the check.  Reset by the next vgs_set_points. */
vgs_status vgs_set_grid_covering(vgs_ctx* ctx, const vgs_grid_state* g);
/* this rank owns the voxels whose centre lies in [lo, hi) in x and y; others are halo (computed redundantly,
 * their own connections are not trusted).  Components are then built from the connections that have an owned
 * endpoint, cluster sizes count owned voxels, and point labels wait for vgs_apply_root_labels. */
vgs_status vgs_set_owned_region(vgs_ctx* ctx, const double* lo_xy, const double* hi_xy);
/* A rank's cloud = the points it loaded itself + the border strips of the other ranks, assembled IN RANK ORDER: strips of lower
 * ranks, own points, strips of higher ranks -- the order in which a single process would have inserted them, because a voxel's
 * attributes depend on the order of its points (sequential float sums, the normal's flip looks at the voxel's first point, VS:1364,
 * 1394).  Points [first, first + n_own) are the rank's own load.  A voxel that holds points of both kinds (an object that reaches over
 * a tile's edge) becomes a boundary voxel on the rank that owns it and on the rank that loaded the foreign points, so the latter
 * learns the label of those points from the owner's record.  Call after vgs_set_points*; n_own = -1 switches it off. */
vgs_status vgs_set_own_point_range(vgs_ctx* ctx, int64_t first, int64_t n_own);
/* after vgs_segment: (global voxel code, local component root) records of the boundary voxels -- both endpoints of
 * every connection that crosses the ownership border, and every owned voxel with a halo voxel in its neighbourhood
 * (a possible closestCheck target of the neighbouring rank); duplicates possible.  Records of all ranks that share
 * a code name the same segment.  Two-call protocol: code == NULL returns the count. */
vgs_status vgs_get_boundary(vgs_ctx* ctx, int64_t* n_records, uint64_t* code, int32_t* root);
/* local component roots that contain owned voxels, with the number of owned voxels (two-call protocol) */
vgs_status vgs_get_owned_roots(vgs_ctx* ctx, int64_t* n_roots, int32_t* root, int32_t* owned_voxels);
/* final labels: label[k] for local root root[k] (-1 = dropped); points of halo voxels and of unlisted roots get -1 */
vgs_status vgs_apply_root_labels(vgs_ctx* ctx, const int32_t* root, const int32_t* label, int64_t n_roots);
/* Compact form of the two calls above, everything but the border stays on the GPU: the boundary records without
 * duplicates (one per boundary voxel: code, local root, number of owned voxels of that root), and the number of
 * local components that touch no boundary record and pass the `> voxels_min` filter on their own (their labels are
 * local_base + rank in ascending root order).  Two-call protocol: code == NULL computes and returns the counts. */
vgs_status vgs_get_boundary_roots(vgs_ctx* ctx, int64_t* n_records, uint64_t* code, int32_t* root, int32_t* owned_voxels,
                                  int64_t* n_kept_local);
/* final labels of a tile: label[k] for boundary root root[k] (-1 = dropped), local_base + rank for the kept
 * components without boundary records, -1 for everything else and for the points of halo voxels */
vgs_status vgs_apply_tile_labels(vgs_ctx* ctx, int32_t local_base, const int32_t* root, const int32_t* label, int64_t n_roots);

#ifdef __cplusplus
}
#endif
#endif /* VGS_H_ */
