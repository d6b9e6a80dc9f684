// capi.hip -- the C-ABI of include/vgs.h: parameter surface, context lifetime, stage drivers with the
// reference's call-order contract turned into status codes, result getters (two-call size queries,
// caller-allocated outputs).  Host code only; kernels live in the stage files.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <string>
#include <vector>

#include <dlfcn.h>

#include "vgs_context.hpp"

static thread_local std::string g_create_err;

static const char* const VGS_STAGE_NAMES[VGS_T_COUNT] = {"vgs:voxelize", "vgs:features", "vgs:adjacency", "vgs:localcut", "vgs:merge", "vgs:labels",
                                                         "vgs:total", "vgs:localcut_kernel", "vgs:supervoxels", "vgs:localcut_bulk", "", ""};
// A stage = one roctx range (rocprofv3 --marker-trace shows the stages on the host timeline).  The marker library is looked up at
// run time (dlopen: the profiler SDK's own library, the older libroctx64 otherwise): a host without it still builds and loads the
// engine, and the ranges are then no-ops.
struct RoctxApi {
  int (*push)(const char*) = nullptr;
  int (*pop)() = nullptr;
  RoctxApi() {
    for (const char* name : {"librocprofiler-sdk-roctx.so.1", "librocprofiler-sdk-roctx.so", "libroctx64.so.4", "libroctx64.so"}) {
      void* h = dlopen(name, RTLD_LAZY | RTLD_LOCAL);
      if (!h) continue;
      push = (int (*)(const char*))dlsym(h, "roctxRangePushA");
      pop = (int (*)())dlsym(h, "roctxRangePop");
      if (push && pop) return;
      push = nullptr; pop = nullptr;
    }
  }
};
static const RoctxApi& roctx_api() { static const RoctxApi api; return api; }
struct RoctxRange {
  bool on;
  explicit RoctxRange(const char* name) : on(roctx_api().push != nullptr) { if (on) roctx_api().push(name); }
  ~RoctxRange() { if (on) roctx_api().pop(); }
};

// Time a stage with HIP events on the context's stream.  The stage is NOT waited for here (round 3): a stage function waits where
// it needs a number on the host, and a wait only for the clock left the GPU idle for a host round trip after voxelize, adjacency and
// the local cut.  The pair of events is read when the times are asked for or at the end of vgs_segment (vgs_resolve_times).
// wait = true: the caller gets the stage complete (a single stage called through the C-ABI: the getters that may follow copy
// with blocking calls that do not order against the context's stream); false: inside vgs_run / vgs_segment, whose last stage ends
// with a read-back behind everything.
template <typename F>
static vgs_status timed(vgs_ctx* c, int slot, F&& f, bool wait = true) {
  RoctxRange range(VGS_STAGE_NAMES[slot]);
  VGS_HIP_TRY(c, hipSetDevice(c->device));
  VGS_HIP_TRY(c, hipEventRecord(c->tev[slot][0], c->stream));
  c->tev_pending[slot] = false;
  vgs_status s = f();
  if (s != VGS_OK) return s;
  VGS_HIP_TRY(c, hipEventRecord(c->tev[slot][1], c->stream));
  c->tev_pending[slot] = true;
  if (wait) VGS_HIP_TRY(c, hipEventSynchronize(c->tev[slot][1]));
  return VGS_OK;
}

static vgs_status vgs_resolve_times(vgs_ctx* c) {
  for (int slot = 0; slot < VGS_T_COUNT; ++slot) {
    if (!c->tev_pending[slot]) continue;
    c->tev_pending[slot] = false;
    VGS_HIP_TRY(c, hipEventSynchronize(c->tev[slot][1]));
    float ms = 0.f;
    VGS_HIP_TRY(c, hipEventElapsedTime(&ms, c->tev[slot][0], c->tev[slot][1]));
    c->times[slot] = ms;
  }
  return VGS_OK;
}

void vgs_read_env_knobs(vgs_ctx* c) {
  VgsKnobs& k = c->K;
  auto geti = [](const char* n, int d) { const char* v = getenv(n); return v ? atoi(v) : d; };
  auto getf = [](const char* n, float d) { const char* v = getenv(n); return v ? (float)atof(v) : d; };
  auto has = [](const char* n) { return getenv(n) != nullptr; };
  k.a1_max = geti("VGS_A1MAX", k.a1_max); k.shell0 = getf("VGS_SHELL0", k.shell0); k.cap_frac = getf("VGS_CAPFRAC", k.cap_frac);
  k.dbg_stop = geti("VGS_DBG_STOP", k.dbg_stop); k.max_rounds = geti("VGS_ROUNDS", k.max_rounds); k.dbg_max_m = geti("VGS_DBG_MAXM", k.dbg_max_m); k.dbg_xl_from = geti("VGS_DBG_XL_FROM", k.dbg_xl_from);
  k.near_min_own = geti("VGS_NEARMINOWN", k.near_min_own); k.fv_blocks = geti("VGS_FV_BLOCKS", k.fv_blocks); k.only_class = geti("VGS_ONLY_CLASS", k.only_class);
  k.no_dense = has("VGS_NO_DENSE"); k.no_overlap = has("VGS_NO_OVERLAP"); k.no_near = has("VGS_NO_NEAR"); k.no_adjmasks = has("VGS_NO_ADJMASKS"); k.no_connbits = has("VGS_NO_CONNBITS"); k.no_c0 = has("VGS_NO_C0"); k.no_pg_xl = has("VGS_NO_PG_XL"); k.no_sort32 = has("VGS_NO_SORT32"); k.no_grow_prefix = has("VGS_NO_GROW_PREFIX"); k.no_adj_wide = has("VGS_NO_ADJ_WIDE"); k.vccs_pingpong = has("VGS_VCCS_PINGPONG"); k.no_vccs_tiles = has("VGS_NO_VCCS_TILES"); k.no_early_union = has("VGS_NO_EARLY_UNION"); k.no_packed_sort = has("VGS_NO_PACKED_SORT"); k.no_pairlists = has("VGS_NO_PAIRLISTS"); k.no_vote = has("VGS_NO_VOTE"); k.vote_force = geti("VGS_VOTE_FORCE", 0); k.cross_lds_kb = geti("VGS_CROSS_LDS", k.cross_lds_kb); k.no_tile_early = has("VGS_NO_TILE_EARLY"); { int vp = geti("VGS_VOTE_PERIOD", k.vote_period); if (vp >= 2 && vp <= 4096 && (vp & (vp - 1)) == 0) k.vote_period = vp; } k.pg_wide = geti("VGS_PG_WIDE", k.pg_wide); k.pg_wide_frac = geti("VGS_PG_WIDEFRAC", k.pg_wide_frac); k.pg_min_frac = geti("VGS_PG_MINFRAC", k.pg_min_frac); k.ho_grid = geti("VGS_HO_GRID", k.ho_grid); k.vccs_nbr_normals = has("VGS_VCCS_NBR_NORMALS");
  k.debug = has("VGS_DEBUG");
}

// The hand-over kernels of the local cut end inside the merge stage's timed region (vgs_localcut_finish measures by how much):
// that tail belongs to the local cut in the reported stage times.
static void vgs_charge_localcut_tail(vgs_ctx* c) {
  (void)vgs_resolve_times(c);   // the merge stage ended with a read-back: every event of the step has completed
  const double t = (double)c->lc_tail.tail_ms;
  if (t > 0.0 && t < c->times[VGS_T_MERGE]) { c->times[VGS_T_LOCALCUT] += t; c->times[VGS_T_MERGE] -= t; }
  c->lc_tail.tail_ms = 0.f;
}

extern "C" {

vgs_status vgs_params_default_vgs(vgs_params* p) {
  if (!p) return VGS_E_ARG;
  std::memset(p, 0, sizeof(*p));
  p->method = 2;
  p->voxel_size = 0.15f; p->graph_size = 0.5f;
  p->sig_p = p->sig_n = p->sig_o = p->sig_e = p->sig_c = 0.2f;
  p->sig_w = 2.0f; p->cut_thred = 0.3f;
  p->points_min = 10; p->adjacency_min = 3; p->voxels_min = 3;
  p->seed_size = 0.25f; p->color_impt = 0.0f; p->spatial_impt = 0.25f; p->normal_impt = 0.75f;
  p->q7_count_as_index = 1;
  p->device = 0;
  return VGS_OK;
}

vgs_status vgs_params_default_svgs(vgs_params* p) {
  vgs_status s = vgs_params_default_vgs(p);
  if (s != VGS_OK) return s;
  p->method = 3;
  p->voxel_size = 0.05f; p->seed_size = 0.25f; p->graph_size = 0.5f;
  p->sig_w = 1.0f; p->cut_thred = 0.5f;
  // createSupervoxels (SS:265-284) calls pcl::SupervoxelClustering::extract + refineSupervoxels(5): supervoxels in PCL's own order are what
  // SVGS means.  (Round 6: the synchronous variant, mode 0, stays as an opt-in approximation -- its FINAL segments agree with this mode's on
  // 92-97 % of the points only, far outside P2: tests/test_gpu_vccs.py::test_synchronous_variant_is_not_within_p2_of_pcl_order.)
  p->vccs_mode = 1;
  return VGS_OK;
}

// inputTaskTxtFile (IOC:148-169): every line into a vector; the drivers then index fixed line numbers.
// Unlike the reference, CR is stripped (the shipped task files are CRLF).
vgs_status vgs_parse_task_file(const char* path, vgs_params* p, char* in_name, char* out_name, int name_cap) {
  if (!path || !p) return VGS_E_ARG;
  std::ifstream f(path);
  if (!f.is_open()) return VGS_E_IO;
  std::vector<std::string> v;
  std::string line;
  while (std::getline(f, line)) {
    while (!line.empty() && (line.back() == '\r' || line.back() == '\n')) line.pop_back();
    v.push_back(line);
  }
  auto num = [&](size_t k) -> double { return k < v.size() ? std::atof(v[k].c_str()) : 0.0; };
  auto inum = [&](size_t k) -> int { return k < v.size() ? std::atoi(v[k].c_str()) : 0; };
  if (v.size() < 51) return VGS_E_IO;
  const int method = inum(24);
  if (method == 3) {
    if (v.size() < 61) return VGS_E_IO;
    vgs_params_default_svgs(p);
    // T:108-125
    p->voxel_size = (float)num(28); p->seed_size = (float)num(30); p->graph_size = (float)num(32);
    p->sig_p = (float)num(34); p->sig_n = (float)num(36); p->sig_o = (float)num(38); p->sig_e = (float)num(40);
    p->sig_c = (float)num(42); p->sig_w = (float)num(44);
    p->color_impt = (float)num(46); p->spatial_impt = (float)num(48); p->normal_impt = (float)num(50);
    p->cut_thred = (float)num(52);
    p->points_min = (int)num(54); p->voxels_min = inum(58); p->adjacency_min = inum(60);
  } else {
    vgs_params_default_vgs(p);
    // T:25-37
    p->voxel_size = (float)num(28); p->graph_size = (float)num(30);
    p->sig_p = (float)num(32); p->sig_n = (float)num(34); p->sig_o = (float)num(36); p->sig_e = (float)num(38);
    p->sig_c = (float)num(40); p->sig_w = (float)num(42); p->cut_thred = (float)num(44);
    p->points_min = inum(46); p->adjacency_min = inum(48); p->voxels_min = inum(50);
    p->method = 2;
  }
  if (in_name && name_cap > 0) { std::strncpy(in_name, v[15].c_str(), name_cap - 1); in_name[name_cap - 1] = 0; }
  if (out_name && name_cap > 0) { std::strncpy(out_name, v[21].c_str(), name_cap - 1); out_name[name_cap - 1] = 0; }
  return VGS_OK;
}

vgs_status vgs_create(const vgs_params* p, vgs_ctx** out) {
  if (!p || !out) return VGS_E_ARG;
  *out = nullptr;
  if (!(p->voxel_size > 0.f) || !(p->graph_size > 0.f) || !(p->sig_w != 0.f) || (p->method != 2 && p->method != 3)) {
    g_create_err = "vgs_create: voxel_size, graph_size must be > 0, sig_w != 0, method 2 or 3";
    return VGS_E_ARG;
  }
  if (p->vccs_mode != 0 && p->vccs_mode != 1) { g_create_err = "vgs_create: vccs_mode must be 0 (synchronous variant) or 1 (PCL's order)"; return VGS_E_ARG; }
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev <= 0) {
    g_create_err = std::string("vgs_create: no HIP device (") + hipGetErrorString(e) + "); this engine has no CPU path";
    return VGS_E_HIP;
  }
  if (p->device < 0 || p->device >= ndev) { g_create_err = "vgs_create: bad device ordinal"; return VGS_E_ARG; }
  vgs_ctx* c = new (std::nothrow) vgs_ctx();
  if (!c) return VGS_E_NOMEM;
  c->P = *p;
  c->device = p->device;
  // the side streams carry few, long-running workgroups that must not queue behind the bulk class: highest priority
  int prio_lo = 0, prio_hi = 0;
  if (hipSetDevice(c->device) != hipSuccess || hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess ||
      hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi) != hipSuccess ||
      hipStreamCreateWithPriority(&c->stream2, hipStreamNonBlocking, prio_hi) != hipSuccess ||
      hipStreamCreateWithPriority(&c->stream3, hipStreamNonBlocking, prio_hi) != hipSuccess ||
      hipStreamCreateWithPriority(&c->stream4, hipStreamNonBlocking, prio_hi) != hipSuccess) {
    g_create_err = "vgs_create: hipSetDevice/hipStreamCreate failed";
    vgs_destroy(c);
    return VGS_E_HIP;
  }
  if (hipHostMalloc(&c->pin, 4096, hipHostMallocDefault) != hipSuccess) c->pin = nullptr;   // read-backs fall back to pageable copies
  bool ok = hipStreamCreateWithFlags(&c->s_h2d, hipStreamNonBlocking) == hipSuccess &&
            hipStreamCreateWithFlags(&c->s_d2h, hipStreamNonBlocking) == hipSuccess &&
            hipEventCreateWithFlags(&c->ev_h2d, hipEventDisableTiming) == hipSuccess &&
            hipEventCreateWithFlags(&c->ev_d2h, hipEventDisableTiming) == hipSuccess &&
            hipEventCreateWithFlags(&c->ev_rb, hipEventDisableTiming) == hipSuccess &&
            hipEventCreateWithFlags(&c->ev_ho, hipEventDisableTiming) == hipSuccess &&
            hipEventCreateWithFlags(&c->ev_ho2, hipEventDisableTiming) == hipSuccess;
  for (int i = 0; ok && i < 16; ++i) ok = hipEventCreate(&c->ev[i]) == hipSuccess;
  for (int i = 0; ok && i < VGS_T_COUNT; ++i) ok = hipEventCreate(&c->tev[i][0]) == hipSuccess && hipEventCreate(&c->tev[i][1]) == hipSuccess;
  if (!ok) { g_create_err = "vgs_create: hipStreamCreate/hipEventCreate failed"; vgs_destroy(c); return VGS_E_HIP; }   // frees what exists
  vgs_read_env_knobs(c);
  *out = c;
  return VGS_OK;
}

void vgs_destroy(vgs_ctx* c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  // the side streams may still carry hand-over kernels if a stage sequence was cut short
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  if (c->stream2) (void)hipStreamSynchronize(c->stream2);
  if (c->stream3) (void)hipStreamSynchronize(c->stream3);
  if (c->stream4) (void)hipStreamSynchronize(c->stream4);
  if (c->s_h2d) (void)hipStreamSynchronize(c->s_h2d);
  if (c->s_d2h) (void)hipStreamSynchronize(c->s_d2h);
  c->xyz_buf[0].release(); c->xyz_buf[1].release(); c->pt_label_alt.release(); c->grow_state.release();
  c->code_a.release(); c->code_b.release(); c->perm_a.release(); c->perm_b.release(); c->sort_tmp.release();
  c->head_flag.release(); c->pt_vox.release(); c->vox_code.release(); c->vox_start.release();
  c->xs.release(); c->ys.release(); c->zs.release();
  c->node.release(); c->used_ids.release(); c->used_rank.release();
  c->hkey.release(); c->hval.release(); c->offsets.release(); c->adj_masks.release(); c->adj_gtab.release(); c->adj_nvals.release(); c->adj_nrank.release(); c->adj_key.release(); c->adj_off.release(); c->adj_cnt.release(); c->adj_mused.release();
  c->nl_cnt.release(); c->nl_tot.release(); c->nl_ent.release(); c->lc_ctab.release(); c->pl_state.release(); c->pl_ent.release(); c->pl_work.release();
  c->cl_off.release(); c->cl_idx.release(); c->conn.release(); c->evals.release(); c->lc_pending.release(); c->lc_defer.release(); c->lc_defer_flag.release(); c->csize.release(); c->attach.release(); c->cc_flags.release(); c->parent.release(); c->csz.release();
  c->vc_cen.release(); c->vc_nrm.release(); c->vc_dist.release(); c->vc_state.release(); c->vc_nbr.release(); c->vc_nbr4.release(); c->vc_tile_start.release(); c->vc_cell.release(); c->vc_plive.release(); c->vc_tile_of.release(); c->vc_tchg.release(); c->vc_nbr_tiles.release(); c->vc_halo.release(); c->vc_tile_meta.release(); c->vc_pool.release(); c->vc_ring.release(); c->vc_dbg.release(); c->vc_label.release();
  c->vc_seedkey.release(); c->vc_sums.release(); c->vc_count.release(); c->vc_accu.release(); c->vc_live.release(); c->vc_alive.release();
  c->sv_label.release(); c->sv_key_a.release(); c->sv_key_b.release(); c->cell_code_a.release(); c->cell_code_b.release();
  c->cell_id_a.release(); c->cell_id_b.release(); c->cell_start.release();
  c->owned.release(); c->straddle.release(); c->mixsrc.release(); c->bnd_code.release(); c->bnd_root.release(); c->root_label.release();
  c->bnd_code2.release(); c->bnd_root2.release(); c->bnd_cnt.release(); c->broot.release();
  c->kept_rank.release(); c->vox_label.release(); c->pt_label.release(); c->counters.release(); c->work_ids.release();
  if (c->pin) (void)hipHostFree(c->pin);
  for (int i = 0; i < 16; ++i) if (c->ev[i]) (void)hipEventDestroy(c->ev[i]);
  for (int i = 0; i < VGS_T_COUNT; ++i) for (int k = 0; k < 2; ++k) if (c->tev[i][k]) (void)hipEventDestroy(c->tev[i][k]);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  if (c->stream2) (void)hipStreamDestroy(c->stream2);
  if (c->stream3) (void)hipStreamDestroy(c->stream3);
  if (c->stream4) (void)hipStreamDestroy(c->stream4);
  if (c->s_h2d) (void)hipStreamDestroy(c->s_h2d);
  if (c->s_d2h) (void)hipStreamDestroy(c->s_d2h);
  if (c->ev_h2d) (void)hipEventDestroy(c->ev_h2d);
  if (c->ev_rb) (void)hipEventDestroy(c->ev_rb);
  if (c->ev_ho) (void)hipEventDestroy(c->ev_ho);
  if (c->ev_ho2) (void)hipEventDestroy(c->ev_ho2);
  if (c->ev_d2h) (void)hipEventDestroy(c->ev_d2h);
  delete c;
}

vgs_status vgs_set_params(vgs_ctx* c, const vgs_params* p) {
  if (!c || !p) return VGS_E_ARG;
  if (!(p->voxel_size > 0.f) || !(p->graph_size > 0.f) || !(p->sig_w != 0.f)) { c->err = "vgs_set_params: voxel_size, graph_size > 0, sig_w != 0"; return VGS_E_ARG; }
  if (p->method != c->P.method || p->device != c->P.device) { c->err = "vgs_set_params: method and device are fixed at vgs_create"; return VGS_E_ARG; }
  if (p->vccs_mode != 0 && p->vccs_mode != 1) { c->err = "vgs_set_params: vccs_mode must be 0 (synchronous variant) or 1 (PCL's order)"; return VGS_E_ARG; }
  int keep = ST_SEGMENTED;
  const vgs_params& o = c->P;
  if (p->cut_thred != o.cut_thred || p->sig_p != o.sig_p || p->sig_n != o.sig_n || p->sig_o != o.sig_o || p->sig_e != o.sig_e ||
      p->sig_c != o.sig_c || p->sig_w != o.sig_w || p->adjacency_min != o.adjacency_min || p->voxels_min != o.voxels_min ||
      p->q7_count_as_index != o.q7_count_as_index)
    keep = ST_ADJACENCY;
  if (p->graph_size != o.graph_size) keep = ST_FEATURES;
  if (p->points_min != o.points_min) keep = ST_VOXELS;
  if (p->voxel_size != o.voxel_size || p->seed_size != o.seed_size || p->color_impt != o.color_impt ||
      p->spatial_impt != o.spatial_impt || p->normal_impt != o.normal_impt || p->vccs_mode != o.vccs_mode)
    keep = ST_POINTS;
  if (c->stage > keep) c->stage = keep;
  // labels from svgs_supervoxels depend on voxel_size / seed_size / the three importances; a caller's own labelling does not
  if (keep == ST_POINTS && c->P.method == 3 && !c->sv_labels_external) { c->sv_have_labels = false; c->sv_max_label = 0; c->sv_label_n = -1; }
  c->P = *p;
  return VGS_OK;
}

const char* vgs_last_error_string(const vgs_ctx* c) { return c ? c->err.c_str() : g_create_err.c_str(); }

static vgs_status set_points_common(vgs_ctx* c, int64_t n, int32_t stride_bytes) {
  if (n < 0 || (stride_bytes != 12 && stride_bytes != 16)) { c->err = "set_points: n >= 0 and stride_bytes 12 or 16 required"; return VGS_E_ARG; }
  if (n >= (int64_t)1 << 31) { c->err = "set_points: more than 2^31-1 points"; return VGS_E_UNSUPPORTED; }
  c->N = n;
  c->stride_f = stride_bytes / 4;
  c->stage = ST_POINTS;
  c->grid_covers = false;   // (vgs_set_grid_covering vouches for one cloud)
  c->counts[VGS_N_POINTS] = n;
  for (int i = 0; i < VGS_T_COUNT; ++i) { c->times[i] = 0; c->tev_pending[i] = false; }
  // a supervoxel labelling belongs to the cloud it was made for (SS:279-331 rebuilds it per createSupervoxels call)
  c->sv_have_labels = false; c->sv_labels_external = false; c->sv_max_label = 0; c->sv_label_n = -1;
  c->n_own = -1; c->own_first = 0;   // tiles: vgs_set_own_point_range follows the cloud
  return VGS_OK;
}

vgs_status vgs_set_points(vgs_ctx* c, const float* xyz_host, int64_t n, int32_t stride_bytes) {
  if (!c || (!xyz_host && n > 0)) return VGS_E_ARG;
  VGS_HIP_TRY(c, hipSetDevice(c->device));
  vgs_status s = set_points_common(c, n, stride_bytes);
  if (s != VGS_OK) return s;
  DevBuf<float>& buf = c->xyz_buf[c->xyz_cur];   // a staged cloud, if any, sits in the other one
  VGS_HIP_TRY(c, buf.ensure((size_t)n * c->stride_f + 4));
  if (n > 0) VGS_HIP_TRY(c, hipMemcpyAsync(buf.p, xyz_host, (size_t)n * stride_bytes, hipMemcpyHostToDevice, c->stream));
  VGS_HIP_TRY(c, hipStreamSynchronize(c->stream));
  c->xyz = buf.p;
  return VGS_OK;
}

// ---- a sequence of clouds: the next upload and the last download overlap the stages of the current cloud --------------
vgs_status vgs_stage_points(vgs_ctx* c, const float* xyz_host, int64_t n, int32_t stride_bytes) {
  if (!c || (!xyz_host && n > 0)) return VGS_E_ARG;
  if (n < 0 || (stride_bytes != 12 && stride_bytes != 16)) { c->err = "vgs_stage_points: n >= 0 and stride_bytes 12 or 16 required"; return VGS_E_ARG; }
  if (n >= (int64_t)1 << 31) { c->err = "vgs_stage_points: more than 2^31-1 points"; return VGS_E_UNSUPPORTED; }
  VGS_HIP_TRY(c, hipSetDevice(c->device));
  if (c->staged) VGS_HIP_TRY(c, hipEventSynchronize(c->ev_h2d));   // a staged cloud that was never committed is replaced
  DevBuf<float>& buf = c->xyz_buf[1 - c->xyz_cur];
  VGS_HIP_TRY(c, buf.ensure((size_t)n * (stride_bytes / 4) + 4));
  if (n > 0) VGS_HIP_TRY(c, hipMemcpyAsync(buf.p, xyz_host, (size_t)n * stride_bytes, hipMemcpyHostToDevice, c->s_h2d));
  VGS_HIP_TRY(c, hipEventRecord(c->ev_h2d, c->s_h2d));
  c->staged = true; c->staged_n = n; c->staged_stride = stride_bytes;
  return VGS_OK;
}

vgs_status vgs_commit_points(vgs_ctx* c) {
  if (!c) return VGS_E_ARG;
  if (!c->staged) { c->err = "vgs_commit_points: no staged cloud (vgs_stage_points first)"; return VGS_E_STATE; }
  VGS_HIP_TRY(c, hipSetDevice(c->device));
  VGS_HIP_TRY(c, hipStreamWaitEvent(c->stream, c->ev_h2d, 0));   // the stages wait on the device, the host does not
  vgs_status s = set_points_common(c, c->staged_n, c->staged_stride);
  if (s != VGS_OK) return s;
  c->xyz_cur = 1 - c->xyz_cur;
  c->xyz = c->xyz_buf[c->xyz_cur].p;
  c->staged = false;
  return VGS_OK;
}

vgs_status vgs_host_alloc(void** p, uint64_t bytes) {
  if (!p) return VGS_E_ARG;
  *p = nullptr;
  return hipHostMalloc(p, bytes > 0 ? (size_t)bytes : 1, hipHostMallocDefault) == hipSuccess ? VGS_OK : VGS_E_NOMEM;
}
vgs_status vgs_host_free(void* p) { return (!p || hipHostFree(p) == hipSuccess) ? VGS_OK : VGS_E_HIP; }
vgs_status vgs_host_register(void* p, uint64_t bytes) {
  if (!p || bytes == 0) return VGS_E_ARG;
  return hipHostRegister(p, (size_t)bytes, hipHostRegisterDefault) == hipSuccess ? VGS_OK : VGS_E_HIP;
}
vgs_status vgs_host_unregister(void* p) { return (p && hipHostUnregister(p) == hipSuccess) ? VGS_OK : VGS_E_HIP; }

vgs_status vgs_set_points_device(vgs_ctx* c, const float* xyz_dev, int64_t n, int32_t stride_bytes) {
  if (!c || (!xyz_dev && n > 0)) return VGS_E_ARG;
  VGS_HIP_TRY(c, hipSetDevice(c->device));
  vgs_status s = set_points_common(c, n, stride_bytes);
  if (s != VGS_OK) return s;
  c->xyz = xyz_dev;
  return VGS_OK;
}

}  // extern "C"

static vgs_status do_voxelize(vgs_ctx* c, bool wait) {
  if (!c) return VGS_E_ARG;
  if (c->stage < ST_POINTS) { c->err = "vgs_voxelize: no input cloud (setInputCloud/getCloudPointNum first)"; return VGS_E_STATE; }
  vgs_status s = timed(c, VGS_T_VOXELIZE, [&] { return vgs_stage_voxelize(c); }, wait);
  if (s == VGS_OK) c->stage = ST_VOXELS;
  return s;
}
static vgs_status do_features(vgs_ctx* c, bool wait) {
  if (!c) return VGS_E_ARG;
  if (c->stage < ST_VOXELS) { c->err = "vgs_features: voxel table missing (setVoxelCenters/getVoxelNum must precede calcualteVoxelCloudAttributes)"; return VGS_E_STATE; }
  vgs_status s = timed(c, VGS_T_FEATURES, [&] { return vgs_stage_features(c); }, wait);
  if (s == VGS_OK) c->stage = ST_FEATURES;
  return s;
}
static vgs_status do_adjacency(vgs_ctx* c, bool wait) {
  if (!c) return VGS_E_ARG;
  if (c->stage < ST_FEATURES) { c->err = "vgs_adjacency: attributes missing"; return VGS_E_STATE; }
  vgs_status s = timed(c, VGS_T_ADJACENCY, [&] { return vgs_stage_adjacency(c); }, wait);
  if (s == VGS_OK) c->stage = ST_ADJACENCY;
  return s;
}

extern "C" {

vgs_status vgs_voxelize(vgs_ctx* c) { return do_voxelize(c, true); }
vgs_status vgs_features(vgs_ctx* c) { return do_features(c, true); }
vgs_status vgs_adjacency(vgs_ctx* c) { return do_adjacency(c, true); }

vgs_status vgs_segment(vgs_ctx* c) {
  if (!c) return VGS_E_ARG;
  if (c->stage < ST_ADJACENCY) { c->err = "vgs_segment: adjacency missing (findAllVoxelAdjacency first)"; return VGS_E_STATE; }
  if (c->U > 0 && vgs_unused_are_inert(c->P) != c->adj_pruned) {  // sigma/cut changed what the rows must hold
    vgs_status sa = timed(c, VGS_T_ADJACENCY, [&] { return vgs_stage_adjacency(c); }, false);
    if (sa != VGS_OK) return sa;
  }
  vgs_status s = timed(c, VGS_T_LOCALCUT, [&] { return vgs_stage_localcut(c); }, false);   // the merge stage's kernels queue up behind it
  if (s != VGS_OK) return s;
  s = timed(c, VGS_T_MERGE, [&] { return vgs_stage_merge(c); });   // ends with a read-back behind everything
  if (s == VGS_OK) { c->stage = ST_SEGMENTED; vgs_charge_localcut_tail(c); }
  return s;
}

vgs_status vgs_run(vgs_ctx* c) {
  if (!c) return VGS_E_ARG;
  if (c->P.method == 3) {
    if (!c->sv_have_labels) {
      vgs_status s = svgs_supervoxels(c);
      if (s != VGS_OK) return s;
    }
    return svgs_segment(c);
  }
  vgs_status s;
  if ((s = do_voxelize(c, false)) != VGS_OK) return s;
  if ((s = do_features(c, false)) != VGS_OK) return s;
  if ((s = do_adjacency(c, false)) != VGS_OK) return s;
  if ((s = vgs_segment(c)) != VGS_OK) return s;
  c->times[VGS_T_TOTAL] = c->times[VGS_T_VOXELIZE] + c->times[VGS_T_FEATURES] + c->times[VGS_T_ADJACENCY] + c->times[VGS_T_LOCALCUT] +
                          c->times[VGS_T_MERGE];
  return VGS_OK;
}

// ---- SVGS -------------------------------------------------------
vgs_status svgs_set_supervoxel_labels(vgs_ctx* c, const int32_t* labels_host, int32_t max_label) {
  if (!c || (!labels_host && c->N > 0)) return VGS_E_ARG;
  if (c->P.method != 3) { c->err = "svgs_set_supervoxel_labels: context was created for method 2 (VGS)"; return VGS_E_STATE; }
  if (c->stage < ST_POINTS) { c->err = "svgs_set_supervoxel_labels: set the input cloud first"; return VGS_E_STATE; }
  VGS_HIP_TRY(c, hipSetDevice(c->device));
  VGS_HIP_TRY(c, c->sv_label.ensure(c->N > 0 ? c->N : 1));
  if (c->N > 0) VGS_HIP_TRY(c, hipMemcpy(c->sv_label.p, labels_host, (size_t)c->N * 4, hipMemcpyHostToDevice));
  c->sv_max_label = max_label;
  c->sv_have_labels = true; c->sv_labels_external = true; c->sv_label_n = c->N;
  if (c->stage > ST_POINTS) c->stage = ST_POINTS;
  return VGS_OK;
}

vgs_status svgs_supervoxels(vgs_ctx* c) {
  if (!c) return VGS_E_ARG;
  if (c->P.method != 3) { c->err = "svgs_supervoxels: context was created for method 2 (VGS)"; return VGS_E_STATE; }
  if (c->stage < ST_POINTS) { c->err = "svgs_supervoxels: set the input cloud first"; return VGS_E_STATE; }
  if (!(c->P.seed_size > c->P.voxel_size)) { c->err = "svgs_supervoxels: seed_size must exceed voxel_size"; return VGS_E_ARG; }
  c->sv_have_labels = false; c->sv_labels_external = false; c->sv_label_n = -1;
  vgs_status s = timed(c, VGS_T_SUPERVOXEL, [&] { return vgs_stage_vccs(c); });
  if (s == VGS_OK) { c->stage = ST_POINTS; c->sv_label_n = c->N; }
  return s;
}

vgs_status svgs_get_supervoxel_labels(vgs_ctx* c, int32_t* labels, int32_t* max_label) {
  if (!c || !labels || !max_label) return VGS_E_ARG;
  if (!c->sv_have_labels || c->sv_label_n != c->N) { c->err = "svgs_get_supervoxel_labels: no supervoxel labelling of this cloud yet"; return VGS_E_STATE; }
  VGS_HIP_TRY(c, hipSetDevice(c->device));
  if (c->N > 0) VGS_HIP_TRY(c, hipMemcpy(labels, c->sv_label.p, (size_t)c->N * 4, hipMemcpyDeviceToHost));
  *max_label = c->sv_max_label;
  return VGS_OK;
}

vgs_status svgs_segment(vgs_ctx* c) {
  if (!c) return VGS_E_ARG;
  if (c->P.method != 3) { c->err = "svgs_segment: context was created for method 2 (VGS)"; return VGS_E_STATE; }
  if (c->stage < ST_POINTS || !c->sv_have_labels) { c->err = "svgs_segment: needs the input cloud and supervoxel labels (createSupervoxels)"; return VGS_E_STATE; }
  if (c->sv_label_n != c->N) { c->err = "svgs_segment: the supervoxel labels were made for a different cloud"; return VGS_E_STATE; }
  vgs_status s;
  if ((s = timed(c, VGS_T_VOXELIZE, [&] { return vgs_stage_svgs_group(c); }, false)) != VGS_OK) return s;       // SS:279-331
  if ((s = timed(c, VGS_T_FEATURES, [&] { return vgs_stage_features(c); }, false)) != VGS_OK) return s;         // SS:1238-1303
  if ((s = timed(c, VGS_T_ADJACENCY, [&] { return vgs_stage_svgs_neighbours(c); }, false)) != VGS_OK) return s;  // SS:1477-1521
  if ((s = timed(c, VGS_T_LOCALCUT, [&] { return vgs_stage_localcut(c); }, false)) != VGS_OK) return s;          // SS:384-413
  if ((s = timed(c, VGS_T_MERGE, [&] { return vgs_stage_merge(c); })) != VGS_OK) return s;                // SS:416-420
  c->stage = ST_SEGMENTED; vgs_charge_localcut_tail(c);
  c->times[VGS_T_TOTAL] = c->times[VGS_T_VOXELIZE] + c->times[VGS_T_FEATURES] + c->times[VGS_T_ADJACENCY] + c->times[VGS_T_LOCALCUT] +
                          c->times[VGS_T_MERGE];
  return VGS_OK;
}

// ---- results ------------------------------------------------------------------------------
vgs_status vgs_get_counts(vgs_ctx* c, int64_t* counts) {
  if (!c || !counts) return VGS_E_ARG;
  if (c->stage >= ST_ADJACENCY && c->U > 0 && c->counts[VGS_N_ADJ] == 0) {
    std::vector<uint32_t> cnt((size_t)c->U);
    VGS_HIP_TRY(c, hipMemcpy(cnt.data(), c->adj_mused.p, cnt.size() * 4, hipMemcpyDeviceToHost));
    int64_t e = 0;
    for (uint32_t x : cnt) e += x;
    c->counts[VGS_N_ADJ] = e;
  }
  if (c->stage >= ST_SEGMENTED && c->U > 0 && c->counts[VGS_N_PAIRS] < 0) {
    std::vector<uint32_t> ev((size_t)c->U);
    VGS_HIP_TRY(c, hipMemcpy(ev.data(), c->evals.p, ev.size() * 4, hipMemcpyDeviceToHost));
    int64_t tot = 0;
    for (uint32_t x : ev) tot += x;
    c->counts[VGS_N_PAIRS] = tot;
  }
  for (int i = 0; i < VGS_N_COUNTS; ++i) counts[i] = c->counts[i];
  return VGS_OK;
}

vgs_status vgs_get_stage_times(vgs_ctx* c, double* ms) {
  if (!c || !ms) return VGS_E_ARG;
  { vgs_status sr = vgs_resolve_times(c); if (sr != VGS_OK) return sr; }
  if (c->P.method == 2 && c->stage >= ST_SEGMENTED)
    c->times[VGS_T_TOTAL] = c->times[VGS_T_VOXELIZE] + c->times[VGS_T_FEATURES] + c->times[VGS_T_ADJACENCY] + c->times[VGS_T_LOCALCUT] + c->times[VGS_T_MERGE];
  for (int i = 0; i < VGS_T_COUNT; ++i) ms[i] = c->times[i];
  return VGS_OK;
}

vgs_status vgs_get_schedule_counters(vgs_ctx* c, int64_t* out) { return vgs_get_schedule_counters_ex(c, out, 8); }

vgs_status vgs_get_schedule_counters_ex(vgs_ctx* c, int64_t* out, int32_t n) {
  if (!c || !out || n < 0 || n > 16) return VGS_E_ARG;
  for (int i = 0; i < n; ++i) out[i] = c->lc_diag[i];
  return VGS_OK;
}

vgs_status vgs_get_bbox(vgs_ctx* c, double* b) {
  if (!c || !b) return VGS_E_ARG;
  if (c->stage < ST_VOXELS) { c->err = "vgs_get_bbox: voxelize first"; return VGS_E_STATE; }
  for (int a = 0; a < 3; ++a) { b[a] = c->box.min[a]; b[3 + a] = c->box.max[a]; }
  return VGS_OK;
}

vgs_status vgs_get_voxel_table(vgs_ctx* c, uint32_t* key, int32_t* start, int32_t* point_idx) {
  if (!c) return VGS_E_ARG;
  if (c->stage < ST_VOXELS) { c->err = "vgs_get_voxel_table: voxelize first"; return VGS_E_STATE; }
  VGS_HIP_TRY(c, hipSetDevice(c->device));
  if (key && c->V > 0) {
    std::vector<uint64_t> code((size_t)c->V);
    VGS_HIP_TRY(c, hipMemcpy(code.data(), c->vox_code.p, code.size() * 8, hipMemcpyDeviceToHost));
    for (int64_t v = 0; v < c->V; ++v) {
      key[3 * v + 0] = vm_compact21(code[v] >> 2);
      key[3 * v + 1] = vm_compact21(code[v] >> 1);
      key[3 * v + 2] = vm_compact21(code[v]);
    }
  }
  if (start) {
    if (c->V > 0) VGS_HIP_TRY(c, hipMemcpy(start, c->vox_start.p, (size_t)(c->V + 1) * 4, hipMemcpyDeviceToHost));
    else start[0] = 0;
  }
  if (point_idx && c->Nf > 0) VGS_HIP_TRY(c, hipMemcpy(point_idx, c->perm_b.p, (size_t)c->Nf * 4, hipMemcpyDeviceToHost));
  return VGS_OK;
}

vgs_status vgs_get_voxel_centers(vgs_ctx* c, float* center) {
  if (!c || !center) return VGS_E_ARG;
  if (c->stage < ST_VOXELS) { c->err = "vgs_get_voxel_centers: voxelize first"; return VGS_E_STATE; }
  if (c->V == 0) return VGS_OK;
  std::vector<uint64_t> code((size_t)c->V);
  VGS_HIP_TRY(c, hipMemcpy(code.data(), c->vox_code.p, code.size() * 8, hipMemcpyDeviceToHost));
  const float res_f = c->P.voxel_size;
  const float mn[3] = {(float)c->box.min[0], (float)c->box.min[1], (float)c->box.min[2]};
  for (int64_t v = 0; v < c->V; ++v) {
    center[3 * v + 0] = vm_voxel_center(vm_compact21(code[v] >> 2), res_f, mn[0]);
    center[3 * v + 1] = vm_voxel_center(vm_compact21(code[v] >> 1), res_f, mn[1]);
    center[3 * v + 2] = vm_voxel_center(vm_compact21(code[v]), res_f, mn[2]);
  }
  return VGS_OK;
}

vgs_status vgs_get_point_voxel(vgs_ctx* c, int32_t* out) {
  if (!c || !out) return VGS_E_ARG;
  if (c->stage < ST_VOXELS) { c->err = "vgs_get_point_voxel: voxelize first"; return VGS_E_STATE; }
  if (c->N == 0) return VGS_OK;
  std::vector<uint32_t> perm((size_t)c->N), pv((size_t)c->N);
  VGS_HIP_TRY(c, hipMemcpy(perm.data(), c->perm_b.p, perm.size() * 4, hipMemcpyDeviceToHost));
  VGS_HIP_TRY(c, hipMemcpy(pv.data(), c->pt_vox.p, pv.size() * 4, hipMemcpyDeviceToHost));
  for (int64_t j = 0; j < c->N; ++j) out[perm[j]] = (pv[j] == 0xffffffffu) ? -1 : (int32_t)pv[j];
  return VGS_OK;
}

vgs_status vgs_get_attributes(vgs_ctx* c, float* centroid, float* normal, float* eigen8, uint8_t* used) {
  if (!c) return VGS_E_ARG;
  if (c->stage < ST_FEATURES) { c->err = "vgs_get_attributes: features first"; return VGS_E_STATE; }
  if (c->V == 0) return VGS_OK;
  std::vector<NodeRec> nd((size_t)c->V);
  VGS_HIP_TRY(c, hipMemcpy(nd.data(), c->node.p, nd.size() * sizeof(NodeRec), hipMemcpyDeviceToHost));
  for (int64_t v = 0; v < c->V; ++v) {
    for (int a = 0; a < 3; ++a) {
      if (centroid) centroid[3 * v + a] = nd[v].c[a];
      if (normal) normal[3 * v + a] = nd[v].n[a];
    }
    if (eigen8) for (int a = 0; a < 8; ++a) eigen8[8 * v + a] = nd[v].f[a];
    if (used) used[v] = (nd[v].flags & VGS_F_EIG) ? 1 : 0;
  }
  return VGS_OK;
}

}  // extern "C"

// the per-node lists of vgs_get_lists as host vectors (also the input of the reference-order cluster walk)
// ordered = true (which >= 1): every list in the reference's own element order, the merge history of the local cut
// (cutorder.hip); false: the members in the order of the adjacency row
static vgs_status vgs_build_lists(vgs_ctx* c, int32_t which, std::vector<std::vector<int32_t>>& L, bool ordered = false) {
  const int64_t V = c->V, U = c->U;
  if (which == 0 && c->P.method == 2 && V > 0) {
    // findAllVoxelAdjacency builds a list for EVERY voxel, used or not, with every neighbour (VS:236-263).  The hot path
    // keeps rows for the used voxels only (and only their used neighbours when unused ones are inert), so the full lists
    // are built here on request: the general kernel over all voxel ids, a chunk of rows at a time.
    VGS_HIP_TRY(c, hipSetDevice(c->device));
    L.assign((size_t)V, {});
    const int64_t chunk = std::min<int64_t>(V, 65536);
    DevBuf<uint64_t> fk; DevBuf<uint32_t> fc, fn, fid;
    VGS_HIP_TRY(c, fk.ensure((size_t)chunk * c->adj_stride)); VGS_HIP_TRY(c, fc.ensure(chunk)); VGS_HIP_TRY(c, fn.ensure(chunk)); VGS_HIP_TRY(c, fid.ensure(chunk));
    std::vector<uint32_t> ids((size_t)chunk), cnt((size_t)chunk);
    std::vector<uint64_t> keys((size_t)chunk * c->adj_stride);
    vgs_status st = VGS_OK;
    for (int64_t v0 = 0; v0 < V && st == VGS_OK; v0 += chunk) {
      const int64_t m = std::min(chunk, V - v0);
      for (int64_t k = 0; k < m; ++k) ids[(size_t)k] = (uint32_t)(v0 + k);
      if (hipMemcpy(fid.p, ids.data(), (size_t)m * 4, hipMemcpyHostToDevice) != hipSuccess) { st = VGS_E_HIP; break; }
      st = vgs_run_adjacency(c, true, fk.p, fc.p, fn.p, c->adj_r2, fid.p, m);
      if (st == VGS_OK && hipStreamSynchronize(c->stream) != hipSuccess) st = VGS_E_HIP;
      if (st == VGS_OK && (hipMemcpy(cnt.data(), fc.p, (size_t)m * 4, hipMemcpyDeviceToHost) != hipSuccess ||
                           hipMemcpy(keys.data(), fk.p, (size_t)m * c->adj_stride * 8, hipMemcpyDeviceToHost) != hipSuccess)) st = VGS_E_HIP;
      if (st != VGS_OK) break;
      for (int64_t k = 0; k < m; ++k) {
        std::vector<int32_t>& l = L[(size_t)(v0 + k)];
        l.resize(cnt[(size_t)k]);
        for (uint32_t q = 0; q < cnt[(size_t)k]; ++q) l[q] = (int32_t)(uint32_t)keys[(size_t)k * c->adj_stride + q];
      }
    }
    fk.release(); fc.release(); fn.release(); fid.release();
    if (st == VGS_E_HIP) c->err = "vgs_get_lists: full adjacency pass failed";
    return st;
  }
  std::vector<uint32_t> used_ids((size_t)U), cnt((size_t)U);
  std::vector<uint64_t> keys;
  std::vector<uint8_t> flag;
  std::vector<int32_t> attach;
  if (ordered && which >= 1 && U > 0) {
    // the reference's own element order: the lists are put together on the device (cutorder.hip: merge-history order, crossValidation's
    // filter in that order) and come down compact -- ids only
    VGS_HIP_TRY(c, hipSetDevice(c->device));
    VGS_HIP_TRY(c, hipMemcpy(used_ids.data(), c->used_ids.p, (size_t)U * 4, hipMemcpyDeviceToHost));
    std::vector<uint16_t> ord;
    std::vector<uint32_t> kk, lcnt;
    std::vector<int32_t> lids;
    const uint8_t* lflag = which == 1 ? (const uint8_t*)nullptr : c->conn.p + (size_t)U * c->adj_stride;
    vgs_status so = vgs_cut_order(c, ord, kk, true, lflag, &lcnt, &lids);
    if (so != VGS_OK) return so;
    L.assign((size_t)V, {});
    size_t at = 0;
    for (int64_t u = 0; u < U; ++u) {
      L[used_ids[(size_t)u]].assign(lids.begin() + (ptrdiff_t)at, lids.begin() + (ptrdiff_t)(at + lcnt[(size_t)u]));
      at += lcnt[(size_t)u];
    }
    if (which == 3) {
      attach.resize((size_t)V);
      VGS_HIP_TRY(c, hipMemcpy(attach.data(), c->attach.p, (size_t)V * 4, hipMemcpyDeviceToHost));
      for (int64_t i = 0; i < V; ++i)   // closestCheck appends (VS:2293-2294): i gets its target, the target gets i, in voxel order
        if (attach[i] >= 0) { L[i].push_back(attach[i]); L[attach[i]].push_back((int32_t)i); }
    }
    return VGS_OK;
  }
  if (U > 0) {
    VGS_HIP_TRY(c, hipSetDevice(c->device));
    VGS_HIP_TRY(c, hipMemcpy(used_ids.data(), c->used_ids.p, (size_t)U * 4, hipMemcpyDeviceToHost));
    keys.resize((size_t)U * c->adj_stride);
    VGS_HIP_TRY(c, hipMemcpy(cnt.data(), c->adj_cnt.p, (size_t)U * 4, hipMemcpyDeviceToHost));
    VGS_HIP_TRY(c, hipMemcpy(keys.data(), c->adj_key.p, keys.size() * 8, hipMemcpyDeviceToHost));
    if (which >= 1) {
      flag.resize((size_t)U * c->adj_stride);
      const uint8_t* src = c->conn.p + (which >= 2 ? (size_t)U * c->adj_stride : 0);
      VGS_HIP_TRY(c, hipMemcpy(flag.data(), src, flag.size(), hipMemcpyDeviceToHost));
    }
    if (which == 3) {
      attach.resize((size_t)V);
      VGS_HIP_TRY(c, hipMemcpy(attach.data(), c->attach.p, (size_t)V * 4, hipMemcpyDeviceToHost));
    }
  }
  // per-voxel lists; closestCheck appends (VS:2293-2294): i gets its target, the target gets i, in voxel order
  L.assign((size_t)V, {});
  for (int64_t u = 0; u < U; ++u) {
    const uint32_t i = used_ids[u];
    for (uint32_t k = 0; k < cnt[u]; ++k) {
      const size_t s = (size_t)u * c->adj_stride + k;
      if (which == 0 || flag[s]) L[i].push_back((int32_t)(uint32_t)keys[s]);
    }
  }
  if (which == 3 && !attach.empty())   // no used voxel: nothing was cut, nothing re-attached
    for (int64_t i = 0; i < V; ++i)
      if (attach[i] >= 0) { L[i].push_back(attach[i]); L[attach[i]].push_back((int32_t)i); }
  return VGS_OK;
}

extern "C" {

vgs_status vgs_get_lists(vgs_ctx* c, int32_t which, int64_t* offsets, int32_t* idx) {
  return vgs_get_lists_ordered(c, which, VGS_ORDER_VOXEL_ID, offsets, idx);
}

vgs_status vgs_get_lists_ordered(vgs_ctx* c, int32_t which, int32_t order, int64_t* offsets, int32_t* idx) {
  if (!c || !offsets || which < 0 || which > 3 || (order != VGS_ORDER_VOXEL_ID && order != VGS_ORDER_REFERENCE)) return VGS_E_ARG;
  const int need = which == 0 ? ST_ADJACENCY : ST_SEGMENTED;
  if (c->stage < need) { c->err = "vgs_get_lists: stage not reached"; return VGS_E_STATE; }
  const int64_t V = c->V;
  std::vector<std::vector<int32_t>> L;
  vgs_status sb = vgs_build_lists(c, which, L, order == VGS_ORDER_REFERENCE);
  if (sb != VGS_OK) return sb;
  int64_t o = 0;
  for (int64_t v = 0; v < V; ++v) {
    offsets[v] = o;
    if (idx) std::memcpy(idx + o, L[v].data(), L[v].size() * 4);
    o += (int64_t)L[v].size();
  }
  offsets[V] = o;
  return VGS_OK;
}

vgs_status vgs_get_adjacency_counts(vgs_ctx* c, int32_t* n_all) {
  if (!c || !n_all) return VGS_E_ARG;
  if (c->stage < ST_ADJACENCY) { c->err = "vgs_get_adjacency_counts: adjacency first"; return VGS_E_STATE; }
  for (int64_t v = 0; v < c->V; ++v) n_all[v] = 0;
  if (c->U == 0) return VGS_OK;
  VGS_HIP_TRY(c, hipSetDevice(c->device));
  std::vector<uint32_t> ids((size_t)c->U), cnt((size_t)c->U);
  VGS_HIP_TRY(c, hipMemcpy(ids.data(), c->used_ids.p, ids.size() * 4, hipMemcpyDeviceToHost));
  VGS_HIP_TRY(c, hipMemcpy(cnt.data(), c->adj_mused.p, cnt.size() * 4, hipMemcpyDeviceToHost));
  for (int64_t u = 0; u < c->U; ++u) n_all[ids[u]] = (int32_t)cnt[u];
  return VGS_OK;
}

vgs_status vgs_get_node_labels(vgs_ctx* c, int32_t* root, int32_t* kept_label) {
  if (!c) return VGS_E_ARG;
  if (c->stage < ST_SEGMENTED) { c->err = "vgs_get_node_labels: segment first"; return VGS_E_STATE; }
  if (c->V == 0) return VGS_OK;
  if (root) VGS_HIP_TRY(c, hipMemcpy(root, c->parent.p, (size_t)c->V * 4, hipMemcpyDeviceToHost));
  if (kept_label) VGS_HIP_TRY(c, hipMemcpy(kept_label, c->vox_label.p, (size_t)c->V * 4, hipMemcpyDeviceToHost));
  return VGS_OK;
}

vgs_status vgs_get_point_labels(vgs_ctx* c, int32_t* labels) {
  if (!c || !labels) return VGS_E_ARG;
  if (c->stage < ST_SEGMENTED) { c->err = "vgs_get_point_labels: segment first"; return VGS_E_STATE; }
  { vgs_status se = vgs_ensure_point_labels(c); if (se != VGS_OK) return se; }
  if (c->N > 0) VGS_HIP_TRY(c, hipMemcpy(labels, c->pt_label.p, (size_t)c->N * 4, hipMemcpyDeviceToHost));
  return VGS_OK;
}

vgs_status vgs_get_point_labels_async(vgs_ctx* c, int32_t* labels) {
  if (!c || !labels) return VGS_E_ARG;
  if (c->stage < ST_SEGMENTED) { c->err = "vgs_get_point_labels_async: segment first"; return VGS_E_STATE; }
  VGS_HIP_TRY(c, hipSetDevice(c->device));
  if (c->d2h_open) { VGS_HIP_TRY(c, hipEventSynchronize(c->ev_d2h)); c->d2h_open = false; }   // one download in flight per context
  if (c->N == 0) return VGS_OK;
  { vgs_status se = vgs_ensure_point_labels(c); if (se != VGS_OK) return se; }
  // device-side order: the copy follows the kernel that wrote the labels (ev[15], recorded behind it on the context's stream by the
  // merge stage and by vgs_apply_*_labels), whatever the host has or has not waited for
  if (c->labels_event_valid) VGS_HIP_TRY(c, hipStreamWaitEvent(c->s_d2h, c->ev[15], 0));
  VGS_HIP_TRY(c, hipMemcpyAsync(labels, c->pt_label.p, (size_t)c->N * 4, hipMemcpyDeviceToHost, c->s_d2h));
  VGS_HIP_TRY(c, hipEventRecord(c->ev_d2h, c->s_d2h));
  c->d2h_open = true; c->d2h_src = c->pt_label.p;
  return VGS_OK;
}

vgs_status vgs_wait_point_labels(vgs_ctx* c) {
  if (!c) return VGS_E_ARG;
  if (c->d2h_open) { VGS_HIP_TRY(c, hipSetDevice(c->device)); VGS_HIP_TRY(c, hipEventSynchronize(c->ev_d2h)); c->d2h_open = false; }
  return VGS_OK;
}

vgs_status vgs_get_point_labels_device(vgs_ctx* c, const int32_t** labels_dev) {
  if (!c || !labels_dev) return VGS_E_ARG;
  if (c->stage < ST_SEGMENTED) { c->err = "vgs_get_point_labels_device: segment first"; return VGS_E_STATE; }
  { vgs_status se = vgs_ensure_point_labels(c); if (se != VGS_OK) return se; }
  *labels_dev = c->pt_label.p;
  return VGS_OK;
}

vgs_status vgs_get_clusters_ordered(vgs_ctx* c, int32_t order, int64_t* offsets, int32_t* point_idx) {
  if (!c || !offsets || (order != VGS_ORDER_VOXEL_ID && order != VGS_ORDER_REFERENCE)) return VGS_E_ARG;
  if (c->stage < ST_SEGMENTED) { c->err = "vgs_get_clusters: segment first (drawColorMapofPointsinClusters precedes getClusterIdx, VS:1006)"; return VGS_E_STATE; }
  if (order == VGS_ORDER_REFERENCE && c->have_region) { c->err = "vgs_get_clusters: reference order needs the whole cloud in one context (not a tile)"; return VGS_E_STATE; }
  VGS_HIP_TRY(c, hipSetDevice(c->device));
  const int64_t K = c->counts[VGS_N_KEPT];
  if (order == VGS_ORDER_VOXEL_ID) {
    // the lists are made on the device (clusters.hip: one stable sort of the leaf order by label); only the answer comes down
    vgs_status sd = vgs_clusters_on_device(c);
    if (sd != VGS_OK) return sd;
    VGS_HIP_TRY(c, hipMemcpy(offsets, c->cl_off.p, ((size_t)K + 1) * sizeof(int64_t), hipMemcpyDeviceToHost));
    if (point_idx && offsets[K] > 0) VGS_HIP_TRY(c, hipMemcpy(point_idx, c->cl_idx.p, (size_t)offsets[K] * sizeof(int32_t), hipMemcpyDeviceToHost));
    return VGS_OK;
  }
  std::vector<int64_t> cnt((size_t)K + 1, 0);
  std::vector<uint32_t> perm((size_t)c->Nf), pv((size_t)c->Nf);
  std::vector<int32_t> vl((size_t)c->V);
  if (c->Nf > 0) {
    VGS_HIP_TRY(c, hipMemcpy(perm.data(), c->perm_b.p, perm.size() * 4, hipMemcpyDeviceToHost));
    VGS_HIP_TRY(c, hipMemcpy(pv.data(), c->pt_vox.p, pv.size() * 4, hipMemcpyDeviceToHost));
    VGS_HIP_TRY(c, hipMemcpy(vl.data(), c->vox_label.p, vl.size() * 4, hipMemcpyDeviceToHost));
  }
  for (int64_t j = 0; j < c->Nf; ++j) { const int32_t l = vl[pv[j]]; if (l >= 0) cnt[l + 1]++; }
  for (int64_t k = 0; k < K; ++k) cnt[k + 1] += cnt[k];
  for (int64_t k = 0; k <= K; ++k) offsets[k] = cnt[k];
  if (!point_idx) return VGS_OK;
  std::vector<int64_t> cur(cnt.begin(), cnt.end() - 1);
  if (order == VGS_ORDER_VOXEL_ID) {
    // sorted positions run over voxels in ascending id and points in ascending index inside a voxel
    for (int64_t j = 0; j < c->Nf; ++j) { const int32_t l = vl[pv[j]]; if (l >= 0) point_idx[cur[l]++] = (int32_t)perm[j]; }
    return VGS_OK;
  }
  // Reference order (output formatting on the host, not the hot path): clusteringVoxels scans the nodes in ascending id and
  // starts recursionSearch at every unclustered one -- a pre-order walk over the final connect lists in list order -- and
  // appends the seed LAST (VS:2032-2053, 2064-2080; SS:2079-2103); a cluster's points are its nodes' point lists in that
  // node order, each in ascending point index (VS:981-999, SS:2109-2126).  The kept clusters keep the order of their seeds,
  // which is the order of the labels (ascending smallest node id).
  std::vector<std::vector<int32_t>> L;
  vgs_status sb = vgs_build_lists(c, 3, L, true);
  if (sb != VGS_OK) return sb;
  const int64_t V = c->V;
  std::vector<uint32_t> vstart((size_t)V + 1, 0);
  if (V > 0) VGS_HIP_TRY(c, hipMemcpy(vstart.data(), c->vox_start.p, ((size_t)V + 1) * 4, hipMemcpyDeviceToHost));
  std::vector<uint8_t> clustered((size_t)V, 0);
  std::vector<std::pair<int32_t, size_t>> stack;   // (node whose list is being scanned, position): the recursion, unrolled
  std::vector<int32_t> members;
  for (int64_t i = 0; i < V; ++i) {
    if (clustered[i]) continue;
    clustered[i] = 1;
    members.clear();
    stack.clear();
    stack.emplace_back((int32_t)i, 0);
    while (!stack.empty()) {
      auto& top = stack.back();
      const std::vector<int32_t>& lst = L[top.first];
      if (top.second >= lst.size()) { stack.pop_back(); continue; }
      const int32_t v = lst[top.second++];
      if (!clustered[v]) { members.push_back(v); clustered[v] = 1; stack.emplace_back(v, 0); }
    }
    members.push_back((int32_t)i);
    const int32_t l = vl[i];
    if (l < 0) continue;   // dropped by the size filter (VS:969)
    for (int32_t v : members)
      for (uint32_t j = vstart[v]; j < vstart[v + 1]; ++j) point_idx[cur[l]++] = (int32_t)perm[j];
  }
  return VGS_OK;
}

vgs_status vgs_get_clusters(vgs_ctx* c, int64_t* offsets, int32_t* point_idx) {
  return vgs_get_clusters_ordered(c, VGS_ORDER_VOXEL_ID, offsets, point_idx);
}

// ---- multi-GPU ---------------------------------------------------------------

}  // extern "C"
