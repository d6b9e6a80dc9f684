// tiles.cpp -- libvgs_tiles.so: the native tiled driver of include/vgs_tiles.h.  Host C++ over the C-ABI of
// include/vgs.h; the collectives are RCCL calls (ncclAllGather / ncclBroadcast on the caller's communicator) or, for tests
// that run several ranks on one GPU, threads of one process meeting in shared memory.  The Python twin of this file is
// vgs-svgs-segmentation_amd/dist.py (same protocol, same labels); protocol notes live there and in DESIGN.md section 6.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <numeric>
#include <string>
#include <vector>

#include "../../include/vgs_tiles.h"

namespace {

// ------------------------------------------------------------------------------------------------ communicators
struct Comm {
  int rank = 0, world = 1;
  std::string err;
  virtual ~Comm() {}
  virtual bool all_gather(const void* send, void* recv, size_t bytes_per_rank) = 0;   // recv: world * bytes_per_rank, rank order
  virtual bool bcast(void* buf, size_t bytes, int root) = 0;
};

// threads of one process (tests: several ranks on one GPU)
struct LocalGroup {
  int world;
  std::mutex m;
  std::condition_variable cv;
  int waiting = 0;
  uint64_t generation = 0;
  bool broken = false;
  std::vector<const void*> slot;
  explicit LocalGroup(int w) : world(w), slot((size_t)w, nullptr) {}
  bool barrier() {
    std::unique_lock<std::mutex> lk(m);
    if (broken) return false;
    const uint64_t g = generation;
    if (++waiting == world) { waiting = 0; ++generation; cv.notify_all(); return true; }
    cv.wait(lk, [&] { return generation != g || broken; });
    return !broken;
  }
  void abort() { std::lock_guard<std::mutex> lk(m); broken = true; cv.notify_all(); }
};

struct LocalComm : Comm {
  LocalGroup* g;
  LocalComm(LocalGroup* grp, int r) : g(grp) { rank = r; world = grp->world; }
  bool all_gather(const void* send, void* recv, size_t bytes) override {
    g->slot[(size_t)rank] = send;
    if (!g->barrier()) { err = "local group aborted"; return false; }
    for (int r = 0; r < world; ++r) std::memcpy((char*)recv + (size_t)r * bytes, g->slot[(size_t)r], bytes);
    if (!g->barrier()) { err = "local group aborted"; return false; }
    return true;
  }
  bool bcast(void* buf, size_t bytes, int root) override {
    if (rank == root) g->slot[(size_t)root] = buf;
    if (!g->barrier()) { err = "local group aborted"; return false; }
    if (rank != root) std::memcpy(buf, g->slot[(size_t)root], bytes);
    if (!g->barrier()) { err = "local group aborted"; return false; }
    return true;
  }
};

// RCCL: payloads are staged through device buffers (the collectives move device memory over xGMI)
struct RcclComm : Comm {
  ncclComm_t comm;
  int device;
  hipStream_t stream = nullptr;
  char *d_send = nullptr, *d_recv = nullptr;
  size_t cap_send = 0, cap_recv = 0;
  RcclComm(ncclComm_t c, int r, int w, int dev) : comm(c), device(dev) { rank = r; world = w; }
  ~RcclComm() override {
    (void)hipSetDevice(device);
    if (d_send) (void)hipFree(d_send);
    if (d_recv) (void)hipFree(d_recv);
    if (stream) (void)hipStreamDestroy(stream);
  }
  bool hip_ok(hipError_t e, const char* what) { if (e != hipSuccess) { err = std::string(what) + ": " + hipGetErrorString(e); return false; } return true; }
  bool nccl_ok(ncclResult_t e, const char* what) { if (e != ncclSuccess) { err = std::string(what) + ": " + ncclGetErrorString(e); return false; } return true; }
  bool ensure(size_t send_bytes, size_t recv_bytes) {
    if (!hip_ok(hipSetDevice(device), "hipSetDevice")) return false;
    if (!stream && !hip_ok(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking), "hipStreamCreate")) return false;
    if (send_bytes > cap_send) { if (d_send) (void)hipFree(d_send); d_send = nullptr; if (!hip_ok(hipMalloc((void**)&d_send, send_bytes + 256), "hipMalloc")) return false; cap_send = send_bytes + 256; }
    if (recv_bytes > cap_recv) { if (d_recv) (void)hipFree(d_recv); d_recv = nullptr; if (!hip_ok(hipMalloc((void**)&d_recv, recv_bytes + 256), "hipMalloc")) return false; cap_recv = recv_bytes + 256; }
    return true;
  }
  bool all_gather(const void* send, void* recv, size_t bytes) override {
    if (!ensure(bytes, bytes * (size_t)world)) return false;
    return hip_ok(hipMemcpyAsync(d_send, send, bytes, hipMemcpyHostToDevice, stream), "H2D") &&
           nccl_ok(ncclAllGather(d_send, d_recv, bytes, ncclChar, comm, stream), "ncclAllGather") &&
           hip_ok(hipMemcpyAsync(recv, d_recv, bytes * (size_t)world, hipMemcpyDeviceToHost, stream), "D2H") &&
           hip_ok(hipStreamSynchronize(stream), "hipStreamSynchronize");
  }
  bool bcast(void* buf, size_t bytes, int root) override {
    if (!ensure(bytes, bytes)) return false;
    return hip_ok(hipMemcpyAsync(d_send, buf, bytes, hipMemcpyHostToDevice, stream), "H2D") &&
           nccl_ok(ncclBroadcast(d_send, d_recv, bytes, ncclChar, root, comm, stream), "ncclBroadcast") &&
           hip_ok(hipMemcpyAsync(buf, d_recv, bytes, hipMemcpyDeviceToHost, stream), "D2H") &&
           hip_ok(hipStreamSynchronize(stream), "hipStreamSynchronize");
  }
};

// the caller's own transport (MPI, gloo, a test harness): two C callbacks over host buffers
struct CallbackComm : Comm {
  vgs_tiles_callbacks cb;
  CallbackComm(const vgs_tiles_callbacks& c, int r, int w) : cb(c) { rank = r; world = w; }
  bool all_gather(const void* send, void* recv, size_t bytes) override {
    if (cb.all_gather(cb.user, send, recv, (uint64_t)bytes) != 0) { err = "caller's all_gather callback failed"; return false; }
    return true;
  }
  bool bcast(void* buf, size_t bytes, int root) override {
    if (cb.bcast(cb.user, buf, (uint64_t)bytes, root) != 0) { err = "caller's bcast callback failed"; return false; }
    return true;
  }
};

// arrays of different lengths: one size exchange + one padded payload exchange
template <typename T>
bool all_gather_varlen(Comm& c, const std::vector<T>& mine, std::vector<std::vector<T>>& out) {
  std::vector<int64_t> sizes((size_t)c.world);
  const int64_t n = (int64_t)mine.size();
  if (!c.all_gather(&n, sizes.data(), sizeof(int64_t))) return false;
  const int64_t m = std::max<int64_t>(*std::max_element(sizes.begin(), sizes.end()), 1);
  std::vector<T> pad((size_t)m, T()), all((size_t)m * c.world);
  std::copy(mine.begin(), mine.end(), pad.begin());
  if (!c.all_gather(pad.data(), all.data(), (size_t)m * sizeof(T))) return false;
  out.assign((size_t)c.world, {});
  for (int r = 0; r < c.world; ++r) out[(size_t)r].assign(all.begin() + (size_t)r * m, all.begin() + (size_t)r * m + sizes[(size_t)r]);
  return true;
}

// ------------------------------------------------------------------------------------------------ boundary merge
struct RankRecords { std::vector<uint64_t> code; std::vector<int32_t> root, cnt; };

// Global segments from the compact per-rank results (vgs_get_boundary_roots): the same tables on every rank from the same
// gathered inputs.  base[r] + rank-in-root-order labels the components that touch no boundary voxel (on the GPU), the
// components named by records get dense labels behind them, in the order of their first (rank, root) node.
void merge_boundary_compact(const std::vector<RankRecords>& rec, const std::vector<int64_t>& kept_local, int voxels_min,
                            std::vector<int64_t>& base, std::vector<std::vector<int32_t>>& uroot, std::vector<std::vector<int32_t>>& ulabel,
                            int64_t& kept_total) {
  const int world = (int)rec.size();
  base.assign((size_t)world, 0);
  for (int r = 1; r < world; ++r) base[(size_t)r] = base[(size_t)r - 1] + kept_local[(size_t)r - 1];
  int64_t next_label = 0;
  for (int r = 0; r < world; ++r) next_label += kept_local[(size_t)r];
  uroot.assign((size_t)world, {}); ulabel.assign((size_t)world, {});
  std::vector<int64_t> node_cnt, offset((size_t)world + 1, 0);
  std::vector<std::pair<uint64_t, int64_t>> rec_code_node;   // (code, node) of every record
  for (int r = 0; r < world; ++r) {
    const RankRecords& R = rec[(size_t)r];
    std::vector<int32_t>& ur = uroot[(size_t)r];
    ur = R.root;
    std::sort(ur.begin(), ur.end());
    ur.erase(std::unique(ur.begin(), ur.end()), ur.end());
    std::vector<int64_t> cnt_of(ur.size(), -1);
    for (size_t k = 0; k < R.root.size(); ++k) {
      const size_t j = (size_t)(std::lower_bound(ur.begin(), ur.end(), R.root[k]) - ur.begin());
      if (cnt_of[j] < 0) cnt_of[j] = R.cnt[k];   // first record of that root (every record of a root carries the same count)
      rec_code_node.emplace_back(R.code[k], offset[(size_t)r] + (int64_t)j);
    }
    for (int64_t x : cnt_of) node_cnt.push_back(x < 0 ? 0 : x);
    offset[(size_t)r + 1] = offset[(size_t)r] + (int64_t)ur.size();
  }
  const int64_t n = offset[(size_t)world];
  kept_total = next_label;
  if (n == 0) return;
  std::vector<int64_t> parent((size_t)n);
  std::iota(parent.begin(), parent.end(), 0);
  auto find = [&](int64_t x) { while (parent[(size_t)x] != x) { parent[(size_t)x] = parent[(size_t)parent[(size_t)x]]; x = parent[(size_t)x]; } return x; };
  std::stable_sort(rec_code_node.begin(), rec_code_node.end(), [](const auto& a, const auto& b) { return a.first < b.first; });
  for (size_t k = 1; k < rec_code_node.size(); ++k)
    if (rec_code_node[k].first == rec_code_node[k - 1].first) {   // records of all ranks that share a code name the same segment
      int64_t a = find(rec_code_node[k - 1].second), b = find(rec_code_node[k].second);
      if (a != b) { if (a < b) parent[(size_t)b] = a; else parent[(size_t)a] = b; }   // the smaller node index is the root = the component's first node
    }
  std::vector<int64_t> total((size_t)n, 0);
  for (int64_t x = 0; x < n; ++x) total[(size_t)find(x)] += node_cnt[(size_t)x];
  std::vector<int64_t> label_of((size_t)n, -1);
  for (int64_t x = 0; x < n; ++x)   // roots in ascending node order = components in the order of their first node
    if (parent[(size_t)x] == x && total[(size_t)x] > voxels_min) label_of[(size_t)x] = next_label++;
  for (int r = 0; r < world; ++r) {
    const size_t m = uroot[(size_t)r].size();
    ulabel[(size_t)r].resize(m);
    for (size_t j = 0; j < m; ++j) ulabel[(size_t)r][j] = (int32_t)label_of[(size_t)find(offset[(size_t)r] + (int64_t)j)];
  }
  kept_total = next_label;
}

}  // namespace

// ------------------------------------------------------------------------------------------------ driver
struct vgs_tiles {
  vgs_params P;
  Comm* comm = nullptr;
  vgs_ctx* ctx = nullptr;
  int tiles_x = 1, tiles_y = 1;
  double pitch = 0, cx = 0, cy = 0;
  double lo[2] = {0, 0}, hi[2] = {0, 0};
  std::vector<float> local;   // tile + halo, packed xyz
  int64_t own_first = 0, n_own = 0, n_outside = 0, n_records = 0, kept = 0;
  int64_t exch_sent = 0, exch_recv = 0; int exch_calls = 0;   // the last run's boundary exchange: bytes this rank sent / received, collectives it took
  double times[VGS_TILES_T_COUNT] = {0};   // last run, milliseconds of host wall time per phase (vgs_tiles_get_times)
  int strict_region = 0;      // VGS_TILES_OPT_STRICT_REGION
  int fail_phase = 0;         // tests (VGS_TILES_FAIL_RANK / VGS_TILES_FAIL_AT): 1 grid, 2 stages, 3 points, 4 upload (behind the last collective of set_points)
  vgs_status pending = VGS_OK;   // a local failure behind a call's last collective: the status word of the next collective carries it
  bool warned_outside = false;
  std::string err;
};

static vgs_status tfail(vgs_tiles* t, vgs_status s, const std::string& msg) { t->err = msg; return s; }
#define TCTX(call) do { vgs_status s_ = (call); if (s_ != VGS_OK) return tfail(t, s_, std::string(#call) + ": " + vgs_last_error_string(t->ctx)); } while (0)
#define TCOMM(call) do { if (!(call)) return tfail(t, VGS_E_HIP, std::string("collective failed: ") + t->comm->err); } while (0)
// A LOCAL failure between two collectives must not make this rank leave while its peers are already inside the next collective
// (they would wait for ever: ADVICE r3).  So a failing step only records its status (`carry`), the rank still takes part in the
// next collective with that status as a word of the payload, and every rank sees it there and returns: the failing rank its own
// error, the others VGS_E_PEER naming the rank.  The caller's process then exits non-zero and the launcher ends the job.
#define TCARRY(call) do { if (carry == VGS_OK) { vgs_status s_ = (call); if (s_ != VGS_OK) { carry = s_; t->err = std::string(#call) + ": " + vgs_last_error_string(t->ctx); } } } while (0)

static vgs_status agreed(vgs_tiles* t, vgs_status mine, int first_bad_rank, const char* phase) {
  if (mine != VGS_OK) return mine;   // t->err already says what failed here
  if (first_bad_rank >= 0) return tfail(t, VGS_E_PEER, std::string("rank ") + std::to_string(first_bad_rank) + " failed in the " + phase + " phase; this rank stops with it");
  return VGS_OK;
}

static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

extern "C" {

vgs_status vgs_tiles_local_group_create(int world, void** group) {
  if (world < 1 || !group) return VGS_E_ARG;
  *group = new LocalGroup(world);
  return VGS_OK;
}
void vgs_tiles_local_group_destroy(void* group) { delete (LocalGroup*)group; }
void vgs_tiles_local_group_abort(void* group) { if (group) ((LocalGroup*)group)->abort(); }

vgs_status vgs_tiles_create(const vgs_params* p, int comm_kind, void* comm_handle, int rank, int world, int tiles_x, int tiles_y, double pitch,
                            double center_x, double center_y, vgs_tiles** out) {
  if (!p || !out || !comm_handle || world < 1 || rank < 0 || rank >= world || tiles_x < 1 || tiles_y < 1 || tiles_x * tiles_y != world) return VGS_E_ARG;
  if (p->method != 2) return VGS_E_UNSUPPORTED;   // the tiled path is VGS (BASELINE.json configs[4])
  *out = nullptr;
  vgs_tiles* t = new vgs_tiles();
  t->P = *p;
  t->tiles_x = tiles_x; t->tiles_y = tiles_y; t->pitch = pitch; t->cx = center_x; t->cy = center_y;
  if (comm_kind == VGS_TILES_COMM_LOCAL) t->comm = new LocalComm((LocalGroup*)comm_handle, rank);
  else if (comm_kind == VGS_TILES_COMM_RCCL) t->comm = new RcclComm((ncclComm_t)comm_handle, rank, world, p->device);
  else if (comm_kind == VGS_TILES_COMM_CALLBACKS) {
    const vgs_tiles_callbacks* cb = (const vgs_tiles_callbacks*)comm_handle;
    if (!cb->all_gather || !cb->bcast) { delete t; return VGS_E_ARG; }
    t->comm = new CallbackComm(*cb, rank, world);
  } else { delete t; return VGS_E_ARG; }
  {
    // failure injection for the tests of the agreed-status protocol; read once, here
    const char* fr = std::getenv("VGS_TILES_FAIL_RANK");
    const char* fa = std::getenv("VGS_TILES_FAIL_AT");
    if (fr && fa && std::atoi(fr) == rank) t->fail_phase = !std::strcmp(fa, "grid") ? 1 : !std::strcmp(fa, "stages") ? 2 : !std::strcmp(fa, "points") ? 3 : !std::strcmp(fa, "upload") ? 4 : 0;
  }
  vgs_status s = vgs_create(p, &t->ctx);
  if (s != VGS_OK) { delete t->comm; delete t; return s; }
  *out = t;
  return VGS_OK;
}

void vgs_tiles_destroy(vgs_tiles* t) {
  if (!t) return;
  if (t->ctx) vgs_destroy(t->ctx);
  delete t->comm;
  delete t;
}

const char* vgs_tiles_last_error_string(const vgs_tiles* t) { return t ? t->err.c_str() : "null handle"; }
vgs_ctx* vgs_tiles_context(vgs_tiles* t) { return t ? t->ctx : nullptr; }

vgs_status vgs_tiles_set_option(vgs_tiles* t, int32_t option, int64_t value) {
  if (!t) return VGS_E_ARG;
  if (option == VGS_TILES_OPT_STRICT_REGION) { t->strict_region = value != 0; return VGS_OK; }
  return tfail(t, VGS_E_ARG, "unknown option");
}

vgs_status vgs_tiles_get_times(vgs_tiles* t, double* ms, int32_t n) {
  if (!t || !ms || n < 0 || n > VGS_TILES_T_COUNT) return VGS_E_ARG;
  for (int i = 0; i < n; ++i) ms[i] = t->times[i];
  return VGS_OK;
}

vgs_status vgs_tiles_set_points(vgs_tiles* t, const float* xyz, int64_t n, int32_t stride_bytes) {
  if (!t || (!xyz && n > 0) || n < 0 || (stride_bytes != 12 && stride_bytes != 16)) return VGS_E_ARG;
  Comm& c = *t->comm;
  const int sf = stride_bytes / 4;
  if (!(t->pitch > 0)) {   // the largest x-extent over the ranks
    float mn = 3.0e38f, mx = -3.0e38f;
    for (int64_t i = 0; i < n; ++i) { const float x = xyz[i * sf]; if (x == x) { mn = std::min(mn, x); mx = std::max(mx, x); } }
    double ext = n > 0 ? (double)(mx - mn) : 0.0;
    std::vector<double> all((size_t)c.world);
    TCOMM(c.all_gather(&ext, all.data(), sizeof(double)));
    t->pitch = *std::max_element(all.begin(), all.end());
  }
  // ownership rectangle [lo, hi): outer edges are open ended
  const double big = 1.0e30;
  const int i = c.rank % t->tiles_x, j = c.rank / t->tiles_x;
  const double x0 = t->cx + (i - t->tiles_x / 2.0) * t->pitch, y0 = t->cy + (j - t->tiles_y / 2.0) * t->pitch;
  t->lo[0] = i > 0 ? x0 : -big; t->lo[1] = j > 0 ? y0 : -big;
  t->hi[0] = i < t->tiles_x - 1 ? x0 + t->pitch : big; t->hi[1] = j < t->tiles_y - 1 ? y0 + t->pitch : big;
  const double h = 2.0 * (double)t->P.graph_size + (double)t->P.voxel_size;
  // border strips of this rank's cloud, all-gathered; every rank keeps what falls into its own rectangle widened by the halo
  std::vector<float> strip;
  t->n_outside = 0;
  for (int64_t k = 0; k < n; ++k) {
    const double x = (double)xyz[k * sf], y = (double)xyz[k * sf + 1];
    if (x < t->lo[0] || x >= t->hi[0] || y < t->lo[1] || y >= t->hi[1]) ++t->n_outside;
    if (x < t->lo[0] + h || x >= t->hi[0] - h || y < t->lo[1] + h || y >= t->hi[1] - h) { strip.push_back(xyz[k * sf]); strip.push_back(xyz[k * sf + 1]); strip.push_back(xyz[k * sf + 2]); }
  }
  // Points a rank holds beyond its own region may come back unlabelled (their voxels are owned, and cut, elsewhere; found by
  // tools/fuzz_tiles.py): a caller with an arbitrary partition is told so once -- or, with VGS_TILES_OPT_STRICT_REGION, every
  // rank refuses the cloud (the count travels in the strips' size exchange below, so the ranks agree on it)
  if (t->n_outside > 0 && !t->strict_region && !t->warned_outside) {
    t->warned_outside = true;
    std::fprintf(stderr, "[vgs_tiles] rank %d: %lld of %lld points lie outside this rank's region; they may come back unlabelled (-1). "
                         "Load points by region, or set VGS_TILES_OPT_STRICT_REGION to make this an error.\n", c.rank, (long long)t->n_outside, (long long)n);
  }
  // a failure behind the last collective of the previous call (the label write-back at the end of vgs_tiles_run) has reached nobody
  // yet: it travels in this call's status word (it used to be cleared below and never arrived)
  vgs_status carry = t->pending;
  if (t->fail_phase == 3 && carry == VGS_OK) { carry = VGS_E_STATE; t->err = "failure requested by VGS_TILES_FAIL_RANK / VGS_TILES_FAIL_AT=points"; }
  if (t->strict_region && t->n_outside > 0 && carry == VGS_OK) {
    carry = VGS_E_ARG;
    t->err = std::to_string(t->n_outside) + " points lie outside this rank's region (VGS_TILES_OPT_STRICT_REGION)";
  }
  {
    // agreed status before the payload: one word per rank
    int64_t st = (int64_t)carry;
    std::vector<int64_t> all((size_t)c.world);
    TCOMM(c.all_gather(&st, all.data(), sizeof(int64_t)));
    int bad = -1;
    for (int r = 0; r < c.world && bad < 0; ++r) if (all[(size_t)r] != 0) bad = r;
    vgs_status a = agreed(t, carry, bad, "point loading");
    if (a != VGS_OK) return a;
  }
  std::vector<std::vector<float>> strips;
  TCOMM(all_gather_varlen(c, strip, strips));
  // the local cloud in RANK ORDER: strips of lower ranks, own points, strips of higher ranks -- the order in which one process
  // would have inserted the points (a voxel's attributes depend on the order of its points; see dist.py)
  t->local.clear();
  auto take_strip = [&](int r) {
    const std::vector<float>& s = strips[(size_t)r];
    for (size_t k = 0; k + 2 < s.size(); k += 3) {
      const double x = (double)s[k], y = (double)s[k + 1];
      if (x >= t->lo[0] - h && x < t->hi[0] + h && y >= t->lo[1] - h && y < t->hi[1] + h) { t->local.push_back(s[k]); t->local.push_back(s[k + 1]); t->local.push_back(s[k + 2]); }
    }
  };
  for (int r = 0; r < c.rank; ++r) take_strip(r);
  t->own_first = (int64_t)(t->local.size() / 3);
  for (int64_t k = 0; k < n; ++k) { t->local.push_back(xyz[k * sf]); t->local.push_back(xyz[k * sf + 1]); t->local.push_back(xyz[k * sf + 2]); }
  for (int r = c.rank + 1; r < c.world; ++r) take_strip(r);
  t->n_own = n;
  // failures from here on are local and no collective follows inside this call: the rank returns its error AND keeps it
  // (t->pending), so that a caller who goes on to vgs_tiles_run all the same joins the grid's collective with that status in its
  // word instead of leaving its peers inside it
  t->pending = VGS_OK;
  auto local_step = [&](vgs_status s, const char* what) {
    if (s != VGS_OK && t->pending == VGS_OK) { t->pending = s; t->err = std::string(what) + ": " + vgs_last_error_string(t->ctx); }
  };
  if (t->fail_phase == 4) { t->pending = VGS_E_STATE; t->err = "failure requested by VGS_TILES_FAIL_RANK / VGS_TILES_FAIL_AT=upload"; }
  if (t->pending == VGS_OK) local_step(vgs_set_points(t->ctx, t->local.data(), (int64_t)(t->local.size() / 3), 12), "vgs_set_points");
  if (t->pending == VGS_OK) local_step(vgs_set_owned_region(t->ctx, t->lo, t->hi), "vgs_set_owned_region");
  if (t->pending == VGS_OK) local_step(vgs_set_own_point_range(t->ctx, t->own_first, t->n_own), "vgs_set_own_point_range");
  return t->pending;
}

// The shared grid: what inserting the ranks' clouds one after the other does to the octree box.  One all-gather
// of the clouds' bounding boxes lets every rank replay the growth on the host wherever the box alone decides it
// (vgs_grid_advance_bbox); only a rank whose cloud leaves the step open scans its points on the GPU and broadcasts the state --
// except rank 0, whose box starts at its first point: it scans first and its state travels with its bounding box.
// `carry`: see TCARRY -- a local failure travels as a status word of the next collective.
static vgs_status chain_grid(vgs_tiles* t, vgs_status& carry) {
  Comm& c = *t->comm;
  float bb[6] = {0, 0, 0, 0, 0, 0};
  int64_t nf = 0;
  if (t->fail_phase == 1) { carry = VGS_E_STATE; t->err = "failure requested by VGS_TILES_FAIL_RANK / VGS_TILES_FAIL_AT=grid"; }
  TCARRY(vgs_points_bbox(t->ctx, bb, &nf));
  auto pack = [](const vgs_grid_state& g, double* v) { for (int a = 0; a < 3; ++a) { v[a] = g.min[a]; v[3 + a] = (double)g.shift[a]; } v[6] = g.depth; v[7] = g.defined; };
  auto unpack = [](vgs_grid_state& g, const double* v) { for (int a = 0; a < 3; ++a) { g.min[a] = v[a]; g.shift[a] = (uint64_t)v[3 + a]; } g.depth = (int32_t)v[6]; g.defined = (int32_t)v[7]; };
  vgs_grid_state g0;
  vgs_grid_state_init(&g0);
  if (c.rank == 0 && nf > 0) TCARRY(vgs_grid_advance(t->ctx, &g0));
  double mine[16];
  for (int a = 0; a < 6; ++a) mine[a] = (double)bb[a];
  mine[6] = (double)nf;
  pack(g0, mine + 7);
  mine[15] = (double)carry;   // status word
  std::vector<double> all((size_t)16 * c.world);
  TCOMM(c.all_gather(mine, all.data(), sizeof(mine)));
  {
    int bad = -1;
    for (int r = 0; r < c.world && bad < 0; ++r) if (all[(size_t)16 * r + 15] != 0.0) bad = r;
    vgs_status a = agreed(t, carry, bad, "grid");
    if (a != VGS_OK) return a;
  }
  vgs_grid_state g;
  vgs_grid_state_init(&g);
  for (int r = 0; r < c.world; ++r) {
    const double* v = all.data() + (size_t)16 * r;
    if (v[6] == 0) continue;   // no finite point: the cloud changes nothing
    if (r == 0) { unpack(g, v + 7); continue; }
    float box[6];
    for (int a = 0; a < 6; ++a) box[a] = (float)v[a];   // float values, exactly as gathered
    int32_t need = 0;
    vgs_status s = vgs_grid_advance_bbox(&g, (double)t->P.voxel_size, box, &need);   // host arithmetic on gathered values: fails on every rank alike
    if (s != VGS_OK) return tfail(t, s, "vgs_grid_advance_bbox");
    if (!need) continue;
    double buf[9] = {0};
    if (c.rank == r) { TCARRY(vgs_grid_advance(t->ctx, &g)); pack(g, buf); buf[8] = (double)carry; }
    TCOMM(c.bcast(buf, sizeof(buf), r));
    if (buf[8] != 0.0) return agreed(t, carry, r, "grid");
    unpack(g, buf);
  }
  TCARRY(vgs_set_grid_covering(t->ctx, &g));   // (every rank's cloud went into g, scanned or by its bounding box) a failure here travels with the boundary records
  return VGS_OK;
}

vgs_status vgs_tiles_run(vgs_tiles* t) {
  if (!t) return VGS_E_ARG;
  Comm& c = *t->comm;
  vgs_status carry = t->pending;   // a local failure behind the last collective of the previous call (upload, label write-back) travels now
  double t0 = now_ms();
  vgs_status s = chain_grid(t, carry);
  if (s != VGS_OK) return s;
  double t1 = now_ms();
  t->times[VGS_TILES_T_GRID] = t1 - t0;
  if (t->fail_phase == 2 && carry == VGS_OK) { carry = VGS_E_STATE; t->err = "failure requested by VGS_TILES_FAIL_RANK / VGS_TILES_FAIL_AT=stages"; }
  TCARRY(vgs_voxelize(t->ctx)); TCARRY(vgs_features(t->ctx)); TCARRY(vgs_adjacency(t->ctx)); TCARRY(vgs_segment(t->ctx));
  double t2 = now_ms();
  t->times[VGS_TILES_T_STAGES] = t2 - t1;
  // only the border leaves the GPU: unique boundary voxels (code, local root, owned voxels of that root) and the number of
  // components that are local to this tile
  int64_t n = 0, nkl = 0;
  TCARRY(vgs_get_boundary_roots(t->ctx, &n, nullptr, nullptr, nullptr, &nkl));
  if (carry != VGS_OK) { n = 0; nkl = 0; }
  RankRecords mine;
  mine.code.resize((size_t)std::max<int64_t>(n, 1)); mine.root.resize(mine.code.size()); mine.cnt.resize(mine.code.size());
  if (n > 0) TCARRY(vgs_get_boundary_roots(t->ctx, &n, mine.code.data(), mine.root.data(), mine.cnt.data(), &nkl));
  if (carry != VGS_OK) { n = 0; nkl = 0; }
  mine.code.resize((size_t)n); mine.root.resize((size_t)n); mine.cnt.resize((size_t)n);
  t->n_records = n;
  double t3 = now_ms();
  t->times[VGS_TILES_T_RECORDS] = t3 - t2;
  // the one data-path exchange: header (record count, local segment count, this rank's status) + records, 3 words each
  constexpr size_t HDR = 3;
  std::vector<int64_t> payload(HDR + 3 * (size_t)n);
  payload[0] = n; payload[1] = nkl; payload[2] = (int64_t)carry;
  for (int64_t k = 0; k < n; ++k) { payload[HDR + (size_t)k] = (int64_t)mine.code[(size_t)k]; payload[HDR + (size_t)(n + k)] = mine.root[(size_t)k]; payload[HDR + (size_t)(2 * n + k)] = mine.cnt[(size_t)k]; }
  // one collective of a fixed size when every rank's payload fits; the gathered headers tell every rank alike when not
  const size_t cap = 3 * 8192 + HDR;
  std::vector<std::vector<int64_t>> gathered((size_t)c.world);
  {
    std::vector<int64_t> fixed(cap, 0), all(cap * (size_t)c.world);
    std::copy(payload.begin(), payload.begin() + (ptrdiff_t)std::min(payload.size(), cap), fixed.begin());
    TCOMM(c.all_gather(fixed.data(), all.data(), cap * sizeof(int64_t)));
    {
      int bad = -1;
      for (int r = 0; r < c.world && bad < 0; ++r) if (all[cap * (size_t)r + 2] != 0) bad = r;
      vgs_status a = agreed(t, carry, bad, "stages");
      if (a != VGS_OK) return a;
    }
    bool fits = true;
    for (int r = 0; r < c.world; ++r) fits = fits && (HDR + 3 * (size_t)all[cap * (size_t)r]) <= cap;
    t->exch_sent = (int64_t)(cap * sizeof(int64_t)); t->exch_recv = t->exch_sent * c.world; t->exch_calls = 1;
    if (fits) for (int r = 0; r < c.world; ++r) gathered[(size_t)r].assign(all.begin() + (ptrdiff_t)(cap * (size_t)r), all.begin() + (ptrdiff_t)(cap * (size_t)r + HDR + 3 * (size_t)all[cap * (size_t)r]));
    else {
      // (a rank with more than 8192 boundary voxels -- a 10 M-point tile has 10^4 to 10^5: the sizes are known from the headers, but the
      // helper's own size word keeps the code path one; the payload is padded to the largest rank's)
      int64_t mx = 0;
      for (int r = 0; r < c.world; ++r) mx = std::max<int64_t>(mx, (int64_t)(HDR + 3 * (size_t)all[cap * (size_t)r]));
      TCOMM(all_gather_varlen(c, payload, gathered));
      t->exch_sent += (int64_t)sizeof(int64_t) * (1 + mx); t->exch_recv += (int64_t)sizeof(int64_t) * (1 + mx) * c.world; t->exch_calls += 2;
    }
  }
  double t4 = now_ms();
  t->times[VGS_TILES_T_EXCHANGE] = t4 - t3;
  std::vector<RankRecords> rec((size_t)c.world);
  std::vector<int64_t> kept_local((size_t)c.world);
  for (int r = 0; r < c.world; ++r) {
    const std::vector<int64_t>& gr = gathered[(size_t)r];
    const int64_t m = gr[0];
    kept_local[(size_t)r] = gr[1];
    RankRecords& R = rec[(size_t)r];
    R.code.resize((size_t)m); R.root.resize((size_t)m); R.cnt.resize((size_t)m);
    for (int64_t k = 0; k < m; ++k) { R.code[(size_t)k] = (uint64_t)gr[HDR + (size_t)k]; R.root[(size_t)k] = (int32_t)gr[HDR + (size_t)(m + k)]; R.cnt[(size_t)k] = (int32_t)gr[HDR + (size_t)(2 * m + k)]; }
  }
  std::vector<int64_t> base;
  std::vector<std::vector<int32_t>> uroot, ulabel;
  merge_boundary_compact(rec, kept_local, t->P.voxels_min, base, uroot, ulabel, t->kept);
  double t5 = now_ms();
  t->times[VGS_TILES_T_MERGE] = t5 - t4;
  std::vector<int32_t>& br = uroot[(size_t)c.rank];
  std::vector<int32_t>& bl = ulabel[(size_t)c.rank];
  int32_t dummy = 0;
  // (a failure here is local and the run's last collective is behind us: this rank returns it and keeps it in t->pending; the
  // peers -- whose results are complete -- see it in the status word of the next run's first collective, or at the launcher)
  {
    vgs_status sa = vgs_apply_tile_labels(t->ctx, (int32_t)base[(size_t)c.rank], br.empty() ? &dummy : br.data(), bl.empty() ? &dummy : bl.data(), (int64_t)br.size());
    if (sa != VGS_OK) { t->pending = sa; return tfail(t, sa, std::string("vgs_apply_tile_labels: ") + vgs_last_error_string(t->ctx)); }
  }
  double t6 = now_ms();
  t->times[VGS_TILES_T_LABELS] = t6 - t5;
  t->times[VGS_TILES_T_TOTAL] = t6 - t0;
  return VGS_OK;
}

vgs_status vgs_tiles_get_point_labels(vgs_tiles* t, int32_t* labels, int64_t* kept_global) {
  if (!t || (!labels && t->n_own > 0)) return VGS_E_ARG;
  std::vector<int32_t> all(t->local.size() / 3 + 1);
  TCTX(vgs_get_point_labels(t->ctx, all.data()));
  std::copy(all.begin() + (ptrdiff_t)t->own_first, all.begin() + (ptrdiff_t)(t->own_first + t->n_own), labels);   // halo points belong to other ranks
  if (kept_global) *kept_global = t->kept;
  return VGS_OK;
}

// host arithmetic only (tests): the boundary merge on flattened per-rank records
vgs_status vgs_tiles_merge_boundary(int world, const int64_t* rec_off, const uint64_t* code, const int32_t* root, const int32_t* cnt,
                                    const int64_t* kept_local, int voxels_min, int64_t* base, int64_t* uoff, int32_t* uroot, int32_t* ulabel,
                                    int64_t* kept_total) {
  if (world < 1 || !rec_off || !kept_local || !base || !uoff || !kept_total) return VGS_E_ARG;
  std::vector<RankRecords> rec((size_t)world);
  std::vector<int64_t> kl(kept_local, kept_local + world);
  for (int r = 0; r < world; ++r) {
    rec[(size_t)r].code.assign(code + rec_off[r], code + rec_off[r + 1]);
    rec[(size_t)r].root.assign(root + rec_off[r], root + rec_off[r + 1]);
    rec[(size_t)r].cnt.assign(cnt + rec_off[r], cnt + rec_off[r + 1]);
  }
  std::vector<int64_t> b;
  std::vector<std::vector<int32_t>> ur, ul;
  merge_boundary_compact(rec, kl, voxels_min, b, ur, ul, *kept_total);
  uoff[0] = 0;
  for (int r = 0; r < world; ++r) {
    base[r] = b[(size_t)r];
    uoff[r + 1] = uoff[r] + (int64_t)ur[(size_t)r].size();
    if (uroot) std::copy(ur[(size_t)r].begin(), ur[(size_t)r].end(), uroot + uoff[r]);
    if (ulabel) std::copy(ul[(size_t)r].begin(), ul[(size_t)r].end(), ulabel + uoff[r]);
  }
  return VGS_OK;
}

vgs_status vgs_tiles_get_exchange(vgs_tiles* t, int64_t* bytes_sent, int64_t* bytes_received, int32_t* collectives) {
  if (!t) return VGS_E_ARG;
  if (bytes_sent) *bytes_sent = t->exch_sent;
  if (bytes_received) *bytes_received = t->exch_recv;
  if (collectives) *collectives = t->exch_calls;
  return VGS_OK;
}

vgs_status vgs_tiles_get_info(vgs_tiles* t, int64_t* n_outside, int64_t* n_local, int64_t* n_boundary_records) {
  if (!t) return VGS_E_ARG;
  if (n_outside) *n_outside = t->n_outside;
  if (n_local) *n_local = (int64_t)(t->local.size() / 3);
  if (n_boundary_records) *n_boundary_records = t->n_records;
  return VGS_OK;
}

}  // extern "C"
