// examples/vgs_tiles_run.cpp -- the native tiled driver (include/vgs_tiles.h) from the command line: one rank per tile.
//
//   RCCL, one process per GPU (started by any launcher that sets RANK, WORLD_SIZE, LOCAL_RANK, MASTER_ADDR, MASTER_PORT --
//   torchrun's variables): rank 0 creates the ncclUniqueId and hands it to the others over a TCP socket on MASTER_PORT + 1,
//   every rank calls ncclCommInitRank and passes the communicator to vgs_tiles_create:
//       vgs_tiles_run --rccl 4x2 --pitch 50 --voxel 0.1 <prefix>
//   Emulation on one GPU (tests): the ranks are threads of this process and meet in shared memory:
//       vgs_tiles_run --emulate 2x2 --pitch 6.1 --voxel 0.1 <prefix>
//   Rank r reads its points from <prefix>.<r>.f32 (packed float32 xyz) and writes one int32 label per point to
//   <prefix>.<r>.labels.i32.  Prints "<world> <kept segments> <points of rank 0> <boundary records of rank 0>".
#include <arpa/inet.h>
#include <netinet/in.h>
#include <sys/socket.h>
#include <unistd.h>

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "vgs_tiles.h"

static bool read_f32(const std::string& path, std::vector<float>& out) {
  FILE* f = std::fopen(path.c_str(), "rb");
  if (!f) return false;
  std::fseek(f, 0, SEEK_END);
  const long bytes = std::ftell(f);
  std::fseek(f, 0, SEEK_SET);
  out.resize((size_t)bytes / 4);
  const size_t got = std::fread(out.data(), 4, out.size(), f);
  std::fclose(f);
  return got == out.size();
}

static bool write_i32(const std::string& path, const std::vector<int32_t>& v) {
  FILE* f = std::fopen(path.c_str(), "wb");
  if (!f) return false;
  const size_t put = std::fwrite(v.data(), 4, v.size(), f);
  std::fclose(f);
  return put == v.size();
}

struct Job { vgs_params p; int tx, ty; double pitch; std::string prefix; };

// one rank: load, run, save; returns 0 on success
static int run_rank(const Job& J, int comm_kind, void* comm, int rank, int world, int64_t* kept, int64_t* n_pts, int64_t* n_rec, std::string* err) {
  std::vector<float> xyz;
  if (!read_f32(J.prefix + "." + std::to_string(rank) + ".f32", xyz)) { *err = "cannot read the points of rank " + std::to_string(rank); return 1; }
  vgs_tiles* t = nullptr;
  vgs_status s = vgs_tiles_create(&J.p, comm_kind, comm, rank, world, J.tx, J.ty, J.pitch, 0.0, 0.0, &t);
  if (s != VGS_OK) { *err = std::string("vgs_tiles_create: ") + vgs_last_error_string(nullptr); return 1; }
  const int64_t n = (int64_t)(xyz.size() / 3);
  std::vector<int32_t> labels((size_t)n + 1);
  int rc = 0;
  if ((s = vgs_tiles_set_points(t, xyz.data(), n, 12)) != VGS_OK || (s = vgs_tiles_run(t)) != VGS_OK ||
      (s = vgs_tiles_get_point_labels(t, labels.data(), kept)) != VGS_OK) {
    *err = std::string("rank ") + std::to_string(rank) + ": " + vgs_tiles_last_error_string(t);
    rc = 1;
  }
  if (rc == 0) {
    labels.resize((size_t)n);
    if (!write_i32(J.prefix + "." + std::to_string(rank) + ".labels.i32", labels)) { *err = "cannot write the labels"; rc = 1; }
    *n_pts = n;
    vgs_tiles_get_info(t, nullptr, nullptr, n_rec);
  }
  vgs_tiles_destroy(t);
  return rc;
}

// ncclUniqueId from rank 0 to everyone over TCP (the launcher gives MASTER_ADDR / MASTER_PORT; port + 1 is used here)
static bool exchange_id(ncclUniqueId* id, int rank, int world, const char* addr, int port) {
  if (world == 1) return true;
  if (rank == 0) {
    const int srv = socket(AF_INET, SOCK_STREAM, 0);
    int one = 1;
    setsockopt(srv, SOL_SOCKET, SO_REUSEADDR, &one, sizeof(one));
    sockaddr_in a{};
    a.sin_family = AF_INET; a.sin_addr.s_addr = htonl(INADDR_ANY); a.sin_port = htons((uint16_t)port);
    if (bind(srv, (sockaddr*)&a, sizeof(a)) != 0 || listen(srv, world) != 0) { close(srv); return false; }
    for (int k = 1; k < world; ++k) {
      const int fd = accept(srv, nullptr, nullptr);
      if (fd < 0) { close(srv); return false; }
      const bool ok = write(fd, id, sizeof(*id)) == (ssize_t)sizeof(*id);
      close(fd);
      if (!ok) { close(srv); return false; }
    }
    close(srv);
    return true;
  }
  sockaddr_in a{};
  a.sin_family = AF_INET; a.sin_port = htons((uint16_t)port);
  if (inet_pton(AF_INET, addr, &a.sin_addr) != 1) return false;
  for (int attempt = 0; attempt < 600; ++attempt) {   // rank 0 may not be listening yet
    const int fd = socket(AF_INET, SOCK_STREAM, 0);
    if (connect(fd, (sockaddr*)&a, sizeof(a)) == 0) {
      size_t got = 0;
      while (got < sizeof(*id)) { const ssize_t r = read(fd, (char*)id + got, sizeof(*id) - got); if (r <= 0) break; got += (size_t)r; }
      close(fd);
      return got == sizeof(*id);
    }
    close(fd);
    usleep(100000);
  }
  return false;
}

int main(int argc, char** argv) {
  Job J;
  vgs_params_default_vgs(&J.p);
  J.p.voxel_size = 0.1f;
  J.tx = 1; J.ty = 1; J.pitch = 0.0;
  int mode = -1;   // 0 rccl, 1 emulate
  for (int a = 1; a < argc; ++a) {
    if ((!std::strcmp(argv[a], "--rccl") || !std::strcmp(argv[a], "--emulate")) && a + 1 < argc) {
      mode = !std::strcmp(argv[a], "--emulate") ? 1 : 0;
      if (std::sscanf(argv[++a], "%dx%d", &J.tx, &J.ty) != 2) { std::fprintf(stderr, "layout must be <tiles_x>x<tiles_y>\n"); return 2; }
    } else if (!std::strcmp(argv[a], "--pitch") && a + 1 < argc) J.pitch = std::atof(argv[++a]);
    else if (!std::strcmp(argv[a], "--voxel") && a + 1 < argc) J.p.voxel_size = (float)std::atof(argv[++a]);
    else if (!std::strcmp(argv[a], "--graph") && a + 1 < argc) J.p.graph_size = (float)std::atof(argv[++a]);
    else if (argv[a][0] != '-') J.prefix = argv[a];
    else { std::fprintf(stderr, "unknown argument %s\n", argv[a]); return 2; }
  }
  if (mode < 0 || J.prefix.empty()) { std::fprintf(stderr, "usage: %s (--rccl|--emulate) <tx>x<ty> [--pitch m] [--voxel m] [--graph m] <prefix>\n", argv[0]); return 2; }
  const int world = J.tx * J.ty;
  int64_t kept = 0, n_pts = 0, n_rec = 0;
  if (mode == 1) {
    void* group = nullptr;
    vgs_tiles_local_group_create(world, &group);
    std::vector<std::thread> th;
    std::vector<int> rc((size_t)world, 0);
    std::vector<std::string> errs((size_t)world);
    std::vector<int64_t> k((size_t)world), np((size_t)world), nr((size_t)world);
    for (int r = 0; r < world; ++r)
      th.emplace_back([&, r] {
        rc[(size_t)r] = run_rank(J, VGS_TILES_COMM_LOCAL, group, r, world, &k[(size_t)r], &np[(size_t)r], &nr[(size_t)r], &errs[(size_t)r]);
        if (rc[(size_t)r]) vgs_tiles_local_group_abort(group);   // the other ranks must not wait for this one
      });
    for (auto& t : th) t.join();
    vgs_tiles_local_group_destroy(group);
    for (int r = 0; r < world; ++r) if (rc[(size_t)r]) { std::fprintf(stderr, "error: %s\n", errs[(size_t)r].c_str()); return 1; }
    for (int r = 1; r < world; ++r) if (k[(size_t)r] != k[0]) { std::fprintf(stderr, "error: ranks disagree on the number of segments\n"); return 1; }
    kept = k[0]; n_pts = np[0]; n_rec = nr[0];
  } else {
    const char* e_rank = std::getenv("RANK"); const char* e_world = std::getenv("WORLD_SIZE"); const char* e_local = std::getenv("LOCAL_RANK");
    const char* e_addr = std::getenv("MASTER_ADDR"); const char* e_port = std::getenv("MASTER_PORT");
    const int rank = e_rank ? std::atoi(e_rank) : 0;
    if ((e_world ? std::atoi(e_world) : 1) != world) { std::fprintf(stderr, "WORLD_SIZE does not match the %dx%d layout\n", J.tx, J.ty); return 2; }
    J.p.device = e_local ? std::atoi(e_local) : 0;
    if (hipSetDevice(J.p.device) != hipSuccess) { std::fprintf(stderr, "hipSetDevice(%d) failed\n", J.p.device); return 1; }
    ncclUniqueId id;
    if (rank == 0 && ncclGetUniqueId(&id) != ncclSuccess) { std::fprintf(stderr, "ncclGetUniqueId failed\n"); return 1; }
    if (!exchange_id(&id, rank, world, e_addr ? e_addr : "127.0.0.1", (e_port ? std::atoi(e_port) : 29500) + 1)) { std::fprintf(stderr, "rank %d: cannot exchange the ncclUniqueId\n", rank); return 1; }
    ncclComm_t comm;
    if (ncclCommInitRank(&comm, world, id, rank) != ncclSuccess) { std::fprintf(stderr, "rank %d: ncclCommInitRank failed\n", rank); return 1; }
    std::string err;
    const int rc = run_rank(J, VGS_TILES_COMM_RCCL, (void*)comm, rank, world, &kept, &n_pts, &n_rec, &err);
    ncclCommDestroy(comm);
    if (rc) { std::fprintf(stderr, "error: %s\n", err.c_str()); return 1; }
    if (rank != 0) return 0;
  }
  std::printf("%d %ld %ld %ld\n", world, (long)kept, (long)n_pts, (long)n_rec);
  return 0;
}
