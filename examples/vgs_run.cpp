// examples/vgs_run.cpp -- the task-file front end the reference implies: read a task file
// (Task_File_VGS.txt / Task_File_SVGS.txt layout), dispatch on its "Method" entry (line 24: 2 = VGS, 3 = SVGS), load
// the input PCD, segment, write the coloured clusters as PCD -- what `main` around the reference's `test` drivers does
// with input_vector[12]/[15] (input path / name) and [18]/[21] (output path / name), minus the viewer.
//   usage: vgs_run <task file> [--in <file.pcd|.ply>] [--out <file.pcd>] [--seed <n>] [--ascii] [--debug-meshes <prefix>]
// --debug-meshes (VGS only) also writes the reference's voxel drawings as <prefix>_voxels.ply, _clustered_voxels.ply, _normals.ply.
// --in / --out replace the path + name entries of the task file (the shipped ones hold Windows paths).
// Prints "<method> <points> <voxels> <supervoxels> <all clusters> <kept clusters> <labelled points>".
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "drivers.hpp"
#include "point_clouds_io.hpp"

int main(int argc, char** argv) {
  if (argc < 2) { std::fprintf(stderr, "usage: %s <task file> [--in file.pcd] [--out file.pcd] [--seed n] [--ascii]\n", argv[0]); return 2; }
  std::string in_file, out_file, debug_prefix;
  uint64_t seed = 0;
  bool ascii = false;
  for (int a = 2; a < argc; ++a) {
    if (!std::strcmp(argv[a], "--in") && a + 1 < argc) in_file = argv[++a];
    else if (!std::strcmp(argv[a], "--out") && a + 1 < argc) out_file = argv[++a];
    else if (!std::strcmp(argv[a], "--seed") && a + 1 < argc) seed = std::strtoull(argv[++a], nullptr, 10);
    else if (!std::strcmp(argv[a], "--ascii")) ascii = true;
    else if (!std::strcmp(argv[a], "--debug-meshes") && a + 1 < argc) debug_prefix = argv[++a];
    else { std::fprintf(stderr, "unknown argument %s\n", argv[a]); return 2; }
  }
  const std::vector<std::string> task = inputTaskTxtFile(argv[1]);
  if (task.size() < 51) { std::fprintf(stderr, "%s: not a task file (%zu lines)\n", argv[1], task.size()); return 2; }
  const int method = std::atoi(task[24].c_str());
  if (method != 2 && method != 3) { std::fprintf(stderr, "%s: method %d is neither 2 (VGS) nor 3 (SVGS)\n", argv[1], method); return 2; }
  if (method == 3 && task.size() < 61) { std::fprintf(stderr, "%s: SVGS task files have 61 lines\n", argv[1]); return 2; }
  if (in_file.empty()) in_file = task[12] + task[15];
  if (out_file.empty()) {
    std::string name = task[21];
    if (name.size() >= 4) name.replace(name.size() - 4, 4, ".pcd");  // test:78 / test:163
    out_file = task[18] + name;
  }
  PCXYZPtr cloud(new PCXYZ);
  const bool ply = in_file.size() > 4 && in_file.substr(in_file.size() - 4) == ".ply";
  if ((ply ? inputPointCloudData2(in_file, cloud) : inputPointCloudData(in_file, cloud)) != 0) return 1;
  std::vector<std::vector<int>> clusters;
  DriverSummary sum;
  try {
    if (method == 2) { if (segmentationVGS(cloud, task, clusters, &sum, debug_prefix) != 0) { std::fprintf(stderr, "cannot write the debug meshes\n"); return 1; } }
    else segmentationSVGS(cloud, task, clusters, &sum);
  } catch (const std::exception& e) {
    std::fprintf(stderr, "error: %s\n", e.what());
    return 1;
  }
  if (saveColoredClusters(out_file, cloud, clusters, seed, !ascii) != 0) return 1;
  std::printf("%d %ld %ld %ld %ld %ld %ld\n", method, sum.points, sum.voxels, sum.supervoxels, sum.clusters, sum.kept, sum.labelled);
  return 0;
}
