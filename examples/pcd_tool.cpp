// examples/pcd_tool.cpp -- host-only exerciser of include/point_clouds_io.hpp (no GPU, no libvgs_hip.so):
//   pcd_tool convert <in.pcd|in.ply> <out.pcd> [ascii|binary] read any supported PCD or PLY, write x y z
//   pcd_tool colour <in.pcd> <clusters.txt> <out.pcd> <seed> clusters.txt: one cluster per line, point indices
//   pcd_tool task <task file>                                 print "<lines> <method> <input name> <output name>"
//   pcd_tool scene <TOWN|PC1M|URB10M|URB80M> <points|0> <out.pcd> a synthetic BASELINE scene (include/vgs_scenes.hpp), binary PCD
#include <cstdio>
#include <fstream>
#include <sstream>

#include "point_clouds_io.hpp"
#include "vgs_scenes.hpp"

int main(int argc, char** argv) {
  if (argc < 3) return 2;
  const std::string cmd = argv[1];
  if (cmd == "convert" && argc >= 4) {
    PCXYZPtr c(new PCXYZ);
    const std::string in = argv[2];
    const bool ply = in.size() > 4 && in.substr(in.size() - 4) == ".ply";
    if ((ply ? inputPointCloudData2(in, c) : inputPointCloudData(in, c)) != 0) return 1;
    const bool binary = argc > 4 && std::string(argv[4]) == "binary";
    return outputPointCloudData(argv[3], c, binary) == 0 ? 0 : 1;
  }
  if (cmd == "colour" && argc >= 6) {
    PCXYZPtr c(new PCXYZ);
    if (inputPointCloudData(argv[2], c) != 0) return 1;
    std::vector<std::vector<int>> clusters;
    std::ifstream f(argv[3]);
    std::string line;
    while (std::getline(f, line)) { std::istringstream ss(line); std::vector<int> v; int x; while (ss >> x) v.push_back(x); clusters.push_back(v); }
    return saveColoredClusters(argv[4], c, clusters, std::strtoull(argv[5], nullptr, 10), true) == 0 ? 0 : 1;
  }
  if (cmd == "scene" && argc >= 5) {
    std::vector<float> xyz;
    if (!vgs_scenes::make_scene(argv[2], std::atoll(argv[3]), xyz)) return 2;
    PCXYZPtr c(new PCXYZ);
    c->points.resize(xyz.size() / 3);
    for (size_t i = 0; i < c->points.size(); ++i) c->points[i] = pcl::PointXYZ(xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]);
    return outputPointCloudData(argv[4], c, true) == 0 ? 0 : 1;
  }
  if (cmd == "task") {
    const auto t = inputTaskTxtFile(argv[2]);
    if (t.size() < 25) return 1;
    std::printf("%zu %d %s %s\n", t.size(), std::atoi(t[24].c_str()), t[15].c_str(), t[21].c_str());
    return 0;
  }
  return 2;
}
