// examples/segmentation_vgs.cpp -- runs the reference's driver segmentationVGS (examples/drivers.hpp, reference
// `test`:9-86) on a raw float32 xyz file.
//   usage: segmentation_vgs <points.f32> [Task_File_VGS.txt] [--ctor-res R]
// --ctor-res: construct the class with resolution R and let setVoxelSize set the task file's afterwards (tests)
// Prints "<points> <voxels> <all clusters> <kept clusters> <labelled points>".
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <string>
#include <vector>

#include "drivers.hpp"
#include "point_clouds_io.hpp"

using std::string;
using std::vector;

int main(int argc, char** argv) {
  if (argc < 2) { std::fprintf(stderr, "usage: %s points.f32 [task file]\n", argv[0]); return 2; }
  std::ifstream f(argv[1], std::ios::binary);
  if (!f) { std::fprintf(stderr, "cannot open %s\n", argv[1]); return 2; }
  f.seekg(0, std::ios::end);
  const size_t bytes = (size_t)f.tellg();
  f.seekg(0);
  vector<float> raw(bytes / 4);
  f.read((char*)raw.data(), (std::streamsize)(raw.size() * 4));
  PCXYZPtr cloud(new PCXYZ);
  cloud->points.resize(raw.size() / 3);
  for (size_t i = 0; i < cloud->points.size(); ++i) cloud->points[i] = pcl::PointXYZ(raw[3 * i], raw[3 * i + 1], raw[3 * i + 2]);
  vector<string> task;
  double ctor_res = 0.0;
  for (int a = 2; a < argc; ++a) {
    if (string(argv[a]) == "--ctor-res" && a + 1 < argc) ctor_res = std::atof(argv[++a]);
    else task = inputTaskTxtFile(argv[a]);
  }
  DriverSummary sum;
  vector<vector<int>> clusters;
  try {
    segmentationVGS(cloud, task, clusters, &sum, string(), ctor_res);
  } catch (const std::exception& e) {
    std::fprintf(stderr, "error: %s\n", e.what());
    return 1;
  }
  std::printf("%ld %ld %ld %ld %ld\n", sum.points, sum.voxels, sum.clusters, sum.kept, sum.labelled);
  return 0;
}
