// examples/segmentation_vgs.cpp -- the reference's driver segmentationVGS (reference `test`:9-86) written
// against include/vgs_segmentation.hpp: same objects, same call order, same parameter unpacking from the task
// vector (lines 28..50).  File IO and the viewer of the reference are replaced by a raw float32 xyz file.
//   usage: segmentation_vgs <points.f32> [Task_File_VGS.txt]
// Prints "<points> <voxels> <all clusters> <kept clusters> <labelled points>".
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <string>
#include <vector>

#include "vgs_segmentation.hpp"

using std::string;
using std::vector;

static vector<string> inputTaskTxtFile(const string& path) {  // point_clouds_IO.cpp:148-169 (CR stripped)
  vector<string> v;
  std::ifstream f(path);
  string line;
  while (std::getline(f, line)) {
    while (!line.empty() && (line.back() == '\r' || line.back() == '\n')) line.pop_back();
    v.push_back(line);
  }
  return v;
}

int segmentationVGS(PCXYZPtr input_cloud, const vector<string>& input_vector, long out[5]) {
  float voxel_size = 0.15f, graph_size = 0.5f, sig_p = 0.2f, sig_n = 0.2f, sig_o = 0.2f, sig_e = 0.2f, sig_c = 0.2f, sig_w = 2.0f,
        cut_thred = 0.3f;
  int points_min = 10, adjacency_min = 3, voxels_min = 3;
  if (input_vector.size() > 50) {  // test:25-37
    voxel_size = (float)std::atof(input_vector[28].c_str());
    graph_size = (float)std::atof(input_vector[30].c_str());
    sig_p = (float)std::atof(input_vector[32].c_str());
    sig_n = (float)std::atof(input_vector[34].c_str());
    sig_o = (float)std::atof(input_vector[36].c_str());
    sig_e = (float)std::atof(input_vector[38].c_str());
    sig_c = (float)std::atof(input_vector[40].c_str());
    sig_w = (float)std::atof(input_vector[42].c_str());
    cut_thred = (float)std::atof(input_vector[44].c_str());
    points_min = std::atoi(input_vector[46].c_str());
    adjacency_min = std::atoi(input_vector[48].c_str());
    voxels_min = std::atoi(input_vector[50].c_str());
  }
  double min_x = 0, min_y = 0, min_z = 0, max_x = 0, max_y = 0, max_z = 0;

  // Voxelization (test:51-57)
  pcl::VoxelBasedSegmentation<pcl::PointXYZ> voxel_structure(voxel_size);
  voxel_structure.setInputCloud(input_cloud);
  voxel_structure.getCloudPointNum(input_cloud);
  voxel_structure.addPointsFromInputCloud();
  voxel_structure.setVoxelSize(voxel_size, points_min, voxels_min, adjacency_min);
  voxel_structure.getBoundingBox(min_x, min_y, min_z, max_x, max_y, max_z);
  voxel_structure.setBoundingBox(min_x, min_y, min_z, max_x, max_y, max_z);
  // centres (test:60-62)
  voxel_structure.setVoxelCenters();
  auto voxel_centers = voxel_structure.getVoxelCenters();
  const int voxels = voxel_structure.getVoxelNum();
  // features, adjacency, segmentation (test:65-71)
  voxel_structure.calcualteVoxelCloudAttributes(input_cloud);
  voxel_structure.findAllVoxelAdjacency(graph_size);
  voxel_structure.segmentVoxelCloudWithGraphModel(cut_thred, sig_p, sig_n, sig_o, sig_e, sig_c, sig_w);
  // output (test:74-76)
  vector<int32_t> labels = voxel_structure.drawColorMapofPointsinClusters();
  vector<vector<int>> clusters_points_idx = voxel_structure.getClusterIdx();
  long labelled = 0;
  for (auto& c : clusters_points_idx) labelled += (long)c.size();
  out[0] = (long)input_cloud->points.size(); out[1] = voxels; out[2] = voxel_structure.getClusterNum();
  out[3] = (long)clusters_points_idx.size(); out[4] = labelled;
  (void)voxel_centers; (void)labels;
  return 0;
}

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
  if (argc > 2) task = inputTaskTxtFile(argv[2]);
  long out[5];
  try {
    segmentationVGS(cloud, task, out);
  } catch (const std::exception& e) {
    std::fprintf(stderr, "error: %s\n", e.what());
    return 1;
  }
  std::printf("%ld %ld %ld %ld %ld\n", out[0], out[1], out[2], out[3], out[4]);
  return 0;
}
