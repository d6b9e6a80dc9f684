// examples/drivers.hpp -- the reference's two drivers (`test`:9-86 segmentationVGS, `test`:91-170 segmentationSVGS)
// written against include/vgs_segmentation.hpp: same objects, same call order, same parameter unpacking from the
// task vector (0-based line index).  The viewer calls (showColoredClusters) are dropped; saving is left to the caller.
#ifndef VGS_EXAMPLE_DRIVERS_HPP_
#define VGS_EXAMPLE_DRIVERS_HPP_

#include <cstdlib>
#include <string>
#include <vector>

#include "vgs_segmentation.hpp"
#include "vgs_debug_export.hpp"

struct DriverSummary { long points = 0, voxels = 0, supervoxels = 0, clusters = 0, kept = 0, labelled = 0; };

// debug_prefix: when not empty, the reference's three voxel drawings (VS:510, 654, 1016) are written as
// <prefix>_voxels.ply, <prefix>_clustered_voxels.ply, <prefix>_normals.ply
inline int segmentationVGS(PCXYZPtr input_cloud, const std::vector<std::string>& input_vector,
                           std::vector<std::vector<int>>& clusters_points_idx, DriverSummary* sum = nullptr,
                           const std::string& debug_prefix = std::string(), double ctor_resolution = 0.0) {
  float voxel_size = 0.15f, graph_size = 0.5f, sig_p = 0.2f, sig_n = 0.2f, sig_o = 0.2f, sig_e = 0.2f, sig_c = 0.2f, sig_w = 2.0f,
        cut_thred = 0.3f;
  int points_min = 10, adjacency_min = 3, voxels_min = 3;
  if (input_vector.size() > 50) {  // the twelve numbers sit on every second line from 28 on (test:25-37)
    auto num = [&](size_t line) { return std::atof(input_vector[line].c_str()); };
    float* const f[] = {&voxel_size, &graph_size, &sig_p, &sig_n, &sig_o, &sig_e, &sig_c, &sig_w, &cut_thred};
    for (size_t k = 0; k < 9; ++k) *f[k] = (float)num(28 + 2 * k);
    points_min = (int)num(46); adjacency_min = (int)num(48); voxels_min = (int)num(50);
  }
  double min_x = 0, min_y = 0, min_z = 0, max_x = 0, max_y = 0, max_z = 0;

  // Voxelization (test:51-57)
  // (ctor_resolution: tests only -- an octree resolution that differs from the one setVoxelSize is given afterwards)
  pcl::VoxelBasedSegmentation<pcl::PointXYZ> voxel_structure(ctor_resolution > 0.0 ? ctor_resolution : (double)voxel_size);
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
  PCXYZRGBPtr clustered_cloud(new PCXYZRGB);
  voxel_structure.drawColorMapofPointsinClusters(clustered_cloud);
  clusters_points_idx = voxel_structure.getClusterIdx();
  if (sum) {
    sum->points = (long)input_cloud->points.size(); sum->voxels = voxels; sum->clusters = voxel_structure.getClusterNum();
    sum->kept = (long)clusters_points_idx.size();
    sum->labelled = 0;
    for (auto& c : clusters_points_idx) sum->labelled += (long)c.size();
  }
  (void)voxel_centers;
  if (!debug_prefix.empty()) {
    pcl::PolygonMesh::Ptr colored_voxels(new pcl::PolygonMesh), clustered_voxels(new pcl::PolygonMesh), normes_voxels(new pcl::PolygonMesh);
    drawColorMapofVoxels(voxel_structure.ctx(), voxel_size, colored_voxels);
    drawColorMapofClusteredVoxels(voxel_structure.ctx(), voxel_size, clustered_voxels);
    drawNormofVoxels(voxel_structure.ctx(), voxel_size, normes_voxels);
    if (savePolygonMeshPLY(debug_prefix + "_voxels.ply", *colored_voxels) != 0 ||
        savePolygonMeshPLY(debug_prefix + "_clustered_voxels.ply", *clustered_voxels) != 0 ||
        savePolygonMeshPLY(debug_prefix + "_normals.ply", *normes_voxels) != 0)
      return -1;
  }
  return 0;
}

inline int segmentationSVGS(PCXYZPtr input_cloud, const std::vector<std::string>& input_vector,
                            std::vector<std::vector<int>>& clusters_points_idx, DriverSummary* sum = nullptr) {
  // Task_File_SVGS.txt values
  float voxel_size = 0.05f, seed_size = 0.25f, graph_size = 0.5f, sig_p = 0.2f, sig_n = 0.2f, sig_o = 0.2f, sig_e = 0.2f, sig_c = 0.2f,
        sig_w = 1.0f, sig_a = 0.0f, sig_b = 0.25f, cut_thred = 0.5f;
  int points_min = 0, voxels_min = 3, adjacency_min = 3;
  if (input_vector.size() > 60) {  // every second line from 28 on (test:108-125)
    auto num = [&](size_t line) { return std::atof(input_vector[line].c_str()); };
    float* const f[] = {&voxel_size, &seed_size, &graph_size, &sig_p, &sig_n, &sig_o, &sig_e, &sig_c, &sig_w, &sig_a, &sig_b};
    for (size_t k = 0; k < 11; ++k) *f[k] = (float)num(28 + 2 * k);
    sig_c = (float)num(50);  // the reference reuses sig_c for the normal importance (test:120): the convexity sigma of line 42 is lost
    cut_thred = (float)num(52);
    points_min = (int)num(54); voxels_min = (int)num(58); adjacency_min = (int)num(60);
  } else {
    sig_c = 0.75f;  // normal importance of Task_File_SVGS.txt, carried in sig_c as above
  }
  double min_x = 0, min_y = 0, min_z = 0, max_x = 0, max_y = 0, max_z = 0;

  pcl::SuperVoxelBasedSegmentation<pcl::PointXYZ> supervoxel_structure(voxel_size);  // test:138
  supervoxel_structure.setInputCloud(input_cloud);
  supervoxel_structure.getCloudPointNum(input_cloud);
  supervoxel_structure.addPointsFromInputCloud();
  supervoxel_structure.setVoxelSize(voxel_size, points_min);                          // test:144-146
  supervoxel_structure.setSupervoxelSize(seed_size, voxels_min, points_min, adjacency_min);
  supervoxel_structure.setGraphSize(seed_size * 2, graph_size);
  supervoxel_structure.getBoundingBox(min_x, min_y, min_z, max_x, max_y, max_z);      // test:148-149
  supervoxel_structure.setBoundingBox(min_x, min_y, min_z, max_x, max_y, max_z);
  supervoxel_structure.setSupervoxelCentersCentroids();                               // test:152-153
  supervoxel_structure.getVoxelNum();
  supervoxel_structure.segmentSupervoxelCloudWithGraphModel(sig_a, sig_b, sig_c, cut_thred, sig_p, sig_n, sig_o, sig_e, sig_c, sig_w);  // test:156
  PCXYZRGBPtr clustered_cloud(new PCXYZRGB);
  supervoxel_structure.drawColorMapofPointsinClusters(clustered_cloud);               // test:159-160
  clusters_points_idx = supervoxel_structure.getClusterIdx();
  if (sum) {
    sum->points = (long)input_cloud->points.size(); sum->voxels = supervoxel_structure.getVoxelNum();
    sum->supervoxels = supervoxel_structure.getSuperVoxelNum(); sum->clusters = supervoxel_structure.getClusterNum();
    sum->kept = (long)clusters_points_idx.size();
    sum->labelled = 0;
    for (auto& c : clusters_points_idx) sum->labelled += (long)c.size();
  }
  return 0;
}

#endif  // VGS_EXAMPLE_DRIVERS_HPP_
