// include/vgs_segmentation.hpp -- host-side C++ mirror of the reference's two class templates over the
// C-ABI of include/vgs.h.  Same class names, member names, argument meaning and call order as
//   pcl::VoxelBasedSegmentation<PointT>       (reference voxel_segmentation.h:57-2305)
//   pcl::SuperVoxelBasedSegmentation<PointT>  (reference supervoxel_segmentation.h:58-2308)
// so that the reference's driver functions segmentationVGS / segmentationSVGS (reference `test`:9-170) compile
// against this header after replacing the PCL cloud types by the two small types below (PCL is not a
// dependency of this engine).  Every member forwards to one or two C entry points; no computation happens here.
//
// What is deliberately NOT mirrored: PCD/PLY IO, viewers, the draw*
// mesh helpers, FPFH and weighted-covariance dead code.  drawColorMapofPointsinClusters is kept because the
// reference makes it obligatory before getClusterIdx (VS:947-1014): here it returns per-point labels.
//
// Error behaviour: the reference's members are void and unchecked; here a failed call throws std::runtime_error
// carrying vgs_last_error_string (bad call order -> VGS_E_STATE instead of reading uninitialised members).
#ifndef VGS_SEGMENTATION_HPP_
#define VGS_SEGMENTATION_HPP_

#include <cstdint>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "vgs.h"

namespace pcl {

struct PointXYZ {  // 16 bytes like pcl::PointXYZ (x, y, z, padding)
  float x, y, z, pad;
  PointXYZ() : x(0), y(0), z(0), pad(1.0f) {}
  PointXYZ(float x_, float y_, float z_) : x(x_), y(y_), z(z_), pad(1.0f) {}
};

struct PointXYZRGB {  // x, y, z, padding, then the colour packed as PCL packs it: 0x00RRGGBB in one 32-bit word
  float x, y, z, pad;
  uint32_t rgba;
  float pad2[3];
  PointXYZRGB() : x(0), y(0), z(0), pad(1.0f), rgba(0), pad2{0, 0, 0} {}
};

template <typename PointT>
struct PointCloud {
  typedef std::shared_ptr<PointCloud<PointT>> Ptr;
  std::vector<PointT> points;
  uint32_t width = 0, height = 1;
  size_t size() const { return points.size(); }
};

}  // namespace pcl

typedef pcl::PointCloud<pcl::PointXYZ>::Ptr PCXYZPtr;
typedef pcl::PointCloud<pcl::PointXYZ> PCXYZ;
typedef pcl::PointCloud<pcl::PointXYZRGB> PCXYZRGB;
typedef pcl::PointCloud<pcl::PointXYZRGB>::Ptr PCXYZRGBPtr;

namespace vgs_color {
// one colour per cluster from a 64-bit LCG (Knuth's MMIX constants), seeded: the reference draws its colours with
// rand() after srand(time(0)) (VS:960, point_clouds_IO.cpp:36), so its files differ from run to run; these do not
struct Palette {
  uint64_t s;
  explicit Palette(uint64_t seed) : s(seed * 0x9E3779B97F4A7C15ull + 0xD1B54A32D192ED03ull) {}
  uint32_t next() {
    s = s * 6364136223846793005ull + 1442695040888963407ull;
    return (uint32_t)(s >> 33) & 0x00ffffffu;  // 0x00RRGGBB
  }
};
// cluster by cluster, member by member (VS:963-1009, point_clouds_IO.cpp:39-61); points of no cluster are absent
inline void color_clusters(const PCXYZ& in, const std::vector<std::vector<int>>& clusters, uint64_t seed, PCXYZRGB& out) {
  Palette pal(seed);
  size_t total = 0;
  for (const auto& c : clusters) total += c.size();
  out.points.clear();
  out.points.reserve(total);
  for (const auto& c : clusters) {
    const uint32_t colour = pal.next();
    for (int idx : c) {
      pcl::PointXYZRGB q;
      const pcl::PointXYZ& p = in.points[(size_t)idx];
      q.x = p.x; q.y = p.y; q.z = p.z; q.rgba = colour;
      out.points.push_back(q);
    }
  }
  out.width = (uint32_t)out.points.size(); out.height = 1;
}
}  // namespace vgs_color

namespace pcl {

namespace vgs_detail {
inline void check(vgs_ctx* c, vgs_status s, const char* what) {
  if (s != VGS_OK) throw std::runtime_error(std::string(what) + ": " + vgs_last_error_string(c));
}
struct CtxDeleter { void operator()(vgs_ctx* c) const { vgs_destroy(c); } };
}  // namespace vgs_detail

template <typename PointT>
class VoxelBasedSegmentation {
 public:
  struct Weight_Index { float Weight; int Index; };                                     // VS:64-68
  static bool godown(const Weight_Index& a, const Weight_Index& b) { return a.Weight > b.Weight; }  // VS:70
  static bool riseup(const Weight_Index& a, const Weight_Index& b) { return a.Weight < b.Weight; }  // VS:76

  explicit VoxelBasedSegmentation(double input_resolution) {                              // VS:84
    vgs_params_default_vgs(&p_);
    p_.voxel_size = (float)input_resolution;
    vgs_ctx* c = nullptr;
    vgs_status s = vgs_create(&p_, &c);
    if (s != VGS_OK) throw std::runtime_error(std::string("vgs_create: ") + vgs_last_error_string(nullptr));
    ctx_.reset(c);
  }

  // inherited from pcl::octree::OctreePointCloud in the reference (test:52-56)
  void setInputCloud(const PCXYZPtr& cloud) { cloud_ = cloud; }
  void addPointsFromInputCloud() {
    if (!cloud_) throw std::runtime_error("addPointsFromInputCloud before setInputCloud");
    chk(vgs_set_points(ctx(), &cloud_->points[0].x, (int64_t)cloud_->points.size(), (int32_t)sizeof(PointT)), "vgs_set_points");
    chk(vgs_voxelize(ctx()), "vgs_voxelize");
    adj_off_.clear(); adj_idx_.clear();
  }
  void getBoundingBox(double& min_x, double& min_y, double& min_z, double& max_x, double& max_y, double& max_z) {
    double b[6];
    chk(vgs_get_bbox(ctx(), b), "vgs_get_bbox");
    min_x = b[0]; min_y = b[1]; min_z = b[2]; max_x = b[3]; max_y = b[4]; max_z = b[5];
  }

  int getCloudPointNum(const PCXYZPtr& input_data) { cloud_ = input_data; return (int)input_data->points.size(); }  // VS:94
  int getVoxelNum() { return (int)count(VGS_N_VOXELS); }                 // VS:104
  int getClusterNum() { return (int)count(VGS_N_CLUSTERS); }                              // VS:111
  std::vector<std::vector<int>> getClusterIdx() {                                        // VS:117
    std::vector<std::vector<int>> out;
    if (!drawn_) return out;  // clusters_point_idx_ is filled by drawColorMapofPointsinClusters only (VS:1006)
    const int64_t k = count(VGS_N_KEPT);
    std::vector<int64_t> off((size_t)k + 1);
    // element for element what the reference's clusters_point_idx_ holds: DFS order of the nodes, seed last
    chk(vgs_get_clusters_ordered(ctx(), VGS_ORDER_REFERENCE, off.data(), nullptr), "vgs_get_clusters");
    std::vector<int32_t> idx((size_t)off[k] + 1);
    chk(vgs_get_clusters_ordered(ctx(), VGS_ORDER_REFERENCE, off.data(), idx.data()), "vgs_get_clusters");
    out.resize((size_t)k);
    for (int64_t i = 0; i < k; ++i) out[(size_t)i].assign(idx.begin() + off[i], idx.begin() + off[i + 1]);
    return out;
  }

  // VS:124-131 only STORES its arguments: the octree keeps binning with the constructor's resolution (VS:84), and so does the
  // engine -- voxel_size is not forwarded.  The stored value is what the reference's debug meshes draw their boxes with
  // (VS:561-582) and what getVoxelCenterFromOctreeKey scales the keys by (VS:2106-2108); with a value other than the
  // constructor's the reference's centres no longer lie in their voxels, which the engine does not reproduce: its centres
  // are the octree's own ("setVoxelSize").
  void setVoxelSize(double input_resolution, int points_num_min, int voxels_num_min, int voxels_adj_min) {  // VS:124
    voxel_resolution_ = (float)input_resolution; p_.points_min = points_num_min; p_.voxels_min = voxels_num_min;
    p_.adjacency_min = voxels_adj_min;
    chk(vgs_set_params(ctx(), &p_), "vgs_set_params");
  }
  float getVoxelResolution() const { return voxel_resolution_ > 0.0f ? voxel_resolution_ : p_.voxel_size; }   // (debug meshes)
  void setBoundingBox(double, double, double, double, double, double) {}                 // VS:133 (the engine keeps the octree's box)
  // VS:146.  The voxel table is built by addPointsFromInputCloud with the constructor's resolution and stays as it is.
  void setVoxelCenters() {}   // (the table is built by addPointsFromInputCloud)
  std::vector<PointXYZ> getVoxelCenters() {                                              // VS:191
    const int64_t v = count(VGS_N_VOXELS);
    std::vector<float> c((size_t)v * 3 + 1);
    chk(vgs_get_voxel_centers(ctx(), c.data()), "vgs_get_voxel_centers");
    std::vector<PointXYZ> out((size_t)v);
    for (int64_t i = 0; i < v; ++i) out[(size_t)i] = PointXYZ(c[3 * i], c[3 * i + 1], c[3 * i + 2]);
    return out;
  }
  void calcualteVoxelCloudAttributes(const PCXYZPtr&) { chk(vgs_features(ctx()), "vgs_features"); }  // VS:290 (sic)
  void findAllVoxelAdjacency(float graph_size) {                                         // VS:223
    p_.graph_size = graph_size;
    chk(vgs_set_params(ctx(), &p_), "vgs_set_params");
    chk(vgs_adjacency(ctx()), "vgs_adjacency");
    adj_off_.clear(); adj_idx_.clear();
  }
  std::vector<int> getOneVoxelAdjacency(int voxel_id) {                                  // VS:268
    // the reference keeps voxels_adjacency_idx_ in memory and a caller loops over the voxels: the lists are fetched once per
    // findAllVoxelAdjacency, not once per call
    if (adj_off_.empty()) {
      const int64_t v = count(VGS_N_VOXELS);
      adj_off_.assign((size_t)v + 1, 0);
      chk(vgs_get_lists(ctx(), 0, adj_off_.data(), nullptr), "vgs_get_lists");
      adj_idx_.assign((size_t)adj_off_[v] + 1, 0);
      chk(vgs_get_lists(ctx(), 0, adj_off_.data(), adj_idx_.data()), "vgs_get_lists");
    }
    if (voxel_id < 0 || (size_t)voxel_id + 1 >= adj_off_.size()) throw std::out_of_range("getOneVoxelAdjacency: voxel id");
    return std::vector<int>(adj_idx_.begin() + adj_off_[voxel_id], adj_idx_.begin() + adj_off_[voxel_id + 1]);
  }
  void segmentVoxelCloudWithGraphModel(float cut_thred, float sig_p, float sig_n, float sig_o, float sig_e, float sig_c,
                                       float sig_w) {                                    // VS:372
    p_.cut_thred = cut_thred; p_.sig_p = sig_p; p_.sig_n = sig_n; p_.sig_o = sig_o; p_.sig_e = sig_e; p_.sig_c = sig_c;
    p_.sig_w = sig_w;
    chk(vgs_set_params(ctx(), &p_), "vgs_set_params");
    chk(vgs_segment(ctx()), "vgs_segment");
  }
  // VS:947 "This is obligatory!": fills the per-cluster point lists; here it also returns one label per point
  std::vector<int32_t> drawColorMapofPointsinClusters() {
    std::vector<int32_t> lab((size_t)count(VGS_N_POINTS) + 1);
    chk(vgs_get_point_labels(ctx(), lab.data()), "vgs_get_point_labels");
    lab.pop_back();
    drawn_ = true;
    return lab;
  }
  // the reference's signature (VS:947): the kept clusters as a coloured cloud, one colour each (seeded palette)
  void drawColorMapofPointsinClusters(const PCXYZRGBPtr& colored_cloud, uint64_t seed = 0) {
    drawn_ = true;
    if (colored_cloud && cloud_) vgs_color::color_clusters(*cloud_, getClusterIdx(), seed, *colored_cloud);
  }

  vgs_ctx* ctx() { return ctx_.get(); }

 private:
  void chk(vgs_status s, const char* what) { vgs_detail::check(ctx_.get(), s, what); }
  int64_t count(int which) {
    int64_t c[VGS_N_COUNTS];
    chk(vgs_get_counts(ctx(), c), "vgs_get_counts");
    return c[which];
  }
  vgs_params p_;
  std::unique_ptr<vgs_ctx, vgs_detail::CtxDeleter> ctx_;
  PCXYZPtr cloud_;
  bool drawn_ = false;
  float voxel_resolution_ = 0.0f;  // setVoxelSize's value (VS:127): stored, never binned with
  std::vector<int64_t> adj_off_;   // getOneVoxelAdjacency's copy of the lists
  std::vector<int32_t> adj_idx_;
};

template <typename PointT>
class SuperVoxelBasedSegmentation {
 public:
  explicit SuperVoxelBasedSegmentation(double input_resolution) {                         // SS:85
    vgs_params_default_svgs(&p_);
    p_.voxel_size = (float)input_resolution;
    vgs_ctx* c = nullptr;
    vgs_status s = vgs_create(&p_, &c);
    if (s != VGS_OK) throw std::runtime_error(std::string("vgs_create: ") + vgs_last_error_string(nullptr));
    ctx_.reset(c);
  }
  void setInputCloud(const PCXYZPtr& cloud) { cloud_ = cloud; }
  void addPointsFromInputCloud() {                                                       // test:142 (own octree: bookkeeping only)
    if (!cloud_) throw std::runtime_error("addPointsFromInputCloud before setInputCloud");
    chk(vgs_set_points(ctx(), &cloud_->points[0].x, (int64_t)cloud_->points.size(), (int32_t)sizeof(PointT)), "vgs_set_points");
  }
  int getCloudPointNum(const PCXYZPtr& input_data) { cloud_ = input_data; return (int)input_data->points.size(); }  // SS:101
  int getVoxelNum() { return (int)count(VGS_N_VOXELS); }                                  // SS:111
  int getSuperVoxelNum() { return (int)count(VGS_N_SUPERVOXELS); }                        // SS:118
  int getClusterNum() { return (int)count(VGS_N_CLUSTERS); }                              // SS:124
  std::vector<std::vector<int>> getClusterIdx() {                                        // SS:130 (no draw call needed, SS:2124)
    const int64_t k = count(VGS_N_KEPT);
    std::vector<int64_t> off((size_t)k + 1);
    // element for element what the reference's clusters_point_idx_ holds: DFS order of the nodes, seed last
    chk(vgs_get_clusters_ordered(ctx(), VGS_ORDER_REFERENCE, off.data(), nullptr), "vgs_get_clusters");
    std::vector<int32_t> idx((size_t)off[k] + 1);
    chk(vgs_get_clusters_ordered(ctx(), VGS_ORDER_REFERENCE, off.data(), idx.data()), "vgs_get_clusters");
    std::vector<std::vector<int>> out((size_t)k);
    for (int64_t i = 0; i < k; ++i) out[(size_t)i].assign(idx.begin() + off[i], idx.begin() + off[i + 1]);
    return out;
  }
  void setVoxelSize(double input_resolution, int points_num_min) {                       // SS:143
    p_.voxel_size = (float)input_resolution; p_.points_min = points_num_min;
    chk(vgs_set_params(ctx(), &p_), "vgs_set_params");
  }
  void setSupervoxelSize(double input_resolution, int voxels_num_min, int, int adjacency_num_min) {  // SS:150
    p_.seed_size = (float)input_resolution; p_.voxels_min = voxels_num_min; p_.adjacency_min = adjacency_num_min;
    chk(vgs_set_params(ctx(), &p_), "vgs_set_params");
  }
  void setGraphSize(double /*small_resolution: consumed nowhere, SS:1438-1475*/, double large_resolution) {  // SS:159
    p_.graph_size = (float)large_resolution;
    chk(vgs_set_params(ctx(), &p_), "vgs_set_params");
  }
  void getBoundingBox(double& min_x, double& min_y, double& min_z, double& max_x, double& max_y, double& max_z) {  // test:145 (inherited from the octree)
    double b[6];
    chk(vgs_voxelize(ctx()), "vgs_voxelize");   // the class's own octree at voxel_resolution_
    chk(vgs_get_bbox(ctx(), b), "vgs_get_bbox");
    min_x = b[0]; min_y = b[1]; min_z = b[2]; max_x = b[3]; max_y = b[4]; max_z = b[5];
  }
  void setBoundingBox(double, double, double, double, double, double) {}                 // SS:166
  void setSupervoxelCentersCentroids() {}                                                // SS:178 (own-octree bookkeeping, unused by the result)
  // the supervoxel labelling pcl::SupervoxelClustering would produce may also be supplied by the caller
  void setSupervoxelLabels(const std::vector<int32_t>& labels, int max_label) {
    chk(svgs_set_supervoxel_labels(ctx(), labels.data(), max_label), "svgs_set_supervoxel_labels");
  }
  void segmentSupervoxelCloudWithGraphModel(float sig_a, float sig_b, float sig_l, float cut_thred, float sig_p, float sig_n,
                                            float sig_o, float sig_e, float sig_c, float sig_w) {  // SS:362
    p_.color_impt = sig_a; p_.spatial_impt = sig_b; p_.normal_impt = sig_l; p_.cut_thred = cut_thred;
    p_.sig_p = sig_p; p_.sig_n = sig_n; p_.sig_o = sig_o; p_.sig_e = sig_e; p_.sig_c = sig_c; p_.sig_w = sig_w;
    chk(vgs_set_params(ctx(), &p_), "vgs_set_params");
    chk(vgs_run(ctx()), "vgs_run");  // createSupervoxels (unless the caller's labelling of THIS cloud is in place), then the graph stages
  }
  std::vector<int32_t> drawColorMapofPointsinClusters() {                                // SS:613
    std::vector<int32_t> lab((size_t)count(VGS_N_POINTS) + 1);
    chk(vgs_get_point_labels(ctx(), lab.data()), "vgs_get_point_labels");
    lab.pop_back();
    return lab;
  }
  void drawColorMapofPointsinClusters(const PCXYZRGBPtr& colored_cloud, uint64_t seed = 0) {  // the reference's signature (SS:613)
    if (colored_cloud && cloud_) vgs_color::color_clusters(*cloud_, getClusterIdx(), seed, *colored_cloud);
  }
  vgs_ctx* ctx() { return ctx_.get(); }

 private:
  void chk(vgs_status s, const char* what) { vgs_detail::check(ctx_.get(), s, what); }
  int64_t count(int which) {
    int64_t c[VGS_N_COUNTS];
    chk(vgs_get_counts(ctx(), c), "vgs_get_counts");
    return c[which];
  }
  vgs_params p_;
  std::unique_ptr<vgs_ctx, vgs_detail::CtxDeleter> ctx_;
  PCXYZPtr cloud_;
};

}  // namespace pcl

#endif  // VGS_SEGMENTATION_HPP_
