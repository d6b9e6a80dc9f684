// vgs_debug_export.hpp -- the reference's voxel debug meshes on top of the class mirror:
//   drawColorMapofVoxels          voxel_segmentation.h:510-652   one coloured box per used voxel
//   drawColorMapofClusteredVoxels voxel_segmentation.h:654-...   boxes coloured by cluster (all voxels of a cluster alike)
//   drawNormofVoxels              voxel_segmentation.h:1016-1104 one segment per used voxel along its normal
// as free functions over a small PolygonMesh (XYZRGB vertices + index polygons) with a PLY writer.  Host-side only:
// they read the engine's voxel table through the C-ABI getters.  Colours come from the seeded palette of
// vgs_segmentation.hpp (the reference seeds rand() with the wall clock).
#ifndef VGS_DEBUG_EXPORT_HPP_
#define VGS_DEBUG_EXPORT_HPP_

#include <cstdio>
#include <fstream>
#include <string>
#include <vector>

#include "vgs_segmentation.hpp"

namespace pcl {
struct Vertices { std::vector<uint32_t> vertices; };
struct PolygonMesh {
  typedef std::shared_ptr<PolygonMesh> Ptr;
  PointCloud<PointXYZRGB> cloud;
  std::vector<Vertices> polygons;
};
}  // namespace pcl

namespace vgs_debug {

struct VoxelDump {   // what the three drawings need, fetched once
  std::vector<float> center, centroid, normal;
  std::vector<uint8_t> used;
  std::vector<int32_t> root, kept;
  float voxel_size = 0;
  int64_t V = 0;
};

inline VoxelDump fetch(vgs_ctx* ctx, float voxel_size, bool with_labels) {
  VoxelDump d;
  int64_t c[VGS_N_COUNTS];
  if (vgs_get_counts(ctx, c) != VGS_OK) throw std::runtime_error(vgs_last_error_string(ctx));
  d.V = c[VGS_N_VOXELS];
  d.voxel_size = voxel_size;
  const size_t V = (size_t)d.V;
  d.center.resize(3 * V + 3); d.centroid.resize(3 * V + 3); d.normal.resize(3 * V + 3); d.used.resize(V + 1);
  std::vector<float> eig(8 * V + 8);
  if (vgs_get_voxel_centers(ctx, d.center.data()) != VGS_OK || vgs_get_attributes(ctx, d.centroid.data(), d.normal.data(), eig.data(), d.used.data()) != VGS_OK)
    throw std::runtime_error(vgs_last_error_string(ctx));
  if (with_labels) {
    d.root.resize(V + 1); d.kept.resize(V + 1);
    if (vgs_get_node_labels(ctx, d.root.data(), d.kept.data()) != VGS_OK) throw std::runtime_error(vgs_last_error_string(ctx));
  }
  return d;
}

inline void add_box(pcl::PolygonMesh& m, const float* c, float size, uint32_t rgb) {
  const uint32_t base = (uint32_t)m.cloud.points.size();
  const float h = 0.5f * size;
  for (int k = 0; k < 8; ++k) {
    pcl::PointXYZRGB p;
    p.x = c[0] + ((k & 1) ? h : -h); p.y = c[1] + ((k & 2) ? h : -h); p.z = c[2] + ((k & 4) ? h : -h);
    p.rgba = rgb;
    m.cloud.points.push_back(p);
  }
  static const int Q[6][4] = {{0, 1, 3, 2}, {4, 6, 7, 5}, {0, 4, 5, 1}, {2, 3, 7, 6}, {0, 2, 6, 4}, {1, 5, 7, 3}};  // six quads, outward
  for (const auto& q : Q) {
    pcl::Vertices v;
    for (int k = 0; k < 4; ++k) v.vertices.push_back(base + (uint32_t)q[k]);
    m.polygons.push_back(v);
  }
}

}  // namespace vgs_debug

// one randomly coloured box per used voxel (VS:510)
inline void drawColorMapofVoxels(vgs_ctx* ctx, float voxel_size, pcl::PolygonMesh::Ptr out, uint64_t seed = 0) {
  const vgs_debug::VoxelDump d = vgs_debug::fetch(ctx, voxel_size, false);
  vgs_color::Palette pal(seed);
  out->cloud.points.clear(); out->polygons.clear();
  for (int64_t v = 0; v < d.V; ++v)
    if (d.used[(size_t)v]) vgs_debug::add_box(*out, &d.center[3 * (size_t)v], voxel_size, pal.next());
}

// boxes coloured by cluster; voxels of dropped clusters (<= voxels_min) are left out (VS:654)
inline void drawColorMapofClusteredVoxels(vgs_ctx* ctx, float voxel_size, pcl::PolygonMesh::Ptr out, uint64_t seed = 0) {
  const vgs_debug::VoxelDump d = vgs_debug::fetch(ctx, voxel_size, true);
  out->cloud.points.clear(); out->polygons.clear();
  std::vector<uint32_t> colour;   // per kept cluster label
  vgs_color::Palette pal(seed);
  for (int64_t v = 0; v < d.V; ++v) {
    const int32_t k = d.kept[(size_t)v];
    if (k < 0) continue;
    while ((size_t)k >= colour.size()) colour.push_back(pal.next());
    vgs_debug::add_box(*out, &d.center[3 * (size_t)v], voxel_size, colour[(size_t)k]);
  }
}

// the normal of every used voxel as a two-vertex polygon from its centroid (VS:1016)
inline void drawNormofVoxels(vgs_ctx* ctx, float voxel_size, pcl::PolygonMesh::Ptr out, uint64_t seed = 0) {
  const vgs_debug::VoxelDump d = vgs_debug::fetch(ctx, voxel_size, false);
  out->cloud.points.clear(); out->polygons.clear();
  const uint32_t rgb = vgs_color::Palette(seed).next();
  for (int64_t v = 0; v < d.V; ++v) {
    if (!d.used[(size_t)v]) continue;
    const float* c = &d.centroid[3 * (size_t)v];
    const float* n = &d.normal[3 * (size_t)v];
    pcl::PointXYZRGB a, b;
    a.x = c[0]; a.y = c[1]; a.z = c[2]; a.rgba = rgb;
    b.x = c[0] + voxel_size * n[0]; b.y = c[1] + voxel_size * n[1]; b.z = c[2] + voxel_size * n[2]; b.rgba = rgb;
    const uint32_t base = (uint32_t)out->cloud.points.size();
    out->cloud.points.push_back(a); out->cloud.points.push_back(b);
    pcl::Vertices e; e.vertices = {base, base + 1};
    out->polygons.push_back(e);
  }
}

// ASCII PLY: polygons with two vertices become an edge element, the others faces
inline int savePolygonMeshPLY(const std::string& name, const pcl::PolygonMesh& m) {
  std::ofstream f(name);
  if (!f.is_open()) return -1;
  size_t n_edge = 0, n_face = 0;
  for (const auto& p : m.polygons) (p.vertices.size() == 2 ? n_edge : n_face)++;
  f << "ply\nformat ascii 1.0\nelement vertex " << m.cloud.points.size()
    << "\nproperty float x\nproperty float y\nproperty float z\nproperty uchar red\nproperty uchar green\nproperty uchar blue\n";
  if (n_face) f << "element face " << n_face << "\nproperty list uchar int vertex_indices\n";
  if (n_edge) f << "element edge " << n_edge << "\nproperty int vertex1\nproperty int vertex2\n";
  f << "end_header\n";
  char buf[128];
  for (const auto& p : m.cloud.points) {
    const int k = std::snprintf(buf, sizeof buf, "%.9g %.9g %.9g %u %u %u\n", p.x, p.y, p.z, (p.rgba >> 16) & 255u, (p.rgba >> 8) & 255u, p.rgba & 255u);
    f.write(buf, k);
  }
  for (const auto& p : m.polygons)
    if (p.vertices.size() != 2) { f << p.vertices.size(); for (uint32_t v : p.vertices) f << ' ' << v; f << '\n'; }
  for (const auto& p : m.polygons)
    if (p.vertices.size() == 2) f << p.vertices[0] << ' ' << p.vertices[1] << '\n';
  return f.good() ? 0 : -1;
}

#endif  // VGS_DEBUG_EXPORT_HPP_
