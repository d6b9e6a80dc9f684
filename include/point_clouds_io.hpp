// point_clouds_io.hpp -- the file formats either side of the segmentation path: what the
// reference's point_clouds_IO.h / point_clouds_IO.cpp do through pcl::io, on this repo's own cloud types.
//
//   inputPointCloudData(name, cloud)        point_clouds_IO.h:64-79   PCD reader: DATA ascii | binary | binary_compressed
//   inputPointCloudData2(name, cloud)       point_clouds_IO.h:81-95   PLY reader: ascii | binary_little_endian | binary_big_endian
//   outputPointCloudData(name, cloud)       point_clouds_IO.h:98-108  PCD writer (ascii like pcl::io::savePCDFile's default,
//                                                                      or binary)
//   saveColoredClusters(name, cloud, idx)   point_clouds_IO.cpp:23-70 XYZRGB PCD, one colour per cluster
//   inputTaskTxtFile(name)                  point_clouds_IO.cpp:148-169 every line of the task file (CR stripped)
//
// Differences from the reference, on purpose:
//   * colours come from a SEEDED generator (the reference calls srand(time(0)), point_clouds_IO.cpp:36): the same
//     clusters give the same file; pass another seed for another palette;
//   * the VTK viewer (point_clouds_IO.cpp:79-145) is not provided;
//   * a file that cannot be read returns -1 with a message on stderr (the reference prints PCL_ERROR and returns -1).
// Host-only, header-only, no GPU code: usable with or without libvgs_hip.so.
#ifndef POINT_CLOUDS_IO_HPP_
#define POINT_CLOUDS_IO_HPP_

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <new>
#include <sstream>
#include <string>
#include <vector>

#include "vgs_segmentation.hpp"

namespace vgs_io {

struct PcdField { std::string name; int size = 4; char type = 'F'; int count = 1; int offset = 0; };
struct PcdHeader {
  std::vector<PcdField> fields;
  uint64_t width = 0, height = 1, points = 0;
  std::string data;   // ascii | binary | binary_compressed
  int point_size = 0; // bytes of one point in the binary layouts
};

// LZF decompression (the format of Marc Lehmann's liblzf, which PCD's binary_compressed uses): control byte < 32 starts a
// literal run of ctrl+1 bytes, otherwise a back reference of length (ctrl >> 5) + 2 (7 = one more length byte follows)
// at distance ((ctrl & 31) << 8 | next) + 1.  Returns the number of bytes written, 0 on malformed input.
inline size_t lzf_decompress(const unsigned char* in, size_t in_len, unsigned char* out, size_t out_len) {
  size_t ip = 0, op = 0;
  while (ip < in_len) {
    unsigned ctrl = in[ip++];
    if (ctrl < 32) {
      const size_t run = ctrl + 1;
      if (op + run > out_len || ip + run > in_len) return 0;
      std::memcpy(out + op, in + ip, run);
      op += run; ip += run;
    } else {
      size_t len = ctrl >> 5;
      if (len == 7) { if (ip >= in_len) return 0; len += in[ip++]; }
      if (ip >= in_len) return 0;
      const size_t dist = (((size_t)(ctrl & 31)) << 8 | in[ip++]) + 1;
      len += 2;
      if (dist > op || op + len > out_len) return 0;
      for (size_t k = 0; k < len; ++k, ++op) out[op] = out[op - dist];  // may overlap: byte by byte
    }
  }
  return op;
}

inline bool parse_pcd_header(std::istream& f, PcdHeader& h, std::string& err) {
  std::string line;
  std::vector<int> sizes, counts;
  std::vector<char> types;
  std::vector<std::string> names;
  bool have_points = false;
  while (std::getline(f, line)) {
    while (!line.empty() && (line.back() == '\r' || line.back() == '\n')) line.pop_back();
    if (line.empty() || line[0] == '#') continue;
    std::istringstream ss(line);
    std::string key;
    ss >> key;
    if (key == "VERSION") continue;
    if (key == "FIELDS" || key == "COLUMNS") { std::string s; while (ss >> s) names.push_back(s); }
    else if (key == "SIZE") { int s; while (ss >> s) sizes.push_back(s); }
    else if (key == "TYPE") { char c; while (ss >> c) types.push_back(c); }
    else if (key == "COUNT") { int c; while (ss >> c) counts.push_back(c); }
    else if (key == "WIDTH") ss >> h.width;
    else if (key == "HEIGHT") ss >> h.height;
    else if (key == "VIEWPOINT") continue;
    else if (key == "POINTS") { ss >> h.points; have_points = true; }
    else if (key == "DATA") { ss >> h.data; break; }
    else { err = "unknown PCD header entry '" + key + "'"; return false; }
  }
  if (h.data.empty()) { err = "PCD header has no DATA line"; return false; }
  if (names.empty() || sizes.size() != names.size() || types.size() != names.size()) { err = "PCD header: FIELDS / SIZE / TYPE do not match"; return false; }
  if (counts.empty()) counts.assign(names.size(), 1);
  if (counts.size() != names.size()) { err = "PCD header: COUNT does not match FIELDS"; return false; }
  if (!have_points) h.points = h.width * h.height;
  int off = 0;
  for (size_t k = 0; k < names.size(); ++k) {
    PcdField fd; fd.name = names[k]; fd.size = sizes[k]; fd.type = types[k]; fd.count = counts[k]; fd.offset = off;
    if (fd.size != 1 && fd.size != 2 && fd.size != 4 && fd.size != 8) { err = "PCD header: unsupported SIZE"; return false; }
    // COUNT < 1 would give a negative or zero field offset (reads before the point), a huge COUNT an overflowed point size
    if (fd.count < 1 || fd.count > (1 << 16)) { err = "PCD header: COUNT must be in [1, 65536]"; return false; }
    off += fd.size * fd.count;
    if (off > (1 << 16)) { err = "PCD header: a point takes more than 65536 bytes"; return false; }
    h.fields.push_back(fd);
  }
  h.point_size = off;
  return true;
}

inline double pcd_value(const unsigned char* p, const PcdField& f) {
  switch (f.type) {
    case 'F': if (f.size == 4) { float v; std::memcpy(&v, p, 4); return v; } else { double v; std::memcpy(&v, p, 8); return v; }
    case 'I': if (f.size == 1) { int8_t v; std::memcpy(&v, p, 1); return v; } if (f.size == 2) { int16_t v; std::memcpy(&v, p, 2); return v; }
              if (f.size == 4) { int32_t v; std::memcpy(&v, p, 4); return v; } { int64_t v; std::memcpy(&v, p, 8); return (double)v; }
    default:  if (f.size == 1) { uint8_t v; std::memcpy(&v, p, 1); return v; } if (f.size == 2) { uint16_t v; std::memcpy(&v, p, 2); return v; }
              if (f.size == 4) { uint32_t v; std::memcpy(&v, p, 4); return v; } { uint64_t v; std::memcpy(&v, p, 8); return (double)v; }
  }
}

// reads the x, y, z fields (any order, any numeric type) of a PCD file; other fields are skipped
inline int read_pcd_xyz(const std::string& name, std::vector<pcl::PointXYZ>& pts, uint32_t* width, uint32_t* height, std::string& err) {
  std::ifstream f(name, std::ios::binary);
  if (!f.is_open()) { err = "cannot open " + name; return -1; }
  PcdHeader h;
  if (!parse_pcd_header(f, h, err)) return -1;
  int fx = -1, fy = -1, fz = -1;
  for (size_t k = 0; k < h.fields.size(); ++k) {
    if (h.fields[k].name == "x") fx = (int)k; else if (h.fields[k].name == "y") fy = (int)k; else if (h.fields[k].name == "z") fz = (int)k;
  }
  if (fx < 0 || fy < 0 || fz < 0) { err = "PCD file has no x / y / z fields"; return -1; }
  const size_t n = (size_t)h.points;
  uint64_t left = 0;   // bytes of the file behind the header
  {
    // a corrupt header must not turn into a huge allocation: every point takes at least two bytes of the file
    const std::streampos here = f.tellg();
    f.seekg(0, std::ios::end);
    left = (uint64_t)(f.tellg() - here);
    f.seekg(here);
    if (h.data != "binary_compressed" && (uint64_t)n > left) { err = "PCD header announces more points than the file can hold"; return -1; }
    if (h.data == "binary" && (uint64_t)n * (uint64_t)h.point_size > left) { err = "PCD binary body is truncated"; return -1; }
    // LZF expands a byte at most 264 / 3 times (a three-byte back reference of the maximum length)
    if (h.data == "binary_compressed" && (uint64_t)n * (uint64_t)h.point_size > 88u * left + 64u) { err = "PCD header announces more points than the file can hold"; return -1; }
  }
  try {
  pts.assign(n, pcl::PointXYZ());
  if (h.data == "ascii") {
    std::string line;
    size_t i = 0;
    std::vector<double> vals;
    while (i < n && std::getline(f, line)) {
      if (line.empty() || line == "\r") continue;
      vals.clear();
      const char* s = line.c_str();
      char* e = nullptr;
      while (true) { const double v = std::strtod(s, &e); if (e == s) break; vals.push_back(v); s = e; }
      size_t col = 0; double xyz[3] = {0, 0, 0};
      for (size_t k = 0; k < h.fields.size(); ++k) {
        if (col >= vals.size()) { err = "PCD ascii row " + std::to_string(i) + " is short"; return -1; }
        if ((int)k == fx) xyz[0] = vals[col]; else if ((int)k == fy) xyz[1] = vals[col]; else if ((int)k == fz) xyz[2] = vals[col];
        col += (size_t)h.fields[k].count;
      }
      pts[i++] = pcl::PointXYZ((float)xyz[0], (float)xyz[1], (float)xyz[2]);
    }
    if (i != n) { err = "PCD ascii body has " + std::to_string(i) + " of " + std::to_string(n) + " points"; return -1; }
  } else if (h.data == "binary") {
    std::vector<unsigned char> buf(n * (size_t)h.point_size);
    f.read((char*)buf.data(), (std::streamsize)buf.size());
    if ((size_t)f.gcount() != buf.size()) { err = "PCD binary body is truncated"; return -1; }
    for (size_t i = 0; i < n; ++i) {
      const unsigned char* p = buf.data() + i * (size_t)h.point_size;
      pts[i] = pcl::PointXYZ((float)pcd_value(p + h.fields[fx].offset, h.fields[fx]), (float)pcd_value(p + h.fields[fy].offset, h.fields[fy]),
                             (float)pcd_value(p + h.fields[fz].offset, h.fields[fz]));
    }
  } else if (h.data == "binary_compressed") {
    // two 32-bit sizes, then the LZF stream of the points stored field by field (structure of arrays)
    uint32_t csize = 0, usize = 0;
    f.read((char*)&csize, 4); f.read((char*)&usize, 4);
    if (!f || (size_t)usize != n * (size_t)h.point_size) { err = "PCD binary_compressed: bad size words"; return -1; }
    if ((uint64_t)csize + 8u > left) { err = "PCD binary_compressed body is truncated"; return -1; }
    std::vector<unsigned char> in(csize), out(usize);
    f.read((char*)in.data(), csize);
    if ((size_t)f.gcount() != (size_t)csize) { err = "PCD binary_compressed body is truncated"; return -1; }
    if (usize > 0 && lzf_decompress(in.data(), csize, out.data(), usize) != usize) { err = "PCD binary_compressed: LZF stream is corrupt"; return -1; }
    std::vector<size_t> start(h.fields.size());
    size_t acc = 0;
    for (size_t k = 0; k < h.fields.size(); ++k) { start[k] = acc; acc += n * (size_t)(h.fields[k].size * h.fields[k].count); }
    for (size_t i = 0; i < n; ++i) {
      auto at = [&](int k) { return pcd_value(out.data() + start[(size_t)k] + i * (size_t)(h.fields[(size_t)k].size * h.fields[(size_t)k].count), h.fields[(size_t)k]); };
      pts[i] = pcl::PointXYZ((float)at(fx), (float)at(fy), (float)at(fz));
    }
  } else {
    err = "unsupported PCD DATA '" + h.data + "'";
    return -1;
  }
  } catch (const std::bad_alloc&) {
    err = "PCD file is too large for this host's memory";
    return -1;
  }
  if (width) *width = (uint32_t)h.width;
  if (height) *height = (uint32_t)h.height;
  return 0;
}

// PLY: the vertex element's x, y, z properties (any scalar type); ascii, binary_little_endian and binary_big_endian;
// list properties inside the vertex element are not supported, other elements (faces) are ignored
inline int read_ply_xyz(const std::string& name, std::vector<pcl::PointXYZ>& pts, std::string& err) {
  std::ifstream f(name, std::ios::binary);
  if (!f.is_open()) { err = "cannot open " + name; return -1; }
  std::string line;
  if (!std::getline(f, line) || line.substr(0, 3) != "ply") { err = "not a PLY file"; return -1; }
  struct Prop { std::string name; int size; char kind; };  // kind: f float, i signed, u unsigned
  auto type_of = [](const std::string& t, Prop& p) -> bool {
    static const struct { const char* n; int size; char kind; } T[] = {
        {"float", 4, 'f'}, {"float32", 4, 'f'}, {"double", 8, 'f'}, {"float64", 8, 'f'}, {"char", 1, 'i'}, {"int8", 1, 'i'},
        {"uchar", 1, 'u'}, {"uint8", 1, 'u'}, {"short", 2, 'i'}, {"int16", 2, 'i'}, {"ushort", 2, 'u'}, {"uint16", 2, 'u'},
        {"int", 4, 'i'}, {"int32", 4, 'i'}, {"uint", 4, 'u'}, {"uint32", 4, 'u'}};
    for (const auto& e : T) if (t == e.n) { p.size = e.size; p.kind = e.kind; return true; }
    return false;
  };
  std::string format;
  std::vector<Prop> props;
  size_t n = 0;
  bool in_vertex = false, seen_vertex = false, vertex_first = true, ended = false;
  while (std::getline(f, line)) {
    while (!line.empty() && (line.back() == '\r' || line.back() == '\n')) line.pop_back();
    std::istringstream ss(line);
    std::string key;
    ss >> key;
    if (key == "format") ss >> format;
    else if (key == "element") {
      std::string el; size_t cnt = 0;
      ss >> el >> cnt;
      in_vertex = (el == "vertex");
      if (in_vertex) { n = cnt; seen_vertex = true; } else if (!seen_vertex && cnt > 0) vertex_first = false;
    } else if (key == "property" && in_vertex) {
      std::string t; ss >> t;
      if (t == "list") { err = "PLY: list property in the vertex element"; return -1; }
      Prop p; ss >> p.name;
      if (!type_of(t, p)) { err = "PLY: unknown property type " + t; return -1; }
      props.push_back(p);
    } else if (key == "end_header") { ended = true; break; }
  }
  if (!ended || !seen_vertex) { err = "PLY header has no vertex element"; return -1; }
  if (!vertex_first) { err = "PLY: an element precedes the vertices"; return -1; }
  int ix = -1, iy = -1, iz = -1, stride = 0;
  std::vector<int> off(props.size());
  for (size_t k = 0; k < props.size(); ++k) {
    off[k] = stride; stride += props[k].size;
    if (props[k].name == "x") ix = (int)k; else if (props[k].name == "y") iy = (int)k; else if (props[k].name == "z") iz = (int)k;
  }
  if (ix < 0 || iy < 0 || iz < 0) { err = "PLY vertex element has no x / y / z"; return -1; }
  {
    const std::streampos here = f.tellg();
    f.seekg(0, std::ios::end);
    const uint64_t left = (uint64_t)(f.tellg() - here);
    f.seekg(here);
    if ((uint64_t)n > left) { err = "PLY header announces more vertices than the file can hold"; return -1; }
  }
  pts.assign(n, pcl::PointXYZ());
  if (format == "ascii") {
    for (size_t i = 0; i < n; ++i) {
      if (!std::getline(f, line)) { err = "PLY ascii body is short"; return -1; }
      const char* s = line.c_str(); char* e = nullptr;
      double xyz[3] = {0, 0, 0};
      for (size_t k = 0; k < props.size(); ++k) {
        const double v = std::strtod(s, &e);
        if (e == s) { err = "PLY ascii row " + std::to_string(i) + " is short"; return -1; }
        s = e;
        if ((int)k == ix) xyz[0] = v; else if ((int)k == iy) xyz[1] = v; else if ((int)k == iz) xyz[2] = v;
      }
      pts[i] = pcl::PointXYZ((float)xyz[0], (float)xyz[1], (float)xyz[2]);
    }
  } else if (format == "binary_little_endian" || format == "binary_big_endian") {
    const bool swap = (format == "binary_big_endian");
    std::vector<unsigned char> buf(n * (size_t)stride);
    f.read((char*)buf.data(), (std::streamsize)buf.size());
    if ((size_t)f.gcount() != buf.size()) { err = "PLY binary body is truncated"; return -1; }
    auto val = [&](const unsigned char* p, const Prop& pr) -> double {
      unsigned char b[8];
      for (int k = 0; k < pr.size; ++k) b[k] = swap ? p[pr.size - 1 - k] : p[k];
      PcdField fd; fd.size = pr.size; fd.type = pr.kind == 'f' ? 'F' : (pr.kind == 'i' ? 'I' : 'U');
      return pcd_value(b, fd);
    };
    for (size_t i = 0; i < n; ++i) {
      const unsigned char* p = buf.data() + i * (size_t)stride;
      pts[i] = pcl::PointXYZ((float)val(p + off[ix], props[ix]), (float)val(p + off[iy], props[iy]), (float)val(p + off[iz], props[iz]));
    }
  } else {
    err = "unsupported PLY format '" + format + "'";
    return -1;
  }
  return 0;
}

inline void write_pcd_header(std::ostream& f, bool rgb, size_t n, bool binary) {
  f << "# .PCD v0.7 - Point Cloud Data file format\nVERSION 0.7\n";
  if (rgb) f << "FIELDS x y z rgb\nSIZE 4 4 4 4\nTYPE F F F F\nCOUNT 1 1 1 1\n";
  else f << "FIELDS x y z\nSIZE 4 4 4\nTYPE F F F\nCOUNT 1 1 1\n";
  f << "WIDTH " << n << "\nHEIGHT 1\nVIEWPOINT 0 0 0 1 0 0 0\nPOINTS " << n << "\nDATA " << (binary ? "binary" : "ascii") << "\n";
}

typedef vgs_color::Palette Palette;  // the seeded per-cluster colours (vgs_segmentation.hpp)

}  // namespace vgs_io

// ---- the reference's function names (point_clouds_IO.h) ----

inline int inputPointCloudData(const std::string& dataName, PCXYZPtr dataCloud) {  // point_clouds_IO.h:64
  std::string err;
  if (!dataCloud || vgs_io::read_pcd_xyz(dataName, dataCloud->points, &dataCloud->width, &dataCloud->height, err) != 0) {
    std::fprintf(stderr, "Couldn't read the PCD file! (%s)\n", err.c_str());
    return -1;
  }
  return 0;
}

inline int inputPointCloudData2(const std::string& dataName, PCXYZPtr dataCloud) {  // point_clouds_IO.h:81 (PLY)
  std::string err;
  if (!dataCloud || vgs_io::read_ply_xyz(dataName, dataCloud->points, err) != 0) {
    std::fprintf(stderr, "Couldn't read the PLY file! (%s)\n", err.c_str());
    return -1;
  }
  dataCloud->width = (uint32_t)dataCloud->points.size(); dataCloud->height = 1;
  return 0;
}

inline int outputPointCloudData(const std::string& outName, PCXYZPtr dataCloud, bool binary = false) {  // point_clouds_IO.h:98
  std::ofstream f(outName, std::ios::binary);
  if (!f.is_open() || !dataCloud) { std::fprintf(stderr, "Couldn't save the PCD file!\n"); return -1; }
  const size_t n = dataCloud->points.size();
  vgs_io::write_pcd_header(f, false, n, binary);
  if (binary) {
    for (const auto& p : dataCloud->points) f.write((const char*)&p.x, 12);
  } else {
    char buf[96];
    for (const auto& p : dataCloud->points) { const int k = std::snprintf(buf, sizeof buf, "%.9g %.9g %.9g\n", p.x, p.y, p.z); f.write(buf, k); }
  }
  return f.good() ? 0 : -1;
}

inline int saveColoredClusters(const std::string& fileoutpath_name, PCXYZRGBPtr colored_cloud, bool binary = true) {  // point_clouds_IO.cpp:72
  std::ofstream f(fileoutpath_name, std::ios::binary);
  if (!f.is_open() || !colored_cloud) { std::fprintf(stderr, "Couldn't save the PCD file!\n"); return -1; }
  const size_t n = colored_cloud->points.size();
  vgs_io::write_pcd_header(f, true, n, binary);
  if (binary) {
    for (const auto& p : colored_cloud->points) { f.write((const char*)&p.x, 12); f.write((const char*)&p.rgba, 4); }
  } else {
    char buf[128];
    for (const auto& p : colored_cloud->points) {
      float packed; std::memcpy(&packed, &p.rgba, 4);  // PCL prints the packed colour as a float
      const int k = std::snprintf(buf, sizeof buf, "%.9g %.9g %.9g %.9g\n", p.x, p.y, p.z, packed); f.write(buf, k);
    }
  }
  return f.good() ? 0 : -1;
}

inline PCXYZRGBPtr colorClusters(PCXYZPtr input_cloud, const std::vector<std::vector<int>>& clusters_points_idx, uint64_t seed = 0) {
  PCXYZRGBPtr out(new PCXYZRGB);
  vgs_color::color_clusters(*input_cloud, clusters_points_idx, seed, *out);
  return out;
}

inline int saveColoredClusters(const std::string& fileoutpath_name, PCXYZPtr input_cloud,
                               const std::vector<std::vector<int>>& clusters_points_idx, uint64_t seed = 0, bool binary = true) {  // point_clouds_IO.cpp:23
  return saveColoredClusters(fileoutpath_name, colorClusters(input_cloud, clusters_points_idx, seed), binary);
}

inline std::vector<std::string> inputTaskTxtFile(const std::string& pathname_file) {  // point_clouds_IO.cpp:148
  std::vector<std::string> v;
  std::ifstream f(pathname_file);
  std::string line;
  while (std::getline(f, line)) {
    while (!line.empty() && (line.back() == '\r' || line.back() == '\n')) line.pop_back();  // the shipped task files have CRLF ends
    v.push_back(line);
  }
  return v;
}

#endif  // POINT_CLOUDS_IO_HPP_
