// brick_table.hpp -- "which voxel sits in lattice cell (x, y, z)?" answered from an L2-resident table.
// Shared by the adjacency stage (adjacency.hip) and the supervoxel stage (vccs.hip).
#ifndef BRICK_TABLE_HPP_
#define BRICK_TABLE_HPP_

#include "vgs_context.hpp"

// Voxels are sorted by (descending) x-major Morton code, so the voxels of one 4x4x4 brick (code >> 6) are contiguous
// in the voxel array.  One 32-byte entry per brick -- key, occupancy mask, used mask, id of the brick's first voxel --
// answers "which voxel sits in lattice cell c, and is it used?" with one hash probe:
//     id(c) = first + popcount(occupancy >> (local + 1))        (ids ascend while the local code descends)
// About V/9 bricks: the table is a few MB and stays in the 4 MB L2 of each XCD, where the per-voxel hash (24 MB at
// 10 M points) was served from the fabric, and the used flag no longer costs a 64-byte node read per neighbour.
struct Brick { unsigned long long key; unsigned long long occ; unsigned long long used; uint32_t first; uint32_t pad; };

// the packed brick coordinates are highly regular: fold the three fields with odd multipliers before the final multiply
__device__ __forceinline__ uint32_t hash_slot(uint64_t key, uint32_t hbits) {
  const uint32_t bx = (uint32_t)key & 0x1fffffu, by = (uint32_t)(key >> 21) & 0x1fffffu, bz = (uint32_t)(key >> 42);
  uint32_t h = bx * 0x9E3779B1u + by * 0x85EBCA77u + bz * 0xC2B2AE3Du;
  h ^= h >> 15;
  return (h * 0x2C1B3C6Du) >> (32 - hbits);
}

// A brick is named by its lattice coordinates (voxel key >> 2 per axis) packed 21 bits each, + 1 so that 0 means empty;
// a voxel's bit inside the brick is the low 6 bits of its Morton code (z0 y0 x0 z1 y1 x1 from bit 0).  Both are cheap
// to form from (nx, ny, nz): the neighbour search never spreads a full Morton code.
__device__ __forceinline__ unsigned long long brick_key(uint32_t nx, uint32_t ny, uint32_t nz) {
  return (((unsigned long long)(nz >> 2) << 42) | ((unsigned long long)(ny >> 2) << 21) | (unsigned long long)(nx >> 2)) + 1ull;
}
__device__ __forceinline__ int brick_local(uint32_t nx, uint32_t ny, uint32_t nz) {
  return (int)((nz & 1u) | ((ny & 1u) << 1) | ((nx & 1u) << 2) | ((nz & 2u) << 2) | ((ny & 2u) << 3) | ((nx & 2u) << 4));
}

// Every brick gets a slot with the OR of its voxels' bits; the brick's first voxel is the smallest id.  Voxels are
// sorted by Morton code, so the voxels of a brick are neighbours in the array: a wavefront first ORs the bits of each run
// of equal brick keys among its lanes (segmented scan, six shuffle steps) and only the last lane of a run goes to the
// table -- about nine times fewer probes and atomics than one insert per voxel.
static __global__ void k_brick_insert(const uint64_t* __restrict__ vox_code, const NodeRec* __restrict__ node, int64_t V,
                               Brick* __restrict__ table, uint32_t hbits) {
  const int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int lane = threadIdx.x & 63;
  unsigned long long key = 0ull, occ = 0ull, usd = 0ull;   // key 0 = no voxel in this lane
  uint32_t first = 0xffffffffu;
  if (v < V) {
    const uint64_t code = vox_code[v];
    key = brick_key(vm_compact21(code >> 2), vm_compact21(code >> 1), vm_compact21(code));
    occ = 1ull << (code & 63ull);  // == brick_local of the same coordinates
    if (node && (node[v].flags & VGS_F_EIG)) usd = occ;
    first = (uint32_t)v;
  }
  for (int o = 1; o < 64; o <<= 1) {
    const unsigned long long ko = __shfl_up(key, o, 64), oo = __shfl_up(occ, o, 64), uo = __shfl_up(usd, o, 64);
    const uint32_t fo = (uint32_t)__shfl_up((int)first, o, 64);
    if (lane >= o && ko == key) { occ |= oo; usd |= uo; first = fo < first ? fo : first; }   // runs are contiguous: equal keys o lanes apart span one run
  }
  const unsigned long long knext = __shfl_down(key, 1, 64);
  if (key == 0ull || (lane < 63 && knext == key)) return;   // not the last lane of its run
  const uint32_t mask = (1u << hbits) - 1u;
  uint32_t s = hash_slot(key, hbits);
  while (true) {
    const unsigned long long prev = atomicCAS(&table[s].key, 0ull, key);
    if (prev == 0ull || prev == key) break;
    s = (s + 1) & mask;
  }
  atomicOr(&table[s].occ, occ);
  if (usd) atomicOr(&table[s].used, usd);
  atomicMin(&table[s].first, first);
}

// voxel id in lattice cell (nx, ny, nz), -1 if empty; *is_used tells whether that voxel has > points_min points
__device__ __forceinline__ int brick_find(const Brick* __restrict__ table, uint32_t hbits, uint32_t nx, uint32_t ny, uint32_t nz, bool* is_used) {
  const unsigned long long key = brick_key(nx, ny, nz);
  const uint32_t mask = (1u << hbits) - 1u;
  uint32_t s = hash_slot(key, hbits);
  while (true) {
    const unsigned long long k = table[s].key;
    if (k == key) break;
    if (k == 0ull) return -1;
    s = (s + 1) & mask;
  }
  const unsigned long long occ = table[s].occ;
  const int local = brick_local(nx, ny, nz);
  if (!((occ >> local) & 1ull)) return -1;
  *is_used = ((table[s].used >> local) & 1ull) != 0;
  const unsigned long long above = (local == 63) ? 0ull : (occ >> (local + 1));
  return (int)(table[s].first + (uint32_t)__popcll(above));
}


// builds the table for the context's current voxel set (vox_code, V) in c->hkey; node may be null (no used mask)
// the empty table: key / occupancy / used = 0, first = 0xffffffff (a minimum over the brick's voxel ids) -- one kernel writing whole 32-byte
// entries (round 5: was a fill plus a strided 2-D fill of the `first` words, 25 us in front of k_brick_insert at the bench scene)
static __global__ void k_brick_clear(Brick* __restrict__ table, size_t H) {
  const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= H) return;
  ulonglong2* p = (ulonglong2*)&table[e];
  p[0] = make_ulonglong2(0ull, 0ull);
  p[1] = make_ulonglong2(0ull, 0x00000000ffffffffull);   // used = 0; first = 0xffffffff, pad = 0
}

static inline vgs_status vgs_build_bricks(vgs_ctx* c, const NodeRec* node) {
  const int64_t V = c->V;
  // the number of bricks is not known without a pass, V/4 slots would already be generous; size by V/2
  uint32_t hbits = 4;
  while ((1ull << hbits) < (uint64_t)(V / 2 + 16)) ++hbits;
  c->hbits = hbits;
  const size_t H = (size_t)1 << hbits;
  VGS_HIP_TRY(c, c->hkey.ensure(H * (sizeof(Brick) / 8)));
  hipLaunchKernelGGL(k_brick_clear, dim3((unsigned)((H + 255) / 256)), dim3(256), 0, c->stream, (Brick*)c->hkey.p, H);
  hipLaunchKernelGGL(k_brick_insert, dim3((unsigned)((V + 255) / 256)), dim3(256), 0, c->stream, c->vox_code.p, node, V, (Brick*)c->hkey.p, hbits);
  return VGS_OK;
}

#endif  // BRICK_TABLE_HPP_
