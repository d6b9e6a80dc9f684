// nearlist.hpp -- near-pair lists shared by nearlist.hip (builder) and the local cut (consumer).
//
// The local cut of voxel i (localcut_wave.hpp) examines, in its first shell, every pair (a, b) of i's neighbours whose
// centroids are closer than about 1.5 voxels -- the same few pairs for every i that has both in its neighbourhood, about
// 70 times each on a surface.  Voxels sit on a lattice, so these pairs are known per VOXEL: for every used voxel a the
// builder lists its used neighbours b at most NL_REACH lattice steps away (Chebyshev) with
//     centroid distance^2  d2(a, b) < (NL_REACH * voxel_size)^2   and   w(a, b) > 1 - cut or w(b, a) > 1 - cut,
// sorted by d2, as (d2, w(a, b), w(b, a), packed lattice offset) -- and every pair ONCE (round 3): in the list of the voxel from
// which the other lies at a lexicographically positive offset (dz, dy, dx).  The weight is symmetric only up to an ulp and the
// cut needs w(first, second) in the order of i's adjacency row, so an entry carries both orientations and the reader picks.
// (Rounds 2 and early 3 kept each pair in both voxels' lists, one orientation each; a reader then found half of the entries it
// looked at pointing the wrong way.  A voxel with a complete neighbourhood has exactly half of its near partners at positive
// offsets -- four of eight on a surface --, so five lanes per vertex read the first shell where nine were needed.)
// The cut then walks the lists of its vertices instead of testing all n^2/2 pairs, and takes the stored weight instead of
// evaluating it.
// Exactness: a listed voxel's centroid lies inside its voxel's cube widened by NL_CUBE_TOL voxel sizes (the builder checks
// it, counting the rounding of the float cube centre as well; a voxel that fails -- float sums of coordinates kilometres from
// the origin -- gets NL_NONE), so two listed voxels more than R lattice steps apart on some axis are farther apart
// than (R - 2 * NL_CUBE_TOL) voxel sizes.  The builder finds a voxel's partners in its adjacency row, i.e. inside the search
// ball: the lists are complete up to R steps only if every offset of at most R steps lies strictly inside the ball,
// 3 R^2 < (graph_size / voxel_size)^2 (nearlist.hip: nl_reach_steps, R <= NL_REACH).  Every pair with
// d2 < d2max = (R * voxel_size)^2 * slack is then in the lists, and the cut reads them no farther (entries beyond d2max
// are never taken: the general enumeration covers d2 >= d2max).  Pairs at or below the singleton threshold 1 - cut are
// never stored by phase A of the cut either (fact S).
#ifndef NEARLIST_HPP_
#define NEARLIST_HPP_

#include <stdint.h>

#ifndef NL_S
#define NL_S 16
#endif
// NL_S: entries per voxel (pairs at positive offsets); a voxel with more is marked NL_NONE and its neighbourhoods take the general path
#define NL_REACH 2    // Chebyshev reach of the lists in lattice steps
#define NL_NONE 0xffu
#define NL_CUBE_TOL 0.9e-3f   // centroid may sit this many voxel sizes outside its cube (per axis) and still be listed
#define NL_D2_SLACK 0.998f    // (1 - 2 * 1e-3 / NL_REACH)^2 rounded down: what the tolerance above costs in reach
#define NL_BALL 5     // largest lattice offset of a neighbour the consumer's offset map can hold ((2*5+1)^3 bytes of LDS)

struct NearLists {
  const uint8_t* cnt;    // [V]        number of entries, NL_NONE = no list
  const uint32_t* tot;   // [V]        heavy near pairs the voxel is part of, in either voxel's list (scheduling only)
  const float4* ent;     // [V * NL_S] (squared centroid distance, vm_pair_weight(a, b), vm_pair_weight(b, a), bits of the lattice
                         //            offset of b from a: (dx+2) | (dy+2) << 4 | (dz+2) << 8), ascending distance; unused entries
                         //            hold d2 = +inf, so a reader needs no count: "d2 < shell radius" ends the list; a voxel
                         //            without a list (NL_NONE) has a NaN distance in entry 0
  float d2max;           // lists are complete for shells up to this squared centroid distance
  int enabled;           // the lists exist
  int direct;            // the search ball fits the one-wavefront classes' direct offset map (NL_BALL)
};

// signed difference of two 10-bit lattice coordinates (exact for |difference| < 512)
#if defined(__HIPCC__)
__device__ __forceinline__ int nl_diff10(uint32_t a, uint32_t b) { return (int)(((a - b + 512u) & 1023u)) - 512; }
#endif

#endif
