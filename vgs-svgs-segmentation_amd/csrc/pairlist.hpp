// pairlist.hpp -- per-voxel lists of ALL heavy pairs inside the search ball ("pair lists"), shared by pairlist.hip (builder)
// and localcut_pg.hpp (consumer).
//
// The weight of a pair depends on the pair only (VS:1597-1740), yet the reference -- and every class of the
// local cut that evaluates weights itself -- computes it again in every neighbourhood that holds both voxels: about 60 times at a
// ball of five voxels, several hundred times at ten.  The near-pair lists (nearlist.hpp) remove that for the pairs at most two
// lattice steps apart, which is all a smooth surface at r = 5 needs.  Everything that leaves that regime -- neighbourhoods in
// clutter or under range noise, where no segment freezes early and every heavy pair is wanted; balls of ten voxels, where the
// scan runs past two lattice steps; solid volumes with thousands of neighbours -- gets its weights from HERE instead:
//
//   for a used voxel a, every used voxel b of a's adjacency row (centre distance below graph_size: the FLANN predicate of
//   adjacency.hip) at a lexicographically POSITIVE lattice offset (dz, dy, dx) with
//       w(a, b) > 1 - cut   or   w(b, a) > 1 - cut          (either orientation: the weight is symmetric only up to an ulp)
//   as one 16-byte entry (w(a, b), w(b, a), packed lattice offset of b from a, id of b), the entries of a voxel sorted by
//   key = max(w(a, b), w(b, a)) DESCENDING, stored in one contiguous chunk of a shared pool (pl_idx[a] = start, count).
//
// Sorted by weight, a list is read as a prefix: "every pair heavier than L" is the entries in front of the first key <= L, so
// a cut takes its edges in exact bands of descending weight -- no distance shells, no bound between distance and weight, no
// edges evaluated and then carried because they turned out lighter than the shell's level.  A pair is evaluated ONCE per step.
//
// Rows are built on demand (k_pl_mark / k_pl_worklist): the rows of the voxels that belong to a neighbourhood the pair-list
// kernel is going to cut.  The bulk class of a smooth scene never asks for them.
//
// What the lists do NOT hold: (1) pairs at or below 1 - cut (phase B of the cut evaluates the few it needs); (2) pairs of one
// neighbourhood that are not in each other's ball (centre distance >= graph_size).  Their centroids are at least
// graph_size - sqrt(3) * voxel_size apart (a listed voxel's centroid lies in its cube: the builder checks it, as nearlist.hip
// does), so by fact (U) of localcut_wave.hpp they weigh at most PairLists::w_ring: bands above that level are complete from
// the lists alone, and a cut that has to go below it evaluates those "ring" pairs itself, once.
#ifndef PAIRLIST_HPP_
#define PAIRLIST_HPP_

#include <stdint.h>

#define PL_NOT_BUILT 0xffffffffu   // pl_idx[v].y: no list yet (the state every run starts from)
#define PL_UNUSABLE 0xfffffffeu    // the row cannot have one (no lattice offsets, centroid outside its cube, pool exhausted)
#define PL_CUBE_TOL 0.9e-3f        // as NL_CUBE_TOL: a centroid may sit this many voxel sizes outside its cube

struct PairLists {
  const uint2* idx;       // [V] x = first entry, y = number of entries or PL_NOT_BUILT / PL_UNUSABLE
  const float4* ent;      // (w(a, b), w(b, a), bits: packed offset of b from a as the rows' adj_off holds it, bits: voxel id of b)
  const uint8_t* any;     // [V] 1 = the voxel is part of at least one heavy pair inside its ball (0xff = none)
  float w_ring;           // every pair of voxels NOT in each other's ball weighs at most this (+inf: no such bound)
};

// A launch that is queued before anybody knows whether it is wanted: it looks at ONE word on the device and returns at once when the
// word does not hold the value it was queued for.  The word is written once per run by k_ho_lists (localcut.hip) when the hand-over lists
// of the one-wavefront classes are complete: LC_FEW -- the dense kernel of localcut_dense.hpp evaluates the hand-overs' pairs itself and
// crossValidation's first pass runs beside it; LC_MANY (more than 1 / pg_min_frac of the used voxels) -- the pair lists are built,
// k_localcut_pg reads them, and crossValidation waits for them (its first pass would put off every row).  word == null: always open.
#define LC_FEW 1u
#define LC_MANY 2u
struct LcGate { const unsigned int* word; unsigned int want; };
#if defined(__HIPCC__)
__device__ __forceinline__ bool lc_gate_open(const LcGate& g) { return g.word == nullptr || *g.word == g.want; }
// ... for a launch that may run BEFORE the word is written and is harmless when it ran in vain (crossValidation's first pass): undecided counts as LC_FEW
__device__ __forceinline__ bool lc_gate_open_early(const LcGate& g) { return g.word == nullptr || __hip_atomic_load(g.word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != LC_MANY; }
#endif

#endif
