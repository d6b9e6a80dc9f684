// regsort.hpp -- descending sort of up to 64*K 64-bit keys of ONE wavefront with the keys in registers: a bitonic network whose
// comparators never touch LDS.  Key e lives in register e % K of lane e / K, so the strides used most (1 .. K/2: every merge
// ends with them) are compare-exchanges between registers of one lane, lane strides 1, 2, 4, 8 are DPP moves inside a row of
// 16 lanes (quad_perm, two bank-masked row shifts for 4, row_ror:8), and the two widest -- used once or twice per sort -- are
// gfx950's v_permlane16_swap / v_permlane32_swap.  The LDS network it replaces in the one-wavefront classes of the local cut
// reads and writes every key once per stage (36 stages for 256 keys: a third of the kernel's LDS instructions, all 8 bytes wide);
// this one loads the list once and stores it once.  Slots at and behind cnt are the key 0, below every real key (real keys
// carry weight bits > 0 in the upper word; dropped entries ARE 0 and end up behind the real ones, as in the LDS network).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#ifndef REGSORT_OPAQUE_LANE
#define REGSORT_OPAQUE_LANE 1
#endif

namespace regsort {

// lanes whose bit `b` is clear
__host__ __device__ constexpr uint64_t lanes_bit_clear(int b) {
  uint64_t m = 0;
  for (int l = 0; l < 64; ++l) if (((l >> b) & 1) == 0) m |= 1ull << l;
  return m;
}

template <int CTRL, int BANK = 0xf>
__device__ __forceinline__ uint32_t dpp(uint32_t old, uint32_t src) {
  return (uint32_t)__builtin_amdgcn_update_dpp((int)old, (int)src, CTRL, 0xf, BANK, false);
}
template <int CTRL>   // every lane is written: no previous value to keep (saves the copy update_dpp needs for it)
__device__ __forceinline__ uint32_t dpp_all(uint32_t src) {
  return (uint32_t)__builtin_amdgcn_mov_dpp((int)src, CTRL, 0xf, 0xf, false);
}

// value of the key held by lane ^ (1 << J), J = 0..3
template <int J>
__device__ __forceinline__ uint64_t other_lane(uint64_t x) {
  uint32_t lo = (uint32_t)x, hi = (uint32_t)(x >> 32), olo, ohi;
  if constexpr (J == 0) { olo = dpp_all<0xB1>(lo); ohi = dpp_all<0xB1>(hi); }          // quad_perm [1,0,3,2]
  else if constexpr (J == 1) { olo = dpp_all<0x4E>(lo); ohi = dpp_all<0x4E>(hi); }     // quad_perm [2,3,0,1]
  else if constexpr (J == 2) {                                                         // lanes 0-3, 8-11 of a row read l + 4, the others l - 4
    olo = dpp_all<0x104>(lo); olo = dpp<0x114, 0xa>(olo, lo);
    ohi = dpp_all<0x104>(hi); ohi = dpp<0x114, 0xa>(ohi, hi);
  } else { olo = dpp_all<0x128>(lo); ohi = dpp_all<0x128>(hi); }                       // row_ror:8
  return ((uint64_t)ohi << 32) | olo;
}

// One stage on lane stride 1 << J: `keepmax` = this lane keeps the larger key of the pair.
template <int K, int J>
__device__ __forceinline__ void lane_stage(uint64_t (&k)[K], bool keepmax) {
#pragma unroll
  for (int r = 0; r < K; ++r) {
    if constexpr (J <= 3) {
      const uint64_t o = other_lane<J>(k[r]);
      const bool take = (o > k[r]) == keepmax;
      k[r] = take ? o : k[r];
    } else {
      // the swap leaves the key of the lower lane of the pair in a, of the upper lane in b -- in both lanes
      const uint32_t lo = (uint32_t)k[r], hi = (uint32_t)(k[r] >> 32);
      uint32_t alo, blo, ahi, bhi;
      if constexpr (J == 4) {
        auto s = __builtin_amdgcn_permlane16_swap(lo, lo, false, false); alo = s[0]; blo = s[1];
        auto t = __builtin_amdgcn_permlane16_swap(hi, hi, false, false); ahi = t[0]; bhi = t[1];
      } else {
        auto s = __builtin_amdgcn_permlane32_swap(lo, lo, false, false); alo = s[0]; blo = s[1];
        auto t = __builtin_amdgcn_permlane32_swap(hi, hi, false, false); ahi = t[0]; bhi = t[1];
      }
      const uint64_t a = ((uint64_t)ahi << 32) | alo, b = ((uint64_t)bhi << 32) | blo;
      const bool take_b = (b > a) == keepmax;
      k[r] = take_b ? b : a;
    }
  }
}

// One stage of the bitonic merge of blocks of 2^S keys, stride 2^J.  Blocks whose index bit S is clear sort descending, the
// others ascending; the last merge (S == LOGN) is descending everywhere.
template <int K, int LOGK, int LOGN, int S, int J>
__device__ __forceinline__ void stage(uint64_t (&k)[K], int lane) {
  if constexpr (J >= LOGK) {
    // lane stride 2^(J - LOGK); descending block <=> bit (S - LOGK) of the lane clear
    constexpr int JB = J - LOGK;
    const bool lower = ((lane >> JB) & 1) == 0;
    bool desc = true;
    if constexpr (S != LOGN) desc = ((lane >> (S - LOGK)) & 1) == 0;
    lane_stage<K, JB>(k, lower == desc);
  } else {
    // register stride 2^J: pairs (r, r + 2^J) with bit J of r clear
#pragma unroll
    for (int r = 0; r < K; ++r) {
      if ((r >> J) & 1) continue;
      const int r2 = r + (1 << J);
      bool desc = true;
      if constexpr (S == LOGN) desc = true;
      else if constexpr (S < LOGK) desc = ((r >> S) & 1) == 0;   // known at compile time once unrolled
      else desc = ((lane >> (S - LOGK)) & 1) == 0;
      const bool swap = (k[r2] > k[r]) == desc;
      const uint64_t x = k[r], y = k[r2];
      k[r] = swap ? y : x;
      k[r2] = swap ? x : y;
    }
  }
}

template <int K, int LOGK, int LOGN, int S, int J>
struct Strides {
  static __device__ __forceinline__ void run(uint64_t (&k)[K], int lane) {
    stage<K, LOGK, LOGN, S, J>(k, lane);
    Strides<K, LOGK, LOGN, S, J - 1>::run(k, lane);
  }
};
template <int K, int LOGK, int LOGN, int S>
struct Strides<K, LOGK, LOGN, S, -1> {
  static __device__ __forceinline__ void run(uint64_t (&)[K], int) {}
};

template <int K, int LOGK, int LOGN, int S>
struct Net {
  static __device__ __forceinline__ void run(uint64_t (&k)[K], int lane) {
    Net<K, LOGK, LOGN, S - 1>::run(k, lane);
    Strides<K, LOGK, LOGN, S, S - 1>::run(k, lane);
  }
};
template <int K, int LOGK, int LOGN>
struct Net<K, LOGK, LOGN, 0> {
  static __device__ __forceinline__ void run(uint64_t (&)[K], int) {}
};

constexpr int ilog2(int v) { int l = 0; while ((1 << l) < v) ++l; return l; }

__device__ __forceinline__ void lds_fence() {   // orders this wavefront's LDS accesses (one wavefront owns the list)
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
  __builtin_amdgcn_wave_barrier();
}

// Sorts lk[0, cnt) (LDS, cnt <= 64 * K) descending.  The caller orders the list's earlier writes before the call and its later
// reads after it (lds_fence).  FINAL_MERGE_ONLY: lk[0, cnt) is already bitonic (the last merge of the network alone).
template <int K, bool FINAL_MERGE_ONLY = false>
__device__ __forceinline__ void sort_desc(uint64_t* lk, int cnt, int lane) {
  constexpr int LOGK = ilog2(K), LOGN = LOGK + 6;
  uint64_t k[K];
#if REGSORT_OPAQUE_LANE
  // the stages' lane predicates depend on the lane alone: hoisted out of the caller's loops they would sit in forty SGPR pairs
  // for the whole kernel; behind this they are recomputed where they are used (two or three instructions per stage)
  asm volatile("" : "+v"(lane));
#endif
#pragma unroll
  for (int r = 0; r < K; ++r) { const int e = lane * K + r; k[r] = e < cnt ? lk[e] : 0ull; }
  if constexpr (FINAL_MERGE_ONLY) Strides<K, LOGK, LOGN, LOGN, LOGN - 1>::run(k, lane);
  else Net<K, LOGK, LOGN, LOGN>::run(k, lane);
#pragma unroll
  for (int r = 0; r < K; ++r) { const int e = lane * K + r; if (e < cnt) lk[e] = k[r]; }
}

// ---- the same sort on 32-bit keys (round 6) -------------------------------------------------------------------------------------------
// A comparator of the network above is v_cmp_gt_u64 and four v_cndmask_b32 -- five instructions of the half-rate class (tools/valu_rate.hip:
// 4.1 cycles each against 2.2 for full-rate ones), two DPP moves more across lanes; 87 % of the sort's instructions.  The lists of phase A
// hold weights in (thr0, 1] with thr0 >= 0.5: 2^23 float values at most, so  key32 = (weight bits - wbase) << 9 | (511 - slot)  orders 512
// slots by weight in ONE word -- and a comparator on one word is v_max_u32 + v_min_u32, full rate, with the DPP move folded into either.
// Directions: instead of selecting "keep the larger / the smaller" per block (a select per key and stage), the lanes of an ascending block
// hold their keys COMPLEMENTED while it is one -- sorting ~k descending is sorting k ascending -- so every lane-dependent comparator runs
// one way; the complement moves once per merge level (one v_xor per key).  What is left of the half-rate class is one select per key and
// cross-lane stage (the lower lane of a pair keeps the maximum, the upper one the minimum).
// The 64-bit keys come back through the slots: position i takes lk[slot of the i-th key32].  Keys of equal weight are then in slot order,
// not in the order of their low words: runs of equal high words (rare: two weights of one neighbourhood with the same 23 bits) are put
// right by adjacent exchanges in registers until none is left -- the result is the descending order of the full 64-bit keys, exactly what
// sort_desc gives.  A key outside the window (never in phase A; the check is one compare per key) sends the list to sort_desc.
template <int J>
__device__ __forceinline__ uint32_t other_lane32(uint32_t x) {
  if constexpr (J == 0) return dpp_all<0xB1>(x);                                       // quad_perm [1,0,3,2]
  else if constexpr (J == 1) return dpp_all<0x4E>(x);                                  // quad_perm [2,3,0,1]
  else if constexpr (J == 2) { uint32_t o = dpp_all<0x104>(x); return dpp<0x114, 0xa>(o, x); }
  else return dpp_all<0x128>(x);                                                       // row_ror:8
}
// lane stride 1 << J, every block descending in what the lanes hold: the lower lane of a pair keeps the larger key
template <int K, int J>
__device__ __forceinline__ void lane_stage32(uint32_t (&k)[K], bool lower) {
#pragma unroll
  for (int r = 0; r < K; ++r) {
    uint32_t a, b;
    if constexpr (J <= 3) { a = k[r]; b = other_lane32<J>(k[r]); }
    else if constexpr (J == 4) { auto s = __builtin_amdgcn_permlane16_swap(k[r], k[r], false, false); a = s[0]; b = s[1]; }
    else { auto s = __builtin_amdgcn_permlane32_swap(k[r], k[r], false, false); a = s[0]; b = s[1]; }
    const uint32_t mx = a > b ? a : b, mn = a > b ? b : a;   // v_max_u32 / v_min_u32
    k[r] = lower ? mx : mn;
  }
}
template <int K, int LOGK, int LOGN, int S, int J>
__device__ __forceinline__ void stage32(uint32_t (&k)[K], int lane) {
  if constexpr (J >= LOGK) {
    lane_stage32<K, J - LOGK>(k, ((lane >> (J - LOGK)) & 1) == 0);
  } else {
#pragma unroll
    for (int r = 0; r < K; ++r) {
      if ((r >> J) & 1) continue;
      const int r2 = r + (1 << J);
      // S < LOGK: the block's direction is a bit of the register index, known here; otherwise descending (complemented where ascending)
      const bool desc = (S >= LOGK) || (((r >> S) & 1) == 0);
      const uint32_t x = k[r], y = k[r2];
      const uint32_t mx = x > y ? x : y, mn = x > y ? y : x;
      k[r] = desc ? mx : mn;
      k[r2] = desc ? mn : mx;
    }
  }
}
template <int K, int LOGK, int LOGN, int S, int J>
struct Strides32 {
  static __device__ __forceinline__ void run(uint32_t (&k)[K], int lane) {
    stage32<K, LOGK, LOGN, S, J>(k, lane);
    Strides32<K, LOGK, LOGN, S, J - 1>::run(k, lane);
  }
};
template <int K, int LOGK, int LOGN, int S>
struct Strides32<K, LOGK, LOGN, S, -1> {
  static __device__ __forceinline__ void run(uint32_t (&)[K], int) {}
};
template <int K, int LOGK, int LOGN, int S>
struct Net32 {
  static __device__ __forceinline__ void run(uint32_t (&k)[K], int lane) {
    Net32<K, LOGK, LOGN, S - 1>::run(k, lane);
    if constexpr (S >= LOGK) {
      // the lanes whose block changes direction between level S - 1 and level S complement their keys: ascending at level L (LOGK <= L <
      // LOGN) are the lanes with bit L - LOGK set; below LOGK and at LOGN nobody is
      int f = 0;
      if constexpr (S > LOGK && S > 1) f ^= lane >> (S - 1 - LOGK);   // (level 0, single keys, has no direction)
      if constexpr (S < LOGN) f ^= lane >> (S - LOGK);
      const uint32_t m = 0u - (uint32_t)(f & 1);
#pragma unroll
      for (int r = 0; r < K; ++r) k[r] ^= m;
    }
    Strides32<K, LOGK, LOGN, S, S - 1>::run(k, lane);
  }
};
template <int K, int LOGK, int LOGN>
struct Net32<K, LOGK, LOGN, 0> {
  static __device__ __forceinline__ void run(uint32_t (&)[K], int) {}
};

// lk[0, cnt) descending, cnt <= 64 * K <= 512, as sort_desc<K> leaves it.  wbase: every non-zero key's high word is expected in
// [wbase, wbase + 2^23).  Returns false -- list untouched -- when a key lies outside that window (uniform): the caller takes sort_desc.
// FIX_TIES false: the low words of equal high words already descend with the slot (pairlist.hip: the low word IS the complemented slot).
template <int K, bool FIX_TIES = true>
__device__ __forceinline__ bool sort_desc32(uint64_t* lk, int cnt, int lane, uint32_t wbase) {
  constexpr int LOGK = ilog2(K), LOGN = LOGK + 6;
  static_assert(64 * K <= 512, "nine bits of slot");
  uint32_t k[K];
#if REGSORT_OPAQUE_LANE
  asm volatile("" : "+v"(lane));
#endif
  bool outside = false;
#pragma unroll
  for (int r = 0; r < K; ++r) {
    const int e = lane * K + r;
    const uint32_t hi = e < cnt ? ((const uint32_t*)lk)[2 * e + 1] : 0u;   // (a dropped entry is 0: its high word is)
    const uint32_t d = hi - wbase;
    outside = outside || (hi != 0u && d >= (1u << 23));
    k[r] = hi != 0u ? ((d << 9) | (uint32_t)(511 - e)) : 0u;
  }
  if (__ballot(outside) != 0ull) return false;
  Net32<K, LOGK, LOGN, LOGN>::run(k, lane);
  // does any key carry the weight of the next one?  (uniform answer; where they are is found again in LDS by the rare list that has some:
  // a mark per position kept in a register across the write-back cost the bulk kernel of the local cut four spilled registers)
  bool tie = false;
  if constexpr (FIX_TIES) {
#pragma unroll
    for (int r = 0; r + 1 < K; ++r) tie = tie || (k[r + 1] != 0u && ((k[r] ^ k[r + 1]) >> 9) == 0u);
    const uint32_t nxt = (uint32_t)__shfl_down((int)k[0], 1, 64);
    tie = tie || (lane < 63 && nxt != 0u && ((k[K - 1] ^ nxt) >> 9) == 0u);
    tie = __ballot(tie) != 0ull;
  }
  {
    // the low words come back through the slots; the high word of a key is in its key32 (no second register per key)
    uint32_t lo[K];
#pragma unroll
    for (int r = 0; r < K; ++r) lo[r] = k[r] != 0u ? ((const uint32_t*)lk)[2 * (511 - (int)(k[r] & 511u))] : 0u;
    lds_fence();   // every slot has been read
#pragma unroll
    for (int r = 0; r < K; ++r) {
      const int e = lane * K + r;
      if (e < cnt) lk[e] = k[r] != 0u ? (((uint64_t)(wbase + (k[r] >> 9)) << 32) | lo[r]) : 0ull;
    }
  }
  if constexpr (FIX_TIES) {
    if (tie) {
      // adjacent exchanges inside the runs of equal high words, even pairs then odd pairs, until a pass moves nothing (runs are two or
      // three keys long)
      bool moved = true;
      while (__ballot(moved) != 0ull) {
        moved = false;
#pragma unroll
        for (int par = 0; par < 2; ++par) {
          lds_fence();
          for (int e = 2 * lane + par; e + 1 < cnt; e += 128) {
            const uint64_t x = lk[e], y = lk[e + 1];
            if ((uint32_t)(x >> 32) == (uint32_t)(y >> 32) && x < y) { lk[e] = y; lk[e + 1] = x; moved = true; }
          }
        }
      }
      lds_fence();
    }
  }
  return true;
}

// 64*K < cnt <= 128*K keys with the register budget of K per lane: both halves sorted descending one after the other, the first
// step of their merge as a mirror step through LDS (slot i against slot 128*K - 1 - i; slots at and behind cnt are keys below
// every real one and never move, so the list need not reach 128*K slots), then the rest of the merge on each half in registers.
template <int K>
__device__ __forceinline__ void sort_desc_two_halves(uint64_t* lk, int cnt, int lane) {
  constexpr int H = 64 * K;
#pragma nounroll
  for (int h = 0; h < 2; ++h) sort_desc<K>(lk + H * h, h ? cnt - H : H, lane);
  lds_fence();
#pragma unroll
  for (int q = 0; q < K; ++q) {
    const int i = lane + 64 * q, j = 2 * H - 1 - i;
    if (j < cnt) {
      const uint64_t x = lk[i], y = lk[j];
      if (x < y) { lk[i] = y; lk[j] = x; }
    }
  }
  lds_fence();
#pragma nounroll
  for (int h = 0; h < 2; ++h) sort_desc<K, true>(lk + H * h, h ? cnt - H : H, lane);
}

// lk[0, cnt) of a workgroup of NWAVES wavefronts (cnt <= 512 * NWAVES; every thread calls, sync() is the workgroup barrier):
// wavefront w sorts keys [512 w, 512 w + 512) in registers; the merges of 1024, 2048 and 4096 keys then take their mirror step and
// their strides of 512 and more through LDS, all wavefronts together (the one-direction form: every comparator puts the larger key
// at the lower index, slots at and behind cnt never move), and the strides 256 .. 1 again in registers, every wavefront on its own
// 512 keys.  Against a network that runs every stage through LDS this is a dozen passes over the list and as many barriers instead
// of 66 (2048 keys).
template <int NWAVES, typename Sync>
__device__ __forceinline__ void sort_desc_block(uint64_t* lk, int cnt, int wave, int lane, Sync sync) {
  int np = 512;
  while (np < cnt) np <<= 1;
  const int tid = wave * 64 + lane;
  auto cmpx = [&](int lo, int hi) {
    if (hi >= cnt) return;
    const uint64_t x = lk[lo], y = lk[hi];
    if (x < y) { lk[lo] = y; lk[hi] = x; }
  };
  const int nblk = (cnt + 511) >> 9;
  const int mine = cnt - 512 * wave < 512 ? cnt - 512 * wave : 512;   // keys of this wavefront's block
  uint64_t* const blk_keys = lk + 512 * wave;
  sync();
  if (wave < nblk) {
    if (mine <= 64) sort_desc<1>(blk_keys, mine, lane);
    else if (mine <= 128) sort_desc<2>(blk_keys, mine, lane);
    else if (mine <= 256) sort_desc<4>(blk_keys, mine, lane);
    else sort_desc<8>(blk_keys, mine, lane);
  }
  sync();
  for (int size = 1024, sbit = 10; (size >> 1) < cnt; size <<= 1, ++sbit) {
    for (int t = tid; t < (np >> 1); t += 64 * NWAVES) {
      const int blk = t >> (sbit - 1), i = t & ((size >> 1) - 1);
      cmpx((blk << sbit) + i, (blk << sbit) + size - 1 - i);
    }
    sync();
    for (int sl = sbit - 2; sl >= 9; --sl) {
      const int strd = 1 << sl;
      for (int t = tid; t < (np >> 1); t += 64 * NWAVES) {
        const int lo = ((t >> sl) << (sl + 1)) | (t & (strd - 1));
        cmpx(lo, lo + strd);
      }
      sync();
    }
    if (wave < nblk) sort_desc<8, true>(blk_keys, mine, lane);
    sync();
  }
}

}  // namespace regsort
