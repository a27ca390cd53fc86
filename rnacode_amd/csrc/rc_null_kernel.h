// rc_null_kernel.h -- device pieces shared by the kernels and the k_null<N-1, LDSC, EXACT> template.
// Included by rc_kernels.hip (native path) and by the rc_null_*.hip units, each of which
// instantiates k_null for a range of N-1 so that the ranges compile in parallel.
#pragma once
#include <hip/hip_runtime.h>

#include <type_traits>
#include <utility>

#include "rc_device.h"
#include "rc_launch.h"

namespace rc {

// ------------------------------------------------------------------------------------------
// shared device pieces
// ------------------------------------------------------------------------------------------

// The reference's MAX and MAX3 (score.h:28-29) are "x > y ? x : y": with a NaN operand they return y -- a NaN in y goes through, a NaN
// in x is dropped --, where v_max_f32 / fmaxf return the operand that is a number.  NaNs only exist in blocks whose background models
// are NaN (an alignment without one of the four nucleotides: HKY85 with a zero frequency, score.c:204-244): k_prep_lut flags those
// kFlagNan, and the kernels give them the reference's order of operands and its macro, a compare and a select per maximum, in a path
// of their own (SEM = true); every other block keeps the one-instruction maximum (found by tools/soak_parity.py, round 5:
// tests/data/soak_hss_positive_delta.json).
template <bool SEM> __device__ __forceinline__ float ref_max(float x, float y) {
  if constexpr (SEM) return x > y ? x : y;
  else return fmaxf(x, y);
}
template <bool SEM> __device__ __forceinline__ float ref_max3(float x, float y, float z) { return ref_max<SEM>(ref_max<SEM>(x, y), z); }

// One step of the 3-state max-plus recurrence for one sequence k (score.c:506-533).
// zc: 0 -> z = 0, 1 -> z = +1, 2 -> z = -1 (wave-uniform).
template <bool SEM = false>
__device__ __forceinline__ void pair_step(int zc, float sig, float Delta, float Omega, float omega,
                                          float &s0, float &s1, float &s2) {
  if (zc == 0) {
    s0 = s0 + sig;
    s1 = s1 + omega;
    s2 = s2 + omega;
  } else if (zc == 1) {
    const float n0 = ref_max<SEM>(s0 + Delta, s2 + Omega);
    const float n1 = ref_max<SEM>(s0 + Omega, s1 + Delta);
    const float n2 = ref_max<SEM>(s1 + Omega, s2 + Delta);
    s0 = n0; s1 = n1; s2 = n2;
  } else {
    const float n0 = ref_max<SEM>(s0 + Delta, s1 + Omega);
    const float n1 = ref_max<SEM>(s1 + Delta, s2 + Omega);
    const float n2 = ref_max<SEM>(s2 + Delta, s0 + Omega);
    s0 = n0; s1 = n1; s2 = n2;
  }
}

// getHSS's serial state machine for one entry (score.c:892-959), emission reduced to a callback.
struct ScanState {
  float cm;  // currMax
  int ss, se;
};

template <typename Emit>
__device__ __forceinline__ void scan_step(ScanState &st, float v, int a, int j, bool last, float tieThr, Emit emit) {
  if (v > 0.0f || last) {
    if ((st.cm > 0.0f && st.se < a) || last) {
      if (st.se - st.ss >= 2) emit(st);
      st.cm = v; st.ss = a; st.se = j;
    } else if (v > st.cm || (fabsf(v - st.cm) < tieThr && (j - a) >= (st.se - st.ss))) {
      st.cm = v; st.ss = a; st.se = j;
    }
  }
}

// reverse the three 2-bit fields of a codon index
// getHSS (score.c:864-974) over the rows [aLo, aHi) of one strand x frame, the rows' entries at row[(a - aLo) * sites + j]:
// the state machine only looks at positive entries and at the frame's final one (score.c:898); the wavefront finds them with a
// ballot and walks the set bits in order (every lane the same walk, the entry's value comes from its lane by a shuffle; lane 0
// writes the records).  No LDS: these kernels run beside k_null, whose workgroups hold all of a CU's LDS.
__device__ __forceinline__ void native_scan_rows(const float *rows, int aLo, int aHi, int sites, int s, int f, float tieThr, ScanState &st, int &n,
                                                 DevHss *out, int hssCap, int lane) {
  for (int a = aLo; a < aHi; a++) {
    for (int j0 = a; j0 < sites; j0 += kWave) {
      const int j = j0 + lane;
      const float v = (j < sites) ? rows[static_cast<size_t>(a - aLo) * sites + j] : 0.0f;
      const bool last = (a == sites - 1) && (j == sites - 1);
      unsigned long long m = __ballot((j < sites) && (v > 0.0f || last));
      while (m) {
        const int t = __builtin_ctzll(m);
        m &= m - 1ull;
        const int jj = j0 + t;
        const float vt = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), t));   // (t is wave-uniform: a v_readlane, not a ds_bpermute round trip per entry)
        scan_step(st, vt, a, jj, (a == sites - 1) && (jj == sites - 1), tieThr, [&](const ScanState &e) {
          if (out && lane == 0 && n < hssCap) out[n] = DevHss{s, f, e.ss, e.se, e.cm};
          n++;
        });
      }
    }
  }
}

__device__ __forceinline__ uint32_t codon_flip(uint32_t c) { return ((c & 3u) << 4) | (c & 12u) | ((c >> 4) & 3u); }

// ------------------------------------------------------------------------------------------
// k_null<NK, LDSC>: the null-distribution loop body, lane = sample
// ------------------------------------------------------------------------------------------

// Per-sequence registers as distinct scalar members (not an array): keeps every chain in its own
// VGPR across the wave-uniform z branches instead of a register tuple that is copied at each join.
template <int K> struct KReg { float s0, s1, s2; int lut; };
template <typename Seq> struct KRegs;
template <int... K> struct KRegs<std::integer_sequence<int, K...>> : KReg<K>... {
  template <int I> __device__ __forceinline__ KReg<I> &at() { return static_cast<KReg<I> &>(*this); }
};
template <int NK, typename F, int... K>
__device__ __forceinline__ void static_for_impl(F &&f, std::integer_sequence<int, K...>) {
  (f(std::integral_constant<int, K>{}), ...);
}
template <int NK, typename F> __device__ __forceinline__ void static_for(F &&f) {
  static_for_impl<NK>(static_cast<F &&>(f), std::make_integer_sequence<int, NK>{});
}

// x / NK, correctly rounded.  Power of two: exact scaling.  Otherwise Markstein's two-FMA
// correction of x * RN(1/NK), proven equal to IEEE division for every NK in 2..31 and every
// |x| in {0} U [2^-100, 2^100) by exhaustion (tools/verify_const_div.c).  Blocks whose score
// tables could leave that range are flagged by the host (DevBlock::exact_div) and are scored by
// the EXACT instantiation, which uses '/'.
template <int NK, bool EXACT> __device__ __forceinline__ float div_by_nk(float x, float nkf) {
  if constexpr ((NK & (NK - 1)) == 0) {
    return x * (1.0f / static_cast<float>(NK));
  } else if constexpr (EXACT) {
    return x / nkf;
  } else {
    constexpr float y = 1.0f / static_cast<float>(NK);
    const float q0 = x * y;
    const float r = __builtin_fmaf(-static_cast<float>(NK), q0, x);
    return __builtin_fmaf(r, y, q0);
  }
}

// getHSS's state machine as the null samples need it (score.c:892-959): only the best emitted score
// matters.  The serial rule per entry (a, j) with value v > 0 is
//   open segment ended before row a (segmentEnd < a, currMax > 0)  -> report it, start (v, a, j)
//   else v > currMax, or |v - currMax| < 1e-4 and j - a >= segmentEnd - segmentStart -> replace.
// Restated per row so that an entry costs 4 compares and 2 selects:
//  * A segment that ended before row a can no longer change, and it is reported either at the next
//    positive entry or at the frame's final entry -- so reporting it at the START of row a and
//    turning the state into "no segment" (currMax 0, which any positive v replaces through v > currMax)
//    yields the same set of reported values.
//  * Within row a the state is either carried in (segment (ss0, se0) with se0 >= a) or was set in
//    this row at some j' (segment (a, j')).  "j - a >= segmentEnd - segmentStart" is j >= se0 - ss0 + a
//    for the former and j >= j' (always true later in the row) for the latter: one threshold Q per
//    lane, kept as 2(se0 - ss0 + a) when carried in and 2j' + 1 when set in this row, compared with
//    2j + 1; its low bit tells at the end of the row which of the two happened.
//  * v <= currMax together with |v - currMax| < thr is fl(v - currMax) > -thr.
struct SampleScan {
  float cm;        // currMax
  float Q;         // see above, as a float (integers < 2^24 are exact): valid inside a row
  uint32_t se;     // segmentEnd, valid between rows
  uint32_t len;    // segmentEnd - segmentStart, valid between rows
};
__device__ __forceinline__ void sample_scan_row_begin(SampleScan &st, float &best, uint32_t a) {
  const bool done = (st.cm > 0.0f) & (st.se < a);                 // score.c:900
  best = (done & (st.len >= 2u) & (st.cm > best)) ? st.cm : best;  // minSegmentLength, score.c:902
  st.cm = done ? 0.0f : st.cm;
  st.len = done ? 0u : st.len;
  st.Q = static_cast<float>(2u * (st.len + a));
}
// every entry except the frame's final one; j2 = 2j + 1, wave-uniform but held in a VGPR as a float: the
// caller advances it with a double-rate v_add_f32 instead of copying a scalar counter per entry
// (Round 6 built the step with a rare-arm branch -- v > currMax and the band test fl(v - currMax) > -thr as two compares into scalar lane masks,
// the tie rule's other two compares only where some lane sits in the band (score.c:953-954), the two selects written out: same decisions, two
// v_cmp fewer in the common arm, a scalar and-not, compare and branch more.  Product builds A/B on the headline: 42.68 against 42.80 ms per launch --
// nothing, and two registers more in the one-row kernels (a wavefront per SIMD lost at 11, 23 and 24 rows).  Taken back; profiles/r06/ab_scan_tie_branch.txt.)
__device__ __forceinline__ void sample_scan_step(SampleScan &st, float v, float j2, float negTieThr) {
  const float d = v - st.cm;
  const bool upd = (v > st.cm) | ((v > 0.0f) & (d > negTieThr) & (st.Q <= j2));   // score.c:953-954
  st.cm = upd ? v : st.cm;
  st.Q = upd ? j2 : st.Q;
}
__device__ __forceinline__ void sample_scan_row_end(SampleScan &st, uint32_t a) {
  const uint32_t q = static_cast<uint32_t>(st.Q);
  const bool inrow = (q & 1u) != 0u;
  const uint32_t j = q >> 1;
  st.se = inrow ? j : st.se;
  st.len = inrow ? j - a : st.len;
}
// the final entry of a frame is entered unconditionally and always reports the open segment
__device__ __forceinline__ void sample_scan_last(const SampleScan &st, float &best) {
  best = ((st.len >= 2u) & (st.cm > best)) ? st.cm : best;
}

// Read-only inputs are separate __restrict__ kernel parameters (not members of the by-value
// argument block) so that the compiler may prove them unclobbered and fetch the wave-uniform
// block tables (z words, tree records, masks) with scalar loads.
// (The body is a device function so that two kernels can share it: k_null, and k_null_occ, the same code compiled for more
// wavefronts per SIMD -- see below.)
// MODE 1 and 2 (tiny batches; see "rows split over workgroups" below) cut an item's two phases into pieces that many workgroups take:
// MODE 2 simulates a RANGE OF SITES of a (block, 64-sample group) and leaves the sigma codes in the item's scratch; MODE 1 then scores
// one ROW RANGE of one strand x frame part from those codes, its S values going to a buffer instead of through the scan;
// k_null_rowscan folds them afterwards.  MODE 0: the whole item (or one strand x frame part of it) in one workgroup.
struct DeepFetch {};   // `kind` of a cell in the span loops that fetch TWO cells ahead (see "two cells of distance" in null_body)

template <int NK, bool LDSC, bool EXACT, bool DUAL, int MODE = 0>
__device__ __forceinline__ void null_body(const NullArgs &A, const uint8_t *__restrict__ blob,
                                          const DevBlock *__restrict__ dblocks, const int *__restrict__ classBlocks,
                                          const uint32_t *__restrict__ flags, const uint32_t *__restrict__ Ustream, const uint8_t *__restrict__ pairTab,
                                          uint32_t *__restrict__ scratch, float *__restrict__ maxima) {
  constexpr int N = NK + 1;
  constexpr int CPW = 5;                  // 6-bit sigma codes per 32-bit word
  constexpr int NCW = (NK + CPW - 1) / CPW;
  constexpr bool kTableLookups = !LDSC && !EXACT;   // sigma look-ups from tables in LDS (the launches that read their codes from L2)
  constexpr int kTableWords = kTableLookups ? NK * kWave : 0;
  // The last code word of a site holds CNT = NK - 5 (NCW - 1) codes; with one or two codes it is kept as a byte or a halfword in the
  // staging scratch (a field shifted down by 6t is a look-up address either way: one code is (code << 2) < 256): a site of twelve rows
  // takes 9 bytes per sample instead of 12, one of seven rows 5 instead of 8 -- that much less to write, to re-read from L2, and to hold
  // in the suffix cache.  Layout per (strand, position): [NCW - 1][64] dwords, then [64] tail values of LB bytes.  (code_layout, rc_device.h)
  constexpr int LB = code_tail_bytes(NK);
  constexpr int PSW = code_pos_words(NK);                  // dwords per (strand, position)
  using TailT = std::conditional_t<LB == 1, uint8_t, std::conditional_t<LB == 2, uint16_t, uint32_t>>;
  constexpr int ZW = (NK + 31) / 32;      // 64-bit words per z entry (2 bits per sequence)
  using Regs = KRegs<std::make_integer_sequence<int, NK>>;
  // DUAL: two rows of S per pass (see "rows a and a + 1 together" below): for instantiations whose codes are staged in LDS and whose
  // register budget has room for a second row's states and a 32-entry row buffer at four wavefronts per SIMD.  The host picks it per
  // launch (NullArgs::dual): up to 127 VGPRs instead of 74, i.e. no other kernel shares a SIMD with four such wavefronts.
  static_assert(!DUAL || (!EXACT && NK <= kDualRowsMaxNK), "two rows per pass: fast division, few sequences");
  static_assert(!(LDSC && EXACT), "the EXACT instantiation uses no LDS");
  constexpr bool ROWS = MODE == 1, SIM = MODE == 2;
#ifndef RC_DEEP_FETCH_MAX_NK
#define RC_DEEP_FETCH_MAX_NK 31
#endif
  constexpr bool kDeep = !LDSC && !EXACT && !DUAL && MODE == 0 && NK <= RC_DEEP_FETCH_MAX_NK;   // codes from L2, one row per pass: long spans fetch two cells ahead
  static_assert(MODE == 0 || (!LDSC && !EXACT && !DUAL), "rows split over workgroups: the one-row kernel that reads its codes from L2");
  // phase A: threshold table of the tree (nnodes x 64 B); phase B with LDSC: [site][NCW][64] sigma codes
  // of the current strand x frame
  extern __shared__ uint32_t lds_codes[];

  const int lane = threadIdx.x;
  // Work items -- (block, 64-sample group), or with comboSplit (small batches) one strand x frame part
  // of such an item, so that 6x more wavefronts exist; each part recomputes phase A and the six partial
  // maxima meet in an atomic max -- are pulled from shared counters, not strided: blocks of one launch
  // differ in length, and wavefronts that share a SIMD do not advance at the same pace.
  // There is one queue per XCD, holding the sample groups g with g mod 8 == xcd: a group reads only its
  // own 64-sample slice of the MT19937 stream, so each XCD's L2 keeps 1/8 of the stream resident instead
  // of every XCD cycling through all of it.  A workgroup whose own queue is empty steals from the next.
  unsigned int xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  const int home = static_cast<int>(xcc & 7u);
  const int rowParts = ROWS ? A.rowParts : 1;
  const int parts = SIM ? A.simParts : A.comboSplit ? 6 * rowParts : 1;
  const int nBlocks = A.nBlocksPtr ? *A.nBlocksPtr : A.nClassBlocks;
  // The tail of the launch: when the queues have run dry, the workgroups that are still busy hold one item each and the others would
  // idle until the slowest has done its six strand x frame parts.  Instead an item taken near the end of a queue ("late") is
  // published once its sigma codes are in the staging scratch, and its parts are claimed one by one through a claim word, by its
  // owner and by any workgroup that found all queues empty and reads the codes from the owner's scratch.  The partial maxima
  // meet in an atomic max (the maxima are preset to -1).  A.steal (zeroed before the launch), G = gridDim.x:
  //   [0] published items, [1] late items still being simulated, [16 + slot] claim word of workgroup `slot` (0 nothing, 1..6 next
  //   unclaimed part + 1, > 6 all taken), [16 + G + 2 slot] its (block, group), [16 + 3 G + i] the i-th published item: slot + 1, XCD in the top byte.
  // Queue counters only grow, so once a workgroup has seen all of them exhausted no workgroup starts another item and a
  // published scratch is not rewritten while helpers read it.  Helpers poll two words and sleep in between: thousands of them
  // sweeping all claim words slowed the owners' simulation more than the sharing gained.
  // (the pointers are re-derived from the kernel arguments where they are needed: nothing of this lives in scalar registers across the DP)
  const bool stealing = A.steal != nullptr && !A.comboSplit;
  constexpr int kStealHdr = 16;
  auto claim_part = [&](int slot) -> int {   // the next unclaimed part of the item published in `slot`, 6 = none left
    unsigned int v = 0;
    if (lane == 0) v = atomicAdd(A.steal + kStealHdr + slot, 1u);
    v = static_cast<unsigned int>(__builtin_amdgcn_readfirstlane(static_cast<int>(v)));
    return (v >= 1u && v <= 6u) ? static_cast<int>(v) - 1 : 6;
  };
  auto steal_word = [&](size_t at) -> unsigned int { return __hip_atomic_load(A.steal + at, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
  int helpCursor = 0;   // published items before this one have no part left
  bool helping = false;
  for (int qi = 0;;) {
    int bi, grp, onlyCombo = -1, firstCombo = 0, rowPart = 0, itemIndex = 0;
    bool shared = false;
    int slot = static_cast<int>(blockIdx.x);   // whose staging scratch holds the item's codes
    if (!helping) {
      if (qi >= 8) {
        if (!stealing) break;
        helping = true;
        continue;
      }
      const int q = (home + qi) & 7;
      const int g0 = A.gLo + ((q - A.gLo) & 7);      // sample groups g0, g0 + 8, ... < gHi: those congruent to q
      const int gq = g0 < A.gHi ? ((A.gHi - 1 - g0) >> 3) + 1 : 0;
      const int perBlock = gq * parts;
      const int nq = nBlocks * perBlock;
      int t = nq;
      if (nq > 0) {
        if (lane == 0) t = static_cast<int>(atomicAdd((SIM ? A.simCounter : A.workCounter) + q, 1u));
        t = __builtin_amdgcn_readfirstlane(t);
      }
      if (t >= nq) { qi++; continue; }
      bi = classBlocks[t / perBlock];
      // blocks another launch takes care of (score tables outside the fast division's range -> the EXACT launch)
      // or that need no more samples (--stop-early decided)
      if (A.skipMask && (flags[bi] & A.skipMask)) continue;
      const int sub = t % perBlock;
      grp = g0 + 8 * (sub / parts);
      onlyCombo = A.comboSplit ? (sub % parts) / rowParts : -1;
      if constexpr (MODE != 0) { rowPart = SIM ? sub % parts : (sub % parts) % rowParts; itemIndex = (t / perBlock) * (A.gHi - A.gLo) + (grp - A.gLo); }
      if constexpr (MODE != 0) slot = itemIndex;   // the codes of a split item live in the ITEM's scratch: the simulation's pieces write them, the row ranges read them
      // items per queue that may still be running when the queues run dry (generous: a wrong guess costs six atomics)
      if (stealing && nq - t <= static_cast<int>(gridDim.x >> 3) + 16) {
        shared = true;
        if (lane == 0) atomicAdd(A.steal + 1, 1u);
      }
    } else {
      // a published item with unclaimed parts: 64 list entries per step, one per lane
      const size_t G = gridDim.x;
      const int np = __builtin_amdgcn_readfirstlane(static_cast<int>(steal_word(0)));
      int victim = -1;
      while (helpCursor < np) {
        const int i = helpCursor + lane;
        const unsigned int ent = i < np ? steal_word(kStealHdr + 3 * G + i) : 0u;        // 0: the entry is not written yet
        // only items of this XCD: owner and helper then share an L2, and the owner's codes need no write-back to memory
        const bool foreign = ent != 0u && (ent >> 24) != static_cast<unsigned int>(home);
        const unsigned int e = ent & 0xFFFFFFu;
        const unsigned int v = (e && !foreign) ? steal_word(kStealHdr + e - 1) : 0u;
        const unsigned long long open = __ballot(v >= 1u && v <= 6u);
        if (open) {   // spread the helpers over the open items of this step
          int pick = static_cast<int>(blockIdx.x % static_cast<unsigned int>(__builtin_popcountll(open)));
          unsigned long long m = open;
          while (pick--) m &= m - 1ull;
          victim = __builtin_amdgcn_readlane(static_cast<int>(e), __builtin_ctzll(m)) - 1;   // wave-uniform (a scalar): everything below branches on it
          break;
        }
        const unsigned long long notDone = ~__ballot(e != 0u && (v > 6u || foreign));
        const int adv = notDone ? __builtin_ctzll(notDone) : kWave;   // leading entries whose parts are all taken: never looked at again
        helpCursor += adv;
        if (adv < kWave) break;
      }
      if (victim < 0) {
        // nothing to take right now: done if nothing can come any more
        if (!A.stealWait) break;
        if (helpCursor >= np && steal_word(1) == 0u && static_cast<int>(steal_word(0)) == np) break;
        __builtin_amdgcn_s_sleep(127);
        continue;
      }
      firstCombo = claim_part(victim);
      if (firstCombo >= 6) continue;
      __atomic_thread_fence(__ATOMIC_ACQUIRE);   // the owner's codes and its (block, group)
      shared = true;
      slot = victim;
      bi = __builtin_amdgcn_readfirstlane(static_cast<int>(steal_word(kStealHdr + G + 2 * victim)));
      grp = __builtin_amdgcn_readfirstlane(static_cast<int>(steal_word(kStealHdr + G + 2 * victim + 1)));
    }
    uint32_t *__restrict__ cod = scratch + static_cast<size_t>(slot) * A.scratchStride;
    const DevBlock *__restrict__ db = dblocks + bi;
    const bool nanSem = EXACT && (flags[bi] & kFlagNan) != 0u;   // NaN score tables (only ever in the EXACT launches): see ref_max
    const int cols = db->cols, L = db->L, nnodes = db->nnodes;
    const float Delta = db->Delta, Omega = db->Omega, omega = db->omega, nkf = db->nkf;
    const NodeRec *__restrict__ nodes = reinterpret_cast<const NodeRec *>(blob + db->off_nodes);
    const uint32_t *__restrict__ qtipw = reinterpret_cast<const uint32_t *>(blob + db->off_qtip);   // u8[32] read as words: scalar loads
    const uint32_t *__restrict__ refcolw = reinterpret_cast<const uint32_t *>(blob + db->off_refcol);   // u16[L+1] (forward strand) read as words
    const unsigned long long *__restrict__ zw = reinterpret_cast<const unsigned long long *>(blob + db->off_zw);
    const unsigned long long *__restrict__ zmask = reinterpret_cast<const unsigned long long *>(blob + db->off_zmask);
    const int zmWords = db->zm_words;
    const uint32_t *__restrict__ maskw = reinterpret_cast<const uint32_t *>(blob + db->off_mask);
    const float *__restrict__ lut = reinterpret_cast<const float *>(blob + db->off_lut);
    const int sidx = grp * kWave + lane;
    const int L1 = L + 1;

    // ---- phase A: simulate the L reference-residue columns (evolve.c:416-433), SPI sites per pass
    // so that the stream loads and threshold look-ups of independent sites overlap; emit sigma
    // codes for both strands.  Node states of a site are packed 2 bits per node into NODEW
    // registers; the per-branch threshold table sits in LDS (the region phase B reuses for codes).
    if (!ROWS && !helping && (!kProfiling || !(A.debugSkip & 2))) {
      constexpr int SPI = 4;
      constexpr int NODEW = (2 * N - 1 + 15) / 16;   // 16 nodes per 32-bit word
      // The EXACT instantiation uses no LDS at all and reads both tables from global memory (L1/L2): its launches are
      // mostly empty (blocks flagged on the device) and must not wait for LDS that another batch's k_null holds.
      const U4 *thrTab;
      const uint8_t *pairLut;
      int thrStride;   // U4 per node
      if constexpr (EXACT) {
        thrTab = reinterpret_cast<const U4 *>(nodes);
        thrStride = sizeof(NodeRec) / sizeof(U4);
        pairLut = pairTab;
      } else {
        thrTab = reinterpret_cast<const U4 *>(lds_codes);
        thrStride = 4;
        // behind it the 64 x 64 codon-pair -> sigma-code table (4 KB): the 2 NK gathers per site are LDS
        // byte reads instead of global ones
        pairLut = reinterpret_cast<const uint8_t *>(lds_codes + nnodes * 16);
        const uint32_t *src = reinterpret_cast<const uint32_t *>(nodes);
        for (int idx = lane; idx < nnodes * 16; idx += kWave) lds_codes[idx] = src[(idx >> 4) * (sizeof(NodeRec) / 4) + (idx & 15)];
        const uint32_t *psrc = reinterpret_cast<const uint32_t *>(pairTab);
#pragma unroll
        for (int x = 0; x < 64 * 64 / 4 / kWave; x++) lds_codes[nnodes * 16 + x * kWave + lane] = psrc[x * kWave + lane];
      }
      const uint32_t *__restrict__ nodew = reinterpret_cast<const uint32_t *>(nodes);
      uint32_t win[N];
#pragma unroll
      for (int r = 0; r < N; r++) win[r] = 0;
      unsigned int clamped = 0;
      auto get_state = [&](const uint32_t (&ns)[NODEW], int q) -> uint32_t {   // q is wave-uniform
        uint32_t word = ns[0];
#pragma unroll
        for (int x = 1; x < NODEW; x++) if ((q >> 4) == x) word = ns[x];
        return (word >> (2 * (q & 15))) & 3u;
      };
      // MODE 2: the sites [iLo, iHi] of this piece (sites are independent; a codon needs its three columns, so the window is filled
      // from two sites earlier and nothing is emitted for those)
      int iLo = 1, iHi = L;
      if constexpr (SIM) { iLo = 1 + static_cast<int>(static_cast<long long>(L) * rowPart / parts); iHi = static_cast<int>(static_cast<long long>(L) * (rowPart + 1) / parts); }
      for (int i0 = (SIM && iLo > 3) ? iLo - 2 : 1; i0 <= iHi; i0 += SPI) {
        size_t ubase[SPI];
        uint32_t ns[SPI][NODEW];
#pragma unroll
        for (int t = 0; t < SPI; t++) {
          const int i = (i0 + t <= L) ? i0 + t : L;     // tail: repeat the last site, results unused
          const uint32_t col = (refcolw[i >> 1] >> (16 * (i & 1))) & 0xFFFFu;
          ubase[t] = static_cast<size_t>(col) * A.Spad + sidx;
#pragma unroll
          for (int x = 0; x < NODEW; x++) ns[t][x] = 0;
        }
        const size_t qstride = static_cast<size_t>(cols) * A.Spad;
        auto node_step = [&](int q, const uint32_t (&u)[SPI]) {
          const uint32_t bp = nodew[q * 20 + 16];                                // NodeRec as 20 words: scalar loads
          const int pq = static_cast<int>(nodew[q * 20 + 17] & 255u);
          auto evolve = [&](auto withBase) {   // seqgen/evolve.c:416-433 for node q at the pass's sites
#pragma unroll
            for (int t = 0; t < SPI; t++) {
              const uint32_t ps = (q == 0) ? 0u : get_state(ns[t], pq);
              const U4 th = thrTab[q * thrStride + ps];
              uint32_t st = (u[t] > th.x) + (u[t] > th.y) + (u[t] > th.z);
              if constexpr (decltype(withBase)::value) st += (bp >> (2 * ps)) & 3u;
              if constexpr (SIM) clamped += (u[t] > th.w) & (i0 + t >= iLo) & (i0 + t <= iHi); else clamped += (u[t] > th.w);
              const uint32_t bits = st << (2 * (q & 15));
#pragma unroll
              for (int x = 0; x < NODEW; x++) if (NODEW == 1 || (q >> 4) == x) ns[t][x] |= bits;
            }
          };
          // base offsets exist only for degenerate cumulative rows: a real (scalar) branch, not a select
          if (__builtin_expect(bp != 0u, 0)) evolve(std::true_type{}); else evolve(std::false_type{});
        };
        if constexpr (SIM) {
          // Rows split over workgroups is the path of a chip that is nearly empty (one wavefront per SIMD at most): nothing hides the
          // round trip of a node's draws, and 2N - 2 of them per pass in a chain were 0.55 of the 0.67 ms a one-block launch took.  The draws
          // of eight nodes are fetched together here (32 loads in flight) -- the form that was slower where sixteen wavefronts share a CU.
          constexpr int QB = 8;
          for (int q0 = 0; q0 < nnodes; q0 += QB) {
            uint32_t ub[QB][SPI];
#pragma unroll
            for (int x = 0; x < QB; x++) {
              const int q = q0 + x < nnodes ? q0 + x : nnodes - 1;
#pragma unroll
              for (int t = 0; t < SPI; t++) ub[x][t] = Ustream[ubase[t] + q * qstride];
            }
#pragma unroll
            for (int x = 0; x < QB; x++) if (q0 + x < nnodes) node_step(q0 + x, ub[x]);
          }
        } else {
          for (int q = 0; q < nnodes; q++) {
            uint32_t u[SPI];
#pragma unroll
            for (int t = 0; t < SPI; t++) u[t] = Ustream[ubase[t] + q * qstride];
            node_step(q, u);
          }
        }
#pragma unroll
        for (int t = 0; t < SPI; t++) {
          const int i = i0 + t;
          if (i <= L) {
#pragma unroll
            for (int r = 0; r < N; r++) win[r] = (win[r] << 2) | get_state(ns[t], static_cast<int>((qtipw[r >> 2] >> (8 * (r & 3))) & 255u));   // low 6 bits = the codon; older sites drift out at the top
            if (i >= 3 && (!SIM || (i >= iLo && i <= iHi))) {
              const int ip = L + 3 - i;  // reverse-strand position whose codon is these three columns
              const uint32_t aF = win[0] & 63u;
              const uint32_t aR = codon_flip(aF) ^ 63u;
#pragma unroll
              for (int w = 0; w < NCW; w++) {
                uint32_t wf = 0, wr = 0;
                const uint32_t mF = maskw[(0 * L1 + i) * NCW + w], mR = maskw[(1 * L1 + ip) * NCW + w];   // wave-uniform
#pragma unroll
                for (int c = 0; c < CPW; c++) {
                  const int k = w * CPW + c;  // 0-based index of sequence k+1
                  if (k < NK) {
                    const uint32_t bF = win[k + 1] & ((mF >> (6 * c)) & 63u);
                    const uint32_t bR = (codon_flip(win[k + 1]) ^ 63u) & ((mR >> (6 * c)) & 63u);
                    // field c sits at bits [6c+7 : 6c+2]: shifted down by 6c it is a ds_bpermute address
                    wf |= static_cast<uint32_t>(pairLut[aF * 64 + bF]) << (6 * c + 2);
                    wr |= static_cast<uint32_t>(pairLut[aR * 64 + bR]) << (6 * c + 2);
                  }
                }
                if (w < NCW - 1) {
                  cod[static_cast<size_t>(0 * L1 + i) * PSW + w * kWave + lane] = wf;
                  cod[static_cast<size_t>(1 * L1 + ip) * PSW + w * kWave + lane] = wr;
                } else {
                  reinterpret_cast<TailT *>(cod + static_cast<size_t>(0 * L1 + i) * PSW + (NCW - 1) * kWave)[lane] = static_cast<TailT>(wf);
                  reinterpret_cast<TailT *>(cod + static_cast<size_t>(1 * L1 + ip) * PSW + (NCW - 1) * kWave)[lane] = static_cast<TailT>(wr);
                }
              }
            }
          }
        }
      }
      // padding lanes do not count; with comboSplit the six parts of an item redo the same simulation: the first one reports
      if (clamped && sidx < A.sampleN && (SIM || onlyCombo <= 0)) atomicAdd(A.clampCount, static_cast<unsigned long long>(clamped));
    }
    if constexpr (SIM) continue;   // the scoring is another launch's (MODE 1)
    if (!helping && shared) {   // a late item: its codes are in the scratch, open it to the workgroups that have run out of work
      const size_t G = gridDim.x;
      if (lane == 0) {
        __hip_atomic_store(A.steal + kStealHdr + G + 2 * blockIdx.x, static_cast<unsigned int>(bi), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(A.steal + kStealHdr + G + 2 * blockIdx.x + 1, static_cast<unsigned int>(grp), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      // Publishing the codes: an agent-scope release, the documented hand-off to another CU (MI355X_MICROARCH.md, inter-workgroup
      // visibility): it writes this XCD's dirty L2 lines back.  (Helpers are taken from the owner's XCD and the L1 writes through, so a
      // workgroup-scope release did in practice on gfx950 what is needed, 1-1.5 % faster on streams of small sub-batches, nothing on
      // 10 000 blocks -- profiles/r04/ab_tail_release.txt; not worth an undocumented dependency, and removed in round 5.)
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      __builtin_amdgcn_s_waitcnt(0);
      if (lane == 0) {
        atomicExch(A.steal + kStealHdr + blockIdx.x, 1u);
        const unsigned int idx = atomicAdd(A.steal, 1u);
        __hip_atomic_store(A.steal + kStealHdr + 3 * G + idx, (blockIdx.x + 1u) | (static_cast<unsigned int>(home) << 24), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        atomicSub(A.steal + 1, 1u);
      }
      firstCombo = claim_part(slot);
    }

    // ---- phase B: both strands x 3 frames: DP over (start a, end j) + serial HSS scan
    float best = -1.0f;
    unsigned long long statCells = 0, statNoPos = 0, statPosLanes = 0, statPristine = 0, statEvent = 0, statTail = 0;   // RC_PROFILING builds only
    // omega and Delta as per-lane values: v_add_f32 with two VGPR operands issues at the full rate,
    // with an SGPR operand at ~60 % of it (tools/microbench.hip)
    float omega_v, Delta_v = 0.0f;
    asm volatile("v_mov_b32 %0, %1" : "=v"(omega_v) : "s"(omega));
    float rcpNk_v = 0.0f, negNk_v = 0.0f;
    if constexpr (!EXACT && (NK & (NK - 1)) != 0) {
      asm volatile("v_mov_b32 %0, %1" : "=v"(rcpNk_v) : "s"(1.0f / static_cast<float>(NK)));
      asm volatile("v_mov_b32 %0, %1" : "=v"(negNk_v) : "s"(-static_cast<float>(NK)));
    }
    if constexpr (EXACT) asm volatile("v_mov_b32 %0, %1" : "=v"(Delta_v) : "s"(Delta));
    // the strand x frame parts c = 3 s + f: all six in turn, the one part of a split item, or -- a shared item -- those this
    // workgroup gets from the item's claim word
    if (!kProfiling || !(A.debugSkip & 1)) {
      Regs R;
      int curS = -1;
      for (int c = onlyCombo >= 0 ? onlyCombo : firstCombo; c < 6; c = onlyCombo >= 0 ? 6 : (shared ? claim_part(slot) : c + 1)) {
        const int s = c >= 3 ? 1 : 0, f = c - 3 * s;
        if (s != curS) {
          static_for<NK>([&](auto kc) {  // 64-entry sigma table of sequence k spread over the 64 lanes
            constexpr int k = decltype(kc)::value;
            R.template at<k>().lut = __float_as_int(lut[(s * NK + k) * kLutSize + lane]);
            // codes from L2: the tables go to LDS, [k][64] from byte 0 on -- a look-up is a ds_read_b32 at code x 4 (+ k x 256 in the
            // instruction's offset) instead of a ds_bpermute_b32: 0.58..0.91 look-ups per ns and CU against 0.34..0.38
            // (tools/mb_lds_lookup.hip), and the look-ups are what bounds these launches (C5: 0.38 LDS instructions per ns and CU)
            if constexpr (kTableLookups) lds_codes[k * kWave + lane] = static_cast<uint32_t>(R.template at<k>().lut);
          });
          curS = s;
        }
        const int sites = (L - f) / 3;
        if (sites <= 0) continue;
        // the code words of this strand x frame: site j at position 3 + f + 3 j of the strand (wave-uniform base, the lane added per access)
        const uint32_t *__restrict__ sbase = cod + static_cast<size_t>(s * L1 + 3 + f) * PSW;
        constexpr int kSiteWords = 3 * PSW;                   // dwords from one site of a frame to the next
        auto word_at = [&](int j, int x) -> uint32_t {        // word x of site j from the staging scratch
          const uint32_t *p = sbase + static_cast<size_t>(j) * kSiteWords;
          if (x < NCW - 1) return p[x * kWave + lane];
          return static_cast<uint32_t>(reinterpret_cast<const TailT *>(p + (NCW - 1) * kWave)[lane]);
        };
        const unsigned long long *__restrict__ zbase = zw + static_cast<size_t>(s * L1 + 3 + f) * ZW;               // site j: [3j * ZW + word]
        const unsigned long long *__restrict__ zany = zmask + static_cast<size_t>(s * 3 + f) * zmWords;            // bit j: some z != 0 at site j
        // Codes from L2 (the strand x frame's words do not fit the LDS at the occupancy this kernel needs): row a reads the words of
        // the sites a..sites-1, so site j is read by j + 1 rows -- the LAST sites are the ones read most.  Whatever LDS the launch
        // was given beyond phase A's tables (A.cacheSites sites' worth) therefore holds the words of the frame's suffix [jc, sites),
        // in the scratch's own layout (narrow tail values); the rows read those from LDS and only the sites before jc from L2 (C5, 100
        // sites, 17 of them cached: 31 % of the reads; 8 rows x 150 columns, 21 of 50: 67 %).
        int jc = sites;
        constexpr int kCacheSiteBytes = PSW * 4;              // bytes of one cached site
        if constexpr (LDSC) {                                 // staged: all dwords, [site][NCW][64] (conflict-free ds_read_b32)
          for (int idx = 0; idx < sites * NCW; idx++) lds_codes[idx * kWave + lane] = word_at(idx / NCW, idx % NCW);
          __builtin_amdgcn_s_waitcnt(0);   // everything landed: the row loops below issue no vector memory ops
        } else if constexpr (!EXACT) {
          jc = sites - (A.cacheSites < sites ? A.cacheSites : sites);
          for (int j = jc; j < sites; j++) {
            uint32_t *dst = lds_codes + kTableWords + (j - jc) * PSW;
#pragma unroll
            for (int x = 0; x < NCW - 1; x++) dst[x * kWave + lane] = word_at(j, x);
            reinterpret_cast<TailT *>(dst + (NCW - 1) * kWave)[lane] = static_cast<TailT>(word_at(j, NCW - 1));
          }
          __builtin_amdgcn_s_waitcnt(0);
        }
        const uint32_t ldsBase = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(lds_codes + kTableWords));   // (the codes: behind the sigma tables, if those are in LDS)
        const uint32_t ldsLane = ldsBase + static_cast<uint32_t>(lane) * 4u;                                  // a lane's dwords
        const uint32_t ldsTail = ldsBase + (NCW - 1) * kWave * 4u + static_cast<uint32_t>(lane) * LB;         // its tail value (suffix cache)
        const uint32_t laneOff = static_cast<uint32_t>(lane) * 4u, laneTail = static_cast<uint32_t>(lane) * LB;
        // asm fetches of a site's code words: the compiler does not know them as memory operations and places no waits of its own
        // (RC_PLAIN_FETCH, a build for cross-checks -- tools/ab_plain_fetch.sh: the same fetches as plain loads, which the compiler sees and
        // waits for itself: slower, and free of the hazard the asm form carries -- a register it does not know to be pending must not be
        // copied or spilled in front of its wait, tests/test_codegen_cpu.py.  Results must not differ.)
        [[maybe_unused]] const uint8_t *ldsBytesAll = reinterpret_cast<const uint8_t *>(lds_codes + kTableWords) - ldsBase;   // (so that the asm form's byte addresses index it)
        auto fetch_staged = [&](uint32_t (&w)[NCW], int j) {        // LDSC: [site][NCW][64] dwords
          const uint32_t at = ldsLane + static_cast<uint32_t>(j) * (NCW * kWave * 4u);
#pragma unroll
          for (int x = 0; x < NCW; x++) {
#ifdef RC_PLAIN_FETCH
            w[x] = *reinterpret_cast<const uint32_t *>(ldsBytesAll + at + x * kWave * 4);
#else
            asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(w[x]) : "v"(at), "n"(x * kWave * 4));
#endif
          }
        };
        auto fetch_cached = [&](uint32_t (&w)[NCW], int j) {        // suffix cache: site j >= jc, the scratch's layout
          const uint32_t off = static_cast<uint32_t>(j - jc) * kCacheSiteBytes;
#ifdef RC_PLAIN_FETCH
          if constexpr (NCW > 1) {
#pragma unroll
            for (int x = 0; x < NCW - 1; x++) w[x] = *reinterpret_cast<const uint32_t *>(ldsBytesAll + ldsLane + off + x * kWave * 4);
          }
          w[NCW - 1] = static_cast<uint32_t>(*reinterpret_cast<const TailT *>(ldsBytesAll + ldsTail + off));
#else
          if constexpr (NCW > 1) {
            const uint32_t at = ldsLane + off;
#pragma unroll
            for (int x = 0; x < NCW - 1; x++) asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(w[x]) : "v"(at), "n"(x * kWave * 4));
          }
          const uint32_t att = ldsTail + off;
          if constexpr (LB == 1) asm volatile("ds_read_u8 %0, %1" : "=v"(w[NCW - 1]) : "v"(att));
          else if constexpr (LB == 2) asm volatile("ds_read_u16 %0, %1" : "=v"(w[NCW - 1]) : "v"(att));
          else asm volatile("ds_read_b32 %0, %1" : "=v"(w[NCW - 1]) : "v"(att));
#endif
        };
        auto fetch_global = [&](uint32_t (&w)[NCW], const uint32_t *p) {   // p: wave-uniform, settled in scalar registers five wait states ago
          const uint32_t lo = laneOff + 0u, lt = laneTail + 0u;   // (copies: an asm operand alone does not make a generic lambda capture the variable)
#ifdef RC_PLAIN_FETCH
          const uint8_t *pb = reinterpret_cast<const uint8_t *>(p);
#pragma unroll
          for (int x = 0; x < NCW - 1; x++) w[x] = *reinterpret_cast<const uint32_t *>(pb + lo + x * kWave * 4);
          w[NCW - 1] = static_cast<uint32_t>(*reinterpret_cast<const TailT *>(pb + lt + (NCW - 1) * kWave * 4));
          return;
#endif
#pragma unroll
          for (int x = 0; x < NCW - 1; x++) asm volatile("global_load_dword %0, %1, %2 offset:%3" : "=v"(w[x]) : "v"(lo), "s"(p), "n"(x * kWave * 4));
          if constexpr (LB == 1) asm volatile("global_load_ubyte %0, %1, %2 offset:%3" : "=v"(w[NCW - 1]) : "v"(lt), "s"(p), "n"((NCW - 1) * kWave * 4));
          else if constexpr (LB == 2) asm volatile("global_load_ushort %0, %1, %2 offset:%3" : "=v"(w[NCW - 1]) : "v"(lt), "s"(p), "n"((NCW - 1) * kWave * 4));
          else asm volatile("global_load_dword %0, %1, %2 offset:%3" : "=v"(w[NCW - 1]) : "v"(lt), "s"(p), "n"((NCW - 1) * kWave * 4));
        };
        // the words of site j, complete when it returns (a row's first site)
        auto load_words = [&](int j, uint32_t (&w)[NCW]) {
          if constexpr (LDSC) {
#pragma unroll
            for (int x = 0; x < NCW; x++) w[x] = lds_codes[(j * NCW + x) * kWave + lane];
          } else if constexpr (!EXACT) {
            if (j >= jc) {   // wave-uniform
              fetch_cached(w, j);
            } else {
              const uint32_t *p = sbase + static_cast<size_t>(j) * kSiteWords;
              asm volatile("s_nop 4" : "+s"(p));   // a VMEM instruction must not read an SGPR within five wait states of the scalar write
              fetch_global(w, p);
            }
#pragma unroll
            for (int x = 0; x < NCW; x++) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" : "+v"(w[x]));
          } else {
#pragma unroll
            for (int x = 0; x < NCW; x++) w[x] = word_at(j, x);
          }
        };
        // sigma of every sequence at the site whose code words are w: a 64-entry table look-up per
        // sequence through ds_bpermute (address bits [7:2] select the source lane).  Written as asm so
        // that all look-ups of a code word are in flight together and ONE wait covers them (the
        // compiler would put a wait in front of every use); the "+v" operands tie the wait between
        // the look-ups and their first use.
        // With the codes in LDS the words of site jn (the next cell) are fetched here as well, INTO w,
        // right after the look-ups have been issued (they read w at issue): no second register set and
        // no copy per cell.  LDS operations complete in order, so lgkmcnt(NCW) leaves exactly those
        // NCW reads in flight; the next call starts by waiting for them.
        // Codes from L2 (neither staged nor EXACT): the same scheme with global loads.  The words of site jn are fetched INTO w behind
        // the look-ups -- from the LDS suffix cache if jn >= jc, else with global_load_dword (scalar base + lane offset: the address
        // costs no vector instruction) -- and the NEXT call starts by waiting for them: a whole cell of other work hides the L2 / HBM
        // latency.  (As plain C++ loads into a second register set the compiler's own wait-count placement waited for them twenty
        // instructions after they had been issued: its analysis merges the loop-entry state, where the first words are still on
        // their way, into the loop.)
        auto lookup = [&](uint32_t (&w)[NCW], float (&sig)[NK], int jn, auto scalarLoadsNearby, auto kind) {   // (kind: unused)
          if constexpr (LDSC) {
#pragma unroll
            for (int x = 0; x < NCW; x++) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(w[x]));
          }
          // the global fetch's scalar base, settled HERE: a VMEM instruction must not read an SGPR within five wait states of the scalar
          // instruction that wrote it, and the compiler's hazard recognizer does not look inside inline asm -- the look-ups below lie between
          const uint32_t *pnext = sbase + static_cast<size_t>(jn) * kSiteWords;
          if constexpr (!LDSC && !EXACT) asm volatile("" : "+s"(pnext));
          static_for<NK>([&](auto kc) {
            constexpr int k = decltype(kc)::value;
            constexpr int t = k % CPW;
            // codes from L2: the words were fetched by the previous call and arrive in order -- wait for each one where it is first used
            // (any LDS read of the suffix cache was covered by that call's closing wait)
            // (a deep cell: the other register set's NCW fetches were issued behind these and stay in flight)
            constexpr int kYounger = std::is_same_v<std::decay_t<decltype(kind)>, DeepFetch> ? NCW : 0;
            if constexpr (!LDSC && !EXACT && t == 0) asm volatile("s_waitcnt vmcnt(%1)" : "+v"(w[k / CPW]) : "n"(NCW - 1 - k / CPW + kYounger));
            uint32_t addr;
            if constexpr (t == 0) addr = w[k / CPW]; else addr = w[k / CPW] >> (6 * t);   // bits [7:2] select the lane
            float got;
            if constexpr (kTableLookups) {
              uint32_t at;
              asm volatile("v_and_b32 %0, 0xfc, %1" : "=v"(at) : "v"(addr));
              asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(got) : "v"(at), "n"(k * kWave * 4));
            } else {
              const int lutv = R.template at<k>().lut;
              asm volatile("ds_bpermute_b32 %0, %1, %2" : "=v"(got) : "v"(addr), "v"(lutv));
            }
            sig[k] = got;
          });
          // one wait for the look-ups; operands of the last code word pin its position.  leave: LDS reads that were issued behind
          // the look-ups and may stay in flight (LDS operations complete in order)
          auto wait_lookups = [&](auto leave) {
            constexpr int k0 = (NCW - 1) * CPW;
            constexpr int cnt = NK - k0;   // look-ups of the last code word: 1..5, each one an operand of the wait (no padding copies)
            // (scalar loads share the counter and may complete out of order: where the compiler could have one
            // in flight -- the event cell reads z -- wait for everything)
            constexpr int left = (decltype(leave)::value > 0 && !decltype(scalarLoadsNearby)::value) ? decltype(leave)::value : 0;
            if constexpr (cnt == 1) asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(sig[k0]) : "n"(left));
            else if constexpr (cnt == 2) asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(sig[k0]), "+v"(sig[k0 + 1]) : "n"(left));
            else if constexpr (cnt == 3) asm volatile("s_waitcnt lgkmcnt(%3)" : "+v"(sig[k0]), "+v"(sig[k0 + 1]), "+v"(sig[k0 + 2]) : "n"(left));
            else if constexpr (cnt == 4)
              asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(sig[k0]), "+v"(sig[k0 + 1]), "+v"(sig[k0 + 2]), "+v"(sig[k0 + 3]) : "n"(left));
            else
              asm volatile("s_waitcnt lgkmcnt(%5)" : "+v"(sig[k0]), "+v"(sig[k0 + 1]), "+v"(sig[k0 + 2]), "+v"(sig[k0 + 3]), "+v"(sig[k0 + 4]) : "n"(left));
            // earlier words: make their values depend on the wait as well (no instruction emitted)
            static_for<NK>([&](auto kc) {
              constexpr int k = decltype(kc)::value;
              if constexpr (k < k0) {
                float b = sig[k];
                asm volatile("" : "+v"(b), "+v"(sig[k0]));
                sig[k] = b;
              }
            });
          };
          if constexpr (LDSC) {
            fetch_staged(w, jn);
            wait_lookups(std::integral_constant<int, NCW>{});
          } else if constexpr (!EXACT) {
            // the fetch goes out as early as it can (the look-ups have read w when they were issued): the next call waits for it, and
            // nearly a whole cell of other work lies between
            // (Also built and taken back: the sites before jc through a RING of four LDS slots filled by LDS-DMA -- global_load_lds_*,
            // memory -> LDS with no register in between, three cells ahead --, so that a cell only ever reads LDS and no register holds
            // data the compiler does not know to be pending.  Bit-exact on the suite, and slower: 7 / 12 / 13 rows x 150 columns
            // 36.6 / 65.4 / 69.3 -> 39.0 / 69.9 / 74.1 ms, C5 587 -> 644 ms (profiles/r04/ab_l2path_v5_*): three 256-byte DMA pieces per
            // cell cost more issue time than the deeper prefetch gains, and the ring's 3 KB come out of the suffix cache.)
            // Which of the two is decided per cell by a wave-uniform branch.  (Cutting every row loop at jc - 1 instead, so that each
            // loop has one kind of fetch compiled in, was built and taken back: with two loops per span and two event cells the
            // register allocator moved w between registers at the joins -- BEFORE the wait that makes w valid --, results were wrong
            // wherever a row had frame-shift events, and the launches were 5-12 % slower on top.)  The wait for the look-ups is common
            // to both arms and behind them (inside the arms its register ties became eleven moves per arm); it covers the LDS arm's reads.
            if constexpr (std::is_same_v<std::decay_t<decltype(kind)>, DeepFetch>) fetch_global(w, pnext);   // (the caller keeps jn in front of the suffix cache)
            else if (jn >= jc) fetch_cached(w, jn);
            else fetch_global(w, pnext);
            wait_lookups(std::integral_constant<int, 0>{});
          } else {
            wait_lookups(std::integral_constant<int, 0>{});
          }
        };
        // Only positive S values ever reach a sample's result (getHSS ignores the rest and the frame's
        // final entry is never reported), and for Delta < 0 a positive S equals sum/(N-1): the fast
        // instantiation skips the MAX3 with Delta; blocks with Delta >= 0 are routed to EXACT.
        // the constants of the constant-divisor division as per-lane registers: a literal operand goes over the constant bus like an
        // SGPR, and such VALU instructions issue at the single rate (tools/microbench.hip)
        auto finish = [&](float sum) -> float {
          if constexpr (kProfiling) {   // how often a wave-uniform "no lane has sum > 0" test could skip the division and the scan step
            const unsigned long long pos = __ballot(sum > 0.0f);
            statCells++; statNoPos += (pos == 0ull); statPosLanes += __builtin_popcountll(pos);
          }
          if constexpr (EXACT) {
            const float top = (sum > Delta_v) ? sum : Delta_v;
            return div_by_nk<NK, EXACT>(top, nkf);
          } else {
            if constexpr ((NK & (NK - 1)) == 0) return div_by_nk<NK, EXACT>(sum, nkf);
            else {   // div_by_nk's sequence on register operands
              const float q0 = sum * rcpNk_v;
              const float r = __builtin_fmaf(negNk_v, q0, sum);
              return __builtin_fmaf(r, rcpNk_v, q0);
            }
          }
        };
        // codon without a frame shift in any sequence (z == 0 for all k, score.c:506-510): straight-line
        auto fast_cell = [&](uint32_t (&w)[NCW], int jn, auto kind) -> float {
          float sig[NK];
          lookup(w, sig, jn, std::false_type{}, kind);
          float sum = 0.0f;
          static_for<NK>([&](auto kc) {
            constexpr int k = decltype(kc)::value;
            auto &r = R.template at<k>();
            r.s0 = r.s0 + sig[k];
            r.s1 = r.s1 + omega_v;
            r.s2 = r.s2 + omega_v;
            const float m = fmaxf(fmaxf(r.s0, r.s1), r.s2);
            sum = (k == 0) ? m : sum + m;
          });
          return finish(sum);
        };
        // the same before the row has met its first event: states 1 and 2 of EVERY sequence have only
        // ever received "+ omega" from 0, so they all hold the one value w (identical operations on
        // identical operands); MAX3(s0, w, w) = max(s0, w).  w advances once per codon.
        auto pristine_cell = [&](uint32_t (&w)[NCW], float &wv, int jn, auto kind) -> float {
          float sig[NK];
          lookup(w, sig, jn, std::false_type{}, kind);
          wv = wv + omega_v;
          float sum = 0.0f;
          static_for<NK>([&](auto kc) {
            constexpr int k = decltype(kc)::value;
            auto &r = R.template at<k>();
            r.s0 = r.s0 + sig[k];
            const float m = fmaxf(r.s0, wv);
            sum = (k == 0) ? m : sum + m;
          });
          return finish(sum);
        };
        // the same after the row's LAST event: states 1 and 2 only ever receive "+ omega" from here to the end of the
        // row and are never read individually again, so one value per sequence, t = max(s1, s2), stands for both:
        // fl(max(s1, s2) + omega) == max(fl(s1 + omega), fl(s2 + omega)) because x -> fl(x + omega) is monotone, and
        // MAX3(s0, s1, s2) == max(s0, t).  The caller folds s2 into s1 first; one add per sequence and codon less.
        auto tail_cell = [&](uint32_t (&w)[NCW], int jn, auto kind) -> float {
          float sig[NK];
          lookup(w, sig, jn, std::false_type{}, kind);
          float sum = 0.0f;
          static_for<NK>([&](auto kc) {
            constexpr int k = decltype(kc)::value;
            auto &r = R.template at<k>();
            r.s0 = r.s0 + sig[k];
            r.s1 = r.s1 + omega_v;
            const float m = fmaxf(r.s0, r.s1);
            sum = (k == 0) ? m : sum + m;
          });
          return finish(sum);
        };
        // codon where some sequence has z != 0 (score.c:512-533), wave-uniform branch per sequence
        auto event_cell = [&](uint32_t (&w)[NCW], int j, int jn, auto kind) -> float {
          float sig[NK];
          lookup(w, sig, jn, std::true_type{}, kind);
          unsigned long long z[ZW];
#pragma unroll
          for (int x = 0; x < ZW; x++) z[x] = zbase[static_cast<size_t>(3 * j) * ZW + x];
          float sum = 0.0f;
          static_for<NK>([&](auto kc) {
            constexpr int k = decltype(kc)::value;
            auto &r = R.template at<k>();
            pair_step(static_cast<int>((z[k >> 5] >> (2 * (k & 31))) & 3ull), sig[k], Delta, Omega, omega, r.s0, r.s1, r.s2);
            const float m = fmaxf(fmaxf(r.s0, r.s1), r.s2);
            sum = (k == 0) ? m : sum + m;
          });
          return finish(sum);
        };
        // every codon of a block with NaN score tables: the recurrence as the reference writes it, its MAX macro and operand order (ref_max)
        auto nan_cell = [&](uint32_t (&w)[NCW], int j, int jn, auto kind) -> float {
          float sig[NK];
          lookup(w, sig, jn, std::true_type{}, kind);
          unsigned long long z[ZW];
#pragma unroll
          for (int x = 0; x < ZW; x++) z[x] = zbase[static_cast<size_t>(3 * j) * ZW + x];
          float sum = 0.0f;
          static_for<NK>([&](auto kc) {
            constexpr int k = decltype(kc)::value;
            auto &r = R.template at<k>();
            pair_step<true>(static_cast<int>((z[k >> 5] >> (2 * (k & 31))) & 3ull), sig[k], Delta, Omega, omega, r.s0, r.s1, r.s2);
            const float m = ref_max3<true>(r.s0, r.s1, r.s2);
            sum = (k == 0) ? m : sum + m;
          });
          return finish(sum);
        };
        // first site >= j (and < end) with an event, or end
        auto next_event = [&](int j, int end) -> int {
          while (j < end) {
            const unsigned long long mword = zany[j >> 6] >> (j & 63);
            if (mword) { const int e = j + __builtin_ctzll(mword); return e < end ? e : end; }
            j = (j | 63) + 1;
          }
          return end;
        };
        SampleScan st{0.0f, 0.0f, 0u, 0u};
        float two_v;
        asm volatile("v_mov_b32 %0, 2.0" : "=v"(two_v));
        const float negTie = -A.tieThr;
        if constexpr (DUAL) {
          // ---- Two rows per pass.  sigma of (site j, sequence k) is the same for every row a <= j, and the look-up -- a shift and a
          // ds_bpermute per sequence -- is what the cell loop waits for most (tools/ab_lookup.sh: four of the five look-ups left out,
          // 49.1 -> 43.7 ms).  Rows a and a + 1 therefore walk their common sites together on one set of look-ups.  getHSS reads the
          // entries row by row, so row a's entries go through the scan as they are made and row a + 1's S values wait in a register
          // buffer (32 floats, indexed with the wave-uniform site number: s_set_gpr_idx) until row a is done; a row longer than the
          // buffer has its first cells made twice -- states only beside row a, then again with their scan when its turn comes.
          // Every row's additions and maxima are the ones of the one-row loop in the same order: results are bit-identical.
          // A pair needs a site a without a frame-shift event (both rows then change their cell kind at the same sites) and two more
          // rows below it; other rows go alone.
          typedef float RowBuf __attribute__((ext_vector_type(32)));
          constexpr int kBuf = 32;
          RowBuf buf = {};
          Regs RB;   // row a + 1 (its lut members are not used)
          auto pristine2 = [&](uint32_t (&w)[NCW], float &wvA, float &wvB, int jn, float &vB, auto kd) -> float {
            float sig[NK];
            lookup(w, sig, jn, std::false_type{}, kd);
            wvA = wvA + omega_v;
            wvB = wvB + omega_v;
            float sumA = 0.0f, sumB = 0.0f;
            static_for<NK>([&](auto kc) {
              constexpr int k = decltype(kc)::value;
              auto &ra = R.template at<k>();
              auto &rb = RB.template at<k>();
              ra.s0 = ra.s0 + sig[k];
              const float ma = fmaxf(ra.s0, wvA);
              sumA = (k == 0) ? ma : sumA + ma;
              rb.s0 = rb.s0 + sig[k];
              const float mb = fmaxf(rb.s0, wvB);
              sumB = (k == 0) ? mb : sumB + mb;
            });
            vB = finish(sumB);
            return finish(sumA);
          };
          auto fast2 = [&](uint32_t (&w)[NCW], int jn, float &vB, auto kd) -> float {
            float sig[NK];
            lookup(w, sig, jn, std::false_type{}, kd);
            float sumA = 0.0f, sumB = 0.0f;
            static_for<NK>([&](auto kc) {
              constexpr int k = decltype(kc)::value;
              auto &ra = R.template at<k>();
              auto &rb = RB.template at<k>();
              ra.s0 = ra.s0 + sig[k];
              ra.s1 = ra.s1 + omega_v;
              ra.s2 = ra.s2 + omega_v;
              const float ma = fmaxf(fmaxf(ra.s0, ra.s1), ra.s2);
              sumA = (k == 0) ? ma : sumA + ma;
              rb.s0 = rb.s0 + sig[k];
              rb.s1 = rb.s1 + omega_v;
              rb.s2 = rb.s2 + omega_v;
              const float mb = fmaxf(fmaxf(rb.s0, rb.s1), rb.s2);
              sumB = (k == 0) ? mb : sumB + mb;
            });
            vB = finish(sumB);
            return finish(sumA);
          };
          auto tail2 = [&](uint32_t (&w)[NCW], int jn, float &vB, auto kd) -> float {
            float sig[NK];
            lookup(w, sig, jn, std::false_type{}, kd);
            float sumA = 0.0f, sumB = 0.0f;
            static_for<NK>([&](auto kc) {
              constexpr int k = decltype(kc)::value;
              auto &ra = R.template at<k>();
              auto &rb = RB.template at<k>();
              ra.s0 = ra.s0 + sig[k];
              ra.s1 = ra.s1 + omega_v;
              const float ma = fmaxf(ra.s0, ra.s1);
              sumA = (k == 0) ? ma : sumA + ma;
              rb.s0 = rb.s0 + sig[k];
              rb.s1 = rb.s1 + omega_v;
              const float mb = fmaxf(rb.s0, rb.s1);
              sumB = (k == 0) ? mb : sumB + mb;
            });
            vB = finish(sumB);
            return finish(sumA);
          };
          auto event2 = [&](uint32_t (&w)[NCW], int j, int jn, float &vB) -> float {
            float sig[NK];
            lookup(w, sig, jn, std::true_type{}, std::true_type{});
            unsigned long long z[ZW];
#pragma unroll
            for (int x = 0; x < ZW; x++) z[x] = zbase[static_cast<size_t>(3 * j) * ZW + x];
            float sumA = 0.0f, sumB = 0.0f;
            static_for<NK>([&](auto kc) {
              constexpr int k = decltype(kc)::value;
              auto &ra = R.template at<k>();
              auto &rb = RB.template at<k>();
              const int zc = static_cast<int>((z[k >> 5] >> (2 * (k & 31))) & 3ull);
              pair_step(zc, sig[k], Delta, Omega, omega, ra.s0, ra.s1, ra.s2);
              const float ma = fmaxf(fmaxf(ra.s0, ra.s1), ra.s2);
              sumA = (k == 0) ? ma : sumA + ma;
              pair_step(zc, sig[k], Delta, Omega, omega, rb.s0, rb.s1, rb.s2);
              const float mb = fmaxf(fmaxf(rb.s0, rb.s1), rb.s2);
              sumB = (k == 0) ? mb : sumB + mb;
            });
            vB = finish(sumB);
            return finish(sumA);
          };
          int a = 0;
          int pendRow = -1, pendExtra = 0, pendN = 0;   // row a + 1 of the last pair: its first pendExtra cells are still to be made, then pendN buffered values to scan
          while (a < sites || pendRow >= 0) {
            if (pendRow < 0 && a + 2 < sites && !((zany[a >> 6] >> (a & 63)) & 1ull)) {
              // ---- rows a and a + 1 together
              sample_scan_row_begin(st, best, static_cast<uint32_t>(a));
              static_for<NK>([&](auto kc) {
                constexpr int k = decltype(kc)::value;
                auto &r = R.template at<k>();
                r.s0 = r.s1 = r.s2 = 0.0f;
                auto &rb = RB.template at<k>();
                rb.s0 = rb.s1 = rb.s2 = 0.0f;
              });
              uint32_t wcur[NCW];
              load_words(a, wcur);
              int j = a;
              float j2f = static_cast<float>(2 * a + 1);
              float wvA = 0.0f, wvB = 0.0f;
              asm volatile("v_mov_b32 %0, 0" : "=v"(wvA));
              asm volatile("v_mov_b32 %0, 0" : "=v"(wvB));
              {   // row a's first cell (site a: no event)
                const float v = pristine_cell(wcur, wvA, a + 1, std::true_type{});
                sample_scan_step(st, v, j2f, negTie);
                asm volatile("v_add_f32 %0, %1, %0" : "+v"(j2f) : "v"(two_v));
                j = a + 1;
              }
              const int extra = (sites - 1 - a > kBuf) ? sites - 1 - a - kBuf : 0;   // row a + 1 has sites - 1 - a cells: the last kBuf are buffered
              const int b0 = a + 1 + extra;
              auto keep = [&](float vB, int jj) {   // (a cell before b0 lands in entry 0 and is overwritten by the one that belongs there)
                const int idx = jj > b0 ? jj - b0 : 0;
                buf[idx] = vB;
              };
              // one cell pair (rows a and a + 1 at site jj) on the register set w: row a's value through the scan, row a + 1's into the buffer
              auto pair_cell = [&](uint32_t (&w)[NCW], int jj, int jn, auto kd, auto &&fn) {
                float vB;
                const float v = fn(w, jn, vB, kd);
                sample_scan_step(st, v, j2f, negTie);
                keep(vB, jj);
                const float two_c = two_v;   // (a copy: an asm operand alone does not make a generic lambda capture the variable)
                asm volatile("v_add_f32 %0, %1, %0" : "+v"(j2f) : "v"(two_c));
              };
              // the cells j .. e - 1 of one kind; codes from L2: in front of the suffix cache with two cells of fetch distance on two register sets
              // (see "Two cells of distance" at the one-row loops below)
              auto dspan = [&](int e, auto &&fn) {
                if constexpr (!LDSC && !EXACT && MODE == 0) {
                  const int lim = e < jc - 2 ? e : jc - 2;
                  if (lim - j >= 4) {
                    uint32_t wB[NCW];
                    {
                      const uint32_t *p = sbase + static_cast<size_t>(j + 1) * kSiteWords;
                      asm volatile("s_nop 4" : "+s"(p));   // a VMEM instruction must not read an SGPR within five wait states of the scalar write
                      fetch_global(wB, p);
                    }
                    for (; j + 2 <= lim; j += 2) {
                      pair_cell(wcur, j, j + 2, DeepFetch{}, fn);
                      pair_cell(wB, j + 1, j + 3, DeepFetch{}, fn);
                    }
#pragma unroll
                    for (int x = 0; x < NCW; x++) asm volatile("s_waitcnt vmcnt(0)" : "+v"(wB[x]));   // (pending registers are not given away)
                  }
                }
                for (; j < e; j++) pair_cell(wcur, j, j + 1 < sites ? j + 1 : j, std::true_type{}, fn);
              };
              {
                dspan(next_event(j, sites), [&](uint32_t (&w)[NCW], int jn, float &vB, auto kd) { return pristine2(w, wvA, wvB, jn, vB, kd); });
                static_for<NK>([&](auto kc) {
                  constexpr int k = decltype(kc)::value;
                  auto &r = R.template at<k>();
                  r.s1 = r.s2 = wvA;
                  auto &rb = RB.template at<k>();
                  rb.s1 = rb.s2 = wvB;
                });
              }
              while (j < sites) {
                if ((zany[j >> 6] >> (j & 63)) & 1ull) {
                  const int jev = j;
                  pair_cell(wcur, j, j + 1 < sites ? j + 1 : j, std::true_type{}, [&](uint32_t (&w)[NCW], int jn, float &vB, auto) { return event2(w, jev, jn, vB); });
                  j++;
                }
                const int e = next_event(j, sites);
                if (e == sites) {
                  static_for<NK>([&](auto kc) {
                    constexpr int k = decltype(kc)::value;
                    auto &r = R.template at<k>();
                    r.s1 = fmaxf(r.s1, r.s2);
                    auto &rb = RB.template at<k>();
                    rb.s1 = fmaxf(rb.s1, rb.s2);
                  });
                  dspan(e, [&](uint32_t (&w)[NCW], int jn, float &vB, auto kd) { return tail2(w, jn, vB, kd); });
                }
                dspan(e, [&](uint32_t (&w)[NCW], int jn, float &vB, auto kd) { return fast2(w, jn, vB, kd); });
              }
#pragma unroll
              for (int x = 0; x < NCW; x++) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(wcur[x]));
              sample_scan_row_end(st, static_cast<uint32_t>(a));
              pendRow = a + 1; pendExtra = extra; pendN = sites - b0;
              a += 2;
            } else {
              // ---- one row: row `a`, or the row left over from a pair (its first pendExtra cells, then its buffered values)
              const bool pend = pendRow >= 0;
              const int row = pend ? pendRow : a;
              const int jend = pend ? row + pendExtra : ((row == sites - 1) ? sites - 1 : sites);
              float j2f = static_cast<float>(2 * row + 1);
                sample_scan_row_begin(st, best, static_cast<uint32_t>(row));
                static_for<NK>([&](auto kc) {
                  constexpr int k = decltype(kc)::value;
                  auto &r = R.template at<k>();
                  r.s0 = r.s1 = r.s2 = 0.0f;
                });
                uint32_t wcur[NCW];
                load_words(row, wcur);
                int j = row;
                {   // pristine segment: up to the row's first event
                  const int e = next_event(j, jend);
                  float wv = 0.0f;
                  asm volatile("v_mov_b32 %0, 0" : "=v"(wv));   // keep w a per-lane register (full-rate v_add/v_max operands)
                  for (; j < e; j++) {
                    const int jn = j + 1 < sites ? j + 1 : j;
                    uint32_t wnext[NCW];
                    if constexpr (EXACT) load_words(jn, wnext);
                    if constexpr (kProfiling) statPristine++;
                    const float v = pristine_cell(wcur, wv, jn, std::true_type{});
                    sample_scan_step(st, v, j2f, negTie);
                    asm volatile("v_add_f32 %0, %1, %0" : "+v"(j2f) : "v"(two_v));
                    if constexpr (EXACT) {
      #pragma unroll
                      for (int x = 0; x < NCW; x++) wcur[x] = wnext[x];
                    }
                  }
                  static_for<NK>([&](auto kc) {
                    constexpr int k = decltype(kc)::value;
                    auto &r = R.template at<k>();
                    r.s1 = r.s2 = wv;
                  });
                }
                while (j < jend) {
                  if ((zany[j >> 6] >> (j & 63)) & 1ull) {               // event codon
                    const int jn = j + 1 < sites ? j + 1 : j;
                    uint32_t wnext[NCW];
                    if constexpr (EXACT) load_words(jn, wnext);
                    if constexpr (kProfiling) statEvent++;
                    const float v = event_cell(wcur, j, jn, std::true_type{});
                    sample_scan_step(st, v, j2f, negTie);
                    asm volatile("v_add_f32 %0, %1, %0" : "+v"(j2f) : "v"(two_v));
                    if constexpr (EXACT) {
      #pragma unroll
                      for (int x = 0; x < NCW; x++) wcur[x] = wnext[x];
                    }
                    j++;
                  }
                  const int e = next_event(j, jend);
                  if (e == jend) {                                       // no further event in this row: the tail
                    static_for<NK>([&](auto kc) {
                      constexpr int k = decltype(kc)::value;
                      auto &r = R.template at<k>();
                      r.s1 = fmaxf(r.s1, r.s2);
                    });
                    for (; j < e; j++) {
                      const int jn = j + 1 < sites ? j + 1 : j;
                      uint32_t wnext[NCW];
                      if constexpr (EXACT) load_words(jn, wnext);
                      if constexpr (kProfiling) statTail++;
                      const float v = tail_cell(wcur, jn, std::true_type{});
                      sample_scan_step(st, v, j2f, negTie);
                      asm volatile("v_add_f32 %0, %1, %0" : "+v"(j2f) : "v"(two_v));
                      if constexpr (EXACT) {
      #pragma unroll
                        for (int x = 0; x < NCW; x++) wcur[x] = wnext[x];
                      }
                    }
                  }
                  for (; j < e; j++) {                                   // tight loop: no events
                    const int jn = j + 1 < sites ? j + 1 : j;
                    uint32_t wnext[NCW];
                    if constexpr (EXACT) load_words(jn, wnext);
                    const float v = fast_cell(wcur, jn, std::true_type{});
                    sample_scan_step(st, v, j2f, negTie);
                    asm volatile("v_add_f32 %0, %1, %0" : "+v"(j2f) : "v"(two_v));
                    if constexpr (EXACT) {
      #pragma unroll
                      for (int x = 0; x < NCW; x++) wcur[x] = wnext[x];
                    }
                  }
                }
                if constexpr (!EXACT) {   // the last cell's fetch-ahead must have landed before its registers are given away
      #pragma unroll
                  for (int x = 0; x < NCW; x++) {
                    if constexpr (LDSC) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(wcur[x]));
                    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" : "+v"(wcur[x]));
                  }
                }
              if (pend) {
                // the buffered values, entry 0 first: unrolled, so that every entry is a register operand of its scan step (an
                // indexed read costs three instructions and 9 cycles of the SIMD, tools/microbench_gpr_idx.hip)
                [&]() {
#pragma unroll
                  for (int idx = 0; idx < kBuf; idx++) {
                    if (idx >= pendN) return;
                    sample_scan_step(st, buf[idx], j2f, negTie);
                    asm volatile("v_add_f32 %0, %1, %0" : "+v"(j2f) : "v"(two_v));
                  }
                }();
                pendRow = -1;
              } else {
                a += 1;
              }
              sample_scan_row_end(st, static_cast<uint32_t>(row));
            }
          }
        } else {
          // ---- Rows split over workgroups (ROWS; batches so small that even their strand x frame parts leave most of the chip idle: a
          // caller that scores block by block, RNAcode.c:164-216 through the shim).  A part's triangle of cells is cut into rowParts row
          // ranges of equal cell counts; each range is a work item of its own, its S values go to the item's buffer in getHSS's entry
          // order (row a's sites - a entries behind those of the rows before it, [entry][64 lanes]) and k_null_rowscan runs the serial
          // fold over the finished buffer.  Same cells, same fold, same order: bit-identical; the DP's latency is divided by rowParts.
          int aLo = 0, aHi = sites;
          float *sdst = nullptr;
          if constexpr (ROWS) {
            const long long total = static_cast<long long>(sites) * (sites + 1) / 2;
            auto row_at = [&](long long cellsBefore) {   // the first row with at least that many entries in front of it
              int a = 0;
              long long cum = 0;
              while (a < sites && cum < cellsBefore) { cum += sites - a; a++; }
              return a;
            };
            aLo = row_at(total * rowPart / rowParts);
            aHi = rowPart + 1 == rowParts ? sites : row_at(total * (rowPart + 1) / rowParts);
            const long long before = static_cast<long long>(aLo) * sites - static_cast<long long>(aLo) * (aLo - 1) / 2;
            sdst = A.sbuf + (static_cast<size_t>(itemIndex) * 6 + c) * A.sbufStride + static_cast<size_t>(before) * kWave + lane;
          }
          for (int a = aLo; a < aHi; a++) {
            if constexpr (!ROWS) sample_scan_row_begin(st, best, static_cast<uint32_t>(a));
            static_for<NK>([&](auto kc) {
              constexpr int k = decltype(kc)::value;
              auto &r = R.template at<k>();
              r.s0 = r.s1 = r.s2 = 0.0f;
            });
            uint32_t wcur[NCW];
            load_words(a, wcur);
            const int jend = (a == sites - 1) ? sites - 1 : sites;   // the frame's final entry is handled below
            int j = a;
            float j2f = static_cast<float>(2 * a + 1);   // 2j + 1 of the cell being entered
            // what becomes of a cell's S value: the item's buffer (rows split over workgroups), or getHSS's fold
            auto take = [&](float v) {
              if constexpr (ROWS) { *sdst = v; sdst += kWave; }
              else {
                sample_scan_step(st, v, j2f, negTie);
                asm volatile("v_add_f32 %0, %1, %0" : "+v"(j2f) : "v"(two_v));
              }
            };
            // one cell on wcur: the site whose words are fetched behind the look-ups, EXACT's second register set, the cell, its scan step
            auto one = [&](auto &&cell) {   // cell(w, jn, kind) -> S
              const int jn = j + 1 < sites ? j + 1 : j;
              uint32_t wnext[NCW];
              if constexpr (EXACT) load_words(jn, wnext);
              take(cell(wcur, jn, std::true_type{}));
              if constexpr (EXACT) {
#pragma unroll
                for (int x = 0; x < NCW; x++) wcur[x] = wnext[x];
              }
            };
            // The cells j .. e - 1 of one kind.  Two cells of distance (kDeep; round 6): a fetch issued behind a cell's look-ups has the rest of
            // that cell -- some 60 % of it -- to arrive before the next cell waits for it, against an L2 / HBM round trip of one to several cells.
            // In a span's part in front of the suffix cache the cells therefore go in PAIRS on two register sets, wcur and wB: a cell refills its own
            // set for the cell after next, the wait at a cell's head leaves the other set's NCW fetches in flight (lookup: kYounger), nothing is
            // copied.  The pair loop is entered with one extra fetch (cell j + 1 into wB) and left with wB drained and dropped -- wcur then holds
            // cell j as the single-set loop expects it, which fetches cell j + 1 once more: two fetches per span for nothing, from four cells on.
            auto span = [&](int e, auto &&cell) {
              if constexpr (kDeep) {
                const int lim = e < jc - 2 ? e : jc - 2;   // a deep cell j fetches site j + 2, its partner j + 3: both in front of the suffix cache
                if (lim - j >= 4) {
                  uint32_t wB[NCW];
                  {
                    const uint32_t *p = sbase + static_cast<size_t>(j + 1) * kSiteWords;
                    asm volatile("s_nop 4" : "+s"(p));   // a VMEM instruction must not read an SGPR within five wait states of the scalar write
                    fetch_global(wB, p);
                  }
                  for (; j + 2 <= lim; j += 2) {
                    take(cell(wcur, j + 2, DeepFetch{}));
                    take(cell(wB, j + 3, DeepFetch{}));
                  }
#pragma unroll
                  for (int x = 0; x < NCW; x++) asm volatile("s_waitcnt vmcnt(0)" : "+v"(wB[x]));   // (pending registers are not given away)
                }
              }
              for (; j < e; j++) one(cell);
            };
            if constexpr (EXACT) {
              if (nanSem)   // (the whole row: none of the shortcuts below holds with NaN states)
                for (; j < jend; j++) {
                  const int jj = j;
                  one([&](uint32_t (&w)[NCW], int jn, auto kd) { return nan_cell(w, jj, jn, kd); });
                }
            }
            {   // pristine segment: up to the row's first event
              const int e = next_event(j, jend);
              float wv = 0.0f;
              asm volatile("v_mov_b32 %0, 0" : "=v"(wv));   // keep w a per-lane register (full-rate v_add/v_max operands)
              span(e, [&](uint32_t (&w)[NCW], int jn, auto kd) {
                if constexpr (kProfiling) statPristine++;
                return pristine_cell(w, wv, jn, kd);
              });
              static_for<NK>([&](auto kc) {
                constexpr int k = decltype(kc)::value;
                auto &r = R.template at<k>();
                r.s1 = r.s2 = wv;
              });
            }
            while (j < jend) {
              if ((zany[j >> 6] >> (j & 63)) & 1ull) {               // event codon
                if constexpr (kProfiling) statEvent++;
                const int jj = j;
                one([&](uint32_t (&w)[NCW], int jn, auto kd) { return event_cell(w, jj, jn, kd); });
                j++;
              }
              const int e = next_event(j, jend);
              if (e == jend) {                                       // no further event in this row: the tail
                static_for<NK>([&](auto kc) {
                  constexpr int k = decltype(kc)::value;
                  auto &r = R.template at<k>();
                  r.s1 = fmaxf(r.s1, r.s2);
                });
                span(e, [&](uint32_t (&w)[NCW], int jn, auto kd) {
                  if constexpr (kProfiling) statTail++;
                  return tail_cell(w, jn, kd);
                });
              }
              span(e, [&](uint32_t (&w)[NCW], int jn, auto kd) { return fast_cell(w, jn, kd); });   // tight loop: no events
            }
            if constexpr (!EXACT) {   // the last cell's fetch-ahead must have landed before its registers are given away
#pragma unroll
              for (int x = 0; x < NCW; x++) {
                if constexpr (LDSC) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(wcur[x]));
                else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" : "+v"(wcur[x]));
              }
            }
            if constexpr (ROWS) { if (a == sites - 1) sdst += kWave; }   // (the frame's final entry has a slot of its own; it is never computed)
            else sample_scan_row_end(st, static_cast<uint32_t>(a));
          }
        }
        if constexpr (!ROWS) sample_scan_last(st, best);   // entry (sites-1, sites-1): its own value can never be reported
      }
    }
    if constexpr (kProfiling) {
      if (A.cellStats && lane == 0) {
        atomicAdd(A.cellStats, statCells); atomicAdd(A.cellStats + 1, statNoPos); atomicAdd(A.cellStats + 2, statPosLanes);
        atomicAdd(A.cellStats + 3, statPristine); atomicAdd(A.cellStats + 4, statEvent); atomicAdd(A.cellStats + 5, statTail);
      }
    }
    if (!ROWS && sidx < A.sampleN) {   // (ROWS: k_null_rowscan writes the maxima)
      float *dst = maxima + static_cast<size_t>(db->out_index) * A.sampleN + sidx;
      // scores are positive floats or the -1 sentinel: integer order == float order on that set
      // (parts of one item scored by several workgroups: the maxima are preset to -1, which no partial maximum needs to replace)
      if (A.comboSplit || shared) { if (best > -1.0f) atomicMax(reinterpret_cast<int *>(dst), __float_as_int(best)); }
      else *dst = best;
    }
  }
}

// A wavefront issues one vector instruction per ~8 cycles whatever its instruction-level parallelism (tools/microbench.hip, profiles/r06/
// valu_issue_costs_r06.txt): a SIMD needs four wavefronts to issue at its rate, and every wavefront the registers cost below four costs a quarter of
// it.  The one-row kernels take 131..168 VGPRs at 17..24 rows (three wavefronts) and 174..229 at 25..32 (two) when the register allocator is left
// alone.  Held to 128 VGPRs at 17..22 rows and to 168 at 25..30 it spills 3..35 registers and keeps them out of the straight-line cells (one scratch
// access in one kind of cell at 22 rows, two at 30; tests/test_codegen_cpu.py counts them and audits the registers the asm fetches leave pending):
// 20 rows x 150 / 300 columns 0.307 / 0.388 -> 0.338 / 0.431 of the issue roof, 28 rows 0.252 / 0.325 -> 0.310 / 0.362.  Not at 23, 24 rows (three
// scratch accesses per cell: 0.313 / 0.392 -> 0.308 / 0.373) nor at 31, 32 (six: 0.245 / 0.315 -> 0.213 / 0.237, and the audit finds a pending
// register copied).  profiles/r06/rows_sweep_before.txt, rows_sweep_forced_occupancy.txt, rows_sweep_after.txt.
// 7, 10 and 11 rows: the second register set of the two-cell fetch distance cost these three a wavefront (78 -> 83, 90 -> 97, 95 -> 103 VGPRs); told
// what to aim for, the register allocator fits them into 78 / 89 / 94 again without a single spill.
constexpr int null_min_waves(int NK, bool EXACT, bool DUAL, int MODE) { return (EXACT || DUAL || MODE != 0) ? 1 : NK == 6 ? 6 : (NK == 9 || NK == 10) ? 5 : (NK >= 16 && NK <= 21) ? 4 : (NK >= 24 && NK <= 29) ? 3 : 1; }
template <int NK, bool LDSC, bool EXACT, bool DUAL = false, int MODE = 0>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(null_min_waves(NK, EXACT, DUAL, MODE))))
void k_null(NullArgs A, const uint8_t *__restrict__ blob,
                                             const DevBlock *__restrict__ dblocks, const int *__restrict__ classBlocks,
                                             const uint32_t *__restrict__ flags, const uint32_t *__restrict__ Ustream, const uint8_t *__restrict__ pairTab,
                                             uint32_t *__restrict__ scratch, float *__restrict__ maxima) {
  null_body<NK, LDSC, EXACT, DUAL, MODE>(A, blob, dblocks, classBlocks, flags, Ustream, pairTab, scratch, maxima);
}

// The instantiations that read their code words from L2 live on occupancy (4000 blocks x 150 columns x 9 rows: 49.4 ms at 20
// workgroups per CU, 64.0 at 12).  k_null_occ<NK> is k_null<NK, false, false> compiled for one more wavefront per SIMD than its
// registers allow by themselves: 6..8 other sequences (81-85 VGPRs) for six (80 VGPRs, a handful of spills: 24 workgroups per CU
// instead of 20), 10..12 (101-113 VGPRs) for five (96 VGPRs: 20 instead of 16).  Round 3 measured +1..7 % on uniform batches and
// -5 % on a stream of ten row-count classes, whose native-block, table and fit kernels ran in the registers these wavefronts take --
// so the host launches it only for batches of ONE row-count class (rc_schedule.cpp, plain_plan).
constexpr int hi_occ_waves(int NK) { return (NK >= 6 && NK <= 8) ? 6 : (NK >= 10 && NK <= 12) ? 5 : 0; }
template <int NK>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(hi_occ_waves(NK) ? hi_occ_waves(NK) : 1)))
void k_null_occ(NullArgs A, const uint8_t *__restrict__ blob, const DevBlock *__restrict__ dblocks, const int *__restrict__ classBlocks,
                const uint32_t *__restrict__ flags, const uint32_t *__restrict__ Ustream, const uint8_t *__restrict__ pairTab,
                uint32_t *__restrict__ scratch, float *__restrict__ maxima) {
  null_body<NK, false, false, false>(A, blob, dblocks, classBlocks, flags, Ustream, pairTab, scratch, maxima);
}


// launch / occupancy of one N-1 value, used by the range units
// dynamic LDS of the instantiations that do not stage codes: at least phase A's threshold table (64 B per node) + pair table; the
// host may give more (NullArgs::plainLds), which phase B uses as its suffix cache
template <int NK> constexpr size_t phase_a_lds() { return NK < kFastRows ? static_cast<size_t>(kPhaseALds) : static_cast<size_t>(2 * (NK + 1)) * 64 + 64 * 64; }

template <int NK>
inline int null_occupancy_one(size_t ldsBytes) {
  int nb = 0;
  hipError_t e = ldsBytes ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_null<NK, true, false>, 64, ldsBytes)
                          : hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_null<NK, false, false>, 64, phase_a_lds<NK>());
  if (e != hipSuccess) return 0;
  return nb;
}
// codes from L2 with `dynLds` bytes of dynamic LDS; hiOcc: the k_null_occ build (0 if this N-1 has none)
template <int NK>
inline int null_occupancy_plain_one(size_t dynLds, int hiOcc) {
  int nb = 0;
  hipError_t e;
  if (hiOcc == 2) {   // the two-row kernel with its codes from L2
    if constexpr (NK >= 3 && NK <= kDualRowsMaxNK) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_null<NK, false, false, true>, 64, dynLds);
    else return 0;
  } else if (hiOcc) {
    if constexpr (hi_occ_waves(NK) != 0) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_null_occ<NK>, 64, dynLds);
    else return 0;
  } else e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_null<NK, false, false>, 64, dynLds);
  if (e != hipSuccess) return 0;
  return nb;
}

// exact: the class contains blocks flagged exact_div (never staged in LDS: a rare parameter choice)
template <int NK>
inline void launch_null_one(const NullArgs &a, int grid, size_t ldsBytes, bool exact, hipStream_t stream) {
  if (exact) {
    hipLaunchKernelGGL((k_null<NK, false, true>), dim3(grid), dim3(64), 0, stream, a, a.blob, a.dblocks,
                       a.classBlocks, a.flags, a.U, a.pair, a.scratch, a.maxima);
  } else if (ldsBytes) {
    if constexpr (NK <= kDualRowsMaxNK) {
      if (a.dual && ldsBytes <= 48 * 1024) {
        hipLaunchKernelGGL((k_null<NK, true, false, true>), dim3(grid), dim3(64), ldsBytes, stream, a, a.blob, a.dblocks,
                           a.classBlocks, a.flags, a.U, a.pair, a.scratch, a.maxima);
        return;
      }
    }
    if (ldsBytes > 48 * 1024)
      (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_null<NK, true, false>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(ldsBytes));
    hipLaunchKernelGGL((k_null<NK, true, false>), dim3(grid), dim3(64), ldsBytes, stream, a, a.blob, a.dblocks,
                       a.classBlocks, a.flags, a.U, a.pair, a.scratch, a.maxima);
  } else {
    const size_t dyn = a.plainLds ? static_cast<size_t>(a.plainLds) : phase_a_lds<NK>();
    if constexpr (NK < kFastRows) {
      if (a.rowParts > 1) {   // an item's simulation split by sites (a.simGrid workgroups), then its scoring split by rows; k_null_rowscan (rc_kernels.hip) follows on the same stream
        hipLaunchKernelGGL((k_null<NK, false, false, false, 2>), dim3(a.simGrid), dim3(64), phase_a_lds<NK>(), stream, a, a.blob, a.dblocks,
                           a.classBlocks, a.flags, a.U, a.pair, a.scratch, a.maxima);
        hipLaunchKernelGGL((k_null<NK, false, false, false, 1>), dim3(grid), dim3(64), dyn, stream, a, a.blob, a.dblocks,
                           a.classBlocks, a.flags, a.U, a.pair, a.scratch, a.maxima);
        return;
      }
    }
    if constexpr (NK >= 3 && NK <= kDualRowsMaxNK) {
      if (a.hiOcc == 2) {
        hipLaunchKernelGGL((k_null<NK, false, false, true>), dim3(grid), dim3(64), dyn, stream, a, a.blob, a.dblocks,
                           a.classBlocks, a.flags, a.U, a.pair, a.scratch, a.maxima);
        return;
      }
    }
    if constexpr (hi_occ_waves(NK) != 0) {
      if (a.hiOcc) {
        hipLaunchKernelGGL((k_null_occ<NK>), dim3(grid), dim3(64), dyn, stream, a, a.blob, a.dblocks,
                           a.classBlocks, a.flags, a.U, a.pair, a.scratch, a.maxima);
        return;
      }
    }
    hipLaunchKernelGGL((k_null<NK, false, false>), dim3(grid), dim3(64), dyn, stream, a, a.blob, a.dblocks,
                       a.classBlocks, a.flags, a.U, a.pair, a.scratch, a.maxima);
  }
}

// one translation unit per range [LO, HI]
#define RC_DEFINE_NULL_RANGE(SUFFIX, LO, HI)                                                        \
  template <int NK> static int occ_rec_##SUFFIX(int nk, size_t lds) {                                \
    if constexpr (NK > HI) return 0;                                                                 \
    else return nk == NK ? null_occupancy_one<NK>(lds) : occ_rec_##SUFFIX<NK + 1>(nk, lds);           \
  }                                                                                                  \
  template <int NK> static bool launch_rec_##SUFFIX(int nk, const NullArgs &a, int grid, size_t lds, bool exact, hipStream_t st) { \
    if constexpr (NK > HI) return false;                                                             \
    else {                                                                                           \
      if (nk == NK) { launch_null_one<NK>(a, grid, lds, exact, st); return true; }                   \
      return launch_rec_##SUFFIX<NK + 1>(nk, a, grid, lds, exact, st);                               \
    }                                                                                                \
  }                                                                                                  \
  template <int NK> static int occp_rec_##SUFFIX(int nk, size_t lds, int hiOcc) {                    \
    if constexpr (NK > HI) return 0;                                                                 \
    else return nk == NK ? null_occupancy_plain_one<NK>(lds, hiOcc) : occp_rec_##SUFFIX<NK + 1>(nk, lds, hiOcc); \
  }                                                                                                  \
  int null_occupancy_##SUFFIX(int nk, size_t lds) { return occ_rec_##SUFFIX<LO>(nk, lds); }          \
  int null_occupancy_plain_##SUFFIX(int nk, size_t lds, int hiOcc) { return occp_rec_##SUFFIX<LO>(nk, lds, hiOcc); } \
  bool launch_null_##SUFFIX(int nk, const NullArgs &a, int grid, size_t lds, bool exact, hipStream_t st) { \
    return launch_rec_##SUFFIX<LO>(nk, a, grid, lds, exact, st);                                     \
  }


// Blocks with more than 32 rows (N-1 in [32, 63]): one instantiation per N-1, the EXACT one without code staging -- the 4 N-1
// per-sequence registers no longer fit the register file (the compiler parks part of them in AGPRs), so this is the slow but
// complete path for the rare wide block; the reference accepts up to 500 rows (rnaz_utils.h:7).
#define RC_DEFINE_NULL_WIDE_RANGE(SUFFIX, LO, HI)                                                   \
  template <int NK> static int occ_rec_##SUFFIX(int nk) {                                            \
    if constexpr (NK > HI) return 0;                                                                 \
    else {                                                                                           \
      if (nk == NK) {                                                                                \
        int nb = 0;                                                                                  \
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_null<NK, false, true>, 64, 0) != hipSuccess) return 0; \
        return nb;                                                                                   \
      }                                                                                              \
      return occ_rec_##SUFFIX<NK + 1>(nk);                                                           \
    }                                                                                                \
  }                                                                                                  \
  template <int NK> static bool launch_rec_##SUFFIX(int nk, const NullArgs &a, int grid, hipStream_t st) { \
    if constexpr (NK > HI) return false;                                                             \
    else {                                                                                           \
      if (nk == NK) {                                                                                \
        hipLaunchKernelGGL((k_null<NK, false, true>), dim3(grid), dim3(64), 0, st, a, a.blob, a.dblocks, a.classBlocks, \
                           a.flags, a.U, a.pair, a.scratch, a.maxima);                                \
        return true;                                                                                 \
      }                                                                                              \
      return launch_rec_##SUFFIX<NK + 1>(nk, a, grid, st);                                           \
    }                                                                                                \
  }                                                                                                  \
  int null_occupancy_##SUFFIX(int nk, size_t) { return occ_rec_##SUFFIX<LO>(nk); }                   \
  int null_occupancy_plain_##SUFFIX(int, size_t, int) { return 0; }                                  \
  bool launch_null_##SUFFIX(int nk, const NullArgs &a, int grid, size_t, bool, hipStream_t st) {     \
    return launch_rec_##SUFFIX<LO>(nk, a, grid, st);                                                 \
  }

}  // namespace rc
