// rc_null_tiled.h -- the null-distribution DP for blocks of 28..64 rows: the sequences in TILES of KT, the cell's running sum carried from tile to tile.
//
// Why: a wavefront issues one vector instruction per ~8 cycles whatever its instruction-level parallelism (tools/microbench.hip: v_add_f32 at
// 1 / 2 / 4 wavefronts per SIMD = 9.0 / 4.6 / 2.8 cycles per instruction), so a SIMD needs FOUR wavefronts to issue at its rate, and k_null<N-1>,
// which keeps 3 (N-1) states in registers, has 3 from 17 rows on, 2 from 28, 1 from 37 (tools/rows_sweep.py, profiles/r06/rows_sweep_before.txt:
// 0.39 / 0.25..0.32 / 0.11..0.18 of the issue roof; k_generic_dp, whose states live in memory, 0.19..0.24).  Here a pass over a row scores
// KT <= 15 sequences (45 state registers: four wavefronts per SIMD) and leaves, per cell, the sum of their maxima in a row buffer; the next pass
// starts from that value and adds its own sequences' maxima -- score.c:830-845's additions in score.c's order, k = 1, 2, ..., only interrupted:
// bit-identical.  The last pass divides and feeds getHSS's fold (SampleScan).  Per cell and pass: ceil(KT / 5) code words in, one float in,
// one out.
//
// Same decomposition as the other sampling kernels (lane = null sample, 64 samples of a block per wavefront, items from per-XCD queues) and two
// launches per round of items like the wide-block path: k_generic_sim<true> leaves an item's sigma codes in the tile-major layout below,
// k_tiled_dp<KT> scores them.  Sequences a block's last tile has too many of (T KT - (N-1)) are scored as zeros: sigma table 0, z = 0, so s0 = 0,
// the gap states never above 0 (omega <= 0: the host sends other blocks elsewhere, rc_device.h block_class) and the maximum +0.0.
// (Two ROWS per pass on one fetch of the code words -- most of what a pass reads -- was built and measured in round 6: bit-identical, 30 % fewer
// bytes, 18 % fewer instructions, and slower: 72..90 state registers spill at four wavefronts per SIMD (0.16 of the issue roof) and run 0.22..0.25 at
// three, against 0.27..0.30 here.  profiles/r06/ab_tiled_two_rows.txt, docs/history/tiled_two_rows_per_pass.patch.)
// (Also measured and taken back in round 6, each 2..4 % slower: the last two cells of a pass fetching the next pass's first two -- the other three
// wavefronts of the SIMD already hide that round trip --, and the look-ups written out with ONE s_waitcnt behind them instead of the compiler's
// thirteen: the cell is bound by its vector instructions (4.5 cycles of SIMD time per VALU instruction at 0.30), not by scalar issue slots.
// profiles/r06/ab_tiled_tile_events.txt.)
// SEM = true: the instantiation for blocks with NaN score tables (the reference's MAX macro, rc_null_kernel.h ref_max), a second launch that takes
// only those.  Division and the maximum with Delta are the plain ones (one per cell and row, in the last pass).
#pragma once
#include "rc_null_kernel.h"

namespace rc {

// an item's codes: u32 [2][L + 1][WT][64]: strand, reference position, word, lane; word t * WPT + x holds the codes of the sequences
// t KT + 5 x .. + 4 (tile t), six bits each, field c at bits [6c + 7 : 6c + 2] -- shifted down by 6c and masked with 0xfc it is the byte offset of the
// sequence's sigma in its 64-entry table
struct TiledLayout {
  int KT, T, WPT, WT;
  __host__ __device__ TiledLayout(int NK, int kt) : KT(kt), T((NK + kt - 1) / kt), WPT((kt + 4) / 5), WT(T * WPT) {}
  __host__ __device__ size_t codes_bytes(int L) const { return static_cast<size_t>(2) * (L + 1) * WT * kWave * sizeof(uint32_t); }
};
// a DP workgroup's scratch: the row buffer, one float per end codon and lane
__host__ __device__ inline size_t tiled_state_bytes(int L) { return (static_cast<size_t>(L) / 3 + 8) * kWave * sizeof(float); }

// LDS of a workgroup (static, so that the look-ups' table addresses are instruction offsets): the current tile's sigma tables [KT][64], then the row
// buffer's last kTiledLds / 256 - KT end codons -- every row ends there, so these are the entries passed on most often; the earlier ones live in
// the workgroup's scratch in global memory.  10 KB: sixteen workgroups per CU.
constexpr int kTiledLdsBytes = 10240;

template <int KT, bool SEM>
__device__ __forceinline__ void tiled_body(const NullArgs &A, const uint8_t *__restrict__ blob, const DevBlock *__restrict__ dblocks,
                                           const int *__restrict__ classBlocks, const uint32_t *__restrict__ flags, uint8_t *__restrict__ scratch,
                                           float *__restrict__ maxima) {
  constexpr int WPT = (KT + 4) / 5;
  constexpr int PC = kTiledLdsBytes / (kWave * static_cast<int>(sizeof(float))) - KT;   // end codons of the row buffer kept in LDS
  __shared__ float tab[KT * kWave];
  __shared__ float rowLds[PC * kWave];
  using Regs = KRegs<std::make_integer_sequence<int, KT>>;
  const int lane = threadIdx.x;
  float *__restrict__ partial = reinterpret_cast<float *>(scratch + static_cast<size_t>(blockIdx.x) * A.scratchStride) + lane;   // [end codon][64]
  unsigned int xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  const int home = static_cast<int>(xcc & 7u);
  const int nBlocks = A.nClassBlocks;
  // (the launch for blocks with NaN tables: k_prep_lut counted the class's flagged blocks -- none as a rule, and then no item's flag need be looked at)
  if (A.onlyMask && A.nBlocksPtr && *A.nBlocksPtr == 0) return;
  const int parts = A.comboSplit ? 6 : 1;   // few items: one strand x frame part each (maxima meet in an atomic max), as in k_null
  for (int qi = 0; qi < 8;) {
    const int q = (home + qi) & 7;
    const int g0 = A.gLo + ((q - A.gLo) & 7);
    const int gq = g0 < A.gHi ? ((A.gHi - 1 - g0) >> 3) + 1 : 0;
    const int perBlock = gq * parts;
    const int nq = nBlocks * perBlock;
    int t0 = nq;
    if (nq > 0) {
      if (lane == 0) t0 = static_cast<int>(atomicAdd(A.workCounter + q, 1u));
      t0 = __builtin_amdgcn_readfirstlane(t0);
    }
    if (t0 >= nq) { qi++; continue; }
    const int bi = classBlocks[t0 / perBlock];
    const uint32_t fl = flags[bi];
    if (A.skipMask && (fl & A.skipMask)) continue;
    if (A.onlyMask && !(fl & A.onlyMask)) continue;
    const int sub = t0 % perBlock;
    const int grp = g0 + 8 * (sub / parts);
    const int onlyCombo = A.comboSplit ? sub % parts : -1;
    const DevBlock *__restrict__ db = dblocks + bi;
    const int NK = db->NK, L = db->L, L1 = L + 1, ZW = db->zw_words, zmWords = db->zm_words;
    const TiledLayout lay(NK, KT);
    const int T = lay.T, WT = lay.WT;
    const float Delta = db->Delta, Omega = db->Omega, omega = db->omega, nkf = db->nkf;
    const unsigned long long *__restrict__ zw = reinterpret_cast<const unsigned long long *>(blob + db->off_zw);
    const unsigned long long *__restrict__ zmask = reinterpret_cast<const unsigned long long *>(blob + db->off_zmask);
    const float *__restrict__ lut = reinterpret_cast<const float *>(blob + db->off_lut);
    const uint32_t *__restrict__ codes = reinterpret_cast<const uint32_t *>(A.codesAll + (static_cast<size_t>(t0 / perBlock) * (A.gHi - A.gLo) + static_cast<size_t>(grp - A.gLo)) * A.codesStride) + lane;
    const int sidx = grp * kWave + lane;
    float best = -1.0f;
    const float negTie = -A.tieThr;
    float omega_v, two_v;
    asm volatile("v_mov_b32 %0, %1" : "=v"(omega_v) : "s"(omega));   // per-lane copies: a v_add_f32 with two VGPR operands issues at the full rate (tools/microbench.hip)
    asm volatile("v_mov_b32 %0, 2.0" : "=v"(two_v));
    for (int s = 0; s < 2; s++) {
      for (int f = 0; f < 3; f++) {
        const int sites = (L - f) / 3;
        if (sites <= 0) continue;
        if (onlyCombo >= 0 && onlyCombo != s * 3 + f) continue;
        const unsigned long long *__restrict__ zbase = zw + static_cast<size_t>(s * L1 + 3 + f) * ZW;        // site j: [3 j ZW + word]
        const unsigned long long *__restrict__ zany = zmask + static_cast<size_t>(s * 3 + f) * zmWords;    // bit j: some z != 0 at site j
        const uint32_t *__restrict__ cbase = codes + static_cast<size_t>(s * L1 + 3 + f) * WT * kWave;        // site j: + 3 j WT 64
        const size_t siteStride = static_cast<size_t>(3) * WT * kWave;
        const int jp = sites > PC ? sites - PC : 0;   // row-buffer entries from here on are in LDS
        auto next_event = [&](int j, int end) -> int {   // first site >= j (and < end) with an event, or end
          while (j < end) {
            const unsigned long long mword = zany[j >> 6] >> (j & 63);
            if (mword) { const int e = j + __builtin_ctzll(mword); return e < end ? e : end; }
            j = (j | 63) + 1;
          }
          return end;
        };
        SampleScan st{0.0f, 0.0f, 0u, 0u};
        for (int a = 0; a < sites; a++) {
          sample_scan_row_begin(st, best, static_cast<uint32_t>(a));
          const int jend = (a == sites - 1) ? sites - 1 : sites;   // the frame's final entry is never reported (score.c:892-900)
          const int fe = next_event(a, jend);                      // the row's first codon with a frame shift in some sequence
          for (int t = 0; t < T; t++) {
            const bool first = t == 0, last = t == T - 1;
            const uint32_t *__restrict__ ct = cbase + static_cast<size_t>(t) * WPT * kWave;
            Regs R;
            static_for<KT>([&](auto kc) {
              auto &r = R.template at<decltype(kc)::value>();
              r.s0 = r.s1 = r.s2 = 0.0f;
            });
            // The words of a cell's site and the sum carried into it are fetched TWO cells ahead, into two register sets used in turn: set A holds the
            // cell being scored, set B the next one, and a cell refills its own set for the cell after next as soon as its look-ups have read it --
            // nothing is copied in a loop of cell pairs; a span of odd length ends with one exchange of the two sets.  (One cell of distance left
            // the memory system at ~3.5 TB/s with the SIMDs waiting; rings of three sets, and cells written for every ring position in ONE loop
            // over all kinds of cell, cost the register allocator, held to 128 VGPRs, 44..190 spills inside the cells.)
            uint32_t wA[WPT], wB[WPT];
            float pA, pB;
            auto fetch = [&](int jj, uint32_t (&w)[WPT], float &pv) {
              const int jc = jj < sites ? jj : sites - 1;
#pragma unroll
              for (int x = 0; x < WPT; x++) w[x] = ct[static_cast<size_t>(jc) * siteStride + x * kWave];
              pv = 0.0f;   // the first pass: score.c:830, sum = 0
              if (!first) {
                if (jc >= jp) pv = rowLds[(jc - jp) * kWave + lane];
                else pv = partial[static_cast<size_t>(jc) * kWave];
              }
            };
            fetch(a, wA, pA);
            fetch(a + 1, wB, pB);
            // tile t's tables: [k][64] floats, zeros for the sequences the tile has too many of (behind the row's first fetches: one round trip, not two)
            static_for<KT>([&](auto kc) {
              constexpr int k = decltype(kc)::value;
              const int kk = t * KT + k;
              tab[k * kWave + lane] = kk < NK ? lut[(static_cast<size_t>(s) * NK + kk) * kLutSize + lane] : 0.0f;
            });
            float j2f = static_cast<float>(2 * a + 1);
            // one cell on the register set (w, pv): the look-ups, the set refilled for the cell after next, `step` on every sequence, the sum onward
            auto cell = [&](int j, uint32_t (&w)[WPT], float &pv, auto &&step) {
              float sig[KT];
              static_for<KT>([&](auto kc) {
                constexpr int k = decltype(kc)::value;
                // (the field shifted down and masked is the byte offset of the sigma in the sequence's table)
                const uint32_t at = (w[k / 5] >> (6 * (k % 5))) & 0xfcu;
                sig[k] = *reinterpret_cast<const float *>(reinterpret_cast<const uint8_t *>(tab + k * kWave) + at);
              });
              float sum = pv;
              fetch(j + 2, w, pv);
              static_for<KT>([&](auto kc) {
                constexpr int k = decltype(kc)::value;
                sum = sum + step(kc, sig[k], R.template at<k>());
              });
              if (last) {
                // score.c:843 with S[b][i-1] = S[b][i-2] = 0: max(sum, Delta) / (N - 1).  Only positive values ever reach a sample's result, and for
                // Delta < 0 a positive one is sum / (N - 1): the maximum is left out (batches with Delta >= 0 take the SEM instantiation)
                const float v = SEM ? ref_max<true>(sum, Delta) / nkf : sum / nkf;
                sample_scan_step(st, v, j2f, negTie);
                asm volatile("v_add_f32 %0, %1, %0" : "+v"(j2f) : "v"(two_v));
              } else if (j >= jp) {
                rowLds[(j - jp) * kWave + lane] = sum;
              } else {
                partial[static_cast<size_t>(j) * kWave] = sum;
              }
            };
            int j = a;
            // the cells j .. e - 1 through `step`, in pairs on sets A and B; an odd span's last cell on A, then the sets change places
            auto span = [&](int e, auto &&step) {
              for (; j + 2 <= e; j += 2) {
                cell(j, wA, pA, step);
                cell(j + 1, wB, pB, step);
              }
              if (j < e) {
                cell(j, wA, pA, step);
                j++;
#pragma unroll
                for (int x = 0; x < WPT; x++) { const uint32_t tw = wA[x]; wA[x] = wB[x]; wB[x] = tw; }
                const float tp = pA; pA = pB; pB = tp;
              }
            };
            // the tile's 2 KT bits of a codon's z entries (two bits per sequence, 128 per codon)
            auto tile_z = [&](int je) -> uint32_t {
              const unsigned long long *__restrict__ zj = zbase + static_cast<size_t>(3 * je) * ZW;
              const int sb = 2 * t * KT, wi = sb >> 6, sh = sb & 63;   // the tile's 2 KT bits start in word wi at bit sh; they may run on into the next word
              const unsigned long long lo = zj[wi] >> sh, hi = (sh && wi + 1 < ZW) ? zj[wi + 1] << (64 - sh) : 0ull;
              return static_cast<uint32_t>(lo | hi) & ((1u << (2 * KT)) - 1u);
            };
            auto event_step = [&](uint32_t zt) {
              return [&, zt](auto kc, float sg, auto &r) -> float {
                constexpr int k = decltype(kc)::value;
                pair_step<SEM>(static_cast<int>((zt >> (2 * k)) & 3u), sg, Delta, Omega, omega, r.s0, r.s1, r.s2);
                return ref_max3<SEM>(r.s0, r.s1, r.s2);
              };
            };
            if constexpr (SEM) {   // NaN tables: every cell through the recurrence as written (score.c:512-533)
              while (j < jend) span(j + 1, event_step(tile_z(j)));
            } else {
              // A codon with a frame shift in some sequence (zany) is an event for THIS pass only if the sequence is one of the tile's: for the others'
              // codons z = 0 throughout the tile, and the cell is an ordinary one -- a tile none of whose sequences shifts in the row stays in the
              // two-instruction form below to the row's end.  (score.c:512-533 with z = 0 is the ordinary cell: same operations, same order.)
              uint32_t zt = 0;
              auto next_tile_event = [&](int from) -> int {   // first codon >= from with a frame shift in one of the tile's sequences (zt: its bits), or jend
                for (int e = from;;) {
                  e = next_event(e, jend);
                  if (e >= jend) return jend;
                  zt = tile_z(e);
                  if (zt) return e;
                  e++;
                }
              };
              int e = fe < jend ? next_tile_event(fe) : jend;
              {   // up to there: states 1 and 2 of EVERY sequence of the tile hold the row's running sum of omega
                float wv = 0.0f;
                asm volatile("v_mov_b32 %0, 0" : "=v"(wv));
                span(e, [&](auto kc, float sg, auto &r) -> float {
                  if constexpr (decltype(kc)::value == 0) wv = wv + omega_v;   // (once per cell: the first sequence's turn)
                  r.s0 = r.s0 + sg;
                  return fmaxf(r.s0, wv);
                });
                static_for<KT>([&](auto kc) {
                  auto &r = R.template at<decltype(kc)::value>();
                  r.s1 = r.s2 = wv;
                });
              }
              while (j < jend) {   // j: a codon with z != 0 in one of the tile's sequences (score.c:512-533), wave-uniform branch per sequence
                span(j + 1, event_step(zt));
                e = next_tile_event(j);
                span(e, [&](auto, float sg, auto &r) -> float {
                  r.s0 = r.s0 + sg;
                  r.s1 = r.s1 + omega_v;
                  r.s2 = r.s2 + omega_v;
                  return fmaxf(fmaxf(r.s0, r.s1), r.s2);
                });
              }
            }
          }
          sample_scan_row_end(st, static_cast<uint32_t>(a));
        }
        sample_scan_last(st, best);
      }
    }
    if (sidx < A.sampleN) {
      float *dst = maxima + static_cast<size_t>(db->out_index) * A.sampleN + sidx;
      if (A.comboSplit) { if (best > -1.0f) atomicMax(reinterpret_cast<int *>(dst), __float_as_int(best)); }   // positive floats or the -1 sentinel: integer order == float order
      else *dst = best;
    }
  }
}

template <int KT, bool SEM>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4)))
void k_tiled_dp(NullArgs A, const uint8_t *__restrict__ blob, const DevBlock *__restrict__ dblocks, const int *__restrict__ classBlocks,
                const uint32_t *__restrict__ flags, uint8_t *__restrict__ scratch, float *__restrict__ maxima) {
  tiled_body<KT, SEM>(A, blob, dblocks, classBlocks, flags, scratch, maxima);
}

}  // namespace rc
