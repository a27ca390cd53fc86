// rc_device.h -- data layout shared by the host-side block preparation and the HIP kernels.
//
// Everything that is constant for one alignment block (tree, HKY thresholds, gap pattern,
// background-model inputs) is packed by the host into one byte blob per batch; a DevBlock header
// per block holds offsets into that blob.  The blob has two parts: the host-written part, which is
// the only thing that crosses PCIe (one copy per batch), and a device-only arena behind it for the
// tables the GPU derives itself (gap-pattern tables, reverse-complement rows, expected-score look-up
// tables, native sigma).
// See DESIGN.md "Data layout in HBM".
#pragma once
#include <cstdint>

namespace rc {

constexpr int kMaxRows = 500;     // N <= 500 (RC_MAX_ROWS, the reference's MAX_NUM_NAMES, rnaz_utils.h:7)
constexpr int kTemplRows = 64;    // N <= 64: kernels instantiated per N-1 (k_null<N-1>, k_native_dp<N-1>); above: k_generic_sim + k_generic_dp
constexpr int kFastRows = 32;     // N <= 32: register-resident fast instantiations; 33..64 the EXACT instantiation only
constexpr int kGenericDefaultMinRows = 37;   // rc_ctx::genericMinRows: from this many rows on the generic kernels are the faster ones (RC_GENERIC_MIN_ROWS overrides, 65 = never below 65)
constexpr int kGenericClass = kTemplRows;   // row-count classes are indexed by N-1 (2..63); all wider blocks share this one
// k_tiled_dp<KT> (rc_null_tiled.h): blocks of kTiledMinRows rows and more whose sequences besides the reference are scored in T tiles of KT (12..15)
// with the running sum of a cell carried from tile to tile -- one class per tile size behind the generic one
constexpr int kTiledMinKT = 12, kTiledMaxKT = 15;
constexpr int kTiledClass0 = kGenericClass + 1;                                   // class of KT = kTiledMinKT
constexpr int kClassSlots = kTiledClass0 + (kTiledMaxKT - kTiledMinKT + 1);
constexpr int kTiledDefaultMinRows = 32;     // rc_ctx::tiledMinRows (RC_TILED_MIN_ROWS; 65 = never)
constexpr int kTiledDefaultMaxL = 250;       // rc_ctx::tiledMaxL (RC_TILED_MAX_L): ... and only blocks of at most this many reference residues (200 below 37 rows)
// Tile size for N-1 sequences: the one that costs least by T (9 KT + 20) -- about nine instructions per sequence and cell, twenty per cell and
// pass (the carried sum in and out, the code words, the loop), padded sequences (T KT - (N-1) of them, scored as zeros) included.
constexpr int tiled_kt(int NK) {
  int best = kTiledMaxKT, bestCost = 1 << 30;
  for (int kt = kTiledMaxKT; kt >= kTiledMinKT; kt--) {
    const int T = (NK + kt - 1) / kt, cost = T * (9 * kt + 20);
    if (cost < bestCost) { bestCost = cost; best = kt; }
  }
  return best;
}
// The class of a block: N-1 for the kernels instantiated per row count, one class per tile size, kGenericClass for the rest.  Which kernel is the
// fastest where is measured (tools/rows_sweep.py, profiles/r06/rows_sweep_*.txt; fraction of the FP32 issue roof at 150 / 300 columns):
//   k_null<N-1>      3 (N-1) states in registers: 4 wavefronts per SIMD up to 22 rows (0.45..0.33 / 0.44..0.38), 3 to 30 (0.31 / 0.39..0.35), 2 to 36
//                    (0.24 / 0.30), ONE from 37 on (0.11..0.18) -- and a wavefront issues one vector instruction per ~8 cycles: a SIMD needs four
//   k_generic_dp     states parked in memory, any N, 4 wavefronts: 0.19 / 0.22..0.24 / 0.24..0.26 / 0.27..0.31 at 100 / 150 / 200 / 300 columns
//   k_tiled_dp<KT>   tiles of KT sequences, the cell's sum carried between them, 4 wavefronts: 0.22..0.26 / 0.26..0.29 / 0.26..0.29 / 0.25..0.29 -- it
//                    moves ~85 bytes per cell and sequence where the others move ~50, less where the row buffer's tail in LDS covers most of a row
// So: templated to 31 rows (36 for long blocks), tiled from 32 rows on for blocks of up to 200 reference residues and from 37 rows on up to 250
// (65..111 rows: up to 200 again; from 112 rows on every block), generic for the rest from 37 rows on.
// The tiled kernels score the sequences a tile's size exceeds its share by as zeros, which needs omega <= 0 (a gap state never above 0): other blocks
// stay where they were.  (RC_GENERIC_MIN_ROWS at or below the tiled range's start switches the tiled kernels off as well: the all-generic test mode.)
// Wider blocks (65 rows and more, no per-row-count kernel): tiled up to tiledMaxRows rows and tiledWideMaxL reference residues, generic beyond
// (RC_TILED_MAX_ROWS, RC_TILED_WIDE_MAX_L, RC_TILED_ANY_L_MIN_ROWS).  65 / 80 / 100 / 128 rows, tiled against generic (profiles/r06/rows_sweep_wide_tiled.txt):
// 0.25 / 0.24 / 0.23 / 0.22 against 0.21 / 0.21 / 0.21 / 0.18 at 100 columns, 0.27 / 0.26 / 0.26 / 0.24 against 0.23 / 0.23 / 0.22 / 0.21 at 150,
// level at 200 (0.27 / 0.27 / 0.27 / 0.25 against 0.26 / 0.28 / 0.28 / 0.24), generic ahead at 300 (0.28 against 0.30..0.32).
// From tiledAnyLMinRows rows on the tiled kernels win at every length: the generic kernel parks three states per sequence and eight-cell group in
// memory, and that traffic grows with N where the tiled kernel's carried sum does not (tools/time_wide.py, profiles/r06/very_wide_tiled.txt: 128 x 300
// 255 -> 230 ms, 200 x 90 150 -> 103, 300 x 300 510 -> 289, 500 x 150 451 -> 186; 100 x 300 the other way, 0.31 against 0.285 of the issue roof).
struct ClassRule { int genericMinRows, tiledMinRows, tiledMaxL, tiledMaxRows, tiledWideMaxL, tiledAnyLMinRows; };
constexpr int kTiledDefaultMaxRows = kMaxRows;
constexpr int kTiledDefaultWideMaxL = 200;
constexpr int kTiledDefaultAnyLMinRows = 112;
constexpr ClassRule kDefaultClassRule{kGenericDefaultMinRows, kTiledDefaultMinRows, kTiledDefaultMaxL, kTiledDefaultMaxRows, kTiledDefaultWideMaxL,
                                      kTiledDefaultAnyLMinRows};
constexpr int block_class(int N, int L, float omega, const ClassRule &r) {
  // (below 37 rows the templated kernels still have two wavefronts per SIMD and win on long blocks)
  const int maxL = N > kTemplRows ? (N >= r.tiledAnyLMinRows ? (1 << 30) : r.tiledWideMaxL) : N >= kGenericDefaultMinRows ? r.tiledMaxL : r.tiledMaxL * 4 / 5;
  const int maxRows = r.tiledMaxRows > kTemplRows ? r.tiledMaxRows : kTemplRows;
  if (N >= r.tiledMinRows && r.tiledMinRows <= kTemplRows && N <= maxRows && L <= maxL && r.genericMinRows > r.tiledMinRows && omega <= 0.0f) return kTiledClass0 + tiled_kt(N - 1) - kTiledMinKT;
  return N < r.genericMinRows ? N - 1 : kGenericClass;
}
constexpr int kMaxNodes = 2 * kMaxRows;
constexpr int kWave = 64;         // CDNA wavefront
constexpr int kLutSize = 64;      // sigma codes are indices into a 64-entry per-(strand,k) table
constexpr int kFastNodes = 2 * kFastRows;
constexpr int kPhaseALds = kFastNodes * 64 + 64 * 64;   // k_null phase A: threshold table + codon-pair table (bytes)

// One node of the guide tree in Seq-Gen's evolution order (seqgen/evolve.c:400-433: pre-order,
// branch1, branch2, then branch0 of an unrooted root).  thr[p][j]: integer form of the
// cumulative transition row for parent state p (see rc_host.cpp, threshold_of): with
// u = genrand_int32(), child state = base[p] + [u > t0] + [u > t1] + [u > t2];
// u > t3 marks a draw past the end of the cumulative vector (clamped to state 3).
struct alignas(16) U4 { uint32_t x, y, z, w; };

struct alignas(16) NodeRec {
  U4 thr[4];        // thr[p] = {t0, t1, t2, t3} for parent state p
  uint32_t basepack;   // base[p] in bits 2p..2p+1 (non-zero only for degenerate cumulative rows)
  uint16_t parent;     // index of the parent node in this order (root: 0); the instantiated kernels (< 256 nodes) read its low byte
  int16_t tiprow;      // alignment row of this tip, -1 for internal nodes
  uint32_t gctrl;      // k_generic_sim: where the parent's state is -- node_ctrl(parent) | bit 31 if the parent is one of this node's group of eight
  uint8_t pad[4];
};
// k_generic_sim keeps the node states of a site 2 bits each, eight nodes per 16-bit word, [word][lane] in LDS: bits 0..3 the position
// of node q's two bits in its word, bits 7..13 the word's byte offset (q >> 3) * 128
constexpr uint32_t node_ctrl(uint32_t q) { return (2u * (q & 7u)) | ((q >> 3) << 7); }
static_assert(sizeof(NodeRec) == 80, "NodeRec layout");

// Background model of one (strand, row) (bgModel, score.h:34-44).  The host fills the HKY85 matrix
// P (probHKY needs libm's exp), freqs, kappa and dist; k_prep_models runs calculateBG's two
// 4096-term accumulations (score.c:107-193) and fills scores and probs.
struct alignas(16) ModelRec {
  float P[16];         // P[4*i + j] = probHKY(i, j, dist, freqs, kappa)
  float scores[4];
  float probs[4];
  float kappa, dist;
  float freqs[4];
  float pad[2];
};
static_assert(sizeof(ModelRec) == 128, "ModelRec layout");

struct DevBlock {
  int32_t N, NK, cols, L;
  int32_t nnodes, status, ref_start, ref_length;
  float Delta, Omega, omega, nkf;          // nkf = (float)(N-1)
  float stop0, stopk;
  int32_t code_zero, code_stop0, code_stopk;         // sigma-code values of the special cases
  int32_t zw_words;      // 64-bit words per z entry: ceil(NK / 32)
  int32_t zm_words;      // ceil((L/3) / 64)
  uint32_t out_index;    // position of this block in the batch (rows of the maxima matrix)
  // offsets (bytes) into the batch blob; host-written part
  uint64_t off_nodes;    // NodeRec[nnodes]
  uint64_t off_qtip;     // u8[N padded to 16] node index of row r's tip (k_null reads words); meaningful for N <= 64 (fewer than 256 nodes)
  uint64_t off_qtip16;   // u16[N] the same for any N (k_generic_sim)
  uint64_t off_chars;    // u8[N][cols] native characters, upper-cased (RNAcode.c:121-128)
  uint64_t off_models;   // ModelRec[2][N]
  // device-only arena
  uint64_t off_chars_rev; // u8[N][cols] the reverse complement (rnaz_utils.c:316-348), k_prep_gaps
  uint64_t off_refcol;   // u16[2][L+1]  strand s: 0-based column of reference residue i (1..L) in that strand's alignment (k_prep_gaps)
  uint64_t off_zw;       // u64[2][L+1][zw_words]  2 bits per k (bit 2((k-1)%32) of word (k-1)/32): 0 -> z=0, 1 -> z=+1, 2 -> z=-1 (k_prep_gaps)
  uint64_t off_mask;     // u32[2][L+1][ceil(NK/5)] 6-bit AND masks (field k%5 of word k/5) zeroing the codon fields where row k has '-' (k_prep_gaps)
  uint64_t off_zmask;    // u64[2][3][zm_words] bit j of (strand, frame): some sequence has z != 0 at codon site j (k_prep_gaps)
  uint64_t off_lut;      // f32[2][NK][64] sigma value per code (k_prep_lut)
  uint64_t off_sigma;    // f32[2][NK][L+1] native sigma (k_native_sigma)
  uint64_t off_chain;    // f32[L/3 + 40] W[3 + t] = omega added t times to 0 (W[0..3] = 0): the gap states of a sequence without a frame shift, t cells into a row (k_prep_lut; k_generic_dp)
};

// k_null's staging scratch: the sigma codes of one (strand, reference position) for the 64 samples of a wavefront -- five 6-bit codes
// per dword, [word][lane]; the LAST word of a position is stored as a byte or a halfword where it holds only one or two codes.
constexpr int code_words(int NK) { return (NK + 4) / 5; }
constexpr int code_tail_bytes(int NK) { return NK - 5 * (code_words(NK) - 1) == 1 ? 1 : NK - 5 * (code_words(NK) - 1) == 2 ? 2 : 4; }
constexpr int code_pos_words(int NK) { return (code_words(NK) - 1) * kWave + code_tail_bytes(NK) * (kWave / 4); }   // dwords per (strand, position)

// per-block flag word (device array next to the DevBlocks)
constexpr int kDualRowsMaxNK = 5;        // k_null walks two rows of S per pass for blocks of up to this many sequences besides the reference (rc_null_kernel.h).
                                         // (Round 5 instantiated the L2-fed two-row kernel for 6..12 as well: 138..205 VGPRs = 3 or 2 wavefronts per SIMD, and 22..57 % SLOWER
                                         // than the one-row kernels at 5 or 6 -- 7 / 10 / 13 rows x 150 columns 36.0 / 51.2 / 63.3 -> 44.1 / 65.0 / 85.1 ms, C5 532 -> 832 ms;
                                         // profiles/r05/ab_dual_l2_wide_slower.txt.  These launches live on occupancy.)
constexpr uint32_t kFlagExact = 1u;     // score tables outside the range proven for div_by_nk -> scored by the EXACT instantiation
constexpr uint32_t kFlagNan = 4u;       // NaN among the block's score tables (always with kFlagExact): the reference's NaN-order-dependent MAX macro applies (rc_null_kernel.h, ref_max)
constexpr uint32_t kFlagStopped = 2u;   // --stop-early: enough samples beat the native score, the remaining ones are not simulated

// what a sigma code means (PairTable, shared by all blocks of a context): kind 0 unused, 1 score code
// (value = b - scores[h]), 2 zero, 3 stop in the reference codon, 4 stop in the other codon
struct CodeInfo { int8_t kind, b, h, pad; };

// HSS record produced by the native scan kernel
struct DevHss {
  int32_t strand, frame, startSite, endSite;
  float score;
};

}  // namespace rc
