// rc_device.h -- data layout shared by the host-side block preparation and the HIP kernels.
//
// Everything that is constant for one alignment block (tree, HKY thresholds, gap pattern,
// expected-score tables) is packed by the host into one byte blob per batch; a DevBlock header
// per block holds offsets into that blob.  See DESIGN.md "Data layout in HBM".
#pragma once
#include <cstdint>

namespace rc {

constexpr int kMaxRows = 32;      // N <= 32 (RC_MAX_ROWS)
constexpr int kMaxNodes = 2 * kMaxRows;
constexpr int kWave = 64;         // CDNA wavefront
constexpr int kLutSize = 64;      // sigma codes are indices into a 64-entry per-(strand,k) table
constexpr int kPhaseALds = kMaxNodes * 64 + 64 * 64;   // k_null phase A: threshold table + codon-pair table (bytes)

// One node of the guide tree in Seq-Gen's evolution order (seqgen/evolve.c:400-433: pre-order,
// branch1, branch2, then branch0 of an unrooted root).  thr[p][j]: integer form of the
// cumulative transition row for parent state p (see rc_host.cpp, threshold_of): with
// u = genrand_int32(), child state = base[p] + [u > t0] + [u > t1] + [u > t2];
// u > t3 marks a draw past the end of the cumulative vector (clamped to state 3).
struct alignas(16) U4 { uint32_t x, y, z, w; };

struct alignas(16) NodeRec {
  U4 thr[4];        // thr[p] = {t0, t1, t2, t3} for parent state p
  uint32_t basepack;   // base[p] in bits 2p..2p+1 (non-zero only for degenerate cumulative rows)
  uint8_t parent;      // index of the parent node in this order (root: 0)
  int8_t tiprow;       // alignment row of this tip, -1 for internal nodes
  uint8_t pad[10];
};
static_assert(sizeof(NodeRec) == 80, "NodeRec layout");

struct DevBlock {
  int32_t N, NK, cols, L;
  int32_t nnodes, status, ref_start, ref_length;
  float Delta, Omega, omega, nkf;          // nkf = (float)(N-1)
  float stop0, stopk;
  int32_t code_zero, code_stop0, code_stopk;         // sigma-code values of the special cases
  int32_t exact_div;                                 // 1: score tables outside the range proven for div_by_nk -> use '/'
  // offsets (bytes) into the batch blob
  uint32_t off_nodes;    // NodeRec[nnodes]
  uint32_t off_qtip;     // u8[32]       node index of row r's tip (padded: k_null reads words)
  uint32_t off_refcol;   // u16[2][L+1]  strand s: 0-based column of reference residue i (1..L) in that strand's alignment
  uint32_t off_zw;       // u64[2][L+1]  2 bits per k (bit 2(k-1)): 0 -> z=0, 1 -> z=+1, 2 -> z=-1
  uint32_t off_mask;     // u32[2][L+1][ceil(NK/5)] 6-bit AND masks (field k%5 of word k/5) zeroing the codon fields where row k has '-'
  uint32_t off_lut;      // f32[2][NK][64] sigma value per code
  uint32_t off_chars;    // u8[2][N][cols] native characters (strand 1 = reverse complement, rnaz_utils.c:316)
  uint32_t off_sigma;    // f32[2][NK][L+1] native sigma (written by k_native_sigma)
  uint32_t out_index;    // position of this block in the batch (rows of the maxima matrix)
  uint32_t off_zmask;    // u64[2][3][zm_words] bit j of (strand, frame): some sequence has z != 0 at codon site j
  int32_t zm_words;      // ceil((L/3) / 64)
  uint32_t pad1[1];
};

// HSS record produced by the native scan kernel
struct DevHss {
  int32_t strand, frame, startSite, endSite;
  float score;
};

}  // namespace rc
