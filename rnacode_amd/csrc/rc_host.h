// rc_host.h -- host-side, block-constant preparation for the device scoring path.
#pragma once
#include <atomic>
#include <cstdint>
#include <string>
#include <vector>

#include "../../include/rnacode_hip.h"
#include "rc_device.h"
#include "rc_tables.h"

namespace rc {

// Guide tree in evolution order (node 0 = root).
struct Tree {
  int nnodes = 0;
  bool rooted = false;
  std::vector<int> parent;          // parent[q] < q, parent[0] = 0
  std::vector<double> length;       // branch to parent as read by "%lf" (treefile.c:431)
  std::vector<int> tip;             // tip number in order of appearance, -1 for internal
  std::vector<char> namepool;       // tip labels back to back (white space removed, as the reference's reader does)
  std::vector<int> nameOff, nameLen;   // by tip number: where its label sits in namepool
  std::vector<int> tipnode;         // node index by tip number
  int ntips() const { return static_cast<int>(tipnode.size()); }
  bool tip_is(int k, const char *name) const;
  void clear();                     // empty, capacity kept
};
bool parse_newick(const char *s, Tree &t, std::string &err);

// sigma-code space shared by all blocks of a context (depends only on the BLOSUM choice)
struct PairTable {
  int nB = 0;                 // distinct matrix values
  int bval[32];               // sorted distinct values
  int code_zero = 0, code_stop0 = 0, code_stopk = 0;
  uint8_t pair[64 * 64];      // [codonA][codonB] -> code  (calculateSigma, score.c:406-425, minus the 'N' test)
  int nat_of_slot[64];        // code -> natural number 3*bIdx + (h-1) (or 3nB.. for the specials), -1 = unused
  CodeInfo info[64];          // the same, in the form k_prep_lut reads
  bool is_score_code(int code) const { return nat_of_slot[code] >= 0 && nat_of_slot[code] < 3 * nB; }
  int h_of_code(int code) const { return nat_of_slot[code] % 3 + 1; }
  int b_of_code(int code) const { return bval[nat_of_slot[code] / 3]; }
  void build(const CodeTables &ct);
};

// Where prepare_block writes: the host-written part of a batch blob (a caller-owned buffer, pinned in
// stream mode) and the device-only arena behind it.  Blocks are prepared by several threads; each takes its
// share with one atomic add per part.
struct BlobArena {
  uint8_t *host = nullptr;
  size_t hostCap = 0;               // bytes; also the offset of the device-only arena in the device blob
  std::atomic<size_t> hostUsed{0};
  size_t devCap = 0;
  std::atomic<size_t> devUsed{0};
};

// what the host keeps per block once the tables are in the blob
struct BlockMeta {
  int status = RC_OK;
  int N = 0, NK = 0, cols = 0, L = 0;
  int ref_start = 0, ref_length = 0;
};

// upper bounds of a block's share of the two blob parts, from its shape alone (L <= cols, nodes <= 2N-1)
void block_footprint(int N, int cols, size_t *hostBytes, size_t *devBytes);

// getModels' inputs, gap pattern, tree thresholds of one block -> blob + DevBlock header.
// Returns RC_OK, RC_ERR_SKIP (block the reference driver skips) or RC_ERR_ARG / RC_ERR_UNSUPPORTED; never throws.
int prepare_block(const rc_block &in, const rc_params &par, const PairTable &pt, BlobArena &arena, uint32_t out_index,
                  DevBlock &db, BlockMeta &meta, std::string &err);

// true if every gap parameter lies in the range for which the kernels' constant-divisor division was proven
// (tools/verify_const_div.c) and Delta < 0; otherwise the whole batch is scored by the EXACT instantiation
bool params_in_fast_range(const rc_params &par);

// tree + kappa estimator (rc_tree.cpp on host threads, rc_tree_kernel.hip on the GPU; both run rc_tree_core.h)
struct PatternSet {
  int N = 0, P = 0;
  std::vector<uint8_t> mask;   // [N][P] allowed-state masks of the distinct alignment columns
  std::vector<double> w;       // [P] how many columns show the pattern
};
bool compress_patterns(const std::vector<std::string> &rows, PatternSet &ps, std::string &err);
bool compress_patterns(const char *const *rows, int N, int cols, PatternSet &ps, std::string &err);   // rows of `cols` characters each
std::string newick_of(int N, int root, const int *nchild, const int *child, const double *len, const std::vector<std::string> &names);
bool fit_tree(const std::vector<std::string> &rows, const std::vector<std::string> &names, std::string &newick,
              float &kappa, double *lnl_out, std::string &err);

bool tree_lnl(const std::vector<std::string> &rows, const std::vector<std::string> &names, const char *newick, float kappa,
              double *lnl_out, std::string &err);

// extreme_fit.c / RNAcode.c:182 pieces that stay on the host
float pvalue_of(float score, float mu, float lambda);
// which of rc_refexp.h's two variants this host's exp() is, bit for bit, on 20 000 arguments: 2 fused, 1 generic, 0 neither
int exp_mode_of_host();

// smallest float t with: for every float x, (x < t) == ((double)x < d)
float float_threshold_lt(double d);

}  // namespace rc
