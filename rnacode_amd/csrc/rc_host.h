// rc_host.h -- host-side, block-constant preparation for the device scoring path.
#pragma once
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
  std::vector<std::string> tipname; // by tip number
  std::vector<int> tipnode;         // node index by tip number
};
bool parse_newick(const char *s, Tree &t, std::string &err);

// sigma-code space shared by all blocks of a context (depends only on the BLOSUM choice)
struct PairTable {
  int nB = 0;                 // distinct matrix values
  int bval[32];               // sorted distinct values
  int code_zero = 0, code_stop0 = 0, code_stopk = 0;
  uint8_t pair[64 * 64];      // [codonA][codonB] -> code  (calculateSigma, score.c:406-425, minus the 'N' test)
  int nat_of_slot[64];        // code -> natural number 3*bIdx + (h-1) (or 3nB.. for the specials), -1 = unused
  bool is_score_code(int code) const { return nat_of_slot[code] >= 0 && nat_of_slot[code] < 3 * nB; }
  int h_of_code(int code) const { return nat_of_slot[code] % 3 + 1; }
  int b_of_code(int code) const { return bval[nat_of_slot[code] / 3]; }
  void build(const CodeTables &ct);
};

struct HostBlock {
  int status = RC_OK;
  int N = 0, NK = 0, cols = 0, L = 0;
  int ref_start = 0, ref_length = 0;
  int exact_div = 0;                      // see rc_host.cpp in_div_range
  std::vector<std::string> rows[2];       // upper-cased forward rows, reverse-complement rows
  std::vector<rc_model> models[2];        // getModels for both strands
  std::vector<uint16_t> refcol[2];        // [L+1]
  std::vector<uint64_t> zw[2];            // [L+1]
  std::vector<uint8_t> mask[2];           // [NK][L+1]
  std::vector<float> lut[2];              // [NK][64]
  std::vector<NodeRec> nodes;
  std::vector<uint8_t> qtip;              // [N]
};

// Returns RC_OK, RC_ERR_SKIP (block the reference driver skips) or RC_ERR_ARG / RC_ERR_UNSUPPORTED.
int prepare_block(const rc_block &in, const rc_params &par, const CodeTables &ct, const PairTable &pt,
                  HostBlock &out, std::string &err);

// tree + kappa estimator (rc_tree.cpp on host threads, rc_tree_kernel.hip on the GPU; both run rc_tree_core.h)
struct PatternSet {
  int N = 0, P = 0;
  std::vector<uint8_t> mask;   // [N][P] allowed-state masks of the distinct alignment columns
  std::vector<double> w;       // [P] how many columns show the pattern
};
bool compress_patterns(const std::vector<std::string> &rows, PatternSet &ps, std::string &err);
std::string newick_of(int N, int root, const int *nchild, const int *child, const double *len, const std::vector<std::string> &names);
bool fit_tree(const std::vector<std::string> &rows, const std::vector<std::string> &names, std::string &newick,
              float &kappa, double *lnl_out, std::string &err);

bool tree_lnl(const std::vector<std::string> &rows, const std::vector<std::string> &names, const char *newick, float kappa,
              double *lnl_out, std::string &err);

// extreme_fit.c / RNAcode.c:182 pieces that stay on the host
float pvalue_of(float score, float mu, float lambda);

// smallest float t with: for every float x, (x < t) == ((double)x < d)
float float_threshold_lt(double d);

}  // namespace rc
