// rc_tree.cpp -- tree + kappa estimator for one alignment block.
//
// Stands in for what RNAcode obtains from its bundled PhyML through src/treeML.c:35-152: a BIONJ
// topology from pairwise ML distances and HKY85 maximum-likelihood branch lengths and ts/tv ratio
// kappa (started at 4.0) on that fixed topology, written as Newick with "%f" branch lengths.  It is
// a from-scratch implementation of that model (not of PhyML's code paths): re-entrant, no global
// state, so blocks can be fitted on host threads.  Agreement with PhyML is statistical, not
// bitwise: both converge to the same likelihood optimum within their own tolerances.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "rc_host.h"
#include "rc_tree_core.h"

namespace rc {
namespace {

using treefit::kBlMin;

// one host thread does all the lanes' work
struct SerialExec {
  int lane() const { return 0; }
  int nlanes() const { return 1; }
  double sum(double x) const { return x; }
  void sum2(double &, double &) const {}
  void exp3(double x0, double x1, double x2, double &e0, double &e1, double &e2) const { e0 = std::exp(x0); e1 = std::exp(x1); e2 = std::exp(x2); }
  void sync() const {}
  void mark(int) const {}
  void count(int) const {}
  int broadcast_root(int r) const { return r; }
};

// host-side owner of the arrays a treefit::Work points into
struct Fitter {
  treefit::Work k{};
  std::vector<uint8_t> mask;
  std::vector<double> w, D, V, len, e3, dnI, upI, kc, tmpD;
  std::vector<treefit::Coef> coef;
  treefit::OptState opt{};
  std::vector<int> parent, nchild, child, preorder, tmpI;

  void load(const PatternSet &ps) {
    mask = ps.mask; w = ps.w;
    k.N = ps.N; k.P = ps.P;
  }
  void allocate(int nn) {
    k.nn = nn;
    const int N = k.N;
    D.assign(static_cast<size_t>(N) * N, 0.0); V = D;
    len.assign(nn, kBlMin); parent.assign(nn, -1); nchild.assign(nn, 0); child.assign(static_cast<size_t>(nn) * 3, -1); preorder.assign(nn, 0);
    const size_t sz = static_cast<size_t>(std::max(nn - N, 1)) * 4 * k.P;   // columns of the internal nodes only
    dnI.assign(sz, 0.0); upI.assign(sz, 0.0); kc.assign(static_cast<size_t>(4) * k.P, 0.0);
    coef.assign(nn, treefit::Coef{}); e3.assign(static_cast<size_t>(3) * nn, 0.0);
    k.mask = mask.data(); k.w = w.data(); k.D = D.data(); k.V = V.data(); k.len = len.data();
    k.parent = parent.data(); k.nchild = nchild.data(); k.child = child.data(); k.preorder = preorder.data();
    k.dnI = dnI.data(); k.upI = upI.data(); k.kc = kc.data(); k.coef = coef.data(); k.e3 = e3.data(); k.opt = &opt;
    tmpD.assign(N, 0.0); tmpI.assign(static_cast<size_t>(2) * N + nn, 0);
    k.tmpD = tmpD.data(); k.tmpI = tmpI.data();
  }
};

}  // namespace

// distinct alignment columns (as allowed-state masks) in the order of their first appearance, with their multiplicities
bool compress_patterns(const char *const *rows, int N, int cols, PatternSet &ps, std::string &err) {
  if (N < 3) { err = "at least three sequences are needed"; return false; }
  if (N > treefit::kMaxTips) { err = "too many sequences for the tree estimator"; return false; }
  // per-thread scratch: column keys (N mask bytes each), an open-addressing table of first appearances
  static thread_local std::vector<uint8_t> keys;
  static thread_local std::vector<int> table, first;
  static const struct MaskLut {
    uint8_t m[256];
    MaskLut() { for (int c = 0; c < 256; c++) m[c] = static_cast<uint8_t>(treefit::state_mask(static_cast<char>(std::toupper(c)))); }
  } lut;
  keys.resize(static_cast<size_t>(cols) * N);
  for (int r = 0; r < N; r++) {
    const uint8_t *row = reinterpret_cast<const uint8_t *>(rows[r]);
    for (int c = 0; c < cols; c++) keys[static_cast<size_t>(c) * N + r] = lut.m[row[c]];
  }
  size_t tsize = 16;
  while (tsize < 2 * static_cast<size_t>(cols)) tsize <<= 1;
  table.assign(tsize, -1);
  first.clear();
  ps.N = N; ps.P = 0; ps.w.clear();
  for (int c = 0; c < cols; c++) {
    const uint8_t *key = keys.data() + static_cast<size_t>(c) * N;
    uint64_t h = 1469598103934665603ull;
    for (int r = 0; r < N; r++) { h ^= key[r]; h *= 1099511628211ull; }
    size_t at = static_cast<size_t>(h ^ (h >> 29)) & (tsize - 1);
    for (;;) {
      const int p = table[at];
      if (p < 0) { table[at] = ps.P++; first.push_back(c); ps.w.push_back(1.0); break; }
      if (std::memcmp(keys.data() + static_cast<size_t>(first[p]) * N, key, static_cast<size_t>(N)) == 0) { ps.w[p] += 1.0; break; }
      at = (at + 1) & (tsize - 1);
    }
  }
  ps.mask.resize(static_cast<size_t>(N) * ps.P);
  for (int p = 0; p < ps.P; p++) {
    const uint8_t *key = keys.data() + static_cast<size_t>(first[p]) * N;
    for (int r = 0; r < N; r++) ps.mask[static_cast<size_t>(r) * ps.P + p] = key[r];
  }
  return true;
}

bool compress_patterns(const std::vector<std::string> &rows, PatternSet &ps, std::string &err) {
  const int N = static_cast<int>(rows.size());
  if (N < 3) { err = "at least three sequences are needed"; return false; }
  const int cols = static_cast<int>(rows[0].size());
  std::vector<const char *> ptr(N);
  for (int r = 0; r < N; r++) {
    if (static_cast<int>(rows[r].size()) != cols) { err = "Sequences are of unequal length."; return false; }
    ptr[r] = rows[r].c_str();
  }
  return compress_patterns(ptr.data(), N, cols, ps, err);
}

static void write_node(int v, int N, int root, const int *nchild, const int *child, const double *len,
                       const std::vector<std::string> &names, std::string &out) {
  char buf[64];
  if (v < N) out += names[v];
  else {
    out += '(';
    for (int c = 0; c < nchild[v]; c++) { if (c) out += ','; write_node(child[v * 3 + c], N, root, nchild, child, len, names, out); }
    out += ')';
  }
  if (v != root) { std::snprintf(buf, sizeof buf, ":%f", len[v]); out += buf; }   // phyml/utilities.c:921,952
}

std::string newick_of(int N, int root, const int *nchild, const int *child, const double *len, const std::vector<std::string> &names) {
  std::string out;
  write_node(root, N, root, nchild, child, len, names, out);
  out += ';';
  return out;
}

bool fit_tree(const std::vector<std::string> &rows, const std::vector<std::string> &names, std::string &newick,
              float &kappa, double *lnl_out, std::string &err) {
  PatternSet ps;
  if (!compress_patterns(rows, ps, err)) return false;
  Fitter f;
  f.load(ps);
  f.allocate(2 * ps.N - 2);
  SerialExec ex;
  const double l = treefit::fit(f.k, ex);
  newick = newick_of(ps.N, f.k.root, f.k.nchild, f.k.child, f.k.len, names);
  kappa = static_cast<float>(f.k.mod.kappa);
  if (lnl_out) *lnl_out = l;
  return true;
}

// log-likelihood of a GIVEN tree (topology + branch lengths) and kappa under the same model and
// data handling as fit_tree: lets tests compare a PhyML tree with ours on equal terms
bool tree_lnl(const std::vector<std::string> &rows, const std::vector<std::string> &names, const char *newick, float kappa,
              double *lnl_out, std::string &err) {
  Tree t;
  if (!parse_newick(newick, t, err)) return false;
  const int N = static_cast<int>(rows.size());
  if (t.ntips() != N) { err = "tree and alignment differ in the number of sequences"; return false; }
  PatternSet ps;
  if (!compress_patterns(rows, ps, err)) return false;
  if (t.nnodes > treefit::kMaxNodes) { err = "tree too large"; return false; }
  Fitter f;
  f.load(ps);
  f.allocate(t.nnodes);
  SerialExec ex;
  treefit::base_freqs(f.k, ex);
  f.k.mod.set(f.k.pi, kappa);
  // map: tips of the parsed tree -> row index (ids 0..N-1), internal nodes -> N..
  std::vector<int> id(t.nnodes, -1);
  int next = N;
  for (int q = 0; q < t.nnodes; q++) {
    if (t.tip[q] >= 0) {
      int r = -1;
      for (int i = 0; i < N; i++) if (t.tip_is(t.tip[q], names[i].c_str())) r = i;
      if (r < 0) { err = "tip not found among the row names"; return false; }
      id[q] = r;
    } else id[q] = next++;
  }
  if (next != t.nnodes) { err = "unexpected tree shape"; return false; }
  f.k.root = id[0];
  for (int q = 1; q < t.nnodes; q++) {
    const int par = id[t.parent[q]], v = id[q];
    if (f.nchild[par] >= 3) { err = "only bi- and trifurcations are supported"; return false; }
    f.parent[v] = par;
    f.child[par * 3 + f.nchild[par]++] = v;
    f.len[v] = std::max(t.length[q], kBlMin);
  }
  for (int q = 0, n = 0; q < t.nnodes; q++) f.preorder[n++] = id[q];   // parse order is a pre-order
  // (the parsed tree numbers its internal nodes from N in parse order -- the root is node N, not the last one -- and the columns
  // are indexed by node - N either way)
  *lnl_out = treefit::total_lnl(f.k, ex);
  return true;
}

}  // namespace rc
