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
#include <map>
#include <string>
#include <vector>

#include "rc_host.h"

namespace rc {
namespace {

constexpr double kBlMin = 1e-6, kBlMax = 100.0, kDistMax = 2.0;   // phyml/utilities.h:78,80,88

// allowed-state mask of a character: A=1 C=2 G=4 T=8 (IUPAC codes as in phyml Get_Base_Freqs)
int state_mask(char c) {
  switch (c) {
    case 'A': return 1; case 'C': return 2; case 'G': return 4; case 'T': case 'U': return 8;
    case 'M': return 3; case 'R': return 5; case 'W': return 9; case 'S': return 6; case 'Y': return 10; case 'K': return 12;
    case 'B': return 14; case 'D': return 13; case 'H': return 11; case 'V': return 7;
    default: return 15;   // N X ? - and anything unknown: missing data
  }
}

struct Hky {
  double pi[4], kappa, beta, grp[4];
  void set(const double p[4], double k) {
    for (int i = 0; i < 4; i++) pi[i] = p[i];
    kappa = k;
    const double R = pi[0] + pi[2], Y = pi[1] + pi[3];
    grp[0] = grp[2] = R; grp[1] = grp[3] = Y;
    beta = 1.0 / (2.0 * (R * Y + kappa * (pi[0] * pi[2] + pi[1] * pi[3])));   // mean rate 1
  }
  void probs(double t, double P[4][4]) const {
    const double e1 = std::exp(-beta * t);
    for (int i = 0; i < 4; i++) {
      const double e2 = std::exp(-beta * t * (1.0 + grp[i] * (kappa - 1.0)));
      for (int j = 0; j < 4; j++) {
        if (i == j) P[i][j] = pi[j] + pi[j] * (1.0 / grp[j] - 1.0) * e1 + (grp[j] - pi[j]) / grp[j] * e2;
        else if (((i ^ j) & 1) == 0) P[i][j] = pi[j] + pi[j] * (1.0 / grp[j] - 1.0) * e1 - pi[j] / grp[j] * e2;
        else P[i][j] = pi[j] * (1.0 - e1);
      }
    }
  }
};

// 1-D minimiser on [lo, hi] (golden section with parabolic steps)
template <typename F>
double minimise(F f, double lo, double hi, double x0, double xtol, int itmax, double *fmin) {
  const double cg = 0.3819660112501051;
  double a = lo, b = hi, x = std::min(std::max(x0, lo), hi), w = x, v = x;
  double fx = f(x), fw = fx, fv = fx, d = 0, e = 0;
  for (int it = 0; it < itmax; it++) {
    const double xm = 0.5 * (a + b), tol1 = xtol * std::fabs(x) + 1e-10, tol2 = 2 * tol1;
    if (std::fabs(x - xm) <= tol2 - 0.5 * (b - a)) break;
    bool golden = true;
    if (std::fabs(e) > tol1) {
      double r = (x - w) * (fx - fv), q = (x - v) * (fx - fw), p = (x - v) * q - (x - w) * r;
      q = 2 * (q - r);
      if (q > 0) p = -p;
      q = std::fabs(q);
      const double etemp = e;
      e = d;
      if (!(std::fabs(p) >= std::fabs(0.5 * q * etemp) || p <= q * (a - x) || p >= q * (b - x))) {
        d = p / q;
        const double u = x + d;
        if (u - a < tol2 || b - u < tol2) d = (xm >= x) ? tol1 : -tol1;
        golden = false;
      }
    }
    if (golden) { e = (x >= xm) ? a - x : b - x; d = cg * e; }
    const double u = (std::fabs(d) >= tol1) ? x + d : x + ((d > 0) ? tol1 : -tol1);
    const double fu = f(u);
    if (fu <= fx) {
      if (u >= x) a = x; else b = x;
      v = w; fv = fw; w = x; fw = fx; x = u; fx = fu;
    } else {
      if (u < x) a = u; else b = u;
      if (fu <= fw || w == x) { v = w; fv = fw; w = u; fw = fu; }
      else if (fu <= fv || v == x || v == w) { v = u; fv = fu; }
    }
  }
  if (fmin) *fmin = fx;
  return x;
}

struct Fitter {
  int N = 0, P = 0;                       // tips, site patterns
  std::vector<std::vector<uint8_t>> pat;  // [tip][pattern] state mask
  std::vector<double> w;                  // pattern weights
  double pi[4];
  Hky mod;
  // unrooted binary tree: nodes 0..N-1 tips, N..2N-3 internal; stored rooted at `root` (internal, 3 children)
  int nn = 0, root = 0;
  std::vector<int> parent;
  std::vector<std::vector<int>> child;
  std::vector<double> len;                // branch to parent

  void compress(const std::vector<std::string> &rows) {
    N = static_cast<int>(rows.size());
    const int cols = static_cast<int>(rows[0].size());
    std::map<std::string, int> seen;
    pat.assign(N, {});
    for (int c = 0; c < cols; c++) {
      std::string key(N, ' ');
      for (int r = 0; r < N; r++) key[r] = static_cast<char>(state_mask(rows[r][c]));
      auto it = seen.find(key);
      if (it == seen.end()) {
        seen.emplace(key, P);
        for (int r = 0; r < N; r++) pat[r].push_back(static_cast<uint8_t>(key[r]));
        w.push_back(1.0);
        P++;
      } else w[it->second] += 1.0;
    }
  }

  // empirical base frequencies; ambiguous characters are shared out by the current estimate
  // (8 rounds, as phyml/utilities.c:1926-1990)
  void base_freqs() {
    double f[4] = {0.25, 0.25, 0.25, 0.25};
    for (int it = 0; it < 8; it++) {
      double cnt[4] = {0, 0, 0, 0};
      for (int r = 0; r < N; r++)
        for (int p = 0; p < P; p++) {
          const int m = pat[r][p];
          double tot = 0;
          for (int s = 0; s < 4; s++) if (m >> s & 1) tot += f[s];
          for (int s = 0; s < 4; s++) if (m >> s & 1) cnt[s] += w[p] * f[s] / tot;
        }
      const double sum = cnt[0] + cnt[1] + cnt[2] + cnt[3];
      for (int s = 0; s < 4; s++) f[s] = cnt[s] / sum;
    }
    for (int s = 0; s < 4; s++) pi[s] = f[s];
  }

  // ML distance of two rows from the joint frequencies of their unambiguous sites
  double pair_distance(int x, int y) const {
    double F[4][4] = {{0}}, tot = 0;
    for (int p = 0; p < P; p++) {
      const int a = pat[x][p], b = pat[y][p];
      if ((a & (a - 1)) || (b & (b - 1))) continue;   // ambiguous in either row: hidden
      F[__builtin_ctz(a)][__builtin_ctz(b)] += w[p];
      tot += w[p];
    }
    if (tot <= 0) return -1.0;
    double diff = 0;
    for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) if (i != j) diff += F[i][j];
    if (diff == 0) return 0.0;
    auto nll = [&](double d) {
      double Pm[4][4], s = 0;
      mod.probs(d, Pm);
      for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) if (F[i][j] > 0) s -= F[i][j] * std::log(std::max(Pm[i][j], 1e-300));
      return s;
    };
    const double d = minimise(nll, kBlMin, 5.0, std::min(1.0, 1.5 * diff / tot + 0.01), 1e-6, 100, nullptr);
    return std::min(d, kDistMax);
  }

  // BIONJ (Gascuel 1997) on the distance matrix; builds the rooted-at-last-join representation.
  // Conventions that decide near-ties follow the reference's bundled implementation
  // (phyml/bionj.c:25-395): slots are scanned i ascending, j < i; a pair replaces the incumbent only
  // if its Q is smaller by more than 1e-5; the merged cluster keeps slot i; a lambda outside [0,1]
  // is replaced by 0.5.
  void bionj(const std::vector<std::vector<double>> &D0) {
    nn = 2 * N - 2;
    parent.assign(nn, -1); child.assign(nn, {}); len.assign(nn, kBlMin);
    std::vector<std::vector<double>> D = D0, V = D0;   // by slot
    std::vector<int> node(N), on(N, 1);
    for (int i = 0; i < N; i++) node[i] = i;
    int r = N, next = N;
    while (r > 3) {
      std::vector<double> S(N, 0.0);
      for (int i = 0; i < N; i++) if (on[i]) for (int k = 0; k < N; k++) if (k != i && on[k]) S[i] += D[i][k];
      double qmin = 1e300; int x = -1, y = -1;
      for (int i = 0; i < N; i++) if (on[i]) for (int j = 0; j < i; j++) if (on[j]) {
        const double q = (r - 2.0) * D[i][j] - S[i] - S[j];
        if (q < qmin - 1e-5) { x = i; y = j; qmin = q; }
      }
      const double vxy = V[x][y];
      const double lx = 0.5 * (D[x][y] + (S[x] - S[y]) / (r - 2.0)), ly = 0.5 * (D[x][y] + (S[y] - S[x]) / (r - 2.0));
      double lam = 0.5;
      if (vxy != 0.0) {
        double t = 0;
        for (int k = 0; k < N; k++) if (k != x && k != y && on[k]) t += V[y][k] - V[x][k];
        lam = 0.5 + t / (2.0 * (r - 2) * vxy);
        if (lam > 1.0 || lam < 0.0) lam = 0.5;
      }
      for (int k = 0; k < N; k++) if (k != x && k != y && on[k]) {
        const double d = lam * (D[x][k] - lx) + (1 - lam) * (D[y][k] - ly);
        const double v = lam * V[x][k] + (1 - lam) * V[y][k] - lam * (1 - lam) * vxy;
        D[x][k] = D[k][x] = d;
        V[x][k] = V[k][x] = v;
      }
      const int u = next++;
      parent[node[x]] = u; parent[node[y]] = u; child[u] = {node[x], node[y]};
      len[node[x]] = std::max(lx, kBlMin); len[node[y]] = std::max(ly, kBlMin);
      node[x] = u; on[y] = 0; r--;
    }
    int s3[3], c = 0;
    for (int i = 0; i < N; i++) if (on[i]) s3[c++] = i;
    root = next++;
    const double dxy = D[s3[0]][s3[1]], dxz = D[s3[0]][s3[2]], dyz = D[s3[1]][s3[2]];
    len[node[s3[0]]] = std::max(0.5 * (dxy - dyz + dxz), kBlMin);
    len[node[s3[1]]] = std::max(0.5 * (dyz - dxz + dxy), kBlMin);
    len[node[s3[2]]] = std::max(0.5 * (dxz - dxy + dyz), kBlMin);
    for (int k = 0; k < 3; k++) { parent[node[s3[k]]] = root; child[root].push_back(node[s3[k]]); }
  }

  // flat work arrays [node][pattern][state], allocated once per fit
  std::vector<double> downA, upA, msgA;
  std::vector<int> preorder;
  double *dn(int v) { return downA.data() + static_cast<size_t>(v) * P * 4; }
  double *upp(int v) { return upA.data() + static_cast<size_t>(v) * P * 4; }
  double *ms(int v) { return msgA.data() + static_cast<size_t>(v) * P * 4; }
  const double *dn(int v) const { return downA.data() + static_cast<size_t>(v) * P * 4; }
  const double *upp(int v) const { return upA.data() + static_cast<size_t>(v) * P * 4; }

  void prepare() {
    const size_t sz = static_cast<size_t>(nn) * P * 4;
    downA.assign(sz, 0.0); upA.assign(sz, 0.0); msgA.assign(sz, 0.0);
    preorder.clear();
    std::vector<int> stack{root};
    while (!stack.empty()) { int v = stack.back(); stack.pop_back(); preorder.push_back(v); for (int c : child[v]) stack.push_back(c); }
    for (int t = 0; t < N; t++) {
      double *L = dn(t);
      for (int p = 0; p < P; p++) for (int s = 0; s < 4; s++) L[p * 4 + s] = (pat[t][p] >> s & 1) ? 1.0 : 0.0;
    }
  }

  // conditional likelihoods of the subtree below each node (down) and of the rest of the tree as seen
  // from the top of each node's branch (up, excluding pi)
  void partials() {
    const int PS = P * 4;
    for (auto it = preorder.rbegin(); it != preorder.rend(); ++it) {
      const int v = *it;
      if (v >= N) {
        double *d = dn(v);
        for (int x = 0; x < PS; x++) d[x] = 1.0;
        for (int c : child[v]) { const double *m = ms(c); for (int x = 0; x < PS; x++) d[x] *= m[x]; }
      }
      if (v != root) {
        double Pm[4][4];
        mod.probs(len[v], Pm);
        const double *d = dn(v);
        double *m = ms(v);
        for (int p = 0; p < P; p++) for (int i = 0; i < 4; i++)
          m[p * 4 + i] = Pm[i][0] * d[p * 4] + Pm[i][1] * d[p * 4 + 1] + Pm[i][2] * d[p * 4 + 2] + Pm[i][3] * d[p * 4 + 3];
      }
    }
    for (int v : preorder) {
      if (v == root) continue;
      const int par = parent[v];
      double *u = upp(v);
      for (int x = 0; x < PS; x++) u[x] = 1.0;
      for (int c : child[par]) if (c != v) { const double *m = ms(c); for (int x = 0; x < PS; x++) u[x] *= m[x]; }
      if (par != root) {   // what comes from above the parent, through the parent's own branch
        double Pm[4][4], W[4][4];
        mod.probs(len[par], Pm);
        for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) W[i][j] = Pm[j][i] * mod.pi[j] / mod.pi[i];
        const double *up_par = upp(par);
        for (int p = 0; p < P; p++)
          for (int i = 0; i < 4; i++)
            u[p * 4 + i] *= W[i][0] * up_par[p * 4] + W[i][1] * up_par[p * 4 + 1] + W[i][2] * up_par[p * 4 + 2] + W[i][3] * up_par[p * 4 + 3];
      }
    }
  }

  double edge_lnl(int v, double t) const {
    double Pm[4][4], lnl = 0;
    mod.probs(t, Pm);
    for (int p = 0; p < P; p++) {
      double s = 0;
      for (int i = 0; i < 4; i++) {
        double m = 0;
        for (int j = 0; j < 4; j++) m += Pm[i][j] * dn(v)[p * 4 + j];
        s += mod.pi[i] * upp(v)[p * 4 + i] * m;
      }
      lnl += w[p] * std::log(std::max(s, 1e-300));
    }
    return lnl;
  }

  double lnl() {
    partials();
    const int v = child[root][0];
    return edge_lnl(v, len[v]);
  }

  double optimise() {
    prepare();
    double prev = lnl();
    for (int round = 0; round < 60; round++) {
      for (int v = 0; v < nn; v++) {
        if (v == root) continue;
        partials();
        double fmin;
        len[v] = minimise([&](double t) { return -edge_lnl(v, t); }, kBlMin, kBlMax, len[v], 1e-5, 60, &fmin);
      }
      {
        const double k = minimise([&](double kk) { Hky save = mod; mod.set(pi, kk); const double r = -lnl(); mod = save; return r; },
                                  0.1, 100.0, mod.kappa, 1e-5, 60, nullptr);
        mod.set(pi, k);
      }
      const double cur = lnl();
      if (std::fabs(cur - prev) < 1e-4) { prev = cur; break; }   // PhyML stops at 1e-3 (min_diff_lk_global)
      prev = cur;
    }
    return prev;
  }

  void write(int v, const std::vector<std::string> &names, std::string &out) const {
    char buf[64];
    if (v < N) out += names[v];
    else {
      out += '(';
      for (size_t c = 0; c < child[v].size(); c++) { if (c) out += ','; write(child[v][c], names, out); }
      out += ')';
    }
    if (v != root) { std::snprintf(buf, sizeof buf, ":%f", len[v]); out += buf; }   // phyml/utilities.c:921,952
  }
};

}  // namespace

bool fit_tree(const std::vector<std::string> &rows_in, const std::vector<std::string> &names, std::string &newick,
              float &kappa, double *lnl_out, std::string &err) {
  const int N = static_cast<int>(rows_in.size());
  if (N < 3) { err = "at least three sequences are needed"; return false; }
  std::vector<std::string> rows = rows_in;
  for (auto &r : rows) {
    if (r.size() != rows[0].size()) { err = "Sequences are of unequal length."; return false; }
    for (auto &c : r) c = static_cast<char>(std::toupper(static_cast<unsigned char>(c)));
  }
  Fitter f;
  f.compress(rows);
  f.base_freqs();
  f.mod.set(f.pi, 4.0);   // treeML.c:75
  std::vector<std::vector<double>> D(N, std::vector<double>(N, 0.0));
  double dmax = 0;
  for (int i = 0; i < N; i++) for (int j = i + 1; j < N; j++) { D[i][j] = D[j][i] = f.pair_distance(i, j); dmax = std::max(dmax, D[i][j]); }
  for (int i = 0; i < N; i++) for (int j = 0; j < N; j++) if (i != j && D[i][j] < 0) D[i][j] = std::max(dmax, 0.1);   // no shared sites
  if (std::getenv("RC_TREE_DEBUG")) {
    std::fprintf(stderr, "freqs %.10f %.10f %.10f %.10f\n", f.pi[0], f.pi[1], f.pi[2], f.pi[3]);
    for (int i = 0; i < N; i++) { for (int j = 0; j < N; j++) std::fprintf(stderr, "%.8f ", D[i][j]); std::fprintf(stderr, "\n"); }
  }
  f.bionj(D);
  const double l = f.optimise();
  newick.clear();
  f.write(f.root, names, newick);
  newick += ';';
  kappa = static_cast<float>(f.mod.kappa);
  if (lnl_out) *lnl_out = l;
  return true;
}

// log-likelihood of a GIVEN tree (topology + branch lengths) and kappa under the same model and
// data handling as fit_tree: lets tests compare a PhyML tree with ours on equal terms
bool tree_lnl(const std::vector<std::string> &rows_in, const std::vector<std::string> &names, const char *newick, float kappa,
              double *lnl_out, std::string &err) {
  Tree t;
  if (!parse_newick(newick, t, err)) return false;
  const int N = static_cast<int>(rows_in.size());
  if (static_cast<int>(t.tipname.size()) != N) { err = "tree and alignment differ in the number of sequences"; return false; }
  std::vector<std::string> rows = rows_in;
  for (auto &r : rows) for (auto &c : r) c = static_cast<char>(std::toupper(static_cast<unsigned char>(c)));
  Fitter f;
  f.compress(rows);
  f.base_freqs();
  f.mod.set(f.pi, kappa);
  f.nn = t.nnodes;
  f.parent.assign(f.nn, -1); f.child.assign(f.nn, {}); f.len.assign(f.nn, kBlMin);
  // map: tips of the parsed tree -> row index (ids 0..N-1), internal nodes -> N..
  std::vector<int> id(t.nnodes, -1);
  int next = N;
  for (int q = 0; q < t.nnodes; q++) {
    if (t.tip[q] >= 0) {
      int r = -1;
      for (int k = 0; k < N; k++) if (names[k] == t.tipname[t.tip[q]]) r = k;
      if (r < 0) { err = "tip not found among the row names"; return false; }
      id[q] = r;
    } else id[q] = next++;
  }
  if (next != t.nnodes) { err = "unexpected tree shape"; return false; }
  f.root = id[0];
  for (int q = 1; q < t.nnodes; q++) {
    f.parent[id[q]] = id[t.parent[q]];
    f.child[id[t.parent[q]]].push_back(id[q]);
    f.len[id[q]] = std::max(t.length[q], kBlMin);
  }
  f.prepare();
  *lnl_out = f.lnl();
  return true;
}

}  // namespace rc
