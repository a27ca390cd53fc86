// rc_tree_core.h -- the tree + kappa estimator as one piece of source that
// runs on a host thread (rc_tree.cpp, Exec = SerialExec) and as one wavefront per alignment block on
// the GPU (rc_tree_kernel.hip, Exec = WaveExec): same algorithm, the only difference is the order in
// which per-site terms are summed.
//
// Model (what RNAcode gets from its bundled PhyML through src/treeML.c:35-152): BIONJ topology from
// pairwise HKY85 ML distances (kappa 4.0, empirical base frequencies), then maximum-likelihood
// branch lengths and kappa on that topology; ambiguity codes as partial information, gaps as missing.
//
// Exec supplies: lane(), nlanes() (site patterns p = lane, lane + nlanes, ... belong to a lane for the
// whole fit, so per-pattern arrays need no synchronisation), sum(x) / sum2(a, b) (the same totals on every lane),
// exp3 (three exponentials at once), sync() (makes lane 0's serial work visible), mark(phase) / count(counter)
// (profiling builds of the kernel: cycles per phase, see tools/tree_phases.sh; nothing otherwise).
#pragma once
#include <cmath>
#include <cstdint>

#ifdef __HIPCC__
#define RC_HD __host__ __device__
#else
#define RC_HD
#endif

namespace rc {
namespace treefit {

constexpr double kBlMin = 1e-6, kBlMax = 100.0, kDistMax = 2.0;   // phyml/utilities.h:78,80,88
// The kernel keeps BIONJ's bookkeeping in per-lane arrays, so on the device the estimator stops at 64 tips; host code
// (rc_fit_tree, rc_tree_lnl, and the wider blocks that rc_fit_trees_device hands to host threads) goes up to the library's 500 rows.
constexpr int kMaxTipsDevice = 64;
#if defined(__HIP_DEVICE_COMPILE__)
constexpr int kMaxTips = kMaxTipsDevice;
#else
constexpr int kMaxTips = 512;
#endif
constexpr int kMaxNodes = 2 * kMaxTips - 1;   // a rooted input tree (tree_lnl) has 2N-1 nodes

// allowed-state mask of a character: A=1 C=2 G=4 T=8 (IUPAC as in phyml Get_Base_Freqs); the rest is missing data
RC_HD inline int state_mask(char c) {
  switch (c) {
    case 'A': case 'a': return 1; case 'C': case 'c': return 2; case 'G': case 'g': return 4;
    case 'T': case 't': case 'U': case 'u': return 8;
    case 'M': return 3; case 'R': return 5; case 'W': return 9; case 'S': return 6; case 'Y': return 10; case 'K': return 12;
    case 'B': return 14; case 'D': return 13; case 'H': return 11; case 'V': return 7;
    default: return 15;
  }
}

// HKY85 with mean rate 1.  States A C G T = 0..3; purines (A, G: class 0) and pyrimidines (C, T: class 1) are the model's two classes:
// the class of state i is i & 1.  P(t) has three exponentials, e1 = exp(-beta t) and e_g = exp(-beta c_g t) with
// c_g = 1 + G_g (kappa - 1), G_g the class frequency:
//   P_ij = pi_j + pi_j (1/G_g - 1) e1 + ((G_g - pi_j) / G_g) e_g      i == j (g the class of both)
//        = pi_j + pi_j (1/G_g - 1) e1 - (pi_j / G_g) e_g              i != j in one class (transition)
//        = pi_j (1 - e1)                                              different classes (transversion)
// so the image of a conditional-likelihood column d across a branch needs no matrix:
//   (P d)_i = a S + b_g S_g + e_g d_i,   S_g = sum_{j in g} pi_j d_j,  S = S_R + S_Y,  a = 1 - e1,  b_g = (e1 - e_g) / G_g
// (twelve operations and five branch constants instead of sixteen products of a 4 x 4 matrix), and by reversibility the same map
// carries a column DOWN a branch (pi_j P_ji / pi_i = P_ij).
struct Hky {
  double pi[4], kappa, beta, G[2], c[2];
  RC_HD void set(const double p[4], double k) {
    for (int i = 0; i < 4; i++) pi[i] = p[i];
    kappa = k;
    G[0] = pi[0] + pi[2]; G[1] = pi[1] + pi[3];
    beta = 1.0 / (2.0 * (G[0] * G[1] + kappa * (pi[0] * pi[2] + pi[1] * pi[3])));   // mean rate 1
    c[0] = 1.0 + G[0] * (kappa - 1.0); c[1] = 1.0 + G[1] * (kappa - 1.0);
  }
};
struct Coef { double a, bR, bY, eR, eY; };
RC_HD inline Coef coef_of(const Hky &m, double e1, double eR, double eY) {
  Coef c;
  c.a = 1.0 - e1; c.bR = (e1 - eR) / m.G[0]; c.bY = (e1 - eY) / m.G[1]; c.eR = eR; c.eY = eY;
  return c;
}
RC_HD inline void transfer(const Hky &m, const Coef &c, const double d[4], double out[4]) {
  const double sR = m.pi[0] * d[0] + m.pi[2] * d[2], sY = m.pi[1] * d[1] + m.pi[3] * d[3];
  const double base = c.a * (sR + sY);
  const double rR = base + c.bR * sR, rY = base + c.bY * sY;
  out[0] = rR + c.eR * d[0]; out[2] = rR + c.eR * d[2];
  out[1] = rY + c.eY * d[1]; out[3] = rY + c.eY * d[3];
}

// 1-D minimiser on [lo, hi]: golden section with parabolic steps, as a resumable object -- start(), then feed() the value at `u`
// until it returns false; the minimiser is `x`.  (Written this way so that the code that evaluates the objective exists once: the
// kernel's instruction stream has to fit the instruction cache.)
struct Brent {
  double a, b, x, w, v, fx, fw, fv, d, e, xtol, u;
  int it, itmax;
  bool first;
  RC_HD void start(double lo, double hi, double x0, double xtol_, int itmax_) {
    a = lo; b = hi; x = x0 < lo ? lo : (x0 > hi ? hi : x0); w = x; v = x;
    fx = fw = fv = 0; d = 0; e = 0; xtol = xtol_; itmax = itmax_; it = 0; first = true; u = x;
  }
  // takes f(u); true: evaluate at the new u, false: done
  RC_HD bool feed(double fu) {
    if (first) { fx = fw = fv = fu; first = false; }
    else {
      if (fu <= fx) {
        if (u >= x) a = x; else b = x;
        v = w; fv = fw; w = x; fw = fx; x = u; fx = fu;
      } else {
        if (u < x) a = u; else b = u;
        if (fu <= fw || w == x) { v = w; fv = fw; w = u; fw = fu; }
        else if (fu <= fv || v == x || v == w) { v = u; fv = fu; }
      }
      it++;
    }
    if (it >= itmax) return false;
    const double cg = 0.3819660112501051;
    const double xm = 0.5 * (a + b), tol1 = xtol * fabs(x) + 1e-10, tol2 = 2 * tol1;
    if (fabs(x - xm) <= tol2 - 0.5 * (b - a)) return false;
    bool golden = true;
    if (fabs(e) > tol1) {
      double r = (x - w) * (fx - fv), q = (x - v) * (fx - fw), p = (x - v) * q - (x - w) * r;
      q = 2 * (q - r);
      if (q > 0) p = -p;
      q = fabs(q);
      const double etemp = e;
      e = d;
      if (!(fabs(p) >= fabs(0.5 * q * etemp) || p <= q * (a - x) || p >= q * (b - x))) {
        d = p / q;
        const double un = x + d;
        if (un - a < tol2 || b - un < tol2) d = (xm >= x) ? tol1 : -tol1;
        golden = false;
      }
    }
    if (golden) { e = (x >= xm) ? a - x : b - x; d = cg * e; }
    u = (fabs(d) >= tol1) ? x + d : x + ((d > 0) ? tol1 : -tol1);
    return true;
  }
};

template <typename F>
RC_HD double minimise(F f, double lo, double hi, double x0, double xtol, int itmax) {
  Brent br;
  br.start(lo, hi, x0, xtol, itmax);
  double val;
  do { val = f(br.u); } while (br.feed(val));
  return br.x;
}

// The same minimiser on a bracket around the starting point, [x0 / factor, x0 * factor] inside [lo, hi]: a branch length or kappa
// that is being re-optimised sits close to its optimum, and golden-section steps across all of [1e-6, 100] cost three times the
// evaluations.  If the minimum comes out at an end of the bracket that is not an end of [lo, hi], the search goes on from there
// with a wider bracket.
template <typename F>
RC_HD double minimise_local(F f, double lo, double hi, double x0, double factor, double xtol, int itmax) {
  double x = x0 < lo ? lo : (x0 > hi ? hi : x0);
  for (int attempt = 0; attempt < 8; attempt++) {
    double a = x / factor, b = x * factor;
    if (a < lo) a = lo;
    if (b > hi) b = hi;
    Brent br;
    br.start(a, b, x, xtol, itmax);
    double val;
    do { val = f(br.u); } while (br.feed(val));
    x = br.x;
    const double tol = 4 * (xtol * fabs(x) + 1e-10);
    const bool atLow = (x - a) <= tol && a > lo, atHigh = (b - x) <= tol && b < hi;
    if (!atLow && !atHigh) break;
    factor *= 4.0;
  }
  return x;
}

// minimise_local as a resumable object (start(), then feed() the value at `u` until it returns false; the minimiser is `x`): the
// search for kappa below evaluates its objective -- a pass over the whole tree -- at ONE place in the code.
struct LocalMin {
  Brent br;
  double lo, hi, factor, xtol, a, b, x, u;
  int itmax, attempt;
  RC_HD void bracket() {
    a = x / factor; b = x * factor;
    if (a < lo) a = lo;
    if (b > hi) b = hi;
    br.start(a, b, x, xtol, itmax);
    u = br.u;
  }
  RC_HD void start(double lo_, double hi_, double x0, double factor_, double xtol_, int itmax_) {
    lo = lo_; hi = hi_; factor = factor_; xtol = xtol_; itmax = itmax_; attempt = 0;
    x = x0 < lo ? lo : (x0 > hi ? hi : x0);
    bracket();
  }
  RC_HD bool feed(double fu) {
    if (br.feed(fu)) { u = br.u; return true; }
    x = br.x;
    const double tol = 4 * (xtol * fabs(x) + 1e-10);
    const bool atLow = (x - a) <= tol && a > lo, atHigh = (b - x) <= tol && b < hi;
    if ((!atLow && !atHigh) || ++attempt >= 8) return false;
    factor *= 4.0;
    bracket();
    return true;
  }
};

// the state of optimise() between two evaluations of the likelihood: wave-uniform, and kept in memory (on the GPU: LDS) rather than
// in forty registers across the tree pass
struct OptState {
  LocalMin lm;
  double prev, kap, h, f0, f1;
  int what, round, probe;
};

// everything one fit works on; the arrays are provided by the caller (host vectors / the kernel's LDS and, for long blocks, global scratch)
struct Work {
  int N, P, nn, root;
  const uint8_t *mask;   // [N][P] allowed-state masks
  const double *w;       // [P] pattern weights
  double pi[4];
  Hky mod;
  // tree (filled by bionj): nodes 0..N-1 tips, N.. internal
  int *parent;           // [nn]
  int *nchild;           // [nn]
  int *child;            // [nn][3]
  int *preorder;         // [nn]
  double *len;           // [nn]
  double *D, *V;         // [N][N]
  double *tmpD;          // [N]           bionj's row sums
  int *tmpI;             // [2 N + nn]    bionj's slot tables and the stack of its pre-order walk
  Coef *coef;            // [nn]          what each node's branch does to a column under the current model and length
  double *e3;            // [3 nn]        the three exponentials of every branch (coef_all)
  // Conditional-likelihood columns, [4][P] each, of the INTERNAL nodes only (a tip's column is its mask):
  //   dn(v): the subtree below v;  up(v): the rest of the tree as seen from the top of v's branch (without pi);
  // kc: the four per-pattern constants of the branch being optimised (see branch_constants)
  double *dnI, *upI;     // [nn - N][4][P]
  double *kc;            // [4][P]
  OptState *opt;         // optimise()'s state
};

// empirical base frequencies; ambiguous characters are shared out by the current estimate (8 rounds,
// as phyml/utilities.c:1926-1990).  An unambiguous character adds its weight to its state whatever the estimate is, so those are
// counted once; a block without ambiguity codes or gaps needs no rounds at all.
template <typename Exec>
RC_HD void base_freqs(Work &k, Exec &ex) {
  double base[4] = {0, 0, 0, 0}, amb = 0;
  for (int p = ex.lane(); p < k.P; p += ex.nlanes())
    for (int r = 0; r < k.N; r++) {
      const int m = k.mask[static_cast<size_t>(r) * k.P + p];
      if ((m & (m - 1)) == 0) { for (int s = 0; s < 4; s++) base[s] += (m >> s & 1) ? k.w[p] : 0.0; }
      else amb += 1.0;
    }
  for (int s = 0; s < 4; s++) base[s] = ex.sum(base[s]);
  amb = ex.sum(amb);
  double f[4] = {0.25, 0.25, 0.25, 0.25};
  for (int it = 0; it < 8; it++) {
    double cnt[4] = {0, 0, 0, 0};
    if (amb > 0) {
      for (int p = ex.lane(); p < k.P; p += ex.nlanes())
        for (int r = 0; r < k.N; r++) {
          const int m = k.mask[static_cast<size_t>(r) * k.P + p];
          if ((m & (m - 1)) == 0) continue;
          double tot = 0;
          for (int s = 0; s < 4; s++) tot += (m >> s & 1) ? f[s] : 0.0;
          const double q = k.w[p] / tot;
          for (int s = 0; s < 4; s++) cnt[s] += (m >> s & 1) ? q * f[s] : 0.0;
        }
      for (int s = 0; s < 4; s++) cnt[s] = ex.sum(cnt[s]);
    }
    for (int s = 0; s < 4; s++) cnt[s] += base[s];
    const double sum = cnt[0] + cnt[1] + cnt[2] + cnt[3];
    for (int s = 0; s < 4; s++) f[s] = cnt[s] / sum;
    if (!(amb > 0)) break;   // nothing depends on the estimate: the first round's answer is every round's
  }
  for (int s = 0; s < 4; s++) k.pi[s] = f[s];
}

// ML distance of two rows from the joint frequencies of their unambiguous sites (one lane does a pair).  The sixteen entries of
// P(d) are nine distinct functions of d (four identities, four transitions, one transversion factor 1 - e1 times pi_j), so an
// evaluation of the likelihood costs three exponentials and nine logarithms.
RC_HD inline double pair_distance(const Work &k, int x, int y) {
  uint32_t cnt[16];
  for (int c = 0; c < 16; c++) cnt[c] = 0;
  for (int p = 0; p < k.P; p++) {
    const int a = k.mask[static_cast<size_t>(x) * k.P + p], b = k.mask[static_cast<size_t>(y) * k.P + p];
    if ((a & (a - 1)) || (b & (b - 1))) continue;   // ambiguous in either row: hidden
    const int ia = (a == 1) ? 0 : (a == 2) ? 1 : (a == 4) ? 2 : 3, ib = (b == 1) ? 0 : (b == 2) ? 1 : (b == 4) ? 2 : 3;
    const uint32_t wi = static_cast<uint32_t>(k.w[p]);   // a multiplicity: a small integer
    const int idx = ia * 4 + ib;
    // compile-time subscripts: a run-time subscript would move the counters into private memory
    for (int c = 0; c < 16; c++) cnt[c] += (idx == c) ? wi : 0u;
  }
  double F[4][4], tot = 0, diff = 0;
  for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) { F[i][j] = static_cast<double>(cnt[i * 4 + j]); tot += F[i][j]; if (i != j) diff += F[i][j]; }
  if (tot <= 0) return -1.0;
  if (diff == 0) return 0.0;
  const Hky &mod = k.mod;
  // transversions: log P_ij = log pi_j + log(1 - e1); their pi part is a constant of the search
  double Ftv = 0, Ctv = 0;
  for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) if (((i ^ j) & 1) && F[i][j] > 0) { Ftv += F[i][j]; Ctv += F[i][j] * log(mod.pi[j]); }
  auto lg = [](double v) { return log(v > 1e-300 ? v : 1e-300); };
  auto nll = [&](double d) {
    const double e1 = exp(-mod.beta * d), eg[2] = {exp(-mod.beta * mod.c[0] * d), exp(-mod.beta * mod.c[1] * d)};
    double s = Ctv;
    if (Ftv > 0) s += Ftv * lg(1.0 - e1);
    for (int i = 0; i < 4; i++) {
      const int g = i & 1, j = i ^ 2;   // j: the other state of i's class
      const double x1 = mod.pi[i] * (1.0 / mod.G[g] - 1.0);
      if (F[i][i] > 0) s += F[i][i] * lg(mod.pi[i] + x1 * e1 + (mod.G[g] - mod.pi[i]) / mod.G[g] * eg[g]);
      if (F[j][i] > 0) s += F[j][i] * lg(mod.pi[i] + x1 * e1 - mod.pi[i] / mod.G[g] * eg[g]);
    }
    return -s;
  };
  double x0 = 1.5 * diff / tot + 0.01;
  if (x0 > 1.0) x0 = 1.0;
  const double d = minimise(nll, kBlMin, 5.0, x0, 1e-6, 100);
  return d < kDistMax ? d : kDistMax;
}

// BIONJ (Gascuel 1997), serial.  Conventions that decide near-ties follow the reference's bundled
// implementation (phyml/bionj.c:25-395): slots scanned i ascending, j < i; a pair replaces the
// incumbent only if its Q is smaller by more than 1e-5; the merged cluster keeps slot i; a lambda
// outside [0,1] becomes 0.5.
RC_HD inline void bionj(Work &k) {
  const int N = k.N;
  double *D = k.D, *V = k.V;
  for (int i = 0; i < N * N; i++) V[i] = D[i];
  int *node = k.tmpI, *on = k.tmpI + N;
  double *S = k.tmpD;
  for (int i = 0; i < N; i++) { node[i] = i; on[i] = 1; }
  for (int v = 0; v < k.nn; v++) { k.parent[v] = -1; k.nchild[v] = 0; k.len[v] = kBlMin; }
  int r = N, next = N;
  while (r > 3) {
    for (int i = 0; i < N; i++) { S[i] = 0; if (on[i]) for (int c = 0; c < N; c++) if (c != i && on[c]) S[i] += D[i * N + c]; }
    double qmin = 1e300; int x = -1, y = -1;
    for (int i = 0; i < N; i++) if (on[i]) for (int j = 0; j < i; j++) if (on[j]) {
      const double q = (r - 2.0) * D[i * N + j] - S[i] - S[j];
      if (q < qmin - 1e-5) { x = i; y = j; qmin = q; }
    }
    const double vxy = V[x * N + y];
    const double lx = 0.5 * (D[x * N + y] + (S[x] - S[y]) / (r - 2.0)), ly = 0.5 * (D[x * N + y] + (S[y] - S[x]) / (r - 2.0));
    double lam = 0.5;
    if (vxy != 0.0) {
      double t = 0;
      for (int c = 0; c < N; c++) if (c != x && c != y && on[c]) t += V[y * N + c] - V[x * N + c];
      lam = 0.5 + t / (2.0 * (r - 2) * vxy);
      if (lam > 1.0 || lam < 0.0) lam = 0.5;
    }
    for (int c = 0; c < N; c++) if (c != x && c != y && on[c]) {
      const double d = lam * (D[x * N + c] - lx) + (1 - lam) * (D[y * N + c] - ly);
      const double v = lam * V[x * N + c] + (1 - lam) * V[y * N + c] - lam * (1 - lam) * vxy;
      D[x * N + c] = D[c * N + x] = d;
      V[x * N + c] = V[c * N + x] = v;
    }
    const int u = next++;
    k.parent[node[x]] = u; k.parent[node[y]] = u;
    k.child[u * 3 + 0] = node[x]; k.child[u * 3 + 1] = node[y]; k.nchild[u] = 2;
    k.len[node[x]] = lx > kBlMin ? lx : kBlMin; k.len[node[y]] = ly > kBlMin ? ly : kBlMin;
    node[x] = u; on[y] = 0; r--;
  }
  int s3[3], c = 0;
  for (int i = 0; i < N; i++) if (on[i]) s3[c++] = i;
  k.root = next++;
  const double dxy = D[s3[0] * N + s3[1]], dxz = D[s3[0] * N + s3[2]], dyz = D[s3[1] * N + s3[2]];
  const double l0 = 0.5 * (dxy - dyz + dxz), l1 = 0.5 * (dyz - dxz + dxy), l2 = 0.5 * (dxz - dxy + dyz);
  k.len[node[s3[0]]] = l0 > kBlMin ? l0 : kBlMin;
  k.len[node[s3[1]]] = l1 > kBlMin ? l1 : kBlMin;
  k.len[node[s3[2]]] = l2 > kBlMin ? l2 : kBlMin;
  for (int i = 0; i < 3; i++) { k.parent[node[s3[i]]] = k.root; k.child[k.root * 3 + i] = node[s3[i]]; }
  k.nchild[k.root] = 3;
  // pre-order
  int *stack = k.tmpI + 2 * N, sp = 0, n = 0;
  stack[sp++] = k.root;
  while (sp) {
    const int v = stack[--sp];
    k.preorder[n++] = v;
    for (int i = 0; i < k.nchild[v]; i++) stack[sp++] = k.child[v * 3 + i];
  }
}

// ---- conditional likelihoods.  A lane owns its patterns for the whole fit, so columns need no synchronisation: what a lane reads
// it has written itself.

// the column below node v at pattern p: a tip's is its mask, an internal node's is stored
RC_HD inline void load_down(const Work &k, int v, int p, double d[4]) {
  if (v < k.N) {
    const int m = k.mask[static_cast<size_t>(v) * k.P + p];
    for (int s = 0; s < 4; s++) d[s] = (m >> s & 1) ? 1.0 : 0.0;
  } else {
    const double *c = k.dnI + static_cast<size_t>(v - k.N) * 4 * k.P + p;
    for (int s = 0; s < 4; s++) d[s] = c[static_cast<size_t>(s) * k.P];
  }
}

// the three exponentials and the transfer constants of every branch under the current model and lengths: one exponential per
// lane (on the GPU 3 (nn - 1) of them at once instead of one branch after the other on every lane)
template <typename Exec>
RC_HD void coef_all(Work &k, Exec &ex) {
  for (int q = ex.lane(); q < 3 * k.nn; q += ex.nlanes()) {
    const int v = q / 3, kind = q - 3 * v;
    const double rate = kind == 0 ? 1.0 : (kind == 1 ? k.mod.c[0] : k.mod.c[1]);   // (no run-time subscript: the model stays in registers)
    k.e3[q] = exp(-k.mod.beta * rate * k.len[v]);
  }
  ex.sync();
  for (int v = ex.lane(); v < k.nn; v += ex.nlanes()) k.coef[v] = coef_of(k.mod, k.e3[3 * v], k.e3[3 * v + 1], k.e3[3 * v + 2]);
  ex.sync();
}

// out *= the image of node x's column across x's branch, at pattern p
RC_HD inline void times_image(const Work &k, int x, const Coef &cx, int p, double out[4]) {
  double d[4], m[4];
  load_down(k, x, p, d);
  transfer(k.mod, cx, d, m);
  for (int s = 0; s < 4; s++) out[s] *= m[s];
}
// up to three nodes whose images meet in a product (a node's children; a node's siblings): ids and branch constants as plain
// members -- an array indexed by a loop counter would live in the lane's private memory, i.e. in global memory
struct Trio {
  int x0, x1, x2;        // node ids, -1: slot not used
  // (the branch constants are read where they are used: three sets of five held across the pattern loop were thirty registers, and
  // the loop runs once for a block of up to 64 patterns)
  RC_HD void product(const Work &k, int p, double out[4]) const {
    for (int s = 0; s < 4; s++) out[s] = 1.0;
    if (x0 >= 0) times_image(k, x0, k.coef[x0], p, out);
    if (x1 >= 0) times_image(k, x1, k.coef[x1], p, out);
    if (x2 >= 0) times_image(k, x2, k.coef[x2], p, out);
  }
};
// the children of v except `except`, child c in slot c (a slot chosen at run time would be a pointer chosen at run time, and the
// structure would live in private memory)
RC_HD inline Trio children_of(const Work &k, int v, int except) {
  Trio t;
  const int nc = k.nchild[v];
  const int a = k.child[v * 3], b = nc > 1 ? k.child[v * 3 + 1] : -1, c = nc > 2 ? k.child[v * 3 + 2] : -1;
  t.x0 = (nc > 0 && a != except) ? a : -1;
  t.x1 = (b >= 0 && b != except) ? b : -1;
  t.x2 = (c >= 0 && c != except) ? c : -1;
  return t;
}

// dn(v) of one internal node from its children's columns
template <typename Exec>
RC_HD void refresh_down_node(Work &k, Exec &ex, int v) {
  const Trio ch = children_of(k, v, -1);
  double *dst = k.dnI + static_cast<size_t>(v - k.N) * 4 * k.P;
  for (int p = ex.lane(); p < k.P; p += ex.nlanes()) {
    double d[4];
    ch.product(k, p, d);
    for (int s = 0; s < 4; s++) dst[static_cast<size_t>(s) * k.P + p] = d[s];
  }
}

// every dn column from the tips up, and the log-likelihood read at the root; needs coef_all first
template <typename Exec>
RC_HD double full_down(Work &k, Exec &ex) {
  double lnl = 0;
  for (int o = k.nn - 1; o >= 0; o--) {
    const int v = k.preorder[o];
    if (v < k.N) continue;
    if (v != k.root) { refresh_down_node(k, ex, v); continue; }
    const Trio ch = children_of(k, v, -1);
    for (int p = ex.lane(); p < k.P; p += ex.nlanes()) {
      double d[4];
      ch.product(k, p, d);
      const double s = k.mod.pi[0] * d[0] + k.mod.pi[1] * d[1] + k.mod.pi[2] * d[2] + k.mod.pi[3] * d[3];
      lnl += k.w[p] * log(s > 1e-300 ? s : 1e-300);
    }
  }
  return ex.sum(lnl);
}

// log-likelihood of the current tree and model
template <typename Exec>
RC_HD double total_lnl(Work &k, Exec &ex) {
  coef_all(k, ex);
  return full_down(k, ex);
}

// Branch v seen from its two ends: d = dn(v), u = up(v).  Its site likelihood as a function of the branch's length t is
//   s(t) = sum_i pi_i u_i (P(t) d)_i = k0 + k1 e^{-beta t} + k2 e^{-beta c_R t} + k3 e^{-beta c_Y t}
// with, U_g = sum_{i in g} pi_i u_i, S_g = sum_{i in g} pi_i d_i, T_g = sum_{i in g} pi_i u_i d_i, q_g = S_g U_g / G_g:
//   k0 = (S_R + S_Y)(U_R + U_Y),  k1 = q_R + q_Y - k0,  k2 = T_R - q_R,  k3 = T_Y - q_Y
// -- four numbers per pattern, made once per visit of the branch; Newton's iterations then never touch a column.
// Also leaves up(v) in its column if v is an internal node (its children will need it).
template <typename Exec>
RC_HD void branch_constants(Work &k, Exec &ex, int v) {
  const int par = k.parent[v];
  const Trio sib = children_of(k, par, v);
  const bool above = par != k.root;
  const double *upPar = above ? k.upI + static_cast<size_t>(par - k.N) * 4 * k.P : nullptr;
  double *upV = v >= k.N ? k.upI + static_cast<size_t>(v - k.N) * 4 * k.P : nullptr;
  const Hky &m = k.mod;
  for (int p = ex.lane(); p < k.P; p += ex.nlanes()) {
    double u[4];
    sib.product(k, p, u);
    if (above) {   // what comes from above the parent, through the parent's own branch
      double a[4], t[4];
      for (int s = 0; s < 4; s++) a[s] = upPar[static_cast<size_t>(s) * k.P + p];
      transfer(m, k.coef[par], a, t);
      for (int s = 0; s < 4; s++) u[s] *= t[s];
    }
    if (upV) for (int s = 0; s < 4; s++) upV[static_cast<size_t>(s) * k.P + p] = u[s];
    double d[4];
    load_down(k, v, p, d);
    const double u0 = m.pi[0] * u[0], u1 = m.pi[1] * u[1], u2 = m.pi[2] * u[2], u3 = m.pi[3] * u[3];
    const double UR = u0 + u2, UY = u1 + u3;
    const double SR = m.pi[0] * d[0] + m.pi[2] * d[2], SY = m.pi[1] * d[1] + m.pi[3] * d[3];
    const double TR = u0 * d[0] + u2 * d[2], TY = u1 * d[1] + u3 * d[3];
    const double k0 = (SR + SY) * (UR + UY), qR = SR * UR / m.G[0], qY = SY * UY / m.G[1];
    k.kc[p] = k0;
    k.kc[static_cast<size_t>(k.P) + p] = qR + qY - k0;
    k.kc[2 * static_cast<size_t>(k.P) + p] = TR - qR;
    k.kc[3 * static_cast<size_t>(k.P) + p] = TY - qY;
  }
}

// Maximum-likelihood length of the branch whose constants are in kc, by Newton's method on d lnL / dt: t <- t - f'/f'' while
// f'' < 0, a step along the gradient by a factor 4 where the curve is not concave, every step kept within a factor 8 of the current
// length and inside [kBlMin, kBlMax]; stops when the length moves by less than 1e-5 of itself.  Three or four evaluations where a
// bracketing search needs ten: near its optimum lnL(t) is very nearly a parabola.
template <typename Exec>
RC_HD double edge_newton(const Work &k, Exec &ex, double t) {
  const double b = k.mod.beta, cR = k.mod.c[0], cY = k.mod.c[1];
  const double *k0 = k.kc, *k1 = k.kc + k.P, *k2 = k.kc + 2 * static_cast<size_t>(k.P), *k3 = k.kc + 3 * static_cast<size_t>(k.P);
  for (int it = 0; it < 40; it++) {
    double e1, eR, eY, g = 0, h = 0;
    ex.count(10);
    ex.exp3(-b * t, -b * cR * t, -b * cY * t, e1, eR, eY);
    for (int p = ex.lane(); p < k.P; p += ex.nlanes()) {
      const double A = k1[p] * e1, B = k2[p] * eR, C = k3[p] * eY;
      const double s0 = k0[p] + A + B + C;
      if (s0 > 1e-300) {
        const double r = 1.0 / s0;
        const double r1 = -b * (A + cR * B + cY * C) * r;                 // s'/s
        const double r2 = b * b * (A + cR * cR * B + cY * cY * C) * r;    // s''/s
        g += k.w[p] * r1;
        h += k.w[p] * (r2 - r1 * r1);
      }
    }
    ex.sum2(g, h);
    double tn;
    if (h < 0) tn = t - g / h;
    else tn = g > 0 ? t * 4.0 : t * 0.25;
    if (tn > t * 8.0) tn = t * 8.0;
    if (tn < t * 0.125) tn = t * 0.125;
    if (tn < kBlMin) tn = kBlMin;
    if (tn > kBlMax) tn = kBlMax;
    const bool done = fabs(tn - t) <= 1e-5 * t + 1e-10;
    t = tn;
    if (done) break;
  }
  return t;
}

// One pass over the branches in the tree's pre-order (parents before children; what PhyML's own branch optimiser walks too): at
// node v everything above it has its new lengths already and everything below is untouched, so up(v) is one product of the
// siblings' images and the parent's up column, and dn(v) is still the column the last bottom-up pass left.  When a subtree is
// finished (the next node of the order hangs off one of v's ancestors) the dn columns on the way up to that ancestor are rebuilt
// from their children: three images per branch and round, where recomputing "whatever is stale" cost two per level of the tree.
// Needs coef_all + full_down first; leaves every dn column current again.
template <typename Exec>
RC_HD void branch_pass(Work &k, Exec &ex) {
  for (int o = 1; o < k.nn; o++) {
    const int v = k.preorder[o];
    ex.mark(7);
    branch_constants(k, ex, v);
    ex.mark(5);
    const double t = edge_newton(k, ex, k.len[v]);
    ex.mark(6);
    double e1, eR, eY;
    ex.exp3(-k.mod.beta * t, -k.mod.beta * k.mod.c[0] * t, -k.mod.beta * k.mod.c[1] * t, e1, eR, eY);
    k.len[v] = t;                          // (every lane stores the same values)
    k.coef[v] = coef_of(k.mod, e1, eR, eY);
    ex.sync();
    const int stop = o + 1 < k.nn ? k.parent[k.preorder[o + 1]] : k.root;
    for (int a = v; a != stop; a = k.parent[a]) if (a >= k.N) refresh_down_node(k, ex, a);
  }
  ex.mark(7);
}

// coordinate-wise ML: every branch, then kappa, until the log-likelihood moves by less than 1e-4
// (PhyML's Round_Optimize stops at 1e-3, phyml/utilities.c:5207).  Branch lengths by Newton's method (edge_newton).  kappa, whose
// every evaluation is a pass over the whole tree: a Newton step from three likelihood values one per cent apart (the round's closing
// evaluation shows what it achieved, the next round refines it); where the curve is not convex there, or the step would more than
// double or halve kappa, the bracketing minimiser on a bracket around the current value instead (a factor 4 either way in the first
// round, 2 later).  Written as one loop around ONE evaluation of the likelihood -- what the value is for is a small state -- so that
// the code of the tree pass exists once (the kernel's instruction stream has to fit the instruction cache).
template <typename Exec>
RC_HD double optimise(Work &k, Exec &ex) {
  enum { kClose, kProbe, kSearch };   // the value is: the round's closing likelihood / one of the three probes / an evaluation of the search
  OptState &st = *k.opt;
  st.what = kClose; st.round = -1; st.probe = 0;
  st.prev = 0; st.kap = k.mod.kappa; st.h = 0; st.f0 = 0; st.f1 = 0;
  for (;;) {
    ex.sync();
    ex.mark(8);
    const double lnl = total_lnl(k, ex);
    ex.mark(4); ex.count(9);
    bool close = false;
    if (st.what == kClose) {
      if (st.round >= 0 && fabs(lnl - st.prev) < 1e-4) return lnl;
      st.prev = lnl;
      if (++st.round >= 60) return lnl;
      ex.sync();
      ex.count(11);
      branch_pass(k, ex);
      st.kap = k.mod.kappa; st.h = 0.01 * st.kap;
      st.what = kProbe; st.probe = 0;
      k.mod.set(k.pi, st.kap - st.h);
    } else if (st.what == kProbe) {
      const double f = -lnl;
      if (st.probe == 0) { st.f0 = f; st.probe = 1; k.mod.set(k.pi, st.kap); }
      else if (st.probe == 1) { st.f1 = f; st.probe = 2; k.mod.set(k.pi, st.kap + st.h); }
      else {
        const double kap = st.kap, h = st.h, f0 = st.f0, f1 = st.f1;
        const double c = f0 - 2 * f1 + f;
        const double kn = c > 0 ? kap + 0.5 * h * (f0 - f) / c : 0.0;
        if (c > 0 && kn > 0.5 * kap && kn < 2.0 * kap && kn > 0.1 && kn < 100.0) { st.kap = kn; close = true; }
        else { st.lm.start(0.1, 100.0, kap, st.round == 0 ? 4.0 : 2.0, 1e-5, 60); st.what = kSearch; k.mod.set(k.pi, st.lm.u); }
      }
    } else {
      if (st.lm.feed(-lnl)) k.mod.set(k.pi, st.lm.u);
      else { st.kap = st.lm.x; close = true; }
    }
    if (close) { k.mod.set(k.pi, st.kap); st.what = kClose; }
  }
}

// the whole fit; D must hold no data yet.  Returns the log-likelihood.
template <typename Exec>
RC_HD double fit(Work &k, Exec &ex) {
  ex.mark(0);
  base_freqs(k, ex);
  ex.mark(1);
  k.mod.set(k.pi, 4.0);   // treeML.c:75
  const int N = k.N, npairs = N * (N - 1) / 2;
  for (int q = ex.lane(); q < npairs; q += ex.nlanes()) {   // pair q -> (i, j), i < j
    int i = 0, rem = q;
    while (rem >= N - 1 - i) { rem -= N - 1 - i; i++; }
    const int j = i + 1 + rem;
    const double d = pair_distance(k, i, j);
    k.D[i * N + j] = k.D[j * N + i] = d;
  }
  ex.sync();
  ex.mark(2);
  if (ex.lane() == 0) {
    double dmax = 0;
    for (int i = 0; i < N; i++) { k.D[i * N + i] = 0; for (int j = 0; j < N; j++) if (i != j && k.D[i * N + j] > dmax) dmax = k.D[i * N + j]; }
    for (int i = 0; i < N * N; i++) if (k.D[i] < 0) k.D[i] = dmax > 0.1 ? dmax : 0.1;   // pairs without a shared site
    bionj(k);
  }
  ex.sync();
  k.root = ex.broadcast_root(k.root);
  ex.mark(3);
  return optimise(k, ex);
}

}  // namespace treefit
}  // namespace rc
