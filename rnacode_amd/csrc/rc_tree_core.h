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
// whole fit, so per-pattern arrays need no synchronisation), sum(x) (the same total on every lane),
// sync() (makes lane 0's serial work visible).
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

struct Hky {
  double pi[4], kappa, beta, grp[4];
  RC_HD void set(const double p[4], double k) {
    for (int i = 0; i < 4; i++) pi[i] = p[i];
    kappa = k;
    const double R = pi[0] + pi[2], Y = pi[1] + pi[3];
    grp[0] = grp[2] = R; grp[1] = grp[3] = Y;
    beta = 1.0 / (2.0 * (R * Y + kappa * (pi[0] * pi[2] + pi[1] * pi[3])));   // mean rate 1
  }
  RC_HD void probs(double t, double P[4][4]) const {
    const double e1 = exp(-beta * t);
    // purines (A, G) share one exponential, pyrimidines (C, T) the other: grp[0] == grp[2], grp[1] == grp[3]
    const double eR = exp(-beta * t * (1.0 + grp[0] * (kappa - 1.0))), eY = exp(-beta * t * (1.0 + grp[1] * (kappa - 1.0)));
    for (int i = 0; i < 4; i++) {
      const double e2 = (i & 1) ? eY : eR;
      for (int j = 0; j < 4; j++) {
        if (i == j) P[i][j] = pi[j] + pi[j] * (1.0 / grp[j] - 1.0) * e1 + (grp[j] - pi[j]) / grp[j] * e2;
        else if (((i ^ j) & 1) == 0) P[i][j] = pi[j] + pi[j] * (1.0 / grp[j] - 1.0) * e1 - pi[j] / grp[j] * e2;
        else P[i][j] = pi[j] * (1.0 - e1);
      }
    }
  }
  // P(t) with dP/dt and d2P/dt2 (the same three exponentials: d/dt exp(-beta c t) = -beta c exp(-beta c t))
  RC_HD void probs_d(double t, double P[4][4], double P1[4][4], double P2[4][4]) const {
    const double cR = 1.0 + grp[0] * (kappa - 1.0), cY = 1.0 + grp[1] * (kappa - 1.0);
    const double e1 = exp(-beta * t), eR = exp(-beta * t * cR), eY = exp(-beta * t * cY);
    const double d1 = -beta * e1, dd1 = beta * beta * e1;
    for (int i = 0; i < 4; i++) {
      const double c = (i & 1) ? cY : cR, e2 = (i & 1) ? eY : eR;
      const double d2 = -beta * c * e2, dd2 = beta * beta * c * c * e2;
      for (int j = 0; j < 4; j++) {
        if (i == j) {
          const double x = pi[j] * (1.0 / grp[j] - 1.0), y = (grp[j] - pi[j]) / grp[j];
          P[i][j] = pi[j] + x * e1 + y * e2; P1[i][j] = x * d1 + y * d2; P2[i][j] = x * dd1 + y * dd2;
        } else if (((i ^ j) & 1) == 0) {
          const double x = pi[j] * (1.0 / grp[j] - 1.0), y = pi[j] / grp[j];
          P[i][j] = pi[j] + x * e1 - y * e2; P1[i][j] = x * d1 - y * d2; P2[i][j] = x * dd1 - y * dd2;
        } else {
          P[i][j] = pi[j] * (1.0 - e1); P1[i][j] = -pi[j] * d1; P2[i][j] = -pi[j] * dd1;
        }
      }
    }
  }
};

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

// everything one fit works on; the arrays are provided by the caller (host vectors / GPU scratch)
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
  int *tmpI;             // [2 N + nn]    bionj's slot tables and the stack of its pre-order walk (kept out of the kernel's private memory:
                         //               a kernel with 1.8 KB of scratch per lane gets fewer wavefronts from the runtime than its registers allow)
  double *down, *msg, *up;   // [nn][4][P]
  // which of those columns are current (bit v of word v / 64), see refresh_down / refresh_up: a branch-length change makes only
  // part of them stale, and recomputing exactly the stale ones gives the same numbers as recomputing everything
  static constexpr int kSetWords = (kMaxNodes + 63) / 64;
  uint64_t okDn[kSetWords], okMs[kSetWords], okUp[kSetWords];
  // On the device a set is two words: the word is picked by comparison, not by a run-time index, so that the sets stay in registers
  // (a run-time index would put them, and with them a scratch allocation per lane, into private memory).
  RC_HD static bool bit(const uint64_t *m, int v) {
    if constexpr (kSetWords <= 2) { uint64_t w = m[0]; for (int x = 1; x < kSetWords; x++) if ((v >> 6) == x) w = m[x]; return (w >> (v & 63)) & 1ull; }
    else return (m[v >> 6] >> (v & 63)) & 1ull;
  }
  RC_HD static void set_bit(uint64_t *m, int v) {
    if constexpr (kSetWords <= 2) { for (int x = 0; x < kSetWords; x++) m[x] |= ((v >> 6) == x) ? (1ull << (v & 63)) : 0ull; }
    else m[v >> 6] |= 1ull << (v & 63);
  }
  RC_HD static void clear_bit(uint64_t *m, int v) {
    if constexpr (kSetWords <= 2) { for (int x = 0; x < kSetWords; x++) m[x] &= ((v >> 6) == x) ? ~(1ull << (v & 63)) : ~0ull; }
    else m[v >> 6] &= ~(1ull << (v & 63));
  }
  RC_HD double &dn(int v, int s, int p) const { return down[(static_cast<size_t>(v) * 4 + s) * P + p]; }
  RC_HD double &ms(int v, int s, int p) const { return msg[(static_cast<size_t>(v) * 4 + s) * P + p]; }
  RC_HD double &upp(int v, int s, int p) const { return up[(static_cast<size_t>(v) * 4 + s) * P + p]; }
};

// empirical base frequencies; ambiguous characters are shared out by the current estimate (8 rounds,
// as phyml/utilities.c:1926-1990)
template <typename Exec>
RC_HD void base_freqs(Work &k, Exec &ex) {
  double f[4] = {0.25, 0.25, 0.25, 0.25};
  for (int it = 0; it < 8; it++) {
    double cnt[4] = {0, 0, 0, 0};
    for (int p = ex.lane(); p < k.P; p += ex.nlanes())
      for (int r = 0; r < k.N; r++) {
        const int m = k.mask[static_cast<size_t>(r) * k.P + p];
        double tot = 0;
        for (int s = 0; s < 4; s++) if (m >> s & 1) tot += f[s];
        for (int s = 0; s < 4; s++) if (m >> s & 1) cnt[s] += k.w[p] * f[s] / tot;
      }
    for (int s = 0; s < 4; s++) cnt[s] = ex.sum(cnt[s]);
    const double sum = cnt[0] + cnt[1] + cnt[2] + cnt[3];
    for (int s = 0; s < 4; s++) f[s] = cnt[s] / sum;
  }
  for (int s = 0; s < 4; s++) k.pi[s] = f[s];
}

// ML distance of two rows from the joint frequencies of their unambiguous sites (one lane does a pair)
RC_HD inline double pair_distance(const Work &k, int x, int y) {
  double F[4][4] = {{0}}, tot = 0;
  for (int p = 0; p < k.P; p++) {
    const int a = k.mask[static_cast<size_t>(x) * k.P + p], b = k.mask[static_cast<size_t>(y) * k.P + p];
    if ((a & (a - 1)) || (b & (b - 1))) continue;   // ambiguous in either row: hidden
    const int ia = (a == 1) ? 0 : (a == 2) ? 1 : (a == 4) ? 2 : 3, ib = (b == 1) ? 0 : (b == 2) ? 1 : (b == 4) ? 2 : 3;
    // F[ia][ib] += w with compile-time subscripts (x + 0.0 == x for these sums): a run-time subscript would move F into private memory
    for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) F[i][j] += (i == ia && j == ib) ? k.w[p] : 0.0;
    tot += k.w[p];
  }
  if (tot <= 0) return -1.0;
  double diff = 0;
  for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) if (i != j) diff += F[i][j];
  if (diff == 0) return 0.0;
  const Hky &mod = k.mod;
  auto nll = [&](double d) {
    double Pm[4][4], s = 0;
    mod.probs(d, Pm);
    for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) if (F[i][j] > 0) s -= F[i][j] * log(Pm[i][j] > 1e-300 ? Pm[i][j] : 1e-300);
    return s;
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

template <typename Exec>
RC_HD void tip_partials(Work &k, Exec &ex) {
  for (int t = 0; t < k.N; t++)
    for (int p = ex.lane(); p < k.P; p += ex.nlanes())
      for (int s = 0; s < 4; s++) k.dn(t, s, p) = (k.mask[static_cast<size_t>(t) * k.P + p] >> s & 1) ? 1.0 : 0.0;
}

// Conditional likelihoods of the subtree below each node (down), their images across the node's branch (msg), and the rest of
// the tree as seen from the top of each node's branch (up, excluding pi).  What depends on what:
//   down(v), msg(v): the branches strictly below v (msg(v) also v's own)
//   up(v):           every branch outside v's subtree except v's own
// so after branch v changed, msg(v), down/msg of v's ancestors and up of every node that is neither v nor an ancestor of v are
// stale; after the model changed everything is.
RC_HD inline void invalidate_all(Work &k) {
  for (int x = 0; x < Work::kSetWords; x++) k.okDn[x] = k.okMs[x] = k.okUp[x] = 0;
  for (int t = 0; t < k.N; t++) Work::set_bit(k.okDn, t);   // tip columns are data (tip_partials)
}
RC_HD inline void branch_changed(Work &k, int v) {
  uint64_t keep[Work::kSetWords];
  for (int x = 0; x < Work::kSetWords; x++) keep[x] = 0;
  Work::set_bit(keep, v);
  Work::clear_bit(k.okMs, v);
  for (int a = k.parent[v];; a = k.parent[a]) {
    Work::set_bit(keep, a);
    Work::clear_bit(k.okDn, a);
    Work::clear_bit(k.okMs, a);
    if (a == k.root) break;
  }
  for (int x = 0; x < Work::kSetWords; x++) k.okUp[x] &= keep[x];
}

// every stale down / msg column, children before parents
template <typename Exec>
RC_HD void refresh_down(Work &k, Exec &ex) {
  for (int o = k.nn - 1; o >= 0; o--) {
    const int v = k.preorder[o];
    if (v >= k.N && !Work::bit(k.okDn, v)) {
      for (int p = ex.lane(); p < k.P; p += ex.nlanes())
        for (int s = 0; s < 4; s++) {
          double d = 1.0;
          for (int c = 0; c < k.nchild[v]; c++) d *= k.ms(k.child[v * 3 + c], s, p);
          k.dn(v, s, p) = d;
        }
      Work::set_bit(k.okDn, v);
    }
    if (v != k.root && !Work::bit(k.okMs, v)) {
      double Pm[4][4];
      k.mod.probs(k.len[v], Pm);
      for (int p = ex.lane(); p < k.P; p += ex.nlanes()) {
        const double d0 = k.dn(v, 0, p), d1 = k.dn(v, 1, p), d2 = k.dn(v, 2, p), d3 = k.dn(v, 3, p);
        for (int i = 0; i < 4; i++) k.ms(v, i, p) = Pm[i][0] * d0 + Pm[i][1] * d1 + Pm[i][2] * d2 + Pm[i][3] * d3;
      }
      Work::set_bit(k.okMs, v);
    }
  }
}

// the stale up columns on the path from the root to `target` (parents before children); needs refresh_down first
template <typename Exec>
RC_HD void refresh_up(Work &k, Exec &ex, int target) {
  uint64_t path[Work::kSetWords];
  for (int x = 0; x < Work::kSetWords; x++) path[x] = 0;
  for (int a = target; a != k.root; a = k.parent[a]) Work::set_bit(path, a);
  for (int o = 0; o < k.nn; o++) {
    const int v = k.preorder[o];
    if (v == k.root || !Work::bit(path, v) || Work::bit(k.okUp, v)) continue;
    const int par = k.parent[v];
    double W[4][4];
    if (par != k.root) {   // what comes from above the parent, through the parent's own branch
      double Pm[4][4];
      k.mod.probs(k.len[par], Pm);
      for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) W[i][j] = Pm[j][i] * k.mod.pi[j] / k.mod.pi[i];
    }
    for (int p = ex.lane(); p < k.P; p += ex.nlanes()) {
      double u[4] = {1.0, 1.0, 1.0, 1.0};
      for (int c = 0; c < k.nchild[par]; c++) {
        const int sib = k.child[par * 3 + c];
        if (sib != v) for (int s = 0; s < 4; s++) u[s] *= k.ms(sib, s, p);
      }
      if (par != k.root) {
        const double a0 = k.upp(par, 0, p), a1 = k.upp(par, 1, p), a2 = k.upp(par, 2, p), a3 = k.upp(par, 3, p);
        for (int i = 0; i < 4; i++) u[i] *= W[i][0] * a0 + W[i][1] * a1 + W[i][2] * a2 + W[i][3] * a3;
      }
      for (int s = 0; s < 4; s++) k.upp(v, s, p) = u[s];
    }
    Work::set_bit(k.okUp, v);
  }
}

template <typename Exec>
RC_HD double edge_lnl(const Work &k, Exec &ex, int v, double t) {
  double Pm[4][4], lnl = 0;
  k.mod.probs(t, Pm);
  for (int p = ex.lane(); p < k.P; p += ex.nlanes()) {
    const double d0 = k.dn(v, 0, p), d1 = k.dn(v, 1, p), d2 = k.dn(v, 2, p), d3 = k.dn(v, 3, p);
    double s = 0;
    for (int i = 0; i < 4; i++)
      s += k.mod.pi[i] * k.upp(v, i, p) * (Pm[i][0] * d0 + Pm[i][1] * d1 + Pm[i][2] * d2 + Pm[i][3] * d3);
    lnl += k.w[p] * log(s > 1e-300 ? s : 1e-300);
  }
  return ex.sum(lnl);
}

// Maximum-likelihood length of branch v by Newton's method on d lnL / dt (what the columns below and above the branch make it a
// function of t alone): t <- t - f'/f'' while f'' < 0, a step along the gradient by a factor 4 where the curve is not concave,
// every step kept within a factor 8 of the current length and inside [kBlMin, kBlMax]; stops when the length moves by less than
// 1e-5 of itself.  Three or four evaluations where a bracketing search needs ten: near its optimum lnL(t) is very nearly a parabola.
template <typename Exec>
RC_HD double edge_newton(const Work &k, Exec &ex, int v, double t) {
  for (int it = 0; it < 40; it++) {
    double Pm[4][4], P1[4][4], P2[4][4], g = 0, h = 0;
    k.mod.probs_d(t, Pm, P1, P2);
    for (int p = ex.lane(); p < k.P; p += ex.nlanes()) {
      const double d0 = k.dn(v, 0, p), d1 = k.dn(v, 1, p), d2 = k.dn(v, 2, p), d3 = k.dn(v, 3, p);
      double s0 = 0, s1 = 0, s2 = 0;
      for (int i = 0; i < 4; i++) {
        const double u = k.mod.pi[i] * k.upp(v, i, p);
        s0 += u * (Pm[i][0] * d0 + Pm[i][1] * d1 + Pm[i][2] * d2 + Pm[i][3] * d3);
        s1 += u * (P1[i][0] * d0 + P1[i][1] * d1 + P1[i][2] * d2 + P1[i][3] * d3);
        s2 += u * (P2[i][0] * d0 + P2[i][1] * d1 + P2[i][2] * d2 + P2[i][3] * d3);
      }
      if (s0 > 1e-300) {
        const double r1 = s1 / s0;
        g += k.w[p] * r1;
        h += k.w[p] * (s2 / s0 - r1 * r1);
      }
    }
    g = ex.sum(g);
    h = ex.sum(h);
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

// log-likelihood of the current tree and model, read at the root's first branch (its up column needs no other up column)
template <typename Exec>
RC_HD double total_lnl(Work &k, Exec &ex) {
  const int v = k.child[k.root * 3];
  refresh_down(k, ex);
  refresh_up(k, ex, v);
  return edge_lnl(k, ex, v, k.len[v]);
}

// coordinate-wise ML: every branch, then kappa, until the log-likelihood moves by less than 1e-4
// (PhyML's Round_Optimize stops at 1e-3, phyml/utilities.c:5207).  Branch lengths by Newton's method (edge_newton); kappa, whose
// every evaluation is a pass over the whole tree, by the bracketing minimiser on a bracket around the current value
// (minimise_local: a factor 4 either way in the first round, 2 later).
template <typename Exec>
RC_HD double optimise(Work &k, Exec &ex) {
  tip_partials(k, ex);
  invalidate_all(k);
  double prev = total_lnl(k, ex);
  for (int round = 0; round < 60; round++) {
    const double factor = round == 0 ? 4.0 : 2.0;
    for (int v = 0; v < k.nn; v++) {
      if (v == k.root) continue;
      refresh_down(k, ex);
      refresh_up(k, ex, v);
      k.len[v] = edge_newton(k, ex, v, k.len[v]);
      branch_changed(k, v);
    }
    // kappa: every evaluation is a pass over the whole tree.  A Newton step from three values one per cent apart (the round's closing
    // total_lnl shows what it achieved, the next round refines it); where the curve is not convex there, or the step would more than
    // double or halve kappa, the bracketing minimiser instead.
    auto at_kappa = [&](double kk) {
      k.mod.set(k.pi, kk);
      invalidate_all(k);
      return -total_lnl(k, ex);
    };
    double kap = k.mod.kappa;
    {
      const double h = 0.01 * kap;
      double fs[3];
      for (int q = 0; q < 3; q++) fs[q] = at_kappa(kap + (q - 1) * h);   // one call site: the objective's code exists once here
      const double c = fs[0] - 2 * fs[1] + fs[2];
      const double kn = c > 0 ? kap + 0.5 * h * (fs[0] - fs[2]) / c : 0.0;
      if (c > 0 && kn > 0.5 * kap && kn < 2.0 * kap && kn > 0.1 && kn < 100.0) kap = kn;
      else kap = minimise_local(at_kappa, 0.1, 100.0, kap, factor, 1e-5, 60);
    }
    k.mod.set(k.pi, kap);
    invalidate_all(k);
    const double cur = total_lnl(k, ex);
    if (fabs(cur - prev) < 1e-4) { prev = cur; break; }
    prev = cur;
  }
  return prev;
}

// the whole fit; D must hold no data yet.  Returns the log-likelihood.
template <typename Exec>
RC_HD double fit(Work &k, Exec &ex) {
  base_freqs(k, ex);
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
  if (ex.lane() == 0) {
    double dmax = 0;
    for (int i = 0; i < N; i++) { k.D[i * N + i] = 0; for (int j = 0; j < N; j++) if (i != j && k.D[i * N + j] > dmax) dmax = k.D[i * N + j]; }
    for (int i = 0; i < N * N; i++) if (k.D[i] < 0) k.D[i] = dmax > 0.1 ? dmax : 0.1;   // pairs without a shared site
    bionj(k);
  }
  ex.sync();
  k.root = ex.broadcast_root(k.root);
  return optimise(k, ex);
}

}  // namespace treefit
}  // namespace rc
