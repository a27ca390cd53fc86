// rc_host.cpp -- block-constant preparation on the host (compiled with -ffp-contract=off).
//
// What the reference computes once per alignment block before / around the hot loops is
// restated here from its arithmetic definition and turned into the tables the kernels read:
//   background models      score.c:107-344 (calculateBG, probHKY, countFreqsMono, getModels)
//   tip distances          treeSimulate.c:118-251
//   Newick reader          seqgen/treefile.c:301-530
//   HKY85 branch matrices  seqgen/nucmodels.c:87-131,187-196,302-362, model.c:116-119
//   codon windows, z       misc.c:186-289 (getBlock, pos2col, getSeqLength)
//   reverse complement     rnaz_utils.c:316-348
// Float/double promotions follow the reference's C expressions exactly;
// f32()/f64() below make every rounding step explicit.
#include "rc_host.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace rc {

static inline float f32(double x) { return static_cast<float>(x); }
static inline double f64(float x) { return static_cast<double>(x); }

// ---------------------------------------------------------------------------------- Newick

namespace {
struct NwParser {
  const char *s;
  size_t pos = 0;
  Tree *t;
  std::string err;
  void ws() { while (s[pos] == ' ' || s[pos] == '\t' || s[pos] == '\n' || s[pos] == '\r') pos++; }
  int new_node(int parent) {
    int q = t->nnodes++;
    t->parent.push_back(parent < 0 ? 0 : parent);
    t->length.push_back(0.0);
    t->tip.push_back(-1);
    return q;
  }
  // subtree := '(' subtree ',' subtree [',' subtree]? ')' [label] | name ;  then [':' length ['[' param ']']]
  bool subtree(int parent, bool is_root, int *out) {
    ws();
    int q;
    if (s[pos] == '(') {
      pos++;
      q = new_node(parent);
      int nchild = 0;
      for (;;) {
        int c;
        if (nchild == (is_root ? 3 : 2)) { err = "tree contains a node that is not a bifurcation"; return false; }
        if (!subtree(q, false, &c)) return false;
        nchild++;
        ws();
        if (s[pos] == ',') { pos++; continue; }
        if (s[pos] == ')') { pos++; break; }
        err = "malformed Newick string";
        return false;
      }
      if (nchild < 2) { err = "internal node with fewer than two children"; return false; }
      if (is_root) t->rooted = (nchild == 2);
      while (s[pos] && s[pos] != ':' && s[pos] != ',' && s[pos] != ')' && s[pos] != ';') pos++;  // label
    } else {
      q = new_node(parent);
      std::string name;
      while (s[pos] && s[pos] != ':' && s[pos] != ',' && s[pos] != ')' && name.size() < 256) {
        char ch = s[pos++];
        if (!(ch == ' ' || ch == '\t' || ch == '\n' || ch == '\r')) name.push_back(ch);
      }
      if (name.empty()) { err = "empty tip label"; return false; }
      t->tip[q] = static_cast<int>(t->tipname.size());
      t->tipname.push_back(name);
      t->tipnode.push_back(q);
    }
    ws();
    if (s[pos] == ':') {
      pos++;
      char *end = nullptr;
      double len = std::strtod(s + pos, &end);
      if (end == s + pos) { err = "unable to read branch length"; return false; }
      pos = static_cast<size_t>(end - s);
      t->length[q] = len;
      ws();
      if (s[pos] == '[') { while (s[pos] && s[pos] != ']') pos++; if (s[pos]) pos++; }
    }
    *out = q;
    return true;
  }
};
}  // namespace

bool parse_newick(const char *s, Tree &t, std::string &err) {
  t = Tree();
  if (!s) { err = "no tree"; return false; }
  NwParser p{s, 0, &t, {}};
  while (s[p.pos] && s[p.pos] != '(') p.pos++;
  if (!s[p.pos]) { err = "no tree found"; return false; }
  int root;
  if (!p.subtree(-1, true, &root)) { err = p.err; return false; }
  t.length[0] = 0.0;
  return true;
}

// ---------------------------------------------------------------------------------- models

// countFreqsMono (score.c:255-280): all rows, '-' skipped, unknown characters count as A
static void mono_freqs(const std::vector<std::string> &rows, float freqs[4]) {
  unsigned long cnt[4] = {0, 0, 0, 0}, total = 0;
  for (const auto &r : rows)
    for (unsigned char c : r) {
      if (c == '-') continue;
      cnt[nt_code(c)]++;
      total++;
    }
  // the reference increments float counters; counts stay far below 2^24 so they are exact
  for (int i = 0; i < 4; i++) freqs[i] = static_cast<float>(cnt[i]) / static_cast<float>(total);
}

// patristic distance row 0 <-> every row, float accumulator += double length (treeSimulate.c:164-182)
static bool tip_distances(const Tree &t, const std::vector<int> &rowtip, std::vector<float> &dist) {
  const int N = static_cast<int>(rowtip.size());
  dist.assign(N, 0.0f);
  std::vector<char> mark(t.nnodes);
  for (int k = 1; k < N; k++) {
    std::fill(mark.begin(), mark.end(), 0);
    int a = t.tipnode[rowtip[0]], b = t.tipnode[rowtip[k]];
    for (int x = a;; x = t.parent[x]) { mark[x] = 1; if (x == 0) break; }
    int lca = b;
    while (!mark[lca]) lca = t.parent[lca];
    float sum = 0.0f;
    for (int x = a; x != lca; x = t.parent[x]) sum = f32(f64(sum) + t.length[x]);
    for (int x = b; x != lca; x = t.parent[x]) sum = f32(f64(sum) + t.length[x]);
    dist[k] = sum;
  }
  return true;
}

// probHKY (score.c:204-244) for all 16 entries at once
static void hky_probs(float d, const float pi[4], float kappa, float P[4][4]) {
  const float piA = pi[0], piC = pi[1], piG = pi[2], piT = pi[3];
  const float piR = piA + piG;
  const float piY = piT + piC;
  float tv_sum = piA * piC;
  tv_sum = tv_sum + piC * piG;
  tv_sum = tv_sum + piA * piT;
  tv_sum = tv_sum + piG * piT;
  const float ts_sum = piC * piT + piA * piG;
  const float denom = tv_sum + kappa * ts_sum;
  const float r = f32(1.0 / (2.0 * f64(denom)));
  const float l = r * d;
  const float k1 = kappa * piY + piR;
  const float k2 = kappa * piR + piY;
  const float e1 = f32(std::exp(f64(-l)));
  const float e22 = f32(std::exp(f64(-k2 * l)));
  const float e21 = f32(std::exp(f64(-k1 * l)));
  const float yr = piY / piR, ry = piR / piY;
  auto stay = [&](float p, float ratio, float q, float e) {  // p*(1.+ratio*e1) + q*e
    return f32(f64(p) * (1.0 + f64(ratio * e1)) + f64(q * e));
  };
  auto ts = [&](float p, float ratio, float q, float e) {    // p*(1.+ratio*e1) - q*e
    return f32(f64(p) * (1.0 + f64(ratio * e1)) - f64(q * e));
  };
  auto tv = [&](float p) { return f32(f64(p) * (1.0 - f64(e1))); };
  P[0][0] = stay(piA, yr, piG / piR, e22);
  P[0][2] = ts(piG, yr, piG / piR, e22);
  P[2][0] = ts(piA, yr, piA / piR, e22);
  P[2][2] = stay(piG, yr, piA / piR, e22);
  P[1][1] = stay(piC, ry, piT / piY, e21);
  P[1][3] = ts(piT, ry, piT / piY, e21);
  P[3][1] = ts(piC, ry, piC / piY, e21);
  P[3][3] = stay(piT, ry, piC / piY, e21);
  P[0][1] = P[2][1] = tv(piC);
  P[0][3] = P[2][3] = tv(piT);
  P[1][0] = P[3][0] = tv(piA);
  P[1][2] = P[3][2] = tv(piG);
}

// calculateBG (score.c:107-193): two 4096-term float accumulations in the loop order a1,a2,a3,b1,b2,b3
static void expected_scores(rc_model &m, const CodeTables &ct) {
  float P[4][4];
  hky_probs(m.dist, m.freqs, m.kappa, P);
  float fA[64], pAB;
  for (int a = 0; a < 64; a++) fA[a] = m.freqs[a >> 4] * m.freqs[(a >> 2) & 3] * m.freqs[a & 3];
  float probStop = 0.0f;
  for (int a = 0; a < 64; a++)
    for (int b = 0; b < 64; b++) {
      if (ct.pep[a] != -1 && ct.pep[b] != -1) continue;
      pAB = P[a >> 4][b >> 4] * P[(a >> 2) & 3][(b >> 2) & 3] * P[a & 3][b & 3];
      pAB = pAB * fA[a];
      probStop = probStop + pAB;
    }
  float counts[4] = {0, 0, 0, 0}, scores[4] = {0, 0, 0, 0};
  const float keep = 1.0f - probStop;   // (1 - probStop): int 1 converted to float
  for (int a = 0; a < 64; a++) {
    if (ct.pep[a] == -1) continue;
    for (int b = 0; b < 64; b++) {
      if (ct.pep[b] == -1) continue;
      const int x = a ^ b;
      const int h = ((x & 48) != 0) + ((x & 12) != 0) + ((x & 3) != 0);
      pAB = P[a >> 4][b >> 4] * P[(a >> 2) & 3][(b >> 2) & 3] * P[a & 3][b & 3];
      pAB = pAB * fA[a];
      pAB = pAB / keep;
      const float sc = static_cast<float>(ct.blosum[ct.pep[a]][ct.pep[b]]);
      counts[h] = counts[h] + pAB;
      scores[h] = scores[h] + sc * pAB;
    }
  }
  for (int h = 0; h < 4; h++) {
    m.scores[h] = scores[h] / counts[h];
    m.probs[h] = counts[h];
  }
}

// ---------------------------------------------------------------------------------- Seq-Gen HKY85

namespace {
struct SeqGenHky {
  double fA, fC, fG, fT, fR, fY;
  double t1[4], t2[4], t3[4];
  double beta, betaR, betaY;
  double addFreq[4];
  SeqGenHky(const float pi[4], float kap) {
    fA = f64(pi[0]); fC = f64(pi[1]); fG = f64(pi[2]); fT = f64(pi[3]);   // treeSimulate.c:70-73
    fR = fA + fG;
    fY = fC + fT;
    const double fAG = fA * fG, fCT = fC * fT;
    const double tstv = f64(kap) * (fAG + fCT) / (fR * fY);                 // treeSimulate.c:80
    const double f[4] = {fA, fC, fG, fT};
    const double grp[4] = {fR, fY, fR, fY};
    for (int x = 0; x < 4; x++) {                                           // nucmodels.c:105-116
      t1[x] = f[x] * ((1 / grp[x]) - 1);
      t2[x] = (grp[x] - f[x]) / grp[x];
      t3[x] = f[x] / grp[x];
    }
    const double kappa = (tstv * fR * fY) / (fAG + fCT);                    // nucmodels.c:122
    beta = -1.0 / (2 * (fR * fY + kappa * (fAG + fCT)));
    betaR = beta * (1.0 + fR * (kappa - 1));
    betaY = beta * (1.0 + fY * (kappa - 1));
    addFreq[0] = fA;                                                        // model.c:116-119
    addFreq[1] = addFreq[0] + fC;
    addFreq[2] = addFreq[1] + fG;
    addFreq[3] = addFreq[2] + fT;
  }
  // cumulative rows of P(len) (nucmodels.c:187-196,302-362)
  void cumulative(double len, double M[4][4]) const {
    const double aa = std::exp(beta * len), bbR = std::exp(betaR * len), bbY = std::exp(betaY * len);
    const double f[4] = {fA, fC, fG, fT};
    const double bb[4] = {bbR, bbY, bbR, bbY};
    for (int i = 0; i < 4; i++)
      for (int j = 0; j < 4; j++) {
        if (i == j) M[i][j] = f[j] + (t1[j] * aa) + (t2[j] * bb[j]);
        else if (((i ^ j) & 1) == 0) M[i][j] = f[j] + (t1[j] * aa) - (t3[j] * bb[j]);
        else M[i][j] = f[j] * (1 - aa);
      }
    for (int i = 0; i < 4; i++) { M[i][1] += M[i][0]; M[i][2] += M[i][1]; M[i][3] += M[i][2]; }
  }
};

// Seq-Gen draws r = genrand_int32() * (1.0/4294967295.0) and walks the cumulative vector while
// r > P[j] (evolve.c:167-175, twister.c:162-166).  r is monotone in the 32-bit integer u, so the
// comparison has an exact integer form: r > P  <=>  u > threshold_of(P).
// Returns the largest u with u*c <= P, or -1 if even u = 0 gives r > P.
int64_t threshold_of(double P) {
  const double c = 1.0 / 4294967295.0;
  if (!(0.0 <= P)) return -1;
  if (4294967295.0 * c <= P) return 4294967295LL;
  int64_t lo = 0, hi = 4294967295LL;   // lo satisfies u*c <= P, hi does not
  while (hi - lo > 1) {
    const int64_t mid = lo + (hi - lo) / 2;
    if (static_cast<double>(static_cast<uint32_t>(mid)) * c <= P) lo = mid; else hi = mid;
  }
  return lo;
}

void pack_thresholds(const double cum[4], U4 *out, uint32_t *base) {
  uint32_t thr[4];
  int64_t T[4];
  for (int j = 0; j < 4; j++) T[j] = threshold_of(cum[j]);
  for (int j = 1; j < 4; j++) T[j] = std::max(T[j], T[j - 1]);   // "first j with r <= P[j]" needs only the running max
  int b = 0;
  while (b < 3 && T[b] < 0) b++;
  *base = static_cast<uint32_t>(b);
  int m = 0;
  for (int j = b; j < 3; j++) thr[m++] = static_cast<uint32_t>(T[j]);
  for (; m < 3; m++) thr[m] = 0xFFFFFFFFu;
  thr[3] = T[3] < 0 ? 0u : static_cast<uint32_t>(T[3]);
  out->x = thr[0]; out->y = thr[1]; out->z = thr[2]; out->w = thr[3];
}
}  // namespace

// ---------------------------------------------------------------------------------- sigma codes

void PairTable::build(const CodeTables &ct) {
  bool seen[64] = {false};
  for (int i = 0; i < 20; i++)
    for (int j = 0; j < 20; j++) seen[ct.blosum[i][j] + 32] = true;
  nB = 0;
  for (int v = -32; v < 32; v++) if (seen[v + 32]) bval[nB++] = v;
  const int nCodes = 3 * nB + 3;
  // "natural" numbering: 3*bIdx + (h-1), then zero / stop0 / stopk
  auto natural = [&](int a, int b) {
    const int x = a ^ b;
    const int h = ((x & 48) != 0) + ((x & 12) != 0) + ((x & 3) != 0);
    if (h == 0) return 3 * nB;                          // score.c:409
    if (ct.pep[a] == -1) return 3 * nB + 1;             // score.c:414
    if (ct.pep[b] == -1) return 3 * nB + 2;             // score.c:418
    const int v = ct.blosum[ct.pep[a]][ct.pep[b]];
    int bi = 0;
    while (bval[bi] != v) bi++;
    return 3 * bi + (h - 1);
  };
  // The kernel looks sigma up with ds_bpermute, whose source lanes l and l+32 share a bank: two
  // DIFFERENT codes in one bank cost an extra LDS cycle per wavefront.  Number the codes so that
  // the 32 most frequent ones (under a generic neutral model: uniform codons, 20 % per-site
  // divergence) get distinct banks and the rare rest shares banks with the rarest of those.
  std::vector<double> weight(nCodes, 0.0);
  for (int a = 0; a < 64; a++)
    for (int b = 0; b < 64; b++) {
      const int x = a ^ b;
      const int h = ((x & 48) != 0) + ((x & 12) != 0) + ((x & 3) != 0);
      weight[natural(a, b)] += std::pow(0.8, 3 - h) * std::pow(0.2 / 3.0, h);
    }
  std::vector<int> order(nCodes);
  for (int i = 0; i < nCodes; i++) order[i] = i;
  std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return weight[x] > weight[y]; });
  std::vector<int> slot(nCodes);
  for (int r = 0; r < nCodes; r++) slot[order[r]] = r < 32 ? r : 32 + (31 - (r - 32));
  for (int i = 0; i < 64; i++) { nat_of_slot[i] = -1; }
  for (int i = 0; i < nCodes; i++) nat_of_slot[slot[i]] = i;
  code_zero = slot[3 * nB]; code_stop0 = slot[3 * nB + 1]; code_stopk = slot[3 * nB + 2];
  for (int a = 0; a < 64; a++)
    for (int b = 0; b < 64; b++) pair[a * 64 + b] = static_cast<uint8_t>(slot[natural(a, b)]);
}

// ---------------------------------------------------------------------------------- gap pattern

// getBlock (misc.c:186-246) reduced to what depends on the gap pattern only: z for every
// reference position i >= 3 and sequence k, and which of the three codon columns are '-' in k.
static void gap_tables(const std::vector<std::string> &rows, int L, std::vector<uint16_t> &refcol,
                       std::vector<uint64_t> &zw, std::vector<uint8_t> &mask) {
  const int N = static_cast<int>(rows.size()), NK = N - 1;
  const std::string &r0 = rows[0];
  refcol.assign(L + 1, 0);
  for (int c = 0, p = 0; c < static_cast<int>(r0.size()); c++)
    if (r0[c] != '-') { p++; if (p <= L) refcol[p] = static_cast<uint16_t>(c); }
  zw.assign(L + 1, 0);
  mask.assign(static_cast<size_t>(NK) * (L + 1), 0);
  for (int i = 3; i <= L; i++) {
    const int start = (i > 3) ? refcol[i - 3] + 1 : 0, end = refcol[i];
    int g0 = 0;
    for (int c = start; c <= end; c++) g0 += (r0[c] == '-');
    for (int k = 1; k < N; k++) {
      const std::string &rk = rows[k];
      int gk = 0;
      for (int c = start; c <= end; c++) gk += (rk[c] == '-');
      const int diff = std::abs(gk - g0) % 3;                     // misc.c:230-244
      zw[i] |= static_cast<uint64_t>(diff) << (2 * (k - 1));      // 0 -> z=0, 1 -> z=+1, 2 -> z=-1
      uint8_t m = 0;
      if (rk[refcol[i - 2]] != '-') m |= 0x30;
      if (rk[refcol[i - 1]] != '-') m |= 0x0C;
      if (rk[refcol[i]] != '-') m |= 0x03;
      mask[static_cast<size_t>(k - 1) * (L + 1) + i] = m;
    }
  }
}

// ---------------------------------------------------------------------------------- block

// Every DP value is a float sum of table entries and gap parameters.  If each of them is 0 or
// has 2^-76 <= |v| < 2^40 it is a multiple of 2^-99, so is every partial sum, hence any non-zero
// sum has |x| >= 2^-99; with fewer than 2^20 terms |x| < 2^60.  That lies inside the range
// [2^-100, 2^100) in which the kernel's 3-instruction division by N-1 was proven equal to IEEE
// division by exhaustion (tools/verify_const_div.c).  Anything else (denormal or astronomically
// large penalties) is scored by the kernel instantiation that uses '/'.
static bool in_div_range(float v) {
  const float a = std::fabs(v);
  return a == 0.0f || (a >= 1.3234889800848443e-23f && a < 1099511627776.0f);
}

int prepare_block(const rc_block &in, const rc_params &par, const CodeTables &ct, const PairTable &pt,
                  HostBlock &hb, std::string &err) {
  hb = HostBlock();
  const int N = in.n_rows, cols = in.n_cols;
  if (N < 1 || cols < 1 || !in.rows || !in.names) { err = "empty block"; return hb.status = RC_ERR_ARG; }
  hb.N = N; hb.NK = N - 1; hb.cols = cols;
  hb.ref_start = in.ref_start; hb.ref_length = in.ref_length;
  hb.rows[0].resize(N);
  for (int r = 0; r < N; r++) {
    if (!in.rows[r] || static_cast<int>(std::strlen(in.rows[r])) != cols) { err = "Sequences are of unequal length."; return hb.status = RC_ERR_ARG; }
    hb.rows[0][r].assign(in.rows[r], cols);
    for (auto &c : hb.rows[0][r]) c = static_cast<char>(std::toupper(static_cast<unsigned char>(c)));  // RNAcode.c:121-128
  }
  int L = 0;
  for (char c : hb.rows[0][0]) L += (c != '-');                   // getSeqLength, misc.c:272-289
  hb.L = L;
  if (N <= 2 || L < 3) return hb.status = RC_ERR_SKIP;             // RNAcode.c:142-150
  if (N > kMaxRows) { err = "more than RC_MAX_ROWS rows"; return hb.status = RC_ERR_UNSUPPORTED; }
  if (cols > 65535) { err = "more than 65535 columns"; return hb.status = RC_ERR_UNSUPPORTED; }

  // reverse complement (rnaz_utils.c:316-348): T,U->A  A->T  C<->G, everything else unchanged
  hb.rows[1].resize(N);
  for (int r = 0; r < N; r++) {
    std::string &d = hb.rows[1][r];
    d.resize(cols);
    for (int j = 0; j < cols; j++) {
      char c = hb.rows[0][r][cols - 1 - j];
      switch (c) { case 'T': case 'U': c = 'A'; break; case 'A': c = 'T'; break; case 'C': c = 'G'; break; case 'G': c = 'C'; break; default: break; }
      d[j] = c;
    }
  }

  Tree t;
  if (!parse_newick(in.newick, t, err)) return hb.status = RC_ERR_ARG;
  if (static_cast<int>(t.tipname.size()) != N) { err = "tree and alignment differ in the number of sequences"; return hb.status = RC_ERR_ARG; }
  if (t.nnodes > kMaxNodes) { err = "tree too large"; return hb.status = RC_ERR_UNSUPPORTED; }
  std::vector<int> rowtip(N, -1);
  for (int r = 0; r < N; r++) {
    for (int k = 0; k < N; k++) if (t.tipname[k] == in.names[r]) { rowtip[r] = k; break; }
    if (rowtip[r] < 0) { err = std::string("row name not found in tree: ") + in.names[r]; return hb.status = RC_ERR_ARG; }
  }
  {
    std::vector<char> used(N, 0);
    for (int r = 0; r < N; r++) { if (used[rowtip[r]]) { err = "duplicate row names"; return hb.status = RC_ERR_ARG; } used[rowtip[r]] = 1; }
  }

  // getModels for both strands (RNAcode.c:164-165)
  std::vector<float> dist;
  tip_distances(t, rowtip, dist);
  for (int s = 0; s < 2; s++) {
    float freqs[4];
    mono_freqs(hb.rows[s], freqs);
    hb.models[s].resize(N);
    for (int j = 0; j < N; j++) {
      rc_model &m = hb.models[s][j];
      m.dist = dist[j];
      m.kappa = in.kappa;
      std::memcpy(m.freqs, freqs, sizeof freqs);
      expected_scores(m, ct);
    }
    gap_tables(hb.rows[s], L, hb.refcol[s], hb.zw[s], hb.mask[s]);
    hb.lut[s].assign(static_cast<size_t>(hb.NK) * kLutSize, 0.0f);
    for (int k = 1; k < N; k++) {
      float *lut = &hb.lut[s][static_cast<size_t>(k - 1) * kLutSize];
      for (int code = 0; code < kLutSize; code++) {   // (float)matrix[pepA][pepB] - scores[h], score.c:422-425
        if (!pt.is_score_code(code)) continue;
        lut[code] = static_cast<float>(pt.b_of_code(code)) - hb.models[s][k].scores[pt.h_of_code(code)];
      }
      lut[pt.code_zero] = 0.0f;
      lut[pt.code_stop0] = par.stopPenalty_0;
      lut[pt.code_stopk] = par.stopPenalty_k;
      for (int code = 0; code < kLutSize; code++) if (!in_div_range(lut[code])) hb.exact_div = 1;
    }
  }

  // simulation tables: null alignments use the forward frequencies and kappa (score.c:996-998)
  SeqGenHky hky(hb.models[0][0].freqs, hb.models[0][0].kappa);
  hb.nodes.assign(t.nnodes, NodeRec());
  for (int q = 0; q < t.nnodes; q++) {
    NodeRec &nr = hb.nodes[q];
    std::memset(&nr, 0, sizeof nr);
    nr.parent = static_cast<uint8_t>(t.parent[q]);
    nr.tiprow = -1;
    double M[4][4];
    if (q != 0) hky.cumulative(t.length[q], M);                      // evolve.c:404 (scale 1.0), :292
    for (int p = 0; p < 4; p++) {
      uint32_t base = 0;
      pack_thresholds(q == 0 ? hky.addFreq : M[p], &nr.thr[p], &base);
      nr.basepack |= base << (2 * p);
    }
  }
  if (!in_div_range(par.Delta) || !in_div_range(par.Omega) || !in_div_range(par.omega)) hb.exact_div = 1;
  if (!(par.Delta < 0.0f)) hb.exact_div = 1;   // the fast kernel drops MAX3(sum, Delta, Delta): only valid for Delta < 0
  hb.qtip.assign(N, 0);
  for (int r = 0; r < N; r++) {
    const int q = t.tipnode[rowtip[r]];
    hb.qtip[r] = static_cast<uint8_t>(q);
    hb.nodes[q].tiprow = static_cast<int8_t>(r);
  }
  return hb.status = RC_OK;
}

float pvalue_of(float score, float parMu, float parLambda) {   // RNAcode.c:182
  const float arg = (-1) * parLambda * (score - parMu);
  return f32(1 - std::exp((-1) * std::exp(f64(arg))));
}

float float_threshold_lt(double d) {
  float t = static_cast<float>(d);
  if (static_cast<double>(t) < d) t = std::nextafterf(t, INFINITY);
  return t;
}

}  // namespace rc
