// rc_eps.h -- colored alignment plots of high-scoring segments in EPS for the native driver (rc_cli.cpp --eps).
//
// colorAln / colorHSS of the reference (src/postscript.c:38-680), byte for byte: the alignment in blocks of 60 columns with names,
// coordinates, ruler and conservation bars, then the segment -- extended left and right to the next stop codon of the reference
// sequence (extendRegion, src/misc.c:555-626) -- colored codon by codon from the backtracked state path (rc_batch_backtrack,
// score.c:558-797).  Same layout code as rnacode_amd/eps.py (which tests/test_eps_cpu.py pins to EPS files the reference wrote);
// tests/test_gpu_dropin.py compares the two drivers' files byte for byte.  Plain C++ on the public C-ABI only: the genetic code and
// the BLOSUM matrix come from rc_code_tables.
#pragma once
#include <cmath>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <algorithm>
#include <functional>
#include <set>
#include <string>
#include <tuple>
#include <vector>

namespace rceps {

// setParameters(), postscript.c:19-36
constexpr int kColumnWidth = 60;
constexpr double kFontW = 6.0, kFontH = 6.5, kLine = kFontH + 2, kBlock = 3.5 * kFontH, kCons = kFontH * 0.5, kSS = 12.0, kRuler = 2.0;
constexpr double kNameStep = 3 * kFontW, kNumStep = kFontW, kMaxConsBar = 2.5 * kFontH, kStartY = 2.0, kNamesX = kFontW;

struct Tables { int32_t pep[64]; int32_t matrix[400]; };

// what rc_batch_backtrack gives for one strand and range: states / transitions [row][position], pitch = columns + 1
struct Path { std::vector<int32_t> states, transitions; int pitch = 0; bool valid = false; };
using Backtrack = std::function<Path(char strand, int b, int e)>;

struct Row { std::string name, seq; int start = 0; };
struct Hss { char strand; int frame, start, end; float pvalue; };

inline void put(std::string &out, const char *fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  const int n = std::vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  if (n < static_cast<int>(sizeof buf)) { out.append(buf, n > 0 ? n : 0); return; }
  std::vector<char> big(n + 1);
  va_start(ap, fmt);
  std::vsnprintf(big.data(), big.size(), fmt, ap);
  va_end(ap);
  out.append(big.data(), n);
}
inline std::string f1(double x) { char b[64]; std::snprintf(b, sizeof b, "%.1f", x); return b; }

inline int nt(char c) {   // ntMap: score.c:41, RNAcode.c:94-98 -- everything that is not C, G, T, U is 0
  switch (c) { case 'C': case 'c': return 1; case 'G': case 'g': return 2; case 'T': case 't': case 'U': case 'u': return 3; default: return 0; }
}
inline int seq_length(const std::string &s, size_t upto = std::string::npos) {
  int n = 0;
  for (size_t i = 0; i < s.size() && i < upto; i++) n += (s[i] != '-');
  return n;
}
// map[l] = 1-based column of the l-th residue (pos2col, misc.c:250-270); map[0] unused
inline std::vector<int> pos2col(const std::string &s) {
  std::vector<int> m{0};
  for (size_t c = 0; c < s.size(); c++) if (s[c] != '-') m.push_back(static_cast<int>(c) + 1);
  return m;
}
// the columns of the codon ending at reference position x (getBlock, misc.c:186-245)
inline void get_block(int x, const std::string &s0, const std::string &sk, const std::vector<int> &map0, std::string &b0, std::string &bk) {
  const int start = x > 3 ? map0[x - 3] + 1 : 1, end = map0[x];
  b0 = s0.substr(start - 1, end - start + 1);
  bk = sk.substr(start - 1, end - start + 1);
}
inline void codons(const std::string &b0, const std::string &bk, std::string &a, std::string &b) {
  a.clear(); b.clear();
  for (size_t i = 0; i < b0.size(); i++) if (b0[i] != '-') { a.push_back(b0[i]); b.push_back(bk[i]); }
}
inline int pep_of(const Tables &t, const std::string &codon) { return t.pep[16 * nt(codon[0]) + 4 * nt(codon[1]) + nt(codon[2])]; }
inline char translate(const std::string &codon, const Tables &t) {   // translateSeq, code.c:103-128; decodeAA, code.c:236-266
  static const char *aa = "ARNDCQEGHILKMFPSTWYV";
  for (char ch : codon) if (!std::strchr("ACGTUacgtu", ch)) return '?';
  const int p = pep_of(t, codon);
  return p == -1 ? '*' : aa[p];
}
inline std::vector<std::string> rev_rows(const std::vector<std::string> &rows) {   // revAln, rnaz_utils.c:316-348
  std::vector<std::string> out;
  for (const std::string &r : rows) {
    std::string s(r.rbegin(), r.rend());
    for (char &c : s) { switch (c) { case 'T': case 'U': c = 'A'; break; case 'A': c = 'T'; break; case 'C': c = 'G'; break; case 'G': c = 'C'; break; default: break; } }
    out.push_back(s);
  }
  return out;
}
// rnaz_utils.c:249-264 with encode_char of librna/pair_mat.h:26-38 ("_ACGUTXKI")
inline std::string consensus(const std::vector<std::string> &rows) {
  static const std::string order = "_ACGUTXKI";
  std::string out;
  for (size_t c = 0; c < rows[0].size(); c++) {
    int freq[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (const std::string &r : rows) {
      const size_t at = order.find(r[c]);
      int code = at == std::string::npos ? 0 : static_cast<int>(at);
      if (code > 4) code -= 1;
      freq[code]++;
    }
    int best = 0, fm = 0;
    for (int s = 0; s < 8; s++) if (freq[s] > fm) { best = s; fm = freq[s]; }
    out.push_back(order[best]);
  }
  return out;
}
// walk codon by codon from the segment to the next stop codon of the reference row (extendRegion, misc.c:555-626)
inline int extend_region(const std::vector<std::string> &rows, int pos, int direction, const Tables &t) {
  const std::string &seq = rows[0];
  const int L = seq_length(seq);
  const std::vector<int> map0 = pos2col(seq);
  int x = direction == 0 ? pos + 2 : pos;
  std::string b0, bk, codon;
  for (;;) {
    get_block(x, seq, rows[1], map0, b0, bk);
    codon.clear();
    for (char ch : b0) if (ch != '-') codon.push_back(ch);
    if (pep_of(t, codon) == -1) break;
    if (direction == 0) { if (x - 3 < 3) break; x -= 3; }
    else { if (x + 3 > L) break; x += 3; }
  }
  return direction == 0 ? x - 2 : x;
}

// colorHSS, postscript.c:334-680
inline void color_hss(std::string &out, const std::vector<std::string> &rows, const Path &bt, const std::string &label, int b, int i,
                      int column_width, double seqs_x, const Tables &t) {
  static const char *colors[2][6] = {{"0.0 0.0", "0.0 0.2", "0.0 0.4", "0.0 0.6", "0.0 0.8", "0.0 1"},          // colorMatrix, postscript.c:355-359: red (radical)
                                     {"0.32 0.1", "0.32 0.2", "0.32 0.4", "0.32 0.6", "0.32 0.8", "0.32 1"}};   // green (synonymous)
  const int N = static_cast<int>(rows.size());
  const std::string &seq0 = rows[0];
  const std::vector<int> map0 = pos2col(seq0);
  std::vector<int> syn, nonsyn;
  std::string b0, bk, ca, cb;
  for (int x = b + 2; x < i + 3; x += 3) {
    int s = 0, ns = 0;
    std::set<std::tuple<int, int, int>> seen;
    for (int k = 1; k < N; k++) {
      get_block(x, seq0, rows[k], map0, b0, bk);
      codons(b0, bk, ca, cb);
      const int pa = pep_of(t, ca);
      if (cb.find('-') != std::string::npos) continue;
      const auto key = std::make_tuple(nt(cb[0]), nt(cb[1]), nt(cb[2]));
      if (!seen.insert(key).second) continue;
      const int pb = pep_of(t, cb);
      if (pa != -1 && pb != -1) {
        const int score = t.matrix[20 * pa + pb];
        if (ca != cb) { if (pa == pb) s++; else if (score < 0) ns++; }
      } else ns++;
    }
    syn.push_back(s); nonsyn.push_back(ns);
  }
  const double row_pitch = kLine * (N + 2) + kBlock + kCons + kRuler;
  int at = 0;
  for (int x = b + 2; x < i + 3; x += 3, at++) {
    const char *syn_color = colors[1][std::min(syn[at], 5)], *nonsyn_color = colors[0][std::min(nonsyn[at], 5)];
    for (int k = 0; k < N; k++) {
      get_block(x, seq0, rows[k], map0, b0, bk);
      codons(b0, bk, ca, cb);
      const int pa = pep_of(t, ca);
      int score, pb;
      if (cb.find('-') != std::string::npos) { score = -1; pb = -99; }   // contains gap
      else {
        pb = pep_of(t, cb);
        if (pa != -1 && pb != -1) score = t.matrix[20 * pa + pb];
        else { score = -1; pb = 99; }                                      // stop
      }
      const int blen = static_cast<int>(b0.size());
      for (int ii = 0; ii < blen; ii++) {
        const int curr_col = map0[x] - blen + ii;
        const int block = static_cast<int>(std::ceil(static_cast<float>(curr_col + 1) / static_cast<float>(column_width)));
        const double xx = seqs_x + (curr_col - (block - 1) * column_width) * kFontW;
        const double yy = kStartY + (block - 1) * row_pitch + kSS * block + (k + 1) * kLine;
        if (k == 0 && !label.empty() && x == b + 2 && ii == 0) {
          out += "0.15 0.5 0.6 sethsbcolor\n/Helvetica findfont\n[8 0 0 -8 0 0] makefont setfont\n";
          out += "(" + label + ") " + f1(xx) + " " + f1(yy - 2 * kLine) + " string\n";
          out += "0.0 setgray\n";
        }
        if (k == 0 && !label.empty()) {   // translation line
          const double off_l = ii == 0 ? 0.5 : 0.0, off_r = ii == blen - 1 ? 0.5 : 0.0;
          out += f1(xx + off_l) + " " + f1(yy - 1) + " " + f1(xx + kFontW - off_r) + " " + f1(yy - kLine - 1) + " 0.15 0.5 box\n";
        }
        if (k == 0 && ii == static_cast<int>(std::ceil(blen / 2.0)) - 1) {
          out += "/Courier findfont\n[10 0 0 -10 0 0] makefont setfont\n";
          out += std::string("(") + translate(ca, t) + ") " + f1(xx) + " " + f1(yy - kLine) + " string\n";
        }
        const std::string box = f1(xx) + " " + f1(yy - 1) + " " + f1(xx + kFontW) + " " + f1(yy + kFontH + 1);
        if (k == 0) {
          out += box + " 0.0 0.0 box\n/Courier-Bold findfont\n[10 0 0 -10 0 0] makefont setfont\n";
          out += std::string("(") + bk[ii] + ") " + f1(xx) + " " + f1(yy) + " string\n";
        }
        if (k > 0) {
          const int st = bt.states[static_cast<size_t>(k) * bt.pitch + x], tr = bt.transitions[static_cast<size_t>(k) * bt.pitch + x];
          if (st == 0 && tr == 0) {   // in frame
            if (score >= 0) out += box + " " + (ca != cb ? syn_color : "0.0 0.0") + " box\n";
            else {
              if (pb == 99) out += box + " 0.6 1.0 box\n";
              if (pb == -99) out += box + " 0.0 0.0 box\n";
              if (pb != 99 && pb != -99) out += box + " " + nonsyn_color + " box\n";
            }
            out += (pa == pb && ca != cb) ? "/Courier-Bold findfont\n" : "/Courier findfont\n";
            out += "[10 0 0 -10 0 0] makefont setfont\n";
          }
          if (tr == 2) out += "/Courier-Bold findfont\n[10 0 0 -10 0 0] makefont setfont\n0.2 setgray\n" + box + " box2\n0.8 setgray\n";
          if (tr == 1 || (tr == 0 && st != 0)) out += "/Courier findfont\n[10 0 0 -10 0 0] makefont setfont\n0.8 setgray\n" + box + " box2\n0 setgray\n";
        }
        out += std::string("(") + bk[ii] + ") " + f1(xx) + " " + f1(yy) + " string\n";
        out += "0 setgray\n";
      }
    }
  }
}

// colorAln, postscript.c:38-332: the EPS text for one high-scoring segment of a block (rows upper-cased, as RNAcode.c:121-128 leaves them)
inline std::string color_aln(const std::vector<Row> &block, const Hss &hss, const Backtrack &backtrack, const Tables &t) {
  std::vector<std::string> rows;
  size_t max_name = 0;
  for (const Row &r : block) { rows.push_back(r.seq); max_name = std::max(max_name, r.name.size()); }
  const int N = static_cast<int>(rows.size()), length = static_cast<int>(rows[0].size());
  const double seqs_x = kNamesX + max_name * kFontW + kNameStep;
  const int max_num = 10;
  int column_width = kColumnWidth, tmp_columns = kColumnWidth;
  if (length < column_width) { column_width = length; tmp_columns = length; }
  const double image_w = std::ceil(kNamesX + (max_name + tmp_columns + max_num) * kFontW + 2 * kNameStep + kFontW + kNumStep);
  const double image_h = kStartY + std::ceil(static_cast<float>(length) / static_cast<float>(column_width)) * ((N + 2) * kLine + kBlock + kCons + kSS + kRuler);
  std::string out;
  put(out, "%%!PS-Adobe-3.0 EPSF-3.0\n%%%%BoundingBox: %d %d %d %d\n%%%%EndComments\n", 0, 0, static_cast<int>(image_w), static_cast<int>(image_h));
  out += "%Created by RNAcode; visit wash.github.com/rnacode\n"
         "% draws box in color given by hue and saturation\n"
         "/box { % x1 y1 x2 y2 hue saturation\n"
         "  gsave\n"
         "  dup 0.3 mul 1 exch sub sethsbcolor\n"
         "  exch 3 index sub exch 2 index sub rectfill\n"
         "  grestore\n"
         "} def\n"
         "% draws a box in current color\n"
         "/box2 { % x1 y1 x2 y2\n"
         "  exch 3 index sub exch 2 index sub rectfill\n"
         "} def\n"
         "/string { % (Text) x y\n"
         " 6 add\n"
         " moveto\n"
         "  show\n"
         "} def\n";
  put(out, "0 %d translate\n", static_cast<int>(image_h));
  out += "1 -1 scale\n/Courier findfont\n[10 0 0 -10 0 0] makefont setfont\n";

  std::string ruler(static_cast<size_t>(2 * length + 16), '.');
  for (int i = 0; i < length; i++)
    if ((i + 1) % 10 == 0 && (i + 1) % column_width != 0) {
      std::string digits = std::to_string(i + 1).substr(0, static_cast<size_t>(std::max(length - 1, 0)));   // snprintf(tmpBuffer, length, ...)
      ruler.replace(static_cast<size_t>(i), digits.size(), digits);
    }
  ruler.resize(static_cast<size_t>(length));

  const std::string cons = consensus(rows);
  double curr_y = kStartY;
  int curr_pos = 0;
  while (curr_pos < length) {
    out += "0 setgray\n";
    curr_y += kSS + kLine;
    for (const Row &r : block) {
      const std::string chunk = r.seq.substr(static_cast<size_t>(curr_pos), static_cast<size_t>(column_width));
      int match = seq_length(r.seq, static_cast<size_t>(curr_pos) + chunk.size());
      if (hss.strand == '+') match += r.start; else match = seq_length(r.seq) - match + 1;
      out += "(" + r.name + ") " + f1(kNamesX) + " " + f1(curr_y) + " string\n";
      put(out, "(%i) %s %s string\n", match, f1(seqs_x + kFontW * chunk.size() + kNumStep).c_str(), f1(curr_y).c_str());
      curr_y += kLine;
    }
    curr_y += kRuler;
    const std::string chunk = ruler.substr(static_cast<size_t>(curr_pos), static_cast<size_t>(column_width));
    for (int twice = 0; twice < 2; twice++) out += "(" + chunk + ") " + f1(seqs_x) + " " + f1(curr_y) + " string\n";
    curr_y += kLine;
    curr_y += kCons;
    out += "0.6 setgray\n";
    for (int i = curr_pos; i < std::min(curr_pos + column_width, length); i++) {
      int match = 0;
      for (const std::string &s : rows) {
        if (cons[i] == s[i]) match++;
        if (cons[i] == 'U' && s[i] == 'T') match++;
        if (cons[i] == 'T' && s[i] == 'U') match++;
      }
      float score = static_cast<float>(match - 1) / static_cast<float>(N - 1);
      if (cons[i] == '-' || cons[i] == '_' || cons[i] == '.') score = 0.0f;
      float bar = static_cast<float>(kMaxConsBar) * score;
      if (bar == 0.0f) bar = 1.0f;
      const double xx = seqs_x + (i - curr_pos) * kFontW;
      const float top = (static_cast<float>(curr_y) + static_cast<float>(kMaxConsBar)) - bar;
      out += f1(xx) + " " + f1(static_cast<double>(top)) + " " + f1(xx + kFontW) + " " + f1(curr_y + kMaxConsBar) + " box2\n";
    }
    curr_y += kBlock;
    curr_pos += column_width;
  }

  out += "0.0 setgray\n";
  const std::vector<std::string> curr = hss.strand == '+' ? rows : rev_rows(rows);
  for (int part = 0; part < 3; part++) {
    int b, e;
    std::string label;
    if (part == 0) { b = extend_region(curr, hss.start, 0, t); e = hss.start - 1; }
    else if (part == 1) {
      b = hss.start; e = hss.end;
      const double p = static_cast<double>(hss.pvalue);
      char ps[64];
      if (p < 0.001) { if (p < 10e-16) std::snprintf(ps, sizeof ps, "<1e-16\n"); else std::snprintf(ps, sizeof ps, "%9.1e\n", p); }
      else std::snprintf(ps, sizeof ps, "%9.3f\n", p);
      char lb[128];
      std::snprintf(lb, sizeof lb, "Frame %c%i p =%s", hss.strand, hss.frame + 1, ps);
      label = lb;
    } else {
      b = hss.end + 1; e = extend_region(curr, hss.end, 1, t);
      if (b >= e) break;
    }
    Path bt;
    if (b + 2 < e + 3) bt = backtrack(hss.strand, b, e);
    color_hss(out, curr, bt, label, b, e, column_width, seqs_x, t);
  }
  out += "showpage\n";
  return out;
}

}  // namespace rceps
