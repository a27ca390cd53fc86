// rc_cli.cpp -- rnacode_hip: a native stand-alone driver around the scoring path.
//
// The reference's driver (src/RNAcode.c:52-231) reads alignment blocks, gets a tree and kappa per block from PhyML, scores
// the block and prints its high-scoring segments.  This one does the same on top of the public C-ABI only
// (include/rnacode_hip.h): MAF / ClustalW readers (src/rnaz_utils.c:44-234), trees from a sidecar or fitted on the GPU
// (rc_fit_trees_device), scoring as a stream of sub-batches (rc_stream_*), listings in the reference's three formats
// (src/misc.c:392-552 printResults, footer RNAcode.c:223-228) byte for byte.  Same options as `python -m rnacode_amd.cli`
// (which additionally draws the EPS plots):
//
//   rnacode_hip [-n N] [-p CUTOFF] [-g | -t] [-b] [-r] [-s] [-m 62|90] [-c D,O,o,S] [-o OUT] [-l SPECIES,...] [--trees SIDECAR]
//               [-e [-i EPS_CUTOFF] [-d EPS_DIR]] [--seed-base S] [--device D | --gpus N [--devices D0,D1,...]] [--sub-blocks B] [FILE]
//
// --gpus N (SURVEY.md section 8e inside ONE process, no collective library needed): N contexts on N host threads, each with
// usable CPUs / N block-preparation threads.  Many blocks: the sub-batches of the stream are dealt to the GPUs in turn, every GPU
// fits the trees of and scores its own sub-batches, and one writer emits the listings in input order (the hit counter runs across
// blocks, misc.c:392-552).  Few blocks (fewer than two per GPU): every GPU simulates a slice of every block's SAMPLE range
// (rc_params.seed_base + first sample, so the union is the single-GPU sample set bit for bit), the slices meet on the host and
// are fitted there (rc_evd_fit), as getExtremeValuePars does after its loop (score.c:1004-1052).
//
// Quirk kept from the reference: the 4th value of --pars goes to stopPenalty_0 (RNAcode.c:318).
#include <algorithm>
#include <atomic>
#include <cctype>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <sstream>
#include <condition_variable>
#include <deque>
#include <memory>
#include <mutex>
#include <string>
#include <string_view>
#include <thread>
#include <vector>
#include <map>

#include <sys/stat.h>
#include <unistd.h>

#include "../../include/rnacode_hip.h"
#include "rc_eps.h"

namespace {

// A row's name and residues are views into the input's bytes (the reader ends each with a NUL in place: no copy of a 94 MB file, no
// 1.2 million small allocations); a ClustalW file's rows, concatenated from its pieces, live in g_owned.
struct Row {
  std::string_view name, seq;
  int start = 0, length = 0;
  char strand = '?';
};
std::deque<std::string> g_owned;
struct Block {
  std::vector<Row> rows;
  std::string tree;     // empty: none
  float kappa = 0.0f;
  int index = 0;        // position in the input file (the --trees sidecar has one entry per block READ, before --limit drops any)
};

double now();
double g_t0 = 0.0;   // start of main
std::thread *g_ctxThread = nullptr;   // the thread that brings up the HIP context: an exit waits for it

// Leaves at once, like the success path: other threads (tree fits with kernels in flight, the printer) may still be running, and
// std::exit would run the HIP runtime's teardown and the static destructors under them.
[[noreturn]] void die(const std::string &msg) {
  if (g_ctxThread && g_ctxThread->joinable() && g_ctxThread->get_id() != std::this_thread::get_id()) g_ctxThread->join();
  std::fprintf(stderr, "ERROR: %s\n", msg.c_str());
  std::fflush(stdout);
  std::fflush(stderr);
  _exit(1);
}

std::vector<std::string> fields(const std::string &line) {
  std::vector<std::string> f;
  std::istringstream is(line);
  std::string w;
  while (is >> w) f.push_back(w);
  return f;
}

void check_equal(const Block &b) {
  for (const Row &r : b.rows)
    if (r.seq.size() != b.rows[0].seq.size()) die("Sequences are of unequal length.");
}

// Blocks of a MAF stream: 's' lines need exactly 7 fields; 'i', 'e', 'q' and comment lines are skipped; a block ends at the
// next 'a' line (rnaz_utils.c:132-234).  One pass over the file's bytes, fields cut at white space in place (a 94 MB file of
// 100 000 blocks: 0.34 s with a string per line and a stream per line's fields, the largest serial stage of the run).
void read_maf_range(char *p, char *end, std::vector<Block> &out) {
  Block cur;
  auto space = [](char c) { return c == ' ' || c == '\t' || c == '\r' || c == '\v' || c == '\f'; };
  while (p < end) {
    char *eol = static_cast<char *>(std::memchr(p, '\n', static_cast<size_t>(end - p)));
    if (!eol) eol = end;
    char *f[8][2];   // up to eight fields: begin, end
    int nf = 0;
    for (char *q = p; q < eol;) {
      while (q < eol && space(*q)) q++;
      if (q >= eol) break;
      char *b0 = q;
      if (nf == 6) {   // the seventh field of an 's' line is the sequence, nearly always the rest of the line: look for white space in bulk
        char *e = eol;
        while (e > b0 && space(e[-1])) e--;
        const size_t len = static_cast<size_t>(e - b0);
        if (!std::memchr(b0, ' ', len) && !std::memchr(b0, '\t', len) && !std::memchr(b0, '\r', len) && !std::memchr(b0, '\v', len) && !std::memchr(b0, '\f', len)) {
          f[6][0] = b0; f[6][1] = e; nf = 7;
          break;
        }
      }
      while (q < eol && !space(*q)) q++;
      if (nf < 8) { f[nf][0] = b0; f[nf][1] = q; }
      nf++;
    }
    p = eol < end ? eol + 1 : end;
    if (nf == 0 || f[0][0][0] == '#') continue;
    const bool one = f[0][1] - f[0][0] == 1;
    const char tag = one ? f[0][0][0] : 0;
    if (tag == 'i' || tag == 'e' || tag == 'q') continue;
    if (tag == 's') {
      if (nf != 7) die("Invalid MAF format (number of fields in 's' line not correct)");
      if (f[4][0][0] != '+' && f[4][0][0] != '-') die("Invalid MAF format (strand)");
      cur.rows.emplace_back();
      Row &r = cur.rows.back();
      r.start = std::atoi(f[2][0]); r.length = std::atoi(f[3][0]);   // (atoi stops at the white space behind the field)
      r.strand = f[4][0][0];
      // name and residues stay where they are, each ended by a NUL in place of the white space (or line end) behind it -- the buffer
      // has one spare byte behind its last line
      r.name = std::string_view(f[1][0], static_cast<size_t>(f[1][1] - f[1][0])); *f[1][1] = 0;
      r.seq = std::string_view(f[6][0], static_cast<size_t>(f[6][1] - f[6][0])); *f[6][1] = 0;
      continue;
    }
    if (tag == 'a') {
      if (!cur.rows.empty()) { check_equal(cur); const size_t k = cur.rows.size(); out.push_back(std::move(cur)); cur = Block(); cur.rows.reserve(k); }
      else cur = Block();
    }
  }
  if (!cur.rows.empty()) { check_equal(cur); out.push_back(std::move(cur)); }
}

// A large file is cut at lines that start with "a" + white space -- where a block ends for the sequential reader too -- and the
// pieces are parsed side by side.
std::vector<Block> read_maf(char *p, size_t size) {
  char *end = p + size;
  unsigned nt = std::min<unsigned>(static_cast<unsigned>(std::max(1, rc_host_cpus())), 16u);   // (the CPUs this process may use, not the machine's)
  if (size < (4u << 20)) nt = 1;
  std::vector<char *> cut{p};
  for (unsigned t = 1; t < nt; t++) {
    char *q = p + size / nt * t;
    while (q + 2 < end) {
      q = static_cast<char *>(std::memchr(q, '\n', static_cast<size_t>(end - q)));
      if (!q || q + 2 >= end) { q = end; break; }
      if (q[1] == 'a' && (q[2] == ' ' || q[2] == '\t' || q[2] == '\n' || q[2] == '\r')) { q++; break; }
      q++;
    }
    if (q + 2 >= end) break;
    if (q > cut.back()) cut.push_back(q);
  }
  cut.push_back(end);
  std::vector<std::vector<Block>> part(cut.size() - 1);
  std::vector<std::thread> th;
  for (size_t t = 1; t + 1 < cut.size(); t++) th.emplace_back([&, t] { read_maf_range(cut[t], cut[t + 1], part[t]); });
  read_maf_range(cut[0], cut[1], part[0]);
  for (auto &x : th) x.join();
  std::vector<Block> out = std::move(part[0]);
  size_t total = out.size();
  for (size_t t = 1; t < part.size(); t++) total += part[t].size();
  out.reserve(total);
  for (size_t t = 1; t < part.size(); t++) for (Block &b : part[t]) out.push_back(std::move(b));
  return out;
}

// A ClustalW file holds one block; rows get start = length = 0 (rnaz_utils.c:44-117)
std::vector<Block> read_clustal(const std::vector<std::string> &lines) {
  std::vector<std::string> names, seqs;
  size_t nn = 0;
  for (const std::string &raw : lines) {
    std::string line = raw;
    while (!line.empty() && (line.back() == '\n' || line.back() == '\r')) line.pop_back();
    if (line.compare(0, 7, "CLUSTAL") == 0) {
      if (!names.empty()) break;
      continue;
    }
    if (line.size() < 4 || std::isspace(static_cast<unsigned char>(line[0]))) { nn = 0; continue; }
    const std::vector<std::string> f = fields(line);
    if (f.size() < 2) continue;
    const std::string name = f[0].substr(0, 99);
    if (nn == names.size()) { names.push_back(name); seqs.push_back(f[1]); }
    else {
      if (names[nn] != name) die("Inconsistent sequence names in CLUSTAL file");
      seqs[nn] += f[1];
    }
    nn++;
  }
  std::vector<Block> out;
  if (!names.empty()) {
    Block b;
    for (size_t i = 0; i < names.size(); i++) {
      Row r;
      g_owned.push_back(names[i]); r.name = g_owned.back();
      g_owned.push_back(seqs[i]); r.seq = g_owned.back();
      b.rows.push_back(r);
    }
    check_equal(b);
    out.push_back(b);
  }
  return out;
}

// The whole input in one buffer that lives as long as the process (the rows are views into it): a regular file is read by several
// threads side by side (pread), anything else as it comes.
std::vector<Block> read_alignment(FILE *in) {
  char *text = nullptr;
  size_t size = 0;
  struct stat sp;
  const int fd = fileno(in);
  // (a regular file is read from where its descriptor stands -- a stdin redirected from a file that something has read the head of --, not from offset 0)
  const off_t from = lseek(fd, 0, SEEK_CUR);
  if (fstat(fd, &sp) == 0 && S_ISREG(sp.st_mode) && from >= 0 && sp.st_size > from) {
    size = static_cast<size_t>(sp.st_size - from);
    text = static_cast<char *>(std::malloc(size + 1));
    if (!text) die("out of memory");
    // threads: the CPUs this process may use (affinity mask and cgroup quota: rc_host_cpus), not the machine's
    const unsigned nt = size < (8u << 20) ? 1u : std::min<unsigned>(static_cast<unsigned>(std::max(1, rc_host_cpus())), 8u);
    std::vector<size_t> got(nt, 0);
    auto piece = [&](unsigned t) {
      size_t lo = size / nt * t, hi = t + 1 == nt ? size : size / nt * (t + 1), at = lo;
      while (at < hi) {
        const ssize_t k = pread(fd, text + at, hi - at, from + static_cast<off_t>(at));
        if (k <= 0) break;
        at += static_cast<size_t>(k);
      }
      got[t] = at - lo;
    };
    std::vector<std::thread> th;
    for (unsigned t = 1; t < nt; t++) th.emplace_back(piece, t);
    piece(0);
    for (auto &x : th) x.join();
    size_t have = 0;
    for (unsigned t = 0; t < nt; t++) { const size_t want = (t + 1 == nt ? size : size / nt * (t + 1)) - size / nt * t; have += got[t]; if (got[t] != want) break; }
    size = have;   // (a file that shrank while it was read: what came in order)
  } else {
    size_t cap = 1 << 20;
    text = static_cast<char *>(std::malloc(cap + 1));
    if (!text) die("out of memory");
    for (size_t k; (k = std::fread(text + size, 1, cap - size, in)) > 0;) {
      size += k;
      if (size == cap) { cap *= 2; text = static_cast<char *>(std::realloc(text, cap + 1)); if (!text) die("out of memory"); }
    }
  }
  text[size] = 0;
  if (std::getenv("RC_CLI_TIMES")) std::fprintf(stderr, "[rnacode_hip] %.1f MB in memory at %.3f s\n", size / 1e6, now() - g_t0);
  const std::string_view all(text, size);
  size_t at = all.find_first_not_of(" \t\r\n");
  if (at != std::string_view::npos) at = all.rfind('\n', at) == std::string_view::npos ? 0 : all.rfind('\n', at) + 1;   // start of the first non-blank line
  if (at != std::string_view::npos && all.compare(at, 7, "CLUSTAL") == 0) {
    std::vector<std::string> lines;
    std::istringstream is(std::string(text, size));
    std::string line;
    while (std::getline(is, line)) lines.push_back(line);
    return read_clustal(lines);
  }
  return read_maf(text, size);
}

// printResults (misc.c:392-552); the HSS counter runs across blocks and is not advanced after a --best-only break
struct Listing {
  FILE *out = stdout;
  int fmt = 0;          // 0 default table, 1 GTF, 2 tabular
  float cutoff = 1.0f;
  bool bestOnly = false, bestRegion = false;
  bool eps = false;
  float epsCutoff = 0.05f;
  std::string epsDir = "eps";
  int hitCounter = 0;

  // The HSS of a block in the order printResults walks them, the hidden ones marked (misc.c:400-433).  `res` is filtered and, with
  // --best-region, re-ordered in place; `order` indexes it by descending score.
  void arrange(std::vector<rc_hss> &res, std::vector<char> &hide, std::vector<size_t> &order) const {
    res.erase(std::remove_if(res.begin(), res.end(), [](const rc_hss &h) { return !(h.score > 0.0f); }), res.end());
    hide.assign(res.size(), 0);
    if (bestRegion) {   // misc.c:408-433: sort by start codon, hide the weaker of two overlapping HSS
      std::stable_sort(res.begin(), res.end(), [](const rc_hss &a, const rc_hss &b) { return a.startSite < b.startSite; });
      size_t curr = 0;
      for (size_t nxt = 1; nxt < res.size(); nxt++) {
        if (!(res[curr].endSite <= res[nxt].startSite)) {
          if (res[curr].score > res[nxt].score) hide[nxt] = 1;
          else { hide[curr] = 1; curr = nxt; }
        } else curr = nxt;
      }
    }
    order.resize(res.size());
    for (size_t i = 0; i < order.size(); i++) order[i] = i;
    std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) { return res[a].score > res[b].score; });
  }
  // positions in `order` of the HSS that get a line (the loop of misc.c:444-547 without its output)
  std::vector<size_t> listed(const std::vector<rc_hss> &res, const std::vector<char> &hide, const std::vector<size_t> &order) const {
    std::vector<size_t> out;
    if (res.empty() || res[order[0]].pvalue > cutoff) return out;
    for (size_t idx : order) {
      if (!(res[idx].pvalue < cutoff)) break;
      if (hide[idx]) continue;
      out.push_back(idx);
      if (bestOnly) break;
    }
    return out;
  }

  // epsText: for the k-th listed HSS the EPS file's text, or empty (drawn while the block's batch was alive; misc.c:461-474 writes
  // hss-<counter>.eps in front of the line)
  void block(std::vector<rc_hss> res, const std::string &refName, const std::vector<std::string> *epsText = nullptr) {
    std::vector<char> hide;
    std::vector<size_t> order;
    arrange(res, hide, order);
    if (res.empty() || res[order[0]].pvalue > cutoff) {
      if (fmt == 0) std::fprintf(out, "\nNo significant coding regions found.\n");
      return;
    }
    if (fmt == 0) {
      std::fprintf(out, "\n%6s%5s%7s%6s%6s%12s%12s%12s%9s%9s\n", " HSS # ", "Frame", "Length", "From", "To", "Name", "Start", "End", "Score", "P");
      std::fprintf(out, "======================================================================================\n");
    }
    size_t k = 0;
    for (size_t idx : listed(res, hide, order)) {
      const rc_hss &h = res[idx];
      const double p = static_cast<double>(h.pvalue);
      if (epsText && k < epsText->size() && !(*epsText)[k].empty()) {
        struct stat sp;
        if (stat(epsDir.c_str(), &sp) != 0 && mkdir(epsDir.c_str(), S_IRWXU | S_IROTH | S_IRGRP) != 0)
          std::fprintf(stderr, "WARNING: Could not create directory: %s", epsDir.c_str());
        const std::string fn = epsDir + "/hss-" + std::to_string(hitCounter) + ".eps";
        if (FILE *f = std::fopen(fn.c_str(), "w")) { std::fwrite((*epsText)[k].data(), 1, (*epsText)[k].size(), f); std::fclose(f); }
        else std::fprintf(stderr, "ERROR: Can't open output file %s\n", fn.c_str());
      }
      k++;
      const int length = h.endSite - h.startSite + 1;
      const char strand[2] = {static_cast<char>(h.strand), 0};
      if (fmt == 0) {
        std::fprintf(out, "%6i %4s%i%7i%6i%6i%12s%12i%12i%9.2f", hitCounter, strand, h.frame + 1, length, h.startSite + 1, h.endSite + 1,
                     refName.c_str(), h.startGenomic, h.endGenomic, static_cast<double>(h.score));
        if (p < 0.001) {
          if (p < 10e-16) std::fprintf(out, "   <1e-16\n"); else std::fprintf(out, "% 9.1e\n", p);
        } else std::fprintf(out, "% 9.3f\n", p);
      } else if (fmt == 1) {
        const size_t dot = refName.find('.');
        const std::string name = dot == std::string::npos ? refName : refName.substr(dot + 1);
        std::fprintf(out, "%s\t%s\t%s\t%i\t%i\t%.2f|%.2e\t%s\t%s\t%s%i%s\n", name.c_str(), "RNAcode", "CDS", h.startGenomic + 1, h.endGenomic + 1,
                     static_cast<double>(h.score), p, strand, ".", "gene_id \"Gene", hitCounter, "\"; transcript_id \"transcript 0\";");
      } else {
        std::fprintf(out, "%i\t%s\t%i\t%i\t%i\t%i\t%s\t%i\t%i\t%7.3f\t", hitCounter, strand, h.frame + 1, length, h.startSite + 1, h.endSite + 1,
                     refName.c_str(), h.startGenomic, h.endGenomic, static_cast<double>(h.score));
        if (p < 0.001) std::fprintf(out, "% 9.3e\n", p); else std::fprintf(out, "% 9.3f\n", p);
      }
      if (bestOnly) break;
      hitCounter++;
    }
  }
};

void usage() {
  std::fprintf(stderr, "usage: rnacode_hip [-n N] [-p CUTOFF] [-g | -t] [-b] [-r] [-s] [-m 62|90] [-c D,O,o,S] [-o OUT] [-l SPECIES,...] [--trees SIDECAR]\n"
                       "                   [-e [-i EPS_CUTOFF] [-d EPS_DIR]] [--seed-base S] [--device D | --gpus N [--devices D0,D1,...]] [--sub-blocks B]\n"
                       "                   [--dump-blocks] [FILE]\n");
}

double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// results of one sub-batch on their way to the writer
struct Job {
  int seq = 0;                       // position of the sub-batch in the input: the writer takes them in this order
  std::vector<int> blockIdx;         // input block of each of its blocks
  std::vector<rc_hss> hss;
  std::vector<int64_t> offs;
  std::vector<int> status;
  std::vector<std::string> why;
  std::vector<std::vector<std::string>> eps;   // per block: the EPS texts of its listed HSS (--eps)
};

// everything the threads share
struct Run {
  rc_params par;
  Listing list;
  std::vector<Block> blocks;
  std::vector<rc_block> rb;          // rc_block views of `blocks` (newick / kappa filled in when the trees are there)
  int cap = 64;                      // room for a Newick text
  bool haveSidecar = false;
  rceps::Tables tables;
  // the writer: one thread, jobs in input order
  std::mutex jm;
  std::condition_variable jcv;
  std::map<int, std::unique_ptr<Job>> jobs;
  int nextSeq = 0, totalSeq = -1;    // totalSeq: number of jobs to expect, known when the plan is
  double tList = 0;
  // failure of any worker: the first message wins, everybody stops
  std::mutex em;
  std::string error;
  std::atomic<bool> failed{false};
  void fail(const std::string &msg) {
    { std::lock_guard<std::mutex> lk(em); if (error.empty()) error = msg; }
    failed.store(true);
    jcv.notify_all();
  }
};

void writer_thread(Run &R) {
  for (;;) {
    std::unique_ptr<Job> j;
    {
      std::unique_lock<std::mutex> lk(R.jm);
      R.jcv.wait(lk, [&] { return R.failed.load() || R.jobs.count(R.nextSeq) || (R.totalSeq >= 0 && R.nextSeq >= R.totalSeq); });
      if (R.failed.load()) return;
      if (!R.jobs.count(R.nextSeq)) return;
      j = std::move(R.jobs[R.nextSeq]);
      R.jobs.erase(R.nextSeq);
      R.nextSeq++;
    }
    const double t = now();
    for (size_t i = 0; i < j->blockIdx.size(); i++) {
      const int st = j->status[i];
      const Block &blk = R.blocks[j->blockIdx[i]];
      if (st == RC_ERR_SKIP) {   // RNAcode.c:142-150
        std::fprintf(stderr, "Skipping alignment. %s\n", blk.rows.size() <= 2 ? "There must be at least three sequences in the alignment." : "Too short.");
        continue;
      }
      if (st != RC_OK) {         // RNAcode.c:153-156
        std::fprintf(stderr, "Skipping alignment. Failed to build ML tree. (%s)\n", j->why[i].empty() ? "not scored" : j->why[i].c_str());
        continue;
      }
      R.list.block(std::vector<rc_hss>(j->hss.begin() + j->offs[i], j->hss.begin() + j->offs[i + 1]), std::string(blk.rows[0].name),
                   j->eps.empty() ? nullptr : &j->eps[i]);
    }
    R.tList += now() - t;
  }
}

void post(Run &R, std::unique_ptr<Job> j) {
  { std::lock_guard<std::mutex> lk(R.jm); const int s = j->seq; R.jobs[s] = std::move(j); }
  R.jcv.notify_all();
}

// --eps: the EPS texts of a block's listed HSS, drawn while its batch is alive (the backtracked paths come from the device)
std::vector<std::string> draw_block(const Run &R, rc_batch *b, int i, const Block &blk, const std::vector<rc_hss> &hss, std::string &err) {
  std::vector<rc_hss> res = hss;
  std::vector<char> hide;
  std::vector<size_t> order;
  R.list.arrange(res, hide, order);
  std::vector<std::string> texts;
  std::vector<rceps::Row> rows;
  for (const Row &r : blk.rows) rows.push_back(rceps::Row{std::string(r.name), std::string(r.seq), r.start});
  const int cols = static_cast<int>(blk.rows[0].seq.size()), N = static_cast<int>(blk.rows.size());
  for (size_t idx : R.list.listed(res, hide, order)) {
    const rc_hss &h = res[idx];
    if (!(h.pvalue < R.list.epsCutoff)) { texts.emplace_back(); continue; }   // misc.c:462
    auto bt = [&](char strand, int lo, int hi) {
      rceps::Path p;
      p.pitch = cols + 1;
      p.states.assign(static_cast<size_t>(N) * p.pitch, 0);
      p.transitions.assign(static_cast<size_t>(N) * p.pitch, 0);
      std::vector<int32_t> z(static_cast<size_t>(N) * p.pitch, 0);
      if (rc_batch_backtrack(b, i, strand == '+' ? 0 : 1, lo, hi, p.states.data(), z.data(), p.transitions.data()) != RC_OK) err = rc_last_error();
      else p.valid = true;
      return p;
    };
    texts.push_back(rceps::color_aln(rows, rceps::Hss{static_cast<char>(h.strand), h.frame, h.start, h.end, h.pvalue}, bt, R.tables));
  }
  return texts;
}

// what the writer needs from a finished batch
std::unique_ptr<Job> collect(const Run &R, rc_batch *b, int seq, const std::vector<int> &blockIdx, std::string &err) {
  std::unique_ptr<Job> j(new Job());
  j->seq = seq;
  j->blockIdx = blockIdx;
  const int m = rc_batch_size(b);
  j->offs.assign(static_cast<size_t>(m) + 1, 0);
  if (rc_batch_hss_all(b, nullptr, 0, j->offs.data()) != RC_OK) { err = rc_last_error(); return nullptr; }
  j->hss.resize(static_cast<size_t>(std::max<int64_t>(j->offs[m], 1)));
  if (rc_batch_hss_all(b, j->hss.data(), j->offs[m], j->offs.data()) != RC_OK) { err = rc_last_error(); return nullptr; }
  j->status.resize(m);
  j->why.resize(m);
  for (int i = 0; i < m; i++) {
    j->status[i] = rc_batch_status(b, i);
    if (j->status[i] != RC_OK && j->status[i] != RC_ERR_SKIP) { const char *why = rc_batch_block_error(b, i); j->why[i] = why ? why : ""; }
  }
  if (R.list.eps) {
    j->eps.resize(m);
    for (int i = 0; i < m && err.empty(); i++)
      if (j->status[i] == RC_OK)
        j->eps[i] = draw_block(R, b, i, R.blocks[blockIdx[i]], std::vector<rc_hss>(j->hss.begin() + j->offs[i], j->hss.begin() + j->offs[i + 1]), err);
    if (!err.empty()) return nullptr;
  }
  return j;
}

struct Times { double ctx = 0, trees = 0, treeWait = 0, submit = 0, wait = 0, fetch = 0; };

// One GPU's share of a many-block input: its sub-batches (ranges of the input, in input order), trees fitted on this GPU in chunks
// that run ahead of the scoring on a thread of their own, scoring as a stream, results to the writer.
struct Worker {
  int id = 0, device = 0;
  rc_ctx *ctx = nullptr;
  std::vector<std::pair<int, int>> parts;   // (first input block, count) of each of its sub-batches
  std::vector<int> seqs;                    // their positions in the input's sequence of sub-batches
  Times t;
};

void run_worker(Run &R, Worker &W, int subBlocks) {
  // the worker's blocks, contiguous in its own arrays
  std::vector<int> idx;
  for (auto &p : W.parts) for (int i = 0; i < p.second; i++) idx.push_back(p.first + i);
  const int n = static_cast<int>(idx.size());
  if (n == 0) return;
  std::vector<rc_block> rb(n);
  for (int i = 0; i < n; i++) rb[i] = R.rb[idx[i]];
  std::vector<std::string> tree(n);
  std::vector<float> kappa(n, 0.0f);
  if (R.haveSidecar) for (int i = 0; i < n; i++) { tree[i] = R.blocks[idx[i]].tree; kappa[i] = R.blocks[idx[i]].kappa; }
  // What treeML() hands over (RNAcode.c:153): tree + kappa, fitted on the GPU (one wavefront per block, latency-bound: a call costs
  // about the same for 100 blocks as for the 2048 the chip holds at once) -- on a thread of its own, in chunks that run ahead of the
  // scoring: 2048 blocks first so that the first sub-batches can go, then doubling up to 8192 (a remainder of less than half a
  // chunk goes with the chunk before it).  The scoring loop below only waits when it has caught up with the fits.
  std::mutex tm;
  std::condition_variable tcv;
  int fitted = R.haveSidecar ? n : 0;      // blocks [0, fitted) have their tree (or a sidecar entry, or none to be had)
  std::string treeErr;
  std::thread treeThread;
  if (!R.haveSidecar) treeThread = std::thread([&] {
    std::vector<char> nwk;
    std::vector<float> kap;
    int at = 0;
    for (int chunk = std::max(subBlocks, 2048); at < n && !R.failed.load(); chunk = std::min(2 * chunk, std::max(subBlocks, 8192))) {
      const int mf = (n - at <= chunk + chunk / 2) ? n - at : chunk;
      nwk.assign(static_cast<size_t>(mf) * R.cap, 0);
      kap.assign(mf, 0.0f);
      const double t0 = now();
      const int r = rc_fit_trees_device(W.ctx, rb.data() + at, mf, nwk.data(), R.cap, kap.data(), nullptr);
      W.t.trees += now() - t0;
      if (r >= 0)
        for (int i = 0; i < mf; i++) { tree[at + i] = nwk.data() + static_cast<size_t>(i) * R.cap; kappa[at + i] = kap[i]; }
      {
        std::lock_guard<std::mutex> lk(tm);
        if (r < 0) { treeErr = rc_last_error(); fitted = n; }
        else fitted = at + mf;
      }
      tcv.notify_all();
      if (r < 0) return;
      at += mf;
    }
    { std::lock_guard<std::mutex> lk(tm); fitted = n; }
    tcv.notify_all();
  });
  auto finish = [&](const std::string &msg) {
    if (!msg.empty()) R.fail(msg);
    if (treeThread.joinable()) treeThread.join();
  };
  rc_stream *stream = nullptr;   // (created beside the first tree fits: it brings up the HIP streams of the scoring pipeline)
  if (rc_stream_create(W.ctx, &R.par, 3, &stream) != RC_OK) { finish(rc_last_error()); return; }
  auto trees_ready = [&](int upto, bool wait) {   // are the trees of blocks [0, upto) there?
    std::unique_lock<std::mutex> lk(tm);
    if (wait) tcv.wait(lk, [&] { return fitted >= upto; });
    return fitted >= upto;
  };
  size_t sent = 0, taken = 0;
  int next = 0;
  std::deque<std::pair<int, int>> inflight;   // (part index, first local block)
  while ((sent < W.parts.size() || rc_stream_pending(stream) > 0) && !R.failed.load()) {
    while (sent < W.parts.size() && rc_stream_pending(stream) < 3) {
      const int m = W.parts[sent].second;
      // the fits have not got this far: take a finished batch first if there is one, else wait for them
      if (!trees_ready(next + m, false)) {
        if (rc_stream_pending(stream) > 0) break;
        const double t0 = now();
        trees_ready(next + m, true);
        W.t.treeWait += now() - t0;
      }
      { std::lock_guard<std::mutex> lk(tm); if (!treeErr.empty()) { finish(treeErr); rc_stream_destroy(stream); return; } }
      for (int i = next; i < next + m; i++) { rb[i].newick = tree[i].empty() ? nullptr : tree[i].c_str(); rb[i].kappa = kappa[i]; }
      const double t0 = now();
      if (rc_stream_submit(stream, rb.data() + next, m) != RC_OK) { finish(rc_last_error()); rc_stream_destroy(stream); return; }
      W.t.submit += now() - t0;
      inflight.emplace_back(static_cast<int>(sent), next);
      next += m;
      sent++;
    }
    rc_batch *b = nullptr;
    double t0 = now();
    if (rc_stream_next(stream, &b) != RC_OK) { finish(rc_last_error()); rc_stream_destroy(stream); return; }
    W.t.wait += now() - t0;
    t0 = now();
    const auto part = inflight.front();
    inflight.pop_front();
    std::vector<int> blockIdx(idx.begin() + part.second, idx.begin() + part.second + W.parts[part.first].second);
    std::string err;
    std::unique_ptr<Job> j = collect(R, b, W.seqs[part.first], blockIdx, err);
    W.t.fetch += now() - t0;
    rc_stream_recycle(stream, b);
    if (!j) { finish(err); rc_stream_destroy(stream); return; }
    post(R, std::move(j));
    taken++;
  }
  finish("");
  rc_stream_destroy(stream);
}

// Few blocks, several GPUs: every GPU simulates the samples [lo, hi) of EVERY block (seed_base + lo: sample s of a block is seeded
// seed_base + s whichever GPU simulates it), the slices meet on the host, and the fit runs on the gathered rows -- getExtremeValuePars
// (score.c:976-1064) with its loop :1004-1048 split and its fit :1050 after the gather.  --stop-early's verdict needs only the
// gathered row (the count of samples above the best native score only grows along the loop, :1036-1042).
bool run_sample_split(Run &R, std::vector<Worker> &W, std::string &err) {
  const int n = static_cast<int>(R.blocks.size()), G = static_cast<int>(W.size());
  const int sampleN = R.par.sampleN, groups = (sampleN + 63) / 64;
  // trees: once, on the first GPU
  std::vector<char> nwk;
  std::vector<float> kap(n, 0.0f);
  if (!R.haveSidecar) {
    nwk.assign(static_cast<size_t>(n) * R.cap, 0);
    const double t0 = now();
    if (rc_fit_trees_device(W[0].ctx, R.rb.data(), n, nwk.data(), R.cap, kap.data(), nullptr) < 0) { err = rc_last_error(); return false; }
    W[0].t.trees += now() - t0;
    for (int i = 0; i < n; i++) { R.blocks[i].tree = nwk.data() + static_cast<size_t>(i) * R.cap; R.blocks[i].kappa = kap[i]; }
  }
  for (int i = 0; i < n; i++) { R.rb[i].newick = R.blocks[i].tree.empty() ? nullptr : R.blocks[i].tree.c_str(); R.rb[i].kappa = R.blocks[i].kappa; }
  std::vector<rc_batch *> batch(G, nullptr);
  std::vector<std::vector<float>> slice(G);
  std::vector<int> lo(G), hi(G);
  std::vector<std::string> werr(G);
  std::vector<std::thread> th;
  for (int g = 0; g < G; g++) {
    const int glo = groups * g / G, ghi = groups * (g + 1) / G;   // whole wavefront groups: only the last slice is ragged
    lo[g] = std::min(sampleN, glo * 64); hi[g] = std::min(sampleN, ghi * 64);
    if (hi[g] <= lo[g] && g > 0) continue;
    th.emplace_back([&, g] {
      rc_params p = R.par;
      p.sampleN = std::max(1, hi[g] - lo[g]); p.seed_base = R.par.seed_base + static_cast<uint32_t>(lo[g]); p.stopEarly = 0;
      const double t0 = now();
      if (rc_batch_create(W[g].ctx, R.rb.data(), n, &p, &batch[g]) != RC_OK || rc_batch_run(batch[g]) != RC_OK) { werr[g] = rc_last_error(); return; }
      slice[g].resize(static_cast<size_t>(n) * p.sampleN);
      if (rc_batch_maxima_all(batch[g], slice[g].data()) != RC_OK) werr[g] = rc_last_error();
      W[g].t.wait += now() - t0;
    });
  }
  for (auto &t : th) t.join();
  for (int g = 0; g < G; g++) if (!werr[g].empty()) { err = werr[g]; return false; }
  // the native HSS lists are the same on every GPU: the first one's, with p-values from the fit of the gathered row
  std::vector<int> blockIdx(n);
  for (int i = 0; i < n; i++) blockIdx[i] = i;
  std::unique_ptr<Job> j(new Job());
  j->seq = 0; j->blockIdx = blockIdx;
  j->offs.assign(static_cast<size_t>(n) + 1, 0);
  if (rc_batch_hss_all(batch[0], nullptr, 0, j->offs.data()) != RC_OK) { err = rc_last_error(); return false; }
  j->hss.resize(static_cast<size_t>(std::max<int64_t>(j->offs[n], 1)));
  if (rc_batch_hss_all(batch[0], j->hss.data(), j->offs[n], j->offs.data()) != RC_OK) { err = rc_last_error(); return false; }
  j->status.resize(n); j->why.resize(n);
  std::vector<double> row(sampleN);
  for (int i = 0; i < n; i++) {
    j->status[i] = rc_batch_status(batch[0], i);
    if (j->status[i] != RC_OK) { if (j->status[i] != RC_ERR_SKIP) { const char *why = rc_batch_block_error(batch[0], i); j->why[i] = why ? why : ""; } continue; }
    for (int g = 0; g < G; g++) {
      if (!batch[g]) continue;
      const int w = std::max(1, hi[g] - lo[g]);
      for (int s = lo[g]; s < hi[g]; s++) row[s] = static_cast<double>(slice[g][static_cast<size_t>(i) * w + (s - lo[g])]);
    }
    float best = -1.0f;
    for (int64_t k = j->offs[i]; k < j->offs[i + 1]; k++) best = std::max(best, j->hss[k].score);
    int rc = -1;
    double mu = 0, lambda = 0;
    bool stopped = false;
    if (R.par.stopEarly) {   // score.c:992,1036-1042
      int better = 0;
      for (int s = 0; s < sampleN; s++) better += (static_cast<float>(row[s]) > best);
      stopped = better > static_cast<int>(R.par.cutoff * static_cast<float>(sampleN));
    }
    if (!stopped) rc = rc_evd_fit(W[0].ctx, row.data(), sampleN, &mu, &lambda) == 1 ? 1 : -1;
    const float mu32 = static_cast<float>(mu), lam32 = static_cast<float>(lambda);   // *parMu = mu, score.c:1051-1052
    for (int64_t k = j->offs[i]; k < j->offs[i + 1]; k++) j->hss[k].pvalue = rc == 1 ? rc_pvalue(j->hss[k].score, mu32, lam32) : 99.0f;   // RNAcode.c:180-188
  }
  if (R.list.eps) {
    j->eps.resize(n);
    for (int i = 0; i < n && err.empty(); i++)
      if (j->status[i] == RC_OK)
        j->eps[i] = draw_block(R, batch[0], i, R.blocks[i], std::vector<rc_hss>(j->hss.begin() + j->offs[i], j->hss.begin() + j->offs[i + 1]), err);
  }
  for (rc_batch *b : batch) if (b) rc_batch_destroy(b);
  if (!err.empty()) return false;
  post(R, std::move(j));
  return true;
}

}  // namespace

int main(int argc, char **argv) {
  Run R;
  rc_default_params(&R.par);
  rc_params &par = R.par;
  Listing &list = R.list;
  std::string file, outfile, trees, limit, devicesArg;
  int device = 0, subBlocks = 0, gpus = 1;   // subBlocks 0: the library's schedule
  bool dumpBlocks = false;
  for (int a = 1; a < argc; a++) {
    const std::string o = argv[a];
    auto val = [&]() -> const char * { if (a + 1 >= argc) { usage(); std::exit(2); } return argv[++a]; };
    if (o == "-o" || o == "--outfile") outfile = val();
    else if (o == "-g" || o == "--gtf") list.fmt = 1;
    else if (o == "-t" || o == "--tabular") list.fmt = 2;
    else if (o == "-b" || o == "--best-only") list.bestOnly = true;
    else if (o == "-r" || o == "--best-region") list.bestRegion = true;
    else if (o == "-s" || o == "--stop-early") par.stopEarly = 1;
    else if (o == "-n" || o == "--num-samples") par.sampleN = std::atoi(val());
    else if (o == "-p" || o == "--cutoff") par.cutoff = static_cast<float>(std::atof(val()));
    else if (o == "-m" || o == "--blosum") par.blosum = std::atoi(val());
    else if (o == "-c" || o == "--pars") {
      float *dst[4] = {&par.Delta, &par.Omega, &par.omega, &par.stopPenalty_0};
      std::stringstream ss(val());
      std::string item;
      for (int i = 0; i < 4 && std::getline(ss, item, ','); i++) *dst[i] = static_cast<float>(std::atof(item.c_str()));
    } else if (o == "-l" || o == "--limit") limit = val();
    else if (o == "--trees") trees = val();
    else if (o == "--seed-base") par.seed_base = static_cast<uint32_t>(std::strtoul(val(), nullptr, 10));
    else if (o == "--device") device = std::atoi(val());
    else if (o == "--gpus") gpus = std::max(1, std::atoi(val()));
    else if (o == "--devices") devicesArg = val();
    else if (o == "--sub-blocks") subBlocks = std::max(1, std::atoi(val()));
    else if (o == "--dump-blocks") dumpBlocks = true;
    else if (o == "-h" || o == "--help") { usage(); return 0; }
    else if (o == "-e" || o == "--eps") list.eps = true;
    else if (o == "-i" || o == "--eps-cutoff") list.epsCutoff = static_cast<float>(std::atof(val()));
    else if (o == "-d" || o == "--eps-dir") list.epsDir = val();
    else if (!o.empty() && o[0] == '-' && o != "-") { usage(); return 2; }
    else file = o;
  }
  if (par.blosum != 62 && par.blosum != 90) die("Currently only BLOSUM62 and BLOSUM90 are supported.");
  list.cutoff = par.cutoff;
  std::vector<int> devices;
  if (!devicesArg.empty()) {
    std::stringstream ss(devicesArg);
    std::string item;
    while (std::getline(ss, item, ',')) if (!item.empty()) devices.push_back(std::atoi(item.c_str()));
    if (static_cast<int>(devices.size()) != gpus) {
      if (gpus == 1) gpus = static_cast<int>(devices.size());
      else die("--devices must name one device per GPU of --gpus");
    }
  } else for (int g = 0; g < gpus; g++) devices.push_back(gpus == 1 ? device : g);

  const double tMain = now();
  g_t0 = tMain;
  double tRead = now();
  // the HIP runtime and the contexts come up (0.1-0.3 s) on threads of their own while this one reads and parses the input
  std::vector<Worker> W(gpus);
  std::vector<int> ctxRc(gpus, RC_OK);
  std::vector<std::string> ctxErr(gpus);
  std::thread ctxThread;
  if (!dumpBlocks) ctxThread = std::thread([&] {
    std::vector<std::thread> more;
    auto up = [&](int g) {
      const double t = now();
      W[g].id = g; W[g].device = devices[g];
      ctxRc[g] = rc_ctx_create(devices[g], &W[g].ctx);
      if (ctxRc[g] != RC_OK) ctxErr[g] = rc_last_error();
      W[g].t.ctx = now() - t;
    };
    for (int g = 1; g < gpus; g++) more.emplace_back(up, g);
    up(0);
    for (auto &t : more) t.join();
  });
  g_ctxThread = &ctxThread;
  std::vector<Block> &blocks = R.blocks;
  if (file.empty() || file == "-") blocks = read_alignment(stdin);
  else {
    FILE *in = std::fopen(file.c_str(), "rb");
    if (!in) die("Could not open input file " + file);
    blocks = read_alignment(in);
    std::fclose(in);
  }
  const int nRead = static_cast<int>(blocks.size());
  for (int i = 0; i < nRead; i++) blocks[i].index = i;
  if (std::getenv("RC_CLI_TIMES")) std::fprintf(stderr, "[rnacode_hip] %d blocks read and parsed in %.3f s\n", nRead, now() - tRead);
  if (list.eps)   // the plots show the rows as main() leaves them: upper-cased (RNAcode.c:121-128; the library upper-cases its own copy)
    for (Block &b : blocks) for (Row &r : b.rows) for (size_t x = 0; x < r.seq.size(); x++) { char &c = const_cast<char &>(r.seq[x]); c = static_cast<char>(std::toupper(static_cast<unsigned char>(c))); }
  if (!limit.empty()) {   // pruneAln (rnaz_utils.c:724-752, RNAcode.c:130-132): rows whose name starts with a listed string stay
    std::vector<std::string> keep;
    std::stringstream ss(limit);
    std::string item;
    while (std::getline(ss, item, ',')) if (!item.empty()) keep.push_back(item);
    std::vector<Block> kept;
    for (Block &b : blocks) {
      Block nb;
      nb.tree = b.tree; nb.kappa = b.kappa; nb.index = b.index;
      for (const Row &r : b.rows)
        if (std::any_of(keep.begin(), keep.end(), [&](const std::string &x) { return r.name.compare(0, x.size(), x) == 0; })) nb.rows.push_back(r);
      if (nb.rows.empty()) {   // (the reference dereferences the missing first row here)
        std::fprintf(stderr, "Skipping alignment. There must be at least three sequences in the alignment.\n");
        continue;
      }
      kept.push_back(std::move(nb));
    }
    blocks.swap(kept);
  }
  const int n = static_cast<int>(blocks.size());
  if (dumpBlocks) {   // reader check (tests): what was parsed, one record per row, no device needed
    for (int i = 0; i < n; i++) {
      std::printf("B %d\n", i);
      for (const Row &r : blocks[i].rows) std::printf("S %s %d %d %c %s\n", r.name.data(), r.start, r.length, r.strand, r.seq.data());
      std::printf("E\n");
    }
    return 0;
  }
  if (!trees.empty()) {   // one '<newick> TAB <kappa>' line per block of the input file, '-' for blocks without a tree
    std::ifstream in(trees);
    if (!in) die("Could not open " + trees);
    std::vector<int> at(nRead, -1);   // input index -> kept block (--limit may have dropped some)
    for (int i = 0; i < n; i++) at[blocks[i].index] = i;
    std::string line;
    int i = 0;
    while (std::getline(in, line)) {
      if (line.find_first_not_of(" \t\r\n") == std::string::npos) continue;
      if (i < nRead && at[i] >= 0) {
        const size_t tab = line.find('\t');
        if (tab != std::string::npos) { blocks[at[i]].tree = line.substr(0, tab); blocks[at[i]].kappa = static_cast<float>(std::atof(line.c_str() + tab + 1)); }
      }
      i++;
    }
    if (i != nRead) die(std::to_string(nRead) + " alignment blocks but " + std::to_string(i) + " sidecar entries");
    R.haveSidecar = true;
  }

  if (!outfile.empty()) { list.out = std::fopen(outfile.c_str(), "w"); if (!list.out) die("Could not open " + outfile); }
  if (list.eps && rc_code_tables(par.blosum, R.tables.pep, R.tables.matrix) != RC_OK) die(rc_last_error());
  tRead = now() - tRead;
  if (ctxThread.joinable()) ctxThread.join();
  for (int g = 0; g < gpus; g++) if (ctxRc[g] != RC_OK) die(ctxErr[g]);
  const auto t0 = std::chrono::steady_clock::now();
  if (gpus > 1) {   // each context its share of the CPUs this process may use
    const int threads = std::max(1, rc_host_cpus() / gpus);
    for (int g = 0; g < gpus; g++) (void)rc_ctx_set_host_threads(W[g].ctx, threads);
  }

  // rc_block views of the blocks
  std::vector<std::vector<const char *>> rowPtr(n), namePtr(n);
  R.rb.resize(n);
  for (int i = 0; i < n; i++) {
    for (const Row &r : blocks[i].rows) { rowPtr[i].push_back(r.seq.data()); namePtr[i].push_back(r.name.data()); }   // (NUL-terminated: see Row)
    rc_block &b = R.rb[i];
    std::memset(&b, 0, sizeof b);
    b.n_rows = static_cast<int>(blocks[i].rows.size());
    b.n_cols = static_cast<int>(blocks[i].rows[0].seq.size());
    b.rows = rowPtr[i].data(); b.names = namePtr[i].data();
    b.ref_start = blocks[i].rows[0].start; b.ref_length = blocks[i].rows[0].length;
  }
  // per block: room for the longest Newick text -- per tip its name, ':' and a "%f" length, per internal node two brackets, a comma
  // and a length (a "%f" of a length <= 100 has at most 10 characters)
  for (const Block &b : blocks) {
    size_t need = 16;
    for (const Row &r : b.rows) need += r.name.size() + 32;
    R.cap = std::max(R.cap, static_cast<int>(need));
  }

  std::thread writer([&] { writer_thread(R); });
  // few blocks on several GPUs: split the sample range instead of the blocks
  int scorable = 0;
  for (const Block &b : blocks) scorable += b.rows.size() >= 3;
  const bool sampleSplit = gpus > 1 && scorable < 2 * gpus && (par.sampleN + 63) / 64 >= gpus;
  std::string mode = gpus == 1 ? "one GPU" : sampleSplit ? "sample ranges over the GPUs" : "sub-batches dealt to the GPUs in turn";
  if (n == 0) {
    { std::lock_guard<std::mutex> lk(R.jm); R.totalSeq = 0; }
    R.jcv.notify_all();
  } else if (sampleSplit) {
    { std::lock_guard<std::mutex> lk(R.jm); R.totalSeq = 1; }
    std::string err;
    if (!run_sample_split(R, W, err)) R.fail(err);
  } else {
    // sub-batches: the library's schedule (small first, then doubling, whole rounds of the chip; every row count is a launch of its
    // own, so more classes mean larger sub-batches), or --sub-blocks B of equal size; dealt to the GPUs in turn
    std::vector<int32_t> plan;
    if (subBlocks > 0) { for (int at = 0; at < n; at += subBlocks) plan.push_back(std::min(subBlocks, n - at)); }
    else {
      std::vector<char> seen(RC_MAX_ROWS + 2, 0);
      int classes = 0;
      for (const Block &b : blocks) { const size_t r = std::min<size_t>(b.rows.size(), RC_MAX_ROWS + 1); if (!seen[r]) { seen[r] = 1; classes++; } }
      plan.resize(256);
      const int k = rc_stream_plan(W[0].ctx, &par, n, std::max(1, classes), plan.data(), static_cast<int32_t>(plan.size()));
      if (k < 0) die(rc_last_error());
      plan.resize(static_cast<size_t>(k));
      int covered = 0;
      for (int32_t v : plan) covered += v;
      if (covered < n) plan.push_back(n - covered);
      subBlocks = 2048;
    }
    int at = 0;
    for (size_t k = 0; k < plan.size(); k++) {
      Worker &w = W[k % gpus];
      w.parts.emplace_back(at, plan[k]);
      w.seqs.push_back(static_cast<int>(k));
      at += plan[k];
    }
    { std::lock_guard<std::mutex> lk(R.jm); R.totalSeq = static_cast<int>(plan.size()); }
    R.jcv.notify_all();
    std::vector<std::thread> th;
    for (int g = 1; g < gpus; g++) th.emplace_back([&, g] { run_worker(R, W[g], subBlocks); });
    run_worker(R, W[0], subBlocks);
    for (auto &t : th) t.join();
  }
  R.jcv.notify_all();
  writer.join();
  if (R.failed.load()) { g_ctxThread = nullptr; die(R.error); }
  if (std::getenv("RC_CLI_TIMES")) {
    Times s;
    for (const Worker &w : W) { s.ctx = std::max(s.ctx, w.t.ctx); s.trees += w.t.trees; s.treeWait += w.t.treeWait; s.submit += w.t.submit; s.wait += w.t.wait; s.fetch += w.t.fetch; }
    std::fprintf(stderr, "[rnacode_hip] read %.3f s, context (beside the reading) %.3f s, trees (their own thread) %.3f s of which the scoring waited %.3f s, submit %.3f s, wait %.3f s, results %.3f s, listing (its own thread) %.3f s, main() so far %.3f s%s\n",
                 tRead, s.ctx, s.trees, s.treeWait, s.submit, s.wait, s.fetch, R.tList, now() - tMain,
                 gpus > 1 ? (", " + std::to_string(gpus) + " GPUs (sums over them): " + mode).c_str() : "");
  }
  if (list.fmt == 0) {   // RNAcode.c:223-228
    const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    std::fprintf(list.out, "\n%i alignment(s) scored in %.2f seconds. Parameters used:\nN=%i, Delta=%.2f, Omega=%.2f, omega=%.2f, stop penalty=%.2f\n\n",
                 nRead, secs, par.sampleN, static_cast<double>(par.Delta), static_cast<double>(par.Omega), static_cast<double>(par.omega),
                 static_cast<double>(par.stopPenalty_k));
  }
  // Everything has been written; every batch has been waited for.  Tearing the HIP runtime down (streams, code objects, the device
  // context) takes longer than the operating system needs to reclaim the process, so a driver that is done leaves at once
  // (RC_CLI_TEARDOWN=1: the orderly way, for leak checkers).
  if (list.out != stdout) std::fclose(list.out);
  std::fflush(stdout);
  std::fflush(stderr);
  if (!std::getenv("RC_CLI_TEARDOWN")) {
    close(STDOUT_FILENO);   // a reader of the listing sees its end now, not when the kernel has released the process's GPU resources
    _exit(0);
  }
  for (Worker &w : W) rc_ctx_destroy(w.ctx);
  return 0;
}
