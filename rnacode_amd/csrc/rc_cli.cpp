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
//               [--seed-base S] [--device D] [--sub-blocks B] [FILE]
//
// Quirk kept from the reference: the 4th value of --pars goes to stopPenalty_0 (RNAcode.c:318).
#include <algorithm>
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
#include <thread>
#include <vector>

#include <unistd.h>

#include "../../include/rnacode_hip.h"

namespace {

struct Row {
  std::string name, seq;
  int start = 0, length = 0;
  char strand = '?';
};
struct Block {
  std::vector<Row> rows;
  std::string tree;     // empty: none
  float kappa = 0.0f;
  int index = 0;        // position in the input file (the --trees sidecar has one entry per block READ, before --limit drops any)
};

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
// next 'a' line (rnaz_utils.c:132-234)
std::vector<Block> read_maf(const std::vector<std::string> &lines) {
  std::vector<Block> out;
  Block cur;
  for (const std::string &line : lines) {
    const std::vector<std::string> f = fields(line);
    if (f.empty() || f[0][0] == '#') continue;
    if (f[0] == "i" || f[0] == "e" || f[0] == "q") continue;
    if (f[0] == "s") {
      if (f.size() != 7) die("Invalid MAF format (number of fields in 's' line not correct)");
      if (f[4][0] != '+' && f[4][0] != '-') die("Invalid MAF format (strand)");
      Row r;
      r.name = f[1]; r.seq = f[6]; r.start = std::atoi(f[2].c_str()); r.length = std::atoi(f[3].c_str()); r.strand = f[4][0];
      cur.rows.push_back(r);
      continue;
    }
    if (f[0] == "a") {
      if (!cur.rows.empty()) { check_equal(cur); out.push_back(cur); }
      cur = Block();
    }
  }
  if (!cur.rows.empty()) { check_equal(cur); out.push_back(cur); }
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
    for (size_t i = 0; i < names.size(); i++) { Row r; r.name = names[i]; r.seq = seqs[i]; b.rows.push_back(r); }
    check_equal(b);
    out.push_back(b);
  }
  return out;
}

std::vector<Block> read_alignment(std::istream &in) {
  std::vector<std::string> lines;
  std::string line;
  while (std::getline(in, line)) lines.push_back(line);
  for (const std::string &l : lines) {
    if (l.find_first_not_of(" \t\r\n") == std::string::npos) continue;
    if (l.compare(0, 7, "CLUSTAL") == 0) return read_clustal(lines);
    break;
  }
  return read_maf(lines);
}

// printResults (misc.c:392-552); the HSS counter runs across blocks and is not advanced after a --best-only break
struct Listing {
  FILE *out = stdout;
  int fmt = 0;          // 0 default table, 1 GTF, 2 tabular
  float cutoff = 1.0f;
  bool bestOnly = false, bestRegion = false;
  int hitCounter = 0;

  void block(std::vector<rc_hss> res, const std::string &refName) {
    res.erase(std::remove_if(res.begin(), res.end(), [](const rc_hss &h) { return !(h.score > 0.0f); }), res.end());
    std::vector<char> hide(res.size(), 0);
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
    std::vector<size_t> order(res.size());
    for (size_t i = 0; i < order.size(); i++) order[i] = i;
    std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) { return res[a].score > res[b].score; });
    if (res.empty() || res[order[0]].pvalue > cutoff) {
      if (fmt == 0) std::fprintf(out, "\nNo significant coding regions found.\n");
      return;
    }
    if (fmt == 0) {
      std::fprintf(out, "\n%6s%5s%7s%6s%6s%12s%12s%12s%9s%9s\n", " HSS # ", "Frame", "Length", "From", "To", "Name", "Start", "End", "Score", "P");
      std::fprintf(out, "======================================================================================\n");
    }
    for (size_t idx : order) {
      const rc_hss &h = res[idx];
      const double p = static_cast<double>(h.pvalue);
      if (!(h.pvalue < cutoff)) break;
      if (hide[idx]) continue;
      const int length = h.endSite - h.startSite + 1;
      const char strand[2] = {static_cast<char>(h.strand), 0};
      if (fmt == 0) {
        std::fprintf(out, "%6i %4s%i%7i%6i%6i%12s%12i%12i%9.2f", hitCounter, strand, h.frame + 1, length, h.startSite + 1, h.endSite + 1,
                     refName.c_str(), h.startGenomic, h.endGenomic, static_cast<double>(h.score));
        if (p < 0.001) {
          if (p < 10e-16) std::fprintf(out, "   <1e-16\n"); else std::fprintf(out, "% 9.1e\n", p);
        } else std::fprintf(out, "% 9.3f\n", p);
      } else if (fmt == 1) {
        const size_t k = refName.find('.');
        const std::string name = k == std::string::npos ? refName : refName.substr(k + 1);
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
                       "                   [--seed-base S] [--device D] [--sub-blocks B] [--dump-blocks] [FILE]\n");
}

}  // namespace

int main(int argc, char **argv) {
  rc_params par;
  rc_default_params(&par);
  Listing list;
  std::string file, outfile, trees, limit;
  int device = 0, subBlocks = 0;   // 0: 2048, or 512 per distinct row count if that is more
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
    else if (o == "--sub-blocks") subBlocks = std::max(1, std::atoi(val()));
    else if (o == "--dump-blocks") dumpBlocks = true;
    else if (o == "-h" || o == "--help") { usage(); return 0; }
    else if (o == "-e" || o == "--eps") die("EPS plots are drawn by `python -m rnacode_amd.cli -e`");
    else if (!o.empty() && o[0] == '-' && o != "-") { usage(); return 2; }
    else file = o;
  }
  if (par.blosum != 62 && par.blosum != 90) die("Currently only BLOSUM62 and BLOSUM90 are supported.");
  list.cutoff = par.cutoff;

  auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  const double tMain = now();
  double tRead = now(), tCtx = 0, tTrees = 0, tTreeWait = 0, tSubmit = 0, tWait = 0, tList = 0, tFetch = 0;
  // the HIP runtime and the context come up (0.1-0.3 s) on a thread of their own while this one reads and parses the input
  rc_ctx *ctx = nullptr;
  int ctxRc = RC_OK;
  std::string ctxErr;
  std::thread ctxThread;
  if (!dumpBlocks) ctxThread = std::thread([&] {
    const double t = now();
    ctxRc = rc_ctx_create(device, &ctx);
    if (ctxRc != RC_OK) ctxErr = rc_last_error();
    tCtx = now() - t;
  });
  g_ctxThread = &ctxThread;
  std::vector<Block> blocks;
  if (file.empty() || file == "-") blocks = read_alignment(std::cin);
  else {
    std::ifstream in(file);
    if (!in) die("Could not open input file " + file);
    blocks = read_alignment(in);
  }
  const int nRead = static_cast<int>(blocks.size());
  for (int i = 0; i < nRead; i++) blocks[i].index = i;
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
      for (const Row &r : blocks[i].rows) std::printf("S %s %d %d %c %s\n", r.name.c_str(), r.start, r.length, r.strand, r.seq.c_str());
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
  }

  if (!outfile.empty()) { list.out = std::fopen(outfile.c_str(), "w"); if (!list.out) die("Could not open " + outfile); }
  tRead = now() - tRead;
  if (ctxThread.joinable()) ctxThread.join();
  if (ctxRc != RC_OK) die(ctxErr);
  const auto t0 = std::chrono::steady_clock::now();

  // rc_block views of the blocks
  std::vector<std::vector<const char *>> rowPtr(n), namePtr(n);
  std::vector<rc_block> rb(n);
  for (int i = 0; i < n; i++) {
    for (const Row &r : blocks[i].rows) { rowPtr[i].push_back(r.seq.c_str()); namePtr[i].push_back(r.name.c_str()); }
    rc_block &b = rb[i];
    std::memset(&b, 0, sizeof b);
    b.n_rows = static_cast<int>(blocks[i].rows.size());
    b.n_cols = static_cast<int>(blocks[i].rows[0].seq.size());
    b.rows = rowPtr[i].data(); b.names = namePtr[i].data();
    b.ref_start = blocks[i].rows[0].start; b.ref_length = blocks[i].rows[0].length;
  }

  // sub-batches: the library's schedule (small first, then doubling, whole rounds of the chip; every row count is a launch of its
  // own, so more classes mean larger sub-batches), or --sub-blocks B of equal size
  std::vector<int32_t> plan;
  if (subBlocks > 0) { for (int at = 0; at < n; at += subBlocks) plan.push_back(std::min(subBlocks, n - at)); }
  else {
    std::vector<char> seen(RC_MAX_ROWS + 2, 0);
    int classes = 0;
    for (const Block &b : blocks) { const size_t r = std::min<size_t>(b.rows.size(), RC_MAX_ROWS + 1); if (!seen[r]) { seen[r] = 1; classes++; } }
    plan.resize(256);
    const int k = rc_stream_plan(ctx, &par, n, std::max(1, classes), plan.data(), static_cast<int32_t>(plan.size()));
    if (k < 0) die(rc_last_error());
    plan.resize(static_cast<size_t>(k));
    subBlocks = 2048;
  }
  size_t planAt = 0;
  rc_stream *stream = nullptr;   // (created below, beside the first tree fits: it brings up the HIP streams of the scoring pipeline)
  // per block: room for the longest Newick text -- per tip its name, ':' and a "%f" length, per internal node two brackets, a comma
  // and a length (a "%f" of a length <= 100 has at most 10 characters)
  int cap = 64;
  for (const Block &b : blocks) {
    size_t need = 16;
    for (const Row &r : b.rows) need += r.name.size() + 32;
    cap = std::max(cap, static_cast<int>(need));
  }
  int next = 0, base = 0;
  // What treeML() hands over (RNAcode.c:153): tree + kappa, fitted on the GPU (one wavefront per block, latency-bound: a call costs
  // about the same for 100 blocks as for the 2048 the chip holds at once) -- on a thread of its own, in chunks that run ahead of the
  // scoring: 2048 blocks first so that the first sub-batches can go, then doubling up to 8192 (a remainder of less than half a
  // chunk goes with the chunk before it).  The scoring loop below only waits when it has caught up with the fits.
  std::mutex tm;
  std::condition_variable tcv;
  int fitted = trees.empty() ? 0 : n;      // blocks [0, fitted) have their tree (or a sidecar entry, or none to be had)
  std::string treeErr;
  std::thread treeThread;
  if (trees.empty() && n > 0) treeThread = std::thread([&] {
    std::vector<char> nwk;
    std::vector<float> kap;
    int at = 0;
    for (int chunk = std::max(subBlocks, 2048); at < n; chunk = std::min(2 * chunk, std::max(subBlocks, 8192))) {
      const int mf = (n - at <= chunk + chunk / 2) ? n - at : chunk;
      nwk.assign(static_cast<size_t>(mf) * cap, 0);
      kap.assign(mf, 0.0f);
      const double t = now();
      const int r = rc_fit_trees_device(ctx, rb.data() + at, mf, nwk.data(), cap, kap.data(), nullptr);
      tTrees += now() - t;
      if (r >= 0)
        for (int i = 0; i < mf; i++) { blocks[at + i].tree = nwk.data() + static_cast<size_t>(i) * cap; blocks[at + i].kappa = kap[i]; }
      {
        std::lock_guard<std::mutex> lk(tm);
        if (r < 0) { treeErr = rc_last_error(); fitted = n; }
        else fitted = at + mf;
      }
      tcv.notify_all();
      if (r < 0) return;
      at += mf;
    }
  });
  if (rc_stream_create(ctx, &par, 3, &stream) != RC_OK) { if (treeThread.joinable()) treeThread.join(); die(rc_last_error()); }
  auto trees_ready = [&](int upto, bool wait) {   // are the trees of blocks [0, upto) there?
    std::unique_lock<std::mutex> lk(tm);
    if (wait) tcv.wait(lk, [&] { return fitted >= upto; });
    if (!treeErr.empty()) { lk.unlock(); if (treeThread.joinable()) treeThread.join(); die(treeErr); }
    return fitted >= upto;
  };
  // The listing is written by a thread of its own, batch by batch in input order: formatting 10^5 lines takes as long as the GPU
  // needs for the blocks behind them, and this thread has the next sub-batch to prepare.
  struct Job {
    int base = 0, m = 0;
    std::vector<rc_hss> hss;
    std::vector<int64_t> offs;
    std::vector<int> status;
    std::vector<std::string> why;
  };
  std::mutex jm;
  std::condition_variable jcv;
  std::deque<std::unique_ptr<Job>> jobs;
  bool jobsDone = false;
  std::thread printer([&] {
    for (;;) {
      std::unique_ptr<Job> j;
      {
        std::unique_lock<std::mutex> lk(jm);
        jcv.wait(lk, [&] { return jobsDone || !jobs.empty(); });
        if (jobs.empty()) return;
        j = std::move(jobs.front());
        jobs.pop_front();
      }
      const double t = now();
      for (int i = 0; i < j->m; i++) {
        const int st = j->status[i];
        const Block &blk = blocks[j->base + i];
        if (st == RC_ERR_SKIP) {   // RNAcode.c:142-150
          std::fprintf(stderr, "Skipping alignment. %s\n", blk.rows.size() <= 2 ? "There must be at least three sequences in the alignment." : "Too short.");
          continue;
        }
        if (st != RC_OK) {         // RNAcode.c:153-156
          std::fprintf(stderr, "Skipping alignment. Failed to build ML tree. (%s)\n", j->why[i].empty() ? "not scored" : j->why[i].c_str());
          continue;
        }
        list.block(std::vector<rc_hss>(j->hss.begin() + j->offs[i], j->hss.begin() + j->offs[i + 1]), blk.rows[0].name);
      }
      tList += now() - t;
    }
  });
  while (next < n || rc_stream_pending(stream) > 0) {
    while (next < n && rc_stream_pending(stream) < 3) {
      const int m = planAt < plan.size() ? std::min<int>(plan[planAt], n - next) : n - next;
      // the fits have not got this far: take a finished batch first if there is one, else wait for them
      if (!trees_ready(next + m, false)) {
        if (rc_stream_pending(stream) > 0) break;
        const double t = now();
        trees_ready(next + m, true);
        tTreeWait += now() - t;
      }
      planAt++;
      for (int i = next; i < next + m; i++) { rb[i].newick = blocks[i].tree.empty() ? nullptr : blocks[i].tree.c_str(); rb[i].kappa = blocks[i].kappa; }
      const double t = now();
      if (rc_stream_submit(stream, rb.data() + next, m) != RC_OK) die(rc_last_error());
      tSubmit += now() - t;
      next += m;
    }
    rc_batch *b = nullptr;
    double t = now();
    if (rc_stream_next(stream, &b) != RC_OK) die(rc_last_error());
    tWait += now() - t;
    t = now();
    std::unique_ptr<Job> j(new Job());
    j->base = base;
    j->m = rc_batch_size(b);
    j->offs.assign(static_cast<size_t>(j->m) + 1, 0);
    if (rc_batch_hss_all(b, nullptr, 0, j->offs.data()) != RC_OK) die(rc_last_error());
    j->hss.resize(static_cast<size_t>(std::max<int64_t>(j->offs[j->m], 1)));
    if (rc_batch_hss_all(b, j->hss.data(), j->offs[j->m], j->offs.data()) != RC_OK) die(rc_last_error());
    j->status.resize(j->m);
    j->why.resize(j->m);
    for (int i = 0; i < j->m; i++) {
      j->status[i] = rc_batch_status(b, i);
      if (j->status[i] != RC_OK && j->status[i] != RC_ERR_SKIP) { const char *why = rc_batch_block_error(b, i); j->why[i] = why ? why : ""; }
    }
    base += j->m;
    tFetch += now() - t;
    rc_stream_recycle(stream, b);
    { std::lock_guard<std::mutex> lk(jm); jobs.push_back(std::move(j)); }
    jcv.notify_one();
  }
  { std::lock_guard<std::mutex> lk(jm); jobsDone = true; }
  jcv.notify_one();
  printer.join();
  if (treeThread.joinable()) treeThread.join();
  rc_stream_destroy(stream);
  if (std::getenv("RC_CLI_TIMES"))
    std::fprintf(stderr, "[rnacode_hip] read %.3f s, context (beside the reading) %.3f s, trees (their own thread) %.3f s of which the scoring waited %.3f s, submit %.3f s, wait %.3f s, results %.3f s, listing (its own thread) %.3f s, main() so far %.3f s\n",
                 tRead, tCtx, tTrees, tTreeWait, tSubmit, tWait, tFetch, tList, now() - tMain);
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
  rc_ctx_destroy(ctx);
  return 0;
}
