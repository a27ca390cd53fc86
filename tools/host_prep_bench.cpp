// host_prep_bench.cpp -- times prepare_block (rc_host.cpp: the host part of a batch's preparation) on one thread, no GPU needed,
// and prints a hash of everything it wrote, so that a change to the preparation can be shown to leave the tables byte-identical.
//   g++ -O3 -std=c++17 -ffp-contract=off -o /tmp/host_prep_bench tools/host_prep_bench.cpp rnacode_amd/csrc/rc_host.cpp -Iinclude
//   python -c "from rnacode_amd.synth import synth_blocks; from rnacode_amd.alnio import write_harness_blocks;
//              write_harness_blocks([b.upper() for b in synth_blocks(1250, 6, 120, seed=1)], '/tmp/c3.blk')"
//   /tmp/host_prep_bench /tmp/c3.blk 20
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>

#include "../rnacode_amd/csrc/rc_host.h"

using namespace rc;

struct Blk {
  std::vector<std::string> names, seqs;
  std::string tree;
  float kappa = 0;
  int start = 0, length = 0;
};

int main(int argc, char **argv) {
  if (argc < 2) { std::fprintf(stderr, "usage: %s blocks.blk [repeats]\n", argv[0]); return 2; }
  const int reps = argc > 2 ? std::atoi(argv[2]) : 10;
  std::ifstream in(argv[1]);
  std::vector<Blk> blks;
  std::string line;
  Blk cur;
  while (std::getline(in, line)) {
    if (line.empty()) continue;
    std::istringstream ss(line);
    std::string tag;
    ss >> tag;
    if (tag == "B") cur = Blk();
    else if (tag == "S") {
      std::string name, strand, seq;
      int st, len, full;
      ss >> name >> st >> len >> strand >> full >> seq;
      if (cur.names.empty()) { cur.start = st; cur.length = len; }
      cur.names.push_back(name); cur.seqs.push_back(seq);
    } else if (tag == "T") { ss >> cur.tree; }
    else if (tag == "K") { ss >> cur.kappa; }
    else if (tag == "E") blks.push_back(cur);
  }
  const int n = static_cast<int>(blks.size());
  std::vector<std::vector<const char *>> rowp(n), namep(n);
  std::vector<rc_block> rb(n);
  size_t hostSum = 0, devSum = 0;
  for (int i = 0; i < n; i++) {
    for (auto &s : blks[i].seqs) rowp[i].push_back(s.c_str());
    for (auto &s : blks[i].names) namep[i].push_back(s.c_str());
    std::memset(&rb[i], 0, sizeof(rc_block));
    rb[i].n_rows = static_cast<int>(blks[i].seqs.size());
    rb[i].n_cols = static_cast<int>(blks[i].seqs[0].size());
    rb[i].rows = rowp[i].data(); rb[i].names = namep[i].data();
    rb[i].ref_start = blks[i].start; rb[i].ref_length = blks[i].length;
    rb[i].newick = blks[i].tree.c_str(); rb[i].kappa = blks[i].kappa;
    size_t h = 0, d = 0;
    block_footprint(rb[i].n_rows, rb[i].n_cols, &h, &d);
    hostSum += h; devSum += d;
  }
  rc_params par;
  par.Delta = -10.0f; par.Omega = -4.0f; par.omega = -2.0f; par.stopPenalty_0 = -9999.0f; par.stopPenalty_k = -8.0f;
  par.blosum = 62; par.sampleN = 1000; par.cutoff = 1.0f; par.stopEarly = 0; par.seed_base = 42;
  CodeTables ct(62);
  PairTable pt;
  pt.build(ct);
  std::vector<uint8_t> host(hostSum + 256);
  std::vector<DevBlock> db(n);
  std::vector<BlockMeta> meta(n);
  double best = 1e30;
  unsigned long long hash = 0;
  for (int r = 0; r < reps; r++) {
    std::memset(host.data(), 0, host.size());
    BlobArena arena;
    arena.host = host.data(); arena.hostCap = host.size(); arena.devCap = devSum + 256;
    const auto t0 = std::chrono::steady_clock::now();
    int bad = 0;
    for (int i = 0; i < n; i++) {
      std::string err;
      const int st = prepare_block(rb[i], par, pt, arena, static_cast<uint32_t>(i), db[i], meta[i], err);
      bad += (st != RC_OK);
    }
    const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    best = std::min(best, us / n);
    if (r == 0) {
      hash = 1469598103934665603ull;
      auto mix = [&](const uint8_t *p, size_t len) { for (size_t x = 0; x < len; x++) { hash ^= p[x]; hash *= 1099511628211ull; } };
      // the host-written sections block by block (independent of where they sit in the blob)
      for (int i = 0; i < n; i++) {
        if (meta[i].status != RC_OK) continue;
        const DevBlock &d = db[i];
        mix(host.data() + d.off_nodes, sizeof(NodeRec) * d.nnodes);
        mix(host.data() + d.off_qtip, d.N);
        mix(host.data() + d.off_qtip16, 2 * d.N);
        mix(host.data() + d.off_chars, static_cast<size_t>(d.N) * d.cols);
        mix(host.data() + d.off_models, sizeof(ModelRec) * 2 * d.N);
        const int hdr[8] = {d.N, d.NK, d.cols, d.L, d.nnodes, d.zw_words, d.zm_words, d.status};
        mix(reinterpret_cast<const uint8_t *>(hdr), sizeof hdr);
      }
      std::printf("%d blocks, %d not OK, %zu host bytes\n", n, bad, arena.hostUsed.load());
    }
  }
  std::printf("best of %d: %.2f us per block on one thread; table hash %016llx\n", reps, best, hash);
  return 0;
}
