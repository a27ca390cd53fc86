// rc_schedule.cpp -- which kernels one run of a batch launches, with what grids, LDS and scratch: the native block's kernels and the
// null-sampling launches per row-count class (rc_batch.cpp's batch_run_async calls in here).
#include "rc_runtime.h"


// persistent workgroups of k_native_dp for `items` (block, strand x frame) items whose longest block has smax codons: each keeps 64 rows
// of S (64 * smax floats), all of them together at most 2 GiB (very long blocks run on fewer workgroups)
// mode 0: beside a k_null that leaves registers free on every SIMD (RC_NATIVE_WAVES_PER_CU, 8); 1: with the chip to itself, every wavefront slot
size_t native_grid(const rc_ctx *c, size_t items, int smax, int mode) {
  const size_t perGroup = static_cast<size_t>(kWave) * std::max(1, smax) * sizeof(float);
  const size_t byMemory = std::max<size_t>(1, (static_cast<size_t>(2) << 30) / perGroup);
  const size_t perCU = mode == 1 ? 32 : static_cast<size_t>(c->nativeWavesPerCU);
  return std::max<size_t>(1, std::min(std::min(items, perCU * c->numCU), byMemory));
}

// A k_null launch that walks two rows per pass (rc_null_kernel.h; N-1 <= kDualRowsMaxNK, codes staged in LDS) needs up to 127 VGPRs:
// four such wavefronts take a SIMD's whole register file, and any other kernel's wavefront on that SIMD displaces one of them (measured:
// the native-block kernels beside it, even one workgroup per CU, 42.5 -> 50.3 ms; an LDS request sized for 15 workgroups per CU did
// not help).  So the two-row instantiations are used where nothing has to run beside them: in batches made of such classes only
// ("fat": every class 3 <= N-1 <= 5 with LDS staging; N-1 = 2 needs 95 VGPRs, leaves a quarter of the registers free and is always
// two-row), whose native-block kernels are queued in front of k_null on the same stream with every wavefront slot to themselves
// (1.9 ms instead of 2.9 at C3) -- resident batches, and sub-batches of a stream that are long enough for the bubble this leaves at
// every sub-batch boundary to be paid back (RC_FAT_STREAM_MIN_ITEMS).  Everything else keeps the one-row instantiations (74 VGPRs)
// and the small kernels beside them.
bool fat_class(const rc_batch *b, const rc_ctx *c, int NK, int maxL) {
  if (NK < 3 || NK > kDualRowsMaxNK || b->allExact) return false;
  const size_t lds = static_cast<size_t>(maxL / 3) * ((NK + 4) / 5) * kWave * sizeof(uint32_t);
  return lds <= c->ldsMaxBytes;   // (longer blocks: the two-row kernel from L2 gains 3-4 % on the launch and loses it again to the native-block kernels queued in front)
}

int launch_native_block(const RunEnv &R) {
  rc_batch *b = R.b; rc_ctx *c = R.c;
  // ---- native block: sigma -> DP per N-1 class -> scan, on the native stream: these kernels are small and
  // latency-bound, the null sampling fills the chip beside them; the fit waits for both
  hipStream_t st = R.nativeFirst ? R.cs : stream_native(c);
  if (!st) return fail(RC_ERR_DEVICE, "hipStreamCreate failed");
  if (!R.nativeFirst) HIP_TRY(hipStreamWaitEvent(st, b->evStart, 0));
  HIP_TRY(b->evN0.record(st));
  NativeArgs na{};
  na.blob = b->dblob.as<uint8_t>(); na.dblocks = R.dblocks; na.blocks = R.okList; na.pair = R.tp.pair;
  na.hss = b->dhss.as<DevHss>(); na.hssCount = b->dhssCount.as<int>(); na.hssCap = b->hssCap; na.tieThr = c->tieThr;
  na.flags = b->dflags.as<uint32_t>();
  HIP_TRY(hipMemsetAsync(b->dhssCount.p, 0, static_cast<size_t>(R.n) * 6 * sizeof(int), st));
  launch_native_sigma(na, static_cast<int>(b->okBlocks.size()), st); b->nl[3]++;
  // Groups of blocks with one row count, each a contiguous piece of its class's list: a per-row-count class is one group; a tiled class (row
  // counts mixed, its list sorted by row count: rc_batch.cpp) one per row count -- the native block of its blocks goes through k_native_dp<N-1>
  // like any block of up to 64 rows (k_native_dp_generic, one wavefront per matrix with its states in memory, took 7 ms for 1662 blocks of
  // 40 x 150 where k_native_dp<39> takes 0.4).
  struct Group { int NK; const int *list; size_t count; int smax; };
  std::vector<Group> groups;
  for (auto &kv : b->classes) {
    if (kv.first == kGenericClass) continue;
    const int *list = R.classList + b->classOff[kv.first];
    const auto &mem = kv.second;
    for (size_t at = 0; at < mem.size();) {
      size_t end = at;
      int smax = 1;
      const int nk = b->meta[mem[at]].NK;
      while (end < mem.size() && b->meta[mem[end]].NK == nk) { smax = std::max(smax, b->meta[mem[end]].L / 3); end++; }
      groups.push_back(Group{nk, list + at, end - at, smax});
      at = end;
    }
  }
  {   // one 64-row buffer per persistent workgroup, shared by the group launches (same stream)
    size_t need = 0;
    for (const Group &g : groups) if (g.NK < kTemplRows) need = std::max(need, native_grid(c, g.count * 6, g.smax, R.nativeMode) * kWave * g.smax);
    HIP_TRY(b->dnativeTile.ensure(std::max<size_t>(need, 4) * sizeof(float)));
  }
  {   // more than 64 rows: states in a scratch, at most 256 blocks per launch share it -- the generic class, and the blocks of that width in the tiled ones
    struct Wide { const int *list; size_t count; int maxNK, smax; };
    std::vector<Wide> wides;
    if (b->classes.count(kGenericClass)) {
      const auto &mem = b->classes[kGenericClass];
      Wide w{R.classList + b->classOff[kGenericClass], mem.size(), 0, 1};
      for (int bi : mem) { w.maxNK = std::max(w.maxNK, b->meta[bi].NK); w.smax = std::max(w.smax, b->meta[bi].L / 3); }
      wides.push_back(w);
    }
    for (const Group &g : groups) if (g.NK >= kTemplRows) wides.push_back(Wide{g.list, g.count, g.NK, g.smax});
    const int chunk = 256;
    auto stride_of = [](const Wide &w) { return static_cast<size_t>(3) * w.maxNK * kWave + static_cast<size_t>(kWave) * w.smax; };   // states, then 64 rows of S
    size_t scratchFloats = 0;
    for (const Wide &w : wides) scratchFloats = std::max(scratchFloats, stride_of(w) * 6 * std::min<size_t>(chunk, w.count));
    if (scratchFloats) HIP_TRY(b->dnativeScratch.ensure(scratchFloats * sizeof(float)));   // (once: the launches below share it in stream order)
    for (const Wide &w : wides) {
      NativeArgs nc = na;
      nc.blocks = w.list;
      const size_t stride = stride_of(w);
      for (size_t at = 0; at < w.count; at += chunk) {
        NativeArgs ng = nc;
        ng.blocks = nc.blocks + at;
        launch_native_dp_generic(ng, static_cast<int>(std::min<size_t>(chunk, w.count - at)), b->dnativeScratch.as<float>(), stride, st);
        b->nl[3]++;
      }
    }
  }
  for (const Group &g : groups) {
    if (g.NK >= kTemplRows) continue;   // (above)
    // up to 64 rows: DP and getHSS fused, 64 rows of one matrix at a time through a per-workgroup buffer (persistent grid)
    NativeArgs nc = na;
    nc.blocks = g.list;
    nc.nItems = static_cast<int>(g.count) * 6;
    nc.tileStride = static_cast<size_t>(kWave) * g.smax;
    const int grid = static_cast<int>(native_grid(c, static_cast<size_t>(nc.nItems), g.smax, R.nativeMode));
    nc.tile = b->dnativeTile.as<float>();
    // every matrix of the group kept for a scan with one lane per matrix, where that fits in a sixteenth of the device memory
    // (headline: 10 000 blocks x 6 x 40 x 40 floats = 384 MB); otherwise DP and scan fused, 64 rows at a time
    const size_t allFloats = static_cast<size_t>(nc.nItems) * g.smax * g.smax;
    if (allFloats * sizeof(float) <= c->totalMem / 16 && nc.nItems >= 6 * 64 && (g.NK <= 16 || g.NK > 32)) {   // (17..32: see k_native_dp)
      HIP_TRY(b->dnativeAll.ensure(std::max<size_t>(allFloats, 4) * sizeof(float)));
      nc.sAll = b->dnativeAll.as<float>(); nc.sAllSites = g.smax;
    }
    if (!launch_native_dp(g.NK, nc, grid, st))
      return fail(RC_ERR_UNSUPPORTED, "no native DP kernel for this number of rows");
    b->nl[3]++;
    if (nc.sAll) { launch_native_scan(nc, st); b->nl[3]++; }
  }
  HIP_TRY(hipMemsetAsync(b->dhssOffsets.as<int>() + R.slots, 0, sizeof(int), st));
  launch_hss_pack(b->dhss.as<DevHss>(), b->dhssCount.as<int>(), b->hssCap, R.slots, b->dhssPacked.as<DevHss>(), b->dhssOffsets.as<int>(),
                  b->dhssOffsets.as<int>() + R.slots, st);
  HIP_TRY(b->evN1.record(st));
  return RC_OK;
}

// null sampling of the sample groups [gLo, gHi) for every class; phase 0 or 1 selects the work counters
// sizeOnly: no launch, *sizeOnly = uint32 of staging scratch this round needs (the rounds of a run share one allocation made up front)
// tail sharing of k_null (rc_null_kernel.h): per launch one claim word and a (block, group) pair per workgroup
size_t steal_slots(const rc_ctx *c) { return static_cast<size_t>(c->numCU) * 32; }
size_t steal_words(const rc_ctx *c) { return 16 + 4 * steal_slots(c); }   // header, claim words, (block, group) pairs, list of published slots

// Launch shape of a k_null that reads its code words from L2 (rc_null_kernel.h): which build -- k_null_occ, one more wavefront per SIMD,
// for batches of one row-count class: round 3 measured +1..7 % there and -5 % on a stream of ten classes, whose small kernels ran in the
// registers those wavefronts take --, how many workgroups per CU, and how much LDS each gets: what phase A needs at least, and with the
// suffix cache everything the occupancy leaves (160 KB / workgroups per CU), which phase B fills with the most re-read code words.
static rc_ctx::PlainPlan plain_plan(rc_ctx *c, int NK, int maxNodes, int wantHi) {   // wantHi: 1 the k_null_occ build, 2 the two-row kernel
  const auto key = std::make_tuple(NK, maxNodes, wantHi);
  auto it = c->plainPlans.find(key);
  if (it != c->plainPlans.end()) return it->second;
  rc_ctx::PlainPlan pp{0, 0, 0};
  for (int hi = wantHi; hi >= 0 && pp.occ <= 0; hi = hi == 2 ? 0 : hi - 1) {
    const size_t minLds = hi ? al256(static_cast<size_t>(maxNodes) * 64 + 64 * 64) : static_cast<size_t>(kPhaseALds);
    const int occ = null_occupancy_plain(NK, minLds, hi);
    if (occ <= 0) continue;
    size_t lds = minLds;
    {
      size_t budget = (c->ldsPerCU / static_cast<size_t>(occ)) & ~static_cast<size_t>(255);
      while (budget > minLds && null_occupancy_plain(NK, budget, hi) < occ) budget -= 256;
      lds = std::max(budget, minLds);
    }
    pp = rc_ctx::PlainPlan{occ, lds, hi};
  }
  c->plainPlans[key] = pp;
  return pp;
}

int launch_null_groups(const RunEnv &R, int gLo, int gHi, int phase, uint32_t extraSkip, size_t *sizeOnly) {
  rc_batch *b = R.b; rc_ctx *c = R.c;
  const int groupsHere = gHi - gLo;
  struct Plan { int NK, grid, gridExact, comboSplit; size_t stride, ldsBytes, need; double itemCost; int hiOcc, cacheSites; size_t plainLds;
                int rowParts, simParts, rowSimGrid; size_t sbufStride, sbufFloats;   // rows split over workgroups (k_null<.., ROWS> + k_null_rowscan)
                int simGrid, roundBlocks; size_t codesBytes, stateBytes;   // (the last four: the wide-block class in two launches)
                int tiledKT; size_t ldsDp; };   // a tiled class (k_generic_sim<true> + k_tiled_dp<KT>): its tile size, the DP's LDS (sigma tables)
  std::vector<Plan> plans;
  trace("null: plan", b);
  size_t needSum = 0, needMax = 0, sbufSum = 0;
  // Several row-count classes: first planned as launches that run side by side (no strand x frame split: together they fill the chip);
  // if their staging areas do not fit side by side, planned again as one launch after the other, each filling the chip by itself.
  bool together = b->classes.size() > 1 && !c->serialNative;
  // ... unless all of them together are too few to fill the chip: then every item is split into its six parts, as for a single small class
  const bool splitAll = static_cast<double>(b->okBlocks.size()) * groupsHere <= c->splitFactor * c->numCU * 16;
 plan_again:
  plans.clear(); needSum = 0; needMax = 0; sbufSum = 0;
  for (auto &kv : b->classes) {
    Plan p{};
    p.NK = kv.first;
    int maxLc = 0, maxNKc = 0;
    for (int bi : kv.second) { maxLc = std::max(maxLc, b->meta[bi].L); maxNKc = std::max(maxNKc, b->meta[bi].NK); }
    p.itemCost = static_cast<double>(maxNKc) * maxLc * maxLc;   // (cell, sequence) steps of the class's longest item, up to a factor
    if (kv.first >= kGenericClass) {   // one class for every block of more than 64 rows, whatever its N: k_generic_sim, then k_generic_dp;
                                       // one per tile size for the blocks of 28..64 rows: k_generic_sim<true>, then k_tiled_dp<KT> (rc_null_tiled.h)
      int maxN = 0, maxNodes = 0;
      for (int bi : kv.second) { maxN = std::max(maxN, b->meta[bi].N); maxNodes = std::max(maxNodes, b->db[bi].nnodes); }
      p.ldsBytes = null_generic_lds_bytes(maxN, maxNodes);
      int occD = 1, occS = 1;
      if (kv.first == kGenericClass) {
        p.codesBytes = null_generic_codes_bytes(maxN, maxLc, maxNodes);
        p.stateBytes = null_generic_state_bytes(maxN, maxLc, maxNodes);
        p.ldsDp = p.ldsBytes;
        const std::pair<int, size_t> keyD(kGenericClass, p.ldsBytes), keyS(-1, p.ldsBytes);
        if (!c->occ.count(keyD)) { c->occ[keyD] = generic_dp_occupancy(p.ldsBytes); c->occ[keyS] = generic_sim_occupancy(p.ldsBytes); }
        occD = std::max(1, c->occ[keyD]); occS = std::max(1, c->occ[keyS]);
      } else {
        p.tiledKT = kTiledMinKT + (kv.first - kTiledClass0);
        for (int bi : kv.second) {
          p.codesBytes = std::max(p.codesBytes, null_tiled_codes_bytes(b->meta[bi].NK, p.tiledKT, b->meta[bi].L));
        }
        p.stateBytes = null_tiled_state_bytes(maxLc);
        p.ldsDp = 0;   // (the kernel's LDS is static: the current tile's sigma tables and the tail of the row buffer, 10 KB -- sixteen workgroups per CU)
        const std::pair<int, size_t> keyD(kv.first, p.ldsDp), keyS(-1, p.ldsBytes);
        if (!c->occ.count(keyD)) c->occ[keyD] = tiled_dp_occupancy(p.tiledKT, p.ldsDp);
        if (!c->occ.count(keyS)) c->occ[keyS] = generic_sim_occupancy(p.ldsBytes);
        occD = std::max(1, c->occ[keyD]); occS = std::max(1, c->occ[keyS]);
      }
      const long long slotsD = static_cast<long long>(c->numCU) * occD;
      const long long items = static_cast<long long>(kv.second.size()) * groupsHere;
      // The sigma codes of every item of a round lie in the scratch at once (N x L x 2 bytes per sample; 5 MB per item at 100 x 300): as few
      // rounds as the scratch budget allows, of equal numbers of blocks.  (Rounds of a whole number of the DP's wavefront slots, as before the
      // items went in parts, left 257 blocks of 64 x 300 -- 4112 items, all within the budget -- a second round of 16 items: a simulation and a
      // DP of single chains on an empty chip, 180 ms against 127 for 256 blocks.)
      const long long budgetBlocks = std::max<long long>(1, static_cast<long long>(c->genericScratchWords * sizeof(uint32_t) / p.codesBytes) / groupsHere);
      const long long nBlocksHere = static_cast<long long>(kv.second.size());
      const long long rounds = (nBlocksHere + budgetBlocks - 1) / budgetBlocks;
      p.roundBlocks = static_cast<int>((nBlocksHere + rounds - 1) / rounds);
      const long long roundItems = static_cast<long long>(p.roundBlocks) * groupsHere;
      // few items: every item's DP is split into its six strand x frame parts to fill the chip (maxima meet in an atomic max)
      // ... and so is every item of a round that fills the chip only a few times over: the simulation is a launch of its own here, the parts redo
      // nothing, and items of minutes-long cost otherwise quantise badly -- 257 blocks of 64 x 300 are 4112 items on 4096 wavefront slots and took as
      // long as 8192 (334 ms against 168 for 294 blocks of 56 x 300: tools/rows_sweep.py)
      p.comboSplit = ((together ? splitAll : static_cast<double>(items) <= c->splitFactor * c->numCU * occD) || roundItems < 8 * slotsD) ? 1 : 0;
      p.grid = static_cast<int>(std::min<long long>(roundItems * (p.comboSplit ? 6 : 1), slotsD));
      p.simGrid = static_cast<int>(std::min<long long>(roundItems, static_cast<long long>(c->numCU) * occS));
      p.need = (static_cast<size_t>(roundItems) * p.codesBytes + static_cast<size_t>(p.grid) * p.stateBytes + 3) / 4 + 64;
      needSum += p.need; needMax = std::max(needMax, p.need);
      plans.push_back(p);
      continue;
    }
    const int NCW = (p.NK + 4) / 5;   // 5 six-bit sigma codes per word
    p.stride = static_cast<size_t>(2) * (maxLc + 1) * code_pos_words(p.NK);   // both strands, positions 0..L, [word][lane] with a narrow last word (rc_device.h)
    // sigma codes of one strand x frame are staged in LDS when that still leaves >= 12 wavefronts per
    // CU (RC_LDS_MAX_BYTES overrides the per-wavefront budget); otherwise they are read from the
    // per-workgroup scratch in global memory.
    p.ldsBytes = static_cast<size_t>(maxLc / 3) * NCW * kWave * sizeof(uint32_t);
    const bool exactOnly = b->allExact || p.NK >= kFastRows;   // wide blocks (N > 32) only have the EXACT instantiation
    // The two-row kernel with its codes staged in LDS runs 160 KB / staged bytes workgroups per CU, and its time goes almost with
    // the inverse of that number (6 / 8 / 10 / 12 / 16 per CU: 93.5 / 71.6 / 59.6 / 51.6 / 42.5 ms at the headline's shape).  Where
    // staging leaves twelve or fewer (blocks of more than ~135 columns), the same kernel reading its codes from L2 behind a suffix
    // cache runs sixteen and is faster: 6 rows x 150 columns 77.3 -> 71.7 ms; at 120 columns (fifteen staged) it is slower, 42.6 -> 47.0.
    // A batch so small that even its strand x frame parts leave most of the chip idle (a caller that scores block by block,
    // RNAcode.c:164-216 through the shim: one block at n = 1000 is 96 parts on 4096 wavefront slots, and the launch lasts as long as
    // the DP of one part, ~2300 cells in a chain): every part's rows are split into up to eight ranges of equal cell counts, each a
    // work item of k_null<.., ROWS>, which leaves the S values in a buffer; k_null_rowscan folds them in getHSS's order afterwards.
    p.rowParts = 1; p.simParts = 1; p.rowSimGrid = 0; p.sbufStride = 0; p.sbufFloats = 0;
    if (c->rowSplit && !exactOnly && p.NK < kFastRows && !R.streaming) {
      const long long slots = static_cast<long long>(c->numCU) * 16;
      const long long partsAll = static_cast<long long>(together ? b->okBlocks.size() : kv.second.size()) * groupsHere * 6;
      if ((!together || splitAll) && partsAll * 2 <= slots) {
        const size_t smax = static_cast<size_t>(maxLc) / 3;
        p.rowParts = static_cast<int>(std::min<long long>(8, slots / partsAll));
        p.sbufStride = smax * (smax + 1) / 2 * kWave;
        p.sbufFloats = p.sbufStride * 6 * kv.second.size() * static_cast<size_t>(groupsHere);
        if (p.sbufFloats * sizeof(float) > (static_cast<size_t>(256) << 20) || smax < 45) { p.rowParts = 1; p.sbufStride = p.sbufFloats = 0; }   // (under ~1000 cells per part the second kernel and the eightfold simulation cost more than the split saves: 4 x 76: 0.11 -> 0.19 ms)
      }
    }
    bool dualL2 = false;
    if (p.rowParts == 1 && !exactOnly && p.NK >= 3 && p.NK <= kDualRowsMaxNK && R.fat && p.ldsBytes <= c->ldsMaxBytes)
      dualL2 = c->ldsPerCU / std::max<size_t>(p.ldsBytes, kPhaseALds) <= 12;
    const bool twoRows = p.NK <= kDualRowsMaxNK && (p.NK == 2 || R.fat);   // (the launch's own test, below)
    if (p.ldsBytes > (twoRows ? c->ldsMaxBytes : std::min(c->ldsMaxBytes, c->ldsMaxBytesOneRow)) || exactOnly || dualL2 || p.rowParts > 1 || (p.NK > kDualRowsMaxNK && !c->stageManyRows)) p.ldsBytes = 0;
    else p.ldsBytes = std::max<size_t>(p.ldsBytes, kPhaseALds);   // phase A keeps the tree's threshold table and the pair table there
    const std::pair<int, size_t> key(p.NK, p.ldsBytes);
    if (!c->occ.count(key)) c->occ[key] = null_occupancy(p.NK, p.ldsBytes);
    if (c->occ[key] <= 0 && p.ldsBytes) { p.ldsBytes = 0; const std::pair<int, size_t> k2(p.NK, 0); if (!c->occ.count(k2)) c->occ[k2] = null_occupancy(p.NK, 0); }
    int occ = std::max(1, c->occ[std::pair<int, size_t>(p.NK, p.ldsBytes)]);
    if (!p.ldsBytes && !exactOnly) {   // codes from L2
      int maxNodes = 0;
      for (int bi : kv.second) maxNodes = std::max(maxNodes, b->db[bi].nnodes);
      const int wantHi = dualL2 ? 2 : (c->highOccupancy == 2 || (c->highOccupancy == 1 && b->classes.size() == 1)) ? 1 : 0;
      const rc_ctx::PlainPlan pp = plain_plan(c, p.NK, maxNodes, wantHi);
      if (pp.occ > 0) {
        occ = pp.occ; p.hiOcc = pp.hiOcc; p.plainLds = pp.lds;
        // (the first NK x 256 bytes of that LDS hold the sigma tables during the DP: the look-ups of these launches read them there)
        const size_t tables = static_cast<size_t>(p.NK) * kWave * sizeof(float);
        p.cacheSites = pp.lds > tables ? static_cast<int>((pp.lds - tables) / (static_cast<size_t>(code_pos_words(p.NK)) * sizeof(uint32_t))) : 0;
      }
    }
    if (c->gridCapPerCU > 0) occ = std::min(occ, c->gridCapPerCU);
    int items = static_cast<int>(kv.second.size()) * groupsHere;
    // small batches: split every item into its six strand x frame parts to fill the chip
    p.comboSplit = (p.rowParts > 1 || (together ? splitAll : static_cast<double>(items) <= c->splitFactor * c->numCU * occ)) ? 1 : 0;
    if (p.rowParts > 1) {   // the simulation of an item in site ranges of about sixteen sites, as many as fill a quarter of the chip
      const long long its = static_cast<long long>(kv.second.size()) * groupsHere;
      p.simParts = static_cast<int>(std::max<long long>(1, std::min<long long>({16, maxLc / 16, static_cast<long long>(c->numCU) * 4 / std::max<long long>(its, 1)})));
      p.rowSimGrid = static_cast<int>(std::min<long long>(its * p.simParts, static_cast<long long>(c->numCU) * occ));
    }
    if (p.comboSplit) items *= 6 * p.rowParts;
    p.grid = std::min(items, c->numCU * occ);
    // very long blocks: bound the sigma-code staging area (4 GiB of uint32) by running fewer workgroups
    p.grid = static_cast<int>(std::max<size_t>(1, std::min<size_t>(p.grid, (static_cast<size_t>(1) << 30) / std::max<size_t>(p.stride, 1))));
    // blocks flagged by k_prep_lut go through the EXACT instantiation in a second launch on the same stream; their
    // number is only known on the device, the launch is a few idle workgroups when there are none
    p.gridExact = exactOnly ? 0 : std::min(p.grid, c->numCU);
    p.need = p.stride * std::max(p.grid, p.gridExact);
    if (p.rowParts > 1) p.need = p.stride * std::max<size_t>(kv.second.size() * static_cast<size_t>(groupsHere), static_cast<size_t>(p.gridExact));   // one scratch per ITEM
    needSum += p.need; needMax = std::max(needMax, p.need);
    sbufSum += p.sbufFloats;
    plans.push_back(p);
  }
  // Several row-count classes: their launches are independent (own work queues, own part of the staging
  // scratch), so they go on separate streams and share the chip -- a small class no longer waits for the tail
  // of the previous one.  One class, a scratch that would exceed 4 GiB, or RC_SERIAL_NATIVE: this run's stream only.
  // the classes with the longest items first (longest-processing-time-first across the launches too)
  std::stable_sort(plans.begin(), plans.end(), [](const Plan &x, const Plan &y) { return x.itemCost > y.itemCost; });
  if (together && needSum > c->togetherWords) { together = false; goto plan_again; }
  const bool concurrent = together;
  const size_t need = concurrent ? needSum : needMax;
  if (sizeOnly) { *sizeOnly = need; b->sbufNeed = std::max(b->sbufNeed, sbufSum); return RC_OK; }
  size_t sbufOff = 0;
  if (need * sizeof(uint32_t) > b->dscratch.cap) return fail(RC_ERR_ARG, "internal: staging scratch was not sized for this round");
  Event &fork = phase ? b->evMid : b->evS0;
  size_t scratchOff = 0;
  if (concurrent) while (b->classDone.size() < static_cast<size_t>(kMaxRounds) * plans.size()) b->classDone.emplace_back(new Event());
  trace("null: planned", b);
  std::vector<Event *> joins;
  for (size_t pi = 0; pi < plans.size(); pi++) {
    const Plan &p = plans[pi];
    const auto &members = b->classes[p.NK];
    hipStream_t st = R.cs;
    if (concurrent) {
      // Three streams for the classes' launches, whatever their number -- this run's own and two more: as ONE resident batch the ten-class
      // workload takes 84..86 ms on 1, 2, 3 or 5 of them (the launches are persistent grids that share the chip by their sizes, not by their
      // queues); as a stream of 1024-block sub-batches 141 / 129 / 100 / 103 ms (consecutive sub-batches overlap through the queues); and a
      // stream costs a fresh process 10..14 ms to create -- the first submit of the ten-class file 0.12..0.14 s with five, 0.08..0.10 with two --
      // and is a hardware queue more for the device to schedule.  (RC_CLASS_STREAMS: their number, this run's included;
      // profiles/r06/class_streams.txt.)
      const size_t slot = pi % std::min<size_t>(plans.size(), static_cast<size_t>(c->classStreamCount));
      if (slot > 0) {
        while (c->classStreams.size() < slot) {
          hipStream_t ns = nullptr;
          HIP_TRY(hipStreamCreateWithPriority(&ns, hipStreamNonBlocking, c->classPrio));
          c->classStreams.push_back(ns);
        }
        st = c->classStreams[slot - 1];
      }
      if (pi == 0) trace("null: streams", b);
      if (st != R.cs) HIP_TRY(hipStreamWaitEvent(st, fork, 0));   // the memsets / the stop marks
    }
    NullArgs a{};
    a.blob = R.blob; a.dblocks = R.dblocks;
    a.classBlocks = R.classList + b->classOff[p.NK];
    a.nClassBlocks = static_cast<int>(members.size());
    a.flags = b->dflags.as<uint32_t>();
    a.gLo = gLo; a.gHi = gHi; a.sampleN = R.sampleN; a.Spad = R.Spad;
    a.U = c->d_U; a.pair = R.tp.pair; a.scratch = b->dscratch.as<uint32_t>() + scratchOff; a.scratchStride = p.stride;
    a.maxima = b->maxPtr; a.clampCount = reinterpret_cast<unsigned long long *>(b->dcounters.as<uint32_t>() + kCntClamp);
    a.tieThr = c->tieThr; a.debugSkip = c->debugSkip; a.comboSplit = p.comboSplit;
    a.cellStats = c->d_cellStats.as<unsigned long long>();
    unsigned int *work = b->dcounters.as<unsigned int>() + kCntWork + static_cast<size_t>(phase) * 3 * kClassSlots * 8;
    {
      const double cost = p.itemCost * static_cast<double>(members.size()) * groupsHere;
      if (cost > b->nullKernelCost) {
        b->nullKernelCost = cost;
        if (p.NK == kGenericClass) b->nullKernel = "rc::k_generic_dp";
        else if (p.NK > kGenericClass) b->nullKernel = "rc::k_tiled_dp<" + std::to_string(p.tiledKT) + ", false>";
        else {
          const bool ex = b->allExact || p.NK >= kFastRows;
          const bool dual = !ex && p.ldsBytes && p.ldsBytes <= 48 * 1024 && p.NK <= kDualRowsMaxNK && (p.NK == 2 || R.fat);
          if (!ex && !p.ldsBytes && p.hiOcc == 2) b->nullKernel = "rc::k_null<" + std::to_string(p.NK) + ", false, false, true, 0>";
          else if (!ex && !p.ldsBytes && p.hiOcc) b->nullKernel = "rc::k_null_occ<" + std::to_string(p.NK) + ">";
          else if (p.rowParts > 1) b->nullKernel = "rc::k_null<" + std::to_string(p.NK) + ", false, false, false, 1>";
          else b->nullKernel = "rc::k_null<" + std::to_string(p.NK) + (!ex && p.ldsBytes ? ", true" : ", false") + (ex ? ", true" : ", false") + (dual ? ", true, 0>" : ", false, 0>");
        }
      }
    }
    if (p.NK >= kGenericClass) {
      const bool tiled = p.NK > kGenericClass;
      a.skipMask = extraSkip;
      a.tiledKT = p.tiledKT; a.plainLds = static_cast<int>(p.ldsDp);
      a.workCounter = work + p.NK * 8;
      uint8_t *base = reinterpret_cast<uint8_t *>(a.scratch);
      const size_t codesAllBytes = (static_cast<size_t>(p.roundBlocks) * groupsHere * p.codesBytes + 255) & ~static_cast<size_t>(255);
      a.codesAll = base; a.codesStride = p.codesBytes;
      a.scratchStride = p.stateBytes;   // bytes for these kernels
      unsigned int *simWork = work + kClassSlots * 8 + p.NK * 8;   // (the class's unused "exact" queue counters)
      unsigned int *nanWork = work + 2 * kClassSlots * 8 + p.NK * 8;   // (... and its unused "split simulation" ones: the tiled classes' launch for blocks with NaN tables)
      const int *list = a.classBlocks;
      for (size_t at = 0; at < members.size(); at += static_cast<size_t>(p.roundBlocks)) {
        if (at) {   // the queues of the round before are spent
          HIP_TRY(hipMemsetAsync(a.workCounter, 0, 8 * sizeof(unsigned int), st));
          HIP_TRY(hipMemsetAsync(simWork, 0, 8 * sizeof(unsigned int), st));
          if (tiled) HIP_TRY(hipMemsetAsync(nanWork, 0, 8 * sizeof(unsigned int), st));
        }
        NullArgs r = a;
        r.classBlocks = list + at;
        r.nClassBlocks = static_cast<int>(std::min<size_t>(static_cast<size_t>(p.roundBlocks), members.size() - at));
        NullArgs sim = r;
        sim.workCounter = simWork;
        launch_generic_sim(sim, p.simGrid, p.ldsBytes, st, tiled);
        if (tiled) {
          // blocks with NaN score tables (flagged on the device by k_prep_lut, none as a rule) are left to a second launch of the instantiation with
          // the reference's NaN-order-dependent maxima: a few workgroups that look at every item's flag
          // (gap parameters outside the fast kernels' range -- Delta >= 0: the maximum with Delta counts -- : that instantiation for every block)
          r.skipMask = extraSkip | (b->allExact ? 0u : kFlagNan);
          if (!launch_tiled_dp(p.tiledKT, b->allExact, r, p.grid, p.ldsDp, base + codesAllBytes, st)) return fail(RC_ERR_UNSUPPORTED, "no tiled kernel for this tile size");
          b->nl[2] += 2;
          if (!b->allExact) {
            NullArgs e = r;
            e.skipMask = extraSkip; e.onlyMask = kFlagNan; e.workCounter = nanWork;
            e.nBlocksPtr = b->dcounters.as<int>() + kCntExact + p.NK;   // the class's blocks flagged by k_prep_lut (every NaN block is one)
            (void)launch_tiled_dp(p.tiledKT, true, e, std::min(p.grid, c->numCU), p.ldsDp, base + codesAllBytes, st);
            b->nl[2]++;
          }
        } else {
          launch_generic_dp(r, p.grid, p.ldsBytes, base + codesAllBytes, st);
          b->nl[2] += 2;
        }
      }
    } else if (b->allExact || p.NK >= kFastRows) {
      a.skipMask = extraSkip;
      a.workCounter = work + kClassSlots * 8 + p.NK * 8;
      if (!launch_null(p.NK, a, p.grid, 0, true, st)) return fail(RC_ERR_UNSUPPORTED, "no null kernel for this number of rows");
      b->nl[2]++;
    } else {
      a.skipMask = kFlagExact | extraSkip;
      a.workCounter = work + p.NK * 8;
      a.dual = (p.ldsBytes && p.NK <= kDualRowsMaxNK && (p.NK == 2 || R.fat)) ? 1 : 0;
      a.hiOcc = p.hiOcc; a.plainLds = static_cast<int>(p.plainLds); a.cacheSites = p.cacheSites;
      if (p.rowParts > 1 && b->dsbuf.p && (sbufOff + p.sbufFloats) * sizeof(float) <= b->dsbuf.cap) {
        a.rowParts = p.rowParts; a.sbuf = b->dsbuf.as<float>() + sbufOff; a.sbufStride = p.sbufStride;
        a.simParts = p.simParts; a.simGrid = p.rowSimGrid; a.simCounter = work + 2 * kClassSlots * 8 + p.NK * 8;
        sbufOff += p.sbufFloats;
      } else if (p.rowParts > 1) return fail(RC_ERR_ARG, "internal: the row buffer was not sized for this round");
      // tail sharing: one slot per workgroup of this launch in the zeroed dsteal area (see batch_run_async)
      if (c->tailSharing && !p.comboSplit && static_cast<size_t>(p.grid) <= steal_slots(c) && b->dsteal.p) {
        const size_t idx = static_cast<size_t>(phase) * b->classes.size() + pi;
        if ((idx + 1) * steal_words(c) * sizeof(uint32_t) <= b->dsteal.cap) a.steal = b->dsteal.as<unsigned int>() + idx * steal_words(c);
        a.stealWait = R.streaming ? 0 : 1;
      }
      if (!launch_null(p.NK, a, p.grid, p.ldsBytes, false, st)) return fail(RC_ERR_UNSUPPORTED, "no null kernel for this number of rows");
      b->nl[2]++;
      if (a.rowParts > 1) launch_null_rowscan(a, static_cast<int>(members.size()) * groupsHere, st);
      NullArgs e = a;
      e.classBlocks = b->dexact.as<int>() + b->classOff[p.NK];
      e.nBlocksPtr = b->dcounters.as<int>() + kCntExact + p.NK;
      e.nClassBlocks = 0;
      e.skipMask = extraSkip;
      e.comboSplit = 0;
      e.steal = nullptr;
      e.dual = 0; e.hiOcc = 0; e.plainLds = 0; e.cacheSites = 0; e.rowParts = 0; e.sbuf = nullptr; e.simParts = 0; e.simGrid = 0; e.simCounter = nullptr;
      e.workCounter = work + kClassSlots * 8 + p.NK * 8;
      if (!launch_null(p.NK, e, p.gridExact, 0, true, st)) return fail(RC_ERR_UNSUPPORTED, "no null kernel for this number of rows");
    }
    if (pi == 0) trace("null: first class", b);
    if (concurrent) {
      scratchOff += p.need;
      if (st != R.cs) {   // (this run's stream waits for the others when every class is queued: a wait in between would hold its own next class back)
        Event &done = *b->classDone[static_cast<size_t>(phase) * plans.size() + pi];
        HIP_TRY(done.record(st));
        joins.push_back(&done);
      }
    }
  }
  for (Event *done : joins) HIP_TRY(hipStreamWaitEvent(R.cs, *done, 0));
  return RC_OK;
}

