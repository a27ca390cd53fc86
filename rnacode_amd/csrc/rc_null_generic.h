// rc_null_generic.h -- the null-distribution loop body and the native DP for blocks with more rows than the register-resident
// kernels take (N > 64; the reference accepts up to MAX_NUM_NAMES = 500 rows, rnaz_utils.h:7: 100-way genome alignments).
//
// Same decomposition as k_null (lane = null sample, 64 samples of one block per wavefront, work items from per-XCD queues) and
// the same arithmetic in the same order, but nothing is unrolled over the sequences: the node states of the simulation and the
// sigma codes live in a per-workgroup scratch in global memory ([...][lane], coalesced), tables are read from the blob.  A DP row
// is cut into tiles of kGenericTile end codons; within a tile the loop runs sequence by sequence -- the three states of the
// current sequence in registers (parked in the scratch between tiles), its 64-entry sigma table spread over the lanes and read
// with ds_bpermute like in k_null -- and adds each sequence's max3 to the tile's sums, which stay in registers, in the reference's
// order k = 0, 1, ...; the HSS scan then walks the tile.
// Tried on top of this and dropped (tools/time_wide.py, 1024 blocks of 100 x 300: 1.60 s): unguarded steps with the row's last
// tile computed in full (35 % more steps at this length: 1.68 s), groups of eight steps skipped past the row end (2.1 s: the
// branches cost more than the steps), the next sequence's loads issued during the current one's steps (1.7-2.2 s: 13 more live
// registers, and the pass is issue-bound, not latency-bound), tiles of 16 or 24 end codons, 3 or 5 wavefronts per SIMD.
// Round 4, measured and taken back: the thirteen operands of a sequence's pass (sigma table entry, eight code words,
// frame-shift word, three parked states) fetched one sequence AHEAD with the look-ups in two halves of sixteen -- 256 blocks of
// 100 x 300 at n = 1000: 412 -> 628 ms at four wavefronts per SIMD (51 spilled registers), 565 at three, 580 at two.  The counters say
// the pass is neither issue-bound (0.33 VALU instructions per ns and SIMD, C3: 0.71) nor short of bandwidth (3.2 TB/s); what it waits
// for is the round trip of the parked states and code words at four wavefronts per SIMD, and the registers that would hide it are
// the ones the tile's 32 sums occupy.  The form that would fix it keeps the states of <= 16 sequences resident over a whole row
// and carries the per-cell k-sum between chunks of sequences (ascending k, as the reference adds): another kernel.
// It is the slow, complete path for rare inputs; the division is the plain '/'.
#pragma once
#include "rc_null_kernel.h"

namespace rc {

// scratch layout of one workgroup (bytes), for a class whose largest block has N rows, L reference residues, nn tree nodes
struct GenericLayout {
  size_t offCodes;   // u32 [6][NK][nW][64]    sigma codes per strand x frame and sequence, four consecutive codons per word (byte j & 3)
  size_t offState;   // f32 [3][NK][64]        DP states of every sequence at the last tile boundary
  size_t total;
  int nW;            // code words per (strand x frame, sequence): a tile reads eight words from any codon on, so 31 codons of padding
  __host__ __device__ GenericLayout(int N, int L, int nn) {
    const size_t NK = static_cast<size_t>(N) - 1;
    nW = ((L / 3 + 31) >> 2) + 1;
    offCodes = 0;
    offState = (offCodes + 6 * NK * static_cast<size_t>(nW) * kWave * sizeof(uint32_t) + 255) & ~static_cast<size_t>(255);
    (void)nn;
    total = (offState + 3 * NK * kWave * sizeof(float) + 255) & ~static_cast<size_t>(255);
  }
};

constexpr int kGenericTile = 32;

// LDS of one workgroup: the node states of the site being simulated, 2 bits each, 16 nodes per word [ceil(nn/16)][64], then the
// last three states of every row, 6 bits in a byte [N][64]
__host__ __device__ inline size_t generic_lds_bytes(int N, int nn) {
  return (static_cast<size_t>((nn + 15) >> 4) * sizeof(uint32_t) + static_cast<size_t>(N)) * kWave;
}

__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_null_generic(NullArgs A, const uint8_t *__restrict__ blob, const DevBlock *__restrict__ dblocks,
                                                     const int *__restrict__ classBlocks, const uint32_t *__restrict__ flags,
                                                     const uint32_t *__restrict__ Ustream, const uint8_t *__restrict__ pairTab,
                                                     uint8_t *__restrict__ scratch, float *__restrict__ maxima) {
  extern __shared__ uint32_t ldsGeneric[];
  const int lane = threadIdx.x;
  uint8_t *__restrict__ mine = scratch + static_cast<size_t>(blockIdx.x) * A.scratchStride;   // scratchStride in bytes here
  unsigned int xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  const int home = static_cast<int>(xcc & 7u);
  const int nBlocks = A.nClassBlocks;
  const int parts = A.comboSplit ? 6 : 1;   // few items: one strand x frame part each (phase A recomputed, maxima meet in an atomic max), as in k_null
  for (int qi = 0; qi < 8;) {
    const int q = (home + qi) & 7;
    const int g0 = A.gLo + ((q - A.gLo) & 7);
    const int gq = g0 < A.gHi ? ((A.gHi - 1 - g0) >> 3) + 1 : 0;
    const int perBlock = gq * parts;
    const int nq = nBlocks * perBlock;
    int t = nq;
    if (nq > 0) {
      if (lane == 0) t = static_cast<int>(atomicAdd(A.workCounter + q, 1u));
      t = __builtin_amdgcn_readfirstlane(t);
    }
    if (t >= nq) { qi++; continue; }
    const int bi = classBlocks[t / perBlock];
    if (A.skipMask && (flags[bi] & A.skipMask)) continue;
    const int sub = t % perBlock;
    const int grp = g0 + 8 * (sub / parts);
    const int onlyCombo = A.comboSplit ? sub % parts : -1;
    const DevBlock *__restrict__ db = dblocks + bi;
    const int N = db->N, NK = db->NK, cols = db->cols, L = db->L, nnodes = db->nnodes, L1 = L + 1;
    const int NCW = (NK + 4) / 5, ZW = db->zw_words;
    const float Delta = db->Delta, Omega = db->Omega, omega = db->omega, nkf = db->nkf;
    const NodeRec *__restrict__ nodes = reinterpret_cast<const NodeRec *>(blob + db->off_nodes);
    const uint16_t *__restrict__ qtip = reinterpret_cast<const uint16_t *>(blob + db->off_qtip16);
    const uint16_t *__restrict__ refcol = reinterpret_cast<const uint16_t *>(blob + db->off_refcol);
    const unsigned long long *__restrict__ zw = reinterpret_cast<const unsigned long long *>(blob + db->off_zw);
    const uint32_t *__restrict__ maskw = reinterpret_cast<const uint32_t *>(blob + db->off_mask);
    const float *__restrict__ lut = reinterpret_cast<const float *>(blob + db->off_lut);
    const GenericLayout lay(N, L, nnodes);
    uint8_t *__restrict__ codes = mine + lay.offCodes;
    uint32_t *nodeSt = ldsGeneric + lane;                                                                     // word w: [w * 64]
    uint8_t *win = reinterpret_cast<uint8_t *>(ldsGeneric + static_cast<size_t>((nnodes + 15) >> 4) * kWave) + lane;   // row r: [r * 64]
    float *__restrict__ dp = reinterpret_cast<float *>(mine + lay.offState);
    const int sidx = grp * kWave + lane;

    // ---- simulate the L reference-residue columns (evolve.c:416-433) and emit sigma codes for both strands
    unsigned int clamped = 0;
    for (int r = 0; r < N; r++) win[r * kWave] = 0;
    for (int i = 1; i <= L; i++) {
      const size_t ubase = static_cast<size_t>(refcol[i]) * A.Spad + sidx;
      const size_t qstride = static_cast<size_t>(cols) * A.Spad;
      uint32_t cur = 0;   // the word of node states being filled (nodes come parents first)
      for (int q0 = 0; q0 < nnodes; q0 += 8) {
        uint32_t uu[8];   // the draws of eight nodes in flight: their latency, not the arithmetic, is the cost of a node
#pragma unroll
        for (int t = 0; t < 8; t++) uu[t] = Ustream[ubase + static_cast<size_t>(q0 + t < nnodes ? q0 + t : nnodes - 1) * qstride];
#pragma unroll
        for (int t = 0; t < 8; t++) {
          const int qn = q0 + t;
          if (qn < nnodes) {
            const NodeRec &nr = nodes[qn];
            const int par = nr.parent;
            uint32_t ps = 0u;
            if (qn != 0) {
              const uint32_t pw = (par >> 4) == (qn >> 4) ? cur : nodeSt[(par >> 4) * kWave];
              ps = (pw >> (2 * (par & 15))) & 3u;
            }
            // the four threshold rows are wave-uniform (scalar loads); the parent's state picks one per lane
            const U4 t0 = nr.thr[0], t1 = nr.thr[1], t2 = nr.thr[2], t3 = nr.thr[3];
            const bool p1 = ps == 1u, p2 = ps == 2u, p3 = ps == 3u;
            const uint32_t thx = p3 ? t3.x : p2 ? t2.x : p1 ? t1.x : t0.x, thy = p3 ? t3.y : p2 ? t2.y : p1 ? t1.y : t0.y,
                           thz = p3 ? t3.z : p2 ? t2.z : p1 ? t1.z : t0.z, thw = p3 ? t3.w : p2 ? t2.w : p1 ? t1.w : t0.w;
            const uint32_t u = uu[t];
            const uint32_t st = (u > thx) + (u > thy) + (u > thz) + ((nr.basepack >> (2 * ps)) & 3u);
            clamped += (u > thw);
            cur |= (st & 3u) << (2 * (qn & 15));
            if ((qn & 15) == 15 || qn == nnodes - 1) { nodeSt[(qn >> 4) * kWave] = cur; cur = 0u; }
          }
        }
      }
      for (int r = 0; r < N; r++) {
        const int tip = qtip[r];
        const uint32_t stt = (nodeSt[(tip >> 4) * kWave] >> (2 * (tip & 15))) & 3u;
        win[r * kWave] = static_cast<uint8_t>(((static_cast<uint32_t>(win[r * kWave]) << 2) | stt) & 63u);
      }
      if (i >= 3) {
        const int ip = L + 3 - i;  // reverse-strand position whose codon is these three columns
        const uint32_t aF = win[0];
        const uint32_t aR = codon_flip(aF) ^ 63u;
        for (int k = 0; k < NK; k++) {
          const uint32_t mF = (maskw[(0 * L1 + i) * NCW + k / 5] >> (6 * (k % 5))) & 63u;
          const uint32_t mR = (maskw[(1 * L1 + ip) * NCW + k / 5] >> (6 * (k % 5))) & 63u;
          const uint32_t b = win[(k + 1) * kWave];
          const uint32_t bF = b & mF, bR = (codon_flip(b) ^ 63u) & mR;
          // column i closes codon j = i/3 - 1 of frame i % 3 (i = 3j + 3 + f)
          codes[(((static_cast<size_t>(0 + i % 3) * NK + k) * lay.nW + ((i / 3 - 1) >> 2)) * kWave + lane) * 4 + ((i / 3 - 1) & 3)] = pairTab[aF * 64 + bF];
          codes[(((static_cast<size_t>(3 + ip % 3) * NK + k) * lay.nW + ((ip / 3 - 1) >> 2)) * kWave + lane) * 4 + ((ip / 3 - 1) & 3)] = pairTab[aR * 64 + bR];
        }
      }
    }
    if (clamped && sidx < A.sampleN && onlyCombo <= 0) atomicAdd(A.clampCount, static_cast<unsigned long long>(clamped));   // (the six parts of a split item redo the same simulation)

    // ---- both strands x 3 frames: DP over (start a, end j) + serial HSS scan (score.c:496-535, 830-845, 888-962)
    float best = -1.0f;
    const float negTie = -A.tieThr;
    for (int s = 0; s < 2; s++)
      for (int f = 0; f < 3; f++) {
        const int sites = (L - f) / 3;
        if (sites <= 0) continue;
        if (onlyCombo >= 0 && onlyCombo != s * 3 + f) continue;
        SampleScan st{0.0f, 0.0f, 0u, 0u};
        for (int a = 0; a < sites; a++) {
          sample_scan_row_begin(st, best, static_cast<uint32_t>(a));
          const int jend = (a == sites - 1) ? sites - 1 : sites;   // the frame's final entry is never reported (score.c:892-900)
          // tiles start at multiples of four codons (the code words hold four), so a row's first tile may begin before a
          for (int j0 = a & ~3; j0 < jend; j0 += kGenericTile) {
            float sum[kGenericTile];
#pragma unroll
            for (int u = 0; u < kGenericTile; u++) sum[u] = 0.0f;
            const bool first = j0 <= a, more = j0 + kGenericTile < jend;
            const int lo = first ? a - j0 : 0, hi = jend - j0 < kGenericTile ? jend - j0 : kGenericTile;
            const uint32_t valid = (hi == kGenericTile ? 0xffffffffu : (1u << hi) - 1u) & ~((1u << lo) - 1u);   // bit u: a <= j0 + u < jend
            if (valid == 0u) break;   // the frame's last row: its only cell is the unreported final entry
            const bool full = valid == 0xffffffffu;
            const int jl = (j0 + (lane & (kGenericTile - 1)) < jend) ? j0 + (lane & (kGenericTile - 1)) : jend - 1;   // this lane's end codon for the z fetch
            for (int k = 0; k < NK; k++) {
              const int lutv = __float_as_int(lut[(static_cast<size_t>(s) * NK + k) * kLutSize + lane]);
              const uint32_t *__restrict__ ck = reinterpret_cast<const uint32_t *>(codes) + ((static_cast<size_t>(s * 3 + f) * NK + k) * lay.nW + (j0 >> 2)) * kWave + lane;
              // z of sequence k at the tile's end codons (score.c:506-533): lane u fetches the word of codon j0 + u, two ballots
              // turn the 2-bit values into wave-uniform masks -- almost always both zero (a sequence has few frame shifts)
              const unsigned long long zword = zw[(static_cast<size_t>(s * L1 + 3 + f) + static_cast<size_t>(3 * jl)) * ZW + (k >> 5)];
              const uint32_t zcl = static_cast<uint32_t>(zword >> (2 * (k & 31))) & 3u;
              float *__restrict__ dk = dp + static_cast<size_t>(k) * kWave + lane;
              float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f;
              if (!first) { s0 = dk[0]; s1 = dk[static_cast<size_t>(NK) * kWave]; s2 = dk[static_cast<size_t>(2 * NK) * kWave]; }
              uint32_t cw[kGenericTile / 4];
#pragma unroll
              for (int w = 0; w < kGenericTile / 4; w++) cw[w] = ck[w * kWave];
              float sig[kGenericTile];
#pragma unroll
              for (int u = 0; u < kGenericTile; u++)
                sig[u] = __int_as_float(__builtin_amdgcn_ds_bpermute(static_cast<int>(((cw[u >> 2] >> (8 * (u & 3))) & 0xffu) << 2), lutv));
              const uint32_t z0 = static_cast<uint32_t>(__ballot(zcl & 1u)) & valid, z1 = static_cast<uint32_t>(__ballot(zcl & 2u)) & valid;
              // FULL: all kGenericTile end codons exist (no guards); CLEAN: no frame shift of this sequence among them
              auto steps = [&](auto fullTag, auto cleanTag) {
                constexpr bool FULL = decltype(fullTag)::value, CLEAN = decltype(cleanTag)::value;
#pragma unroll
                for (int u = 0; u < kGenericTile; u++) {
                  if (FULL || ((valid >> u) & 1u)) {
                    if constexpr (CLEAN) { s0 = s0 + sig[u]; s1 = s1 + omega; s2 = s2 + omega; }
                    else pair_step(static_cast<int>(((z0 >> u) & 1u) | (((z1 >> u) & 1u) << 1)), sig[u], Delta, Omega, omega, s0, s1, s2);
                    const float m = fmaxf(fmaxf(s0, s1), s2);
                    sum[u] = (k == 0) ? m : sum[u] + m;
                  }
                }
              };
              if ((z0 | z1) == 0u) {
                if (full) steps(std::true_type{}, std::true_type{}); else steps(std::false_type{}, std::true_type{});
              } else {
                steps(std::false_type{}, std::false_type{});
              }
              if (more) { dk[0] = s0; dk[static_cast<size_t>(NK) * kWave] = s1; dk[static_cast<size_t>(2 * NK) * kWave] = s2; }
            }
#pragma unroll
            for (int u = 0; u < kGenericTile; u++) {
              if ((valid >> u) & 1u) {   // a <= j0 + u < jend
                const float v = fmaxf(sum[u], Delta) / nkf;
                sample_scan_step(st, v, static_cast<float>(2 * (j0 + u) + 1), negTie);
              }
            }
          }
          sample_scan_row_end(st, static_cast<uint32_t>(a));
        }
        sample_scan_last(st, best);
      }
    if (sidx < A.sampleN) {
      float *dst = maxima + static_cast<size_t>(db->out_index) * A.sampleN + sidx;
      if (A.comboSplit) atomicMax(reinterpret_cast<int *>(dst), __float_as_int(best));   // positive floats or the -1 sentinel: integer order == float order
      else *dst = best;
    }
  }
}

// native block, any number of rows: one wavefront per (block, strand x frame), lane = start codon, the states of the lane's row
// in a scratch [3][NK][64] per workgroup (score.c:441-556, 811-848); like k_native_dp, 64 rows of S at a time go through a buffer
// [64][sites] behind the states and are scanned (getHSS) before the next 64
__global__ __launch_bounds__(64) void k_native_dp_generic(NativeArgs A, float *__restrict__ scratch, size_t scratchStride) {
  const int lane = threadIdx.x;
  const int bi = A.blocks[blockIdx.x / 6];
  const int combo = blockIdx.x % 6, s = combo / 3, f = combo % 3;
  const DevBlock *__restrict__ db = A.dblocks + bi;
  const int L = db->L, L1 = L + 1, NK = db->NK, ZW = db->zw_words;
  const float Delta = db->Delta, Omega = db->Omega, omega = db->omega, nkf = db->nkf;
  const unsigned long long *zw = reinterpret_cast<const unsigned long long *>(A.blob + db->off_zw);
  const float *sigma = reinterpret_cast<const float *>(A.blob + db->off_sigma);
  const int sites = (L - f) / 3, smax = L / 3;
  float *dp = scratch + static_cast<size_t>(blockIdx.x) * scratchStride;
  float *tile = dp + static_cast<size_t>(3) * NK * kWave;
  float *full = A.fullS ? A.fullS + static_cast<size_t>(combo) * smax * smax : nullptr;
  DevHss *out = A.fullS ? nullptr : A.hss + (static_cast<size_t>(bi) * 6 + combo) * A.hssCap;
  int n = 0;
  ScanState st{0.0f, -1, -1};
  for (int a0 = 0; a0 < sites; a0 += kWave) {
    const int a = a0 + lane;
    for (int k = 0; k < 3 * NK; k++) dp[static_cast<size_t>(k) * kWave + lane] = 0.0f;
    for (int j = a0; j < sites; j++) {
      const int i = 3 * j + 3 + f;
      const unsigned long long *z = zw + static_cast<size_t>(s * L1 + i) * ZW;
      if (a <= j && a < sites) {
        float sum = 0.0f;
        for (int k = 0; k < NK; k++) {
          float s0 = dp[(0 * static_cast<size_t>(NK) + k) * kWave + lane], s1 = dp[(1 * static_cast<size_t>(NK) + k) * kWave + lane],
                s2 = dp[(2 * static_cast<size_t>(NK) + k) * kWave + lane];
          pair_step(static_cast<int>((z[k >> 5] >> (2 * (k & 31))) & 3ull), sigma[(static_cast<size_t>(s) * NK + k) * L1 + i], Delta, Omega, omega, s0, s1, s2);
          dp[(0 * static_cast<size_t>(NK) + k) * kWave + lane] = s0;
          dp[(1 * static_cast<size_t>(NK) + k) * kWave + lane] = s1;
          dp[(2 * static_cast<size_t>(NK) + k) * kWave + lane] = s2;
          sum = sum + fmaxf(fmaxf(s0, s1), s2);
        }
        const float v = fmaxf(sum, Delta) / nkf;
        tile[static_cast<size_t>(lane) * sites + j] = v;
        if (full) full[static_cast<size_t>(a) * sites + j] = v;
      }
    }
    __syncthreads();   // the rows written by the other lanes
    native_scan_rows(tile, a0, (a0 + kWave < sites) ? a0 + kWave : sites, sites, s, f, A.tieThr, st, n, out, A.hssCap, lane);
    __syncthreads();
  }
  if (out && lane == 0) A.hssCount[static_cast<size_t>(bi) * 6 + combo] = n;
}

}  // namespace rc
