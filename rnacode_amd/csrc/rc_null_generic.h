// rc_null_generic.h -- the null-distribution loop body and the native DP for blocks with more rows than the register-resident
// kernels take (N > 64; the reference accepts up to MAX_NUM_NAMES = 500 rows, rnaz_utils.h:7: 100-way genome alignments).
//
// Same decomposition as k_null (lane = null sample, 64 samples of one block per wavefront, work items from per-XCD queues) and
// the same arithmetic in the same order, but nothing is unrolled over the sequences: the per-sequence DP states, the node states
// of the simulation and the sigma codes live in a per-workgroup scratch in global memory ([...][lane], coalesced), tables are
// read from the blob.  It is the slow, complete path for rare inputs -- correctness over speed; the division is the plain '/'.
#pragma once
#include "rc_null_kernel.h"

namespace rc {

// scratch layout of one workgroup (bytes), for a class whose largest block has N rows, L reference residues, nn tree nodes
struct GenericLayout {
  size_t offCodes;   // u8  [2][L+1][NK][64]   sigma codes of both strands
  size_t offNode;    // u8  [nn][64]           node states of the site being simulated
  size_t offWin;     // u8  [N][64]            last three states of every row, 6 bits
  size_t offState;   // f32 [3][NK][64]        DP states
  size_t total;
  __host__ __device__ GenericLayout(int N, int L, int nn) {
    const size_t NK = static_cast<size_t>(N) - 1, L1 = static_cast<size_t>(L) + 1;
    offCodes = 0;
    offNode = offCodes + 2 * L1 * NK * kWave;
    offWin = offNode + static_cast<size_t>(nn) * kWave;
    offState = (offWin + static_cast<size_t>(N) * kWave + 255) & ~static_cast<size_t>(255);
    total = (offState + 3 * NK * kWave * sizeof(float) + 255) & ~static_cast<size_t>(255);
  }
};

__global__ __launch_bounds__(64) void k_null_generic(NullArgs A, const uint8_t *__restrict__ blob, const DevBlock *__restrict__ dblocks,
                                                     const int *__restrict__ classBlocks, const uint32_t *__restrict__ flags,
                                                     const uint32_t *__restrict__ Ustream, const uint8_t *__restrict__ pairTab,
                                                     uint8_t *__restrict__ scratch, float *__restrict__ maxima) {
  const int lane = threadIdx.x;
  uint8_t *__restrict__ mine = scratch + static_cast<size_t>(blockIdx.x) * A.scratchStride;   // scratchStride in bytes here
  unsigned int xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  const int home = static_cast<int>(xcc & 7u);
  const int nBlocks = A.nClassBlocks;
  for (int qi = 0; qi < 8;) {
    const int q = (home + qi) & 7;
    const int g0 = A.gLo + ((q - A.gLo) & 7);
    const int gq = g0 < A.gHi ? ((A.gHi - 1 - g0) >> 3) + 1 : 0;
    const int nq = nBlocks * gq;
    int t = nq;
    if (nq > 0) {
      if (lane == 0) t = static_cast<int>(atomicAdd(A.workCounter + q, 1u));
      t = __builtin_amdgcn_readfirstlane(t);
    }
    if (t >= nq) { qi++; continue; }
    const int bi = classBlocks[t / gq];
    if (A.skipMask && (flags[bi] & A.skipMask)) continue;
    const int grp = g0 + 8 * (t % gq);
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
    uint8_t *__restrict__ nodeSt = mine + lay.offNode;
    uint8_t *__restrict__ win = mine + lay.offWin;
    float *__restrict__ dp = reinterpret_cast<float *>(mine + lay.offState);
    const int sidx = grp * kWave + lane;

    // ---- simulate the L reference-residue columns (evolve.c:416-433) and emit sigma codes for both strands
    unsigned int clamped = 0;
    for (int r = 0; r < N; r++) win[r * kWave + lane] = 0;
    for (int i = 1; i <= L; i++) {
      const size_t ubase = static_cast<size_t>(refcol[i]) * A.Spad + sidx;
      const size_t qstride = static_cast<size_t>(cols) * A.Spad;
      for (int qn = 0; qn < nnodes; qn++) {
        const NodeRec &nr = nodes[qn];
        const uint32_t ps = qn == 0 ? 0u : nodeSt[static_cast<size_t>(nr.parent) * kWave + lane];
        const U4 th = nr.thr[ps];
        const uint32_t u = Ustream[ubase + qn * qstride];
        uint32_t st = (u > th.x) + (u > th.y) + (u > th.z) + ((nr.basepack >> (2 * ps)) & 3u);
        clamped += (u > th.w);
        nodeSt[static_cast<size_t>(qn) * kWave + lane] = static_cast<uint8_t>(st);
      }
      for (int r = 0; r < N; r++) {
        const uint32_t w = ((static_cast<uint32_t>(win[r * kWave + lane]) << 2) | nodeSt[static_cast<size_t>(qtip[r]) * kWave + lane]) & 63u;
        win[r * kWave + lane] = static_cast<uint8_t>(w);
      }
      if (i >= 3) {
        const int ip = L + 3 - i;  // reverse-strand position whose codon is these three columns
        const uint32_t aF = win[lane];
        const uint32_t aR = codon_flip(aF) ^ 63u;
        for (int k = 0; k < NK; k++) {
          const uint32_t mF = (maskw[(0 * L1 + i) * NCW + k / 5] >> (6 * (k % 5))) & 63u;
          const uint32_t mR = (maskw[(1 * L1 + ip) * NCW + k / 5] >> (6 * (k % 5))) & 63u;
          const uint32_t b = win[(k + 1) * kWave + lane];
          const uint32_t bF = b & mF, bR = (codon_flip(b) ^ 63u) & mR;
          codes[((0 * static_cast<size_t>(L1) + i) * NK + k) * kWave + lane] = pairTab[aF * 64 + bF];
          codes[((1 * static_cast<size_t>(L1) + ip) * NK + k) * kWave + lane] = pairTab[aR * 64 + bR];
        }
      }
    }
    if (clamped && sidx < A.sampleN) atomicAdd(A.clampCount, static_cast<unsigned long long>(clamped));

    // ---- both strands x 3 frames: DP over (start a, end j) + serial HSS scan (score.c:496-535, 830-845, 888-962)
    float best = -1.0f;
    const float negTie = -A.tieThr;
    for (int s = 0; s < 2; s++)
      for (int f = 0; f < 3; f++) {
        const int sites = (L - f) / 3;
        if (sites <= 0) continue;
        SampleScan st{0.0f, 0.0f, 0u, 0u};
        for (int a = 0; a < sites; a++) {
          sample_scan_row_begin(st, best, static_cast<uint32_t>(a));
          for (int k = 0; k < 3 * NK; k++) dp[static_cast<size_t>(k) * kWave + lane] = 0.0f;
          const int jend = (a == sites - 1) ? sites - 1 : sites;   // the frame's final entry is never reported (score.c:892-900)
          for (int j = a; j < jend; j++) {
            const int i = 3 * j + 3 + f;
            const unsigned long long *z = zw + static_cast<size_t>(s * L1 + i) * ZW;
            float sum = 0.0f;
            for (int k = 0; k < NK; k++) {
              float s0 = dp[(0 * static_cast<size_t>(NK) + k) * kWave + lane], s1 = dp[(1 * static_cast<size_t>(NK) + k) * kWave + lane],
                    s2 = dp[(2 * static_cast<size_t>(NK) + k) * kWave + lane];
              const uint32_t code = codes[((static_cast<size_t>(s) * L1 + i) * NK + k) * kWave + lane];
              const float sig = lut[(static_cast<size_t>(s) * NK + k) * kLutSize + code];
              pair_step(static_cast<int>((z[k >> 5] >> (2 * (k & 31))) & 3ull), sig, Delta, Omega, omega, s0, s1, s2);
              dp[(0 * static_cast<size_t>(NK) + k) * kWave + lane] = s0;
              dp[(1 * static_cast<size_t>(NK) + k) * kWave + lane] = s1;
              dp[(2 * static_cast<size_t>(NK) + k) * kWave + lane] = s2;
              const float m = fmaxf(fmaxf(s0, s1), s2);
              sum = (k == 0) ? m : sum + m;
            }
            const float v = fmaxf(sum, Delta) / nkf;
            sample_scan_step(st, v, static_cast<float>(2 * j + 1), negTie);
          }
          sample_scan_row_end(st, static_cast<uint32_t>(a));
        }
        sample_scan_last(st, best);
      }
    if (sidx < A.sampleN) maxima[static_cast<size_t>(db->out_index) * A.sampleN + sidx] = best;
  }
}

// native block, any number of rows: one wavefront per (block, strand x frame), lane = start codon, the states of the lane's row
// in a scratch [3][NK][64] per workgroup (score.c:441-556, 811-848)
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
  float *S = A.S + A.offS[bi] + static_cast<size_t>(combo) * smax * smax;
  float *dp = scratch + static_cast<size_t>(blockIdx.x) * scratchStride;
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
        S[static_cast<size_t>(a) * sites + j] = fmaxf(sum, Delta) / nkf;
      }
    }
  }
}

}  // namespace rc
