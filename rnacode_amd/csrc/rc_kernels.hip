// rc_kernels.hip -- hand-written HIP kernels for gfx950 (MI355X) of the RNAcode scoring path.
//
//   k_mt_stream       MT19937 output streams, one wavefront per seed          (seqgen/twister.c:73-152)
//   k_prep_models     calculateBG's two 4096-term accumulations per (block, strand, row)  (score.c:107-193)
//   k_prep_lut        sigma value of every codon-pair code per (block, strand, k)         (score.c:406-425)
//   k_prep_gaps       reverse-complement rows, codon windows and z of every reference position    (rnaz_utils.c:316-348, misc.c:186-289)
//   k_null<NK>        simulate -> codon-pair codes -> 3-state DP -> HSS scan -> per-sample maximum,
//                     one LANE per null sample, 64 samples of one block per wavefront
//                                                                             (score.c:976-1064 loop body)
//   k_native_sigma    calculateSigma for the native block                     (score.c:375-426)
//   k_native_dp<NK>   pairwise + multiple score matrix of the native block, one lane per start codon
//                                                                             (score.c:441-556, 811-848)
//                     (getHSS, score.c:864-974, runs inside k_native_dp: native_scan_rows in rc_null_kernel.h)
//   k_sk_row          one row b of Sk for backtracking                        (score.c:496-535)
//   k_evd_fit         Gumbel maximum-likelihood fit per block                 (extreme_fit.c:157-251)
//
// Numerics: every add/max of the DP is a single IEEE binary32 operation in the reference's
// order; compile with -ffp-contract=off; float division is hipcc's correctly rounded default.
#include "rc_null_kernel.h"
#include "rc_null_generic.h"
#include "rc_refexp.h"

namespace rc {

// ------------------------------------------------------------------------------------------
// MT19937 stream: U[d * Spad + s] = d-th output of init_genrand(seedBase + s)
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_mt_stream(uint32_t seedBase, int Spad, int D, uint32_t *__restrict__ U) {
  __shared__ uint32_t mt[624];
  const int lane = threadIdx.x;
  const int s = blockIdx.x;
  if (lane == 0) {  // init_genrand: a serial recurrence (twister.c:73-89)
    uint32_t x = seedBase + static_cast<uint32_t>(s);
    mt[0] = x;
    for (int i = 1; i < 624; i++) {
      x = 1812433253u * (x ^ (x >> 30)) + static_cast<uint32_t>(i);
      mt[i] = x;
    }
  }
  __syncthreads();
  for (int base = 0; base < D; base += 624) {
    // regenerate the 624 words (twister.c:127-141).  Ascending chunks of 64 are safe in place:
    // word kk needs old kk, old kk+1 and (kk+397) mod 624, which is old for kk < 227 and already
    // new (written >= 164 words earlier) otherwise; word 623 needs the new word 0.
    for (int c = 0; c < 624; c += 64) {
      const int kk = c + lane;
      uint32_t nv = 0;
      if (kk < 624) {
        const int k1 = (kk + 1 == 624) ? 0 : kk + 1;
        const int km = (kk + 397 >= 624) ? kk + 397 - 624 : kk + 397;
        const uint32_t y = (mt[kk] & 0x80000000u) | (mt[k1] & 0x7fffffffu);
        nv = mt[km] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
      }
      __syncthreads();
      if (kk < 624) mt[kk] = nv;
      __syncthreads();
    }
    for (int c = 0; c < 624; c += 64) {  // tempering (twister.c:145-151)
      const int kk = c + lane;
      if (kk < 624 && base + kk < D) {
        uint32_t y = mt[kk];
        y ^= (y >> 11);
        y ^= (y << 7) & 0x9d2c5680u;
        y ^= (y << 15) & 0xefc60000u;
        y ^= (y >> 18);
        U[static_cast<size_t>(base + kk) * Spad + s] = y;
      }
    }
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------------
// block-constant tables derived on the device
// ------------------------------------------------------------------------------------------

// calculateBG (score.c:107-193) for one model per lane and Hamming class per wavefront: two 4096-term float accumulations in the loop
// order a1,a2,a3,b1,b2,b3, every product and sum a single binary32 operation in the reference's order
// (SURVEY appendix A.4).  The codon a is a run-time loop (its three HKY85 rows, 12 floats, are re-read per a);
// the 64 codons b are unrolled, so row entries, stop tests and Hamming classes of b are compile-time and the
// kernel needs no LDS -- it runs beside k_null, whose workgroups hold all of a CU's LDS.
__device__ constexpr int kPepOf(int c) {   // genetic code as in rc_tables.h: amino-acid index in "ARNDCQEGHILKMFPSTWYV", -1 = stop
  constexpr char code[] = "KNKNTTTTRSRSIIMIQHQHPPPPRRRRLLLLEDEDAAAAGGGGVVVV*Y*YSSSS*CWCLFLF";
  constexpr char order[] = "ARNDCQEGHILKMFPSTWYV";
  for (int i = 0; i < 20; i++) if (order[i] == code[c]) return i;
  return -1;
}

template <int B, typename F> __device__ __forceinline__ void for_codon_b(F &&f) {
  if constexpr (B < 64) { f(std::integral_constant<int, B>{}); for_codon_b<B + 1>(static_cast<F &&>(f)); }
}

__global__ __launch_bounds__(64) void k_prep_models(PrepArgs A) {
  __builtin_amdgcn_s_setprio(3);   // a short latency-bound kernel beside k_null: the SIMD issues its instructions first
  // Four wavefronts per 64 models, one per Hamming class: a wavefront is a serial instruction stream (one VALU instruction per ~9 cycles
  // when it has a SIMD to itself, fewer beside k_null), and this kernel's latency is what the first k_null of a stream waits for.  The
  // class of a term is wave-uniform (a is the loop counter, b is unrolled), so a wavefront skips the terms of the other classes with a
  // scalar branch, division included; its two accumulators see exactly the terms they saw when all four classes shared one lane.
  const int lane = threadIdx.x;
  const int hq = blockIdx.x & 3;
  int t = (blockIdx.x >> 2) * kWave + lane;
  const bool active = t < A.nModels;
  if (!active) t = A.nModels - 1;
  int lo = 0, hi = A.nBlocks;                       // block whose [prefix, next prefix) holds t
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (A.modelPrefix[mid] <= t) lo = mid; else hi = mid;
  }
  const DevBlock *db = A.dblocks + A.blocks[lo];
  ModelRec *m = reinterpret_cast<ModelRec *>(A.blob + db->off_models) + (t - A.modelPrefix[lo]);
  const float *__restrict__ P = m->P;
  const float f0 = m->freqs[0], f1 = m->freqs[1], f2 = m->freqs[2], f3 = m->freqs[3];
  const int *__restrict__ blosum = A.blosum;
  auto freq = [&](int x) { return x == 0 ? f0 : x == 1 ? f1 : x == 2 ? f2 : f3; };   // x is wave-uniform
  float probStop = 0.0f;
  for (int a = 0; a < 64; a++) {
    const int a1 = a >> 4, a2 = (a >> 2) & 3, a3 = a & 3;
    const bool stopA = A.pep[a] == -1;
    const float fa = freq(a1) * freq(a2) * freq(a3);
    float r1[4], r2[4], r3[4];
#pragma unroll
    for (int x = 0; x < 4; x++) { r1[x] = P[a1 * 4 + x]; r2[x] = P[a2 * 4 + x]; r3[x] = P[a3 * 4 + x]; }
    for_codon_b<0>([&](auto bc) {
      constexpr int b = decltype(bc)::value;
      constexpr bool stopB = kPepOf(b) == -1;
      if (stopA || stopB) {
        float pAB = r1[b >> 4] * r2[(b >> 2) & 3] * r3[b & 3];
        pAB = pAB * fa;
        probStop = probStop + pAB;
      }
    });
  }
  float cq = 0.0f, sq = 0.0f;
  const float keep = 1.0f - probStop;   // (1 - probStop): int 1 converted to float
  for (int a = 0; a < 64; a++) {
    const int pa = A.pep[a];
    if (pa == -1) continue;
    const int a1 = a >> 4, a2 = (a >> 2) & 3, a3 = a & 3;
    const float fa = freq(a1) * freq(a2) * freq(a3);
    float r1[4], r2[4], r3[4];
#pragma unroll
    for (int x = 0; x < 4; x++) { r1[x] = P[a1 * 4 + x]; r2[x] = P[a2 * 4 + x]; r3[x] = P[a3 * 4 + x]; }
    for_codon_b<0>([&](auto bc) {
      constexpr int b = decltype(bc)::value;
      constexpr int pb = kPepOf(b);
      if constexpr (pb != -1) {
        const int x = a ^ b;
        const int h = ((x & 48) != 0) + ((x & 12) != 0) + ((x & 3) != 0);
        if (h == hq) {   // wave-uniform: a scalar branch
          float pAB = r1[b >> 4] * r2[(b >> 2) & 3] * r3[b & 3];
          pAB = pAB * fa;
          pAB = pAB / keep;
          const float sc = static_cast<float>(blosum[pa * 20 + pb]);
          cq = cq + pAB;
          sq = sq + sc * pAB;
        }
      }
    });
  }
  if (active) {
    m->scores[hq] = sq / cq;
    m->probs[hq] = cq;
  }
}

// The same for a FEW models (a caller that scores block by block; the first sub-batch of a stream): k_prep_models gives a model to a
// lane, and its 4096-term chain is then the latency of the whole preparation (0.25 ms, a quarter of a one-block batch's run).  Here a
// WAVEFRONT takes one (model, Hamming class): the terms of its accumulation are made 64 at a time, one per lane, in the order of the
// reference's loops (the lists below), and only the additions run in sequence, lane by lane through v_readlane -- the same operands into
// the same chains in the same order, a tenth of the latency at twenty times the instructions (hence only below 2048 models).
struct PrepLists { uint16_t ab[4096]; int off[6]; };   // (a << 6 | b): pairs with a stop codon, then the pairs of Hamming class 0, 1, 2, 3; a outer, b inner
constexpr PrepLists make_prep_lists() {
  PrepLists l{};
  int n = 0;
  l.off[0] = 0;
  for (int a = 0; a < 64; a++) for (int b = 0; b < 64; b++) if (kPepOf(a) == -1 || kPepOf(b) == -1) l.ab[n++] = static_cast<uint16_t>(a << 6 | b);
  for (int h = 0; h < 4; h++) {
    l.off[1 + h] = n;
    for (int a = 0; a < 64; a++) for (int b = 0; b < 64; b++) {
      if (kPepOf(a) == -1 || kPepOf(b) == -1) continue;
      const int x = a ^ b;
      if (((x & 48) != 0) + ((x & 12) != 0) + ((x & 3) != 0) == h) l.ab[n++] = static_cast<uint16_t>(a << 6 | b);
    }
  }
  l.off[5] = n;
  return l;
}
__device__ const PrepLists kPrepLists = make_prep_lists();

__global__ __launch_bounds__(64) void k_prep_models_few(PrepArgs A) {
  __builtin_amdgcn_s_setprio(3);
  const int lane = threadIdx.x;
  const int hq = blockIdx.x & 3, t = blockIdx.x >> 2;   // the grid is 4 x nModels
  int lo = 0, hi = A.nBlocks;
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (A.modelPrefix[mid] <= t) lo = mid; else hi = mid;
  }
  const DevBlock *db = A.dblocks + A.blocks[lo];
  ModelRec *m = reinterpret_cast<ModelRec *>(A.blob + db->off_models) + (t - A.modelPrefix[lo]);
  const float *__restrict__ P = m->P;
  const float f0 = m->freqs[0], f1 = m->freqs[1], f2 = m->freqs[2], f3 = m->freqs[3];
  auto freq = [&](int x) { return x == 0 ? f0 : x == 1 ? f1 : x == 2 ? f2 : f3; };
  auto term = [&](int ab) {   // P(a -> b) x f(a), the products in calculateBG's order
    const int a = ab >> 6, b = ab & 63;
    const int a1 = a >> 4, a2 = (a >> 2) & 3, a3 = a & 3;
    const float fa = freq(a1) * freq(a2) * freq(a3);
    float pAB = P[a1 * 4 + (b >> 4)] * P[a2 * 4 + ((b >> 2) & 3)] * P[a3 * 4 + (b & 3)];
    pAB = pAB * fa;
    return pAB;
  };
  auto add_in_order = [&](float &acc, float v, int cnt) {   // acc += v of lane 0, 1, ..., cnt - 1
    for (int l = 0; l < cnt; l++) acc = acc + __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l));
  };
  float probStop = 0.0f;
  for (int base = kPrepLists.off[0]; base < kPrepLists.off[1]; base += kWave) {
    const int cnt = kPrepLists.off[1] - base < kWave ? kPrepLists.off[1] - base : kWave;
    const float v = lane < cnt ? term(kPrepLists.ab[base + lane]) : 0.0f;
    add_in_order(probStop, v, cnt);
  }
  const float keep = 1.0f - probStop;
  float cq = 0.0f, sq = 0.0f;
  for (int base = kPrepLists.off[1 + hq]; base < kPrepLists.off[2 + hq]; base += kWave) {
    const int cnt = kPrepLists.off[2 + hq] - base < kWave ? kPrepLists.off[2 + hq] - base : kWave;
    float pAB = 0.0f, sp = 0.0f;
    if (lane < cnt) {
      const int ab = kPrepLists.ab[base + lane];
      pAB = term(ab);
      pAB = pAB / keep;
      const float sc = static_cast<float>(A.blosum[A.pep[ab >> 6] * 20 + A.pep[ab & 63]]);
      sp = sc * pAB;
    }
    for (int l = 0; l < cnt; l++) {
      cq = cq + __int_as_float(__builtin_amdgcn_readlane(__float_as_int(pAB), l));
      sq = sq + __int_as_float(__builtin_amdgcn_readlane(__float_as_int(sp), l));
    }
  }
  if (lane == 0) {
    m->scores[hq] = sq / cq;
    m->probs[hq] = cq;
  }
}

// same test as rc_host.cpp in_div_range: 0, or 2^-76 <= |v| < 2^40
__device__ __forceinline__ bool in_div_range_dev(float v) {
  const float a = fabsf(v);
  return a == 0.0f || (a >= 1.3234889800848443e-23f && a < 1099511627776.0f);
}

// sigma per code: (float)matrix[pepA][pepB] - scores[h] (score.c:422-425), 0, or a stop penalty.  A block
// with a value outside the range proven for the constant-divisor division is flagged and appended to its
// class's list for the EXACT launch.
__global__ __launch_bounds__(64) void k_prep_lut(PrepArgs A) {   // one wavefront per block, no LDS (it runs beside k_null)
  __builtin_amdgcn_s_setprio(3);   // a short latency-bound kernel beside k_null: the SIMD issues its instructions first
  const int bi = A.blocks[blockIdx.x];
  const DevBlock *db = A.dblocks + bi;
  const int N = db->N, NK = db->NK;
  const ModelRec *models = reinterpret_cast<const ModelRec *>(A.blob + db->off_models);
  float *lut = reinterpret_cast<float *>(A.blob + db->off_lut);
  const int total = 2 * NK * kLutSize;
  bool mine = false, nan = false;
  for (int idx = threadIdx.x; idx < total; idx += kWave) {
    const int code = idx % kLutSize, k = (idx / kLutSize) % NK, s = idx / (kLutSize * NK);
    const CodeInfo ci = A.codeInfo[code];
    float v = 0.0f;
    if (ci.kind == 1) v = static_cast<float>(ci.b) - models[s * N + k + 1].scores[ci.h];
    else if (ci.kind == 3) v = db->stop0;
    else if (ci.kind == 4) v = db->stopk;
    lut[idx] = v;
    mine |= !in_div_range_dev(v);
    nan |= v != v;
  }
  const int cls = block_class(N, db->L, db->omega, A.rule);
  if (cls == kGenericClass && threadIdx.x == 0) {   // k_generic_dp's running sums of omega, added the way the DP adds them
    float *W = reinterpret_cast<float *>(A.blob + db->off_chain);
    const int nW = db->L / 3 + 40;
    const float omega = db->omega;
    float c = 0.0f;
    for (int i = 0; i < nW; i++) {
      if (i > 3) c = c + omega;
      W[i] = c;
    }
  }
  const bool bad = __ballot(mine) != 0ull, anyNan = __ballot(nan) != 0ull;
  if (threadIdx.x == 0 && bad) {
    A.flags[bi] |= kFlagExact | (anyNan ? kFlagNan : 0u);
    const int at = atomicAdd(A.exactCount + cls, 1);
    A.exactList[A.classOff[cls] + at] = bi;
  }
}


// The gap pattern of a block, both strands, from its uploaded characters: what getBlock / pos2col (misc.c:186-269) derive per
// reference position i >= 3 and sequence k, reduced to what the DP needs -- z (misc.c:230-244) and which of the codon's three
// columns hold '-' in row k -- plus the reverse-complement rows themselves (rnaz_utils.c:316-348) for k_native_sigma.
// One wavefront per (block, strand), no LDS (it runs beside k_null).  Strand 1 is read through the mirrored column index: a
// reverse complement keeps '-' where it was.  The window of position i is the columns (col(i-3), col(i)] (from column 0 for
// i == 3): exactly three reference residues, so the reference row's own gap count there is the window's length minus 3.
__global__ __launch_bounds__(64) void k_prep_gaps(PrepArgs A) {
  __builtin_amdgcn_s_setprio(3);   // a short latency-bound kernel beside k_null: the SIMD issues its instructions first
  const int bi = A.blocks[blockIdx.x >> 1];
  const int s = blockIdx.x & 1;
  const int lane = threadIdx.x;
  const DevBlock *__restrict__ db = A.dblocks + bi;
  const int N = db->N, NK = db->NK, cols = db->cols, L = db->L, L1 = L + 1;
  const int zww = db->zw_words, ncw = (NK + 4) / 5, zmw = db->zm_words;
  const uint8_t *__restrict__ fwd = A.blob + db->off_chars;
  if (s == 1) {   // T,U -> A  A -> T  C <-> G, everything else unchanged; columns reversed
    uint8_t *rev = A.blob + db->off_chars_rev;
    for (int r = 0; r < N; r++)
      for (int c = lane; c < cols; c += kWave) {
        const uint8_t ch = fwd[static_cast<size_t>(r) * cols + c];
        const uint8_t o = (ch == 'T' || ch == 'U') ? 'A' : ch == 'A' ? 'T' : ch == 'C' ? 'G' : ch == 'G' ? 'C' : ch;
        rev[static_cast<size_t>(r) * cols + (cols - 1 - c)] = o;
      }
  }
  auto is_gap = [&](int r, int c) { return fwd[static_cast<size_t>(r) * cols + (s ? cols - 1 - c : c)] == '-'; };
  uint16_t *refcol = reinterpret_cast<uint16_t *>(A.blob + db->off_refcol) + static_cast<size_t>(s) * L1;
  unsigned long long *zw = reinterpret_cast<unsigned long long *>(A.blob + db->off_zw) + static_cast<size_t>(s) * L1 * zww;
  uint32_t *maskw = reinterpret_cast<uint32_t *>(A.blob + db->off_mask) + static_cast<size_t>(s) * L1 * ncw;
  unsigned long long *zm = reinterpret_cast<unsigned long long *>(A.blob + db->off_zmask) + static_cast<size_t>(s) * 3 * zmw;
  // refcol[p] = column of the p-th reference residue (1-based): an ordered compaction of row 0's non-gap columns
  if (lane == 0) refcol[0] = 0;
  int seen = 0;
  for (int c0 = 0; c0 < cols; c0 += kWave) {
    const int c = c0 + lane;
    const bool res = c < cols && !is_gap(0, c);
    const unsigned long long m = __ballot(res);
    const int p = seen + __builtin_popcountll(m & ((1ull << lane) - 1ull)) + 1;
    if (res && p <= L) refcol[p] = static_cast<uint16_t>(c);
    seen += __builtin_popcountll(m);
  }
  // positions 0..2 have no codon ending there: all-zero entries
  if (lane < 3 && lane <= L) {
    for (int w = 0; w < zww; w++) zw[static_cast<size_t>(lane) * zww + w] = 0ull;
    for (int w = 0; w < ncw; w++) maskw[static_cast<size_t>(lane) * ncw + w] = 0u;
  }
  __syncthreads();   // one wavefront: makes refcol, written by other lanes, visible
  for (int f = 0; f < 3; f++) {
    const int sites = (L - f) / 3;
    for (int j0 = 0; j0 < zmw * kWave; j0 += kWave) {
      const int j = j0 + lane;
      const bool active = j < sites;
      bool any = false;
      if (active) {
        const int i = 3 * j + 3 + f;
        const int c1 = refcol[i - 2], c2 = refcol[i - 1], c3 = refcol[i];
        const int lo = (i > 3) ? refcol[i - 3] + 1 : 0, hi = c3 + 1;
        const int g0 = (hi - lo) - 3;
        unsigned long long zword = 0ull;
        uint32_t mword = 0u;
        for (int k = 0; k < NK; k++) {
          int gk = 0;
          for (int c = lo; c < hi; c++) gk += is_gap(k + 1, c) ? 1 : 0;
          int diff = gk - g0;
          diff = (diff < 0 ? -diff : diff) % 3;                 // misc.c:230-244: 0 -> z = 0, 1 -> z = +1, 2 -> z = -1
          any |= diff != 0;
          zword |= static_cast<unsigned long long>(diff) << (2 * (k & 31));
          const uint32_t m = (is_gap(k + 1, c1) ? 0u : 0x30u) | (is_gap(k + 1, c2) ? 0u : 0x0Cu) | (is_gap(k + 1, c3) ? 0u : 0x03u);
          mword |= m << (6 * (k % 5));
          if ((k & 31) == 31 || k == NK - 1) { zw[static_cast<size_t>(i) * zww + (k >> 5)] = zword; zword = 0ull; }
          if (k % 5 == 4 || k == NK - 1) { maskw[static_cast<size_t>(i) * ncw + k / 5] = mword; mword = 0u; }
        }
      }
      const unsigned long long bits = __ballot(active && any);
      if (lane == 0) zm[static_cast<size_t>(f) * zmw + (j0 >> 6)] = bits;
    }
  }
}

void launch_prep(const PrepArgs &a, hipStream_t stream) {
  if (a.nBlocks <= 0) return;
  hipLaunchKernelGGL(k_prep_gaps, dim3(2 * a.nBlocks), dim3(kWave), 0, stream, a);
  if (a.nModels <= 2048) hipLaunchKernelGGL(k_prep_models_few, dim3(4 * a.nModels), dim3(kWave), 0, stream, a);
  else hipLaunchKernelGGL(k_prep_models, dim3(4 * ((a.nModels + kWave - 1) / kWave)), dim3(kWave), 0, stream, a);
  hipLaunchKernelGGL(k_prep_lut, dim3(a.nBlocks), dim3(kWave), 0, stream, a);
}

// ------------------------------------------------------------------------------------------
// native block
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_native_sigma(NativeArgs A) {
  __builtin_amdgcn_s_setprio(3);   // a short latency-bound kernel beside k_null: the SIMD issues its instructions first
  const int bi = A.blocks[blockIdx.x];
  const DevBlock *__restrict__ db = A.dblocks + bi;
  const int NK = db->NK, cols = db->cols, L = db->L, L1 = L + 1;
  uint8_t *blob = A.blob;
  const uint16_t *refcol = reinterpret_cast<const uint16_t *>(blob + db->off_refcol);
  const uint8_t *chars = blob + db->off_chars, *charsRev = blob + db->off_chars_rev;
  const float *lut = reinterpret_cast<const float *>(blob + db->off_lut);
  float *sigma = reinterpret_cast<float *>(blob + db->off_sigma);
  const int total = 2 * NK * L1;
  for (int idx = threadIdx.x; idx < total; idx += blockDim.x) {
    const int i = idx % L1, k = (idx / L1) % NK, s = idx / (L1 * NK);
    float sg = 0.0f;
    if (i >= 3) {
      const uint8_t *r0 = (s ? charsRev : chars);
      const uint8_t *rk = (s ? charsRev : chars) + static_cast<size_t>(k + 1) * cols;
      const uint16_t *rc = refcol + s * L1;
      uint32_t a = 0, b = 0;
      bool anyN = false;
#pragma unroll
      for (int t = 2; t >= 0; t--) {
        const int c = rc[i - t];
        const uint8_t ca = r0[c], cb = rk[c];
        anyN |= (ca == 'N') | (cb == 'N');            // score.c:400-404
        const uint32_t xa = (ca == 'C') ? 1u : (ca == 'G') ? 2u : (ca == 'T' || ca == 'U') ? 3u : 0u;
        const uint32_t xb = (cb == 'C') ? 1u : (cb == 'G') ? 2u : (cb == 'T' || cb == 'U') ? 3u : 0u;
        a = (a << 2) | xa;
        b = (b << 2) | xb;
      }
      if (!anyN) sg = lut[(s * NK + k) * kLutSize + A.pair[a * 64 + b]];
    }
    sigma[idx] = sg;
  }
}

// Native block, up to 64 rows: pairwise + multiple score matrix (score.c:441-556, 811-848) and getHSS in one pass.  A persistent
// grid of single-wavefront workgroups takes the (block, strand x frame) items in turn; lane = start codon, 64 rows of S at a time
// go through a per-workgroup buffer (global memory, L2-resident) and are scanned in the reference's order before the next 64:
// the matrices themselves are never materialised (they were 6 (L/3)^2 floats per block: 24 GB for 10^5 blocks of 300 columns).
template <int NK>
__global__ __launch_bounds__(64) void k_native_dp(NativeArgs A) {
  __builtin_amdgcn_s_setprio(3);   // a short latency-bound kernel beside k_null: the SIMD issues its instructions first
  const int lane = threadIdx.x;
  float *__restrict__ tile = A.tile + static_cast<size_t>(blockIdx.x) * A.tileStride;
  for (int item = blockIdx.x; item < A.nItems; item += gridDim.x) {
    const int bi = A.blocks[item / 6];
    const int combo = item % 6, s = combo / 3, f = combo % 3;
    const DevBlock *__restrict__ db = A.dblocks + bi;
    const int L = db->L, L1 = L + 1;
    const float Delta = db->Delta, Omega = db->Omega, omega = db->omega, nkf = db->nkf;
    const unsigned long long *zw = reinterpret_cast<const unsigned long long *>(A.blob + db->off_zw);
    const float *sigma = reinterpret_cast<const float *>(A.blob + db->off_sigma);
    const int sites = (L - f) / 3, smax = L / 3;
    float *full = A.fullS ? A.fullS + static_cast<size_t>(combo) * smax * smax : nullptr;
    // (17..32 other sequences: this fully unrolled kernel spills already; it stays exactly as it was -- scalar loads, scan inside)
    constexpr bool kOld = NK > 16 && NK <= 32;
    float *__restrict__ all = (!kOld && A.sAll) ? A.sAll + static_cast<size_t>(item) * A.sAllSites * A.sAllSites : nullptr;
    DevHss *out = A.fullS ? nullptr : A.hss + (static_cast<size_t>(bi) * 6 + combo) * A.hssCap;
    int n = 0;
    ScanState st{0.0f, -1, -1};
    const bool nanSem = A.flags && (A.flags[bi] & kFlagNan);   // NaN score tables: the reference's MAX macro, operand order and all (ref_max)
    for (int a0 = 0; a0 < sites; a0 += kWave) {
      const int a = a0 + lane;
      float s0[NK], s1[NK], s2[NK];
#pragma unroll
      for (int k = 0; k < NK; k++) s0[k] = s1[k] = s2[k] = 0.0f;
      // (17..32 other sequences: the lane-fetched operands are 2 (N-1) registers more than this fully unrolled kernel has -- 21 rows x 90
      // columns 2.5 -> 4.5 ms with them, spilled; those keep the scalar loads)
      if constexpr (NK <= 16 || NK > 32) {
      // sigma and z of 64 end codons at a time, one codon per lane (vector loads, all in flight together), handed to the cell loop
      // with v_readlane: wave-uniform scalar loads inside that loop were a round trip per end codon, and the kernel is nothing else
      constexpr int ZP = (NK + 15) / 16;   // z of 16 sequences, 2 bits each, per register
      for (int jc = a0; jc < sites; jc += kWave) {
        const int jl = jc + lane < sites ? jc + lane : sites - 1, il = 3 * jl + 3 + f;
        float sgl[NK];
        uint32_t zl[ZP];
#pragma unroll
        for (int k = 0; k < NK; k++) sgl[k] = sigma[(s * NK + k) * L1 + il];
#pragma unroll
        for (int x = 0; x < ZP; x++) zl[x] = reinterpret_cast<const uint32_t *>(zw)[(static_cast<size_t>(s * L1 + il) * ((NK + 31) / 32)) * 2 + x];
        const int jhi = jc + kWave < sites ? jc + kWave : sites;
        for (int j = jc; j < jhi; j++) {
          const int t = j - jc;
          float sg[NK];
          uint32_t z[ZP];
#pragma unroll
          for (int k = 0; k < NK; k++) sg[k] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(sgl[k]), t));
#pragma unroll
          for (int x = 0; x < ZP; x++) z[x] = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(zl[x]), t));
          if (a <= j && a < sites) {
            float sum = 0.0f;
            uint32_t zany = 0u;
#pragma unroll
            for (int x = 0; x < ZP; x++) zany |= z[x];
            if (__builtin_expect(nanSem, 0)) {
#pragma unroll
              for (int k = 0; k < NK; k++) {
                pair_step<true>(static_cast<int>((z[k >> 4] >> (2 * (k & 15))) & 3u), sg[k], Delta, Omega, omega, s0[k], s1[k], s2[k]);
                sum = sum + ref_max3<true>(s0[k], s1[k], s2[k]);
              }
            } else if (zany == 0u) {   // no sequence has a frame shift at this codon (almost every codon): straight-line code, no case per sequence
#pragma unroll
              for (int k = 0; k < NK; k++) {
                s0[k] = s0[k] + sg[k];
                s1[k] = s1[k] + omega;
                s2[k] = s2[k] + omega;
                sum = sum + fmaxf(fmaxf(s0[k], s1[k]), s2[k]);
              }
            } else {
              // Some sequence is out of frame here -- and stays so until its next gap: in real alignments these are long stretches.  No
              // case per sequence (a wavefront alone on its SIMD waits for the instruction fetch behind every taken branch, and the
              // compiler copied all 3 (N-1) states at every join): the three cases of score.c:506-533 computed side by side, the
              // wave-uniform z selects.  z = +-1: s_x' = max(s_x + Delta, s_y + Omega), y the state before x (z = +1) or behind it (-1).
#pragma unroll
              for (int k = 0; k < NK; k++) {
                const uint32_t zc = (z[k >> 4] >> (2 * (k & 15))) & 3u;
                const bool zero = zc == 0u, one = zc == 1u;
                const float d0 = s0[k] + Delta, d1 = s1[k] + Delta, d2 = s2[k] + Delta;
                const float o0 = s0[k] + Omega, o1 = s1[k] + Omega, o2 = s2[k] + Omega;
                const float n0 = fmaxf(d0, one ? o2 : o1), n1 = fmaxf(d1, one ? o0 : o2), n2 = fmaxf(d2, one ? o1 : o0);
                const float p0 = s0[k] + sg[k], p1 = s1[k] + omega, p2 = s2[k] + omega;
                s0[k] = zero ? p0 : n0;
                s1[k] = zero ? p1 : n1;
                s2[k] = zero ? p2 : n2;
                sum = sum + fmaxf(fmaxf(s0[k], s1[k]), s2[k]);
              }
            }
            const float v = fmaxf(sum, Delta) / nkf;
            if (all) all[static_cast<size_t>(a) * A.sAllSites + j] = v;
            else tile[static_cast<size_t>(lane) * sites + j] = v;
            if (full) full[static_cast<size_t>(a) * sites + j] = v;
          }
        }
      }
      } else {
      for (int j = a0; j < sites; j++) {
        const int i = 3 * j + 3 + f;
        constexpr int ZW = (NK + 31) / 32;
        unsigned long long z[ZW];
#pragma unroll
        for (int x = 0; x < ZW; x++) z[x] = zw[static_cast<size_t>(s * L1 + i) * ZW + x];
        if (a <= j && a < sites) {
          float sum = 0.0f;
#pragma unroll
          for (int k = 0; k < NK; k++) {
            const float sig = sigma[(s * NK + k) * L1 + i];
            if (__builtin_expect(nanSem, 0)) {
              pair_step<true>(static_cast<int>((z[k >> 5] >> (2 * (k & 31))) & 3ull), sig, Delta, Omega, omega, s0[k], s1[k], s2[k]);
              sum = sum + ref_max3<true>(s0[k], s1[k], s2[k]);
            } else {
              pair_step(static_cast<int>((z[k >> 5] >> (2 * (k & 31))) & 3ull), sig, Delta, Omega, omega, s0[k], s1[k], s2[k]);
              sum = sum + fmaxf(fmaxf(s0[k], s1[k]), s2[k]);
            }
          }
          const float v = fmaxf(sum, Delta) / nkf;
          tile[static_cast<size_t>(lane) * sites + j] = v;
          if (full) full[static_cast<size_t>(a) * sites + j] = v;
        }
      }
      }
      if constexpr (!kOld) { if (all) continue; }   // getHSS: k_native_scan
      __syncthreads();   // the rows written by the other lanes (one wavefront: a fence, no waiting for anybody)
      native_scan_rows(tile, a0, (a0 + kWave < sites) ? a0 + kWave : sites, sites, s, f, A.tieThr, st, n, out, A.hssCap, lane);
      __syncthreads();   // all read before the next 64 rows (or the next item) overwrite the buffer
    }
    if (out && lane == 0 && (kOld || !all)) A.hssCount[static_cast<size_t>(bi) * 6 + combo] = n;
  }
}

// getHSS (score.c:864-974) over the matrices k_native_dp left in A.sAll, one LANE per (block, strand x frame): the state machine is serial
// per matrix -- a wavefront per matrix (native_scan_rows) walked the positive entries one by one with 63 lanes looking on, and was two
// thirds of the native-block kernels' time at the headline shape (1.39 ms of which 0.94).  Here 64 matrices advance together; the
// lanes' matrices differ in size, a lane past its own end idles.
__global__ __launch_bounds__(64) void k_native_scan(NativeArgs A) {
  const int item = blockIdx.x * kWave + threadIdx.x;
  if (item >= A.nItems) return;
  const int bi = A.blocks[item / 6];
  const int combo = item % 6, s = combo / 3, f = combo % 3;
  const DevBlock *__restrict__ db = A.dblocks + bi;
  const int sites = (db->L - f) / 3;
  const float *__restrict__ S = A.sAll + static_cast<size_t>(item) * A.sAllSites * A.sAllSites;
  DevHss *out = A.hss + (static_cast<size_t>(bi) * 6 + combo) * A.hssCap;
  int n = 0;
  ScanState st{0.0f, -1, -1};
  for (int a = 0; a < sites; a++) {
    const float *__restrict__ row = S + static_cast<size_t>(a) * A.sAllSites;
    for (int j0 = a; j0 < sites; j0 += 8) {   // eight entries fetched at a time: the loads do not depend on the state machine
      float v8[8];
#pragma unroll
      for (int t = 0; t < 8; t++) v8[t] = row[j0 + t < sites ? j0 + t : sites - 1];
#pragma unroll
      for (int t = 0; t < 8; t++) {
        const int j = j0 + t;
        const float v = v8[t];
        const bool last = (a == sites - 1) && (j == sites - 1);
        if (j < sites && (v > 0.0f || last))
          scan_step(st, v, a, j, last, A.tieThr, [&](const ScanState &e) {
            if (n < A.hssCap) out[n] = DevHss{s, f, e.ss, e.se, e.cm};
            n++;
          });
      }
    }
  }
  A.hssCount[static_cast<size_t>(bi) * 6 + combo] = n;
}

// One row b of Sk for all k (lane = k-1): out[(k-1)*3 + x][t] for i = b+2+3t <= iMax.
__global__ __launch_bounds__(64) void k_sk_row(const uint8_t *blob, const DevBlock *dblocks, const uint32_t *flags, int bi, int s, int b, int iMax,
                                               float *out, int stride) {
  const DevBlock *db = dblocks + bi;
  const bool nanSem = flags && (flags[bi] & kFlagNan);
  const int L1 = db->L + 1;
  const unsigned long long *zw = reinterpret_cast<const unsigned long long *>(blob + db->off_zw);
  const float *sigma = reinterpret_cast<const float *>(blob + db->off_sigma);
  for (int k = threadIdx.x; k < db->NK; k += kWave) {
    float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f;
    int t = 0;
    for (int i = b + 2; i <= iMax; i += 3, t++) {
      const int zc = static_cast<int>((zw[static_cast<size_t>(s * L1 + i) * db->zw_words + (k >> 5)] >> (2 * (k & 31))) & 3ull);
      // lanes follow different k: the branch may diverge here, which is fine for this helper
      if (nanSem) pair_step<true>(zc, sigma[(static_cast<size_t>(s) * db->NK + k) * L1 + i], db->Delta, db->Omega, db->omega, s0, s1, s2);
      else pair_step(zc, sigma[(static_cast<size_t>(s) * db->NK + k) * L1 + i], db->Delta, db->Omega, db->omega, s0, s1, s2);
      out[(static_cast<size_t>(k) * 3 + 0) * stride + t] = s0;
      out[(static_cast<size_t>(k) * 3 + 1) * stride + t] = s1;
      out[(static_cast<size_t>(k) * 3 + 2) * stride + t] = s2;
    }
  }
}

// ------------------------------------------------------------------------------------------
// EVD fit (extreme_fit.c:157-251), one workgroup per block, sums reduced across 256 threads
// ------------------------------------------------------------------------------------------
// The per-block fit comes in two shapes.  THREADS = 64: ONE wavefront without LDS -- in a large batch it is queued behind k_null,
// whose workgroups hold all of a CU's LDS, beside the next batch's k_null, and must not wait for LDS.  THREADS = 256: four
// wavefronts that meet through 128 bytes of LDS -- for small batches, where nothing competes for LDS and the latency of the
// slowest block's fit (up to 300 evaluations of the four sums when Newton's method needs the bisection fallback) is the run time.
// Either way the sums end as identical bits on every thread (xor butterfly, then a fixed-order sum of the wavefronts' values), so
// the Newton / bisection control flow below is uniform.
// Sum over the wavefront, the same bits on every lane, without LDS round trips: inside a row of 16 lanes the partner values come by DPP
// (xor 1, xor 2, then the mirrored half row and the mirrored row -- after the first two steps a quad holds one value, so any lane of the
// other quad / half will do); the four row sums come back through v_readlane and are added in one fixed order.  (The fit evaluates four
// such sums up to 300 times per block: with ds_bpermute shuffles that was most of a small batch's fit time.)
template <int CTRL> __device__ __forceinline__ double dpp_add(double x) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), CTRL, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), CTRL, 0xf, 0xf, false);
  return x + __hiloint2double(hi, lo);
}
__device__ __forceinline__ double row_value(double x, int lane) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(x), lane), __builtin_amdgcn_readlane(__double2loint(x), lane));
}
__device__ __forceinline__ double wave_sum(double v) {
  v = dpp_add<0xB1>(v);    // quad_perm [1,0,3,2]
  v = dpp_add<0x4E>(v);    // quad_perm [2,3,0,1]
  v = dpp_add<0x141>(v);   // row_half_mirror
  v = dpp_add<0x140>(v);   // row_mirror
  return (row_value(v, 0) + row_value(v, 16)) + (row_value(v, 32) + row_value(v, 48));
}

template <int THREADS>
__device__ __forceinline__ void fit_sum4(double &a, double &b, double &c, double &d, double *sh) {
  a = wave_sum(a); b = wave_sum(b); c = wave_sum(c); d = wave_sum(d);
  if constexpr (THREADS > kWave) {
    constexpr int NW = THREADS / kWave;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) { double *w = sh + (threadIdx.x >> 6) * 4; w[0] = a; w[1] = b; w[2] = c; w[3] = d; }
    __syncthreads();
    a = b = c = d = 0.0;
#pragma unroll
    for (int w = 0; w < NW; w++) { a += sh[w * 4 + 0]; b += sh[w * 4 + 1]; c += sh[w * 4 + 2]; d += sh[w * 4 + 3]; }
  }
}

// exp as the reference's C library computes it (rc_refexp.h): the Newton iteration's stopping test is decided by its last bit
__device__ __forceinline__ double fit_exp(int mode, double a) { return mode == 2 ? ref_exp(a, true) : mode == 1 ? ref_exp(a, false) : exp(a); }

// The four sums of Lawless 4.1.6 in the REFERENCE'S order, i = 0 .. n-1 (extreme_fit.c:50-58): every wavefront goes over all n elements
// sixty-four at a time, lane l makes element base + l's terms, and the sums take them in one by one through v_readlane -- the same
// additions in the same order as the reference's loop, hence (with rc_refexp.h's exp) the same bits, on every lane of every wavefront, without
// LDS or barriers.  A sum over n terms in another order differs by ~n 2^-53 relative, which cannot matter to a fit -- except in the
// comparisons that steer it (|f| < 1e-5, the signs in the bisection), on maxima so degenerate that f is only known to its tolerance (the
// class of tests/data/soak_fit_*.json).  So the fast evaluation (lawless416: strided partial sums, a tree over lanes) is what runs, and
// this one decides wherever a difference in the last bits could: fits on few samples, a fit that has not converged after twelve Newton steps
// (one that converges takes < 10; the slow one is replayed from the start in this order), the bisection fallback, and any evaluation whose |f|
// lands within 1e-8 of the tolerance.
template <typename T>
__device__ void lawless416_serial(int mode, const T *x, int n, double lambda, double *f, double *df) {
  const int lane = threadIdx.x & (kWave - 1);
  double e = 0, xe = 0, xxe = 0, xs = 0;
  for (int base = 0; base < n; base += kWave) {
    const int i = base + lane;
    const double xi = i < n ? static_cast<double>(x[i]) : 0.0;
    const double ex = i < n ? fit_exp(mode, -1. * lambda * xi) : 0.0;
    const double t1 = xi * ex, t2 = xi * xi * ex;
    const int cnt = n - base < kWave ? n - base : kWave;
    if (cnt == kWave) {
#pragma unroll
      for (int l = 0; l < kWave; l++) { xs += row_value(xi, l); xe += row_value(t1, l); xxe += row_value(t2, l); e += row_value(ex, l); }
    } else {
      for (int l = 0; l < cnt; l++) { xs += row_value(xi, l); xe += row_value(t1, l); xxe += row_value(t2, l); e += row_value(ex, l); }
    }
  }
  const double total = static_cast<double>(n);
  *f = 1. / lambda - xs / total + xe / e;
  *df = ((xe / e) * (xe / e)) - (xxe / e) - (1. / (lambda * lambda));
}
// sum of exp(-lambda x_i) in the reference's order (extreme_fit.c:238-245)
template <typename T>
__device__ double esum_serial(int mode, const T *x, int n, double lambda) {
  const int lane = threadIdx.x & (kWave - 1);
  double e = 0;
  for (int base = 0; base < n; base += kWave) {
    const int i = base + lane;
    const double ex = i < n ? fit_exp(mode, -1 * lambda * static_cast<double>(x[i])) : 0.0;
    const int cnt = n - base < kWave ? n - base : kWave;
    for (int l = 0; l < cnt; l++) e += row_value(ex, l);
  }
  return e;
}

template <int THREADS, typename T>
__device__ void lawless416(int mode, const T *x, int n, double lambda, double *sh, double *f, double *df) {
  double e = 0, xe = 0, xxe = 0, xs = 0;
  for (int i = threadIdx.x; i < n; i += THREADS) {
    const double xi = static_cast<double>(x[i]);
    const double ex = fit_exp(mode, -1. * lambda * xi);
    xs += xi; xe += xi * ex; xxe += xi * xi * ex; e += ex;
  }
  fit_sum4<THREADS>(e, xe, xxe, xs, sh);
  const double total = static_cast<double>(n);
  *f = 1. / lambda - xs / total + xe / e;
  *df = ((xe / e) * (xe / e)) - (xxe / e) - (1. / (lambda * lambda));
}

constexpr int kFitSerialMaxN = 128;    // fits on at most this many samples: every evaluation in the reference's order
constexpr int kFitSerialFromStep = 12; // a fit on more samples that has not converged after this many Newton steps: again from the start, the same

template <int THREADS, typename T>
__device__ int evd_fit_block(int mode, const T *x, int n, double *sh, double *mu_out, double *lambda_out) {
  double lambda = 0.2, fx, dfx;
  const double tol = 1e-5;
  const bool small = n <= kFitSerialMaxN;
  // one evaluation: the fast sums, and the reference-ordered ones where the outcome of a comparison could hang on the order (see lawless416_serial)
  auto eval = [&](double lam, bool careful) {
    if (careful) { lawless416_serial<T>(mode, x, n, lam, &fx, &dfx); return; }
    lawless416<THREADS, T>(mode, x, n, lam, sh, &fx, &dfx);
    if (fabs(fabs(fx) - tol) < 1e-8) lawless416_serial<T>(mode, x, n, lam, &fx, &dfx);
  };
  // Newton-Raphson (extreme_fit.c:170-191).  A fit that converges takes fewer than ten steps and does not care about the last bits of its sums.  One
  // that has not converged after twelve is on a path only the reference's own rounding leads along (129 maxima of two values, 122 and 7 of them:
  // thirty-four steps through lambda = 130, where exp(-lambda x) is 1e-320 and the sums' order decides whether the next step is finite; a soak
  // found it in round 6, tests/data/soak_fit_two_values.json): it is started again from lambda = 0.2 with every evaluation in the reference's order.
  int it = 0;
  for (int replay = small ? 1 : 0; replay < 2; replay++) {
    const int steps = replay ? 100 : kFitSerialFromStep;
    lambda = 0.2;
    for (it = 0; it < steps; it++) {
      eval(lambda, replay != 0);
      if (fabs(fx) < tol) break;
      lambda = lambda - fx / dfx;
      if (lambda <= 0.) lambda = 0.001;
    }
    if (it < steps) break;
  }
  if (it == 100) {  // bisection fallback (extreme_fit.c:193-236)
    double left = 0.2, right = 0.2, mid = 0.2;
    lambda = 0.2;
    eval(lambda, true);
    if (fx < 0.) {
      do {
        left -= 0.1;
        if (left < 0.) return 0;
        eval(left, true);
      } while (fx < 0.);
    } else {
      do {
        right += 0.1;
        eval(right, true);
        if (right > 100.) return 0;
      } while (fx > 0.);
    }
    for (it = 0; it < 100; it++) {
      mid = (left + right) / 2.;
      eval(mid, true);
      if (fabs(fx) < tol) break;
      if (fx > 0.) left = mid; else right = mid;
    }
    if (it == 100) return 0;
    lambda = mid;
  }
  double e = 0, z0 = 0, z1 = 0, z2 = 0;
  if (small) e = esum_serial<T>(mode, x, n, lambda);
  else {
    for (int i = threadIdx.x; i < n; i += THREADS) e += fit_exp(mode, -1 * lambda * static_cast<double>(x[i]));
    fit_sum4<THREADS>(e, z0, z1, z2, sh);
  }
  *mu_out = -1. * log(e / static_cast<double>(n)) / lambda;
  *lambda_out = lambda;
  return 1;
}

// best native HSS = results[0].score after the sort, -1 if none (RNAcode.c:176-178)
__device__ __forceinline__ float best_native(const FitArgs &A, int bi) {
  float m = -1.0f;
  for (int c = 0; c < 6; c++) {
    const int n = min(A.hssCount[static_cast<size_t>(bi) * 6 + c], A.hssCap);
    const DevHss *h = A.hss + (static_cast<size_t>(bi) * 6 + c) * A.hssCap;
    for (int i = 0; i < n; i++) if (h[i].score > 0.0f) m = fmaxf(m, h[i].score);
  }
  return m;
}

// --stop-early (score.c:992,1036-1042): the reference leaves the sample loop as soon as more than
// (int)(cutoff * sampleN) samples beat the best native score.  That count only grows, so a block is decided
// once the samples simulated so far exceed it; the remaining sample groups are then not run for it.
__global__ __launch_bounds__(64) void k_stop_mark(FitArgs A) {
  __builtin_amdgcn_s_setprio(3);   // a short latency-bound kernel beside k_null: the SIMD issues its instructions first
  const int bi = A.blocks[blockIdx.x];
  const DevBlock *db = A.dblocks + bi;
  const float *x = A.maxima + static_cast<size_t>(db->out_index) * A.sampleN;
  const float maxNative = best_native(A, bi);     // every lane: a few records
  int better = 0;
  for (int i = threadIdx.x; i < A.firstN; i += kWave) better += (x[i] > maxNative) ? 1 : 0;
  for (int off = 32; off > 0; off >>= 1) better += __shfl_down(better, off, 64);
  // written either way: the bit of an earlier run of this batch must not survive (a block decided late in that run would otherwise
  // be left out from the second round on and be fitted on too few samples)
  if (threadIdx.x == 0) A.flags[bi] = (A.flags[bi] & ~kFlagStopped) | (better > A.stopCutoff ? kFlagStopped : 0u);
}

template <int THREADS>
__global__ __launch_bounds__(THREADS) void k_evd_fit(FitArgs A) {
  __builtin_amdgcn_s_setprio(3);   // a short latency-bound kernel beside k_null: the SIMD issues its instructions first
  __shared__ double sh[THREADS > kWave ? (THREADS / kWave) * 4 : 1];   // not referenced (and not allocated) for THREADS = 64
  const int bi = A.blocks[blockIdx.x];
  const DevBlock *db = A.dblocks + bi;
  const float *x = A.maxima + static_cast<size_t>(db->out_index) * A.sampleN;
  const float maxNative = best_native(A, bi);     // every thread: a few records
  double better = 0, scored = 0, z1 = 0, z2 = 0;
  for (int i = threadIdx.x; i < A.sampleN; i += THREADS) {
    better += (x[i] > maxNative) ? 1.0 : 0.0;   // samples not simulated hold -1
    scored += (x[i] != -1.0f) ? 1.0 : 0.0;
  }
  fit_sum4<THREADS>(better, scored, z1, z2, sh);
  FitOut o;
  o.better = static_cast<int>(better);
  o.rc = -1; o.mu = 0; o.lambda = 0;
  const bool stopped = A.stopEarly && (o.better > A.stopCutoff);  // score.c:1040-1042
  // No sample has a high-scoring segment (every maximum is the -1 of score.c:1028: a block of one or two codons).  EVDMaxLikelyFit then
  // evaluates Lawless 4.1.6 some 1100 times to no avail -- f(lambda) = (1/lambda + 1) - 1 > 0 for every lambda it tries: Newton doubles
  // lambda until exp overflows and the sums turn NaN (100 steps), the bracketing walks right from 0.2 to 100 in steps of 0.1 and
  // gives up (extreme_fit.c:193-216) -- and returns 0.  Same verdict here without the walk: it was 1.2 of the 4 ms that
  // genomic-preprocessed.maf (one such block among 33) takes.
  const bool nothingScored = scored == 0.0;
  if (!stopped && !nothingScored) {
    double mu, lambda;
    if (evd_fit_block<THREADS, float>(A.expMode, x, A.sampleN, sh, &mu, &lambda) == 1) { o.rc = 1; o.mu = mu; o.lambda = lambda; }
  }
  if (threadIdx.x == 0) A.out[db->out_index] = o;
}

// EVDMaxLikelyFit on caller-supplied doubles (rc_evd_fit): the same fit instantiated for binary64 inputs.
__global__ __launch_bounds__(256) void k_evd_fit_f64(const double *x, int n, FitOut *out, int mode) {
  __shared__ double sh[(256 / kWave) * 4];
  double mu = 0, lambda = 0;
  const int rc = evd_fit_block<256, double>(mode, x, n, sh, &mu, &lambda);
  if (threadIdx.x == 0) { out->rc = rc; out->mu = rc ? mu : 0; out->lambda = lambda; out->better = 0; }
}

// getHSS's fold (SampleScan, rc_null_kernel.h) over the S values a ROWS launch of k_null left in its buffers: one wavefront per
// (block, 64-sample group, strand x frame), lane = sample, entries in the reference's order (row by row), eight loads ahead of the
// fold.  The maxima meet those of the other five parts in an atomic max on the preset -1.
__global__ __launch_bounds__(64) void k_null_rowscan(NullArgs A, const DevBlock *__restrict__ dblocks, const int *__restrict__ classBlocks,
                                                      const uint32_t *__restrict__ flags, float *__restrict__ maxima) {
  const int lane = threadIdx.x;
  const int groupsHere = A.gHi - A.gLo;
  const int part = blockIdx.x % 6, item = blockIdx.x / 6;
  const int cb = item / groupsHere, grp = A.gLo + item % groupsHere;
  if (cb >= A.nClassBlocks) return;
  const int bi = classBlocks[cb];
  if (A.skipMask && (flags[bi] & A.skipMask)) return;
  const DevBlock *__restrict__ db = dblocks + bi;
  const int f = part % 3, sites = (db->L - f) / 3;
  if (sites <= 0) return;
  const float *__restrict__ src = A.sbuf + (static_cast<size_t>(item) * 6 + part) * A.sbufStride + lane;
  const float negTie = -A.tieThr;
  SampleScan st{0.0f, 0.0f, 0u, 0u};
  float best = -1.0f;
  // The entries as one stream of sites (sites + 1) / 2 slots, 64 loads in flight at a time (a wavefront that waits for eight loads
  // per round trip spends a microsecond per eight entries: 0.4 ms for one block of 10 x 201); the row bookkeeping is scalar.
  constexpr int kAhead = 64;
  const int total = sites * (sites + 1) / 2;
  int a = 0, slotsLeft = sites;
  float j2f = 1.0f;
  sample_scan_row_begin(st, best, 0u);
  for (int base = 0; base < total; base += kAhead) {
    float v[kAhead];
#pragma unroll
    for (int u = 0; u < kAhead; u++) v[u] = (base + u < total) ? src[static_cast<size_t>(base + u) * kWave] : 0.0f;
#pragma unroll
    for (int u = 0; u < kAhead; u++) {
      if (base + u < total) {
        if (a < sites - 1) { sample_scan_step(st, v[u], j2f, negTie); j2f += 2.0f; }   // (the last row's one slot: the frame's final entry, never computed)
        if (--slotsLeft == 0) {
          sample_scan_row_end(st, static_cast<uint32_t>(a));
          a++;
          if (a < sites) { sample_scan_row_begin(st, best, static_cast<uint32_t>(a)); slotsLeft = sites - a; j2f = static_cast<float>(2 * a + 1); }
        }
      }
    }
  }
  sample_scan_last(st, best);
  const int sidx = grp * kWave + lane;
  if (sidx < A.sampleN && best > -1.0f)
    atomicMax(reinterpret_cast<int *>(maxima + static_cast<size_t>(db->out_index) * A.sampleN + sidx), __float_as_int(best));
}

// ------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------
void launch_mt_stream(uint32_t seedBase, int Spad, int D, uint32_t *U, hipStream_t stream) {
  hipLaunchKernelGGL(k_mt_stream, dim3(Spad), dim3(64), 0, stream, seedBase, Spad, D, U);
}

// k_null is instantiated in eight units (rc_null_a..h.hip), one per range of N-1; e..h hold the wide blocks (N-1 >= 32)
#define RC_NULL_UNIT(S) int null_occupancy_##S(int, size_t); int null_occupancy_plain_##S(int, size_t, int); bool launch_null_##S(int, const NullArgs &, int, size_t, bool, hipStream_t);
RC_NULL_UNIT(a) RC_NULL_UNIT(b) RC_NULL_UNIT(c) RC_NULL_UNIT(d) RC_NULL_UNIT(e) RC_NULL_UNIT(f) RC_NULL_UNIT(g) RC_NULL_UNIT(h)
#undef RC_NULL_UNIT

int null_occupancy(int NK, size_t ldsBytes) {
  if (NK < 2 || NK > kTemplRows - 1) return 0;
  return NK <= 6 ? null_occupancy_a(NK, ldsBytes) : NK <= 12 ? null_occupancy_b(NK, ldsBytes)
       : NK <= 21 ? null_occupancy_c(NK, ldsBytes) : NK <= 31 ? null_occupancy_d(NK, ldsBytes)
       : NK <= 39 ? null_occupancy_e(NK, ldsBytes) : NK <= 47 ? null_occupancy_f(NK, ldsBytes)
       : NK <= 55 ? null_occupancy_g(NK, ldsBytes) : null_occupancy_h(NK, ldsBytes);
}

int null_occupancy_plain(int NK, size_t dynLds, int hiOcc) {
  if (NK < 2 || NK >= kFastRows) return 0;
  return NK <= 6 ? null_occupancy_plain_a(NK, dynLds, hiOcc) : NK <= 12 ? null_occupancy_plain_b(NK, dynLds, hiOcc)
       : NK <= 21 ? null_occupancy_plain_c(NK, dynLds, hiOcc) : null_occupancy_plain_d(NK, dynLds, hiOcc);
}

// N-1 >= 32: only the EXACT instantiation without code staging exists (ldsBytes and exact are ignored)
bool launch_null(int NK, const NullArgs &a, int grid, size_t ldsBytes, bool exact, hipStream_t stream) {
  if (NK < 2 || NK > kTemplRows - 1) return false;
  return NK <= 6 ? launch_null_a(NK, a, grid, ldsBytes, exact, stream) : NK <= 12 ? launch_null_b(NK, a, grid, ldsBytes, exact, stream)
       : NK <= 21 ? launch_null_c(NK, a, grid, ldsBytes, exact, stream) : NK <= 31 ? launch_null_d(NK, a, grid, ldsBytes, exact, stream)
       : NK <= 39 ? launch_null_e(NK, a, grid, ldsBytes, exact, stream) : NK <= 47 ? launch_null_f(NK, a, grid, ldsBytes, exact, stream)
       : NK <= 55 ? launch_null_g(NK, a, grid, ldsBytes, exact, stream) : launch_null_h(NK, a, grid, ldsBytes, exact, stream);
}

#define RC_FOR_NK(X) \
  X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15) X(16) X(17) X(18) X(19) X(20) X(21) \
  X(22) X(23) X(24) X(25) X(26) X(27) X(28) X(29) X(30) X(31) X(32) X(33) X(34) X(35) X(36) X(37) X(38) X(39) X(40) X(41) \
  X(42) X(43) X(44) X(45) X(46) X(47) X(48) X(49) X(50) X(51) X(52) X(53) X(54) X(55) X(56) X(57) X(58) X(59) X(60) X(61) X(62) X(63)

void launch_null_rowscan(const NullArgs &a, int items, hipStream_t stream) {
  if (items <= 0) return;
  hipLaunchKernelGGL(k_null_rowscan, dim3(items * 6), dim3(64), 0, stream, a, a.dblocks, a.classBlocks, a.flags, a.maxima);
}

void launch_native_sigma(const NativeArgs &a, int nblocks, hipStream_t stream) {
  hipLaunchKernelGGL(k_native_sigma, dim3(nblocks), dim3(256), 0, stream, a);
}

bool launch_native_dp(int NK, const NativeArgs &a, int grid, hipStream_t stream) {
  switch (NK) {
#define X(n) case n: hipLaunchKernelGGL(k_native_dp<n>, dim3(grid), dim3(64), 0, stream, a); return true;
    RC_FOR_NK(X)
#undef X
    default: return false;
  }
}

void launch_native_scan(const NativeArgs &a, hipStream_t stream) {
  hipLaunchKernelGGL(k_native_scan, dim3((a.nItems + kWave - 1) / kWave), dim3(64), 0, stream, a);
}

// The HSS records sit in a [block][strand x frame][cap] array of which a few per cent are used: pack the
// used ones for the copy to the host.  offsets[t] = where the records of slot t start in `packed`.
__global__ __launch_bounds__(256) void k_hss_pack(const DevHss *__restrict__ hss, const int *__restrict__ count, int cap, int slots,
                                                  DevHss *__restrict__ packed, int *__restrict__ offsets, int *__restrict__ total) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= slots) return;
  const int c = min(count[t], cap);
  const int off = c > 0 ? atomicAdd(total, c) : 0;
  offsets[t] = off;
  for (int i = 0; i < c; i++) packed[off + i] = hss[static_cast<size_t>(t) * cap + i];
}

void launch_hss_pack(const DevHss *hss, const int *count, int cap, int slots, DevHss *packed, int *offsets, int *total,
                     hipStream_t stream) {
  if (slots <= 0) return;
  hipLaunchKernelGGL(k_hss_pack, dim3((slots + 255) / 256), dim3(256), 0, stream, hss, count, cap, slots, packed, offsets, total);
}


// A batch's small results, written by the GPU straight into the batch's pinned host buffers (mapped into the device's address
// space).  One LDS-free kernel instead of five hipMemcpyAsync: the runtime turns such small device-to-host copies into blit
// kernels, and those found no place on the chip beside another batch's k_null until its tail -- in a stream the results of a
// sub-batch, and with them the next sub-batch on the same compute stream, were held back by a millisecond or more.
__global__ __launch_bounds__(64) void k_results_out(ResultsOutArgs a) {
  __builtin_amdgcn_s_setprio(3);   // a short latency-bound kernel beside k_null: the SIMD issues its instructions first
  const size_t t = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x, step = static_cast<size_t>(gridDim.x) * blockDim.x;
#pragma unroll
  for (int x = 0; x < ResultsOutArgs::kParts; x++)
    for (size_t i = t; i < a.words[x]; i += step) a.dst[x][i] = a.src[x][i];
}

void launch_results_out(const ResultsOutArgs &a, hipStream_t stream) {
  size_t most = 0;
  for (int x = 0; x < ResultsOutArgs::kParts; x++) most = std::max(most, a.words[x]);
  if (!most) return;
  const int grid = static_cast<int>(std::min<size_t>(256, (most + 255) / 256));
  hipLaunchKernelGGL(k_results_out, dim3(grid), dim3(64), 0, stream, a);
}

void launch_sk_row(const uint8_t *blob, const DevBlock *dblocks, const uint32_t *flags, int bi, int s, int b, int iMax, float *out, int stride,
                   hipStream_t stream) {
  hipLaunchKernelGGL(k_sk_row, dim3(1), dim3(64), 0, stream, blob, dblocks, flags, bi, s, b, iMax, out, stride);
}

size_t null_generic_lds_bytes(int N, int nnodes) { return generic_lds_bytes(N, nnodes); }

size_t null_generic_codes_bytes(int N, int L, int nnodes) { return generic_codes_bytes(N, L, nnodes); }
size_t null_generic_state_bytes(int N, int L, int nnodes) { return generic_state_bytes(N, L, nnodes); }
int generic_sim_occupancy(size_t ldsBytes) {
  int nb = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_generic_sim<false>, 64, ldsBytes) != hipSuccess) return 0;
  return nb;
}
int generic_dp_occupancy(size_t ldsBytes) {
  int nb = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_generic_dp, 64, ldsBytes) != hipSuccess) return 0;
  return nb;
}
void launch_generic_sim(const NullArgs &a, int grid, size_t ldsBytes, hipStream_t stream, bool tiled) {
  if (tiled) {
    hipLaunchKernelGGL(k_generic_sim<true>, dim3(grid), dim3(64), ldsBytes, stream, a, a.blob, a.dblocks, a.classBlocks, a.flags, a.U, a.pair, static_cast<uint8_t *>(nullptr), a.maxima);
    return;
  }
  hipLaunchKernelGGL(k_generic_sim<false>, dim3(grid), dim3(64), ldsBytes, stream, a, a.blob, a.dblocks, a.classBlocks, a.flags, a.U, a.pair, static_cast<uint8_t *>(nullptr), a.maxima);
}
void launch_generic_dp(const NullArgs &a, int grid, size_t ldsBytes, uint8_t *scratchBytes, hipStream_t stream) {
  hipLaunchKernelGGL(k_generic_dp, dim3(grid), dim3(64), ldsBytes, stream, a, a.blob, a.dblocks, a.classBlocks, a.flags, a.U, a.pair, scratchBytes, a.maxima);
}

void launch_native_dp_generic(const NativeArgs &a, int nblocks, float *scratch, size_t scratchStride, hipStream_t stream) {
  hipLaunchKernelGGL(k_native_dp_generic, dim3(nblocks * 6), dim3(64), 0, stream, a, scratch, scratchStride);
}

void launch_stop_mark(const FitArgs &a, int nblocks, hipStream_t stream) {
  hipLaunchKernelGGL(k_stop_mark, dim3(nblocks), dim3(64), 0, stream, a);
}

void launch_evd_fit(const FitArgs &a, int nblocks, bool latency, hipStream_t stream) {
  // a small batch with the chip to itself: four wavefronts per block (latency); otherwise one LDS-free wavefront per block
  // (see fit_sum4: beside another batch's k_null the LDS variant waits milliseconds for its 128 bytes)
  // (128 threads: 1.64 ms for the 33 blocks of genomic-preprocessed.maf at n = 1000, 256: 1.27 ms, 1024: 3.1 ms)
  if (latency && nblocks <= 2048) hipLaunchKernelGGL(k_evd_fit<256>, dim3(nblocks), dim3(256), 0, stream, a);
  else hipLaunchKernelGGL(k_evd_fit<64>, dim3(nblocks), dim3(64), 0, stream, a);
}

void launch_evd_fit_f64(const double *x, int n, FitOut *out, int expMode, hipStream_t stream) {
  hipLaunchKernelGGL(k_evd_fit_f64, dim3(1), dim3(256), 0, stream, x, n, out, expMode);
}

}  // namespace rc
