// rc_kernels.hip -- hand-written HIP kernels for gfx950 (MI355X) of the RNAcode scoring path.
//
//   k_mt_stream       MT19937 output streams, one wavefront per seed          (seqgen/twister.c:73-152)
//   k_null<NK>        simulate -> codon-pair codes -> 3-state DP -> HSS scan -> per-sample maximum,
//                     one LANE per null sample, 64 samples of one block per wavefront
//                                                                             (score.c:976-1064 loop body)
//   k_native_sigma    calculateSigma for the native block                     (score.c:375-426)
//   k_native_dp<NK>   pairwise + multiple score matrix of the native block, one lane per start codon
//                                                                             (score.c:441-556, 811-848)
//   k_native_scan     getHSS on the native S matrix                           (score.c:864-974)
//   k_sk_row          one row b of Sk for backtracking                        (score.c:496-535)
//   k_evd_fit         Gumbel maximum-likelihood fit per block                 (extreme_fit.c:157-251)
//
// Numerics: every add/max of the DP is a single IEEE binary32 operation in the reference's
// order; compile with -ffp-contract=off; float division is hipcc's correctly rounded default.
#include <hip/hip_runtime.h>

#include <type_traits>
#include <utility>

#include "rc_device.h"
#include "rc_launch.h"

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
// shared device pieces
// ------------------------------------------------------------------------------------------

// One step of the 3-state max-plus recurrence for one sequence k (score.c:506-533).
// zc: 0 -> z = 0, 1 -> z = +1, 2 -> z = -1 (wave-uniform).
__device__ __forceinline__ void pair_step(int zc, float sig, float Delta, float Omega, float omega,
                                          float &s0, float &s1, float &s2) {
  if (zc == 0) {
    s0 = s0 + sig;
    s1 = s1 + omega;
    s2 = s2 + omega;
  } else if (zc == 1) {
    const float n0 = fmaxf(s0 + Delta, s2 + Omega);
    const float n1 = fmaxf(s0 + Omega, s1 + Delta);
    const float n2 = fmaxf(s1 + Omega, s2 + Delta);
    s0 = n0; s1 = n1; s2 = n2;
  } else {
    const float n0 = fmaxf(s0 + Delta, s1 + Omega);
    const float n1 = fmaxf(s1 + Delta, s2 + Omega);
    const float n2 = fmaxf(s2 + Delta, s0 + Omega);
    s0 = n0; s1 = n1; s2 = n2;
  }
}

// getHSS's serial state machine for one entry (score.c:892-959), emission reduced to a callback.
struct ScanState {
  float cm;  // currMax
  int ss, se;
};

template <typename Emit>
__device__ __forceinline__ void scan_step(ScanState &st, float v, int a, int j, bool last, float tieThr, Emit emit) {
  if (v > 0.0f || last) {
    if ((st.cm > 0.0f && st.se < a) || last) {
      if (st.se - st.ss >= 2) emit(st);
      st.cm = v; st.ss = a; st.se = j;
    } else if (v > st.cm || (fabsf(v - st.cm) < tieThr && (j - a) >= (st.se - st.ss))) {
      st.cm = v; st.ss = a; st.se = j;
    }
  }
}

// reverse the three 2-bit fields of a codon index
__device__ __forceinline__ uint32_t codon_flip(uint32_t c) { return ((c & 3u) << 4) | (c & 12u) | ((c >> 4) & 3u); }

// ------------------------------------------------------------------------------------------
// k_null<NK, LDSC>: the null-distribution loop body, lane = sample
// ------------------------------------------------------------------------------------------

// Per-sequence registers as distinct scalar members (not an array): keeps every chain in its own
// VGPR across the wave-uniform z branches instead of a register tuple that is copied at each join.
template <int K> struct KReg { float s0, s1, s2; int lut; };
template <typename Seq> struct KRegs;
template <int... K> struct KRegs<std::integer_sequence<int, K...>> : KReg<K>... {
  template <int I> __device__ __forceinline__ KReg<I> &at() { return static_cast<KReg<I> &>(*this); }
};
template <int NK, typename F, int... K>
__device__ __forceinline__ void static_for_impl(F &&f, std::integer_sequence<int, K...>) {
  (f(std::integral_constant<int, K>{}), ...);
}
template <int NK, typename F> __device__ __forceinline__ void static_for(F &&f) {
  static_for_impl<NK>(static_cast<F &&>(f), std::make_integer_sequence<int, NK>{});
}

// x / NK, correctly rounded.  Power of two: exact scaling.  Otherwise Markstein's two-FMA
// correction of x * RN(1/NK), proven equal to IEEE division for every NK in 2..31 and every
// |x| in {0} U [2^-100, 2^100) by exhaustion (tools/verify_const_div.c).  Blocks whose score
// tables could leave that range are flagged by the host (DevBlock::exact_div) and are scored by
// the EXACT instantiation, which uses '/'.
template <int NK, bool EXACT> __device__ __forceinline__ float div_by_nk(float x, float nkf) {
  if constexpr ((NK & (NK - 1)) == 0) {
    return x * (1.0f / static_cast<float>(NK));
  } else if constexpr (EXACT) {
    return x / nkf;
  } else {
    constexpr float y = 1.0f / static_cast<float>(NK);
    const float q0 = x * y;
    const float r = __builtin_fmaf(-static_cast<float>(NK), q0, x);
    return __builtin_fmaf(r, y, q0);
  }
}

// getHSS's state machine as the null samples need it: only the best emitted score matters, so the
// segment start is carried as len = segmentEnd - segmentStart (score.c:892-959).
// key = segmentEnd << 16 | (segmentEnd - segmentStart); codon sites < 32767.  Before the first
// segment of a frame segmentEnd holds the sentinel 0x7FFF, which makes "currMax > 0 && segmentEnd < i"
// (score.c:900) equivalent to key < (i << 16): currMax is positive exactly when a segment was opened.
struct SampleScan { float cm; uint32_t key; };
constexpr uint32_t kScanInitKey = 0x7FFFu << 16;
// every entry except the frame's final one
__device__ __forceinline__ void sample_scan_step(SampleScan &st, float &best, float v, uint32_t a, uint32_t span, uint32_t newkey,
                                                 float tieThr) {
  const uint32_t len = st.key & 0xFFFFu;
  const bool pos = v > 0.0f;
  const bool fresh = st.key < (a << 16);
  const bool rep = (v > st.cm) | ((fabsf(v - st.cm) < tieThr) & (len <= span));   // (j-i) >= (segEnd-segStart), score.c:953-954
  const bool emit = pos & fresh & (len >= 2u);                                    // minSegmentLength, score.c:902
  best = (emit & (st.cm > best)) ? st.cm : best;
  const bool upd = pos & (fresh | rep);
  st.cm = upd ? v : st.cm;
  st.key = upd ? newkey : st.key;
}
// the final entry of a frame is entered unconditionally and always reports the open segment
__device__ __forceinline__ void sample_scan_last(const SampleScan &st, float &best) {
  best = (((st.key & 0xFFFFu) >= 2u) & (st.cm > best)) ? st.cm : best;
}

// Read-only inputs are separate __restrict__ kernel parameters (not members of the by-value
// argument block) so that the compiler may prove them unclobbered and fetch the wave-uniform
// block tables (z words, tree records, masks) with scalar loads.
template <int NK, bool LDSC, bool EXACT>
__global__ __launch_bounds__(64) void k_null(NullArgs A, const uint8_t *__restrict__ blob,
                                             const DevBlock *__restrict__ dblocks, const int *__restrict__ classBlocks,
                                             const uint32_t *__restrict__ Ustream, const uint8_t *__restrict__ pairTab,
                                             uint32_t *__restrict__ scratch, float *__restrict__ maxima) {
  constexpr int N = NK + 1;
  constexpr int CPW = 5;                  // 6-bit sigma codes per 32-bit word
  constexpr int NCW = (NK + CPW - 1) / CPW;
  using Regs = KRegs<std::make_integer_sequence<int, NK>>;
  // phase A: threshold table of the tree (nnodes x 64 B); phase B with LDSC: [site][NCW][64] sigma codes
  // of the current strand x frame
  extern __shared__ uint32_t lds_codes[];

  const int lane = threadIdx.x;
  uint32_t *__restrict__ cod = scratch + static_cast<size_t>(blockIdx.x) * A.scratchStride;
  // comboSplit (small batches): an item is one (block, group, strand x frame) so that 6x more
  // wavefronts exist; each recomputes phase A and the six partial maxima meet in an atomic max.
  const int perBlock = A.groups * (A.comboSplit ? 6 : 1);
  const int nItems = A.nClassBlocks * perBlock;

  for (int item = blockIdx.x; item < nItems; item += gridDim.x) {
    const int bi = classBlocks[item / perBlock];
    const int sub = item % perBlock;
    const int grp = A.comboSplit ? sub / 6 : sub;
    const int onlyCombo = A.comboSplit ? sub % 6 : -1;
    const DevBlock *__restrict__ db = dblocks + bi;
    const int cols = db->cols, L = db->L, nnodes = db->nnodes;
    const float Delta = db->Delta, Omega = db->Omega, omega = db->omega, nkf = db->nkf;
    const NodeRec *__restrict__ nodes = reinterpret_cast<const NodeRec *>(blob + db->off_nodes);
    const uint8_t *__restrict__ qtip = blob + db->off_qtip;
    const uint16_t *__restrict__ refcol = reinterpret_cast<const uint16_t *>(blob + db->off_refcol);
    const unsigned long long *__restrict__ zw = reinterpret_cast<const unsigned long long *>(blob + db->off_zw);
    const unsigned long long *__restrict__ zmask = reinterpret_cast<const unsigned long long *>(blob + db->off_zmask);
    const int zmWords = db->zm_words;
    const uint8_t *__restrict__ mask = blob + db->off_mask;
    const float *__restrict__ lut = reinterpret_cast<const float *>(blob + db->off_lut);
    const int sidx = grp * kWave + lane;
    const int L1 = L + 1;

    // ---- phase A: simulate the L reference-residue columns (evolve.c:416-433), SPI sites per pass
    // so that the stream loads and threshold look-ups of independent sites overlap; emit sigma
    // codes for both strands.  Node states of a site are packed 2 bits per node into NODEW
    // registers; the per-branch threshold table sits in LDS (the region phase B reuses for codes).
    if (!(A.debugSkip & 2)) {
      constexpr int SPI = 4;
      constexpr int NODEW = (2 * N - 1 + 15) / 16;   // 16 nodes per 32-bit word
      U4 *ldsThr = reinterpret_cast<U4 *>(lds_codes);
      {
        const uint32_t *src = reinterpret_cast<const uint32_t *>(nodes);
        for (int idx = lane; idx < nnodes * 16; idx += kWave) lds_codes[idx] = src[(idx >> 4) * (sizeof(NodeRec) / 4) + (idx & 15)];
      }
      uint32_t win[N];
#pragma unroll
      for (int r = 0; r < N; r++) win[r] = 0;
      unsigned int clamped = 0;
      auto get_state = [&](const uint32_t (&ns)[NODEW], int q) -> uint32_t {   // q is wave-uniform
        uint32_t word = ns[0];
#pragma unroll
        for (int x = 1; x < NODEW; x++) if ((q >> 4) == x) word = ns[x];
        return (word >> (2 * (q & 15))) & 3u;
      };
      for (int i0 = 1; i0 <= L; i0 += SPI) {
        size_t ubase[SPI];
        uint32_t ns[SPI][NODEW];
#pragma unroll
        for (int t = 0; t < SPI; t++) {
          const int i = (i0 + t <= L) ? i0 + t : L;     // tail: repeat the last site, results unused
          ubase[t] = static_cast<size_t>(refcol[i]) * A.Spad + sidx;
#pragma unroll
          for (int x = 0; x < NODEW; x++) ns[t][x] = 0;
        }
        const size_t qstride = static_cast<size_t>(cols) * A.Spad;
        for (int q = 0; q < nnodes; q++) {
          const int pq = nodes[q].parent;
          const uint32_t bp = nodes[q].basepack;
          uint32_t u[SPI];
#pragma unroll
          for (int t = 0; t < SPI; t++) u[t] = Ustream[ubase[t] + q * qstride];
#pragma unroll
          for (int t = 0; t < SPI; t++) {
            const uint32_t ps = (q == 0) ? 0u : get_state(ns[t], pq);
            const U4 th = ldsThr[q * 4 + ps];
            uint32_t st = (u[t] > th.x) + (u[t] > th.y) + (u[t] > th.z);
            if (bp) st += (bp >> (2 * ps)) & 3u;
            clamped += (u[t] > th.w);
            const uint32_t bits = st << (2 * (q & 15));
#pragma unroll
            for (int x = 0; x < NODEW; x++) if (NODEW == 1 || (q >> 4) == x) ns[t][x] |= bits;
          }
        }
#pragma unroll
        for (int t = 0; t < SPI; t++) {
          const int i = i0 + t;
          if (i <= L) {
#pragma unroll
            for (int r = 0; r < N; r++) win[r] = ((win[r] << 2) | get_state(ns[t], qtip[r])) & 63u;
            if (i >= 3) {
              const int ip = L + 3 - i;  // reverse-strand position whose codon is these three columns
              const uint32_t aF = win[0];
              const uint32_t aR = codon_flip(aF) ^ 63u;
#pragma unroll
              for (int w = 0; w < NCW; w++) {
                uint32_t wf = 0, wr = 0;
#pragma unroll
                for (int c = 0; c < CPW; c++) {
                  const int k = w * CPW + c;  // 0-based index of sequence k+1
                  if (k < NK) {
                    const uint32_t bF = win[k + 1] & mask[(0 * NK + k) * L1 + i];
                    const uint32_t bR = (codon_flip(win[k + 1]) ^ 63u) & mask[(1 * NK + k) * L1 + ip];
                    wf |= static_cast<uint32_t>(pairTab[aF * 64 + bF]) << (6 * c);
                    wr |= static_cast<uint32_t>(pairTab[aR * 64 + bR]) << (6 * c);
                  }
                }
                cod[(static_cast<size_t>(0 * L1 + i) * NCW + w) * kWave + lane] = wf;
                cod[(static_cast<size_t>(1 * L1 + ip) * NCW + w) * kWave + lane] = wr;
              }
            }
          }
        }
      }
      if (clamped) atomicAdd(A.clampCount, static_cast<unsigned long long>(clamped));
    }

    // ---- phase B: both strands x 3 frames: DP over (start a, end j) + serial HSS scan
    float best = -1.0f;
    // omega and Delta as per-lane values: v_add_f32 with two VGPR operands issues at the full rate,
    // with an SGPR operand at ~60 % of it (tools/microbench.hip)
    float omega_v, Delta_v;
    asm volatile("v_mov_b32 %0, %1" : "=v"(omega_v) : "s"(omega));
    asm volatile("v_mov_b32 %0, %1" : "=v"(Delta_v) : "s"(Delta));
    if (!(A.debugSkip & 1))
    for (int s = 0; s < 2; s++) {
      if (onlyCombo >= 0 && onlyCombo / 3 != s) continue;
      Regs R;
      static_for<NK>([&](auto kc) {  // 64-entry sigma table of sequence k spread over the 64 lanes
        constexpr int k = decltype(kc)::value;
        R.template at<k>().lut = __float_as_int(lut[(s * NK + k) * kLutSize + lane]);
      });
      for (int f = 0; f < 3; f++) {
        const int sites = (L - f) / 3;
        if (sites <= 0) continue;
        if (onlyCombo >= 0 && onlyCombo % 3 != f) continue;
        const uint32_t *__restrict__ cbase = cod + (static_cast<size_t>(s * L1 + 3 + f) * NCW) * kWave + lane;  // site j: + j*3*NCW*64
        const unsigned long long *__restrict__ zbase = zw + s * L1 + 3 + f;                                       // site j: [3j]
        const unsigned long long *__restrict__ zany = zmask + static_cast<size_t>(s * 3 + f) * zmWords;            // bit j: some z != 0 at site j
        if constexpr (LDSC) {
          for (int idx = 0; idx < sites * NCW; idx++)
            lds_codes[idx * kWave + lane] = cbase[static_cast<size_t>(idx / NCW) * 3 * NCW * kWave + (idx % NCW) * kWave];
          __builtin_amdgcn_s_waitcnt(0);   // everything landed: the row loops below issue no vector memory ops
        }
        auto load_words = [&](int j, uint32_t (&w)[NCW]) {
#pragma unroll
          for (int x = 0; x < NCW; x++) {
            if constexpr (LDSC) w[x] = lds_codes[(j * NCW + x) * kWave + lane];
            else w[x] = cbase[(static_cast<size_t>(j) * 3 * NCW + x) * kWave];
          }
        };
        // sigma of every sequence at the site whose code words are w (all table look-ups first: their
        // LDS round trips overlap).  ds_bpermute uses address bits [7:2] as the source lane.
        auto lookup = [&](const uint32_t (&w)[NCW], float (&sig)[NK]) {
          static_for<NK>([&](auto kc) {
            constexpr int k = decltype(kc)::value;
            constexpr int t = k % CPW;
            const uint32_t addr = (t == 0) ? (w[k / CPW] << 2) : (w[k / CPW] >> (6 * t - 2));
            sig[k] = __int_as_float(__builtin_amdgcn_ds_bpermute(static_cast<int>(addr), R.template at<k>().lut));
          });
          if constexpr (LDSC) __builtin_amdgcn_s_waitcnt(0xC07F);   // one lgkmcnt(0) instead of one wait per look-up
        };
        // S[a][j] = MAX3(sum, 0+Delta, 0+Delta)/(N-1) (score.c:841-843).  The reference starts the sum
        // from 0; 0 + m == m except for the sign of a zero, which nothing observes.
        auto finish = [&](float sum) -> float {
          const float top = (sum > Delta_v) ? sum : Delta_v;
          return div_by_nk<NK, EXACT>(top, nkf);
        };
        // codon without a frame shift in any sequence (z == 0 for all k, score.c:506-510): straight-line
        auto fast_cell = [&](const uint32_t (&w)[NCW]) -> float {
          float sig[NK];
          lookup(w, sig);
          float sum = 0.0f;
          static_for<NK>([&](auto kc) {
            constexpr int k = decltype(kc)::value;
            auto &r = R.template at<k>();
            r.s0 = r.s0 + sig[k];
            r.s1 = r.s1 + omega_v;
            r.s2 = r.s2 + omega_v;
            const float m = fmaxf(fmaxf(r.s0, r.s1), r.s2);
            sum = (k == 0) ? m : sum + m;
          });
          return finish(sum);
        };
        // codon where some sequence has z != 0 (score.c:512-533), wave-uniform branch per sequence
        auto event_cell = [&](const uint32_t (&w)[NCW], int j) -> float {
          float sig[NK];
          lookup(w, sig);
          const unsigned long long z = zbase[3 * j];
          float sum = 0.0f;
          static_for<NK>([&](auto kc) {
            constexpr int k = decltype(kc)::value;
            auto &r = R.template at<k>();
            pair_step(static_cast<int>((z >> (2 * k)) & 3ull), sig[k], Delta, Omega, omega, r.s0, r.s1, r.s2);
            const float m = fmaxf(fmaxf(r.s0, r.s1), r.s2);
            sum = (k == 0) ? m : sum + m;
          });
          return finish(sum);
        };
        // first site >= j (and < end) with an event, or end
        auto next_event = [&](int j, int end) -> int {
          while (j < end) {
            const unsigned long long mword = zany[j >> 6] >> (j & 63);
            if (mword) { const int e = j + __builtin_ctzll(mword); return e < end ? e : end; }
            j = (j | 63) + 1;
          }
          return end;
        };
        SampleScan st{0.0f, kScanInitKey};
        for (int a = 0; a < sites; a++) {
          static_for<NK>([&](auto kc) {
            constexpr int k = decltype(kc)::value;
            auto &r = R.template at<k>();
            r.s0 = r.s1 = r.s2 = 0.0f;
          });
          uint32_t wcur[NCW];
          load_words(a, wcur);
          const int jend = (a == sites - 1) ? sites - 1 : sites;   // the frame's final entry is handled below
          int j = a;
          while (j < jend) {
            const int e = next_event(j, jend);
#pragma unroll 2
            for (; j < e; j++) {                                   // tight loop: no events
              uint32_t wnext[NCW];
              load_words(j + 1 < sites ? j + 1 : j, wnext);
              const float v = fast_cell(wcur);
              sample_scan_step(st, best, v, static_cast<uint32_t>(a), static_cast<uint32_t>(j - a), (static_cast<uint32_t>(j) << 16) | static_cast<uint32_t>(j - a), A.tieThr);
#pragma unroll
              for (int x = 0; x < NCW; x++) wcur[x] = wnext[x];
            }
            if (j < jend) {
              uint32_t wnext[NCW];
              load_words(j + 1 < sites ? j + 1 : j, wnext);
              const float v = event_cell(wcur, j);
              sample_scan_step(st, best, v, static_cast<uint32_t>(a), static_cast<uint32_t>(j - a), (static_cast<uint32_t>(j) << 16) | static_cast<uint32_t>(j - a), A.tieThr);
#pragma unroll
              for (int x = 0; x < NCW; x++) wcur[x] = wnext[x];
              j++;
            }
          }
        }
        sample_scan_last(st, best);   // entry (sites-1, sites-1): its own value can never be reported
      }
    }
    if (sidx < A.sampleN) {
      float *dst = maxima + static_cast<size_t>(db->out_index) * A.sampleN + sidx;
      // scores are positive floats or the -1 sentinel: integer order == float order on that set
      if (A.comboSplit) atomicMax(reinterpret_cast<int *>(dst), __float_as_int(best));
      else *dst = best;
    }
  }
}

// ------------------------------------------------------------------------------------------
// native block
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_native_sigma(NativeArgs A) {
  const int bi = A.blocks[blockIdx.x];
  const DevBlock *__restrict__ db = A.dblocks + bi;
  const int N = db->N, NK = db->NK, cols = db->cols, L = db->L, L1 = L + 1;
  uint8_t *blob = A.blob;
  const uint16_t *refcol = reinterpret_cast<const uint16_t *>(blob + db->off_refcol);
  const uint8_t *chars = blob + db->off_chars;
  const float *lut = reinterpret_cast<const float *>(blob + db->off_lut);
  float *sigma = reinterpret_cast<float *>(blob + db->off_sigma);
  const int total = 2 * NK * L1;
  for (int idx = threadIdx.x; idx < total; idx += blockDim.x) {
    const int i = idx % L1, k = (idx / L1) % NK, s = idx / (L1 * NK);
    float sg = 0.0f;
    if (i >= 3) {
      const uint8_t *r0 = chars + static_cast<size_t>(s * N + 0) * cols;
      const uint8_t *rk = chars + static_cast<size_t>(s * N + k + 1) * cols;
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

template <int NK>
__global__ __launch_bounds__(64) void k_native_dp(NativeArgs A) {
  const int lane = threadIdx.x;
  const int bi = A.blocks[blockIdx.x / 6];
  const int combo = blockIdx.x % 6, s = combo / 3, f = combo % 3;
  const DevBlock *__restrict__ db = A.dblocks + bi;
  const int L = db->L, L1 = L + 1;
  const float Delta = db->Delta, Omega = db->Omega, omega = db->omega, nkf = db->nkf;
  const unsigned long long *zw = reinterpret_cast<const unsigned long long *>(A.blob + db->off_zw);
  const float *sigma = reinterpret_cast<const float *>(A.blob + db->off_sigma);
  const int sites = (L - f) / 3, smax = L / 3;
  float *S = A.S + A.offS[bi] + static_cast<size_t>(combo) * smax * smax;
  for (int a0 = 0; a0 < sites; a0 += kWave) {
    const int a = a0 + lane;
    float s0[NK], s1[NK], s2[NK];
#pragma unroll
    for (int k = 0; k < NK; k++) s0[k] = s1[k] = s2[k] = 0.0f;
    for (int j = a0; j < sites; j++) {
      const int i = 3 * j + 3 + f;
      const unsigned long long z = zw[s * L1 + i];
      if (a <= j && a < sites) {
        float sum = 0.0f;
#pragma unroll
        for (int k = 0; k < NK; k++) {
          const float sig = sigma[(s * NK + k) * L1 + i];
          pair_step(static_cast<int>((z >> (2 * k)) & 3ull), sig, Delta, Omega, omega, s0[k], s1[k], s2[k]);
          sum = sum + fmaxf(fmaxf(s0[k], s1[k]), s2[k]);
        }
        S[static_cast<size_t>(a) * sites + j] = fmaxf(sum, Delta) / nkf;
      }
    }
  }
}

__global__ __launch_bounds__(64) void k_native_scan(NativeArgs A) {
  __shared__ float row[kWave];
  const int lane = threadIdx.x;
  const int slot = blockIdx.x;            // position in the blocks list
  const int bi = A.blocks[slot / 6];
  const int combo = slot % 6, s = combo / 3, f = combo % 3;
  const DevBlock *__restrict__ db = A.dblocks + bi;
  const int L = db->L;
  const int sites = (L - f) / 3, smax = L / 3;
  const float *S = A.S + A.offS[bi] + static_cast<size_t>(combo) * smax * smax;
  DevHss *out = A.hss + (static_cast<size_t>(bi) * 6 + combo) * A.hssCap;
  int n = 0;
  ScanState st{0.0f, -1, -1};
  for (int a = 0; a < sites; a++) {
    for (int j0 = a; j0 < sites; j0 += kWave) {
      __syncthreads();
      if (j0 + lane < sites) row[lane] = S[static_cast<size_t>(a) * sites + j0 + lane];
      __syncthreads();
      if (lane == 0) {
        const int jend = min(sites, j0 + kWave);
        for (int j = j0; j < jend; j++) {
          const bool last = (a == sites - 1) && (j == sites - 1);
          scan_step(st, row[j - j0], a, j, last, A.tieThr, [&](const ScanState &e) {
            if (n < A.hssCap) out[n] = DevHss{s, f, e.ss, e.se, e.cm};
            n++;
          });
        }
      }
    }
  }
  if (lane == 0) A.hssCount[static_cast<size_t>(bi) * 6 + combo] = n;
}

// One row b of Sk for all k (lane = k-1): out[(k-1)*3 + x][t] for i = b+2+3t <= iMax.
__global__ __launch_bounds__(64) void k_sk_row(const uint8_t *blob, const DevBlock *dblocks, int bi, int s, int b, int iMax,
                                               float *out, int stride) {
  const DevBlock *db = dblocks + bi;
  const int k = threadIdx.x;
  if (k >= db->NK) return;
  const int L1 = db->L + 1;
  const unsigned long long *zw = reinterpret_cast<const unsigned long long *>(blob + db->off_zw);
  const float *sigma = reinterpret_cast<const float *>(blob + db->off_sigma);
  float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f;
  int t = 0;
  for (int i = b + 2; i <= iMax; i += 3, t++) {
    const int zc = static_cast<int>((zw[s * L1 + i] >> (2 * k)) & 3ull);
    // lanes follow different k: the branch may diverge here, which is fine for this helper
    pair_step(zc, sigma[(s * db->NK + k) * L1 + i], db->Delta, db->Omega, db->omega, s0, s1, s2);
    out[(k * 3 + 0) * stride + t] = s0;
    out[(k * 3 + 1) * stride + t] = s1;
    out[(k * 3 + 2) * stride + t] = s2;
  }
}

// ------------------------------------------------------------------------------------------
// EVD fit (extreme_fit.c:157-251), one workgroup per block, sums reduced across 256 threads
// ------------------------------------------------------------------------------------------
struct Sums4 { double e, xe, xxe, x; };

__device__ __forceinline__ double block_sum(double v, double *sh) {
  const int tid = threadIdx.x;
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  __syncthreads();
  if ((tid & 63) == 0) sh[tid >> 6] = v;
  __syncthreads();
  double r = 0.0;
  for (int w = 0; w < static_cast<int>(blockDim.x >> 6); w++) r += sh[w];
  return r;
}

// all four sums of Lawless 4.1.6 in one pass over the workgroup (one barrier pair per evaluation)
__device__ __forceinline__ void block_sum4(double &a, double &b, double &c, double &d, double *sh) {
  const int tid = threadIdx.x, nw = static_cast<int>(blockDim.x >> 6);
  for (int off = 32; off > 0; off >>= 1) {
    a += __shfl_down(a, off, 64); b += __shfl_down(b, off, 64);
    c += __shfl_down(c, off, 64); d += __shfl_down(d, off, 64);
  }
  __syncthreads();
  if ((tid & 63) == 0) { sh[(tid >> 6) * 4 + 0] = a; sh[(tid >> 6) * 4 + 1] = b; sh[(tid >> 6) * 4 + 2] = c; sh[(tid >> 6) * 4 + 3] = d; }
  __syncthreads();
  a = b = c = d = 0.0;
  for (int w = 0; w < nw; w++) { a += sh[w * 4 + 0]; b += sh[w * 4 + 1]; c += sh[w * 4 + 2]; d += sh[w * 4 + 3]; }
}

__device__ void lawless416(const float *x, int n, double lambda, double *sh, double *f, double *df) {
  double e = 0, xe = 0, xxe = 0, xs = 0;
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    const double xi = static_cast<double>(x[i]);
    const double ex = exp(-1. * lambda * xi);
    xs += xi; xe += xi * ex; xxe += xi * xi * ex; e += ex;
  }
  block_sum4(e, xe, xxe, xs, sh);
  const double total = static_cast<double>(n);
  *f = 1. / lambda - xs / total + xe / e;
  *df = ((xe / e) * (xe / e)) - (xxe / e) - (1. / (lambda * lambda));
}

__device__ int evd_fit_block(const float *x, int n, double *sh, double *mu_out, double *lambda_out) {
  double lambda = 0.2, fx, dfx;
  const double tol = 1e-5;
  int it;
  for (it = 0; it < 100; it++) {
    lawless416(x, n, lambda, sh, &fx, &dfx);
    if (fabs(fx) < tol) break;
    lambda = lambda - fx / dfx;
    if (lambda <= 0.) lambda = 0.001;
  }
  if (it == 100) {  // bisection fallback (extreme_fit.c:193-236)
    double left = 0.2, right = 0.2, mid = 0.2;
    lambda = 0.2;
    lawless416(x, n, lambda, sh, &fx, &dfx);
    if (fx < 0.) {
      do {
        left -= 0.1;
        if (left < 0.) return 0;
        lawless416(x, n, left, sh, &fx, &dfx);
      } while (fx < 0.);
    } else {
      do {
        right += 0.1;
        lawless416(x, n, right, sh, &fx, &dfx);
        if (right > 100.) return 0;
      } while (fx > 0.);
    }
    for (it = 0; it < 100; it++) {
      mid = (left + right) / 2.;
      lawless416(x, n, mid, sh, &fx, &dfx);
      if (fabs(fx) < tol) break;
      if (fx > 0.) left = mid; else right = mid;
    }
    if (it == 100) return 0;
    lambda = mid;
  }
  double e = 0;
  for (int i = threadIdx.x; i < n; i += blockDim.x) e += exp(-1 * lambda * static_cast<double>(x[i]));
  e = block_sum(e, sh);
  *mu_out = -1. * log(e / static_cast<double>(n)) / lambda;
  *lambda_out = lambda;
  return 1;
}

__global__ __launch_bounds__(256) void k_evd_fit(FitArgs A) {
  __shared__ double sh[16];
  __shared__ float maxNativeSh;
  const int bi = A.blocks[blockIdx.x];
  const DevBlock *db = A.dblocks + bi;
  const float *x = A.maxima + static_cast<size_t>(db->out_index) * A.sampleN;
  if (threadIdx.x == 0) {  // best native HSS = results[0].score after the sort, -1 if none (RNAcode.c:176-178)
    float m = -1.0f;
    for (int c = 0; c < 6; c++) {
      const int n = min(A.hssCount[static_cast<size_t>(bi) * 6 + c], A.hssCap);
      const DevHss *h = A.hss + (static_cast<size_t>(bi) * 6 + c) * A.hssCap;
      for (int i = 0; i < n; i++) if (h[i].score > 0.0f) m = fmaxf(m, h[i].score);
    }
    maxNativeSh = m;
  }
  __syncthreads();
  const float maxNative = maxNativeSh;
  double better = 0;
  for (int i = threadIdx.x; i < A.sampleN; i += blockDim.x) better += (x[i] > maxNative) ? 1.0 : 0.0;
  better = block_sum(better, sh);
  FitOut o;
  o.better = static_cast<int>(better);
  o.rc = -1; o.mu = 0; o.lambda = 0;
  const bool stopped = A.stopEarly && (o.better > A.stopCutoff);  // score.c:1040-1042
  if (!stopped) {
    double mu, lambda;
    if (evd_fit_block(x, A.sampleN, sh, &mu, &lambda) == 1) { o.rc = 1; o.mu = mu; o.lambda = lambda; }
  }
  if (threadIdx.x == 0) A.out[db->out_index] = o;
}

// EVDMaxLikelyFit on caller-supplied doubles (rc_evd_fit): same code path with float inputs would
// lose bits, so this variant reads doubles.
__global__ __launch_bounds__(256) void k_evd_fit_f64(const double *x, int n, FitOut *out) {
  __shared__ double sh[8];
  // reuse the float path by templating would duplicate code; the sums are written out here
  double lambda = 0.2, fx = 0, dfx = 0;
  const double tol = 1e-5;
  auto law = [&](double lam, double *f, double *df) {
    double e = 0, xe = 0, xxe = 0, xs = 0;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
      const double xi = x[i];
      const double ex = exp(-1. * lam * xi);
      xs += xi; xe += xi * ex; xxe += xi * xi * ex; e += ex;
    }
    e = block_sum(e, sh); xe = block_sum(xe, sh); xxe = block_sum(xxe, sh); xs = block_sum(xs, sh);
    *f = 1. / lam - xs / static_cast<double>(n) + xe / e;
    *df = ((xe / e) * (xe / e)) - (xxe / e) - (1. / (lam * lam));
  };
  int it, rc = 1;
  for (it = 0; it < 100; it++) {
    law(lambda, &fx, &dfx);
    if (fabs(fx) < tol) break;
    lambda = lambda - fx / dfx;
    if (lambda <= 0.) lambda = 0.001;
  }
  if (it == 100) {
    double left = 0.2, right = 0.2, mid = 0.2;
    law(0.2, &fx, &dfx);
    if (fx < 0.) {
      do { left -= 0.1; if (left < 0.) { rc = 0; break; } law(left, &fx, &dfx); } while (fx < 0.);
    } else {
      do { right += 0.1; law(right, &fx, &dfx); if (right > 100.) { rc = 0; break; } } while (fx > 0.);
    }
    if (rc) {
      for (it = 0; it < 100; it++) {
        mid = (left + right) / 2.;
        law(mid, &fx, &dfx);
        if (fabs(fx) < tol) break;
        if (fx > 0.) left = mid; else right = mid;
      }
      if (it == 100) rc = 0;
      lambda = mid;
    }
  }
  double mu = 0;
  if (rc) {
    double e = 0;
    for (int i = threadIdx.x; i < n; i += blockDim.x) e += exp(-1 * lambda * x[i]);
    e = block_sum(e, sh);
    mu = -1. * log(e / static_cast<double>(n)) / lambda;
  }
  if (threadIdx.x == 0) { out->rc = rc; out->mu = mu; out->lambda = lambda; out->better = 0; }
}

// ------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------
void launch_mt_stream(uint32_t seedBase, int Spad, int D, uint32_t *U, hipStream_t stream) {
  hipLaunchKernelGGL(k_mt_stream, dim3(Spad), dim3(64), 0, stream, seedBase, Spad, D, U);
}

template <int NK>
static int null_occupancy_t(size_t ldsBytes) {
  int nb = 0;
  hipError_t e = ldsBytes ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_null<NK, true, false>, 64, ldsBytes)
                          : hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_null<NK, false, false>, 64, kMaxNodes * 64);
  if (e != hipSuccess) return 0;
  return nb;
}

#define RC_FOR_NK(X) \
  X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15) X(16) X(17) X(18) X(19) X(20) X(21) \
  X(22) X(23) X(24) X(25) X(26) X(27) X(28) X(29) X(30) X(31)

int null_occupancy(int NK, size_t ldsBytes) {
  switch (NK) {
#define X(n) case n: return null_occupancy_t<n>(ldsBytes);
    RC_FOR_NK(X)
#undef X
    default: return 0;
  }
}

// exact: the class contains blocks flagged exact_div (never staged in LDS: a rare parameter choice)
bool launch_null(int NK, const NullArgs &a, int grid, size_t ldsBytes, bool exact, hipStream_t stream) {
#define RC_NULL_ARGS a, a.blob, a.dblocks, a.classBlocks, a.U, a.pair, a.scratch, a.maxima
  switch (NK) {
#define X(n)                                                                                                   \
  case n:                                                                                                      \
    if (exact) {                                                                                               \
      hipLaunchKernelGGL((k_null<n, false, true>), dim3(grid), dim3(64), kMaxNodes * 64, stream, RC_NULL_ARGS); \
    } else if (ldsBytes) {                                                                                     \
      if (ldsBytes > 48 * 1024)                                                                                \
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_null<n, true, false>),                      \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(ldsBytes));     \
      hipLaunchKernelGGL((k_null<n, true, false>), dim3(grid), dim3(64), ldsBytes, stream, RC_NULL_ARGS);      \
    } else {                                                                                                   \
      hipLaunchKernelGGL((k_null<n, false, false>), dim3(grid), dim3(64), kMaxNodes * 64, stream, RC_NULL_ARGS); \
    }                                                                                                          \
    return true;
    RC_FOR_NK(X)
#undef X
#undef RC_NULL_ARGS
    default: return false;
  }
}

void launch_native_sigma(const NativeArgs &a, int nblocks, hipStream_t stream) {
  hipLaunchKernelGGL(k_native_sigma, dim3(nblocks), dim3(256), 0, stream, a);
}

bool launch_native_dp(int NK, const NativeArgs &a, int nblocks, hipStream_t stream) {
  switch (NK) {
#define X(n) case n: hipLaunchKernelGGL(k_native_dp<n>, dim3(nblocks * 6), dim3(64), 0, stream, a); return true;
    RC_FOR_NK(X)
#undef X
    default: return false;
  }
}

void launch_native_scan(const NativeArgs &a, int nblocks, hipStream_t stream) {
  hipLaunchKernelGGL(k_native_scan, dim3(nblocks * 6), dim3(64), 0, stream, a);
}

void launch_sk_row(const uint8_t *blob, const DevBlock *dblocks, int bi, int s, int b, int iMax, float *out, int stride,
                   hipStream_t stream) {
  hipLaunchKernelGGL(k_sk_row, dim3(1), dim3(64), 0, stream, blob, dblocks, bi, s, b, iMax, out, stride);
}

void launch_evd_fit(const FitArgs &a, int nblocks, hipStream_t stream) {
  hipLaunchKernelGGL(k_evd_fit, dim3(nblocks), dim3(256), 0, stream, a);
}

void launch_evd_fit_f64(const double *x, int n, FitOut *out, hipStream_t stream) {
  hipLaunchKernelGGL(k_evd_fit_f64, dim3(1), dim3(256), 0, stream, x, n, out);
}

}  // namespace rc
