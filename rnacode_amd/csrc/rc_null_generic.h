// rc_null_generic.h -- the null-distribution loop body and the native DP for blocks with more rows than the register-resident
// kernels take (N > 64; the reference accepts up to MAX_NUM_NAMES = 500 rows, rnaz_utils.h:7: 100-way genome alignments).
//
// Same decomposition as k_null (lane = null sample, 64 samples of one block per wavefront, work items from per-XCD queues) and
// the same arithmetic in the same order, but nothing is unrolled over the sequences: the node states of the simulation and the
// sigma codes live in a per-workgroup scratch in global memory ([...][lane], coalesced), tables are read from the blob.  A DP row
// is cut into tiles of kGenericTile end codons; within a tile the loop runs sequence by sequence and adds each sequence's max3 to
// the tile's sums, which stay in registers, in the reference's order k = 0, 1, ...; the HSS scan then walks the tile.
// Round 4 -- 1024 blocks of 100 x 300 at n = 1000: 1.60 -> 1.03 s, 256 blocks: 412 -> 328 ms:
//  * a sequence that has met no frame shift since the row's start has s1 == s2 == the row's running sum of omega, the same number
//    in every lane: its pass keeps s0 only (add / max / add per cell instead of three adds and a max3) and parks s0 only;
//  * rows go in pairs on one set of look-ups and code words (half the ds_bpermutes, half the code traffic);
//  * the look-ups and cells of a group of eight are written out (the compiler waited for every look-up behind its issue), the
//    codes are stored times four so that the code word itself is a ds_bpermute address (tools/mb_bpermute_addr.hip);
//  * the sums are distinct scalars updated in ONE place per group (an array, or an update per kind of pass, and the register
//    allocator keeps two sets of sums and copies all of them at both ends of every pass);
//  * scalar work of the common pass cut to one decision word; the parked s0 stored behind the next pass's loads.
// Round 5 -- 1024 blocks: 1.03 -> 0.93 s, 500 blocks of 200 x 90: 225 -> 189 ms, 16 blocks of 100 x 300: 67 -> 50 ms:
//  * the simulation rebuilt around what it cost, the CU's one scalar unit and chains of dependent loads (its ISA had ~50 scalar
//    instructions and nine scalar loads per tree node, the parent's index fetched by a VECTOR load with a full wait behind it, one
//    table gather round trip per sequence and site): thresholds by ds_bpermute out of one coalesced word per four nodes, control
//    words ready-made by the host, operands of the next eight nodes and of eight sequences in flight (see the phase below);
//    alone it takes 117 ms of the 1024-block run, the DP 727 ms;
//  * the two phases as two launches per round of items for every batch size, not one kernel (see generic_body).
// What bounds the DP, measured: without its memory operations it runs at 0.50 s; its access pattern alone reaches 6.3 TB/s over a
// 20 GB footprint (tools/mb_footprint.hip) and a pass moves 3 KB for 64 cell pairs (eight code words, two parked states in, two
// out): ~3.1 TB per 1024 blocks, every byte from HBM (5 MB of codes per wavefront x 4096 wavefronts); more wavefronts per SIMD
// (5, 6), the next pass's operands fetched ahead into LDS (global_load_lds_dword: bit-exact, 1.14 -> 1.20 s), and this round the
// next pass's operands fetched ahead into registers by loads the compiler does not see (code words into the registers of the words
// just used, no wait in the common pass but the counted ones; needs three wavefronts per SIMD, at four the register allocator's
// spill reloads in the rare passes put a full wait into the common one): 933 ms against 930 -- latency is not what is left; for
// 16 blocks (a quarter of the chip's wavefront slots) it is, 50 -> 39 ms, not shipped: a register a pending load writes must not
// be copied before its wait, and the compiler placed such copies (one parity failure in five runs).
// Tried earlier and dropped (1024 blocks: 1.60 s then): unguarded steps with the row's last tile computed in full, tiles of 16 or
// 24 end codons, 3 or 5 wavefronts per SIMD, operands fetched one sequence ahead in registers (51 spilled registers).
// It is the slow, complete path for rare inputs; the division is the plain '/'.
#pragma once
#include "rc_null_kernel.h"
#include "rc_null_tiled.h"

namespace rc {

// layout of an item's codes, and behind them of a DP workgroup's scratch (bytes), for a class whose largest block has N rows, L reference residues, nn tree nodes
struct GenericLayout {
  size_t offCodes;   // u32 [6][NK][nW][64]    sigma codes x 4 per strand x frame and sequence, four consecutive codons per word (byte j & 3)
  size_t offState;   // f32 [2][1 + 3 NK][64]  DP states of every sequence at the last tile boundary, for the two rows of a pass (row 0 of each: spare)
  size_t offRow;     // f32 [L/3 + 8][64]      the second row's S values, kept until the first row has gone through the scan
  size_t total;
  int nW;            // code words per (strand x frame, sequence): a tile reads eight words from any codon on, so 31 codons of padding
  __host__ __device__ GenericLayout(int N, int L, int nn) {
    const size_t NK = static_cast<size_t>(N) - 1;
    nW = ((L / 3 + 31) >> 2) + 1;
    offCodes = 0;
    offState = (offCodes + 6 * NK * static_cast<size_t>(nW) * kWave * sizeof(uint32_t) + 255) & ~static_cast<size_t>(255);
    (void)nn;
    offRow = (offState + 2 * (3 * NK + 1) * kWave * sizeof(float) + 255) & ~static_cast<size_t>(255);
    total = (offRow + (static_cast<size_t>(L) / 3 + 8) * kWave * sizeof(float) + 255) & ~static_cast<size_t>(255);
  }
};

// two launches: bytes of an item's codes, and of a DP workgroup's states + row buffer
__host__ __device__ inline size_t generic_codes_bytes(int N, int L, int nn) { return GenericLayout(N, L, nn).offState; }
__host__ __device__ inline size_t generic_state_bytes(int N, int L, int nn) { const GenericLayout l(N, L, nn); return l.total - l.offState; }

constexpr int kGenericTile = 32;
#ifndef RC_GENERIC_WAVES
#define RC_GENERIC_WAVES 4   // wavefronts per SIMD of k_generic_dp
#endif

// The cell sums of a tile as distinct scalar members, not an array: an array becomes one 32-register tuple, and every join of the
// three kinds of pass then copies all of it (the same lesson as KRegs in rc_null_kernel.h).
template <int U> struct CellSum { float v; };
template <typename Seq> struct CellSums;
template <int... U> struct CellSums<std::integer_sequence<int, U...>> : CellSum<U>... {
  template <int I> __device__ __forceinline__ float &at() { return static_cast<CellSum<I> &>(*this).v; }
};

// Eight cells of one sequence.  The look-ups first -- the sequence's 64-entry sigma table is spread over the lanes in `lut`, a code
// is stored times four, one per byte, and ds_bpermute_b32 reads address bits 2..7 only (tools/mb_bpermute_addr.hip), so the code
// word itself, or the word shifted right by 8, 16, 24, is the address -- then the cells; m0..m7 receive max(s0, s1, s2) of each
// cell.  Written out because the compiler, given the same in C++, waits for every look-up right behind its issue.  The sums are
// not touched here: every kind of pass hands its m values to ONE add8 behind the join -- with an update of the sums in each kind
// the register allocator keeps two sets of sums and copies all of them at both ends of every pass.
#define RC_LOOKUPS8                                  \
  "ds_bpermute_b32 %[m0], %[wa], %[lut]\n\t"         \
  "v_lshrrev_b32 %[m1], 8, %[wa]\n\t"                \
  "v_lshrrev_b32 %[m2], 16, %[wa]\n\t"               \
  "ds_bpermute_b32 %[m1], %[m1], %[lut]\n\t"         \
  "v_lshrrev_b32 %[m3], 24, %[wa]\n\t"               \
  "ds_bpermute_b32 %[m2], %[m2], %[lut]\n\t"         \
  "v_lshrrev_b32 %[m5], 8, %[wb]\n\t"                \
  "ds_bpermute_b32 %[m3], %[m3], %[lut]\n\t"         \
  "v_lshrrev_b32 %[m6], 16, %[wb]\n\t"               \
  "ds_bpermute_b32 %[m4], %[wb], %[lut]\n\t"         \
  "v_lshrrev_b32 %[m7], 24, %[wb]\n\t"               \
  "ds_bpermute_b32 %[m5], %[m5], %[lut]\n\t"         \
  "ds_bpermute_b32 %[m6], %[m6], %[lut]\n\t"         \
  "ds_bpermute_b32 %[m7], %[m7], %[lut]\n\t"
#define RC_M_OUT [m0] "=&v"(m0), [m1] "=&v"(m1), [m2] "=&v"(m2), [m3] "=&v"(m3), [m4] "=&v"(m4), [m5] "=&v"(m5), [m6] "=&v"(m6), [m7] "=&v"(m7)
#define RC_WAIT4 "s_waitcnt lgkmcnt(4)\n\t"
#define RC_WAIT0 "s_waitcnt lgkmcnt(0)\n\t"

// a sequence that has met no frame shift in this row: s1 == s2 == w, the row's running sum of omega (the same number in every
// lane and for every such sequence; kept in a VGPR and advanced here -- a table of it in SGPRs costs more than the adds)
// (o0..o3: what the first four cells add to w -- omega, or zero for the cells before the row's start in its first tile)
__device__ __forceinline__ void shared_cells8(float &s0, float &w, float &m0, float &m1, float &m2, float &m3, float &m4, float &m5, float &m6,
                                              float &m7, uint32_t wa, uint32_t wb, int lut, float omega, float o0, float o1, float o2, float o3) {
#define RC_CELL(N, OM) \
  "v_add_f32 %[s0], %[s0], %[m" #N "]\n\t"      \
  "v_add_f32 %[w], %[" #OM "], %[w]\n\t"        \
  "v_max_f32 %[m" #N "], %[w], %[s0]\n\t"
  asm volatile(RC_LOOKUPS8 RC_WAIT4 RC_CELL(0, o0) RC_CELL(1, o1) RC_CELL(2, o2) RC_CELL(3, o3) RC_WAIT0 RC_CELL(4, om) RC_CELL(5, om)
               RC_CELL(6, om) RC_CELL(7, om)
               : [s0] "+v"(s0), [w] "+v"(w), RC_M_OUT
               : [wa] "v"(wa), [wb] "v"(wb), [lut] "v"(lut), [om] "s"(omega), [o0] "s"(o0), [o1] "s"(o1), [o2] "s"(o2), [o3] "s"(o3));
#undef RC_CELL
}

// two rows a, a + 1 of such a sequence on one set of look-ups (sigma of cell j is the same for every row): row a + 1 is one cell
// behind in w -- wx holds row a's w on entry and on exit, row a + 1 uses the value of the cell before
__device__ __forceinline__ void dual_cells8(float &sa, float &sb, float &wx, float &m0, float &m1, float &m2, float &m3, float &m4, float &m5,
                                            float &m6, float &m7, float &n0, float &n1, float &n2, float &n3, float &n4, float &n5, float &n6,
                                            float &n7, uint32_t wa, uint32_t wb, int lut, float omega, float o0, float o1, float o2, float o3) {
  float wy;
#define RC_CELL(N, NEW, OLD, OM) \
  "v_add_f32 %[sa], %[sa], %[m" #N "]\n\t"      \
  "v_add_f32 %[sb], %[sb], %[m" #N "]\n\t"      \
  "v_add_f32 %[" #NEW "], %[" #OM "], %[" #OLD "]\n\t" \
  "v_max_f32 %[m" #N "], %[" #NEW "], %[sa]\n\t" \
  "v_max_f32 %[n" #N "], %[" #OLD "], %[sb]\n\t"
  asm volatile(RC_LOOKUPS8 RC_WAIT4 RC_CELL(0, wy, wx, o0) RC_CELL(1, wx, wy, o1) RC_CELL(2, wy, wx, o2) RC_CELL(3, wx, wy, o3) RC_WAIT0
               RC_CELL(4, wy, wx, om) RC_CELL(5, wx, wy, om) RC_CELL(6, wy, wx, om) RC_CELL(7, wx, wy, om)
               : [sa] "+v"(sa), [sb] "+v"(sb), [wx] "+v"(wx), [wy] "=&v"(wy), RC_M_OUT, [n0] "=&v"(n0), [n1] "=&v"(n1), [n2] "=&v"(n2),
                 [n3] "=&v"(n3), [n4] "=&v"(n4), [n5] "=&v"(n5), [n6] "=&v"(n6), [n7] "=&v"(n7)
               : [wa] "v"(wa), [wb] "v"(wb), [lut] "v"(lut), [om] "s"(omega), [o0] "s"(o0), [o1] "s"(o1), [o2] "s"(o2), [o3] "s"(o3));
#undef RC_CELL
}

// a sequence that has met a frame shift earlier in the row but has none in these cells: three states
__device__ __forceinline__ void general_cells8(float &s0, float &s1, float &s2, float &m0, float &m1, float &m2, float &m3, float &m4, float &m5,
                                               float &m6, float &m7, uint32_t wa, uint32_t wb, int lut, float omega) {
#define RC_CELL(N) \
  "v_add_f32 %[s0], %[s0], %[m" #N "]\n\t"      \
  "v_add_f32 %[s1], %[om], %[s1]\n\t"           \
  "v_add_f32 %[s2], %[om], %[s2]\n\t"           \
  "v_max3_f32 %[m" #N "], %[s0], %[s1], %[s2]\n\t"
  asm volatile(RC_LOOKUPS8 RC_WAIT4 RC_CELL(0) RC_CELL(1) RC_CELL(2) RC_CELL(3) RC_WAIT0 RC_CELL(4) RC_CELL(5) RC_CELL(6) RC_CELL(7)
               : [s0] "+v"(s0), [s1] "+v"(s1), [s2] "+v"(s2), RC_M_OUT
               : [wa] "v"(wa), [wb] "v"(wb), [lut] "v"(lut), [om] "s"(omega));
#undef RC_CELL
}

// only the look-ups: the cells of a tile in which the sequence does have a frame shift are compiler-made
__device__ __forceinline__ void lookup8(uint32_t wa, uint32_t wb, int lut, float &m0, float &m1, float &m2, float &m3, float &m4, float &m5, float &m6, float &m7) {
  asm volatile(RC_LOOKUPS8 RC_WAIT0 : RC_M_OUT : [wa] "v"(wa), [wb] "v"(wb), [lut] "v"(lut));
}
#undef RC_LOOKUPS8
#undef RC_M_OUT
#undef RC_WAIT4
#undef RC_WAIT0

// the one place where the cell sums of a tile change (score.c:512: sum += m)
__device__ __forceinline__ void add8(float &c0, float &c1, float &c2, float &c3, float &c4, float &c5, float &c6, float &c7,
                                     float m0, float m1, float m2, float m3, float m4, float m5, float m6, float m7) {
  asm volatile("v_add_f32 %0, %0, %8\n\tv_add_f32 %1, %1, %9\n\tv_add_f32 %2, %2, %10\n\tv_add_f32 %3, %3, %11\n\t"
               "v_add_f32 %4, %4, %12\n\tv_add_f32 %5, %5, %13\n\tv_add_f32 %6, %6, %14\n\tv_add_f32 %7, %7, %15\n\t"
               : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7)
               : "v"(m0), "v"(m1), "v"(m2), "v"(m3), "v"(m4), "v"(m5), "v"(m6), "v"(m7));
}

// LDS of one workgroup: the node states of the site being simulated, 2 bits each, eight nodes per 16-bit word [ceil(nn/8)][64] (rc_device.h,
// node_ctrl), then the last three states of every row, 6 bits in a byte [N][64]
__host__ __device__ inline size_t generic_lds_bytes(int N, int nn) {
  return (static_cast<size_t>((nn + 15) >> 4) * sizeof(uint32_t) + static_cast<size_t>(N)) * kWave;
}

// Two launches per round of items: k_generic_sim (SIM) writes the sigma codes of item u = (block of the round) x groups + (group - gLo) to
// codesAll + u * codesStride, k_generic_dp reads them there -- an item whole, or, in a batch of so few items that each is split into its six
// strand x frame parts to fill the chip, one part per workgroup.  (Until round 5 large batches ran both phases in one kernel: wavefronts
// in the simulation and in the DP side by side on a CU cost more than the two in turn -- 1024 blocks of 100 x 300: 972 ms against 117 ms
// + 727 ms alone, 928 ms as two launches per round of 4096 items; 500 blocks of 200 x 90: 225 -> 189 ms; 40 blocks of 500 x 60: 63.5 -> 45.1.)
// TILED (with SIM): the codes in k_tiled_dp's layout (rc_null_tiled.h) instead, for blocks of up to 64 rows.
template <bool SIM, bool TILED = false>
__device__ __forceinline__ void generic_body(const NullArgs &A, const uint8_t *__restrict__ blob, const DevBlock *__restrict__ dblocks,
                                             const int *__restrict__ classBlocks, const uint32_t *__restrict__ flags,
                                             const uint32_t *__restrict__ Ustream, const uint8_t *__restrict__ pairTab,
                                             uint8_t *__restrict__ scratch, float *__restrict__ maxima, uint32_t *ldsGeneric) {
  const int lane = threadIdx.x;
  uint8_t *__restrict__ mine = scratch + static_cast<size_t>(blockIdx.x) * A.scratchStride;   // scratchStride in bytes here
  unsigned int xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  const int home = static_cast<int>(xcc & 7u);
  const int nBlocks = A.nClassBlocks;
  const int parts = (A.comboSplit && !SIM) ? 6 : 1;   // few items: one strand x frame part each (maxima meet in an atomic max), as in k_null
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
    const int onlyCombo = (A.comboSplit && !SIM) ? sub % parts : -1;
    const DevBlock *__restrict__ db = dblocks + bi;
    const bool nanSem = !SIM && (flags[bi] & kFlagNan) != 0u;   // NaN score tables: every cell through the written-out recurrence with the reference's MAX macro (rc_null_kernel.h, ref_max)
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
    uint8_t *__restrict__ codes = A.codesAll + (static_cast<size_t>(t / perBlock) * (A.gHi - A.gLo) + static_cast<size_t>(grp - A.gLo)) * A.codesStride;
    uint8_t *win = reinterpret_cast<uint8_t *>(ldsGeneric + static_cast<size_t>((nnodes + 15) >> 4) * kWave) + lane;   // row r: [r * 64]
    float *__restrict__ dp = reinterpret_cast<float *>(mine) + kWave;   // (the DP's scratch: the states and the row buffer; the row in front: see the parked s0 below)
    const int sidx = grp * kWave + lane;

    // ---- simulate the L reference-residue columns (evolve.c:416-433) and emit sigma codes for both strands
    // The CU's one scalar unit and the latency of dependent loads are what this phase costs, so nothing here is derived per node or per
    // row by scalar arithmetic: a node's sixteen thresholds come as ONE coalesced word per lane for four nodes (lane = node x parent
    // state x threshold) and reach the lane through four ds_bpermutes addressed by the parent's state; where the parent's state is
    // comes ready-made from the host (NodeRec::gctrl); the draws and thresholds of the next eight nodes are fetched while these eight
    // are worked; a row's tip position and its two codon masks are computed by the lane of that row and read with v_readlane.
    if constexpr (SIM) {
    unsigned int clamped = 0;
    const uint32_t *__restrict__ nodew = reinterpret_cast<const uint32_t *>(nodes);   // a NodeRec as 20 words
    const uint32_t *__restrict__ refw = reinterpret_cast<const uint32_t *>(refcol);
    uint8_t *ldsBytes = reinterpret_cast<uint8_t *>(ldsGeneric);
    const uint32_t laneSt = static_cast<uint32_t>(lane) * 2u;   // node states: 16-bit word g of lane l at g * 128 + 2 l
    const int thrLane = lane & 15, thrNode = lane >> 4;
    const size_t qstride = static_cast<size_t>(cols) * A.Spad;
    const int lastNode = nnodes - 1;
    for (int r = 0; r < N; r++) win[r * kWave] = 0;
    for (int i = 1; i <= L; i++) {
      const uint32_t col = (refw[i >> 1] >> (16 * (i & 1))) & 0xffffu;
      const uint32_t *__restrict__ up = Ustream + (static_cast<size_t>(col) * A.Spad + sidx);
      uint32_t uu[8], thrA, thrB, ctlV;
      auto fetch = [&](int q0, uint32_t (&u)[8], uint32_t &ta, uint32_t &tb, uint32_t &cv) {
        if (q0 + 8 <= nnodes) {
#pragma unroll
          for (int t = 0; t < 8; t++) { u[t] = *up; up += qstride; }
        } else {
#pragma unroll
          for (int t = 0; t < 8; t++) {
            u[t] = 0u;
            if (q0 + t < nnodes) { u[t] = *up; up += qstride; }
          }
        }
        const int qa = q0 + thrNode < lastNode ? q0 + thrNode : lastNode, qb = q0 + 4 + thrNode < lastNode ? q0 + 4 + thrNode : lastNode;
        ta = nodew[qa * 20 + thrLane];
        tb = nodew[qb * 20 + thrLane];
        const int qc = q0 + (lane & 7) < lastNode ? q0 + (lane & 7) : lastNode;
        cv = nodew[qc * 20 + 16 + ((lane >> 3) & 3)];   // lane t: node q0 + t's basepack, lane 16 + t: its gctrl (scalar loads share the counter of the LDS look-ups)
      };
      fetch(0, uu, thrA, thrB, ctlV);
      for (int q0 = 0; q0 < nnodes; q0 += 8) {
        uint32_t un[8], tnA = 0u, tnB = 0u, cnV = 0u;
        if (q0 + 8 < nnodes) fetch(q0 + 8, un, tnA, tnB, cnV);
        const bool full = q0 + 8 <= nnodes;
        uint32_t cur = 0u;   // the states of this group of eight
        auto node = [&](auto tc) {
          constexpr int t = decltype(tc)::value;
          const uint32_t bp = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(ctlV), t));
          const uint32_t ctl = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(ctlV), 16 + t));
          uint32_t pw = cur;
          if (static_cast<int>(ctl) >= 0) pw = *reinterpret_cast<const uint16_t *>(ldsBytes + ((ctl & 0x7f80u) | laneSt));
          const uint32_t ps = (pw >> (ctl & 31u)) & 3u;   // (bit 4 of a control word is zero)   // (the root: its own group, nothing set yet: 0, and its four rows are the same)
          const int at = static_cast<int>(ps << 4) + (t & 3) * 64;
          const int tv = static_cast<int>(t < 4 ? thrA : thrB);
          const uint32_t thx = static_cast<uint32_t>(__builtin_amdgcn_ds_bpermute(at, tv)), thy = static_cast<uint32_t>(__builtin_amdgcn_ds_bpermute(at + 4, tv)),
                         thz = static_cast<uint32_t>(__builtin_amdgcn_ds_bpermute(at + 8, tv)), thw = static_cast<uint32_t>(__builtin_amdgcn_ds_bpermute(at + 12, tv));
          const uint32_t u = uu[t];
          const uint32_t st = (u > thx) + (u > thy) + (u > thz) + ((bp >> (2 * ps)) & 3u);
          clamped += (u > thw);
          cur |= (st & 3u) << (2 * t);
        };
        if (full) {
          static_for<8>(node);
        } else {
          static_for<8>([&](auto tc) { if (q0 + decltype(tc)::value < nnodes) node(tc); });
        }
        *reinterpret_cast<uint16_t *>(ldsBytes + (static_cast<uint32_t>(q0 >> 3) * 128u + laneSt)) = static_cast<uint16_t>(cur);
#pragma unroll
        for (int t = 0; t < 8; t++) uu[t] = un[t];
        thrA = tnA; thrB = tnB; ctlV = cnV;
      }
      // the rows: the codon window of each takes in the new state; from the third column on a window is a codon, and with the
      // reference row's it gives the sigma code of this column's codon on either strand
      const int ip = L + 3 - i;  // reverse-strand position whose codon is these three columns
      const int jF = i / 3 - 1, jR = ip / 3 - 1;   // column i closes codon j = i/3 - 1 of frame i % 3 (i = 3j + 3 + f)
      const uint32_t offF = (static_cast<uint32_t>(0 + i % 3) * NK * lay.nW + static_cast<uint32_t>(jF >> 2)) * (kWave * 4) + lane * 4 + (jF & 3);
      const uint32_t offR = (static_cast<uint32_t>(3 + ip % 3) * NK * lay.nW + static_cast<uint32_t>(jR >> 2)) * (kWave * 4) + lane * 4 + (jR & 3);
      const uint32_t *__restrict__ mrowF = maskw + static_cast<size_t>(0 * L1 + i) * NCW, *__restrict__ mrowR = maskw + static_cast<size_t>(1 * L1 + (i >= 3 ? ip : L)) * NCW;
      const bool emit = i >= 3;
      uint32_t aF, aR;
      {
        const uint32_t c = node_ctrl(qtip[0]);
        const uint32_t wv = *reinterpret_cast<const uint16_t *>(ldsBytes + ((c & 0x7f80u) | laneSt));
        aF = ((static_cast<uint32_t>(win[0]) << 2) | ((wv >> (c & 31u)) & 3u)) & 63u;
        win[0] = static_cast<uint8_t>(aF);
        aR = codon_flip(aF) ^ 63u;
      }
      const uint32_t aF64 = aF * 64u, aR64 = aR * 64u;
      uint8_t *__restrict__ kp = codes;   // sequence k's words of strand x frame 0, word 0
      const size_t kstep = static_cast<size_t>(lay.nW) * (kWave * 4);
      // (TILED: a chunk of sequences is a whole number of tiles, so that no word of the tiled layout straddles two chunks)
      const int kChunk = TILED ? (kWave / A.tiledKT) * A.tiledKT : kWave;
      for (int k0 = 0; k0 < NK; k0 += kChunk) {
        // lane l works out what sequence k0 + l needs: the word and the bits of its tip's state, its two masks
        const int kl = k0 + lane < NK ? k0 + lane : NK - 1;
        const uint32_t tctl = node_ctrl(qtip[kl + 1]);
        const uint32_t k5 = static_cast<uint32_t>(kl) / 5u, sh = 6u * (static_cast<uint32_t>(kl) - 5u * k5);
        const uint32_t mFv = (mrowF[k5] >> sh) & 63u, mRv = (mrowR[k5] >> sh) & 63u;
        const int nk = NK - k0 < kWave ? NK - k0 : kWave;
        if constexpr (TILED) {
          // A word of the tiled layout holds up to five sequences of one tile: window update as below, then
          // the codes of the word's sequences packed, field c at bits [6c + 7 : 6c + 2], one dword per strand
          const TiledLayout tl(NK, A.tiledKT);
          uint32_t *__restrict__ cwp = reinterpret_cast<uint32_t *>(codes) + lane;
          const size_t posF = static_cast<size_t>(0 * L1 + i) * tl.WT * kWave, posR = static_cast<size_t>(1 * L1 + (emit ? ip : L)) * tl.WT * kWave;
          auto rowsT = [&](auto cnt, int kk, int widx) {   // sequences kk .. kk + CNT - 1 = the fields of word widx
            constexpr int CNT = decltype(cnt)::value;
            uint32_t bb[CNT], cc[CNT], wv[CNT];
            uint8_t *wrow = win + (kk + 1) * kWave;
#pragma unroll
            for (int t = 0; t < CNT; t++) {
              cc[t] = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(tctl), kk - k0 + t));
              wv[t] = *reinterpret_cast<const uint16_t *>(ldsBytes + ((cc[t] & 0x7f80u) | laneSt));
              bb[t] = wrow[t * kWave];
            }
#pragma unroll
            for (int t = 0; t < CNT; t++) bb[t] = ((bb[t] << 2) | ((wv[t] >> (cc[t] & 31u)) & 3u)) & 63u;
#pragma unroll
            for (int t = 0; t < CNT; t++) wrow[t * kWave] = static_cast<uint8_t>(bb[t]);
            if (emit) {
              uint32_t wf = 0u, wr = 0u;
#pragma unroll
              for (int t = 0; t < CNT; t++) {
                const uint32_t mF = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(mFv), kk - k0 + t));
                const uint32_t mR = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(mRv), kk - k0 + t));
                wf |= static_cast<uint32_t>(pairTab[aF64 + (bb[t] & mF)]) << (6 * t + 2);
                wr |= static_cast<uint32_t>(pairTab[aR64 + ((codon_flip(bb[t]) ^ 63u) & mR)]) << (6 * t + 2);
              }
              cwp[posF + static_cast<size_t>(widx) * kWave] = wf;
              cwp[posR + static_cast<size_t>(widx) * kWave] = wr;
            }
          };
          const int t0 = k0 / tl.KT, t1 = t0 + kChunk / tl.KT < tl.T ? t0 + kChunk / tl.KT : tl.T;   // this chunk's tiles
          for (int t = t0; t < t1; t++)
            for (int x = 0; x < tl.WPT; x++) {
              const int kk = t * tl.KT + 5 * x;
              int cnt = tl.KT - 5 * x;
              cnt = cnt > 5 ? 5 : cnt;
              cnt = cnt > NK - kk ? NK - kk : cnt;
              const int widx = t * tl.WPT + x;
              if (cnt >= 5) rowsT(std::integral_constant<int, 5>{}, kk, widx);
              else if (cnt == 4) rowsT(std::integral_constant<int, 4>{}, kk, widx);
              else if (cnt == 3) rowsT(std::integral_constant<int, 3>{}, kk, widx);
              else if (cnt == 2) rowsT(std::integral_constant<int, 2>{}, kk, widx);
              else if (cnt == 1) rowsT(std::integral_constant<int, 1>{}, kk, widx);
            }
          continue;
        }
        auto rows = [&](auto cnt, int kk) {   // sequences k0 + kk .. + CNT - 1
          constexpr int CNT = decltype(cnt)::value;
          uint32_t bb[CNT], pf[CNT], pr[CNT];
          uint8_t *wrow = win + (k0 + kk + 1) * kWave;
          uint32_t cc[CNT], wv[CNT];
#pragma unroll
          for (int t = 0; t < CNT; t++) {   // (all the reads in front of the writes: the compiler cannot tell the two LDS arrays apart)
            cc[t] = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(tctl), kk + t));
            wv[t] = *reinterpret_cast<const uint16_t *>(ldsBytes + ((cc[t] & 0x7f80u) | laneSt));
            bb[t] = wrow[t * kWave];
          }
#pragma unroll
          for (int t = 0; t < CNT; t++) bb[t] = ((bb[t] << 2) | ((wv[t] >> (cc[t] & 31u)) & 3u)) & 63u;
#pragma unroll
          for (int t = 0; t < CNT; t++) wrow[t * kWave] = static_cast<uint8_t>(bb[t]);
          if (emit) {
#pragma unroll
            for (int t = 0; t < CNT; t++) {
              const uint32_t mF = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(mFv), kk + t));
              const uint32_t mR = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(mRv), kk + t));
              pf[t] = pairTab[aF64 + (bb[t] & mF)];
              pr[t] = pairTab[aR64 + ((codon_flip(bb[t]) ^ 63u) & mR)];
            }
#pragma unroll
            for (int t = 0; t < CNT; t++) {
              kp[offF] = static_cast<uint8_t>(pf[t] << 2);
              kp[offR] = static_cast<uint8_t>(pr[t] << 2);
              kp += kstep;
            }
          }
        };
        int kk = 0;
        for (; kk + 8 <= nk; kk += 8) rows(std::integral_constant<int, 8>{}, kk);
        for (; kk < nk; kk++) rows(std::integral_constant<int, 1>{}, kk);
      }
    }
    if (clamped && sidx < A.sampleN) atomicAdd(A.clampCount, static_cast<unsigned long long>(clamped));
    }
    if constexpr (SIM) continue;

    // ---- both strands x 3 frames: DP over (start a, end j) + serial HSS scan (score.c:496-535, 830-845, 888-962)
    // A sequence that has met no frame shift since the row's start has s1 == s2 == the row's running sum of omega, the same
    // number for every such sequence and every sample: W[3 + t] after t cells (k_prep_lut, off_chain).  Its pass keeps s0 only,
    // a cell is add / max against the wave-uniform W / add to the cell's sum, and only s0 is parked between tiles.  The first
    // tile in which a sequence meets a frame shift, and every later tile of that row, takes the three-state pass.
    // Rows go in pairs a, a + 1 (a even: both start in the same code word): where a sequence is in the first state in both, the
    // two rows share its look-ups and its code words -- half the ds_bpermutes and half the code traffic, which are what bound the
    // pass.  getHSS reads S row by row: row a's entries go through the scan as they are made, row a + 1's wait in `rowbuf`.
    float best = -1.0f;
    const float negTie = -A.tieThr;
    const float *__restrict__ chainW = reinterpret_cast<const float *>(blob + db->off_chain);
    const uint32_t zeroCodes = (static_cast<uint32_t>(db->code_zero) << 2) * 0x01010101u;   // four copies of the code whose sigma is 0
    float *__restrict__ rowbuf = reinterpret_cast<float *>(mine + (lay.offRow - lay.offState)) + lane;
    float *__restrict__ dpB = dp + static_cast<size_t>(3 * NK + 1) * kWave;
    for (int s = 0; s < 2; s++)
      for (int f = 0; f < 3; f++) {
        const int sites = (L - f) / 3;
        if (sites <= 0) continue;
        if (onlyCombo >= 0 && onlyCombo != s * 3 + f) continue;
        SampleScan st{0.0f, 0.0f, 0u, 0u};
        for (int a = 0; a < sites; a += 2) {
          const int b = a + 1;   // the second row of the pair; b == sites: there is none
          sample_scan_row_begin(st, best, static_cast<uint32_t>(a));
          const int jendA = (a == sites - 1) ? sites - 1 : sites;   // the frame's final entry is never reported (score.c:892-900)
          const int jendB = (b >= sites - 1) ? sites - 1 : sites;   // (b == sites: no cell passes b <= j < sites - 1)
          // bit k & 31 of lane k >> 5: sequence k has met a frame shift in this row (a block with NaN tables: every sequence, from the start -- no pass
          // of it is ever the common one, at no cost to the common pass's decision)
          uint32_t stickyA = nanSem ? 0xffffffffu : 0u, stickyB = stickyA;
          // tiles start at multiples of four codons (the code words hold four), so a row's first tile may begin before a
          for (int j0 = a & ~3; j0 < jendA; j0 += kGenericTile) {
            const bool first = j0 <= a, more = j0 + kGenericTile < jendA;
            const int loA = first ? a - j0 : 0, hiA = jendA - j0 < kGenericTile ? jendA - j0 : kGenericTile;
            const int loB = first ? b - j0 : 0, hiB = jendB - j0 < kGenericTile ? (jendB - j0 > 0 ? jendB - j0 : 0) : kGenericTile;
            const uint32_t validA = (hiA == kGenericTile ? 0xffffffffu : (1u << hiA) - 1u) & ~((1u << loA) - 1u);   // bit u: a <= j0 + u < jendA
            const uint32_t validB = hiB > loB ? (hiB == kGenericTile ? 0xffffffffu : (1u << hiB) - 1u) & ~((1u << loB) - 1u) : 0u;
            if (validA == 0u) break;   // the frame's last row: its only cell is the unreported final entry
            const int ng = (hiA + 7) >> 3;   // groups of eight cells; the cells of the last group past the row's end are computed and dropped
            CellSums<std::make_integer_sequence<int, kGenericTile>> sumA, sumB;
            static_for<kGenericTile>([&](auto uc) {   // score.c:509: sum = 0, then += in the order k = 0, 1, ...
              sumA.template at<decltype(uc)::value>() = 0.0f;
              sumB.template at<decltype(uc)::value>() = 0.0f;
            });
            const float wInA = chainW[j0 - a + 3], wInB = chainW[j0 - a + 2];   // the rows' running sums of omega before the tile's first cell
            float oA[4], oB[4];   // what the tile's first four cells add to them: nothing before the row's start
#pragma unroll
            for (int u = 0; u < 4; u++) { oA[u] = u < loA ? 0.0f : omega; oB[u] = u < loB ? 0.0f : omega; }
            const int jl = (j0 + (lane & (kGenericTile - 1)) < jendA) ? j0 + (lane & (kGenericTile - 1)) : jendA - 1;   // this lane's end codon for the z fetch
            // The s0 values a pass parks are stored behind the NEXT pass's loads: vmcnt counts in order, so a wait for loads issued
            // behind a store is also a wait for that store's round trip.
            // Every pass stores (a wait that has to serve a path with the stores and one without is a wait for everything): the first
            // one of a tile, and all of a row's last tile, into the spare row in front of the states.
            float heldA = 0.0f, heldB = 0.0f;
            const int heldStride = more ? kWave : 0;
            for (int k = 0; k < NK; k++) {
              const int lutv = __float_as_int(lut[(static_cast<size_t>(s) * NK + k) * kLutSize + lane]);
              const uint32_t *__restrict__ ck = reinterpret_cast<const uint32_t *>(codes) + ((static_cast<size_t>(s * 3 + f) * NK + k) * lay.nW + (j0 >> 2)) * kWave + lane;
              float *__restrict__ dkA = dp + static_cast<size_t>(k) * kWave + lane;
              float *__restrict__ dkB = dpB + static_cast<size_t>(k) * kWave + lane;
              // only the words of the groups of eight cells this tile has (ng of four): a row's last tile is half empty on average, and every
              // byte of these words comes from HBM -- rows of 50 codons (150 columns) had 41 % of their code words fetched for nothing
              uint32_t cw[kGenericTile / 4];
              cw[0] = ck[0];
              cw[1] = ck[kWave];
#pragma unroll
              for (int g = 1; g < kGenericTile / 8; g++) {
                if (g < ng) { cw[2 * g] = ck[(2 * g) * kWave]; cw[2 * g + 1] = ck[(2 * g + 1) * kWave]; }
              }
              // z of sequence k at the tile's end codons (score.c:506-533): lane u fetches the half word of codon j0 + u that holds it,
              // two ballots turn the 2-bit values into wave-uniform masks -- almost always both zero (a sequence has few frame shifts)
              // (fetched behind the code words: it is needed first, and a wait for it in front of their issue would be two round trips)
              const uint32_t zhalf = reinterpret_cast<const uint32_t *>(zw)[((static_cast<size_t>(s * L1 + 3 + f) + static_cast<size_t>(3 * jl)) * ZW + (k >> 5)) * 2 + ((k >> 4) & 1)];
              const uint32_t zcl = (zhalf >> (2 * (k & 15))) & 3u;
              const uint32_t zz0 = static_cast<uint32_t>(__ballot(zcl & 1u)), zz1 = static_cast<uint32_t>(__ballot(zcl & 2u));
              const uint32_t swA = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(stickyA), k >> 5));
              const uint32_t swB = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(stickyB), k >> 5));
              // The scalar unit is shared by the CU's four SIMDs (tools/microbench.hip: 1.3 instructions per ns and CU) and is what
              // bounds this loop: the two-row pass decides on ONE word; what the other passes need to know is worked out on their side
              // of the branch (`fb`), behind an empty asm that keeps the compiler from hoisting it in front of the branch.
              const uint32_t bitK = 1u << (k & 31);
              const bool dual = (((zz0 | zz1) & validA) | ((swA | swB) & bitK)) == 0u;   // (validB is a subset of validA)
              uint32_t fb = 0u;   // not dual: bit 0 row a in the first state, bit 1 row a + 1, bit 2 a frame shift of this sequence in row a's cells, bit 3 in row a + 1's
              float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f, b0 = 0.0f, b1 = 0.0f, b2 = 0.0f;
              uint32_t cwA0 = cw[0], cwB0 = cw[0];   // the first word as the two rows see it
              if (__builtin_expect(dual, 1)) {
                if (!first) {
                  a0 = dkA[0];
                  b0 = dkB[0];
                } else {
                  // the cells before the rows' start add sigma = 0, so that s0 stays zero; both rows go on row a's word: row a + 1
                  // starts at minus sigma of cell a, and is exactly zero behind that cell
                  if (loA != 0) cwA0 = (cw[0] & (0xffffffffu << (8 * loA))) | (zeroCodes >> (32 - 8 * loA));
                  b0 = -__int_as_float(__builtin_amdgcn_ds_bpermute(static_cast<int>(cw[0] >> (8 * loA)), lutv));
                }
              } else {
                uint32_t zA = (zz0 | zz1) & validA, zB = (zz0 | zz1) & validB, tA = swA & bitK, tB = swB & bitK;
                asm volatile("" : "+s"(zA), "+s"(zB), "+s"(tA), "+s"(tB));
                const bool evA = zA != 0u || nanSem, evB = zB != 0u || nanSem, wasA = tA != 0u, wasB = tB != 0u, sharedA = !wasA && !evA, sharedB = !wasB && !evB;
                fb = (sharedA ? 1u : 0u) | (sharedB ? 2u : 0u) | (evA ? 4u : 0u) | (evB ? 8u : 0u);
                if (!first) {
                  a0 = dkA[0];
                  b0 = dkB[0];
                  if (!sharedA) {
                    a1 = a2 = wInA;
                    if (wasA) { a1 = dkA[static_cast<size_t>(NK) * kWave]; a2 = dkA[static_cast<size_t>(2 * NK) * kWave]; }
                  }
                  if (!sharedB) {
                    b1 = b2 = wInB;
                    if (wasB) { b1 = dkB[static_cast<size_t>(NK) * kWave]; b2 = dkB[static_cast<size_t>(2 * NK) * kWave]; }
                  }
                } else {
                  // (the three-state pass skips the cells before a row's start instead)
                  if (sharedA && loA != 0) cwA0 = (cw[0] & (0xffffffffu << (8 * loA))) | (zeroCodes >> (32 - 8 * loA));
                  if (sharedB) cwB0 = (cw[0] & (0xffffffffu << (8 * loB))) | (zeroCodes >> (32 - 8 * loB));   // (loB is 1 or 3)
                }
                if (evA && !wasA) stickyA |= (lane == (k >> 5)) ? bitK : 0u;
                if (evB && !wasB) stickyB |= (lane == (k >> 5)) ? bitK : 0u;
              }
              float wA = wInA, wB = wInB;   // (a row in the first state only)
              dp[(k - 1) * heldStride - (kWave - heldStride) + lane] = heldA;
              dpB[(k - 1) * heldStride - (kWave - heldStride) + lane] = heldB;
              static_for<kGenericTile / 8>([&](auto gc) {
                constexpr int g = decltype(gc)::value, u0 = 8 * g;
                if (g == 0 || g < ng) {
                  float m[8], n[8];
                  const uint32_t cA = g == 0 ? cwA0 : cw[2 * g], cB = g == 0 ? cwB0 : cw[2 * g];
                  if (__builtin_expect(dual, 1)) {
                    dual_cells8(a0, b0, wA, m[0], m[1], m[2], m[3], m[4], m[5], m[6], m[7], n[0], n[1], n[2], n[3], n[4], n[5], n[6], n[7], cA,
                                cw[2 * g + 1], lutv, omega, g == 0 ? oA[0] : omega, g == 0 ? oA[1] : omega, g == 0 ? oA[2] : omega,
                                g == 0 ? oA[3] : omega);
                  } else {
                    // one row after the other
                    auto row = [&](bool shared, bool ev, uint32_t valid, uint32_t w0, const float (&o)[4], float &w, float &s0, float &s1, float &s2, float (&mm)[8]) {
                      if (shared) {
                        shared_cells8(s0, w, mm[0], mm[1], mm[2], mm[3], mm[4], mm[5], mm[6], mm[7], w0, cw[2 * g + 1], lutv, omega, g == 0 ? o[0] : omega,
                                      g == 0 ? o[1] : omega, g == 0 ? o[2] : omega, g == 0 ? o[3] : omega);
                      } else if (!ev) {   // (never a row's first tile: that one is here only for an event)
                        general_cells8(s0, s1, s2, mm[0], mm[1], mm[2], mm[3], mm[4], mm[5], mm[6], mm[7], w0, cw[2 * g + 1], lutv, omega);
                      } else {
                        lookup8(w0, cw[2 * g + 1], lutv, mm[0], mm[1], mm[2], mm[3], mm[4], mm[5], mm[6], mm[7]);
                        static_for<8>([&](auto vc) {
                          constexpr int v = decltype(vc)::value, u = u0 + v;
                          const float sg = mm[v];
                          mm[v] = 0.0f;   // a cell outside the row adds zero to a sum that is never read
                          if ((valid >> u) & 1u) {
                            // (the reference's MAX macro for every block that comes here: equal to v_max_f32 where no operand is a NaN, and what a block
                            // with NaN tables needs -- rc_null_kernel.h, ref_max; two instructions per maximum in these few cells)
                            pair_step<true>(static_cast<int>(((zz0 >> u) & 1u) | (((zz1 >> u) & 1u) << 1)), sg, Delta, Omega, omega, s0, s1, s2);
                            mm[v] = ref_max3<true>(s0, s1, s2);
                          }
                        });
                      }
                    };
                    row((fb & 1u) != 0u, (fb & 4u) != 0u, validA, cA, oA, wA, a0, a1, a2, m);
                    row((fb & 2u) != 0u, (fb & 8u) != 0u, validB, cB, oB, wB, b0, b1, b2, n);
                  }
                  add8(sumA.template at<u0>(), sumA.template at<u0 + 1>(), sumA.template at<u0 + 2>(), sumA.template at<u0 + 3>(),
                       sumA.template at<u0 + 4>(), sumA.template at<u0 + 5>(), sumA.template at<u0 + 6>(), sumA.template at<u0 + 7>(), m[0], m[1],
                       m[2], m[3], m[4], m[5], m[6], m[7]);
                  add8(sumB.template at<u0>(), sumB.template at<u0 + 1>(), sumB.template at<u0 + 2>(), sumB.template at<u0 + 3>(),
                       sumB.template at<u0 + 4>(), sumB.template at<u0 + 5>(), sumB.template at<u0 + 6>(), sumB.template at<u0 + 7>(), n[0], n[1],
                       n[2], n[3], n[4], n[5], n[6], n[7]);
                }
              });
              heldA = a0; heldB = b0;   // (stored by the next pass, or behind the loop)
              if (!dual && more) {
                if (!(fb & 1u)) { dkA[static_cast<size_t>(NK) * kWave] = a1; dkA[static_cast<size_t>(2 * NK) * kWave] = a2; }
                if (!(fb & 2u)) { dkB[static_cast<size_t>(NK) * kWave] = b1; dkB[static_cast<size_t>(2 * NK) * kWave] = b2; }
              }
            }
            if (more) { dp[static_cast<size_t>(NK - 1) * kWave + lane] = heldA; dpB[static_cast<size_t>(NK - 1) * kWave + lane] = heldB; }
            static_for<kGenericTile>([&](auto uc) {
              constexpr int u = decltype(uc)::value;
              if ((validA >> u) & 1u) {   // a <= j0 + u < jendA
                const float v = fmaxf(sumA.template at<u>(), Delta) / nkf;
                sample_scan_step(st, v, static_cast<float>(2 * (j0 + u) + 1), negTie);
              }
              if ((validB >> u) & 1u) rowbuf[static_cast<size_t>(j0 + u) * kWave] = fmaxf(sumB.template at<u>(), Delta) / nkf;
            });
          }
          sample_scan_row_end(st, static_cast<uint32_t>(a));
          if (b < sites) {   // row a + 1 through the scan, eight entries fetched at a time
            sample_scan_row_begin(st, best, static_cast<uint32_t>(b));
            for (int j = b; j < jendB; j += 8) {
              float v[8];
#pragma unroll
              for (int t = 0; t < 8; t++) v[t] = rowbuf[static_cast<size_t>(j + t < jendB ? j + t : jendB - 1) * kWave];
#pragma unroll
              for (int t = 0; t < 8; t++)
                if (j + t < jendB) sample_scan_step(st, v[t], static_cast<float>(2 * (j + t) + 1), negTie);
            }
            sample_scan_row_end(st, static_cast<uint32_t>(b));
          }
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

template <bool TILED>
__global__ __launch_bounds__(64) void k_generic_sim(NullArgs A, const uint8_t *__restrict__ blob, const DevBlock *__restrict__ dblocks,
                                                    const int *__restrict__ classBlocks, const uint32_t *__restrict__ flags,
                                                    const uint32_t *__restrict__ Ustream, const uint8_t *__restrict__ pairTab,
                                                    uint8_t *__restrict__ scratch, float *__restrict__ maxima) {
  extern __shared__ uint32_t ldsGeneric[];
  generic_body<true, TILED>(A, blob, dblocks, classBlocks, flags, Ustream, pairTab, scratch, maxima, ldsGeneric);
}
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(RC_GENERIC_WAVES, RC_GENERIC_WAVES))) void k_generic_dp(NullArgs A, const uint8_t *__restrict__ blob, const DevBlock *__restrict__ dblocks,
                                                   const int *__restrict__ classBlocks, const uint32_t *__restrict__ flags,
                                                   const uint32_t *__restrict__ Ustream, const uint8_t *__restrict__ pairTab,
                                                   uint8_t *__restrict__ scratch, float *__restrict__ maxima) {
  extern __shared__ uint32_t ldsGeneric[];
  generic_body<false>(A, blob, dblocks, classBlocks, flags, Ustream, pairTab, scratch, maxima, ldsGeneric);
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
  const bool nanSem = A.flags && (A.flags[bi] & kFlagNan);   // NaN score tables: the reference's MAX macro (rc_null_kernel.h, ref_max)
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
          if (nanSem) pair_step<true>(static_cast<int>((z[k >> 5] >> (2 * (k & 31))) & 3ull), sigma[(static_cast<size_t>(s) * NK + k) * L1 + i], Delta, Omega, omega, s0, s1, s2);
          else pair_step(static_cast<int>((z[k >> 5] >> (2 * (k & 31))) & 3ull), sigma[(static_cast<size_t>(s) * NK + k) * L1 + i], Delta, Omega, omega, s0, s1, s2);
          dp[(0 * static_cast<size_t>(NK) + k) * kWave + lane] = s0;
          dp[(1 * static_cast<size_t>(NK) + k) * kWave + lane] = s1;
          dp[(2 * static_cast<size_t>(NK) + k) * kWave + lane] = s2;
          sum = sum + (nanSem ? ref_max3<true>(s0, s1, s2) : fmaxf(fmaxf(s0, s1), s2));
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
