// pole_kernels.hip -- questions with a row at the pole of the lack term, re-evaluated the reference's way BEHIND the sweep (gfx950).
//
// lack = -sum invD^2 / log2(p) (reference PqaCore/CEEvalQsSubtaskConsider.cpp:117) has a pole at p -> 1.  A quiz a handful of
// consistent answers deep -- where real quizzes end -- has its posterior on one target, p = 1 - 1e-7 ... 1 - 1e-16: there the last
// place of p = l * (1 / W_k), i.e. the ORDER in which W_k was summed, moves log2 p by 1.6e-16 absolute and the priority by
// 1e-16 / (1 - p) relative -- the ninth digit at 1 - 1e-7, the first at 1 - 1e-16 -- and the velocity term (p - prior)^2 (:119-127)
// of that element, a difference of two numbers next to 1, follows with a third of that.  The sweeps sum W_k as a GPU sums (lane
// partials and butterflies); the reference sums four serial Kahan lanes down the row (SRPlatform/Interface/SRAccumVectDbl256.h:
// 40-46) and folds them with PreciseSum (:62-92): T / 4 DEPENDENT steps per row, which no sweep can afford for every row
//  -- but only the rows AT the pole need it.
//
// So every sweep only WATCHES (a compare per row; eval_kernels.hip, cluster_kernels.hip, batch_kernels.hip): a question with a row
// whose largest posterior element is within 2^-10 of 1, or holds a quarter of the row while the answer hardly moves the posterior
// (pole_device.h: kNearOneHi, kQuarterHi, kSmallV -- why there), leaves its sums in memory and an entry in the suspect list, and
// the sweep's finisher, seeing suspects, leaves the publication of the result to this kernel, which is launched behind every
// watching sweep (an empty list: a few hundred threads read one word and leave).  Here a WAVE takes one suspect at a time:
//   * the operands of the listed rows -- the answer rows, mD, the priors, the gap words -- come in chunk by chunk by LDS-DMA,
//     kFixDepth chunks ahead of the one worked on (no register is held across the wait);
//   * the lanes form the likelihoods (A * invD) * prior again in place -- bit for bit the sweep's and the reference's (:72-82) --
//     and keep each row's largest element;
//   * four lanes per row run the reference-order sums, all rows of the question side by side; PreciseSum folds the four;
//   * for the row's largest element: Log2Hot by the reference's own operation sequence (SRVectMath.h:87-135) on
//     p = l * (1 / W_k) with the reference-order W_k, and its entropy, lack and velocity terms replace what pass 2 had added;
//   * the epilogue (:134-207) again, the priority stored (and handed to the host where the sweep hands priorities over).
// The last workgroup to finish then does what the sweep's finisher left undone: the argmax / the reference's selector / the
// flag of the hand-over, and empties the list.
// In a late quiz EVERY question has such rows (the target's likelihood is all of W_k whatever the answer): the fix then re-reads
// the cube once and runs T / 4 dependent steps per question, four questions per CU and SIMD at a time -- one to two more sweeps'
// time; the in-kernel fix of round 4 (rows of up to 4096 targets only, lists of 62 / 128 suspects) cost 4.5 - 6.5.
#include <cmath>

#include "eval_device.h"
#include "pole_device.h"
#include "pqa_device.h"
#include "pqa_kernels.h"

namespace pqa {

static __device__ double gLog2TableP[kLog2TableDoubles];   // this translation unit's copy of the Log2Hot table
static __device__ double gLog2Entry0RefP;                   // its entry 0 as the reference has it (eval_kernels.hip: gLog2Entry0Ref)

hipError_t UploadLog2TablePole(const double *hostTable) {
  const double entry0 = std::log2(1.0 + 0x1p-11) * 9.9999999999999927e-01;   // SRVectMath.cpp:31,42
  const hipError_t e = hipMemcpyToSymbol(HIP_SYMBOL(gLog2Entry0RefP), &entry0, sizeof(double));
  if (e != hipSuccess) return e;
  return hipMemcpyToSymbol(HIP_SYMBOL(gLog2TableP), hostTable, kLog2TableDoubles * sizeof(double));
}

namespace {

#ifndef PQA_FIX_WAVES
#define PQA_FIX_WAVES 4
#endif
#ifndef PQA_FIX_DEPTH
#define PQA_FIX_DEPTH 2
#endif
constexpr int kFixThreads = 256;                   // four waves, each by itself
constexpr int kFixChunk = 2 * kWave;               // targets of a row per LDS chunk: one 16-byte pair per lane
constexpr int kFixRowStride = kFixChunk + 4;       // doubles between the rows of a chunk (the rows' chain lanes on different banks)
constexpr int kFixRed = 12;                        // doubles of scratch per row
constexpr int kFixDepth = PQA_FIX_DEPTH;           // chunks whose operands are in flight (LDS-DMA) ahead of the one worked on

// LDS-DMA (eval_kernels.hip: dma16): 16 (4) bytes per lane from global memory straight into LDS -- the row base in an SGPR descriptor,
// the lane's byte offset in a VGPR, M0 = the wave-uniform LDS byte address; lane i lands at M0 + 16 i (4 i).  Counted by vmcnt,
// invisible to the compiler's own bookkeeping: waited for explicitly (wait_vmcnt).  Reads beyond the descriptor's range return 0.
typedef unsigned int dma_rsrc_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ dma_rsrc_t dma_rsrc(const void *row, int64_t bytes) {
  const uint64_t base = (uint64_t)(uintptr_t)row;
  return dma_rsrc_t{(unsigned)__builtin_amdgcn_readfirstlane((unsigned)base),
                    (unsigned)__builtin_amdgcn_readfirstlane((unsigned)(base >> 32)) & 0xFFFFu,
                    (unsigned)__builtin_amdgcn_readfirstlane((unsigned)bytes), 0x00020000u};
}
__device__ __forceinline__ void dma16(dma_rsrc_t rsrc, unsigned byteOffset, unsigned ldsDst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(byteOffset), "s"(rsrc), "s"(__builtin_amdgcn_readfirstlane(ldsDst)) : "memory");
}
__device__ __forceinline__ void dma4(dma_rsrc_t rsrc, unsigned byteOffset, unsigned ldsDst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dword %1, %2, 0 offen lds\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(byteOffset), "s"(rsrc), "s"(__builtin_amdgcn_readfirstlane(ldsDst)) : "memory");
}
// s_waitcnt vmcnt(n) for a wave-uniform n (the instruction takes an immediate; fewer than asked for is always safe)
__device__ __forceinline__ void wait_vmcnt(int n) {
#define PQA_VM(k) case k: asm volatile("s_waitcnt vmcnt(" #k ")" ::: "memory"); break;
  switch (n) {
    PQA_VM(0) PQA_VM(1) PQA_VM(2) PQA_VM(3) PQA_VM(4) PQA_VM(5) PQA_VM(6) PQA_VM(7) PQA_VM(8) PQA_VM(9) PQA_VM(10) PQA_VM(11) PQA_VM(12)
    PQA_VM(13) PQA_VM(14) PQA_VM(15) PQA_VM(16) PQA_VM(17) PQA_VM(18) PQA_VM(19) PQA_VM(20) PQA_VM(21) PQA_VM(22) PQA_VM(23) PQA_VM(24)
    PQA_VM(25) PQA_VM(26) PQA_VM(27) PQA_VM(28) PQA_VM(29) PQA_VM(30) PQA_VM(31) PQA_VM(32) PQA_VM(33) PQA_VM(34) PQA_VM(35) PQA_VM(36)
    PQA_VM(37) PQA_VM(38) PQA_VM(39) PQA_VM(40) PQA_VM(41) PQA_VM(42) PQA_VM(43) PQA_VM(44) PQA_VM(45) PQA_VM(46) PQA_VM(47) PQA_VM(48)
    default: asm volatile("s_waitcnt vmcnt(48)" ::: "memory"); break;
  }
#undef PQA_VM
}

// log2hot (pqa_device.h) with the table in global memory: the same operations on the same table entries, so the same bits -- what
// pass 2 of the sweep took for an element.  (This kernel's LDS is the rows' chunks; one lane per row needs the function.)
__device__ __forceinline__ double log2hot_global(double x, const double *__restrict__ tbl) {
  const uint32_t lo = (uint32_t)d2u(x);
  const uint32_t hi = (uint32_t)(d2u(x) >> 32);
  const double de = u2d(0x4330000000000000ULL | (uint64_t)(hi >> 20)) - 4503599627371519.0;
  const uint32_t idx = (hi >> 10) & 1023u;
  const double y = tbl[2 * idx], c1 = tbl[2 * idx + 1];
  const uint32_t mhi = (hi & 0x800ffc00u) | 0x3ff00200u;
  const uint32_t zhi = (hi & 0x800fffffu) | 0x3ff00000u;
  const double m = u2d((uint64_t)mhi << 32);
  const double z = u2d(((uint64_t)zhi << 32) | lo);
  const double w = (z - m) * c1;
  double c = fma(w, -0x1.55046a143789p-4, 0x1.47fd3ffac83b4p-3);
  c = fma(w, c, -0x1.62e42fefa39efp-2);
  c = fma(w, c, 1.0);
  const double log2z = fma(w, c, y);
  return log2z + de;
}

struct Best {
  double p;
  int64_t i;
};
__device__ __forceinline__ void best_merge(Best &b, double op, int64_t oi) {
  if (oi >= 0 && (b.i < 0 || op > b.p || (op == b.p && oi < b.i))) { b.p = op; b.i = oi; }
}

// ------------------------------------------------------------------------------------------------------------------
// The gate (PoleFix::gate): where only the ARGMAX leaves the engine, a listed question needs the fix only if it can still win.
// What the fix changes in a question's sums is bounded by how far the two sums of a row can differ and how close to 1 the row's
// largest posterior element is.  With d the relative difference of the reference-order W_k and the sweep's -- the sweep's is a
// pairwise tree over lane sums of at most 2 x 10 terms, depth <= 30 additions of non-negative terms: within 30 u of the exact sum,
// the reference's Kahan lanes within 2 u + O(n u^2); d <= 33 u, taken as kGateD = 4e-15 -- and g a lower bound of 1 - p for that
// element (PoleEntry::gap, from the sweep's watch):
//   * lack (:117): the element's term id^2 / log2 p moves by |log2 pFast - log2 pRef| / |log2 pRef| of itself, and itself is at most the
//     whole (one-signed) lack sum: |log2 pRef| >= (g - 2 d) / ln 2, the difference <= 1.01 d / ln 2 + 3e-17 (the two Log2Hot forms) --
//     per listed row (1.1 d + 3e-17) / (g - 2 d) of the lack sum, hence of the priority (:207);
//   * entropy (:113-114, :175-181): the element's term moves by at most W_k (1.5 d + 2e-17), sum W_k by d of itself: avgH by at most 21.5 d
//     (|avgH| <= log2 of the targets <= 20), 2^(-2 avgH) by 30 d;
//   * velocity (:119-127, :156-177): |sqrt V_new - sqrt V_old| <= |pRef - pFast| (both roots are at least the element's own difference),
//     so W_k sqrt V_k moves by at most 2.6 d W_k, avgV by 4.1 d, ln avgV by 4.1 d / avgV, vComp^9 (:191, :207) by 38 d vComp / avgV.
// The bound of a question is twice the sum (kGateSafety) plus 2e-9 (what the fixed priorities themselves are held to); a question
// whose gap is not known (other sweeps than the register shapes', an element that rounds to 1) has none and is always redone.
// pole_bounds_kernel computes it per entry from the sums the sweep left -- the epilogue's averages again, a thread per entry -- stores
// it in the entry and raises PoleHeader::floorBits to the best LOWER bound priority x (1 - bound) over the listed questions (positive
// doubles order as their bit patterns).  The fix then skips every entry whose UPPER bound priority x (1 + bound) is below that floor:
// the question that is the reference's maximum is never skipped (its upper bound is at least its reference priority, which is at
// least every other question's, which is at least the floor), and what the final argmax compares are fixed priorities of the
// questions that could win and untouched ones that could not.
// ------------------------------------------------------------------------------------------------------------------
constexpr double kGateD = 4e-15, kGateSafety = 2.0, kGateSlack = 2e-9;

__global__ __launch_bounds__(256) void pole_bounds_kernel(PoleFix a) {
  const uint32_t n = a.list->count;
  PoleEntry *entries = reinterpret_cast<PoleEntry *>(a.list + 1);
  const int64_t K = a.K;
  double lo = 0.0;
  for (uint32_t e = blockIdx.x * blockDim.x + threadIdx.x; e < n; e += gridDim.x * blockDim.x) {
    const PoleEntry en = entries[e];
    const float g = __uint_as_float(en.gap);
    const double *rec = a.sums + (size_t)en.q * a.sumsStride;
    double bound = __builtin_huge_val();
    const double pri = a.priority[en.q];
    if (en.gap != 0u && g > 4.0 * kGateD && K <= 31) {
      double totW = 0.0, wv = 0.0;
      for (int64_t k = 0; k < K; k++) { totW += rec[a.wOff + k]; wv += rec[a.vOff + k]; }   // (a.secondIsWV: W_k sqrt V_k)
      const double avgV = wv / totW;
      if (avgV > 1e-100 && totW > 0.0) {
        const double lnV = log_pos(avgV);
        const double vComp = 1.0 / (0.34657359027997265470861606072909 - lnV + a.vCompTail);
        const int rows = en.rowMask != 0u ? __popc(en.rowMask) : (int)K;
        const double lack = rows * (1.1 * kGateD + 3e-17) / ((double)g - 2.0 * kGateD);
        const double vel = 38.0 * kGateD * vComp / avgV;
        bound = kGateSafety * (lack + 30.0 * kGateD + vel) + kGateSlack;
      }
    }
    if (!(bound < 0.25)) bound = __builtin_huge_val();         // (nothing to gain: redone)
    float bf = (float)bound;
    if ((double)bf < bound) bf = __uint_as_float(__float_as_uint(bf) + 1u);   // (rounded up)
    entries[e].gap = __float_as_uint(bf);
    if (bound < 0.25 && pri > 0.0) { const double l = pri * (1.0 - bound); lo = l > lo ? l : lo; }
  }
  lo = wave_max_d(lo);
  if (threadIdx.x % kWave == 0 && lo > 0.0) atomicMax(&a.list->floorBits, (unsigned long long)d2u(lo));
}

// One WAVE per suspect, no workgroup barriers: every lane stages a pair of targets of the listed rows per chunk (the loads of the next
// chunk in flight while this one is worked on), then four lanes per row run the rows' reference-order sums over the chunk -- T / 4
// dependent Kahan steps per row in all, the only serial part -- while a SIMD's other waves fill the gaps of the chain with their own
// suspects' work.  LDS per wave: one chunk of the rows (RMAX KB) and a few dozen doubles: a CU holds twenty suspects at once.
// (Round 5's first form gave a suspect a workgroup -- three staging waves, one chain wave, a barrier per chunk: 30 us per suspect
// whatever the row length, three suspects per CU.)
// RMAX: rows of a question side by side (the launcher takes the smallest of 2 / 5 / 8 / 16 that holds the answers).
template <int RMAX>
__global__ __launch_bounds__(kFixThreads) __attribute__((amdgpu_waves_per_eu(PQA_FIX_WAVES, PQA_FIX_WAVES))) void pole_fixup_kernel(PoleFix a) {
  extern __shared__ double smem[];
  const uint32_t n = a.list->count;   // (written by the sweep: a kernel boundary ago)
  if (n == 0) return;
  const int tid = threadIdx.x, lane = tid % kWave;
  const int wave = (int)__builtin_amdgcn_readfirstlane(tid / kWave);
  const int64_t K = a.K, ldT = a.ldT, nT = 4 * ((a.T + 3) >> 2);
  constexpr int R = RMAX;
  constexpr unsigned kFixSlotBytes = (R + 2) * 1024u + 16u;
  // this wave's LDS: the DMA ring | the rows' likelihoods of one chunk [R][kFixRowStride] | per row {four lanes' sums, corrections, largest element, its place, dH, dL} [R][kFixRed] | the suspect's record [2 K + 2]
  double *ring = smem + (size_t)wave * a.waveLds;              // [kFixDepth] slots of {R answer rows, mD, prior: a KB each; four gap words}
  double *buf = ring + (size_t)kFixDepth * (kFixSlotBytes / 8);
  double *red = buf + (size_t)R * kFixRowStride;
  double *rec = red + (size_t)R * kFixRed;
  double *recW = rec, *recV = rec + K;
  const PoleEntry *entries = reinterpret_cast<const PoleEntry *>(a.list + 1);
  const int nChunks = (int)((nT + kFixChunk - 1) / kFixChunk);
  const int wgWaves = (int)(blockDim.x / kWave);               // (four; fewer where dozens of rows side by side take the LDS)
  const uint32_t nWaves = gridDim.x * (uint32_t)wgWaves;
  for (uint32_t e = blockIdx.x * (uint32_t)wgWaves + wave; e < n; e += nWaves) {
    // (the entry is the same for every lane: said once, so that everything derived from it -- the row and record pointers --
    //  lives in scalar registers)
    PoleEntry en = entries[e];
    en.q = __builtin_amdgcn_readfirstlane(en.q);
    en.b = __builtin_amdgcn_readfirstlane(en.b);
    en.rowMask = __builtin_amdgcn_readfirstlane(en.rowMask);
    const int64_t qLocal = en.q;                               // position in the priority vector; the cube's question is qFirst + it
    const double *prior = a.prior;
    if (a.slots != nullptr) {
      const uint64_t pp = (uint64_t)(uintptr_t)a.slots[en.b].prior;
      prior = reinterpret_cast<const double *>((uintptr_t)(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(pp >> 32)) << 32) |
                                                              (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)pp)));
    }
    if (a.gate) {   // (the bound pole_bounds_kernel left in the entry: a question that cannot be the maximum keeps the sweep's priority)
      const float bf = __uint_as_float((uint32_t)__builtin_amdgcn_readfirstlane((int)en.gap));
      const double pri = a.priority[qLocal], floorP = u2d(__hip_atomic_load(&a.list->floorBits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
      if (bf < 0.25f && pri * (1.0 + (double)bf) < floorP) continue;
    }
    uint32_t rowMask = en.rowMask;
    if (a.maskDense != nullptr) rowMask = __builtin_amdgcn_readfirstlane(a.maskDense[qLocal]);
    if (K > 31 || rowMask == 0) rowMask = K >= 32 ? 0xFFFFFFFFu : (1u << K) - 1u;   // (not known, or dozens of answers: every row)
    // the suspect's record of sums, in LDS while it is worked on: W_k | second | sum l log2 p | lack
    double *recG = a.sums + (size_t)(a.bySlot ? (int64_t)e : qLocal) * a.sumsStride;
    for (int i = lane; i < 2 * (int)K + 2; i += kWave)
      rec[i] = recG[i < K ? a.wOff + i : i < 2 * K ? a.vOff + (i - (int)K) : i == 2 * K ? a.hOff : a.lOff];
    const double *qBase = a.cube + (a.qFirst + qLocal) * (K + 1) * ldT;
    const double *rowD = qBase + K * ldT;
    double dHsum = 0.0, dLsum = 0.0;                           // (lane 0: what the near-1 elements change in the entropy and lack sums)
    for (int64_t kBase = 0; kBase < K; kBase += 32) {          // (more than 32 answers: every row, 32 at a time)
      uint32_t rest = K > 31 ? (K - kBase >= 32 ? 0xFFFFFFFFu : (1u << (K - kBase)) - 1u) : rowMask;
      while (rest != 0) {
        // ---- a batch of up to R listed rows
        uint32_t rowOff[R];                                    // (element offset of the row within the question's block)
        int nb = 0;
#pragma unroll
        for (int r = 0; r < R; r++) {
          rowOff[r] = 0;
          if (rest != 0) { rowOff[r] = (uint32_t)(((int)kBase + __builtin_ctz(rest)) * ldT); rest &= rest - 1; nb = r + 1; }
        }
        // lane 4 r + cc: lane cc of row r's accumulator (SRAccumVectDbl256.h:40-46); every lane: the largest likelihood it has formed
        // per row, and where (chunk and element of its pair)
        double sum = 0.0, corr = 0.0;
        double mx[R];
        int mxAt[R];
#pragma unroll
        for (int r = 0; r < R; r++) { mx[r] = 0.0; mxAt[r] = 0; }
        // The operands of the rows' likelihoods (:72-82) arrive by LDS-DMA -- global memory straight into this wave's ring in LDS, no
        // registers held while they fly -- kFixDepth chunks ahead of the chunk being worked on: per chunk the mD row, the prior, the
        // listed answer rows (16 bytes per lane each) and the chunk's four gap words.  A lane's pair of chunk c lies at byte
        // 16 lane of the row's KB in slot c % kFixDepth.
        const unsigned ringAddr = (unsigned)(uintptr_t)ring;
        const int G = nb + 3;                                  // loads per chunk
        dma_rsrc_t rsA[R];
#pragma unroll
        for (int r = 0; r < R; r++) rsA[r] = dma_rsrc(qBase + rowOff[r], ldT * 8);
        const dma_rsrc_t rsD = dma_rsrc(rowD, ldT * 8), rsP = dma_rsrc(prior, ldT * 8), rsG = dma_rsrc(a.tgap, ((nT + 31) >> 5) * 4);
        auto request = [&](int c) __attribute__((always_inline)) {
          const unsigned slotAddr = ringAddr + (unsigned)(c % kFixDepth) * kFixSlotBytes;
          const unsigned off = (unsigned)c * (kFixChunk * 8u) + 16u * (unsigned)lane;
#pragma unroll
          for (int r = 0; r < R; r++)
            if (r < nb) dma16(rsA[r], off, slotAddr + (unsigned)r * 1024u);
          dma16(rsD, off, slotAddr + (unsigned)R * 1024u);
          dma16(rsP, off, slotAddr + (unsigned)(R + 1) * 1024u);
          if (lane < 4) dma4(rsG, (unsigned)c * 16u + 4u * (unsigned)lane, slotAddr + (unsigned)(R + 2) * 1024u);
        };
        for (int c = 0; c < kFixDepth && c < nChunks; c++) request(c);
        for (int c = 0; c < nChunks; c++) {
          const int later = nChunks - 1 - c < kFixDepth - 1 ? nChunks - 1 - c : kFixDepth - 1;   // chunks requested behind this one
          wait_vmcnt(later * G);
          const char *slot = reinterpret_cast<const char *>(ring) + (size_t)(c % kFixDepth) * kFixSlotBytes;
          const int64_t t0 = (int64_t)c * kFixChunk + 2 * lane;
          if (t0 < nT) {
            const double2 dv = *reinterpret_cast<const double2 *>(slot + R * 1024 + 16 * lane);
            const double2 pv = *reinterpret_cast<const double2 *>(slot + (R + 1) * 1024 + 16 * lane);
            const uint32_t gw = reinterpret_cast<const uint32_t *>(slot + (R + 2) * 1024)[lane >> 4] >> ((2 * lane) & 31);
            const bool g0 = gw & 1u, g1 = gw & 2u;
            const double id0 = g0 ? 0.0 : div_nr(1.0, dv.x), id1 = g1 ? 0.0 : div_nr(1.0, dv.y);   // :74
            const double p0 = g0 ? 0.0 : pv.x, p1 = g1 ? 0.0 : pv.y;                                  // :103
#pragma unroll
            for (int r = 0; r < R; r++)
              if (r < nb) {
                const double2 av = *reinterpret_cast<const double2 *>(slot + r * 1024 + 16 * lane);
                const double l0 = (av.x * id0) * p0, l1 = (av.y * id1) * p1;                          // :81-82
                *reinterpret_cast<double2 *>(buf + (size_t)r * kFixRowStride + 2 * lane) = make_double2(l0, l1);
                const double lm = l1 > l0 ? l1 : l0;
                if (lm > mx[r]) { mx[r] = lm; mxAt[r] = 2 * c + (l1 > l0 ? 1 : 0); }
              }
          }
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (the slot has been read: its next chunk may land)
          if (c + kFixDepth < nChunks) request(c + kFixDepth);
          __builtin_amdgcn_wave_barrier();                     // (one wave, LDS in order: the chains read what the lanes wrote)
          if (lane < 4 * nb) {
            const double *src = buf + (size_t)(lane >> 2) * kFixRowStride + (lane & 3);
            const int64_t left = nT / 4 - (int64_t)c * (kFixChunk / 4);
            const int steps = (int)(left < kFixChunk / 4 ? left : kFixChunk / 4);
            int j = 0;
            for (; j + 8 <= steps; j += 8) {                   // (eight elements requested at once, added in order)
              double x[8];
#pragma unroll
              for (int u = 0; u < 8; u++) x[u] = src[4 * (j + u)];
#pragma unroll
              for (int u = 0; u < 8; u++) {
                const double y = x[u] - corr;
                const double t = sum + y;
                corr = (t - sum) - y;
                sum = t;
              }
            }
            for (; j < steps; j++) {
              const double y = src[4 * j] - corr;
              const double t = sum + y;
              corr = (t - sum) - y;
              sum = t;
            }
          }
          __builtin_amdgcn_wave_barrier();                     // (... and the next chunk is written behind these reads)
        }
        // ---- per row: the four lanes' results, the largest element and its place
        if (lane < 4 * nb) {
          double *out = red + (size_t)(lane >> 2) * kFixRed;
          out[lane & 3] = sum;
          out[4 + (lane & 3)] = corr;
        }
#pragma unroll
        for (int r = 0; r < R; r++)
          if (r < nb) {
            const double wmx = wave_max_d(mx[r]);
            double *cw = red + (size_t)r * kFixRed + 8;
            if (lane == 0) { cw[0] = 0.0; cw[1] = 0.0; }
            if (mx[r] == wmx && mx[r] > 0.0) {                 // (behind lane 0's zeros; lanes that tie hold equal values: any one's place serves)
              cw[0] = mx[r];
              cw[1] = (double)((mxAt[r] >> 1) * kFixChunk + 2 * lane + (mxAt[r] & 1));
            }
          }
        __builtin_amdgcn_wave_barrier();
        if (lane < nb) {
          uint32_t off = rowOff[0];                            // (rowOff[lane] without a dynamic register index)
#pragma unroll
          for (int r = 1; r < R; r++) off = lane == r ? rowOff[r] : off;
          const int kk = (int)(off / (uint32_t)ldT);
          double *rr = red + (size_t)lane * kFixRed;
          const double cand = rr[8];
          const int candT = (int)rr[9];
          const double Wx = precise_sum4(rr, rr + 4);          // :88
          const double invWx = div_nr(1.0, Wx);                // :91
          const double pRef = cand * invWx;                    // :97
          double dH = 0.0, dL = 0.0;
          if (cand > 0.0 && (uint32_t)(d2u(pRef) >> 32) >= kQuarterHi) {     // (the row is listed: next to 1, or a quarter and a vanishing velocity sum)
            const double dAt = rowD[candT], prh = prior[candT];   // (one round trip for both)
            const double Wf = recW[kk];                        // the sweep's W_k
            const double pFast = cand * div_nr(1.0, Wf);       // what pass 2 took for this element
            const double lFast = log2hot_global(pFast, gLog2TableP);
            const double lRef = log2hot_ref(pRef, gLog2TableP, gLog2Entry0RefP);   // :106
            dH = cand * lRef - cand * lFast;                   // :113-114 (weighted by W_k: eval_epilogue)
            const double candId = div_nr(1.0, dAt);            // :74 (not a gap: its likelihood is positive)
            const double id2 = candId * candId;
            dL = div_fast(id2, lRef) - div_fast(id2, lFast);   // :117 (pass 2's quotient was within 2^-48.8 of the second one)
            // :119-127 the element's velocity term: a difference of two numbers next to 1
            const double dF = pFast - prh, dR = pRef - prh;
            const double vOld = a.secondIsWV ? [&] { const double sv = div_fast(recV[kk], Wf); return sv * sv; }() : recV[kk];
            double vNew = (vOld - dF * dF) + dR * dR;
            if (!(vNew > 0.0)) vNew = dR * dR;
            recV[kk] = a.secondIsWV ? Wx * sqrt(vNew) : vNew;  // :156-157
            recW[kk] = Wx;                                     // :90
          }
          rr[10] = dH;
          rr[11] = dL;
        }
        __builtin_amdgcn_wave_barrier();
        if (lane == 0)
          for (int r = 0; r < nb; r++) { dHsum += red[(size_t)r * kFixRed + 10]; dLsum += red[(size_t)r * kFixRed + 11]; }
        __builtin_amdgcn_wave_barrier();                       // (red[] is written again by the next batch)
      }
    }
    if (lane == 0) {
      if (a.priority != nullptr || a.priorityT != nullptr || a.slots != nullptr) {
        const double pri = eval_epilogue(recW, -(rec[2 * K] + dHsum), recV, K, rec[2 * K + 1] + dLsum, a.vCompTail);   // :130-207
        double *dst = a.priorityT != nullptr ? a.priorityT + (size_t)qLocal * a.Bp + en.b
                      : a.slots != nullptr ? a.slots[en.b].priority + qLocal : a.priority + qLocal;
        __hip_atomic_store(dst, pri, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        TaggedPriority *hp = a.slots != nullptr ? a.slots[en.b].hostPriority : a.hostPriority;
        if (hp != nullptr) {
          typedef unsigned int u4 __attribute__((ext_vector_type(4)));
          const uint64_t w0 = d2u(pri), w1 = a.hostTag;
          const u4 x = {(unsigned)w0, (unsigned)(w0 >> 32), (unsigned)w1, (unsigned)(w1 >> 32)};
          asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(hp + qLocal), "v"(x) : "memory");
        }
      } else {                                                 // (the caller's epilogue kernel follows: cluster_kernels.hip)
        for (int k = 0; k < K; k++) { recG[a.wOff + k] = recW[k]; recG[a.vOff + k] = recV[k]; }
        recG[a.hOff] = rec[2 * K] + dHsum;
        recG[a.lOff] = rec[2 * K + 1] + dLsum;
      }
      if (a.maskDense != nullptr) a.maskDense[qLocal] = 0;
      if (a.dirty != nullptr) a.dirty[en.b] = 1u;
    }
    __builtin_amdgcn_wave_barrier();                           // (rec[] is the next suspect's)
  }
  double *misc = smem, *pub = smem + 8;                        // (the waves' LDS is free by the barrier below)
  // ---- the last workgroup to get here publishes what the sweep's finisher left to this kernel, and empties the list
  if (a.hostPriority != nullptr || a.slots != nullptr) __threadfence_system();   // (the host's records among the corrected priorities)
  else __threadfence();
  __syncthreads();
  if (tid == 0) {
    const uint32_t arrived = atomicAdd(&a.list->arrived, 1u);
    misc[2] = arrived == gridDim.x - 1 ? 1.0 : 0.0;
  }
  __syncthreads();
  if (misc[2] == 0.0) return;
  __threadfence();
  if (a.fs.scratch != nullptr && a.slots == nullptr) {
    const int64_t nQ = a.nQ;
    const bool sampled = a.fs.sampleSubtasks > 0;
    uint64_t seqValue = a.fs.seqValue, flagValue = a.fs.flagValue;
    if (a.fs.tagCell != nullptr) seqValue = flagValue = __hip_atomic_load(a.fs.tagCell, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    double outP = 0.0;
    int64_t outI = 0;
    if (!sampled) {
      // the argmax over every evaluated question (maximum priority, lowest index on ties, NaN never wins: eval_kernels.hip)
      Best b{0.0, -1};
      for (int64_t j = tid; j < nQ; j += blockDim.x) {
        const int64_t q = a.qFirst + j;
        if (bit_test(a.qgap, q) || bit_test(a.asked, q)) continue;
        double p = __hip_atomic_load(a.priority + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (p != p) p = -__builtin_huge_val();
        best_merge(b, p, j);
      }
      for (int m = kWave / 2; m >= 1; m >>= 1) {
        const double op = __shfl_xor(b.p, m, kWave);
        const int64_t oi = __shfl_xor(b.i, m, kWave);
        best_merge(b, op, oi);
      }
      Best *wb = reinterpret_cast<Best *>(pub);
      if (lane == 0) wb[wave] = b;
      __syncthreads();
      if (tid == 0) {
        for (int w = 1; w < wgWaves; w++) best_merge(b, wb[w].p, wb[w].i);
        outP = b.i < 0 ? 0.0 : b.p;
        outI = b.i < 0 ? -1 : b.i + a.fs.outBase;
      }
    } else if (a.fs.hostPriority == nullptr) {
      // the reference's selector over the corrected vector (the sweep's own workgroup 0 would have run it)
      const SampledPick r = select_sampled_wg_lds<true>(a.priority, a.qgap, a.asked, a.qFirst, nQ, a.fs.sampleSubtasks, a.fs.sampleRnd, pub);
      outP = r.priority;
      outI = r.index + a.fs.outBase;
    }
    if (tid == 0) {
      a.fs.out->priority = outP;
      a.fs.out->index = outI;
      if (a.fs.seq != nullptr) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");          // system scope: the record (and the handed-over priorities) before the flag
        __hip_atomic_store(a.fs.seq, flagValue, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      }
      if (a.fs.tagCell != nullptr) {
        uint64_t next = seqValue + 1;
        if ((uint32_t)next == 0) next++;
        __hip_atomic_store(a.fs.tagCell, next, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  }
  if (a.fs.scratch != nullptr && a.slots != nullptr && a.nSlots > 0) {
    // a grid.y = quiz launch: every quiz's result (the finishers that saw the list empty have published theirs already -- the same)
    const bool handOver = a.fs.sampleSubtasks > 0;            // (the priorities went to the host as tagged records: the flags only)
    for (int b = wave; b < a.nSlots; b += wgWaves) {
      const QuizSlot qs = a.slots[b];
      Best best{0.0, -1};
      if (!handOver) {
        for (int64_t j = lane; j < a.nQ; j += kWave) {
          const int64_t q = a.qFirst + j;
          if (bit_test(a.qgap, q) || bit_test(qs.asked, q)) continue;
          double p = __hip_atomic_load(qs.priority + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (p != p) p = -__builtin_huge_val();
          best_merge(best, p, j);
        }
        for (int m = kWave / 2; m >= 1; m >>= 1) {
          const double op = __shfl_xor(best.p, m, kWave);
          const int64_t oi = __shfl_xor(best.i, m, kWave);
          best_merge(best, op, oi);
        }
      }
      if (lane == 0) {
        qs.out->priority = handOver || best.i < 0 ? 0.0 : best.p;
        qs.out->index = handOver ? 0 : best.i < 0 ? -1 : best.i + a.fs.outBase;
        if (qs.seq != nullptr) {
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
          __hip_atomic_store(qs.seq, a.fs.flagValue, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
      }
    }
  }
  if (tid == 0) {
    a.list->arrived = 0;
    a.list->floorBits = 0ull;
    __hip_atomic_store(&a.list->count, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

}  // namespace

size_t PoleListBytes(int64_t capacity) { return sizeof(PoleHeader) + (size_t)capacity * sizeof(PoleEntry); }

// Launched behind every sweep that watches (the caller set `fix` up for the sweep's own shape).  selectLdsDoubles: LDS the
// reference's selector needs when this kernel has to run it (0: not asked for).
hipError_t LaunchPoleFixup(const PoleFix &fix, hipStream_t stream) {
  if (fix.list == nullptr || fix.sums == nullptr) return hipErrorInvalidValue;
  PoleFix a = fix;
  a.rows = a.K <= 2 ? 2 : a.K <= 5 ? 5 : a.K <= 8 ? 8 : 16;
  void (*kern)(PoleFix) = a.rows == 2 ? pole_fixup_kernel<2> : a.rows == 5 ? pole_fixup_kernel<5> : a.rows == 8 ? pole_fixup_kernel<8> : pole_fixup_kernel<16>;
  a.waveLds = (int)(((size_t)kFixDepth * (((size_t)a.rows + 2) * 1024 + 16) / 8 + (size_t)a.rows * kFixRowStride + (size_t)a.rows * kFixRed + 2 * (size_t)a.K + 2 + 1) / 2 * 2);
  int wgWaves = kFixThreads / kWave;
  while (wgWaves > 1 && (size_t)wgWaves * a.waveLds * sizeof(double) > 150 * 1024) wgWaves--;   // (sixteen rows side by side: two waves)
  size_t shmem = (size_t)wgWaves * a.waveLds * sizeof(double);
  if (shmem < 512) shmem = 512;
  if (a.fs.scratch != nullptr && a.fs.sampleSubtasks > 0 && a.fs.hostPriority == nullptr) {
    const size_t need = ((size_t)select_sampled_lds_doubles(a.nQ, a.fs.sampleSubtasks) + 8) * sizeof(double);
    if (need > shmem) shmem = need;
  }
  if (shmem > 160 * 1024) return hipErrorInvalidValue;
  static LaunchCache caches[4];   // (per kernel: the shape is a function of the LDS size, up to selections with very many subtasks)
  LaunchCache &cache = caches[a.rows == 2 ? 0 : a.rows == 5 ? 1 : a.rows == 8 ? 2 : 3];
  const int dev = LaunchCache::Device();
  int perCU = 0;
  if (!cache.Get(dev, shmem, &perCU)) {
    if (shmem > 64 * 1024) {
      const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
      if (e != hipSuccess) return e;
    }
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&perCU, kern, wgWaves * kWave, shmem) != hipSuccess || perCU < 1) perCU = 1;
    cache.Put(dev, shmem, perCU);
  }
  int64_t grid = (int64_t)cache.NumCUs(dev) * perCU;
  const int64_t wgs = (fix.capacity + wgWaves - 1) / wgWaves;   // (a wave per suspect)
  if (fix.capacity > 0 && grid > wgs) grid = wgs;
  if (grid < 1) grid = 1;
  if (a.gate) {
    if (a.priority == nullptr || a.slots != nullptr || a.bySlot || !a.secondIsWV) a.gate = 0;   // (what pole_bounds_kernel reads)
    else hipLaunchKernelGGL(pole_bounds_kernel, dim3((unsigned)((fix.capacity + 255) / 256 < 1 ? 1 : (fix.capacity + 255) / 256 > 64 ? 64 : (fix.capacity + 255) / 256)),
                            dim3(256), 0, stream, a);
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3((unsigned)(wgWaves * kWave)), shmem, stream, a);
  return hipGetLastError();
}

}  // namespace pqa
