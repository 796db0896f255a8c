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
// whose largest posterior element is within 2^-10 of 1 (pole_device.h: kNearOneHi -- why there) leaves its sums in memory and an entry in the suspect list, and the sweep's finisher, seeing a non-empty list, leaves
// the publication of the result to this kernel, which is launched behind every watching sweep (an empty list: a few hundred
// threads read one word and leave).  Here a workgroup takes one suspect at a time:
//   * waves 1 - 3 form the likelihoods (A * invD) * prior of the listed rows again, chunk by chunk, into LDS -- bit for bit the
//     sweep's and the reference's (:72-82) -- and keep each row's largest element;
//   * wave 0 runs the rows' reference-order sums side by side, four lanes per row (the chunks double-buffered against the
//     stagers), PreciseSum folds the four;
//   * for the element within 2^-10 of 1: Log2Hot by the reference's own operation sequence (SRVectMath.h:87-135) on
//     p = l * (1 / W_k) with the reference-order W_k, and its entropy, lack and velocity terms replace what pass 2 had added;
//   * the epilogue (:134-207) again, the priority stored (and handed to the host where the sweep hands priorities over).
// The last workgroup to finish then does what the sweep's finisher left undone: the argmax / the reference's selector / the
// flag of the hand-over, and empties the list.
// In a late quiz EVERY question has such rows (the target's likelihood is all of W_k whatever the answer): the fix then re-reads
// the cube once and runs T / 4 dependent steps per question, a few questions per CU at a time -- about one more sweep's time,
// whatever the row length; the in-kernel fix of round 4 (rows of up to 4096 targets only, lists of 62 / 128 suspects) cost 4.5 - 6.5.
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
constexpr int kFixThreads = 256;                   // four waves, each by itself
constexpr int kFixChunk = 2 * kWave;               // targets of a row per LDS chunk: one 16-byte pair per lane
constexpr int kFixRowStride = kFixChunk + 4;       // doubles between the rows of a chunk (the rows' chain lanes on different banks)
constexpr int kFixRed = 12;                        // doubles of scratch per row

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
  // this wave's LDS: the rows' chunk [R][kFixRowStride] | per row {four lanes' sums, corrections, largest element, its place, dH, dL} [R][kFixRed] | the suspect's record [2 K + 2]
  double *buf = smem + (size_t)wave * a.waveLds;
  double *red = buf + (size_t)R * kFixRowStride;
  double *rec = red + (size_t)R * kFixRed;
  double *recW = rec, *recV = rec + K;
  const PoleEntry *entries = reinterpret_cast<const PoleEntry *>(a.list + 1);
  const int nChunks = (int)((nT + kFixChunk - 1) / kFixChunk);
  const uint32_t nWaves = gridDim.x * (kFixThreads / kWave);
  for (uint32_t e = blockIdx.x * (kFixThreads / kWave) + wave; e < n; e += nWaves) {
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
        struct Chunk { double2 dv, pv, av[R]; uint32_t gw; };
        // the loads of chunk c (:72-82's operands), a pair of targets per lane
        auto request = [&](Chunk &ch, int c) __attribute__((always_inline)) {
          const int64_t t0 = (int64_t)c * kFixChunk + 2 * lane;
          const int64_t tc = t0 < nT ? t0 : 0;                 // (beyond the row: any valid pair, not used)
          ch.dv = *reinterpret_cast<const double2 *>(rowD + tc);
          ch.pv = *reinterpret_cast<const double2 *>(prior + tc);
          ch.gw = a.tgap[tc >> 5] >> (tc & 31);
#pragma unroll
          for (int r = 0; r < R; r++)
            if (r < nb) ch.av[r] = *reinterpret_cast<const double2 *>(qBase + rowOff[r] + tc);
        };
        // chunk c: the rows' likelihoods as pass 1 forms them, into LDS
        auto stage = [&](const Chunk &ch, int c) __attribute__((always_inline)) {
          const int64_t t0 = (int64_t)c * kFixChunk + 2 * lane;
          if (t0 < nT) {
            const bool g0 = ch.gw & 1u, g1 = ch.gw & 2u;
            const double id0 = g0 ? 0.0 : div_nr(1.0, ch.dv.x), id1 = g1 ? 0.0 : div_nr(1.0, ch.dv.y);   // :74
            const double p0 = g0 ? 0.0 : ch.pv.x, p1 = g1 ? 0.0 : ch.pv.y;                                // :103
#pragma unroll
            for (int r = 0; r < R; r++)
              if (r < nb) {
                const double l0 = (ch.av[r].x * id0) * p0, l1 = (ch.av[r].y * id1) * p1;                  // :81-82
                *reinterpret_cast<double2 *>(buf + (size_t)r * kFixRowStride + 2 * lane) = make_double2(l0, l1);
                const double lm = l1 > l0 ? l1 : l0;
                if (lm > mx[r]) { mx[r] = lm; mxAt[r] = 2 * c + (l1 > l0 ? 1 : 0); }
              }
          }
        };
        Chunk ch;
        request(ch, 0);
        for (int c = 0; c < nChunks; c++) {
          stage(ch, c);
          if (c + 1 < nChunks) request(ch, c + 1);             // (in flight during the chains below, and the other waves' work)
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
      for (int64_t j = tid; j < nQ; j += kFixThreads) {
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
        for (int w = 1; w < kFixThreads / kWave; w++) best_merge(b, wb[w].p, wb[w].i);
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
    for (int b = wave; b < a.nSlots; b += kFixThreads / kWave) {
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
  a.waveLds = (int)(((size_t)a.rows * kFixRowStride + (size_t)a.rows * kFixRed + 2 * (size_t)a.K + 2 + 1) / 2 * 2);
  size_t shmem = (size_t)(kFixThreads / kWave) * a.waveLds * sizeof(double);
  if (shmem < 512) shmem = 512;
  if (a.fs.scratch != nullptr && a.fs.sampleSubtasks > 0 && a.fs.hostPriority == nullptr) {
    const size_t need = ((size_t)select_sampled_lds_doubles(a.nQ, a.fs.sampleSubtasks) + 8) * sizeof(double);
    if (need > shmem) shmem = need;
  }
  if (shmem > 160 * 1024) return hipErrorInvalidValue;
  static LaunchCache cache;   // (the shape is a function of the LDS size, up to selections with very many subtasks)
  const int dev = LaunchCache::Device();
  int perCU = 0;
  if (!cache.Get(dev, shmem, &perCU)) {
    if (shmem > 64 * 1024) {
      const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
      if (e != hipSuccess) return e;
    }
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&perCU, kern, kFixThreads, shmem) != hipSuccess || perCU < 1) perCU = 1;
    cache.Put(dev, shmem, perCU);
  }
  int64_t grid = (int64_t)cache.NumCUs(dev) * perCU;
  const int64_t wgs = (fix.capacity + kFixThreads / kWave - 1) / (kFixThreads / kWave);   // (a wave per suspect)
  if (fix.capacity > 0 && grid > wgs) grid = wgs;
  if (grid < 1) grid = 1;
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(kFixThreads), shmem, stream, a);
  return hipGetLastError();
}

}  // namespace pqa
