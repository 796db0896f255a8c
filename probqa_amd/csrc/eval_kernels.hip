// eval_kernels.hip -- the next-question priority sweep on CDNA4 (gfx950).
//
// Replaces CEEvalQsSubtaskConsider<SRDoubleNumber>::Run (reference: PqaCore/CEEvalQsSubtaskConsider.cpp:41-217).
//
// Shape: one workgroup of WPQ wavefronts per candidate question (persistent over a grid-stride of questions).
// Lane `tid` owns the target pairs p = tid + j*(64*WPQ), j < NP, i.e. every global load is a fully coalesced
// 16 B/lane (1 KiB/wave) read down the target axis of one sA row.  The row's likelihoods and 1/D stay in REGISTERS
// between the two passes, and the next row is prefetched into registers while the current one is reduced and
// log2'ed, so every byte of the cube is read from HBM exactly once: algorithmic traffic = Q*(K+1)*ldT*8 B per sweep.
//
// The kernel sits on the fp64-ALU / HBM ridge (about 200 wave-cycles of fp64 work per 614 B of cube), so the ALU side
// is trimmed: scale-free exact division (div_nr), DPP/permlane all-reduces instead of ds_bpermute shuffles, one
// workgroup barrier per answer row (only the answer weight W_k is needed by everyone before pass 2; entropy, velocity
// and lack partials go to LDS and are combined once per question).  Reductions: plain fp64 partial sums per lane
// (<= 2*NP terms) followed by butterflies, i.e. a pairwise tree with 64*WPQ leaves (error ~1 ulp of the sum, measured
// 4e-14 max on the priorities against the reference's 4-lane Kahan chains).
// The epilogue (weighted averages over answers, velocity component, integer powers) is the reference's scalar code,
// run by one lane with the reference's Kahan lane order.
//
// This is a reduction with a nonlinear inner function (table log2 + two divisions per element): no MFMA.
#include "pqa_device.h"
#include "eval_device.h"
#include "pqa_kernels.h"
#include "prior_device.h"
#include "pole_device.h"

namespace pqa {

static __device__ double gLog2Table[kLog2TableDoubles];  // {log2(midpoint), 1/(2*midpoint)} per bucket
hipError_t UploadLog2Table(const double *hostTable) {
  return hipMemcpyToSymbol(HIP_SYMBOL(gLog2Table), hostTable, kLog2TableDoubles * sizeof(double));
}

// (prior_device.h: reference_order_sum, TopRequest -- the posterior update that eval_questions_f64_upd runs ahead of its sweep)
struct EvalArgs {
  const double *cube;
  const double *prior;
  const uint32_t *tgap;
  const uint32_t *qgap;
  const uint32_t *asked;
  double *priority;
  double *poleScratch;   // KbView::poleScratch (single-quiz launches), or null: the sums of questions that passed the pole watch, [question][2 K + 2]
  PoleHeader *poleList;  // KbView::poleList: ... and their entries (pole_kernels.hip is launched behind the sweep); null: no watch
  bool serverWatch;          // the resident kernel watches too (nothing can be launched behind a step: a step that found a row says so --
                             // index -4 -- and the host takes the launched path)
  bool serverNoWatch;        // ... not reported for this request (ServerMailbox: kServerNoWatch)
  int64_t K, T, ldT, qFirst, qLimit;
  double vCompTail;  // ln(sqrt 2) / (nValidTargets + 1)^2, PqaCore/CEEvalQsSubtaskConsider.cpp:191
  FusedSelect fs;
  const QuizSlot *slots;  // batched launch: per-quiz pointers, indexed by blockIdx.y (nullptr: a single quiz)
  int maxGrid;            // host side only: KbView::maxGrid
  int poleNoFollow;       // KbView::poleNoFollow (measurement hook): the register-form sweep watches but lists and defers nothing, and no fix is launched behind it
  int poleGate;           // the launch is a fused argmax of one quiz and KbView::poleGate is on: the watch also tracks the listed questions' gaps (PoleEntry::gap)
  // eval_questions_f64_upd only: the answer whose posterior update runs in the sweep's prologue (sweep_body, FUSE)
  const double *updRowA, *updRowD;   // sA[q][a][.], mD[q][.] of the answered question
  int64_t updQuestion;               // its index (local): asked from this sweep on
  int64_t updVects, updWorkers;      // ceil(T / 4), the subtasks of the reference's sum
  int64_t updT;
  TopRequest updTop;                 // the new posterior's best targets, listed by workgroup 0 (count 0: none)
};

// batched launch: this workgroup's quiz replaces the per-quiz fields of the arguments
__device__ __forceinline__ void select_quiz(EvalArgs &a) {
  if (a.slots == nullptr) return;
  const QuizSlot s = a.slots[blockIdx.y];
  a.prior = s.prior;
  a.asked = s.asked;
  a.priority = s.priority;
  a.fs.out = s.out;
  a.fs.seq = s.seq;
  a.fs.hostPriority = s.hostPriority;
  a.fs.scratch += (size_t)blockIdx.y * (size_t)a.fs.scratchStride;
}

// Fused argmax: a selection is ONE launch and -- with out/seq in host-coherent memory -- needs no copy and no stream
// synchronisation.  Every workgroup keeps the best of its own questions in registers while it sweeps (maximum priority,
// lowest index on ties, NaN never wins) and publishes ONE 16-byte record {priority, launch tag : index} with a single
// write-through store.  There is no arrival counter: the epilogue wave of workgroup 0 is the finisher -- it polls the
// tags of all records (one coalesced load per 64 workgroups) until every one carries this launch's tag, reduces the
// records and publishes the winner.  After the last workgroup's store lands, the tail is one poll, one read of the
// priorities and the write to the host: device-scope round trips (each ~1 us across the XCDs' L2s) are what the tail
// of a 19 us kernel is made of, and a counter tree costs five of them in series.
// Never a deadlock: only the finisher waits, and it waits for workgroups that need nothing from it.
// Visibility: records are written with sc1 (write-through past the XCD's L2) and polled with sc1 loads; the two words
// of a record are written by one 16-byte store and the priority word is read only after its tag has been seen.
__device__ __forceinline__ void store_priority(double *p, double v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

struct Best {
  double p;
  int64_t i;  // position in priority[], < 0: none
};

__device__ __forceinline__ void best_merge(Best &b, double op, int64_t oi) {
  if (oi >= 0 && (b.i < 0 || op > b.p || (op == b.p && oi < b.i))) { b.p = op; b.i = oi; }
}
__device__ __forceinline__ void best_offer(Best &b, double p, int64_t i) {
  best_merge(b, p != p ? -__builtin_huge_val() : p, i);   // NaN never wins over a number
}
__device__ __forceinline__ Best wave_best(Best b) {       // every lane ends up with the wave's best
#pragma unroll
  for (int m = kWave / 2; m >= 1; m >>= 1) {
    const double op = __shfl_xor(b.p, m, kWave);
    const int64_t oi = __shfl_xor(b.i, m, kWave);
    best_merge(b, op, oi);
  }
  return b;
}

// Called by every lane of ONE wave per workgroup (the wave that stored the workgroup's priorities) with the lanes'
// running bests.  Record word 1: launch tag (low 32 bits of seqValue) << 32 | index (0xFFFFFFFF: none).
// UNI (resident kernel): the record, the result and the flag are stored by every lane of the wave (same value, same address)
// instead of by lane 0 -- inside the resident loop a lane-0-only store after the finisher's poll loop is parked by the
// structurizer behind the loop exit of its wave while the other lanes run on to the next step's barrier: a deadlock
// (tools/server_rt.hip reproduces it).
template <bool UNI = false>
__device__ __forceinline__ void fused_select(const EvalArgs &a, Best mine, int lane, bool *allReported = nullptr, bool wgSuspect = false) {
  if (a.fs.scratch == nullptr) return;
  const bool sampled = a.fs.sampleSubtasks > 0;
  // sampled: the finisher's workgroup reads every workgroup's priorities afterwards -- they must be visible before the record
  // (they are write-through stores, store_priority: waiting for their acknowledgements is enough -- a release fence here
  //  makes every workgroup write the whole L2 back, 7 us per launch)
  if (sampled && a.fs.hostPriority == nullptr) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (host hand-over: tagged records)
  const Best wg = wave_best(mine);
  uint64_t seqValue = a.fs.seqValue, flagValue = a.fs.flagValue;
  if (a.fs.tagCell != nullptr)  // graph replay: the finisher of the previous replay left this launch's tag here
    seqValue = flagValue = __hip_atomic_load(a.fs.tagCell, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const uint64_t tag = (uint64_t)(uint32_t)seqValue << 32;
  SelectResult *rec = a.fs.scratch;
  if (UNI || lane == 0) {
    typedef unsigned int u4 __attribute__((ext_vector_type(4)));
    // (bit 31 of the index word: a question of this workgroup passed the pole watch -- the finisher learns it with the record)
    const uint64_t w0 = d2u(wg.p), w1 = tag | (uint32_t)(wg.i < 0 ? 0x7FFFFFFFu : (uint32_t)wg.i) | (wgSuspect ? 0x80000000u : 0u);
    const u4 v = {(unsigned)w0, (unsigned)(w0 >> 32), (unsigned)w1, (unsigned)(w1 >> 32)};
    asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(rec + blockIdx.x), "v"(v) : "memory");
  }
  if (blockIdx.x != 0) return;
  const unsigned grid = gridDim.x;
  // One poll = one round trip: each lane fetches its 16 records of a 1024-record chunk with 16-byte loads (tag and
  // priority arrive together, consistent with the 16-byte store) that are all in flight at once.
  // bounded (~20 s): a record can only stay stale if a workgroup of this launch died; then the host gets index -3
  typedef unsigned int u4 __attribute__((ext_vector_type(4)));
  bool complete = true;
  uint32_t anySuspect = 0;
  Best b{0.0, -1};
  for (unsigned chunk = 0; chunk < grid && complete; chunk += 16 * kWave) {
    const SelectResult *p0 = rec + chunk + lane, *p1 = p0 + 4 * kWave, *p2 = p0 + 8 * kWave, *p3 = p0 + 12 * kWave;
    u4 r[16];
    bool all = false;
    for (unsigned spin = 0; spin < (1u << 24) && !all; spin++) {
      asm volatile(
          "global_load_dwordx4 %0, %16, off sc1\n\tglobal_load_dwordx4 %1, %16, off offset:1024 sc1\n\t"
          "global_load_dwordx4 %2, %16, off offset:2048 sc1\n\tglobal_load_dwordx4 %3, %16, off offset:3072 sc1\n\t"
          "global_load_dwordx4 %4, %17, off sc1\n\tglobal_load_dwordx4 %5, %17, off offset:1024 sc1\n\t"
          "global_load_dwordx4 %6, %17, off offset:2048 sc1\n\tglobal_load_dwordx4 %7, %17, off offset:3072 sc1\n\t"
          "global_load_dwordx4 %8, %18, off sc1\n\tglobal_load_dwordx4 %9, %18, off offset:1024 sc1\n\t"
          "global_load_dwordx4 %10, %18, off offset:2048 sc1\n\tglobal_load_dwordx4 %11, %18, off offset:3072 sc1\n\t"
          "global_load_dwordx4 %12, %19, off sc1\n\tglobal_load_dwordx4 %13, %19, off offset:1024 sc1\n\t"
          "global_load_dwordx4 %14, %19, off offset:2048 sc1\n\tglobal_load_dwordx4 %15, %19, off offset:3072 sc1\n\t"
          "s_waitcnt vmcnt(0)"
          : "=&v"(r[0]), "=&v"(r[1]), "=&v"(r[2]), "=&v"(r[3]), "=&v"(r[4]), "=&v"(r[5]), "=&v"(r[6]), "=&v"(r[7]),
            "=&v"(r[8]), "=&v"(r[9]), "=&v"(r[10]), "=&v"(r[11]), "=&v"(r[12]), "=&v"(r[13]), "=&v"(r[14]), "=&v"(r[15])
          : "v"(p0), "v"(p1), "v"(p2), "v"(p3)
          : "memory");
      bool mine = true;
#pragma unroll
      for (int u = 0; u < 16; u++)
        if (chunk + lane + u * kWave < grid) mine = mine && r[u][3] == (unsigned)(tag >> 32);
      all = __all(mine);
      if (!all) __builtin_amdgcn_s_sleep(2);
    }
    complete = all;
#pragma unroll
    for (int u = 0; u < 16; u++)
      if (chunk + lane + u * kWave < grid) {
        const uint64_t pw = (uint64_t)r[u][0] | ((uint64_t)r[u][1] << 32);
        anySuspect |= r[u][2] >> 31;
        best_merge(b, u2d(pw), (r[u][2] & 0x7FFFFFFFu) == 0x7FFFFFFFu ? -1 : (int64_t)(r[u][2] & 0x7FFFFFFFu));
      }
  }
  b = wave_best(b);
  // questions that passed the pole watch (any workgroup's: their entries were in the list before their workgroups reported): the
  // fix launched behind this sweep publishes the result -- pole_kernels.hip
  // A workgroup's question passed the pole watch: the fix launched behind this sweep publishes the result (pole_kernels.hip) -- or,
  // the resident kernel, behind which nothing can be launched: the step's answer says so (index -4; the caller takes the launched path)
  const bool suspects = __any(anySuspect != 0);
  const bool listed = !UNI && suspects && complete && a.poleList != nullptr, lazy = a.fs.lazyFix != 0;   // (lazy: the caller launches the fix when told to -- index -4, as the resident kernel's)
  const bool deferred = listed && !lazy, redo = (UNI && suspects && !a.serverNoWatch) || (listed && lazy);
  if (sampled) {           // the selection follows (sweep_body); only whether the sweep is complete (and whether it publishes) is handed on
    if (allReported != nullptr && (UNI || lane == 0)) { allReported[0] = complete; allReported[1] = deferred; allReported[2] = redo; }
    return;
  }
  if (deferred) return;
  if (UNI || lane == 0) {
    a.fs.out->priority = b.i < 0 ? 0.0 : b.p;
    a.fs.out->index = !complete ? -3 : redo ? -4 : b.i < 0 ? -1 : b.i + a.fs.outBase;
    if (a.fs.seq != nullptr) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");  // system scope: the record is visible to the host before the flag
      __hip_atomic_store(a.fs.seq, flagValue, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    if (a.fs.tagCell != nullptr) {  // every workgroup has read the cell (it published): the next replay gets the next tag
      uint64_t next = seqValue + 1;
      if ((uint32_t)next == 0) next++;
      __hip_atomic_store(a.fs.tagCell, next, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// One element pair of pass 2 (:95-128).  lh: likelihoods, id: 1/D, pr: masked priors.
__device__ __forceinline__ void pass2_pair(double2 lh, double2 id, double2 pr, double invWk, const double *tbl,
                                           double &hW, double &v, double &accL) {
  const double p0 = lh.x * invWk, p1 = lh.y * invWk;           // :97
  const double d0 = p0 - pr.x, d1 = p1 - pr.y;                 // :119 (first: log2hot may then rework p's registers in place)
  const double l20 = log2hot(p0, tbl), l21 = log2hot(p1, tbl); // :106 (gap lanes: p = 0 -> -1023, contributes -0)
  hW = fma(lh.x, l20, hW);                                     // :113-114 weighted by W_k (see eval_epilogue)
  hW = fma(lh.y, l21, hW);
  // :117 lack += invD^2 / log2(p) for both elements over one reciprocal: (ix^2*l21 + iy^2*l20) / (l20*l21), the
  // reciprocal by v_rcp_f64 + one Newton step (2^-48.8, below the sum's own rounding).  The squares are written
  // (ix*l)*ix: as ix^2*l they are invariant over the answers and the compiler hoists 2*NP of them into registers.
  const double prod = l20 * l21;
  double r = __builtin_amdgcn_rcp(prod);
  r = fma(r, fma(-prod, r, 1.0), r);
  const double num = fma(id.y * l20, id.y, (id.x * l21) * id.x);
  accL = fma(num, r, accL);
  v = fma(d0, d0, v);                                          // :126-127
  v = fma(d1, d1, v);
}

// ------------------------------------------------------------------------------------------------------------------
// Rows at the pole of the lack term.  lack = -sum invD^2 / log2(p) (:117) has a pole at p -> 1: with a posterior element at
// p = 1 - 1e-7 the last place of p = l * (1 / W_k) -- i.e. the ORDER W_k was summed in -- moves the priority by 1.6e-9 relative,
// and by more the closer to 1 (a quiz's last states).  The sweep's W_k is a plain per-lane sum and a butterfly; the reference's is
// four serial Kahan lanes down the row (SRAccumVectDbl256.h:40-46) and PreciseSum (:62-92), which no wave reproduces at speed
// (SURVEY F4).  Only the rows AT the pole need it, so the sweep only WATCHES, and what it finds is redone behind it by
// pole_kernels.hip (the whole story is told there).
// The watch costs an integer add and compare per ROW and bar: an element within 2^-10 of 1 is nearly all of W_k, so the lane that
// holds it has a pass-1 sum of at least (1 - 2^-9) of W_k (all terms are >= 0, sums of them only grow); the wider bar -- a quarter of
// W_k -- is for rows whose velocity sum all but vanishes (pole_device.h: kSmallV).  Per lane, on the high words of the two sums, no
// vote and no branch: a row that passes sets its bit in the lane's word of the question, and the lanes that have any OR them into
// LDS at the question's end (sweep_body; the streaming kernel below still votes per row: its rows are long).  (A lane whose several
// elements together hold the row's mass passes too: the fix looks at the row's largest element before it changes anything.)
// Round 4 watched every element PAIR (one v_max3_u32 each and a register for the running maximum: +2 - 5 % on the short-row
// shapes, 3.7 % at 10000 targets) and fixed rows of up to 4096 targets inside the sweep, lists of 62 suspects per workgroup;
// round 5's first form voted per row in front of the W exchange (a branch there: +6 - 8 % at 1000 targets).  Measured on the way and
// dropped: the reference's order for every row at the end of the sweep (5 - 8 sweeps per sweep in a late quiz), the correction
// inside the row loop from the registers (+10 - 50 % in EVERY state: the blocks between pass 1 and pass 2 cost the loop its
// registers), the fix as a called function (+13 %: scratch).
// ------------------------------------------------------------------------------------------------------------------
constexpr double kNearOneShare = 1.0 - 0x1p-9;   // (of W_k, in the streaming kernel's votes: see above; the bar is pole_device.h: kNearOneHi)
constexpr double kQuarterShare = 0.2499;         // ... and the wider watch (a strict compare: a wave of padding lanes, all sums 0, does not pass) for rows with a vanishing velocity sum (pole_device.h: kSmallV)
constexpr int kSusDoubles = 4;                 // LDS: words [0], [1] by question parity (the answer rows that passed the watch, a bit each), [4] the list slot -- an EVEN count of doubles: the LDS priors behind it are read as 16-byte pairs

// LDS-DMA: 16 bytes per lane from global memory straight into LDS, no destination VGPRs (buffer_load_dwordx4 ... offen lds:
// row base in an SGPR descriptor, the lane's 32-bit byte offset in a VGPR -- no 64-bit address pairs either); completion is
// counted by vmcnt but invisible to hipcc's own bookkeeping (wait for it explicitly).  M0 = wave-uniform LDS byte address
// of the destination (16-byte aligned); lane i lands at M0 + 16*i.  Checked beyond 64 KiB by tools/glds_test.hip.
typedef unsigned int dma_rsrc_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ dma_rsrc_t dma_rsrc(const void *row, int64_t bytes) {
  const uint64_t base = (uint64_t)(uintptr_t)row;
  return dma_rsrc_t{(unsigned)__builtin_amdgcn_readfirstlane((unsigned)base),
                    (unsigned)__builtin_amdgcn_readfirstlane((unsigned)(base >> 32)) & 0xFFFFu, (unsigned)bytes, 0x00020000u};
}
__device__ __forceinline__ void dma16(dma_rsrc_t rsrc, unsigned byteOffset, unsigned ldsDst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(byteOffset), "s"(rsrc), "s"(__builtin_amdgcn_readfirstlane(ldsDst)) : "memory");
}

// ------------------------------------------------------------------------------------------------------------------
// Register-resident sweep: WPQ waves per question, NP target pairs per lane.  Requires ldT <= 128*WPQ*NP.
// PRLDS: keep the masked prior vector in LDS instead of registers (long rows: frees 4*NP VGPRs).
//
// The cube is consumed as ONE continuous stream of rows per workgroup -- mD(q), sA(q,0..K-1), mD(q'), sA(q',0) ... --
// through a single row-sized register ring `ring`: the instant a lane has consumed its pair j of the current row it
// re-issues the load of pair j of the NEXT row of the stream into the same registers, across answer and question
// boundaries alike.  So a full row (16*NP*64*WPQ bytes) is always outstanding per workgroup while pass 2 runs, and the
// memory pipe never idles at a row or question boundary.
//
// LDS (doubles): log2 table [2048] | W exchange [2][WPQ] | partials [2][K+2][WPQ] | pending [kPend][2K+3] |
//                running argmax [64][2] | suspects of the pole watch [kSusDoubles] | prior [ldT + 2] if PRLDS, else (KiB-aligned) the mD landing row [NP*64*WPQ pairs] |
//                deferred lane sums [K+2][64*WPQ] (eval_defers_sums) | the resident kernel's request line [8]
//   partial rows: V_k (K rows), sum W_k*H_k (1 row), lack (1 row); the leading [2] alternates per question.
//   pending: per finished question W_k[K], V_k[K], sum WH, lack, question index.  The scalar epilogue (exp2, log,
//   divisions: ~1 us of dependent fp64 code) is not run per question by one lane while 511 wait at the next barrier;
//   finished questions queue up here and wave 0 runs up to kPend epilogues at once, one per lane.
// ------------------------------------------------------------------------------------------------------------------
constexpr int kPend = 32;

// (+ for register-prior shapes: the landing row of the next question's mD, np*64*wpq pairs, on a KiB boundary)
__host__ __device__ constexpr size_t eval_lds_fixed_doubles(int wpq, int64_t K) {
  return kLog2TableDoubles + 2 * (size_t)wpq + 2 * (size_t)(K + 2) * wpq + (size_t)kPend * (2 * (size_t)K + 3) + 2 * kWave + kSusDoubles;
}
static_assert(kSusDoubles % 2 == 0 && kLog2TableDoubles % 2 == 0 && (kPend * 3) % 2 == 0,
              "the LDS priors behind the fixed part are read as 16-byte pairs: an odd count of doubles in front of them halves their rate (10000 x 5 x 10000: 906 -> 1250 us)");
__host__ __device__ constexpr size_t eval_md_row_offset_bytes(int wpq, int64_t K) {
  return (eval_lds_fixed_doubles(wpq, K) * sizeof(double) + 1023) / 1024 * 1024;
}
__host__ __device__ constexpr size_t eval_lds_doubles(int wpq, int64_t K, bool prLds, int64_t ldT) {
  return kLog2TableDoubles + 2 * (size_t)wpq + 2 * (size_t)(K + 2) * wpq + (size_t)kPend * (2 * (size_t)K + 3) + 2 * kWave + kSusDoubles +
         (prLds ? (size_t)ldT + 2 : 0);
}

// Shapes of up to 8 waves keep the lanes' partial sums of a question -- K velocity sums, the entropy sum, the
// lack sum -- in LDS ((K + 2) x threads doubles behind the mD landing row / the LDS priors) and reduce them once per question, all waves
// together, instead of one wave butterfly per sum and row: 18 VALU instructions per row and wave become one ds_write.
// (It costs four registers, which takes the two 4-pair shapes from 165 / 167 to 169 and from three waves per SIMD to two --
// 41 instead of 37 us at 2000 targets: wg256_np4's kernel is held to three waves, eval_questions_f64_occ3 (35.7 us).  The
// two-wave wg128_np4, the shape of batched launches, keeps the per-row butterflies: 92 k vs 84 k selections/s deferred.)
__host__ __device__ constexpr bool eval_defers_sums(int wpq, int np, bool prLds) { return wpq <= 8 && !(wpq == 2 && np == 4); }
__host__ __device__ constexpr size_t eval_deferred_bytes(int wpq, int np, int64_t K, bool prLds) {
  return eval_defers_sums(wpq, np, prLds) ? (size_t)(K + 2) * wpq * kWave * sizeof(double) : 0;
}

// epilogue wave only: one epilogue per lane over the queued questions; each lane keeps the best of its own
__device__ __forceinline__ void flush_pending(const EvalArgs &a, const double *pend, int nPend, int lane, Best &best) {
  if (lane < nPend) {
    const int64_t K = a.K;
    const double *rec = pend + (size_t)lane * (2 * K + 3);
    const int64_t q = reinterpret_cast<const int64_t *>(rec)[2 * K + 2];
    const double pri = eval_epilogue(rec, -rec[2 * K], rec + K, K, rec[2 * K + 1], a.vCompTail);  // :130
    store_priority(a.priority + (q - a.qFirst), pri);
    // hand-over of the priority vector to the host (FusedSelect::hostPriority): every workgroup delivers its own questions as it
    // finishes them, one 16-byte record {priority, launch tag} each, and waits for nothing -- a gather by the finisher's workgroup
    // after the last record (a round of loads past the L2s, a burst over the host link, a fence) put 5 us behind the sweep's 13.5,
    // and every workgroup waiting for its own stores' acknowledgements before reporting still 3.5
    if (a.fs.hostPriority != nullptr && a.fs.sampleSubtasks > 0) {
      typedef unsigned int u4 __attribute__((ext_vector_type(4)));
      const uint64_t w0 = d2u(pri), w1 = a.fs.seqValue;
      const u4 x = {(unsigned)w0, (unsigned)(w0 >> 32), (unsigned)w1, (unsigned)(w1 >> 32)};
      asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(a.fs.hostPriority + (q - a.qFirst)), "v"(x) : "memory");
    }
    best_offer(best, pri, q - a.qFirst);
  }
}

// The sweep of one launch (SERVER = false) or of one step of the resident kernel (SERVER = true: the Log2Hot table is
// already in LDS after the first step, and branches on the wave number are made provably uniform).
// What changes between two steps of the resident kernel without a kernel boundary in between: the quiz's posterior and its
// asked bits (rewritten by RecordAnswer's kernel, possibly on another XCD, whose L2 is not coherent with this one's).  The
// resident kernel reads exactly these with agent-scope (sc1) loads, which are served past the non-coherent cache levels.
// The cube and the gap bitmaps change only in operations that stop the resident kernel first (hip_engine_server.cpp: StopServer).
// Tried instead: an acquire fence in every wave at the start of a step (what a kernel boundary does) -- correct, 44 us per
// step against 22; per-XCD copies made by one leader workgroup per XCD and read with workgroup-scope (sc0) loads -- no
// faster (the step is bound by pulling the 48 MB cube through the L2s, ~13 us, not by these 8 KB) and not coherent.
__device__ __forceinline__ double uniform_double(double x) {   // a wave-uniform value, into scalar registers
  const uint64_t u = d2u(x);
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)u), hi = __builtin_amdgcn_readfirstlane((uint32_t)(u >> 32));
  return u2d(((uint64_t)hi << 32) | lo);
}
template <bool COH>
__device__ __forceinline__ uint32_t load_word(const uint32_t *p) {
  if constexpr (COH) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else return *p;
}
template <bool COH>
__device__ __forceinline__ double2 load_pair(const double2 *p) {
  if constexpr (COH) {
    const double *d = reinterpret_cast<const double *>(p);
    return make_double2(__hip_atomic_load(d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT),
                        __hip_atomic_load(d + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
  } else {
    return *p;
  }
}

// FUSE: RecordAnswer's posterior update in the prologue (single-quiz launches of the register-prior shapes with rows of up to 1024
// targets).  A lone client's critical path is RecordAnswer -> posterior kernel -> the sweep of the NextQuestion that follows
// (launched speculatively right behind it): the sweep cannot start before the posterior kernel has finished, 5 us of kernel
// and a dispatch gap for 8 KB of arithmetic.  Here EVERY workgroup of the sweep computes the posterior itself -- the old prior and
// the answered question's two rows are 24 KB from L2, the element operations and the reference-order sum are those of
// record_answer_body, so every workgroup gets the same bits that kernel would have written -- and sweeps with it; the last
// workgroup also lists the posterior's best targets right away (ListTopTargets is the client's next call), and workgroup 0, as the
// finisher that has seen every workgroup's record (so nobody reads the old prior any more), stores the posterior over the old prior
// at the end.
template <int WPQ, int NP, bool PRLDS, bool SERVER, bool DEFER, bool FUSE = false, bool POLE = false, int KC = 0>
__device__ __forceinline__ void sweep_body(EvalArgs a, bool copyTable) {
  constexpr int kThreads = WPQ * kWave;
  constexpr int NPR = PRLDS ? 1 : NP;
  constexpr bool kStreamHint = !SERVER && WPQ >= 8;            // the shapes for rows beyond 4096 targets: see row_load
  extern __shared__ double smem[];
  const int64_t ldT = a.ldT, K = KC > 0 ? KC : a.K;             // (KC: the answer count as a constant)
  double *tbl = smem;
  if (!lds_table_at_zero(tbl)) __builtin_trap();  // log2hot addresses the table absolutely
  double *redW = tbl + kLog2TableDoubles;
  double *partAll = redW + 2 * WPQ;
  double *pend = partAll + 2 * (K + 2) * WPQ;
  Best *bestLds = reinterpret_cast<Best *>(pend + kPend * (2 * K + 3));  // wave 0's per-lane running argmax
  uint32_t *susWords = reinterpret_cast<uint32_t *>(bestLds + kWave);   // [0], [1] by question parity: the listed rows (low half) and the rows with an element of a quarter (high half); [4]: the entry's place in the list
  double2 *prLds = reinterpret_cast<double2 *>(reinterpret_cast<double *>(bestLds + kWave) + kSusDoubles);
  // landing row of the NEXT question's mD (register-prior shapes): lane-private 16-byte slots, slot j of thread tid at
  // mdRow[j*kThreads + tid]; filled by LDS-DMA while the current question's last answer is in pass 2
  constexpr bool kMdLds = !PRLDS;
  double2 *mdRow = reinterpret_cast<double2 *>(reinterpret_cast<char *>(smem) + eval_md_row_offset_bytes(WPQ, K));
  const int nPart = (int)(K + 2);
  const int recLen = (int)(2 * K + 3);
  int tidRaw = threadIdx.x;
  // resident kernel: nothing derived from the thread index may be hoisted out of the step loop (56 VGPRs of loop
  // invariants otherwise, and a workgroup less per CU)
  if constexpr (SERVER) asm volatile("" : "+v"(tidRaw));
  const int tid = tidRaw, lane = tid % kWave;
  const int wave = SERVER ? (int)__builtin_amdgcn_readfirstlane(tid / kWave) : tid / kWave;
  const unsigned mdRowWaveAddr = (unsigned)(uintptr_t)mdRow + (unsigned)wave * 1024u;
  constexpr bool kDefer = DEFER;
  double *vdump = PRLDS ? reinterpret_cast<double *>(prLds) + ldT + 2
                        : reinterpret_cast<double *>(reinterpret_cast<char *>(mdRow) + (size_t)NP * kThreads * 16);   // kDefer only
  // ---- prologue.  Everything it needs from memory is independent of everything else, so all of it is requested
  // before anything is used: one memory round trip instead of a dozen dependent ones (with one question per workgroup,
  // as at 1000 x 1000, the prologue is on the critical path of the whole launch).  That includes the mD row of the first
  // candidate question, requested before its gap / asked bits are known (a skipped candidate costs one wasted row).
  const int nPairs = (int)(ldT >> 1);
  const int64_t qStride = (K + 1) * ldT, rowBytes = ldT * 8;
  uint32_t poff[NP];     // byte offset of the lane's pair j within a row (see row_load)
  uint32_t gapWord[NP];
  double2 pr[NPR], prRaw[NP];
  double2 invD[NP], ring[NP];
  const int64_t q0 = a.qFirst + blockIdx.x;
  const bool haveQ0 = q0 < a.qLimit;
  const int64_t q0c = haveQ0 ? q0 : a.qLimit - 1;  // (unconditional loads: a branch here would end the batch)
  const uint32_t q0Gap = a.qgap[q0c >> 5], q0Asked = load_word<SERVER>(a.asked + (q0c >> 5));
#pragma unroll
  for (int j = 0; j < NP; j++) {
    const int p = tid + j * kThreads;
    const int pc = p < nPairs ? p : (nPairs - 1);  // clamped: out-of-row lanes re-read the last pair and are masked
    poff[j] = (uint32_t)pc * 16u;
    gapWord[j] = a.tgap[pc >> 4];                  // both bits of the pair (targets 2pc, 2pc+1) sit in one word
    prRaw[j] = load_pair<SERVER>(reinterpret_cast<const double2 *>(a.prior) + pc);
  }
  auto head_of_stream = [&](int64_t qq) __attribute__((always_inline)) {
    if constexpr (kMdLds) {   // mD to its LDS landing row, the first answer row to the ring
      const RowRsrc rowA = row_rsrc(a.cube + qq * qStride, rowBytes);
      const dma_rsrc_t md = dma_rsrc(a.cube + qq * qStride + K * ldT, rowBytes);
#pragma unroll
      for (int j = 0; j < NP; j++) {
        ring[j] = row_load<kStreamHint>(rowA, poff[j]);
        dma16(md, poff[j], mdRowWaveAddr + (unsigned)j * (kThreads * 16u));
      }
    } else {
      const RowRsrc rowD = row_rsrc(a.cube + qq * qStride + K * ldT, rowBytes);
#pragma unroll
      for (int j = 0; j < NP; j++) ring[j] = row_load<kStreamHint>(rowD, poff[j]);
    }
  };
  if (haveQ0) head_of_stream(q0);
  static_assert((kLog2TableDoubles / 2) % kThreads == 0, "the table copy is an exact number of 16-byte loads per thread");
  constexpr int kTblPerThread = kLog2TableDoubles / 2 / kThreads;
  double2 tv[kTblPerThread];
  if constexpr (!FUSE) {
    if (copyTable) {
      const double2 *src = reinterpret_cast<const double2 *>(gLog2Table);
#pragma unroll
      for (int i = 0; i < kTblPerThread; i++) tv[i] = src[tid + i * kThreads];
#pragma unroll
      for (int i = 0; i < kTblPerThread; i++) reinterpret_cast<double2 *>(tbl)[tid + i * kThreads] = tv[i];
    }
  }
  if constexpr (FUSE) {
    static_assert(!PRLDS && !SERVER && NP * kThreads <= 512, "rows of up to 1024 targets: the table's LDS holds the update's scratch first");
    // scratch in the LDS the Log2Hot table takes afterwards: the un-normalised values [1024] | the sum's partials [8 * 51 + 1] | the listing's scratch
    double *stage = tbl, *sums = tbl + 1024;
    TopScratch *topScratch = reinterpret_cast<TopScratch *>(tbl + 1440);
    static_assert(1440 * sizeof(double) + sizeof(TopScratch) <= kLog2TableDoubles * sizeof(double), "update scratch within the table's LDS");
    double2 av[NP], dv[NP];
    const RowRsrc ra = row_rsrc(a.updRowA, rowBytes), rd = row_rsrc(a.updRowD, rowBytes);
#pragma unroll
    for (int j = 0; j < NP; j++) { av[j] = row_load(ra, poff[j]); dv[j] = row_load(rd, poff[j]); }   // (every workgroup reads these two rows: no streaming hint)
#pragma unroll
    for (int j = 0; j < NP; j++) {
      const int p = tid + j * kThreads;
      if (p < nPairs) {
        const int sh = (2 * p) & 31;
        const double x0 = prRaw[j].x * (av[j].x / dv[j].x), x1 = prRaw[j].y * (av[j].y / dv[j].y);   // CERecordAnswerSubtaskMul.cpp:31, :34
        stage[2 * p] = ((gapWord[j] >> sh) & 1u) ? 0.0 : x0;                                            // :35-37
        stage[2 * p + 1] = ((gapWord[j] >> (sh + 1)) & 1u) ? 0.0 : x1;
      }
    }
    const double total = reference_order_sum<true>(stage, a.updVects, a.updWorkers, sums);   // (sixteen values of a chain requested at once: its 17 LDS round trips in a row were 1 us)
    // the table, requested now (held across the whole update its values went through scratch memory, 12 MB per launch): it
    // arrives while the divisions run
    const double2 *tblSrc = reinterpret_cast<const double2 *>(gLog2Table);
    const double2 tv0 = tblSrc[tid], tv1 = tblSrc[tid + kThreads], tv2 = tblSrc[tid + 2 * kThreads], tv3 = tblSrc[tid + 3 * kThreads];
    static_assert(kTblPerThread == 4, "four 16-byte table loads per thread");
#pragma unroll
    for (int j = 0; j < NP; j++) {
      const int p = tid + j * kThreads;
      if (p < nPairs) {
        double2 v = make_double2(stage[2 * p], stage[2 * p + 1]);
        if (2 * p < 4 * a.updVects) v.x = v.x / total;                                                  // CEDivTargPriors :19
        if (2 * p + 1 < 4 * a.updVects) v.y = v.y / total;
        prRaw[j] = v;
        if (blockIdx.x == gridDim.x - 1) { stage[2 * p] = v.x; stage[2 * p + 1] = v.y; }   // (the thread's own elements)
      }
    }
    // the listing is the LAST workgroup's: of a grid that strides over the questions it has the fewest, while workgroup 0 -- the
    // finisher -- has the most and is what the launch waits for
    if (blockIdx.x == gridDim.x - 1) {
      if (tid == 0) const_cast<uint32_t *>(a.asked)[a.updQuestion >> 5] |= 1u << (a.updQuestion & 31);   // CEQuiz::RecordAnswer, PqaCore/CEQuiz.h:92
      if (a.updTop.count > 0) {
        __syncthreads();
        top_targets_publish<true>(stage, a.tgap, a.updT, a.updTop.count, a.updTop.out, a.updTop.nOut, a.updTop.flag, a.updTop.flagValue, topScratch);
      }
    }
    __syncthreads();   // the scratch is the table's from here on
    double2 *tblDst = reinterpret_cast<double2 *>(tbl);
    tblDst[tid] = tv0; tblDst[tid + kThreads] = tv1; tblDst[tid + 2 * kThreads] = tv2; tblDst[tid + 3 * kThreads] = tv3;
  }
  // Per-lane constants of the sweep: masked priors and gap flags of the lane's targets.
  uint32_t gapBits = 0;  // bit 2j / 2j+1 : target pair j element 0 / 1 is a gap (or beyond the row)
#pragma unroll
  for (int j = 0; j < NP; j++) {
    const int p = tid + j * kThreads;
    const bool inRow = p < nPairs;
    const int sh = (2 * (inRow ? p : nPairs - 1)) & 31;
    const bool g0 = !inRow || ((gapWord[j] >> sh) & 1u), g1 = !inRow || ((gapWord[j] >> (sh + 1)) & 1u);
    gapBits |= (g0 ? 1u : 0u) << (2 * j) | (g1 ? 1u : 0u) << (2 * j + 1);
    double2 pv = prRaw[j];
    pv.x = g0 ? 0.0 : pv.x;                        // :103 andnot(gapMask, prior)
    pv.y = g1 ? 0.0 : pv.y;
    if constexpr (PRLDS) { if (inRow) prLds[p] = pv; } else { pr[j] = pv; }
  }
  // out-of-row lanes read the all-zero pair stored right after the row (their cube loads are clamped instead)
  if constexpr (PRLDS) { if (tid == 0) prLds[nPairs] = make_double2(0.0, 0.0); }

  auto next_valid = [&](int64_t q) {               // :54 gap / asked questions get priority 0 and leave the stream
    while (q < a.qLimit && (bit_test(a.qgap, q) || ((load_word<SERVER>(a.asked + (q >> 5)) >> (q & 31)) & 1u) || (FUSE && q == a.updQuestion))) {
      if (tid == 0) store_priority(a.priority + (q - a.qFirst), 0.0);
      q += gridDim.x;
    }
    return q;
  };
  if (tid < 4) susWords[tid] = tid < 2 ? 0u : kGapNoneBits;   // ([2], [3]: the smallest gap of the question's listed rows, by parity -- EvalArgs::poleGate)
  int64_t q = q0;
  if (haveQ0 && ((((q0Gap | q0Asked) >> (q0 & 31)) & 1u) || (FUSE && q0 == a.updQuestion))) {     // the first candidate is skipped: restart the stream
    q = next_valid(q0);
    if (q < a.qLimit) head_of_stream(q);
  }
  __syncthreads();

  int phase = 0, qpar = 0, nPend = 0;
  // The pole watch (see above; template POLE: the launcher takes the variant when the engine gave it KbView::poleList -- option
  // pole_fix, default on).  The resident kernel watches too (EvalArgs::serverWatch); nothing can be launched behind a step, so a
  // step that met such a row answers "redo" (index -4, fused_select) and the host launches that quiz's selections from there on.
  constexpr bool kWatch = POLE;
  // (the resident kernel only has to know WHETHER a step met such a row, so its lanes decide for themselves and keep one bit over the
  //  whole step: a lane's sum of nearly all of W_k, or of a quarter of it while the lane's OWN share of the row's velocity sum is
  //  within kSmallV -- if the row's sum is, so is the share of the lane that holds the element: every row the launched sweeps would
  //  list passes, few others do -- and the workgroup asks once, at the step's end; the per-question bookkeeping below cost its step 0.5 us)
  constexpr bool kListWatch = kWatch && !SERVER;
  [[maybe_unused]] int32_t stepNear = INT32_MIN, stepSmall = INT32_MIN;   // (the largest hi(lane's sum) - hi(W_k) of the step: over all rows / over the rows with the lane's velocity share within kSmallV)
  const bool watchOn = SERVER ? a.serverWatch : a.poleList != nullptr;
  bool wgSuspect = false;                                     // (a question of this workgroup has passed: into its record)
  if (wave == 0) bestLds[lane] = Best{0.0, -1};   // only wave 0 ever touches these
  // mD in the ring (the shapes with the priors in LDS: no room for a landing row): 1/D pair by pair, each consumed pair refilled from
  // the question's first answer row.  At the head of a question that row's whole latency stands between 1/D and pass 1 -- so, with
  // the answer count a constant (the rows' loop unrolled: no branch in the pairs' loop, the ring no phi), every question but a
  // workgroup's first has it done inside the LAST answer's pass 2 of the question before it (kHeadInTail): behind pair j's pass 2
  // the old 1/D of the pair is dead and the next question's mD pair, requested during that answer's pass 1, has long arrived; the
  // first answer row then has the rest of that pass 2 and the question's reduction to arrive in.
  constexpr bool kHeadInTail = !kMdLds && KC > 0;
  [[maybe_unused]] bool firstOfStream = true;
  auto head_pair = [&](const RowRsrc &rowA, int j) __attribute__((always_inline)) {
    invD[j].x = ((gapBits >> (2 * j)) & 1) ? 0.0 : div_nr(1.0, ring[j].x);      // :74 andnot(gapMask, 1/D)
    invD[j].y = ((gapBits >> (2 * j + 1)) & 1) ? 0.0 : div_nr(1.0, ring[j].y);
    ring[j] = row_load<kStreamHint>(rowA, poff[j]);
  };
  auto head_from_ring = [&](const double *qBase) __attribute__((always_inline)) {
    const RowRsrc rowA = row_rsrc(qBase, rowBytes);
#pragma unroll
    for (int j = 0; j < NP; j++) {
      head_pair(rowA, j);
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  while (q < a.qLimit) {
    const int64_t qn = next_valid(q + gridDim.x);
    const double *qBase = a.cube + q * qStride;
    if constexpr (kMdLds) {
      // mD arrived in LDS while the previous question finished, and the ring already holds this question's first answer
      // (both requested during that question's last pass 1): nothing is waited for here.  With mD in the ring instead,
      // the first answer row could only be requested now and its whole latency (~9k cycles of a 55k-cycle question at
      // 10000 targets) stood between 1/D and pass 1.
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
      for (int j = 0; j < NP; j++) {
        const double2 dv = mdRow[j * kThreads + tid];
        invD[j].x = ((gapBits >> (2 * j)) & 1) ? 0.0 : div_nr(1.0, dv.x);           // :74 andnot(gapMask, 1/D)
        invD[j].y = ((gapBits >> (2 * j + 1)) & 1) ? 0.0 : div_nr(1.0, dv.y);
        __builtin_amdgcn_sched_barrier(0);
      }
    } else if (!kHeadInTail || firstOfStream) {
      head_from_ring(qBase);
      firstOfStream = false;
    }
    double *part = partAll + qpar * (nPart * WPQ);
    double *rec = pend + nPend * recLen;
    double accL = 0, hW = 0;
    [[maybe_unused]] int32_t rowGap = INT32_MIN;
    [[maybe_unused]] uint32_t watchRows = 0;                   // (pole watch, per lane: the rows in which this lane's sum is nearly all of W_k below, a quarter of it above)
#pragma unroll KC > 0 ? KC : 1
    for (int64_t k = 0; k < K; k++) {
      // ---- pass 1 (:66-87): likelihoods into registers, W_k; each consumed pair is refilled from the next stream row
      double2 lh[NP];
      double s0 = 0, s1 = 0;
      // next row of the stream: the next answer, else the next question's mD row; at the very end of the stream the
      // (cache-resident, row-sized) prior vector stands in, so that the refill stays unconditional -- a conditional
      // refill makes the ring a phi and costs a full vmcnt(0) + 2*NP moves per row
      const bool lastRow = k + 1 == K;
      const bool moreQuestions = qn < a.qLimit;
      // (register-prior shapes: the row after a question's last answer is the next question's FIRST answer; its mD row goes
      //  to LDS, below)
      const double *rowNext = !lastRow ? qBase + (k + 1) * ldT
                              : !moreQuestions ? a.prior
                              : a.cube + qn * qStride + (kMdLds ? 0 : K * ldT);
      const RowRsrc rowN = row_rsrc(rowNext, rowBytes);
#pragma unroll
      for (int j = 0; j < NP; j++) {
        double2 pv;
        if constexpr (PRLDS) pv = prLds[min(tid + j * kThreads, nPairs)]; else pv = pr[j];
        lh[j].x = (ring[j].x * invD[j].x) * pv.x;              // :81-82 (gap lanes: invD = 0 and prior = 0)
        lh[j].y = (ring[j].y * invD[j].y) * pv.y;
        s0 += lh[j].x;  // <= 2*NP terms per lane: plain sums, then the butterfly -- a 64*WPQ-leaf pairwise tree
        s1 += lh[j].y;
#ifdef PQA_ABLATE_REFILL   // measurement only (wrong values): the stream's rows are not loaded -- what the sweep costs without its memory
        asm volatile("" : "+v"(ring[j].x), "+v"(ring[j].y));
#else
        ring[j] = row_load<kStreamHint>(rowN, poff[j]);
#endif
        __builtin_amdgcn_sched_barrier(0);
      }
      if (kMdLds && lastRow && moreQuestions) {
        // after the loop, not inside it: hipcc does not count these, and a counted wait for a ring pair with fresh DMA
        // requests behind it would wait for them too
        const dma_rsrc_t mdNext = dma_rsrc(a.cube + qn * qStride + K * ldT, rowBytes);
#pragma unroll
        for (int j = 0; j < NP; j++) dma16(mdNext, poff[j], mdRowWaveAddr + (unsigned)j * (kThreads * 16u));
      }
      const double sLane = s0 + s1;
      double Wk = wave_sum(sLane);                             // :88
      if constexpr (WPQ > 1) {
        double *buf = redW + phase * WPQ;
        if (lane == 0) buf[wave] = Wk;
        __syncthreads();
        Wk = row_sum<WPQ>(buf[lane % WPQ]);
        phase ^= 1;
      }
      if constexpr (kWatch) {
        // Does this lane's sum reach a quarter of W_k / nearly all of it (a share of 1 - 2^-9)?  On the high words -- an integer add
        // and compare each, the bars a little (2^-9 ... 2^-8) on the generous side; lanes of padding, sum 0, do not pass -- per lane and
        // without a branch: a bit per row in one word per question (the rows with an element of a quarter above, the rows to list
        // below), OR-ed into LDS at the question's end by the lanes that have any.  (A vote and a branch per row in front of the
        // exchange cost the launched 1000-target sweep 6 - 8 %; this form nothing measurable there, ~0.5 us of the resident step's 16.)
        const uint32_t hs = (uint32_t)(d2u(sLane) >> 32), hw = (uint32_t)(d2u(Wk) >> 32);
        if constexpr (SERVER) {
          rowGap = (int32_t)(hs - hw);                         // (high words of positive doubles: no overflow; both bars are looked at once, at the step's end)
          stepNear = max(stepNear, rowGap);
        } else {
          const uint32_t kb = (uint32_t)(k < 15 ? k : 15);
          watchRows |= (hs + 0x00201000u >= hw ? 0x10000u << kb : 0u) | (hs + 0x00001000u >= hw ? 1u << kb : 0u);   // (per lane: no vote, no scalar result to wait for)
        }
      }
      const double invWk = div_nr(1.0, Wk);                    // :91
      if constexpr (kListWatch) {
        // Gated fix (pole_kernels.hip): the row's largest element is at most this lane's sum, so 1 - p >= (W_k - sum) / W_k -- an exact
        // difference where it matters.  A wave-uniform branch (launches that do not gate skip it) around a rare divergent one: the lane
        // whose sum is nearly all of W_k lowers the question's word right here -- nothing is carried over the rows.  (Behind the row's
        // exchange barrier: the word was reset for this question before any wave passed it.)
        if (a.poleGate) {
          if ((uint32_t)(d2u(sLane) >> 32) + 0x00001000u >= (uint32_t)(d2u(Wk) >> 32))
            atomicMin(&susWords[2 + qpar], pole_gap_bits((Wk - sLane) * invWk));
        }
      }
      // ---- pass 2 (:95-128)
      double v = 0;
      // (kHeadInTail, the last answer: the next question's 1/D and the request for its first answer row, pair by pair behind this
      //  pass 2 -- where no question follows, the prior vector stands in for both rows, as it does for the refill above)
      [[maybe_unused]] const RowRsrc rowHead = row_rsrc(moreQuestions ? a.cube + qn * qStride : a.prior, rowBytes);
#pragma unroll
      for (int j = 0; j < NP; j++) {
        double2 pv;
        if constexpr (PRLDS) pv = prLds[min(tid + j * kThreads, nPairs)]; else pv = pr[j];
        pass2_pair(lh[j], invD[j], pv, invWk, tbl, hW, v, accL);
        // Pin the accumulators here: without an opaque use the compiler sinks the whole lack chain (and every log2 it
        // needs) below the loop, which costs 8 live VGPRs per pair; and keep the interleave to one pair at a time.
        asm volatile("" : "+v"(accL), "+v"(hW), "+v"(v));
        if constexpr (kHeadInTail) { if (lastRow) head_pair(rowHead, j); }
        __builtin_amdgcn_sched_barrier(0);
      }
      if constexpr (kWatch && SERVER) stepSmall = max(stepSmall, v <= kSmallV ? rowGap : INT32_MIN);
      if constexpr (kDefer) {
        vdump[k * kThreads + tid] = v;                         // :132, reduced with the question's other sums below
        if (tid == 0) rec[k] = Wk;                             // :90
      } else {
        v = wave_sum(v);
        if (lane == 0) {
          if (wave == 0) rec[k] = Wk;                           // :90
          part[k * WPQ + wave] = v;                              // :132
        }
      }
    }
    bool suspect = false;   // workgroup-uniform: a row of this question passed the pole watch
    if constexpr (kDefer) {
      // (the dump is single-buffered: a wave that runs ahead writes it again only behind the next question's first W
      //  barrier, which no wave passes before every wave has read here)
      vdump[K * kThreads + tid] = hW;
      vdump[(K + 1) * kThreads + tid] = accL;
      if constexpr (kListWatch) {                                   // (rare)
        if (watchRows != 0) atomicOr(&susWords[qpar], watchRows);   // (the lanes that hold such a sum: one or two of a late quiz's wave)
      }
      __syncthreads();
      uint32_t wideRows = 0;                                     // workgroup-uniform: the rows with an element of a quarter
      if constexpr (kListWatch) {
        const uint32_t w = watchOn ? susWords[qpar] : 0u;
        suspect = (w & 0xFFFFu) != 0;
        wideRows = w >> 16;
      }
      // every 32 lanes take one of the K + 2 sums: threads/32 partials each, then a 32-lane butterfly
      constexpr int kGroups = kThreads / 32;
      const int l32 = tid & 31;
      for (int r = tid >> 5; r < nPart; r += kGroups) {
        const double *src = vdump + r * kThreads + l32;
        double acc = src[0];
#pragma unroll
        for (int i = 1; i < kGroups; i++) acc += src[32 * i];
        acc += mov_dpp<kDppXor1>(acc);
        acc += mov_dpp<kDppXor2>(acc);
        acc += mov_dpp<kDppHalfMirror>(acc);
        acc += mov_dpp<kDppMirror>(acc);
        const Pair p = swap16(acc);
        acc = p.a + p.b;
        if constexpr (kListWatch) {                                  // (a row whose velocity sum all but vanishes, with an element of a quarter: pole_device.h)
          if (l32 == 0 && r < K && acc <= kSmallV && ((wideRows >> (r < 15 ? r : 15)) & 1u)) atomicOr(&susWords[qpar], 1u << (r < 15 ? r : 15));
        }
        if (r < K) acc = rec[r] * sqrt(acc);                   // :156-157
        if (l32 == 0) rec[K + r] = acc;
      }
      if constexpr (kListWatch) {
        if (wideRows != 0) { __syncthreads(); suspect = (susWords[qpar] & 0xFFFFu) != 0; }   // (the bits the other waves' lanes have just set)
        if constexpr (!SERVER) { if (a.poleNoFollow) suspect = false; }   // (measurement hook: the watch runs, nothing is listed or deferred -- no fix follows)
      }
      if (tid == 0) {
        reinterpret_cast<int64_t *>(rec)[2 * K + 2] = q;
        if constexpr (kListWatch) { susWords[qpar ^ 1] = 0; susWords[2 + (qpar ^ 1)] = kGapNoneBits; }   // (the other parity's flags: read by everybody before this question's barrier, set again only behind the next question's)
        if constexpr (!SERVER) {
          if (suspect) susWords[4] = pole_list_append(a.poleList, (uint32_t)(q - a.qFirst), K <= 15 ? susWords[qpar] & 0xFFFFu : 0u, a.slots != nullptr ? blockIdx.y : 0u,
                                                      a.poleGate ? susWords[2 + qpar] : 0u);
        }
      }
      wgSuspect = wgSuspect || suspect;
      if (!SERVER && suspect) {
        // the question's sums as they are, for the fix behind the sweep (the question is queued like any other: its priority stands until
        // then): by question, or -- the quizzes of a grid.y launch share the buffer -- by its entry in the list
        __syncthreads();
        const int64_t at = a.slots != nullptr ? (int64_t)susWords[4] : q - a.qFirst;
        for (int i = tid; i < 2 * (int)K + 2; i += kThreads) a.poleScratch[at * (2 * K + 2) + i] = rec[i];
      }
      if (nPend + 1 == kPend) {
        __syncthreads();                                       // the records of other waves
        if (wave == 0) flush_pending(a, pend, kPend, lane, bestLds[lane]);
      }
    } else {
      hW = wave_sum(hW);
      accL = wave_sum(accL);
      if (lane == 0) {
        part[K * WPQ + wave] = hW;
        part[(K + 1) * WPQ + wave] = accL;
      }
      if constexpr (kListWatch) {                                   // (rare)
        if (watchRows != 0) atomicOr(&susWords[qpar], watchRows);   // (the lanes that hold such a sum: one or two of a late quiz's wave)
      }
      if constexpr (WPQ > 1) __syncthreads();
      uint32_t wideRows = 0;
      if constexpr (kListWatch) {
        const uint32_t w = watchOn ? susWords[qpar] : 0u;
        suspect = (w & 0xFFFFu) != 0;
        wideRows = w >> 16;
      }
      if (wave == 0) {
        // combine the waves' partials in wave order, one partial row per lane, and queue the question
        for (int r = lane; r < nPart; r += kWave) {
          double acc = part[r * WPQ];
          for (int w2 = 1; w2 < WPQ; w2++) acc += part[r * WPQ + w2];
          if constexpr (kListWatch) {                                // (a row whose velocity sum all but vanishes, with an element of a quarter)
            if (r < K && acc <= kSmallV && ((wideRows >> (r < 15 ? r : 15)) & 1u)) atomicOr(&susWords[qpar], 1u << (r < 15 ? r : 15));
          }
          if (r < K) acc = rec[r] * sqrt(acc);                   // :156-157, one answer per lane
          rec[K + r] = acc;
        }
        if constexpr (kListWatch) { if (wideRows != 0) suspect = (susWords[qpar] & 0xFFFFu) != 0; }   // (this wave's own atomics: in order)
        if constexpr (kListWatch && !SERVER) { if (a.poleNoFollow) suspect = false; }   // (measurement hook, as above)
        if (lane == 0) {
          reinterpret_cast<int64_t *>(rec)[2 * K + 2] = q;
          if constexpr (kListWatch) { susWords[qpar ^ 1] = 0; susWords[2 + (qpar ^ 1)] = kGapNoneBits; }
          if constexpr (!SERVER) {
            if (suspect) susWords[4] = pole_list_append(a.poleList, (uint32_t)(q - a.qFirst), K <= 15 ? susWords[qpar] & 0xFFFFu : 0u, a.slots != nullptr ? blockIdx.y : 0u,
                                                        a.poleGate ? susWords[2 + qpar] : 0u);
          }
        }
        wgSuspect = wgSuspect || suspect;
        if (!SERVER && suspect) {   // (the record is this wave's own work: no barrier)
          const int64_t at = a.slots != nullptr ? (int64_t)susWords[4] : q - a.qFirst;
          for (int i = lane; i < 2 * (int)K + 2; i += kWave) a.poleScratch[at * (2 * K + 2) + i] = rec[i];
        }
        if (nPend + 1 == kPend) flush_pending(a, pend, kPend, lane, bestLds[lane]);
      }
    }
    nPend = nPend + 1 == kPend ? 0 : nPend + 1;
    qpar ^= 1;
    q = qn;
  }
  if constexpr (kWatch && SERVER) { if (stepNear >= -0x00001000 || stepSmall >= -0x00201000) susWords[0] = 1u; }   // (rare; any lane of any wave: the bars of the launched sweeps' watch)
  if constexpr (kDefer || (kWatch && SERVER)) __syncthreads();   // the last questions' records, written by other waves
  if constexpr (kWatch && SERVER) {
    if (wave == 0) { wgSuspect = watchOn && susWords[0] != 0u; if (lane == 0) susWords[0] = 0u; }   // (the next step's lanes write behind its rows' barriers)
  }
  bool *allReported = reinterpret_cast<bool *>(redW);         // (the W exchange buffer is free now)
  if (wave == 0) flush_pending(a, pend, nPend, lane, bestLds[lane]);
  if (wave == 0) fused_select<SERVER>(a, bestLds[lane], lane, allReported, wgSuspect);
  if constexpr (FUSE) {
    if (blockIdx.x == 0) {
      // every workgroup has reported (fused_select above has seen their records): the old prior has no readers left
      __syncthreads();
#pragma unroll
      for (int j = 0; j < NP; j++) {
        const int p = tid + j * kThreads;
        if (p < nPairs) reinterpret_cast<double2 *>(const_cast<double *>(a.prior))[p] = pr[j];
      }
    }
  }
  if (a.fs.scratch != nullptr && a.fs.sampleSubtasks > 0 && a.fs.hostPriority != nullptr) {
    // ---- the reference's selector on the HOST: the priorities are on their way to host-coherent memory (flush_pending: tagged
    // records) and fused_select above has seen every workgroup's record, so workgroup 0 only raises the flag.  Launched and
    // resident form alike (resident: the stores of the last step are made by the whole wave).  Asked and gap questions have no
    // entry: the selector skips them by the bitmaps.
    if (blockIdx.x == 0) {
      __syncthreads();
      const bool complete = allReported[0], deferred = allReported[1], redo = allReported[2];
      if (!deferred && (SERVER ? wave == 0 : tid == 0)) {
        a.fs.out->priority = 0.0;
        a.fs.out->index = !complete ? -3 : redo ? -4 : 0;
        if (a.fs.seq != nullptr) {
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
          __hip_atomic_store(a.fs.seq, a.fs.flagValue, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
      }
    }
    return;
  }
  if constexpr (!SERVER) {
    if (a.fs.scratch != nullptr && a.fs.sampleSubtasks > 0 && blockIdx.x == 0) {
      // ---- the reference's selector, by this workgroup, over what every workgroup has written (fused_select above made
      // sure they all have); the Log2Hot table's LDS holds the subtask totals
      __syncthreads();
      const bool complete = allReported[0];
      if (allReported[1]) return;                              // (deferred to the fix behind the sweep: fused_select)
      const SampledPick r = select_sampled_wg_lds<true>(a.priority, a.qgap, a.asked, a.qFirst, a.qLimit - a.qFirst,
                                                        a.fs.sampleSubtasks, a.fs.sampleRnd, tbl);
      if (tid == 0) {
        a.fs.out->priority = r.priority;
        a.fs.out->index = complete ? r.index + a.fs.outBase : -3;
        if (a.fs.seq != nullptr) {
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
          __hip_atomic_store(a.fs.seq, a.fs.flagValue, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
      }
    }
  }
}

// DEFER: the question's lane sums leave the row loop (eval_defers_sums) -- the default where the shape has a deferred form
// and the (K + 2) x threads doubles fit beside the rest of its LDS; a knowledge base with dozens of answers per question
// falls back to the form without.
// KC (here and in the kernels below): the answer count as a constant -- five, where the caller's cube has five answers per question
// (every configuration of BASELINE.json): the loop over the answers has no trip-count registers and is unrolled.  Round 5, same box,
// launched sweep: 1000 targets (four waves of two pairs) 14.1 -> 13.2 us (the resident step 16.3-16.7 -> 15.1-15.6), 1500 29.1 -> 27.3,
// 3000 108 -> 96, 4000 179 -> 158, 6000 388 -> 328, 8000 653 -> 556 us (-15 %); the shapes with the priors in LDS -1 % or nothing.
template <int WPQ, int NP, bool PRLDS, bool DEFER, bool POLE = false, int KC = 0>
__global__ __launch_bounds__(WPQ * 64) void eval_questions_f64(EvalArgs a) {
  select_quiz(a);
  sweep_body<WPQ, NP, PRLDS, false, DEFER, false, POLE, KC>(a, true);
}
// The same kernel held to three waves per SIMD (168 VGPRs): the two 4-pair shapes need 169 with the deferred sums, and a
// register spilled costs them less than a wave of occupancy does.
template <int WPQ, int NP, bool PRLDS, bool DEFER, bool POLE = false>
__global__ __launch_bounds__(WPQ * 64) __attribute__((amdgpu_waves_per_eu(3, 3))) void eval_questions_f64_occ3(EvalArgs a) {
  select_quiz(a);
  sweep_body<WPQ, NP, PRLDS, false, DEFER, false, POLE>(a, true);
}

// RecordAnswer's posterior update + the sweep of the NextQuestion that follows, one launch (sweep_body: FUSE)
template <int WPQ, int NP, bool DEFER, bool POLE = false, int KC = 0>
__global__ __launch_bounds__(WPQ * 64) void eval_questions_f64_upd(EvalArgs a) {
  sweep_body<WPQ, NP, false, false, DEFER, true, POLE, KC>(a, true);
}

// ------------------------------------------------------------------------------------------------------------------
// Resident form of the sweep (pqa_kernels.h: ServerMailbox).  Control flow around the steps is wave-uniform by
// construction: whole waves poll (one address -> one transaction) and whole waves store, every wait is bounded, and the
// kernel's lifetime is bounded by idleTicks whatever the host does.
// ------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint64_t uniform64(uint64_t x) {
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)x), hi = __builtin_amdgcn_readfirstlane((uint32_t)(x >> 32));
  return ((uint64_t)hi << 32) | lo;
}
// A request is one 64-byte line -- 16 dwords, the same layout in the host's mailbox, in the device's hand-off block and in
// LDS: {req / go, prior, asked, out, flag, flagValue, outBase, stop}.  Lane l of a wave moves dword l & 15, so that a poll, a
// hand-off or a fetch of all the fields is ONE memory transaction (six dependent reads over PCIe cost 10 us; one costs 1.7).
// The writers store the sequence number last (host: program order; workgroup 0: one store instruction for the line), so a
// line that shows a new sequence number shows that request's fields.
static_assert(offsetof(ServerMailbox, stop) == 56 && offsetof(ServerMailbox, state) == 64 && sizeof(ServerCtl) == 64, "request line");
__device__ __forceinline__ uint64_t line_u64(uint32_t v, int i) {   // qword i of the line a wave holds
  return (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)v, 2 * i) |
         ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)v, 2 * i + 1) << 32);
}

template <int WPQ, int NP, bool DEFER, int KC>
// Three workgroups per CU with the 154 registers the kernel wants (3 x 160 of the 512 per SIMD lane).  Capped at 128 (waves_per_eu
// (4, 4), until round 3) it spilled 17 registers and a step took 19.0 us instead of 17.0.  The 32 registers left per lane are what the
// 256-thread posterior kernels that must run beside the resident sweep fit into (prior_kernels.hip: kSmallThreads; 24 - 30 each).
// (KC = 5, round 5: 162 registers, 168 allocated.  A quiz through the resident sweep -- NextQuestion and RecordAnswer in turn, 1000 x 5 x
//  1000 -- took 73.4 us per pair before and 73.5 after on one box: measured, because the margin above is gone.)
__global__ __launch_bounds__(WPQ * 64) __attribute__((amdgpu_waves_per_eu(3, 3)))
void eval_server_f64(EvalArgs a, ServerMailbox *mb, uint32_t *requestLine, int everyonePolls, ServerCtl *ctl, uint64_t lastSeq,
                     uint64_t idleTicks, unsigned stepOffsetBytes) {
  extern __shared__ double smem[];
  uint32_t *step = reinterpret_cast<uint32_t *>(reinterpret_cast<char *>(smem) + stepOffsetBytes);   // 16 dwords
  const int wave = (int)__builtin_amdgcn_readfirstlane(threadIdx.x / kWave);
  const unsigned word = threadIdx.x & 15;
  const bool first = blockIdx.x == 0;
  // requestLine: where the host writes requests -- the first line of the mailbox, or (everyonePolls) a line of host-visible
  // DEVICE memory: then a poll is a local read instead of one over PCIe, cheap enough for every workgroup to watch the
  // line itself instead of waiting for workgroup 0 to hand the request on (-2 us per step)
  uint32_t *mbWord = requestLine + word, *ctlWord = reinterpret_cast<uint32_t *>(ctl) + word;
  uint64_t last = lastSeq;
  bool copyTable = true;
  for (;;) {
    if (wave == 0) {
      uint32_t v = 0;
      uint64_t go = last;
      const uint64_t t0 = wall_clock64();
      if (first) {
        for (;;) {
          v = __hip_atomic_load(mbWord, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          go = line_u64(v, 0);
          if (go != last) break;
          const bool stop = line_u64(v, 7) != 0;
          if (stop || wall_clock64() - t0 > idleTicks) {
            // leaving: say so, then look once more -- the host posts first and reads `state` second (tools/server_rt.hip)
            __hip_atomic_store(&mb->state, kServerExiting, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "");
            v = __hip_atomic_load(mbWord, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            go = line_u64(v, 0);
            if (go != last && !stop) {
              __hip_atomic_store(&mb->state, kServerRunning, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            } else {
              go = ~0ull;
              v = 0xFFFFFFFFu;     // (every dword of the hand-off line, the sequence number included)
            }
            break;
          }
        }
        if (go != ~0ull) __hip_atomic_store(&mb->taken, go, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if (!everyonePolls || go == ~0ull)
          __hip_atomic_store(ctlWord, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // the whole line: one store instruction
      } else if (everyonePolls) {
        for (unsigned it = 0;; it++) {
          v = __hip_atomic_load(mbWord, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          go = line_u64(v, 0);
          if (go != last && go != 0) break;
          if ((it & 15) == 15) {   // workgroup 0 says "leave" through the hand-off line
            const uint32_t v2 = __hip_atomic_load(ctlWord, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (line_u64(v2, 0) == ~0ull) { v = v2; break; }
            if (wall_clock64() - t0 > 8 * idleTicks + 100000000ull) { v = 0xFFFFFFFFu; break; }   // workgroup 0 is gone
          }
          __builtin_amdgcn_s_sleep(1);
        }
      } else {
        for (;;) {
          v = __hip_atomic_load(ctlWord, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          go = line_u64(v, 0);
          if (go != last && go != 0) break;
          if (wall_clock64() - t0 > 8 * idleTicks + 100000000ull) { v = 0xFFFFFFFFu; break; }   // workgroup 0 is gone
          __builtin_amdgcn_s_sleep(1);
        }
      }
      step[word] = v;   // lanes l, l+16, .. store the same dword
    }
    __syncthreads();
    const uint64_t go = uniform64(reinterpret_cast<const uint64_t *>(step)[0]);
    if (go == ~0ull) break;
    EvalArgs b = a;
    b.prior = reinterpret_cast<const double *>(uniform64(reinterpret_cast<const uint64_t *>(step)[1]));
    b.asked = reinterpret_cast<const uint32_t *>(uniform64(reinterpret_cast<const uint64_t *>(step)[2]));
    b.fs.out = reinterpret_cast<SelectResult *>(uniform64(reinterpret_cast<const uint64_t *>(step)[3]));
    b.fs.seq = reinterpret_cast<uint64_t *>(uniform64(reinterpret_cast<const uint64_t *>(step)[4]));
    b.fs.flagValue = uniform64(reinterpret_cast<const uint64_t *>(step)[5]);
    const uint64_t ob = uniform64(reinterpret_cast<const uint64_t *>(step)[6]);
    b.fs.outBase = (int64_t)(ob & ~(kServerHandOver | kServerNoWatch));
    b.serverNoWatch = (ob & kServerNoWatch) != 0;
    b.fs.sampleSubtasks = (ob & kServerHandOver) ? 1 : 0;   // the priority vector goes to the host (FusedSelect::hostPriority), no argmax
    b.fs.seqValue = go;
    // (as with the thread index: nothing derived from the launch constants may be hoisted out of the step loop)
    asm volatile("" : "+s"(b.cube), "+s"(b.tgap), "+s"(b.qgap), "+s"(b.priority), "+s"(b.fs.scratch), "+s"(b.K), "+s"(b.ldT),
                 "+s"(b.qFirst), "+s"(b.qLimit));
    __syncthreads();   // the step block may be rewritten only after everybody has read it
    const uint64_t tA = wall_clock64();   // 100 MHz
    sweep_body<WPQ, NP, false, true, DEFER, false, true, KC>(b, copyTable);
    copyTable = false;
    last = go;
    if (first && wave == 0) {
      // device-side duration of the step -- request in hand to answer published (workgroup 0's epilogue wave is the finisher, so
      // its sweep ends with the answer) -- then the request it belongs to: two posted stores to the host's mailbox, never read
      // here.  What bench.py reports as the dominant kernel's time on the resident path (ServerMailbox::pad).
      __hip_atomic_store(&mb->pad[0], wall_clock64() - tA, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      __hip_atomic_store(&mb->pad[1], go, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
      __hip_atomic_store(&mb->done, go, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    __syncthreads();   // closes every divergent region of the step before the next one's wait
  }
  if (first && wave == 0) __hip_atomic_store(&mb->state, kServerExited, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// ------------------------------------------------------------------------------------------------------------------
// Streaming fallback for rows too long to keep on chip (ldT > 16384): one 256-thread workgroup per question, pass 2
// re-reads the sA row (served by L2 / Infinity Cache when it can).  Same arithmetic per element.
// ------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void eval_questions_f64_stream(EvalArgs a) {
  select_quiz(a);
  constexpr int WPQ = 4, kThreads = 256;
  extern __shared__ double smem[];
  const int64_t ldT = a.ldT, K = a.K;
  double *tbl = smem;
  if (!lds_table_at_zero(tbl)) __builtin_trap();  // log2hot addresses the table absolutely
  double *redW = tbl + kLog2TableDoubles;
  double *wkAll = redW + 2 * WPQ;
  double *partAll = wkAll + 2 * K;
  const int nPart = (int)(K + 2);
  uint32_t *watchWords = reinterpret_cast<uint32_t *>(partAll + 2 * nPart * WPQ);   // by question parity: the rows that passed the pole watch [0], [1], its wider form [2], [3]
  const int tid = threadIdx.x, lane = tid % kWave, wave = tid / kWave;
  for (int i = tid; i < kLog2TableDoubles; i += kThreads) tbl[i] = gLog2Table[i];
  if (tid < 4) watchWords[tid] = 0;
  __syncthreads();
  const int64_t qStride = (K + 1) * ldT;
  const int64_t nPairs = ldT >> 1;
  int phase = 0, qpar = 0;
  const bool watch = a.poleList != nullptr;
  bool wgSuspect = false;                                     // (thread 0's: a question of this workgroup passed the watch)
  Best best{0.0, -1};
  for (int64_t q = a.qFirst + blockIdx.x; q < a.qLimit; q += gridDim.x) {
    if (bit_test(a.qgap, q) || bit_test(a.asked, q)) {
      if (tid == 0) store_priority(a.priority + (q - a.qFirst), 0.0);
      continue;
    }
    const double *qBase = a.cube + q * qStride;
    const double2 *rowD = reinterpret_cast<const double2 *>(qBase + K * ldT);
    const double2 *prior2 = reinterpret_cast<const double2 *>(a.prior);
    double *wk = wkAll + qpar * K;
    double *part = partAll + qpar * (nPart * WPQ);
    double accL = 0, hW = 0;
    uint32_t poleRows = 0, quarterRows = 0;
    for (int64_t k = 0; k < K; k++) {
      const double2 *rowA = reinterpret_cast<const double2 *>(qBase + k * ldT);
      double s0 = 0, c0 = 0, s1 = 0, c1 = 0;
      for (int64_t p = tid; p < nPairs; p += kThreads) {
        const bool g0 = bit_test(a.tgap, 2 * p), g1 = bit_test(a.tgap, 2 * p + 1);
        const double2 d = rowD[p], av = rowA[p], pv = prior2[p];
        const double x0 = g0 ? 0.0 : (av.x * div_nr(1.0, d.x)) * pv.x;
        const double x1 = g1 ? 0.0 : (av.y * div_nr(1.0, d.y)) * pv.y;
        {
          const double y = x0 - c0;
          const double t = s0 + y;
          c0 = (t - s0) - y;
          s0 = t;
        }
        {
          const double y = x1 - c1;
          const double t = s1 + y;
          c1 = (t - s1) - y;
          s1 = t;
        }
      }
      const double sLane = (s0 - c0) + (s1 - c1);
      double Wk = wave_sum(sLane);
      {
        double *buf = redW + phase * WPQ;
        if (lane == 0) buf[wave] = Wk;
        __syncthreads();
        Wk = row_sum<WPQ>(buf[lane % WPQ]);
        phase ^= 1;
      }
      // the pole watch (see sweep_body): a lane that holds a quarter / nearly all of W_k
      if (watch && __any(sLane > Wk * kQuarterShare)) {
        quarterRows |= 1u << (k < 31 ? (int)k : 31);
        if (__any(sLane >= Wk * kNearOneShare)) poleRows |= 1u << (k < 31 ? (int)k : 31);
      }
      const double invWk = div_nr(1.0, Wk);
      double v = 0;
      for (int64_t p = tid; p < nPairs; p += kThreads) {
        const bool g0 = bit_test(a.tgap, 2 * p), g1 = bit_test(a.tgap, 2 * p + 1);
        const double2 d = rowD[p], av = rowA[p];
        double2 pv = prior2[p], id, lh;
        id.x = g0 ? 0.0 : div_nr(1.0, d.x);
        id.y = g1 ? 0.0 : div_nr(1.0, d.y);
        pv.x = g0 ? 0.0 : pv.x;
        pv.y = g1 ? 0.0 : pv.y;
        lh.x = (av.x * id.x) * pv.x;
        lh.y = (av.y * id.y) * pv.y;
        pass2_pair(lh, id, pv, invWk, tbl, hW, v, accL);
      }
      v = wave_sum(v);
      if (lane == 0) {
        if (wave == 0) wk[k] = Wk;
        part[k * WPQ + wave] = v;
      }
    }
    hW = wave_sum(hW);
    accL = wave_sum(accL);
    if (lane == 0) {
      part[K * WPQ + wave] = hW;
      part[(K + 1) * WPQ + wave] = accL;
      if (quarterRows != 0) { atomicOr(&watchWords[2 + qpar], quarterRows); if (poleRows != 0) atomicOr(&watchWords[qpar], poleRows); }
    }
    __syncthreads();
    if (tid == 0) {
      for (int r = 0; r < nPart; r++) {
        double acc = part[r * WPQ];
        for (int w2 = 1; w2 < WPQ; w2++) acc += part[r * WPQ + w2];
        if (r < K && acc <= kSmallV && ((watchWords[2 + qpar] >> (r < 31 ? r : 31)) & 1u)) watchWords[qpar] |= 1u << (r < 31 ? r : 31);   // (a vanishing velocity sum: pole_device.h)
        part[r] = r < K ? wk[r] * sqrt(acc) : acc;
      }
      const double pri = eval_epilogue(wk, -part[K], part, K, part[K + 1], a.vCompTail);
      store_priority(a.priority + (q - a.qFirst), pri);
      best_offer(best, pri, q - a.qFirst);
      watchWords[qpar ^ 1] = 0;                                // (set again only behind the next question's barriers)
      watchWords[2 + (qpar ^ 1)] = 0;
      if (watch && watchWords[qpar] != 0) {
        wgSuspect = true;
        // the question's sums as they are, for the fix behind the sweep (pole_kernels.hip)
        const uint32_t at = pole_list_append(a.poleList, (uint32_t)(q - a.qFirst), K <= 31 ? watchWords[qpar] : 0u, a.slots != nullptr ? blockIdx.y : 0u);
        double *ps = a.poleScratch + (a.slots != nullptr ? (int64_t)at : q - a.qFirst) * (2 * K + 2);
        for (int r = 0; r < K; r++) { ps[r] = wk[r]; ps[K + r] = part[r]; }
        ps[2 * K] = part[K];
        ps[2 * K + 1] = part[K + 1];
      }
    }
    qpar ^= 1;
  }
  if (wave == 0) fused_select(a, best, lane, nullptr, wgSuspect);   // lane 0 carries the workgroup's best, the other lanes none
}

// ------------------------------------------------------------------------------------------------------------------
// Float engines, batched argmax: the fp32 sweep's best questions of every quiz RE-RANKED IN FP64 (BASELINE configs[4]; north_star:
// "bit-exact for the argmax index").  fp32's 2^-23 reaches a priority amplified by the state's conditioning (DESIGN section 5:
// 1e-6 .. 3e-3 relative), so the fp32 argmax may name a question whose priority is 1e-4 below the fp64 oracle's pick.  The fp32
// sweep therefore only NOMINATES: batch_topk_kernel takes each quiz's kRerank best questions out of the sweep's priority matrix,
// batch_rerank_kernel evaluates exactly those -- one workgroup per (candidate, quiz), the reference's formula in fp64 on the
// engine's (rounded) cube rows, the streaming sweep's element arithmetic -- and batch_repick_kernel returns the fp64 argmax of
// the candidates (maximum, lowest index on ties).  kRerank x (K + 1) rows per quiz: microseconds beside the sweep.
// ------------------------------------------------------------------------------------------------------------------
constexpr int kRerank = 8;

// one workgroup per quiz: kRerank rounds of "the best available question after the previous winner" in the order (priority
// descending, index ascending); priorityT is [Q][Bp], quiz-minor
__global__ __launch_bounds__(256) void batch_topk_kernel(const double *__restrict__ priorityT, int64_t Q, int Bp, const uint32_t *__restrict__ qgap,
                                                         const QuizSlot *__restrict__ slots, int64_t *__restrict__ cand) {
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid % kWave, wave = tid / kWave;
  const uint32_t *asked = slots[b].asked;
  __shared__ double sp[4];
  __shared__ long long si[4];
  double lastP = __builtin_huge_val();
  long long lastI = -1;
  for (int r = 0; r < kRerank; r++) {
    double bp = 0;
    long long bi = -1;
    for (int64_t q = tid; q < Q; q += 256) {
      if (bit_test(qgap, q) || bit_test(asked, q)) continue;
      double p = priorityT[(size_t)q * (size_t)Bp + (size_t)b];
      if (p != p) p = -__builtin_huge_val();                                      // NaN never wins
      if (!(p < lastP || (p == lastP && q > lastI))) continue;                    // taken in an earlier round
      if (bi < 0 || p > bp) { bp = p; bi = q; }                                   // (q ascending: the first of equal values stays)
    }
    for (int m = kWave / 2; m >= 1; m >>= 1) {
      const double op = __shfl_xor(bp, m, kWave);
      const long long oi = __shfl_xor(bi, m, kWave);
      if (oi >= 0 && (bi < 0 || op > bp || (op == bp && oi < bi))) { bp = op; bi = oi; }
    }
    if (lane == 0) { sp[wave] = bp; si[wave] = bi; }
    __syncthreads();
    bp = sp[0]; bi = si[0];
    for (int w = 1; w < 4; w++)
      if (si[w] >= 0 && (bi < 0 || sp[w] > bp || (sp[w] == bp && si[w] < bi))) { bp = sp[w]; bi = si[w]; }
    __syncthreads();
    if (tid == 0) cand[(size_t)b * kRerank + r] = bi;
    if (bi < 0) { for (int r2 = r + 1; r2 < kRerank && tid == 0; r2++) cand[(size_t)b * kRerank + r2] = -1; break; }
    lastP = bp; lastI = bi;
  }
}

// blockIdx.x = candidate, blockIdx.y = quiz.  E: the cube's element type (float for Float engines); every element is converted
// to fp64 (exact) and the question is evaluated as eval_questions_f64_stream evaluates it.
template <typename E>
__global__ __launch_bounds__(256) void batch_rerank_kernel(const E *__restrict__ cube, const uint32_t *__restrict__ tgap,
                                                           const QuizSlot *__restrict__ slots, const int64_t *__restrict__ cand,
                                                           double *__restrict__ candPri, int64_t K, int64_t ldT, double vCompTail) {
  constexpr int WPQ = 4, kThreads = 256;
  extern __shared__ double smem[];
  double *tbl = smem;
  if (!lds_table_at_zero(tbl)) __builtin_trap();  // log2hot addresses the table absolutely
  double *redW = tbl + kLog2TableDoubles;
  double *wk = redW + 2 * WPQ;
  double *part = wk + K;
  const int nPart = (int)(K + 2);
  const int tid = threadIdx.x, lane = tid % kWave, wave = tid / kWave;
  const int b = blockIdx.y;
  const int64_t q = cand[(size_t)b * kRerank + blockIdx.x];
  if (q < 0) {
    if (tid == 0) candPri[(size_t)b * kRerank + blockIdx.x] = 0.0;
    return;
  }
  for (int i = tid; i < kLog2TableDoubles; i += kThreads) tbl[i] = gLog2Table[i];
  __syncthreads();
  const double *prior = slots[b].prior;
  const E *qBase = cube + q * (K + 1) * ldT;
  const E *rowD = qBase + K * ldT;
  const int64_t nPairs = ldT >> 1;
  int phase = 0;
  double accL = 0, hW = 0;
  for (int64_t k = 0; k < K; k++) {
    const E *rowA = qBase + k * ldT;
    double s0 = 0, c0 = 0, s1 = 0, c1 = 0;
    for (int64_t p = tid; p < nPairs; p += kThreads) {
      const bool g0 = bit_test(tgap, 2 * p), g1 = bit_test(tgap, 2 * p + 1);
      const double x0 = g0 ? 0.0 : ((double)rowA[2 * p] * div_nr(1.0, (double)rowD[2 * p])) * prior[2 * p];
      const double x1 = g1 ? 0.0 : ((double)rowA[2 * p + 1] * div_nr(1.0, (double)rowD[2 * p + 1])) * prior[2 * p + 1];
      { const double y = x0 - c0; const double t = s0 + y; c0 = (t - s0) - y; s0 = t; }
      { const double y = x1 - c1; const double t = s1 + y; c1 = (t - s1) - y; s1 = t; }
    }
    double Wk = wave_sum((s0 - c0) + (s1 - c1));
    {
      double *buf = redW + phase * WPQ;
      if (lane == 0) buf[wave] = Wk;
      __syncthreads();
      Wk = row_sum<WPQ>(buf[lane % WPQ]);
      phase ^= 1;
    }
    const double invWk = div_nr(1.0, Wk);
    double v = 0;
    for (int64_t p = tid; p < nPairs; p += kThreads) {
      const bool g0 = bit_test(tgap, 2 * p), g1 = bit_test(tgap, 2 * p + 1);
      double2 pv, id, lh;
      id.x = g0 ? 0.0 : div_nr(1.0, (double)rowD[2 * p]);
      id.y = g1 ? 0.0 : div_nr(1.0, (double)rowD[2 * p + 1]);
      pv.x = g0 ? 0.0 : prior[2 * p];
      pv.y = g1 ? 0.0 : prior[2 * p + 1];
      lh.x = ((double)rowA[2 * p] * id.x) * pv.x;
      lh.y = ((double)rowA[2 * p + 1] * id.y) * pv.y;
      pass2_pair(lh, id, pv, invWk, tbl, hW, v, accL);
    }
    v = wave_sum(v);
    if (lane == 0) {
      if (wave == 0) wk[k] = Wk;
      part[k * WPQ + wave] = v;
    }
  }
  hW = wave_sum(hW);
  accL = wave_sum(accL);
  if (lane == 0) {
    part[K * WPQ + wave] = hW;
    part[(K + 1) * WPQ + wave] = accL;
  }
  __syncthreads();
  if (tid == 0) {
    for (int r = 0; r < nPart; r++) {
      double acc = part[r * WPQ];
      for (int w2 = 1; w2 < WPQ; w2++) acc += part[r * WPQ + w2];
      part[r] = r < K ? wk[r] * sqrt(acc) : acc;
    }
    candPri[(size_t)b * kRerank + blockIdx.x] = eval_epilogue(wk, -part[K], part, K, part[K + 1], vCompTail);
  }
}

// every quiz's winner among its re-ranked candidates; the result and then the flag go to host-coherent memory
__global__ __launch_bounds__(256) void batch_repick_kernel(const int64_t *__restrict__ cand, const double *__restrict__ candPri,
                                                           const QuizSlot *__restrict__ slots, int nSlots, int64_t outBase, uint64_t flagValue) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= nSlots) return;
  double bp = 0.0;
  int64_t bq = -1;
  for (int r = 0; r < kRerank; r++) {
    const int64_t q = cand[(size_t)b * kRerank + r];
    if (q < 0) continue;
    double p = candPri[(size_t)b * kRerank + r];
    if (p != p) p = -__builtin_huge_val();
    if (bq < 0 || p > bp || (p == bp && q < bq)) { bp = p; bq = q; }
  }
  const QuizSlot s = slots[b];
  s.out->priority = bq < 0 ? 0.0 : bp;
  s.out->index = bq < 0 ? -1 : bq + outBase;
  if (s.seq != nullptr) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");             // system scope: the record before the flag
    __hip_atomic_store(s.seq, flagValue, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

size_t BatchRerankScratchBytes() { return (size_t)256 * kRerank * (sizeof(int64_t) + sizeof(double)); }

hipError_t LaunchBatchRerank(const KbView &kb, const QuizSlot *slots, int nSlots, int Bp, const double *priorityT, void *scratch,
                             int64_t outBase, uint64_t flagValue, hipStream_t stream) {
  if (kb.elem != 4 || nSlots <= 0 || nSlots > 256 || scratch == nullptr || priorityT == nullptr) return hipErrorInvalidValue;
  int64_t *cand = static_cast<int64_t *>(scratch);
  double *candPri = reinterpret_cast<double *>(cand + 256 * kRerank);
  const double nT = (double)(kb.nValidTargets + 1);            // PqaCore/CEEvalQsSubtaskConsider.cpp:191
  const double vCompTail = 0.34657359027997265470861606072909 / (nT * nT);
  hipLaunchKernelGGL(batch_topk_kernel, dim3((unsigned)nSlots), dim3(256), 0, stream, priorityT, kb.Q, Bp, kb.qgap, slots, cand);
  const size_t shmem = (size_t)(kLog2TableDoubles + 2 * 4 + kb.K + (kb.K + 2) * 4) * sizeof(double);
  hipLaunchKernelGGL(batch_rerank_kernel<float>, dim3(kRerank, (unsigned)nSlots), dim3(256), shmem, stream, static_cast<const float *>(kb.cube),
                     kb.tgap, slots, cand, candPri, kb.K, kb.ldT, vCompTail);
  hipLaunchKernelGGL(batch_repick_kernel, dim3((unsigned)((nSlots + 255) / 256)), dim3(256), 0, stream, cand, candPri, slots, nSlots, outBase, flagValue);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------------------------
// log2hot() on an array, for the tests of the device function against the reference's SRVectMathTest.Log2Hot criteria
// and against the oracle's operation-for-operation restatement (the sweep only ever shows it through priorities).
// ------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void log2hot_array_kernel(const double *x, double *out, int64_t n) {
  extern __shared__ double smem[];
  double *tbl = smem;
  if (!lds_table_at_zero(tbl)) __builtin_trap();
  for (int i = threadIdx.x; i < kLog2TableDoubles; i += 256) tbl[i] = gLog2Table[i];
  __syncthreads();
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) out[i] = log2hot(x[i], tbl);
}

hipError_t LaunchLog2HotArray(const double *x, double *out, int64_t n, hipStream_t stream) {
  if (n <= 0) return hipSuccess;
  const int64_t blocks = (n + 255) / 256;
  hipLaunchKernelGGL(log2hot_array_kernel, dim3((unsigned)(blocks < 2048 ? blocks : 2048)), dim3(256),
                     kLog2TableDoubles * sizeof(double), stream, x, out, n);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------------------------
// launch
// ------------------------------------------------------------------------------------------------------------------
namespace {

struct Variant {
  int id, wpq, np;
  bool prLds;
  const char *name;
};
// capacity in targets = 128 * wpq * np
const Variant kVariants[] = {
    {1, 1, 8, false, "wave_per_question_np8"}, {2, 4, 2, false, "wg256_np2"},   {3, 4, 4, false, "wg256_np4"},
    {4, 4, 8, false, "wg256_np8"},             {5, 8, 8, true, "wg512_np8_prlds"}, {6, 16, 5, true, "wg1024_np5_prlds"},
    {7, 16, 8, true, "wg1024_np8_prlds"},      {8, 2, 4, false, "wg128_np4"},   {9, 8, 5, false, "wg512_np5"},
    {10, 8, 10, true, "wg512_np10_prlds"},     {11, 16, 5, false, "wg1024_np5"}, {12, 8, 10, false, "wg512_np10"},
    {13, 4, 3, false, "wg256_np3"},            {14, 4, 5, false, "wg256_np5"},  {15, 4, 6, false, "wg256_np6"},
    {16, 8, 6, false, "wg512_np6"},            {17, 8, 7, false, "wg512_np7"},  {18, 8, 8, false, "wg512_np8"},
    {19, 8, 9, false, "wg512_np9"},            {20, 4, 10, true, "wg256_np10_prlds"}, {21, 4, 10, false, "wg256_np10"},
    {22, 4, 6, true, "wg256_np6_prlds"},       {23, 4, 8, true, "wg256_np8_prlds"}, {24, 4, 9, true, "wg256_np9_prlds"},
    {99, 4, 0, false, "stream256"},
};

// Default shape: the smallest capacity that holds a row (idle lanes are pure loss: at 8000 targets the np10 shape wastes a
// fifth of its lanes), 256-thread workgroups up to 4096 targets (two workgroups per CU), 512 threads beyond.
int pick_variant(int64_t ldT, int variant) {
  if (variant != 0) return variant;
  if (ldT <= 1024) return 2;
  if (ldT <= 1536) return 13;
  if (ldT <= 2048) return 3;
  if (ldT <= 2560) return 14;
  if (ldT <= 3072) return 15;
  if (ldT <= 4096) return 4;
  // 4097..5120 targets: four waves of 9 / 10 pairs per lane with the priors in LDS, not eight waves of 5 -- half as many waves
  // at the row barrier and half as many wave reductions per element (194 vs 218 us at 4500 targets, 234 vs 250 at 5000)
  if (ldT <= 4608) return 24;
  if (ldT <= 5120) return 20;
  if (ldT <= 6144) return 16;
  if (ldT <= 7168) return 17;
  if (ldT <= 8192) return 18;
  if (ldT <= 9216) return 19;
  if (ldT <= 10240) return 10;   // priors in LDS: no spills at 10 pairs per lane (950 vs 1028 us at 10000 x 5 x 10000)
  if (ldT <= 16384) return 7;
  return 99;
}


constexpr size_t kLdsPerCU = 160 * 1024;   // gfx950
template <int WPQ, int NP, bool PRLDS>
constexpr size_t eval_base_lds_bytes(int64_t K, int64_t ldT) {
  return PRLDS ? eval_lds_doubles(WPQ, K, true, ldT) * sizeof(double)
               : eval_md_row_offset_bytes(WPQ, K) + (size_t)NP * WPQ * kWave * 16;
}

// The fix behind a watching single-quiz sweep (pole_kernels.hip): the questions in the launch's suspect list, their sums in
// poleScratch as the sweep left them -- W_k [K] | W_k sqrt(V_k) [K] | sum l log2 p | lack sum.
// Where only the argmax leaves the engine -- a fused argmax of ONE quiz -- the fix is gated (PoleFix::gate); the sweep of such a launch tracks
// the gaps the gate needs (EvalArgs::poleGate is set from this by the launch wrappers: sweep and fix always agree).
static bool eval_gates(const EvalArgs &args) {
  // (rows of up to 1024 targets: a question's fix is latency there -- ~20 us whatever the number of questions redone -- and the kernel
  //  that bounds them one launch more: 71.3 vs 71.9 us per late selection at 1000 x 5 x 1000, 60.8 vs 55.0 half-way; from 1500 targets on
  //  the gate pays: 86 vs 95, 116 vs 123 (2000), 181 vs 262 (3000), 245 vs 377 (4000), 988 vs 1771 us (10000) -- tools/gate_probe.py)
  return args.poleGate && args.poleList != nullptr && args.slots == nullptr && args.fs.scratch != nullptr && args.fs.sampleSubtasks == 0 &&
         args.fs.hostPriority == nullptr && !args.poleNoFollow && args.ldT > 1024;
}
hipError_t launch_pole_fixup(const EvalArgs &args, hipStream_t stream, int nBatch = 1) {
  PoleFix f{};
  f.cube = args.cube; f.tgap = args.tgap; f.qgap = args.qgap; f.asked = args.asked; f.prior = args.prior;
  f.list = args.poleList; f.sums = args.poleScratch; f.sumsStride = 2 * args.K + 2;
  f.wOff = 0; f.vOff = (int)args.K; f.hOff = (int)(2 * args.K); f.lOff = (int)(2 * args.K + 1); f.secondIsWV = 1;
  f.priority = args.priority;
  f.K = args.K; f.T = args.T; f.ldT = args.ldT; f.qFirst = args.qFirst; f.nQ = args.qLimit - args.qFirst;
  f.capacity = f.nQ;
  f.vCompTail = args.vCompTail;
  f.fs = args.fs;
  if (args.fs.scratch != nullptr && args.fs.sampleSubtasks > 0 && args.fs.hostPriority != nullptr) { f.hostPriority = args.fs.hostPriority; f.hostTag = args.fs.seqValue; }
  f.gate = eval_gates(args) ? 1 : 0;
  if (args.slots != nullptr) {   // a grid.y = quiz launch: the records by entry, every quiz's own vectors, and every quiz's publication
    f.slots = args.slots; f.nSlots = nBatch; f.bySlot = 1; f.prior = nullptr; f.asked = nullptr; f.priority = nullptr;
    f.capacity = f.nQ * (int64_t)nBatch;
    f.hostTag = args.fs.seqValue;
  }
  return LaunchPoleFixup(f, stream);
}

template <int WPQ, int NP, bool PRLDS, bool DEFER, bool POLE = false, int KC = 0>
hipError_t launch_reg_form(const EvalArgs &args, int64_t nQ, int nBatch, hipStream_t stream) {
  // the variant with the pole watch where the engine asked for it (single-quiz launches)
  if constexpr (!POLE) {
    if (args.poleList != nullptr) return launch_reg_form<WPQ, NP, PRLDS, DEFER, true, KC>(args, nQ, nBatch, stream);
  }
  // five answers as a constant (eval_questions_f64's comment): the shapes with the priors in registers, and the 10000-target shape
  // (the other LDS-prior shapes measured within 1 %: not built)
  // (four pairs per lane: 2000 targets 39.3 -> 41.5 us with it -- the shape sits at its register cap, eval_questions_f64_occ3 -- left out)
  if constexpr (KC == 0 && DEFER && NP != 4 && (!PRLDS || (WPQ == 8 && NP == 10))) {
    if (args.K == 5) return launch_reg_form<WPQ, NP, PRLDS, DEFER, POLE, 5>(args, nQ, nBatch, stream);
  }
  const size_t shmem = eval_base_lds_bytes<WPQ, NP, PRLDS>(args.K, args.ldT) + (DEFER ? eval_deferred_bytes(WPQ, NP, args.K, PRLDS) : 0);
  auto kern = [] {
    if constexpr (NP == 4 && WPQ == 4 && (DEFER || POLE)) return eval_questions_f64_occ3<WPQ, NP, PRLDS, DEFER, POLE>;   // (5 and 6 pairs: slower with the spills)
    else return eval_questions_f64<WPQ, NP, PRLDS, DEFER, POLE, KC>;
  }();
  // attribute and occupancy are properties of (kernel, LDS size, device): asked once per device, not on every launch
  static LaunchCache cache;
  const int dev = LaunchCache::Device();
  int cachedPerCU = 0;
  if (!cache.Get(dev, shmem, &cachedPerCU)) {
    if (shmem > 64 * 1024) {
      const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
      if (e != hipSuccess) return e;
    }
    int perCU = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&perCU, kern, WPQ * 64, shmem) != hipSuccess || perCU < 1) perCU = 1;
    // measured at 1000 x 5 x 1000 (wg256_np2): three resident workgroups per CU striding over the questions run the sweep in
    // 14.9 us, four (every question its own workgroup) in 15.7, two in 17.5 -- do not rely on the register count to say 3
    if (NP <= 2 && perCU > 3) perCU = 3;
    cachedPerCU = perCU;
    cache.Put(dev, shmem, perCU);
  }
  const int gNumCUs = cache.NumCUs(dev);
  // one question per workgroup while they all fit on the chip at once; otherwise a resident grid that strides
  const int64_t resident = (int64_t)gNumCUs * cachedPerCU;
  int64_t resGrid = nQ < resident ? nQ : resident;
  const int64_t maxRecords = args.slots != nullptr ? args.fs.scratchStride : kFusedMaxGrid;
  if (args.fs.scratch != nullptr && resGrid > maxRecords) resGrid = maxRecords;  // one winner record per workgroup
  if (args.maxGrid > 0 && resGrid > args.maxGrid) resGrid = args.maxGrid;   // (test hook: KbView::maxGrid)
  hipLaunchKernelGGL(kern, dim3((unsigned)resGrid, (unsigned)nBatch), dim3(WPQ * 64), shmem, stream, args);
  const hipError_t le = hipGetLastError();
  if constexpr (POLE) { if (le == hipSuccess && !args.poleNoFollow && !args.fs.lazyFix) return launch_pole_fixup(args, stream, nBatch); }
  return le;
}

template <int WPQ, int NP, bool PRLDS>
hipError_t launch_reg(const EvalArgs &args, int64_t nQ, int nBatch, hipStream_t stream) {
  if constexpr (eval_defers_sums(WPQ, NP, PRLDS)) {
    if (eval_base_lds_bytes<WPQ, NP, PRLDS>(args.K, args.ldT) + eval_deferred_bytes(WPQ, NP, args.K, PRLDS) <= kLdsPerCU)
      return launch_reg_form<WPQ, NP, PRLDS, true>(args, nQ, nBatch, stream);
  }
  return launch_reg_form<WPQ, NP, PRLDS, false>(args, nQ, nBatch, stream);
}

hipError_t launch_variant(const EvalArgs &args, int64_t ldT, int variant, int nBatch, hipStream_t stream) {
  const int64_t nQ = args.qLimit - args.qFirst;
  int v = pick_variant(ldT, variant);
  // Many quizzes per launch over short rows: there are workgroups enough to fill the chip whatever their size, so the shape
  // with fewer instructions per question wins -- two waves of four pairs per lane pay half as many wave reductions per
  // element as four waves of two (93.0 k vs 84.8 k selections/s at 1000 x 5 x 1000 from 8 quizzes per launch up; for a
  // single quiz it is the other way round, 15.1 vs 14.75 us).
  if (variant == 0 && v == 2 && nBatch >= 8) v = 8;
  int wpq = 0, np = 0;
  for (const Variant &x : kVariants)
    if (x.id == v) { wpq = x.wpq; np = x.np; }
  if (v != 99 && (wpq == 0 || ldT > (int64_t)128 * wpq * np)) return hipErrorInvalidValue;
  switch (v) {
    case 1: return launch_reg<1, 8, false>(args, nQ, nBatch, stream);
    case 2: return launch_reg<4, 2, false>(args, nQ, nBatch, stream);
    case 3: return launch_reg<4, 4, false>(args, nQ, nBatch, stream);
    case 4: return launch_reg<4, 8, false>(args, nQ, nBatch, stream);
    case 5: return launch_reg<8, 8, true>(args, nQ, nBatch, stream);
    case 6: return launch_reg<16, 5, true>(args, nQ, nBatch, stream);
    case 7: return launch_reg<16, 8, true>(args, nQ, nBatch, stream);
    case 8: return launch_reg<2, 4, false>(args, nQ, nBatch, stream);
    case 9: return launch_reg<8, 5, false>(args, nQ, nBatch, stream);
    case 10: return launch_reg<8, 10, true>(args, nQ, nBatch, stream);
    case 11: return launch_reg<16, 5, false>(args, nQ, nBatch, stream);
    case 12: return launch_reg<8, 10, false>(args, nQ, nBatch, stream);
    case 13: return launch_reg<4, 3, false>(args, nQ, nBatch, stream);
    case 14: return launch_reg<4, 5, false>(args, nQ, nBatch, stream);
    case 15: return launch_reg<4, 6, false>(args, nQ, nBatch, stream);
    case 16: return launch_reg<8, 6, false>(args, nQ, nBatch, stream);
    case 17: return launch_reg<8, 7, false>(args, nQ, nBatch, stream);
    case 18: return launch_reg<8, 8, false>(args, nQ, nBatch, stream);
    case 19: return launch_reg<8, 9, false>(args, nQ, nBatch, stream);
    case 20: return launch_reg<4, 10, true>(args, nQ, nBatch, stream);
    case 21: return launch_reg<4, 10, false>(args, nQ, nBatch, stream);
    case 22: return launch_reg<4, 6, true>(args, nQ, nBatch, stream);
    case 23: return launch_reg<4, 8, true>(args, nQ, nBatch, stream);
    case 24: return launch_reg<4, 9, true>(args, nQ, nBatch, stream);
    case 99: {
      const size_t shmem = eval_lds_doubles(4, args.K, false, 0) * sizeof(double);
      int64_t maxBlocks = 256 * 8;
      if (args.slots != nullptr && maxBlocks > args.fs.scratchStride) maxBlocks = args.fs.scratchStride;
      if (args.maxGrid > 0 && maxBlocks > args.maxGrid) maxBlocks = args.maxGrid;
      const unsigned grid = (unsigned)(nQ < maxBlocks ? nQ : maxBlocks);
      hipLaunchKernelGGL(eval_questions_f64_stream, dim3(grid, (unsigned)nBatch), dim3(256), shmem, stream, args);
      const hipError_t le = hipGetLastError();
      if (le == hipSuccess && args.poleList != nullptr && !args.fs.lazyFix) return launch_pole_fixup(args, stream, nBatch);
      return le;
    }
    default: return hipErrorInvalidValue;
  }
}

}  // namespace

const char *EvalVariantName(const KbView &kb, int variant) {
  const int v = pick_variant(kb.ldT, variant);
  for (const Variant &x : kVariants)
    if (x.id == v) return x.name;
  return "unknown";
}

static EvalArgs make_args(const KbView &kb, int64_t qFirst, int64_t qLimit) {
  EvalArgs args{};
  args.cube = static_cast<const double *>(kb.cube);   // Double engines only (Float: batch_kernels.hip)
  args.tgap = kb.tgap;
  args.qgap = kb.qgap;
  args.K = kb.K;
  args.T = kb.T;
  args.ldT = kb.ldT;
  args.poleScratch = kb.poleScratch;
  args.poleList = kb.poleScratch != nullptr ? kb.poleList : nullptr;
  args.qFirst = qFirst;
  args.qLimit = qLimit;
  const double nT = (double)(kb.nValidTargets + 1);  // PqaCore/CEEvalQsSubtaskConsider.cpp:191
  args.vCompTail = 0.34657359027997265470861606072909 / (nT * nT);
  args.fs = FusedSelect{nullptr, nullptr, nullptr, 0, 0, 0, 0, nullptr, 0, 0, nullptr, nullptr};
  args.slots = nullptr;
  args.maxGrid = kb.maxGrid;
  args.poleNoFollow = kb.poleNoFollow;
  args.poleGate = kb.poleGate;   // (the engine's option; the launch wrappers clear it where the launch is no fused single-quiz argmax: finish_args)
  return args;
}
// once a launch's selection fields are set: the sweep tracks gaps only where its fix will be gated
static void finish_args(EvalArgs &args) { args.poleGate = eval_gates(args) ? 1 : 0; }

hipError_t LaunchEvalQuestions(const KbView &kb, const double *prior, const uint32_t *asked, int64_t qFirst,
                               int64_t qLimit, double *priority, int variant, const FusedSelect *fused,
                               hipStream_t stream) {
  if (qLimit <= qFirst) return hipSuccess;
  EvalArgs args = make_args(kb, qFirst, qLimit);
  args.prior = prior;
  args.asked = asked;
  args.priority = priority;
  if (fused) args.fs = *fused;
  finish_args(args);
  return launch_variant(args, kb.ldT, variant, 1, stream);
}

hipError_t LaunchEvalPoleFixup(const KbView &kb, const double *prior, const uint32_t *asked, double *priority, const FusedSelect &fused,
                               hipStream_t stream) {
  EvalArgs args = make_args(kb, 0, kb.Q);
  if (args.poleList == nullptr) return hipErrorInvalidValue;
  args.prior = prior;
  args.asked = asked;
  args.priority = priority;
  args.fs = fused;
  finish_args(args);
  return launch_pole_fixup(args, stream, 1);
}

// ---- the posterior update of one answer + the sweep over the new posterior (eval_questions_f64_upd).  Whole-cube Double engines,
// the 2-pair shape (rows of up to 1024 targets), a fused selection (its finisher is what knows when the old prior may be replaced).
bool EvalFusesUpdate(const KbView &kb, int variant, int64_t nWorkers) {
  const int64_t nVects = (kb.T + 3) >> 2, quot = nVects / nWorkers, rem = nVects % nWorkers;
  const int64_t nSubtasks = quot == 0 ? rem : nWorkers;
  return kb.elem == 8 && pick_variant(kb.ldT, variant) == 2 && kb.ldT <= 1024 && nSubtasks <= 51 && kb.maxGrid == 0 &&
         eval_base_lds_bytes<4, 2, false>(kb.K, kb.ldT) + eval_deferred_bytes(4, 2, kb.K, false) <= kLdsPerCU;
}

hipError_t LaunchEvalQuestionsWithUpdate(const KbView &kb, double *prior, uint32_t *asked, double *priority, int variant, const FusedSelect &fused,
                                         int64_t iQuestion, int64_t iAnswer, int64_t nWorkers, RatedTargetDev *topOut, int64_t *topN,
                                         uint64_t *topFlag, uint64_t topFlagValue, int64_t topCount, hipStream_t stream) {
  if (!EvalFusesUpdate(kb, variant, nWorkers) || fused.scratch == nullptr || nWorkers < 1) return hipErrorInvalidValue;
  EvalArgs args = make_args(kb, 0, kb.Q);
  args.prior = prior;
  args.asked = asked;
  args.priority = priority;
  args.fs = fused;
  const double *cube = static_cast<const double *>(kb.cube);
  args.updRowA = cube + (iQuestion * (kb.K + 1) + iAnswer) * kb.ldT;   // CERecordAnswerSubtaskMul.cpp:25
  args.updRowD = cube + (iQuestion * (kb.K + 1) + kb.K) * kb.ldT;      // :26
  args.updQuestion = iQuestion;
  args.updVects = (kb.T + 3) >> 2;
  args.updWorkers = nWorkers;
  args.updT = kb.T;
  args.updTop = TopRequest{reinterpret_cast<TopOut *>(topOut), topN, topFlag, topFlagValue, topOut ? topCount : 0};
  finish_args(args);
  constexpr int WPQ = 4, NP = 2;
  const size_t shmem = eval_base_lds_bytes<WPQ, NP, false>(args.K, args.ldT) + eval_deferred_bytes(WPQ, NP, args.K, false);
  const bool pole = args.poleList != nullptr;
  auto kern = args.K == 5 ? (pole ? eval_questions_f64_upd<WPQ, NP, true, true, 5> : eval_questions_f64_upd<WPQ, NP, true, false, 5>)
                          : (pole ? eval_questions_f64_upd<WPQ, NP, true, true> : eval_questions_f64_upd<WPQ, NP, true, false>);
  static LaunchCache cache;
  const int dev = LaunchCache::Device();
  int cachedPerCU = 0;
  if (!cache.Get(dev, shmem, &cachedPerCU)) {
    if (shmem > 64 * 1024) {
      const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
      if (e != hipSuccess) return e;
    }
    int perCU = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&perCU, kern, WPQ * 64, shmem) != hipSuccess || perCU < 1) perCU = 1;
    if (perCU > 3) perCU = 3;   // as launch_reg_form
    cachedPerCU = perCU;
    cache.Put(dev, shmem, perCU);
  }
  const int64_t resident = (int64_t)cache.NumCUs(dev) * cachedPerCU;
  int64_t grid = kb.Q < resident ? kb.Q : resident;
  if (grid > kFusedMaxGrid) grid = kFusedMaxGrid;
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(WPQ * 64), shmem, stream, args);
  const hipError_t le = hipGetLastError();
  if (pole && le == hipSuccess && !args.fs.lazyFix) return launch_pole_fixup(args, stream);
  return le;
}

size_t EvalBatchPoleBytes(const KbView &kb, int nSlots) {
  if (kb.elem != 8 || kb.poleList == nullptr) return 0;
  const size_t cap = (size_t)kb.Q * (size_t)nSlots;
  return kBatchPoleClear + cap * sizeof(PoleEntry) + cap * (size_t)(2 * kb.K + 2) * sizeof(double);
}

hipError_t LaunchEvalQuestionsBatch(const KbView &kb, const QuizSlot *slots, int nSlots, int64_t qFirst, int64_t qLimit,
                                    int variant, const FusedSelect &fused, hipStream_t stream, void *pole) {
  if (qLimit <= qFirst || nSlots <= 0) return hipSuccess;
  if (slots == nullptr || fused.scratch == nullptr || fused.scratchStride <= 0) return hipErrorInvalidValue;
  EvalArgs args = make_args(kb, qFirst, qLimit);
  args.fs = fused;
  args.slots = slots;
  // the pole watch of a grid.y launch: one list for the batch's quizzes (an entry names its quiz), the records by entry; without the
  // caller's buffer the quizzes keep the sweep's own sums
  args.poleScratch = nullptr;
  args.poleList = nullptr;
  if (pole != nullptr && kb.poleList != nullptr && qFirst == 0 && qLimit == kb.Q) {
    char *p = static_cast<char *>(pole);
    args.poleList = reinterpret_cast<PoleHeader *>(p + 256 * sizeof(uint32_t));
    args.poleScratch = reinterpret_cast<double *>(p + kBatchPoleClear + (size_t)kb.Q * (size_t)nSlots * sizeof(PoleEntry));
  }
  finish_args(args);   // (a grid.y = quiz launch: never gated)
  return launch_variant(args, kb.ldT, variant, nSlots, stream);
}

template <int WPQ, int NP, bool DEFER, int KC>
static hipError_t launch_server_form(const EvalArgs &args, ServerMailbox *mb, void *requestLine, bool everyonePolls, ServerCtl *ctl,
                                uint64_t lastSeq, uint64_t idleTicks, hipStream_t stream) {
  const size_t stepOffset = eval_base_lds_bytes<WPQ, NP, false>(args.K, args.ldT) + (DEFER ? eval_deferred_bytes(WPQ, NP, args.K, false) : 0);
  const size_t shmem = stepOffset + 64;
  auto kern = eval_server_f64<WPQ, NP, DEFER, KC>;
  static LaunchCache cache;
  const int dev = LaunchCache::Device();
  int cachedPerCU = 0;
  if (!cache.Get(dev, shmem, &cachedPerCU)) {
    if (shmem > 64 * 1024) {
      const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
      if (e != hipSuccess) return e;
    }
    int perCU = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&perCU, kern, WPQ * 64, shmem) != hipSuccess || perCU < 1) perCU = 1;
    if (NP <= 2 && perCU > 3) perCU = 3;   // as launch_reg (measured for the resident kernel too: 46.0 k selections/s at three per CU, 43.9 k at four)
    cachedPerCU = perCU;
    cache.Put(dev, shmem, perCU);
  }
  const int gNumCUs = cache.NumCUs(dev);
  // every workgroup must be resident at once: the steps are collective
  const int64_t nQ = args.qLimit - args.qFirst, resident = (int64_t)gNumCUs * cachedPerCU;
  int64_t grid = nQ < resident ? nQ : resident;
  if (grid > kFusedMaxGrid) grid = kFusedMaxGrid;
  if (args.maxGrid > 0 && grid > args.maxGrid) grid = args.maxGrid;
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(WPQ * 64), shmem, stream, args, mb, reinterpret_cast<uint32_t *>(requestLine),
                     everyonePolls ? 1 : 0, ctl, lastSeq, idleTicks, (unsigned)stepOffset);
  return hipGetLastError();
}

template <int WPQ, int NP>
static hipError_t launch_server(const EvalArgs &args, ServerMailbox *mb, void *requestLine, bool everyonePolls, ServerCtl *ctl,
                                uint64_t lastSeq, uint64_t idleTicks, hipStream_t stream) {
  if (eval_defers_sums(WPQ, NP, false) && eval_base_lds_bytes<WPQ, NP, false>(args.K, args.ldT) + eval_deferred_bytes(WPQ, NP, args.K, false) + 64 <= kLdsPerCU)
    return args.K == 5 && WPQ == 4 && NP == 2 ? launch_server_form<WPQ, NP, true, (WPQ == 4 && NP == 2 ? 5 : 0)>(args, mb, requestLine, everyonePolls, ctl, lastSeq, idleTicks, stream)
                       : launch_server_form<WPQ, NP, true, 0>(args, mb, requestLine, everyonePolls, ctl, lastSeq, idleTicks, stream);
  return launch_server_form<WPQ, NP, false, 0>(args, mb, requestLine, everyonePolls, ctl, lastSeq, idleTicks, stream);
}

static int server_variant(const KbView &kb, int variant) {
  const int v = pick_variant(kb.ldT, variant);
  return v == 2 || (v == 8 && variant == 8) ? v : 0;   // wg256_np2: rows up to 1024 targets (wg128_np4: on request)
}
bool EvalVariantFusesSampled(const KbView &kb, int variant, int64_t nSubtasks) {   // a register shape, and the selection's LDS fits
  return pick_variant(kb.ldT, variant) != 99 && select_sampled_lds_doubles(kb.Q, nSubtasks) <= kLog2TableDoubles;
}
bool EvalVariantHasFinisherWorkgroup(const KbView &kb, int variant) { return pick_variant(kb.ldT, variant) != 99; }   // a register shape
bool EvalServerSupported(const KbView &kb, int variant) { return server_variant(kb, variant) != 0; }

hipError_t LaunchEvalServer(const KbView &kb, int64_t qFirst, int64_t qLimit, double *priority, int variant,
                            SelectResult *scratch, ServerMailbox *mailbox, void *requestLine, bool everyonePolls,
                            ServerCtl *ctl, uint64_t lastSeq, uint64_t idleTicks, TaggedPriority *hostPriority, hipStream_t stream,
                            bool watch) {
  if (qLimit <= qFirst || scratch == nullptr || mailbox == nullptr || requestLine == nullptr || ctl == nullptr)
    return hipErrorInvalidValue;
  EvalArgs args = make_args(kb, qFirst, qLimit);
  args.poleScratch = nullptr;   // (nothing can be launched behind a step of the resident kernel: it only says that the step wants redoing)
  args.poleList = nullptr;
  args.serverWatch = watch;
  args.priority = priority;
  args.fs.scratch = scratch;
  args.fs.hostPriority = hostPriority;
  args.poleGate = 0;
  switch (server_variant(kb, variant)) {
    case 2: return launch_server<4, 2>(args, mailbox, requestLine, everyonePolls, ctl, lastSeq, idleTicks, stream);
    case 8: return launch_server<2, 4>(args, mailbox, requestLine, everyonePolls, ctl, lastSeq, idleTicks, stream);
    default: return hipErrorNotSupported;
  }
}

}  // namespace pqa
