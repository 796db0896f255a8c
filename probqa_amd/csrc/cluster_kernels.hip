// cluster_kernels.hip -- the single-quiz priority sweep for LONG rows (more than 16384 targets), fp32 and fp64, gfx950.
//
// Reference: PqaCore/CEEvalQsSubtaskConsider.cpp:41-217.  The sweep makes two passes over a question's K answer rows -- pass 1
// W_k = sum_t (A/D) prior, pass 2 everything that needs 1/W_k -- and a row of 10^5 targets fits neither the registers nor the LDS
// of one workgroup: the plain streaming forms (eval_kernels.hip "stream256", batch_kernels.hip eval_questions_f32_stream) read
// every answer row twice and the mD row once per pass and answer (1.3 / 1.7 TB/s of cube at 2000 x 5 x 100000, fp64 / fp32).
//
// Here a question is swept by a CLUSTER of C workgroups, each owning one contiguous slice of the target axis:
//   * pass 1: a workgroup streams its slice of the question's K + 1 rows once (all requested before any is used), keeps the
//     likelihoods (A * invD) * prior of its slice in LDS, 1/D and the masked priors in registers, and publishes its partial W_k;
//   * the members exchange their partials as 16-byte records {value, question count of the cluster} written with one
//     write-through store each: no atomics, no fences, no counter.  Every member polls all C x K records of the question with one
//     coalesced round of loads past the L2s (value and tag arrive together) until all carry the question's tag, and adds them in
//     slice order -- the same W_k, bit for bit, in all of them;
//   * pass 2 runs on the LDS copy: posterior, log2, entropy / velocity sums, the lack term with one reciprocal per target over
//     the answers of the question (batch_kernels.hip); the partial sums go out the same way, and ONE member (they take turns)
//     folds them into the question's totals right after the NEXT question's exchange -- by then every member has published
//     them, so it never waits;
//   * the fp64 epilogues (:134-207) run in a second small kernel, one thread per question.
// The cube is read ONCE.  Two workgroups of different clusters share a CU, so that one streams while the other waits for its
// cluster.  The grid is exactly the number of workgroups the device holds at once (they wait for each other).
// Slices, partial sums and their order are fixed by (ldT, K, C) alone: results do not depend on timing.
// Three kernels: eval_cluster_kernel (question by question, as above), eval_cluster_ahead_kernel (round 4: pass 1 a question ahead of
// the exchange; any number of answers), eval_cluster_five_kernel (round 5: that form for questions of five answers -- 256 threads of
// two units, the loop's loads issued and waited for by hand; the default where it applies).  Options cluster_form / cluster_shape.
#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstdlib>

#include "eval_device.h"
#include "pole_device.h"
#include "pqa_device.h"
#include "pqa_kernels.h"

namespace pqa {

static __device__ double gLog2TableC[kLog2TableDoubles];   // this translation unit's copy of the Log2Hot table

hipError_t UploadLog2TableCluster(const double *hostTable) {
  return hipMemcpyToSymbol(HIP_SYMBOL(gLog2TableC), hostTable, kLog2TableDoubles * sizeof(double));
}

namespace {

constexpr int kClusterThreads = 512;
constexpr int kMaxWaves = kClusterThreads / kWave;
constexpr int kMaxK = 16;
constexpr int kExchangeDoubles = 1024;   // LDS: xch[C][K + 2] doubles -- clusters of up to 1024 / (K + 2) members
constexpr bool kClusterAheadByDefault = true;

template <typename R> struct Vec;
template <> struct Vec<float> { typedef float4 type; static constexpr int N = 4; };
template <> struct Vec<double> { typedef double2 type; static constexpr int N = 2; };

template <typename R> struct NumC;
template <> struct NumC<double> {
  static constexpr bool kTable = true;
  static __device__ __forceinline__ double log2p(double p, const double *tbl) { return log2hot(p, tbl); }
  static __device__ __forceinline__ double rcp(double x) {      // 2^-48.8: below the rounding of the sum it feeds
    const double r = __builtin_amdgcn_rcp(x);
    return fma(r, fma(-x, r, 1.0), r);
  }
  static __device__ __forceinline__ double inv(double x) { return div_nr(1.0, x); }   // exact quotient (:74, :91)
};
template <> struct NumC<float> {
  static constexpr bool kTable = false;
  static __device__ __forceinline__ float log2p(float p, const double *) {   // (batch_kernels.hip: Num<float>::log2p)
    return __builtin_amdgcn_fmed3f(__builtin_amdgcn_logf(p), -127.0f, -4.2992253e-08f);
  }
  static __device__ __forceinline__ float rcp(float x) { return __builtin_amdgcn_rcpf(x); }
  static __device__ __forceinline__ float inv(float x) { const float r = __builtin_amdgcn_rcpf(x); return fmaf(r, fmaf(-x, r, 1.0f), r); }
};

// a 16-byte unit of a cube row, with the hint that it will not be read again (the cube streams: pqa_device.h, row_load)
template <typename V> __device__ __forceinline__ V load_unit(const V *p) {
  typedef unsigned int u4n __attribute__((ext_vector_type(4)));
  return __builtin_bit_cast(V, __builtin_nontemporal_load(reinterpret_cast<const u4n *>(p)));
}
template <typename R> __device__ __forceinline__ R &at(typename Vec<R>::type &v, int e) { return reinterpret_cast<R *>(&v)[e]; }
template <typename R> __device__ __forceinline__ R at(const typename Vec<R>::type &v, int e) { return reinterpret_cast<const R *>(&v)[e]; }

__device__ __forceinline__ double wave_sum_d(double v) { return wave_sum(v); }   // (DPP + permlane swaps, pqa_device.h)

struct alignas(16) ExRec { double value; unsigned long long tag; };

struct ClusterArgs {
  const void *cube;           // R [Q][K+1][ldT]
  const double *prior;
  const uint32_t *tgap, *qgap, *asked;
  int64_t K, Q, ldT;
  int C, nClusters;           // workgroups per cluster, clusters (grid = C * nClusters)
  int sliceUnits;             // 16-byte units of a row per slice
  // exchange, per cluster g: recW[g][2][C][kMaxK], recS[g][2][C][kMaxK + 2] records (the [2]: parity of the question count)
  ExRec *recW, *recS;
  unsigned long long tagBase;  // launch number << 32: records of earlier launches never match
  double *totals;             // [Q][2 kMaxK + 2]: W_k | V_k | sum W_k H_k | lack
  double *priority;           // skipped questions get their 0 here
  // the pole watch (Double engines; pole_kernels.hip redoes the listed questions between this sweep and its epilogues): the list,
  // and per question the rows in which a member saw a posterior element within 2^-10 of 1 (several members may)
  PoleHeader *poleList;
  uint32_t *poleMask;
  unsigned long long *candSum;   // [Q][kMaxK]: the bits of the largest THREAD sum of likelihoods in each answer row, where some thread held a quarter of its wave's
                              // (cluster_watch_kernel, behind the sweep, compares it with W_k and lists the question's rows)
};
// The pole watch of the long-row sweeps (fp64; pole_kernels.hip redoes what it lists) leaves pass 2 -- at the edge of its 128
// registers -- alone.  Pass 1 votes, answer row by answer row: does a thread's sum of likelihoods (two elements) reach a quarter of
// its WAVE's sum?  That is necessary for one of its elements to hold a quarter of W_k, and never true on a fresh quiz's rows.  The
// votes of a question's rows are OR-ed as masks (no branch per row: the rows' butterflies overlap); where any was cast (rare, until
// the posterior has settled) the waves put the largest thread sum of every row into candSum[q][k] (an atomic max of the bits).
// Behind the sweep cluster_watch_kernel holds those against W_k as the sweeps of eval_kernels.hip do their lanes' sums: a share of
// 1 - 2^-9 lists the row, a quarter lists it if its velocity sum all but vanishes (pole_device.h: kSmallV).
// (Measured on the way at 2000 x 5 x 100000, fp64, form that runs ahead, 3.16 ms without any watch: the largest element tracked
//  inside pass 2 -- one integer max per element -- 4.12 ms, the scalar registers it pushed out going through VGPR lanes inside the
//  loop; the second stage as a rarely taken loop over the LDS copy in front of pass 2 3.37 ms, 3.29 of it for the code's mere presence.)
// the vote on the high words: four times the thread's sum, and a little (2^-9 ... 2^-8) more, reaches the wave's; all sums 0 does not pass
__device__ __forceinline__ unsigned long long cluster_cand_vote(double sum, double waveSum) {
  return __ballot((uint32_t)(d2u(sum) >> 32) >= (uint32_t)(d2u(waveSum) >> 32) - 0x00201000u);
}
__device__ __forceinline__ void cluster_cand_publish(unsigned long long *candSum, int64_t q, int64_t k, double sum, int lane) {
  const double mx = wave_max_d(sum);
  if (lane == 0) atomicMax(&candSum[q * kMaxK + k], d2u(mx));
}

// The members of a cluster run on different XCDs, whose L2s are not coherent with each other: records are written through (sc1)
// and read past the L2s (sc1), 16 bytes at a time -- value and tag travel together (as the sweep's winner records do,
// eval_kernels.hip: fused_select).
typedef unsigned int u4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void put_record(ExRec *p, double v, unsigned long long tag) {
  const unsigned long long w0 = d2u(v);
  const u4 x = {(unsigned)w0, (unsigned)(w0 >> 32), (unsigned)tag, (unsigned)(tag >> 32)};
  asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(x) : "memory");
}
// A record read past the L2s WITHOUT the compiler's wait accounting: the compiler drains the vector-memory counter wherever control
// flow has merged (s_waitcnt vmcnt(0) in front of the first use of a polled record, of pass 2's first LDS read, ...), i.e. it waits
// there for the next-but-one question's ROWS, requested a moment earlier and not needed for a whole iteration -- an HBM round trip
// (~2900 cycles of an iteration's 19 900, by the kernel's own clock) in series with everything else.  Loads issued here are waited
// for by wait_records<N>: "at most N younger vector-memory operations outstanding", N = the row loads issued behind them (loads
// return in order, so the records are in; a smaller N only waits longer).
// THE PRICE: between such a load and its wait the compiler believes the destination registers hold the value.  Were it to spill or
// copy them in that window (live-range splitting under register pressure) it would save stale contents and hand the registers to
// something else, which the landing load then overwrites -- seen in round 5 as a memory fault of a 512-thread form of the kernel
// below, built at the edge of its 128 registers.  Kernels that use these loads are therefore built with registers to spare, and
// tests/test_build_lint.py holds their compiled form to it: no scratch, no AGPR copies, at least 16 VGPRs unused.
__device__ __forceinline__ u4 get_record_async(const ExRec *p) {
  u4 x;
  asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(x) : "v"(p));
  return x;
}
// ... and the rows themselves, requested a whole iteration before pass 1 takes them: issued here (a streaming 16-byte load, as
// load_unit's) and waited for by rows_arrived at the top of pass 1 -- the compiler sees no vector-memory load inside the loop and
// puts no wait of its own there.
template <typename V> __device__ __forceinline__ u4 load_unit_async(const V *p) {
  u4 x;
  asm volatile("global_load_dwordx4 %0, %1, off nt" : "=v"(x) : "v"(p));
  return x;
}
// (The registers are named behind the wait, unit by unit.  Nothing in the language keeps the compiler from copying one of them between
//  the wait and the statement that names it -- tools/vmem_hazards.py reads the BUILT kernel and fails the build lint if it did.  One
//  statement that holds the wait and all twelve registers was measured: 1.5-4 % slower, the allocator has them all tied at once.)
template <int NU, int NR> __device__ __forceinline__ void rows_arrived(u4 (&d)[NU], u4 (&rows)[NR][NU]) {
  asm volatile("s_waitcnt vmcnt(0)");
#pragma unroll
  for (int j = 0; j < NU; j++) {   // (the uses stand behind these empty statements, which stand behind the wait)
    asm volatile("" : "+v"(d[j]));
#pragma unroll
    for (int k = 0; k < NR; k++) asm volatile("" : "+v"(rows[k][j]));
  }
}
template <int N, int M> __device__ __forceinline__ void wait_records(u4 (&x)[M]) {
  static_assert(M == 2 || M == 4, "records per thread");
  if constexpr (M == 2) asm volatile("s_waitcnt vmcnt(%2)" : "+v"(x[0]), "+v"(x[1]) : "n"(N));
  else asm volatile("s_waitcnt vmcnt(%4)" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]) : "n"(N));
}
// records asked for and waited for in ONE statement: there is no stretch of code in which the compiler believes the registers loaded
// while they are not (the retry rounds of a poll; the wait also drains whatever else is in flight -- retries are rare)
template <int M> __device__ __forceinline__ void get_records_now(const ExRec *(&p)[M], u4 (&x)[M]) {
  static_assert(M == 2 || M == 4, "records per thread");
  if constexpr (M == 2)
    asm volatile("global_load_dwordx4 %0, %2, off sc1\n\tglobal_load_dwordx4 %1, %3, off sc1\n\ts_waitcnt vmcnt(0)"
                 : "=&v"(x[0]), "=&v"(x[1]) : "v"(p[0]), "v"(p[1]) : "memory");
  else
    asm volatile("global_load_dwordx4 %0, %4, off sc1\n\tglobal_load_dwordx4 %1, %5, off sc1\n\tglobal_load_dwordx4 %2, %6, off sc1\n\t"
                 "global_load_dwordx4 %3, %7, off sc1\n\ts_waitcnt vmcnt(0)"
                 : "=&v"(x[0]), "=&v"(x[1]), "=&v"(x[2]), "=&v"(x[3]) : "v"(p[0]), "v"(p[1]), "v"(p[2]), "v"(p[3]) : "memory");
}
template <typename R, int NU>   // NU: 16-byte units of a slice per thread; two workgroups per CU (<= 128 registers)
__global__ __launch_bounds__(kClusterThreads, 4) void eval_cluster_kernel(ClusterArgs a) {
  typedef typename Vec<R>::type V;
  constexpr int VN = Vec<R>::N;
  constexpr int NW = kClusterThreads / kWave;
  extern __shared__ double smem[];
  const double *tbl = smem;
  // LDS (all dynamic: the Log2Hot table must sit at address 0): table | likelihoods [K][sliceUnits] | the waves' partials
  // [kMaxK + 2][NW] | W_k [kMaxK] | the waves' votes [NW] | the members' partials [C][K + 2]
  V *lhL = reinterpret_cast<V *>(smem + (NumC<R>::kTable ? kLog2TableDoubles : 0));
  double (*red)[NW] = reinterpret_cast<double (*)[NW]>(lhL + (size_t)a.K * a.sliceUnits);
  double *wTot = &red[kMaxK + 2][0];
  double *xch = wTot + kMaxK + NW / 2;                          // [C][K + 2]: the members' partials of one question
  if constexpr (NumC<R>::kTable) {
    if (!lds_table_at_zero(tbl)) __builtin_trap();            // log2hot addresses the table absolutely
    for (int i = threadIdx.x; i < kLog2TableDoubles; i += kClusterThreads) smem[i] = gLog2TableC[i];
  }
  const int tid = threadIdx.x, lane = tid % kWave, wave = tid / kWave;
  const int64_t K = a.K, ldT = a.ldT;
  const int C = a.C, g = blockIdx.x / C, m = blockIdx.x % C;   // cluster, member (= slice)
  const int nUnits = (int)(ldT / VN), SU = a.sliceUnits;
  const int u0 = m * SU;                                        // first unit of the slice
  const int nMine = u0 >= nUnits ? 0 : (nUnits - u0 < SU ? nUnits - u0 : SU);
  const R *cube = static_cast<const R *>(a.cube);
  const int64_t qStride = (K + 1) * ldT;
  // ---- the thread's units of the slice, its masked priors (:103) and gap bits
  int ui[NU];            // unit index within the row (clamped; units beyond the slice are masked)
  uint32_t gapBits[NU];
  V pr[NU];
#pragma unroll
  for (int j = 0; j < NU; j++) {
    const int s = tid + j * kClusterThreads;
    const bool in = s < nMine;
    ui[j] = in ? u0 + s : (nUnits - 1);
    const int64_t t0 = (int64_t)ui[j] * VN;
    const uint32_t gbits = in ? (a.tgap[t0 >> 5] >> (t0 & 31)) & ((1u << VN) - 1) : (1u << VN) - 1;   // (bits past T are set)
    gapBits[j] = gbits;
#pragma unroll
    for (int e = 0; e < VN; e++) at<R>(pr[j], e) = ((gbits >> e) & 1) ? (R)0 : (R)a.prior[t0 + e];
  }
  ExRec *recW = a.recW + (size_t)g * 2 * C * kMaxK, *recS = a.recS + (size_t)g * 2 * C * (kMaxK + 2);
  unsigned long long round = 0;                                 // questions this cluster has swept
  const unsigned long long tagBase = a.tagBase;
  int64_t qPrev = -1;                                           // the previous question, whose pass-2 partials are still to be folded
  // All C x n records of one exchange (n values per member, tag = the question's count), one coalesced round of loads per poll,
  // into xch[member][n].  Returns once every record carries the tag.
  // The loads are compiler-visible buffer loads with the sc1 bit (past the L2s), not inline assembly: `between` -- the request for
  // the next question's rows -- is issued BEHIND the first round of record loads and the wait for that round leaves the younger
  // row loads in flight (loads return in order: behind the rows, the first poll waited for all of them -- 4 us per question).
  // Every wave polls for ITS records (thread t: records t and t + 512) until they all carry the tag, then ONE barrier (round 3).
  // Polling in lock step -- two barriers and a vote of the eight waves per round -- made a round ~2 us and an exchange three
  // rounds on average (13.5 M vector loads per launch against 7 M of rows in the counters): 6 of a question's 9 us.
  auto gather = [&](const ExRec *recs, int stride, int n, unsigned long long tag, auto &&between) {
    const int total = C * n;                                    // <= 1024 (cluster_shape)
    const RowRsrc rs = row_rsrc(recs, (int64_t)C * stride * (int64_t)sizeof(ExRec));
    constexpr int kAuxSc1 = 16;
    uint32_t off[2];
#pragma unroll
    for (int u = 0; u < 2; u++) {
      const int r = tid + u * kClusterThreads;
      const int rc = r < total ? r : 0;
      const int mm = rc / n, kk = rc - mm * n;
      off[u] = (uint32_t)((mm * stride + kk) * (int)sizeof(ExRec));
    }
    unsigned spins = 0;
    for (bool first = true;; first = false) {
      u32x4_t x[2];
#pragma unroll
      for (int u = 0; u < 2; u++)
        if (u == 0 || total > kClusterThreads) x[u] = __builtin_amdgcn_raw_buffer_load_b128(rs, off[u], 0, kAuxSc1);
      if (first) between();
      int ok = 1;
#pragma unroll
      for (int u = 0; u < 2; u++) {
        const int r = tid + u * kClusterThreads;
        if ((u == 0 || total > kClusterThreads) && r < total) {
          const unsigned long long t = (unsigned long long)x[u][2] | ((unsigned long long)x[u][3] << 32);
          if (t == tag) xch[r] = u2d((unsigned long long)x[u][0] | ((unsigned long long)x[u][1] << 32)); else ok = 0;
        }
      }
      if (__all(ok)) break;                                     // (this wave's records: the other waves wait for theirs)
      __builtin_amdgcn_s_sleep(1);
      if (++spins > (1u << 26)) __builtin_trap();               // (minutes: a member died -- no silent hang)
    }
    __syncthreads();
  };
  // sums over the members of the n values of xch[member][n], into out[0 .. n): wave w takes the columns w, w + NW, ...; lane l adds
  // the members l, l + 64, ... and the wave's DPP tree the lanes (one member after the other it is C dependent LDS round trips:
  // 7600 cycles at 98 members).  A fixed order, the same in every member of the cluster.
  auto sum_members = [&](int n, double *out) {
    for (int col = wave; col < n; col += NW) {
      double s = 0.0;
      for (int i = lane; i < C; i += kWave) s += xch[i * n + col];
      s = wave_sum(s);
      if (lane == 0) out[col] = s;
    }
  };
  // the pass-2 partials of question `qq` (count `cnt`), folded into its totals; W_k passed along
  auto fold = [&](int64_t qq, unsigned long long cnt, const double *wOfQ) {
    gather(recS + (size_t)(cnt & 1) * C * (kMaxK + 2), kMaxK + 2, (int)K + 2, tagBase + cnt + 1, [] {});
    sum_members((int)K + 2, red[0]);                            // (red[] is free between the exchange and pass 2)
    __syncthreads();
    if (tid < K + 2) {
      const double s = red[0][tid];
      double *tot = a.totals + (size_t)qq * (2 * kMaxK + 2);
      if (tid < K) { tot[tid] = wOfQ[tid]; tot[kMaxK + tid] = s; }
      else tot[2 * kMaxK + (tid - K)] = s;
    }
    __syncthreads();
  };

  auto next_valid = [&](int64_t q) {    // :54 gap / asked questions get priority 0 and are skipped (the same decision in every member)
    while (q < a.Q && (bit_test(a.qgap, q) || bit_test(a.asked, q))) {
      if (m == 0 && tid == 0) a.priority[q] = 0.0;
      q += a.nClusters;
    }
    return q;
  };
  auto load_units = [&](const R *rowPtr, V (&dst)[NU]) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < NU; j++) dst[j] = load_unit(reinterpret_cast<const V *>(rowPtr) + ui[j]);
  };
  if constexpr (NumC<R>::kTable) __syncthreads();
  int64_t q = next_valid(g);
  // A thread's share of a row is one or two 16-byte units, i.e. a row costs it one memory round trip whatever its length: requested
  // a row ahead the K + 1 rows of a question are K + 1 round trips in a row (12 of the 15 us a question took).  So ALL rows of the
  // NEXT question (up to kRowsAhead answer rows; the few beyond are fetched on the spot) are requested at once, right after pass 1
  // has consumed the current ones, and fly during the exchange and pass 2.
  constexpr int kRowsAhead = NU == 1 ? 8 : 5;   // (two units per thread: five rows are what 128 registers hold)
  V dN[NU], rows[kRowsAhead][NU];
  auto request_question = [&](int64_t qq) __attribute__((always_inline)) {
    const R *base = cube + qq * qStride;
    load_units(base + K * ldT, dN);
#pragma unroll
    for (int k = 0; k < kRowsAhead; k++)
      if (k < K) load_units(base + k * ldT, rows[k]);
  };
  if (q < a.Q) request_question(q);
  while (q < a.Q) {
    const int par = (int)(round & 1);
    const R *qb = cube + q * qStride;
    const int64_t qNext = next_valid(q + a.nClusters);
    // ---- pass 1 (:66-88) on the slice of the K + 1 rows
    V id[NU];
#pragma unroll
    for (int j = 0; j < NU; j++)
#pragma unroll
      for (int e = 0; e < VN; e++) at<R>(id[j], e) = ((gapBits[j] >> e) & 1) ? (R)0 : NumC<R>::inv(at<R>(dN[j], e));   // :74
    [[maybe_unused]] unsigned long long candVotes = 0;
    auto pass1_row = [&](int64_t k, const V (&row)[NU]) __attribute__((always_inline)) {
      R s = (R)0;
#pragma unroll
      for (int j = 0; j < NU; j++) {
        V lh;
#pragma unroll
        for (int e = 0; e < VN; e++) {
          at<R>(lh, e) = (at<R>(row[j], e) * at<R>(id[j], e)) * at<R>(pr[j], e);   // :81-82
          s += at<R>(lh, e);
        }
        const int sl = tid + j * kClusterThreads;
        if (sl < SU) lhL[k * SU + sl] = lh;
      }
      const double sw = wave_sum_d((double)s);
      if (lane == 0) red[k][wave] = sw;
      if constexpr (NumC<R>::kTable) candVotes |= cluster_cand_vote((double)s, sw);   // (the pole watch)
    };
#pragma unroll
    for (int k = 0; k < kRowsAhead; k++)
      if (k < K) pass1_row(k, rows[k]);
    for (int64_t k = kRowsAhead; k < K; k++) {                  // (more than kRowsAhead answers)
      V late[NU];
      load_units(qb + k * ldT, late);
      pass1_row(k, late);
    }
    if constexpr (NumC<R>::kTable) {
      if (candVotes != 0 && a.candSum != nullptr) {             // (rare: the rows' thread sums again, from the LDS copy this thread wrote)
        for (int64_t k = 0; k < K; k++) {
          R sk = (R)0;
#pragma unroll
          for (int j = 0; j < NU; j++) {
            const int sl = tid + j * kClusterThreads;
            if (sl < SU) {
              const V lh = lhL[k * SU + sl];
#pragma unroll
              for (int e = 0; e < VN; e++) sk += at<R>(lh, e);
            }
          }
          cluster_cand_publish(a.candSum, q, k, (double)sk, lane);
        }
      }
    }
    __syncthreads();
    if (tid < K) {
      double w = 0.0;
      for (int i = 0; i < NW; i++) w += red[tid][i];
      put_record(recW + ((size_t)par * C + m) * kMaxK + tid, w, tagBase + round + 1);
    }
    // ---- the cluster meets: everybody's partials, in slice order
    gather(recW + (size_t)par * C * kMaxK, kMaxK, (int)K, tagBase + round + 1, [&]() __attribute__((always_inline)) {
      if (qNext < a.Q) request_question(qNext);                 // the next question's rows fly during the exchange and pass 2
    });
    sum_members((int)K, red[8]);                                // W_k of this question, parked (rows 8.. of red[]; fold uses rows 0..2) while wTot still holds the previous one's
    __syncthreads();                                            // (xch is read)
    // the member whose turn it is folds the PREVIOUS question's pass-2 partials: every member published them before it published
    // this question's W partials, which have all just been seen
    if (qPrev >= 0 && (int)((round - 1) % (unsigned long long)C) == m) fold(qPrev, round - 1, wTot);
    if (tid < K) wTot[tid] = red[8][tid];
    __syncthreads();
    // ---- pass 2 (:95-128) from LDS, answer by answer; the lack term's N / D pairs (batch_kernels.hip) run across the answers
    V accN[NU], accD[NU];
    R hW = (R)0, accL = (R)0;
    for (int64_t k = 0; k < K; k++) {
      const R invWk = (R)div_fast(1.0, wTot[k]);                // :91
      R vk = (R)0;
#pragma unroll
      for (int j = 0; j < NU; j++) {
        const int sl = tid + j * kClusterThreads;
        if (sl < SU) {
          const V lh = lhL[k * SU + sl];
#pragma unroll
          for (int e = 0; e < VN; e++) {
            const R l = at<R>(lh, e), pi = at<R>(pr[j], e);
            const R p = l * invWk;                              // :97
            const R l2 = NumC<R>::log2p(p, tbl);                // :106
            hW = fma(l, l2, hW);                                // :113-114 weighted by W_k (eval_epilogue)
            const R dd = p - pi;                                // :119
            vk = fma(dd, dd, vk);                               // :126-127
            // :117 sum_k 1 / log2 p_k = N / D, built answer by answer
            if (k == 0) { at<R>(accN[j], e) = (R)1; at<R>(accD[j], e) = l2; }
            else { at<R>(accN[j], e) = fma(at<R>(accN[j], e), l2, at<R>(accD[j], e)); at<R>(accD[j], e) = at<R>(accD[j], e) * l2; }
          }
        }
      }
      const double s = wave_sum_d((double)vk);
      if (lane == 0) red[k][wave] = s;
    }
#pragma unroll
    for (int j = 0; j < NU; j++) {
      const int sl = tid + j * kClusterThreads;
      if (sl < SU) {
#pragma unroll
        for (int e = 0; e < VN; e++) {
          const R i1 = at<R>(id[j], e);
          accL = fma((i1 * at<R>(accN[j], e)) * i1, NumC<R>::rcp(at<R>(accD[j], e)), accL);
        }
      }
    }
    {
      const double s1 = wave_sum_d((double)hW), s2 = wave_sum_d((double)accL);
      if (lane == 0) { red[K][wave] = s1; red[K + 1][wave] = s2; }
    }
    __syncthreads();
    if (tid < K + 2) {
      double s = 0.0;
      for (int i = 0; i < NW; i++) s += red[tid][i];
      put_record(recS + ((size_t)par * C + m) * (kMaxK + 2) + tid, s, tagBase + round + 1);
    }
    qPrev = q;
    round++;
    q = qNext;
    __syncthreads();                                            // red[] and the LDS rows are free again
  }
  // the last question's partials: its turn-taker waits for them (the only wait of its kind)
  if (qPrev >= 0 && (int)((round - 1) % (unsigned long long)C) == m) fold(qPrev, round - 1, wTot);
}

// ------------------------------------------------------------------------------------------------------------------
// The same sweep with the exchange of question i under the work of its neighbours (round 4).  In the form above a
// member's question is a chain -- pass 1, publish, wait for everybody's partials (a write-through, a read past the L2s and the skew
// of ~100 members: 2.7 of a question's 8.1 us), pass 2 -- and only the CU's second workgroup fills the wait.  Here a member runs pass
// 1 of the NEXT question before it asks for the current one's partials: the likelihoods of question i + 1 wait in REGISTERS (one
// 16-byte unit per thread and answer) while pass 2 of question i still owns the LDS copy, and move there when it is done -- the LDS
// and the slices stay what they are (two workgroups per CU), no second LDS copy (which would halve the slices and double the members
// and the records of every exchange, or cost the second workgroup).  By the time a member polls for question i its partials have been
// on their way for a whole pass 2 + pass 1.
//   iteration i:  pass 1 (q[i+1]) from the arrived rows -> its likelihoods in registers, publish its partial W   |  request the rows of q[i+2]
//                 gather W (q[i])  ->  the turn-taker folds q[i-2]  ->  pass 2 (q[i]) from LDS, publish its sums  |  likelihoods of q[i+1] -> LDS
// Records: a member is now up to two questions ahead of another (it publishes W of q[i+1] when it has seen everybody's W of q[i-1]),
// so the record buffers have FOUR slots (question count mod 4) instead of two, and the fold lags two questions (everybody published
// the sums of q[i-2] before its W of q[i], which have all been seen).  One unit per thread (NU = 1): rows of up to 512 units per member.
template <typename R>
__global__ __launch_bounds__(kClusterThreads, 4) void eval_cluster_ahead_kernel(ClusterArgs a) {
  typedef typename Vec<R>::type V;
  constexpr int VN = Vec<R>::N;
  constexpr int NW = kClusterThreads / kWave;
  constexpr int kRows = 5;                                      // answer rows requested ahead (further ones are fetched on the spot)
  extern __shared__ double smem[];
  const double *tbl = smem;
  V *lhL = reinterpret_cast<V *>(smem + (NumC<R>::kTable ? kLog2TableDoubles : 0));
  double (*red)[NW] = reinterpret_cast<double (*)[NW]>(lhL + (size_t)a.K * a.sliceUnits);
  double *wTot = &red[kMaxK + 2][0];
  double *xch = wTot + kMaxK + NW / 2;                          // [C][K + 2]: the members' partials of one question
  double *wHist = xch + kExchangeDoubles;                       // [4][kMaxK]: W_k of the last four questions (the fold lags two)
  double *wInv = wHist + 4 * kMaxK;                             // [kMaxK]: 1 / W_k of the question in pass 2
  float *park = reinterpret_cast<float *>(wInv + kMaxK);        // fp32: [K][threads] -- the lanes' pass-1 sums, added by column (kParkPass1)
  constexpr bool kParkPass1 = !NumC<R>::kTable;                 // (Float engines have the Log2Hot table's 16 KB of LDS to spare)
  if constexpr (NumC<R>::kTable) {
    if (!lds_table_at_zero(tbl)) __builtin_trap();            // log2hot addresses the table absolutely
    for (int i = threadIdx.x; i < kLog2TableDoubles; i += kClusterThreads) smem[i] = gLog2TableC[i];
  }
  const int tid = threadIdx.x, lane = tid % kWave, wave = tid / kWave;
  const int64_t K = a.K, ldT = a.ldT;
  const int C = a.C, g = blockIdx.x / C, m = blockIdx.x % C;   // cluster, member (= slice)
  const int nUnits = (int)(ldT / VN), SU = a.sliceUnits;
  const int u0 = m * SU;
  const int nMine = u0 >= nUnits ? 0 : (nUnits - u0 < SU ? nUnits - u0 : SU);
  const R *cube = static_cast<const R *>(a.cube);
  const int64_t qStride = (K + 1) * ldT;
  const bool in = tid < nMine, inSlice = tid < SU;
  const int ui = in ? u0 + tid : (nUnits - 1);                  // unit index within the row (clamped; units beyond the slice are masked)
  const int64_t t0 = (int64_t)ui * VN;
  const uint32_t gapBits = in ? (a.tgap[t0 >> 5] >> (t0 & 31)) & ((1u << VN) - 1) : (1u << VN) - 1;   // (bits past T are set)
  V pr;
#pragma unroll
  for (int e = 0; e < VN; e++) at<R>(pr, e) = ((gapBits >> e) & 1) ? (R)0 : (R)a.prior[t0 + e];      // :103
  ExRec *recW = a.recW + (size_t)g * 4 * C * kMaxK, *recS = a.recS + (size_t)g * 4 * C * (kMaxK + 2);
  const unsigned long long tagBase = a.tagBase;

  auto gather = [&](const ExRec *recs, int stride, int n, unsigned long long tag) {   // (as eval_cluster_kernel's)
    const int total = C * n;
    const RowRsrc rs = row_rsrc(recs, (int64_t)C * stride * (int64_t)sizeof(ExRec));
    constexpr int kAuxSc1 = 16;
    uint32_t off[2];
#pragma unroll
    for (int u = 0; u < 2; u++) {
      const int r = tid + u * kClusterThreads;
      const int rc = r < total ? r : 0;
      const int mm = rc / n, kk = rc - mm * n;
      off[u] = (uint32_t)((mm * stride + kk) * (int)sizeof(ExRec));
    }
    unsigned spins = 0;
    for (;;) {
      u32x4_t x[2];
#pragma unroll
      for (int u = 0; u < 2; u++)
        if (u == 0 || total > kClusterThreads) x[u] = __builtin_amdgcn_raw_buffer_load_b128(rs, off[u], 0, kAuxSc1);
      int ok = 1;
#pragma unroll
      for (int u = 0; u < 2; u++) {
        const int r = tid + u * kClusterThreads;
        if ((u == 0 || total > kClusterThreads) && r < total) {
          const unsigned long long t = (unsigned long long)x[u][2] | ((unsigned long long)x[u][3] << 32);
          if (t == tag) xch[r] = u2d((unsigned long long)x[u][0] | ((unsigned long long)x[u][1] << 32)); else ok = 0;
        }
      }
      if (__all(ok)) break;
      __builtin_amdgcn_s_sleep(1);
      if (++spins > (1u << 26)) __builtin_trap();               // (minutes: a member died -- no silent hang)
    }
    __syncthreads();
  };
  auto sum_members = [&](int n, double *out) {
    for (int col = wave; col < n; col += NW) {
      double s = 0.0;
      for (int i = lane; i < C; i += kWave) s += xch[i * n + col];
      s = wave_sum(s);
      if (lane == 0) out[col] = s;
    }
  };
  auto fold = [&](int64_t qq, unsigned long long cnt) {
    gather(recS + (size_t)(cnt & 3) * C * (kMaxK + 2), kMaxK + 2, (int)K + 2, tagBase + cnt + 1);
    sum_members((int)K + 2, red[0]);
    __syncthreads();
    if (tid < K + 2) {
      const double s = red[0][tid];
      double *tot = a.totals + (size_t)qq * (2 * kMaxK + 2);
      if (tid < K) { tot[tid] = wHist[(cnt & 3) * kMaxK + tid]; tot[kMaxK + tid] = s; }
      else tot[2 * kMaxK + (tid - K)] = s;
    }
    __syncthreads();
  };
  auto next_valid = [&](int64_t q) {    // :54
    while (q < a.Q && (bit_test(a.qgap, q) || bit_test(a.asked, q))) {
      if (m == 0 && tid == 0) a.priority[q] = 0.0;
      q += a.nClusters;
    }
    return q;
  };
  V dN, rows[kRows];
  auto request_question = [&](int64_t qq) __attribute__((always_inline)) {
    const R *base = cube + qq * qStride;
    dN = load_unit(reinterpret_cast<const V *>(base + K * ldT) + ui);
#pragma unroll
    for (int k = 0; k < kRows; k++)
      if (k < K) rows[k] = load_unit(reinterpret_cast<const V *>(base + k * ldT) + ui);
  };
  // pass 1 of question qq (count cnt) on the rows that have arrived: 1/D and the likelihoods of the first kRows answers into idOut /
  // lhOut (registers), further answers' likelihoods straight into LDS when `direct` (the first question) or not at all here -- they
  // are formed again when the registers move to LDS (store_question); the partial W_k published.
  auto pass1 = [&](int64_t qq, unsigned long long cnt, V &idOut, V (&lhOut)[kRows]) __attribute__((always_inline)) {
#pragma unroll
    for (int e = 0; e < VN; e++) at<R>(idOut, e) = ((gapBits >> e) & 1) ? (R)0 : NumC<R>::inv(at<R>(dN, e));   // :74
    [[maybe_unused]] unsigned long long candVotes = 0;
    auto row_sum = [&](int64_t k, const V &row, V &lh, bool late) __attribute__((always_inline)) {
      R sum = (R)0;
#pragma unroll
      for (int e = 0; e < VN; e++) {
        at<R>(lh, e) = (at<R>(row, e) * at<R>(idOut, e)) * at<R>(pr, e);   // :81-82
        sum += at<R>(lh, e);
      }
      if constexpr (kParkPass1) {
        park[k * kClusterThreads + tid] = (float)sum;           // (a butterfly of six DPP steps per answer otherwise: a sixth of the iteration's instructions)
      } else {
        const double sw = wave_sum_d((double)sum);
        if (lane == 0) red[k][wave] = sw;
        if constexpr (NumC<R>::kTable) { const unsigned long long vote = cluster_cand_vote((double)sum, sw); candVotes |= vote; if (late && vote != 0 && a.candSum != nullptr) cluster_cand_publish(a.candSum, qq, k, (double)sum, lane); }   // (the pole watch)
      }
    };
#pragma unroll
    for (int k = 0; k < kRows; k++)
      if (k < K) row_sum(k, rows[k], lhOut[k], false);
    for (int64_t k = kRows; k < K; k++) {                       // (more than kRows answers: summed here, formed again for LDS later)
      const V late = load_unit(reinterpret_cast<const V *>(cube + qq * qStride + k * ldT) + ui);
      V lh;
      row_sum(k, late, lh, true);
    }
    if constexpr (NumC<R>::kTable) {
      if (candVotes != 0 && a.candSum != nullptr) {             // (rare: the rows' thread sums again, from the registers; rows beyond kRows published theirs)
#pragma unroll
        for (int k = 0; k < kRows; k++)
          if (k < K) {
            R sk = (R)0;
#pragma unroll
            for (int e = 0; e < VN; e++) sk += at<R>(lhOut[k], e);
            cluster_cand_publish(a.candSum, qq, k, (double)sk, lane);
          }
      }
    }
    __syncthreads();
    if constexpr (kParkPass1) {
      const int grp = tid >> 5, l32 = tid & 31;                 // sixteen groups of 32 lanes, a column (answer) each
      for (int col = grp; col < K; col += kClusterThreads / 32) {
        const float *src = park + col * kClusterThreads + l32;
        double acc = 0.0;
#pragma unroll
        for (int i = 0; i < kClusterThreads / 32; i++) acc += (double)src[32 * i];
        acc += mov_dpp<kDppXor1>(acc);
        acc += mov_dpp<kDppXor2>(acc);
        acc += mov_dpp<kDppHalfMirror>(acc);
        acc += mov_dpp<kDppMirror>(acc);
        const Pair pq = swap16(acc);
        acc = pq.a + pq.b;
        if (l32 == 0) put_record(recW + ((size_t)(cnt & 3) * C + m) * kMaxK + col, acc, tagBase + cnt + 1);
      }
    } else {
      if (tid < K) {
        double w = 0.0;
        for (int i = 0; i < NW; i++) w += red[tid][i];
        put_record(recW + ((size_t)(cnt & 3) * C + m) * kMaxK + tid, w, tagBase + cnt + 1);
      }
    }
  };
  // the likelihoods of question qq into LDS (pass 2 of the question before it is done with the copy: the caller's barrier)
  auto store_question = [&](int64_t qq, const V &idOf, const V (&lhOf)[kRows]) __attribute__((always_inline)) {
    if (inSlice) {
#pragma unroll
      for (int k = 0; k < kRows; k++)
        if (k < K) lhL[k * SU + tid] = lhOf[k];
      for (int64_t k = kRows; k < K; k++) {
        const V late = load_unit(reinterpret_cast<const V *>(cube + qq * qStride + k * ldT) + ui);
        V lh;
#pragma unroll
        for (int e = 0; e < VN; e++) at<R>(lh, e) = (at<R>(late, e) * at<R>(idOf, e)) * at<R>(pr, e);
        lhL[k * SU + tid] = lh;
      }
    }
  };

  if constexpr (NumC<R>::kTable) __syncthreads();
  unsigned long long round = 0;                                 // count of the question in pass 2 (the cluster's questions so far)
  int64_t q = next_valid(g), qPrev = -1, qPrev2 = -1;
  V id, idNext, lhNext[kRows];
  if (q < a.Q) {
    request_question(q);
    pass1(q, 0, idNext, lhNext);                                // (the first question: nothing to hide behind)
    store_question(q, idNext, lhNext);
    id = idNext;
  }
  int64_t qNext = q < a.Q ? next_valid(q + a.nClusters) : a.Q;
  if (qNext < a.Q) request_question(qNext);
  while (q < a.Q) {
    // ---- pass 1 of the NEXT question; then its successor's rows are requested into the registers that are free again
    const int64_t qNext2 = qNext < a.Q ? next_valid(qNext + a.nClusters) : a.Q;
    if (qNext < a.Q) {
      pass1(qNext, round + 1, idNext, lhNext);
      if (qNext2 < a.Q) request_question(qNext2);
    }
    // ---- everybody's partial W of THIS question (published an iteration ago), in slice order
    gather(recW + (size_t)(round & 3) * C * kMaxK, kMaxK, (int)K, tagBase + round + 1);
    // (two forms of the same sums, each kept where it measured faster on one box at 2000 x 5 x 100000: the sums written straight to
    //  where pass 2 and the fold look, one barrier less -- fp32 1517 -> 1489 us, fp64 3206 -> 3262 us)
    constexpr bool kDirectSums = !NumC<R>::kTable;
    if constexpr (kDirectSums) {
      for (int col = wave; col < K; col += NW) {
        double w = 0.0;
        for (int i = lane; i < C; i += kWave) w += xch[i * (int)K + col];
        w = wave_sum(w);
        if (lane == 0) {
          wInv[col] = div_fast(1.0, w);                         // :91, once per workgroup (every thread formed it: a fifth of pass 2's instructions)
          wHist[(round & 3) * kMaxK + col] = w;
        }
      }
      __syncthreads();                                          // (xch is read, W is there)
    } else {
      sum_members((int)K, red[8]);
      __syncthreads();                                          // (xch is read)
    }
    // the turn-taker folds the question before the previous one: every member published its sums before it published its W of this
    // question, which have all just been seen
    if (qPrev2 >= 0 && (int)((round - 2) % (unsigned long long)C) == m) fold(qPrev2, round - 2);
    if constexpr (!kDirectSums) {
      if (tid < K) {
        const double w = red[8][tid];
        wInv[tid] = div_fast(1.0, w);                           // :91, once per workgroup
        wHist[(round & 3) * kMaxK + tid] = w;
      }
      __syncthreads();
    }
    // ---- pass 2 (:95-128) from LDS, answer by answer.  The lanes' sums are not reduced wave by wave (K + 2 butterflies of six DPP
    // steps each were a quarter of the iteration's instructions): a lane parks its K + 2 sums in the LDS slots of its own unit --
    // 16 bytes per answer row, dead once the row's likelihoods have been read -- and sixteen groups of 32 lanes add one column each.
    double *slot = reinterpret_cast<double *>(lhL);             // unit u of row k: slot[(k * SU + u) * 2 + {0, 1}]
    V accN, accD;
    R hW = (R)0, accL = (R)0;
    for (int64_t k = 0; k < K; k++) {
      const R invWk = (R)wInv[k];                               // :91 (formed once per workgroup, above)
      R vk = (R)0;
      if (inSlice) {
        const V lh = lhL[k * SU + tid];
#pragma unroll
        for (int e = 0; e < VN; e++) {
          const R l = at<R>(lh, e), pi = at<R>(pr, e);
          const R p = l * invWk;                                // :97
          const R l2 = NumC<R>::log2p(p, tbl);                  // :106
          hW = fma(l, l2, hW);                                  // :113-114 weighted by W_k (eval_epilogue)
          const R dd = p - pi;                                  // :119
          vk = fma(dd, dd, vk);                                 // :126-127
          if (k == 0) { at<R>(accN, e) = (R)1; at<R>(accD, e) = l2; }   // :117 sum_k 1 / log2 p_k = N / D, answer by answer
          else { at<R>(accN, e) = fma(at<R>(accN, e), l2, at<R>(accD, e)); at<R>(accD, e) = at<R>(accD, e) * l2; }
        }
        slot[(k * SU + tid) * 2] = (double)vk;
      }
    }
    if (inSlice) {
#pragma unroll
      for (int e = 0; e < VN; e++) {
        const R i1 = at<R>(id, e);
        accL = fma((i1 * at<R>(accN, e)) * i1, NumC<R>::rcp(at<R>(accD, e)), accL);
      }
      slot[tid * 2 + 1] = (double)hW;                           // (row 0's second double)
      if (K > 1) slot[(SU + tid) * 2 + 1] = (double)accL;       // (row 1's)
    }
    if (K == 1) {                                               // (one answer: one row of slots -- the lack sum the old way)
      const double s2 = wave_sum_d((double)accL);
      if (lane == 0) red[0][wave] = s2;
    }
    __syncthreads();                                            // (the slots are complete)
    {
      const int grp = tid >> 5, l32 = tid & 31;                 // column grp of the K + 2: V_k (k < K) | sum l log2 p | lack
      if (grp < K + 2) {
        const double *src = grp < K ? slot + (size_t)grp * SU * 2 : grp == K ? slot + 1 : slot + (size_t)SU * 2 + 1;
        double acc = 0.0;
        if (grp == K + 1 && K == 1) {
          if (l32 < NW) acc = red[0][l32];
        } else {
          for (int u = l32; u < SU; u += 32) acc += src[(size_t)u * 2];
        }
        acc += mov_dpp<kDppXor1>(acc);
        acc += mov_dpp<kDppXor2>(acc);
        acc += mov_dpp<kDppHalfMirror>(acc);
        acc += mov_dpp<kDppMirror>(acc);
        const Pair pq = swap16(acc);
        acc = pq.a + pq.b;
        if (l32 == 0) put_record(recS + ((size_t)(round & 3) * C + m) * (kMaxK + 2) + grp, acc, tagBase + round + 1);
      }
    }
    __syncthreads();                                            // (the slots are read: the LDS copy is free)
    // ---- the next question's likelihoods move from the registers to LDS
    if (qNext < a.Q) { store_question(qNext, idNext, lhNext); id = idNext; }
    qPrev2 = qPrev;
    qPrev = q;
    round++;
    q = qNext;
    qNext = qNext2;
    if constexpr (!kDirectSums) __syncthreads();                // (not needed for the data: the next reader of the LDS copy is pass 2, behind pass 1's and the exchange's barriers)
  }
  __syncthreads();
  // the last two questions' sums: their turn-takers wait for them
  if (qPrev2 >= 0 && (int)((round - 2) % (unsigned long long)C) == m) fold(qPrev2, round - 2);
  if (qPrev >= 0 && (int)((round - 1) % (unsigned long long)C) == m) fold(qPrev, round - 1);
}

// ------------------------------------------------------------------------------------------------------------------
// The form that runs ahead for questions of exactly FIVE answers -- every configuration of BASELINE.json -- in round 5: TPB threads of
// NU 16-byte units each (WPE waves per SIMD resident: the register budget), and the loop's loads issued and waited for by hand
// (get_record_async), so that the next-but-one question's rows stay in flight from their request to the pass 1 that takes them.
//   * 256 threads x 2 units: the reductions, the exchange and the barriers are per wave, two units per thread amortise them over
//     twice the elements (same slices, same LDS, two workgroups per CU; 2 waves per SIMD with up to 256 registers);
//   * by the kernel's own clock (s_memtime around the phases, 2000 x 5 x 100000 fp64, 512 x 1) an iteration of 19 900 cycles was:
//     pass 1 of the next question 6550, the wait for the members' records 2700, their sums and the fold 1700, pass 2 -- the only part
//     that is the sweep's arithmetic -- 3800, barrier 950, the column sums 2400, the likelihoods' move to LDS 1800.  The 2700 were not
//     the records: the compiler's s_waitcnt vmcnt(0) in front of the first use of a polled record also waited for the rows requested
//     a moment earlier, and so did one in front of pass 2's first prior (loop-carried "maybe pending" from loads before the loop).
// With the answer count a constant the loops over the answers have no trip-count registers and no on-the-spot loads; other answer
// counts keep eval_cluster_ahead_kernel above (the hand-made waits measured there too, as one kernel with a switch: 3000 x 3 x 60000
// fp64 +8 %, 2000 x 8 x 50000 fp32 +3 % against it).
template <typename R, int TPB, int NU, int WPE, int KC>
__global__ __launch_bounds__(TPB, WPE) void eval_cluster_five_kernel(ClusterArgs a) {
  typedef typename Vec<R>::type V;
  constexpr int VN = Vec<R>::N;
  constexpr int NW = TPB / kWave;
  constexpr int NG = TPB / 32;                                  // groups of 32 lanes: the column sums
  constexpr int kRows = KC;                                     // answer rows requested ahead: all of the question's
  extern __shared__ double smem[];
  const double *tbl = smem;
  V *lhL = reinterpret_cast<V *>(smem + (NumC<R>::kTable ? kLog2TableDoubles : 0));
  double (*red)[kMaxWaves] = reinterpret_cast<double (*)[kMaxWaves]>(lhL + (size_t)a.K * a.sliceUnits);
  double *wTot = &red[kMaxK + 2][0];
  double *xch = wTot + kMaxK + kMaxWaves / 2;                   // [C][K + 2]: the members' partials of one question
  double *wHist = xch + kExchangeDoubles;                       // [4][kMaxK]: W_k of the last four questions (the fold lags two)
  double *wInv = wHist + 4 * kMaxK;                             // [kMaxK]: 1 / W_k of the question in pass 2
  float *park = reinterpret_cast<float *>(wInv + kMaxK);        // fp32: [K][threads] -- the threads' pass-1 sums, added by column (kParkPass1)
  constexpr bool kParkPass1 = !NumC<R>::kTable;                 // (Float engines have the Log2Hot table's 16 KB of LDS to spare)
  if constexpr (NumC<R>::kTable) {
    if (!lds_table_at_zero(tbl)) __builtin_trap();            // log2hot addresses the table absolutely
    for (int i = threadIdx.x; i < kLog2TableDoubles; i += TPB) smem[i] = gLog2TableC[i];
  }
  const int tid = threadIdx.x, lane = tid % kWave, wave = tid / kWave;
  constexpr int64_t K = KC;                                     // (the launcher sends nothing else here)
  const int64_t ldT = a.ldT;
  const int C = a.C, g = blockIdx.x / C, m = blockIdx.x % C;   // cluster, member (= slice)
  const int nUnits = (int)(ldT / VN), SU = a.sliceUnits;
  const int u0 = m * SU;
  const int nMine = u0 >= nUnits ? 0 : (nUnits - u0 < SU ? nUnits - u0 : SU);
  const R *cube = static_cast<const R *>(a.cube);
  const int64_t qStride = (K + 1) * ldT;
  // the thread's units of the slice: tid, tid + TPB, ... (clamped; units beyond the slice are masked), gap bits, masked priors (:103)
  int ui[NU];
  uint32_t gapBits[NU];
  V pr[NU];
#pragma unroll
  for (int j = 0; j < NU; j++) {
    const int s = tid + j * TPB;
    const bool in = s < nMine;
    ui[j] = in ? u0 + s : (nUnits - 1);
    const int64_t t0 = (int64_t)ui[j] * VN;
    gapBits[j] = in ? (a.tgap[t0 >> 5] >> (t0 & 31)) & ((1u << VN) - 1) : (1u << VN) - 1;   // (bits past T are set)
#pragma unroll
    for (int e = 0; e < VN; e++) at<R>(pr[j], e) = ((gapBits[j] >> e) & 1) ? (R)0 : (R)a.prior[t0 + e];
  }
  // (what was loaded up to here is USED here, on every path: otherwise the compiler, which cannot tell whether these loads were waited
  //  for on the way into the loop, puts s_waitcnt vmcnt(0) in front of pass 2's first use of a prior -- and waits there for the rows)
#pragma unroll
  for (int j = 0; j < NU; j++) {
    asm volatile("" : "+v"(gapBits[j]));
#pragma unroll
    for (int e = 0; e < VN; e++) asm volatile("" : "+v"(at<R>(pr[j], e)));
  }
  const bool inSlice0 = tid < SU;                               // (a thread whose first unit lies beyond the slice has none in it)
  const int nSlots = SU < TPB ? SU : TPB;                       // pass 2 parks a thread's sums in the slots of its FIRST unit
  ExRec *recW = a.recW + (size_t)g * 4 * C * kMaxK, *recS = a.recS + (size_t)g * 4 * C * (kMaxK + 2);
  const unsigned long long tagBase = a.tagBase;

  // All C x n records of one exchange (eval_cluster_kernel's gather), in two halves: the loads are ISSUED in front of the request for
  // the next-but-one question's rows and looked at behind it, with a wait that leaves those rows in flight (get_record_async)
  constexpr int kRounds = 2 * kClusterThreads / TPB;            // records per thread: C x n <= 1024
  struct Polled { u4 x[kRounds]; };
  auto record_of = [&](const ExRec *recs, int stride, int n, int u) __attribute__((always_inline)) {
    const int r = tid + u * TPB;
    const int rc = r < C * n ? r : 0;
    const int mm = rc / n, kk = rc - mm * n;
    return recs + (mm * stride + kk);
  };
  auto gather_issue = [&](const ExRec *recs, int stride, int n, Polled &px) __attribute__((always_inline)) {
#pragma unroll
    for (int u = 0; u < kRounds; u++) px.x[u] = get_record_async(record_of(recs, stride, n, u));
  };
  // one look at the thread's polled records: those that carry the tag go to xch; 1 if all of them did
  auto take_records = [&](int total, unsigned long long tag, const Polled &px) __attribute__((always_inline)) {
    int ok = 1;
#pragma unroll
    for (int u = 0; u < kRounds; u++) {
      const int r = tid + u * TPB;
      if (r < total) {
        const unsigned long long t = (unsigned long long)px.x[u][2] | ((unsigned long long)px.x[u][3] << 32);
        if (t == tag) xch[r] = u2d((unsigned long long)px.x[u][0] | ((unsigned long long)px.x[u][1] << 32)); else ok = 0;
      }
    }
    return ok;
  };
  // ... and again until every record is there: each round's loads are issued AND waited for in one statement (get_records_now), so
  // that no register is believed loaded before it is -- and the first look's registers never meet these in a phi
  auto poll_records = [&](const ExRec *recs, int stride, int n, unsigned long long tag) {
    const int total = C * n;
    unsigned spins = 0;
    for (;;) {
      Polled px;
      const ExRec *at[kRounds];
#pragma unroll
      for (int u = 0; u < kRounds; u++) at[u] = record_of(recs, stride, n, u);
      get_records_now(at, px.x);
      if (__all(take_records(total, tag, px))) break;
      __builtin_amdgcn_s_sleep(1);
      if (++spins > (1u << 26)) __builtin_trap();               // (minutes: a member died -- no silent hang)
    }
  };
  auto gather = [&](const ExRec *recs, int stride, int n, unsigned long long tag) {
    poll_records(recs, stride, n, tag);
    __syncthreads();
  };
  auto sum_members = [&](int n, double *out) {
    for (int col = wave; col < n; col += NW) {
      double s = 0.0;
      for (int i = lane; i < C; i += kWave) s += xch[i * n + col];
      s = wave_sum(s);
      if (lane == 0) out[col] = s;
    }
  };
  auto fold = [&](int64_t qq, unsigned long long cnt) {
    gather(recS + (size_t)(cnt & 3) * C * (kMaxK + 2), kMaxK + 2, (int)K + 2, tagBase + cnt + 1);
    sum_members((int)K + 2, red[0]);
    __syncthreads();
    if (tid < K + 2) {
      const double s = red[0][tid];
      double *tot = a.totals + (size_t)qq * (2 * kMaxK + 2);
      if (tid < K) { tot[tid] = wHist[(cnt & 3) * kMaxK + tid]; tot[kMaxK + tid] = s; }
      else tot[2 * kMaxK + (tid - K)] = s;
    }
    __syncthreads();
  };
  auto next_valid = [&](int64_t q) {    // :54
    while (q < a.Q && (bit_test(a.qgap, q) || bit_test(a.asked, q))) {
      if (m == 0 && tid == 0) a.priority[q] = 0.0;
      q += a.nClusters;
    }
    return q;
  };
  u4 dN[NU], rows[kRows][NU];                                  // (raw: the loads are inline assembly, get_record_async's comment)
  auto request_question = [&](int64_t qq) __attribute__((always_inline)) {
    const R *base = cube + qq * qStride;
#pragma unroll
    for (int j = 0; j < NU; j++) dN[j] = load_unit_async(reinterpret_cast<const V *>(base + K * ldT) + ui[j]);
#pragma unroll
    for (int k = 0; k < kRows; k++)
#pragma unroll
      for (int j = 0; j < NU; j++) rows[k][j] = load_unit_async(reinterpret_cast<const V *>(base + k * ldT) + ui[j]);
  };
  // pass 1 of question qq (count cnt), its rows having arrived (rows_arrived, the caller): 1/D and the likelihoods into idOut / lhOut
  // (registers); the partial W_k published.
  auto pass1 = [&](int64_t qq, unsigned long long cnt, V (&idOut)[NU], V (&lhOut)[kRows][NU]) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < NU; j++)
#pragma unroll
      for (int e = 0; e < VN; e++) at<R>(idOut[j], e) = ((gapBits[j] >> e) & 1) ? (R)0 : NumC<R>::inv(at<R>(__builtin_bit_cast(V, dN[j]), e));   // :74
    [[maybe_unused]] unsigned long long candVotes = 0;
    auto row_sum = [&](int k, const V (&row)[NU], V (&lh)[NU]) __attribute__((always_inline)) {
      R sum = (R)0;
#pragma unroll
      for (int j = 0; j < NU; j++)
#pragma unroll
        for (int e = 0; e < VN; e++) {
          at<R>(lh[j], e) = (at<R>(row[j], e) * at<R>(idOut[j], e)) * at<R>(pr[j], e);   // :81-82
          sum += at<R>(lh[j], e);
        }
      if constexpr (kParkPass1) {
        park[k * TPB + tid] = (float)sum;                       // (a butterfly of six DPP steps per answer otherwise: a sixth of the iteration's instructions)
      } else {
        const double sw = wave_sum_d((double)sum);
        if (lane == 0) red[k][wave] = sw;
        if constexpr (NumC<R>::kTable) candVotes |= cluster_cand_vote((double)sum, sw);   // (the pole watch)
      }
    };
#pragma unroll
    for (int k = 0; k < kRows; k++) {
      V row[NU];
#pragma unroll
      for (int j = 0; j < NU; j++) row[j] = __builtin_bit_cast(V, rows[k][j]);
      row_sum(k, row, lhOut[k]);
    }
    if constexpr (NumC<R>::kTable) {
      if (candVotes != 0 && a.candSum != nullptr) {             // (rare: the rows' thread sums again, from the registers)
#pragma unroll
        for (int k = 0; k < kRows; k++) {
          R sk = (R)0;
#pragma unroll
          for (int j = 0; j < NU; j++)
#pragma unroll
            for (int e = 0; e < VN; e++) sk += at<R>(lhOut[k][j], e);
          cluster_cand_publish(a.candSum, qq, k, (double)sk, lane);
        }
      }
    }
    __syncthreads();
    if constexpr (kParkPass1) {
      const int grp = tid >> 5, l32 = tid & 31;                 // groups of 32 lanes, a column (answer) each
      for (int col = grp; col < K; col += NG) {
        const float *src = park + col * TPB + l32;
        double acc = 0.0;
#pragma unroll
        for (int i = 0; i < NG; i++) acc += (double)src[32 * i];
        acc += mov_dpp<kDppXor1>(acc);
        acc += mov_dpp<kDppXor2>(acc);
        acc += mov_dpp<kDppHalfMirror>(acc);
        acc += mov_dpp<kDppMirror>(acc);
        const Pair pq = swap16(acc);
        acc = pq.a + pq.b;
        if (l32 == 0) put_record(recW + ((size_t)(cnt & 3) * C + m) * kMaxK + col, acc, tagBase + cnt + 1);
      }
    } else {
      if (tid < K) {
        double w = 0.0;
        for (int i = 0; i < NW; i++) w += red[tid][i];
        put_record(recW + ((size_t)(cnt & 3) * C + m) * kMaxK + tid, w, tagBase + cnt + 1);
      }
    }
  };
  // the likelihoods of question qq into LDS (pass 2 of the question before it is done with the copy: the caller's barrier)
  auto store_question = [&](int64_t qq, const V (&idOf)[NU], const V (&lhOf)[kRows][NU]) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < NU; j++) {
      const int s = tid + j * TPB;
      if (s < SU) {
#pragma unroll
        for (int k = 0; k < kRows; k++)
          lhL[k * SU + s] = lhOf[k][j];
      }
    }
  };

  if constexpr (NumC<R>::kTable) __syncthreads();
  unsigned long long round = 0;                                 // count of the question in pass 2 (the cluster's questions so far)
  int64_t q = next_valid(g), qPrev = -1, qPrev2 = -1;
  V id[NU], idNext[NU], lhNext[kRows][NU];
  if (q < a.Q) {
    request_question(q);
    rows_arrived(dN, rows);
    pass1(q, 0, idNext, lhNext);                                // (the first question: nothing to hide behind)
    store_question(q, idNext, lhNext);
#pragma unroll
    for (int j = 0; j < NU; j++) id[j] = idNext[j];
  }
  int64_t qNext = q < a.Q ? next_valid(q + a.nClusters) : a.Q;
  if (qNext < a.Q) request_question(qNext);
  while (q < a.Q) {
    // ---- pass 1 of the NEXT question; then its successor's rows are requested into the registers that are free again
    const int64_t qNext2 = qNext < a.Q ? next_valid(qNext + a.nClusters) : a.Q;
    Polled pw;
    rows_arrived(dN, rows);                                     // (whether or not there is a next question: no path carries a request past here)
    if (qNext < a.Q) pass1(qNext, round + 1, idNext, lhNext);
    // ---- everybody's partial W of THIS question (published an iteration ago), in slice order: the records are asked for BEFORE the
    // next-but-one question's rows are requested (into the registers pass 1 has just freed), so that their wait leaves those in flight.
    // (Asked for already in front of pass 1's arithmetic, 16 registers held across it: 2921 against 2910 us -- no gain, not kept.)
    // (Each answer's group of 32 lanes reading ITS records itself, no staging in LDS and a barrier less: the reads are 256 bytes apart
    //  instead of one coalesced round -- fp64 2670 -> 2757 us, fp32 1110 -> 1170-1220; not kept.)
    // (the request, its wait and the look at the records stand TOGETHER in one branch: the wait's count is an immediate, and a wait
    //  behind a merge would be a second test of the same condition -- tools/vmem_hazards.py follows the built code's paths without
    //  knowing that two branches test the same thing.  Without a request behind them the records are asked for and waited for in
    //  one statement: a tied wait there had the compiler copy the registers to where the other branch leaves them BEFORE the wait.)
    int seen;
    if (qNext2 < a.Q) {
      gather_issue(recW + (size_t)(round & 3) * C * kMaxK, kMaxK, (int)K, pw);
      request_question(qNext2);
      wait_records<(kRows + 1) * NU>(pw.x);                     // (the rows just requested stay in flight)
      seen = take_records(C * (int)K, tagBase + round + 1, pw);
    } else {
      const ExRec *at[kRounds];
#pragma unroll
      for (int u = 0; u < kRounds; u++) at[u] = record_of(recW + (size_t)(round & 3) * C * kMaxK, kMaxK, (int)K, u);
      get_records_now(at, pw.x);
      seen = take_records(C * (int)K, tagBase + round + 1, pw);
    }
    if (!__all(seen)) poll_records(recW + (size_t)(round & 3) * C * kMaxK, kMaxK, (int)K, tagBase + round + 1);
    __syncthreads();
    // W_k over the members, a column per group of 32 lanes (the five in one round), written straight to where pass 2 and the fold look;
    // 1 / W_k (:91) once per workgroup (every thread formed it: a fifth of pass 2's instructions)
    {
      const int grp = tid >> 5, l32 = tid & 31;
      for (int col = grp; col < K; col += NG) {
        double w = 0.0;
        for (int i = l32; i < C; i += 32) w += xch[i * (int)K + col];
        w += mov_dpp<kDppXor1>(w);
        w += mov_dpp<kDppXor2>(w);
        w += mov_dpp<kDppHalfMirror>(w);
        w += mov_dpp<kDppMirror>(w);
        const Pair pq = swap16(w);
        w = pq.a + pq.b;
        if (l32 == 0) {
          wInv[col] = div_fast(1.0, w);
          wHist[(round & 3) * kMaxK + col] = w;
        }
      }
    }
    __syncthreads();                                            // (xch is read, W is there)
    // the turn-taker folds the question before the previous one: every member published its sums before it published its W of this
    // question, which have all just been seen
    if (qPrev2 >= 0 && (int)((round - 2) % (unsigned long long)C) == m) fold(qPrev2, round - 2);
    // ---- pass 2 (:95-128) from LDS, answer by answer.  The threads' sums are not reduced wave by wave (K + 2 butterflies of six DPP
    // steps each were a quarter of the iteration's instructions): a thread parks its K + 2 sums in the LDS slots of its first unit --
    // 16 bytes per answer row, dead once the row's likelihoods have been read -- and groups of 32 lanes add one column each.
    double *slot = reinterpret_cast<double *>(lhL);             // unit u of row k: slot[(k * SU + u) * 2 + {0, 1}]
    V accN[NU], accD[NU];
    R hW = (R)0, accL = (R)0;
#pragma unroll                                                 // (the five answers in one stretch: -6 % against a loop)
    for (int k = 0; k < (int)K; k++) {
      const R invWk = (R)wInv[k];                               // :91 (formed once per workgroup, above)
      R vk = (R)0;
      V lh[NU];
#pragma unroll
      for (int j = 0; j < NU; j++) {
        const int s = tid + j * TPB;
        lh[j] = lhL[k * SU + (s < SU ? s : 0)];
      }
#pragma unroll
      for (int j = 0; j < NU; j++) {
        if (tid + j * TPB < SU) {
#pragma unroll
          for (int e = 0; e < VN; e++) {
            const R l = at<R>(lh[j], e), pi = at<R>(pr[j], e);
            const R p = l * invWk;                              // :97
            const R l2 = NumC<R>::log2p(p, tbl);                // :106
            hW = fma(l, l2, hW);                                // :113-114 weighted by W_k (eval_epilogue)
            const R dd = p - pi;                                // :119
            vk = fma(dd, dd, vk);                               // :126-127
            if (k == 0) { at<R>(accN[j], e) = (R)1; at<R>(accD[j], e) = l2; }   // :117 sum_k 1 / log2 p_k = N / D, answer by answer
            else { at<R>(accN[j], e) = fma(at<R>(accN[j], e), l2, at<R>(accD[j], e)); at<R>(accD[j], e) = at<R>(accD[j], e) * l2; }
          }
        }
      }
      if (inSlice0) slot[(k * SU + tid) * 2] = (double)vk;
    }
#pragma unroll
    for (int j = 0; j < NU; j++) {
      if (tid + j * TPB < SU) {
#pragma unroll
        for (int e = 0; e < VN; e++) {
          const R i1 = at<R>(id[j], e);
          accL = fma((i1 * at<R>(accN[j], e)) * i1, NumC<R>::rcp(at<R>(accD[j], e)), accL);
        }
      }
    }
    if (inSlice0) {
      slot[tid * 2 + 1] = (double)hW;                           // (row 0's second double)
      if (K > 1) slot[(SU + tid) * 2 + 1] = (double)accL;       // (row 1's)
    }
    if (K == 1) {                                               // (one answer: one row of slots -- the lack sum the old way)
      const double s2 = wave_sum_d((double)accL);
      if (lane == 0) red[0][wave] = s2;
    }
    __syncthreads();                                            // (the slots are complete)
    {
      const int grp = tid >> 5, l32 = tid & 31;                 // column c of the K + 2: V_k (k < K) | sum l log2 p | lack
      for (int col = grp; col < K + 2; col += NG) {
        const double *src = col < K ? slot + (size_t)col * SU * 2 : col == K ? slot + 1 : slot + (size_t)SU * 2 + 1;
        double acc = 0.0;
        if (col == K + 1 && K == 1) {
          if (l32 < NW) acc = red[0][l32];
        } else {
          double acc1 = 0.0, acc2 = 0.0, acc3 = 0.0;             // (four chains: nSlots / 32 dependent additions were the phase)
          int u = l32;
          for (; u + 96 < nSlots; u += 128) { acc += src[(size_t)u * 2]; acc1 += src[(size_t)(u + 32) * 2]; acc2 += src[(size_t)(u + 64) * 2]; acc3 += src[(size_t)(u + 96) * 2]; }
          for (; u < nSlots; u += 32) acc += src[(size_t)u * 2];
          acc = (acc + acc1) + (acc2 + acc3);
        }
        acc += mov_dpp<kDppXor1>(acc);
        acc += mov_dpp<kDppXor2>(acc);
        acc += mov_dpp<kDppHalfMirror>(acc);
        acc += mov_dpp<kDppMirror>(acc);
        const Pair pq = swap16(acc);
        acc = pq.a + pq.b;
        if (l32 == 0) put_record(recS + ((size_t)(round & 3) * C + m) * (kMaxK + 2) + col, acc, tagBase + round + 1);
      }
    }
    __syncthreads();                                            // (the slots are read: the LDS copy is free)
    // ---- the next question's likelihoods move from the registers to LDS
    if (qNext < a.Q) {
      store_question(qNext, idNext, lhNext);
#pragma unroll
      for (int j = 0; j < NU; j++) id[j] = idNext[j];
    }
    qPrev2 = qPrev;
    qPrev = q;
    round++;
    q = qNext;
    qNext = qNext2;
  }
  rows_arrived(dN, rows);                                       // (nothing is in flight here -- said where the checker of the built code sees it)
  __syncthreads();
  // the last two questions' sums: their turn-takers wait for them
  if (qPrev2 >= 0 && (int)((round - 2) % (unsigned long long)C) == m) fold(qPrev2, round - 2);
  if (qPrev >= 0 && (int)((round - 1) % (unsigned long long)C) == m) fold(qPrev, round - 1);
}

// The pole watch's verdict, behind the sweep (a thread per question and answer row; nearly all of them read one zero word and leave):
// the largest thread sum a wave reported for the row against W_k -- the bars of the sweeps in eval_kernels.hip (kNearOneShare,
// kQuarterShare there; the shares are of two elements' sum here, of a lane's there: bounds for the largest element both).
__global__ __launch_bounds__(256) void cluster_watch_kernel(const double *__restrict__ totals, unsigned long long *__restrict__ candSum,
                                                            uint32_t *__restrict__ poleMask, PoleHeader *list, int64_t K, int64_t Q) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  const int64_t q = i / kMaxK, k = i % kMaxK;
  if (q >= Q || k >= K) return;
  const unsigned long long bits = candSum[i];
  if (bits == 0ull) return;
  candSum[i] = 0ull;                                            // (the next launch finds it cleared)
  const double *tot = totals + (size_t)q * (2 * kMaxK + 2);
  const double top = __longlong_as_double((long long)bits), w = tot[k];
  const bool list1 = top >= w * (1.0 - 0x1p-9) || (top > w * 0.2499 && tot[kMaxK + k] <= kSmallV);
  if (list1 && atomicOr(&poleMask[q], 1u << k) == 0u) pole_list_append(list, (uint32_t)q, 0u, 0u);
}

// :134-207, one thread per question
__global__ __launch_bounds__(256) void cluster_epilogue_kernel(const double *__restrict__ totals, const uint32_t *__restrict__ qgap,
                                                               const uint32_t *__restrict__ asked, double *__restrict__ priority,
                                                               int64_t K, int64_t Q, double vCompTail) {
  const int64_t q = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (q >= Q || bit_test(qgap, q) || bit_test(asked, q)) return;
  const double *tot = totals + (size_t)q * (2 * kMaxK + 2);
  double mW[kMaxK], mWV[kMaxK];
  for (int64_t k = 0; k < K; k++) { mW[k] = tot[k]; mWV[k] = tot[k] * sqrt(tot[kMaxK + k]); }   // :156-157
  priority[q] = eval_epilogue(mW, -tot[2 * kMaxK], mWV, K, tot[2 * kMaxK + 1], vCompTail);
}

constexpr size_t kFixedLdsBytes = ((size_t)(kMaxK + 2) * (kClusterThreads / kWave) + kMaxK + kClusterThreads / kWave / 2) * sizeof(double);   // red[][] + wTot[] + votes[]
constexpr size_t kExchangeLdsBytes = kExchangeDoubles * sizeof(double);
constexpr size_t kAheadLdsBytes = 5 * kMaxK * sizeof(double);   // eval_cluster_ahead_kernel: wHist, wInv (+ fp32: K x threads floats, park)
struct ClusterShape { int C, nClusters, sliceUnits, nu, tpb, perCU; size_t shmem; bool ahead; };

// The shapes of the form that runs ahead (engine option cluster_shape; KbView::clusterShape): threads x units per thread, and how
// many workgroups share a CU.  Reductions, exchanges and barriers are per WAVE: two units per thread amortise them over twice the
// elements (at half the waves per SIMD and twice the registers per thread).
struct AheadVariant { int tpb, nu, perCU; };
constexpr AheadVariant kAheadVariants[] = {
    {512, 1, 2},   // 1: round 4's kernel (eval_cluster_ahead_kernel) -- four waves per SIMD, 128 registers, any number of answers
    {256, 2, 2},   // 2: eval_cluster_five_kernel -- the same slices and LDS, half the waves, two units per thread; five answers
    // (measured and taken out in round 5: one workgroup of 512 x 2 or 512 x 3 per CU, slices two and three times as long, half and a
    //  third of the members -- 2000 x 5 x 100000: fp64 3.74 / 3.93 ms against 3.48, fp32 1.67 / 1.61 against 1.31)
};
constexpr int kAheadVariantCount = (int)(sizeof(kAheadVariants) / sizeof(kAheadVariants[0]));
constexpr int kAheadDefaultF64 = 2, kAheadDefaultF32 = 2;     // (where the shape is not built -- other than five answers -- shape 1)

constexpr int kFiveMinK = 2, kFiveMaxK = 5;                   // answer counts eval_cluster_five_kernel is built for (six: 254 registers, eight: spills)
template <typename R>
const void *ahead_kernel_of(int variant, int64_t K) {
  if (variant == 2) {                                           // (two to five answers only: cluster_shape_of)
    switch (K) {
      case 2: return reinterpret_cast<const void *>(eval_cluster_five_kernel<R, 256, 2, 2, 2>);
      case 3: return reinterpret_cast<const void *>(eval_cluster_five_kernel<R, 256, 2, 2, 3>);
      case 4: return reinterpret_cast<const void *>(eval_cluster_five_kernel<R, 256, 2, 2, 4>);
      default: return reinterpret_cast<const void *>(eval_cluster_five_kernel<R, 256, 2, 2, 5>);
    }
  }
  return reinterpret_cast<const void *>(eval_cluster_ahead_kernel<R>);
}

// does the device hold `perCU` workgroups of the kernel per CU with this much LDS?  (cached per kernel, device and LDS size)
bool occupancy_reaches(LaunchCache &cache, const void *kern, int threads, size_t shmem, int perCU) {
  const int dev = LaunchCache::Device();
  int got = 0;
  if (cache.Get(dev, shmem, &got)) return got >= perCU;
  hipError_t e = hipSuccess;
  if (shmem > 64 * 1024) e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
  if (e == hipSuccess) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&got, kern, threads, shmem);
  if (e != hipSuccess) { got = 0; (void)hipGetLastError(); }   // (not this launch's error: the caller falls back to another form)
  cache.Put(dev, shmem, got);
  return got >= perCU;
}
template <typename R, int NU>
bool occupancy_two(size_t shmem) {
  static LaunchCache cache;   // (per instantiation and device)
  return occupancy_reaches(cache, reinterpret_cast<const void *>(eval_cluster_kernel<R, NU>), kClusterThreads, shmem, 2);
}
template <typename R>
bool occupancy_ahead(int variant, int64_t K, size_t shmem) {
  static LaunchCache cache[kAheadVariantCount + 1][kFiveMaxK + 1];   // (per kernel: the 256 x 2 shape is one per answer count)
  const AheadVariant &v = kAheadVariants[variant - 1];
  return occupancy_reaches(cache[variant][variant == 2 ? (int)K : 0], ahead_kernel_of<R>(variant, K), v.tpb, shmem, v.perCU);
}

// Slices as long as the workgroups' LDS allows (72 KB each where two share a CU, the fp64 table included): the fewer members a
// cluster has, the fewer partials every member adds per question.
// variant: 0 = the question-by-question form (eval_cluster_kernel), 1.. = kAheadVariants[variant - 1] of the form that runs ahead.
template <typename R>
bool cluster_shape_of(const KbView &kb, int nCU, int variant, ClusterShape *out) {
  constexpr int VN = Vec<R>::N;
  const bool ahead = variant > 0;
  if (kb.K > kMaxK || kb.K < 1 || variant > kAheadVariantCount) return false;
  if (variant == 2 && (kb.K < kFiveMinK || kb.K > kFiveMaxK)) return false;   // (256 x 2 is built for questions of two to five answers)
  const AheadVariant v = ahead ? kAheadVariants[variant - 1] : AheadVariant{kClusterThreads, NumC<R>::kTable ? 1 : 2, 2};
  const int64_t nUnits = kb.ldT / VN;
  const size_t ldsPerWg = v.perCU == 2 ? 72 * 1024 : 144 * 1024;
  const size_t tableBytes = NumC<R>::kTable ? kLog2TableDoubles * sizeof(double) : 0;
  const size_t parkBytes = ahead && !NumC<R>::kTable ? (size_t)kb.K * kClusterThreads * sizeof(float) : 0;
  if (ldsPerWg < tableBytes + kFixedLdsBytes + kExchangeLdsBytes + kAheadLdsBytes + parkBytes + 64 * 16 * (size_t)kb.K) return false;
  const size_t budget = ldsPerWg - tableBytes - kFixedLdsBytes - kExchangeLdsBytes - (ahead ? kAheadLdsBytes + parkBytes : 0);
  int64_t maxUnits = (int64_t)(budget / ((size_t)kb.K * 16));
  // question by question: fp32 up to two units per thread (fewer members per cluster: 1797 vs 2018 us at 2000 x 5 x 100000), fp64 one --
  // with two the next question's rows do not fit the 128 registers beside pass 2 and spill (6160 vs 4459 us)
  maxUnits = std::min<int64_t>(maxUnits, (int64_t)v.nu * v.tpb);
  maxUnits = maxUnits / kWave * kWave;
  if (maxUnits < kWave) return false;
  const int64_t C = (nUnits + maxUnits - 1) / maxUnits;
  const int capacity = v.perCU * nCU;
  // A cluster's members wait for each other, so they must become resident together.  Workgroups of a launch are dispatched in
  // order: at any time a launch has at most ONE incomplete cluster on the device (its frontier), every other resident cluster is
  // complete and finishes its questions whatever else happens -- so even several such launches in flight at once (shards of one
  // engine on one device, two processes on one GPU) keep making progress as long as their frontiers together do not fill the
  // device.  Clusters of at most a quarter of the device: three launches at once can never.
  if (C > capacity / 4 || (size_t)C * (kb.K + 2) * sizeof(double) > kExchangeLdsBytes) return false;
  int64_t su = ((nUnits + C - 1) / C + kWave - 1) / kWave * kWave;        // whole waves of units
  out->C = (int)C;
  out->nClusters = (int)std::max<int64_t>(1, std::min<int64_t>(capacity / C, kb.Q));
  out->sliceUnits = (int)su;
  out->tpb = v.tpb;
  out->perCU = v.perCU;
  out->nu = ahead ? v.nu : (su <= kClusterThreads ? 1 : 2);
  out->ahead = ahead;
  out->shmem = tableBytes + (size_t)kb.K * su * 16 + kFixedLdsBytes + kExchangeLdsBytes + (ahead ? kAheadLdsBytes + parkBytes : 0);
  if (ahead) return occupancy_ahead<R>(variant, kb.K, out->shmem);
  return out->nu == 1 ? occupancy_two<R, 1>(out->shmem) : occupancy_two<R, 2>(out->shmem);
}
// KbView::clusterForm (engine option cluster_form): 0 = the default below, 1 = the question-by-question form, 2 = pass 1 a question
// ahead; KbView::clusterShape (option cluster_shape): which of kAheadVariants the form that runs ahead takes, 0 = the default
template <typename R>
int cluster_variant(const KbView &kb, int nCU, ClusterShape *out) {   // -1: not supported; 0: question by question; 1..: kAheadVariants
  const bool ahead = kb.clusterForm == 0 ? kClusterAheadByDefault : kb.clusterForm == 2;
  if (ahead) {
    const int want = kb.clusterShape > 0 ? kb.clusterShape : (NumC<R>::kTable ? kAheadDefaultF64 : kAheadDefaultF32);
    if (cluster_shape_of<R>(kb, nCU, want, out)) return want;
    if (want != 1 && cluster_shape_of<R>(kb, nCU, 1, out)) return 1;
  }
  return cluster_shape_of<R>(kb, nCU, 0, out) ? 0 : -1;
}
template <typename R>
bool cluster_shape(const KbView &kb, int nCU, ClusterShape *out) { return cluster_variant<R>(kb, nCU, out) >= 0; }

int device_cus() {
  int dev = 0, nCU = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&nCU, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || nCU <= 0) nCU = 256;
  return nCU;
}

}  // namespace

// Rows longer than the register shapes take (ldT > 16384), up to 16 answers, two workgroups per CU resident.
bool EvalClusterSupported(const KbView &kb) {
  if (kb.ldT <= kb.clusterFrom) return false;
  ClusterShape s;
  return kb.elem == 4 ? cluster_shape<float>(kb, device_cus(), &s) : cluster_shape<double>(kb, device_cus(), &s);
}

const char *EvalClusterKernelName(const KbView &kb) {
  static thread_local char name[48];
  ClusterShape s{};
  const bool ok = kb.elem == 4 ? cluster_shape<float>(kb, device_cus(), &s) : cluster_shape<double>(kb, device_cus(), &s);
  if (!ok) return "stream";
  char shape[16] = "";
  if (s.ahead && !(s.tpb == 512 && s.nu == 1)) std::snprintf(shape, sizeof(shape), "_%dx%d", s.tpb, s.nu);
  std::snprintf(name, sizeof(name), "%s_cluster%d_x%d%s%s", kb.elem == 4 ? "f32" : "f64", s.C, s.nClusters, s.ahead ? "_ahead" : "", shape);
  return name;
}

// bytes of exchange scratch the launch needs (records + totals; cleared once by the caller); 0 if the shape is not supported
size_t EvalClusterScratchBytes(const KbView &kb) {
  ClusterShape s{};
  const bool ok = kb.elem == 4 ? cluster_shape<float>(kb, device_cus(), &s) : cluster_shape<double>(kb, device_cus(), &s);
  if (!ok) return 0;
  const size_t perCluster = (size_t)4 * s.C * (2 * kMaxK + 2) * sizeof(ExRec);   // (four record slots: the form that runs ahead; the other uses two)
  return (size_t)s.nClusters * perCluster + (size_t)kb.Q * (2 * kMaxK + 2) * sizeof(double) + 256 + ((size_t)kb.Q + 2) * sizeof(uint32_t) + (size_t)kb.Q * kMaxK * sizeof(unsigned long long);
}

hipError_t LaunchEvalCluster(const KbView &kb, const double *prior, const uint32_t *asked, double *priority, void *scratch, hipStream_t stream) {
  ClusterShape s{};
  const bool f32 = kb.elem == 4;
  const int variant = f32 ? cluster_variant<float>(kb, device_cus(), &s) : cluster_variant<double>(kb, device_cus(), &s);
  if (variant < 0 || scratch == nullptr) return hipErrorInvalidValue;
  char *p = static_cast<char *>(scratch);
  ClusterArgs a{};
  a.cube = kb.cube; a.prior = prior; a.tgap = kb.tgap; a.qgap = kb.qgap; a.asked = asked;
  a.K = kb.K; a.Q = kb.Q; a.ldT = kb.ldT; a.C = s.C; a.nClusters = s.nClusters; a.sliceUnits = s.sliceUnits;
  a.recW = reinterpret_cast<ExRec *>(p);
  a.recS = a.recW + (size_t)s.nClusters * 4 * s.C * kMaxK;
  a.totals = reinterpret_cast<double *>(a.recS + (size_t)s.nClusters * 4 * s.C * (kMaxK + 2));
  a.priority = priority;
  // (the rows' mask words: behind the totals; the caller cleared the scratch once, the fix leaves them cleared)
  const bool watch = !f32 && kb.poleList != nullptr && kb.poleScratch != nullptr;
  a.poleList = watch ? kb.poleList : nullptr;
  a.poleMask = watch ? reinterpret_cast<uint32_t *>(reinterpret_cast<char *>(a.totals) + (size_t)kb.Q * (2 * kMaxK + 2) * sizeof(double) + 256) : nullptr;
  a.candSum = watch ? reinterpret_cast<unsigned long long *>(a.poleMask + ((kb.Q + 1) & ~(int64_t)1)) : nullptr;
  static std::atomic<unsigned long long> launches{0};
  a.tagBase = (launches.fetch_add(1) + 1) << 32;
  hipError_t e = hipSuccess;
  const dim3 grid((unsigned)(s.C * s.nClusters));
  if (variant > 0) {
    void *params[] = {&a};
    e = hipLaunchKernel(f32 ? ahead_kernel_of<float>(variant, kb.K) : ahead_kernel_of<double>(variant, kb.K), grid, dim3((unsigned)s.tpb), params, s.shmem, stream);
    if (e != hipSuccess) return e;
  } else if (f32) {
    if (s.nu == 1) hipLaunchKernelGGL((eval_cluster_kernel<float, 1>), grid, dim3(kClusterThreads), s.shmem, stream, a);
    else hipLaunchKernelGGL((eval_cluster_kernel<float, 2>), grid, dim3(kClusterThreads), s.shmem, stream, a);
  } else {
    if (s.nu == 1) hipLaunchKernelGGL((eval_cluster_kernel<double, 1>), grid, dim3(kClusterThreads), s.shmem, stream, a);
    else hipLaunchKernelGGL((eval_cluster_kernel<double, 2>), grid, dim3(kClusterThreads), s.shmem, stream, a);
  }
  e = hipGetLastError();
  if (e != hipSuccess) return e;
  if (watch) {
    hipLaunchKernelGGL(cluster_watch_kernel, dim3((unsigned)((kb.Q * kMaxK + 255) / 256)), dim3(256), 0, stream, a.totals, a.candSum, a.poleMask,
                       a.poleList, kb.K, kb.Q);
    // the questions on the list, redone in the reference's order where their totals lie (W_k | V_k | sum l log2 p | lack); the
    // epilogues below then see the corrected totals
    PoleFix f{};
    f.cube = static_cast<const double *>(kb.cube); f.tgap = kb.tgap; f.qgap = kb.qgap; f.asked = asked; f.prior = prior;
    f.list = a.poleList; f.maskDense = a.poleMask; f.sums = a.totals; f.sumsStride = 2 * kMaxK + 2;
    f.wOff = 0; f.vOff = kMaxK; f.hOff = 2 * kMaxK; f.lOff = 2 * kMaxK + 1; f.secondIsWV = 0;
    f.K = kb.K; f.T = kb.T; f.ldT = kb.ldT; f.qFirst = 0; f.nQ = kb.Q; f.capacity = kb.Q;
    e = LaunchPoleFixup(f, stream);
    if (e != hipSuccess) return e;
  }
  const double nT = (double)(kb.nValidTargets + 1);             // PqaCore/CEEvalQsSubtaskConsider.cpp:191
  hipLaunchKernelGGL(cluster_epilogue_kernel, dim3((unsigned)((kb.Q + 255) / 256)), dim3(256), 0, stream, a.totals, kb.qgap, asked,
                     priority, kb.K, kb.Q, 0.34657359027997265470861606072909 / (nT * nT));
  return hipGetLastError();
}

}  // namespace pqa
