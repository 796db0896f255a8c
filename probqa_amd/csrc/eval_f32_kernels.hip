// eval_f32_kernels.hip -- the single-quiz priority sweep of Float engines with the question's rows held in registers, gfx950.
//
// Reference: PqaCore/CEEvalQsSubtaskConsider.cpp:41-217 (one question at a time: pass 1 W_k = sum_t (A/D) prior, pass 2 the
// posterior, its log2, the entropy / lack / velocity sums), on an fp32 cube.  The arithmetic of one element is that of the fp32
// batched sweep (batch_kernels.hip: v_log_f32 clamped to the Float analogue of Log2Hot's range, v_rcp_f32 for the lack term, one
// reciprocal per pair of targets); the epilogue is the shared fp64 one (eval_device.h).
//
// What the plain streaming form (batch_kernels.hip: eval_questions_f32_stream) does per question is read every answer row twice
// (pass 2 again, from L2) and the mD row once per pass and answer, with four barriers and a handful of wave reductions between
// any two rows: 2.1-2.3 TB/s of cube at 10000 x 5 x 10000, a quarter of the HBM rate, and SLOWER than a Double engine on twice the
// bytes.  Here, as in the fp64 register shapes (eval_kernels.hip):
//   * a thread owns NQ quads of targets (16 bytes each: quad i = tid + j NT of every row) for the whole launch: the masked prior
//     of its targets is loaded ONCE per workgroup, 1/D once per question, an answer row once -- pass 2 runs on the registers pass
//     1 ran on;
//   * the next row of the stream (next answer, or the next question's mD) is requested before pass 1 of the current one, so a
//     full row per workgroup is in flight while the current one is computed on;
//   * per row one exchange of the waves' W partials (one barrier); the velocity sums are reduced per answer, the entropy / lack
//     sums stay lane-local over the question; finished questions queue up and wave 0 runs their fp64 epilogues 32 at a time, one
//     per lane (one lane running one epilogue while the workgroup waits cost a fifth of a question at 10000 targets).
// HBM traffic = the cube once; 4.2-4.8 TB/s of cube for rows of 4000..16000 targets (f32_shape below), 580 us at
// 10000 x 5 x 10000 against 1046 us for the streaming form and 911 us for a Double engine.
#include <algorithm>
#include <cstdio>
#include <cstdlib>

#include "eval_device.h"
#include "pqa_device.h"
#include "pqa_kernels.h"

namespace pqa {

namespace {

__device__ __forceinline__ float rcp_nr_f32(float x) {       // 2^-22.5 -> full fp32 precision
  const float r = __builtin_amdgcn_rcpf(x);
  return fmaf(r, fmaf(-x, r, 1.0f), r);
}
__device__ __forceinline__ float log2p_f32(float p) {        // (batch_kernels.hip: Num<float>::log2p)
  return __builtin_amdgcn_fmed3f(__builtin_amdgcn_logf(p), -127.0f, -4.2992253e-08f);
}

// The fp64 epilogue (exp2, log, four divisions) as a real call: inlined, its ~90 registers come on top of the rows a thread holds
// (the 1024-thread shape spilled); called, it saves what it clobbers to the stack -- once per question, one lane of one wave.
__device__ __attribute__((noinline)) double epilogue_call(const double *rec, double whSum, int64_t K, double lackSum, double vCompTail) {
  return eval_epilogue(rec, whSum, rec + K, K, lackSum, vCompTail);
}

// LDS-DMA (eval_kernels.hip: dma16): 16 bytes per lane from global memory straight into LDS, no destination registers; M0 = the
// wave-uniform LDS byte address, lane i lands at M0 + 16 i; counted by vmcnt, invisible to the compiler's bookkeeping.
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
// one word of a bitmap by a SCALAR load (a vector load here would be counted with the DMA loads in flight, and the compiler's wait
// for it would wait for them all): the word's address is wave-uniform
__device__ __forceinline__ uint32_t scalar_word(const uint32_t *p) {
  uint32_t v;
  const uint64_t addr = (uint64_t)(uintptr_t)p;
  const uint64_t au = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(addr >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)addr);
  asm volatile("s_load_dword %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(au) : "memory");
  return v;
}

struct F32Args {
  const float *cube;          // [Q][K+1][ldT]
  const double *prior;        // the quiz's posterior, fp64
  const uint32_t *tgap, *qgap, *asked;
  double *priority;
  int64_t K, Q, ldT;
  double vCompTail;
};

constexpr int kMaxK = 16;     // answers whose per-row sums fit the LDS layout below (more: the streaming form)
constexpr int kPend = 32;                   // finished questions queued for their epilogues
constexpr int kPendLen = 2 * kMaxK + 3;     // W_k [K] | W_k sqrt(V_k) [K] | sum W_k H_k | lack | question index

// NQ quads per thread, blockDim.x = NT threads (a multiple of 64): NT x NQ is fitted to the row (f32_shape) -- a thread without a
// quad of the row re-reads the last one and masks it, i.e. issues loads for nothing.
// The rows a workgroup reads form ONE stream -- per question its mD row, then its answer rows -- and the kernel is a loop over that
// stream with the next row requested ahead, across answer and question boundaries alike (two rows ahead: measured, no faster).
template <int NQ>
__global__ __launch_bounds__(1024) void eval_questions_f32_reg(F32Args a) {
  constexpr int kMaxWaves = 1024 / kWave;
  const int NT = (int)blockDim.x, NW = NT / kWave;
  // LDS: W exchange [2][NW] floats | per-question partials [kMaxK + 2][NW] floats | W_k [kMaxK] doubles | the queue of finished questions
  __shared__ float wx[2][kMaxWaves];
  __shared__ float part[kMaxK + 2][kMaxWaves];
  __shared__ double rec[kMaxK];
  __shared__ double pend[kPend * kPendLen];
  const int tid = threadIdx.x, lane = tid % kWave, wave = tid / kWave;
  const int64_t K = a.K, ldT = a.ldT;
  const int nQuads = (int)(ldT >> 2);
  const int64_t qStride = (K + 1) * ldT;
  // ---- the thread's targets: quad index (clamped: out-of-row quads re-read the last one and are masked), gap bits, masked prior
  int qi[NQ];
  uint32_t gapBits[NQ];
  float4 pr[NQ];
#pragma unroll
  for (int j = 0; j < NQ; j++) {
    const int i = tid + j * NT;
    const bool in = i < nQuads;
    qi[j] = in ? i : nQuads - 1;
    const uint32_t g = in ? (a.tgap[qi[j] >> 3] >> ((4 * qi[j]) & 31)) & 15u : 15u;   // (bits past T are set: padding columns)
    gapBits[j] = g;
    const double2 p0 = reinterpret_cast<const double2 *>(a.prior)[2 * qi[j]], p1 = reinterpret_cast<const double2 *>(a.prior)[2 * qi[j] + 1];
    pr[j] = make_float4((g & 1) ? 0.f : (float)p0.x, (g & 2) ? 0.f : (float)p0.y, (g & 4) ? 0.f : (float)p1.x, (g & 8) ? 0.f : (float)p1.y);   // :103
  }
  auto next_valid = [&](int64_t q) {    // :54 gap / asked questions get priority 0 and leave the stream
    while (q < a.Q && (bit_test(a.qgap, q) || bit_test(a.asked, q))) {
      if (tid == 0) a.priority[q] = 0.0;
      q += gridDim.x;
    }
    return q;
  };
  struct Pos { int64_t q; int64_t r; };   // row r of question q: r = K the mD row (first of the question), then r = 0 .. K - 1
  auto advance = [&](Pos p) {
    if (p.r == K) return Pos{p.q, 0};
    if (p.r + 1 < K) return Pos{p.q, p.r + 1};
    return Pos{next_valid(p.q + gridDim.x), K};
  };
  auto load_row = [&](Pos p, float4 (&dst)[NQ]) __attribute__((always_inline)) {
    if (p.q >= a.Q) return;             // (past the end of the stream)
    const float4 *row = reinterpret_cast<const float4 *>(a.cube + p.q * qStride + p.r * ldT);
#pragma unroll
    for (int j = 0; j < NQ; j++) {      // (non-temporal: the cube streams -- pqa_device.h, row_load)
      typedef unsigned int u4n __attribute__((ext_vector_type(4)));
      dst[j] = __builtin_bit_cast(float4, __builtin_nontemporal_load(reinterpret_cast<const u4n *>(row + qi[j])));
    }
  };
  int nPend = 0;
  auto flush = [&](int n) {           // wave 0: one queued question per lane (:130-207)
    if (lane < n) {
      const double *pq = pend + (size_t)lane * kPendLen;
      const int64_t qq = reinterpret_cast<const int64_t *>(pq)[2 * K + 2];
      a.priority[qq] = epilogue_call(pq, -pq[2 * K], K, pq[2 * K + 1], a.vCompTail);
    }
  };
  Pos pc{next_valid(blockIdx.x), K}, pa = pc;   // the row being computed on, and the row in ahead[]
  float4 cur[NQ], ahead[NQ], id[NQ];
  load_row(pa, ahead);
  int par = 0;
  float hW = 0.f, accL = 0.f;
  while (pc.q < a.Q) {
    // ---- the stream moves on: the requested row becomes the current one, the next row is requested
#pragma unroll
    for (int j = 0; j < NQ; j++) cur[j] = ahead[j];
    pa = advance(pa);
    load_row(pa, ahead);
    const int64_t q = pc.q, k = pc.r;
    if (k == K) {
      // ---- the question's mD row: 1/D (:74), masked by the target gaps
#pragma unroll
      for (int j = 0; j < NQ; j++) {
        const uint32_t g = gapBits[j];
        id[j] = make_float4((g & 1) ? 0.f : rcp_nr_f32(cur[j].x), (g & 2) ? 0.f : rcp_nr_f32(cur[j].y),
                            (g & 4) ? 0.f : rcp_nr_f32(cur[j].z), (g & 8) ? 0.f : rcp_nr_f32(cur[j].w));
      }
      hW = accL = 0.f;
      pc = pa;
      continue;
    }
    // ---- pass 1 (:66-88): W_k = sum_t (A * invD) * prior; the likelihoods replace the row in its registers
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < NQ; j++) {
      cur[j].x = (cur[j].x * id[j].x) * pr[j].x;                        // :81-82
      cur[j].y = (cur[j].y * id[j].y) * pr[j].y;
      cur[j].z = (cur[j].z * id[j].z) * pr[j].z;
      cur[j].w = (cur[j].w * id[j].w) * pr[j].w;
      s += (cur[j].x + cur[j].y) + (cur[j].z + cur[j].w);
    }
    s = wave_sum_f32(s);
    if (lane == 0) wx[par][wave] = s;
    __syncthreads();                                                    // (the other parity's readers are a barrier behind)
    float Wk = 0.f;
    for (int w = 0; w < NW; w++) Wk += wx[par][w];
    par ^= 1;
    const float invWk = 1.0f / Wk;                                      // :91
    // ---- pass 2 (:95-128) on the same registers
    float v = 0.f;
    auto element = [&](float lh, float pi) __attribute__((always_inline)) {
      const float p = lh * invWk;                                       // :97
      const float l2 = log2p_f32(p);                                    // :106
      hW = fmaf(lh, l2, hW);                                            // :113-114 weighted by W_k (eval_epilogue)
      const float d = p - pi;                                           // :119
      v = fmaf(d, d, v);                                                // :126-127
      return l2;
    };
#pragma unroll
    for (int j = 0; j < NQ; j++) {
      // :117 lack += invD^2 / log2 p, two targets per reciprocal: (ix^2 lb + iy^2 la) / (la lb)
      // (the squares are written (ix lb) ix: as ix^2 lb they are invariant over the answers and get hoisted into 4 NQ registers)
      const float la = element(cur[j].x, pr[j].x), lb = element(cur[j].y, pr[j].y);
      accL = fmaf(fmaf(id[j].x * lb, id[j].x, (id[j].y * la) * id[j].y), __builtin_amdgcn_rcpf(la * lb), accL);
      const float lc = element(cur[j].z, pr[j].z), ld = element(cur[j].w, pr[j].w);
      accL = fmaf(fmaf(id[j].z * ld, id[j].z, (id[j].w * lc) * id[j].w), __builtin_amdgcn_rcpf(lc * ld), accL);
      __builtin_amdgcn_sched_barrier(0);   // one quad's chains in flight at a time: four logarithms cover each other's latency
    }
    v = wave_sum_f32(v);
    if (lane == 0) part[k][wave] = v;
    if (tid == 0) rec[k] = (double)Wk;
    pc = pa;
    if (k + 1 < K) continue;
    // ---- the question's last answer row is done
    hW = wave_sum_f32(hW);
    accL = wave_sum_f32(accL);
    if (lane == 0) { part[K][wave] = hW; part[K + 1][wave] = accL; }
    __syncthreads();
    // The question joins the queue of finished ones: one lane of wave 0 per sum folds the waves' partials (wave order).  The
    // epilogue itself -- ~2 us of dependent fp64 code -- is not run per question by one lane while every other wave waits at the
    // next row's barrier (a fifth of a 10 us question at 10000 targets): wave 0 runs up to kPend of them at once, one per lane.
    // (No barrier behind this: the next writes to part[] and rec[] come after the next row's barrier, which wave 0 reaches later.)
    if (wave == 0) {
      double *pq = pend + (size_t)nPend * kPendLen;
      if (lane < K + 2) {
        double sum = 0.0;
        for (int w = 0; w < NW; w++) sum += (double)part[lane][w];
        if (lane < K) { pq[lane] = rec[lane]; pq[K + lane] = rec[lane] * sqrt(sum); }   // W_k, W_k sqrt(V_k) (:156-157)
        else pq[K + lane] = sum;                                                        // [2K] sum W_k H_k, [2K + 1] lack
      }
      if (lane == 0) reinterpret_cast<int64_t *>(pq)[2 * K + 2] = q;
    }
    nPend++;
    if (nPend == kPend) {
      if (wave == 0) flush(nPend);
      nPend = 0;
    }
  }
  if (wave == 0 && nPend > 0) flush(nPend);
}

// The same sweep with the stream of rows landing in LDS by DMA, D rows ahead of the one computed on (round 5).  The
// register form above keeps ONE row per workgroup in flight -- in the registers that will compute on it -- and a workgroup of ten
// waves has a CU to itself: 40 KB in flight per CU at 10000 targets, which at ~2 us of memory latency is 5 TB/s for the chip, and
// that is what it reached (0.59 of the HBM peak).  Here a row is requested D rows before its turn and costs no registers while it
// flies (the 16 that `ahead` took are gone too): D x 40 KB per CU in flight.  Everything else -- the element arithmetic, one
// exchange per row, the queued epilogues -- is the register form's.
//   * row i of the stream lands in slot i % D of the ring: the quad of thread t, j at byte 16 (t + j NT) of the slot;
//   * where it is used: f32_shape / LaunchEvalQuestionsF32Reg below (rows of 8193 .. 10240 targets, as 512 threads of five quads);
//   * before row i is read the wave waits for all but the (D - 1) NQ youngest of its loads (the stream's end requests the prior
//     vector instead of rows, so that the count stays a constant);
//   * the words of the gap / asked bitmaps that advance() needs come by scalar loads: a vector load there would be waited for with
//     every DMA load in flight behind it.
template <int NQ, int D>
__global__ __launch_bounds__(1024) void eval_questions_f32_dma(F32Args a) {
  constexpr int kMaxWaves = 1024 / kWave;
  const int NT = (int)blockDim.x, NW = NT / kWave;
  __shared__ float wx[2][kMaxWaves];
  __shared__ float part[kMaxK + 2][kMaxWaves];
  __shared__ double rec[kMaxK];
  __shared__ double pend[kPend * kPendLen];
  extern __shared__ float4 ring[];                              // [D][NT * NQ]
  const int tid = threadIdx.x, lane = tid % kWave;
  const int wave = (int)__builtin_amdgcn_readfirstlane(tid / kWave);
  const int64_t K = a.K, ldT = a.ldT;
  const int nQuads = (int)(ldT >> 2);
  const int64_t qStride = (K + 1) * ldT;
  const unsigned slotBytes = (unsigned)NT * NQ * 16u;
  const unsigned ringAddr = (unsigned)(uintptr_t)ring;
  unsigned off[NQ];                                             // byte offset of the thread's quad j within a row (clamped: out-of-row quads re-read the last one and are masked)
  uint32_t gapBits[NQ];
  float4 pr[NQ];
#pragma unroll
  for (int j = 0; j < NQ; j++) {
    const int i = tid + j * NT;
    const bool in = i < nQuads;
    const int qi = in ? i : nQuads - 1;
    off[j] = (unsigned)qi * 16u;
    const uint32_t g = in ? (a.tgap[qi >> 3] >> ((4 * qi) & 31)) & 15u : 15u;
    gapBits[j] = g;
    const double2 p0 = reinterpret_cast<const double2 *>(a.prior)[2 * qi], p1 = reinterpret_cast<const double2 *>(a.prior)[2 * qi + 1];
    pr[j] = make_float4((g & 1) ? 0.f : (float)p0.x, (g & 2) ? 0.f : (float)p0.y, (g & 4) ? 0.f : (float)p1.x, (g & 8) ? 0.f : (float)p1.y);   // :103
  }
  auto unavailable = [&](int64_t q) { return (((scalar_word(a.qgap + (q >> 5)) | scalar_word(a.asked + (q >> 5))) >> (q & 31)) & 1u) != 0; };
  auto next_valid = [&](int64_t q) {    // :54 gap / asked questions get priority 0 and leave the stream
    while (q < a.Q && unavailable(q)) {
      if (tid == 0) a.priority[q] = 0.0;
      q += gridDim.x;
    }
    return q;
  };
  struct Pos { int64_t q; int64_t r; };
  auto advance = [&](Pos p) {
    if (p.q >= a.Q) return p;
    if (p.r == K) return Pos{p.q, 0};
    if (p.r + 1 < K) return Pos{p.q, p.r + 1};
    return Pos{next_valid(p.q + gridDim.x), K};
  };
  auto request = [&](Pos p, int slot) __attribute__((always_inline)) {
    const void *row = p.q < a.Q ? static_cast<const void *>(a.cube + p.q * qStride + p.r * ldT) : static_cast<const void *>(a.prior);
    const dma_rsrc_t rs = dma_rsrc(row, ldT * 4);
    const unsigned dst = ringAddr + (unsigned)slot * slotBytes + (unsigned)wave * 1024u;
#pragma unroll
    for (int j = 0; j < NQ; j++) dma16(rs, off[j], dst + (unsigned)j * ((unsigned)NT * 16u));
  };
  int nPend = 0;
  auto flush = [&](int n) {
    if (lane < n) {
      const double *pq = pend + (size_t)lane * kPendLen;
      const int64_t qq = reinterpret_cast<const int64_t *>(pq)[2 * K + 2];
      a.priority[qq] = epilogue_call(pq, -pq[2 * K], K, pq[2 * K + 1], a.vCompTail);
    }
  };
  Pos pc{next_valid(blockIdx.x), K}, pa = pc;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // (the prologue's own loads: the counted waits below are the DMA loads')
#pragma unroll
  for (int d = 0; d < D; d++) { request(pa, d); pa = advance(pa); }
  float4 cur[NQ], id[NQ];
  int par = 0, slot = 0;
  float hW = 0.f, accL = 0.f;
  while (pc.q < a.Q) {
    // ---- the oldest requested row has landed: into the registers, its slot to the row D ahead
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((D - 1) * NQ) : "memory");
    const float4 *src = ring + (size_t)slot * NT * NQ;
#pragma unroll
    for (int j = 0; j < NQ; j++) cur[j] = src[tid + j * NT];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    request(pa, slot);
    const Pos pn = advance(pc);                                  // the row after this one
    pa = advance(pa);
    slot = slot + 1 == D ? 0 : slot + 1;
    const int64_t q = pc.q, k = pc.r;
    pc = pn;
    if (k == K) {
#pragma unroll
      for (int j = 0; j < NQ; j++) {
        const uint32_t g = gapBits[j];
        id[j] = make_float4((g & 1) ? 0.f : rcp_nr_f32(cur[j].x), (g & 2) ? 0.f : rcp_nr_f32(cur[j].y),
                            (g & 4) ? 0.f : rcp_nr_f32(cur[j].z), (g & 8) ? 0.f : rcp_nr_f32(cur[j].w));   // :74
      }
      hW = accL = 0.f;
      continue;
    }
    // ---- pass 1 (:66-88)
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < NQ; j++) {
      cur[j].x = (cur[j].x * id[j].x) * pr[j].x;                        // :81-82
      cur[j].y = (cur[j].y * id[j].y) * pr[j].y;
      cur[j].z = (cur[j].z * id[j].z) * pr[j].z;
      cur[j].w = (cur[j].w * id[j].w) * pr[j].w;
      s += (cur[j].x + cur[j].y) + (cur[j].z + cur[j].w);
    }
    s = wave_sum_f32(s);
    if (lane == 0) wx[par][wave] = s;
    __syncthreads();
    float Wk = 0.f;
    for (int w = 0; w < NW; w++) Wk += wx[par][w];
    par ^= 1;
    const float invWk = 1.0f / Wk;                                      // :91
    // ---- pass 2 (:95-128) on the same registers
    float v = 0.f;
    auto element = [&](float lh, float pi) __attribute__((always_inline)) {
      const float p = lh * invWk;                                       // :97
      const float l2 = log2p_f32(p);                                    // :106
      hW = fmaf(lh, l2, hW);                                            // :113-114 weighted by W_k (eval_epilogue)
      const float d = p - pi;                                           // :119
      v = fmaf(d, d, v);                                                // :126-127
      return l2;
    };
#pragma unroll
    for (int j = 0; j < NQ; j++) {
      const float la = element(cur[j].x, pr[j].x), lb = element(cur[j].y, pr[j].y);
      accL = fmaf(fmaf(id[j].x * lb, id[j].x, (id[j].y * la) * id[j].y), __builtin_amdgcn_rcpf(la * lb), accL);   // :117, two targets per reciprocal
      const float lc = element(cur[j].z, pr[j].z), ld = element(cur[j].w, pr[j].w);
      accL = fmaf(fmaf(id[j].z * ld, id[j].z, (id[j].w * lc) * id[j].w), __builtin_amdgcn_rcpf(lc * ld), accL);
      __builtin_amdgcn_sched_barrier(0);
    }
    v = wave_sum_f32(v);
    if (lane == 0) part[k][wave] = v;
    if (tid == 0) rec[k] = (double)Wk;
    if (k + 1 < K) continue;
    // ---- the question's last answer row is done (as the register form)
    hW = wave_sum_f32(hW);
    accL = wave_sum_f32(accL);
    if (lane == 0) { part[K][wave] = hW; part[K + 1][wave] = accL; }
    __syncthreads();
    if (wave == 0) {
      double *pq = pend + (size_t)nPend * kPendLen;
      if (lane < K + 2) {
        double sum = 0.0;
        for (int w = 0; w < NW; w++) sum += (double)part[lane][w];
        if (lane < K) { pq[lane] = rec[lane]; pq[K + lane] = rec[lane] * sqrt(sum); }   // W_k, W_k sqrt(V_k) (:156-157)
        else pq[K + lane] = sum;                                                        // [2K] sum W_k H_k, [2K + 1] lack
      }
      if (lane == 0) reinterpret_cast<int64_t *>(pq)[2 * K + 2] = q;
    }
    nPend++;
    if (nPend == kPend) {
      if (wave == 0) flush(nPend);
      nPend = 0;
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // (the requests behind the stream's end)
  if (wave == 0 && nPend > 0) flush(nPend);
}

template <int NQ, int D>
hipError_t launch_dma(const F32Args &args, int nt, int nCU, int64_t maxGrid, hipStream_t stream) {
  auto kern = eval_questions_f32_dma<NQ, D>;
  const size_t shmem = (size_t)D * nt * NQ * 16;
  static LaunchCache cache;
  const int dev = LaunchCache::Device();
  int perCU = 0;
  const size_t key = shmem * 2048 + (size_t)nt;
  if (!cache.Get(dev, key, &perCU)) {
    if (shmem > 48 * 1024) {
      const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
      if (e != hipSuccess) return e;
    }
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&perCU, kern, nt, shmem) != hipSuccess || perCU < 1) perCU = 1;
    cache.Put(dev, key, perCU);
  }
  int64_t grid = std::min<int64_t>(args.Q, (int64_t)nCU * perCU);
  if (maxGrid > 0 && grid > maxGrid) grid = maxGrid;
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3((unsigned)nt), shmem, stream, args);
  return hipGetLastError();
}

template <int NQ>
hipError_t launch_reg(const F32Args &args, int nt, int nCU, int64_t maxGrid, hipStream_t stream) {
  auto kern = eval_questions_f32_reg<NQ>;
  int perCU = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&perCU, kern, nt, 0) != hipSuccess || perCU < 1) perCU = 1;
  int64_t grid = std::min<int64_t>(args.Q, (int64_t)nCU * perCU);
  if (maxGrid > 0 && grid > maxGrid) grid = maxGrid;
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3((unsigned)nt), 0, stream, args);
  return hipGetLastError();
}

// Shape for rows of ldT floats: {threads, quads per thread}; 0 threads = none (the streaming form takes the row).  What wastes
// the memory pipe is a thread without a quad of the row (it re-reads the last one and masks it) or with few of them (the per-row
// exchange is paid per thread: 4000 targets on 1024 x 1: 2.0 TB/s of cube, on 256 x 4: 4.8).  All shapes keep within 128 registers
// (sixteen waves per CU, whatever the workgroup size).  So: up to 4096 targets 256 threads with one to four quads each; above, four
// quads per thread and as many threads as that takes (a multiple of 64, up to 1024: 16384 targets).
// Measured (tools/f32_single_bench.py, TB/s of cube): 20000 x 5 x 4000 4.8, 10000 x 5 x 7000 4.4, 10000 x 5 x 10000 4.2,
// 4000 x 5 x 16000 4.5 -- against 2.0-2.3 for the streaming form and 5.1-5.3 for a Double engine's sweep on twice the bytes.
struct F32Shape { int nt, nq; };
F32Shape f32_shape(int64_t ldT, int64_t K, int variant) {
  if (K > kMaxK) return {0, 0};
  const int64_t nQuads = ldT >> 2;
  if (nQuads > 4096) return {0, 0};
  int nq = nQuads <= 1024 ? (int)((nQuads + 255) / 256) : 4;
  // 8193 .. 10240 targets: 512 threads of five quads -- eight waves, two to a SIMD (640 x 4 is ten: three on two of the SIMDs,
  // two on the others, and the row's barrier waits for the three) -- with the rows by LDS-DMA (eval_questions_f32_dma)
  if (nQuads > 2048 && nQuads <= 2560) nq = 5;
  if (variant >= 1 && variant <= 6) nq = variant;               // (tuning)
  int64_t nt = ((nQuads + nq - 1) / nq + kWave - 1) / kWave * kWave;
  if (nt < 256) nt = 256;
  if (nt > 1024) return {0, 0};
  return {(int)nt, nq};
}

}  // namespace

bool EvalF32RegisterShape(const KbView &kb, int variant) { return kb.elem == 4 && f32_shape(kb.ldT, kb.K, variant).nt != 0; }

const char *EvalF32KernelName(const KbView &kb, int variant) {
  static thread_local char name[40];
  const F32Shape s = f32_shape(kb.ldT, kb.K, variant);
  if (s.nt == 0) return "f32_stream";
  std::snprintf(name, sizeof(name), "f32_wg%d_nq%d", s.nt, s.nq);
  return name;
}

hipError_t LaunchEvalQuestionsF32Reg(const KbView &kb, const double *prior, const uint32_t *asked, double *priority, int variant,
                                     hipStream_t stream) {
  const F32Shape s = f32_shape(kb.ldT, kb.K, variant);
  if (kb.elem != 4 || s.nt == 0) return hipErrorInvalidValue;
  const double nT = (double)(kb.nValidTargets + 1);             // PqaCore/CEEvalQsSubtaskConsider.cpp:191
  F32Args a{static_cast<const float *>(kb.cube), prior, kb.tgap, kb.qgap, asked, priority, kb.K, kb.Q, kb.ldT,
            0.34657359027997265470861606072909 / (nT * nT)};
  int dev = 0, nCU = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&nCU, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || nCU <= 0) nCU = 256;
  // The rows by LDS-DMA where that puts more bytes in flight per CU than the register form's one row per workgroup: measured
  // (tools/f32_single_bench.py, one box, whole selections): 10000 x 5 x 10000 498 us as 640 x 4 in registers, 478 with the rows by
  // DMA, 444 as 512 x 5 by DMA two rows ahead (449 three ahead); NOT for shorter rows, whose register form has two to four
  // workgroups per CU that the ring's LDS would halve (4000 targets 358 -> 389 us, 7000 323 -> 398), nor at 16384 (279 -> 285).
  // PQA_F32_DMA=0 / PQA_F32_DEPTH: the register form / the ring's depth, for measurements.
  static const bool useDma = [] { const char *e = std::getenv("PQA_F32_DMA"); return !(e && e[0] == '0'); }();
  static const int dmaDepth = [] { const char *e = std::getenv("PQA_F32_DEPTH"); return e ? atoi(e) : 2; }();
  if (useDma && s.nq >= 5) {
    const size_t slot = (size_t)s.nt * s.nq * 16;
    const int d = dmaDepth >= 3 && 3 * slot + 12 * 1024 <= 160 * 1024 ? 3 : 2 * slot + 12 * 1024 <= 160 * 1024 ? 2 : 0;
    if (s.nq == 5 && d == 3) return launch_dma<5, 3>(a, s.nt, nCU, kb.maxGrid, stream);
    if (s.nq == 5 && d == 2) return launch_dma<5, 2>(a, s.nt, nCU, kb.maxGrid, stream);
    if (s.nq == 6 && d == 3) return launch_dma<6, 3>(a, s.nt, nCU, kb.maxGrid, stream);
    if (s.nq == 6 && d == 2) return launch_dma<6, 2>(a, s.nt, nCU, kb.maxGrid, stream);
    return hipErrorInvalidValue;
  }
  if (s.nq == 4 && useDma && std::getenv("PQA_F32_DMA4")) return launch_dma<4, 2>(a, s.nt, nCU, kb.maxGrid, stream);   // (measurement: 4 quads by DMA)
  switch (s.nq) {
    case 1: return launch_reg<1>(a, s.nt, nCU, kb.maxGrid, stream);
    case 2: return launch_reg<2>(a, s.nt, nCU, kb.maxGrid, stream);
    case 3: return launch_reg<3>(a, s.nt, nCU, kb.maxGrid, stream);
    case 4: return launch_reg<4>(a, s.nt, nCU, kb.maxGrid, stream);
    default: return hipErrorInvalidValue;
  }
}

}  // namespace pqa
