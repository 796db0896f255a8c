// pqa_device.h -- device-side building blocks shared by the gfx950 kernels: Log2Hot, exact scale-free division,
// DPP wave reductions, bitmap helpers.  Compiled with -ffp-contract=off: fma() appears only where it is written.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>
#include <utility>

namespace pqa {

constexpr int kWave = 64;  // CDNA wavefront

// Compile-time unrolled loop: f(std::integral_constant<int, 0>) ... f(std::integral_constant<int, N-1>), for bodies that
// need the index as a constant expression (immediate operands, static ring slots).
template <typename F, int... I>
__device__ __forceinline__ void static_for_impl(F &&f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, typename F>
__device__ __forceinline__ void static_for(F &&f) {
  static_for_impl(static_cast<F &&>(f), std::make_integer_sequence<int, N>{});
}
constexpr uint64_t kExpMaskUp = 0x7FF0000000000000ULL;
constexpr uint64_t kExp0Up = 0x3FF0000000000000ULL;

__device__ __forceinline__ uint64_t d2u(double d) { return (uint64_t)__double_as_longlong(d); }
__device__ __forceinline__ double u2d(uint64_t u) { return __longlong_as_double((long long)u); }

__device__ __forceinline__ bool bit_test(const uint32_t *__restrict__ bits, int64_t i) {
  return (bits[i >> 5] >> (i & 31)) & 1u;
}

// ---- the cube's element type --------------------------------------------------------------------------------------
// Double engines keep sA / mD as fp64, Float engines (TPqaPrecisionType::Float, reference PqaCore/Interface/PqaCommon.h:17-24)
// as fp32.  The O(T) kernels around the sweep (posterior updates, training, maintenance) read and write single elements
// through these two; their arithmetic is fp64 either way (a Float cube is rounded on the store).
__device__ __forceinline__ double cube_ld(const void *cube, int elem, int64_t i) {
  return elem == 8 ? static_cast<const double *>(cube)[i] : (double)static_cast<const float *>(cube)[i];
}
__device__ __forceinline__ void cube_st(void *cube, int elem, int64_t i, double v) {
  if (elem == 8) static_cast<double *>(cube)[i] = v; else static_cast<float *>(cube)[i] = (float)v;
}

// ---- division ------------------------------------------------------------------------------------------------------
// IEEE-correct quotient for operands that need no scaling (no denormals / overflow in n, d, n/d): v_rcp_f64 (2^-24.4
// accurate on gfx950), one Newton step, then Markstein's residual correction.  37 cycles per wave instead of the 60-70
// of the compiler's div_scale / div_fmas / div_fixup sequence; bit-identical to '/' on 3e9 random operands in the
// ranges the sweep uses (tools/div_test.hip: Log2Hot's (z-m)/(z+m), 1/D, generic quotients).
__device__ __forceinline__ double div_nr(double n, double d) {
  double r = __builtin_amdgcn_rcp(d);
  const double e = fma(-d, r, 1.0);
  r = fma(r, e, r);
  const double q0 = n * r;
  const double rem = fma(-d, q0, n);
  return fma(rem, r, q0);
}

// ---- row loads through a buffer resource ------------------------------------------------------------------------------
// buffer_load_dwordx4 v, voffset, s[rsrc], 0 offen: address = row base (in the SGPR descriptor) + the lane's 32-bit byte
// offset.  No per-load address arithmetic on the VALU (a flat/global load needs a 64-bit add per lane per load), and the
// per-lane state is one VGPR per pair.  Reads beyond `bytes` return 0.
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
using RowRsrc = __amdgpu_buffer_rsrc_t;

// (the row pointer is wave-uniform by construction but hipcc cannot always prove it -- the question index comes out of
// loaded bitmap words -- and a descriptor it believes divergent costs a readfirstlane "waterfall" loop around EVERY load:
// 4 v_readfirstlane + 2 64-bit compares + exec juggling per 16-byte load.  Saying so once per row removes all of it.)
__device__ __forceinline__ RowRsrc row_rsrc(const void *row, int64_t bytes) {
  const uint64_t p = (uint64_t)(uintptr_t)row;
  const uint64_t pu = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(p >> 32)) << 32) |
                      (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)p);
  return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void *>((uintptr_t)pu), (short)0,
                                           __builtin_amdgcn_readfirstlane((int)bytes), 0x00020000);
}
// NT: with the hint that the bytes will not be read again (aux 2 = nt).  The cube streams -- a row is read once per sweep and by one
// workgroup -- and on a cube that lives in HBM the hint saves the L2s the allocation: round 4, two boxes, back to back, 10000^2
// 876 -> 853 and 893 -> 870 us, 8000^2 589 -> 578 and 598 -> 590 us.  A cube that fits the Infinity Cache is better off without it
// (1000^2: the resident step 15.3 -> 15.6 us), so the sweep gives the hint for rows beyond 4096 targets only (the same hint on the
// mD row's LDS-DMA loads changes nothing either way).
template <bool NT = false>
__device__ __forceinline__ double2 row_load(RowRsrc rs, uint32_t byteOffset) {
  return __builtin_bit_cast(double2, __builtin_amdgcn_raw_buffer_load_b128(rs, byteOffset, 0, NT ? 2 : 0));
}

// ---- SRVectMath::Log2Hot ---------------------------------------------------------------------------------------------
// Reference: SRPlatform/Interface/SRVectMath.h:87-135 with the 1024-entry table of SRPlatform/SRVectMath.cpp:30-44 (log2 of
// the bucket midpoint m, entry 0 scaled by 9.9999999999999927e-01 so that log2(1) < 0).  The table is built on the host
// with std::log2 and uploaded once (UploadLog2Table), so the kernels see exactly the host libm's values, like the
// reference's CPU code; each entry carries a second double, (2/ln 2)/(2m).
// log2hot() follows the reference operation for operation -- exponent / mantissa split, bucket midpoint, t = (z-m)/(z+m),
// the two explicit FMAs, + exponent -- except for how the quotient t is formed: the reference divides; here
// z+m = 2m(1+u), u = (z-m)/(2m), so t = u/(1+u) = u - u^2 + u^3 - u^4 ..., and the series t + t^3/3 of :118 with its scaling by
// 2/ln 2 (:122) is evaluated as one degree-4 polynomial (next term 3 u^5, |u| <= 2^-12): 5 fp64 operations instead of a 37-cycle
// exact division and four more.  The truncation reaches the result below 1.1e-17 absolute: the value is the reference's Log2Hot(x) except where that
// perturbation crosses a rounding boundary of the last two operations (a fraction of a percent of arguments, by one
// rounding unit; tools/log2hot_stats.py).  The host
// re-seats table entry 0 so that Log2Hot(1) stays negative under this arithmetic (hip_engine.cpp).  The priority vector
// is compared at 1e-9 relative (measured 4e-14, set by the summation order -- DESIGN.md section 5), not bit for bit.
// tbl: LDS copy of the table; it sits at LDS address 0 (first thing in the dynamic segment of kernels without static
// LDS, checked by the kernels through lds_table_at_zero), so the masked byte offset IS the ds_read address.
constexpr int kLog2TableDoubles = 2048;

__device__ __forceinline__ bool lds_table_at_zero(const double *tbl) {
  return (uint32_t)(uintptr_t)tbl == 0;
}

// The three bit patterns of the mantissa surgery, in registers: the two field masks in SGPRs and the fill pattern in a VGPR, so
// that each of the two results is ONE v_bfi_b32 (a VOP3 takes no literal on gfx9 and one SGPR; as literals the compiler spends
// v_and + v_or on each).  Plain asm without inputs: hoisted out of every loop and shared by all the call sites of a kernel.
struct Log2Bits { uint32_t keepZ, keepM, fill; };
__device__ __forceinline__ Log2Bits log2_bits() {
  Log2Bits k;
  asm("s_mov_b32 %0, 0x800fffff" : "=s"(k.keepZ));            // sign and mantissa: the rest <- exponent 0
  asm("s_mov_b32 %0, 0x800ffc00" : "=s"(k.keepM));            // sign and the top 10 mantissa bits: the rest <- exponent 0, 100...0
  asm("v_mov_b32 %0, 0x3ff00200" : "=v"(k.fill));
  return k;
}

__device__ __forceinline__ double log2hot(double x, const double *__restrict__ tbl) {
  // the bit surgery is done on the high word only (the low mantissa word passes through): 5 integer instructions per element
  // (two shifts, a mask, two bit-field inserts), of the ~26 the element costs at all
  const Log2Bits k = log2_bits();
  const uint32_t lo = (uint32_t)d2u(x);
  const uint32_t hi = (uint32_t)(d2u(x) >> 32);
  // exponent as a double (:96-98, :131; x >= 0 assumed): (2^52 + E) - (2^52 + 1023) with E the biased exponent field
  // planted in the low word of 2^52 -- exact, and an integer shift + one fp64 add instead of shift, add and a
  // quarter-rate v_cvt_f64_i32
  const double de = u2d(0x4330000000000000ULL | (uint64_t)(hi >> 20)) - 4503599627371519.0;
#ifdef PQA_ABLATE_TABLE_CONFLICTS   // measurement only (wrong values): every lane reads its own 16 bytes -- the gather without its bank conflicts
  const uint32_t tblByte = (((hi >> 6) & 0x0u) | ((uint32_t)__lane_id() << 4)) & 0x3FF0u;
#else
  const uint32_t tblByte = (hi >> 6) & 0x3FF0u;                // top 10 mantissa bits (:101-102), times 16
#endif
  (void)tbl;
  typedef double f64x2_t __attribute__((ext_vector_type(2)));
  const f64x2_t yc = *reinterpret_cast<const __attribute__((address_space(3))) f64x2_t *>((uintptr_t)tblByte);
  const uint32_t mhi = (hi & k.keepM) | (k.fill & ~k.keepM);   // bucket midpoint (:108): low 42 bits <- 100...0
  const uint32_t zhi = (hi & k.keepZ) | (k.fill & ~k.keepZ);   // mantissa (and sign) with exponent 0: z in [1,2)
  const double m = u2d((uint64_t)mhi << 32);
  const double z = u2d(((uint64_t)zhi << 32) | lo);
  const double w = (z - m) * yc.y;                             // z - m is exact (same binade, |z-m| < 2^-10); w = C u, u = (z-m)/(2m)
  // :111-122 t = (z-m)/(z+m) = u/(1+u), terms01 = t + t^3/3, log2 z = C * terms01 + y with C = 2/ln 2 -- as ONE polynomial: with
  // t = u - u^2 + u^3 - u^4 + ... and t^3 = u^3 - 3u^4 + ..., terms01 = u - u^2 + (4/3) u^3 - 2 u^4 (next term 3 u^5 <= 3 * 2^-60:
  // 1.1e-17 on the result), and in w = C u (the table's second double is C/(2m)):
  // C * terms01 = w (1 - w/C + (4/3) w^2/C^2 - 2 w^3/C^3).  Five operations for what the quotient (4), the two-term series (3) and
  // the scaling (1) took; the roundings of w and of the bracket reach the result below 3e-19.
  double c = fma(w, -0x1.55046a143789p-4, 0x1.47fd3ffac83b4p-3);   // -(ln 2)^3 / 4, (ln 2)^2 / 3
  c = fma(w, c, -0x1.62e42fefa39efp-2);                        // -(ln 2) / 2
  c = fma(w, c, 1.0);
  const double log2z = fma(w, c, yc.x);                        // :122
  return log2z + de;                                           // :131-133
}

// ---- error-free transformation and compensated values ---------------------------------------------------------------
__device__ __forceinline__ void two_sum(double a, double b, double &s, double &e) {
  s = a + b;
  const double bb = s - a;
  e = (a - (s - bb)) + (b - bb);
}

struct Comp {  // value = s + c, c holds the rounding errors of s
  double s, c;
};

__device__ __forceinline__ Comp comp_merge(Comp a, Comp b) {
  Comp r;
  double e;
  two_sum(a.s, b.s, r.s, e);
  r.c = (a.c + b.c) + e;
  return r;
}

// ---- cross-lane moves -----------------------------------------------------------------------------------------------
// DPP controls (gfx9): quad_perm[1,0,3,2] = lane^1, quad_perm[2,3,0,1] = lane^2, row_half_mirror = 7-lane within 8,
// row_mirror = 15-lane within 16.  Applied in this order to partial sums they form an all-reduce over a 16-lane row
// (after the two quad steps all lanes of a quad agree, so the mirrored partner holds the other quad's / half's sum).
constexpr int kDppXor1 = 0xB1, kDppXor2 = 0x4E, kDppHalfMirror = 0x141, kDppMirror = 0x140;

template <int CTRL>
__device__ __forceinline__ double mov_dpp(double v) {
  const uint64_t b = d2u(v);
  int lo = (int)(uint32_t)b, hi = (int)(uint32_t)(b >> 32);
  lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xF, 0xF, true);  // every lane has a valid source: no `old` copy
  hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xF, 0xF, true);
  return u2d(((uint64_t)(uint32_t)hi << 32) | (uint32_t)lo);
}

// v_permlane16_swap / v_permlane32_swap (gfx950): with both operands = v they return, per lane, this lane's value and
// the value of the lane 16 (resp. 32) away in the paired row (resp. half).
struct Pair { double a, b; };
__device__ __forceinline__ Pair swap16(double v) {
  const uint64_t b = d2u(v);
  const auto r0 = __builtin_amdgcn_permlane16_swap((uint32_t)b, (uint32_t)b, false, false);
  const auto r1 = __builtin_amdgcn_permlane16_swap((uint32_t)(b >> 32), (uint32_t)(b >> 32), false, false);
  return Pair{u2d(((uint64_t)r1[0] << 32) | r0[0]), u2d(((uint64_t)r1[1] << 32) | r0[1])};
}
__device__ __forceinline__ Pair swap32(double v) {
  const uint64_t b = d2u(v);
  const auto r0 = __builtin_amdgcn_permlane32_swap((uint32_t)b, (uint32_t)b, false, false);
  const auto r1 = __builtin_amdgcn_permlane32_swap((uint32_t)(b >> 32), (uint32_t)(b >> 32), false, false);
  return Pair{u2d(((uint64_t)r1[0] << 32) | r0[0]), u2d(((uint64_t)r1[1] << 32) | r0[1])};
}

// All-reduce (sum) over the 64 lanes of a wave: 4 DPP steps inside each row of 16, then rows, then halves.
// Every lane ends with the same bits (each step adds the same two partial sums in both partner lanes).
__device__ __forceinline__ double wave_sum(double v) {
  v += mov_dpp<kDppXor1>(v);
  v += mov_dpp<kDppXor2>(v);
  v += mov_dpp<kDppHalfMirror>(v);
  v += mov_dpp<kDppMirror>(v);
  Pair p = swap16(v);
  v = p.a + p.b;
  p = swap32(v);
  return p.a + p.b;
}

// The same for a float: 4 DPP steps, 2 permlane swaps -- six full-rate instruction pairs instead of the six ds_bpermute round
// trips (~100 cycles each, in series) that __shfl_xor costs.
template <int CTRL>
__device__ __forceinline__ float mov_dpp_f32(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float wave_sum_f32(float v) {
  v += mov_dpp_f32<kDppXor1>(v);
  v += mov_dpp_f32<kDppXor2>(v);
  v += mov_dpp_f32<kDppHalfMirror>(v);
  v += mov_dpp_f32<kDppMirror>(v);
  const uint32_t b = __builtin_bit_cast(uint32_t, v);
  const auto r = __builtin_amdgcn_permlane16_swap(b, b, false, false);
  v = __builtin_bit_cast(float, (uint32_t)r[0]) + __builtin_bit_cast(float, (uint32_t)r[1]);
  const uint32_t c = __builtin_bit_cast(uint32_t, v);
  const auto r2 = __builtin_amdgcn_permlane32_swap(c, c, false, false);
  return __builtin_bit_cast(float, (uint32_t)r2[0]) + __builtin_bit_cast(float, (uint32_t)r2[1]);
}

__device__ __forceinline__ Comp wave_sum_comp(Comp v) {
  v = comp_merge(v, Comp{mov_dpp<kDppXor1>(v.s), mov_dpp<kDppXor1>(v.c)});
  v = comp_merge(v, Comp{mov_dpp<kDppXor2>(v.s), mov_dpp<kDppXor2>(v.c)});
  v = comp_merge(v, Comp{mov_dpp<kDppHalfMirror>(v.s), mov_dpp<kDppHalfMirror>(v.c)});
  v = comp_merge(v, Comp{mov_dpp<kDppMirror>(v.s), mov_dpp<kDppMirror>(v.c)});
  Pair ps = swap16(v.s), pc = swap16(v.c);
  v = comp_merge(Comp{ps.a, pc.a}, Comp{ps.b, pc.b});
  ps = swap32(v.s);
  pc = swap32(v.c);
  return comp_merge(Comp{ps.a, pc.a}, Comp{ps.b, pc.b});
}

// Butterfly over the first W (power of two <= 16) lanes of each 16-lane row, for combining per-wave partials that
// every lane loaded as entry (lane % W).
template <int W>
__device__ __forceinline__ Comp row_sum_comp(Comp v) {
  if constexpr (W >= 2) v = comp_merge(v, Comp{mov_dpp<kDppXor1>(v.s), mov_dpp<kDppXor1>(v.c)});
  if constexpr (W >= 4) v = comp_merge(v, Comp{mov_dpp<kDppXor2>(v.s), mov_dpp<kDppXor2>(v.c)});
  if constexpr (W >= 8) v = comp_merge(v, Comp{mov_dpp<kDppHalfMirror>(v.s), mov_dpp<kDppHalfMirror>(v.c)});
  if constexpr (W >= 16) v = comp_merge(v, Comp{mov_dpp<kDppMirror>(v.s), mov_dpp<kDppMirror>(v.c)});
  return v;
}

template <int W>
__device__ __forceinline__ double row_sum(double v) {
  if constexpr (W >= 2) v += mov_dpp<kDppXor1>(v);
  if constexpr (W >= 4) v += mov_dpp<kDppXor2>(v);
  if constexpr (W >= 8) v += mov_dpp<kDppHalfMirror>(v);
  if constexpr (W >= 16) v += mov_dpp<kDppMirror>(v);
  return v;
}

// shuffle-based variants for the small single-workgroup kernels
__device__ __forceinline__ double shfl_xor_d(double v, int mask) { return __shfl_xor(v, mask, kWave); }

// ---- scalar Kahan state of the reference (SRPlatform/Interface/SRAccumulator.h:15-39, SRAccumVectDbl256.h) -------
struct Kahan1 {
  double sum, corr;
  __device__ __forceinline__ void init(double v) { sum = v; corr = 0; }
  __device__ __forceinline__ void add(double v) {
    const double y = v - corr;
    const double t = sum + y;
    corr = (t - sum) - y;
    sum = t;
  }
  __device__ __forceinline__ double get() const { return sum - corr; }
};

// SRAccumVectDbl256::PreciseSum (SRAccumVectDbl256.h:83-91) over 4 lanes given as arrays
__device__ __forceinline__ double precise_sum4(const double *sum, const double *corr) {
  Kahan1 a;
  a.init(corr[3]);
  for (int i = 2; i >= 0; i--) a.add(corr[i]);
  a.sum = -a.sum;
  a.corr = -a.corr;
  for (int i = 3; i >= 0; i--) a.add(sum[i]);
  return a.get();
}

// ---- the reference's sampled selector for one workgroup (PqaCore/CpuEngine.cpp:362-400) ------------------------------------
// Per-subtask Kahan run lengths, Kahan grand totals, one uniform number, two upper_bounds -- step for step, so that with the
// same priorities, subtask count and random number it returns the reference's question.  Called by every thread of the
// workgroup (barriers inside); the result is valid in thread 0.  COH: the priorities were written by other workgroups of
// the same launch (fused with the sweep) and are read past the non-coherent cache levels.
// std::upper_bound: first element strictly greater than v
__device__ __forceinline__ int64_t upper_bound_d(const double *a, int64_t n, double v) {
  int64_t lo = 0, hi = n;
  while (lo < hi) {
    const int64_t mid = lo + ((hi - lo) >> 1);
    if (!(v < a[mid])) lo = mid + 1; else hi = mid;
  }
  return lo;
}
// SRPoolRunner::CalcSplit bound i (reference: SRPlatform/Interface/SRPoolRunner.h:96-110) in closed form:
// the first `rem` subtasks get quot+1 items.
__device__ __forceinline__ int64_t calc_split_bound(int64_t i, int64_t quot, int64_t rem) {  // end of subtask i
  const int64_t n1 = (i + 1 < rem) ? (i + 1) : rem;
  return (i + 1) * quot + n1;
}
struct SampledPick { double priority; int64_t index; };
template <bool COH>
__device__ __forceinline__ SampledPick select_sampled_wg_impl(const double *priority, const uint32_t *qgap, const uint32_t *asked,
                                                              int64_t qFirst, int64_t n, int64_t nWorkers, uint64_t rnd,
                                                              double *runLength, double *grand) {
  const int64_t quot = n / nWorkers, rem = n % nWorkers;
  const int64_t nSubtasks = (quot == 0) ? rem : nWorkers;  // CalcSplit stops once the items run out
  // per-subtask inclusive Kahan running sums (PqaCore/CEEvalQsSubtaskConsider.cpp:52,212-214)
  for (int64_t s = threadIdx.x; s < nSubtasks; s += blockDim.x) {
    const int64_t first = (s == 0) ? 0 : calc_split_bound(s - 1, quot, rem), limit = calc_split_bound(s, quot, rem);
    Kahan1 acc;
    acc.init(0.0);
    for (int64_t i = first; i < limit; i++) {
      // gap / asked questions only copy the running sum (:54-58); evaluated ones are Kahan-added (:212)
      if (!(bit_test(qgap, qFirst + i) || bit_test(asked, qFirst + i)))
        acc.add(COH ? __hip_atomic_load(priority + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : priority[i]);
      runLength[i] = acc.get();
    }
    grand[s] = acc.get();
  }
  __syncthreads();
  SampledPick r{0.0, -1};
  if (threadIdx.x == 0) {
    Kahan1 accTotG;
    accTotG.init(0.0);                                         // PqaCore/CpuEngine.cpp:362
    // 16 totals at a time: their loads are issued together and the results stored afterwards, so that only the Kahan
    // chain itself (4 dependent operations per subtask) is serial, not an LDS round trip per subtask as well
    for (int64_t s0 = 0; s0 < nSubtasks; s0 += 16) {
      double g[16];
#pragma unroll
      for (int u = 0; u < 16; u++) g[u] = s0 + u < nSubtasks ? grand[s0 + u] : 0.0;
#pragma unroll
      for (int u = 0; u < 16; u++) {
        if (s0 + u < nSubtasks) {
          accTotG.add(g[u]);                                   // :366-367
          g[u] = accTotG.get();                                // :368
        }
      }
#pragma unroll
      for (int u = 0; u < 16; u++)
        if (s0 + u < nSubtasks) grand[s0 + u] = g[u];
    }
    const double totG = grand[nSubtasks - 1];                  // :375
    // SRDoubleNumber::MakeRandom (SRPlatform/Interface/SRDoubleNumber.h:35-39)
    const double selRunLen = totG * (double)rnd / 18446744073709551615.0;  // :379
    int64_t sel;
    const int64_t iWorker = upper_bound_d(grand, nSubtasks, selRunLen);    // :380-381
    if (iWorker >= nSubtasks) {
      sel = n - 1;                                             // :384
    } else {
      const double inWorkerRunLen = selRunLen - ((iWorker == 0) ? 0.0 : grand[iWorker - 1]);  // :388
      const int64_t first = (iWorker == 0) ? 0 : calc_split_bound(iWorker - 1, quot, rem);    // :389
      const int64_t limit = calc_split_bound(iWorker, quot, rem);                              // :390
      sel = first + upper_bound_d(runLength + first, limit - first, inWorkerRunLen);           // :391
      if (sel >= limit) sel = limit - 1;                       // :392-400
    }
    r.priority = totG;
    r.index = sel;
  }
  return r;
}

// The same selection with everything it touches staged in LDS first (`lds`: n + nWorkers + n/64 + 2 doubles): the
// priorities arrive in ONE parallel round of loads instead of one dependent load per item of a subtask's chain -- what
// makes the selection affordable inside the sweep's launch, where those loads go past the caches (sc1, ~2 us each).
template <bool COH>
__device__ __forceinline__ SampledPick select_sampled_wg_lds(const double *priority, const uint32_t *qgap, const uint32_t *asked,
                                                             int64_t qFirst, int64_t n, int64_t nWorkers, uint64_t rnd,
                                                             double *lds) {
  double *run = lds;                                   // priorities, then (in place) the run lengths
  double *grand = lds + n;
  uint32_t *skip = reinterpret_cast<uint32_t *>(grand + nWorkers);   // gap | asked bits of the n questions, local numbering
  const int64_t nWords = (n + 31) >> 5;
  for (int64_t i0 = threadIdx.x; i0 < n; i0 += 8 * (int64_t)blockDim.x) {   // eight loads in flight per thread, not one
    double v[8];
#pragma unroll
    for (int u = 0; u < 8; u++) {
      const int64_t i = i0 + u * (int64_t)blockDim.x;
      v[u] = i < n ? (COH ? __hip_atomic_load(priority + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : priority[i]) : 0.0;
    }
#pragma unroll
    for (int u = 0; u < 8; u++) {
      const int64_t i = i0 + u * (int64_t)blockDim.x;
      if (i < n) run[i] = v[u];
    }
  }
  for (int64_t w = threadIdx.x; w < nWords; w += blockDim.x) {
    uint32_t bits = 0;
    if ((qFirst & 31) == 0) {            // (every caller's case: whole words)
      bits = qgap[(qFirst >> 5) + w] | asked[(qFirst >> 5) + w];
    } else {
      for (int b = 0; b < 32; b++) {
        const int64_t i = (w << 5) + b;
        if (i < n && (bit_test(qgap, qFirst + i) || bit_test(asked, qFirst + i))) bits |= 1u << b;
      }
    }
    skip[w] = bits;
  }
  __syncthreads();
  const int64_t quot = n / nWorkers, rem = n % nWorkers;
  const int64_t nSubtasks = (quot == 0) ? rem : nWorkers;
  for (int64_t s = threadIdx.x; s < nSubtasks; s += blockDim.x) {
    const int64_t first = (s == 0) ? 0 : calc_split_bound(s - 1, quot, rem), limit = calc_split_bound(s, quot, rem);
    Kahan1 acc;
    acc.init(0.0);
    for (int64_t i = first; i < limit; i++) {
      if (!((skip[i >> 5] >> (i & 31)) & 1u)) acc.add(run[i]);   // :54-58 / :212
      run[i] = acc.get();
    }
    grand[s] = acc.get();
  }
  __syncthreads();
  SampledPick r{0.0, -1};
  if (threadIdx.x == 0) {
    Kahan1 accTotG;
    accTotG.init(0.0);                                         // PqaCore/CpuEngine.cpp:362
    // 16 totals at a time: their loads are issued together and the results stored afterwards, so that only the Kahan
    // chain itself (4 dependent operations per subtask) is serial, not an LDS round trip per subtask as well
    for (int64_t s0 = 0; s0 < nSubtasks; s0 += 16) {
      double g[16];
#pragma unroll
      for (int u = 0; u < 16; u++) g[u] = s0 + u < nSubtasks ? grand[s0 + u] : 0.0;
#pragma unroll
      for (int u = 0; u < 16; u++) {
        if (s0 + u < nSubtasks) {
          accTotG.add(g[u]);                                   // :366-367
          g[u] = accTotG.get();                                // :368
        }
      }
#pragma unroll
      for (int u = 0; u < 16; u++)
        if (s0 + u < nSubtasks) grand[s0 + u] = g[u];
    }
    const double totG = grand[nSubtasks - 1];                  // :375
    const double selRunLen = totG * (double)rnd / 18446744073709551615.0;  // :379
    int64_t sel;
    const int64_t iWorker = upper_bound_d(grand, nSubtasks, selRunLen);    // :380-381
    if (iWorker >= nSubtasks) {
      sel = n - 1;                                             // :384
    } else {
      const double inWorkerRunLen = selRunLen - ((iWorker == 0) ? 0.0 : grand[iWorker - 1]);  // :388
      const int64_t first = (iWorker == 0) ? 0 : calc_split_bound(iWorker - 1, quot, rem);    // :389
      const int64_t limit = calc_split_bound(iWorker, quot, rem);                              // :390
      sel = first + upper_bound_d(run + first, limit - first, inWorkerRunLen);                 // :391
      if (sel >= limit) sel = limit - 1;                       // :392-400
    }
    r.priority = totG;
    r.index = sel;
  }
  return r;
}
__host__ __device__ constexpr int64_t select_sampled_lds_doubles(int64_t n, int64_t nWorkers) { return n + nWorkers + n / 64 + 2; }

// ---- top-maxCount targets by probability -------------------------------------------------------------------------------
// Descending probability, lower index first on ties, gaps and probabilities <= 0 never listed (reference
// PqaCore/CEListTopTargetsAlgorithm.cpp:30-95 over CEHeapifyPriorsSubtaskMake.cpp:42-52).  Ties: the reference lists equal
// probabilities in the order its per-thread heaps leave them -- a function of the machine's thread count.  These rounds list them by
// ascending target index; the engine asks for one entry more than its caller, and where that listing shows a tie it is made again
// by the heaps themselves (kb_kernels.hip: LaunchTopTargetsExact).
// For a workgroup of up to 1024 threads: every thread holds its E targets (t = tid + e*blockDim) in registers; a round is one wave
// argmax by DPP / permlane swaps, one LDS exchange and ONE barrier (the per-wave results alternate between two LDS rows by
// round parity), one 16-lane row argmax over the 16 waves' results, after which every thread knows the round's winner and
// its owner retires it: ~0.3 us per round.  T <= 1024*E <= 16384; small maxCount.
struct TopOut {
  int64_t iTarget;
  double prob;
};
// A round = the maximum probability over the workgroup (v_max_f64 butterflies), then the lowest index among the targets
// that hold it (integer-min butterflies, whose DPP form is one instruction per step): two short dependent chains.  The
// first version carried {probability, index} pairs through one butterfly with a four-way comparison per step -- ten
// steps of ~25 dependent instructions, 2.7 us per listed target, three quarters of RecordAnswer's kernel.
// Listed: valid targets by descending probability, lower index first on ties; retired and invalid ones hold -1.
// (v_max_f64 as written: the compiler's fmax quiets signalling NaNs first -- a second v_max_f64 per operand; no NaN reaches these)
__device__ __forceinline__ double max_raw(double a, double b) {
  double r;
  asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ double wave_max(double v) {
  v = max_raw(v, mov_dpp<kDppXor1>(v));
  v = max_raw(v, mov_dpp<kDppXor2>(v));
  v = max_raw(v, mov_dpp<kDppHalfMirror>(v));
  v = max_raw(v, mov_dpp<kDppMirror>(v));
  Pair p = swap16(v);
  v = max_raw(p.a, p.b);
  p = swap32(v);
  return max_raw(p.a, p.b);
}
__device__ __forceinline__ double row_max(double v) {   // over the 16 lanes of a row
  v = max_raw(v, mov_dpp<kDppXor1>(v));
  v = max_raw(v, mov_dpp<kDppXor2>(v));
  v = max_raw(v, mov_dpp<kDppHalfMirror>(v));
  return max_raw(v, mov_dpp<kDppMirror>(v));
}
template <int CTRL>
__device__ __forceinline__ int min_dpp(int v) {
  const int o = __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, true);
  return o < v ? o : v;
}
__device__ __forceinline__ int row_min(int v) {
  v = min_dpp<kDppXor1>(v);
  v = min_dpp<kDppXor2>(v);
  v = min_dpp<kDppHalfMirror>(v);
  return min_dpp<kDppMirror>(v);
}
__device__ __forceinline__ int wave_min(int v) {
  v = row_min(v);
  auto t = __builtin_amdgcn_permlane16_swap((uint32_t)v, (uint32_t)v, false, false);
  v = (int)t[0] < (int)t[1] ? (int)t[0] : (int)t[1];
  t = __builtin_amdgcn_permlane32_swap((uint32_t)v, (uint32_t)v, false, false);
  return (int)t[0] < (int)t[1] ? (int)t[0] : (int)t[1];
}
constexpr int kTopNone = 0x7FFFFFFF;
// The lane's own best candidate: the highest probability it holds, the lowest target among equals.  A lane's targets ascend with e
// wherever its probabilities tie (t = base + tid + e * blockDim over a posterior; position in a sequence of lists that are each
// ascending among equals and cover ascending target ranges), so the first e that holds the maximum is the one.
template <int E>
__device__ __forceinline__ void lane_best(const double (&p)[E], const int (&t)[E], double &bp, int &bt) {
  bp = p[0];
#pragma unroll
  for (int e = 1; e < E; e++) bp = max_raw(bp, p[e]);
  bt = t[E - 1];
#pragma unroll
  for (int e = E - 2; e >= 0; e--) bt = p[e] == bp ? t[e] : bt;
}
// The rounds over what the threads of a WORKGROUP hold: p[e] > 0 for a candidate (anything else, -1, is never listed), t[e] its target.
// Every lane carries its own best from round to round; a round is the all-reduce of those, and only the lane that held the winner --
// one lane of one wave -- looks at its E candidates again.
template <int E>
__device__ __forceinline__ int64_t top_rounds(double (&p)[E], int (&t)[E], int64_t maxCount, TopOut *out, double (*sp)[16], int (*st)[16]) {
  const int lane = threadIdx.x % kWave, wave = threadIdx.x / kWave;
  int64_t listed = 0;
  double bp;
  int bt;
  lane_best<E>(p, t, bp, bt);
  for (int64_t r = 0; r < maxCount; r++) {
    const double wm = wave_max(bp);
    const int wt = wave_min(bp == wm ? bt : kTopNone);
    const int par = (int)(r & 1);
    if (lane == 0) {
      sp[par][wave] = wm;
      st[par][wave] = wt;
    }
    __syncthreads();
    const double mp = sp[par][lane % 16];     // up to 16 waves -> one 16-lane row (absent waves hold -1)
    const int mt = st[par][lane % 16];
    const double gm = row_max(mp);
    if (!(gm > 0.0)) break;                   // nothing left; the same for every thread
    const int gt = row_min(mp == gm ? mt : kTopNone);
    if (threadIdx.x == 0) out[r] = TopOut{gt, gm};   // `out`: LDS staging (top_targets_publish, the batched listing's kernels)
    if (bt == gt) {                           // (targets are unique: this lane held the winner)
#pragma unroll
      for (int e = 0; e < E; e++)
        if (t[e] == gt) p[e] = -1.0;
      lane_best<E>(p, t, bp, bt);
    }
    listed++;
  }
  return listed;
}
// The same for ONE WAVE by itself (no LDS, no barrier): its lanes' candidates only; every lane learns every round's winner.
template <int E>
__device__ __forceinline__ int64_t top_rounds_wave(double (&p)[E], int (&t)[E], int64_t maxCount, TopOut *out) {
  const int lane = threadIdx.x % kWave;
  int64_t listed = 0;
  double bp;
  int bt;
  lane_best<E>(p, t, bp, bt);
  for (int64_t r = 0; r < maxCount; r++) {
    const double gm = wave_max(bp);
    if (!(gm > 0.0)) break;
    const int gt = wave_min(bp == gm ? bt : kTopNone);
    if (lane == 0) out[r] = TopOut{gt, gm};
    if (bt == gt) {
#pragma unroll
      for (int e = 0; e < E; e++)
        if (t[e] == gt) p[e] = -1.0;
      lane_best<E>(p, t, bp, bt);
    }
    listed++;
  }
  return listed;
}
// A target is listed if it is no gap and its probability is > 0: the reference drops `prob <= 0` (PqaCore/CEHeapifyPriorsSubtaskMake.cpp:43-49,
// PqaCore/CERadixSortRatingsSubtaskSort.cpp:78-84) -- a posterior element that underflowed to exactly 0, or that NormalizePriors flushed
// (PqaCore/CENormPriorsSubtaskCorrSum.cpp:32-36), is no candidate.  (A NaN is never listed here.)
template <int E>
__device__ __forceinline__ int64_t top_targets_rounds(const double *prior, const uint32_t *tgap, int64_t tFirst, int64_t tLimit, int64_t maxCount,
                                                      TopOut *out, double (*sp)[16], int (*st)[16]) {
  double p[E];
  int t[E];
#pragma unroll
  for (int e = 0; e < E; e++) {
    const int64_t tt = tFirst + threadIdx.x + (int64_t)e * blockDim.x;
    const bool ok = tt < tLimit && !bit_test(tgap, tt);
    p[e] = ok ? prior[tt] : -1.0;
    if (!(p[e] > 0.0)) p[e] = -1.0;
    t[e] = (int)tt;
  }
  return top_rounds<E>(p, t, maxCount, out, sp, st);
}
// the list, its length and then a flag, into host-coherent memory (T <= 16384, maxCount <= 256).  The winners are staged
// in LDS and leave in one coalesced burst at the end: a store to host memory per round costs more than the round.
// SMALL: only the <= 4 targets per thread forms (a 256-thread launch over <= 1024 targets keeps 40 VGPRs and fits beside
// the resident sweep's workgroups, prior_kernels.hip).
struct TopScratch {          // the listing's LDS: the caller's, so that a kernel whose LDS layout is fixed can place it
  double sp[2][16];
  int st[2][16];
  TopOut staged[256];
};
template <bool SMALL = false>
__device__ __forceinline__ void top_targets_publish(const double *prior, const uint32_t *tgap, int64_t T, int64_t maxCount,
                                                    TopOut *out, int64_t *nOut, uint64_t *flag, uint64_t flagValue,
                                                    TopScratch *scratch) {
  double (*sp)[16] = scratch->sp;
  int (*st)[16] = scratch->st;
  TopOut *staged = scratch->staged;
  if (maxCount > 256) maxCount = 256;
  // waves that do not exist never win a round
  if (threadIdx.x < 32) {
    sp[threadIdx.x >> 4][threadIdx.x & 15] = -1.0;
    st[threadIdx.x >> 4][threadIdx.x & 15] = kTopNone;
  }
  __syncthreads();
  const int64_t perThread = (T + blockDim.x - 1) / blockDim.x;
  int64_t listed;
  if (perThread <= 1) listed = top_targets_rounds<1>(prior, tgap, 0, T, maxCount, staged, sp, st);
  else if (SMALL || perThread <= 4) listed = top_targets_rounds<4>(prior, tgap, 0, T, maxCount, staged, sp, st);
  else listed = top_targets_rounds<16>(prior, tgap, 0, T, maxCount, staged, sp, st);
  __syncthreads();
  if ((int64_t)threadIdx.x < listed) out[threadIdx.x] = staged[threadIdx.x];
  __syncthreads();
  if (threadIdx.x == 0) {
    *nOut = listed;
    if (flag != nullptr) {  // the host polls: no copy, no synchronise
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
      __hip_atomic_store(flag, flagValue, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

}  // namespace pqa
