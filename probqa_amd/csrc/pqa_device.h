// pqa_device.h -- device-side building blocks shared by the gfx950 kernels: Log2Hot, compensated wave / workgroup
// reductions, bitmap helpers.  Compiled with -ffp-contract=off: fma() appears only where it is written.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace pqa {

constexpr int kWave = 64;  // CDNA wavefront
constexpr uint64_t kExpMaskUp = 0x7FF0000000000000ULL;
constexpr uint64_t kExp0Up = 0x3FF0000000000000ULL;

__device__ __forceinline__ uint64_t d2u(double d) { return (uint64_t)__double_as_longlong(d); }
__device__ __forceinline__ double u2d(uint64_t u) { return __longlong_as_double((long long)u); }

__device__ __forceinline__ bool bit_test(const uint32_t *__restrict__ bits, int64_t i) {
  return (bits[i >> 5] >> (i & 31)) & 1u;
}

// The 1024-entry table of SRVectMath::Log2Hot (reference: SRPlatform/SRVectMath.cpp:30-44): log2 of the bucket
// midpoint, entry 0 scaled by 9.9999999999999927e-01 so that log2(1) < 0.  It lives in eval_kernels.hip
// (gLog2Table), is built on the host with std::log2 and uploaded once (UploadLog2Table) -- the values the kernels
// see are exactly the host libm's, like the reference's.

// SRVectMath::Log2Hot (reference: SRPlatform/Interface/SRVectMath.h:87-135), one lane.  tbl points at the LDS copy
// of the table.  Operation-for-operation the reference's sequence (true division, the two explicit FMAs), so the
// result is bit-identical to the CPU for the same x.
__device__ __forceinline__ double log2hot(double x, const double *__restrict__ tbl) {
  const uint64_t ux = d2u(x);
  const int32_t hi = (int32_t)(ux >> 32);
  const uint64_t uz = (ux & ~kExpMaskUp) | kExp0Up;            // mantissa (and sign) with exponent 0: z in [1,2)
  const double z = u2d(uz);
  const int32_t e = (hi >> 20) - 1023;                         // arithmetic shift; x >= 0 assumed (:96-98)
  const int32_t idx = (hi >> 10) & 1023;                       // top 10 mantissa bits (:101-102)
  const double y = tbl[idx];
  const double m = u2d((1ULL << 41) | (uz & ~((1ULL << 42) - 1)));  // bucket midpoint (:108)
  const double t = (z - m) / (z + m);                          // :111-114
  const double t2 = t * t;
  const double t3 = t * t2;
  const double terms01 = fma(1.0 / 3, t3, t);                  // :118
  const double log2z = fma(terms01, 2.8853900817779268147198493620038, y);  // :122
  return log2z + (double)e;                                    // :131-133
}

// ---- error-free transformation and compensated reductions -------------------------------------------------------
__device__ __forceinline__ void two_sum(double a, double b, double &s, double &e) {
  s = a + b;
  const double bb = s - a;
  e = (a - (s - bb)) + (b - bb);
}

struct Comp {  // value = s + c, c holds the rounding errors of s
  double s, c;
};

__device__ __forceinline__ void comp_add(Comp &a, double x) {
  double e;
  two_sum(a.s, x, a.s, e);
  a.c += e;
}

__device__ __forceinline__ Comp comp_merge(Comp a, Comp b) {
  Comp r;
  double e;
  two_sum(a.s, b.s, r.s, e);
  r.c = (a.c + b.c) + e;
  return r;
}

__device__ __forceinline__ double shfl_xor_d(double v, int mask) { return __shfl_xor(v, mask, kWave); }

// Butterfly all-reduce over the 64 lanes; commutative steps => every lane ends with the same bits.
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int m = kWave / 2; m >= 1; m >>= 1) v += shfl_xor_d(v, m);
  return v;
}

__device__ __forceinline__ Comp wave_sum_comp(Comp v) {
#pragma unroll
  for (int m = kWave / 2; m >= 1; m >>= 1) {
    Comp o;
    o.s = shfl_xor_d(v.s, m);
    o.c = shfl_xor_d(v.c, m);
    v = comp_merge(v, o);
  }
  return v;
}

// Workgroup all-reduce of NV doubles over W waves.  scratch: LDS array of 2*W*NV doubles; `phase` alternates 0/1 per
// call so that one barrier per reduction suffices.  W == 1 needs no LDS and no barrier.
template <int W, int NV>
__device__ __forceinline__ void block_sum(double (&v)[NV], double *scratch, int &phase) {
#pragma unroll
  for (int i = 0; i < NV; i++) v[i] = wave_sum(v[i]);
  if constexpr (W > 1) {
    const int wave = threadIdx.x / kWave, lane = threadIdx.x % kWave;
    double *buf = scratch + phase * (W * NV);
    if (lane == 0) {
#pragma unroll
      for (int i = 0; i < NV; i++) buf[wave * NV + i] = v[i];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NV; i++) {
      double acc = buf[i];
#pragma unroll
      for (int w = 1; w < W; w++) acc += buf[w * NV + i];
      v[i] = acc;
    }
    phase ^= 1;
  }
}

template <int W>
__device__ __forceinline__ double block_sum_comp(Comp v, double *scratch, int &phase) {
  v = wave_sum_comp(v);
  if constexpr (W > 1) {
    const int wave = threadIdx.x / kWave, lane = threadIdx.x % kWave;
    double *buf = scratch + phase * (W * 2);
    if (lane == 0) {
      buf[wave * 2] = v.s;
      buf[wave * 2 + 1] = v.c;
    }
    __syncthreads();
    Comp acc = {buf[0], buf[1]};
#pragma unroll
    for (int w = 1; w < W; w++) acc = comp_merge(acc, Comp{buf[w * 2], buf[w * 2 + 1]});
    v = acc;
    phase ^= 1;
  }
  return v.s + v.c;
}

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

}  // namespace pqa
