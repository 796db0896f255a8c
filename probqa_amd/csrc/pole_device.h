// pole_device.h -- rows at the pole of the lack term: what the sweeps' watch (eval_kernels.hip, cluster_kernels.hip,
// batch_kernels.hip) and the fix launched behind them (pole_kernels.hip: the whole story) share -- the bars, the suspect list's
// append, Log2Hot by the reference's own operation sequence.  Reference: PqaCore/CEEvalQsSubtaskConsider.cpp:62-132,
// SRPlatform/Interface/SRAccumVectDbl256.h:40-46, :62-92, SRPlatform/Interface/SRVectMath.h:87-135.
#pragma once
#include "pqa_device.h"
#include "eval_device.h"
#include "pqa_kernels.h"

namespace pqa {

// A row is redone in the reference's order when its largest posterior element is at least 1 - 2^-10.  Two mechanisms set the bar.
// The lack term's pole: the summation order of W_k moves the priority by ~1e-16 / (1 - p): 1e-9 at p = 1 - 1e-7.  The velocity
// term reaches further out: (p - prior)^2 of a target that holds nearly all the mass BEFORE and AFTER the answer is the square of
// a difference of two numbers next to 1 -- |p - prior| ~ (1 - p) x (how much the answer tells this target from the rest) -- and
// p's last place (1.1e-16) is 1e-8 of it at p = 1 - 4e-5, where the likeliest answer's velocity is a sixth of the question's:
// 1.3e-9 of the priority (tests/test_gpu_late.py: the first cases that found it).  At 1 - 2^-10 both are below 1e-11.
constexpr uint32_t kNearOneHi = 0x3FEFF800u;   // high word of 1 - 2^-10
// The velocity term again, further out still: an answer that tells a question's targets apart by less than one part in 10^5 -- the
// near-certain answer of a late quiz, or of a question with few answers -- leaves the posterior where the prior was, and the
// row's velocity sum V_k = sum (p - prior)^2 is then the square of ONE difference |d| ~ sqrt(V_k) carrying p's last place:
// 2.2e-16 / sqrt(V_k) relative, of which up to a half reaches the priority.  Rows with V_k <= kSmallV whose largest element holds
// at least a quarter of the mass (below that no single difference makes up the sum) are redone as well: what the others keep is
// below 1e-10.  (4e-10, the first bar, listed most questions of a quiz three answers deep on 10000 targets for 3e-12 apiece.)
constexpr double kSmallV = 2e-12;
constexpr uint32_t kQuarterHi = 0x3FCE0000u;   // high word of 0.234: the element of a listed row whose terms are corrected

// a sweep's entry for a question that passed its watch (pqa_kernels.h: PoleHeader; one thread)
__device__ __forceinline__ uint32_t pole_list_append(PoleHeader *list, uint32_t q, uint32_t rowMask, uint32_t b, uint32_t gap = 0u) {
  const uint32_t at = atomicAdd(&list->count, 1u);
  reinterpret_cast<PoleEntry *>(list + 1)[at] = PoleEntry{q, rowMask, b, gap};
  return at;
}
// The gap a sweep that tracks it (KbView::poleGate) hands over with an entry: a lower bound of 1 - p for the largest element of the
// question's listed rows, as float bits rounded down (positive floats order as their bit patterns: an LDS atomicMin gathers the lanes').
// A row whose lane sums all stay 2^-8 below W_k -- not "near one" for the watch -- has no element above 1 - 2^-9: kGapNoneBits.
constexpr uint32_t kGapNoneBits = 0x3A7FF000u;   // just below 2^-10
__device__ __forceinline__ uint32_t pole_gap_bits(double gap) {
  const float f = (float)(gap * 0.99999);        // (below the double whatever the rounding of the conversion)
  return f > 0.0f ? __float_as_uint(f) : 0u;     // (0, denormal-flushed or negative: not known -- such a question is always redone)
}

// SRVectMath.h:87-135, operation for operation (oracle: orc_log2hot).  tbl: the Log2Hot table in global memory ({log2 midpoint,
// 1 / (2 midpoint)} per bucket), entry0: its entry 0 as the REFERENCE has it (SRVectMath.cpp:31,42 -- the table's own is re-seated
// for the division-free log2hot, hip_engine.cpp).
__device__ __forceinline__ double log2hot_ref(double x, const double *tbl, double entry0) {
  const uint64_t ux = d2u(x);
  const double z = u2d((ux & ~kExpMaskUp) | kExp0Up);          // :88-89
  const int32_t high32 = (int32_t)(ux >> 32);                  // :92-94
  const int32_t normExps = (high32 >> 20) - 1023;              // :97-98
  const int32_t idx = (high32 >> 10) & 1023;                   // :101-102
  const double y = idx == 0 ? entry0 : tbl[2 * idx];           // :105-106
  const double m = u2d((1ULL << 41) | (d2u(z) & ~((1ULL << 42) - 1)));   // :108
  const double t = div_fast(z - m, z + m);                     // :111-114 (the exact quotient: div_nr, pqa_device.h)
  const double t2 = t * t;                                     // :115
  const double t3 = t * t2;                                    // :117
  const double terms01 = fma(1.0 / 3, t3, t);                  // :118
  const double log2z = fma(terms01, 2.8853900817779268147198493620038, y);   // :122
  return log2z + (double)normExps;                             // :131-133
}

// the wave's maximum in every lane (the DPP / permlane steps of wave_sum: six ds_bpermute round trips otherwise)
__device__ __forceinline__ double wave_max_d(double v) {
  v = fmax(v, mov_dpp<kDppXor1>(v));
  v = fmax(v, mov_dpp<kDppXor2>(v));
  v = fmax(v, mov_dpp<kDppHalfMirror>(v));
  v = fmax(v, mov_dpp<kDppMirror>(v));
  Pair p = swap16(v);
  v = fmax(p.a, p.b);
  p = swap32(v);
  return fmax(p.a, p.b);
}

}  // namespace pqa
