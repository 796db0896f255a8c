"""A SECOND statement of the priority formula, written from the reference's sources and SURVEY.md Appendix A
-- NOT from oracle/pqa_oracle.c -- in plain Python floats: one IEEE-754 operation per Python operation, vector lanes emulated one
by one.  It exists so that the committed fixtures (tests/golden/*.npz, produced by the C oracle) are not single-sourced: a
misreading of PqaCore/CEEvalQsSubtaskConsider.cpp:41-217 shared by the oracle and by the kernels would pass every test; two
independent readings that agree BIT FOR BIT on every priority of the small fixtures make that unlikely.
tests/test_oracle.py::test_independent_restatement runs it (CPU only; a few seconds).

What is restated, with the reference lines each function follows:
  * Log2Hot              SRPlatform/Interface/SRVectMath.h:87-135, table SRPlatform/SRVectMath.cpp:30-44
  * Kahan4 (4 lanes)     SRPlatform/Interface/SRAccumVectDbl256.h:40-55 (Add, Add-at), :62-92 (PreciseSum), :94-133 (PairSum)
  * Kahan1               SRPlatform/Interface/SRAccumulator.h:15-39
  * priority of one question   PqaCore/CEEvalQsSubtaskConsider.cpp:59-207; velocity component :22-32
Conventions shared with the oracle because the reference leaves them to its platform: std::pow(x, 9) and
std::pow(x, -2) as integer powers (the author's own TODO at :206), std::log / std::exp2 / std::log2 from this machine's libm.
fused multiply-add (the three _mm256_fmadd_pd of Log2Hot) is computed exactly in rational arithmetic and rounded once."""
from __future__ import annotations

import ctypes
import ctypes.util
import math
import struct
from fractions import Fraction

_libm = ctypes.CDLL(ctypes.util.find_library("m") or "libm.so.6")
_libm.exp2.restype = ctypes.c_double
_libm.exp2.argtypes = [ctypes.c_double]


def _bits(x: float) -> int:
    return struct.unpack("<Q", struct.pack("<d", x))[0]


def _from_bits(u: int) -> float:
    return struct.unpack("<d", struct.pack("<Q", u & 0xFFFFFFFFFFFFFFFF))[0]


def fma(a: float, b: float, c: float) -> float:
    """a * b + c with ONE rounding (round to nearest even): exact in rationals, then float()."""
    return float(Fraction(a) * Fraction(b) + Fraction(c))


# ---- Log2Hot -----------------------------------------------------------------------------------------------------------------
TBL_BITS = 10                                   # SRVectMath::_cnLog2TblBits
MANT_BITS = 52
EXP0_UP = 0x3FF0000000000000                    # SRNumTraits<double>::_cExponent0Up
EXP_MASK_UP = 0x7FF0000000000000


def _make_table():
    tbl = []
    for i in range(1 << TBL_BITS):              # SRVectMath.cpp:33-41
        iz = EXP0_UP | (i << (MANT_BITS - TBL_BITS))
        izp = iz | (1 << (MANT_BITS - TBL_BITS - 1))
        tbl.append(math.log2(_from_bits(izp)))
    tbl[0] *= 9.9999999999999927e-01            # :42 "so that log2(1) <= 0"
    return tbl


_TABLE = _make_table()
_C2DIVLN2 = 2.8853900817779268147198493620038   # SRVectMath.cpp:10
_COEFF1 = 1.0 / 3


def log2hot(x: float) -> float:
    ux = _bits(x)
    z = _from_bits((ux & ~EXP_MASK_UP) | EXP0_UP)                         # SRVectMath.h:88-89
    high32 = (ux >> 32) & 0xFFFFFFFF
    if high32 & 0x80000000:
        high32 -= 1 << 32                                                # (arithmetic shifts of a signed word: :96, :101)
    norm_exp = (high32 >> (MANT_BITS - 32)) - 1023                       # :96-97
    index = (high32 >> (MANT_BITS - 32 - TBL_BITS)) & ((1 << TBL_BITS) - 1)   # :100-101
    y = _TABLE[index]                                                    # :104-105
    exp2_y = _from_bits((1 << (MANT_BITS - TBL_BITS - 1)) | (_bits(z) & ~((1 << (MANT_BITS - TBL_BITS)) - 1)))   # :107
    t = (z - exp2_y) / (z + exp2_y)                                      # :110-113
    t2 = t * t                                                           # :114
    t3 = t * t2                                                          # :116
    terms01 = fma(_COEFF1, t3, t)                                        # :117
    log2_z = fma(terms01, _C2DIVLN2, y)                                  # :121 (terms01, not terms012)
    return log2_z + float(norm_exp)                                      # :130-132


# ---- the Kahan accumulators ---------------------------------------------------------------------------------------------------
class Kahan1:                                    # SRAccumulator.h
    def __init__(self, v: float):
        self.sum, self.corr = v, 0.0

    def add(self, v: float):
        y = v - self.corr
        t = self.sum + y
        self.corr = (t - self.sum) - y
        self.sum = t

    def neg(self):
        self.sum, self.corr = -self.sum, -self.corr

    def get(self) -> float:
        return self.sum - self.corr


class Kahan4:                                    # SRAccumVectDbl256.h
    def __init__(self):
        self.sum, self.corr = [0.0] * 4, [0.0] * 4

    def add(self, v4):                           # :40-46, lane by lane
        for c in range(4):
            self.add_at(c, v4[c])

    def add_at(self, c: int, v: float):          # :48-54
        y = v - self.corr[c]
        t = self.sum[c] + y
        self.corr[c] = (t - self.sum[c]) - y
        self.sum[c] = t

    def precise_sum(self) -> float:              # :83-91
        a = Kahan1(self.corr[3])
        for i in (2, 1, 0):
            a.add(self.corr[i])
        a.neg()
        for i in (3, 2, 1, 0):
            a.add(self.sum[i])
        return a.get()

    def pair_sum(self, fellow: "Kahan4"):        # :113-132: two Kahan chains side by side (this one, the fellow)
        out = []
        for acc in (self, fellow):
            s, c = acc.corr[3], 0.0
            for i in (2, 1, 0):
                y = acc.corr[i] - c
                t = s + y
                c = (t - s) - y
                s = t
            s, c = -s, -c
            for i in (3, 2, 1, 0):
                y = acc.sum[i] - c
                t = s + y
                c = (t - s) - y
                s = t
            out.append(s - c)
        return out[0], out[1]


# ---- one question --------------------------------------------------------------------------------------------------------------
LN_SQRT2 = 0.34657359027997265470861606072909   # SRMath::_cLnSqrt2 (= _cLnMaxV)
LN0_STAB = -746.0                                # _cLn0Stab


def velocity_component(v: float, n_targets: int) -> float:               # CEEvalQsSubtaskConsider.cpp:22-32
    ln_v = LN0_STAB if v == 0 else math.log(v)
    pow_t = float(n_targets) * n_targets
    return 1 / (LN_SQRT2 - ln_v + LN_SQRT2 / pow_t)


def question_priority(A_q, D_q, prior, gap_t, K: int, T: int, n_valid: int) -> float:
    """A_q[k][t], D_q[t], prior[t], gap_t[t] over the padded target range (multiples of four; padding: gaps)."""
    n_vect = (T + 3) >> 2
    inv_d = [0.0] * (4 * n_vect)
    post = [0.0] * (4 * n_vect)
    acc_tot_w = Kahan1(0.0)                                              # :60
    acc_l = Kahan4()                                                     # :61
    W, H, V2 = [0.0] * K, [0.0] * K, [0.0] * K
    for k in range(K):
        acc = Kahan4()                                                   # :63
        for j in range(n_vect):                                          # :66-87
            lh = [0.0] * 4
            for c in range(4):
                t = 4 * j + c
                if k == 0:
                    inv_d[t] = 0.0 if gap_t[t] else 1.0 / D_q[t]         # :72-76 (andnot: an exact +0 in gap lanes)
                pr_given = A_q[k][t] * inv_d[t]                          # :81
                lh[c] = 0.0 if gap_t[t] else pr_given * prior[t]         # :82
                post[t] = lh[c]                                          # :84
            acc.add(lh)                                                  # :86
        W[k] = acc.precise_sum()                                         # :88
        acc_tot_w.add(W[k])                                              # :89
        inv_w = 1.0 / W[k]                                               # :91
        acc_h, acc_v = Kahan4(), Kahan4()                                # :93-94
        for j in range(n_vect):                                          # :95-128
            hv, lv, vv = [0.0] * 4, [0.0] * 4, [0.0] * 4
            for c in range(4):
                t = 4 * j + c
                p = post[t] * inv_w                                      # :97
                pri = 0.0 if gap_t[t] else prior[t]                      # :103
                l2 = 0.0 if gap_t[t] else log2hot(p)                     # :106
                hv[c] = p * l2                                           # :113
                lv[c] = 0.0 if gap_t[t] else (inv_d[t] * inv_d[t]) / l2  # :116-117
                d = p - pri                                              # :119
                vv[c] = d * d                                            # :126
            acc_h.add(hv)                                                # :114
            acc_l.add(lv)                                                # :117
            acc_v.add(vv)                                                # :127
        sum_h, sum_v = acc_h.pair_sum(acc_v)                             # :130
        H[k], V2[k] = -sum_h, sum_v                                      # :131-132
    tot_w = acc_tot_w.get()                                              # :134
    acc_avg_h, acc_avg_v = Kahan4(), Kahan4()                            # :139-140
    n_vectorized = (K >> 2) << 2                                         # :141-142
    for k0 in range(0, n_vectorized, 4):                                 # :148-159
        acc_avg_h.add([W[k0 + c] * H[k0 + c] for c in range(4)])
        acc_avg_v.add([W[k0 + c] * math.sqrt(V2[k0 + c]) for c in range(4)])
    for k in range(n_vectorized, K):                                     # :163-172
        velocity = math.sqrt(V2[k])
        acc_avg_h.add_at(k - n_vectorized, W[k] * H[k])
        acc_avg_v.add_at(k - n_vectorized, W[k] * velocity)
    avg_h, avg_v = acc_avg_h.pair_sum(acc_avg_v)                         # :175
    avg_h, avg_v = avg_h / tot_w, avg_v / tot_w                          # :176-177
    n_expected = _libm.exp2(avg_h)                                       # :181
    v_comp = velocity_component(avg_v, n_valid + 1)                      # :191
    lack = -acc_l.precise_sum()                                          # :201
    v2 = v_comp * v_comp                                                 # :207 with integer powers (:206)
    v4 = v2 * v2
    v8 = v4 * v4
    v9 = v8 * v_comp
    return lack * v9 * (1.0 / (n_expected * n_expected))


def priorities(A, D, prior, tgaps, qgaps, asked, K: int, Q: int, T: int):
    """priority[q] for every question of the cube A[q][k][t] / D[q][t] (numpy arrays or nested lists), 0 for gap / asked ones."""
    n4 = 4 * ((T + 3) >> 2)
    gap_t = [False] * n4
    for t in range(T, n4):
        gap_t[t] = True                                                  # (GapTracker: positions past the size are gaps)
    for t in tgaps:
        gap_t[t] = True
    pr = [float(prior[t]) if t < T else 0.0 for t in range(n4)]
    n_valid = T - len(set(tgaps))                                        # CpuEngine.cpp:352
    out = [0.0] * Q
    for q in range(Q):
        if q in qgaps or q in asked:                                     # CEEvalQsSubtaskConsider.cpp:54-58
            continue
        A_q = [[float(A[q][k][t]) if t < T else 0.0 for t in range(n4)] for k in range(K)]
        D_q = [float(D[q][t]) if t < T else 1.0 for t in range(n4)]
        out[q] = question_priority(A_q, D_q, pr, gap_t, K, T, n_valid)
    return out
