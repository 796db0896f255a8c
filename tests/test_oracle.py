"""CPU tests of the oracle (oracle/pqa_oracle.c): the reference's own known-answer tests for the pieces it pins, internal
consistency (AVX2 port == scalar restatement), an independent mathematical definition, selection logic, and the
committed golden fixtures."""
import ctypes
import json
import os

import numpy as np
import pytest

import cases
import orclib
from probqa_amd import synth

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


# ---- reference SRPlatformTests/SRVectMathTest.cpp:45-103 (Log2Hot) ---------------------------------------------------
def test_log2hot_reference_assertions(oracle_lib):
    req_prec = 3e-12
    f = oracle_lib.orc_log2hot
    vals = [f(x) for x in (1.0, 0.99999, 0.9999, 0.999)]          # :49-56
    assert all(v <= 0 for v in vals) and vals[0] >= -req_prec
    assert all(f(x) <= -1022.5 for x in (0.0, -1.0, -2.0, -3.0))  # :57-63
    # :65-83 mantissa x exponent sweep (every 16th point of the reference's 2^22 to keep the CPU suite short)
    n_mant_bits, exp_range = 22, 2046
    for i in range(0, 1 << n_mant_bits, 16):
        cur_exp = (i % exp_range) + 1
        bits = (cur_exp << 52) | (i << (52 - n_mant_bits))
        x = np.array([bits], dtype=np.uint64).view(np.float64)[0]
        if x == 1.0:
            continue
        exp = np.log2(x)
        assert abs(f(float(x)) - exp) <= abs(exp) * req_prec
    rng = np.random.default_rng(1)                                # :85-103 random uint64 -> double
    for u in rng.integers(1, 2**63, size=20000, dtype=np.uint64):
        x = float(u)
        exp = np.log2(x)
        assert abs(f(x) - exp) <= abs(exp) * req_prec


def test_log2hot_special_values(oracle_lib):
    f = oracle_lib.orc_log2hot
    assert f(0.0) == -1023.0 and f(2.0 ** -1022) == -1022.0 and f(0.5) == -1.0
    assert -1e-18 < f(1.0) < 0  # strictly negative at 1 (table[0] correction, SRVectMath.cpp:31,42)
    tbl = np.ctypeslib.as_array(oracle_lib.orc_log2hot_table(), shape=(1024,))
    assert np.all(np.diff(tbl) > 0) and 0 < tbl[0] < 1e-3 and tbl[-1] < 1


# ---- reference SRPlatformTests/SRAccumulatorTest.cpp:21-35 -----------------------------------------------------------
def test_kahan4_reference_known_answers(oracle_lib):
    va1, va2 = orclib.OrcKahan4(), orclib.OrcKahan4()
    va1.sum[:] = [16, 32, 64, 128]
    va1.corr[:] = [1, 2, 4, 8]
    va2.sum[:] = [4096, 8192, 16384, 32768]
    va2.corr[:] = [256, 512, 1024, 2048]
    assert oracle_lib.orc_k4_precise_sum(ctypes.byref(va1)) == 225
    assert oracle_lib.orc_k4_full_sum(ctypes.byref(va1)) == 225
    s2 = ctypes.c_double()
    assert oracle_lib.orc_k4_pair_sum(ctypes.byref(va1), ctypes.byref(va2), ctypes.byref(s2)) == 225
    assert s2.value == 57600
    assert oracle_lib.orc_k4_full_sum(ctypes.byref(va2)) == 57600
    assert oracle_lib.orc_k4_precise_sum(ctypes.byref(va2)) == 57600


def test_calc_split(oracle_lib):
    # SRPoolRunner::CalcSplit (SRPoolRunner.h:96-110)
    b = (ctypes.c_int64 * 8)()
    assert oracle_lib.orc_calc_split(10, 4, b) == 4 and list(b)[:4] == [3, 6, 8, 10]
    assert oracle_lib.orc_calc_split(3, 8, b) == 3 and list(b)[:3] == [1, 2, 3]


# ---- reference PqaCoreTests/Dimensions.cpp:60-77: fresh KB values ------------------------------------------------------
def test_fresh_kb_values():
    o = orclib.Oracle(4, 5, 7, 0.3)
    assert (o.A[:, :, :7] == 0.3 * 0.3).all() and (o.D[:, :7] == 0.3 * 0.3 * 4).all() and (o.B[:7] == 0.3).all()
    o.start_quiz(16)
    assert abs(o.priors().sum() - 1) < 1e-15


# ---- internal consistency ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("case", cases.small_cases(), ids=lambda c: c.name)
def test_avx2_port_is_bit_identical_to_scalar(case):
    if not orclib.lib().orc_have_avx2():
        pytest.skip("no AVX2")
    o = case.make_oracle()
    o.start_quiz(16)
    for q, a in case.answers:
        o.record_answer(q, a, 15)
    run, pri = o.eval(32)
    for threads in (1, 4):
        run2, pri2 = o.eval_avx2(threads, 32)
        assert np.array_equal(pri, pri2) and np.array_equal(run, run2)


def math_priority(A, D, prior, n_valid, gaps=()):
    """Independent statement of the priority (README / CEEvalQsSubtaskConsider.cpp:207) in plain numpy with true log2 and
    naive sums, in long double where it is cheap."""
    ld = np.longdouble
    keep = np.ones(A.shape[-1], bool)
    keep[list(gaps)] = False
    A, D, prior = A[..., keep].astype(ld), D[..., keep].astype(ld), prior[keep].astype(ld)
    invD = 1 / D
    P = A * invD[:, None, :] * prior[None, None, :]
    W = P.sum(-1)
    post = P / W[..., None]
    l2 = np.log2(post)
    H = -(post * l2).sum(-1)
    V = np.sqrt(((post - prior) ** 2).sum(-1))
    lack = -((invD[:, None, :] ** 2) / l2).sum((1, 2))
    totW = W.sum(-1)
    avgH, avgV = (W * H).sum(-1) / totW, (W * V).sum(-1) / totW
    c = ld(0.34657359027997265470861606072909)
    vcomp = 1 / (c - np.log(avgV) + c / ld((n_valid + 1) ** 2))
    return (lack * vcomp ** 9 / np.exp2(avgH) ** 2).astype(np.float64)


@pytest.mark.parametrize("case", cases.small_cases(), ids=lambda c: c.name)
def test_priority_matches_mathematical_definition(case):
    o = case.make_oracle()
    o.start_quiz(16)
    for q, a in case.answers:
        o.record_answer(q, a, 15)
    _, pri = o.eval(128)
    A, D, _ = case.kb()
    mp = math_priority(A, D, o.priors(), case.T - len(case.tgaps), case.tgaps)
    live = pri != 0
    assert set(np.nonzero(~live)[0]) == set(case.qgaps) | {q for q, _ in case.answers}
    # the restatement computes the intended quantity; conditioning grows as a posterior approaches 1
    assert cases.rel_err(pri[live], mp[live]).max() < 1e-10


def test_posteriors_match_bayes_rule():
    case = cases.small_cases()[4]
    o = case.make_oracle()
    A, D, B = case.kb()
    o.start_quiz(16)
    p = B / B.sum()
    assert np.allclose(o.priors(), p, rtol=1e-14)
    for q, a in case.answers:
        o.record_answer(q, a, 15)
        p = p * A[q, a] / D[q]
        p /= p.sum()
        assert np.allclose(o.priors(), p, rtol=1e-13)
    # ResumeQuiz recomputes the same posterior through the mantissa/exponent split (CEUpdatePriorsSubtaskMul)
    o2 = case.make_oracle()
    assert o2.resume_quiz(case.answers, 16) == 0
    assert np.allclose(o2.priors(), p, rtol=1e-13)


def test_resume_bug_compat_switch():
    # CEUpdatePriorsSubtaskMul.cpp:53 reads vector 0 of vB for every target vector: identical for a uniform vB,
    # different once vB varies beyond the first four targets.
    case = cases.small_cases()[1]
    o1, o2 = case.make_oracle(), case.make_oracle()
    o1.resume_quiz([(10, 4), (30, 0)], 16, False)
    o2.resume_quiz([(10, 4), (30, 0)], 16, True)
    assert not np.array_equal(o1.priors(), o2.priors())
    o1.B[:] = 0.7
    o2.B[:] = 0.7
    o1.resume_quiz([(10, 4), (30, 0)], 16, False)
    o2.resume_quiz([(10, 4), (30, 0)], 16, True)
    assert np.array_equal(o1.priors(), o2.priors())


def test_resume_all_gaps_underflows():
    o = orclib.Oracle(3, 4, 6, 1.0)
    o.set_target_gaps(range(6))
    assert o.resume_quiz([(0, 1)], 4) == 16  # PqaErrorCode::I64Underflow (CpuEngine.cpp:317-321)


# ---- selection (CpuEngine.cpp:362-406, BaseEngine.cpp:60-124) ------------------------------------------------------------
def test_sampled_selection_follows_the_running_sums():
    case = cases.small_cases()[1]
    o = case.make_oracle()
    o.start_quiz(16)
    o.record_answer(10, 4, 15)
    n_sub = 8
    run, pri = o.eval(n_sub)
    bounds = (ctypes.c_int64 * n_sub)()
    n = orclib.lib().orc_calc_split(case.Q, n_sub, bounds)
    tot = sum(run[bounds[i] - 1] for i in range(n))
    cum = np.cumsum(pri)
    assert abs(cum[-1] - tot) <= 1e-12 * tot
    rng = np.random.default_rng(3)
    for rnd in [0, 2**64 - 1] + [int(x) for x in rng.integers(0, 2**64 - 1, size=200, dtype=np.uint64)]:
        sel = o.select_sampled(run, n_sub, rnd)
        assert pri[sel] > 0  # never a gap / asked question
        target = tot * rnd / (2**64 - 1)
        lo = cum[sel - 1] if sel > 0 else 0.0
        assert lo - 1e-9 * tot <= target <= cum[sel] + 1e-9 * tot or rnd == 2**64 - 1


def test_find_nearest_question_matches_brute_force():
    Q = 300
    o = orclib.Oracle(3, Q, 8, 1.0)
    rng = np.random.default_rng(9)
    avail = rng.random(Q) < 0.05
    avail[[0, 63, 64, 200]] = True
    for q in range(Q):
        if not avail[q]:
            if rng.random() < 0.5:
                o.set_question_gaps([q])
            else:
                orclib.lib()  # asked bits live in the quiz
                o.quiz.contents.asked[q >> 3] |= 1 << (q & 7)
    cand = np.nonzero(avail)[0]
    for mid in range(Q):
        got = o.find_nearest(mid)
        d = np.abs(cand - mid)
        assert got in cand and abs(got - mid) <= d.min() + 64  # pack-granular search: nearest within the 64-bit pack order
        if (cand >> 6 == mid >> 6).any():  # same pack: exact nearest, ties to the lower index
            same = cand[cand >> 6 == mid >> 6]
            ds = np.abs(same - mid)
            best = same[ds == ds.min()].min()
            assert got == best
    o2 = orclib.Oracle(3, 70, 8, 1.0)
    o2.set_question_gaps(range(70))
    assert o2.find_nearest(35) == -1


def test_argmax_lowest_index_on_ties():
    o = orclib.Oracle(3, 9, 8, 1.0)
    o.start_quiz(4)
    _, pri = o.eval(4)
    assert np.ptp(pri) == 0 and o.select_argmax(pri) == 0
    o.set_question_gaps([0])
    _, pri = o.eval(4)
    assert o.select_argmax(pri) == 1


# ---- golden fixtures (generated by tests/golden/make_golden.py from this oracle; they pin it against regressions) ----------
@pytest.mark.parametrize("name", sorted(f[:-5] for f in os.listdir(GOLDEN) if f.endswith(".json")) if os.path.isdir(GOLDEN) else [])
def test_oracle_reproduces_golden_fixture(name):
    import sys
    sys.path.insert(0, GOLDEN)
    import make_golden

    meta = json.load(open(os.path.join(GOLDEN, name + ".json")))
    data = np.load(os.path.join(GOLDEN, name + ".npz"))
    got = make_golden.run_case(make_golden.case_from_meta(meta))
    for key in data.files:
        assert np.array_equal(got[key], data[key]), (name, key)


def test_training_pairing_rules():
    """CETrainOperation::Perform2's three cases and the bucket order of CpuEngine::TrainSpec, against hand-written algebra
    (reference PqaCore/CETrainOperation.cpp:32-83, CETrainSubtaskDistrib.h:46-52, CETrainSubtaskAdd.cpp:17-38)."""
    K, Q, T, init, b = 4, 6, 5, 0.3, 0.7
    A0, D0 = init * init, init * init * K

    def fresh():
        return orclib.Oracle(K, Q, T, init)

    def step(a2, amount):  # one step of `amount` on a cell holding a2
        return a2 + (np.sqrt(a2) * (2 * amount) + amount * amount)

    # same question, same answer: ONE step of 2b (4ab + 4b^2), D gets the same addend
    o = fresh()
    o.record_quiz_target(2, b, aqs=[(1, 3), (1, 3)])
    add = np.sqrt(A0) * (4 * b) + 4 * (b * b)
    assert o.A[1, 3, 2] == A0 + add and o.D[1, 2] == D0 + add
    assert abs(o.A[1, 3, 2] - (init + 2 * b) ** 2) < 1e-14 and abs(o.B[2] - (init + b)) < 1e-15
    # same question, different answers: each cell its own addend, D gets TWICE THE FIRST one's
    o = fresh()
    o.A[1, 0, 2], o.A[1, 3, 2] = 4.0, 9.0
    o.record_quiz_target(2, b, aqs=[(1, 0), (1, 3)])
    add1, add2 = 2.0 * (2 * b) + b * b, 3.0 * (2 * b) + b * b
    assert o.A[1, 0, 2] == 4.0 + add1 and o.A[1, 3, 2] == 9.0 + add2 and o.D[1, 2] == D0 + (add1 + add1)
    # different questions: two independent Perform1 steps; an odd last entry is a Perform1
    o = fresh()
    o.record_quiz_target(0, b, aqs=[(0, 1), (4, 2), (5, 0)])
    for q, a in ((0, 1), (4, 2), (5, 0)):
        assert o.A[q, a, 0] == step(A0, b) and o.D[q, 0] == D0 + (step(A0, b) - A0)
    # RecordQuizTarget pairs by POSITION: (1,3),(2,0) | (1,3): question 1 is stepped twice by b, not once by 2b
    o = fresh()
    o.record_quiz_target(3, b, aqs=[(1, 3), (2, 0), (1, 3)])
    assert o.A[1, 3, 3] == step(step(A0, b), b)
    # Train buckets by question % workers, each bucket consumed newest first in pairs: with 4 workers the entries of question 1
    # (positions 0, 2, 3; all in bucket 1 together with question 5 at position 1) are consumed as (3,2) | (1,0):
    # Perform2(same question 1, same answer) = one 2b step, then Perform2(question 5, question 1) = one b step each
    o = fresh()
    o.train([(1, 3), (5, 0), (1, 3), (1, 3)], 4, b, n_workers=4)
    assert o.A[1, 3, 4] == step(A0 + add, b) and o.A[5, 0, 4] == step(A0, b)
    # with one worker per question nothing pairs up across questions, the pairing within question 1 is (3,2) | (0)
    o2 = fresh()
    o2.train([(1, 3), (5, 0), (1, 3), (1, 3)], 4, b, n_workers=64)
    assert o2.A[1, 3, 4] == o.A[1, 3, 4] and o2.A[5, 0, 4] == o.A[5, 0, 4]


def test_reference_sum_is_not_the_correctly_rounded_sum():
    """Why the sweep's handling of rows at the pole of the lack term (eval_kernels.hip: pole_fix) walks the row in the
    REFERENCE'S ORDER instead of summing it well: the reference's W_k -- four serial Kahan lanes down the row and PreciseSum
    (SRAccumVectDbl256.h:40-46, :62-92) -- differs from the correctly rounded sum by a unit in the last place in 10 - 30 % of rows,
    also where one likelihood carries all but a sliver of the sum (a lane that meets the large element after smaller ones rounds
    its own compensation).  A better sum on the device would therefore still miss the reference's last place that often."""
    import math

    def reference_sum(x):
        n4 = (len(x) + 3) // 4 * 4
        x = np.concatenate([x, np.zeros(n4 - len(x))])
        s, c = [0.0] * 4, [0.0] * 4
        for j in range(0, n4, 4):                       # SRAccumVectDbl256::Add
            for lane in range(4):
                y = x[j + lane] - c[lane]
                t = s[lane] + y
                c[lane] = (t - s[lane]) - y
                s[lane] = t
        ks, kc = c[3], 0.0                              # PreciseSum: the corrections, negated, then the sums
        for i in (2, 1, 0):
            y = c[i] - kc
            t = ks + y
            kc = (t - ks) - y
            ks = t
        ks, kc = -ks, -kc
        for i in (3, 2, 1, 0):
            y = s[i] - kc
            t = ks + y
            kc = (t - ks) - y
            ks = t
        return ks - kc

    rng = np.random.default_rng(20260929)
    dominated = ordinary = 0
    for n in (4, 10, 101):
        for trial in range(300):
            x = np.concatenate([[rng.random() + 0.5], rng.random(n - 1) * 1e-7 / n])     # the other targets: far below 2^-17 of the row
            rng.shuffle(x)
            r, e = reference_sum(x), math.fsum(x)
            assert abs(r - e) <= np.spacing(e)
            dominated += r != e
            y = rng.random(n)
            ordinary += reference_sum(y) != math.fsum(y)
    assert dominated > 30 and ordinary > 30


@pytest.mark.parametrize("name", ["tiny_8x3x12", "gaps_37x5x101", "ragged_50x4x67"])
def test_independent_restatement(name):
    """The committed priorities (tests/golden/*.npz, from oracle/pqa_oracle.c) against a SECOND restatement of
    CEEvalQsSubtaskConsider.cpp:59-207 written from the reference's sources in plain Python (tests/golden/independent_a1.py:
    lane-emulating Kahan accumulators, Log2Hot with exactly rounded fused multiply-adds): BIT FOR BIT, every question, every
    step.  The fixtures are then not single-sourced: a misreading of the reference would have to be made twice."""
    import json
    import os
    import sys

    gdir = os.path.join(os.path.dirname(__file__), "golden")
    sys.path.insert(0, gdir)
    import independent_a1
    import make_golden

    meta = json.load(open(os.path.join(gdir, name + ".json")))
    gold = np.load(os.path.join(gdir, name + ".npz"))
    case = make_golden.case_from_meta(meta)
    A, D, _ = case.kb()
    asked = []
    for step in range(len(case.answers) + 1):
        mine = independent_a1.priorities(A, D, gold[f"priors_{step}"], case.tgaps, case.qgaps, asked, case.K, case.Q, case.T)
        want = gold[f"priority_{step}"]
        bad = [q for q in range(case.Q) if np.float64(mine[q]).tobytes() != np.float64(want[q]).tobytes()]
        assert not bad, (name, step, bad[:5], [(mine[q], want[q]) for q in bad[:3]])
        if step < len(case.answers):
            asked.append(case.answers[step][0])


# ---- f2: ListTopTargets (PqaCore/CEListTopTargetsAlgorithm.cpp:30-95, CEHeapifyPriorsSubtaskMake.cpp:42-88) ------------
@pytest.fixture(scope="module")
def std_heap(tmp_path_factory):
    """libstdc++'s std::make_heap / std::pop_heap behind a C ABI (tests/std_heap_check.cpp, compiled here)."""
    import subprocess

    so = tmp_path_factory.mktemp("stdheap") / "libstdheap.so"
    subprocess.check_call(["g++", "-O1", "-shared", "-fPIC", "-o", str(so), os.path.join(os.path.dirname(__file__), "std_heap_check.cpp")])
    L = ctypes.CDLL(str(so))
    for name in ("std_heap_make", "std_heap_pop"):
        getattr(L, name).restype = None
        getattr(L, name).argtypes = [ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int64), ctypes.c_int64]
    return L


def _heap_call(fn, prob, ids):
    p, i = np.array(prob, dtype=np.float64), np.array(ids, dtype=np.int64)
    fn(p.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), i.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)), len(p))
    return p, i


def test_restated_heap_steps_equal_libstdcxx(oracle_lib, std_heap):
    """The oracle's make_heap / pop_heap move every record exactly as the C++ library's do, ties included (few distinct values)."""
    rng = np.random.default_rng(11)
    for n in list(range(0, 40)) + [63, 64, 65, 127, 128, 1000, 6250]:
        for distinct in (1, 2, 3, 7, 10**9):
            prob = rng.integers(1, distinct + 1, size=n).astype(np.float64) / 8.0
            ids = np.arange(n, dtype=np.int64)
            mine = _heap_call(oracle_lib.orc_heap_make, prob, ids)
            std = _heap_call(std_heap.std_heap_make, prob, ids)
            assert np.array_equal(mine[0], std[0]) and np.array_equal(mine[1], std[1]), (n, distinct)
            p, i = mine
            for m in range(n, max(n - 12, 0), -1):           # a dozen pops of the heap just made
                a = _heap_call(oracle_lib.orc_heap_pop, p[:m], i[:m])
                b = _heap_call(std_heap.std_heap_pop, p[:m], i[:m])
                assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]), (n, distinct, m)
                p[:m], i[:m] = a


def _listing_by_definition(orc, max_count, n_workers, std_heap):
    """RunHeapifyBased restated a second time, in Python over libstdc++'s heap calls (not over the oracle's)."""
    T = orc.T
    gaps = np.unpackbits(np.ctypeslib.as_array(orc.kb.contents.targetGaps, shape=((orc.ldT + 7) // 8,)), bitorder="little")
    bounds = (ctypes.c_int64 * n_workers)()
    n_sub = orc.L.orc_calc_split(T, n_workers, bounds)
    pieces = []
    for s in range(n_sub):
        first, limit = (0 if s == 0 else bounds[s - 1]), bounds[s]
        sel = [(orc.mants[t], t) for t in range(first, limit) if not gaps[t] and not (orc.mants[t] <= 0)]
        p, i = _heap_call(std_heap.std_heap_make, [x[0] for x in sel], [x[1] for x in sel])
        pieces.append([list(p), list(i)])
    head = [(pieces[s][0][0], s) for s in range(n_sub) if pieces[s][0]]
    hp, hs = _heap_call(std_heap.std_heap_make, [h[0] for h in head], [h[1] for h in head])
    hp, hs = list(hp), list(hs)
    out = []
    for _ in range(max_count):
        if not hp:
            break
        s = int(hs[0])
        p, i = pieces[s]
        out.append((int(i[0]), hp[0]))
        if len(p) == 1:
            a, b = _heap_call(std_heap.std_heap_pop, hp, hs)
            hp, hs = list(a[:-1]), list(b[:-1])
            continue
        a, b = _heap_call(std_heap.std_heap_pop, p, i)
        pieces[s] = [list(a[:-1]), list(b[:-1])]
        hp[0] = pieces[s][0][0]
        cur = 0                                              # SRHeapHelper::Down, SRPlatform/Interface/SRHeap.h:16-39
        while True:
            c1 = 2 * cur + 1
            if c1 >= len(hp):
                break
            c2 = c1 + 1
            if c2 >= len(hp):
                if hp[cur] < hp[c1]:
                    hp[cur], hp[c1], hs[cur], hs[c1] = hp[c1], hp[cur], hs[c1], hs[cur]
                break
            hi = c1 if hp[c2] < hp[c1] else c2
            if not (hp[cur] < hp[hi]):
                break
            hp[cur], hp[hi], hs[cur], hs[hi] = hp[hi], hp[cur], hs[hi], hs[cur]
            cur = hi
    return out


def test_list_top_targets_drops_gaps_and_nonpositive_and_orders_descending(std_heap):
    K, Q, T = 3, 4, 203
    orc = orclib.Oracle(K, Q, T, 1.0)
    rng = np.random.default_rng(5)
    orc.set_target_gaps([0, 17, 202])
    orc.start_quiz(16)
    orc.mants[:T] = rng.random(T)
    orc.mants[[5, 6, 100]] = 0.0              # exact zeros: never listed (CEHeapifyPriorsSubtaskMake.cpp:47)
    orc.mants[7] = -0.25                      # (cannot arise; the filter is `prob <= 0`)
    live = [t for t in range(T) if t not in (0, 17, 202, 5, 6, 100, 7)]
    want = sorted(live, key=lambda t: -orc.mants[t])
    for n_workers in (1, 2, 7, 16, 64, 300):
        for mc in (1, 3, 10, len(live), T + 5):
            got = orc.list_top_targets(mc, n_workers)
            assert [t for t, _ in got] == want[:mc], (n_workers, mc)             # distinct values: the order is the sort's
            assert [p for _, p in got] == [orc.mants[t] for t in want[:mc]]
            assert got == _listing_by_definition(orc, mc, n_workers, std_heap)
    # nothing positive: an empty listing
    orc.mants[:T] = 0.0
    assert orc.list_top_targets(10, 16) == []
    orc.close()


def test_list_top_targets_tie_order_is_the_heaps(std_heap):
    """Equal probabilities come out in the order the piece heaps and the head heap give: a function of the pool size, not of
    the target index -- the second restatement (over libstdc++'s heap calls) agrees record for record."""
    K, Q, T = 2, 3, 157
    orc = orclib.Oracle(K, Q, T, 0.5)
    orc.set_target_gaps([3, 80])
    orc.start_quiz(16)                        # a fresh KB: every live target holds the same probability
    orders = {}
    for n_workers in (1, 4, 16):
        got = orc.list_top_targets(12, n_workers)
        assert len(got) == 12 and len({p for _, p in got}) == 1 and len({t for t, _ in got}) == 12
        assert got == _listing_by_definition(orc, 12, n_workers, std_heap)
        orders[n_workers] = [t for t, _ in got]
    assert orders[1] != orders[16] and orders[1] != sorted(orders[1])           # neither index order nor pool-independent
    rng = np.random.default_rng(9)            # a few distinct values, many ties each
    orc.mants[:T] = rng.integers(0, 5, size=T) / 16.0
    for n_workers in (1, 3, 16, 40):
        for mc in (1, 5, 40, 200):
            got = orc.list_top_targets(mc, n_workers)
            assert got == _listing_by_definition(orc, mc, n_workers, std_heap), (n_workers, mc)
            assert [p for _, p in got] == sorted((p for _, p in got), reverse=True) and all(p > 0 for _, p in got)
    orc.close()


def test_list_top_targets_cost_model_switch(oracle_lib):
    """CpuEngine.cpp:423-434: the radix branch only for lists of hundreds (never for the quiz loop's 1..10)."""
    f = oracle_lib.orc_list_top_targets_takes_radix
    assert not f(1000, 16, 10) and not f(1000, 16, 300) and f(1000, 16, 400)
    assert not f(100000, 16, 10) and not f(100000, 16, 256) and f(100000, 16, 3000)
