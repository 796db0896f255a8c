"""The two soaks that found every late offender of rounds 3 and 4, as seeded tests: 200 cases and fifty concurrency rounds in every run of the suite, as
many more as asked for with `pytest -m gpu tests/test_gpu_soak.py --soak N [--soak-first SEED]` (round 4: --soak 13700 is clean,
390 s on one MI355X).  tools/fuzz_more.py and tools/concurrency_soak.py are the same loops from the command line."""
import random
import traceback

import pytest

import test_gpu_fuzz as tf
from probqa_amd import interop

pytestmark = pytest.mark.gpu


def fuzz_cases(first, last, factory, report=print):
    """Seeds first..last-1 of tests/test_gpu_fuzz.py's generator through its single-quiz and batched checks (posteriors bit for bit,
    priorities within 1e-9 / the fp32 bound, the selected questions).  Returns the failures as (leg, seed, case name, what)."""
    bad = []
    for i in range(first, last):
        for leg, fn in (("single", tf.test_random_case), ("batched", tf.test_random_case_batched)):
            try:
                fn(i, factory)
            except BaseException as ex:  # noqa: BLE001
                if type(ex).__name__ == "Skipped":
                    continue
                if isinstance(ex, KeyboardInterrupt):
                    raise
                bad.append((leg, i, tf.random_case(i).name, repr(ex)[:300]))
                report("FAIL %s %d %s %s" % bad[-1])
                report(traceback.format_exc(limit=2))
    return bad


def concurrency_rounds(rounds, factory, seed=20260929, report=print):
    """Learner threads of random counts on one engine (combined sweeps, posted operations, group commit), both selectors: the
    transcript of the argmax loop must be the one-thread transcript whatever the interleaving; with training every quiz must
    finish without an error.  Returns the rounds that did not."""
    rnd = random.Random(seed)
    bad = []
    for r in range(rounds):
        K, Q, T = rnd.choice([(5, 300, 1000), (5, 80, 300), (5, 1000, 1000), (5, 50, 2000), (6, 120, 700), (8, 100, 500)])   # (the client answers 0..4)
        e, err = factory.create_cpu_engine(interop.EngineDefinition(K, Q, T, init_amount=0.1))
        assert e is not None, err
        try:
            e.fill_synthetic(8.0, 0.5, 1000 + r)
            e.set_option("select", 1)
            if rnd.random() < 0.3:
                e.set_option("combine_linger_us", rnd.choice([0, 5, 50]))
            if rnd.random() < 0.2:
                e.set_option("post_always", 1)
            nq, mq = rnd.choice([48, 96, 200]), rnd.choice([6, 12, 25])
            one = interop.run_learners(e, 1, nq, mq, seed=r, train=False)
            nt = rnd.choice([2, 3, 7, 16, 33, 64, 150])
            many = interop.run_learners(e, nt, nq, mq, seed=r, train=False)
            ok = one["errors"] == 0 and many["errors"] == 0 and \
                (many["questions"], many["transcript_hash"]) == (one["questions"], one["transcript_hash"])
            e.set_option("select", rnd.choice([0, 1]))
            tr = interop.run_learners(e, nt, nq, mq, seed=r + 1, train=True)
            ok = ok and tr["errors"] == 0 and tr["quizzes"] == nq
            report("round %2d: %dx%dx%d %3d threads: %s  (%.0f q/s; posted %d, combined %d)" % (
                r, Q, K, T, nt, "ok" if ok else "MISMATCH", tr["questions"] / tr["seconds"], e.get_option("posted_ops"),
                e.get_option("combined_batches")))
            if not ok:
                bad.append((r, (Q, K, T), nt))
        finally:
            e.close()
    return bad


def test_fuzz_soak(factory, soak):
    n, first = soak
    bad = fuzz_cases(first, first + (n if n > 0 else 200), factory)
    assert not bad, bad[:5]


def test_concurrency_soak(factory, soak):
    n, first = soak
    bad = concurrency_rounds(50 if n <= 0 else max(50, n // 50), factory, seed=20260929 + (first if n > 0 else 0))
    assert not bad, bad
