"""Many client threads on ONE engine behind the plain ABI (missing #1).  The reference serves every client's
NextQuestion under a shared lock (PqaCore/CpuEngine.cpp:357-361; contract Interface/IPqaEngine.h:44: no concurrent calls on the
SAME quiz); here concurrent NextQuestion calls of different quizzes are combined into one batched sweep, concurrent RecordAnswers
into one launch (hip_engine.cpp: Combine / FlushUpdates).  Whatever the threads' interleaving, every quiz's transcript must be
the one the same script produces alone -- questions, listed targets and the final posterior bit for bit."""
import threading

import numpy as np
import pytest

import cases
from probqa_amd import interop

pytestmark = pytest.mark.gpu


def _engine(factory, K, Q, T, seed, select):
    case = cases.Case("conc", K, Q, T, seed=seed, qgaps=[5])
    eng = case.make_engine(factory)
    eng.set_option("select", select)
    return eng


@pytest.mark.parametrize("sampled", [False, True], ids=["argmax", "sampled"])
@pytest.mark.parametrize("shape", [(5, 80, 300), (4, 300, 1000)], ids=lambda s: "%dx%dx%d" % (s[1], s[0], s[2]))
def test_64_threads_transcripts_equal_sequential(factory, shape, sampled):
    K, Q, T = shape
    eng = _engine(factory, K, Q, T, 77, 1)
    n_threads, n_steps = 64, 10

    def script(seed, out):
        rng = np.random.default_rng(seed)
        quiz = eng.start_quiz()
        log = []
        for _ in range(n_steps):
            q = eng.next_question_sampled(quiz, int(rng.integers(0, 2**63))) if sampled else eng.next_question(quiz)
            a = int(rng.integers(0, K))
            eng.record_answer(quiz, a)
            top = eng.list_top_targets(quiz, 3)
            log.append((q, a, tuple((t.i_target, t.prob) for t in top)))
        log.append(eng.get_priors(quiz).tobytes())
        eng.release_quiz(quiz)
        out.append((seed, log))

    alone = []
    for seed in range(n_threads):
        script(seed, alone)
    assert eng.get_option("combined_batches") == 0          # one caller at a time: every call by itself
    together = []
    threads = [threading.Thread(target=script, args=(seed, together)) for seed in range(n_threads)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert sorted(together) == sorted(alone)
    eng.close()     # (whether Python threads overlap inside the engine is the interpreter's business; the native client's test asserts it)


def test_native_learner_threads_same_transcripts_and_combined(factory):
    """The learner client of probqa_amd/client/pqa_client.cpp (the reference's PqaClient threads): the digest of all transcripts
    with 48 threads equals the one-thread run's (argmax selector, no training: nothing depends on the interleaving), and the
    engine did combine: sweeps that served several NextQuestion calls, launches that ran several RecordAnswers."""
    eng = _engine(factory, 5, 300, 1000, 5, 1)
    one = interop.run_learners(eng, 1, 96, 12, seed=9, train=False)
    assert one["errors"] == 0 and one["quizzes"] == 96
    assert eng.get_option("combined_batches") == 0
    many = interop.run_learners(eng, 48, 96, 12, seed=9, train=False)
    assert many["errors"] == 0 and many["quizzes"] == 96
    assert (many["questions"], many["guessed_on_top"], many["transcript_hash"]) == (one["questions"], one["guessed_on_top"], one["transcript_hash"])
    assert eng.get_option("combined_batches") > 0 and eng.get_option("combined_max_batch") > 4
    assert eng.get_option("update_max_flush") > 1
    # with training at the end of every quiz and the sampled selector the transcripts depend on the interleaving; the run must
    # still complete without an error and teach the cube (most guesses end on top)
    eng.set_option("select", 0)
    trained = interop.run_learners(eng, 48, 192, 30, seed=10, train=True)
    assert trained["errors"] == 0 and trained["quizzes"] == 192 and trained["guessed_on_top"] > 96
    eng.close()


def test_combining_can_be_switched_off(factory):
    eng = _engine(factory, 5, 40, 100, 3, 1)
    eng.set_option("combine", 0)
    many = interop.run_learners(eng, 16, 48, 8, seed=2, train=False)
    assert many["errors"] == 0 and eng.get_option("combined_batches") == 0 and eng.get_option("update_max_flush") <= 1
    eng.set_option("combine", 1)
    again = interop.run_learners(eng, 16, 48, 8, seed=2, train=False)
    assert again["transcript_hash"] == many["transcript_hash"]
    eng.close()


@pytest.mark.parametrize("option", [("server", 1), ("use_graph", 1), ("speculate", 0), ("combine_linger_us", 0), ("host_sampled", 0)],
                         ids=lambda o: "%s=%d" % o)
def test_native_learners_under_every_selection_path(option, factory):
    """The combining of concurrent calls beside the engine's other selection paths -- the resident sweep and graph replay serve one
    quiz at a time (a combined batch falls back to them request by request), speculation off, no lingering, the selector kernel
    instead of the host's selector: 32 learner threads, the digest of all transcripts equals the one-thread run's."""
    eng = _engine(factory, 5, 300, 1000, 6, 1)
    eng.set_option(*option)
    one = interop.run_learners(eng, 1, 64, 10, seed=4, train=False)
    many = interop.run_learners(eng, 32, 64, 10, seed=4, train=False)
    assert one["errors"] == 0 and many["errors"] == 0
    assert (many["questions"], many["transcript_hash"]) == (one["questions"], one["transcript_hash"])
    eng.set_option("select", 0)                      # the sampled selector: interleaving-dependent, must simply work
    again = interop.run_learners(eng, 32, 64, 10, seed=5, train=True)
    assert again["errors"] == 0 and again["quizzes"] == 64
    eng.close()
