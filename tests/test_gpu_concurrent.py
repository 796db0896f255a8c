"""Many client threads on ONE engine behind the plain ABI (missing #1).  The reference serves every client's
NextQuestion under a shared lock (PqaCore/CpuEngine.cpp:357-361; contract Interface/IPqaEngine.h:44: no concurrent calls on the
SAME quiz); here concurrent NextQuestion calls of different quizzes are combined into one batched sweep, concurrent RecordAnswers
into one launch (hip_engine_combine.cpp: Combine, hip_engine_update.cpp: FlushUpdates).  Whatever the threads' interleaving, every quiz's transcript must be
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
    assert eng.get_option("posted_ops") > 0 and eng.get_option("posted_drains") > 0     # calls that found the engine taken
    # with training at the end of every quiz and the sampled selector the transcripts depend on the interleaving; the run must
    # still complete without an error and teach the cube (most guesses end on top)
    eng.set_option("select", 0)
    trained = interop.run_learners(eng, 48, 192, 30, seed=10, train=True)
    assert trained["errors"] == 0 and trained["quizzes"] == 192 and trained["guessed_on_top"] > 96
    assert eng.get_option("train_batch_calls") > eng.get_option("train_batches") > 0     # RecordQuizTarget calls that shared a launch
    eng.close()


def test_tied_listings_from_many_threads(factory):
    """A cube trained without noise: the posteriors tie, so most ListTopTargets calls of the learner threads take the reference's heaps
    on the device (hip_engine_update.cpp: ListTopTargetsExact) -- from 1 and from 32 threads the same transcripts; and Python threads
    listing three targets each step get, quiz by quiz, what the same script gets alone."""
    K, Q, T = 5, 200, 1000
    eng, err = factory.create_cpu_engine(interop.EngineDefinition(K, Q, T, init_amount=0.1))
    assert err is None
    eng.set_option("workers", cases.WORKERS)
    eng.set_option("select", 1)
    eng.fill_synthetic(8.0, 0.0, 11)
    one = interop.run_learners(eng, 1, 64, 10, seed=4, train=False)
    before = eng.get_option("top_exact_listings")
    many = interop.run_learners(eng, 32, 64, 10, seed=4, train=False)
    assert one["errors"] == 0 and many["errors"] == 0
    assert (many["questions"], many["guessed_on_top"], many["transcript_hash"]) == (one["questions"], one["guessed_on_top"], one["transcript_hash"])
    assert eng.get_option("top_exact_listings") > before        # (ties were met)

    def script(seed, out):
        rng = np.random.default_rng(seed)
        quiz = eng.start_quiz()
        log = []
        for _ in range(6):
            q = eng.next_question(quiz)
            a = int(rng.integers(0, K))
            eng.record_answer(quiz, a)
            log.append((q, a, tuple((t.i_target, t.prob) for t in eng.list_top_targets(quiz, 3))))
        eng.release_quiz(quiz)
        out.append((seed, log))

    alone, together = [], []
    for seed in range(24):
        script(seed, alone)
    threads = [threading.Thread(target=script, args=(seed, together)) for seed in range(24)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert sorted(together) == sorted(alone)
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


@pytest.mark.parametrize("option", [("server", 1), ("speculate", 0), ("top_cache", 0)], ids=lambda o: "%s=%d" % o)
def test_clients_that_never_list_targets(option, factory):
    """clients that go RecordAnswer -> NextQuestion with no ListTopTargets in between.  The deferred updates
    are then launched by the selection itself (FlushUpdates' batched launch right ahead of the request to the resident sweep, or by
    the drain on a holder's way out); the resident sweep is not ordered behind the engine's stream and must not read a posterior
    that such a launch is still writing.  Digest of all transcripts: 32 threads = one thread; repeated, since a race shows up
    now and then."""
    eng = _engine(factory, 5, 300, 1000, 8, 1)
    eng.set_option(*option)
    one = interop.run_learners(eng, 1, 64, 10, seed=6, train=False, list_targets=False)
    assert one["errors"] == 0 and one["questions"] == 640
    for rep in range(4):
        many = interop.run_learners(eng, 32, 64, 10, seed=6, train=False, list_targets=False)
        assert many["errors"] == 0
        assert (many["questions"], many["transcript_hash"]) == (one["questions"], one["transcript_hash"]), rep
    assert eng.get_option("update_max_flush") > 1
    eng.close()


def test_release_of_a_quiz_inside_a_selection_is_refused_not_waited_for(factory):
    """ReleaseQuiz of a quiz whose NextQuestion is in flight on another thread (the caller's error, IPqaEngine.h:44) must
    not hang the engine -- the posted form is refused or succeeds, whichever the timing gives; the engine serves on."""
    eng = _engine(factory, 5, 300, 1000, 8, 1)
    quizzes = [eng.start_quiz() for _ in range(24)]
    stop = threading.Event()
    errors = []

    def asker(z):
        try:
            for _ in range(6):
                eng.next_question(z)
                eng.record_answer(z, 1)
        except interop.PqaException as e:      # (its quiz was released under it: absent id)
            errors.append(str(e))

    def releaser():
        for z in quizzes[:6]:
            try:
                eng.release_quiz(z)
            except interop.PqaException as e:
                errors.append(str(e))

    ts = [threading.Thread(target=asker, args=(z,)) for z in quizzes] + [threading.Thread(target=releaser)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(timeout=60)
    assert not any(t.is_alive() for t in ts), "the engine hangs"
    z = eng.start_quiz()
    assert eng.next_question(z) >= 0
    eng.close()


def test_posted_operations_equal_direct_calls(factory):
    """RecordAnswer and ListTopTargets in their posted form (what a call does that finds the engine taken: hip_engine.h, posted
    operations) against the direct form: the same transcripts, top lists and errors.  Option post_always forces the form."""
    a = _engine(factory, 5, 60, 200, 4, 1)
    b = _engine(factory, 5, 60, 200, 4, 1)
    b.set_option("post_always", 1)
    for eng in (a, b):
        eng.set_option("select", 1)
    qa, qb = a.start_quiz(), b.start_quiz()
    for step in range(12):
        na, nb = a.next_question(qa), b.next_question(qb)
        assert na == nb
        a.record_answer(qa, (step * 3) % 5)
        b.record_answer(qb, (step * 3) % 5)
        for want in (1, 5, 32, 40):      # (40: more than a quiz's pinned lines hold -- the posted form hands it back)
            ta, tb = a.list_top_targets(qa, want), b.list_top_targets(qb, want)
            assert [(t.i_target, t.prob) for t in ta] == [(t.i_target, t.prob) for t in tb]
    assert b.get_option("posted_ops") >= 12 * 5 and a.get_option("posted_ops") == 0
    # RecordQuizTarget, ReleaseQuiz and StartQuiz in the posted form: the trained cube gives the same priorities to a new quiz
    a.record_quiz_target(qa, 7)
    b.record_quiz_target(qb, 7)
    a.release_quiz(qa)
    b.release_quiz(qb)
    qa, qb = a.start_quiz(), b.start_quiz()
    assert np.array_equal(a.eval_priorities(qa), b.eval_priorities(qb)) and np.array_equal(a.get_priors(qa), b.get_priors(qb))
    with pytest.raises(interop.PqaException):
        b.release_quiz(4321)
    with pytest.raises(interop.PqaException):
        b.record_quiz_target(qb, 10**6)
    # the errors come back through the operation
    for eng in (a, b):
        with pytest.raises(interop.PqaException) as e1:
            eng.record_answer(qa if eng is a else qb, 0)            # no active question
        assert "active question" in str(e1.value)
        with pytest.raises(interop.PqaException):
            eng.record_answer(12345, 0)                              # no such quiz
        with pytest.raises(interop.PqaException):
            eng.list_top_targets(12345, 3)
        eng.next_question(qa if eng is a else qb)
        with pytest.raises(interop.PqaException):
            eng.record_answer(qa if eng is a else qb, 5)             # answer out of range
    a.close(); b.close()


def test_training_calls_that_arrive_together_share_a_launch(factory):
    """RecordQuizTarget from many threads at once (the end of the reference's learner loop, PqaClient.cpp:214-222): the calls a
    drain finds posted go out in ONE launch (kb_kernels.hip: train_batch_inline_kernel -- a workgroup per call, different targets;
    a target met twice closes the batch).  Held to the same calls made one after the other on a second engine: the whole cube bit
    for bit (different targets touch disjoint cells, so the order of arrival does not matter), including a question answered twice
    in one quiz (Perform2's same-question rules) and a target that two quizzes share."""
    n = 24
    a = _engine(factory, 5, 60, 200, 9, 1)
    b = _engine(factory, 5, 60, 200, 9, 1)
    a.set_option("post_always", 1)
    quizzes = {}
    for eng in (a, b):
        zs = []
        for i in range(n):
            z = eng.start_quiz()
            for step in range(2 + i % 4):
                qq = (7 * i + 11 * step) % 60 if not (i % 5 == 0 and step == 1) else (7 * i) % 60     # (every fifth quiz repeats its first question)
                if qq == 5:
                    qq = 6                                                                            # (5 is a gap of the fixture)
                eng.set_active_question(z, qq)
                eng.record_answer(z, (i + step) % 5)
            zs.append(z)
        quizzes[id(eng)] = zs
    targets = [3 + 7 * i for i in range(n)]
    targets[-1] = targets[0]                       # one target twice: the second call closes the first's batch
    start = threading.Barrier(n - 1)

    def train(i):
        start.wait()
        a.record_quiz_target(quizzes[id(a)][i], targets[i], 1.0 + 0.25 * (i % 3))

    # (the two calls that share a target are not concurrent: their order would show in mD's roundings)
    ts = [threading.Thread(target=train, args=(i,)) for i in range(n - 1)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    a.record_quiz_target(quizzes[id(a)][n - 1], targets[n - 1], 1.0 + 0.25 * ((n - 1) % 3))
    for i in range(n):
        b.record_quiz_target(quizzes[id(b)][i], targets[i], 1.0 + 0.25 * (i % 3))
    for x, y in zip(a.get_kb(), b.get_kb()):
        assert np.array_equal(x, y)
    assert a.get_option("train_batch_calls") == n and b.get_option("train_batch_calls") == 0
    a.close(); b.close()
