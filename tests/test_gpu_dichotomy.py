"""The reference's end-to-end regression, PqaCoreTests/DichotomyTest.cpp:10-100, on the HIP engine through the C ABI:
1000 questions x 5 answers x 1000 targets, initAmount 0.1, the +-32 "binary search" answer rule (:50-64); each quiz
asks questions (reference selector: priority-proportional sampling) until the guessed target shows up in the top 10,
then trains on it (RecordQuizTarget).  The reference trains for > 3 M questions and demands >= 98 % of 10 000 trials
(:33,:99); this shortened form trains for the ~150-250 k questions its README reports as sufficient (README.md:37) and
applies the same 98 % bar to the following 1000 quizzes.  It exercises NextQuestion, RecordAnswer, ListTopTargets,
RecordQuizTarget and ReleaseQuiz on the device exactly as the reference's test drives its CPU engine."""
import numpy as np
import pytest

from probqa_amd import interop, synth

pytestmark = pytest.mark.gpu


def play(eng, rng, n_targets, width, n_quizzes, max_len=100, top=10):
    hits, lens = 0, 0
    for _ in range(n_quizzes):
        guess = int(rng.integers(n_targets))                       # DichotomyTest.cpp:40
        quiz = eng.start_quiz()                                    # :41
        for j in range(max_len):                                   # :45
            q = eng.next_question(quiz)                            # :46
            eng.record_answer(quiz, synth.dichotomy_answer(q, guess, width))  # :50-68
            if guess in [t.i_target for t in eng.list_top_targets(quiz, top)]:  # :71-82
                hits += 1
                break
        lens += j + 1
        eng.record_quiz_target(quiz, guess)                        # :94
        eng.release_quiz(quiz)                                     # :96
    return hits / n_quizzes, lens / n_quizzes


@pytest.mark.parametrize("n,width,train,trials,bar", [(200, 6, 1500, 500, 0.98), (1000, 32, 12000, 1000, 0.98)],
                         ids=["200x5x200", "1000x5x1000_reference_dims"])
def test_dichotomy_learning(factory, n, width, train, trials, bar):
    eng, err = factory.create_cpu_engine(interop.EngineDefinition(5, n, n, init_amount=0.1))  # :12-18
    assert err is None
    eng.set_option("seed", 20260928)
    rng = np.random.default_rng(20260928)
    early, early_len = play(eng, rng, n, width, 200)
    play(eng, rng, n, width, train - 200)
    acc, avg_len = play(eng, rng, n, width, trials)
    print("dichotomy %dx5x%d: first 200 quizzes %.3f (len %.1f); after %d quizzes / %d questions: top-10 in 100 = %.4f, "
          "mean quiz length %.2f" % (n, n, early, early_len, train, eng.get_total_questions_asked(), acc, avg_len))
    assert early < 0.9          # it really had to learn
    assert acc >= bar           # DichotomyTest.cpp:99
    assert avg_len < 15
    eng.close()
