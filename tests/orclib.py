"""ctypes access to the CPU oracle (oracle/libpqa_oracle.so) for tests, smoke() and bench.py's cpu_baseline leg.
TEST INFRASTRUCTURE ONLY -- the product (probqa_amd/) never imports this module."""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB = os.path.join(ORACLE_DIR, "libpqa_oracle.so")


class OrcKB(ctypes.Structure):
    _fields_ = [("nAnswers", ctypes.c_int64), ("nQuestions", ctypes.c_int64), ("nTargets", ctypes.c_int64),
                ("ldT", ctypes.c_int64), ("A", ctypes.POINTER(ctypes.c_double)), ("D", ctypes.POINTER(ctypes.c_double)),
                ("B", ctypes.POINTER(ctypes.c_double)), ("targetGaps", ctypes.POINTER(ctypes.c_uint8)),
                ("questionGaps", ctypes.POINTER(ctypes.c_uint8)), ("nTargetGaps", ctypes.c_int64)]


class OrcQuiz(ctypes.Structure):
    _fields_ = [("mants", ctypes.POINTER(ctypes.c_double)), ("exps", ctypes.POINTER(ctypes.c_int64)),
                ("asked", ctypes.POINTER(ctypes.c_uint8))]


class OrcAQ(ctypes.Structure):
    _fields_ = [("iQuestion", ctypes.c_int64), ("iAnswer", ctypes.c_int64)]


class OrcRatedTarget(ctypes.Structure):
    _fields_ = [("iTarget", ctypes.c_int64), ("prob", ctypes.c_double)]


class OrcKahan4(ctypes.Structure):
    _fields_ = [("sum", ctypes.c_double * 4), ("corr", ctypes.c_double * 4)]


_lib = None


def build():
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR])


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB):
        build()
    L = ctypes.CDLL(LIB)
    pKB, pQz, pd = ctypes.POINTER(OrcKB), ctypes.POINTER(OrcQuiz), ctypes.POINTER(ctypes.c_double)
    i64, u64 = ctypes.c_int64, ctypes.c_uint64
    sig = {
        "orc_log2hot": (ctypes.c_double, [ctypes.c_double]),
        "orc_log2hot_table": (pd, []),
        "orc_k4_reset": (None, [ctypes.POINTER(OrcKahan4)]),
        "orc_k4_add": (None, [ctypes.POINTER(OrcKahan4), pd]),
        "orc_k4_precise_sum": (ctypes.c_double, [ctypes.POINTER(OrcKahan4)]),
        "orc_k4_pair_sum": (ctypes.c_double, [ctypes.POINTER(OrcKahan4), ctypes.POINTER(OrcKahan4), pd]),
        "orc_k4_full_sum": (ctypes.c_double, [ctypes.POINTER(OrcKahan4)]),
        "orc_calc_split": (i64, [i64, i64, ctypes.POINTER(i64)]),
        "orc_kb_create": (pKB, [i64, i64, i64, ctypes.c_double]),
        "orc_kb_destroy": (None, [pKB]),
        "orc_kb_set_target_gap": (None, [pKB, i64, ctypes.c_int]),
        "orc_kb_set_question_gap": (None, [pKB, i64, ctypes.c_int]),
        "orc_kb_train": (None, [pKB, i64, ctypes.POINTER(OrcAQ), i64, ctypes.c_double]),
        "orc_kb_train_workers": (None, [pKB, i64, ctypes.POINTER(OrcAQ), i64, ctypes.c_double, i64]),
        "orc_kb_record_quiz_target": (None, [pKB, i64, ctypes.POINTER(OrcAQ), i64, ctypes.c_double]),
        "orc_quiz_create": (pQz, [pKB]),
        "orc_quiz_destroy": (None, [pQz]),
        "orc_eval_subtask": (None, [pKB, pQz, i64, i64, i64, pd, pd]),
        "orc_eval_all": (None, [pKB, pQz, i64, pd, pd]),
        "orc_select_sampled": (i64, [pKB, pQz, i64, pd, u64]),
        "orc_find_nearest_question": (i64, [pKB, pQz, i64]),
        "orc_select_argmax": (i64, [pKB, pQz, pd]),
        "orc_start_quiz": (None, [pKB, pQz, i64]),
        "orc_record_answer": (None, [pKB, pQz, i64, i64, i64]),
        "orc_resume_quiz": (ctypes.c_int, [pKB, pQz, i64, ctypes.POINTER(OrcAQ), i64, ctypes.c_int]),
        "orc_eval_all_avx2_mt": (None, [pKB, pQz, i64, i64, pd, pd]),
        "orc_have_avx2": (ctypes.c_int, []),
        "orc_list_top_targets": (i64, [pKB, pQz, i64, i64, ctypes.POINTER(OrcRatedTarget)]),
        "orc_list_top_targets_takes_radix": (ctypes.c_int, [i64, i64, i64]),
        "orc_heap_make": (None, [pd, ctypes.POINTER(i64), i64]),
        "orc_heap_pop": (None, [pd, ctypes.POINTER(i64), i64]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)
        fn.restype, fn.argtypes = res, args
    _lib = L
    return L


def _dp(a: np.ndarray):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))


class Oracle:
    """A CPU-oracle knowledge base plus one quiz, with numpy views of its arrays."""

    def __init__(self, K: int, Q: int, T: int, init: float = 1.0):
        self.L = lib()
        self.K, self.Q, self.T = K, Q, T
        self.kb = self.L.orc_kb_create(K, Q, T, init)
        self.ldT = self.kb.contents.ldT
        self.A = np.ctypeslib.as_array(self.kb.contents.A, shape=(Q, K, self.ldT))
        self.D = np.ctypeslib.as_array(self.kb.contents.D, shape=(Q, self.ldT))
        self.B = np.ctypeslib.as_array(self.kb.contents.B, shape=(self.ldT,))
        self.quiz = self.L.orc_quiz_create(self.kb)
        self.mants = np.ctypeslib.as_array(self.quiz.contents.mants, shape=(self.ldT,))
        self.answers = []

    def close(self):
        if self.kb:
            self.L.orc_quiz_destroy(self.quiz)
            self.L.orc_kb_destroy(self.kb)
            self.kb = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_kb(self, A: np.ndarray, D: np.ndarray, B: np.ndarray):
        self.A[:, :, : self.T] = A
        self.D[:, : self.T] = D
        self.B[: self.T] = B

    def set_target_gaps(self, ts):
        for t in ts:
            self.L.orc_kb_set_target_gap(self.kb, int(t), 1)

    def set_question_gaps(self, qs):
        for q in qs:
            self.L.orc_kb_set_question_gap(self.kb, int(q), 1)

    def train(self, aqs, i_target: int, amount: float = 1.0, n_workers: int = 16):
        """CpuEngine::TrainSpec for a thread pool of n_workers (repeated questions pair up by its bucket order)."""
        arr = (OrcAQ * max(len(aqs), 1))(*[OrcAQ(q, a) for q, a in aqs])
        self.L.orc_kb_train_workers(self.kb, len(aqs), arr, i_target, amount, n_workers)

    def record_quiz_target(self, i_target: int, amount: float = 1.0, aqs=None):
        """RecordQuizTarget over the quiz's answers (or the given ones), in order."""
        aqs = self.answers if aqs is None else aqs
        arr = (OrcAQ * max(len(aqs), 1))(*[OrcAQ(q, a) for q, a in aqs])
        self.L.orc_kb_record_quiz_target(self.kb, len(aqs), arr, i_target, amount)

    def start_quiz(self, n_workers: int = 16):
        self.answers = []
        self.L.orc_start_quiz(self.kb, self.quiz, n_workers)

    def resume_quiz(self, aqs, n_workers: int = 16, bug_compat: bool = False) -> int:
        arr = (OrcAQ * max(len(aqs), 1))(*[OrcAQ(q, a) for q, a in aqs])
        self.answers = list(aqs)
        return self.L.orc_resume_quiz(self.kb, self.quiz, len(aqs), arr, n_workers, 1 if bug_compat else 0)

    def record_answer(self, q: int, a: int, n_workers: int = 15):
        self.answers.append((q, a))
        self.L.orc_record_answer(self.kb, self.quiz, q, a, n_workers)

    def priors(self) -> np.ndarray:
        return self.mants[: self.T].copy()

    def eval(self, n_subtasks: int = 128):
        run = np.empty(self.Q)
        pri = np.empty(self.Q)
        self.L.orc_eval_all(self.kb, self.quiz, n_subtasks, _dp(run), _dp(pri))
        return run, pri

    def eval_avx2(self, n_threads: int, n_subtasks: int | None = None):
        run = np.empty(self.Q)
        pri = np.empty(self.Q)
        self.L.orc_eval_all_avx2_mt(self.kb, self.quiz, n_threads, n_subtasks or 8 * n_threads, _dp(run), _dp(pri))
        return run, pri

    def select_sampled(self, run: np.ndarray, n_subtasks: int, rnd: int) -> int:
        return self.L.orc_select_sampled(self.kb, self.quiz, n_subtasks, _dp(run), rnd)

    def select_argmax(self, pri: np.ndarray) -> int:
        return self.L.orc_select_argmax(self.kb, self.quiz, _dp(pri))

    def list_top_targets(self, max_count: int, n_workers: int = 16):
        """CpuEngine::ListTopTargetsSpec's heapify branch for a pool of n_workers threads: [(target, prob)], descending."""
        dest = (OrcRatedTarget * max(max_count, 1))()
        n = self.L.orc_list_top_targets(self.kb, self.quiz, max_count, n_workers, dest)
        return [(dest[i].iTarget, dest[i].prob) for i in range(n)]

    def find_nearest(self, q: int) -> int:
        return self.L.orc_find_nearest_question(self.kb, self.quiz, q)
