"""ctypes binding of libPqaCore.so (the MI355X engine) with the reference wrapper's Python API.

The class and method names, argument meaning and error behaviour mirror the reference's
``Interop/Python/ProbQAInterop/ProbQA.py`` (structs :72-116, prototypes :119-296, ``PqaEngine`` :423-741,
``PqaEngineFactory`` :743-783) so that code written against it runs unchanged; the reference module itself cannot be
imported on Linux (it subclasses ``ctypes.WinDLL`` and loads ``DLLs/PqaCore.dll``).  Everything below the C ABI is
HIP; there is no CPU fallback: loading fails loudly when the library is missing, and engine creation returns an
error when no GPU is present.

Additive, MI355X-specific methods (``eval_priorities``, ``next_question_argmax`` ...) bind ``include/PqaHipExt.h``.
"""
from __future__ import annotations

import ctypes
import os
from enum import Enum
from typing import List, Optional, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PQACORE_LIB", os.path.join(_HERE, "libPqaCore.so"))


class CiEngineDefinition(ctypes.Structure):  # reference PqaCInterop.h:10-19
    _pack_ = 8
    _fields_ = [("nAnswers", ctypes.c_int64), ("nQuestions", ctypes.c_int64), ("nTargets", ctypes.c_int64),
                ("precType", ctypes.c_uint8), ("precExponent", ctypes.c_uint16), ("precMantissa", ctypes.c_uint32),
                ("initAmount", ctypes.c_double), ("memPoolMaxBytes", ctypes.c_uint64)]


class CiAnsweredQuestion(ctypes.Structure):
    _pack_ = 8
    _fields_ = [("iQuestion", ctypes.c_int64), ("iAnswer", ctypes.c_int64)]


class CiEngineDimensions(ctypes.Structure):
    _pack_ = 8
    _fields_ = [("nAnswers", ctypes.c_int64), ("nQuestions", ctypes.c_int64), ("nTargets", ctypes.c_int64)]


class CiRatedTarget(ctypes.Structure):
    _pack_ = 8
    _fields_ = [("iTarget", ctypes.c_int64), ("prob", ctypes.c_double)]


class CiAddQorTParam(ctypes.Structure):
    _pack_ = 8
    _fields_ = [("index", ctypes.c_int64), ("initAmount", ctypes.c_double)]


class CiHipShard(ctypes.Structure):  # include/PqaHipExt.h
    _pack_ = 8
    _fields_ = [("qFirst", ctypes.c_int64), ("qTotal", ctypes.c_int64), ("device", ctypes.c_int32),
                ("reserved", ctypes.c_int32)]


class CiHipSelection(ctypes.Structure):
    _pack_ = 8
    _fields_ = [("priority", ctypes.c_double), ("iQuestion", ctypes.c_int64)]


_vp, _i64, _u64, _u8, _dbl = ctypes.c_void_p, ctypes.c_int64, ctypes.c_uint64, ctypes.c_uint8, ctypes.c_double
_pvp = ctypes.POINTER(ctypes.c_void_p)
_pi64 = ctypes.POINTER(ctypes.c_int64)
_pdbl = ctypes.POINTER(ctypes.c_double)
_pAQ = ctypes.POINTER(CiAnsweredQuestion)

# name -> (restype, argtypes): the 40 reference exports (PqaCInterop.h:48-108) ...
REFERENCE_EXPORTS = {
    "CiDebugBreak": (None, []),
    "Logger_Init": (_u8, [_pvp, ctypes.c_char_p]),
    "CiReleaseString": (None, [_vp]),
    "CiGetPqaEngineFactory": (_vp, []),
    "PqaEngineFactory_CreateCpuEngine": (_vp, [_vp, _pvp, ctypes.POINTER(CiEngineDefinition)]),
    "PqaEngineFactory_LoadCpuEngine": (_vp, [_vp, _pvp, ctypes.c_char_p, _u64]),
    "CiReleasePqaError": (None, [_vp]),
    "PqaError_ToString": (_vp, [_vp, _u8]),
    "CiReleasePqaEngine": (None, [_vp]),
    "PqaEngine_Train": (_vp, [_vp, _i64, _pAQ, _i64, _dbl]),
    "PqaEngine_QuestionPermFromComp": (_u8, [_vp, _i64, _pi64]),
    "PqaEngine_QuestionCompFromPerm": (_u8, [_vp, _i64, _pi64]),
    "PqaEngine_TargetPermFromComp": (_u8, [_vp, _i64, _pi64]),
    "PqaEngine_TargetCompFromPerm": (_u8, [_vp, _i64, _pi64]),
    "PqaEngine_QuizPermFromComp": (_u8, [_vp, _i64, _pi64]),
    "PqaEngine_QuizCompFromPerm": (_u8, [_vp, _i64, _pi64]),
    "PqaEngine_EnsurePermQuizGreater": (_u8, [_vp, _i64]),
    "PqaEngine_RemapQuizPermId": (_u8, [_vp, _i64, _i64]),
    "PqaEngine_GetTotalQuestionsAsked": (_u64, [_vp, _pvp]),
    "PqaEngine_CopyDims": (_u8, [_vp, ctypes.POINTER(CiEngineDimensions)]),
    "PqaEngine_StartQuiz": (_i64, [_vp, _pvp]),
    "PqaEngine_ResumeQuiz": (_i64, [_vp, _pvp, _i64, _pAQ]),
    "PqaEngine_NextQuestion": (_i64, [_vp, _pvp, _i64]),
    "PqaEngine_RecordAnswer": (_vp, [_vp, _i64, _i64]),
    "PqaEngine_ClearOldQuizzes": (_vp, [_vp, _i64, _dbl]),
    "PqaEngine_GetActiveQuestionId": (_i64, [_vp, _pvp, _i64]),
    "PqaEngine_SetActiveQuestion": (_vp, [_vp, _i64, _i64]),
    "PqaEngine_ListTopTargets": (_i64, [_vp, _pvp, _i64, _i64, ctypes.POINTER(CiRatedTarget)]),
    "PqaEngine_RecordQuizTarget": (_vp, [_vp, _i64, _i64, _dbl]),
    "PqaEngine_ReleaseQuiz": (_vp, [_vp, _i64]),
    "PqaEngine_SaveKB": (_vp, [_vp, ctypes.c_char_p, _u8]),
    "PqaEngine_StartMaintenance": (_vp, [_vp, ctypes.c_bool]),
    "PqaEngine_FinishMaintenance": (_vp, [_vp]),
    "PqaEngine_AddQsTs": (_vp, [_vp, _i64, ctypes.POINTER(CiAddQorTParam), _i64, ctypes.POINTER(CiAddQorTParam)]),
    "PqaEngine_RemoveQuestions": (_vp, [_vp, _i64, _pi64]),
    "PqaEngine_RemoveTargets": (_vp, [_vp, _i64, _pi64]),
    "PqaEngine_Compact": (_vp, [_vp, _pi64, ctypes.POINTER(_pi64), _pi64, ctypes.POINTER(_pi64)]),
    "CiReleaseCompaction": (None, [_pi64]),
    "PqaEngine_Shutdown": (_vp, [_vp, ctypes.c_char_p]),
    "PqaEngine_SetLogger": (_vp, [_vp, _vp]),
}
# ... and the additive ones (include/PqaHipExt.h)
HIP_EXPORTS = {
    "PqaEngineFactory_CreateHipEngine": (_vp, [_vp, _pvp, ctypes.POINTER(CiEngineDefinition)]),
    "PqaEngineFactory_CreateHipEngineSharded": (_vp, [_vp, _pvp, ctypes.POINTER(CiEngineDefinition),
                                                      ctypes.POINTER(CiHipShard)]),
    "PqaEngineFactory_LoadHipEngine": (_vp, [_vp, _pvp, ctypes.c_char_p, _u64]),
    "PqaHip_SetOption": (_vp, [_vp, ctypes.c_char_p, _i64]),
    "PqaHip_GetOption": (_i64, [_vp, ctypes.c_char_p]),
    "PqaHip_EvalKernelName": (ctypes.c_char_p, [_vp]),
    "PqaHip_SetKB": (_vp, [_vp, _pdbl, _pdbl, _pdbl]),
    "PqaHip_GetKB": (_vp, [_vp, _pdbl, _pdbl, _pdbl]),
    "PqaHip_FillSynthetic": (_vp, [_vp, _dbl, _dbl, _u64]),
    "PqaHip_SetTargetGaps": (_vp, [_vp, _i64, _pi64]),
    "PqaHip_SetQuestionGaps": (_vp, [_vp, _i64, _pi64]),
    "PqaEngine_EvalPriorities": (_vp, [_vp, _i64, _pdbl, _i64]),
    "PqaEngine_NextQuestionArgmax": (_i64, [_vp, _pvp, _i64]),
    "PqaHip_Log2Hot": (_vp, [_vp, _pdbl, _pdbl, _i64]),
    "PqaEngine_EvalPrioritiesBatch": (_vp, [_vp, _i64, _pi64, _pdbl]),
    "PqaHip_SelectArgmaxBatch": (_vp, [_vp, _i64, _pi64, ctypes.POINTER(CiHipSelection)]),
    "PqaEngine_NextQuestionArgmaxBatch": (_vp, [_vp, _i64, _pi64, _pi64]),
    "PqaEngine_NextQuestionSampled": (_i64, [_vp, _pvp, _i64, _u64]),
    "PqaHip_GetPriors": (_vp, [_vp, _i64, _pdbl, _i64]),
    "PqaHip_GetStream": (_vp, [_vp]),
    "PqaHip_SetStream": (_vp, [_vp, _vp]),
    "PqaHip_Synchronize": (_vp, [_vp]),
    "PqaHip_Quiesce": (_vp, [_vp]),
    "PqaHip_EnqueueSelectArgmax": (_vp, [_vp, _i64, _vp]),
    "PqaHip_EnqueueSelectArgmaxFlag": (_vp, [_vp, _i64, _vp, _vp, ctypes.c_uint64]),
    "PqaHip_HostRegister": (_vp, [_vp, _i64, _pvp]),
    "PqaHip_HostUnregister": (_vp, [_vp]),
    "PqaHip_PickWhenAll": (_vp, [_vp, _i64, _i64, ctypes.c_uint64, ctypes.c_double, _pdbl, _pi64]),
    "PqaHip_SelectThroughSlots": (_vp, [_vp, _i64, _vp, _vp, _i64, _i64, _i64, ctypes.c_uint64, ctypes.c_double, _pdbl, _pi64]),
    "PqaHip_SelectArgmaxRccl": (_vp, [_vp, _i64, _vp, _i64, _pdbl, _pi64]),
    "PqaHip_EnqueueEval": (_vp, [_vp, _i64]),
    "PqaHip_GetPriorDevicePtr": (_vp, [_vp, _i64, _pvp, _pi64]),
    "PqaHip_RecordAnswerRemote": (_vp, [_vp, _i64, _i64]),
    "PqaEngine_RecordAnswerBatch": (_vp, [_vp, _i64, _pi64, _pi64]),
    "PqaEngine_StartQuizBatch": (_vp, [_vp, _i64, _pi64]),
    "PqaEngine_ListTopTargetsBatch": (_vp, [_vp, _i64, _pi64, _i64, ctypes.POINTER(CiRatedTarget), _pi64]),
    "PqaHip_HostLogicProbe": (_i64, [ctypes.c_char_p, _pi64, _i64, _pi64, _i64]),
}

_lib = None


def load_library(path: Optional[str] = None) -> ctypes.CDLL:
    """Load libPqaCore.so and declare every prototype.  Raises OSError if the HIP library is not built."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or LIB_PATH
    if not os.path.exists(p):
        raise OSError(f"{p} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                      "(there is no CPU fallback)")
    lib = ctypes.CDLL(p, mode=ctypes.RTLD_GLOBAL)
    for table in (REFERENCE_EXPORTS, HIP_EXPORTS):
        for name, (res, args) in table.items():
            fn = getattr(lib, name)  # AttributeError if a declared symbol is not exported
            fn.restype = res
            fn.argtypes = args
    if path is None:
        _lib = lib
    return lib


class PqaClientStats(ctypes.Structure):  # probqa_amd/client/pqa_client.cpp
    _fields_ = [("nQuizzes", ctypes.c_int64), ("nQuestions", ctypes.c_int64), ("nGuessedOnTop", ctypes.c_int64),
                ("nErrors", ctypes.c_int64), ("seconds", ctypes.c_double), ("transcriptHash", ctypes.c_uint64)]


CLIENT_LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libPqaClient.so")
_client = None


def run_learners(engine: "PqaEngine", n_threads: int, n_quizzes: int, max_questions: int = 30, seed: int = 1, train: bool = True,
                 list_targets: bool = True) -> dict:
    """The reference's learner client (PqaClient/PqaClient.cpp:150-245) as native threads on ONE engine, through the C ABI only:
    `n_threads` threads share `n_quizzes` quizzes (StartQuiz, NextQuestion / RecordAnswer / ListTopTargets(1) until the guess is
    on top or `max_questions` were asked, RecordQuizTarget if `train`, ReleaseQuiz).  `list_targets=False`: clients that go from
    RecordAnswer straight to the next NextQuestion."""
    global _client
    load_library()
    if _client is None:
        if not os.path.exists(CLIENT_LIB_PATH):
            raise OSError(f"{CLIENT_LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'`")
        _client = ctypes.CDLL(CLIENT_LIB_PATH)
        _client.PqaClient_RunLearners.restype = ctypes.c_int64
        _client.PqaClient_RunLearners.argtypes = [_vp, _i64, _i64, _i64, ctypes.c_uint64, _i64, ctypes.POINTER(PqaClientStats)]
    st = PqaClientStats()
    rc = _client.PqaClient_RunLearners(engine.c_engine, n_threads, n_quizzes, max_questions, seed, (1 if train else 0) | (0 if list_targets else 2), ctypes.byref(st))
    if rc != 0:
        raise PqaException("PqaClient_RunLearners refused its arguments")
    return {"quizzes": st.nQuizzes, "questions": st.nQuestions, "guessed_on_top": st.nGuessedOnTop, "errors": st.nErrors,
            "seconds": st.seconds, "transcript_hash": st.transcriptHash, "threads": n_threads}


def time_selections_native(engine: "PqaEngine", i_quiz: int, n_warm: int, n: int):
    """(seconds, last selected question) of n synchronous PqaEngine_NextQuestion calls made by native code (libPqaClient.so): the
    step of bench.py without the Python wrapper around every call."""
    global _client
    load_library()
    if _client is None:
        _client = ctypes.CDLL(CLIENT_LIB_PATH)
        _client.PqaClient_RunLearners.restype = ctypes.c_int64
        _client.PqaClient_RunLearners.argtypes = [_vp, _i64, _i64, _i64, ctypes.c_uint64, _i64, ctypes.POINTER(PqaClientStats)]
    _client.PqaClient_TimeSelections.restype = ctypes.c_double
    _client.PqaClient_TimeSelections.argtypes = [_vp, _i64, _i64, _i64, ctypes.POINTER(_i64)]
    last = _i64(-1)
    dt = _client.PqaClient_TimeSelections(engine.c_engine, i_quiz, n_warm, n, ctypes.byref(last))
    if dt < 0:
        raise PqaException("PqaClient_TimeSelections: a call of the ABI failed")
    return dt, int(last.value)


class PqaException(Exception):  # reference ProbQA.py:300-302
    pass


class PrecisionType(Enum):  # reference PqaCommon.h:17-24
    NONE = 0
    FLOAT = 1
    FLOAT_PAIR = 2
    DOUBLE = 3
    DOUBLE_PAIR = 4
    ARBITRARY = 5


class AnsweredQuestion:
    def __init__(self, i_question: int, i_answer: int):
        self.i_question = i_question
        self.i_answer = i_answer

    def __repr__(self):
        return f"[q={self.i_question}, a={self.i_answer}]"


class RatedTarget:
    def __init__(self, i_target: int, prob: float):
        self.i_target = i_target
        self.prob = prob

    def __repr__(self):
        return f"[t={self.i_target}, p={self.prob}]"


class EngineDefinition:
    DEFAULT_MEM_POOL_MAX_BYTES = 512 * 1024 * 1024

    def __init__(self, n_answers: int, n_questions: int, n_targets: int, init_amount=1.0,
                 prec_type=PrecisionType.DOUBLE, prec_exponent=11, prec_mantissa=53,
                 mem_pool_max_bytes=DEFAULT_MEM_POOL_MAX_BYTES):
        self.n_answers, self.n_questions, self.n_targets = n_answers, n_questions, n_targets
        self.init_amount = init_amount
        self.prec_type, self.prec_exponent, self.prec_mantissa = prec_type, prec_exponent, prec_mantissa
        self.mem_pool_max_bytes = mem_pool_max_bytes

    def to_c(self) -> CiEngineDefinition:
        c = CiEngineDefinition()
        c.nAnswers, c.nQuestions, c.nTargets = self.n_answers, self.n_questions, self.n_targets
        c.precType, c.precExponent, c.precMantissa = self.prec_type.value, self.prec_exponent, self.prec_mantissa
        c.initAmount, c.memPoolMaxBytes = self.init_amount, self.mem_pool_max_bytes
        return c


class EngineDimensions:
    def __init__(self, n_answers: int, n_questions: int, n_targets: int):
        self.n_answers, self.n_questions, self.n_targets = n_answers, n_questions, n_targets

    def __repr__(self):
        return f"[nAnswers={self.n_answers}, nQuestions={self.n_questions}, nTargets={self.n_targets}]"


INVALID_PQA_ID = -1


class AddQuestionParam:  # reference ProbQA.py:381-387
    def __init__(self, init_amount=1.0):
        self.i_question = INVALID_PQA_ID
        self.init_amount = init_amount

    def __repr__(self):
        return "[i_question=%d, init_amount=%f]" % (self.i_question, self.init_amount)


class AddTargetParam:  # reference ProbQA.py:390-396
    def __init__(self, init_amount=1.0):
        self.i_target = INVALID_PQA_ID
        self.init_amount = init_amount

    def __repr__(self):
        return "[i_target=%d, init_amount=%f]" % (self.i_target, self.init_amount)


class PqaError:
    """Owns a native PqaError* (reference ProbQA.py:399-420)."""

    @staticmethod
    def factor(c_err) -> Optional["PqaError"]:
        if not c_err:
            return None
        return PqaError(c_err)

    def __init__(self, c_err):
        self.c_err = ctypes.c_void_p(c_err) if not isinstance(c_err, ctypes.c_void_p) else c_err

    def __del__(self):
        if self.c_err and _lib is not None:
            _lib.CiReleasePqaError(self.c_err)
            self.c_err = None

    def __repr__(self):
        return self.to_string(True)

    def to_string(self, with_params: bool) -> str:
        p = _lib.PqaError_ToString(self.c_err, 1 if with_params else 0)
        try:
            return ctypes.cast(p, ctypes.c_char_p).value.decode("utf-8", "replace")
        finally:
            _lib.CiReleaseString(p)


class _ErrSlot(__import__("threading").local):
    """The `void **ppError` argument of the value-returning calls, one per thread (ctypes releases the GIL inside a call): allocating
    a fresh c_void_p and its reference for every NextQuestion is a quarter of a microsecond of a 21 us step."""

    def __init__(self):
        self.err = ctypes.c_void_p()
        self.ref = ctypes.byref(self.err)


_err_slot = _ErrSlot()


def _check(c_err, throw: bool = True) -> Optional[PqaError]:
    if not c_err:       # (the common case of every call: nothing to wrap)
        return None
    err = PqaError.factor(c_err)
    if err and throw:
        raise PqaException(err.to_string(True))
    return err


def host_register(address: int, n_bytes: int) -> int:
    """Make host memory (e.g. a shared-memory segment) writable by this process's GPU; returns its device address."""
    load_library()
    dev = ctypes.c_void_p()
    _check(_lib.PqaHip_HostRegister(ctypes.c_void_p(address), n_bytes, ctypes.byref(dev)))
    return dev.value


def host_unregister(address: int) -> None:
    if _lib is not None:
        _lib.PqaHip_HostUnregister(ctypes.c_void_p(address))


def pick_when_all(address: int, world: int, stride: int, flag_value: int, timeout_s: float = 30.0):
    """Host half of the shared-memory exchange: wait for all flags, then the exact global pick -> (priority, index)."""
    load_library()
    pri, idx = ctypes.c_double(), ctypes.c_int64()
    _check(_lib.PqaHip_PickWhenAll(ctypes.c_void_p(address), world, stride, flag_value, timeout_s,
                                   ctypes.byref(pri), ctypes.byref(idx)))
    return pri.value, idx.value


def _dptr(a: np.ndarray):
    return a.ctypes.data_as(_pdbl)


class PqaEngine:
    """Reference ProbQA.py:423-741 plus the additive MI355X methods."""

    def __init__(self, c_engine):
        self.c_engine = ctypes.c_void_p(c_engine)

    def __del__(self):
        self.close()

    def close(self):
        if getattr(self, "c_engine", None) and _lib is not None:
            _lib.CiReleasePqaEngine(self.c_engine)
            self.c_engine = None

    # ---- id mapping ---------------------------------------------------------------------------------------------
    def __call_id_mapping(self, fn, ids: List[int]) -> List[int]:
        arr = (ctypes.c_int64 * len(ids))(*ids)
        fn(self.c_engine, len(ids), arr)
        return list(arr)

    def question_perm_from_comp(self, ids): return self.__call_id_mapping(_lib.PqaEngine_QuestionPermFromComp, ids)
    def question_comp_from_perm(self, ids): return self.__call_id_mapping(_lib.PqaEngine_QuestionCompFromPerm, ids)
    def target_perm_from_comp(self, ids): return self.__call_id_mapping(_lib.PqaEngine_TargetPermFromComp, ids)
    def target_comp_from_perm(self, ids): return self.__call_id_mapping(_lib.PqaEngine_TargetCompFromPerm, ids)
    def quiz_perm_from_comp(self, ids): return self.__call_id_mapping(_lib.PqaEngine_QuizPermFromComp, ids)
    def quiz_comp_from_perm(self, ids): return self.__call_id_mapping(_lib.PqaEngine_QuizCompFromPerm, ids)

    def ensure_perm_quiz_greater(self, bound: int) -> bool:
        return _lib.PqaEngine_EnsurePermQuizGreater(self.c_engine, bound) != 0

    def remap_quiz_perm_id(self, src_id: int, dest_id: int, throw: bool = True) -> bool:
        ok = _lib.PqaEngine_RemapQuizPermId(self.c_engine, src_id, dest_id) != 0
        if not ok and throw:
            raise PqaException(f"Failed to remap quiz permanent ID {src_id} to {dest_id}")
        return ok

    @staticmethod
    def to_c_answered_questions(answered_questions: List[AnsweredQuestion]):
        n = len(answered_questions)
        arr = (CiAnsweredQuestion * max(n, 1))()
        for i, aq in enumerate(answered_questions):
            arr[i].iQuestion, arr[i].iAnswer = aq.i_question, aq.i_answer
        return arr, n

    # ---- reference operations -----------------------------------------------------------------------------------
    def train(self, answered_questions: List[AnsweredQuestion], i_target: int, amount: float = 1.0,
              throw: bool = True) -> Optional[PqaError]:
        arr, n = self.to_c_answered_questions(answered_questions)
        return _check(_lib.PqaEngine_Train(self.c_engine, n, arr, i_target, amount), throw)

    def get_total_questions_asked(self) -> int:
        c_err = ctypes.c_void_p()
        res = _lib.PqaEngine_GetTotalQuestionsAsked(self.c_engine, ctypes.byref(c_err))
        _check(c_err.value)
        return res

    def copy_dims(self) -> EngineDimensions:
        d = CiEngineDimensions()
        if _lib.PqaEngine_CopyDims(self.c_engine, ctypes.byref(d)) == 0:
            raise PqaException("PqaEngine_CopyDims() failed")
        return EngineDimensions(d.nAnswers, d.nQuestions, d.nTargets)

    def start_quiz(self) -> int:
        c_err = ctypes.c_void_p()
        i_quiz = _lib.PqaEngine_StartQuiz(self.c_engine, ctypes.byref(c_err))
        _check(c_err.value)
        return i_quiz

    def resume_quiz(self, answered_questions: List[AnsweredQuestion]) -> int:
        arr, n = self.to_c_answered_questions(answered_questions)
        c_err = ctypes.c_void_p()
        i_quiz = _lib.PqaEngine_ResumeQuiz(self.c_engine, ctypes.byref(c_err), n, arr)
        _check(c_err.value)
        return i_quiz

    def next_question(self, i_quiz: int) -> int:
        slot = _err_slot
        q = _lib.PqaEngine_NextQuestion(self.c_engine, slot.ref, i_quiz)
        if slot.err.value:
            _check(slot.err.value)
        return q

    def record_answer(self, i_quiz: int, i_answer: int, throw: bool = True) -> Optional[PqaError]:
        return _check(_lib.PqaEngine_RecordAnswer(self.c_engine, i_quiz, i_answer), throw)

    def get_active_question_id(self, i_quiz: int) -> int:
        c_err = ctypes.c_void_p()
        q = _lib.PqaEngine_GetActiveQuestionId(self.c_engine, ctypes.byref(c_err), i_quiz)
        _check(c_err.value)
        return q

    def set_active_question(self, i_quiz: int, i_question: int, throw: bool = True) -> Optional[PqaError]:
        return _check(_lib.PqaEngine_SetActiveQuestion(self.c_engine, i_quiz, i_question), throw)

    def list_top_targets(self, i_quiz: int, max_count: int) -> List[RatedTarget]:
        arr = (CiRatedTarget * max(max_count, 1))()
        c_err = ctypes.c_void_p()
        n = _lib.PqaEngine_ListTopTargets(self.c_engine, ctypes.byref(c_err), i_quiz, max_count, arr)
        if c_err.value:
            _check(c_err.value)
        if n == 1:      # (the learner loop's ListTopTargets(1))
            return [RatedTarget(arr[0].iTarget, arr[0].prob)]
        return [RatedTarget(arr[i].iTarget, arr[i].prob) for i in range(n)]

    def record_quiz_target(self, i_quiz: int, i_target: int, amount: float = 1.0, throw: bool = True):
        return _check(_lib.PqaEngine_RecordQuizTarget(self.c_engine, i_quiz, i_target, amount), throw)

    def release_quiz(self, i_quiz: int, throw: bool = True):
        return _check(_lib.PqaEngine_ReleaseQuiz(self.c_engine, i_quiz), throw)

    def save_kb(self, file_path: str, b_double_buffer: bool, throw: bool = True):
        return _check(_lib.PqaEngine_SaveKB(self.c_engine, file_path.encode(), 1 if b_double_buffer else 0), throw)

    def start_maintenance(self, force_quizzes: bool, throw: bool = True):
        return _check(_lib.PqaEngine_StartMaintenance(self.c_engine, force_quizzes), throw)

    def finish_maintenance(self, throw: bool = True):
        return _check(_lib.PqaEngine_FinishMaintenance(self.c_engine), throw)

    def add_qs_ts(self, add_questions: List["AddQuestionParam"], add_targets: List["AddTargetParam"],
                  throw: bool = True) -> Optional[PqaError]:
        cq = (CiAddQorTParam * max(len(add_questions), 1))()
        ct = (CiAddQorTParam * max(len(add_targets), 1))()
        for i, p in enumerate(add_questions):
            cq[i].index, cq[i].initAmount = p.i_question, p.init_amount
        for i, p in enumerate(add_targets):
            ct[i].index, ct[i].initAmount = p.i_target, p.init_amount
        err = _check(_lib.PqaEngine_AddQsTs(self.c_engine, len(add_questions), cq, len(add_targets), ct), throw)
        for i, p in enumerate(add_questions):
            p.i_question = cq[i].index
        for i, p in enumerate(add_targets):
            p.i_target = ct[i].index
        return err

    def remove_questions(self, question_ids: List[int], throw: bool = True) -> Optional[PqaError]:
        arr = (ctypes.c_int64 * max(len(question_ids), 1))(*question_ids)
        return _check(_lib.PqaEngine_RemoveQuestions(self.c_engine, len(question_ids), arr), throw)

    def remove_targets(self, target_ids: List[int], throw: bool = True) -> Optional[PqaError]:
        arr = (ctypes.c_int64 * max(len(target_ids), 1))(*target_ids)
        return _check(_lib.PqaEngine_RemoveTargets(self.c_engine, len(target_ids), arr), throw)

    def compact(self) -> Tuple[List[int], List[int]]:
        n_q, n_t = ctypes.c_int64(), ctypes.c_int64()
        p_q, p_t = _pi64(), _pi64()
        _check(_lib.PqaEngine_Compact(self.c_engine, ctypes.byref(n_q), ctypes.byref(p_q), ctypes.byref(n_t),
                                      ctypes.byref(p_t)))
        try:
            return [p_q[i] for i in range(n_q.value)], [p_t[i] for i in range(n_t.value)]
        finally:
            _lib.CiReleaseCompaction(p_q)
            _lib.CiReleaseCompaction(p_t)

    def shutdown(self, save_file_path: Optional[str] = None, throw: bool = True):
        p = save_file_path.encode() if save_file_path else None
        return _check(_lib.PqaEngine_Shutdown(self.c_engine, p), throw)

    def clear_old_quizzes(self, max_count: int, max_age_sec: float, throw: bool = True):
        return _check(_lib.PqaEngine_ClearOldQuizzes(self.c_engine, max_count, max_age_sec), throw)

    # ---- additive (include/PqaHipExt.h) -----------------------------------------------------------------------------
    def set_option(self, name: str, value: int):
        _check(_lib.PqaHip_SetOption(self.c_engine, name.encode(), int(value)))

    def get_option(self, name: str) -> int:
        return _lib.PqaHip_GetOption(self.c_engine, name.encode())

    def eval_kernel_name(self) -> str:
        return _lib.PqaHip_EvalKernelName(self.c_engine).decode()

    def set_kb(self, A: np.ndarray, D: np.ndarray, B: np.ndarray):
        A = np.ascontiguousarray(A, dtype=np.float64)
        D = np.ascontiguousarray(D, dtype=np.float64)
        B = np.ascontiguousarray(B, dtype=np.float64)
        _check(_lib.PqaHip_SetKB(self.c_engine, _dptr(A), _dptr(D), _dptr(B)))

    def get_kb(self, n_local_questions: Optional[int] = None) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
        d = self.copy_dims()
        q = n_local_questions if n_local_questions is not None else d.n_questions
        A = np.empty((q, d.n_answers, d.n_targets)); D = np.empty((q, d.n_targets)); B = np.empty(d.n_targets)
        _check(_lib.PqaHip_GetKB(self.c_engine, _dptr(A), _dptr(D), _dptr(B)))
        return A, D, B

    def fill_synthetic(self, n_train: float, noise_amp: float, seed: int):
        _check(_lib.PqaHip_FillSynthetic(self.c_engine, n_train, noise_amp, seed))

    def set_target_gaps(self, targets: List[int]):
        arr = (ctypes.c_int64 * max(len(targets), 1))(*targets)
        _check(_lib.PqaHip_SetTargetGaps(self.c_engine, len(targets), arr))

    def set_question_gaps(self, questions: List[int]):
        arr = (ctypes.c_int64 * max(len(questions), 1))(*questions)
        _check(_lib.PqaHip_SetQuestionGaps(self.c_engine, len(questions), arr))

    def eval_priorities(self, i_quiz: int, n_local_questions: Optional[int] = None) -> np.ndarray:
        n = n_local_questions if n_local_questions is not None else self.copy_dims().n_questions
        out = np.empty(n, dtype=np.float64)
        _check(_lib.PqaEngine_EvalPriorities(self.c_engine, i_quiz, _dptr(out), n))
        return out

    def next_question_argmax(self, i_quiz: int) -> int:
        slot = _err_slot
        q = _lib.PqaEngine_NextQuestionArgmax(self.c_engine, slot.ref, i_quiz)
        if slot.err.value:
            _check(slot.err.value)
        return q

    def next_question_argmax_batch(self, quizzes) -> List[int]:
        """Argmax NextQuestion of several distinct quizzes with one launch; -1 for a quiz that has run out of questions."""
        n = len(quizzes)
        qs = (ctypes.c_int64 * max(n, 1))(*quizzes)
        out = (ctypes.c_int64 * max(n, 1))()
        _check(_lib.PqaEngine_NextQuestionArgmaxBatch(self.c_engine, n, qs, out))
        return list(out[:n])

    def select_argmax_batch(self, quizzes) -> np.ndarray:
        """This engine's (shard's) winners of a batch: array [n, 2] of (priority, GLOBAL question index or -1)."""
        n = len(quizzes)
        qs = (ctypes.c_int64 * max(n, 1))(*quizzes)
        out = (CiHipSelection * max(n, 1))()
        _check(_lib.PqaHip_SelectArgmaxBatch(self.c_engine, n, qs, out))
        return np.array([(out[i].priority, out[i].iQuestion) for i in range(n)], dtype=np.float64).reshape(n, 2)

    def eval_priorities_batch(self, quizzes, n_local_questions: Optional[int] = None) -> np.ndarray:
        """Priority vectors [len(quizzes), Q] of several distinct quizzes from one sweep that reads the cube once."""
        n = len(quizzes)
        nq = n_local_questions if n_local_questions is not None else self.copy_dims().n_questions
        qs = (ctypes.c_int64 * max(n, 1))(*quizzes)
        out = np.zeros((n, nq), dtype=np.float64)
        _check(_lib.PqaEngine_EvalPrioritiesBatch(self.c_engine, n, qs, _dptr(out)))
        return out

    def enqueue_select_argmax_flag(self, i_quiz: int, out_dev: int, flag_dev: int, flag_value: int) -> None:
        _check(_lib.PqaHip_EnqueueSelectArgmaxFlag(self.c_engine, i_quiz, ctypes.c_void_p(out_dev),
                                                   ctypes.c_void_p(flag_dev), flag_value))

    def select_through_slots(self, i_quiz: int, slots_host: int, slots_dev: int, rank: int, world: int, stride: int,
                             flag_value: int, timeout_s: float = 30.0):
        """One step of the shared-memory exchange (enqueue_select_argmax_flag + pick_when_all) -> (priority, index)."""
        pri, idx = ctypes.c_double(), ctypes.c_int64()
        _check(_lib.PqaHip_SelectThroughSlots(self.c_engine, i_quiz, ctypes.c_void_p(slots_host), ctypes.c_void_p(slots_dev),
                                              rank, world, stride, flag_value, timeout_s, ctypes.byref(pri), ctypes.byref(idx)))
        return pri.value, idx.value

    def select_argmax_rccl(self, i_quiz: int, nccl_comm: int, world: int):
        """This shard's winner all-gathered over the caller's RCCL communicator (an ncclComm_t as an integer) -> (priority, index)."""
        pri, idx = ctypes.c_double(), ctypes.c_int64()
        _check(_lib.PqaHip_SelectArgmaxRccl(self.c_engine, i_quiz, ctypes.c_void_p(nccl_comm), world, ctypes.byref(pri), ctypes.byref(idx)))
        return pri.value, idx.value

    def log2hot(self, x: np.ndarray) -> np.ndarray:
        """The device's Log2Hot over an array (the per-element function of the sweep)."""
        xin = np.ascontiguousarray(x, dtype=np.float64)
        out = np.empty_like(xin)
        _check(_lib.PqaHip_Log2Hot(self.c_engine, _dptr(xin), _dptr(out), xin.size))
        return out

    def next_question_sampled(self, i_quiz: int, rnd: int) -> int:
        c_err = ctypes.c_void_p()
        q = _lib.PqaEngine_NextQuestionSampled(self.c_engine, ctypes.byref(c_err), i_quiz, rnd)
        _check(c_err.value)
        return q

    def get_priors(self, i_quiz: int) -> np.ndarray:
        out = np.empty(self.copy_dims().n_targets, dtype=np.float64)
        _check(_lib.PqaHip_GetPriors(self.c_engine, i_quiz, _dptr(out), out.size))
        return out

    def get_stream(self) -> int:
        return _lib.PqaHip_GetStream(self.c_engine) or 0

    def set_stream(self, stream: int):
        _check(_lib.PqaHip_SetStream(self.c_engine, ctypes.c_void_p(stream)))

    def synchronize(self):
        _check(_lib.PqaHip_Synchronize(self.c_engine))

    def quiesce(self):
        """Everything this engine put on the device has finished; a resident sweep kernel stays (include/PqaHipExt.h)."""
        _check(_lib.PqaHip_Quiesce(self.c_engine))

    def enqueue_select_argmax(self, i_quiz: int, out_ptr: int = 0):
        _check(_lib.PqaHip_EnqueueSelectArgmax(self.c_engine, i_quiz, ctypes.c_void_p(out_ptr)))

    def enqueue_eval(self, i_quiz: int):
        _check(_lib.PqaHip_EnqueueEval(self.c_engine, i_quiz))

    def prior_device_ptr(self, i_quiz: int) -> Tuple[int, int]:
        dev = ctypes.c_void_p()
        ld = ctypes.c_int64()
        _check(_lib.PqaHip_GetPriorDevicePtr(self.c_engine, i_quiz, ctypes.byref(dev), ctypes.byref(ld)))
        return dev.value or 0, ld.value

    def record_answer_batch(self, quizzes, answers):
        """RecordAnswer of several quizzes (each with an active question) in one launch."""
        n = len(quizzes)
        qs = (ctypes.c_int64 * max(n, 1))(*quizzes)
        ans = (ctypes.c_int64 * max(n, 1))(*answers)
        _check(_lib.PqaEngine_RecordAnswerBatch(self.c_engine, n, qs, ans))

    def start_quiz_batch(self, n: int) -> List[int]:
        """n new quizzes, one launch for their priors."""
        out = (ctypes.c_int64 * max(n, 1))()
        _check(_lib.PqaEngine_StartQuizBatch(self.c_engine, n, out))
        return list(out[:n])

    def list_top_targets_batch(self, quizzes, max_count: int) -> List[List[RatedTarget]]:
        """ListTopTargets of several quizzes with one launch sequence; no posterior leaves the device."""
        n = len(quizzes)
        qs = (ctypes.c_int64 * max(n, 1))(*quizzes)
        counts = (ctypes.c_int64 * max(n, 1))()
        arr = (CiRatedTarget * max(n * max_count, 1))()
        _check(_lib.PqaEngine_ListTopTargetsBatch(self.c_engine, n, qs, max_count, arr, counts))
        return [[RatedTarget(arr[i * max_count + j].iTarget, arr[i * max_count + j].prob) for j in range(counts[i])] for i in range(n)]

    def record_answer_remote(self, i_quiz: int, i_answer: int):
        _check(_lib.PqaHip_RecordAnswerRemote(self.c_engine, i_quiz, i_answer))


class PqaEngineFactory:
    """Reference ProbQA.py:743-783."""

    def __init__(self):
        load_library()
        self.c_factory = ctypes.c_void_p(_lib.CiGetPqaEngineFactory())

    def create_cpu_engine(self, eng_def: EngineDefinition) -> Tuple[Optional[PqaEngine], Optional[PqaError]]:
        c_def = eng_def.to_c()
        c_err = ctypes.c_void_p()
        c_engine = _lib.PqaEngineFactory_CreateCpuEngine(self.c_factory, ctypes.byref(c_err), ctypes.byref(c_def))
        return (PqaEngine(c_engine) if c_engine else None), PqaError.factor(c_err.value)

    def create_hip_engine(self, eng_def: EngineDefinition, q_first: int = 0, q_total: Optional[int] = None,
                          device: int = -1) -> PqaEngine:
        c_def = eng_def.to_c()
        shard = CiHipShard(q_first, q_total if q_total is not None else eng_def.n_questions, device, 0)
        c_err = ctypes.c_void_p()
        c_engine = _lib.PqaEngineFactory_CreateHipEngineSharded(self.c_factory, ctypes.byref(c_err),
                                                                ctypes.byref(c_def), ctypes.byref(shard))
        _check(c_err.value)
        return PqaEngine(c_engine)

    def load_cpu_engine(self, file_path: str, mem_pool_max_bytes: int = EngineDefinition.DEFAULT_MEM_POOL_MAX_BYTES):
        c_err = ctypes.c_void_p()
        c_engine = _lib.PqaEngineFactory_LoadCpuEngine(self.c_factory, ctypes.byref(c_err), file_path.encode(),
                                                       mem_pool_max_bytes)
        return (PqaEngine(c_engine) if c_engine else None), PqaError.factor(c_err.value)


class MaintenanceLock:  # reference ProbQA.py:786-796
    def __init__(self, engine: PqaEngine, force_quizzes: bool):
        self.engine, self.force_quizzes = engine, force_quizzes

    def __enter__(self):
        self.engine.start_maintenance(self.force_quizzes)

    def __exit__(self, exc_type, exc_value, exc_trace):
        self.engine.finish_maintenance()
