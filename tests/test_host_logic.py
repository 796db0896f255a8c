"""Host bookkeeping of the engine that needs no device (runs in the CPU suite): the compact <-> permanent id map and the
eviction rule of ClearOldQuizzes, driven through PqaHip_HostLogicProbe and held to a Python model of the reference's
observable behaviour (PqaCore/PermanentIdManager.cpp, PqaCore/BaseEngine.cpp:814-873)."""
import ctypes
import random

import numpy as np
import pytest

from probqa_amd import interop


def probe(what, words, n_out):
    lib = interop.load_library()
    arr = np.ascontiguousarray(words, dtype=np.int64)
    out = np.zeros(max(1, n_out), dtype=np.int64)
    n = lib.PqaHip_HostLogicProbe(what.encode(), arr.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)), len(arr),
                                  out.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)), n_out)
    return n, out[:max(0, n)].tolist()


class LedgerModel:
    """What a caller of the reference's id manager can observe."""

    def __init__(self):
        self.next = 0
        self.perm = []        # slot -> permanent id or -1
        self.slot = {}        # permanent id -> slot

    def run(self, op, a, b, extra):
        if op == 0:
            return self.perm[a] if 0 <= a < len(self.perm) else -1
        if op == 1:
            return self.slot.get(a, -1)
        if op == 2:
            if self.next <= a:
                self.next = a + 1
                return 1
            return 0
        if op == 3:
            if not (0 <= a < len(self.perm)) or self.perm[a] == -1:
                return 0
            del self.slot[self.perm[a]]
            self.perm[a] = -1
            return 1
        if op == 4:
            if not (0 <= a < len(self.perm)) or self.perm[a] != -1:
                return 0
            self.perm[a] = self.next
            self.slot[self.next] = a
            self.next += 1
            return 1
        if op == 5:
            if a < len(self.perm):
                return 0
            while len(self.perm) < a:
                self.slot[self.next] = len(self.perm)
                self.perm.append(self.next)
                self.next += 1
            return 1
        if op == 6:
            if b < 0 or b >= self.next or b in self.slot or a not in self.slot:
                return 0
            s = self.slot.pop(a)
            self.slot[b] = s
            self.perm[s] = b
            return 1
        if op == 7:
            live = [s for s, p in enumerate(self.perm) if p != -1]
            if a != len(live) or sorted(extra) != live:
                return 0
            self.perm = [self.perm[s] for s in extra]
            self.slot = {p: i for i, p in enumerate(self.perm)}
            return 1
        if op == 8:
            return 1
        raise AssertionError(op)


def test_id_ledger_follows_the_reference_sequence():
    """The sequence tests/test_gpu_kb.py drives through an engine, here without one: fresh ids, LIFO slot reuse under new
    permanent ids, the raised floor, a rename into the past, compaction."""
    script = [5, 5, 0,     # five slots: permanent 0..4
              3, 1, 0,     # vacate slot 1
              4, 1, 0,     # reissue: permanent 5
              0, 1, 0, 1, 1, 0, 1, 5, 0,
              2, 100, 0, 2, 50, 0,
              5, 6, 0,     # slot 5: permanent 101
              0, 5, 0,
              6, 101, 77, 1, 77, 0, 1, 101, 0,
              6, 77, 500,  # not into the future
              6, 0, 2,     # not onto a live id
              3, 0, 0, 3, 2, 0,
              7, 4, 4, 5, 1, 3, 4,   # slots 5, 1, 3, 4 survive, in that order
              0, 0, 0, 0, 1, 0, 0, 3, 0, 0, 4, 0, 1, 77, 0, 1, 2, 0,
              8, 0, 0, 1, 5, 0, 5, 5, 0, 0, 4, 0]
    n, out = probe("id_ledger", script, 64)
    assert n == 28
    assert out == [1, 1, 1, 5, -1, 1, 1, 0, 1, 101, 1, 5, -1, 0, 0, 1, 1, 1, 77, 5, 4, -1, 0, -1, 1, 1, 1, 102]
    # (the last one: the issue floor survives the file -- the slot added after the round trip continues the sequence)


@pytest.mark.parametrize("seed", range(12))
def test_id_ledger_against_model(seed):
    rng = random.Random(seed)
    model = LedgerModel()
    script, expect = [], []

    def emit(op, a=0, b=0, extra=()):
        script.extend([op, a, b, *extra])
        expect.append(model.run(op, a, b, list(extra)))

    emit(5, rng.randrange(0, 40))
    for _ in range(3000):
        n = len(model.perm)
        r = rng.random()
        if r < 0.25:
            emit(3, rng.randrange(-1, n + 2))
        elif r < 0.45:
            emit(4, rng.randrange(-1, n + 2))
        elif r < 0.55:
            emit(5, n + rng.randrange(-1, 4))
        elif r < 0.63:
            emit(2, model.next + rng.randrange(-3, 3))
        elif r < 0.75:
            src = rng.choice(list(model.slot)) if model.slot and rng.random() < 0.8 else rng.randrange(-1, model.next + 2)
            emit(6, src, rng.randrange(-1, model.next + 2))
        elif r < 0.78:
            live = [s for s, p in enumerate(model.perm) if p != -1]
            rng.shuffle(live)
            if rng.random() < 0.2 and live:
                live[0] = rng.randrange(-1, n + 1)     # a bad list now and then: nothing may change
            emit(7, len(live), len(live), live)
        elif r < 0.80:
            emit(8)
        elif r < 0.82:
            script.extend([9, 0, 0])
            expect.append(len(model.slot))
        elif r < 0.90:
            emit(0, rng.randrange(-2, n + 2))
        else:
            emit(1, rng.randrange(-2, model.next + 2))
    # the whole map at the end, both directions
    for s in range(len(model.perm)):
        emit(0, s)
    for p in range(model.next):
        emit(1, p)
    n, out = probe("id_ledger", script, len(expect))
    assert n == len(expect)
    bad = [i for i in range(n) if out[i] != expect[i]]
    assert not bad, (bad[:5], [script[:0]])


def test_malformed_scripts_are_refused():
    assert probe("id_ledger", [5, 3], 4)[0] == -1
    assert probe("id_ledger", [10, 0, 0], 4)[0] == -1
    assert probe("id_ledger", [7, 3, 3, 0, 1], 4)[0] == -1      # the inline list is cut short
    assert probe("id_ledger", [5, 3, 0, 5, 4, 0], 1)[0] == -1    # output too small
    assert probe("nothing", [], 1)[0] == -1
    assert probe("let_go", [0, 1, 1, 2, 7, 0], 8)[0] == -1


def let_go(now, max_count, max_age, quizzes):
    words = [now, max_count, max_age, len(quizzes)]
    for q, t in quizzes:
        words += [q, t]
    n, out = probe("let_go", words, len(quizzes) + 1)
    assert n == out[0] + 1
    return out[1:]


def test_clear_old_quizzes_rule():
    now = 1_000_000
    quizzes = [(0, now - 50), (1, now - 5), (3, now - 400), (4, now - 5), (7, now - 20), (9, now)]
    assert let_go(now, 10, 1e9, quizzes) == []
    assert let_go(now, 10, 100, quizzes) == [3]                       # older than 100 s
    assert let_go(now, 3, 100, quizzes) == [3, 0, 7]                  # then the longest-unused until three remain
    assert let_go(now, 2, 1e9, quizzes) == [3, 0, 7, 1]               # equal ages: registry order
    assert let_go(now, 0, 1e9, quizzes) == [3, 0, 7, 1, 4, 9]
    assert let_go(now, 5, 0, quizzes) == [0, 1, 3, 4, 7]              # everything used before `now`, in registry order
    assert let_go(now, 0, 10, []) == []


@pytest.mark.parametrize("seed", range(6))
def test_clear_old_quizzes_rule_random(seed):
    rng = random.Random(100 + seed)
    now = 2_000_000
    ids = sorted(rng.sample(range(500), rng.randrange(1, 200)))
    quizzes = [(q, now - rng.randrange(0, 30)) for q in ids]
    max_count, max_age = rng.randrange(0, 220), rng.randrange(0, 30)
    gone = let_go(now, max_count, max_age, quizzes)
    aged = [q for q, t in quizzes if now - t > max_age]
    assert gone[:len(aged)] == aged
    rest = [(q, t) for q, t in quizzes if now - t <= max_age]
    extra = gone[len(aged):]
    assert len(extra) == max(0, len(rest) - max_count) and len(set(gone)) == len(gone)
    kept = [(q, t) for q, t in rest if q not in extra]
    age = dict(quizzes)
    if extra and kept:
        assert max(age[q] for q in extra) <= min(t for _, t in kept)      # nobody kept is older than somebody released
    assert [age[q] for q in extra] == sorted(age[q] for q in extra)       # longest-unused first
