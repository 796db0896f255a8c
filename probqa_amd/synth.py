"""Deterministic synthetic knowledge bases (numpy), bit-identical to the device generator `fill_synth_kernel`
(probqa_amd/csrc/kb_kernels.hip).

Definition ("binary-search-trained + noise"; the answer rule is the one of the reference's
PqaCoreTests/DichotomyTest.cpp:50-64 with the +-32 window scaled by T/1000):

    x(q)   = floor(q * T / Qtotal)                 w = max(1, floor(32 * T / 1000))
    ans    = 0 if t < x-w; 1 if x-w <= t < x; 2 if t == x; 3 if x < t <= x+w; 4 otherwise   (clamped to K-1)
    u(i)   = (splitmix64(seed + i * 0x9E3779B97F4A7C15) >> 11) * 2^-53          in [0,1)
    a      = init (+ nTrain if k == ans) + noiseAmp * u((q*K + k)*T + t)         added in this order
    A[q,k,t] = a*a        (the cube stores squares, reference PqaCore/CETrainOperation.cpp:15-25)
    D[q,t]   = ((A[q,0,t] + A[q,1,t]) + ...) in k order
    B[t]     = (init + nTrain) + noiseAmp * u'(t),   u' uses seed ^ 0x5851F42D4C957F2D

Only IEEE fp64 multiply/add, so host and device agree to the bit.
"""
from __future__ import annotations

import numpy as np

_M = np.uint64(0xFFFFFFFFFFFFFFFF)
_GOLD = np.uint64(0x9E3779B97F4A7C15)


def splitmix64(x: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        z = (x + _GOLD) & _M
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M
        return z ^ (z >> np.uint64(31))


def hash_unit(seed: int, idx: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        x = (np.uint64(seed) + idx.astype(np.uint64) * _GOLD) & _M
    return (splitmix64(x) >> np.uint64(11)).astype(np.float64) * 2.0 ** -53


def synthetic_kb(K: int, Q: int, T: int, init: float, n_train: float, noise_amp: float, seed: int,
                 q_offset: int = 0, q_total: int | None = None):
    """Return (A[Q,K,T], D[Q,T], B[T]) for questions q_offset .. q_offset+Q of a KB with q_total questions."""
    q_total = Q if q_total is None else q_total
    w = max(1, (32 * T) // 1000)
    qg = (np.arange(Q, dtype=np.int64) + q_offset)[:, None]          # [Q,1]
    t = np.arange(T, dtype=np.int64)[None, :]                         # [1,T]
    x = (qg * T) // q_total
    ans = np.where(t < x - w, 0, np.where(t < x, 1, np.where(t == x, 2, np.where(t <= x + w, 3, 4))))
    ans = np.minimum(ans, K - 1)
    A = np.empty((Q, K, T), dtype=np.float64)
    D = np.zeros((Q, T), dtype=np.float64)
    for k in range(K):
        a = np.full((Q, T), init, dtype=np.float64)
        a = np.where(ans == k, a + n_train, a)
        idx = ((qg * K + k) * T + t).astype(np.uint64)
        a = a + noise_amp * hash_unit(seed, idx)
        A[:, k, :] = a * a
        D = D + A[:, k, :]
    B = (init + n_train) + noise_amp * hash_unit(seed ^ 0x5851F42D4C957F2D, np.arange(T, dtype=np.uint64))
    return A, D, B


def dichotomy_answer(i_question: int, guess: int, width: int = 32) -> int:
    """The trainer's answer rule, reference PqaCoreTests/DichotomyTest.cpp:50-64."""
    if guess < i_question - width:
        return 0
    if guess < i_question:
        return 1
    if guess == i_question:
        return 2
    if guess <= i_question + width:
        return 3
    return 4
