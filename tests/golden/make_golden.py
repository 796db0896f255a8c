"""Generate the golden fixtures of tests/golden/ from the CPU oracle.

The reference holds NO golden priorities / posteriors / selections for this path and cannot be built or
imported here (MSVC/Win32 only), so these vectors are produced by oracle/pqa_oracle.c -- whose pieces are pinned against
the reference's own known-answer tests in tests/test_oracle.py -- and committed so that (a) the oracle cannot drift
silently and (b) the GPU tests have fixed expected outputs.  Inputs are regenerated from the seed in the .json
(probqa_amd/synth.py); outputs live in the .npz:
    priors_<s>      posterior vector after step s (s = 0: StartQuiz, then one RecordAnswer per step)
    priority_<s>    per-question priority vector (0 for gap / asked questions)
    run_<s>         per-subtask Kahan running sums (the reference's _pRunLength), 128 subtasks
    argmax_<s>, margin_<s>   index of the maximum and the relative gap to the runner-up
    sampled_<s>     questions picked by the reference's selector for the random numbers in RNDS
    resume_priors   posterior of ResumeQuiz(all answers), and resume_priors_bug with CEUpdatePriorsSubtaskMul.cpp:53 quirk

Run:  python tests/golden/make_golden.py
"""
from __future__ import annotations

import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import cases  # noqa: E402

RNDS = [0, 1, 2**63, 2**64 - 1, 0x9E3779B97F4A7C15, 0x1234567890ABCDEF, 0xDEADBEEFCAFEF00D]
SUBTASKS = 8 * cases.WORKERS


def case_from_meta(meta) -> cases.Case:
    return cases.Case(meta["name"], meta["K"], meta["Q"], meta["T"], meta["seed"], meta["init"], meta["n_train"],
                      meta["noise"], meta["tgaps"], meta["qgaps"], [tuple(a) for a in meta["answers"]])


def meta_from_case(c: cases.Case):
    return dict(name=c.name, K=c.K, Q=c.Q, T=c.T, seed=c.seed, init=c.init, n_train=c.n_train, noise=c.noise,
                tgaps=c.tgaps, qgaps=c.qgaps, answers=[list(a) for a in c.answers], workers=cases.WORKERS,
                subtasks=SUBTASKS, rnds=[str(r) for r in RNDS])


def run_case(c: cases.Case):
    out = {}
    o = c.make_oracle()
    o.start_quiz(cases.WORKERS)
    for step in range(len(c.answers) + 1):
        out[f"priors_{step}"] = o.priors()
        run, pri = o.eval(SUBTASKS)
        out[f"priority_{step}"], out[f"run_{step}"] = pri, run
        srt = np.sort(pri)[::-1]
        out[f"argmax_{step}"] = np.array(o.select_argmax(pri))
        out[f"margin_{step}"] = np.array((srt[0] - srt[1]) / srt[0] if srt[0] > 0 else 0.0)
        out[f"sampled_{step}"] = np.array([o.select_sampled(run, SUBTASKS, r) for r in RNDS])
        if step < len(c.answers):
            o.record_answer(*c.answers[step], cases.WORKERS - 1)
    if c.answers:
        for bug in (False, True):
            o2 = c.make_oracle()
            o2.resume_quiz(c.answers, cases.WORKERS, bug)
            out["resume_priors_bug" if bug else "resume_priors"] = o2.priors()
    return out


def main():
    for c in cases.small_cases():
        json.dump(meta_from_case(c), open(os.path.join(HERE, c.name + ".json"), "w"), indent=1)
        np.savez_compressed(os.path.join(HERE, c.name + ".npz"), **run_case(c))
        print("wrote", c.name)


if __name__ == "__main__":
    main()
