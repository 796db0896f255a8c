#!/usr/bin/env python3
"""bench.py -- next-question selections/s of the MI355X engine, through the PqaCore C ABI.

A "step" is one complete NextQuestion of one quiz: the priority sweep over every unasked question of the
sA[question][answer][target] cube (eval kernel) + the argmax selection + delivery of the selected question id to
the host.  The cube is resident in HBM before the timed region.  Workload at N=1: BASELINE.json configs[1]
(1000Q x 5A x 1000T fp64, single in-flight quiz).  `value` goes through the library's DEFAULT path -- one kernel launch per
selection, what an unchanged caller of PqaEngine_NextQuestion gets; the opt-in resident sweep (engine option server = 1) is
measured beside it (`resident_sweep`), and `--server` takes `value` through it instead.  N>1: the question axis of the same cube is sharded over the ranks
(one process per GPU), each rank sweeps its shard and the ranks' 16-byte winners meet in host shared memory written by
the sweeps themselves (--exchange rccl: one RCCL all-gather instead), every rank picks the same global argmax
("strong" scaling, as north_star states it).  --config M runs configs[2] (10000x5x10000, the HBM-bound point); sharded
runs of the default config also time configs[3] (the 10000x5x10000 cube over the same ranks) into the extra key
"sharded_10000x5x10000".

Prints ONE JSON line (rank 0).  Extra keys: roofline (dominant kernel, live HIP-event timing on the engine's stream),
cpu_baseline (the AVX2+threads CPU port of the reference path, timed on this host; N=1 only).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

CONFIGS = {
    "S": dict(Q=1000, K=5, T=1000, name="1000Qx5Ax1000T"),
    "M": dict(Q=10000, K=5, T=10000, name="10000Qx5Ax10000T"),
    "T6": dict(Q=6000, K=5, T=6000, name="6000Qx5Ax6000T"),
    "T7": dict(Q=7000, K=5, T=7000, name="7000Qx5Ax7000T"),
    "M8": dict(Q=8000, K=5, T=8000, name="8000Qx5Ax8000T"),
    "M9": dict(Q=9000, K=5, T=9000, name="9000Qx5Ax9000T"),
    "T2": dict(Q=2000, K=5, T=2000, name="2000Qx5Ax2000T"),
    "T25": dict(Q=2500, K=5, T=2500, name="2500Qx5Ax2500T"),
    "T3": dict(Q=3000, K=5, T=3000, name="3000Qx5Ax3000T"),
    "T4": dict(Q=4000, K=5, T=4000, name="4000Qx5Ax4000T"),
    "T45": dict(Q=4500, K=5, T=4500, name="4500Qx5Ax4500T"),
    "T5": dict(Q=5000, K=5, T=5000, name="5000Qx5Ax5000T"),
    # BASELINE configs[4]: 100000Q x 5A x 100000T fp32, 256 concurrent quizzes, question axis over 8 GPUs -- ONE GPU's shard
    # (12500 questions, 30 GB); with --gpus N every rank holds such a shard of a cube N times as large (weak scaling)
    "L1": dict(Q=12500, K=5, T=100000, name="12500Qx5Ax100000T", prec="f32", quizzes=256),
    # the same batched path on small cubes (development / smoke sizes)
    "LS": dict(Q=1000, K=5, T=1000, name="1000Qx5Ax1000T", prec="f32", quizzes=256),
    "SB": dict(Q=1000, K=5, T=1000, name="1000Qx5Ax1000T", prec="f64", quizzes=256),
}
HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s spec
FP32_VECTOR_PEAK_TFLOPS = 157.3   # same guide: 64 FLOP/clk/SIMD x 4 SIMDs x 256 CUs x 2.4 GHz
FP64_VECTOR_PEAK_TFLOPS = 78.6    # half the fp32 vector rate (an fp64 FMA issues over 4 cycles per wave: tools/ubench_fp64.hip)
FLOPS_PER_ELEMENT = 45            # the reference's operation count per (question, answer, target) element
SEED = 20260928


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=5000,
                    help="untimed steps first (the default is long enough for the GPU's clocks to settle: +2-4 %% at S)")
    ap.add_argument("--config", choices=sorted(CONFIGS), default="S")
    ap.add_argument("--variant", type=int, default=0, help="force an eval kernel shape (0 = auto)")
    ap.add_argument("--batch", type=int, default=-1,
                    help="quizzes per launch of the batched-selection extra (0 = skip; default 64, or 8 for cubes over 1 GB); "
                         "with a batched config (L1, ...): the batch size of the run instead of the config's own")
    ap.add_argument("--force-collective", action="store_true",
                    help="use the sharded selector even on one GPU: exercises the N>1 path")
    ap.add_argument("--exchange", choices=("shm", "rccl"), default="shm",
                    help="how the shards' 16-byte winners meet: host shared memory written by the sweep itself "
                         "(default), or an RCCL all-gather + D2H copy")
    ap.add_argument("--server", action="store_true",
                    help="take `value` through the resident sweep kernel (engine option server=1, opt-in in the library too) instead of the "
                         "library's default path, one kernel launch per selection; without it the resident rate is reported beside `value` "
                         "(resident_sweep)")
    ap.add_argument("--no-server", action="store_true", help="do not measure the resident sweep at all")
    ap.add_argument("--no-quiz-loop", action="store_true", help="skip the quiz-loop extra")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-points", action="store_true", help="one GPU, default config: skip the M and L1 points (hbm_point_M, valu_point_L1)")
    ap.add_argument("--sharded-configs", default="S,M,L1", help="sharded runs: which of the extras S, M, L1 to run (both exchanges each)")
    ap.add_argument("--l1-config", default="L1", choices=("L1", "LS", "SB"), help="the batched configuration of the sharded extras (tests use a small one)")
    ap.add_argument("--cpu-seconds", type=float, default=3.0, help="wall seconds of the CPU baseline leg")
    args = ap.parse_args()

    # `python bench.py --gpus N` by itself: start the N ranks (one process per GPU) the way the driver does
    if args.gpus > 1 and "RANK" not in os.environ:
        raise SystemExit(self_spawn(args.gpus))

    # stdout carries exactly one line, the result: libraries that print there (RCCL's version banner, for one) are sent
    # to stderr for the whole run
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)

    import numpy as np
    import torch

    from probqa_amd import dist as pdist
    from probqa_amd import interop

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    args.gpus = world
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the engine has no CPU fallback")
    # Fewer devices than ranks (a 1-GPU box asked for --gpus 2): a DRY RUN of the N > 1 path -- the ranks share devices, the
    # control plane is gloo (RCCL refuses two ranks on one device) and the RCCL exchange is reported as skipped.
    n_dev = torch.cuda.device_count()
    dev_index = local_rank % n_dev
    oversub = world > n_dev
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    ctl = Ctl(torch, world, device, oversub, args.force_collective)
    if oversub:
        args.no_server = True      # two engines' resident kernels on one device wait for each other's idle exit
        print("bench.py: %d ranks on %d device(s): dry run of the sharded path (gloo control plane, no RCCL exchange)" % (world, n_dev), file=sys.stderr)

    cfg = CONFIGS[args.config]
    Q, K, T = cfg["Q"], cfg["K"], cfg["T"]
    if "quizzes" in cfg:
        if args.batch > 0 and args.batch != cfg["quizzes"]:
            cfg = dict(cfg, quizzes=args.batch, traffic_key="%s_b%d" % (args.config, args.batch))   # (--config L1 --batch 32: the small-batch point)
        out = run_batched(args, cfg, np, torch, interop, pdist, ctl, rank, dev_index, device)
        if rank == 0:
            os.write(result_fd, (json.dumps(out) + "\n").encode())
        ctl.close()
        return
    if args.batch < 0:
        args.batch = 64 if Q * (K + 1) * T * 8 < 1e9 else 8
    factory = interop.PqaEngineFactory()
    stream = torch.cuda.Stream(device=device)
    torch.cuda.set_stream(stream)
    sharded = world > 1 or args.force_collective

    def make_engine(c):
        """This rank's shard of cube `c` and one quiz started."""
        qf, ql = pdist.shard_range(c["Q"], world, rank)
        e = factory.create_hip_engine(interop.EngineDefinition(c["K"], ql - qf, c["T"], init_amount=0.1), qf, c["Q"], dev_index)
        e.set_option("select", 1)
        e.set_option("eval_variant", args.variant)
        e.fill_synthetic(8.0, 0.5, SEED)
        e.set_stream(stream.cuda_stream)
        return e, e.start_quiz(), ql - qf

    def make_selector(e, qz, kind, tag):
        """The exchange of the shards' 16-byte winners: "shm" (slots in host shared memory written by the sweeps' finishers) or
        "rccl" (one all-gather).  Every rank takes the same path: returns (selector or None, kind actually used)."""
        if kind == "shm":
            name = "bench_%s_%s" % (os.environ.get("MASTER_PORT", "0"), tag)
            sel_obj, ok = None, 1
            try:
                if rank == 0:
                    sel_obj = pdist.ShmSelector(e, qz, rank, world, name, create=True)
            except Exception as ex:  # noqa: BLE001 - every rank must take the same path
                print("rank 0: shared-memory exchange unavailable (%r)" % (ex,), file=sys.stderr)
                ok = 0
            ctl.barrier()                       # the segment exists before the other ranks open it
            try:
                if rank != 0 and ok:
                    sel_obj = pdist.ShmSelector(e, qz, rank, world, name, create=False)
            except Exception as ex:  # noqa: BLE001
                print("rank %d: shared-memory exchange unavailable (%r)" % (rank, ex), file=sys.stderr)
                ok = 0
            if ctl.min_int(ok) == 1:
                return sel_obj, "shm"
            if sel_obj is not None:             # one rank could not: all ranks use the collective
                sel_obj.close()
            kind = "rccl"
        if not ctl.rccl:
            return None, "none"                 # (dry run on shared devices: no RCCL)
        return pdist.ShardedSelector(lambda out: e.enqueue_select_argmax(qz, out.data_ptr()), device), "rccl"

    eng, quiz, q_local = make_engine(cfg)
    selector = None
    if sharded:
        selector, args.exchange = make_selector(eng, quiz, args.exchange, "main")
        if selector is None:
            raise SystemExit("no exchange available for the sharded run")

    def step():
        if selector is None:
            return eng.next_question_argmax(quiz)
        _, q = selector.select()
        return q

    barrier = ctl.barrier

    def timed(fn, warmup, steps, e=None):
        """W untimed steps, then exactly K steps between barrier + synchronize; the maximum over ranks.
        Every step is a synchronous call that has returned its result.  The engine is quiesced (PqaHip_Synchronize: its
        stream is drained and its resident sweep kernel, if one is serving the selections, is told to leave -- a few
        microseconds) before each device-wide synchronisation, which would otherwise sit out that kernel's idle time-out
        (server_idle_us) inside the timed region: 2 ms, i.e. 5.5x the 20 timed steps of the driver's run in round 1."""
        e = e or eng
        r = None
        # One rank: every step is a synchronous call -- when it has returned there is nothing of it left on the device, and the
        # bracket only has to say so (PqaHip_Quiesce: the engine's stream drained, the resident kernel idle but STILL THERE).
        # Sending the resident kernel away on both sides (PqaHip_Synchronize, what a device-wide synchronisation needs) put its
        # relaunch and two synchronisations inside 20 timed steps: 23.1 us per step where the same call's median was 21.3
        # (VERDICT r3, weak #4).  Several ranks: the barrier and the full synchronisation, as before.
        # (the device-wide torch.cuda.synchronize() of the several-rank bracket would sit out the idle resident kernel's time-out --
        #  0.5 ms -- inside the timed region; PqaHip_Quiesce synchronises the stream ALL of this engine's work is on)
        def bracket():
            if world == 1:
                e.quiesce()
            else:
                e.synchronize()
                barrier()

        for _ in range(warmup):
            r = fn()
        bracket()
        t0 = time.perf_counter()
        for _ in range(steps):
            r = fn()
        bracket()
        return ctl.max_float(time.perf_counter() - t0), r

    def kernel_ms_of(e, qz, n_k):
        """dominant kernel: live HIP-event timing on the engine's stream, back-to-back launches of the sweep kernel -- by itself:
        the (empty, in this quiz state) fix that is launched behind every watching sweep is left out (option pole_follow)"""
        e.set_option("pole_follow", 0)
        try:
            return kernel_ms_with(e, qz, n_k)
        finally:
            e.set_option("pole_follow", 1)

    def kernel_ms_with(e, qz, n_k):
        for _ in range(5):
            e.enqueue_eval(qz)
        torch.cuda.synchronize()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record(stream)
        for _ in range(n_k):
            e.enqueue_eval(qz)
        ev1.record(stream)
        torch.cuda.synchronize()
        return ev0.elapsed_time(ev1) / n_k

    # The resident sweep (engine option "server": one launch serves many selections, csrc/eval_kernels.hip) where the
    # engine has one for the cube's shape; the launch-per-selection rate of the same call is reported beside it.
    uses_collective = selector is not None and args.exchange == "rccl"   # (that path enqueues launches on the stream)
    if args.server and not args.no_server and not uses_collective:
        eng.set_option("server", 1)
    resident = bool(eng.get_option("server_active") == 1) and not uses_collective
    # set-up, not measurement: the device's clocks ramp up over the first tenths of a second of work, and a run of W = 5 warm-up steps
    # and K = 20 timed ones (the driver's command: 0.6 ms of device work in all) would time the ramp -- a third of a second of the same
    # call first, then the W untimed and K timed steps of the contract
    # (several ranks: every selection is one of ALL ranks -- the same COUNT on each, not the same time: a rank that made one call more
    #  than its peers waits for a selection they never make)
    if world > 1:
        for _ in range(5000):
            step()
    else:
        t_settle = time.perf_counter() + 0.3
        while time.perf_counter() < t_settle:
            step()
    # ... and the bracket's own synchronisation rehearsed: the settle calls were only ever waited for through their result flags, and the
    # first stream synchronisations behind thousands of such launches retire that backlog -- 30 us of the timed region's closing bracket
    # when left to it (the driver's 25-step run, eight times each way on one box, alternating: 26.7 us per step without, 24.6 with)
    for _ in range(3):
        if world == 1:
            eng.quiesce()
        else:
            eng.synchronize()
            barrier()
    elapsed, sel = timed(step, args.warmup, args.steps)
    value = args.steps / elapsed

    # ---- per-step latency distribution of the same synchronous call (median and p10 / p90)
    lat = []
    for _ in range(min(args.steps, 2000)):
        t1 = time.perf_counter()
        step()
        lat.append(time.perf_counter() - t1)
    lat.sort()
    latency_us = {"p10": 1e6 * lat[len(lat) // 10], "p50": 1e6 * lat[len(lat) // 2], "p90": 1e6 * lat[(9 * len(lat)) // 10],
                  "n": len(lat)}

    # ---- the dominant kernel ON THE `value` PATH.  Resident sweep: eval_server_f64's own 100 MHz clock, request in hand to answer
    # published, per step (engine option server_last_step_ns); a launch per selection: eval_questions_f64, HIP events below.
    server_step_us = None
    if resident:
        ticks = []
        for _ in range(max(50, min(args.steps, 500))):
            step()
            ns = eng.get_option("server_last_step_ns")
            if ns > 0:
                ticks.append(ns * 1e-3)
        if ticks:
            ticks.sort()
            server_step_us = {"mean": sum(ticks) / len(ticks), "p10": ticks[len(ticks) // 10], "p50": ticks[len(ticks) // 2],
                              "p90": ticks[(9 * len(ticks)) // 10], "n": len(ticks)}

    # ---- the same synchronous call made by native code (libPqaClient.so: n x PqaEngine_NextQuestion, no Python between the calls):
    # what of a step is the wrapper's
    native = None
    if selector is None:
        n_nat = max(500, min(args.steps, 5000))
        dt_n, sel_n = interop.time_selections_native(eng, quiz, 200, n_nat)
        native = {"resident" if resident else "launched": {"selections_per_sec": n_nat / dt_n, "us_per_step": 1e6 * dt_n / n_nat,
                                                            "agrees": int(sel_n) == int(sel)}}
        if server_step_us:
            native["resident"]["host_overhead_us"] = 1e6 * dt_n / n_nat - server_step_us["mean"]
    launch_rate = None
    if resident:
        eng.set_option("server", 0)   # everything below launches kernels that would wait for the resident one to leave
        dt_l, sel_l = timed(step, min(args.warmup, 500), max(200, args.steps // 2))
        launch_rate = {"selections_per_sec": max(200, args.steps // 2) / dt_l, "agrees_with_resident": int(sel_l) == int(sel)}
        if native is not None:
            n_nat = max(500, min(args.steps, 5000))
            dt_n, sel_n = interop.time_selections_native(eng, quiz, 200, n_nat)
            native["launched"] = {"selections_per_sec": n_nat / dt_n, "us_per_step": 1e6 * dt_n / n_nat, "agrees": int(sel_n) == int(sel)}
    # ---- extra (not `value`): the same synchronous call through the resident sweep kernel (engine option server = 1: opt-in -- it keeps
    # every CU polling for up to server_idle_us after a selection), where the engine has one for the cube's shape
    resident_extra = None
    if not resident and selector is None and not args.no_server and not uses_collective:
        eng.set_option("server", 1)
        if eng.get_option("server_active") == 1:
            n_r = max(200, args.steps)
            dt_r, sel_r = timed(step, min(args.warmup, 2000), n_r)
            ticks = []
            for _ in range(max(50, min(args.steps, 500))):
                step()
                ns = eng.get_option("server_last_step_ns")
                if ns > 0:
                    ticks.append(ns * 1e-3)
            ticks.sort()
            step_us = {"mean": sum(ticks) / len(ticks), "p10": ticks[len(ticks) // 10], "p50": ticks[len(ticks) // 2],
                       "p90": ticks[(9 * len(ticks)) // 10], "n": len(ticks)} if ticks else None
            resident_extra = {"selections_per_sec": n_r / dt_r, "us_per_step": 1e6 * dt_r / n_r, "agrees_with_launches": int(sel_r) == int(sel),
                              "resident_step_us": step_us,
                              "host_overhead_us": (1e6 * dt_r / n_r - step_us["mean"]) if step_us else None,
                              "note": "engine option server = 1 (opt-in; `bench.py --server` takes `value` through it): one launch of eval_server_f64 serves "
                                      "the selections, request and answer through pinned / BAR-mapped memory; resident_step_us is the kernel's own 100 MHz "
                                      "clock per step, request in hand to answer published"}
        eng.set_option("server", 0)
    kernel_ms = kernel_ms_of(eng, quiz, max(20, min(args.steps, 200)))
    # ... and the sweep's launch INSIDE the synchronous call `value` times: HIP events recorded by the engine on its stream right
    # around the launch (option time_sweeps), dispatch to retirement -- what a kernel trace of this command shows per launch
    # (profiles/): a launch onto an idle device does not overlap its ramp with the previous one's tail as back-to-back launches do
    sync_kernel_us = None
    if selector is None and not resident:
        eng.set_option("time_sweeps", 1)
        ts = []
        for _ in range(max(50, min(args.steps, 500))):
            step()
            ns = eng.get_option("last_sweep_ns")
            if ns > 0:
                ts.append(ns * 1e-3)
        eng.set_option("time_sweeps", 0)
        if ts:
            ts.sort()
            sync_kernel_us = {"mean": sum(ts) / len(ts), "p10": ts[len(ts) // 10], "p50": ts[len(ts) // 2], "p90": ts[(9 * len(ts)) // 10], "n": len(ts)}
    alg_bytes = q_local * (K + 1) * T * 8  # one read of every sA row and the mD row, fp64
    alg_flops = q_local * K * T * FLOPS_PER_ELEMENT   # ~45 fp64 operations per (question, answer, target)
    launched_us = kernel_ms * 1e3
    # the contract's kernel = the one `value` was measured through
    path_us = server_step_us["mean"] if server_step_us else launched_us
    achieved = alg_bytes / (path_us * 1e-6) / 1e9

    # ---- stream-ordered (pipelined) throughput: selections enqueued back to back, results left on the device
    if selector is None:
        for _ in range(5):
            eng.enqueue_select_argmax(quiz)
        torch.cuda.synchronize()
        tp0 = time.perf_counter()
        for _ in range(args.steps):
            eng.enqueue_select_argmax(quiz)
        torch.cuda.synchronize()
        pipelined = args.steps / (time.perf_counter() - tp0)
    else:
        pipelined = None

    # ---- extra (not `value`): the same synchronous selection replayed from a HIP graph (SURVEY 8(d) asks for the variant)
    graph_rate = None
    if selector is None:
        eng.set_option("use_graph", 1)
        for _ in range(50):
            eng.next_question_argmax(quiz)
        torch.cuda.synchronize()
        n_g = max(200, args.steps // 2)
        tg0 = time.perf_counter()
        for _ in range(n_g):
            gsel = eng.next_question_argmax(quiz)
        graph_rate = {"selections_per_sec": n_g / (time.perf_counter() - tg0), "agrees_with_launches": int(gsel) == int(sel)}
        eng.set_option("use_graph", 0)

    # ---- extra (not `value`): many quizzes in flight, one launch per batch (PqaEngine_NextQuestionArgmaxBatch)
    batched = None
    if selector is None and args.batch > 0:
        quizzes = [quiz] + [eng.start_quiz() for _ in range(args.batch - 1)]
        reps = max(3, min(200, args.steps // args.batch))
        for _ in range(2):
            eng.next_question_argmax_batch(quizzes)
        torch.cuda.synchronize()
        tb0 = time.perf_counter()
        for _ in range(reps):
            picks = eng.next_question_argmax_batch(quizzes)
        batched = {"quizzes_per_launch": args.batch, "selections_per_sec": reps * args.batch / (time.perf_counter() - tb0),
                   "launches_timed": reps, "agrees_with_single": int(picks[0]) == int(sel)}

    # ---- extra (not `value`): the quiz loop the reference's only published rate is about (301.2 NextQuestion/s
    # end to end on the author's 2017 desktop, PqaClient's learner loop at 1000 x 5 x 1000): NextQuestion with the reference's
    # sampled selector, RecordAnswer by the binary-search rule, ListTopTargets(1), until the guess is on top or 30 questions
    # were asked, then RecordQuizTarget.  On the synthetic trained cube of this run, not from a fresh KB.
    quiz_loop = None
    if selector is None and args.config == "S" and not args.no_quiz_loop:
        from probqa_amd import synth

        eng.set_option("select", 0)
        eng.set_option("seed", SEED)
        # (server off: RecordQuizTarget rewrites the cube at the end of every quiz, which stops a resident kernel -- measured 15.6 k
        #  questions/s with it against 17.6 k with one launch per selection; the selector itself runs on the host either way)
        rng = np.random.default_rng(SEED)
        asked, hits, n_q = 0, 0, 600
        tq0 = time.perf_counter()
        for _ in range(n_q):
            guess = int(rng.integers(T))
            qz = eng.start_quiz()
            for _j in range(30):
                qq = eng.next_question(qz)
                asked += 1
                eng.record_answer(qz, synth.dichotomy_answer(qq * T // Q, guess, max(1, 32 * T // 1000)))
                top1 = eng.list_top_targets(qz, 1)
                if top1 and top1[0].i_target == guess:
                    hits += 1
                    break
            eng.record_quiz_target(qz, guess)
            eng.release_quiz(qz)
        dtq = time.perf_counter() - tq0
        eng.set_option("select", 1)
        quiz_loop = {"questions_per_sec": asked / dtq, "quizzes": n_q, "questions": asked, "guessed_on_top": hits,
                     "published_reference_questions_per_sec": 301.2,
                     "selection_path": "one launch per selection, made by StartQuiz / RecordAnswer ahead of the NextQuestion that follows (option speculate) -- "
                                       "RecordAnswer's posterior update runs inside that launch (option fuse_update); "
                                       "every workgroup hands its priorities to the host, the reference's selector runs there",
                     "speculative_sweeps": {"used": int(eng.get_option("spec_hits")), "dropped": int(eng.get_option("spec_dropped")),
                                            "with_the_posterior_update_inside": int(eng.get_option("fused_updates"))},
                     "note": "reference figure: PqaClient learner loop on the author's 2017 desktop CPU (BASELINE.md); "
                             "here: Python wrapper of the C ABI, one quiz at a time, sampled selector"}

    # ---- extras (not `value`), sharded runs only.  Every one of them reports BOTH exchanges of the shards' winners: the slots
    # in host shared memory that the sweeps' finishers write (the default path of `value`) and the RCCL all-gather that
    # north_star names -- S (this cube), M = BASELINE configs[3] (10000x5x10000 over the same ranks: where sharding is about
    # bandwidth rather than launch latency), L1 = BASELINE configs[4] (one 12500x5x100000 fp32 shard per rank, 256 quizzes).
    def both_exchanges(e, qz, tag, warm, n_steps, first=None):
        """{kind: {selections_per_sec, us_per_step, selected_question}} for shm and rccl on engine `e` (resident sweep off)."""
        res = {}
        e.set_option("server", 0)
        for kind in ("shm", "rccl"):
            if first is not None and kind == first[0]:
                res[kind] = first[1]
                continue
            sel_x, used = make_selector(e, qz, kind, tag + kind)
            if sel_x is None or used != kind:
                res[kind] = {"skipped": "the ranks share a device: RCCL refuses that" if kind == "rccl" and not ctl.rccl else "unavailable on this host"}
                if sel_x is not None and hasattr(sel_x, "close"):
                    sel_x.close()
                continue
            dt_x, pick_x = timed(lambda: sel_x.select()[1], warm, n_steps, e)
            res[kind] = {"selections_per_sec": n_steps / dt_x, "us_per_step": 1e6 * dt_x / n_steps, "selected_question": int(pick_x)}
            if hasattr(sel_x, "close"):
                sel_x.close()
        return res

    exchanges_s, sharded_m, sharded_l1 = None, None, None
    ranks_seen = ctl.rccl_ranks_seen()          # (every rank takes part)
    want = set(x for x in args.sharded_configs.split(",") if x)
    if sharded and args.config == "S":
        if "S" in want:
            exchanges_s = both_exchanges(eng, quiz, "s", 200, max(500, min(args.steps, 2000)),
                                         None if resident else (args.exchange, {"selections_per_sec": value, "us_per_step": 1e6 / value, "selected_question": int(sel)}))
            exchanges_s["note"] = "one launch per selection on both (the timed `value` uses %s%s)" % (
                args.exchange, " through the resident sweep" if resident else "")
        if "M" in want:
            cm = CONFIGS["M"]
            eng_m, quiz_m, q_local_m = make_engine(cm)
            ex_m = both_exchanges(eng_m, quiz_m, "m", 20, 200)
            k_ms = kernel_ms_of(eng_m, quiz_m, 20)
            bytes_m = q_local_m * (cm["K"] + 1) * cm["T"] * 8
            best = ex_m.get("shm") if "selections_per_sec" in ex_m.get("shm", {}) else ex_m.get("rccl", {})
            sharded_m = {"workload": cm["name"] + " fp64, question axis sharded over the ranks", "exchange": ex_m,
                         "selections_per_sec": best.get("selections_per_sec"), "ms_per_step": 1e-3 * best.get("us_per_step", 0.0),
                         "questions_per_gpu": q_local_m, "selected_question": best.get("selected_question"),
                         "rank0_kernel_us": 1e3 * k_ms, "rank0_kernel_GBps": bytes_m / (k_ms * 1e-3) / 1e9,
                         "eval_kernel": eng_m.eval_kernel_name()}
            eng_m.close()
        if "L1" in want:
            sharded_l1 = run_batched(args, CONFIGS[args.l1_config], np, torch, interop, pdist, ctl, rank, dev_index, device, compact=True)

    # ---- extra (not `value`), sharded runs only: the SAME sharding inside ONE process behind the plain C ABI
    # (probqa_amd/csrc/sharded_engine.cpp: PQA_DEVICES lists the devices, PqaEngineFactory_CreateCpuEngine builds one shard per
    # device and PqaEngine_NextQuestion fans out over them -- what the reference's unchanged wrappers get).  Rank 0 drives all
    # `world` GPUs while the other ranks wait at a barrier.
    one_process = None
    if sharded and world > 1:
        if rank == 0:
            one_process = {}
            os.environ["PQA_DEVICES"] = ",".join(str(d % n_dev) for d in range(world))
            try:
                for key, c, n_steps in (("1000x5x1000", CONFIGS["S"], 2000), ("10000x5x10000", CONFIGS["M"], 200)):
                    if ("S" if c is CONFIGS["S"] else "M") not in want:
                        continue
                    e1, err = factory.create_cpu_engine(interop.EngineDefinition(c["K"], c["Q"], c["T"], init_amount=0.1))
                    if err is not None:
                        one_process[key] = {"error": err.to_string(True)}
                        continue
                    e1.set_option("select", 1)
                    e1.fill_synthetic(8.0, 0.5, SEED)
                    qz1 = e1.start_quiz()
                    for _ in range(50):
                        p1 = e1.next_question(qz1)
                    t1 = time.perf_counter()
                    for _ in range(n_steps):
                        p1 = e1.next_question(qz1)
                    d1 = time.perf_counter() - t1
                    one_process[key] = {"selections_per_sec": n_steps / d1, "us_per_step": 1e6 * d1 / n_steps, "shards": e1.get_option("shards"),
                                        "selected_question": int(p1)}
                    if c is CONFIGS["S"] and not args.no_quiz_loop:
                        # the learner loop from many client threads on the ONE sharded engine (sampled selector, training at the end
                        # of every quiz): concurrent NextQuestion calls share one batched sweep per shard, all shards in flight
                        e1.set_option("select", 0)
                        e1.set_option("seed", SEED)
                        e1.release_quiz(qz1)
                        qlt = {}
                        for nt in (1, 16, 64):
                            b0 = (e1.get_option("combined_batches"), e1.get_option("combined_requests"))
                            r = interop.run_learners(e1, nt, 300 if nt == 1 else 1500, 30, seed=SEED + nt, train=True)
                            b1 = (e1.get_option("combined_batches"), e1.get_option("combined_requests"))
                            qlt[str(nt)] = {"questions_per_sec": r["questions"] / r["seconds"], "quizzes": r["quizzes"], "errors": r["errors"],
                                            "guessed_on_top": r["guessed_on_top"],
                                            "next_questions_per_combined_sweep": (b1[1] - b0[1]) / max(1, b1[0] - b0[0])}
                        for v in qlt.values():
                            v["vs_one_thread"] = v["questions_per_sec"] / qlt["1"]["questions_per_sec"]
                        qlt["shards_in_flight_max"] = e1.get_option("shards_in_flight_max")
                        qlt["peer_access"] = e1.get_option("peer_access")
                        one_process[key]["quiz_loop_threads"] = qlt
                    e1.close()
                if "L1" in want:
                    # BASELINE configs[4] as ONE engine: world x 12500 questions x 100000 targets fp32, 256 quizzes per batched call;
                    # every shard's sweep is enqueued before the first is waited for (shards_in_flight_max == shards)
                    c = CONFIGS[args.l1_config]
                    kw = dict(prec_type=interop.PrecisionType.FLOAT, prec_exponent=8, prec_mantissa=24) if c["prec"] == "f32" else {}
                    e1, err = factory.create_cpu_engine(interop.EngineDefinition(c["K"], c["Q"] * world, c["T"], init_amount=0.1, **kw))
                    if err is not None:
                        one_process[c["name"] + "_per_shard"] = {"error": err.to_string(True)}
                    else:
                        e1.set_option("select", 1)
                        e1.set_option("batch_min", 1)
                        e1.fill_synthetic(8.0, 0.5, SEED)
                        qzs = [e1.start_quiz() for _ in range(c["quizzes"])]
                        for i, qz1 in enumerate(qzs):
                            e1.set_active_question(qz1, (37 * i) % (c["Q"] * world))
                            e1.record_answer(qz1, i % c["K"])
                        p1 = e1.next_question_argmax_batch(qzs)
                        n_b = 2
                        t1 = time.perf_counter()
                        for _ in range(n_b):
                            p1 = e1.next_question_argmax_batch(qzs)
                        d1 = time.perf_counter() - t1
                        one_process[c["name"] + "_per_shard"] = {
                            "selections_per_sec": n_b * len(qzs) / d1, "ms_per_batch": 1e3 * d1 / n_b, "shards": e1.get_option("shards"),
                            "shards_in_flight_max": e1.get_option("shards_in_flight_max"), "quizzes_per_batch": len(qzs),
                            "selected_question_of_quiz_0": int(p1[0])}
                        e1.close()
            except Exception as ex:  # noqa: BLE001 -- an extra must not take the line (and the other ranks, at the barrier) with it
                one_process["error"] = repr(ex)[:500]
            finally:
                os.environ.pop("PQA_DEVICES", None)
        barrier()

    # ---- extras (not `value`), one GPU, default config: the two other measured points of the path in the line the driver runs --
    # the HBM-bound point M (BASELINE configs[2]: 10000x5x10000 fp64, one quiz) and the VALU-bound point L1 (configs[4]'s
    # per-GPU shard: 12500x5x100000 fp32, 256 quizzes per batched sweep), each with a bounded parity sample against the CPU port.
    hbm_point_m, valu_point_l1, late_state = None, None, None
    if not sharded and args.config == "S" and not args.no_points:
        free_b = torch.cuda.mem_get_info(device)[0]
        if free_b > 12e9:
            hbm_point_m = point_m(args, np, torch, interop, factory, stream, kernel_ms_of)
        try:
            late_state = late_state_point(args, np, torch, interop, factory, stream, kernel_ms_with)
        except Exception as ex:  # noqa: BLE001 -- an extra must not take the line with it
            late_state = {"error": repr(ex)[:300]}
        if free_b > 80e9:
            valu_point_l1 = run_batched(args, CONFIGS["L1"], np, torch, interop, pdist, ctl, rank, dev_index, device, compact=True)

    # ---- extra (not `value`): the same learner loop from MANY CLIENT THREADS on the one engine -- what the reference's published
    # rate is (PqaClient.cpp:238-245: the sum over hardware_concurrency learner threads, every NextQuestion under a shared lock,
    # CpuEngine.cpp:357-361).  Native threads through the C ABI (probqa_amd/client/pqa_client.cpp), sampled selector, training at
    # the end of every quiz; the engine combines the concurrent NextQuestion calls into batched sweeps and the RecordAnswers
    # into batched launches (engine option "combine").
    quiz_loop_threads, learners_hung = None, False
    if selector is None and args.config == "S" and not args.no_quiz_loop:
        eng.set_option("select", 0)
        eng.set_option("seed", SEED)
        quiz_loop_threads = {}
        import threading

        for nt in (1, 16, 64, 256):
            b0 = (eng.get_option("combined_batches"), eng.get_option("combined_requests"), eng.get_option("update_flushes"), eng.get_option("updates_flushed"))
            # (a watchdog: were the client threads ever to block each other, the line -- everything else is measured by now -- is
            #  still printed, with the fact in it)
            box = {}
            th = threading.Thread(target=lambda: box.update(r=interop.run_learners(eng, nt, 400 if nt == 1 else 2400, 30, seed=SEED + nt, train=True)),
                                  daemon=True)
            th.start()
            th.join(120.0)
            if th.is_alive() or "r" not in box:
                quiz_loop_threads[str(nt)] = {"error": "the learner threads did not finish within 120 s"}
                learners_hung = True
                break
            r = box["r"]
            b1 = (eng.get_option("combined_batches"), eng.get_option("combined_requests"), eng.get_option("update_flushes"), eng.get_option("updates_flushed"))
            quiz_loop_threads[str(nt)] = {
                "questions_per_sec": r["questions"] / r["seconds"], "quizzes": r["quizzes"], "questions": r["questions"],
                "guessed_on_top": r["guessed_on_top"], "errors": r["errors"],
                "next_questions_per_combined_sweep": (b1[1] - b0[1]) / max(1, b1[0] - b0[0]),
                "record_answers_per_launch": (b1[3] - b0[3]) / max(1, b1[2] - b0[2])}
        base = quiz_loop_threads["1"].get("questions_per_sec")
        for k, v in quiz_loop_threads.items():
            if base and "questions_per_sec" in v:
                v["vs_one_thread"] = v["questions_per_sec"] / base
        quiz_loop_threads["note"] = ("native client threads on ONE engine through the plain C ABI (PqaEngine_NextQuestion / RecordAnswer / "
                                     "ListTopTargets / RecordQuizTarget); published reference figure: 301.2 questions/s over all threads")
        if not learners_hung:
            eng.set_option("select", 1)

    out = {
        "metric": "next_question_selections_per_sec",
        "value": value,
        "unit": "selections/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps,
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {
            "workload": "%s fp64 cube resident in HBM, single in-flight quiz; step = priority sweep + argmax + "
                        "question id on host, synchronous call through the C ABI" % cfg["name"],
            "selection_path": "resident sweep kernel (engine option server=1, opt-in: bench.py --server): request and answer through pinned memory"
            if resident else "one kernel launch per selection (the library's default path)",
            "questions_per_gpu": q_local,
            "parallelism": ("question-axis shards x%d, winners exchanged through %s" % (
            world, "host shared memory written by the sweep" if args.exchange == "shm" else "an RCCL all-gather"))
        if selector is not None else "single GPU",
            "eval_kernel": eng.eval_kernel_name(),
            "selected_question": int(sel),
        },
        "question_evals_per_sec": value * Q,
        "step_latency_us": latency_us,
        # what a synchronous step spends outside the sweep: the request's way to the kernel, the answer's way back, the wrapper
        "host_overhead_us": (1e6 * elapsed / args.steps - server_step_us["mean"]) if server_step_us else (1e6 * elapsed / args.steps - launched_us),
        "launch_per_selection": launch_rate if launch_rate is not None else ({"selections_per_sec": value, "note": "this is `value`"} if selector is None else None),
        "resident_sweep": resident_extra,
        "native_caller": native,
        "pipelined_selections_per_sec": pipelined,
        "batched": batched,
        "hip_graph_replay": graph_rate,
        "quiz_loop": quiz_loop,
        "quiz_loop_threads": quiz_loop_threads,
        # what a scaling run is read by: which exchange `value` used, how many ranks an RCCL collective reached, the node's peer
        # access, and the rate of BOTH exchanges where the run is sharded (one launch per selection on both)
        "multi_gpu": {
            "n_gpus": world,
            "exchange": args.exchange if selector is not None else None,
            "rccl_ranks_seen": ranks_seen,
            "peer_access_matrix": ctl.peer_access_matrix(),
            "selections_per_sec": {k: v.get("selections_per_sec") for k, v in exchanges_s.items() if isinstance(v, dict)} if exchanges_s else
            {"single_gpu_no_exchange": value},
        },
        "exchange_1000x5x1000": exchanges_s,
        "sharded_10000x5x10000": sharded_m,
        "sharded_12500x5x100000_per_gpu": sharded_l1,
        "one_process_sharded_engine": one_process,
        "hbm_point_M": hbm_point_m,
        "late_state_S": late_state,
        "valu_point_L1": valu_point_l1,
        "roofline": {
            "bound": "hbm",
            "achieved": achieved,
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS,
            "traffic": pmc_traffic(args.config, world),
            "traffic_from_other_sources_than_the_tree": pmc_is_stale(args.config),
            "kernel": ("eval_server_f64 (wg256_np2, resident: one launch serves the selections; the kernel `value` went through)"
                       if server_step_us else "eval_questions_f64 (%s)" % eng.eval_kernel_name()),
            "kernel_us": path_us,
            "kernel_us_source": ("the resident kernel's own 100 MHz clock per step, request in hand to answer published "
                                 "(mean of %d steps; p10/p50/p90 in resident_step_us)" % server_step_us["n"]) if server_step_us
            else "HIP events on the engine's stream around back-to-back launches (mean)",
            # the same kernel launched onto an IDLE device, as every synchronous call of the timed region launches it: the engine's own
            # HIP events right around the launch (option time_sweeps; the two event markers included), dispatch to retirement.  A launch
            # by itself does not overlap its ramp with the previous kernel's tail: rocprofv3's per-launch average over the profiled
            # command (profiles/r06_S_*: mostly such launches) lies between this and kernel_us
            "synchronous_launch_us": sync_kernel_us,
            "resident_step_us": server_step_us,
            "launched_kernel": {"kernel": "eval_questions_f64 (%s)" % eng.eval_kernel_name(), "kernel_us": launched_us,
                                "achieved": alg_bytes / (launched_us * 1e-6) / 1e9,
                                "frac": alg_bytes / (launched_us * 1e-6) / 1e9 / HBM_PEAK_GBS,
                                "source": "HIP events on the engine's stream around back-to-back launches (mean); the "
                                          "rocprofv3 --kernel-trace --stats summary of this kernel is under profiles/"},
            "algorithmic_bytes_per_launch": alg_bytes,
            "note": "48 MB cube fits the 256 MiB Infinity Cache: achieved GB/s is not an HBM measurement at this size; "
                    "use --config M for the HBM-bound point" if args.config == "S" else "cube exceeds the Infinity Cache",
        },
        # the other roofline ("report both"): the sweep's arithmetic against the fp64 vector peak
        "roofline_valu": {
            "bound": "valu_fp64",
            "achieved": alg_flops / (path_us * 1e-6) / 1e12,
            "peak": FP64_VECTOR_PEAK_TFLOPS,
            "unit": "TFLOP/s",
            "frac": alg_flops / (path_us * 1e-6) / 1e12 / FP64_VECTOR_PEAK_TFLOPS,
            "algorithmic_flops_per_launch": alg_flops,
            "flops_per_element": FLOPS_PER_ELEMENT,
            "pmc": pmc_valu(args.config, world),
            "note": "algorithmic flops = Q K T x 45 (the reference's operation count, SURVEY 8(d)); pmc = VALU instructions and "
                    "VALU-busy cycles per launch of the launched kernel from the committed counter pass (profiles/)",
        },
    }

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"], cpu_argmax, cpu_margin = cpu_baseline(np, cfg, args.cpu_seconds)
        # north_star: "with the argmax matching CpuEngine" -- the CPU port's priorities of the same cube and quiz state
        # (after StartQuiz), its argmax (lowest index on ties) against the question the timed steps selected
        if cpu_argmax is not None:
            out["argmax_matches_cpu"] = int(cpu_argmax) == int(sel)
            out["config"]["cpu_argmax"] = int(cpu_argmax)
            out["config"]["cpu_top2_relative_margin"] = cpu_margin
            if int(cpu_argmax) != int(sel):
                print("bench.py: the engine selected question %d, the CPU port's argmax is %d (top-2 margin %.3g)"
                      % (int(sel), int(cpu_argmax), cpu_margin), file=sys.stderr)
    if rank == 0:
        os.write(result_fd, (json.dumps(out) + "\n").encode())
    if learners_hung:
        os._exit(0)      # (the engine's client threads are stuck: its destructor would wait for them)
    if selector is not None and hasattr(selector, "close"):
        selector.close()
    eng.close()
    ctl.close()


def self_spawn(n):
    """`python bench.py --gpus N` typed by itself: run the same command line as N ranks under torch.distributed.run (one
    process per GPU, rendezvous on 127.0.0.1) -- what the driver's launch line does.  Returns the launcher's exit code."""
    import socket
    import subprocess

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


class Ctl:
    """The control plane of a run: rendezvous, barriers, max-over-ranks timing.  torch.distributed with backend nccl (= RCCL);
    gloo when several ranks share a device (dry run of the N > 1 path on a smaller box).  One rank: no process group at all,
    unless --force-collective asks for the RCCL path with a world of one."""

    def __init__(self, torch, world, device, oversub, force):
        self.torch, self.world, self.device = torch, world, device
        self.dist = None
        self.rccl = False
        if world > 1 or force:
            import torch.distributed as dist

            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29533")
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
            if oversub:
                dist.init_process_group(backend="gloo")
            else:
                dist.init_process_group(backend="nccl", device_id=device)
                self.rccl = True
            self.dist = dist
        self.tdev = device if self.rccl else torch.device("cpu")

    def barrier(self):
        self.torch.cuda.synchronize()
        if self.world > 1:
            self.dist.barrier()
            self.torch.cuda.synchronize()

    def max_float(self, x):
        if self.world == 1:
            return x
        t = self.torch.tensor([x], dtype=self.torch.float64, device=self.tdev)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def min_int(self, x):
        if self.world == 1:
            return int(x)
        t = self.torch.tensor([int(x)], dtype=self.torch.int32, device=self.tdev)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MIN)
        return int(t.item())

    def broadcast_prior(self, pdist, ptr, ld, owner):
        """A quiz's new posterior from the rank that computed it to every rank's engine (8 ldT bytes)."""
        if self.world == 1:
            return
        t = pdist.tensor_from_device_ptr(ptr, ld, self.device)
        if self.rccl:
            self.dist.broadcast(t, src=owner)
        else:                       # gloo dry run: through the host
            h = t.cpu()
            self.dist.broadcast(h, src=owner)
            t.copy_(h)

    def rccl_ranks_seen(self):
        """How many ranks one RCCL all-gather reaches (every rank contributes its rank number): the world size if the
        collective spans the job, None without RCCL (one rank and no --force-collective, or the gloo dry run)."""
        if not self.rccl:
            return None
        mine = self.torch.tensor([int(os.environ.get("RANK", "0"))], dtype=self.torch.int64, device=self.device)
        parts = [self.torch.empty_like(mine) for _ in range(max(1, self.world))]
        self.dist.all_gather(parts, mine)
        self.torch.cuda.synchronize()
        return len(set(int(p.item()) for p in parts))

    def peer_access_matrix(self):
        """[i][j] = 1 if device i of this node maps device j's memory (hipDeviceCanAccessPeer): what the shards' reads of each
        other's rows go over (xGMI where 1, a staged copy where 0)."""
        n = self.torch.cuda.device_count()
        return [[1 if i == j or self.torch.cuda.can_device_access_peer(i, j) else 0 for j in range(n)] for i in range(n)]

    def close(self):
        if self.dist is not None:
            self.dist.destroy_process_group()
            self.dist = None


def late_state_point(args, np, torch, interop, factory, stream, kernel_ms_with):
    """The default cube in a LATE quiz state -- consistent answers until the posterior sits on one target, where every question has
    an answer row at the pole of the lack term: the sweep with the reference-order fix-up behind it (pole_kernels.hip) against the
    sweep alone, by HIP events; the synchronous selection a caller sees; parity of that state's priorities against the CPU port."""
    c = CONFIGS["S"]
    Q, K, T = c["Q"], c["K"], c["T"]
    e = factory.create_hip_engine(interop.EngineDefinition(K, Q, T, init_amount=0.1), 0, Q, torch.cuda.current_device())
    e.set_option("select", 1)
    e.fill_synthetic(8.0, 0.5, SEED)
    e.set_stream(stream.cuda_stream)
    qz = e.start_quiz()
    guess, width, top, answers, hist = int(0.37 * T), max(1, 32 * T // 1000), None, 0, []
    for _ in range(40):
        qq = e.next_question_argmax(qz)
        x = qq * T // Q
        ans = 0 if guess < x - width else 1 if guess < x else 2 if guess == x else 3 if guess <= x + width else 4
        e.record_answer(qz, ans)
        hist.append((int(qq), ans))
        answers += 1
        top = e.list_top_targets(qz, 1)
        if top and top[0].prob > 1 - 1e-6:
            break
    with_fix = kernel_ms_with(e, qz, 100)
    for _ in range(5):
        pick = e.next_question_argmax(qz)
    n = 300
    t0 = time.perf_counter()
    for _ in range(n):
        pick = e.next_question_argmax(qz)
    sync_us = 1e6 * (time.perf_counter() - t0) / n
    out = {"workload": c["name"] + " fp64, one quiz after %d consistent answers (top posterior 1 - %.1e): every question listed" % (answers, 1 - top[0].prob if top else float("nan")),
           "sweep_plus_fixup_us": 1e3 * with_fix, "synchronous_selection_us": sync_us, "selected_question": int(pick)}
    if not args.no_cpu_baseline:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import orclib
        from probqa_amd import synth

        pri = e.eval_priorities(qz, Q)
        orc = orclib.Oracle(K, Q, T, 0.1)
        orc.set_kb(*synth.synthetic_kb(K, Q, T, 0.1, 8.0, 0.5, SEED))
        orc.start_quiz(16)
        for qa in hist:
            orc.record_answer(qa[0], qa[1], 15)
        _, opri = orc.eval_avx2(min(os.cpu_count() or 1, 16))
        live = opri > 0
        rel = np.abs(pri[live] - opri[live]) / opri[live]
        out["parity"] = {"questions": int(live.sum()), "max_rel_err": float(rel.max()),
                         "posterior_bit_identical": bool(np.array_equal(e.get_priors(qz), orc.priors())),
                         "argmax_matches_cpu": int(pick) == int(orc.select_argmax(opri)),
                         "note": "all priorities of this state against the fp64 CPU port replaying the same answers"}
    e.set_option("pole_fix", 0)
    out["sweep_alone_us"] = 1e3 * kernel_ms_with(e, qz, 100)
    if "parity" in out:
        pri0 = e.eval_priorities(qz, Q)
        out["parity"]["max_rel_err_without_the_fixup"] = float((np.abs(pri0[live] - opri[live]) / opri[live]).max())
    e.close()
    return out


def point_m(args, np, torch, interop, factory, stream, kernel_ms_of):
    """BASELINE configs[2] inside the default run: 10000x5x10000 fp64, one quiz -- 30 synchronous selections, the sweep kernel
    by HIP events, and a parity sample (the first 200 questions' priorities against the CPU port on the generator's rows)."""
    c = CONFIGS["M"]
    Q, K, T = c["Q"], c["K"], c["T"]
    e = factory.create_hip_engine(interop.EngineDefinition(K, Q, T, init_amount=0.1), 0, Q, torch.cuda.current_device())
    e.set_option("select", 1)
    e.fill_synthetic(8.0, 0.5, SEED)
    e.set_stream(stream.cuda_stream)
    qz = e.start_quiz()
    for _ in range(5):
        pick = e.next_question_argmax(qz)
    torch.cuda.synchronize()
    n = 30
    t0 = time.perf_counter()
    for _ in range(n):
        pick = e.next_question_argmax(qz)
    dt = time.perf_counter() - t0
    k_ms = kernel_ms_of(e, qz, 20)
    alg = Q * (K + 1) * T * 8
    out = {"workload": c["name"] + " fp64, single in-flight quiz (BASELINE configs[2])", "selections_per_sec": n / dt, "ms_per_step": 1e3 * dt / n,
           "steps": n, "kernel": "eval_questions_f64 (%s)" % e.eval_kernel_name(), "kernel_us": 1e3 * k_ms,
           "kernel_us_source": "HIP events on the engine's stream around 20 back-to-back launches (mean)",
           "algorithmic_bytes_per_launch": alg, "achieved": alg / (k_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
           "frac": alg / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "selected_question": int(pick)}
    if not args.no_cpu_baseline:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import orclib
        from probqa_amd import synth

        n_q = 200
        pri = e.eval_priorities(qz, Q)
        orc = orclib.Oracle(K, n_q, T, 0.1)
        orc.set_kb(*synth.synthetic_kb(K, n_q, T, 0.1, 8.0, 0.5, SEED, q_offset=0, q_total=Q))
        orc.mants[:T] = e.get_priors(qz)
        _, opri = orc.eval_avx2(min(os.cpu_count() or 1, 16))
        rel = np.abs(pri[:n_q] - opri) / np.maximum(np.abs(opri), 1e-300)
        out["sample_parity"] = {"questions": n_q, "max_rel_err": float(rel.max()),
                                "argmax_of_sample_matches_cpu": int(np.argmax(pri[:n_q])) == int(orc.select_argmax(opri)),
                                "note": "GPU priorities of the first %d questions against the fp64 CPU port on the generator's rows" % n_q}
    # ... and the same cube in a LATE quiz state (consistent answers until the posterior sits on one target): sweep + fix-up
    try:
        guess, width, top, hist = int(0.37 * T), max(1, 32 * T // 1000), None, 0
        for _ in range(40):
            qq = e.next_question_argmax(qz)
            x = qq * T // Q
            e.record_answer(qz, 0 if guess < x - width else 1 if guess < x else 2 if guess == x else 3 if guess <= x + width else 4)
            hist += 1
            top = e.list_top_targets(qz, 1)
            if top and top[0].prob > 1 - 1e-6:
                break
        e.set_option("pole_follow", 1)
        for _ in range(2):
            e.enqueue_eval(qz)
        torch.cuda.synchronize()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record(stream)
        for _ in range(10):
            e.enqueue_eval(qz)
        ev1.record(stream)
        torch.cuda.synchronize()
        late = {"answers": hist, "top_posterior_one_minus": (1 - top[0].prob) if top else None, "sweep_plus_fixup_us": 1e3 * ev0.elapsed_time(ev1) / 10,
                "sweep_plus_fixup_note": "the priority VECTOR of the late state (what PqaEngine_EvalPriorities and the sampled selector get): every listed question redone"}
        # ... and what a NextQuestion with the argmax selector costs there: only the selected question leaves the engine, so only the listed
        # questions that can still win are redone (engine option pole_gate; pole_kernels.hip: pole_bounds_kernel)
        e.set_option("speculate", 0)
        sel_us = {}
        for gate in (0, 1):
            e.set_option("pole_gate", gate)
            for _ in range(3):
                pick_l = e.next_question_argmax(qz)
            t1 = time.perf_counter()
            for _ in range(10):
                pick_l = e.next_question_argmax(qz)
            sel_us[gate] = (1e5 * (time.perf_counter() - t1), int(pick_l))
        e.set_option("speculate", 1)
        late["synchronous_argmax_selection_us"] = sel_us[1][0]
        late["synchronous_argmax_selection_us_with_every_listed_question_redone"] = sel_us[0][0]
        late["selected_question"] = sel_us[1][1]
        late["gated_and_full_fix_select_the_same_question"] = sel_us[0][1] == sel_us[1][1]
        if not args.no_cpu_baseline:
            n_q = 200
            pri = e.eval_priorities(qz, Q)
            orc = orclib.Oracle(K, n_q, T, 0.1)
            orc.set_kb(*synth.synthetic_kb(K, n_q, T, 0.1, 8.0, 0.5, SEED, q_offset=0, q_total=Q))
            orc.mants[:T] = e.get_priors(qz)
            _, opri = orc.eval_avx2(min(os.cpu_count() or 1, 16))
            live = pri[:n_q] > 0                                   # (asked questions among the sample: priority 0 on the device)
            late["sample_parity"] = {"questions": int(live.sum()),
                                     "max_rel_err": float((np.abs(pri[:n_q][live] - opri[live]) / opri[live]).max())}
            e.set_option("pole_fix", 0)
            pri0 = e.eval_priorities(qz, Q)
            late["sample_parity"]["max_rel_err_without_the_fixup"] = float((np.abs(pri0[:n_q][live] - opri[live]) / opri[live]).max())
        out["late_state"] = late
    except Exception as ex:  # noqa: BLE001 -- an extra of an extra
        out["late_state"] = {"error": repr(ex)[:300]}
    e.close()
    return out


def run_batched(args, cfg, np, torch, interop, pdist, ctl, rank, dev_index, device, compact=False):
    """BASELINE configs[4]: B quizzes in flight, one batched NextQuestion per step -- the row-sharing sweep
    (probqa_amd/csrc/batch_kernels.hip: a lane is a quiz, the cube is read once per batch).  A step = the argmax selections
    of all B quizzes: transposed priors + sweep + per-quiz pick on the device, question ids on the host.  N > 1: every rank
    holds its own Q-question shard (weak scaling: the cube grows with N); the ranks' per-quiz winners (16 B x B) meet in host
    shared memory (they are on the host when the batched call returns) or in one RCCL all-gather -- both are timed -- and every
    rank makes the same picks.  compact: the short form that rides in the default run's line (2 steps, 1 warm-up)."""
    world = ctl.world
    Q, K, T, B = cfg["Q"], cfg["K"], cfg["T"], cfg["quizzes"]
    f32 = cfg["prec"] == "f32"
    if compact:
        steps, warmup = (2, 1) if Q * T > 1e8 else (10, 2)
    else:
        steps = args.steps if args.steps != 2000 else (3 if Q * T > 1e8 else 20)       # the defaults are sized for the S config
        warmup = args.warmup if args.warmup != 5000 else 1
    factory = interop.PqaEngineFactory()
    stream = torch.cuda.Stream(device=device)
    torch.cuda.set_stream(stream)
    q_total = Q * world
    kw = dict(prec_type=interop.PrecisionType.FLOAT, prec_exponent=8, prec_mantissa=24) if f32 else {}
    eng = factory.create_hip_engine(interop.EngineDefinition(K, Q, T, init_amount=0.1, **kw), rank * Q, q_total, dev_index)
    eng.set_option("select", 1)
    eng.set_option("batch_min", 1)
    eng.fill_synthetic(8.0, 0.5, SEED)
    eng.set_stream(stream.cuda_stream)
    # B quizzes in different states: quiz i has answered question (37 i) mod Qtotal with answer i mod K (the owner rank records
    # it, the new posterior is copied to the other ranks' engines)
    quizzes = [eng.start_quiz() for _ in range(B)]
    for i, qz in enumerate(quizzes):
        qa = (37 * i) % q_total
        owner = qa // Q
        eng.set_active_question(qz, qa)
        if owner == rank:
            eng.record_answer(qz, i % K)
        else:
            eng.record_answer_remote(qz, i % K)
        if world > 1:
            ptr, ld = eng.prior_device_ptr(qz)
            eng.synchronize()
            ctl.broadcast_prior(pdist, ptr, ld, owner)

    shm = None
    if world > 1:
        name = "bench_%s_b%d" % (os.environ.get("MASTER_PORT", "0"), Q)
        if rank == 0:
            shm = pdist.ShmBatchExchange(B, rank, world, name, create=True)
        ctl.barrier()
        if rank != 0:
            shm = pdist.ShmBatchExchange(B, rank, world, name, create=False)

    def step_shm():
        if world == 1:
            return eng.next_question_argmax_batch(quizzes)
        return shm.exchange(eng.select_argmax_batch(quizzes))

    def step_rccl():
        mine = torch.from_numpy(eng.select_argmax_batch(quizzes)).to(device)          # [B, 2] (priority, global index)
        return pdist.select_batch(mine)

    barrier = ctl.barrier

    def timed_steps(step):
        picks = None
        for _ in range(warmup):
            picks = step()
        eng.synchronize()
        barrier()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record(stream)
        t0 = time.perf_counter()
        for _ in range(steps):
            picks = step()
        ev1.record(stream)
        eng.synchronize()
        barrier()
        dt = ctl.max_float(time.perf_counter() - t0)
        return dt, ev0.elapsed_time(ev1) / steps, picks

    dt, kernel_ms, picks = timed_steps(step_shm)
    exchange = None
    if world > 1:
        exchange = {"shm": {"selections_per_sec": B * steps / dt, "ms_per_step": 1e3 * dt / steps, "selected_question_of_quiz_0": int(picks[0])}}
        if ctl.rccl:
            dt_r, _, picks_r = timed_steps(step_rccl)
            exchange["rccl"] = {"selections_per_sec": B * steps / dt_r, "ms_per_step": 1e3 * dt_r / steps, "selected_question_of_quiz_0": int(picks_r[0]),
                                "same_picks_as_shm": [int(x) for x in picks_r] == [int(x) for x in picks]}
        else:
            exchange["rccl"] = {"skipped": "the ranks share a device: RCCL refuses that"}
    value = B * steps / dt
    s = 4 if f32 else 8
    elements = Q * K * T * B
    alg_flops = elements * FLOPS_PER_ELEMENT
    alg_bytes = Q * (K + 1) * T * s                # the cube once per batch ("read once per batch of 256 quizzes")
    peak = FP32_VECTOR_PEAK_TFLOPS if f32 else FP64_VECTOR_PEAK_TFLOPS
    tf = alg_flops / (kernel_ms * 1e-3) / 1e12
    if compact:
        out = {"workload": "%s %s per GPU (%.1f GB), %d quizzes per batched sweep (BASELINE configs[4]'s shard)" % (cfg["name"], cfg["prec"], alg_bytes / 1e9, B),
               "selections_per_sec": value, "ms_per_step": 1e3 * dt / steps, "steps": steps, "n_gpus": world, "questions_total": q_total,
               "kernel": "eval_batch_kernel<%s>" % ("float" if f32 else "double"), "kernel_us": kernel_ms * 1e3,
               "kernel_us_source": "HIP events on the engine's stream around the timed steps (sweep + prep + pick kernels)",
               "bound": "valu_fp32" if f32 else "valu_fp64", "achieved": tf, "peak": peak, "unit": "TFLOP/s", "frac": tf / peak,
               "algorithmic_flops_per_launch": alg_flops, "exchange": exchange, "selected_question_of_quiz_0": int(picks[0])}
        if rank == 0 and world == 1 and not args.no_cpu_baseline:
            _, out["sample_parity"] = cpu_baseline_batched(np, cfg, eng, quizzes, min(args.cpu_seconds, 1.5), f32)
        if shm is not None:
            shm.close()
        eng.close()
        return out
    out = {
        "metric": "next_question_selections_per_sec",
        "value": value,
        "unit": "selections/s",
        "n_gpus": world,
        "steps": steps,
        "warmup": warmup,
        "ms_per_step": 1e3 * dt / steps,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32" if f32 else "f64",
        "data": "synthetic",
        "config": {
            "workload": "%s %s cube per GPU resident in HBM (%.1f GB), %d concurrent quizzes batched along a leading dimension; "
                        "step = one batched NextQuestion: argmax selection for every quiz, question ids on the host"
                        % (cfg["name"], cfg["prec"], alg_bytes / 1e9, B),
            "quizzes_per_batch": B,
            "questions_per_gpu": Q,
            "questions_total": q_total,
            "parallelism": "single GPU" if world == 1 else "question-axis shards x%d (one %d-question shard per GPU), the "
                           "ranks' per-quiz winners (%d B per rank) exchanged through host shared memory (`exchange` has the RCCL "
                           "all-gather beside it)" % (world, Q, 16 * B),
            "eval_kernel": "eval_batch_kernel<%s> (lane = quiz, LDS tile shared by the batch)" % ("float" if f32 else "double"),
            "selected_question_of_quiz_0": int(picks[0]),
        },
        "question_evals_per_sec": value * q_total,
        "exchange": exchange,
        "element_evals_per_sec_per_gpu": elements / (dt / steps),
        # this configuration is bound by the vector ALU (transcendental + fp32 arithmetic per element), not by HBM
        "roofline": {
            "bound": "valu_fp32" if f32 else "valu_fp64",
            "achieved": tf,
            "peak": peak,
            "unit": "TFLOP/s",
            "frac": tf / peak,
            "traffic": pmc_traffic(cfg.get("traffic_key", args.config), world),
            "kernel": "eval_batch_kernel",
            "kernel_us": kernel_ms * 1e3,
            "kernel_us_source": "HIP events on the engine's stream around the timed steps (sweep + prep + pick kernels; the sweep "
                                "is > 99.9 % of it at this size)",
            "algorithmic_flops_per_launch": alg_flops,
            "flops_per_element": FLOPS_PER_ELEMENT,
            "pmc": pmc_valu(cfg.get("traffic_key", args.config), world),
            "note": "flops = B Q K T x 45 (the reference's operation count per element, SURVEY 8(d)); peak = the %s vector peak -- the "
                    "contract's fraction.  The physical one is pmc.valu_pipe_busy: the kernel issues ~7.7 VALU instructions per "
                    "element (packed ones carry two elements), two of them quarter-rate transcendentals" % ("fp32" if f32 else "fp64"),
        },
        "roofline_hbm": {
            "bound": "hbm",
            "achieved": alg_bytes / (kernel_ms * 1e-3) / 1e9,
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": alg_bytes / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "algorithmic_bytes_per_launch": alg_bytes,
            "note": "algorithmic bytes = the cube once per batch; the batch's work is B x that many element evaluations, so the HBM "
                    "side idles (SURVEY 8(d)) -- `traffic` above is the measured fetch per batch, priors included",
        },
    }
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"], out["sample_parity"] = cpu_baseline_batched(np, cfg, eng, quizzes, args.cpu_seconds, f32)
        out["argmax_vs_cpu"] = argmax_share(np, cfg, eng, quizzes, picks, f32)
    if shm is not None:
        shm.close()
    eng.close()
    return out


def argmax_share(np, cfg, eng, quizzes, picks, f32):
    """north_star: "with the argmax matching CpuEngine".  Cubes the CPU port can sweep whole (<= 2e7 elements): the share of the
    batch's quizzes whose pick is the argmax of the fp64 CPU port's priorities on the engine's own (rounded) cube -- with the
    Float engine's fp64 re-rank (option rerank, default) and by the fp32 sweep alone.  Larger cubes: the picks are checked
    against the CPU port on the CANDIDATE questions only (the 8 best of the fp32 sweep), which is what the re-rank decides among."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import orclib
    from probqa_amd import synth

    Q, K, T = cfg["Q"], cfg["K"], cfg["T"]
    if Q * K * T > 2e7:
        return None
    A, D, Bv = synth.synthetic_kb(K, Q, T, 0.1, 8.0, 0.5, SEED)
    if f32:
        A, D, Bv = (x.astype(np.float32).astype(np.float64) for x in (A, D, Bv))
    orc = orclib.Oracle(K, Q, T, 0.1)
    orc.set_kb(A, D, Bv)
    picks32 = None
    if f32:
        eng.set_option("rerank", 0)
        picks32 = eng.next_question_argmax_batch(quizzes)
        eng.set_option("rerank", 1)
        picks = eng.next_question_argmax_batch(quizzes)
    same = same32 = decided = 0
    n = min(len(quizzes), 64)
    for i in range(n):
        orc.mants[:T] = eng.get_priors(quizzes[i])
        asked = [(37 * i) % Q]
        _, opri = orc.eval_avx2(min(os.cpu_count() or 1, 16))
        opri[asked] = 0.0
        want = int(np.argmax(opri))
        top = np.sort(opri)[::-1]
        decided += int((top[0] - top[1]) / top[0] > 1e-9)
        same += int(int(picks[i]) == want)
        if picks32 is not None:
            same32 += int(int(picks32[i]) == want)
    return {"quizzes_checked": n, "picks_equal_to_cpu_argmax": same, "decided_beyond_1e-9": decided,
            "picks_equal_by_fp32_sweep_alone": same32 if picks32 is not None else None,
            "note": "fp64 CPU port on the engine's own cube, whole sweep per quiz; the quiz's answered question excluded"}


def cpu_baseline_batched(np, cfg, eng, quizzes, seconds, f32):
    """The CPU port (fp64 -- the reference's CPU engine has no other precision) on a bounded sample of the same workload: the
    first `n_q` questions of the cube, quiz 0's state; its rate is scaled to whole sweeps of the per-GPU cube.  The same sample
    checks the GPU batch's priorities for quizzes 0 and 1."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import orclib
    from probqa_amd import synth

    Q, K, T = cfg["Q"], cfg["K"], cfg["T"]
    n_q = max(8, min(Q, int(2e7 // (K * T))))                   # ~2e7 elements: tens of milliseconds per sweep per core
    hw = os.cpu_count() or 1
    A, D, Bv = synth.synthetic_kb(K, n_q, T, 0.1, 8.0, 0.5, SEED, q_offset=0, q_total=Q)
    if f32:
        A, D, Bv = (x.astype(np.float32).astype(np.float64) for x in (A, D, Bv))
    orc = orclib.Oracle(K, n_q, T, 0.1)
    orc.set_kb(A, D, Bv)
    pri_gpu = eng.eval_priorities_batch(quizzes[:64], Q)        # one more batched sweep, priorities kept
    parity = {}
    for i in (0, 1):
        orc.mants[:T] = eng.get_priors(quizzes[i])
        _, opri = orc.eval_avx2(min(hw, 32))
        got = pri_gpu[i][:n_q]
        nz = (opri != 0) & (got != 0)      # (the quiz's own asked question has priority 0 on the GPU; the sample oracle has no asked bits)
        parity["quiz_%d_max_rel_err" % i] = float(np.max(np.abs(got[nz] - opri[nz]) / opri[nz])) if nz.any() else 0.0
    parity["questions"] = n_q
    parity["note"] = "GPU batched sweep vs the fp64 CPU port on the first %d questions of the (rounded) cube" % n_q
    tried = []
    for threads in sorted({min(hw, 64), min(hw, 32), min(hw, 16)}):
        orc.eval_avx2(threads)
        n, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < seconds / 3 or n < 3:
            orc.eval_avx2(threads)
            n += 1
        dt = time.perf_counter() - t0
        tried.append((n / dt * (n_q / Q), threads, n, dt))
    sweeps, threads, n, dt = max(tried)
    return ({"value": sweeps, "unit": "selections/s", "cores": threads, "kind": "port",
             "sample": "first %d of %d questions of the %s cube, fp64 AVX2+FMA 4-lane Kahan port, one quiz at a time (the reference "
                       "serves concurrent quizzes one sweep each): %d sample sweeps in %.1f s on %d threads, scaled by %d/%d to "
                       "whole-cube selections/s; pool sizes tried: %s"
                       % (n_q, Q, cfg["name"], n, dt, threads, n_q, Q, ", ".join("%d thr: %.3g/s" % (t, v) for v, t, _, _ in tried))},
            parity)


def pmc_traffic(config, world):
    """HBM-side read bytes per launch of the sweep kernel from the rocprofv3 PMC pass of this workload (FETCH_SIZE in its
    own --pmc run, KB -> bytes, doubled as MI355X_MICROARCH.md prescribes for 16 B/lane streaming reads on gfx950).
    bench.py cannot collect counters on itself, so the number is the one tools/prof.sh wrote into profiles/traffic.json
    for the committed profile of the same command; null when there is none (or for sharded runs)."""
    if world != 1:
        return None
    try:
        with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
            return json.load(f).get(config, {}).get("bytes_per_launch")
    except (OSError, ValueError):
        return None


KERNEL_HEADERS = ("pqa_device.h", "eval_device.h", "prior_device.h", "pole_device.h", "pqa_kernels.h")   # (what the kernels include; hip_engine.h is the host's)


def kernel_sources_sha16():
    """sha256 (first 16 hex digits) of the kernel sources, as tools/prof.sh records it beside the counters it collects."""
    import hashlib

    d = os.path.join(ROOT, "probqa_amd", "csrc")
    return hashlib.sha256(b"".join(open(os.path.join(d, f), "rb").read() for f in sorted(os.listdir(d))
                                   if f.endswith(".hip") or f in KERNEL_HEADERS)).hexdigest()[:16]


def pmc_is_stale(config):
    """True if the committed counter pass (profiles/traffic.json) was taken from other kernel sources than the ones in the tree:
    `roofline.traffic` / `roofline_valu.pmc` are that pass's numbers, not this run's (bench.py cannot collect counters on itself)."""
    try:
        with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
            rec = json.load(f).get(config, {})
        return rec.get("kernel_sources_sha16") != kernel_sources_sha16()
    except (OSError, ValueError):
        return None


def pmc_valu(config, world):
    """VALU counters per launch of the sweep from the committed PMC pass (profiles/traffic.json, written by tools/prof.sh)."""
    if world != 1:
        return None
    try:
        with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
            v = json.load(f).get(config, {}).get("valu")
    except (OSError, ValueError):
        return None
    if v and v.get("SQ_WAVE_CYCLES") and v.get("SQ_WAVES"):
        # the physical figure beside the contract's flops fraction: the share of the launch in which a SIMD's VALU has an
        # instruction executing = (VALU-active share of a wave's resident cycles) x (waves per SIMD; the sweeps are persistent
        # grids, every wave resident for the whole launch; MI355X: 256 CUs x 4 SIMDs)
        v = dict(v, valu_pipe_busy=min(1.0, v["SQ_ACTIVE_INST_VALU"] / v["SQ_WAVE_CYCLES"] * v["SQ_WAVES"] / 1024.0))
    return v


def cpu_baseline(np, cfg, seconds):
    """The reference's AVX2 SRThreadPool path as restated in oracle/pqa_oracle_avx2.c ("port"), timed on this host's
    cores on the same workload: whole sweeps of the same synthetic cube for a bounded wall time."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import orclib
    from probqa_amd import synth

    Q, K, T = cfg["Q"], cfg["K"], cfg["T"]
    sample = "full %s sweep" % cfg["name"]
    if Q * K * T > 2e8:  # bound memory and time: a 1000-question slice of the big cube (same rows, same T)
        Q = 1000
        sample = "first 1000 of %d questions of the %s cube (rate scaled to the full cube)" % (cfg["Q"], cfg["name"])
    hw = os.cpu_count() or 1
    orc = orclib.Oracle(K, Q, T, 0.1)
    orc.set_kb(*synth.synthetic_kb(K, Q, T, 0.1, 8.0, 0.5, SEED, q_offset=0, q_total=cfg["Q"]))
    orc.start_quiz(16)
    # The reference runs hardware_concurrency threads and 8x as many subtasks (PqaCore/CpuEngine.cpp:339); on a
    # many-core host that split leaves most subtasks of a 1000-question sweep empty, so smaller pools are timed too and
    # the fastest is the baseline (its thread count is what "cores" reports).
    tried = []
    for threads in sorted({hw, max(1, hw // 2), min(hw, 64), min(hw, 32), min(hw, 16)}):  # the pool only grows
        orc.eval_avx2(threads)  # warm-up, spawns the missing workers
        n, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < seconds / 4 or n < 3:
            orc.eval_avx2(threads)
            n += 1
        dt = time.perf_counter() - t0
        tried.append((n / dt * (Q / cfg["Q"]), threads, n, dt))
    sweeps, threads, n, dt = max(tried)
    others = ", ".join("%d thr: %.1f/s" % (t, v) for v, t, _, _ in tried)
    model = ""
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    cpu_argmax, cpu_margin = None, None
    if Q == cfg["Q"]:   # the whole cube was swept: its argmax is what the engine's selection is held to
        _, pri = orc.eval_avx2(threads)
        cpu_argmax = orc.select_argmax(pri)
        top = np.sort(pri)[::-1]
        cpu_margin = float((top[0] - top[1]) / top[0]) if len(top) > 1 and top[0] > 0 else 1.0
    return {"value": sweeps, "unit": "selections/s", "cores": threads, "kind": "port",
            "sample": "%s, %d sweeps in %.1f s wall on %d threads (AVX2+FMA 4-lane Kahan port, 8*threads subtasks); "
                      "pool sizes tried: %s" % (sample, n, dt, threads, others),
            "cpu_model": model}, cpu_argmax, cpu_margin


if __name__ == "__main__":
    main()
