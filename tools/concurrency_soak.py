"""Soak of the concurrent-client paths (combined sweeps, posted operations, group commit): learner threads of random counts, both
selectors, with and without training, against the one-thread digest where the transcript is interleaving-independent.
usage: python tools/concurrency_soak.py [rounds]      (every round is bounded by the client's own watchdog)"""
import sys, os, time, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from probqa_amd import interop

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rnd = random.Random(20260929)
f = interop.PqaEngineFactory()
t0 = time.time()
for r in range(rounds):
    K, Q, T = rnd.choice([(5, 300, 1000), (5, 80, 300), (5, 1000, 1000), (5, 50, 2000), (6, 120, 700), (8, 100, 500)])   # (the client answers 0..4)
    e, err = f.create_cpu_engine(interop.EngineDefinition(K, Q, T, init_amount=0.1))
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
    ok = one["errors"] == 0 and many["errors"] == 0 and (many["questions"], many["transcript_hash"]) == (one["questions"], one["transcript_hash"])
    e.set_option("select", rnd.choice([0, 1]))
    tr = interop.run_learners(e, nt, nq, mq, seed=r + 1, train=True)
    ok = ok and tr["errors"] == 0 and tr["quizzes"] == nq
    print("round %2d: %dx%dx%d %3d threads: %s  (%.0f q/s; posted %d, combined %d)" % (
        r, Q, K, T, nt, "ok" if ok else "MISMATCH", tr["questions"] / tr["seconds"], e.get_option("posted_ops"), e.get_option("combined_batches")), flush=True)
    e.close()
    if not ok:
        sys.exit(1)
print("soak passed: %d rounds in %.0f s" % (rounds, time.time() - t0))
