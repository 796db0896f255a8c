"""Where the time of the threaded learner loop goes (engine counters of the combined sweeps), per thread count."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from probqa_amd import interop

f = interop.PqaEngineFactory()
e, err = f.create_cpu_engine(interop.EngineDefinition(5, 1000, 1000, init_amount=0.1))
e.fill_synthetic(8.0, 0.5, 20260928)
e.set_option("select", int(os.environ.get("SELECT", "0")))
if "SPIN" in os.environ:
    e.set_option("combine_spin", int(os.environ["SPIN"]))
if "LINGER" in os.environ:
    e.set_option("combine_linger_us", int(os.environ["LINGER"]))
keys = ["combined_batches", "combined_requests", "update_flushes", "updates_flushed", "combined_ns_lock", "combined_ns_launch",
        "combined_ns_device", "combined_ns_relock", "combined_ns_select", "combined_ns_selmu", "combined_ns_readers"]
interop.run_learners(e, 1, 100, 30, seed=1, train=True)
for nt in [int(x) for x in (sys.argv[1:] or ["1", "16", "64", "256"])]:
    b0 = [e.get_option(k) for k in keys]
    r = interop.run_learners(e, nt, 400 if nt == 1 else 2400, 30, seed=nt, train=True)
    d = dict(zip(keys, [e.get_option(k) - x for k, x in zip(keys, b0)]))
    nb = max(1, d["combined_batches"])
    print("threads %3d: %.0f q/s  %d questions in %.3f s | sweeps %d x %.1f req | per sweep us: lock %.1f (selMu %.1f readers %.1f) launch %.1f device %.1f relock %.1f select %.1f | RA/launch %.2f"
          % (nt, r["questions"] / r["seconds"], r["questions"], r["seconds"], d["combined_batches"], d["combined_requests"] / nb,
             d["combined_ns_lock"] / nb / 1e3, d["combined_ns_selmu"] / nb / 1e3, d["combined_ns_readers"] / nb / 1e3, d["combined_ns_launch"] / nb / 1e3, d["combined_ns_device"] / nb / 1e3,
             d["combined_ns_relock"] / nb / 1e3, d["combined_ns_select"] / nb / 1e3, d["updates_flushed"] / max(1, d["update_flushes"])))
