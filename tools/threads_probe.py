"""Where the time of the threaded learner loop goes (engine counters of the combined sweeps), per thread count."""
import sys, os, resource, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from probqa_amd import interop

f = interop.PqaEngineFactory()
e, err = f.create_cpu_engine(interop.EngineDefinition(5, 1000, 1000, init_amount=0.1))
e.fill_synthetic(8.0, 0.5, 20260928)
e.set_option("select", int(os.environ.get("SELECT", "0")))
if "SPIN" in os.environ:
    e.set_option("combine_spin", int(os.environ["SPIN"]))
if "FUSE" in os.environ:
    e.set_option("fuse_update", int(os.environ["FUSE"]))
if "LINGER" in os.environ:
    e.set_option("combine_linger_us", int(os.environ["LINGER"]))
keys = ["combined_batches", "combined_requests", "update_flushes", "updates_flushed", "combined_ns_lock", "combined_ns_launch",
        "combined_ns_device", "combined_ns_relock", "combined_ns_select", "combined_ns_selmu", "combined_ns_readers", "posted_ops", "posted_drains"]
interop.run_learners(e, 1, 100, 30, seed=1, train=True)


def throttled():
    try:
        d = dict(l.split() for l in open("/sys/fs/cgroup/cpu.stat"))
        return int(d.get("nr_throttled", 0)), int(d.get("throttled_usec", 0))
    except OSError:
        return 0, 0


for nt in [int(x) for x in (sys.argv[1:] or ["1", "16", "64", "256"])]:
    b0 = [e.get_option(k) for k in keys]
    ru0, th0 = resource.getrusage(resource.RUSAGE_SELF), throttled()
    r = interop.run_learners(e, nt, 400 if nt == 1 else 2400, 30, seed=nt, train=True)
    ru1, th1 = resource.getrusage(resource.RUSAGE_SELF), throttled()
    cpu = (ru1.ru_utime - ru0.ru_utime) + (ru1.ru_stime - ru0.ru_stime)
    print("            cpu %.2f s (user %.2f sys %.2f) = %.1f cpus, %.0f us cpu per question | throttled %d times, %.0f ms | ctx switches %d vol %d invol"
          % (cpu, ru1.ru_utime - ru0.ru_utime, ru1.ru_stime - ru0.ru_stime, cpu / r["seconds"], cpu / r["questions"] * 1e6,
             th1[0] - th0[0], (th1[1] - th0[1]) / 1e3, ru1.ru_nvcsw - ru0.ru_nvcsw, ru1.ru_nivcsw - ru0.ru_nivcsw))
    d = dict(zip(keys, [e.get_option(k) - x for k, x in zip(keys, b0)]))
    nb = max(1, d["combined_batches"])
    print("threads %3d: %.0f q/s  %d questions in %.3f s | sweeps %d x %.1f req | per sweep us: lock %.1f (selMu %.1f readers %.1f) launch %.1f device %.1f relock %.1f select %.1f | RA/launch %.2f | posted %d in %d drains"
          % (nt, r["questions"] / r["seconds"], r["questions"], r["seconds"], d["combined_batches"], d["combined_requests"] / nb,
             d["combined_ns_lock"] / nb / 1e3, d["combined_ns_selmu"] / nb / 1e3, d["combined_ns_readers"] / nb / 1e3, d["combined_ns_launch"] / nb / 1e3, d["combined_ns_device"] / nb / 1e3,
             d["combined_ns_relock"] / nb / 1e3, d["combined_ns_select"] / nb / 1e3, d["updates_flushed"] / max(1, d["update_flushes"]), d["posted_ops"], d["posted_drains"]))
