import sys, time, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import _bootstrap; _bootstrap.load()
from chrono_wisc_amd import capi, scenes
print("affinity", len(os.sched_getaffinity(0)), "cpu_count", os.cpu_count())
for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us"):
    try: print(f, open(f).read().strip())
    except Exception as e: print(f, "n/a")
print([l for l in open("/proc/cpuinfo") if l.startswith("model name")][0].strip())
o = capi.CApi(os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "oracle/libsph_oracle.so"), "ref_")
s = scenes.dam_break_named("C2", o)
dt = 1e-5
for nt in (16, 32, 64, 128, 8, 1):
    o.set_num_threads(s.solver.ctx, nt)
    s.DoStepDynamics(dt)
    t0 = time.perf_counter(); n = 2 if nt > 1 else 1
    for _ in range(n): s.DoStepDynamics(dt)
    el = (time.perf_counter() - t0) / n
    print(nt, "threads: %.2f s/step, %.3g particle-steps/s" % (el, s.GetCounters().numAllMarkers / el), flush=True)
