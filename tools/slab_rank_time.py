#!/usr/bin/env python3
"""What ONE slab rank of an N-rank run spends per step when it has a GPU to itself -- measured on one GPU.  N slab contexts live in this process
(threads, device-to-device messages) and step in lockstep through sph_slab_step, but a token serialises them: a rank holds it while it computes and hands
it over whenever it enters an exchange (by then the library has drained its stream), so its kernels never share the GPU with another rank's.  The time a
rank holds the token per step = its kernels + its host orchestration + its synchronisation gaps, i.e. the step of a dedicated GPU minus link transfers and
minus waiting for the neighbours.  Per-scope device times of one middle rank follow (profile scopes; every scope drained, so they sum to more).
usage: slab_rank_time.py [label=C3] [nranks=8] [steps=10] [python] [noref]       python: slab.py drives the step instead of sph_slab_step"""
import ctypes as C
import os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import _bootstrap; _bootstrap.load()
import numpy as np
from chrono_wisc_amd import capi, scenes, slab

label = sys.argv[1] if len(sys.argv) > 1 else "C3"
nranks = int(sys.argv[2]) if len(sys.argv) > 2 else 8
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
native = "python" not in sys.argv[4:]
api = capi.CApi()
dt = scenes.DAM_BREAK_CONFIGS[label][2]
token = threading.Lock()


class TokenEndpoint:
    """A ThreadTransport endpoint that gives the GPU token away for the duration of every exchange."""

    def __init__(self, ep):
        self.ep, self.rank, self.nranks, self.tr = ep, ep.rank, ep.nranks, ep.tr
        self.held, self.t_in = 0.0, None

    def _out(self):
        self.held += time.perf_counter() - self.t_in
        token.release()

    def _in(self):
        token.acquire()
        self.t_in = time.perf_counter()

    def exchange(self, sends, recvs):
        self._out()
        try:
            return self.ep.exchange(sends, recvs)
        finally:
            self._in()

    def exchange_ints(self, sends, peers):
        self._out()
        try:
            return self.ep.exchange_ints(sends, peers)
        finally:
            self._in()

    def __getattr__(self, name):
        return getattr(self.ep, name)


tr = slab.ThreadTransport(nranks, device_direct=True)
ranks, eps = [], []
for r in range(nranks):
    s = scenes.dam_break_named(label, api, initialize=False, check_errors=False)
    cuts = s.InitializeSlab(r, nranks)
    eps.append(TokenEndpoint(tr.endpoint(r)))
    ranks.append(slab.SlabRank(s, eps[-1], r, nranks, cuts))
errs = []


def work(i, n):
    try:
        rk, ep = ranks[i], eps[i]
        for _ in range(n):
            ep._in()
            (rk.step_native if native else rk.step)(dt)
            rk.sys.Synchronize()
            ep._out()
    except Exception as e:  # noqa: BLE001
        errs.append(e)
        tr.failed.set()
        if token.locked():
            try:
                token.release()
            except RuntimeError:
                pass


def run(n):
    ts = [threading.Thread(target=work, args=(i, n)) for i in range(nranks)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    if errs:
        raise errs[0]


run(6)  # warm-up: two-round preamble, buffer growth, resident mode entered
for e in eps:
    e.held = 0.0
t0 = time.perf_counter()
run(steps)
wall = time.perf_counter() - t0
single = float("nan")
if "noref" not in sys.argv[4:]:  # (kernel traces: the slab ranks' launches only)
    one = scenes.dam_break_named(label, api, check_errors=False)
    for _ in range(6):
        one.DoStepDynamics(dt)
    one.Synchronize()
    t1 = time.perf_counter()
    for _ in range(steps):
        one.DoStepDynamics(dt)
    one.Synchronize()
    single = (time.perf_counter() - t1) / steps * 1e3
    del one
n_global = ranks[0].sys.n_global
print("%s in %d slabs, %s driver, one rank on the GPU at a time (%d steps):" % (label, nranks, "library" if native else "slab.py", steps))
fit = []
for r, e in zip(ranks, eps):
    c = r.sys.GetCounters()
    nf = int((r.sys.solver.rho_pres_mu_type()[:, 3] < -0.5).sum())
    fit.append((nf, c.numAllMarkers - nf, c.numExtendedParticles - c.numAllMarkers, e.held / steps * 1e3))
    print("  rank %d: %8d owned (%8d fluid, %7d wall), %8d rows: %.3f ms per step (%.3f ns per owned marker)" % (r.rank, c.numAllMarkers, nf, c.numAllMarkers - nf,
          c.numExtendedParticles, e.held / steps * 1e3, e.held / steps / c.numAllMarkers * 1e9))
M = np.array([[f, w, g, 1.0] for f, w, g, _t in fit], dtype=np.float64)
y = np.array([t for *_x, t in fit], dtype=np.float64)
coef, *_ = np.linalg.lstsq(M, y, rcond=None)
print("  least squares over the ranks: ms = %.3e fluid + %.3e wall + %.3e ghost rows + %.3f  =>  wall / fluid = %.2f, ghost / fluid = %.2f" %
      (coef[0], coef[1], coef[2], coef[3], coef[1] / coef[0], coef[2] / coef[0]))
worst = max(e.held for e in eps) / steps * 1e3
print("single context: %.3f ms per step (%.3f ns per marker); slowest rank %.3f ms => %.2f x before link time, sum over ranks %.3f ms (%.2f x the single context)"
      % (single, single / n_global * 1e6, worst, single / worst, sum(e.held for e in eps) / steps * 1e3, sum(e.held for e in eps) / steps * 1e3 / single))
if "noref" in sys.argv[4:]:
    sys.exit(0)
for mid in (ranks[nranks // 2], ranks[-1]):
    api.profile_enable(mid.ctx, 1)
    run(5)
    parts = []
    for name in ("sort", "neigh", "adami", "force", "integrate", "halo", "refresh", "rows"):
        ms, k = C.c_double(), C.c_uint64()
        if api.profile_get(mid.ctx, name.encode(), C.byref(ms), C.byref(k)) == 0 and k.value:
            parts.append("%s %.0f us" % (name, ms.value / 5 * 1e3))
    api.profile_enable(mid.ctx, 0)
    print("  rank %d, device time per step by profile scope: %s" % (mid.rank, ", ".join(parts)))
