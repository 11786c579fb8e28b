"""CPU tests of the drop-in boundary: the product library loads without a GPU, exports every symbol that
include/chsph_mi.h declares, agrees on struct sizes, and refuses to run without a HIP device (no CPU fallback)."""
import ctypes as C
import os
import re
import subprocess

import pytest

from chrono_wisc_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_functions():
    text = open(capi.HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(sph_[a-z0-9_]+)\s*\(", text)))


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(capi.PRODUCT_LIB):
        subprocess.check_call(["make", "-C", os.path.dirname(capi.PRODUCT_LIB)])
    return capi.CApi().lib  # through the binding: it settles which ROCm runtime the process uses before the library is mapped


def test_header_declares_the_expected_surface():
    names = _declared_functions()
    for must in ("sph_create", "sph_add_markers", "sph_add_rigid_body", "sph_initialize", "sph_step", "sph_current_step", "sph_load_body_states",
                 "sph_compute_body_wrenches", "sph_get_body_wrenches", "sph_update_body_markers", "sph_get_positions", "sph_get_velocities",
                 "sph_get_properties", "sph_get_accelerations", "sph_get_neighbor_csr", "sph_last_error", "sph_destroy"):
        assert must in names


def test_library_exports_every_declared_symbol(lib):
    missing = [n for n in _declared_functions() if not hasattr(lib, n)]
    assert not missing, "declared in include/chsph_mi.h but not exported: %s" % missing


def test_no_torch_or_cxx_types_in_the_abi():
    text = re.sub(r"/\*.*?\*/", "", open(capi.HEADER).read(), flags=re.S)  # declarations only
    assert "torch" not in text.lower() and "std::" not in text and "at::" not in text


def test_struct_sizes_match_the_header(lib):
    lib.sph_abi_sizeof.restype = C.c_size_t
    for which, st in enumerate((capi.SphParams, capi.SphCounters, capi.SphBodyState, capi.SphWrench)):
        assert lib.sph_abi_sizeof(which) == C.sizeof(st)
    lib.sph_abi_version.restype = C.c_uint32
    assert lib.sph_abi_version() == 2


def test_defaults_are_initparams(lib):
    p = capi.SphParams()
    lib.sph_params_default(C.byref(p))
    # sph/ChFsiFluidSystemSPH.cpp:81-176
    assert p.rho0 == 1000 and p.d0 == pytest.approx(0.01) and p.d0_multiplier == pytest.approx(1.2)
    assert p.integration_scheme == capi.SPH_SCHEME_RK2 and p.eos_type == capi.SPH_EOS_ISOTHERMAL
    assert p.shifting_method == capi.SPH_SHIFT_XSPH and p.num_proximity_search_steps == 1 and p.dT == -1
    assert p.density_reinit_steps == 2147483647 and p.use_default_limits == 1


def test_product_refuses_to_run_without_a_gpu(lib):
    """Here (no GPU) creating a context must fail loudly; on the GPU box this test is a no-op."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    p = capi.SphParams()
    lib.sph_params_default(C.byref(p))
    lib.sph_create.restype = C.c_void_p
    lib.sph_create_error.restype = C.c_char_p
    ctx = lib.sph_create(C.byref(p), -1)
    assert not ctx
    assert b"no CPU fallback" in lib.sph_create_error()
    with pytest.raises(capi.SphError):
        from chrono_wisc_amd import scenes
        scenes.dam_break_named("tiny", capi.CApi())


def test_product_sources_do_not_reference_the_oracle():
    """The oracle is test infrastructure: nothing under chrono-wisc_amd/ may import, link or call it."""
    pkg = os.path.join(ROOT, "chrono-wisc_amd")
    for dirpath, _d, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp", ".hpp", "Makefile")):
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "libsph_oracle" not in text and "sph_oracle.h" not in text and "ref_step" not in text, os.path.join(dirpath, f)


def test_every_environment_switch_of_the_library_is_documented_in_the_header():
    """The library reads its A/B and test switches with getenv("CHSPH_..."): each of them must be listed in include/chsph_mi.h ("environment switches")."""
    import re
    names = set()
    src = os.path.join(ROOT, "chrono-wisc_amd", "csrc")
    for f in os.listdir(src):
        if f.endswith((".hip", ".h")):
            names |= set(re.findall(r'getenv\("(CHSPH_[A-Z0-9_]+)"\)', open(os.path.join(src, f), errors="ignore").read()))
    assert len(names) > 20
    header = open(os.path.join(ROOT, "include", "chsph_mi.h")).read()
    missing = sorted(n for n in names if n not in header)
    assert not missing, missing


def test_committed_bench_line_follows_the_contract():
    """The JSON line bench.py printed on the GPU box (profiles/r04_final_bench_line.json, the final kernels of round 4): keys and types the driver relies on,
    and the live force-launch figure against the kernel trace of the same command (profiles/r04_final_bench_kernel_stats.csv)."""
    import csv
    import json
    import os
    prof = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "profiles")
    path = os.path.join(prof, "r04_final_bench_line.json")
    d = json.loads(open(path).read().strip().splitlines()[-1])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline",
              "cpu_baseline"):
        assert k in d, k
    assert d["metric"] == "SPH particle-steps/sec" and d["unit"] == "particle-steps/s" and d["higher_is_better"] is True and d["vs_baseline"] is None
    assert d["dtype"] == "f32" and d["data"] == "synthetic" and "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and 0 < r["frac"] < 1
    assert abs(r["achieved"] - r["algorithmic_bytes_per_launch"] / (r["avg_launch_ms"] * 1e-3) / 1e9) < 1e-6 * r["achieved"]
    assert r["traffic"] is None or r["traffic"] > 0
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and c["unit"] == d["unit"] and "sample" in c
    assert c["single_thread"]["cores"] == 1 and 0 < c["single_thread"]["value"] <= c["value"] and c["cpu_model"]  # one thread and all cores, CPU model stated
    assert "check_errors" in d["config"] and "state" in d["config"] and "arithmetic" in d["config"]
    assert abs(d["value"] - 1234797 * d["steps"] / (d["ms_per_step"] * 1e-3 * d["steps"])) < 1e-6 * d["value"]
    # the event-pair figure minus its calibrated bracketing cost is what the line prices the roofline with ...
    assert abs(r["avg_launch_ms"] - (r["avg_launch_raw_ms"] - r["event_pair_overhead_ms"])) < 1e-9 and 0 < r["event_pair_overhead_ms"] < 0.03
    assert r["gather_ceiling"]["floor_ms"] < r["isolated_avg_launch_ms"] and d["scaling"] == "strong"
    assert d["single_gpu_same_workload"]["value"] > 0 and "C3" in d["single_gpu_same_workload"]["workload"]
    # ... and the line printed under rocprofv3 agrees with the profiler's own average duration of that kernel in the same run
    under = json.loads(open(os.path.join(prof, "r04_final_bench_line_under_rocprof.json")).read().strip().splitlines()[-1])
    rows = [row for row in csv.DictReader(open(os.path.join(prof, "r04_final_bench_kernel_stats.csv"))) if "::k_force_fast<" in row["Name"]]
    assert len(rows) == 1
    trace_ms = float(rows[0]["AverageNs"]) * 1e-6
    assert abs(under["roofline"]["avg_launch_ms"] - trace_ms) < 0.05 * trace_ms, (under["roofline"]["avg_launch_ms"], trace_ms)


def test_bench_started_bare_spawns_its_ranks():
    """`python bench.py --gpus N` as the driver starts it (no launcher, WORLD_SIZE unset) must start N ranks itself, as child processes,
    relay rank 0's single JSON line and exit with the children's status.  --rendezvous-only keeps the ranks off the GPU."""
    import json
    import subprocess
    import sys
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--rendezvous-only"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["rank_sum"] == 1.0
    # a world size that does not match --gpus is an error, not a silent single-rank run
    env2 = dict(env, WORLD_SIZE="1", RANK="0")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--rendezvous-only"], env=env2, capture_output=True, text=True, timeout=300)
    assert r.returncode == 2


@pytest.mark.parametrize("args", [("box", 0.3, 0.4, 0.5), ("box", 0.1, 0.4, 0.5), ("sphere", 0.21, 1), ("sphere", 0.21, 0), ("cylinder", 0.12, 0.2, 1), ("cylinder", 0.12, 0.2, 0),
                                  ("cylinder", 0.04, 0.2, 1), ("annulus", 0.1, 0.2, 0.3, 1), ("annulus", 0.1, 0.2, 0.3, 0)])
def test_cxx_bce_generators_match_the_python_mirror(args):
    """The C++ facade's body BCE generators (what the Chrono-side adapter uses for ChBodyGeometry shapes) against the Python mirror,
    point for point in the same order (host-only code: runs without a GPU)."""
    import subprocess
    import numpy as np
    from chrono_wisc_amd import fsi
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
    exe = os.path.join(root, "chrono-wisc_amd", "host", "bce_points_MI")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", os.path.dirname(exe)])
    d0, layers = 0.025, 3
    raw = subprocess.run([exe, repr(d0), str(layers)] + [str(a) for a in args], check=True, capture_output=True).stdout
    got = np.frombuffer(raw, dtype=np.float64).reshape(-1, 3)
    s = fsi.ChFsiFluidSystemSPH(None)
    s.SetInitialSpacing(d0)
    s.SetNumBCELayers(layers)
    kind = args[0]
    if kind == "box":
        want = s.CreatePointsBoxInterior(args[1:4])
    elif kind == "sphere":
        want = s.CreatePointsSphereInterior(args[1], bool(args[2]))
    elif kind == "cylinder":
        want = s.CreatePointsCylinderInterior(args[1], args[2], bool(args[3]))
    else:
        want = s.CreatePointsCylinderAnnulus(args[1], args[2], args[3], bool(args[4]))
    assert got.shape == np.asarray(want).shape and len(got) > 0
    np.testing.assert_allclose(got, want, rtol=0, atol=1e-15)


_ORDER_SCRIPT = r"""
import os, sys, json
sys.path.insert(0, %(root)r)
import _bootstrap
_bootstrap.load()
from chrono_wisc_amd import capi
order = sys.argv[1]
if order == "torch_first":
    import torch
    api = capi.CApi()
elif order == "solver_first":
    api = capi.CApi()
    import torch
elif order == "solver_only":  # a tool that never imports torch: the device-to-device transport's own handle on the HIP runtime (slab._hiplib)
    api = capi.CApi()
    from chrono_wisc_amd import slab
    slab._hiplib()
else:  # raw: what a foreign binding does wrong -- the library straight through the loader, then torch
    import ctypes
    lib = ctypes.CDLL(capi.PRODUCT_LIB, mode=ctypes.RTLD_GLOBAL)
    import torch
    lib.sph_runtime_census.argtypes = [ctypes.c_char_p, ctypes.c_size_t]
    buf = ctypes.create_string_buffer(4096)
    worst = lib.sph_runtime_census(buf, 4096)
    p = capi.SphParams()
    lib.sph_params_default(ctypes.byref(p))
    lib.sph_create.restype = ctypes.c_void_p
    lib.sph_create_error.restype = ctypes.c_char_p
    ctx = lib.sph_create(ctypes.byref(p), -1)
    err = lib.sph_create_error().decode()
    guard = ""
    try:
        capi.assert_single_runtime("test")
    except capi.SphError as e:
        guard = str(e)
    print(json.dumps({"worst": worst, "ctx": bool(ctx), "err": err, "guard": guard, "maps": capi.mapped_runtimes()}))
    sys.exit(0)
buf = __import__("ctypes").create_string_buffer(4096)
worst = api.runtime_census(buf, 4096)
capi.assert_single_runtime("test")
print(json.dumps({"worst": worst, "maps": capi.mapped_runtimes(), "listing": buf.value.decode()}))
"""


def _run_order(order):
    import json
    import subprocess
    import sys
    env = dict(os.environ)
    env.pop("CHSPH_RUNTIME", None)
    out = subprocess.run([sys.executable, "-c", _ORDER_SCRIPT % {"root": ROOT}, order], capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    return json.loads(out.stdout.strip().splitlines()[-1])


@pytest.mark.parametrize("order", ["torch_first", "solver_first", "solver_only"])
def test_one_rocm_runtime_in_either_import_order(order):
    """The round-3 exit abort: the solver loaded before torch mapped /opt/rocm's HIP + HSA runtimes next to torch's bundled ones.  The binding
    now maps torch's bundled libamdhip64 itself before the solver, so both orders end with ONE libamdhip64, ONE libhsa-runtime64 and at
    most one librccl in the process (include/chsph_mi.h, sph_runtime_census)."""
    r = _run_order(order)
    assert r["worst"] == 1, r
    for lib, paths in r["maps"].items():
        assert len(paths) <= 1, (lib, paths)  # (no profiler in this process: one HSA runtime as well)
    assert len(r["maps"]["libamdhip64.so"]) == 1 and len(r["maps"]["libhsa-runtime64.so"]) == 1


def test_a_second_runtime_is_detected_and_refused():
    """What a binding that ignores the rule gets: the census counts two copies, sph_create refuses with a message that names them, and the
    Python guard raises."""
    r = _run_order("raw")
    if r["worst"] <= 1:
        pytest.skip("torch's wheel and the ROCm installation share one runtime on this machine")
    assert not r["ctx"]
    assert "more than one ROCm runtime" in r["err"] and "libamdhip64" in r["err"]
    assert "more than one ROCm runtime" in r["guard"]


def _icosphere(radius, subdiv):
    import numpy as np
    t = (1 + 5 ** 0.5) / 2
    V = [(-1, t, 0), (1, t, 0), (-1, -t, 0), (1, -t, 0), (0, -1, t), (0, 1, t), (0, -1, -t), (0, 1, -t), (t, 0, -1), (t, 0, 1), (-t, 0, -1), (-t, 0, 1)]
    F = [(0, 11, 5), (0, 5, 1), (0, 1, 7), (0, 7, 10), (0, 10, 11), (1, 5, 9), (5, 11, 4), (11, 10, 2), (10, 7, 6), (7, 1, 8), (3, 9, 4), (3, 4, 2), (3, 2, 6),
         (3, 6, 8), (3, 8, 9), (4, 9, 5), (2, 4, 11), (6, 2, 10), (8, 6, 7), (9, 8, 1)]
    V = [np.array(v, dtype=np.float64) / np.linalg.norm(v) for v in V]
    for _ in range(subdiv):
        mid, F2 = {}, []

        def m(a, b):
            k = (min(a, b), max(a, b))
            if k not in mid:
                p = V[a] + V[b]
                V.append(p / np.linalg.norm(p))
                mid[k] = len(V) - 1
            return mid[k]

        for a, b, c in F:
            ab, bc, ca = m(a, b), m(b, c), m(c, a)
            F2 += [(a, ab, ca), (b, bc, ab), (c, ca, bc), (ab, bc, ca)]
        F = F2
    return radius * np.array(V), np.array(F, dtype=np.int32)


def test_triangle_mesh_bce_points():
    """CreatePointsMesh (sph/ChFsiFluidSystemSPH.cpp:2627-2714: every point of the d0 grid, anchored at the bounding-box minimum, that lies inside the
    closed surface by a two-ray parity test): the Python mirror against geometry -- a box mesh must give exactly the grid points strictly inside
    the box, an icosphere the grid points inside the sphere up to the faceting -- and the C++ mirror (what the Chrono-side adapter calls for
    ChBodyGeometry::coll_meshes) against the Python one, point for point."""
    import struct
    import subprocess
    import tempfile
    import numpy as np
    from chrono_wisc_amd import fsi
    d0 = 0.025
    s = fsi.ChFsiFluidSystemSPH(None)
    s.SetInitialSpacing(d0)
    # a box (12 triangles), lower corner at (0.1, -0.2, 0.3)
    lo, sz = np.array([0.1, -0.2, 0.3]), np.array([0.31, 0.2, 0.26])
    V = np.array([[x, y, z] for x in (0, 1) for y in (0, 1) for z in (0, 1)], dtype=np.float64) * sz + lo
    F = np.array([[0, 1, 3], [0, 3, 2], [4, 6, 7], [4, 7, 5], [0, 4, 5], [0, 5, 1], [2, 3, 7], [2, 7, 6], [0, 2, 6], [0, 6, 4], [1, 5, 7], [1, 7, 3]], dtype=np.int32)
    pts = s.CreatePointsMesh(V, F)
    sp = float(np.float32(d0))
    n = [int(np.ceil(sz[k] / sp - 1e-12)) for k in range(3)]          # grid points lo + i * sp < hi
    want = np.array([[lo[0] + i * sp, lo[1] + j * sp, lo[2] + k * sp] for i in range(n[0]) for j in range(n[1]) for k in range(n[2])])
    # The first layers sit ON the faces and are shifted 1e-9 inwards.  In y and z both rays leave them away from that face, so they count as inside; in x
    # the second ray (-3, 0.7, 10) re-crosses the face 3e-10 behind the point, which the hit threshold (t > 1e-6) ignores: an even count, outside --
    # the reference's behaviour for points within 1e-6 of the surface, reproduced as it is.
    inside = np.all((want + 1e-9 > lo) & (want + 1e-9 < lo + sz), axis=1) & (want[:, 0] > lo[0] + 0.5 * sp)
    np.testing.assert_allclose(pts, want[inside], rtol=0, atol=1e-12)
    # an icosphere: all points well inside the inscribed sphere are there, none outside the circumscribed one
    R = 0.2
    Vs, Fs = _icosphere(R, 2)
    ps = s.CreatePointsMesh(Vs, Fs)
    r = np.linalg.norm(ps, axis=1)
    assert r.max() <= R + 1e-9 and len(ps) > 1500
    lo_s = Vs.min(axis=0)
    grid = np.stack(np.meshgrid(*[lo_s[k] + sp * np.arange(int(2 * R / sp) + 2) for k in range(3)], indexing="ij"), axis=-1).reshape(-1, 3)
    deep = grid[np.linalg.norm(grid, axis=1) < 0.95 * R]
    have = {tuple(np.round(p / 1e-9).astype(np.int64)) for p in ps}
    assert all(tuple(np.round(p / 1e-9).astype(np.int64)) in have for p in deep)
    # C++ mirror == Python mirror
    exe = os.path.join(ROOT, "chrono-wisc_amd", "host", "bce_points_MI")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", os.path.dirname(exe)])
    with tempfile.NamedTemporaryFile(suffix=".mesh") as f:
        f.write(struct.pack("<ii", len(Vs), len(Fs)))
        f.write(np.ascontiguousarray(Vs, dtype=np.float64).tobytes())
        f.write(np.ascontiguousarray(Fs, dtype=np.int32).tobytes())
        f.flush()
        raw = subprocess.run([exe, repr(d0), "3", "mesh", f.name], check=True, capture_output=True).stdout
    got = np.frombuffer(raw, dtype=np.float64).reshape(-1, 3)
    np.testing.assert_array_equal(got, ps)
