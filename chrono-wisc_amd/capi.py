"""ctypes binding of the C ABI declared in include/chsph_mi.h.

The struct layouts are parsed from the header itself so that the binding cannot drift
from the ABI.  The same class binds any library that exports the ABI under a symbol
prefix: the product library (`sph_`, chrono-wisc_amd/csrc/libchsph_mi.so).  Tests bind the
CPU oracle through the same class with prefix `ref_`; nothing in this package loads it.
"""
import ctypes as C
import os
import re

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "chsph_mi.h")
PRODUCT_LIB = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc", "libchsph_mi.so")

_CT = {"int32_t": C.c_int32, "uint32_t": C.c_uint32, "uint64_t": C.c_uint64, "float": C.c_float, "double": C.c_double}


def _parse_struct(name, text):
    m = re.search(r"typedef struct %s \{(.*?)\} %s;" % (name, name), text, re.S)
    if not m:
        raise RuntimeError("struct %s not found in %s" % (name, HEADER))
    body = re.sub(r"/\*.*?\*/", "", m.group(1), flags=re.S)
    fields = []
    for decl in body.split(";"):
        decl = decl.strip()
        if not decl:
            continue
        ty, rest = decl.split(None, 1)
        for var in rest.split(","):
            var = var.strip()
            am = re.match(r"(\w+)\[(\d+)\]$", var)
            if am:
                fields.append((am.group(1), _CT[ty] * int(am.group(2))))
            else:
                fields.append((var, _CT[ty]))
    return type(name, (C.Structure,), {"_fields_": fields})


_text = open(HEADER).read()
SphParams = _parse_struct("sph_params", _text)
SphCounters = _parse_struct("sph_counters", _text)
SphBodyState = _parse_struct("sph_body_state", _text)
SphWrench = _parse_struct("sph_wrench", _text)
SphSlabConfig = _parse_struct("sph_slab_config", _text)


# sph_slab_transport (include/chsph_mi.h): function pointers, so it is spelled out here instead of parsed
SLAB_EXCHANGE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.c_int, C.c_void_p)
SLAB_COUNTS_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32))
SLAB_ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p)


class SphSlabTransport(C.Structure):
    _fields_ = [("user", C.c_void_p), ("rank", C.c_int32), ("nranks", C.c_int32), ("stream_ordered", C.c_int32), ("exchange", SLAB_EXCHANGE_FN),
                ("exchange_counts", SLAB_COUNTS_FN), ("allreduce", SLAB_ALLREDUCE_FN)]


_ENUMS = {k: int(v) for k, v in re.findall(r"(SPH_[A-Z0-9_]+) = (-?\d+)", _text)}
globals().update(_ENUMS)


def params_to_dict(p):
    out = {}
    for name, ty in p._fields_:
        v = getattr(p, name)
        out[name] = list(v) if hasattr(v, "__len__") else v
    return out


class SphError(RuntimeError):
    """Raised where the reference throws std::runtime_error (sph/utils/SphUtilsDevice.cuh:98-136)."""


def _fp(a, dtype=np.float32):
    if a is None:
        return None, None
    a = np.ascontiguousarray(a, dtype=dtype)
    return a, a.ctypes.data_as(C.c_void_p)


# ---- one ROCm runtime per process (include/chsph_mi.h, sph_runtime_census) ---------------------------------------------------------
# libchsph_mi.so needs `libamdhip64.so.7`.  PyTorch's wheel bundles its own libamdhip64 / libhsa-runtime64 / librccl with the same
# SONAMEs.  With torch imported first the loader hands the solver torch's copy (one runtime).  With the solver loaded first and torch
# imported later, torch's copies are mapped next to /opt/rocm's: two HIP and two HSA runtimes in one process.  So that the ORDER cannot
# matter in a Python process, the binding maps torch's bundled libamdhip64 itself (without importing torch) before it loads the solver;
# `CHSPH_RUNTIME=system` keeps the ROCm installation's runtime instead (then torch must not be imported into the process at all).
_RUNTIME_LIBS = ("libamdhip64.so", "libhsa-runtime64.so", "librccl.so")


def mapped_runtimes():
    """{library name: sorted distinct paths mapped in this process} read from /proc/self/maps."""
    found = {k: set() for k in _RUNTIME_LIBS}
    try:
        with open("/proc/self/maps") as f:
            for line in f:
                k = line.find("/")
                if k < 0:
                    continue
                path = line[k:].rstrip("\n")
                base = os.path.basename(path)
                for lib in _RUNTIME_LIBS:
                    if base.startswith(lib):
                        found[lib].add(path)
    except OSError:
        pass
    return {k: sorted(v) for k, v in found.items()}


def assert_single_runtime(where="chrono_wisc_amd"):
    """Raise SphError when two copies of the HIP / HSA runtime or of RCCL are mapped in this process."""
    # (a second libhsa-runtime64 alone is what rocprofv3 leaves behind -- its tool library maps the ROCm installation's HSA before the program
    # starts -- and is harmless; two HIP runtimes or two RCCLs are not)
    bad = {k: v for k, v in mapped_runtimes().items() if len(v) > 1 and k != "libhsa-runtime64.so"}
    if bad:
        raise SphError("%s: more than one ROCm runtime is mapped in this process: %s.  Import torch before anything loads "
                       "libchsph_mi.so, or not at all (INTEGRATION.md, 'One ROCm runtime per process')." % (where, bad))


def _bind_torch_runtime():
    """Map torch's bundled HIP runtime (if torch is installed and not yet imported) so that the solver and a later `import torch` share it."""
    import sys
    if os.environ.get("CHSPH_RUNTIME", "") == "system" or "torch" in sys.modules or mapped_runtimes()["libamdhip64.so"]:
        return None  # (a HIP runtime is already mapped: the solver will bind to that one)
    import importlib.util
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return None
    lib = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if not os.path.exists(lib):
        return None
    return C.CDLL(lib, mode=C.RTLD_GLOBAL)


class CApi:
    """One loaded library exporting the chsph_mi ABI under `prefix`."""

    def __init__(self, path=PRODUCT_LIB, prefix="sph_"):
        if not os.path.exists(path):
            raise SphError(
                "%s not found: the HIP extension is not built (run `python -c 'import __graft_entry__ as g; g.build()'`). "
                "There is no CPU fallback for the product path." % path)
        self.path = path
        self.prefix = prefix
        if prefix == "sph_":
            self._runtime = _bind_torch_runtime()
        self.lib = C.CDLL(path, mode=C.RTLD_GLOBAL if prefix == "sph_" else C.RTLD_LOCAL)
        if prefix == "sph_":
            assert_single_runtime("loading %s" % path)
        f = self._f
        vp, sz, i, d = C.c_void_p, C.c_size_t, C.c_int, C.c_double
        f("params_default", None, [C.POINTER(SphParams)])
        if prefix == "sph_":
            f("create", vp, [C.POINTER(SphParams), i])
            f("abi_version", C.c_uint32, [])
            f("abi_sizeof", sz, [i])
            f("runtime_census", i, [C.c_char_p, sz])
            f("create_error", C.c_char_p, [])
            f("synchronize", i, [vp])
            f("set_stream", i, [vp, vp])
            f("get_device_view", i, [vp, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(sz)])
            f("profile_enable", i, [vp, i])
            f("profile_get", i, [vp, C.c_char_p, C.POINTER(d), C.POINTER(C.c_uint64)])
            f("profile_overhead", i, [vp, i, C.POINTER(d)])
            f("bench_rhs", i, [vp, i, C.POINTER(d)])
            f("set_arithmetic", i, [vp, i])
            f("get_arithmetic", i, [vp])
            f("bench_copy", i, [vp, sz, i, C.POINTER(d)])
            f("staged_tile_stats", i, [vp, C.c_uint32 * 4])
            f("debug_search_classes", i, [vp, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)])
            f("debug_kernel_functions", i, [vp, i, sz, vp, vp, C.c_float, vp, vp])
            for which, st in enumerate((SphParams, SphCounters, SphBodyState, SphWrench)):
                if self.lib.sph_abi_sizeof(which) != C.sizeof(st):
                    raise SphError("ABI mismatch for struct %d: header %d vs library %d" % (which, C.sizeof(st), self.lib.sph_abi_sizeof(which)))
        else:
            f("create", vp, [C.POINTER(SphParams)])
            f("set_num_threads", i, [vp, i])
        f("destroy", None, [vp])
        f("last_error", C.c_char_p, [vp])
        f("set_params", i, [vp, C.POINTER(SphParams)])
        f("get_params", i, [vp, C.POINTER(SphParams)])
        f("get_counters", i, [vp, C.POINTER(SphCounters)])
        f("set_step_size", i, [vp, d])
        f("set_check_errors", i, [vp, i])
        f("add_markers", i, [vp, i, sz, vp, vp, vp])
        f("add_rigid_body", i, [vp, sz, vp])
        f("set_marker_stress", i, [vp, sz, sz, vp, vp])
        f("get_stress", i, [vp, vp, sz])
        f("set_marker_consolidation", i, [vp, sz, sz, vp])
        f("get_consolidation", i, [vp, vp, sz])
        f("debug_pow_2p55", i, [vp, vp, vp, sz])
        f("initialize", i, [vp, sz, C.POINTER(SphBodyState)])
        f("step", i, [vp, d, d])
        f("current_step", d, [vp])
        f("get_isph_iterations", i, [vp, C.POINTER(i), C.POINTER(i)])
        f("set_isph_diagnostics", i, [vp, i])
        f("get_isph_g_tensor", i, [vp, vp, vp, sz])
        f("force_proximity_search", i, [vp])
        f("load_body_states", i, [vp, sz, C.POINTER(SphBodyState)])
        f("compute_body_wrenches", i, [vp])
        f("get_body_wrenches", i, [vp, sz, C.POINTER(SphWrench)])
        f("update_body_markers", i, [vp])
        for g in ("positions", "velocities", "properties", "accelerations", "derivatives", "rho_pres_mu_type"):
            f("get_" + g, i, [vp, vp, sz])
        f("get_neighbor_csr", i, [vp, vp, sz, vp, sz, C.POINTER(sz), C.POINTER(sz)])
        f("get_sorted_index", i, [vp, vp, vp, sz])
        f("get_sorted_state", i, [vp, vp, vp, vp, sz])
        # step phases and slab hooks
        f("step_begin", i, [vp, d])
        f("step_needs_search", i, [vp])
        f("step_search", i, [vp])
        f("step_num_stages", i, [vp])
        f("step_stage", i, [vp, i, d])
        f("step_stage_target", i, [vp, i])
        f("step_end", i, [vp])
        f("set_global_ids", i, [vp, sz, vp])
        f("get_global_ids", i, [vp, vp, sz])
        f("slab_configure", i, [vp, C.POINTER(SphSlabConfig)])
        u2 = C.c_uint32 * 2
        f("slab_classify", i, [vp, u2])
        f("slab_apply_migration", i, [vp, u2, u2])
        f("slab_apply_ghosts", i, [vp, u2])
        f("slab_buffer", i, [vp, i, i, i, sz, C.POINTER(vp), C.POINTER(sz), C.POINTER(i)])
        f("slab_halo_pack", i, [vp, i, i])
        f("slab_halo_unpack", i, [vp, i, i])
        f("slab_one_round", i, [vp])
        f("slab_refresh_pack", i, [vp, u2])
        f("slab_refresh_apply", i, [vp, u2])
        f("slab_rigid_subset", i, [vp, C.c_uint32, sz, vp])
        f("slab_wrench_table", i, [vp, C.POINTER(vp), C.POINTER(sz), C.POINTER(i)])
        f("slab_wrench_reduce", i, [vp])
        if prefix == "sph_":  # the whole slab step inside the library + built-in RCCL transport (product only)
            f("slab_step", i, [vp, d, d, C.POINTER(SphSlabTransport)])
            f("slab_rccl_unique_id", sz, [vp, sz])
            f("slab_transport_rccl", C.POINTER(SphSlabTransport), [vp, vp, sz, i, i])
            f("slab_transport_free", None, [C.POINTER(SphSlabTransport)])
            f("slab_group_create", vp, [i])
            f("slab_group_destroy", i, [vp])
            f("slab_transport_inproc", C.POINTER(SphSlabTransport), [vp, vp, i])
            f("slab_body_wrenches", i, [vp, C.POINTER(SphSlabTransport)])
            f("slab_current_step", d, [vp, C.POINTER(SphSlabTransport)])
            f("slab_overlapped_halos", C.c_uint64, [vp])
        # I2SPH phases (slab runs) and per-field ghost exchange
        u = C.c_uint32
        f("isph_phase", i, [vp, i, d])
        f("isph_sweep", i, [vp, i, u])
        f("isph_sweep_ran", i, [vp, i, u])
        f("isph_reduce_buffer", i, [vp, i, u, C.POINTER(vp), C.POINTER(sz), C.POINTER(i)])
        f("slab_field_pack", i, [vp, i, u, i])
        f("slab_field_unpack", i, [vp, i, u, i])

    def _f(self, name, restype, argtypes):
        fn = getattr(self.lib, self.prefix + name)
        fn.restype = restype
        fn.argtypes = argtypes
        setattr(self, name, fn)

    def default_params(self):
        p = SphParams()
        self.params_default(C.byref(p))
        return p


def make_body_states(states):
    """states: iterable of dicts with pos, rot, lin_vel, ang_vel, lin_acc, ang_acc."""
    states = list(states)
    arr = (SphBodyState * max(len(states), 1))()
    for k, s in enumerate(states):
        for name, n in (("pos", 3), ("rot", 4), ("lin_vel", 3), ("ang_vel", 3), ("lin_acc", 3), ("ang_acc", 3)):
            v = s.get(name, (1, 0, 0, 0) if name == "rot" else (0, 0, 0))
            for q in range(n):
                getattr(arr[k], name)[q] = float(v[q])
    return arr, len(states)


class Solver:
    """Thin object wrapper over one context: the calls of ChFsiFluidSystemSPH's private overrides
    (sph/ChFsiFluidSystemSPH.h:555-669) one to one; raises SphError where the reference throws."""

    def __init__(self, api, params, device=-1):
        self.api = api
        if api.prefix == "sph_":
            self.ctx = api.create(C.byref(params), device)
            if not self.ctx:
                raise SphError(api.create_error().decode())
        else:
            self.ctx = api.create(C.byref(params))
            if not self.ctx:
                raise SphError("oracle: create failed")
        self.time = 0.0
        self.frame = 0

    def close(self):
        if getattr(self, "ctx", None):
            self.api.destroy(self.ctx)
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, rc):
        if rc != 0:
            raise SphError(self.api.last_error(self.ctx).decode())

    # -- setup
    def set_params(self, p):
        self._ck(self.api.set_params(self.ctx, C.byref(p)))

    def params(self):
        p = SphParams()
        self._ck(self.api.get_params(self.ctx, C.byref(p)))
        return p

    def counters(self):
        c = SphCounters()
        self._ck(self.api.get_counters(self.ctx, C.byref(c)))
        return c

    def set_step_size(self, h):
        self._ck(self.api.set_step_size(self.ctx, h))

    def set_check_errors(self, on):
        self._ck(self.api.set_check_errors(self.ctx, int(on)))

    def set_arithmetic(self, mode):
        """SPH_ARITH_EXACT (0) / SPH_ARITH_FAST (1); the oracle only knows the exact arithmetic."""
        if self.api.prefix != "sph_":
            if int(mode) != 0:
                raise SphError("the CPU oracle only implements the exact arithmetic")
            return
        self._ck(self.api.set_arithmetic(self.ctx, int(mode)))

    def add_markers(self, type_code, xyz, vel=None, rho_p_mu=None):
        xyz, pxyz = _fp(xyz)
        n = xyz.shape[0] if xyz.ndim == 2 else xyz.size // 3
        vel, pvel = _fp(vel)
        rpm, prpm = _fp(rho_p_mu)
        self._ck(self.api.add_markers(self.ctx, type_code, n, pxyz, pvel, prpm))
        self._n_added = getattr(self, "_n_added", 0) + n

    def counters_before_init(self):
        """number of markers handed over so far (sph_add_markers calls before sph_initialize)"""
        return getattr(self, "_n_added", 0)

    def set_marker_stress(self, first, tau_diag, tau_offdiag):
        d, pd = _fp(tau_diag)
        o, po = _fp(tau_offdiag)
        n = d.shape[0]
        self._ck(self.api.set_marker_stress(self.ctx, first, n, pd, po))

    def set_marker_consolidation(self, first, pc):
        a, pa = _fp(pc)
        self._ck(self.api.set_marker_consolidation(self.ctx, first, a.size, pa))

    def consolidation(self):
        return self._get(self.api.get_consolidation, self.counters().numAllMarkers, 3)

    def pow_2p55(self, x):
        x = np.ascontiguousarray(x, dtype=np.float64)
        out = np.empty_like(x)
        self._ck(self.api.debug_pow_2p55(self.ctx, x.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p), x.size))
        return out

    def stress(self):
        return self._get(self.api.get_stress, self.counters().numAllMarkers, 6)

    def set_global_ids(self, gid):
        g = np.ascontiguousarray(gid, dtype=np.uint32)
        self._ck(self.api.set_global_ids(self.ctx, g.size, g.ctypes.data_as(C.c_void_p)))

    def global_ids(self):
        n = self.counters().numAllMarkers
        g = np.empty(n, dtype=np.uint32)
        self._ck(self.api.get_global_ids(self.ctx, g.ctypes.data_as(C.c_void_p), n))
        return g

    def slab_rigid_subset(self, gid_first_rigid, idx):
        """Slab ranks: stage only the rigid BCE markers `idx` (indices into the concatenated BCE list of all bodies)."""
        g = np.ascontiguousarray(idx, dtype=np.uint32)
        self._ck(self.api.slab_rigid_subset(self.ctx, int(gid_first_rigid), g.size, g.ctypes.data_as(C.c_void_p)))

    def add_rigid_body(self, local_xyz):
        a, pa = _fp(local_xyz)
        n = a.shape[0] if a.ndim == 2 else a.size // 3
        rc = self.api.add_rigid_body(self.ctx, n, pa)
        if rc < 0:
            raise SphError(self.api.last_error(self.ctx).decode())
        return rc

    def initialize(self, body_states=()):
        arr, n = make_body_states(body_states)
        self._ck(self.api.initialize(self.ctx, n, arr))

    # -- stepping (ChFsiFluidSystem::DoStepDynamics, ChFsiFluidSystem.cpp:58-69)
    def do_step(self, h):
        self._ck(self.api.step(self.ctx, self.time, h))
        self.frame += 1
        self.time += h

    def current_step(self):
        return self.api.current_step(self.ctx)

    def isph_iterations(self):
        """(velocity sweeps, pressure sweeps) of the last I2SPH step."""
        v, p = C.c_int(0), C.c_int(0)
        self._ck(self.api.get_isph_iterations(self.ctx, C.byref(v), C.byref(p)))
        return v.value, p.value

    def set_isph_diagnostics(self, on=True):
        """Evaluate calcNormalizedRho_Gi_fillInMatrixIndices (G tensor, normals) in every PreProcessor call of the I2SPH step."""
        self._ck(self.api.set_isph_diagnostics(self.ctx, 1 if on else 0))

    def isph_g_tensor(self):
        """(G (n, 3, 3), normals (n, 3)) of the last PreProcessor call, cell (sorted) order: pair with sorted_index()."""
        n = self.counters().numExtendedParticles
        G = np.empty((n, 3, 3), dtype=np.float32)
        nr = np.empty((n, 3), dtype=np.float32)
        self._ck(self.api.get_isph_g_tensor(self.ctx, G.ctypes.data_as(C.c_void_p), nr.ctypes.data_as(C.c_void_p), n))
        return G, nr

    def synchronize(self):
        if self.api.prefix == "sph_":
            self._ck(self.api.synchronize(self.ctx))

    def load_body_states(self, body_states):
        arr, n = make_body_states(body_states)
        self._ck(self.api.load_body_states(self.ctx, n, arr))

    def compute_body_wrenches(self):
        self._ck(self.api.compute_body_wrenches(self.ctx))

    def get_body_wrenches(self, n):
        arr = (SphWrench * max(n, 1))()
        self._ck(self.api.get_body_wrenches(self.ctx, n, arr))
        return [(np.array(arr[k].force[:]), np.array(arr[k].torque[:])) for k in range(n)]

    def update_body_markers(self):
        self._ck(self.api.update_body_markers(self.ctx))

    # -- accessors
    def _get(self, fn, n, width):
        out = np.empty((n, width), dtype=np.float32)
        self._ck(fn(self.ctx, out.ctypes.data_as(C.c_void_p), n))
        return out

    def positions(self):
        return self._get(self.api.get_positions, self.counters().numAllMarkers, 3)

    def velocities(self):
        return self._get(self.api.get_velocities, self.counters().numAllMarkers, 3)

    def properties(self):
        return self._get(self.api.get_properties, self.counters().numAllMarkers, 3)

    def accelerations(self):
        return self._get(self.api.get_accelerations, self.counters().numFluidMarkers, 3)

    def derivatives(self):
        return self._get(self.api.get_derivatives, self.counters().numAllMarkers, 4)

    def rho_pres_mu_type(self):
        return self._get(self.api.get_rho_pres_mu_type, self.counters().numAllMarkers, 4)

    def neighbor_csr(self):
        rows, nnz = C.c_size_t(), C.c_size_t()
        self._ck(self.api.get_neighbor_csr(self.ctx, None, 0, None, 0, C.byref(rows), C.byref(nnz)))
        offs = np.empty(rows.value + 1, dtype=np.uint32)
        idx = np.empty(nnz.value, dtype=np.uint32)
        self._ck(self.api.get_neighbor_csr(self.ctx, offs.ctypes.data_as(C.c_void_p), offs.size, idx.ctypes.data_as(C.c_void_p), idx.size,
                                           C.byref(rows), C.byref(nnz)))
        return offs, idx

    def sorted_index(self):
        n = self.counters().numExtendedParticles
        s2o = np.empty(n, dtype=np.uint32)
        h = np.empty(n, dtype=np.uint32)
        self._ck(self.api.get_sorted_index(self.ctx, s2o.ctypes.data_as(C.c_void_p), h.ctypes.data_as(C.c_void_p), n))
        return s2o, h

    def sorted_state(self):
        n = self.counters().numExtendedParticles
        pos = np.empty((n, 4), dtype=np.float32)
        vel = np.empty((n, 3), dtype=np.float32)
        rpm = np.empty((n, 4), dtype=np.float32)
        self._ck(self.api.get_sorted_state(self.ctx, pos.ctypes.data_as(C.c_void_p), vel.ctypes.data_as(C.c_void_p), rpm.ctypes.data_as(C.c_void_p), n))
        return pos, vel, rpm
