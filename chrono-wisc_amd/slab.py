"""x-slab decomposition of the SPH step over several ranks (one process per GPU).

The reference solver is single-device; this is the multi-GPU extension BASELINE.json asks for.  The library only
classifies / packs / unpacks (include/chsph_mi.h, "multi-GPU slabs"); this module sequences one step of one rank and
moves the buffers through a transport:

    if search due:  classify -> exchange emigrants -> apply_migration -> exchange ghosts -> apply_ghosts
                    (from the second step of an explicit scheme with a search every step: refresh_pack -> ONE exchange ->
                    refresh_apply, the library keeping its state resident in cell order; include/chsph_mi.h)
    step_begin                       (activity, on the refreshed local marker layout)
    if search due:  step_search
    for each stage: step_stage -> halo exchange of the copy the stage advanced
    step_end                         (copy back, frame++)

Transports:
  DistTransport      torch.distributed point-to-point between slab neighbours: backend "nccl" (= RCCL over xGMI on
                     ROCm) sends straight from the library's device buffers; backend "gloo" stages through host memory
                     (CPU ranks, or several ranks sharing one GPU in tests).
  ThreadTransport    all ranks inside one process, one thread per rank (tests on a single host / a single GPU).
Because cells are hashed on the global grid and ties inside a cell are broken by global marker index, every owned
marker sees the same neighbours in the same order as in a single-device run: results are bit-identical to it.
"""
import ctypes as C
import os
import queue
import threading

import numpy as np

from . import capi

LEFT, RIGHT = capi.SPH_SLAB_LEFT, capi.SPH_SLAB_RIGHT
SEND, RECV = capi.SPH_SLAB_SEND, capi.SPH_SLAB_RECV
MIGRATE, GHOST, HALO = capi.SPH_SLAB_MIGRATE, capi.SPH_SLAB_GHOST, capi.SPH_SLAB_HALO
# CRM (elastic_SPH): the stress state travels in a companion buffer with the same record count and order
CRM_COMPANION = {MIGRATE: capi.SPH_SLAB_MIGRATE_CRM, GHOST: capi.SPH_SLAB_GHOST_CRM, HALO: capi.SPH_SLAB_HALO_CRM,
                 capi.SPH_SLAB_REFRESH: capi.SPH_SLAB_REFRESH_CRM}

_hip = None


def _hiplib():
    global _hip
    if _hip is None:
        # the HIP runtime this process already uses (the solver library's: capi binds it before the library loads).  Asking the loader for
        # "libamdhip64.so" by name would map the ROCm installation's copy as a SECOND runtime whenever the mapped one came from torch's wheel
        # under its full path (tools started without `import torch`; INTEGRATION.md, "One ROCm runtime per process")
        mapped = capi.mapped_runtimes()["libamdhip64.so"]
        _hip = C.CDLL(mapped[0] if mapped else "libamdhip64.so")
        _hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        _hip.hipMemcpy.restype = C.c_int
        _hip.hipStreamSynchronize.argtypes = [C.c_void_p]
        _hip.hipStreamSynchronize.restype = C.c_int
    return _hip


def _to_host(ptr, nbytes, on_device):
    """bytes object holding a copy of the buffer"""
    if nbytes == 0:
        return b""
    buf = (C.c_char * nbytes)()
    if on_device:
        if _hiplib().hipMemcpy(buf, ptr, nbytes, 2) != 0:  # hipMemcpyDeviceToHost
            raise capi.SphError("hipMemcpy D2H failed")
    else:
        C.memmove(buf, ptr, nbytes)
    return bytes(buf)


def _from_host(data, ptr, nbytes, on_device):
    if nbytes == 0:
        return
    assert len(data) == nbytes, "halo size mismatch: got %d bytes, expected %d" % (len(data), nbytes)
    if on_device:
        if _hiplib().hipMemcpy(ptr, data, nbytes, 1) != 0:  # hipMemcpyHostToDevice
            raise capi.SphError("hipMemcpy H2D failed")
    else:
        C.memmove(ptr, data, nbytes)


# ---- slab cuts -----------------------------------------------------------------------------------------------
def cell_column(x, params):
    """Global cell column of a marker, exactly like calcGridPos + the non-periodic clamp of calcGridHash
    (sph/physics/SphGeneral.cuh:325-350), in float32."""
    gx = np.floor((np.asarray(x, dtype=np.float32) - np.float32(params.worldOrigin[0])) / np.float32(params.cellSize[0])).astype(np.int64)
    return np.clip(gx, 0, params.gridSize[0] - 1)


# Relative cost of a step per marker, fitted to the per-rank step times of the C3 dam break cut into 8 slabs, one rank on the GPU at a time and two
# different partitions (tools/slab_rank_time.py, round 4; NOTEBOOK.md 4): 0.415 ns per owned fluid marker, 0.26 ns per ghost row, 0.27 ms per rank.
# A wall BCE row carries no force evaluation and no update, but it is sorted, searched and extrapolated (Adami, four lanes per row), its waves pass
# through the force kernel, and its share of the step came out at that of a fluid row.
WALL_MARKER_WEIGHT = 1.0
GHOST_ROW_WEIGHT = 0.6
# ... and per CELL of a rank's column window (zeroing and scanning cellStart): what the last rank of a dam break pays for its hundreds of dry
# columns (C3 / 8: 425 columns of 80 000 cells, ~0.1 ms of its 1.4 ms step when the cuts ignored it), in fluid rows per cell.
CELL_WEIGHT = 0.006


_HISTOGRAM = object()  # balanced_cuts(hist, ..., weights=_HISTOGRAM): `cx` already is the weighted column histogram


def balanced_cuts(cx, grid_x, nranks, min_width=4, weights=None, ghost_weight=GHOST_ROW_WEIGHT, halo=1, column_cost=0.0):
    """Cut positions [c0=0, c1, ..., c_n=grid_x] minimising the largest per-rank cost, where a rank pays for the (weighted)
    markers of its columns plus `ghost_weight` times those of the `halo` columns beyond each of its cuts (ghost rows are sorted,
    searched and evaluated like owned ones, only not advanced; the resident slab path keeps one ghost column except near walls),
    plus `column_cost` per cell column of its window, empty or not (CELL_WEIGHT x cells per column).  Exact min-max partition by dynamic
    programming over the column histogram (grid_x is a few hundred); deterministic, so every rank computes the same cuts."""
    hist = np.asarray(cx, dtype=np.float64) if weights is _HISTOGRAM else np.bincount(cx, weights=weights, minlength=grid_x).astype(np.float64)
    if grid_x < min_width * nranks:
        raise capi.SphError("domain too narrow in x for %d slabs of at least %d cell columns" % (nranks, min_width))
    W = np.concatenate([[0.0], np.cumsum(hist)])                      # W[c] = weight of columns [0, c)
    c = np.arange(grid_x + 1)
    ghost_l = ghost_weight * (W[c] - W[np.maximum(c - halo, 0)])       # ghosts a rank starting at column c imports from the left
    ghost_r = ghost_weight * (W[np.minimum(c + halo, grid_x)] - W[c])  # ... a rank ending at column c imports from the right
    ghost_l[0] = 0.0
    ghost_r[grid_x] = 0.0
    INF = float("inf")
    best = np.full((nranks + 1, grid_x + 1), INF)
    arg = np.zeros((nranks + 1, grid_x + 1), dtype=np.int64)
    best[0, 0] = 0.0
    for k in range(1, nranks + 1):
        for b in range(k * min_width, grid_x - (nranks - k) * min_width + 1):
            if k == nranks and b != grid_x:
                continue
            a = np.arange((k - 1) * min_width, b - min_width + 1)
            if len(a) == 0:
                continue
            cost = np.maximum(best[k - 1, a], (W[b] - W[a]) + ghost_l[a] + ghost_r[b] + column_cost * (b - a + 4))
            j = int(np.argmin(cost))
            best[k, b], arg[k, b] = cost[j], a[j]
    if not np.isfinite(best[nranks, grid_x]):
        raise capi.SphError("domain too narrow in x for %d slabs of at least %d cell columns" % (nranks, min_width))
    cuts = [grid_x]
    for k in range(nranks, 0, -1):
        cuts.append(int(arg[k, cuts[-1]]))
    return cuts[::-1]


# ---- transports ----------------------------------------------------------------------------------------------
class ThreadTransport:
    """Mailboxes between ranks living in one process (one thread per rank)."""

    def __init__(self, nranks, device_direct=False):
        """device_direct: messages between device buffers are copied device-to-device by the receiver (the ranks share one GPU and
        one address space) instead of being staged through host memory -- what a peer-to-peer transport costs, minus the link."""
        self.nranks = nranks
        self.device_direct = device_direct
        self.box = {(s, d): queue.Queue() for s in range(nranks) for d in range(nranks)}
        self.ack = {(s, d): queue.Queue() for s in range(nranks) for d in range(nranks)}
        self.failed = threading.Event()  # a rank raised: the others stop waiting for its messages

    def endpoint(self, rank):
        return _ThreadEndpoint(self, rank)

    def get(self, src, dst, timeout=600.0):
        waited = 0.0
        while True:
            try:
                return self.box[(src, dst)].get(timeout=0.2)
            except queue.Empty:
                waited += 0.2
                if self.failed.is_set():
                    raise capi.SphError("slab rank %d: a peer rank failed" % dst)
                if waited >= timeout:
                    raise


class _ThreadEndpoint:
    def __init__(self, tr, rank):
        self.tr, self.rank, self.nranks = tr, rank, tr.nranks

    def exchange(self, sends, recvs):
        """sends: {peer: (ptr, nbytes, on_device)}; recvs: {peer: (ptr, nbytes, on_device)}"""
        direct = self.tr.device_direct and all(dev for (_p, _n, dev) in list(sends.values()) + list(recvs.values()))
        if direct:
            for peer, (ptr, nb, dev) in sends.items():
                self.tr.box[(self.rank, peer)].put((ptr, nb))
            for peer, (ptr, nb, dev) in recvs.items():
                src, n_src = self.tr.get(peer, self.rank)
                assert n_src == nb, "halo size mismatch: got %d bytes, expected %d" % (n_src, nb)
                # a device-to-device hipMemcpy is not synchronous with the host: drain the null stream before anybody touches either buffer
                if nb and (_hiplib().hipMemcpy(C.c_void_p(ptr), C.c_void_p(src), nb, 3) != 0 or _hiplib().hipStreamSynchronize(None) != 0):
                    raise capi.SphError("hipMemcpy D2D failed")
                self.tr.ack[(self.rank, peer)].put(1)
            for peer in sends:  # the send buffer is free again only when the receiver has copied it
                q = self.tr.ack[(peer, self.rank)]
                while True:
                    try:
                        q.get(timeout=0.2)
                        break
                    except queue.Empty:
                        if self.tr.failed.is_set():
                            raise capi.SphError("slab rank %d: a peer rank failed" % self.rank)
            return
        for peer, (ptr, nb, dev) in sends.items():
            self.tr.box[(self.rank, peer)].put(_to_host(ptr, nb, dev))
        for peer, (ptr, nb, dev) in recvs.items():
            _from_host(self.tr.get(peer, self.rank), ptr, nb, dev)

    def exchange_ints(self, sends, peers):
        for peer, v in sends.items():
            self.tr.box[(self.rank, peer)].put(int(v))
        return {peer: self.tr.get(peer, self.rank) for peer in peers}

    def allreduce_min_float(self, v):
        for peer in range(self.nranks):
            if peer != self.rank:
                self.tr.box[(self.rank, peer)].put(float(v))
        for peer in range(self.nranks):
            if peer != self.rank:
                v = min(v, self.tr.get(peer, self.rank))
        return v

    def allreduce_sum_array(self, a):
        """Elementwise sum of a float64 array over all ranks (summed in rank order on every rank: identical results)."""
        a = np.ascontiguousarray(a, dtype=np.float64)
        for peer in range(self.nranks):
            if peer != self.rank:
                self.tr.box[(self.rank, peer)].put(a)
        parts = {self.rank: a}
        for peer in range(self.nranks):
            if peer != self.rank:
                parts[peer] = self.tr.get(peer, self.rank)
        out = np.zeros_like(a)
        for r in range(self.nranks):
            out += parts[r]
        return out

    def allreduce_max_words(self, ptr, nwords, on_device):
        """In-place elementwise max of `nwords` non-negative 32-bit integers over all ranks."""
        if self.nranks == 1 or nwords == 0:
            return
        mine = np.frombuffer(_to_host(ptr, 4 * nwords, on_device), dtype=np.int32).copy()
        for peer in range(self.nranks):
            if peer != self.rank:
                self.tr.box[(self.rank, peer)].put(mine)
        out = mine.copy()
        for peer in range(self.nranks):
            if peer != self.rank:
                np.maximum(out, self.tr.get(peer, self.rank), out=out)
        _from_host(out.tobytes(), ptr, 4 * nwords, on_device)


    def allreduce_sum_words(self, ptr, nwords, on_device):
        """In-place elementwise sum of `nwords` 32-bit integers over all ranks (used on tables in which one rank holds the
        bits of an entry and all others zero, so the integer sum reproduces them)."""
        if self.nranks == 1 or nwords == 0:
            return
        mine = np.frombuffer(_to_host(ptr, 4 * nwords, on_device), dtype=np.int32).copy()
        for peer in range(self.nranks):
            if peer != self.rank:
                self.tr.box[(self.rank, peer)].put(mine)
        out = mine.copy()
        for peer in range(self.nranks):
            if peer != self.rank:
                out += self.tr.get(peer, self.rank)
        _from_host(out.tobytes(), ptr, 4 * nwords, on_device)


class InprocGroup:
    """sph_slab_group: the ranks are threads of this process and exchange through the library's in-process transport
    (sph_slab_transport_inproc: the receiver copies on its own stream behind the sender's event -- stream-ordered, no host wait for a
    GPU).  Only sph_slab_step runs over it (SlabRank.step_native); the Python-driven step needs a ThreadTransport."""

    def __init__(self, api, nranks):
        self.api, self.nranks = api, nranks
        self.handle = api.slab_group_create(nranks)
        if not self.handle:
            raise capi.SphError("sph_slab_group_create(%d) failed (1..16 ranks)" % nranks)

    def endpoint(self, rank):
        return _InprocEndpoint(self, rank)

    def close(self):
        if self.handle and self.api.slab_group_destroy(self.handle) == 0:
            self.handle = None


class _InprocEndpoint:
    def __init__(self, group, rank):
        self.inproc_group, self.rank = group, rank


class _DevView:
    """Zero-copy view of a device buffer for torch.as_tensor (CUDA array interface)."""

    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (max(nbytes, 1),), "typestr": "|u1", "data": (int(ptr), False), "version": 2}


class HostMailbox:
    """Same-node side channel for the few host scalars of a step (record counts, the variable time step): one single-slot
    mailbox per ordered rank pair in a POSIX shared-memory segment, sequence/acknowledge handshake, busy-wait.  A count exchange
    through the collective library costs ~80 us per step on the one-GPU box (two tiny sends/receives plus the device round
    trip); this costs a few microseconds, and the GPU idles while the host waits for exactly these numbers.  Ranks on several
    nodes keep the collective path (DistTransport checks the host names)."""
    WORDS = 6  # seq, ack, 4 payload words (int64 / float64 bit patterns)

    def __init__(self, dist, rank, nranks):
        from multiprocessing import resource_tracker, shared_memory
        self.rank, self.nranks = rank, nranks
        nbytes = nranks * nranks * self.WORDS * 8
        name = [None]
        if rank == 0:
            try:
                self.shm = shared_memory.SharedMemory(create=True, size=nbytes)
                self.shm.buf[:nbytes] = bytes(nbytes)
                name[0] = self.shm.name
            except OSError:
                name[0] = None  # no shared memory on this host: every rank falls back together
        dist.broadcast_object_list(name, src=0)
        if name[0] is None:
            raise OSError("no POSIX shared memory")
        if rank != 0:
            self.shm = shared_memory.SharedMemory(name=name[0])
            try:  # the creator unlinks; without this every attaching process tries to as well (and warns) at exit
                resource_tracker.unregister(self.shm._name, "shared_memory")
            except Exception:  # noqa: BLE001
                pass
        self.a = np.ndarray((nranks, nranks, self.WORDS), dtype=np.int64, buffer=self.shm.buf)
        self.f = self.a.view(np.float64)
        self.sent = np.zeros(nranks, dtype=np.int64)
        self.got = np.zeros(nranks, dtype=np.int64)
        dist.barrier()

    def _wait(self, cond, what):
        import time
        t0, spins = None, 0
        while not cond():
            spins += 1
            if spins & 0xFFF == 0:
                now = time.perf_counter()
                t0 = t0 or now
                if now - t0 > 600.0:
                    raise capi.SphError("slab rank %d: timed out waiting for %s" % (self.rank, what))

    def send(self, dst, ints=(), floats=()):
        slot, k = self.a[self.rank, dst], self.sent[dst]
        self._wait(lambda: slot[1] == k, "rank %d to take the previous message" % dst)
        for i, v in enumerate(ints):
            slot[2 + i] = int(v)
        for i, v in enumerate(floats):
            self.f[self.rank, dst, 2 + len(ints) + i] = float(v)
        slot[0] = k + 1  # x86: stores are not reordered with older stores
        self.sent[dst] = k + 1

    def recv(self, src, nints=0, nfloats=0):
        slot, k = self.a[src, self.rank], self.got[src]
        self._wait(lambda: slot[0] == k + 1, "a message from rank %d" % src)
        ints = [int(slot[2 + i]) for i in range(nints)]
        floats = [float(self.f[src, self.rank, 2 + nints + i]) for i in range(nfloats)]
        slot[1] = k + 1
        self.got[src] = k + 1
        return ints, floats

    def close(self):
        try:
            self.a = self.f = None
            self.shm.close()
            if self.rank == 0:
                self.shm.unlink()
        except Exception:  # noqa: BLE001
            pass


class DistTransport:
    """torch.distributed point-to-point between slab neighbours (one rank per process)."""

    def __init__(self, dist, device=None):
        import socket
        import torch
        capi.assert_single_runtime("slab.DistTransport")  # device pointers of the solver are handed to torch below
        self.torch = torch
        self.dist = dist
        self.rank = dist.get_rank()
        self.nranks = dist.get_world_size()
        self.backend = dist.get_backend()
        self.device = device
        self.shared_stream = False  # set by SlabRank when the solver context runs on torch's current stream
        self._views = {}
        self.mailbox = None
        if self.nranks > 1 and os.environ.get("CHSPH_NO_HOST_MAILBOX") is None:
            hosts = [None] * self.nranks
            dist.all_gather_object(hosts, socket.gethostname())
            if len(set(hosts)) == 1:  # one node: host scalars travel through shared memory
                import atexit
                try:
                    self.mailbox = HostMailbox(dist, self.rank, self.nranks)
                    atexit.register(self.mailbox.close)
                except OSError:
                    self.mailbox = None

    def stream_handle(self):
        """Raw HIP stream torch enqueues on for this device (None on CPU ranks): handing it to the solver context orders
        pack -> send/recv -> unpack by the stream, without host synchronisation."""
        if self.device is None:
            return None
        return int(self.torch.cuda.current_stream(self.device).cuda_stream)

    def endpoint(self, rank):
        assert rank == self.rank
        return self

    def _tensor(self, ptr, nbytes, on_device, for_recv):
        """(tensor to hand to torch.distributed, device view to copy a staged receive into or None)"""
        torch = self.torch
        if on_device:
            # zero-copy view of the library buffer; views are cached per pointer (the buffers live across steps, and building a
            # tensor from the CUDA array interface costs more than the send it feeds)
            hit = self._views.get(ptr)
            if hit is None or hit.numel() < nbytes:
                if len(self._views) > 256:
                    self._views.clear()
                hit = torch.as_tensor(_DevView(ptr, max(nbytes, 1)), device=self.device)
                self._views[ptr] = hit
            dv = hit[:nbytes]
            if self.backend == "nccl":
                return dv, None
            # gloo cannot move device memory: stage through the host
            return (torch.empty(nbytes, dtype=torch.uint8) if for_recv else dv.cpu()), (dv if for_recv else None)
        arr = np.ctypeslib.as_array((C.c_uint8 * max(nbytes, 1)).from_address(ptr))[:nbytes]
        return torch.from_numpy(arr), None

    def exchange(self, sends, recvs):
        dist = self.dist
        ops, staged, keep = [], [], []
        for peer, (ptr, nb, dev) in recvs.items():
            if nb == 0:
                continue
            t, dv = self._tensor(ptr, nb, dev, True)
            ops.append(dist.P2POp(dist.irecv, t, peer))
            if dv is not None:
                staged.append((t, dv))
        for peer, (ptr, nb, dev) in sends.items():
            if nb == 0:
                continue
            t, _ = self._tensor(ptr, nb, dev, False)
            keep.append(t)
            ops.append(dist.P2POp(dist.isend, t, peer))
        if ops:
            for w in dist.batch_isend_irecv(ops):
                w.wait()
        for t, dv in staged:
            dv.copy_(t)
        if self.device is not None and not self.shared_stream:
            # the library's kernels run on its own stream: everything torch enqueued must have landed first
            self.torch.cuda.current_stream(self.device).synchronize()

    def allreduce_sum_array(self, a):
        """Elementwise sum of a float64 array over all ranks.  Gathered and summed in rank order on every rank, so that all
        ranks hold bit-identical sums (they derive the new slab cuts from them)."""
        a = np.ascontiguousarray(a, dtype=np.float64)
        parts = [None] * self.nranks
        self.dist.all_gather_object(parts, a)
        out = np.zeros_like(a)
        for r in range(self.nranks):
            out += parts[r]
        return out

    def allreduce_min_float(self, v):
        if self.mailbox is not None:
            for peer in range(self.nranks):
                if peer != self.rank:
                    self.mailbox.send(peer, floats=(v,))
            for peer in range(self.nranks):
                if peer != self.rank:
                    v = min(v, self.mailbox.recv(peer, nfloats=1)[1][0])
            return float(v)
        t = self.torch.tensor([float(v)], dtype=self.torch.float64, device=self.device if self.backend == "nccl" else "cpu")
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MIN)
        return float(t.item())

    def allreduce_max_words(self, ptr, nwords, on_device):
        """In-place elementwise max of `nwords` non-negative 32-bit integers over all ranks (RCCL all-reduce on the
        library's device buffer; staged through the host for gloo)."""
        torch, dist = self.torch, self.dist
        if self.nranks == 1 or nwords == 0:
            return
        if on_device and self.backend == "nccl":
            t = torch.as_tensor(_DevView(ptr, 4 * nwords), device=self.device)[:4 * nwords].view(torch.int32)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            if not self.shared_stream:
                self.torch.cuda.current_stream(self.device).synchronize()
            return
        t = torch.from_numpy(np.frombuffer(_to_host(ptr, 4 * nwords, on_device), dtype=np.int32).copy())
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        _from_host(t.numpy().tobytes(), ptr, 4 * nwords, on_device)

    def allreduce_sum_words(self, ptr, nwords, on_device):
        """In-place elementwise sum of `nwords` 32-bit integers over all ranks (RCCL all-reduce on the library's device
        buffer; staged through the host for gloo).  See _ThreadEndpoint.allreduce_sum_words."""
        torch, dist = self.torch, self.dist
        if self.nranks == 1 or nwords == 0:
            return
        if on_device and self.backend == "nccl":
            t = torch.as_tensor(_DevView(ptr, 4 * nwords), device=self.device)[:4 * nwords].view(torch.int32)
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
            if not self.shared_stream:
                self.torch.cuda.current_stream(self.device).synchronize()
            return
        t = torch.from_numpy(np.frombuffer(_to_host(ptr, 4 * nwords, on_device), dtype=np.int32).copy())
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        _from_host(t.numpy().tobytes(), ptr, 4 * nwords, on_device)

    def exchange_ints(self, sends, peers):
        if self.mailbox is not None:
            for peer, v in sends.items():
                self.mailbox.send(peer, ints=(v,))
            return {peer: self.mailbox.recv(peer, nints=1)[0][0] for peer in peers}
        torch, dist = self.torch, self.dist
        dev = self.device if self.backend == "nccl" else "cpu"
        ops, rt, keep = [], {}, []
        for peer in peers:
            rt[peer] = torch.zeros(1, dtype=torch.int64, device=dev)
            ops.append(dist.P2POp(dist.irecv, rt[peer], peer))
        for peer, v in sends.items():
            t = torch.tensor([int(v)], dtype=torch.int64, device=dev)
            keep.append(t)
            ops.append(dist.P2POp(dist.isend, t, peer))
        if ops:
            for w in dist.batch_isend_irecv(ops):
                w.wait()
        return {peer: int(t.item()) for peer, t in rt.items()}


# ---- one rank ------------------------------------------------------------------------------------------------
class SlabRank:
    """Drives the step phases of one rank's solver context and its exchanges with the two neighbour ranks."""

    def __init__(self, sysS, endpoint, rank, nranks, cuts):
        self.sys = sysS
        self.api = sysS.api
        self.ctx = sysS.solver.ctx
        self.ep = endpoint
        self.rank, self.nranks = rank, nranks
        self.cuts = list(cuts)
        cfg = capi.SphSlabConfig(rank, nranks, cuts[rank], cuts[rank + 1], 2)
        sysS.solver._ck(self.api.slab_configure(self.ctx, C.byref(cfg)))
        # slabs along a periodic x axis form a ring: the first and the last rank exchange across the seam (three ranks or more, checked by slab_configure)
        self.ring = nranks > 1 and bool(sysS.GetParams().x_periodic)
        self.left = (rank - 1) % nranks if self.ring else (rank - 1 if rank > 0 else None)
        self.right = (rank + 1) % nranks if self.ring else (rank + 1 if rank < nranks - 1 else None)
        self.time = 0.0
        self.ps_freq = int(sysS.GetParams().num_proximity_search_steps)
        self.crm = bool(sysS.GetParams().elastic_SPH)
        self._field_bufs = {}  # I2SPH: (send, recv) buffer handles per field, valid for the current ghost set
        self._reduce_bufs = {}
        sysS._slab_rank = self  # OnExchangeSolidForces of a slab rank sums the per-marker wrench table over the ranks
        # device ranks under torch.distributed: run the solver on torch's stream so exchanges need no host round trips
        self.shared_stream = False
        handle = endpoint.stream_handle() if hasattr(endpoint, "stream_handle") else None
        if handle is not None and self.api.prefix == "sph_" and os.environ.get("CHSPH_SLAB_OWN_STREAM") is None:
            self._ck(self.api.set_stream(self.ctx, C.c_void_p(handle)))
            self.shared_stream = True
            endpoint.shared_stream = True

    def _ck(self, rc):
        self.sys.solver._ck(rc)

    def _buf(self, direction, kind, side, nrecords=0):
        ptr, nb, dev = C.c_void_p(), C.c_size_t(), C.c_int()
        self._ck(self.api.slab_buffer(self.ctx, direction, kind, side, nrecords, C.byref(ptr), C.byref(nb), C.byref(dev)))
        return (ptr.value or 0, nb.value, dev.value)

    def _peers(self):
        return [(LEFT, self.left), (RIGHT, self.right)]

    def _exchange_records(self, kind, send_counts):
        # counts first (how many records each neighbour sends me), then the payloads
        sends = {peer: send_counts[side] for side, peer in self._peers() if peer is not None}
        got = self.ep.exchange_ints(sends, [p for _s, p in self._peers() if p is not None])
        recv_counts = [got.get(self.left, 0) if self.left is not None else 0, got.get(self.right, 0) if self.right is not None else 0]
        s, r = {}, {}
        for side, peer in self._peers():
            if peer is None:
                continue
            s[peer] = self._buf(SEND, kind, side)
            r[peer] = self._buf(RECV, kind, side, recv_counts[side])
        self.ep.exchange(s, r)
        if self.crm and kind in CRM_COMPANION:
            s, r = {}, {}
            for side, peer in self._peers():
                if peer is not None:
                    s[peer] = self._buf(SEND, CRM_COMPANION[kind], side)
                    r[peer] = self._buf(RECV, CRM_COMPANION[kind], side, recv_counts[side])
            self.ep.exchange(s, r)
        return recv_counts

    def _refresh_ghost_set(self):
        u2 = C.c_uint32 * 2
        self._field_bufs = {}
        if self.api.slab_one_round(self.ctx):
            # one exchange round: emigrants travel with the ghost layer, the receiver sorts them out by cell column
            counts = u2()
            self._ck(self.api.slab_refresh_pack(self.ctx, counts))
            recv = self._exchange_records(capi.SPH_SLAB_REFRESH, [counts[0], counts[1]])
            self._ck(self.api.slab_refresh_apply(self.ctx, u2(*recv)))
            return
        counts = u2()
        self._ck(self.api.slab_classify(self.ctx, counts))
        recv = self._exchange_records(MIGRATE, [counts[0], counts[1]])
        gcounts = u2()
        self._ck(self.api.slab_apply_migration(self.ctx, u2(*recv), gcounts))
        grecv = self._exchange_records(GHOST, [gcounts[0], gcounts[1]])
        self._ck(self.api.slab_apply_ghosts(self.ctx, u2(*grecv)))

    def _halo(self, y):
        s, r = {}, {}
        for side, peer in self._peers():
            if peer is None:
                continue
            self._ck(self.api.slab_halo_pack(self.ctx, y, side))
            s[peer] = self._buf(SEND, HALO, side)
            r[peer] = self._buf(RECV, HALO, side)
        self.ep.exchange(s, r)
        if self.crm:
            s, r = {}, {}
            for side, peer in self._peers():
                if peer is not None:
                    s[peer] = self._buf(SEND, CRM_COMPANION[HALO], side)
                    r[peer] = self._buf(RECV, CRM_COMPANION[HALO], side)
            self.ep.exchange(s, r)
        for side, peer in self._peers():
            if peer is not None:
                self._ck(self.api.slab_halo_unpack(self.ctx, y, side))

    # -- I2SPH: the phases of include/chsph_mi.h with the ghost exchanges between them
    def _field(self, field, t=0):
        # one pack call and one unpack call for both sides; the buffer handles of a field are fetched once per ghost set (the
        # record counts are those of the ghost layer, so the library does not move the buffers between two searches)
        self._ck(self.api.slab_field_pack(self.ctx, field, t, -1))
        bufs = self._field_bufs.get(field)
        if bufs is None:
            s, r = {}, {}
            for side, peer in self._peers():
                if peer is not None:
                    s[peer] = self._buf(SEND, field, side)
                    r[peer] = self._buf(RECV, field, side)
            bufs = self._field_bufs[field] = (s, r)
        self.ep.exchange(*bufs)
        self._ck(self.api.slab_field_unpack(self.ctx, field, t, -1))

    def _allreduce(self, what, t=0):
        ptr, nw, dev = C.c_void_p(), C.c_size_t(), C.c_int()
        self._ck(self.api.isph_reduce_buffer(self.ctx, what, t, C.byref(ptr), C.byref(nw), C.byref(dev)))
        if self.api.prefix == "sph_" and not self.shared_stream:
            self._ck(self.api.synchronize(self.ctx))  # the record is written on the library's stream
        if what == capi.SPH_ISPH_REDUCE_MINPRESSURE:  # one float, ordinary comparison
            v = float(np.frombuffer(_to_host(ptr.value or 0, 4, dev.value), dtype=np.float32)[0])
            _from_host(np.array([self.ep.allreduce_min_float(v)], dtype=np.float32).tobytes(), ptr.value or 0, 4, dev.value)
            return
        self.ep.allreduce_max_words(ptr.value or 0, nw.value, dev.value)

    def _solve(self, which, field):
        api, ctx = self.api, self.ctx
        max_iter = max(int(self.sys.GetParams().LinearSolver_Max_Iter), 0)
        for t in range(max_iter):
            self._ck(api.isph_sweep(ctx, which, t))
            self._field(field, t)                      # ghosts of the iterate just written
            self._allreduce(capi.SPH_ISPH_REDUCE_RESIDUAL, t)  # every rank's next gate sees the global residual maximum
            if t % 16 == 15 and t + 1 < max_iter:
                ran = api.isph_sweep_ran(ctx, which, t)
                if ran < 0:
                    self._ck(1)
                if not ran:
                    break

    def _step_isph(self, h):
        api, ctx = self.api, self.ctx
        search = api.step_needs_search(ctx)
        if search:
            self._refresh_ghost_set()
        self._ck(api.step_begin(ctx, self.time))
        if search:
            self._ck(api.step_search(ctx))
        self._ck(api.isph_phase(ctx, capi.SPH_ISPH_DENSITY1, h))
        self._allreduce(capi.SPH_ISPH_REDUCE_MAXVEL)
        self._field(capi.SPH_FIELD_RHO_VOL)
        self._ck(api.isph_phase(ctx, capi.SPH_ISPH_VSYSTEM, h))
        self._solve(0, capi.SPH_FIELD_VSTAR)
        self._ck(api.isph_phase(ctx, capi.SPH_ISPH_PSYSTEM, h))
        self._solve(1, capi.SPH_FIELD_Q)
        self._ck(api.isph_phase(ctx, capi.SPH_ISPH_CORRECT, h))
        self._halo(1)
        self._ck(api.isph_phase(ctx, capi.SPH_ISPH_DENSITY2, h))
        self._field(capi.SPH_FIELD_RHO_VOL)
        self._ck(api.isph_phase(ctx, capi.SPH_ISPH_SHIFT, h))
        self._halo(1)
        p = self.sys.GetParams()
        if int(p.Conservative_Form) and int(p.ClampPressure):
            self._ck(api.isph_phase(ctx, capi.SPH_ISPH_CLAMP, h))
            self._allreduce(capi.SPH_ISPH_REDUCE_MINPRESSURE)
        self._ck(api.isph_phase(ctx, capi.SPH_ISPH_FINISH, h))
        self._ck(api.step_end(ctx))
        self.time += h
        self.sys.m_time = self.time
        self.sys.m_frame += 1

    def step(self, h):
        api, ctx = self.api, self.ctx
        if int(self.sys.GetParams().integration_scheme) == capi.SPH_SCHEME_IMPLICIT_SPH:
            return self._step_isph(h)
        search = api.step_needs_search(ctx)
        if search:
            self._refresh_ghost_set()  # changes the local marker layout: must precede the activity pass
        self._ck(api.step_begin(ctx, self.time))
        if search:
            self._ck(api.step_search(ctx))
        ns = api.step_num_stages(ctx)
        for st in range(ns):
            self._ck(api.step_stage(ctx, st, h))
            # ghosts of the advanced copy must match their owners before anyone evaluates a force on it: always
            # after an intermediate stage; after the final one only if the next step re-uses this ghost set
            if st < ns - 1 or self.ps_freq > 1:
                self._halo(api.step_stage_target(ctx, st))
        self._ck(api.step_end(ctx))
        self.time += h
        self.sys.m_time = self.time
        self.sys.m_frame += 1

    # -- the same step with the orchestration inside the library (sph_slab_step): one call per step instead of ~20
    def native_transport(self):
        """sph_slab_transport for this rank: the library's built-in RCCL transport when the endpoint is a torch.distributed `nccl`
        group on device memory (communicator created from an id that rank 0 obtains and the group broadcasts), otherwise
        callbacks into this endpoint's exchange / exchange_ints (in-process threads, gloo staging: host-side copies)."""
        if getattr(self, "_native", None) is not None:
            return self._native
        api, ep = self.api, self.ep
        if hasattr(ep, "inproc_group"):
            t = api.slab_transport_inproc(self.ctx, ep.inproc_group.handle, self.rank)
            if not t:
                self._ck(1)
            self._native, self._native_kind = t, "inproc"
            return t
        if getattr(ep, "backend", None) == "nccl" and ep.device is not None and os.environ.get("CHSPH_SLAB_CALLBACK_TRANSPORT") is None:
            buf = (C.c_uint8 * 256)()
            n = api.slab_rccl_unique_id(buf, 256) if self.rank == 0 else 0
            box = [bytes(buf[:n]) if self.rank == 0 else None]
            ep.dist.broadcast_object_list(box, src=0)
            uid = (C.c_uint8 * len(box[0])).from_buffer_copy(box[0])
            t = api.slab_transport_rccl(self.ctx, uid, len(box[0]), self.rank, self.nranks)
            if not t:
                self._ck(1)
            self._native, self._native_kind = t, "rccl"
            return t

        ordered = bool(getattr(self, "native_ordered", False))

        def exchange(_user, send, sbytes, recv, rbytes, on_device, _stream):
            try:
                if ordered and _hiplib().hipStreamSynchronize(C.c_void_p(_stream)) != 0:
                    raise capi.SphError("hipStreamSynchronize failed")
                s, r = {}, {}
                for side, peer in self._peers():
                    if peer is None:
                        continue
                    if sbytes[side]:
                        s[peer] = (send[side] or 0, sbytes[side], on_device)
                    if rbytes[side]:
                        r[peer] = (recv[side] or 0, rbytes[side], on_device)
                # zero-length messages still travel: the peers post their sends / receives unconditionally
                for side, peer in self._peers():
                    if peer is not None:
                        s.setdefault(peer, (send[side] or 0, 0, on_device))
                        r.setdefault(peer, (recv[side] or 0, 0, on_device))
                ep.exchange(s, r)
                return 0
            except Exception as e:  # noqa: BLE001 -- must not propagate through the C frame
                self._native_error = e
                return 1

        def counts(_user, send, recv):
            try:
                sends = {peer: int(send[side]) for side, peer in self._peers() if peer is not None}
                got = ep.exchange_ints(sends, [p for _s, p in self._peers() if p is not None])
                recv[0] = got.get(self.left, 0) if self.left is not None else 0
                recv[1] = got.get(self.right, 0) if self.right is not None else 0
                return 0
            except Exception as e:  # noqa: BLE001
                self._native_error = e
                return 1

        def allreduce(_user, ptr, nwords, op, on_device, _stream):
            try:
                if ordered and _hiplib().hipStreamSynchronize(C.c_void_p(_stream)) != 0:
                    raise capi.SphError("hipStreamSynchronize failed")
                if op == capi.SPH_SLAB_REDUCE_MAX_U32:
                    ep.allreduce_max_words(ptr or 0, nwords, on_device)
                elif op == capi.SPH_SLAB_REDUCE_SUM_I32:
                    ep.allreduce_sum_words(ptr or 0, nwords, on_device)
                elif op == capi.SPH_SLAB_REDUCE_MIN_F32:
                    vals = np.frombuffer(_to_host(ptr or 0, 4 * nwords, on_device), dtype=np.float32).copy()
                    out = np.array([ep.allreduce_min_float(float(v)) for v in vals], dtype=np.float32)
                    _from_host(out.tobytes(), ptr or 0, 4 * nwords, on_device)
                else:
                    raise capi.SphError("unknown reduction %d" % op)
                return 0
            except Exception as e:  # noqa: BLE001
                self._native_error = e
                return 1

        self._cb = (capi.SLAB_EXCHANGE_FN(exchange), capi.SLAB_COUNTS_FN(counts), capi.SLAB_ALLREDUCE_FN(allreduce))  # keep the thunks alive
        # native_ordered (tests): the callbacks declare themselves stream-ordered -- they drain the stream they are handed before they move
        # anything, so everything is in place when they return -- which lets sph_slab_step take its overlapped path (split force kernel,
        # halo on a second stream) without a second GPU
        t = capi.SphSlabTransport(None, self.rank, self.nranks, 1 if ordered else 0, self._cb[0], self._cb[1], self._cb[2])
        self._native, self._native_kind = C.pointer(t), "callbacks"
        return self._native

    def close(self):
        """Release the library-owned transport of this rank (RCCL communicator, in-process channel membership) BEFORE its context goes:
        the transport holds the context.  (sph_destroy releases what is left, and a later free is a no-op, so the order cannot hurt.)"""
        t, kind = getattr(self, "_native", None), getattr(self, "_native_kind", None)
        if t is not None and kind in ("rccl", "inproc"):
            self.api.slab_transport_free(t)
        self._native = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass

    def step_native(self, h):
        """One step through sph_slab_step: the sequence of step() (explicit schemes) / _step_isph() (I2SPH) run by the library."""
        self._native_error = None
        rc = self.api.slab_step(self.ctx, self.time, h, self.native_transport())
        if rc and self._native_error is not None:
            raise self._native_error
        self._ck(rc)
        self.time += h
        self.sys.m_time = self.time
        self.sys.m_frame += 1

    def exchange_body_wrenches_native(self):
        """exchange_body_wrenches() inside the library (sph_slab_body_wrenches): table, integer all-reduce through the transport, reduction."""
        self._native_error = None
        rc = self.api.slab_body_wrenches(self.ctx, self.native_transport())
        if rc and self._native_error is not None:
            raise self._native_error
        self._ck(rc)

    def current_step_native(self):
        """current_step() inside the library (sph_slab_current_step): the minimum over the ranks is taken on the device."""
        self._native_error = None
        h = self.api.slab_current_step(self.ctx, self.native_transport())
        if h < 0:
            if self._native_error is not None:
                raise self._native_error
            self._ck(1)
        return h

    def exchange_body_wrenches(self):
        """Rigid_Forces_Torques of a slab run (collective): each rank fills the per-marker (force, torque) table for the rigid
        markers it owns, the tables are summed over the ranks, and every rank runs the single-device block reduction on the
        same data -- the wrenches are bit-identical to the single-device ones on every rank."""
        ptr, nb, dev = C.c_void_p(), C.c_size_t(), C.c_int()
        self._ck(self.api.slab_wrench_table(self.ctx, C.byref(ptr), C.byref(nb), C.byref(dev)))
        self.ep.allreduce_sum_words(ptr.value or 0, nb.value // 4, dev.value)
        self._ck(self.api.slab_wrench_reduce(self.ctx))

    def recut(self, min_width=4):
        """Collective, between two steps: move the slab cuts to where the work is now.  Every rank histograms its owned markers
        per cell column (fluid 1, wall WALL_MARKER_WEIGHT), the histograms are summed over the ranks, balanced_cuts() gives new
        cuts, clamped so that every marker's new owner is its old owner or a direct neighbour (new cut i stays inside the two old
        slabs next to it).  The library is re-configured with the new column range; the next step runs the two-round preamble,
        which moves the markers of the re-assigned columns like any other emigrants.  Returns the new cuts."""
        if self.nranks == 1:
            return list(self.cuts)
        p = self.sys.GetParams()
        gx = int(p.gridSize[0])
        s = self.sys.solver
        s.synchronize()
        pos, rpm = s.positions(), s.rho_pres_mu_type()
        w = np.where(rpm[:, 3] < -0.5, 1.0, WALL_MARKER_WEIGHT)
        hist = np.bincount(cell_column(pos[:, 0], p), weights=w, minlength=gx).astype(np.float64)
        hist = self.ep.allreduce_sum_array(hist)
        new = balanced_cuts(hist, gx, self.nranks, min_width=min_width, weights=_HISTOGRAM, column_cost=CELL_WEIGHT * int(p.gridSize[1]) * int(p.gridSize[2]))
        old = self.cuts
        for i in range(1, self.nranks):  # left to right: stay inside the old neighbours, keep the minimum width
            lo = max(old[i - 1], new[i - 1] + min_width)
            hi = min(old[i + 1], gx - min_width * (self.nranks - i))
            new[i] = min(max(new[i], lo), hi)
        if new != old:
            self.cuts = list(new)
            cfg = capi.SphSlabConfig(self.rank, self.nranks, new[self.rank], new[self.rank + 1], 2)
            self._ck(self.api.slab_configure(self.ctx, C.byref(cfg)))
            self.sys.slab_cuts = list(new)
        return list(self.cuts)

    def current_step(self):
        """GetCurrentStepSize under slabs: every rank evaluates 0.3 * min(dt_Courant, dt_acc) over its owned rows; the global
        step is the minimum over ranks (ncclAllReduce(min); the rank holding the minimum reproduces the single-device value)."""
        return self.ep.allreduce_min_float(self.api.current_step(self.ctx))

    # -- results, in global marker order pieces
    def owned(self):
        s = self.sys.solver
        s.synchronize()
        return {"gid": s.global_ids(), "pos": s.positions(), "vel": s.velocities(), "rpm": s.rho_pres_mu_type(), "deriv": s.derivatives()}


def gather_global(pieces, n_total, extra=()):
    """Assemble per-rank `owned()` dicts into global original-order arrays (`extra`: further per-marker arrays in the pieces)."""
    out = {}
    seen = np.zeros(n_total, dtype=np.int32)
    for pc in pieces:
        g = pc["gid"].astype(np.int64)
        seen[g] += 1
        for k in ("pos", "vel", "rpm", "deriv") + tuple(extra):
            if k not in out:
                out[k] = np.zeros((n_total, pc[k].shape[1]), dtype=np.float32)
            out[k][g] = pc[k]
    if not np.all(seen == 1):
        raise capi.SphError("slab gather: %d markers owned by no rank, %d by several" % (int((seen == 0).sum()), int((seen > 1).sum())))
    return out


def run_threads(ranks, nsteps, h, native=False):
    """Step all ranks of an in-process group (ThreadTransport) in lockstep, one thread per rank.  native: through sph_slab_step."""
    errs = []

    def work(r):
        try:
            for _ in range(nsteps):
                (r.step_native if native else r.step)(h)
        except Exception as e:  # noqa: BLE001
            errs.append(e)
            tr = getattr(r.ep, "tr", None)
            if tr is not None and hasattr(tr, "failed"):
                tr.failed.set()

    ts = [threading.Thread(target=work, args=(r,)) for r in ranks]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    if errs:
        raise errs[0]
