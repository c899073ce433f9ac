"""Multi-GPU ICP: one process per GPU, target tiles sharded across ranks, base
KD-tree replicated, ONE exchange per iteration: the all-reduce (sum) of the 10
float64 partial sums.  torch.distributed's "nccl" backend is
RCCL on ROCm (xGMI inside a node); "gloo" runs the same host logic on CPU.

Nothing else on the path communicates: kNN batches and the voxel filter shard
by independent tiles / chunks with no collective.
"""
import numpy as np

from . import _lib as L
from . import icp as _icp
from . import mat


def morton30(pts, lo, hi):
    """30-bit Morton code (host, numpy) used to cut a cloud into spatial tiles."""
    ext = np.maximum((hi - lo).astype(np.float64), 1e-30)
    cells = np.clip(((pts.astype(np.float64) - lo) / ext * 1024.0).astype(np.int64), 0, 1023)

    def spread(v):
        v = v & 0x3FF
        v = (v | (v << 16)) & 0x030000FF
        v = (v | (v << 8)) & 0x0300F00F
        v = (v | (v << 4)) & 0x030C30C3
        v = (v | (v << 2)) & 0x09249249
        return v
    return spread(cells[:, 0]) | (spread(cells[:, 1]) << 1) | (spread(cells[:, 2]) << 2)


def spatial_tiles(points, world):
    """Index arrays of `world` spatial tiles: the cloud in Morton order cut into contiguous,
    equally sized ranges.  The tiles partition the cloud (every point in exactly one tile)."""
    points = np.asarray(points, np.float32).reshape(-1, 3)
    order = np.argsort(morton30(points, points.min(axis=0), points.max(axis=0)), kind="stable")
    bounds = [(len(points) * r) // world for r in range(world + 1)]
    return [order[bounds[r]:bounds[r + 1]] for r in range(world)]


def allreduce_sums(sums10, group=None):
    """Sum of the 10 partial sums over all ranks.  numpy float64[10] (host, any backend that
    supports CPU tensors) or a torch tensor (device tensors go through RCCL)."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return sums10
    if isinstance(sums10, np.ndarray):
        t = torch.from_numpy(np.ascontiguousarray(sums10, dtype=np.float64).copy())
        if dist.get_backend(group) == "nccl":
            t = t.cuda()
        dist.all_reduce(t, group=group)
        return t.cpu().numpy()
    dist.all_reduce(sums10, group=group)
    return sums10


def fit_sharded(partials_fn, MinPairs=0, UpdaterFactory=None, group=None):
    """The reference's Fit loop (icp.go:46-66) with the evaluator split at the exchange:

        sums   = partials_fn(trans, it)       # this rank's 10 float64 sums at the current pose
        sums   = all-reduce(sums)             # the only collective
        ev     = finish_evaluate(sums)        # evaluator.go:156-186, identical on every rank
        trans  = updater.Update(trans, ev)    # updater.go:44-71, identical on every rank

    `it` is the number of pose updates applied so far (0: the raw target is evaluated).
    Returns (trans, Stat).  Raises ErrNotEnoughPairs like Fit.  Host-driven: used by the
    backend-agnostic tests and by hosts that bring their own exchange; bench.py uses
    ShardedIcp below, which keeps the loop on the device."""
    uf = UpdaterFactory or _icp.GradientDescentUpdaterFactory()
    # point-to-plane extension: 30 sums (J^T J upper triangle, J^T r, ...) and a Gauss-Newton update
    plane = isinstance(uf, _icp.GaussNewtonUpdaterFactory)
    finish = _icp.FinishEvaluatePlane if plane else _icp.FinishEvaluate
    updater = uf.New()
    trans = mat.Translate(0, 0, 0)
    stat = _icp.Stat()
    while True:
        sums = allreduce_sums(np.asarray(partials_fn(trans, updater.i), np.float64), group)
        if len(sums) != (30 if plane else 10):
            raise ValueError("partials_fn must return %d sums" % (30 if plane else 10))
        stat.NumIteration += 1
        try:
            ev = finish(sums, MinPairs)
        except L.ErrNotEnoughPairs as e:
            e.trans, e.stat = trans, stat
            raise
        stat.Evaluated = ev
        trans, converged = updater.Update(trans, ev)
        if converged:
            return trans, stat


class Comm:
    """The exchange of the sharded path behind the C ABI (include/pcgx.h, csrc/comm.hip).

    Comm.rccl(rank, world, store): RCCL communicator; the ncclUniqueId travels from rank 0 to the
    others through `store` (anything with set(key, bytes) / get(key) -> bytes, e.g.
    torch.distributed.TCPStore).  Comm.callback(rank, world, fn): the exchange through a host
    function fn(numpy float64 array) that sums the array over the ranks in place (e.g. gloo)."""

    def __init__(self, handle, keep=None, world=1):
        self._h = handle
        self._keep = keep
        self.world = int(world)

    @classmethod
    def rccl(cls, rank, world, store, key="pcgx_comm_id"):
        import ctypes as C
        buf = C.create_string_buffer(128)
        if rank == 0:
            L.check(L.lib().pcgx_comm_unique_id(buf))
            store.set(key, buf.raw)
        else:
            raw = bytes(store.get(key))
            assert len(raw) == 128
            buf = C.create_string_buffer(raw, 128)
        h = C.c_void_p()
        L.check(L.lib().pcgx_comm_init(rank, world, buf, C.byref(h)))
        return cls(h, world=world)

    @classmethod
    def callback(cls, rank, world, fn):
        import ctypes as C
        proto = C.CFUNCTYPE(C.c_int32, C.POINTER(C.c_double), C.c_int32, C.c_void_p)

        def tramp(ptr, count, _user):
            try:
                fn(np.ctypeslib.as_array(ptr, shape=(count,)))
                return 0
            except Exception:   # the C side turns this into PCGX_E_RCCL
                import traceback
                traceback.print_exc()
                return 1
        cb = proto(tramp)
        h = C.c_void_p()
        L.check(L.lib().pcgx_comm_init_callback(rank, world, C.cast(cb, C.c_void_p), None, C.byref(h)))
        return cls(h, keep=cb, world=world)

    @classmethod
    def gloo(cls, group=None):
        """Callback communicator over an initialised torch.distributed group (CPU tensors: gloo)."""
        import torch
        import torch.distributed as dist

        def fn(a):
            t = torch.from_numpy(a)
            dist.all_reduce(t, group=group)
        return cls.callback(dist.get_rank(group), dist.get_world_size(group), fn)

    def close(self):
        if self._h:
            L.check(L.lib().pcgx_comm_free(self._h))
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class ShardedIcp:
    """Device-resident sharded Fit: every rank holds the whole base tree and one tile of the
    target; per iteration the partial sums are all-reduced in place on the device (RCCL) and the
    pose update runs on every GPU redundantly, so no rank ever waits for the host."""

    def __init__(self, base_tree, target_tile, MaxDist, MinPairs=0, Weight=None, Threshold=None,
                 MaxIteration=0, group=None, force_exchange=False, BaseNormals=None, Damping=0.0, comm=None,
                 SumsMode=None):
        """SumsMode: None = float64 sums wherever there is an exchange (one all-reduce per iteration), the reference's sums
        on one rank; icp.SumsReference with comm=...: the reference's sums over the ranks' tiles one after the other
        (the library's default for a sharded Fit: bit-identical to the Fit of the concatenated target).
        BaseNormals: point-to-plane / Gauss-Newton extension; the exchange is then the all-reduce
        of 30 doubles (sum r^2, J^T r, upper triangle of J^T J, sum w, pairs) instead of 10.
        comm: a Comm -- the exchange then runs inside libpcgx.so (pcgx_icp_session_step_sharded: what
        a Go host calls); without it the all-reduce is torch.distributed's on the sums tensor."""
        import torch
        self.torch = torch
        self.group = group
        self.comm = comm
        # the library works on ONE device per process: bind it to torch's current one (fails loudly if it
        # was initialised on another)
        L.check(L.lib().pcgx_init(torch.cuda.current_device()))
        # A stream of our own: the kernels are launched on it through the C ABI and the
        # all-reduce is issued while it is torch's current stream, so RCCL orders itself after
        # the partial sums and the update kernel after RCCL.  (torch's default stream has
        # handle 0, which the C ABI reads as "use the library's stream": never use it here.)
        self.stream = torch.cuda.Stream()
        self.sums = torch.zeros(30 if BaseNormals is not None else 10, dtype=torch.float64, device="cuda")
        torch.cuda.current_stream().synchronize()
        import torch.distributed as dist
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        # force_exchange: take the partials -> all-reduce -> update path even with one rank (tests)
        self.exchange = self.world > 1 or (force_exchange and dist.is_initialized())
        # float64 sums wherever there is an exchange unless asked otherwise (partials / update around torch's all-reduce
        # can only add float64 sums; the library's own step forms the reference's sums over the ranks as well), the
        # reference's own sums (the library's default) on one rank
        sharded = self.exchange or (comm is not None and comm.world > 1)
        self.sess = _icp.IcpSession(base_tree, target_tile, MaxDist, MinPairs, Weight, Threshold, MaxIteration,
                                    d_sums10=self.sums.data_ptr(), BaseNormals=BaseNormals, Damping=Damping,
                                    SumsMode=SumsMode if SumsMode is not None else
                                    (_icp.SumsF64Tree if sharded else _icp.SumsReference))
        self.max_iteration = self.sess.max_iteration

    def step(self):
        """One ICP iteration, enqueued on self.stream."""
        st = self.stream.cuda_stream
        if self.comm is not None:
            L.check(L.lib().pcgx_icp_session_step_sharded(self.sess._h, self.comm._h, L.ptr(st)))
            return
        if not self.exchange:
            self.sess.step(st)  # reduce + update fused: no exchange needed
            return
        with self.torch.cuda.stream(self.stream):
            self.sess.partials(st)
            self.torch.distributed.all_reduce(self.sums, group=self.group)
            self.sess.update(st)

    def reset(self):
        self.sess.reset(self.stream.cuda_stream)

    def fit(self):
        self.reset()
        for _ in range(self.max_iteration):
            self.step()
        return self.result()

    def result(self):
        return self.sess.result(self.stream.cuda_stream)

    def close(self):
        self.sess.close()
