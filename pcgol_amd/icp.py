"""pc/registration/icp mirror: NearestPointCorresponder, PointToPointEvaluator,
GradientDescentUpdaterFactory and PointToPointICPGradient.Fit on the GPU."""
import ctypes as C

import numpy as np

from . import _lib as L
from . import mat
from .kdtree import KDTree

ErrNotEnoughPairs = L.ErrNotEnoughPairs  # evaluator.go:16
ErrNeedGradient = L.ErrNeedGradient      # icp.go:15
ErrSingular = L.ErrSingular              # point-to-plane extension


class PointToPointCorrespondence:  # correspondence.go:8-12
    __slots__ = ("BaseID", "TargetID", "SquaredDistance")

    def __init__(self, b, t, d):
        self.BaseID, self.TargetID, self.SquaredDistance = int(b), int(t), np.float32(d)

    def __eq__(self, o):
        return (self.BaseID, self.TargetID, self.SquaredDistance) == (o.BaseID, o.TargetID, o.SquaredDistance)

    def __repr__(self):
        return "{%d %d %r}" % (self.BaseID, self.TargetID, float(self.SquaredDistance))


class NearestPointCorresponder:  # correspondence.go:18-37
    def __init__(self, MaxDist):
        self.MaxDist = float(MaxDist)

    def PairsArrays(self, base, target):
        target = L.f32c(target).reshape(-1, 3)
        n = len(target)
        b = np.empty(max(n, 1), np.int64)
        t = np.empty(max(n, 1), np.int64)
        d = np.empty(max(n, 1), np.float32)
        m = C.c_int64()
        L.check(L.lib().pcgx_icp_pairs(base._h, L.ptr(target), n, self.MaxDist, base.MinDistSq, L.ptr(b),
                                       L.ptr(t), L.ptr(d), C.byref(m)))
        return b[: m.value], t[: m.value], d[: m.value]

    def Pairs(self, base, target):
        return [PointToPointCorrespondence(*r) for r in zip(*self.PairsArrays(base, target))]


class Evaluated:  # evaluator.go:25-30
    def __init__(self, ev=None):
        self.Value = np.float32(ev.value if ev else 0)
        self.Gradient = np.array(list(ev.gradient) if ev else [0] * 6, np.float32)
        self.Hessian = np.zeros(36, np.float32)  # never written by the reference (:28,:76)
        self.DistRMS = np.float32(ev.dist_rms if ev else 0)
        self.NumPairs = int(ev.num_pairs) if ev else 0


class WeightFn:
    """PointToPointEvaluator.WeightFn (evaluator.go:19-23): the reference takes any closure of the
    squared distance; the device evaluates one of these built-in forms, each in float32 exactly as
    the Go expression reads (include/pcgx.h PCGX_WEIGHT_*).  Calling the object evaluates the same
    expression on the host (numpy float32)."""

    def __init__(self, kind, a=0.0):
        self.kind, self.a = int(kind), np.float32(a)

    def __call__(self, d):
        f, a, d = np.float32, self.a, np.float32(d)
        if self.kind == L.PCGX_WEIGHT_CONSTANT:
            return a
        if self.kind == L.PCGX_WEIGHT_INVERSE:
            return f(f(1) / f(a + d))
        if self.kind == L.PCGX_WEIGHT_HUBER:
            return f(1) if d <= a else f(np.sqrt(np.float64(f(a / d))))
        if self.kind == L.PCGX_WEIGHT_TUKEY:
            if not d < a:
                return f(0)
            u = f(f(1) - f(d / a))
            return f(u * u)
        return f(1)


def WeightConstant(a):
    return WeightFn(L.PCGX_WEIGHT_CONSTANT, a)


def WeightInverse(a):
    return WeightFn(L.PCGX_WEIGHT_INVERSE, a)


def WeightHuber(k_sq):
    return WeightFn(L.PCGX_WEIGHT_HUBER, k_sq)


def WeightTukey(c_sq):
    return WeightFn(L.PCGX_WEIGHT_TUKEY, c_sq)


SumsReference = L.PCGX_SUMS_REFERENCE            # the reference's sequential float32 sums (default; one GPU)
SumsF64Tree = L.PCGX_SUMS_F64_TREE                # fixed-order float64 reduction of the same terms
SumsReferenceChain = L.PCGX_SUMS_REFERENCE_CHAIN  # the reference's sums by one wave (cross-check)


class PointToPointEvaluator:  # evaluator.go:69-76
    def __init__(self, Corresponder, MinPairs=0, WeightFn=None, SumsMode=SumsReference):
        """SumsMode (not in the reference): include/pcgx.h PCGX_SUMS_*; the default forms the sums as
        the Go code does, bit for bit."""
        if WeightFn is not None and not isinstance(WeightFn, globals()["WeightFn"]):
            raise NotImplementedError("a custom WeightFn closure cannot run on the device: use one of the built-in "
                                      "forms (icp.WeightConstant / WeightInverse / WeightHuber / WeightTukey)")
        if not isinstance(Corresponder, NearestPointCorresponder):
            raise TypeError("the GPU evaluator fuses NearestPointCorresponder")
        self.Corresponder = Corresponder
        self.MinPairs = int(MinPairs)
        self.WeightFn = WeightFn
        self.SumsMode = int(SumsMode)

    def HasGradient(self):
        return True

    def HasHessian(self):
        return False

    def Evaluate(self, base, target):
        if not isinstance(base, KDTree):
            raise TypeError("base must be a pcgol_amd KDTree")
        target = L.f32c(target).reshape(-1, 3)
        ev = L.IcpEvaluated()
        p = _params(self.Corresponder.MaxDist, base.MinDistSq, self.MinPairs, np.zeros(6), np.zeros(6), 0, self.WeightFn,
                    self.SumsMode)
        L.check(L.lib().pcgx_icp_evaluate_params(base._h, L.ptr(target), len(target), C.byref(p), C.byref(ev)))
        return Evaluated(ev)


def _params(max_dist, min_dist_sq, min_pairs, weight, threshold, max_iteration, weight_fn=None, sums_mode=0):
    p = L.IcpParams()
    p.max_dist, p.min_dist_sq, p.min_pairs, p.max_iteration = max_dist, min_dist_sq, min_pairs, max_iteration
    p.sums_mode = int(sums_mode)
    if weight_fn is not None:
        p.weight_fn, p.weight_fn_param = weight_fn.kind, float(weight_fn.a)
    for i in range(6):
        p.weight[i] = float(weight[i])
        p.threshold[i] = float(threshold[i])
    return p


class GradientDescentUpdaterFactory:  # updater.go:18-37
    def __init__(self, Weight=None, Threshold=None, MaxIteration=0):
        self.Weight = np.zeros(6, np.float32) if Weight is None else np.asarray(Weight, np.float32)
        self.Threshold = np.zeros(6, np.float32) if Threshold is None else np.asarray(Threshold, np.float32)
        self.MaxIteration = int(MaxIteration)

    def New(self):
        return _GradientDescentUpdater(self)


class _GradientDescentUpdater:  # updater.go:39-71
    def __init__(self, f):
        self.f = f
        self.i = 0

    def Update(self, trans, ev):
        p = _params(0, 0, 0, self.f.Weight, self.f.Threshold, self.f.MaxIteration)
        it = C.c_int32(self.i)
        t = L.f32c(trans).copy()
        g = L.f32c(ev.Gradient)
        conv = C.c_int32()
        L.check(L.lib().pcgx_icp_update(C.byref(p), C.byref(it), L.ptr(g), L.ptr(t), C.byref(conv)))
        self.i = it.value
        return t, bool(conv.value)


def FinishEvaluate(sums10, MinPairs=0):
    """evaluator.go:92-105,156-186 from the 10 (all-reduced) float64 sums."""
    s = np.ascontiguousarray(sums10, dtype=np.float64)
    ev = L.IcpEvaluated()
    L.check(L.lib().pcgx_icp_finish_evaluate(L.ptr(s), int(MinPairs), C.byref(ev)))
    return Evaluated(ev)


class Stat:  # stat.go:3-6
    def __init__(self, st=None):
        self.Evaluated = Evaluated(st.evaluated if st else None)
        self.NumIteration = st.num_iteration if st else 0


class PointToPointICPGradient:  # icp.go:18-67
    def __init__(self, Evaluator, UpdaterFactory=None):
        self.Evaluator = Evaluator
        self.UpdaterFactory = UpdaterFactory

    def Fit(self, base, target):
        """(trans[16], Stat).  Raises ErrNotEnoughPairs with .trans/.stat attached, like icp.go:49-53."""
        ev = self.Evaluator
        if not ev.HasGradient():
            raise ErrNeedGradient(L.PCGX_E_NEED_GRADIENT, "need gradient output of Evaluator")
        uf = self.UpdaterFactory or GradientDescentUpdaterFactory()
        target = L.f32c(target).reshape(-1, 3)
        p = _params(ev.Corresponder.MaxDist, base.MinDistSq, ev.MinPairs, uf.Weight, uf.Threshold, uf.MaxIteration,
                    ev.WeightFn, ev.SumsMode)
        trans = np.empty(16, np.float32)
        st = L.IcpStat()
        rc = L.lib().pcgx_icp_fit(base._h, L.ptr(target), len(target), C.byref(p), L.ptr(trans), C.byref(st))
        if rc == L.PCGX_E_NOT_ENOUGH_PAIRS:
            e = ErrNotEnoughPairs(rc, L.last_error())
            e.trans, e.stat = trans, Stat(st)
            raise e
        L.check(rc)
        return trans, Stat(st)


class IcpSession:
    """Device-resident Fit loop cut at the per-iteration exchange (include/pcgx.h)."""

    def __init__(self, base, target, MaxDist, MinPairs=0, Weight=None, Threshold=None, MaxIteration=0,
                 d_sums10=0, target_on_device=False, nt=None, BaseNormals=None, Damping=0.0, WeightFn=None,
                 SumsMode=SumsReference):
        """BaseNormals (unit normals per base point, id order; a device address when
        target_on_device) selects the point-to-plane / Gauss-Newton extension: the exchange
        vector then has 30 doubles (d_sums10 must point to 30).  SumsMode: PCGX_SUMS_* (default: the
        reference's sequential float32 sums; a session stepped through an exchange forms float64 sums)."""
        w = np.zeros(6, np.float32) if Weight is None else Weight
        th = np.zeros(6, np.float32) if Threshold is None else Threshold
        self.params = _params(MaxDist, base.MinDistSq, MinPairs, w, th, MaxIteration, WeightFn, SumsMode)
        self.max_iteration = MaxIteration or 20
        self.base = base
        self.plane = BaseNormals is not None
        self.n_sums = 30 if self.plane else 10
        if target_on_device:
            tptr, n = L.ptr(int(target)), int(nt)
        else:
            self._t = L.f32c(target).reshape(-1, 3)
            tptr, n = L.ptr(self._t), len(self._t)
        h = C.c_void_p()
        sums = L.ptr(int(d_sums10)) if d_sums10 else None
        if self.plane:
            if target_on_device:
                nptr = L.ptr(int(BaseNormals))
            else:
                self._n = L.f32c(BaseNormals).reshape(-1, 3)
                if len(self._n) != base.Len():
                    raise ValueError("BaseNormals must hold one normal per base point")
                nptr = L.ptr(self._n)
            L.check(L.lib().pcgx_icp_plane_session_create(base._h, nptr, tptr, n, 1 if target_on_device else 0,
                                                          C.byref(self.params), float(Damping), sums, C.byref(h)))
        else:
            L.check(L.lib().pcgx_icp_session_create(base._h, tptr, n, 1 if target_on_device else 0,
                                                    C.byref(self.params), sums, C.byref(h)))
        self._h = h

    def grid_stats(self, stream=0):
        """Measurement aid: (targets, targets left to the walk, point records, cell-bound words,
        lane-slots of the scan loops, targets kept on their partner's certificate without a search)
        of the grid pass of the NEXT iteration; zeros when the base tree has no grid."""
        out = (C.c_int64 * 6)()
        L.check(L.lib().pcgx_debug_icp_grid_stats(self._h, L.ptr(stream) if stream else None, out))
        return tuple(out)

    def partials(self, stream=0):
        L.check(L.lib().pcgx_icp_session_partials(self._h, L.ptr(stream) if stream else None))

    def update(self, stream=0):
        L.check(L.lib().pcgx_icp_session_update(self._h, L.ptr(stream) if stream else None))

    def step(self, stream=0):
        L.check(L.lib().pcgx_icp_session_step(self._h, L.ptr(stream) if stream else None))

    def set_pose(self, trans, it, stream=0):
        t = L.f32c(trans)
        L.check(L.lib().pcgx_icp_session_set_pose(self._h, L.ptr(t), int(it), L.ptr(stream) if stream else None))

    def set_strict(self, on=True):
        """Change the sums after creation (SumsMode sets them at creation; the default is 1).
        True / 1: sequential float32 sums in target order, as the Go code adds them, evaluated in
        parallel by the whole GPU (csrc/strict_sum.h): bit-identical Evaluated / pose at any size;
        2: one wave adding term after term (cross-check); 0: float64 tree.  See include/pcgx.h."""
        L.check(L.lib().pcgx_icp_session_set_strict(self._h, int(on)))

    def strict_stats(self, stream=0):
        """Measurement aid: counters of the parallel strict sums since the last call
        (runs applied, runs failed, tiles recomputed, leaves added serially, tile records failed)."""
        out = np.zeros(64, np.int64)
        L.check(L.lib().pcgx_debug_icp_strict_stats(self._h, L.ptr(stream) if stream else None, L.ptr(out)))
        return out

    def read_sums(self, stream=0):
        out = np.empty(self.n_sums, np.float64)
        L.check(L.lib().pcgx_icp_session_read_sums_n(self._h, L.ptr(out), self.n_sums,
                                                     L.ptr(stream) if stream else None))
        return out

    def hessian(self, stream=0):
        out = np.empty(36, np.float32)
        L.check(L.lib().pcgx_icp_session_hessian(self._h, L.ptr(stream) if stream else None, L.ptr(out)))
        return out

    def reset(self, stream=0):
        L.check(L.lib().pcgx_icp_session_reset(self._h, L.ptr(stream) if stream else None))

    def result(self, stream=0):
        trans = np.empty(16, np.float32)
        st = L.IcpStat()
        conv = C.c_int32()
        rc = L.lib().pcgx_icp_session_result(self._h, L.ptr(stream) if stream else None, L.ptr(trans),
                                             C.byref(st), C.byref(conv))
        if rc == L.PCGX_E_NOT_ENOUGH_PAIRS:
            e = ErrNotEnoughPairs(rc, L.last_error())
            e.trans, e.stat = trans, Stat(st)
            raise e
        L.check(rc)
        stat = Stat(st)
        if self.plane:
            stat.Evaluated.Hessian = self.hessian(stream)
        return trans, stat, bool(conv.value)

    def close(self):
        if getattr(self, "_h", None) and L is not None:  # L is None during interpreter shutdown
            L.lib().pcgx_icp_session_free(self._h)
            self._h = None

    __del__ = close


# ---------------------------------------------------------------------------
# Point-to-plane / Gauss-Newton extension (include/pcgx.h "point-to-plane ICP (extension)").
# NOT in the reference: it fills the reference's unused slots Evaluated.Hessian / HasHessian()
# (evaluator.go:28,35,76) behind the same Evaluator / Updater / Fit shapes.

class PointToPlaneEvaluator:
    """Evaluator whose residual is the distance to the matched base point's tangent plane.
    BaseNormals: (n, 3) unit normals, one per base point in the tree's id order."""

    def __init__(self, Corresponder, BaseNormals, MinPairs=0):
        if not isinstance(Corresponder, NearestPointCorresponder):
            raise TypeError("the GPU evaluator fuses NearestPointCorresponder")
        self.Corresponder = Corresponder
        self.BaseNormals = L.f32c(BaseNormals).reshape(-1, 3)
        self.MinPairs = int(MinPairs)

    def HasGradient(self):
        return True

    def HasHessian(self):
        return True

    def Sums(self, base, target):
        """The 30 float64 sums of one evaluation (what N ranks all-reduce)."""
        s = IcpSession(base, target, self.Corresponder.MaxDist, self.MinPairs, BaseNormals=self.BaseNormals)
        try:
            s.partials()
            return s.read_sums()
        finally:
            s.close()

    def Evaluate(self, base, target):
        return FinishEvaluatePlane(self.Sums(base, target), self.MinPairs)


def FinishEvaluatePlane(sums30, MinPairs=0):
    s = np.ascontiguousarray(sums30, dtype=np.float64)
    ev = L.IcpEvaluated()
    h = np.empty(36, np.float32)
    L.check(L.lib().pcgx_icp_plane_finish_evaluate(L.ptr(s), int(MinPairs), C.byref(ev), L.ptr(h)))
    out = Evaluated(ev)
    out.Hessian = h
    return out


class GaussNewtonUpdaterFactory:
    """Threshold / MaxIteration as GradientDescentUpdaterFactory (zero -> 0.01 / 20); Damping is the
    Levenberg-Marquardt factor on diag(H)."""

    def __init__(self, Threshold=None, MaxIteration=0, Damping=0.0):
        self.Weight = np.zeros(6, np.float32)
        self.Threshold = np.zeros(6, np.float32) if Threshold is None else np.asarray(Threshold, np.float32)
        self.MaxIteration = int(MaxIteration)
        self.Damping = float(Damping)

    def New(self):
        return _GaussNewtonUpdater(self)


class _GaussNewtonUpdater:
    def __init__(self, f):
        self.f = f
        self.i = 0

    def Update(self, trans, ev):
        p = _params(0, 0, 0, self.f.Weight, self.f.Threshold, self.f.MaxIteration)
        it = C.c_int32(self.i)
        t = L.f32c(trans).copy()
        g, h = L.f32c(ev.Gradient), L.f32c(ev.Hessian)
        conv = C.c_int32()
        L.check(L.lib().pcgx_icp_gauss_newton_update(C.byref(p), self.f.Damping, C.byref(it), L.ptr(g), L.ptr(h),
                                                     L.ptr(t), C.byref(conv)))
        self.i = it.value
        return t, bool(conv.value)


class PointToPlaneICP:
    """Fit loop of icp.go:23-67 with PointToPlaneEvaluator + GaussNewtonUpdaterFactory, on the device."""

    def __init__(self, Evaluator, UpdaterFactory=None):
        if not isinstance(Evaluator, PointToPlaneEvaluator):
            raise TypeError("PointToPlaneICP needs a PointToPlaneEvaluator")
        self.Evaluator = Evaluator
        self.UpdaterFactory = UpdaterFactory

    def Fit(self, base, target):
        ev = self.Evaluator
        uf = self.UpdaterFactory or GaussNewtonUpdaterFactory()
        target = L.f32c(target).reshape(-1, 3)
        if len(ev.BaseNormals) != base.Len():
            raise ValueError("BaseNormals must hold one normal per base point")
        p = _params(ev.Corresponder.MaxDist, 0.0, ev.MinPairs, uf.Weight, uf.Threshold, uf.MaxIteration)
        trans = np.empty(16, np.float32)
        st = L.IcpStat()
        h = np.zeros(36, np.float32)
        rc = L.lib().pcgx_icp_plane_fit(base._h, L.ptr(ev.BaseNormals), L.ptr(target), len(target), C.byref(p),
                                        uf.Damping, L.ptr(trans), C.byref(st), L.ptr(h))
        if rc == L.PCGX_E_NOT_ENOUGH_PAIRS:
            e = ErrNotEnoughPairs(rc, L.last_error())
            e.trans, e.stat = trans, Stat(st)
            raise e
        L.check(rc)
        stat = Stat(st)
        stat.Evaluated.Hessian = h
        return trans, stat
