// Package pcgx binds libpcgx.so (include/pcgx.h), the MI355X hot path, behind
// seqsense/pcgol's interfaces: storage.Search, filter.Filter, icp.Evaluator.
//
// The reference's package surface over this binding -- kdtree.New, voxelgrid.New / WithChunkSize,
// icp.NearestPointCorresponder / PointToPointEvaluator / PointToPointICPGradient under their own names and
// signatures -- is in ../pc/storage/kdtree, ../pc/filter/voxelgrid, ../pc/registration/icp.
//
// NOT compiled in the build image (no Go toolchain there); see go/README.md.
package pcgx

/*
#cgo CFLAGS: -I${SRCDIR}/../../include
#cgo LDFLAGS: -lpcgx
#include <stdlib.h>
#include "pcgx.h"
*/
import "C"

import (
	"errors"
	"fmt"
	"math"
	"runtime"
	"sync/atomic"
	"unsafe"

	"github.com/seqsense/pcgol/mat"
	"github.com/seqsense/pcgol/pc"
	"github.com/seqsense/pcgol/pc/filter"
	"github.com/seqsense/pcgol/pc/registration/icp"
	"github.com/seqsense/pcgol/pc/storage"
)

// (import path of this package: github.com/seqsense/pcgol/gpu/pcgx, go/go.mod)

// ErrOutOfRange is returned where the pure-Go filter would panic with
// "index out of range" (pc/filter/voxelgrid/voxelgrid.go:151).
var ErrOutOfRange = errors.New("pcgx: point outside the dense voxel grid")

// ErrNeedsDevice: a sharded Fit was asked of an evaluator that cannot run on the device (a WeightFn closure, a
// Corresponder of the caller's own).
var ErrNeedsDevice = errors.New("pcgx: this evaluator cannot run on the device (WeightFn closure or foreign Corresponder)")

// singlePointCalls counts KDTree.Nearest / Range calls for ONE point: each is a blocking GPU call (tens of
// microseconds), so a loop over them -- the reference's own correspondence.go:25-36, regiongrowing.go:26,47 -- is
// slower than the CPU tree.  Tests of callers that were moved to the batch seams assert the counter stays put.
var singlePointCalls int64

// SinglePointCalls returns how many single-point Nearest / Range calls this process has made.
func SinglePointCalls() int64 { return atomic.LoadInt64(&singlePointCalls) }

func lastError() string {
	buf := make([]byte, 512)
	C.pcgx_last_error((*C.char)(unsafe.Pointer(&buf[0])), C.size_t(len(buf)))
	for i, b := range buf {
		if b == 0 {
			return string(buf[:i])
		}
	}
	return string(buf)
}

// status maps pcgx_status onto the reference's sentinel errors so that
// errors.Is keeps working for callers.
func status(rc C.pcgx_status) error {
	switch rc {
	case C.PCGX_OK:
		return nil
	case C.PCGX_E_NO_POINT:
		return errors.New("no point") // pc/minmax.go:11
	case C.PCGX_E_NOT_ENOUGH_PAIRS:
		return icp.ErrNotEnoughPairs // icp/evaluator.go:16
	case C.PCGX_E_NEED_GRADIENT:
		return icp.ErrNeedGradient // icp/icp.go:15
	case C.PCGX_E_BAD_FIELD:
		return errors.New("invalid field name") // pc/pointcloud.go:115
	case C.PCGX_E_OUT_OF_RANGE:
		return ErrOutOfRange
	default:
		// callers hold the OS thread (runtime.LockOSThread) from the cgo call to here: the C side
		// keeps the message per thread
		return fmt.Errorf("pcgx: %s (status %d)", lastError(), int(rc))
	}
}

// Init selects the GPU of this process (one process per GPU).
func Init(device int) error { return status(C.pcgx_init(C.int32_t(device))) }

func init() {
	// the struct layouts this file was compiled against (include/pcgx.h) must be the library's
	if v := int(C.pcgx_abi_version()); v != C.PCGX_ABI_VERSION {
		panic(fmt.Sprintf("pcgx: libpcgx.so speaks ABI version %d, this package was built against %d", v, int(C.PCGX_ABI_VERSION)))
	}
}

// xyzLayout finds stride and xyz byte offset of a cloud exactly as
// PointCloud.Vec3Iterator does (pc/pointcloud.go:130-150).
func xyzLayout(pp *pc.PointCloud) (stride, off int, err error) {
	state, start := 0, 0
	for i, name := range pp.Fields {
		switch {
		case name == "xyz":
			return pp.Stride(), off, nil
		case name == "x" && state == 0:
			state, start = 1, off
		case name == "y" && state == 1:
			state = 2
		case name == "z" && state == 2:
			return pp.Stride(), start, nil
		default:
			state = 0
		}
		off += pp.Size[i] * pp.Count[i]
	}
	return 0, 0, errors.New("invalid field name")
}

// ---------------------------------------------------------------- KD-tree

// KDTree implements storage.Search on the GPU (replaces kdtree.New /
// KDTree.Nearest, pc/storage/kdtree/kdtree.go:33-56,83-146).
type KDTree struct {
	pc.Vec3RandomAccessor
	t *devTree // shared by the shallow copies With() makes (kdtree.go:58-65)
	// MinDistSq > 0 selects the reference's approximate search (kdtree.go:20-22).
	MinDistSq float32
}

// devTree owns the device handle; the copies of a KDTree share it, the last one garbage-collected (or the
// first Close) releases it.
type devTree struct {
	h *C.pcgx_kdtree
}

// KDTreeOption mirrors kdtree.KDTreeOption (kdtree.go:31): New(ra, opts...) and (*KDTree).With(opts...)
// apply them to the tree value exactly as the reference does.  The reference defines no option of its
// own (its one knob is the exported field MinDistSq); WithMinDistSq is the option form of that field.
type KDTreeOption func(*KDTree)

// WithMinDistSq sets KDTree.MinDistSq (kdtree.go:20-22).
func WithMinDistSq(d float32) KDTreeOption { return func(k *KDTree) { k.MinDistSq = d } }

var _ storage.Search = (*KDTree)(nil)

// cloudLayout says where a Vec3RandomAccessor's points already lie packed in memory, so that New can
// hand the caller's bytes to pcgx_kdtree_build as they are (data, stride, xyz offset -- the record layout
// of pc/pointcloud.go:64-78) instead of copying them out through n Vec3At interface calls: at 1M points
// that loop costs more than the 2.5 ms device build.
func cloudLayout(ra pc.Vec3RandomAccessor) (data unsafe.Pointer, stride, off int, keep interface{}, ok bool) {
	switch v := ra.(type) {
	case pc.Vec3Slice: // pc/vec3slice.go:8-20: []mat.Vec3, 12-byte records
		if len(v) == 0 {
			return nil, 12, 0, nil, true
		}
		return unsafe.Pointer(&v[0]), 12, 0, v, true
	}
	return nil, 0, 0, nil, false
}

// New builds the tree (kdtree.New, kdtree.go:33-56).  A pc.Vec3Slice is uploaded straight from its own
// memory (NewFromPointCloud does the same for a cloud's records); any other Vec3RandomAccessor through one
// packed copy.
func New(ra pc.Vec3RandomAccessor, opts ...KDTreeOption) (*KDTree, error) {
	runtime.LockOSThread() // the error text is thread-local on the C side: call and pcgx_last_error on one OS thread
	defer runtime.UnlockOSThread()
	n := ra.Len()
	data, stride, off, keep, ok := cloudLayout(ra)
	if !ok {
		xyz := packVec3(ra)
		stride, off, keep = 12, 0, xyz
		if n > 0 {
			data = unsafe.Pointer(&xyz[0])
		}
	}
	t := &devTree{}
	rc := C.pcgx_kdtree_build(data, C.int64_t(n), C.int32_t(stride), C.int32_t(off), &t.h)
	runtime.KeepAlive(keep)
	if err := status(rc); err != nil {
		return nil, err
	}
	runtime.SetFinalizer(t, func(t *devTree) { t.release() })
	k := &KDTree{Vec3RandomAccessor: ra, t: t}
	for _, o := range opts {
		o(k)
	}
	return k, nil
}

// NewFromPointCloud builds the tree over a cloud's points straight from its records (Data, stride and xyz
// offset as PointCloud.Vec3Iterator finds them, pc/pointcloud.go:130-163): what kdtree.New(it) does for
// it, _ := pp.Vec3Iterator(), without reading the cloud point by point.  The tree's accessor is that iterator.
func NewFromPointCloud(pp *pc.PointCloud, opts ...KDTreeOption) (*KDTree, error) {
	runtime.LockOSThread()
	defer runtime.UnlockOSThread()
	it, err := pp.Vec3Iterator()
	if err != nil {
		return nil, err
	}
	stride, off, err := xyzLayout(pp)
	if err != nil {
		return nil, err
	}
	var data unsafe.Pointer
	if pp.Points > 0 {
		data = unsafe.Pointer(&pp.Data[0])
	}
	t := &devTree{}
	rc := C.pcgx_kdtree_build(data, C.int64_t(pp.Points), C.int32_t(stride), C.int32_t(off), &t.h)
	runtime.KeepAlive(pp)
	if err := status(rc); err != nil {
		return nil, err
	}
	runtime.SetFinalizer(t, func(t *devTree) { t.release() })
	k := &KDTree{Vec3RandomAccessor: it, t: t}
	for _, o := range opts {
		o(k)
	}
	return k, nil
}

// With creates a shallow copy of the tree with the options applied (kdtree.go:58-65); the copy shares
// the device tree.
func (k *KDTree) With(opts ...KDTreeOption) *KDTree {
	k2 := *k
	for _, o := range opts {
		o(&k2)
	}
	return &k2
}

func (t *devTree) release() {
	runtime.LockOSThread()
	defer runtime.UnlockOSThread()
	runtime.SetFinalizer(t, nil)
	if t.h != nil {
		C.pcgx_kdtree_free(t.h)
		t.h = nil
	}
}

// Close releases the device tree now (for every copy made by With) instead of at garbage collection.
func (k *KDTree) Close() {
	defer runtime.KeepAlive(k) // the finalizer must not free the handle while a call is in flight
	if k.t != nil {
		k.t.release()
	}
}

// DeletePoint removes a point from the tree (KDTree.DeletePoint, kdtree.go:322-332).
func (k *KDTree) DeletePoint(pID int) error {
	runtime.LockOSThread() // the error text is thread-local on the C side: call and pcgx_last_error on one OS thread
	defer runtime.UnlockOSThread()
	defer runtime.KeepAlive(k) // the finalizer must not free the handle while a call is in flight
	id := C.int64_t(pID)
	rc := C.pcgx_kdtree_delete_points(k.t.h, &id, 1)
	if rc == C.PCGX_E_OUT_OF_RANGE {
		return fmt.Errorf("%d does not correspond to any point in the tree", pID) // kdtree.go:324
	}
	return status(rc)
}

// NearestBatch is the batched seam: result i equals k.Nearest(q[i], maxRange).
func (k *KDTree) NearestBatch(q []mat.Vec3, maxRange float32) ([]storage.Neighbor, error) {
	runtime.LockOSThread() // the error text is thread-local on the C side: call and pcgx_last_error on one OS thread
	defer runtime.UnlockOSThread()
	defer runtime.KeepAlive(k) // the finalizer must not free the handle while a call is in flight
	n := len(q)
	out := make([]storage.Neighbor, n)
	if n == 0 {
		return out, nil
	}
	ids := make([]int64, n)
	dsq := make([]float32, n)
	rc := C.pcgx_kdtree_nearest_batch(k.t.h, (*C.float)(unsafe.Pointer(&q[0])), C.int64_t(n),
		C.float(maxRange), C.float(k.MinDistSq), (*C.int64_t)(unsafe.Pointer(&ids[0])), (*C.float)(unsafe.Pointer(&dsq[0])))
	if err := status(rc); err != nil {
		return nil, err
	}
	for i := range out {
		out[i] = storage.Neighbor{ID: int(ids[i]), DistSq: dsq[i]}
	}
	return out, nil
}

// Nearest keeps storage.Search working for single points (one tiny batch).
func (k *KDTree) Nearest(p mat.Vec3, maxRange float32) storage.Neighbor {
	atomic.AddInt64(&singlePointCalls, 1)
	r, err := k.NearestBatch([]mat.Vec3{p}, maxRange)
	if err != nil {
		panic(err)
	}
	return r[0]
}

// RangeBatch: neighbours with DistSq < maxRange^2 of every query, each list sorted by DistSq
// (KDTree.Range, kdtree.go:148-161).  out[i] belongs to q[i].
func (k *KDTree) RangeBatch(q []mat.Vec3, maxRange float32) ([][]storage.Neighbor, error) {
	runtime.LockOSThread() // the error text is thread-local on the C side: call and pcgx_last_error on one OS thread
	defer runtime.UnlockOSThread()
	defer runtime.KeepAlive(k) // the finalizer must not free the handle while a call is in flight
	n := len(q)
	out := make([][]storage.Neighbor, n)
	if n == 0 {
		return out, nil
	}
	counts := make([]int64, n)
	qp := (*C.float)(unsafe.Pointer(&q[0]))
	if err := status(C.pcgx_kdtree_range_count(k.t.h, qp, C.int64_t(n), C.float(maxRange),
		(*C.int64_t)(unsafe.Pointer(&counts[0])))); err != nil {
		return nil, err
	}
	offs := make([]int64, n+1)
	for i, c := range counts {
		offs[i+1] = offs[i] + c
	}
	total := offs[n]
	ids := make([]int64, total+1)
	dsq := make([]float32, total+1)
	if err := status(C.pcgx_kdtree_range_fill(k.t.h, qp, C.int64_t(n), C.float(maxRange),
		(*C.int64_t)(unsafe.Pointer(&offs[0])), (*C.int64_t)(unsafe.Pointer(&ids[0])),
		(*C.float)(unsafe.Pointer(&dsq[0])))); err != nil {
		return nil, err
	}
	for i := range out {
		nb := make([]storage.Neighbor, counts[i])
		for j := range nb {
			nb[j] = storage.Neighbor{ID: int(ids[offs[i]+int64(j)]), DistSq: dsq[offs[i]+int64(j)]}
		}
		out[i] = nb
	}
	return out, nil
}

// Range keeps storage.Search working for single points.
func (k *KDTree) Range(p mat.Vec3, maxRange float32) []storage.Neighbor {
	atomic.AddInt64(&singlePointCalls, 1)
	r, err := k.RangeBatch([]mat.Vec3{p}, maxRange)
	if err != nil {
		panic(err)
	}
	return r[0]
}

// -------------------------------------------------------------- VoxelGrid

type voxelGrid struct {
	leaf  mat.Vec3
	chunk [3]int
}

// NewVoxelGrid replaces voxelgrid.New(leaf, WithChunkSize(chunk))
// (pc/filter/voxelgrid/voxelgrid.go:23-33); chunk {0,0,0} = non-chunked.
func NewVoxelGrid(leaf mat.Vec3, chunk [3]int) filter.Filter {
	return &voxelGrid{leaf: leaf, chunk: chunk}
}

func (f *voxelGrid) Filter(pp *pc.PointCloud) (*pc.PointCloud, error) {
	runtime.LockOSThread() // the error text is thread-local on the C side: call and pcgx_last_error on one OS thread
	defer runtime.UnlockOSThread()
	defer runtime.KeepAlive(f) // the finalizer must not free the handle while a call is in flight
	stride, off, err := xyzLayout(pp)
	if err != nil {
		return nil, err
	}
	n := pp.Points
	if n == 0 {
		return nil, errors.New("no point")
	}
	out := make([]byte, n*stride)
	leaf := [3]C.float{C.float(f.leaf[0]), C.float(f.leaf[1]), C.float(f.leaf[2])}
	chunk := [3]C.int32_t{C.int32_t(f.chunk[0]), C.int32_t(f.chunk[1]), C.int32_t(f.chunk[2])}
	var m C.int64_t
	rc := C.pcgx_voxel_filter(unsafe.Pointer(&pp.Data[0]), C.int64_t(n), C.int32_t(stride), C.int32_t(off),
		&leaf[0], &chunk[0], unsafe.Pointer(&out[0]), &m)
	if err := status(rc); err != nil {
		return nil, err
	}
	newPc := &pc.PointCloud{PointCloudHeader: pp.Clone(), Points: int(m), Data: out[:int(m)*stride]}
	newPc.Width, newPc.Height = int(m), 1
	return newPc, nil
}

// FilterSharded is this rank's share of Filter(pp) over the ranks of c (SURVEY 8(e)): every rank
// passes the same cloud; the ranks' results, rank 0's first, are Filter's output record for record.
// Collective.
func FilterSharded(f filter.Filter, pp *pc.PointCloud, c *Comm) (*pc.PointCloud, error) {
	runtime.LockOSThread()
	defer runtime.UnlockOSThread()
	defer runtime.KeepAlive(c)
	vg, ok := f.(*voxelGrid)
	if !ok {
		return nil, errors.New("pcgx: FilterSharded needs a filter made by NewVoxelGrid")
	}
	stride, off, err := xyzLayout(pp)
	if err != nil {
		return nil, err
	}
	n := pp.Points
	if n == 0 {
		return nil, errors.New("no point")
	}
	out := make([]byte, n*stride)
	leaf := [3]C.float{C.float(vg.leaf[0]), C.float(vg.leaf[1]), C.float(vg.leaf[2])}
	chunk := [3]C.int32_t{C.int32_t(vg.chunk[0]), C.int32_t(vg.chunk[1]), C.int32_t(vg.chunk[2])}
	var m C.int64_t
	rc := C.pcgx_voxel_filter_sharded(c.h, unsafe.Pointer(&pp.Data[0]), C.int64_t(n), C.int32_t(stride), C.int32_t(off),
		&leaf[0], &chunk[0], unsafe.Pointer(&out[0]), &m)
	if err := status(rc); err != nil {
		return nil, err
	}
	part := &pc.PointCloud{PointCloudHeader: pp.Clone(), Points: int(m), Data: out[:int(m)*stride]}
	part.Width, part.Height = int(m), 1
	return part, nil
}

// -------------------------------------------------------------------- ICP

// Evaluator implements icp.Evaluator with the fused GPU correspondence +
// reduction (replaces PointToPointEvaluator + NearestPointCorresponder,
// evaluator.go:91-189, correspondence.go:22-37; default weight only).
type Evaluator struct {
	MaxDist  float32
	MinPairs int
	// Weight replaces PointToPointEvaluator.WeightFn (evaluator.go:72): the reference accepts any Go
	// closure, the device one of the built-in forms below.  Weight.Func() is the matching closure
	// for the CPU evaluator, so both compute the same float32 weights.  Zero value: w = 1.
	Weight WeightFn
	// Sums selects how the nine sums of evaluator.go:122-145 are formed (include/pcgx.h PCGX_SUMS_*).
	// Zero value SumsReference: the reference's own sequential float32 additions, evaluated exactly by
	// the whole GPU -- Evaluate and Fit return what the CPU code returns, bit for bit, at any size.
	// SumsF64Tree: a fixed-order float64 reduction of the same terms (about 3x faster per iteration at
	// 1M pairs; differs from the reference by the reference's own rounding noise, 1.6e-5 on the pose
	// at 1M pairs); it is what FitSharded computes when the target is spread over several ranks.
	Sums int
}

// Evaluator.Sums (include/pcgx.h PCGX_SUMS_*).
const (
	SumsReference      = 0
	SumsF64Tree        = 1
	SumsReferenceChain = 2 // one wave adding term after term: the on-device cross-check
)

// WeightFn kinds (include/pcgx.h PCGX_WEIGHT_*).
const (
	WeightOne      = 0 // DefaultEvaluateWeightFn
	WeightConstant = 1 // a
	WeightInverse  = 2 // 1 / (a + d)
	WeightHuber    = 3 // d <= a ? 1 : sqrt(a / d)
	WeightTukey    = 4 // d < a ? (1 - d/a)^2 : 0
)

type WeightFn struct {
	Kind int
	A    float32
}

// Func is the closure icp.PointToPointEvaluator{WeightFn: ...} takes for the same weights on the CPU.
func (w WeightFn) Func() icp.EvaluateWeightFn {
	a := w.A
	switch w.Kind {
	case WeightConstant:
		return func(float32) float32 { return a }
	case WeightInverse:
		return func(d float32) float32 { return 1 / (a + d) }
	case WeightHuber:
		return func(d float32) float32 {
			if d <= a {
				return 1
			}
			return float32(math.Sqrt(float64(a / d)))
		}
	case WeightTukey:
		return func(d float32) float32 {
			if !(d < a) {
				return 0
			}
			u := 1 - d/a
			return u * u
		}
	}
	return icp.DefaultEvaluateWeightFn
}

var _ icp.Evaluator = (*Evaluator)(nil)

func (Evaluator) HasGradient() bool { return true }
func (Evaluator) HasHessian() bool  { return false }

func packVec3(ra pc.Vec3RandomAccessor) []float32 {
	if s, ok := ra.(pc.Vec3Slice); ok && len(s) > 0 {
		return unsafe.Slice((*float32)(unsafe.Pointer(&s[0])), 3*len(s))
	}
	out := make([]float32, 3*ra.Len())
	for i := 0; i < ra.Len(); i++ {
		v := ra.Vec3At(i)
		copy(out[3*i:], v[:])
	}
	return out
}

func (e *Evaluator) Evaluate(base storage.Search, target pc.Vec3RandomAccessor) (*icp.Evaluated, error) {
	runtime.LockOSThread() // the error text is thread-local on the C side: call and pcgx_last_error on one OS thread
	defer runtime.UnlockOSThread()
	defer runtime.KeepAlive(e) // the finalizer must not free the handle while a call is in flight
	k, ok := base.(*KDTree)
	if !ok {
		return nil, errors.New("pcgx: base must be a *pcgx.KDTree")
	}
	t := packVec3(target)
	var tp *C.float
	if len(t) > 0 {
		tp = (*C.float)(unsafe.Pointer(&t[0]))
	}
	var ev C.pcgx_icp_evaluated
	var p C.pcgx_icp_params
	p.max_dist, p.min_dist_sq, p.min_pairs = C.float(e.MaxDist), C.float(k.MinDistSq), C.int32_t(e.MinPairs)
	p.weight_fn, p.weight_fn_param = C.int32_t(e.Weight.Kind), C.float(e.Weight.A)
	p.sums_mode = C.int32_t(e.Sums)
	rc := C.pcgx_icp_evaluate_params(k.t.h, tp, C.int64_t(target.Len()), &p, &ev)
	runtime.KeepAlive(k)
	if err := status(rc); err != nil {
		return nil, err
	}
	out := &icp.Evaluated{Value: float32(ev.value), DistRMS: float32(ev.dist_rms)}
	for i := 0; i < 6; i++ {
		out.Gradient[i] = float32(ev.gradient[i])
	}
	return out, nil
}

// Corresponder implements icp.PointToPointCorresponder (correspondence.go:14-16) over pcgx_icp_pairs: one batched
// nearest-neighbour pass for all targets, the pairs compacted in target order exactly as
// NearestPointCorresponder.Pairs emits them (correspondence.go:22-37).  The reference's own
// icp.PointToPointEvaluator{Corresponder: &pcgx.Corresponder{MaxDist: d}} then runs its CPU reduction (and any Go
// WeightFn closure) over GPU correspondences.  A base that is not a *pcgx.KDTree is answered by the reference's
// loop over base.Nearest.
type Corresponder struct {
	MaxDist float32
}

var _ icp.PointToPointCorresponder = (*Corresponder)(nil)

func (c *Corresponder) Pairs(base storage.Search, target pc.Vec3RandomAccessor) []icp.PointToPointCorrespondence {
	k, ok := base.(*KDTree)
	if !ok {
		return (&icp.NearestPointCorresponder{MaxDist: c.MaxDist}).Pairs(base, target)
	}
	runtime.LockOSThread()
	defer runtime.UnlockOSThread()
	defer runtime.KeepAlive(k)
	n := target.Len()
	if n == 0 {
		return []icp.PointToPointCorrespondence{}
	}
	t := packVec3(target)
	baseID := make([]int64, n)
	targetID := make([]int64, n)
	dsq := make([]float32, n)
	var np C.int64_t
	rc := C.pcgx_icp_pairs(k.t.h, (*C.float)(unsafe.Pointer(&t[0])), C.int64_t(n), C.float(c.MaxDist), C.float(k.MinDistSq),
		(*C.int64_t)(unsafe.Pointer(&baseID[0])), (*C.int64_t)(unsafe.Pointer(&targetID[0])), (*C.float)(unsafe.Pointer(&dsq[0])), &np)
	if err := status(rc); err != nil {
		panic(err) // the interface has no error result; the reference's loop cannot fail either
	}
	out := make([]icp.PointToPointCorrespondence, int(np))
	for i := range out {
		out[i] = icp.PointToPointCorrespondence{BaseID: int(baseID[i]), TargetID: int(targetID[i]), SquaredDistance: dsq[i]}
	}
	return out
}

// Fit runs the whole PointToPointICPGradient.Fit loop on the device
// (icp.go:23-67) with a GradientDescentUpdaterFactory's parameters.
func Fit(base *KDTree, target pc.Vec3RandomAccessor, e *Evaluator, u *icp.GradientDescentUpdaterFactory) (mat.Mat4, icp.Stat, error) {
	runtime.LockOSThread() // the error text is thread-local on the C side: call and pcgx_last_error on one OS thread
	defer runtime.UnlockOSThread()
	defer runtime.KeepAlive(base)
	var p C.pcgx_icp_params
	p.max_dist, p.min_dist_sq, p.min_pairs = C.float(e.MaxDist), C.float(base.MinDistSq), C.int32_t(e.MinPairs)
	p.weight_fn, p.weight_fn_param = C.int32_t(e.Weight.Kind), C.float(e.Weight.A)
	p.sums_mode = C.int32_t(e.Sums)
	if u != nil {
		for i := 0; i < 6; i++ {
			p.weight[i], p.threshold[i] = C.float(u.Weight[i]), C.float(u.Threshold[i])
		}
		p.max_iteration = C.int32_t(u.MaxIteration)
	}
	t := packVec3(target)
	var tp *C.float
	if len(t) > 0 {
		tp = (*C.float)(unsafe.Pointer(&t[0]))
	}
	var trans mat.Mat4
	var st C.pcgx_icp_stat
	rc := C.pcgx_icp_fit(base.t.h, tp, C.int64_t(target.Len()), &p, (*C.float)(unsafe.Pointer(&trans[0])), &st)
	stat := icp.Stat{NumIteration: int(st.num_iteration)}
	stat.Value, stat.DistRMS = float32(st.evaluated.value), float32(st.evaluated.dist_rms)
	for i := 0; i < 6; i++ {
		stat.Gradient[i] = float32(st.evaluated.gradient[i])
	}
	return trans, stat, status(rc)
}

// ------------------------------------------ point-to-plane ICP (extension)

// ErrSingular: the 6x6 normal equations are not positive definite.
var ErrSingular = errors.New("pcgx: normal equations are not positive definite")

// PlaneEvaluator fills the slots the reference leaves empty (Evaluated.Hessian,
// HasHessian(), evaluator.go:28,35,76): point-to-plane residual r = n.(pt-pb),
// J = {n, pt x n}, Hessian = 2/sum(w) * sum J J^T.  BaseNormals holds one unit
// normal per base point in the tree's id order.  No counterpart in the reference.
type PlaneEvaluator struct {
	MaxDist     float32
	MinPairs    int
	BaseNormals []mat.Vec3
}

var _ icp.Evaluator = (*PlaneEvaluator)(nil)

func (PlaneEvaluator) HasGradient() bool { return true }
func (PlaneEvaluator) HasHessian() bool  { return true }

// Evaluate runs one fused correspondence + 30-sum reduction on the device.
func (e *PlaneEvaluator) Evaluate(base storage.Search, target pc.Vec3RandomAccessor) (*icp.Evaluated, error) {
	runtime.LockOSThread() // the error text is thread-local on the C side: call and pcgx_last_error on one OS thread
	defer runtime.UnlockOSThread()
	defer runtime.KeepAlive(e) // the finalizer must not free the handle while a call is in flight
	k, ok := base.(*KDTree)
	if !ok {
		return nil, errors.New("pcgx: base must be a *pcgx.KDTree")
	}
	if len(e.BaseNormals) != k.Len() || len(e.BaseNormals) == 0 {
		return nil, errors.New("pcgx: one normal per base point is required")
	}
	t := packVec3(target)
	var tp *C.float
	if len(t) > 0 {
		tp = (*C.float)(unsafe.Pointer(&t[0]))
	}
	var p C.pcgx_icp_params
	p.max_dist, p.min_pairs = C.float(e.MaxDist), C.int32_t(e.MinPairs)
	var s *C.pcgx_icp_session
	rc := C.pcgx_icp_plane_session_create(k.t.h, (*C.float)(unsafe.Pointer(&e.BaseNormals[0])), tp,
		C.int64_t(target.Len()), 0, &p, 0, nil, &s)
	if err := status(rc); err != nil {
		return nil, err
	}
	defer C.pcgx_icp_session_free(s)
	if err := status(C.pcgx_icp_session_partials(s, nil)); err != nil {
		return nil, err
	}
	var sums [30]C.double
	if err := status(C.pcgx_icp_session_read_sums_n(s, &sums[0], 30, nil)); err != nil {
		return nil, err
	}
	var ev C.pcgx_icp_evaluated
	out := &icp.Evaluated{}
	rc = C.pcgx_icp_plane_finish_evaluate(&sums[0], C.int32_t(e.MinPairs), &ev, (*C.float)(unsafe.Pointer(&out.Hessian[0])))
	if err := status(rc); err != nil {
		return nil, err
	}
	out.Value = float32(ev.value)
	for i := 0; i < 6; i++ {
		out.Gradient[i] = float32(ev.gradient[i])
	}
	return out, nil
}

// FitPlane runs the whole loop (icp.go:23-67 shape) with the point-to-plane
// evaluator and a Gauss-Newton updater on the device.
func FitPlane(base *KDTree, target pc.Vec3RandomAccessor, e *PlaneEvaluator, threshold mat.Vec6, maxIteration int, damping float32) (mat.Mat4, icp.Stat, error) {
	runtime.LockOSThread() // the error text is thread-local on the C side: call and pcgx_last_error on one OS thread
	defer runtime.UnlockOSThread()
	defer runtime.KeepAlive(base)
	var p C.pcgx_icp_params
	p.max_dist, p.min_pairs, p.max_iteration = C.float(e.MaxDist), C.int32_t(e.MinPairs), C.int32_t(maxIteration)
	for i := 0; i < 6; i++ {
		p.threshold[i] = C.float(threshold[i])
	}
	t := packVec3(target)
	var tp *C.float
	if len(t) > 0 {
		tp = (*C.float)(unsafe.Pointer(&t[0]))
	}
	var trans mat.Mat4
	var st C.pcgx_icp_stat
	stat := icp.Stat{}
	if len(e.BaseNormals) != base.Len() || len(e.BaseNormals) == 0 {
		return trans, stat, errors.New("pcgx: one normal per base point is required")
	}
	rc := C.pcgx_icp_plane_fit(base.t.h, (*C.float)(unsafe.Pointer(&e.BaseNormals[0])), tp, C.int64_t(target.Len()), &p,
		C.float(damping), (*C.float)(unsafe.Pointer(&trans[0])), &st, (*C.float)(unsafe.Pointer(&stat.Hessian[0])))
	stat.NumIteration = int(st.num_iteration)
	stat.Value = float32(st.evaluated.value)
	for i := 0; i < 6; i++ {
		stat.Gradient[i] = float32(st.evaluated.gradient[i])
	}
	if rc == C.PCGX_E_SINGULAR {
		return trans, stat, ErrSingular
	}
	return trans, stat, status(rc)
}

// ------------------------------------------------- bucket grid, segmentation

// BucketGrid is pc/storage/voxelgrid.VoxelGrid filled with Add(point i, i) for
// a whole cloud (voxelgrid.go:15-23,37-45), with pc/segmentation/voxelgrid's
// Segment (voxelgrid.go:39-73) answered from the connected components the
// device computes for the whole grid.
type BucketGrid struct {
	h *C.pcgx_bucket_grid
}

// NewBucketGrid builds the grid over every point of ra.
func NewBucketGrid(resolution float32, size [3]int, origin mat.Vec3, ra pc.Vec3RandomAccessor) (*BucketGrid, error) {
	runtime.LockOSThread() // the error text is thread-local on the C side: call and pcgx_last_error on one OS thread
	defer runtime.UnlockOSThread()
	xyz := packVec3(ra)
	var data unsafe.Pointer
	if len(xyz) > 0 {
		data = unsafe.Pointer(&xyz[0])
	}
	sz := [3]C.int64_t{C.int64_t(size[0]), C.int64_t(size[1]), C.int64_t(size[2])}
	g := &BucketGrid{}
	rc := C.pcgx_bucket_grid_build(data, C.int64_t(ra.Len()), 12, 0, C.float(resolution), &sz[0],
		(*C.float)(unsafe.Pointer(&origin[0])), &g.h)
	if err := status(rc); err != nil {
		return nil, err
	}
	runtime.SetFinalizer(g, func(g *BucketGrid) { g.Close() })
	return g, nil
}

// Close releases the grid.
func (g *BucketGrid) Close() {
	runtime.LockOSThread() // the error text is thread-local on the C side: call and pcgx_last_error on one OS thread
	defer runtime.UnlockOSThread()
	defer runtime.KeepAlive(g) // the finalizer must not free the handle while a call is in flight
	if g.h != nil {
		C.pcgx_bucket_grid_free(g.h)
		g.h = nil
	}
}

func idsToInt(ids []int64) []int {
	out := make([]int, len(ids))
	for i, v := range ids {
		out[i] = int(v)
	}
	return out
}

// Get returns the ids of p's voxel, nil when p is outside the grid (voxelgrid.go:52-58).
func (g *BucketGrid) Get(p mat.Vec3) []int {
	runtime.LockOSThread() // the error text is thread-local on the C side: call and pcgx_last_error on one OS thread
	defer runtime.UnlockOSThread()
	defer runtime.KeepAlive(g) // the finalizer must not free the handle while a call is in flight
	var cnt C.int64_t
	if C.pcgx_bucket_grid_get(g.h, (*C.float)(unsafe.Pointer(&p[0])), nil, 0, &cnt) != C.PCGX_OK || cnt < 0 {
		return nil
	}
	ids := make([]int64, int(cnt)+1)
	C.pcgx_bucket_grid_get(g.h, (*C.float)(unsafe.Pointer(&p[0])), (*C.int64_t)(unsafe.Pointer(&ids[0])), cnt, &cnt)
	return idsToInt(ids[:int(cnt)])
}

// Segment is segmentation/voxelgrid.VoxelGrid.Segment, ids in the reference's own order.
func (g *BucketGrid) Segment(p mat.Vec3) []int {
	runtime.LockOSThread() // the error text is thread-local on the C side: call and pcgx_last_error on one OS thread
	defer runtime.UnlockOSThread()
	defer runtime.KeepAlive(g) // the finalizer must not free the handle while a call is in flight
	var cnt C.int64_t
	if C.pcgx_bucket_grid_segment_bfs(g.h, (*C.float)(unsafe.Pointer(&p[0])), nil, 0, &cnt) != C.PCGX_OK || cnt <= 0 {
		return nil
	}
	ids := make([]int64, int(cnt))
	C.pcgx_bucket_grid_segment_bfs(g.h, (*C.float)(unsafe.Pointer(&p[0])), (*C.int64_t)(unsafe.Pointer(&ids[0])), cnt, &cnt)
	return idsToInt(ids[:int(cnt)])
}

// RegionGrowing is pc/segmentation/regiongrowing.RegionGrowing (regiongrowing.go:13-56) with the
// regions of the whole cloud labelled on the device once per maxRange.
type RegionGrowing struct {
	search   *KDTree
	labels   []uint32
	comp     []int64
	maxRange float32
}

// NewRegionGrowing mirrors regiongrowing.New(search, propertyIter).
func NewRegionGrowing(search *KDTree, propertyIter pc.Uint32RandomAccessor) *RegionGrowing {
	labels := make([]uint32, search.Len())
	for i := range labels {
		labels[i] = propertyIter.Uint32At(i)
	}
	return &RegionGrowing{search: search, labels: labels}
}

// Segment mirrors RegionGrowing.Segment, ids in the reference's own order.
func (r *RegionGrowing) Segment(p mat.Vec3, maxRange float32) []int {
	runtime.LockOSThread() // the error text is thread-local on the C side: call and pcgx_last_error on one OS thread
	defer runtime.UnlockOSThread()
	defer runtime.KeepAlive(r) // the finalizer must not free the handle while a call is in flight
	n := len(r.labels)
	if n == 0 {
		return []int{}
	}
	ids := make([]int64, n)
	var cnt C.int64_t
	if C.pcgx_region_growing_segment_bfs(r.search.t.h, (*C.uint32_t)(unsafe.Pointer(&r.labels[0])),
		(*C.float)(unsafe.Pointer(&p[0])), C.float(maxRange), (*C.int64_t)(unsafe.Pointer(&ids[0])), C.int64_t(n),
		&cnt) != C.PCGX_OK {
		return []int{}
	}
	return idsToInt(ids[:int(cnt)])
}

// SegmentByID returns the same set in ascending id order from region labels of the whole cloud
// (computed once per maxRange): the fast path for many seeds.
func (r *RegionGrowing) SegmentByID(p mat.Vec3, maxRange float32) []int {
	runtime.LockOSThread() // the error text is thread-local on the C side: call and pcgx_last_error on one OS thread
	defer runtime.UnlockOSThread()
	defer runtime.KeepAlive(r) // the finalizer must not free the handle while a call is in flight
	n := len(r.labels)
	if n == 0 {
		return []int{}
	}
	if r.comp == nil || r.maxRange != maxRange {
		comp := make([]int64, n)
		if C.pcgx_region_growing_components(r.search.t.h, (*C.uint32_t)(unsafe.Pointer(&r.labels[0])), C.float(maxRange),
			(*C.int64_t)(unsafe.Pointer(&comp[0]))) != C.PCGX_OK {
			return []int{}
		}
		r.comp, r.maxRange = comp, maxRange
	}
	ids := make([]int64, n)
	var cnt C.int64_t
	if C.pcgx_region_growing_segment(r.search.t.h, (*C.uint32_t)(unsafe.Pointer(&r.labels[0])),
		(*C.int64_t)(unsafe.Pointer(&r.comp[0])), (*C.float)(unsafe.Pointer(&p[0])), C.float(maxRange),
		(*C.int64_t)(unsafe.Pointer(&ids[0])), C.int64_t(n), &cnt) != C.PCGX_OK {
		return []int{}
	}
	return idsToInt(ids[:int(cnt)])
}


// ------------------------------------------ strict sums and the sharded Fit

// FitStrict is Fit with Evaluator.Sums forced to SumsReference, whatever e.Sums says: the evaluator's
// sums formed exactly as the Go code forms them (sequential float32 additions in target order,
// evaluator.go:122-145), so the returned transform and Stat are bit-identical to
// PointToPointICPGradient.Fit on the CPU at any size.  Since SumsReference is the zero value, Fit
// already does this for an Evaluator that does not ask for anything else; the name is kept for callers
// written against the earlier shim, where the float64 reduction was the default.
func FitStrict(base *KDTree, target pc.Vec3RandomAccessor, e *Evaluator, u *icp.GradientDescentUpdaterFactory) (mat.Mat4, icp.Stat, error) {
	strict := *e
	strict.Sums = SumsReference
	return Fit(base, target, &strict, u)
}

// Comm is the exchange of the sharded Fit: one process per GPU, every rank with a replica of the
// base tree and one spatial tile of the target; per iteration the ten float64 partial sums are
// all-reduced over the ranks (RCCL over xGMI, bound by libpcgx.so at run time).
type Comm struct{ h *C.pcgx_comm }

// CommID is generated by rank 0 (NewCommID) and handed to every rank by any channel the host has.
type CommID [128]byte

func NewCommID() (CommID, error) {
	runtime.LockOSThread()
	defer runtime.UnlockOSThread()
	var id CommID
	err := status(C.pcgx_comm_unique_id((*C.pcgx_comm_id)(unsafe.Pointer(&id[0]))))
	return id, err
}

// NewComm is collective: every rank calls it with the same id.
func NewComm(rank, world int, id CommID) (*Comm, error) {
	runtime.LockOSThread()
	defer runtime.UnlockOSThread()
	c := &Comm{}
	if err := status(C.pcgx_comm_init(C.int32_t(rank), C.int32_t(world), (*C.pcgx_comm_id)(unsafe.Pointer(&id[0])), &c.h)); err != nil {
		return nil, err
	}
	return c, nil
}

func (c *Comm) Close() {
	if c.h != nil {
		C.pcgx_comm_free(c.h)
		c.h = nil
	}
}

// InitDevices: one process drives several GPUs (SURVEY 8(b): the reference is one process, icp.go:23).  Slot k of
// the library works on HIP device ids[k] (nil: device k).  A goroutine names the slot its calls are for with
// SetDevice -- after runtime.LockOSThread: the slot is per OS thread, like HIP's current device -- and a tree
// belongs to the slot it was built on.  FitMulti needs none of that from the caller.
func InitDevices(n int, ids []int32) error {
	runtime.LockOSThread()
	defer runtime.UnlockOSThread()
	var p *C.int32_t
	if len(ids) > 0 {
		p = (*C.int32_t)(unsafe.Pointer(&ids[0]))
	}
	return status(C.pcgx_init_devices(C.int32_t(n), p))
}

// SetDevice selects the calling OS thread's device slot (call runtime.LockOSThread first).
func SetDevice(slot int) error { return status(C.pcgx_set_device(C.int32_t(slot))) }

// NewReplicas builds the same tree on the first n device slots (one upload and build per slot).
func NewReplicas(ra pc.Vec3RandomAccessor, n int, opts ...KDTreeOption) ([]*KDTree, error) {
	runtime.LockOSThread()
	defer runtime.UnlockOSThread()
	defer C.pcgx_set_device(0)
	out := make([]*KDTree, 0, n)
	for r := 0; r < n; r++ {
		if err := SetDevice(r); err != nil {
			return nil, err
		}
		k, err := New(ra, opts...)
		if err != nil {
			return nil, err
		}
		out = append(out, k)
	}
	return out, nil
}

// FitMulti is Fit with the target spread over the device slots of THIS process: bases[r] is the replica on slot r
// (NewReplicas), tiles[r] slot r's part of the target.  With the default sums (e.Sums == 0) the result is the
// reference's Fit of the tiles one after the other, bit for bit (icp.go:23-67, evaluator.go:122-145); no second
// process, no communicator to set up.  A slot whose step fails ends the Fit on every slot (the error names it).
func FitMulti(bases []*KDTree, tiles []pc.Vec3RandomAccessor, e *Evaluator, u *icp.GradientDescentUpdaterFactory) (mat.Mat4, icp.Stat, error) {
	runtime.LockOSThread()
	defer runtime.UnlockOSThread()
	n := len(bases)
	if n == 0 || len(tiles) != n {
		return mat.Mat4{}, icp.Stat{}, errors.New("pcgx: FitMulti needs one tile per base replica")
	}
	var p C.pcgx_icp_params
	p.max_dist, p.min_dist_sq, p.min_pairs = C.float(e.MaxDist), C.float(bases[0].MinDistSq), C.int32_t(e.MinPairs)
	p.weight_fn, p.weight_fn_param = C.int32_t(e.Weight.Kind), C.float(e.Weight.A)
	p.sums_mode = C.int32_t(e.Sums)
	if u != nil {
		for i := 0; i < 6; i++ {
			p.weight[i], p.threshold[i] = C.float(u.Weight[i]), C.float(u.Threshold[i])
		}
		p.max_iteration = C.int32_t(u.MaxIteration)
	}
	// the C side reads the arrays of pointers during the call only; they live in C memory (cgo: no Go pointers to Go pointers)
	hb := (*[1 << 20]*C.pcgx_kdtree)(C.malloc(C.size_t(n) * C.size_t(unsafe.Sizeof(uintptr(0)))))
	ht := (*[1 << 20]*C.float)(C.malloc(C.size_t(n) * C.size_t(unsafe.Sizeof(uintptr(0)))))
	hn := (*[1 << 20]C.int64_t)(C.malloc(C.size_t(n) * 8))
	defer C.free(unsafe.Pointer(hb))
	defer C.free(unsafe.Pointer(ht))
	defer C.free(unsafe.Pointer(hn))
	packed := make([][]float32, n)
	var pin runtime.Pinner // the tiles' memory is referenced from C memory for the length of the call
	defer pin.Unpin()
	for r := 0; r < n; r++ {
		packed[r] = packVec3(tiles[r])
		hb[r] = bases[r].t.h
		hn[r] = C.int64_t(tiles[r].Len())
		ht[r] = nil
		if len(packed[r]) > 0 {
			pin.Pin(&packed[r][0])
			ht[r] = (*C.float)(unsafe.Pointer(&packed[r][0]))
		}
	}
	var trans mat.Mat4
	var st C.pcgx_icp_stat
	rc := C.pcgx_icp_fit_multi(C.int32_t(n), (**C.pcgx_kdtree)(unsafe.Pointer(hb)), (**C.float)(unsafe.Pointer(ht)),
		(*C.int64_t)(unsafe.Pointer(hn)), &p, (*C.float)(unsafe.Pointer(&trans[0])), &st)
	runtime.KeepAlive(bases)
	runtime.KeepAlive(packed)
	stat := icp.Stat{NumIteration: int(st.num_iteration)}
	stat.Value, stat.DistRMS = float32(st.evaluated.value), float32(st.evaluated.dist_rms)
	for i := 0; i < 6; i++ {
		stat.Gradient[i] = float32(st.evaluated.gradient[i])
	}
	return trans, stat, status(rc)
}

// FitSharded runs Fit on this rank's tile of the target; every rank returns the same transform.
// With the default sums (e.Sums == 0) that is the reference's Fit of the ranks' tiles one after the other, rank 0's
// first, bit for bit (its sequential float32 sums go round the ranks); SumsF64Tree is one all-reduce of ten float64
// per iteration.  A communicator of one rank is Fit.
func FitSharded(base *KDTree, tile pc.Vec3RandomAccessor, e *Evaluator, u *icp.GradientDescentUpdaterFactory, c *Comm) (mat.Mat4, icp.Stat, error) {
	runtime.LockOSThread()
	defer runtime.UnlockOSThread()
	defer runtime.KeepAlive(base)
	defer runtime.KeepAlive(c)
	var p C.pcgx_icp_params
	p.max_dist, p.min_dist_sq, p.min_pairs = C.float(e.MaxDist), C.float(base.MinDistSq), C.int32_t(e.MinPairs)
	p.weight_fn, p.weight_fn_param = C.int32_t(e.Weight.Kind), C.float(e.Weight.A)
	p.sums_mode = C.int32_t(e.Sums)
	if u != nil {
		for i := 0; i < 6; i++ {
			p.weight[i], p.threshold[i] = C.float(u.Weight[i]), C.float(u.Threshold[i])
		}
		p.max_iteration = C.int32_t(u.MaxIteration)
	}
	t := packVec3(tile)
	var tp *C.float
	if len(t) > 0 {
		tp = (*C.float)(unsafe.Pointer(&t[0]))
	}
	var trans mat.Mat4
	var st C.pcgx_icp_stat
	rc := C.pcgx_icp_fit_sharded(base.t.h, tp, C.int64_t(tile.Len()), &p, c.h, (*C.float)(unsafe.Pointer(&trans[0])), &st)
	stat := icp.Stat{NumIteration: int(st.num_iteration)}
	stat.Value, stat.DistRMS = float32(st.evaluated.value), float32(st.evaluated.dist_rms)
	for i := 0; i < 6; i++ {
		stat.Gradient[i] = float32(st.evaluated.gradient[i])
	}
	return trans, stat, status(rc)
}
