// Package icp is the GPU drop-in for github.com/seqsense/pcgol/pc/registration/icp: every exported name of the
// reference package exists here with the reference's signature.  Value types, interfaces, sentinel errors and the
// updater are the reference's own (aliases: values pass between the two packages unconverted); the three types that
// do the work -- NearestPointCorresponder (correspondence.go:18-37), PointToPointEvaluator (evaluator.go:69-189),
// PointToPointICPGradient (icp.go:18-67) -- run on the device when the base is a GPU tree
// (gpu/pc/storage/kdtree) and on the reference's own code otherwise.
//
// NOT compiled in the build image (no Go toolchain there); see go/README.md.
package icp

import (
	"github.com/seqsense/pcgol/mat"
	"github.com/seqsense/pcgol/pc"
	ref "github.com/seqsense/pcgol/pc/registration/icp"
	"github.com/seqsense/pcgol/pc/storage"

	"github.com/seqsense/pcgol/gpu/pcgx"
)

// The reference's own types under their own names (correspondence.go:8-16, evaluator.go:19-36, updater.go:7-22,
// stat.go:3-6).
type (
	PointToPointCorrespondence    = ref.PointToPointCorrespondence
	PointToPointCorresponder      = ref.PointToPointCorresponder
	EvaluateWeightFn              = ref.EvaluateWeightFn
	Evaluated                     = ref.Evaluated
	Evaluator                     = ref.Evaluator
	UpdaterGradientFactory        = ref.UpdaterGradientFactory
	UpdaterGradient               = ref.UpdaterGradient
	GradientDescentUpdaterFactory = ref.GradientDescentUpdaterFactory
	Stat                          = ref.Stat
)

// The reference's sentinels and defaults (evaluator.go:15-23, icp.go:14-16, updater.go:15-16): errors.Is works
// across both packages.
var (
	ErrNotEnoughPairs        = ref.ErrNotEnoughPairs
	ErrNeedGradient          = ref.ErrNeedGradient
	DefaultEvaluateWeightFn  = ref.DefaultEvaluateWeightFn
	DefaultGradientWeight    = ref.DefaultGradientWeight
	DefaultGradientThreshold = ref.DefaultGradientThreshold
)

// NearestPointCorresponder is icp.NearestPointCorresponder (correspondence.go:18-37): {MaxDist float32}, Pairs in
// ascending TargetID with unmatched targets dropped.  On a GPU tree: one batched nearest-neighbour pass; on any
// other storage.Search: the reference's loop over base.Nearest.
type NearestPointCorresponder = pcgx.Corresponder

// Weight names one of the weight forms the device evaluates (PointToPointEvaluator.Weight below).
type Weight = pcgx.WeightFn

// Weight.Kind and PointToPointEvaluator.Sums (include/pcgx.h PCGX_WEIGHT_*, PCGX_SUMS_*).
const (
	WeightOne      = pcgx.WeightOne
	WeightConstant = pcgx.WeightConstant
	WeightInverse  = pcgx.WeightInverse
	WeightHuber    = pcgx.WeightHuber
	WeightTukey    = pcgx.WeightTukey

	SumsReference = pcgx.SumsReference
	SumsF64Tree   = pcgx.SumsF64Tree
)

// PointToPointEvaluator is icp.PointToPointEvaluator (evaluator.go:69-73): Corresponder, MinPairs, WeightFn mean what
// they mean there.  The two further fields are optional (a keyed literal written for the reference compiles
// unchanged): Weight is a device-side weight form, used when WeightFn is nil; Sums selects how the nine sums are
// formed (zero value: the reference's sequential float32 additions, bit for bit).
//
// Evaluate runs fused on the device -- correspondence, the nine sums, the evaluate tail -- when base is a GPU tree,
// Corresponder a *NearestPointCorresponder and WeightFn nil.  Any other Corresponder, or a WeightFn closure (which
// cannot cross to the GPU), takes the reference's reduction (evaluator.go:91-189) over whatever the Corresponder
// returns -- GPU correspondences when it is a *NearestPointCorresponder on a GPU tree.
type PointToPointEvaluator struct {
	Corresponder PointToPointCorresponder
	MinPairs     int
	WeightFn     EvaluateWeightFn

	Weight Weight
	Sums   int
}

func (PointToPointEvaluator) HasGradient() bool { return true }
func (PointToPointEvaluator) HasHessian() bool  { return false }

var _ Evaluator = (*PointToPointEvaluator)(nil)

// device says whether Evaluate / Fit can run fused on the GPU, and with what.
func (e *PointToPointEvaluator) device(base storage.Search) (*pcgx.KDTree, *pcgx.Evaluator, bool) {
	k, ok := base.(*pcgx.KDTree)
	if !ok || e.WeightFn != nil {
		return nil, nil, false
	}
	c, ok := e.Corresponder.(*NearestPointCorresponder)
	if !ok || c == nil {
		return nil, nil, false
	}
	return k, &pcgx.Evaluator{MaxDist: c.MaxDist, MinPairs: e.MinPairs, Weight: e.Weight, Sums: e.Sums}, true
}

// cpu is the reference's evaluator with this one's settings.
func (e *PointToPointEvaluator) cpu() *ref.PointToPointEvaluator {
	w := e.WeightFn
	if w == nil && e.Weight.Kind != WeightOne {
		w = e.Weight.Func()
	}
	return &ref.PointToPointEvaluator{Corresponder: e.Corresponder, MinPairs: e.MinPairs, WeightFn: w}
}

// Evaluate is PointToPointEvaluator.Evaluate (evaluator.go:91-189).
func (e *PointToPointEvaluator) Evaluate(base storage.Search, target pc.Vec3RandomAccessor) (*Evaluated, error) {
	if k, d, ok := e.device(base); ok {
		return d.Evaluate(k, target)
	}
	return e.cpu().Evaluate(base, target)
}

// PointToPointICPGradient is icp.PointToPointICPGradient (icp.go:18-21).
type PointToPointICPGradient struct {
	Evaluator      Evaluator
	UpdaterFactory UpdaterGradientFactory
}

// Fit is PointToPointICPGradient.Fit (icp.go:23-67).  With a *PointToPointEvaluator that can run on the device (see
// there) and the reference's gradient-descent updater (nil, GradientDescentUpdaterFactory or a pointer to one) the
// whole loop -- Evaluate, Update, re-projection of the original target -- stays on the GPU and returns the
// reference's transform and Stat bit for bit; with any other Evaluator or UpdaterFactory the reference's loop runs,
// calling them as given.
func (r *PointToPointICPGradient) Fit(base storage.Search, target pc.Vec3RandomAccessor) (mat.Mat4, Stat, error) {
	if r.Evaluator == nil || !r.Evaluator.HasGradient() {
		return mat.Mat4{}, Stat{}, ErrNeedGradient // icp.go:24-26
	}
	if e, ok := r.Evaluator.(*PointToPointEvaluator); ok {
		if k, d, ok := e.device(base); ok {
			switch u := r.UpdaterFactory.(type) {
			case nil:
				return pcgx.Fit(k, target, d, nil)
			case GradientDescentUpdaterFactory:
				return pcgx.Fit(k, target, d, &u)
			case *GradientDescentUpdaterFactory:
				return pcgx.Fit(k, target, d, u)
			}
		}
	}
	return (&ref.PointToPointICPGradient{Evaluator: r.Evaluator, UpdaterFactory: r.UpdaterFactory}).Fit(base, target)
}

// FitSharded is Fit on this rank's tile of the target (one process per GPU, every rank with a replica of the base
// tree): with the default sums the reference's Fit of the ranks' tiles one after the other, rank 0's first, bit for
// bit.  FitMulti is the same for ONE process that drives several GPUs (pcgx.InitDevices, pcgx.NewReplicas).
func (r *PointToPointICPGradient) FitSharded(base *pcgx.KDTree, tile pc.Vec3RandomAccessor, c *pcgx.Comm) (mat.Mat4, Stat, error) {
	e, ok := r.Evaluator.(*PointToPointEvaluator)
	if !ok {
		return mat.Mat4{}, Stat{}, ErrNeedGradient
	}
	_, d, ok := e.device(base)
	if !ok {
		return mat.Mat4{}, Stat{}, pcgx.ErrNeedsDevice
	}
	return pcgx.FitSharded(base, tile, d, r.updater(), c)
}

func (r *PointToPointICPGradient) FitMulti(bases []*pcgx.KDTree, tiles []pc.Vec3RandomAccessor) (mat.Mat4, Stat, error) {
	e, ok := r.Evaluator.(*PointToPointEvaluator)
	if !ok || len(bases) == 0 {
		return mat.Mat4{}, Stat{}, ErrNeedGradient
	}
	_, d, ok := e.device(bases[0])
	if !ok {
		return mat.Mat4{}, Stat{}, pcgx.ErrNeedsDevice
	}
	return pcgx.FitMulti(bases, tiles, d, r.updater())
}

func (r *PointToPointICPGradient) updater() *GradientDescentUpdaterFactory {
	switch u := r.UpdaterFactory.(type) {
	case GradientDescentUpdaterFactory:
		return &u
	case *GradientDescentUpdaterFactory:
		return u
	}
	return nil
}
