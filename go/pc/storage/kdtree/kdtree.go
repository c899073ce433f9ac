// Package kdtree is the GPU drop-in for github.com/seqsense/pcgol/pc/storage/kdtree: the same exported names and
// signatures (kdtree.go:14-65,83,148,322), every call answered by libpcgx.so through the cgo package ../../../pcgx.
//
// NOT compiled in the build image (no Go toolchain there); see go/README.md.
package kdtree

import (
	"github.com/seqsense/pcgol/pc"
	"github.com/seqsense/pcgol/pc/storage"

	"github.com/seqsense/pcgol/gpu/pcgx"
)

// KDTree is the reference's type by name and by use (kdtree.go:14-23): it embeds the accessor it was built over,
// exports MinDistSq (> 0: the reference's approximate search, kdtree.go:20-22) and implements storage.Search --
// Nearest (kdtree.go:83), Range (kdtree.go:148) -- plus DeletePoint (kdtree.go:322) and With (kdtree.go:58-65).
// Nearest and Range for ONE point are a blocking GPU call each: code that loops over them (correspondence.go:25-36,
// regiongrowing.go:26,47) should take the batch seams instead -- NearestBatch / RangeBatch here, or the icp and
// segmentation types of this tree, which do.  pcgx.SinglePointCalls() counts such calls (tests guard hot loops with it).
type KDTree = pcgx.KDTree

// KDTreeOption is kdtree.KDTreeOption (kdtree.go:31).
type KDTreeOption = pcgx.KDTreeOption

var _ storage.Search = (*KDTree)(nil)

// New is kdtree.New (kdtree.go:33-56): the same tree (median split, dim = depth % 3, ties by position), built on the
// device.  Like the reference it has no error result; where the reference panics (an empty cloud indexes
// indice[0], kdtree.go:355) or could not go on (no GPU, out of device memory) New panics with the error TryNew returns.
func New(ra pc.Vec3RandomAccessor, opts ...KDTreeOption) *KDTree {
	k, err := pcgx.New(ra, opts...)
	if err != nil {
		panic(err)
	}
	return k
}

// TryNew is New with the error as a result instead of a panic.
func TryNew(ra pc.Vec3RandomAccessor, opts ...KDTreeOption) (*KDTree, error) { return pcgx.New(ra, opts...) }

// NewFromPointCloud is New(it) for it, _ := pp.Vec3Iterator(), uploading the cloud's records as they lie in
// pp.Data instead of reading them point by point.
func NewFromPointCloud(pp *pc.PointCloud, opts ...KDTreeOption) (*KDTree, error) {
	return pcgx.NewFromPointCloud(pp, opts...)
}

// WithMinDistSq is the option form of the exported field MinDistSq (the reference defines no option of its own).
func WithMinDistSq(d float32) KDTreeOption { return pcgx.WithMinDistSq(d) }
