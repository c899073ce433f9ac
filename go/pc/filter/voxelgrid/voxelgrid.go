// Package voxelgrid is the GPU drop-in for github.com/seqsense/pcgol/pc/filter/voxelgrid: New, Options, Option and
// WithChunkSize as the reference declares them (voxelgrid.go:23-33, option.go:7-18); Filter returns the reference's
// output byte for byte (same cells, same record carried over, same float32 centroids, same order).
//
// NOT compiled in the build image (no Go toolchain there); see go/README.md.
package voxelgrid

import (
	"github.com/seqsense/pcgol/mat"
	"github.com/seqsense/pcgol/pc/filter"

	"github.com/seqsense/pcgol/gpu/pcgx"
)

// Options is voxelgrid.Options (option.go:7-10).
type Options struct {
	LeafSize  mat.Vec3
	ChunkSize [3]int
}

// Option is voxelgrid.Option (option.go:12).
type Option func(*Options)

// WithChunkSize is voxelgrid.WithChunkSize (option.go:14-18).
func WithChunkSize(s [3]int) Option {
	return Option(func(o *Options) {
		o.ChunkSize = s
	})
}

// New is voxelgrid.New (voxelgrid.go:23-33).  Unlike the reference's filter the result holds no scratch between
// calls and may be used from several goroutines.  Where the reference panics with an index out of range (a point
// outside its dense array, voxelgrid.go:151) Filter returns pcgx.ErrOutOfRange.
func New(leafSize mat.Vec3, opts ...Option) filter.Filter {
	o := Options{LeafSize: leafSize}
	for _, f := range opts {
		f(&o)
	}
	return pcgx.NewVoxelGrid(o.LeafSize, o.ChunkSize)
}
