// The GPU packages live next to the reference's own: a caller migrates by changing three import paths
//   github.com/seqsense/pcgol/pc/storage/kdtree    -> github.com/seqsense/pcgol/gpu/pc/storage/kdtree
//   github.com/seqsense/pcgol/pc/filter/voxelgrid  -> github.com/seqsense/pcgol/gpu/pc/filter/voxelgrid
//   github.com/seqsense/pcgol/pc/registration/icp  -> github.com/seqsense/pcgol/gpu/pc/registration/icp
// (this directory checked out as <pcgol>/gpu, or kept elsewhere with the replace directive below pointed at a pcgol checkout).
module github.com/seqsense/pcgol/gpu

go 1.21

require github.com/seqsense/pcgol v0.0.0

// replace github.com/seqsense/pcgol => ../
