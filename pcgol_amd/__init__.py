"""pcgol_amd -- MI355X (gfx950) hot path behind seqsense/pcgol's interfaces.

Host-side mirror (Python over the libpcgx.so C ABI, include/pcgx.h) of the
reference packages on the path: pc/storage/kdtree, pc/filter/voxelgrid,
pc/registration/icp.  The Go shim that binds the same C ABI is in go/.
"""
from . import _lib, icp, kdtree, mat, pc, synth, voxelgrid  # noqa: F401
from ._lib import (ErrInvalidField, ErrNeedGradient, ErrNoPoint, ErrNotEnoughPairs,  # noqa: F401
                   PcgxError)
