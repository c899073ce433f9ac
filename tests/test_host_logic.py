"""CPU tests of the host-side mirror (no GPU): cloud layout contract, synthetic inputs,
spatial tiling."""
import numpy as np
import pytest

import oracle as O
from pcgol_amd import ErrInvalidField, pc, synth
from pcgol_amd.distributed import spatial_tiles


def test_xyz_field_discovery():
    """pc/pointcloud.go:130-150: one field "xyz" or consecutive x,y,z; else invalid field name."""
    def hdr(fields, size=None, count=None):
        return pc.PointCloudHeader(fields, size or [4] * len(fields), count or [1] * len(fields))
    assert pc.PointCloud(hdr(["x", "y", "z"]), 0, np.zeros(0, np.uint8)).xyz_offset() == 0
    assert pc.PointCloud(hdr(["label", "x", "y", "z"]), 0, np.zeros(0, np.uint8)).xyz_offset() == 4
    assert pc.PointCloud(hdr(["rgb", "xyz"], [4, 4], [1, 3]), 0, np.zeros(0, np.uint8)).xyz_offset() == 4
    assert pc.PointCloud(hdr(["x", "y", "z", "label"]), 0, np.zeros(0, np.uint8)).Stride() == 16
    with pytest.raises(ErrInvalidField):
        pc.PointCloud(hdr(["x", "z", "y"]), 0, np.zeros(0, np.uint8)).xyz_offset()
    p = pc.PointCloud.from_xyz(np.arange(12, dtype=np.float32).reshape(4, 3))
    assert p.Points == 4 and p.Stride() == 12 and np.array_equal(p.Vec3()[2], [6, 7, 8])
    h = p.PointCloudHeader.Clone()
    h.Fields.append("w")
    assert p.PointCloudHeader.Fields == ["x", "y", "z"]  # Clone is deep (pointcloud.go:20-31)


def test_synth_is_deterministic_and_in_range():
    a = synth.uniform_cloud(1000, 10.0, 2)
    b = synth.uniform_cloud(1000, 10.0, 2)
    assert np.array_equal(a, b) and a.dtype == np.float32
    assert a.min() >= 0 and a.max() < 10.0
    assert not np.array_equal(a, synth.uniform_cloud(1000, 10.0, 3))


def test_synth_pose_matches_oracle():
    m = O.mat4_mul(O.translate(0.02, 0.01, -0.015), O.rotate(0, 0, 1, 0.001))
    assert np.array_equal(synth.icp_pose(), m)
    pts = synth.uniform_cloud(500, 10.0, 9)
    assert np.array_equal(synth.transform_points(m, pts), O.mat4_transform(m, pts))


@pytest.mark.parametrize("world", [1, 2, 3, 8])
def test_spatial_tiles_partition(world):
    pts = synth.uniform_cloud(10007, 10.0, 4)
    tiles = spatial_tiles(pts, world)
    assert len(tiles) == world
    allidx = np.concatenate(tiles)
    assert np.array_equal(np.sort(allidx), np.arange(len(pts)))
    sizes = [len(t) for t in tiles]
    assert max(sizes) - min(sizes) <= 1
    if world > 1:  # tiles are spatially compact: much smaller extent than the cloud's in some axis
        vols = [np.prod(pts[t].max(axis=0) - pts[t].min(axis=0)) for t in tiles]
        assert np.mean(vols) < 0.75 * 1000.0
