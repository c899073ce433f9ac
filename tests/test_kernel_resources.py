"""No kernel of the ICP step (and of the VoxelGrid bucket path) may spill: hipcc's own resource report
(-Rpass-analysis=kernel-resource-usage, tools/kernel_resources.py) for the production instantiations.
HISTORY.md (3.9) records what spilled dwords cost on this chip (5 us phases became 40 us ones); round 3 shipped
strict_sum_kernel with 8 bytes of scratch per lane under its occupancy attribute."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import kernel_resources as KR  # noqa: E402

STEP_KERNELS = {
    "icp.hip": ["icp_grid_kernelILb0ELb0ELb0E", "icp_corr_kernelILb0ELb0ELb1ELb0E"],
    "strict.hip": ["strict_sum_kernelILb1E", "strict_sum_kernelILb0E", "strict_job_kernel", "strict_chain_kernelILb0ELb0E",
                   "strict_chain_kernelILb0ELb1E"],
    "icp_small.hip": ["icp_small_fit_kernelILb0E", "icp_small_fit_kernelILb1E"],
    "voxel_bucket.hip": ["vb_key_hist_kernel", "vb_scatter_kernelILb1ELb0E", "vb_scatter_kernelILb0ELb0E",
                         "vb_bucket_kernelILb0E", "vb_bucket_kernelILb1E"],
}


@pytest.mark.parametrize("source", sorted(STEP_KERNELS))
def test_no_scratch_in_the_hot_kernels(source):
    ks = KR.resources(source)
    for want in STEP_KERNELS[source]:
        hits = {n: r for n, r in ks.items() if want in n}
        assert hits, (source, want, sorted(ks)[:8])
        for name, r in hits.items():
            assert r.get("ScratchSize") == 0, (name, r)
            assert r.get("VGPRs Spill") == 0, (name, r)


def test_the_chain_kernel_fits_a_cu():
    """strict_chain_kernel keeps its chunk's records, their compositions both ways and fourteen candidate tables in LDS
    (154 KB): all three builds of it (plain, self-checking, with the walk ahead of a wait) must stay inside the CU's
    160 KB, and at 256 registers or fewer a lane (eight waves of a workgroup, two per SIMD)."""
    ks = KR.resources("strict.hip")
    hits = {n: r for n, r in ks.items() if "strict_chain_kernel" in n}
    assert len(hits) == 3, sorted(ks)
    for name, r in hits.items():
        assert 0 < r.get("LDS Size", 0) <= 160 * 1024, (name, r)
        assert r.get("VGPRs", 999) <= 256 and r.get("VGPRs Spill") == 0, (name, r)
