"""Per-kernel resource usage of one translation unit, from hipcc's -Rpass-analysis=kernel-resource-usage.

    python tools/kernel_resources.py strict.hip [substring ...]

Prints VGPRs / AGPRs / SGPRs / scratch bytes per lane / LDS bytes / occupancy (waves per SIMD) for every kernel of the
file whose (demangled-ish) name contains one of the substrings (all kernels if none are given).  Also used by
tests/test_kernel_resources.py, which fails when a kernel of the ICP step spills.
"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def resources(source):
    from pcgol_amd import build as B
    src = os.path.join(B.CSRC, source)
    cmd = [B.hipcc()] + B.flags() + ["-c", src, "-o", os.devnull, "-Rpass-analysis=kernel-resource-usage"]
    out = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, check=True).stdout.decode()
    kernels, cur = {}, None
    for line in out.splitlines():
        m = re.search(r"remark:\s+([A-Za-z ]+?)(?:\s*\[[^\]]*\])?:\s+(\S+)", line)
        if not m:
            continue
        key, val = m.group(1).strip(), m.group(2)
        if key == "Function Name":
            cur = kernels.setdefault(val, {})
        elif cur is not None:
            try:
                cur[key] = int(val)
            except ValueError:
                cur[key] = val
    return kernels


def short(name):
    m = re.match(r"_ZN4pcgx(\d+)", name)
    if m:
        n = int(m.group(1))
        start = m.end()
        return name[start:start + n] + ("<" + name[start + n:][:24] + ">" if name[start + n:start + n + 1] == "I" else "")
    return name


if __name__ == "__main__":
    ks = resources(sys.argv[1])
    want = sys.argv[2:]
    for name, r in ks.items():
        if want and not any(w in name for w in want):
            continue
        print("%-60s VGPR %3s AGPR %3s SGPR %3s (spilled: %s SGPRs, %s VGPRs) scratch %4s B/lane LDS %6s B occupancy %s" % (
            short(name), r.get("VGPRs"), r.get("AGPRs"), r.get("TotalSGPRs"), r.get("SGPRs Spill"), r.get("VGPRs Spill"),
            r.get("ScratchSize"), r.get("LDS Size"), r.get("Occupancy")))
