"""Builds pcgol_amd/libpcgx.so (hand-written HIP for gfx950) in-tree with hipcc.

The flags matter for parity: -ffp-contract=off (the Go reference never fuses
multiply-add on amd64), correctly rounded float32 divide/sqrt (hipcc default)
and no fast-math, on both the host and the device side.
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
SO = os.path.join(HERE, "libpcgx.so")
SOURCES = ["core.hip", "knn.hip", "knn_explicit.hip", "knn_grid.hip", "sort.hip", "icp.hip", "icp_small.hip", "strict.hip", "strict_check.hip", "comm.hip", "voxel.hip", "voxel_bucket.hip", "range.hip", "segment.hip", "pcd.hip", "kdtree_build_gpu.hip", "kdtree_build.cpp"]
# every header beside the sources, whoever includes it (round 5's wg_stamps.h was missing from a hand-kept list: an
# edit there changed kernels without changing source_hash), and the public header
HEADERS = sorted(f for f in os.listdir(CSRC) if f.endswith(".h")) + [os.path.join("..", "..", "include", "pcgx.h")]
ARCH = "gfx950"


def hipcc():
    for c in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found: libpcgx.so can only be built with the ROCm toolchain")


def flags():
    # (-align-all-nofallthru-blocks=6: loop heads and branch targets on 64-byte instruction-cache lines -- the walker
    # wave of strict_chain_kernel is one wave's instruction stream, and where its blocks fall was worth 0.4-0.5 us of
    # the ICP step either way, HISTORY.md)
    return ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math",
            "-fPIC", "-fvisibility=hidden", "-Wall", "-Wno-unused-result", "-mllvm", "-align-all-nofallthru-blocks=6",
            "-x", "hip"] + \
        os.environ.get("PCGX_EXTRA_CFLAGS", "").split()  # experiments only (e.g. -DPCGX_WALK_TOP_LEVELS=6)


def source_hash():
    """sha256 (16 hex digits) over the kernel sources: profiles/*_pmc.json records the build its counters were
    collected on, bench.py says whether that is the build it is timing."""
    import hashlib
    h = hashlib.sha256()
    h.update(" ".join(flags()).encode())  # (the compiler's switches are part of what a library was built from)
    for s in sorted(SOURCES + HEADERS):   # (the public header too: its ABI version and struct layouts are compiled in)
        with open(os.path.join(CSRC, s), "rb") as f:
            h.update(os.path.basename(s).encode() + b"\0" + f.read())
    return h.hexdigest()[:16]


def needs_build():
    if not os.path.exists(SO):
        return True
    if stale():
        return True
    t = os.path.getmtime(SO)
    deps = [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    if not force and not needs_build():
        return SO
    objs = []
    procs = []
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    for s in SOURCES:
        o = os.path.join(objdir, os.path.splitext(s)[0] + ".o")
        objs.append(o)
        cmd = [hipcc()] + flags() + ["-c", os.path.join(CSRC, s), "-o", o]
        if verbose:
            print(" ".join(cmd))
        procs.append((s, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    failed = False
    for s, p in procs:
        out = p.communicate()[0].decode()
        if p.returncode != 0:
            failed = True
            sys.stderr.write("== %s ==\n%s\n" % (s, out))
        elif verbose and out.strip():
            print(out)
    if failed:
        raise RuntimeError("hipcc failed")
    cmd = [hipcc(), "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", SO] + objs + ["-lpthread", "-ldl"]
    subprocess.check_call(cmd)  # (-ldl is part of libc on this image; RCCL is bound at run time, csrc/comm.hip)
    with open(SO + ".hash", "w") as f:  # what the library was built from (_lib.py refuses a library older than its sources)
        f.write(source_hash() + "\n")
    return SO


def stale():
    """True when libpcgx.so was built from other sources than the ones beside it (a checkout, an edit): measuring or
    testing such a library says nothing about the tree."""
    try:
        with open(SO + ".hash") as f:
            return f.read().strip() != source_hash()
    except OSError:
        return True


if __name__ == "__main__":
    print(build(force="-f" in sys.argv, verbose=True))
