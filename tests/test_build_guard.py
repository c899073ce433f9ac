"""The binding refuses a library that was not built from the sources beside it (pcgol_amd/build.py: stale())."""
import os
import shutil

from pcgol_amd import build as B


def test_hash_file_matches_sources():
    # the build check (__graft_entry__.build) ran before the tests: the recorded hash is the sources' hash
    if not os.path.exists(B.SO):
        import pytest
        pytest.skip("library not built")
    assert not B.stale()


def test_stale_when_hash_differs(tmp_path, monkeypatch):
    if not os.path.exists(B.SO + ".hash"):
        import pytest
        pytest.skip("library not built")
    so = tmp_path / "libpcgx.so"
    so.write_bytes(b"")
    shutil.copy(B.SO + ".hash", str(so) + ".hash")
    monkeypatch.setattr(B, "SO", str(so))
    assert not B.stale()
    with open(str(so) + ".hash", "w") as f:
        f.write("0123456789abcdef\n")
    assert B.stale()
    os.remove(str(so) + ".hash")
    assert B.stale()   # no record at all: not trusted


def test_every_file_the_kernels_are_built_from_is_hashed():
    """source_hash() ties profiles/*_pmc.json to the build bench.py times: every translation unit under csrc/ is in
    SOURCES, and every file any of them includes with quotes is in SOURCES + HEADERS."""
    import re
    hashed = {os.path.normpath(os.path.join(B.CSRC, f)) for f in B.SOURCES + B.HEADERS}
    units = [f for f in os.listdir(B.CSRC) if f.endswith((".hip", ".cpp"))]
    assert sorted(units) == sorted(B.SOURCES)
    for f in os.listdir(B.CSRC):
        with open(os.path.join(B.CSRC, f)) as fh:
            for inc in re.findall(r'^\s*#\s*include\s+"([^"]+)"', fh.read(), re.M):
                cand = [os.path.normpath(os.path.join(B.CSRC, inc)), os.path.normpath(os.path.join(B.CSRC, "..", "..", "include", inc))]
                assert any(c in hashed for c in cand), "%s includes %s, which source_hash() does not cover" % (f, inc)
