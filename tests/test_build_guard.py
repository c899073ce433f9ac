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
