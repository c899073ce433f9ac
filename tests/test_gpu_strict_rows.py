"""The strict sums' DEVICE pipeline (csrc/strict.hip: strict_sum / strict_job / strict_chain kernels) on rows of
terms no registration produces: pcgx_debug_strict_sum_dev runs the kernels on nine given rows instead of on the
terms of a session's pairs.  Checker: a plain sequential float32 accumulation, what the reference does
(pc/registration/icp/evaluator.go:122-145).  Every sum must agree bit for bit -- with the candidate tables, the
jobs' leaf records, the chain kernel's recomputations from the pairs (no slot left) and the in-kernel self-check
each taking their turn.  The host model of the same arithmetic is tests/test_strict_model.py."""
import numpy as np
import pytest

from pcgol_amd import _lib as L
from test_strict_model import ROWS, same_bits, sequential_f32

pytestmark = pytest.mark.gpu


def device_sums(rows9):
    n = len(rows9[0])
    t = np.ascontiguousarray(np.stack(rows9).astype(np.float32))
    assert t.shape == (9, n)
    out = np.zeros(9, np.float32)
    stats = np.zeros(64, np.int64)
    L.check(L.lib().pcgx_debug_strict_sum_dev(L.ptr(t), n, L.ptr(out), L.ptr(stats)))
    return out, stats


def padded(t, n):
    return np.concatenate([t, np.full(n - len(t), -0.0, np.float32)]) if len(t) < n else t[:n]


NAMES = sorted(ROWS)


@pytest.mark.parametrize("env", [{}, {"PCGX_STRICT_SELFCHECK": "1"}, {"PCGX_STRICT_NOSPEC": "1"},
                                 {"PCGX_STRICT_SLOTS_PER_SHARD": "0"}, {"PCGX_STRICT_SLOTS_PER_SHARD": "1"}])
@pytest.mark.parametrize("first", range(0, len(NAMES), 9))
def test_adversarial_rows_on_the_device(first, env, monkeypatch):
    """Nine of the host model's rows at a time (the last group wraps around), cut / padded to one length."""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    names = [NAMES[(first + k) % len(NAMES)] for k in range(9)]
    n = 70_001
    rows9 = [padded(ROWS[name], n) for name in names]
    got, stats = device_sums(rows9)
    for k, name in enumerate(names):
        want = sequential_f32(rows9[k])
        assert same_bits(got[k], want), (name, env, got[k], want, stats[:8])
    if "PCGX_STRICT_SELFCHECK" in env:
        assert not stats[12:16].any() and stats[6] == 0 and stats[7] == 0, stats[:24]


@pytest.mark.parametrize("n", [1, 2, 31, 32, 33, 2047, 2048, 2049, 4095, 4097, 64 * 2048 + 5, 1_000_003])
def test_ragged_lengths_on_the_device(n):
    rng = np.random.Generator(np.random.PCG64(n + 1))
    rows9 = [(rng.standard_normal(n) * 0.3 * (k + 1) + 0.1 * (k - 4)).astype(np.float32) for k in range(9)]
    got, _ = device_sums(rows9)
    for k in range(9):
        assert same_bits(got[k], sequential_f32(rows9[k])), (n, k)


def test_random_rows_property_on_the_device():
    """Seeded fuzz: random scale, drift and a sprinkling of exact ties, nine different rows per launch."""
    rng = np.random.Generator(np.random.PCG64(5))
    for _ in range(12):
        n = int(rng.integers(1, 300_000))
        rows9 = []
        for k in range(9):
            scale = 10.0 ** rng.uniform(-6, 3)
            drift = scale * rng.uniform(-1, 1) * rng.choice([0, 0.01, 1])
            t = (rng.standard_normal(n) * scale + drift).astype(np.float32)
            if rng.random() < 0.5:
                q = np.float32(2.0 ** np.floor(np.log2(scale)) / 64)
                t = (np.round(t / q) * q).astype(np.float32)   # few significant bits: ties galore
            rows9.append(t)
        got, stats = device_sums(rows9)
        for k in range(9):
            assert same_bits(got[k], sequential_f32(rows9[k])), (n, k, stats[:8])


@pytest.mark.parametrize("env", [{}, {"PCGX_STRICT_SELFCHECK": "1"}])
def test_nine_sums_that_hover_around_zero_at_full_size(env, monkeypatch):
    """1M terms per row, every row a zero-mean random walk of its own scale (hundreds of tiles without a window per
    row), plus rows that cross zero again and again on purpose: the jobs' candidate tables, leaf records and the
    walker's look-ups at the size the bench runs."""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    rng = np.random.Generator(np.random.PCG64(99))
    n = 1_000_000
    rows9 = [(rng.standard_normal(n) * 10.0 ** (k - 5)).astype(np.float32) for k in range(7)]
    saw = (np.arange(n) % 4096 < 2048).astype(np.float32) * 2 - 1          # +1 for 2048 terms, -1 for 2048, ...
    rows9.append((saw * np.float32(0.37) + rng.standard_normal(n).astype(np.float32) * np.float32(1e-3)).astype(np.float32))
    rows9.append((rng.integers(-3, 4, n) * 2.0 ** -20).astype(np.float32))  # exact terms: every guess is exact
    got, stats = device_sums(rows9)
    for k in range(9):
        assert same_bits(got[k], sequential_f32(rows9[k])), (k, env, stats[:8])
    assert stats[24] > 50, stats[:32]   # the case is about tiles without a window
    if env:
        assert not stats[12:16].any() and stats[6] == 0 and stats[7] == 0, stats[:24]


def test_more_tiles_than_one_chunk_of_the_chain_kernel():
    """3M terms per row = 1465 tiles: the chain kernel walks them in three chunks of 512 (records, scans, helpers'
    tables per chunk).  Hovering rows, drifting rows and a row with a NaN in the last chunk."""
    rng = np.random.Generator(np.random.PCG64(123))
    n = 3_000_000
    rows9 = [(rng.standard_normal(n) * 10.0 ** (k - 4) + (k % 3 - 1) * 10.0 ** (k - 6)).astype(np.float32) for k in range(8)]
    last = (rng.standard_normal(n) * 0.01).astype(np.float32)
    last[2_900_000] = np.float32(np.inf)
    last[2_950_000] = np.float32(-np.inf)
    rows9.append(last)
    got, stats = device_sums(rows9)
    for k in range(9):
        assert same_bits(got[k], sequential_f32(rows9[k])), (k, stats[:8])
