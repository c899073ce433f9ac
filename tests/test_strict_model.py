"""The arithmetic behind the parallel STRICT sums (csrc/strict_sum.h), on the CPU.

pcgx_debug_strict_sum_host runs the pipeline of the strict_* kernels in plain host loops, compiled
from the same header the device code uses (leaf summaries, compose, apply, resolve).  Checker: a
plain sequential float32 accumulation, which is what the reference does
(pc/registration/icp/evaluator.go:122-145).  Every case must agree bit for bit."""
import ctypes as C

import numpy as np
import pytest

from pcgol_amd import _lib as L


def sequential_f32(t):
    # np.add.accumulate on float32 adds left to right in float32: s <- fl32(s + t_i)
    if len(t) == 0:
        return np.float32(0.0)
    return np.add.accumulate(np.concatenate([[np.float32(0.0)], t]).astype(np.float32), dtype=np.float32)[-1]


def model(t, mode=0):
    t = np.ascontiguousarray(t, dtype=np.float32)
    out = C.c_float(0)
    stats = np.zeros(8, np.int64)
    L.check(L.lib().pcgx_debug_strict_sum_host(L.ptr(t) if len(t) else None, len(t), mode, C.byref(out), L.ptr(stats)))
    return np.float32(out.value), stats


def same_bits(a, b):
    a, b = np.float32(a), np.float32(b)
    return a.view(np.uint32) == b.view(np.uint32) or (np.isnan(a) and np.isnan(b))


def rows():
    rng = np.random.Generator(np.random.PCG64(11))
    n = 70_000
    out = {}
    out["positive"] = rng.random(n, dtype=np.float32) * np.float32(3e-3)
    out["drift+noise"] = (rng.standard_normal(n) * 0.02 + 0.013).astype(np.float32)
    out["random walk"] = (rng.standard_normal(n) * 0.02).astype(np.float32)          # hovers, changes sign
    out["constant (biased rounding)"] = np.full(n, -0.015, np.float32)
    out["ones"] = np.ones(n, np.float32)                                              # every step exact, lands on 2^k
    out["ties"] = np.concatenate([[np.float32(2 ** 24)], np.ones(n, np.float32)])    # s + 1 is a tie at every step
    out["ties alternating"] = np.concatenate([[np.float32(2 ** 24)], np.tile(np.array([1, 3, -1, 1], np.float32), n // 4)])
    out["few bits"] = (rng.integers(-64, 64, n) * 2.0 ** -12).astype(np.float32)      # structured terms: many ties
    out["coordinates"] = (rng.integers(0, 1 << 24, n).astype(np.float32) / np.float32(1 << 24) * np.float32(10.0)
                          - rng.integers(0, 1 << 24, n).astype(np.float32) / np.float32(1 << 24) * np.float32(10.0))
    wide = np.exp(rng.uniform(-40, 20, n)) * rng.choice([-1.0, 1.0], n)
    out["wide dynamic range"] = wide.astype(np.float32)
    out["cancel to zero"] = np.concatenate([out["drift+noise"], -out["drift+noise"][::-1]])
    out["tiny"] = (rng.standard_normal(n) * 1e-41).astype(np.float32)                # subnormal sums
    z = out["random walk"].copy()
    z[::7] = -0.0
    z[::11] = 0.0
    out["with zeros"] = z
    big = out["drift+noise"].copy()
    big[1234] = np.float32(3e38)
    big[2345] = np.float32(3e38)      # overflows to +inf ...
    out["overflow"] = big
    nn = big.copy()
    nn[40_000] = np.float32(-3e38)
    nn[40_001] = np.float32(-3e38)
    nn[40_002] = np.float32(-3e38)    # ... and inf - inf gives NaN
    out["nan"] = nn
    out["steps"] = np.concatenate([np.full(5000, 1e-3, np.float32), np.full(5000, 7.0, np.float32),
                                   np.full(5000, -7.0, np.float32), np.full(5000, 1e-3, np.float32)])
    return out


ROWS = rows()


@pytest.mark.parametrize("name", sorted(ROWS))
@pytest.mark.parametrize("mode", [0, 1, 2, 3, 4, 6])
def test_model_equals_sequential_sum(name, mode):
    t = ROWS[name]
    got, stats = model(t, mode)
    assert same_bits(got, sequential_f32(t)), (name, mode, got, sequential_f32(t), stats)


@pytest.mark.parametrize("n", [0, 1, 2, 31, 32, 33, 2047, 2048, 2049, 4095, 4097, 64 * 2048 + 5])
def test_ragged_lengths(n):
    rng = np.random.Generator(np.random.PCG64(n + 1))
    t = (rng.standard_normal(n) * 0.3 + 0.1).astype(np.float32)
    got, _ = model(t)
    assert same_bits(got, sequential_f32(t))


def test_records_carry_most_of_the_work():
    """On a well-behaved row nearly every tile is applied through its record (the fast path the GPU
    relies on); the exact recomputation handles the first tile and a few window edges."""
    t = ROWS["drift+noise"]
    got, stats = model(np.tile(t, 8))
    tiles, norec, runs, runfail, resolved = stats[:5]
    assert same_bits(got, sequential_f32(np.tile(t, 8)))
    assert resolved <= 12 and resolved < tiles // 10, stats


def test_random_rows_property():
    """Seeded fuzz: random scale, drift, length and a sprinkling of exact ties."""
    rng = np.random.Generator(np.random.PCG64(5))
    for _ in range(40):
        n = int(rng.integers(1, 30_000))
        scale = 10.0 ** rng.uniform(-6, 3)
        drift = scale * rng.uniform(-1, 1) * rng.choice([0, 0.01, 1])
        t = (rng.standard_normal(n) * scale + drift).astype(np.float32)
        if rng.random() < 0.5:
            q = np.float32(2.0 ** np.floor(np.log2(scale)) / 64)
            t = (np.round(t / q) * q).astype(np.float32)   # few significant bits: ties galore
        got, stats = model(t, int(rng.integers(0, 8)))
        assert same_bits(got, sequential_f32(t)), (n, scale, drift, stats)
