"""Host assignment solver (tscd_amd/csrc_host/lap.cpp) == scipy.optimize.linear_sum_assignment, bit for bit.

scipy is the solver the reference calls (yolox/models/tscd_matching.py:908,935); it is installed here, so it pins the
restatement directly: identical row/column indices on tie-heavy (small integers, constant, fp16-quantised) and
rectangular matrices, which exercise the tie rules of the augmenting-path search."""
import numpy as np
import pytest
from scipy.optimize import linear_sum_assignment


@pytest.mark.parametrize("seed", range(6))
def test_lap_matches_scipy_exactly(seed):
    from tscd_amd._host import lap_f64, lap_f16
    rng = np.random.default_rng(seed)
    for trial in range(120):
        nr, nc = rng.integers(1, 90, 2)
        kind = trial % 5
        if kind == 0:
            c = rng.integers(0, 4, (nr, nc)).astype(float)
        elif kind == 1:
            c = (1 + 0.05 * rng.standard_normal((nr, nc))).astype(np.float16).astype(float)
        elif kind == 2:
            c = np.ones((nr, nc))
        elif kind == 3:
            c = rng.random((nr, nc))
        else:
            c = np.round(rng.random((nr, nc)), 1)
        r0, c0 = linear_sum_assignment(c)
        r1, c1 = lap_f64(c)
        assert np.array_equal(r0, r1) and np.array_equal(c0, c1), (seed, trial, nr, nc)
        if kind == 1:
            h = c.astype(np.float16)
            perm = rng.permutation(nr)
            r2, c2 = lap_f16(np.ascontiguousarray(h).view(np.uint16), perm)
            r3, c3 = linear_sum_assignment(h[perm].astype(float))
            assert np.array_equal(r2, r3) and np.array_equal(c2, c3)


def test_lap_large_and_nan():
    from tscd_amd._host import lap_f16
    rng = np.random.default_rng(7)
    for n, m in ((500, 500), (500, 430), (430, 500)):
        h = (1 + 0.05 * rng.standard_normal((n, m))).astype(np.float16)
        h[3, 5] = np.nan
        ref = h.astype(float)
        ref[np.isnan(ref)] = 0
        r0, c0 = linear_sum_assignment(ref)
        r1, c1 = lap_f16(h.view(np.uint16))
        assert np.array_equal(r0, r1) and np.array_equal(c0, c1)
