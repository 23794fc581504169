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


@pytest.mark.parametrize("seed", range(4))
def test_lap_fixed_point_fp16_variant_matches_scipy(seed):
    """The int32 fixed-point AVX-512 path taken for fp16 matrices (and its fall-back to the double solver outside the
    fixed-point range) returns scipy's indices on the cost shapes CAFM produces: 1 - cos of near-identical rows (dense
    ties), self-match (zero diagonal), slightly negative entries, rectangular, plus out-of-range / inf entries."""
    from tscd_amd._host import lap_f16
    rng = np.random.default_rng(100 + seed)
    for trial in range(60):
        nr, nc = rng.integers(1, 140, 2)
        kind = trial % 6
        if kind == 0:       # near-identical unit rows: costs ~1e-3 with heavy fp16 ties
            base = rng.standard_normal((1, 32))
            a = base + 0.02 * rng.standard_normal((nr, 32)); b = base + 0.02 * rng.standard_normal((nc, 32))
            a /= np.linalg.norm(a, axis=1, keepdims=True); b /= np.linalg.norm(b, axis=1, keepdims=True)
            h = (1 - a @ b.T).astype(np.float16)
        elif kind == 1:     # self-match
            a = rng.standard_normal((nr, 16)); a /= np.linalg.norm(a, axis=1, keepdims=True)
            h = (1 - (a @ a.T).astype(np.float16).astype(np.float32)).astype(np.float16)
        elif kind == 2:
            h = (0.01 * rng.standard_normal((nr, nc))).astype(np.float16)          # both signs
        elif kind == 3:
            h = rng.integers(0, 3, (nr, nc)).astype(np.float16)
        elif kind == 4:     # leaves the fixed-point range -> double solver
            h = (20 * rng.random((nr, nc))).astype(np.float16)
        else:
            h = (1 + 0.05 * rng.standard_normal((nr, nc))).astype(np.float16)
            h[rng.integers(0, nr), rng.integers(0, nc)] = np.nan
        ref = h.astype(float)
        ref[np.isnan(ref)] = 0
        perm = rng.permutation(h.shape[0])
        r0, c0 = linear_sum_assignment(ref[perm])
        r1, c1 = lap_f16(np.ascontiguousarray(h).view(np.uint16), perm)
        assert np.array_equal(r0, r1) and np.array_equal(c0, c1), (seed, trial, kind, h.shape)


def test_lap_on_device_cost_sample():
    """A 192x160 crop of a CAFM cost matrix the MI355X GEMM produced on the bench clip (random-init regime: hundreds of
    exactly tied fp16 entries per row) with the indices scipy returns for it."""
    import os
    from tscd_amd._host import lap_f16
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "cafm_cost_device_sample.npz"))
    r, c = lap_f16(np.ascontiguousarray(z["cost_f16_bits"]))
    assert np.array_equal(r, z["scipy_rows"]) and np.array_equal(c, z["scipy_cols"])
    r0, c0 = linear_sum_assignment(z["cost_f16_bits"].view(np.float16).astype(float))
    assert np.array_equal(c0, z["scipy_cols"])


@pytest.mark.parametrize("seed", range(3))
def test_lap_chain_matches_stepwise_scipy(seed):
    """tscd_lap_chain_f16 (the clip's dependent assignments in one call) against the reference's frame loop written with
    scipy (tscd_matching.py:940-1080): previous rows re-ordered by the previous assignment, unmatched current proposals
    appended when the reference frame was smaller, matched reference rows only when it was larger."""
    from tscd_amd._host import lap_chain_f16, KIND_SELF, KIND_STATE, KIND_PREV
    rng = np.random.default_rng(500 + seed)
    for trial in range(10):
        K = int(rng.integers(2, 6))
        ns = [int(rng.integers(20, 90)) for _ in range(K)]
        first_kind = KIND_SELF if trial % 2 == 0 else KIND_STATE
        n_state = int(rng.integers(20, 90))
        mats, kinds = [], []
        for k in range(K):
            nr = ns[k] if (k == 0 and first_kind == KIND_SELF) else (n_state if k == 0 else ns[k - 1])
            base = rng.standard_normal((1, 16))
            a = base + 0.03 * rng.standard_normal((nr, 16)); b = base + 0.03 * rng.standard_normal((ns[k], 16))
            a /= np.linalg.norm(a, axis=1, keepdims=True); b /= np.linalg.norm(b, axis=1, keepdims=True)
            mats.append(np.ascontiguousarray((1 - a @ b.T).astype(np.float16)))
            kinds.append(first_kind if k == 0 else KIND_PREV)
        flat = np.concatenate([m.reshape(-1) for m in mats])
        got = lap_chain_f16(flat.ctypes.data, [m.shape for m in mats], kinds)
        prev_perm = None
        for k, m in enumerate(mats):
            ref = m.astype(float)
            if kinds[k] == KIND_PREV:
                ref = ref[prev_perm]
            r, c = linear_sum_assignment(ref)
            if kinds[k] == KIND_SELF:
                perm = np.arange(m.shape[1])
            else:
                rest = np.setdiff1d(np.arange(m.shape[1]), c)
                perm = np.concatenate([c, rest])
            assert np.array_equal(got[k][0], r) and np.array_equal(got[k][1], c), (seed, trial, k)
            assert np.array_equal(got[k][2], perm), (seed, trial, k)
            prev_perm = perm


def _chain_reference(mats, kinds, KIND_SELF, KIND_PREV):
    out, prev_perm = [], None
    for k, m in enumerate(mats):
        ref = m.astype(float)
        if kinds[k] == KIND_PREV:
            ref = ref[prev_perm]
        r, c = linear_sum_assignment(ref)
        perm = np.arange(m.shape[1]) if kinds[k] == KIND_SELF else np.concatenate([c, np.setdiff1d(np.arange(m.shape[1]), c)])
        out.append((r, c, perm))
        prev_perm = perm
    return out


@pytest.mark.parametrize("mode", ["ties", "real", "mixed"])
def test_lap_chain_side_by_side_solves_match_in_order_scipy(mode):
    """The chain solves its matrices side by side in identity row order and relabels the answer where the optimum is certified
    unique (exact integer duals), falling back to the in-order solve otherwise.  Whatever the certificate says, the result must be
    the reference's in-order scipy answer: tie-heavy costs (a handful of distinct fp16 values: most optima are NOT unique and the
    tie rules decide), real-valued costs (unique: answered from the side-by-side solve) and mixtures, square and rectangular in
    both directions -- and the strictly sequential chain must agree."""
    from tscd_amd import _host
    from tscd_amd._host import lap_chain_f16, KIND_SELF, KIND_STATE, KIND_PREV
    rng = np.random.default_rng({"ties": 1, "real": 2, "mixed": 3}[mode])
    hits = 0
    for trial in range(60):
        K = int(rng.integers(2, 6))
        ns = [int(rng.integers(3, 40)) for _ in range(K)]
        first_kind = KIND_SELF if trial % 2 == 0 else KIND_STATE
        n_state = int(rng.integers(3, 40))
        mats, kinds = [], []
        for k in range(K):
            nr = ns[k] if (k == 0 and first_kind == KIND_SELF) else (n_state if k == 0 else ns[k - 1])
            tie = mode == "ties" or (mode == "mixed" and rng.random() < 0.5)
            if tie:
                levels = int(rng.integers(2, 6))
                m = rng.integers(0, levels, (nr, ns[k])).astype(np.float16) / 8
                if rng.random() < 0.3:      # a nearly-unique matrix with one tied pair of rows
                    m = rng.random((nr, ns[k])).astype(np.float16)
                    if nr > 1:
                        m[1] = m[0]
            else:
                m = (0.5 + 0.5 * rng.random((nr, ns[k]))).astype(np.float16)
            mats.append(np.ascontiguousarray(m))
            kinds.append(first_kind if k == 0 else KIND_PREV)
        flat = np.concatenate([m.reshape(-1) for m in mats])
        ref = _chain_reference(mats, kinds, KIND_SELF, KIND_PREV)
        for parallel in (2, 0):                    # 2: side-by-side pass on every chain (1 = adaptive default)
            prev = _host._lib.tscd_lap_chain_set_parallel(parallel)
            _host._lib.tscd_lap_chain_spec_hits(1)
            try:
                got = lap_chain_f16(flat.ctypes.data, [m.shape for m in mats], kinds)
            finally:
                _host._lib.tscd_lap_chain_set_parallel(prev)
            if parallel:
                hits += _host._lib.tscd_lap_chain_spec_hits(1)
            else:
                assert _host._lib.tscd_lap_chain_spec_hits(1) == 0
            for k in range(K):
                for a, b in zip(got[k], ref[k]):
                    assert np.array_equal(a, b), (mode, trial, k, parallel, mats[k].shape)
    if _host._lib.tscd_lap_fixed_point_available():
        if mode == "real":
            assert hits > 100          # unique optima are answered from the side-by-side solves
        if mode == "ties":
            assert hits < 60           # tie-heavy optima are (almost) never certified unique


def test_chain_with_index_table_and_latch():
    """tscd_lap_chain_f16_table (the clamped training regime's deferred join): the index table [inv_0 | c_1 | inv_1 | ...] equals what
    AwarePositionRegMatcher.finish builds from the chain's results on the host, and the latch is raised on every path -- on a clip
    outside the regime the call fails, the table holds in-range identity rows and a parked stream would still be released."""
    import threading
    from tscd_amd._host import Latch, lap_chain_f16, KIND_SELF, KIND_PREV
    rng = np.random.default_rng(77)
    n, K = 64, 4
    mats = []
    for k in range(K):
        base = rng.standard_normal((1, 16))
        a = base + 0.03 * rng.standard_normal((n, 16)); b = base + 0.03 * rng.standard_normal((n, 16))
        a /= np.linalg.norm(a, axis=1, keepdims=True); b /= np.linalg.norm(b, axis=1, keepdims=True)
        mats.append(np.ascontiguousarray((1 - (a @ a.T if k == 0 else a @ b.T)).astype(np.float16)))
    mats[0][3] = mats[0][2]                                    # duplicate proposals: the self assignment is not the identity
    mats[0][:, 3] = mats[0][:, 2]
    flat = np.concatenate([m.reshape(-1) for m in mats])
    kinds = [KIND_SELF] + [KIND_PREV] * (K - 1)
    plain = lap_chain_f16(flat.ctypes.data, [m.shape for m in mats], kinds)
    latch = Latch()
    gen = latch.new_generation()
    table = np.full((2 * K - 1) * n, -1, np.int64)
    waited = []
    th = threading.Thread(target=lambda: waited.append(latch.wait(gen, 20000)))
    th.start()
    got = lap_chain_f16(flat.ctypes.data, [m.shape for m in mats], kinds, table_ptr=table.ctypes.data, table_len=len(table), latch=latch, gen=gen)
    th.join()
    assert waited == [True] and latch.generation == gen and latch.timeouts == 0
    for a, b in zip(plain, got):
        assert all(np.array_equal(x, y) for x, y in zip(a, b))
    want = [np.argsort(plain[0][1])]
    for k in range(1, K):
        want += [plain[k][2], np.argsort(plain[k][2])]
    assert np.array_equal(table, np.concatenate(want))
    # outside the regime (a frame of another size): status 5 -> ValueError, identity rows, latch still raised
    gen2 = latch.new_generation()
    bad = [mats[0], np.ascontiguousarray(mats[1][:, :n - 1])]
    flat2 = np.concatenate([m.reshape(-1) for m in bad])
    table2 = np.full(3 * n, -1, np.int64)
    with pytest.raises(ValueError):
        lap_chain_f16(flat2.ctypes.data, [m.shape for m in bad], kinds[:2], table_ptr=table2.ctypes.data, table_len=len(table2), latch=latch, gen=gen2)
    assert latch.wait(gen2, 10) and table2.min() == 0 and table2.max() == n - 1
    # a chain that cannot start (the wait for the cost copy fails) releases the latch as well
    gen3 = latch.new_generation()

    def boom():
        raise RuntimeError("copy failed")
    with pytest.raises(RuntimeError):
        lap_chain_f16(flat.ctypes.data, [m.shape for m in mats], kinds, before=boom, table_ptr=table.ctypes.data, table_len=len(table), latch=latch, gen=gen3)
    assert latch.wait(gen3, 10)
    assert np.array_equal(table, np.arange(len(table)) % n)           # ... and no stale table is handed to the device
    assert not latch.wait(gen3 + 1, 5) and latch.timeouts == 1


def test_lap_fixed_point_on_tie_heavy_matrices_matches_scipy():
    """Matrices with massive exact ties -- quarter-integer costs, constant matrices, rows / columns drawn with replacement from the device
    cost sample of tests/golden (the random-init regime: ~8 columns tied at the minimum of a pass) -- through the fixed-point solver:
    rows and columns identical to scipy's.  The plateau pass (lap.cpp: relax_row_plateau) takes the tie rule from the same sweep as the
    relaxation on 9 passes of 10 here."""
    import os
    from tscd_amd._host import lap_f16
    rng = np.random.default_rng(123)
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "cafm_cost_device_sample.npz"))
    crop = z["cost_f16_bits"].view(np.float16).astype(np.float64)
    for t in range(200):
        nr, nc = int(rng.integers(1, 140)), int(rng.integers(1, 140))
        kind = t % 5
        if kind == 0:
            c = rng.integers(0, 4, (nr, nc)).astype(np.float64) * 0.25
        elif kind == 1:
            c = crop[np.ix_(rng.integers(0, crop.shape[0], nr), rng.integers(0, crop.shape[1], nc))]
        elif kind == 2:
            c = np.round(rng.random((nr, nc)) * 8) / 8
        elif kind == 3:
            c = np.full((nr, nc), 0.5)
        else:
            c = crop[np.ix_(rng.integers(0, crop.shape[0], nr), rng.integers(0, crop.shape[1], nc))] + 0.01 * rng.standard_normal((nr, nc))
        m = np.ascontiguousarray(c.astype(np.float16).view(np.uint16))
        r, cc = lap_f16(m)
        r0, c0 = linear_sum_assignment(m.view(np.float16).astype(np.float64))
        assert np.array_equal(r, r0) and np.array_equal(cc, c0), (t, kind, nr, nc)


@pytest.mark.parametrize("nr,nc", [(60, 2600), (2600, 60), (300, 700), (1100, 1100)])
def test_lap_fixed_point_wide_matrices_with_ties_match_scipy(nr, nc):
    """The tie rule keeps the columns' positions as 16-bit lanes (lap.cpp: pos16, bit 15 = "no row yet"): matrices whose positions need
    more than 8 and 11 bits, rectangular both ways (the transposed orientation included), with exact ties -- scipy's rows and columns."""
    from tscd_amd._host import lap_f16
    rng = np.random.default_rng(nr * 7 + nc)
    for kind in range(3):
        if kind == 0:
            c = rng.integers(0, 3, (nr, nc)).astype(np.float64) * 0.25          # massive ties: free columns at high positions decide
        elif kind == 1:
            c = np.round(rng.random((nr, nc)) * 16) / 16
        else:
            c = np.full((nr, nc), 0.75)
        m = np.ascontiguousarray(c.astype(np.float16).view(np.uint16))
        r, cc = lap_f16(m)
        r0, c0 = linear_sum_assignment(m.view(np.float16).astype(np.float64))
        assert np.array_equal(r, r0) and np.array_equal(cc, c0), (kind, nr, nc)
