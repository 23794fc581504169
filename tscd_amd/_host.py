"""ctypes binding of libtscd_host.so (tscd_amd/csrc_host/lap.cpp): the host-side linear-sum-assignment solver."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libtscd_host.so")
if not os.path.exists(LIB_PATH):
    raise ImportError(f"{LIB_PATH} not found: run `python -m tscd_amd.build`")
_lib = C.CDLL(LIB_PATH)
_lib.tscd_lap_solve_f64.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p]
_lib.tscd_lap_solve_f64.restype = C.c_int
_lib.tscd_lap_solve_f16.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p]
_lib.tscd_lap_solve_f16.restype = C.c_int


def lap_f64(cost):
    """Same contract as scipy.optimize.linear_sum_assignment(cost) (minimisation), bit-identical result."""
    cost = np.ascontiguousarray(cost, dtype=np.float64)
    nr, nc = cost.shape
    k = min(nr, nc)
    a, b = np.empty(k, np.int64), np.empty(k, np.int64)
    st = _lib.tscd_lap_solve_f64(cost.ctypes.data, nr, nc, a.ctypes.data, b.ctypes.data)
    if st:
        raise ValueError("cost matrix is infeasible" if st == 2 else "matrix contains invalid numeric entries")
    return a, b


def lap_f16(cost_u16, row_map=None):
    """cost_u16: C-contiguous uint16 view of an fp16 [R, nc] matrix; row_map (int64 [nr]) selects/permutes its rows.
    NaNs count as 0 (tscd_matching.py:929-930)."""
    assert cost_u16.dtype == np.uint16 and cost_u16.flags.c_contiguous
    nc = cost_u16.shape[1]
    if row_map is not None:
        row_map = np.ascontiguousarray(row_map, dtype=np.int64)
        nr = row_map.shape[0]
    else:
        nr = cost_u16.shape[0]
    k = min(nr, nc)
    a, b = np.empty(k, np.int64), np.empty(k, np.int64)
    st = _lib.tscd_lap_solve_f16(cost_u16.ctypes.data, nc, None if row_map is None else row_map.ctypes.data, nr, nc,
                                 a.ctypes.data, b.ctypes.data)
    if st:
        raise ValueError("assignment failed (status %d)" % st)
    return a, b


_lib.tscd_lap_chain_f16.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
_lib.tscd_lap_chain_f16.restype = C.c_int
_lib.tscd_lap_chain_set_parallel.argtypes = [C.c_int]
_lib.tscd_lap_chain_set_parallel.restype = C.c_int
_lib.tscd_lap_chain_spec_hits.argtypes = [C.c_int]
_lib.tscd_lap_chain_spec_hits.restype = C.c_int64
_lib.tscd_lap_debug_width.argtypes = [C.c_int, C.c_int]
_lib.tscd_lap_debug_width.restype = C.c_int64
_lib.tscd_lap_debug_steps.argtypes = [C.c_int]
_lib.tscd_lap_debug_steps.restype = C.c_int64
if os.environ.get("TSCD_LAP_PARALLEL", "1") == "0":       # diagnostics / A-B runs: strictly in-order chain
    _lib.tscd_lap_chain_set_parallel(0)
KIND_SELF, KIND_STATE, KIND_PREV = 0, 1, 2


_lib.tscd_lap_chain_f16_table.argtypes = [C.c_int] + [C.c_void_p] * 8 + [C.c_int64, C.c_void_p, C.c_uint64]
_lib.tscd_lap_chain_f16_table.restype = C.c_int
_lib.tscd_latch_create.argtypes, _lib.tscd_latch_create.restype = [], C.c_void_p
_lib.tscd_latch_destroy.argtypes, _lib.tscd_latch_destroy.restype = [C.c_void_p], None
_lib.tscd_latch_signal.argtypes, _lib.tscd_latch_signal.restype = [C.c_void_p, C.c_uint64], None
_lib.tscd_latch_generation.argtypes, _lib.tscd_latch_generation.restype = [C.c_void_p], C.c_uint64
_lib.tscd_latch_timeouts.argtypes, _lib.tscd_latch_timeouts.restype = [C.c_void_p], C.c_int64
_lib.tscd_latch_wait.argtypes, _lib.tscd_latch_wait.restype = [C.c_void_p, C.c_uint64, C.c_int], C.c_int
_lib.tscd_latch_ticket.argtypes, _lib.tscd_latch_ticket.restype = [C.c_void_p, C.c_uint64, C.c_int], C.c_void_p
_lib.tscd_latch_set_spin.argtypes, _lib.tscd_latch_set_spin.restype = [C.c_void_p, C.c_int64], C.c_int64
LATCH_WAIT_CB = C.cast(_lib.tscd_latch_wait_cb, C.c_void_p).value      # address of the hipHostFn_t that consumes a ticket


class Latch:
    """A generation counter a device stream can be parked on (csrc_host/lap.cpp: TscdLatch).  ``ticket(gen)`` is the argument of
    one queued wait (LATCH_WAIT_CB frees it); the object lives as long as the process queues waits on it (never destroyed while a
    wait may be pending: it is kept for the life of the process)."""

    def __init__(self):
        self.handle = _lib.tscd_latch_create()
        self.next_gen = 0
        spin = int(os.environ.get("TSCD_LATCH_SPIN_US", "0"))
        if spin > 0:
            _lib.tscd_latch_set_spin(self.handle, spin)

    def new_generation(self):
        self.next_gen += 1
        return self.next_gen

    def ticket(self, gen, timeout_ms=30000):
        return _lib.tscd_latch_ticket(self.handle, gen, timeout_ms)

    def signal(self, gen):
        _lib.tscd_latch_signal(self.handle, gen)

    def wait(self, gen, timeout_ms=30000):
        return _lib.tscd_latch_wait(self.handle, gen, timeout_ms) == 0

    @property
    def generation(self):
        return _lib.tscd_latch_generation(self.handle)

    @property
    def timeouts(self):
        return _lib.tscd_latch_timeouts(self.handle)


def lap_chain_f16(base_ptr, shapes, kinds, before=None, table_ptr=None, table_len=0, latch=None, gen=0):
    """The clip's dependent chain of assignments in ONE foreign call (no Python between the solves; ctypes drops the GIL, so a
    caller on a worker thread leaves the main thread free to queue device work).

    base_ptr: address of the fp16 cost matrices laid out back to back ([nr_k, nc_k] each, C order); shapes: [(nr_k, nc_k)];
    kinds: KIND_* per matrix; before: optional callable run first on the calling thread (e.g. wait for the copy).
    table_ptr / table_len / latch / gen: see tscd_lap_chain_f16_table (include/tscd_host.h) -- the clamped training regime's index
    table written to pinned memory, then the latch raised; the latch is raised on every path, also when the chain fails.
    Returns [(row_ind, col_ind, perm)] per matrix -- perm: the order in which the reference keeps the frame's rows."""
    try:
        if before is not None:
            before()
    except BaseException:
        if table_ptr is not None and table_len > 0 and shapes:     # ... nor feed it a table left over from a clip of another size
            n = max(int(shapes[0][1]), 1)
            np.copyto(np.ctypeslib.as_array(C.cast(table_ptr, C.POINTER(C.c_int64)), (table_len,)), np.arange(table_len, dtype=np.int64) % n)
        if latch is not None:          # never leave a stream parked on a chain that will not run
            latch.signal(gen)
        raise
    K = len(shapes)
    if K == 0:
        if latch is not None:
            latch.signal(gen)
        return []
    ptrs = (C.c_void_p * K)()
    off = 0
    for i, (r, c) in enumerate(shapes):
        ptrs[i] = base_ptr + 2 * off
        off += r * c
    nr = np.array([s[0] for s in shapes], np.int64)
    nc = np.array([s[1] for s in shapes], np.int64)
    kd = np.array(kinds, np.int32)
    npair = np.minimum(nr, nc)
    rows, cols, perms = np.empty(int(npair.sum()), np.int64), np.empty(int(npair.sum()), np.int64), np.empty(int(nc.sum()), np.int64)
    if table_ptr is not None:
        # the clamped regime: the index table is written to pinned memory and the latch raised inside the call (no interpreter
        # between the last solve and the device's go-ahead)
        st = _lib.tscd_lap_chain_f16_table(K, C.cast(ptrs, C.c_void_p), nr.ctypes.data, nc.ctypes.data, kd.ctypes.data, rows.ctypes.data,
                                           cols.ctypes.data, perms.ctypes.data, table_ptr, table_len, latch.handle if latch else None, gen)
    else:
        st = _lib.tscd_lap_chain_f16(K, C.cast(ptrs, C.c_void_p), nr.ctypes.data, nc.ctypes.data, kd.ctypes.data, rows.ctypes.data,
                                     cols.ctypes.data, perms.ctypes.data)
        if latch is not None:
            latch.signal(gen)
    if st:
        raise ValueError("assignment chain failed at matrix %d (status %d)" % (st // 16, st % 16))
    out, a, b = [], 0, 0
    for i in range(K):
        out.append((rows[a:a + npair[i]], cols[a:a + npair[i]], perms[b:b + nc[i]]))
        a += int(npair[i])
        b += int(nc[i])
    return out


_lib.tscd_cocoeval_bbox.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_long, C.c_void_p, C.c_void_p, C.c_void_p,
                                    C.c_void_p, C.c_long, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
_lib.tscd_cocoeval_bbox.restype = C.c_int


def cocoeval_bbox(gt_box, gt_img, gt_cat, gt_crowd, gt_area, dt_box, dt_img, dt_cat, dt_score, n_img, n_cat, full=False):
    """COCO bbox evaluation (csrc_host/cocoeval.cpp).  Boxes xywh float32 [n,4]; image / category ids dense int32 indices.
    -> stats [12] (COCOeval.summarize order) and, with ``full``, precision [10,101,K,4,3] and recall [10,K,4,3]."""
    f32 = lambda a, shape: np.ascontiguousarray(np.asarray(a, dtype=np.float32).reshape(shape))
    i32 = lambda a: np.ascontiguousarray(np.asarray(a, dtype=np.int32).reshape(-1))
    gt_box, dt_box = f32(gt_box, (-1, 4)), f32(dt_box, (-1, 4))
    gt_img, gt_cat, dt_img, dt_cat = i32(gt_img), i32(gt_cat), i32(dt_img), i32(dt_cat)
    gt_crowd = np.ascontiguousarray(np.asarray(gt_crowd, dtype=np.uint8).reshape(-1))
    gt_area, dt_score = f32(gt_area, (-1,)), f32(dt_score, (-1,))
    stats = np.empty(12, np.float32)
    prec = np.empty((10, 101, n_cat, 4, 3), np.float32) if full else None
    rec = np.empty((10, n_cat, 4, 3), np.float32) if full else None
    st = _lib.tscd_cocoeval_bbox(gt_box.ctypes.data, gt_img.ctypes.data, gt_cat.ctypes.data, gt_crowd.ctypes.data, gt_area.ctypes.data,
                                 len(gt_img), dt_box.ctypes.data, dt_img.ctypes.data, dt_cat.ctypes.data, dt_score.ctypes.data,
                                 len(dt_img), n_img, n_cat, stats.ctypes.data, None if prec is None else prec.ctypes.data,
                                 None if rec is None else rec.ctypes.data)
    if st:
        raise ValueError("cocoeval_bbox failed (status %d)" % st)
    return (stats, prec, rec) if full else stats
