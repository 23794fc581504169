/* Host-side C-ABI of the TSCD hot path (libtscd_host.so, built from tscd_amd/csrc_host/lap.cpp with g++).
 *
 * The one step of the path the reference itself runs on the host: the Hungarian assignment of CAFM's frame-to-frame
 * matching, double_match_embds (yolox/models/tscd_matching.py:908-937), which calls
 * scipy.optimize.linear_sum_assignment on a cost matrix copied from the device.  scipy is a third-party dependency of the
 * reference (not vendored); its solver is the shortest-augmenting-path algorithm of D. F. Crouse, "On implementing 2D
 * rectangular assignment algorithms", IEEE TAES 52(4), 2016.  These entry points restate that algorithm with scipy's
 * iteration order and tie rules, so the returned indices equal scipy's for every input (tests/test_lap_host.py).
 *
 * Plain pointers and sizes, no allocation visible to the caller, thread-safe (per-thread scratch).  Return codes:
 * 0 ok, 1 invalid entries (NaN / -inf in the f64 form), 2 infeasible.
 */
#ifndef TSCD_HOST_H
#define TSCD_HOST_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* cost: row-major [nr][nc] float64.  row_ind / col_ind: min(nr, nc) entries, rows ascending (as scipy returns them). */
int tscd_lap_solve_f64(const double* cost, int64_t nr, int64_t nc, int64_t* row_ind, int64_t* col_ind);

/* cost: fp16 bit patterns as the device GEMM (EPI_COST epilogue of tscd_conv2d_nhwc_f16) wrote them, leading dimension
 * ld (elements).  row_map (may be NULL): row r of the solved matrix is cost[row_map[r]] -- the reference solves on the
 * previous frame's rows re-ordered by the previous assignment.  NaN entries count as 0 like the reference's
 * torch.where(isnan).  fp16 costs are exact multiples of 2^-24, so the solver runs in exact int32 fixed point with
 * AVX-512 inner loops when the host has them (tscd_lap_fixed_point_available) and |cost| < 8, otherwise in double. */
int tscd_lap_solve_f16(const uint16_t* cost, int64_t ld, const int64_t* row_map, int64_t nr, int64_t nc, int64_t* row_ind,
                       int64_t* col_ind);
int tscd_lap_fixed_point_available(void);

/* The dependent chain of assignments of one clip in one call (AwarePositionRegMatcher.forward, tscd_matching.py:940-1080).
 * kind[k]: 0 first frame of a clip (self match), 1 against carried-over state (rows as given), 2 against the previous frame
 * (rows gathered by the previous solve's column order; nr[k] == nc[k-1]).  cost[k]: fp16 bits [nr[k]][nc[k]].  Outputs are
 * concatenated over k: row_out / col_out hold min(nr,nc) scipy pairs each, perm_out nc[k] entries (the pairs' columns, then
 * the unmatched columns ascending; the identity for kind 0).  Returns 0, or 16*k + status of the first failing solve. */
int tscd_lap_chain_f16(int K, const uint16_t* const* cost, const int64_t* nr, const int64_t* nc, const int32_t* kind,
                       int64_t* row_out, int64_t* col_out, int64_t* perm_out);

/* The chain's matrices depend on each other only through the ORDER of their rows, so they are solved side by side (identity
 * order, a small persistent thread pool) and the in-order answer is recovered by relabelling wherever the optimum is certified
 * unique from the exact integer duals (no alternating cycle of tight entries, no free move to an unassigned column); a matrix
 * that fails the certificate is solved again strictly in order, so the result is always the reference's.
 * The side-by-side pass is adaptive: after a chain where fewer than half the matrices were certified (tie-heavy costs, e.g. a
 * randomly initialised network) it pauses and is probed again every 64th chain.  tscd_lap_chain_set_parallel(0) forces the
 * strictly sequential chain, (2) the side-by-side pass on every chain, (1) the adaptive default (returns the previous setting);
 * tscd_lap_chain_spec_hits counts matrices answered from the side-by-side solve. */
int tscd_lap_chain_set_parallel(int on);
int64_t tscd_lap_chain_spec_hits(int reset);

/* A latch a device stream can be parked on, and the chain call that raises it.  The train step queues everything that follows
 * the assignments (CAFM, refined heads, the whole backward) while the chain is still being solved on a worker thread: the caller
 * queues tscd_latch_wait_cb with a ticket through include/tscd_hip.h's tscd_stream_host_func -- hipLaunchHostFunc --, then the copy of
 * the index table from pinned memory, then the kernels.  The reference has no counterpart: it blocks the interpreter in
 * linear_sum_assignment (tscd_matching.py:935).  Generations only grow, so one latch serves every step.  tscd_latch_wait
 * returns 0 once generation `gen` is reached, 1 after timeout_ms (counted by tscd_latch_timeouts). */
void* tscd_latch_create(void);
void tscd_latch_destroy(void* latch);
void tscd_latch_signal(void* latch, uint64_t gen);
uint64_t tscd_latch_generation(void* latch);
int64_t tscd_latch_timeouts(void* latch);
int tscd_latch_wait(void* latch, uint64_t gen, int timeout_ms);
/* A wait polls the counter for up to `us` microseconds before it blocks on the condition variable (0, the default: blocks at once):
 * removes the futex wake-up between the worker's signal and the stream's go-ahead at the price of a busy runtime thread. */
int64_t tscd_latch_set_spin(void* latch, int64_t us);
void* tscd_latch_ticket(void* latch, uint64_t gen, int timeout_ms);   /* heap ticket, consumed by tscd_latch_wait_cb */
void tscd_latch_wait_cb(void* ticket);                                /* hipHostFn_t */
/* tscd_lap_chain_f16 for the clamped training regime (K frames of n proposals each, kind = 0,2,2,...): also writes the clip's
 * index table [inv_0 | c_1 | inv_1 | ... ] (n int64 each; c_k = perm_out of frame k, inv_k its inverse, inv_0 the inverse of the
 * first frame's self assignment; table_len = (2K-1)*n) and then raises `latch` to `gen` -- on every path; on failure (status as
 * tscd_lap_chain_f16, 5 = shapes outside the regime) the table holds identity permutations. */
int tscd_lap_chain_f16_table(int K, const uint16_t* const* cost, const int64_t* nr, const int64_t* nc, const int32_t* kind,
                             int64_t* row_out, int64_t* col_out, int64_t* perm_out, int64_t* table, int64_t table_len, void* latch,
                             uint64_t gen);

/* Diagnostics: relaxation passes executed by the fixed-point solver since the last reset. */
int64_t tscd_lap_debug_steps(int reset);
/* Diagnostics: solves answered on the int16 (which = 0) / int32 (which = 1) fixed-point matrix so far. */
int64_t tscd_lap_debug_width(int which, int reset);

/* COCO bounding-box evaluation (replaces the reference's pybind11 extension
 * yolox/layers/cocoeval/cocoeval.cpp:59-199,370-502 and its pycocotools fallback, ovis_evaluator_v2.py:397-402).
 * Boxes are xywh; image and category ids are dense indices 0..n-1; gt_area is the annotation's "area" field.
 * stats12 receives the 12 numbers of COCOeval.summarize() (AP@[.5:.95], AP50, AP75, AP small / medium / large, AR@1 / 10 / 100,
 * AR small / medium / large; -1 where undefined).  Optional outputs: precision [10][101][n_cat][4][3] and recall
 * [10][n_cat][4][3] (IoU threshold, recall threshold, category, area range {all, small, medium, large}, maxDets {1,10,100}),
 * -1 = no ground truth.  Returns 0, 1 on bad sizes, 2 on an out-of-range image / category index. */
int tscd_cocoeval_bbox(const float* gt_box, const int* gt_img, const int* gt_cat, const unsigned char* gt_crowd,
                       const float* gt_area, long n_gt, const float* dt_box, const int* dt_img, const int* dt_cat,
                       const float* dt_score, long n_dt, int n_img, int n_cat, float* stats12, float* precision_out,
                       float* recall_out);

#ifdef __cplusplus
}
#endif
#endif
