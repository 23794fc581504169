// Host-side linear sum assignment for CAFM's frame-to-frame matching (reference: double_match_embds,
// yolox/models/tscd_matching.py:912-937, which calls scipy.optimize.linear_sum_assignment on the host).
//
// scipy is a third-party dependency of the reference; its solver is the shortest-augmenting-path algorithm of
// D. F. Crouse, "On implementing 2D rectangular assignment algorithms", IEEE TAES 52(4), 2016.  This file restates
// that published algorithm with the SAME iteration order and tie rules as scipy's solver (columns scanned in the order
// of a swap-removal "remaining" list that starts reversed; among equal shortest-path costs a column without a row
// wins), so the permutation is identical to scipy's for every input -- tests/test_lap_host.py checks exact equality
// on tie-heavy (fp16-quantised, small-integer) and rectangular matrices.  What changes is the inner loop: instead of
// a scalar scan through the indirection list it is a contiguous, auto-vectorised pass over all columns (AVX-512 /
// AVX2 clones selected at load time) plus a short scalar pass over the tied minima.  500x500 fp16-quantised random
// costs: ~1 ms vs 8-12 ms.
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <vector>
#include <algorithm>
#include <numeric>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <functional>
#include <mutex>
#include <thread>
#include <unistd.h>

namespace {

struct Work {
  std::vector<double> u, v, spc, pathd, dead, blockmin;   // pathd: predecessor row as double (exact for n < 2^53) so the hot loop is all-double
  std::vector<int64_t> col4row, row4col, remaining, pos;
  std::vector<uint8_t> SR;
};

// One relaxation pass from row i: spc[j] = min(spc[j], minVal + cost[i][j] - u_i - v[j]) for columns not yet in SC;
// returns the minimum spc over those columns.  Expression order matches scipy: ((minVal + c) - u_i) - v_j.
// `dead[j]` is 0.0 for live columns and +inf for columns already in SC: adding it keeps live values bit-identical and
// removes dead columns from both the update and the minimum without integer masks (pure-double loop => 8-wide AVX-512).
constexpr int64_t BLK = 32;   // column padding granule (arrays are padded with permanently-dead columns)

// relax_row: spc[j] = min(spc[j], ((minVal + c[j]) - u_i) - v[j]) over live columns, returns the minimum live spc.
// find_ties: visits every live column whose spc equals `lowest`.
// Explicit SIMD (the auto-vectoriser gives up on the select/min mix); the arithmetic is the same IEEE double sequence in
// every variant, so all variants return bit-identical results.
#if defined(__x86_64__)
#include <immintrin.h>
__attribute__((target("avx512f")))
static double relax_row_avx512(const double* crow, const double* v, double* spc, double* pathd, const double* dead, int64_t ncp,
                               double minVal, double ui, double id) {
  const __m512d vmin = _mm512_set1_pd(minVal), vui = _mm512_set1_pd(ui), vid = _mm512_set1_pd(id);
  __m512d low = _mm512_set1_pd(std::numeric_limits<double>::infinity());
  for (int64_t j = 0; j < ncp; j += 8) {
    __m512d d = _mm512_loadu_pd(dead + j);
    __m512d r = _mm512_add_pd(_mm512_sub_pd(_mm512_sub_pd(_mm512_add_pd(vmin, _mm512_loadu_pd(crow + j)), vui), _mm512_loadu_pd(v + j)), d);
    __m512d s = _mm512_loadu_pd(spc + j);
    __mmask8 m = _mm512_cmp_pd_mask(r, s, _CMP_LT_OQ);
    s = _mm512_mask_blend_pd(m, s, r);
    _mm512_storeu_pd(spc + j, s);
    _mm512_mask_storeu_pd(pathd + j, m, vid);
    __m512d e = _mm512_add_pd(s, d);
    low = _mm512_mask_blend_pd(_mm512_cmp_pd_mask(e, low, _CMP_LT_OQ), low, e);
  }
  return _mm512_reduce_min_pd(low);
}
__attribute__((target("avx512f")))
static uint64_t tie_mask8_avx512(const double* spc, const double* dead, int64_t j, double lowest) {
  return _mm512_cmp_pd_mask(_mm512_add_pd(_mm512_loadu_pd(spc + j), _mm512_loadu_pd(dead + j)), _mm512_set1_pd(lowest), _CMP_EQ_OQ);
}
__attribute__((target("avx2")))
static double relax_row_avx2(const double* crow, const double* v, double* spc, double* pathd, const double* dead, int64_t ncp,
                             double minVal, double ui, double id) {
  const __m256d vmin = _mm256_set1_pd(minVal), vui = _mm256_set1_pd(ui), vid = _mm256_set1_pd(id);
  __m256d low = _mm256_set1_pd(std::numeric_limits<double>::infinity());
  for (int64_t j = 0; j < ncp; j += 4) {
    __m256d d = _mm256_loadu_pd(dead + j);
    __m256d r = _mm256_add_pd(_mm256_sub_pd(_mm256_sub_pd(_mm256_add_pd(vmin, _mm256_loadu_pd(crow + j)), vui), _mm256_loadu_pd(v + j)), d);
    __m256d s = _mm256_loadu_pd(spc + j);
    __m256d m = _mm256_cmp_pd(r, s, _CMP_LT_OQ);
    s = _mm256_blendv_pd(s, r, m);
    _mm256_storeu_pd(spc + j, s);
    _mm256_storeu_pd(pathd + j, _mm256_blendv_pd(_mm256_loadu_pd(pathd + j), vid, m));
    __m256d e = _mm256_add_pd(s, d);
    low = _mm256_blendv_pd(low, e, _mm256_cmp_pd(e, low, _CMP_LT_OQ));
  }
  double t[4];
  _mm256_storeu_pd(t, low);
  double a = t[0] < t[1] ? t[0] : t[1], b = t[2] < t[3] ? t[2] : t[3];
  return a < b ? a : b;
}
static const int g_simd = __builtin_cpu_supports("avx512f") ? 2 : (__builtin_cpu_supports("avx2") ? 1 : 0);
#else
static const int g_simd = 0;
#endif

static double relax_row_scalar(const double* crow, const double* v, double* spc, double* pathd, const double* dead, int64_t ncp,
                               double minVal, double ui, double id) {
  double lowest = std::numeric_limits<double>::infinity();
  for (int64_t j = 0; j < ncp; ++j) {
    double r = (minVal + crow[j] - ui - v[j]) + dead[j];
    if (r < spc[j]) { spc[j] = r; pathd[j] = id; }
    double e = spc[j] + dead[j];
    if (e < lowest) lowest = e;
  }
  return lowest;
}

static inline double relax_row(const double* crow, const double* v, double* spc, double* pathd, const double* dead, int64_t ncp,
                               double minVal, double ui, double id) {
#if defined(__x86_64__)
  if (g_simd == 2) return relax_row_avx512(crow, v, spc, pathd, dead, ncp, minVal, ui, id);
  if (g_simd == 1) return relax_row_avx2(crow, v, spc, pathd, dead, ncp, minVal, ui, id);
#endif
  return relax_row_scalar(crow, v, spc, pathd, dead, ncp, minVal, ui, id);
}

int64_t augmenting_path(int64_t nc, int64_t ncp, const double* cost, Work& w, int64_t i, double* p_minVal) {
  double minVal = 0;
  int64_t num_remaining = nc;
  for (int64_t it = 0; it < nc; ++it) {
    w.remaining[it] = nc - it - 1;      // reversed start: a constant matrix solves to the identity (scipy #11602)
    w.pos[nc - it - 1] = it;
  }
  std::fill(w.SR.begin(), w.SR.end(), 0);
  std::fill(w.dead.begin(), w.dead.begin() + nc, 0.0);   // padding columns [nc, ncp) stay dead (+inf)
  std::fill(w.spc.begin(), w.spc.end(), std::numeric_limits<double>::infinity());
  int64_t sink = -1;
  while (sink == -1) {
    w.SR[i] = 1;
    double lowest = relax_row(cost + i * ncp, w.v.data(), w.spc.data(), w.pathd.data(), w.dead.data(), ncp, minVal, w.u[i],
                              (double)i);
    if (lowest == std::numeric_limits<double>::infinity()) return -1;   // infeasible
    // tie rule of the scalar scan: the first column (in `remaining` order) reaching the minimum is taken, and every
    // later tied column that has no row yet replaces it.
    int64_t index = -1, first_pos = std::numeric_limits<int64_t>::max(), last_free_pos = -1;
    auto visit = [&](int64_t j) {
      int64_t p = w.pos[j];
      if (p < first_pos) first_pos = p;
      if (w.row4col[j] == -1 && p > last_free_pos) last_free_pos = p;
    };
#if defined(__x86_64__)
    if (g_simd == 2) {
      for (int64_t j = 0; j < ncp; j += 8) {
        uint64_t m = tie_mask8_avx512(w.spc.data(), w.dead.data(), j, lowest);
        while (m) { visit(j + __builtin_ctzll(m)); m &= m - 1; }
      }
    } else
#endif
    {
      for (int64_t j = 0; j < ncp; ++j)
        if (w.spc[j] + w.dead[j] == lowest) visit(j);
    }
    index = (last_free_pos > first_pos) ? last_free_pos : first_pos;
    minVal = lowest;
    int64_t j = w.remaining[index];
    if (w.row4col[j] == -1) sink = j; else i = w.row4col[j];
    w.dead[j] = std::numeric_limits<double>::infinity();
    --num_remaining;
    int64_t moved = w.remaining[num_remaining];
    w.remaining[index] = moved;
    w.pos[moved] = index;
  }
  *p_minVal = minVal;
  return sink;
}

int solve(int64_t nr, int64_t nc, const double* cost_in, int64_t* a, int64_t* b) {
  if (nr == 0 || nc == 0) return 0;
  const bool transpose = nc < nr;
  std::vector<double> temp;
  const double* cost = cost_in;
  if (transpose) {
    temp.resize((size_t)nr * nc);
    for (int64_t i = 0; i < nr; ++i)
      for (int64_t j = 0; j < nc; ++j) temp[(size_t)j * nr + i] = cost_in[(size_t)i * nc + j];
    std::swap(nr, nc);
    cost = temp.data();
  }
  for (int64_t k = 0; k < nr * nc; ++k)
    if (cost[k] != cost[k] || cost[k] == -std::numeric_limits<double>::infinity()) return 1;   // invalid entries
  // pad rows to a multiple of BLK columns (padding is permanently "dead")
  const int64_t ncp = (nc + BLK - 1) / BLK * BLK;
  std::vector<double> padded((size_t)nr * ncp, 0.0);
  for (int64_t i = 0; i < nr; ++i) std::memcpy(&padded[(size_t)i * ncp], cost + (size_t)i * nc, sizeof(double) * nc);
  Work w;
  w.u.assign(nr, 0.0); w.v.assign(ncp, 0.0); w.spc.resize(ncp); w.blockmin.resize(ncp / BLK);
  w.pathd.assign(ncp, -1.0); w.col4row.assign(nr, -1); w.row4col.assign(ncp, -1);
  w.SR.resize(nr); w.dead.assign(ncp, std::numeric_limits<double>::infinity()); w.remaining.resize(nc); w.pos.resize(ncp);
  for (int64_t curRow = 0; curRow < nr; ++curRow) {
    double minVal;
    int64_t sink = augmenting_path(nc, ncp, padded.data(), w, curRow, &minVal);
    if (sink < 0) return 2;   // infeasible
    w.u[curRow] += minVal;
    for (int64_t i = 0; i < nr; ++i)
      if (w.SR[i] && i != curRow) w.u[i] += minVal - w.spc[w.col4row[i]];
    for (int64_t j = 0; j < nc; ++j)
      if (w.dead[j] != 0.0) w.v[j] -= minVal - w.spc[j];
    int64_t j = sink;
    while (true) {
      int64_t i = (int64_t)w.pathd[j];
      w.row4col[j] = i;
      std::swap(w.col4row[i], j);
      if (i == curRow) break;
    }
  }
  if (transpose) {
    std::vector<int64_t> order(nr);
    std::iota(order.begin(), order.end(), 0);
    std::sort(order.begin(), order.end(), [&](int64_t x, int64_t y) { return w.col4row[x] < w.col4row[y]; });
    for (int64_t i = 0; i < nr; ++i) { a[i] = w.col4row[order[i]]; b[i] = order[i]; }
  } else {
    for (int64_t i = 0; i < nr; ++i) { a[i] = i; b[i] = w.col4row[i]; }
  }
  return 0;
}

// --------------------------------------------------------------------------------------------------------------------
// Fixed-point variant for fp16 cost matrices (what the device GEMM hands over).  Every fp16 value below 8.0 is an exact
// multiple of 2^-24, so costs, dual variables and path lengths are exact int32 in that unit; the double-precision
// sequence above is exact on the same inputs too (all intermediate magnitudes < 2^10 with 2^-24 granularity fit 53 bits),
// hence both variants -- and scipy -- walk through identical states.  16 columns per AVX-512 op, branch-free relaxation and
// tie rule, int16 predecessor stores.  Any magnitude reaching 2^27 units (8.0) aborts
// with status 3 and the caller falls back to the double solver.
#if defined(__x86_64__)
// 64-byte aligned storage: every 512-bit access below stays inside one cache line (no split loads / failed store forwarding)
template <typename T>
struct AlignedBuf {
  T* p = nullptr;
  size_t n = 0;
  AlignedBuf(size_t count, T fill) : n(count) {
    p = (T*)std::aligned_alloc(64, (count * sizeof(T) + 63) / 64 * 64);
    for (size_t i = 0; i < count; ++i) p[i] = fill;
  }
  explicit AlignedBuf(size_t count) : n(count) { p = (T*)std::aligned_alloc(64, (count * sizeof(T) + 63) / 64 * 64); }
  ~AlignedBuf() { std::free(p); }
  AlignedBuf(const AlignedBuf&) = delete;
  AlignedBuf& operator=(const AlignedBuf&) = delete;
  T* data() { return p; }
  T& operator[](size_t i) { return p[i]; }
  T* begin() { return p; }
  T* end() { return p + n; }
};
constexpr int32_t QLIM = 1 << 27, QINF = 1 << 30, QDEADV = -(QINF + (QINF >> 1));
static std::atomic<int64_t> g_steps{0};        // relaxation passes of the last solves (diagnostics; solves may run on pool threads)

// One relaxation pass from row i, mask-free: columns already in SC carry veff = QDEADV (so their candidate exceeds QINF)
// and spc = QINF (their real spc is parked in spcsave), which removes them from the update, the minimum and the tie test
// without predicate loads.  Branch-free on purpose: data-dependent branches here mispredict.
__attribute__((target("avx512f,avx512bw,avx512vl,f16c")))
static inline void set_lane(int32_t* arr, int64_t j, int32_t val) {     // arr 64-byte aligned
  int32_t* blk = arr + (j & ~(int64_t)15);
  _mm512_store_si512(blk, _mm512_mask_mov_epi32(_mm512_load_si512(blk), (__mmask16)(1u << (j & 15)), _mm512_set1_epi32(val)));
}
// positions in `remaining` order as 16-bit lanes (nc < 32000): 32 columns per op in the tie rule, ONE array.  pos16[j] = position of
// column j, bit 15 set while the column has no row (padding: 0x7fff).  Smallest position among the ties = min of (pos16 & 0x7fff);
// LARGEST position among the ties without a row = 0x7fff - min of (pos16 ^ 0xffff) (>= 0x8000 for a column that has a row).
__attribute__((target("avx512f,avx512bw,avx512vl,f16c")))
static inline void set_lane16(uint16_t* arr, int64_t j, uint16_t val) {     // arr 64-byte aligned
  uint16_t* blk = arr + (j & ~(int64_t)31);
  _mm512_store_si512(blk, _mm512_mask_mov_epi16(_mm512_load_si512(blk), (__mmask32)(1u << (j & 31)), _mm512_set1_epi16((short)val)));
}
__attribute__((target("avx512f,avx512bw,avx512vl,f16c")))
static inline uint32_t hmin_epu16(__m512i v) {
  const __m256i a = _mm256_min_epu16(_mm512_castsi512_si256(v), _mm512_extracti64x4_epi64(v, 1));
  const __m128i b = _mm_min_epu16(_mm256_castsi256_si128(a), _mm256_extracti128_si256(a, 1));
  return (uint32_t)_mm_extract_epi16(_mm_minpos_epu16(b), 0) & 0xffffu;
}
template <typename CT>
__attribute__((target("avx512f,avx512bw,avx512vl,f16c")))
static inline __m512i load16(const CT* p) {
  if constexpr (sizeof(CT) == 2) return _mm512_cvtepi16_epi32(_mm256_loadu_si256((const __m256i*)p));
  else return _mm512_loadu_si512(p);
}
template <typename CT>
__attribute__((target("avx512f,avx512bw,avx512vl,f16c")))
static int32_t relax_row_q(const CT* crow, const int32_t* veff, int32_t* spc, int16_t* path, int64_t ncp, int32_t base, int16_t id) {
  const __m512i vbase = _mm512_set1_epi32(base);
  const __m512i vid = _mm512_set1_epi16(id);
  __m512i low0 = _mm512_set1_epi32(QINF), low1 = low0;
  for (int64_t j = 0; j < ncp; j += 32) {
    __m512i r0 = _mm512_sub_epi32(_mm512_add_epi32(load16<CT>(crow + j), vbase), _mm512_loadu_si512(veff + j));
    __m512i r1 = _mm512_sub_epi32(_mm512_add_epi32(load16<CT>(crow + j + 16), vbase), _mm512_loadu_si512(veff + j + 16));
    __m512i s0 = _mm512_loadu_si512(spc + j), s1 = _mm512_loadu_si512(spc + j + 16);
    __mmask16 m0 = _mm512_cmplt_epi32_mask(r0, s0), m1 = _mm512_cmplt_epi32_mask(r1, s1);
    s0 = _mm512_min_epi32(s0, r0);
    s1 = _mm512_min_epi32(s1, r1);
    _mm512_storeu_si512(spc + j, s0);
    _mm512_storeu_si512(spc + j + 16, s1);
    _mm512_mask_storeu_epi16(path + j, (__mmask32)m0 | ((__mmask32)m1 << 16), vid);
    low0 = _mm512_min_epi32(low0, s0);
    low1 = _mm512_min_epi32(low1, s1);
  }
  return _mm512_reduce_min_epi32(_mm512_min_epi32(low0, low1));
}
// The same relaxation for a pass that is expected to stay on the current PLATEAU (the minimum of the previous pass, `level`): path
// lengths never fall below it, so the columns at the minimum are exactly the live columns with spc == level -- known per 16 lanes
// as soon as they are relaxed -- and the tie rule (smallest position among them, largest position among those without a row) is
// accumulated in the same sweep; the two reductions that follow are independent of each other.  On the near-degenerate costs of a
// CAFM at random initialisation 93 % of the passes stay on their plateau (8 tied columns on average); the others (no column at
// `level`: first_pos stays INT32_MAX) take the minimum and the tie rule in two further sweeps, as every pass did before (one relaxing
// sweep, a reduction, one tie sweep that waited for it, two reductions: 55 ns per pass on the box's EPYC 9575F, of which the tie sweep
// and its reductions were 45 %).  Second half of round 4: the positions as 16-bit lanes in ONE array (32 columns per masked min, both
// reductions through phminposuw; four loads and two reductions per 32 columns less) 50.7 -> 48.0 ns, and the next pass's cost row
// prefetched as soon as its index is known -> 45.5 ns (same-box A/B against the previous file, results scipy-checked).
template <typename CT>
__attribute__((target("avx512f,avx512bw,avx512vl,f16c")))
static inline void relax_row_plateau(const CT* crow, const int32_t* veff, int32_t* spc, int16_t* path, const uint16_t* pos16,
                                     int64_t ncp, int32_t base, int16_t id, int32_t level, int32_t& first_pos,
                                     int32_t& last_free_pos) {
  const __m512i vbase = _mm512_set1_epi32(base), vlevel = _mm512_set1_epi32(level);
  const __m512i vid = _mm512_set1_epi16(id);
  const __m512i v7fff = _mm512_set1_epi16(0x7fff), vffff = _mm512_set1_epi16((short)0xffff);
  __m512i pmin = vffff, imin = vffff;
  for (int64_t j = 0; j < ncp; j += 32) {
    __m512i r0 = _mm512_sub_epi32(_mm512_add_epi32(load16<CT>(crow + j), vbase), _mm512_loadu_si512(veff + j));
    __m512i r1 = _mm512_sub_epi32(_mm512_add_epi32(load16<CT>(crow + j + 16), vbase), _mm512_loadu_si512(veff + j + 16));
    __m512i s0 = _mm512_loadu_si512(spc + j), s1 = _mm512_loadu_si512(spc + j + 16);
    __mmask16 m0 = _mm512_cmplt_epi32_mask(r0, s0), m1 = _mm512_cmplt_epi32_mask(r1, s1);
    s0 = _mm512_min_epi32(s0, r0);
    s1 = _mm512_min_epi32(s1, r1);
    _mm512_storeu_si512(spc + j, s0);
    _mm512_storeu_si512(spc + j + 16, s1);
    _mm512_mask_storeu_epi16(path + j, (__mmask32)m0 | ((__mmask32)m1 << 16), vid);
    const __mmask16 e0 = _mm512_cmpeq_epi32_mask(s0, vlevel), e1 = _mm512_cmpeq_epi32_mask(s1, vlevel);
    const __mmask32 e = (__mmask32)e0 | ((__mmask32)e1 << 16);
    const __m512i pf = _mm512_load_si512(pos16 + j);
    pmin = _mm512_mask_min_epu16(pmin, e, pmin, _mm512_and_si512(pf, v7fff));
    imin = _mm512_mask_min_epu16(imin, e, imin, _mm512_xor_si512(pf, vffff));      // free: 0x7fff - pos; with a row: >= 0x8000
  }
  const uint32_t fp = hmin_epu16(pmin), fi = hmin_epu16(imin);
  first_pos = fp >= 0x7fffu ? std::numeric_limits<int32_t>::max() : (int32_t)fp;
  last_free_pos = fi >= 0x8000u ? -1 : (int32_t)(0x7fffu - fi);
}
template <typename CT>
__attribute__((target("avx512f,avx512bw,avx512vl,f16c")))
static inline int32_t min_of(const int32_t* spc, int64_t ncp) {
  __m512i low0 = _mm512_set1_epi32(QINF), low1 = low0;
  for (int64_t j = 0; j < ncp; j += 32) {
    low0 = _mm512_min_epi32(low0, _mm512_load_si512(spc + j));
    low1 = _mm512_min_epi32(low1, _mm512_load_si512(spc + j + 16));
  }
  return _mm512_reduce_min_epi32(_mm512_min_epi32(low0, low1));
}
// fp16 -> integers in units of 2^(E-25) (E = smallest biased exponent among the non-zero entries, at least 1): exact, since
// every fp16 value with exponent field e >= E is a multiple of that unit.  Scaling all costs by one power of two changes no
// comparison and no tie in the algorithm.
template <typename CT>
__attribute__((target("avx512f,avx512bw,avx512vl,f16c")))
static void convert_q(const uint16_t* src, CT* dst, int64_t n, int E) {   // n % 16 == 0
  const __m512 scale = _mm512_set1_ps(std::ldexp(1.0f, 25 - E));
  for (int64_t j = 0; j < n; j += 16) {
    __m512i v = _mm512_cvtps_epi32(_mm512_mul_ps(_mm512_cvtph_ps(_mm256_loadu_si256((const __m256i*)(src + j))), scale));
    if constexpr (sizeof(CT) == 2) _mm256_storeu_si256((__m256i*)(dst + j), _mm512_cvtepi32_epi16(v));
    else _mm512_storeu_si512(dst + j, v);
  }
}
__attribute__((target("avx512f,avx512bw,avx512vl,f16c")))
static void exponent_span(const uint16_t* src, int64_t n, int& emin, int& emax) {   // n % 32 == 0
  const __m512i m15 = _mm512_set1_epi16(0x7fff), one = _mm512_set1_epi16(1);
  __m512i lo = _mm512_set1_epi16(31), hi = _mm512_setzero_si512();
  for (int64_t j = 0; j < n; j += 32) {
    __m512i h = _mm512_and_si512(_mm512_loadu_si512(src + j), m15);
    __mmask32 nz = _mm512_test_epi16_mask(h, h);
    __m512i e = _mm512_max_epu16(_mm512_srli_epi16(h, 10), one);       // subnormals share exponent 1's unit
    lo = _mm512_mask_min_epu16(lo, nz, lo, e);
    hi = _mm512_mask_max_epu16(hi, nz, hi, e);
  }
  alignas(64) uint16_t a[32], b[32];
  _mm512_store_si512(a, lo);
  _mm512_store_si512(b, hi);
  unsigned l = 31, g = 0;
  for (int k = 0; k < 32; ++k) { l = a[k] < l ? a[k] : l; g = b[k] > g ? b[k] : g; }
  emin = (int)l; emax = (int)g;
}

// Uniqueness certificate (exact, integer duals).  At termination (u, v) is dual feasible, every assigned pair is tight
// (c - u - v == 0) and unassigned columns keep v == 0 <= every other v.  With nc - nr dummy rows (cost 0, dual 0: tight exactly
// on the columns with v == 0) the optimal solutions are the perfect matchings of the tight graph, so the real rows' columns
// are the same in every optimal solution iff, in the column digraph  j -> j'  (the row assigned to j has another tight entry
// in j'),  (a) no cycle runs through assigned columns and (b) no assigned column with v == 0 reaches an unassigned column.
// A unique optimum does not depend on the order in which rows / columns are presented: the speculative identity-order solve of
// tscd_lap_chain_f16 is then the reference's answer up to relabelling.
template <typename CT>
static bool unique_optimum(int64_t nr, int64_t nc, int64_t ncp, const CT* c, const int32_t* u, const int32_t* v, const int32_t* col4row,
                           const int32_t* row4col) {
  std::vector<int32_t> start(nc + 1, 0), adj;
  adj.reserve(4 * nr);
  for (int64_t j = 0; j < nc; ++j) {
    const int32_t i = row4col[j];
    if (i >= 0) {
      const CT* crow = c + (int64_t)i * ncp;
      const int32_t ui = u[i];
      for (int64_t t = 0; t < nc; ++t)
        if (t != j && (int32_t)crow[t] - ui - v[t] == 0) adj.push_back((int32_t)t);
    }
    start[j + 1] = (int32_t)adj.size();
  }
  // iterative DFS over assigned columns: colour 0 new / 1 on stack / 2 done; reach[j]: an unassigned column is reachable from j
  std::vector<uint8_t> colour(nc, 0), reach(nc, 0);
  std::vector<int32_t> stack, edge;
  for (int64_t root = 0; root < nc; ++root) {
    if (row4col[root] < 0 || colour[root]) continue;
    stack.assign(1, (int32_t)root);
    edge.assign(1, start[root]);
    colour[root] = 1;
    while (!stack.empty()) {
      const int32_t j = stack.back();
      if (edge.back() < start[j + 1]) {
        const int32_t t = adj[edge.back()++];
        if (row4col[t] < 0) { reach[j] = 1; continue; }
        if (colour[t] == 1) return false;                    // (a) alternating cycle of tight entries
        if (colour[t] == 0) { colour[t] = 1; stack.push_back(t); edge.push_back(start[t]); }
        else reach[j] |= reach[t];
      } else {
        colour[j] = 2;
        stack.pop_back(); edge.pop_back();
        if (!stack.empty()) reach[stack.back()] |= reach[j];
      }
    }
  }
  if (nr < nc)
    for (int64_t j = 0; j < nc; ++j)
      if (row4col[j] >= 0 && v[j] == 0 && reach[j]) return false;   // (b) a row may move to an unassigned column for free
  return true;
}

template <typename CT>
__attribute__((target("avx512f,avx512bw,avx512vl,f16c")))
static int solve_q_t(int64_t nr, int64_t nc, int64_t ncp, const uint16_t* cost, int E, int32_t* col4row_out, int* unique_out);

// cost: fp16 bits, [nr][ncp] with ncp % 32 == 0 (padding columns are never live), nr <= nc.  0 ok, 2 infeasible, 3 overflow.
// The integer matrix is int16 (0.5 MB for 500x500: stays in the core's L2) when the entries' exponents span <= 4 binades --
// a randomly initialised CAFM's costs 1 - cos live in [0.5, 1.2] -- and int32 (unit 2^-24) otherwise: the self-match, whose diagonal is
// ~0, and a CAFM that has started to separate its features (near-duplicate proposals give 1 - cos ~ 1e-3 next to ~1).  On the
// MI355X host (EPYC 9575F) the two widths cost the same per relaxation pass (55 vs 57 ns solo on real 500x500 matrices); a third form
// that kept the fp16 bits and converted them inside the pass was 12 % slower than int32 and is not kept.
static std::atomic<int64_t> g_q16_solves{0}, g_q32_solves{0};      // which fixed-point width the solves took (diagnostics)
__attribute__((target("avx512f,avx512bw,avx512vl,f16c")))
static int solve_q(int64_t nr, int64_t nc, int64_t ncp, const uint16_t* cost, int32_t* col4row_out, int* unique_out = nullptr) {
  int emin, emax;
  exponent_span(cost, nr * ncp, emin, emax);
  if (emax < emin) { emin = emax = 1; }                      // all zeros
  if (emax - emin <= 4) { ++g_q16_solves; return solve_q_t<int16_t>(nr, nc, ncp, cost, emin, col4row_out, unique_out); }
  ++g_q32_solves;
  return solve_q_t<int32_t>(nr, nc, ncp, cost, 1, col4row_out, unique_out);   // unit 2^-24
}

template <typename CT>
__attribute__((target("avx512f,avx512bw,avx512vl,f16c")))
static int solve_q_t(int64_t nr, int64_t nc, int64_t ncp, const uint16_t* cost, int E, int32_t* col4row_out, int* unique_out) {
  int64_t steps = 0;
  // converted once; the buffer is kept per thread (fresh pages cost more than the solve's first rows)
  static thread_local CT* c_buf = nullptr;
  static thread_local size_t c_cap = 0;
  if (c_cap < (size_t)nr * ncp) {
    std::free(c_buf);
    c_cap = (size_t)nr * ncp;
    c_buf = (CT*)std::aligned_alloc(64, (c_cap * sizeof(CT) + 63) / 64 * 64);
  }
  struct { CT* p; CT* data() { return p; } } c32{c_buf};
  convert_q<CT>(cost, c32.data(), nr * ncp, E);
  AlignedBuf<int32_t> veff(ncp, QDEADV), spc(ncp);
  AlignedBuf<uint16_t> pos(ncp, (uint16_t)0x7fff);      // pos16 of the tie rule (see set_lane16)
  AlignedBuf<int16_t> path(ncp, -1);
  std::vector<int32_t> u(nr, 0), v(ncp, 0), spcsave(ncp, 0), col4row(nr, -1), row4col(ncp, -1);
  std::vector<int32_t> remaining(nc), remaining0(nc);
  std::vector<uint16_t> pos0(ncp, (uint16_t)0x7fff);
  std::vector<int32_t> sr_list, dead_list;
  sr_list.reserve(nr); dead_list.reserve(nc);
  for (int64_t it = 0; it < nc; ++it) {
    remaining0[it] = (int32_t)(nc - it - 1);       // reversed start: a constant matrix solves to the identity (scipy #11602)
    pos0[nc - it - 1] = (uint16_t)(it | 0x8000);
    veff[it] = 0;
  }
  for (int64_t curRow = 0; curRow < nr; ++curRow) {
    int32_t minVal = 0;
    int64_t i = curRow, num_remaining = nc, sink = -1;
    std::memcpy(remaining.data(), remaining0.data(), sizeof(int32_t) * nc);
    std::memcpy(pos.data(), pos0.data(), sizeof(uint16_t) * ncp);          // bit 15 cleared where the column already has a row
    std::fill(spc.begin(), spc.end(), QINF);
    sr_list.clear();
    dead_list.clear();
    while (sink == -1) {
      sr_list.push_back((int32_t)i);
      ++steps;
      // tie rule of the scalar scan: the first column (in `remaining` order) reaching the minimum is taken, and every later
      // tied column that has no row yet replaces it -> smallest position among the ties vs. largest among the free ties.
      int32_t lowest, first_pos32 = std::numeric_limits<int32_t>::max(), last_free32 = -1;
      const bool on_plateau = sr_list.size() > 1;      // the first pass of an augmentation starts from path length 0: no plateau yet
      if (on_plateau) {
        relax_row_plateau<CT>(c32.data() + i * ncp, veff.data(), spc.data(), path.data(), pos.data(), ncp, minVal - u[i],
                              (int16_t)i, minVal, first_pos32, last_free32);
        lowest = minVal;
      }
      if (!on_plateau || first_pos32 == std::numeric_limits<int32_t>::max()) {
        // no column at the previous minimum: the path length rises -- minimum and tie rule in sweeps of their own
        lowest = on_plateau ? min_of<CT>(spc.data(), ncp)
                            : relax_row_q<CT>(c32.data() + i * ncp, veff.data(), spc.data(), path.data(), ncp, minVal - u[i], (int16_t)i);
        if (lowest == QINF) return 2;
        if (lowest >= QLIM || lowest <= -QLIM) return 3;
        const __m512i vlow = _mm512_set1_epi32(lowest);
        const __m512i v7fff = _mm512_set1_epi16(0x7fff), vffff = _mm512_set1_epi16((short)0xffff);
        __m512i pmin = vffff, imin = vffff;
        for (int64_t j = 0; j < ncp; j += 32) {
          const __mmask16 m0 = _mm512_cmpeq_epi32_mask(_mm512_load_si512(spc.data() + j), vlow);
          const __mmask16 m1 = _mm512_cmpeq_epi32_mask(_mm512_load_si512(spc.data() + j + 16), vlow);
          const __mmask32 m = (__mmask32)m0 | ((__mmask32)m1 << 16);
          const __m512i pf = _mm512_load_si512(pos.data() + j);
          pmin = _mm512_mask_min_epu16(pmin, m, pmin, _mm512_and_si512(pf, v7fff));
          imin = _mm512_mask_min_epu16(imin, m, imin, _mm512_xor_si512(pf, vffff));
        }
        const uint32_t fp = hmin_epu16(pmin), fi = hmin_epu16(imin);
        first_pos32 = fp >= 0x7fffu ? std::numeric_limits<int32_t>::max() : (int32_t)fp;
        last_free32 = fi >= 0x8000u ? -1 : (int32_t)(0x7fffu - fi);
      }
      const int64_t first_pos = first_pos32, last_free_pos = last_free32;
      const int64_t index = (last_free_pos > first_pos) ? last_free_pos : first_pos;
      minVal = lowest;
      const int64_t j = remaining[index];
      if (row4col[j] == -1) sink = j; else i = row4col[j];
      if (sink == -1) {      // the next pass's cost row (L2-resident): every line on its way before the bookkeeping below
        const char* nrow = (const char*)(c32.data() + i * ncp);
        for (int64_t b = 0; b < (int64_t)(ncp * sizeof(CT)); b += 64) _mm_prefetch(nrow + b, _MM_HINT_T0);
      }
      spcsave[j] = spc[j];
      // single-lane updates of arrays the vector passes re-read right away: write the whole 64-byte block (a 4-byte store
      // followed by an overlapping 64-byte load cannot be store-forwarded and stalls until the store retires)
      set_lane(spc.data(), j, QINF);
      set_lane(veff.data(), j, QDEADV);
      dead_list.push_back((int32_t)j);
      --num_remaining;
      const int32_t moved = remaining[num_remaining];
      remaining[index] = moved;
      set_lane16(pos.data(), moved, (uint16_t)(index | (row4col[moved] == -1 ? 0x8000 : 0)));
    }
    bool over = false;
    u[curRow] += minVal;
    over |= (u[curRow] >= QLIM || u[curRow] <= -QLIM);
    for (int32_t r : sr_list)
      if (r != curRow) {
        u[r] += minVal - spcsave[col4row[r]];
        over |= (u[r] >= QLIM || u[r] <= -QLIM);
      }
    for (int32_t j : dead_list) {
      v[j] -= minVal - spcsave[j];
      veff[j] = v[j];
      over |= (v[j] >= QLIM || v[j] <= -QLIM);
    }
    if (over) return 3;
    int64_t j = sink;
    while (true) {
      const int64_t r = path[j];
      row4col[j] = (int32_t)r;
      pos0[j] &= (uint16_t)0x7fff;
      const int32_t t = col4row[r];
      col4row[r] = (int32_t)j;
      j = t;
      if (r == curRow) break;
    }
  }
  for (int64_t r = 0; r < nr; ++r) col4row_out[r] = col4row[r];
  g_steps.fetch_add(steps, std::memory_order_relaxed);
  if (unique_out) *unique_out = unique_optimum<CT>(nr, nc, ncp, c32.data(), u.data(), v.data(), col4row.data(), row4col.data()) ? 1 : 0;
  return 0;
}
static const bool g_have_q = __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512bw") &&
                             __builtin_cpu_supports("avx512vl") && __builtin_cpu_supports("f16c");

// gathers (and transposes when nc < nr) the fp16 matrix into the padded layout solve_q wants; false when an entry is
// outside the fixed-point range (|x| >= 8, inf).  NaN -> 0 like the reference.
static bool prepare_q(const uint16_t* cost, int64_t ld, const int64_t* row_map, int64_t nr, int64_t nc, bool transpose, int64_t ncp,
                      std::vector<uint16_t>& out) {
  const int64_t R = transpose ? nc : nr;
  out.assign((size_t)R * ncp, 0);
  bool ok = true;
  unsigned bad = 0;
  for (int64_t i = 0; i < nr; ++i) {
    const uint16_t* src = cost + (row_map ? row_map[i] : i) * ld;
    if (!transpose) {
      uint16_t* dst = &out[(size_t)i * ncp];
      for (int64_t j = 0; j < nc; ++j) {           // branch-free so it vectorises
        uint16_t h = src[j], e = h & 0x7c00;
        uint16_t isnan = (uint16_t)((e == 0x7c00) & ((h & 0x3ff) != 0));
        bad |= (unsigned)((e >= 0x4800) & !isnan);   // |x| >= 8 or inf
        dst[j] = isnan ? (uint16_t)0 : h;
      }
    } else {
      for (int64_t j = 0; j < nc; ++j) {
        uint16_t h = src[j], e = h & 0x7c00;
        bool isnan = e == 0x7c00 && (h & 0x3ff);
        bad |= (unsigned)(e >= 0x4800 && !isnan);
        out[(size_t)j * ncp + i] = isnan ? (uint16_t)0 : h;
      }
    }
  }
  ok = bad == 0;
  return ok;
}
#endif

inline double half_to_double(uint16_t h) {
  uint32_t sign = (h >> 15) & 1, exp = (h >> 10) & 0x1f, man = h & 0x3ff;
  double val;
  if (exp == 0) val = std::ldexp((double)man, -24);
  else if (exp == 31) val = man ? std::numeric_limits<double>::quiet_NaN() : std::numeric_limits<double>::infinity();
  else val = std::ldexp((double)(man | 0x400), (int)exp - 25);
  return sign ? -val : val;
}

// Persistent worker threads for the chain's independent solves (their thread_local conversion buffers stay warm; threads are
// never joined -- the pool is leaked at exit on purpose -- and are re-created in a forked child).
class Pool {
 public:
  static Pool& get() {
    static std::mutex gm;
    static Pool* inst = nullptr;
    std::lock_guard<std::mutex> lk(gm);
    if (!inst || inst->pid_ != getpid()) inst = new Pool();
    return *inst;
  }
  // runs jobs[0..n) on the workers (the caller takes part) and returns when all are done
  void run(std::vector<std::function<void()>>& jobs) {
    if (jobs.empty()) return;
    std::unique_lock<std::mutex> lk(m_);
    pending_ += (int)jobs.size();
    for (auto& j : jobs) q_.push_back(&j);
    cv_.notify_all();
    while (true) {
      if (!q_.empty()) {
        std::function<void()>* j = q_.front();
        q_.pop_front();
        lk.unlock();
        (*j)();
        lk.lock();
        if (--pending_ == 0) done_.notify_all();
      } else if (pending_ > 0) {
        done_.wait(lk);
      } else {
        break;
      }
    }
  }

 private:
  Pool() : pid_(getpid()) {
    unsigned hw = std::thread::hardware_concurrency();
    int n = hw > 4 ? 3 : (hw > 1 ? (int)hw - 1 : 0);     // + the calling thread: 4 solves side by side
    for (int i = 0; i < n; ++i) std::thread([this] { loop(); }).detach();
  }
  void loop() {
    std::unique_lock<std::mutex> lk(m_);
    while (true) {
      cv_.wait(lk, [this] { return !q_.empty(); });
      std::function<void()>* j = q_.front();
      q_.pop_front();
      lk.unlock();
      (*j)();
      lk.lock();
      if (--pending_ == 0) done_.notify_all();
    }
  }
  pid_t pid_;
  std::mutex m_;
  std::condition_variable cv_, done_;
  std::deque<std::function<void()>*> q_;
  int pending_ = 0;
};

#if defined(__x86_64__)
// Identity-order solve of one chain matrix on the fixed-point path.  c4r: assignment in the solver's orientation (rows =
// the shorter side).  Returns false when the fixed-point path does not apply (the caller then solves in order).
struct Spec { bool ok = false, transpose = false; int unique = 0; int status = 0; std::vector<int32_t> c4r; };
__attribute__((target("avx512f,avx512bw,avx512vl,f16c")))
static void speculate(const uint16_t* cost, int64_t nr, int64_t nc, Spec& sp) {
  sp.ok = false;
  if (!g_have_q || nr >= 32000 || nc >= 32000 || nr == 0 || nc == 0) return;
  sp.transpose = nc < nr;
  const int64_t R = sp.transpose ? nc : nr, Cc = sp.transpose ? nr : nc, ncp = (Cc + 31) / 32 * 32;
  std::vector<uint16_t> q;
  if (!prepare_q(cost, nc, nullptr, nr, nc, sp.transpose, ncp, q)) return;
  sp.c4r.assign(R, -1);
  sp.status = solve_q(R, Cc, ncp, q.data(), sp.c4r.data(), &sp.unique);
  sp.ok = sp.status == 0;
}
#endif

}  // namespace

extern "C" {
// cost: row-major [nr][nc] float64.  row_ind / col_ind: min(nr,nc) entries, rows ascending (as scipy returns them).
// returns 0 ok, 1 invalid entries (NaN / -inf), 2 infeasible.
int tscd_lap_solve_f64(const double* cost, int64_t nr, int64_t nc, int64_t* row_ind, int64_t* col_ind) {
  return solve(nr, nc, cost, row_ind, col_ind);
}
// fp16 cost matrix as produced by the device GEMM (EPI_COST), optionally with a row gather: row r of the solved matrix
// is cost[row_map[r]] (row_map may be NULL).  NaN entries are replaced by 0 like the reference does.
int tscd_lap_solve_f16(const uint16_t* cost, int64_t ld, const int64_t* row_map, int64_t nr, int64_t nc, int64_t* row_ind,
                       int64_t* col_ind) {
  if (nr == 0 || nc == 0) return 0;
#if defined(__x86_64__)
  if (g_have_q && nr < 32000 && nc < 32000) {
    const bool transpose = nc < nr;
    const int64_t R = transpose ? nc : nr, Cc = transpose ? nr : nc, ncp = (Cc + 31) / 32 * 32;
    std::vector<uint16_t> q;
    if (prepare_q(cost, ld, row_map, nr, nc, transpose, ncp, q)) {
      std::vector<int32_t> c4r(R);
      int st = solve_q(R, Cc, ncp, q.data(), c4r.data());
      if (st == 0) {
        if (transpose) {
          std::vector<int64_t> order(R);
          std::iota(order.begin(), order.end(), 0);
          std::sort(order.begin(), order.end(), [&](int64_t x, int64_t y) { return c4r[x] < c4r[y]; });
          for (int64_t i = 0; i < R; ++i) { row_ind[i] = c4r[order[i]]; col_ind[i] = order[i]; }
        } else {
          for (int64_t i = 0; i < R; ++i) { row_ind[i] = i; col_ind[i] = c4r[i]; }
        }
        return 0;
      }
      if (st == 2) return 2;
      // st == 3: magnitudes left the fixed-point range -> double solver below
    }
  }
#endif
  std::vector<double> c((size_t)nr * nc);
  for (int64_t i = 0; i < nr; ++i) {
    const uint16_t* src = cost + (row_map ? row_map[i] : i) * ld;
    for (int64_t j = 0; j < nc; ++j) {
      double d = half_to_double(src[j]);
      c[(size_t)i * nc + j] = (d != d) ? 0.0 : d;
    }
  }
  return solve(nr, nc, c.data(), row_ind, col_ind);
}
// The whole frame-to-frame matching chain of one clip in a single call (AwarePositionRegMatcher.forward,
// tscd_matching.py:940-1080): frame k's reference rows are frame k-1's rows re-ordered by frame k-1's assignment, so solve k
// needs the column order produced by solve k-1 and nothing from the device.  Running the chain here lets the caller keep its
// own thread free (ctypes drops the GIL for the duration) to queue device work that does not depend on the assignments.
//   kind[k]: 0 = first frame of a clip (cost(cur, cur), identity row order), 1 = matched against carried-over state (rows as
//            given), 2 = matched against the previous frame (rows gathered by the previous frame's column order).
//   cost[k]: fp16 bits [nr[k]][nc[k]] (ld = nc[k]);  for kind 2, nr[k] == nc[k-1].
// Outputs, concatenated over k: row_out / col_out (min(nr,nc) entries each: what scipy returns), perm_out (nc[k] entries:
// col_out followed by the unmatched columns in ascending order when nr[k] < nc[k]; the identity for kind 0) -- the order in
// which the reference keeps frame k's rows.  Returns 0, or 16*k + status of the first failing solve.
static std::atomic<int> g_chain_parallel{1};
static std::atomic<int64_t> g_chain_spec_hits{0};
// 1 (default): chain matrices are solved side by side and relabelled where the optimum is certified unique, adaptively (see
// below); 2: always; 0: strictly in order.  Returns the previous setting.  tscd_lap_chain_spec_hits: matrices answered from the side-by-side solve so far.
int tscd_lap_chain_set_parallel(int on) { return g_chain_parallel.exchange(on); }
int64_t tscd_lap_chain_spec_hits(int reset) { int64_t h = g_chain_spec_hits; if (reset) g_chain_spec_hits = 0; return h; }
}  // extern "C"
// The chain with a hook: done(k) runs as soon as matrix k's row_out / col_out / perm_out entries are final (the deferred join hands
// each frame's part of the index table to the device while the next matrix is being solved).
static int lap_chain_core(int K, const uint16_t* const* cost, const int64_t* nr, const int64_t* nc, const int32_t* kind,
                          int64_t* row_out, int64_t* col_out, int64_t* perm_out, const std::function<void(int)>& on_done);
extern "C" {
int tscd_lap_chain_f16(int K, const uint16_t* const* cost, const int64_t* nr, const int64_t* nc, const int32_t* kind,
                       int64_t* row_out, int64_t* col_out, int64_t* perm_out) {
  return lap_chain_core(K, cost, nr, nc, kind, row_out, col_out, perm_out, [](int) {});
}
}  // extern "C"
static int lap_chain_core(int K, const uint16_t* const* cost, const int64_t* nr, const int64_t* nc, const int32_t* kind,
                          int64_t* row_out, int64_t* col_out, int64_t* perm_out, const std::function<void(int)>& on_done) {
  std::vector<int64_t> prev_perm;
  int64_t o_pair = 0, o_perm = 0;
#if defined(__x86_64__)
  // Solve k needs solve k-1's column order only to ORDER its rows.  All matrices are therefore solved side by side in
  // identity order first; where the optimum is certified unique (unique_optimum: it almost always is for real-valued
  // features) the in-order answer is that assignment relabelled, otherwise the matrix is solved again in order below.
  // Adaptive: the side-by-side solves pay off only when most optima are certified unique (trained features).  Tie-heavy costs
  // (e.g. a randomly initialised network: near-duplicate proposals) fail the certificate and would be solved twice, so after a
  // chain with fewer hits than misses speculation pauses and is only probed again every 64th chain.  g_chain_parallel == 2
  // forces it (tests).
  static std::atomic<int> cooldown{0};
  bool speculate_now = g_chain_parallel == 2 || (g_chain_parallel == 1 && cooldown.load() <= 0);
  if (g_chain_parallel == 1 && !speculate_now) --cooldown;
  std::vector<Spec> spec(K);
  int tried = 0;
  if (speculate_now && K > 1) {
    std::vector<std::function<void()>> jobs;
    for (int k = 0; k < K; ++k)
      if (kind[k] == 2 && nr[k] > 0 && nc[k] > 0) jobs.emplace_back([&, k] { speculate(cost[k], nr[k], nc[k], spec[k]); });
    tried = (int)jobs.size();
    Pool::get().run(jobs);
    int hits = 0;
    for (int k = 0; k < K; ++k) hits += spec[k].ok && spec[k].unique;
    if (g_chain_parallel == 1) cooldown = (2 * hits < tried) ? 64 : 0;
  }
#endif
  for (int k = 0; k < K; ++k) {
    const int64_t R = nr[k], Cn = nc[k], np = R < Cn ? R : Cn;
    const int64_t* row_map = nullptr;
    if (kind[k] == 2) {
      if ((int64_t)prev_perm.size() != R) return 16 * k + 4;
      row_map = prev_perm.data();
    }
    int st = 0;
    bool done = false;
#if defined(__x86_64__)
    if (kind[k] == 2 && spec[k].ok && spec[k].unique) {
      // relabel: presented row i is cost row row_map[i]
      int64_t* ro = row_out + o_pair; int64_t* co = col_out + o_pair;
      if (!spec[k].transpose) {            // every presented row is assigned
        for (int64_t i = 0; i < R; ++i) { ro[i] = i; co[i] = spec[k].c4r[row_map[i]]; }
      } else {                             // c4r[column] = cost row; only Cn of the R presented rows are assigned
        std::vector<int64_t> inv(R);
        for (int64_t i = 0; i < R; ++i) inv[row_map[i]] = i;
        std::vector<std::pair<int64_t, int64_t>> pr(Cn);
        for (int64_t c = 0; c < Cn; ++c) pr[c] = {inv[spec[k].c4r[c]], c};
        std::sort(pr.begin(), pr.end());
        for (int64_t c = 0; c < Cn; ++c) { ro[c] = pr[c].first; co[c] = pr[c].second; }
      }
      ++g_chain_spec_hits;
      done = true;
    }
#endif
    if (!done) st = tscd_lap_solve_f16(cost[k], Cn, row_map, R, Cn, row_out + o_pair, col_out + o_pair);
    if (st != 0) return 16 * k + st;
    std::vector<int64_t> perm(Cn);
    if (kind[k] == 0) {
      std::iota(perm.begin(), perm.end(), 0);
    } else {
      std::vector<uint8_t> taken(Cn, 0);
      for (int64_t i = 0; i < np; ++i) { perm[i] = col_out[o_pair + i]; taken[perm[i]] = 1; }
      int64_t w = np;
      for (int64_t j = 0; j < Cn; ++j)
        if (!taken[j]) perm[w++] = j;
    }
    std::memcpy(perm_out + o_perm, perm.data(), sizeof(int64_t) * Cn);
    prev_perm.swap(perm);
    // when the reference frame had more rows than the current one (R > Cn) only the matched reference rows survive: the next
    // frame's reference is the current frame's Cn rows in column order -- prev_perm holds exactly col_out (np == Cn)
    o_pair += np;
    o_perm += Cn;
    on_done(k);
  }
  return 0;
}
extern "C" {
// ---- a latch the device's stream can be parked on ---------------------------------------------------------------------------
// The train step queues everything that FOLLOWS the assignments (CAFM, the refined heads, the whole backward) while the chain is
// still being solved on a worker thread: the stream is held by a host function (hipLaunchHostFunc -> tscd_latch_wait_cb, run on a
// runtime thread without the interpreter) until the worker has written the clip's index table to pinned memory and raised the
// latch.  A latch carries a generation counter so one object serves every step; a wait gives up after its timeout (the stream then
// continues on the table's previous contents, which are always in-range permutations) and the failure is reported by
// tscd_latch_timeouts / the worker's own status.
struct TscdLatch {
  std::mutex m;
  std::condition_variable cv;
  uint64_t gen = 0;
  std::atomic<uint64_t> agen{0};          // the same counter for the lock-free spin phase of a wait
  std::atomic<int64_t> spin_us{0};        // tscd_latch_set_spin: how long a wait polls before it blocks on the condition variable
  std::atomic<int64_t> timeouts{0};
};
struct TscdLatchTicket { TscdLatch* latch; uint64_t gen; int timeout_ms; };
void* tscd_latch_create(void) { return new TscdLatch(); }
void tscd_latch_destroy(void* l) { delete (TscdLatch*)l; }
void tscd_latch_signal(void* l, uint64_t gen) {
  TscdLatch* L = (TscdLatch*)l;
  {
    std::lock_guard<std::mutex> g(L->m);
    if (gen > L->gen) L->gen = gen;
    L->agen.store(L->gen, std::memory_order_release);
  }
  L->cv.notify_all();
}
// A wait first polls the counter for up to `us` microseconds (no futex wake-up between the worker's signal and the stream's go-ahead:
// the waiting thread is a runtime thread that has nothing else to do), then blocks.  0 (default): block at once.  Returns the previous value.
int64_t tscd_latch_set_spin(void* l, int64_t us) { return ((TscdLatch*)l)->spin_us.exchange(us < 0 ? 0 : us); }
uint64_t tscd_latch_generation(void* l) { TscdLatch* L = (TscdLatch*)l; std::lock_guard<std::mutex> g(L->m); return L->gen; }
int64_t tscd_latch_timeouts(void* l) { return ((TscdLatch*)l)->timeouts.load(); }
// 0 when generation `gen` has been reached, 1 after timeout_ms without it
int tscd_latch_wait(void* l, uint64_t gen, int timeout_ms) {
  TscdLatch* L = (TscdLatch*)l;
  const int64_t spin = L->spin_us.load();
  if (spin > 0) {
    const auto t0 = std::chrono::steady_clock::now();
    for (int it = 0;; ++it) {
      if (L->agen.load(std::memory_order_acquire) >= gen) return 0;
#if defined(__x86_64__)
      _mm_pause();
#endif
      if ((it & 255) == 255 && std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count() >= spin) break;
    }
  }
  std::unique_lock<std::mutex> g(L->m);
  bool ok = L->cv.wait_for(g, std::chrono::milliseconds(timeout_ms), [&] { return L->gen >= gen; });
  if (!ok) ++L->timeouts;
  return ok ? 0 : 1;
}
// heap ticket for one queued wait; tscd_latch_wait_cb (signature of hipHostFn_t) consumes it
void* tscd_latch_ticket(void* l, uint64_t gen, int timeout_ms) { return new TscdLatchTicket{(TscdLatch*)l, gen, timeout_ms}; }
void tscd_latch_wait_cb(void* ticket) {
  TscdLatchTicket* t = (TscdLatchTicket*)ticket;
  tscd_latch_wait(t->latch, t->gen, t->timeout_ms);
  delete t;
}

// tscd_lap_chain_f16 for the clamped training regime (every frame n proposals, kinds 0,2,2,...): additionally writes the clip's
// index table -- [inv_0 | c_1 | inv_1 | c_2 | inv_2 ...], n int64 each: c_k the order in which the reference keeps frame k's rows
// (perm_out), inv_k its inverse, inv_0 the inverse of the first frame's self assignment -- to `table` (pinned host memory the
// device copies from), frame by frame: as soon as matrix k is solved its part is written and the latch raised to gen - (K-1-k), so the
// caller allots K consecutive generations per clip and may park the stream once per frame.  The latch reaches `gen` on EVERY path; on
// failure the parts not yet announced hold identity permutations, so work queued behind the latch stays in range.  table_len must be (2K-1)*n.
int tscd_lap_chain_f16_table(int K, const uint16_t* const* cost, const int64_t* nr, const int64_t* nc, const int32_t* kind,
                             int64_t* row_out, int64_t* col_out, int64_t* perm_out, int64_t* table, int64_t table_len, void* latch,
                             uint64_t gen) {
  int st = 0;
  const int64_t n = K > 0 ? nc[0] : 0;
  bool shape_ok = table && K > 0 && table_len == (2 * (int64_t)K - 1) * n;
  for (int k = 0; k < K && shape_ok; ++k) shape_ok = nr[k] == n && nc[k] == n && kind[k] == (k ? 2 : 0);
  if (!shape_ok) st = 5;
  int finished = 0;                                                    // matrices whose table part is written and announced
  auto done = [&](int k) {
    // frame k's part: inv_0 (k = 0) or [c_k | inv_k]; then generation gen - (K-1-k): a stream parked on frame k goes on while
    // frame k+1 is still being solved
    if (k == 0) {
      for (int64_t i = 0; i < n; ++i) table[col_out[i]] = i;          // inverse of the first frame's own column order
    } else {
      int64_t* t = table + (2 * (int64_t)k - 1) * n;
      const int64_t* pk = perm_out + (int64_t)k * n;
      std::memcpy(t, pk, sizeof(int64_t) * n);
      for (int64_t i = 0; i < n; ++i) t[n + pk[i]] = i;
    }
    finished = k + 1;
    const uint64_t back = (uint64_t)(K - 1 - k);
    if (latch && gen > back) tscd_latch_signal(latch, gen - back);
  };
  if (st == 0) st = lap_chain_core(K, cost, nr, nc, kind, row_out, col_out, perm_out, done);
  if (table && st != 0 && n > 0) {                                     // what was not announced: identity rows
    const int64_t from = finished == 0 ? 0 : (2 * (int64_t)finished - 1) * n;
    for (int64_t i = from; i < table_len; ++i) table[i] = i % n;
  }
  if (latch) tscd_latch_signal(latch, gen);
  return st;
}
// solves answered on the int16 (which = 0) / int32 (which = 1) fixed-point matrix so far
int64_t tscd_lap_debug_width(int which, int reset) {
#if defined(__x86_64__)
  std::atomic<int64_t>& c = which ? g_q32_solves : g_q16_solves;
  int64_t v = c;
  if (reset) c = 0;
  return v;
#else
  return 0;
#endif
}
int64_t tscd_lap_debug_steps(int reset) {
#if defined(__x86_64__)
  int64_t s = g_steps;
  if (reset) g_steps = 0;
  return s;
#else
  return 0;
#endif
}
// 1 when tscd_lap_solve_f16 runs the fixed-point AVX-512 variant on this host
int tscd_lap_fixed_point_available(void) {
#if defined(__x86_64__)
  return g_have_q ? 1 : 0;
#else
  return 0;
#endif
}
}
