// COCO bounding-box evaluation on the host: the per-image matching and the precision / recall
// accumulation of the COCO detection protocol, the part the reference runs through its pybind11 extension
// yolox/layers/cocoeval/cocoeval.cpp:59-199 (COCOevalEvaluateImages) and :370-502 (COCOevalAccumulate), falling back to
// pycocotools.cocoeval (ovis_evaluator_v2.py:397-402).  Neither pycocotools nor the reference's extension can run in this
// image, so this follows the published COCO protocol:
//   * per (image, category): detections sorted by score (stable, descending), truncated to the largest maxDets; ground truth
//     sorted so that ignored boxes (crowd, or area outside the range) come last;
//   * IoU of xywh boxes; against a crowd box IoU = intersection / detection area;
//   * per IoU threshold t in {.50, .55, ... .95}: each detection takes the best still-free ground truth with IoU >= t,
//     preferring non-ignored ones (a crowd box may match several detections); a detection matched to an ignored box, or
//     unmatched with its own area outside the range, is ignored;
//   * per (category, area range, maxDets): scores of all images merged by a stable descending sort, cumulative TP / FP,
//     precision made monotone from the right, sampled at the 101 recall thresholds 0, .01, ... 1;
//   * summary = the 12 numbers of COCOeval.summarize() (AP, AP50, AP75, AP small/medium/large, AR@1/10/100, AR s/m/l).
// Plain C ABI (include/tscd_host.h), no dependencies, one thread per category.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <numeric>
#include <thread>
#include <vector>

namespace {

constexpr int T = 10, R = 101, A = 4, MD = 3;
const double kAreaLo[A] = {0.0, 0.0, 32.0 * 32.0, 96.0 * 96.0};
const double kAreaHi[A] = {1e10, 32.0 * 32.0, 96.0 * 96.0, 1e10};
const int kMaxDets[MD] = {1, 10, 100};

struct Box { float x, y, w, h; };

inline double iou_xywh(const Box& d, const Box& g, bool crowd) {
  const double ix = std::min<double>(d.x + d.w, g.x + g.w) - std::max<double>(d.x, g.x);
  const double iy = std::min<double>(d.y + d.h, g.y + g.h) - std::max<double>(d.y, g.y);
  if (ix <= 0 || iy <= 0) return 0.0;
  const double inter = ix * iy, da = (double)d.w * d.h, ga = (double)g.w * g.h;
  return inter / (crowd ? da : da + ga - inter);
}

struct ImgEval {            // one (image, category, area range): matches at every IoU threshold for <= maxDet detections
  std::vector<float> score;
  std::vector<uint8_t> dt_matched;    // [T][nd]
  std::vector<uint8_t> dt_ignore;     // [T][nd]
  int nd = 0, n_gt_valid = 0;
};

struct CatData {
  std::vector<std::vector<int>> gt_of_img, dt_of_img;   // indices into the flat arrays
};

}  // namespace

extern "C" int tscd_cocoeval_bbox(const float* gt_box, const int* gt_img, const int* gt_cat, const unsigned char* gt_crowd,
                                  const float* gt_area, long n_gt, const float* dt_box, const int* dt_img, const int* dt_cat,
                                  const float* dt_score, long n_dt, int n_img, int n_cat, float* stats12, float* precision_out,
                                  float* recall_out) {
  if (n_img <= 0 || n_cat <= 0 || !stats12) return 1;
  for (long i = 0; i < n_gt; ++i)
    if (gt_img[i] < 0 || gt_img[i] >= n_img || gt_cat[i] < 0 || gt_cat[i] >= n_cat) return 2;
  for (long i = 0; i < n_dt; ++i)
    if (dt_img[i] < 0 || dt_img[i] >= n_img || dt_cat[i] < 0 || dt_cat[i] >= n_cat) return 2;
  const Box* gb = reinterpret_cast<const Box*>(gt_box);
  const Box* db = reinterpret_cast<const Box*>(dt_box);
  std::vector<CatData> cats(n_cat);
  for (auto& c : cats) { c.gt_of_img.resize(n_img); c.dt_of_img.resize(n_img); }
  for (long i = 0; i < n_gt; ++i) cats[gt_cat[i]].gt_of_img[gt_img[i]].push_back((int)i);
  for (long i = 0; i < n_dt; ++i) cats[dt_cat[i]].dt_of_img[dt_img[i]].push_back((int)i);

  // precision[T][R][K][A][M], recall[T][K][A][M]; -1 = undefined (no ground truth of that category / range)
  std::vector<float> precision((size_t)T * R * n_cat * A * MD, -1.f), recall((size_t)T * n_cat * A * MD, -1.f);
  auto P = [&](int t, int r, int k, int a, int m) -> float& { return precision[((((size_t)t * R + r) * n_cat + k) * A + a) * MD + m]; };
  auto RC = [&](int t, int k, int a, int m) -> float& { return recall[(((size_t)t * n_cat + k) * A + a) * MD + m]; };
  const int max_det = kMaxDets[MD - 1];

  auto work = [&](int k) {
    const CatData& c = cats[k];
    for (int a = 0; a < A; ++a) {
      std::vector<ImgEval> evals;
      evals.reserve(n_img);
      for (int im = 0; im < n_img; ++im) {
        const auto& gi = c.gt_of_img[im];
        const auto& di = c.dt_of_img[im];
        if (gi.empty() && di.empty()) continue;
        // ground truth: ignore flag, then non-ignored first (stable)
        std::vector<int> g(gi);
        std::vector<uint8_t> gig(g.size());
        auto ig_of = [&](int idx) { return (uint8_t)(gt_crowd[idx] || gt_area[idx] < kAreaLo[a] || gt_area[idx] > kAreaHi[a]); };
        std::stable_sort(g.begin(), g.end(), [&](int x, int y) { return ig_of(x) < ig_of(y); });
        for (size_t j = 0; j < g.size(); ++j) gig[j] = ig_of(g[j]);
        // detections: score descending (stable), top max_det
        std::vector<int> d(di);
        std::stable_sort(d.begin(), d.end(), [&](int x, int y) { return dt_score[x] > dt_score[y]; });
        if ((int)d.size() > max_det) d.resize(max_det);
        ImgEval e;
        e.nd = (int)d.size();
        e.score.resize(e.nd);
        e.dt_matched.assign((size_t)T * e.nd, 0);
        e.dt_ignore.assign((size_t)T * e.nd, 0);
        for (int j = 0; j < e.nd; ++j) e.score[j] = dt_score[d[j]];
        for (size_t j = 0; j < g.size(); ++j) e.n_gt_valid += gig[j] ? 0 : 1;
        std::vector<double> ious((size_t)e.nd * g.size());
        for (int i = 0; i < e.nd; ++i)
          for (size_t j = 0; j < g.size(); ++j) ious[(size_t)i * g.size() + j] = iou_xywh(db[d[i]], gb[g[j]], gt_crowd[g[j]] != 0);
        std::vector<int> gtm(g.size());
        for (int t = 0; t < T; ++t) {
          const double thr = 0.5 + 0.05 * t;
          std::fill(gtm.begin(), gtm.end(), -1);
          for (int i = 0; i < e.nd; ++i) {
            double best = std::min(thr, 1.0 - 1e-10);
            int m = -1;
            for (size_t j = 0; j < g.size(); ++j) {
              if (gtm[j] >= 0 && !gt_crowd[g[j]]) continue;             // taken (crowd boxes may be matched repeatedly)
              if (m > -1 && !gig[m] && gig[j]) break;                    // a real match exists: do not trade it for an ignored box
              const double v = ious[(size_t)i * g.size() + j];
              if (v < best) continue;
              best = v;
              m = (int)j;
            }
            if (m >= 0) {
              gtm[m] = i;
              e.dt_matched[(size_t)t * e.nd + i] = 1;
              e.dt_ignore[(size_t)t * e.nd + i] = gig[m];
            } else {
              const double da = (double)db[d[i]].w * db[d[i]].h;
              e.dt_ignore[(size_t)t * e.nd + i] = (da < kAreaLo[a] || da > kAreaHi[a]) ? 1 : 0;
            }
          }
        }
        evals.push_back(std::move(e));
      }
      // ---- accumulate ----
      for (int m = 0; m < MD; ++m) {
        const int md = kMaxDets[m];
        long npig = 0;
        std::vector<float> sc;
        std::vector<std::pair<int, int>> src;     // (eval index, detection index)
        for (size_t ei = 0; ei < evals.size(); ++ei) {
          npig += evals[ei].n_gt_valid;
          const int n = std::min(evals[ei].nd, md);
          for (int i = 0; i < n; ++i) { sc.push_back(evals[ei].score[i]); src.emplace_back((int)ei, i); }
        }
        if (npig == 0) continue;
        std::vector<int> order(sc.size());
        std::iota(order.begin(), order.end(), 0);
        std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return sc[x] > sc[y]; });
        for (int t = 0; t < T; ++t) {
          std::vector<double> rc, pr;
          double tp = 0, fp = 0;
          for (int oi : order) {
            const ImgEval& e = evals[src[oi].first];
            const size_t at = (size_t)t * e.nd + src[oi].second;
            if (e.dt_ignore[at]) continue;
            if (e.dt_matched[at]) tp += 1; else fp += 1;
            rc.push_back(tp / npig);
            pr.push_back(tp / (tp + fp + 2.220446049250313e-16));
          }
          RC(t, k, a, m) = rc.empty() ? 0.f : (float)rc.back();
          for (int i = (int)pr.size() - 1; i > 0; --i)
            if (pr[i] > pr[i - 1]) pr[i - 1] = pr[i];
          size_t pos = 0;
          for (int r = 0; r < R; ++r) {
            const double thr = r * 0.01;
            while (pos < rc.size() && rc[pos] < thr) ++pos;             // np.searchsorted(rc, thr, side='left')
            P(t, r, k, a, m) = pos < pr.size() ? (float)pr[pos] : 0.f;
          }
        }
      }
    }
  };
  {
    unsigned hw = std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
    std::vector<std::thread> pool;
    std::vector<int> next(1, 0);
    std::vector<std::vector<int>> share(hw);
    for (int k = 0; k < n_cat; ++k) share[k % hw].push_back(k);
    for (unsigned w = 0; w < hw; ++w)
      pool.emplace_back([&, w] { for (int k : share[w]) work(k); });
    for (auto& th : pool) th.join();
  }
  auto mean_p = [&](int t_lo, int t_hi, int a, int m) {
    double s = 0; long n = 0;
    for (int t = t_lo; t < t_hi; ++t)
      for (int r = 0; r < R; ++r)
        for (int k = 0; k < n_cat; ++k) { float v = P(t, r, k, a, m); if (v > -1.f) { s += v; ++n; } }
    return n ? (float)(s / n) : -1.f;
  };
  auto mean_r = [&](int a, int m) {
    double s = 0; long n = 0;
    for (int t = 0; t < T; ++t)
      for (int k = 0; k < n_cat; ++k) { float v = RC(t, k, a, m); if (v > -1.f) { s += v; ++n; } }
    return n ? (float)(s / n) : -1.f;
  };
  stats12[0] = mean_p(0, T, 0, 2); stats12[1] = mean_p(0, 1, 0, 2); stats12[2] = mean_p(5, 6, 0, 2);
  stats12[3] = mean_p(0, T, 1, 2); stats12[4] = mean_p(0, T, 2, 2); stats12[5] = mean_p(0, T, 3, 2);
  stats12[6] = mean_r(0, 0); stats12[7] = mean_r(0, 1); stats12[8] = mean_r(0, 2);
  stats12[9] = mean_r(1, 2); stats12[10] = mean_r(2, 2); stats12[11] = mean_r(3, 2);
  if (precision_out) std::memcpy(precision_out, precision.data(), precision.size() * sizeof(float));
  if (recall_out) std::memcpy(recall_out, recall.data(), recall.size() * sizeof(float));
  return 0;
}
