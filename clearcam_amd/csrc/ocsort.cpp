// OC-SORT multi-object tracker on the host, behind `OCSort.update(preds, thresh)` (ocsort_tracker/ocsort.py:186-308).
//
// This is the consumer directly behind the detector (clearcam.py:583-585: preds = jit_infer(yolo, frame).numpy();
// tracker.update(preds, thresh)).  The reference is 624 lines of per-track numpy (a 7x7 Kalman filter per object with
// np.linalg.inv, Python loops over tracks, a greedy assignment on an argsort) and costs ~1-3 ms per frame per camera,
// i.e. more host time than the GPU spends on the frame; at 64 cameras per GPU it is the pipeline bottleneck
// (SURVEY.md §8f-2).  Same algorithm here in plain C++ with fixed-size 7-state arithmetic, no allocation per track
// per frame.  Pure host code: no HIP calls.
//
// Behaviour follows the reference statement by statement, including the parts that look accidental, because the
// golden fixtures (tests/golden/ocsort_*.npz, produced by the reference itself) pin them:
//  * detections arrive as float32 and several intermediate values are float32 in numpy 2 (requirements.txt:2) -
//    box -> (x,y,s,r), the speed direction, box areas and centres of detections; they are float here too;
//  * a new track reports its Kalman state until its first matched update (last_observation placeholder);
//  * the observation-centric re-update ("unfreeze", kalmanfilter.py:66-104) restores the filter to the snapshot taken
//    at the first missed frame, replays a linear virtual trajectory, and drops the triggering observation from the
//    observation history;
//  * `score` of a track is the score of the detection that created it; class id is the arg-max of summed scores;
//  * dead tracks are only removed when they moved (speed > 2) or after 600 missed frames (ocsort.py:296-297);
//  * a NaN IoU is "not below the threshold" and therefore matches, as in the reference's comparisons.
// One deliberate difference: track ids count per tracker object; the reference uses one process-wide counter that
// every new OCSort() resets (ocsort.py:64,198), which interleaves ids between cameras.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <memory>
#include <stdexcept>
#include <string>
#include <thread>
#include <utility>
#include <vector>

namespace cc { void set_error(const std::string& msg); }
#ifndef OCP
#define OCP(i)
#endif

namespace {

constexpr int NX = 7, NZ = 4;

struct HistObs { bool present = false, f32 = false; double v[4] = {0, 0, 0, 0}; };

// filterpy-style Kalman filter specialised to F = [I4 | I3;0], H = [I4 | 0]  (ocsort.py:72-82, kalmanfilter.py)
struct Kalman {
  double x[NX];
  double P[NX][NX];
  std::vector<HistObs> history;
  bool observed = false;
  struct Snapshot { double x[NX]; double P[NX][NX]; std::vector<HistObs> history; bool had_saved; };
  std::unique_ptr<Snapshot> saved;

  static const double* Qdiag() { static const double q[NX] = {1, 1, 1, 1, 0.01, 0.01, 0.0001}; return q; }
  static const double* Rdiag() { static const double r[NZ] = {1, 1, 10, 10}; return r; }

  Kalman() {
    for (int i = 0; i < NX; ++i) { x[i] = 0; for (int j = 0; j < NX; ++j) P[i][j] = 0; }
    for (int i = 0; i < NX; ++i) P[i][i] = (i >= 4 ? 1000.0 : 1.0) * 10.0;
  }

  void predict() {
    // x = F x ; P = F P F^T + Q   with F = I + E, E[i][i+4] = 1 for i < 3
    for (int i = 0; i < 3; ++i) x[i] = x[i] + x[i + 4];
    double FP[NX][NX];
    for (int i = 0; i < NX; ++i)
      for (int j = 0; j < NX; ++j) FP[i][j] = P[i][j] + (i < 3 ? P[i + 4][j] : 0.0);
    for (int i = 0; i < NX; ++i)
      for (int j = 0; j < NX; ++j) P[i][j] = FP[i][j] + (j < 3 ? FP[i][j + 4] : 0.0);
    for (int i = 0; i < NX; ++i) P[i][i] += Qdiag()[i];
  }

  static void inv4(const double (&S)[NZ][NZ], double (&SI)[NZ][NZ]) {   // Gauss-Jordan with partial pivoting
    double a[NZ][2 * NZ];
    for (int i = 0; i < NZ; ++i)
      for (int j = 0; j < NZ; ++j) { a[i][j] = S[i][j]; a[i][j + NZ] = i == j ? 1.0 : 0.0; }
    for (int c = 0; c < NZ; ++c) {
      int p = c;
      for (int r = c + 1; r < NZ; ++r) if (std::fabs(a[r][c]) > std::fabs(a[p][c])) p = r;
      if (p != c) for (int j = 0; j < 2 * NZ; ++j) std::swap(a[p][j], a[c][j]);
      const double d = a[c][c];
      for (int j = 0; j < 2 * NZ; ++j) a[c][j] /= d;
      for (int r = 0; r < NZ; ++r) {
        if (r == c) continue;
        const double f = a[r][c];
        if (f != 0.0) for (int j = 0; j < 2 * NZ; ++j) a[r][j] -= f * a[c][j];
      }
    }
    for (int i = 0; i < NZ; ++i) for (int j = 0; j < NZ; ++j) SI[i][j] = a[i][j + NZ];
  }

  void correct(const double (&z)[NZ]) {          // the measurement update proper (kalmanfilter.py:119-129)
    double y[NZ], S[NZ][NZ], SI[NZ][NZ], K[NX][NZ];
    for (int i = 0; i < NZ; ++i) y[i] = z[i] - x[i];
    for (int i = 0; i < NZ; ++i) for (int j = 0; j < NZ; ++j) S[i][j] = P[i][j] + (i == j ? Rdiag()[i] : 0.0);
    inv4(S, SI);
    for (int i = 0; i < NX; ++i)
      for (int j = 0; j < NZ; ++j) { double s = 0; for (int k = 0; k < NZ; ++k) s += P[i][k] * SI[k][j]; K[i][j] = s; }
    for (int i = 0; i < NX; ++i) { double s = 0; for (int k = 0; k < NZ; ++k) s += K[i][k] * y[k]; x[i] += s; }
    // Joseph form: P = (I-KH) P (I-KH)^T + K R K^T
    double A[NX][NX], AP[NX][NX];
    for (int i = 0; i < NX; ++i) for (int j = 0; j < NX; ++j) A[i][j] = (i == j ? 1.0 : 0.0) - (j < NZ ? K[i][j] : 0.0);
    for (int i = 0; i < NX; ++i)
      for (int j = 0; j < NX; ++j) { double s = 0; for (int k = 0; k < NX; ++k) s += A[i][k] * P[k][j]; AP[i][j] = s; }
    for (int i = 0; i < NX; ++i)
      for (int j = 0; j < NX; ++j) {
        double s = 0; for (int k = 0; k < NX; ++k) s += AP[i][k] * A[j][k];
        double kr = 0; for (int k = 0; k < NZ; ++k) kr += K[i][k] * Rdiag()[k] * K[j][k];
        P[i][j] = s + kr;
      }
  }

  void unfreeze() {                              // kalmanfilter.py:66-104
    int i1 = -1, i2 = -1;
    for (int i = (int)history.size() - 1; i >= 0; --i)
      if (history[i].present) { if (i2 < 0) i2 = i; else { i1 = i; break; } }
    if (i1 < 0) throw std::runtime_error("ocsort: unfreeze without two observations");
    const HistObs b1 = history[i1], b2 = history[i2];
    auto wh = [](const HistObs& b, double& w, double& h) {
      if (b.f32) { w = (double)std::sqrt((float)b.v[2] * (float)b.v[3]); h = (double)std::sqrt((float)b.v[2] / (float)b.v[3]); }
      else { w = std::sqrt(b.v[2] * b.v[3]); h = std::sqrt(b.v[2] / b.v[3]); }
    };
    double w1, h1, w2, h2;
    wh(b1, w1, h1); wh(b2, w2, h2);
    const bool f = b1.f32 && b2.f32;             // float32 - float32 stays float32 before the division by an int64
    const double gap = (double)(i2 - i1);
    const double ddx = (f ? (double)((float)b2.v[0] - (float)b1.v[0]) : b2.v[0] - b1.v[0]) / gap;
    const double ddy = (f ? (double)((float)b2.v[1] - (float)b1.v[1]) : b2.v[1] - b1.v[1]) / gap;
    const double ddw = (f ? (double)((float)w2 - (float)w1) : w2 - w1) / gap;
    const double ddh = (f ? (double)((float)h2 - (float)h1) : h2 - h1) / gap;
    // restore the snapshot taken at the first missed frame (self.__dict__ = self.attr_saved)
    std::unique_ptr<Snapshot> s = std::move(saved);
    std::memcpy(x, s->x, sizeof(x)); std::memcpy(P, s->P, sizeof(P));
    history = std::move(s->history);
    observed = true;
    const bool had = s->had_saved;
    saved.reset();
    if (had) { saved.reset(new Snapshot()); saved->had_saved = false; }      // only its truthiness is ever read again
    const int n = i2 - i1;
    for (int i = 0; i < n; ++i) {
      const double bx = b1.v[0] + (i + 1) * ddx, by = b1.v[1] + (i + 1) * ddy;
      const double w = w1 + (i + 1) * ddw, h = h1 + (i + 1) * ddh;
      HistObs o; o.present = true; o.f32 = false;
      o.v[0] = bx; o.v[1] = by; o.v[2] = w * h; o.v[3] = w / h;
      history.push_back(o);
      correct(o.v);
      if (i != n - 1) predict();
    }
  }

  void update(const HistObs* z) {                // kalmanfilter.py:107-129
    if (!z) {
      history.push_back(HistObs());
      if (observed) {
        std::unique_ptr<Snapshot> s(new Snapshot());
        std::memcpy(s->x, x, sizeof(x)); std::memcpy(s->P, P, sizeof(P));
        s->history = history; s->had_saved = (bool)saved;
        saved = std::move(s);
      }
      observed = false;
      return;
    }
    history.push_back(*z);
    if (!observed && saved) unfreeze();
    observed = true;
    correct(z->v);
  }
};

struct Box5 { float v[5]; };                     // x1,y1,x2,y2,score as float32 (a row of `dets`)

struct Track {
  Kalman kf;
  int id = 0, age = 0, hits = 0, hit_streak = 0, time_since_update = 0, delta_t = 3;
  bool has_obs = false;                          // last_observation is a real box (not the [-1]*5 placeholder)
  Box5 last{};
  std::vector<std::pair<int, Box5>> observations;   // age -> box, ages strictly increasing
  std::vector<std::pair<int, float>> occurrences;   // class id -> summed score (float32 sums, as numpy 2 adds them), insertion ordered
  int class_id = 0;
  float score = 0.f;
  double velocity[2] = {0, 0};                   // (dy, dx) unit direction, float32 values once set
  double avg_vel[2] = {0, 0};
  double speed = 0;

  bool last_ok() const {                                      // `last_observation.sum() >= 0`: a real box, not the placeholder
    if (!has_obs) return false;
    float s = 0.f; for (float v : last.v) s += v;
    return s >= 0.f;
  }
  static void box_to_z(const Box5& b, HistObs& z) {          // convert_bbox_to_z in float32 (ocsort.py:22-34)
    const float w = b.v[2] - b.v[0], h = b.v[3] - b.v[1];
    const float x = b.v[0] + w / 2.f, y = b.v[1] + h / 2.f;
    const float s = w * h, r = w / (h + 1e-6f);
    z.present = true; z.f32 = true; z.v[0] = x; z.v[1] = y; z.v[2] = s; z.v[3] = r;
  }
  const Box5* obs_at(int a) const {
    for (auto it = observations.rbegin(); it != observations.rend(); ++it) { if (it->first == a) return &it->second; if (it->first < a) break; }
    return nullptr;
  }
  void add_occurrence(int cls, float w) {
    for (auto& o : occurrences) if (o.first == cls) { o.second += w; return; }
    occurrences.emplace_back(cls, w);
  }
  void update(const Box5* b, float sc, int cls) {            // ocsort.py:107-148
    if (!b) { kf.update(nullptr); return; }
    add_occurrence(cls, sc);
    {                                                        // max(dict, key=dict.get): first key holding the maximum
      size_t best = 0;
      for (size_t i = 1; i < occurrences.size(); ++i) if (occurrences[i].second > occurrences[best].second) best = i;
      class_id = occurrences[best].first;
    }
    if (last_ok()) {
      const Box5* prev = nullptr;
      for (int i = 0; i < delta_t && !prev; ++i) prev = obs_at(age - (delta_t - i));
      if (!prev) prev = &last;
      const float cx1 = (prev->v[0] + prev->v[2]) / 2.f, cy1 = (prev->v[1] + prev->v[3]) / 2.f;
      const float cx2 = (b->v[0] + b->v[2]) / 2.f, cy2 = (b->v[1] + b->v[3]) / 2.f;
      const float dy = cy2 - cy1, dx = cx2 - cx1;
      const float norm = std::sqrt(dy * dy + dx * dx) + 1e-6f;
      velocity[0] = dy / norm; velocity[1] = dx / norm;
      avg_vel[0] += (double)(dy / (float)age); avg_vel[1] += (double)(dx / (float)age);
      speed = std::fabs(avg_vel[0]) + std::fabs(avg_vel[1]);
    }
    last = *b; has_obs = true;
    if (!observations.empty() && observations.back().first == age) observations.back().second = *b;
    else observations.emplace_back(age, *b);
    time_since_update = 0; ++hits; ++hit_streak;
    HistObs z; box_to_z(*b, z);
    kf.update(&z);
  }
  void state_box(double (&o)[4]) const {                     // convert_x_to_bbox (ocsort.py:37-47)
    const double w = std::sqrt(kf.x[2] * kf.x[3]), h = kf.x[2] / w;
    o[0] = kf.x[0] - w / 2.; o[1] = kf.x[1] - h / 2.; o[2] = kf.x[0] + w / 2.; o[3] = kf.x[1] + h / 2.;
  }
  void predict(double (&o)[4]) {                             // ocsort.py:150-162
    if (kf.x[6] + kf.x[2] <= 0) kf.x[6] *= 0.0;
    kf.predict();
    ++age;
    if (time_since_update > 0) hit_streak = 0;
    ++time_since_update;
    state_box(o);
  }
  void k_previous(int k, double (&o)[5]) const {             // k_previous_obs (ocsort.py:11-19)
    if (observations.empty()) { for (double& v : o) v = -1; return; }
    const Box5* p = nullptr;
    for (int i = 0; i < k && !p; ++i) p = obs_at(age - (k - i));
    if (!p) p = &observations.back().second;                 // the entry with the largest age
    for (int i = 0; i < 5; ++i) o[i] = p->v[i];
  }
};

inline double iou_f32_f64(const float* d, const double* t) {             // association.py:3-19 with a float32 first operand
  const double xx1 = std::max((double)d[0], t[0]), yy1 = std::max((double)d[1], t[1]);
  const double xx2 = std::min((double)d[2], t[2]), yy2 = std::min((double)d[3], t[3]);
  // np.maximum / np.minimum propagate NaN
  const bool nan = std::isnan(t[0]) || std::isnan(t[1]) || std::isnan(t[2]) || std::isnan(t[3]);
  if (nan) return std::numeric_limits<double>::quiet_NaN();
  const double w = std::max(0.0, xx2 - xx1), h = std::max(0.0, yy2 - yy1), wh = w * h;
  const float a1 = (d[2] - d[0]) * (d[3] - d[1]);
  const double a2 = (t[2] - t[0]) * (t[3] - t[1]);
  return wh / ((double)a1 + a2 - wh);
}

// greedy assignment on ascending cost (association.py:32-52): walk the pairs in the order of a stable argsort (ties in
// row-major order, NaN last as numpy sorts them) and take every pair whose row and column are still free; stop once all
// rows or all columns are used.  The walk normally ends after a small prefix, so the order is produced lazily from a
// heap on the total order (cost, index) instead of sorting all rows*cols pairs.
void greedy_assign(const std::vector<double>& cost, int rows, int cols, std::vector<std::pair<int, int>>& out) {
  out.clear();
  if (rows == 0 || cols == 0) return;
  auto after = [&](int a, int b) {                     // true if a comes after b in the argsort order
    const double ca = cost[a], cb = cost[b];
    const bool na = std::isnan(ca), nb = std::isnan(cb);
    if (na || nb) return na != nb ? na : a > b;
    if (ca != cb) return ca > cb;
    return a > b;
  };
  std::vector<int> heap(cost.size());
  for (size_t i = 0; i < heap.size(); ++i) heap[i] = (int)i;
  std::make_heap(heap.begin(), heap.end(), after);
  std::vector<char> ru(rows, 0), cu(cols, 0);
  int nr = 0, nc = 0;
  auto end = heap.end();
  while (end != heap.begin()) {
    std::pop_heap(heap.begin(), end, after); --end;
    const int f = *end, r = f / cols, c = f - r * cols;
    if (ru[r] || cu[c]) continue;
    out.emplace_back(r, c);
    ru[r] = cu[c] = 1; ++nr; ++nc;
    if (nr == rows || nc == cols) break;
  }
}

// The same walk for the first association, where the cost -(iou + angle term) would need a sqrt and an acos for each
// of the dets x tracks pairs, and a crowded scene has tens of thousands of overlapping pairs.
// |angle term| <= bound[row] = 0.5*|inertia|*score, so
//  * a pair with IoU exactly 0 costs between -bmax and +bmax: every pair below -bmax has a non-zero IoU;
//  * -(iou + bound[row]) is a lower bound of a pair's cost that needs no transcendental.
// Phase 1 walks the pairs below -bmax in exact argsort order.  Each row keeps its candidates in a small heap keyed on
// the lower bound, and a heap over the row heads yields the global minimum.  A pair is priced exactly only when it
// surfaces with its row and column still free, and goes back under its exact cost (a priced entry on top is the true
// minimum because every other key is a lower bound; unpriced entries win key ties so that an equal-cost pair with a
// smaller index is never overtaken).  When a row is assigned, the rest of its candidates are dropped unseen.
// Phase 2: whatever the walk does afterwards only involves rows and columns that are still free, so only that block is
// priced and sorted.  The assignment sequence is the one a stable argsort of the dense cost matrix gives.
struct Cand { double c; int idx; bool exact; };
inline bool cand_after(const Cand& a, const Cand& b) {       // heap order: true if a must come out after b
  if (a.c != b.c) return a.c > b.c;
  if (a.exact != b.exact) return a.exact;
  return a.idx > b.idx;
}

template <class CostFn>
void greedy_first(std::vector<std::vector<Cand>>& rowc, int rows, int cols, double bmax, CostFn cost,
                  std::vector<std::pair<int, int>>& out) {
  out.clear();
  if (rows == 0 || cols == 0) return;
  std::vector<Cand> heads;                                   // one entry per row with candidates: its current minimum
  for (int r = 0; r < rows; ++r) {
    std::vector<Cand>& v = rowc[r];
    if (v.empty()) continue;
    std::make_heap(v.begin(), v.end(), cand_after);
    heads.push_back(v.front());
  }
  std::make_heap(heads.begin(), heads.end(), cand_after);
  std::vector<char> ru(rows, 0), cu(cols, 0);
  int nr = 0, nc = 0;
  while (!heads.empty()) {
    std::pop_heap(heads.begin(), heads.end(), cand_after);
    const Cand it = heads.back(); heads.pop_back();
    const int r = it.idx / cols, c = it.idx - r * cols;
    std::vector<Cand>& v = rowc[r];                          // `it` is v.front()
    std::pop_heap(v.begin(), v.end(), cand_after); v.pop_back();
    bool row_done = false;
    if (!cu[c]) {
      if (!it.exact) {
        const double k = cost(r, c);
        if (k < -bmax) { v.push_back(Cand{k, it.idx, true}); std::push_heap(v.begin(), v.end(), cand_after); }
      } else {
        out.emplace_back(r, c);
        ru[r] = cu[c] = 1; ++nr; ++nc;
        if (nr == rows || nc == cols) return;
        row_done = true;
      }
    }
    if (!row_done && !v.empty()) { heads.push_back(v.front()); std::push_heap(heads.begin(), heads.end(), cand_after); }
  }
  std::vector<int> fr, fc;
  for (int r = 0; r < rows; ++r) if (!ru[r]) fr.push_back(r);
  for (int c = 0; c < cols; ++c) if (!cu[c]) fc.push_back(c);
  struct Tail { double c; int idx; };
  std::vector<Tail> items;
  items.reserve(fr.size() * fc.size());
  for (int r : fr) for (int c : fc) items.push_back(Tail{cost(r, c), r * cols + c});
  std::sort(items.begin(), items.end(), [](const Tail& a, const Tail& b) {
    const bool na = std::isnan(a.c), nb = std::isnan(b.c);
    if (na || nb) return na != nb ? nb : a.idx < b.idx;
    if (a.c != b.c) return a.c < b.c;
    return a.idx < b.idx;
  });
  for (const Tail& it : items) {
    const int r = it.idx / cols, c = it.idx - r * cols;
    if (ru[r] || cu[c]) continue;
    out.emplace_back(r, c);
    ru[r] = cu[c] = 1; ++nr; ++nc;
    if (nr == rows || nc == cols) return;
  }
}

void setdiff_sorted(std::vector<int>& a, const std::vector<int>& remove) {   // np.setdiff1d: sorted unique difference
  std::sort(a.begin(), a.end());
  a.erase(std::unique(a.begin(), a.end()), a.end());
  std::vector<int> r;
  for (int v : a) if (std::find(remove.begin(), remove.end(), v) == remove.end()) r.push_back(v);
  a.swap(r);
}

}  // namespace

struct cc_ocsort {
  int max_age = 30, min_hits = 3, delta_t = 3, use_byte = 0;
  double iou_threshold = 0.3, inertia = 0.2;
  int frame_count = 0, next_id = 0;
  std::vector<std::unique_ptr<Track>> tracks;
  std::vector<std::vector<Cand>> cand_scratch;
  std::vector<double> iou_scratch;      // dets x tracks matrix, kept between frames (a fresh 0.5 MB vector per frame
                                        // per camera is an mmap/munmap pair and serialises the camera threads in the kernel)
};

namespace {

constexpr double kPi = 3.141592653589793;

void ocsort_update(cc_ocsort& S, const float* rows, int n, double det_thresh, std::vector<double>& out) {
  ++S.frame_count;
  const float thr = (float)det_thresh, low = 0.1f;            // python floats are weak against float32 arrays
  std::vector<Box5> dets, dets2; std::vector<int> cls, cls2;
  for (int i = 0; i < n; ++i) {
    const float* r = rows + (size_t)i * 6;
    Box5 b; std::memcpy(b.v, r, 5 * sizeof(float));
    const int c = (int)r[5];                                  // astype(int) truncates
    if (r[4] > low && r[4] < thr) { dets2.push_back(b); cls2.push_back(c); }
    if (r[4] > thr) { dets.push_back(b); cls.push_back(c); }
  }
  const int D = (int)dets.size(), T = (int)S.tracks.size();
  std::vector<double> trks((size_t)T * 4), vel((size_t)T * 2), lastb((size_t)T * 5), kobs((size_t)T * 5);
  for (int t = 0; t < T; ++t) {
    Track& k = *S.tracks[t];
    double p[4]; k.predict(p);
    std::memcpy(&trks[(size_t)t * 4], p, sizeof(p));
    vel[t * 2] = k.velocity[0]; vel[t * 2 + 1] = k.velocity[1];
    for (int i = 0; i < 5; ++i) lastb[(size_t)t * 5 + i] = k.has_obs ? (double)k.last.v[i] : -1.0;
    double ko[5]; k.k_previous(S.delta_t, ko);
    std::memcpy(&kobs[(size_t)t * 5], ko, sizeof(ko));
  }

  // ---- first association (association.py:54-109)
  std::vector<std::pair<int, int>> matches;
  std::vector<int> um_d, um_t;
  if (T == 0) {
    for (int d = 0; d < D; ++d) um_d.push_back(d);
  } else {
    // IoU for every pair (cheap); the velocity-direction term needs sqrt + acos per pair and is evaluated only where the
    // greedy walk can observe it (see greedy_first).
    std::vector<double>& iou = S.iou_scratch;
    iou.resize((size_t)D * T);
    std::vector<std::vector<Cand>>& rowc = S.cand_scratch;   // per detection: pairs that can cost less than -bmax
    if ((int)rowc.size() < D) rowc.resize(D);
    std::vector<double> bound(D);                            // |angle term| <= 0.5*|inertia|*score
    double bmax = 0.0;
    for (int d = 0; d < D; ++d) { bound[d] = (0.5 * std::fabs(S.inertia)) * std::fabs((double)dets[d].v[4]); bmax = std::max(bmax, bound[d]); }
    std::vector<int> rs(D, 0), cs(T, 0);                     // pairs above the IoU threshold per row / column
    {
      std::vector<double> a2(T); std::vector<char> tnan(T);
      for (int t = 0; t < T; ++t) {
        const double* b = &trks[(size_t)t * 4];
        a2[t] = (b[2] - b[0]) * (b[3] - b[1]);
        tnan[t] = std::isnan(b[0]) || std::isnan(b[1]) || std::isnan(b[2]) || std::isnan(b[3]);
      }
      for (int d = 0; d < D; ++d) {
        const float* v = dets[d].v;
        const double x1 = v[0], y1 = v[1], x2 = v[2], y2 = v[3];
        const double a1 = (double)((v[2] - v[0]) * (v[3] - v[1]));       // float32 product, as numpy computes it
        double* row = &iou[(size_t)d * T];
        std::vector<Cand>& rc = rowc[d];
        rc.clear();
        for (int t = 0; t < T; ++t) {
          const double* b = &trks[(size_t)t * 4];
          const double w = std::min(x2, b[2]) - std::max(x1, b[0]), h = std::min(y2, b[3]) - std::max(y1, b[1]);
          // disjoint boxes with a positive area sum: 0 / positive = 0 exactly, skip the division (most pairs)
          if ((w <= 0.0 || h <= 0.0) && !tnan[t] && a1 + a2[t] > 0.0) { row[t] = 0.0; continue; }
          const double o = iou_f32_f64(v, b);
          row[t] = o;
          if (o > S.iou_threshold) { ++rs[d]; ++cs[t]; }
          if (o != 0.0 && !std::isnan(o)) {                  // zero IoU cannot be below -bmax; NaN sorts last (phase 2)
            const double lb = -(o + bound[d]);
            if (lb < -bmax) rc.push_back(Cand{lb, d * T + t, false});
          }
        }
      }
    }
    std::vector<float> dcx(D), dcy(D);
    for (int d = 0; d < D; ++d) { dcx[d] = (dets[d].v[0] + dets[d].v[2]) / 2.f; dcy[d] = (dets[d].v[1] + dets[d].v[3]) / 2.f; }
    auto pair_cost = [&](int d, int t) -> double {           // -(iou + angle_diff_cost)  (association.py:58-82)
      const double* ko = &kobs[(size_t)t * 5];
      const double cx2 = (ko[0] + ko[2]) / 2.0, cy2 = (ko[1] + ko[3]) / 2.0;
      const double valid = ko[4] < 0 ? 0.0 : 1.0;
      double adc;
      if ((valid == 0.0 || (vel[t * 2] == 0.0 && vel[t * 2 + 1] == 0.0)) && std::isfinite(cx2) && std::isfinite(cy2)) {
        adc = 0.0;                                           // acos(0) is exactly pi/2 in double: the angle term is exactly 0
      } else {
        double dx = (double)dcx[d] - cx2, dy = (double)dcy[d] - cy2;
        const double norm = std::sqrt(dx * dx + dy * dy) + 1e-6;
        dx /= norm; dy /= norm;
        double c = vel[t * 2 + 1] * dx + vel[t * 2] * dy;
        c = c < -1.0 ? -1.0 : (c > 1.0 ? 1.0 : c);           // np.clip (NaN passes through)
        const double ang = (kPi / 2.0 - std::fabs(std::acos(c))) / kPi;
        adc = ((valid * ang) * S.inertia) * (double)dets[d].v[4];
      }
      return -(iou[(size_t)d * T + t] + adc);
    };
    std::vector<std::pair<int, int>> cand;
    if (D > 0) {
      if (*std::max_element(rs.begin(), rs.end()) == 1 && *std::max_element(cs.begin(), cs.end()) == 1) {
        for (int d = 0; d < D; ++d) for (int t = 0; t < T; ++t) if (iou[(size_t)d * T + t] > S.iou_threshold) cand.emplace_back(d, t);
      } else {
        greedy_first(rowc, D, T, bmax, pair_cost, cand);
      }
    }
    std::vector<char> dm(D, 0), tm(T, 0);
    for (auto& m : cand) { dm[m.first] = 1; tm[m.second] = 1; }
    for (int d = 0; d < D; ++d) if (!dm[d]) um_d.push_back(d);
    for (int t = 0; t < T; ++t) if (!tm[t]) um_t.push_back(t);
    for (auto& m : cand) {
      if (iou[(size_t)m.first * T + m.second] < S.iou_threshold) { um_d.push_back(m.first); um_t.push_back(m.second); }
      else matches.push_back(m);
    }
  }
  for (auto& m : matches) S.tracks[m.second]->update(&dets[m.first], dets[m.first].v[4], cls[m.first]);

  // ---- BYTE: low-score detections against the still unmatched tracks (ocsort.py:232-251)
  if (S.use_byte && !dets2.empty() && !um_t.empty()) {
    const int R = (int)dets2.size(), Cn = (int)um_t.size();
    std::vector<double> il((size_t)R * Cn), neg((size_t)R * Cn);
    bool above = false;
    for (int r = 0; r < R; ++r) for (int c = 0; c < Cn; ++c) {
      const double o = iou_f32_f64(dets2[r].v, &trks[(size_t)um_t[c] * 4]);
      il[(size_t)r * Cn + c] = o; neg[(size_t)r * Cn + c] = -o;
    }
    {   // ndarray.max() returns NaN if any element is NaN, and NaN > thr is false
      double mx = -std::numeric_limits<double>::infinity(); bool nan = false;
      for (double v : il) { if (std::isnan(v)) nan = true; else mx = std::max(mx, v); }
      above = !nan && mx > S.iou_threshold;
    }
    if (above) {
      std::vector<std::pair<int, int>> mi; greedy_assign(neg, R, Cn, mi);
      std::vector<int> rm;
      for (auto& m : mi) {
        if (il[(size_t)m.first * Cn + m.second] < S.iou_threshold) continue;
        const int ti = um_t[m.second];
        S.tracks[ti]->update(&dets2[m.first], dets2[m.first].v[4], cls2[m.first]);
        rm.push_back(ti);
      }
      setdiff_sorted(um_t, rm);
    }
  }

  // ---- observation-centric recovery: unmatched detections against the tracks' last observations (ocsort.py:253-277)
  if (!um_d.empty() && !um_t.empty()) {
    const int R = (int)um_d.size(), Cn = (int)um_t.size();
    std::vector<double> il((size_t)R * Cn), neg((size_t)R * Cn);
    for (int r = 0; r < R; ++r) for (int c = 0; c < Cn; ++c) {
      const double o = iou_f32_f64(dets[um_d[r]].v, &lastb[(size_t)um_t[c] * 5]);
      il[(size_t)r * Cn + c] = o; neg[(size_t)r * Cn + c] = -o;
    }
    double mx = -std::numeric_limits<double>::infinity(); bool nan = false;
    for (double v : il) { if (std::isnan(v)) nan = true; else mx = std::max(mx, v); }
    if (!nan && mx > S.iou_threshold) {
      std::vector<std::pair<int, int>> mi; greedy_assign(neg, R, Cn, mi);
      std::vector<int> rd, rt;
      for (auto& m : mi) {
        if (il[(size_t)m.first * Cn + m.second] < S.iou_threshold) continue;
        const int di = um_d[m.first], ti = um_t[m.second];
        S.tracks[ti]->update(&dets[di], dets[di].v[4], cls[di]);
        rd.push_back(di); rt.push_back(ti);
      }
      setdiff_sorted(um_d, rd); setdiff_sorted(um_t, rt);
    }
  }
  for (int t : um_t) S.tracks[t]->update(nullptr, 0.f, 0);

  // ---- births (ocsort.py:282-288)
  for (int d : um_d) {
    std::unique_ptr<Track> k(new Track());
    k->delta_t = S.delta_t;
    HistObs z; Track::box_to_z(dets[d], z);
    for (int i = 0; i < 4; ++i) k->kf.x[i] = z.v[i];
    k->id = S.next_id++;
    k->class_id = cls[d]; k->score = dets[d].v[4];
    k->add_occurrence(cls[d], 1.0f);
    S.tracks.push_back(std::move(k));
  }

  // ---- report + reap, newest track first (ocsort.py:289-308)
  out.clear();
  for (int i = (int)S.tracks.size() - 1; i >= 0; --i) {
    Track& k = *S.tracks[i];
    double d[4];
    if (!k.last_ok()) k.state_box(d);
    else for (int j = 0; j < 4; ++j) d[j] = (double)k.last.v[j];
    if (k.time_since_update < 1 && (k.hit_streak >= S.min_hits || S.frame_count <= S.min_hits)) {
      const double row[9] = {d[0], d[1], d[2] - d[0], d[3] - d[1], (double)(k.id + 1), (double)k.age, (double)k.class_id,
                             (double)k.score, k.speed};
      out.insert(out.end(), row, row + 9);
    }
    if (k.time_since_update > S.max_age && (k.speed > 2 || k.time_since_update > 600)) S.tracks.erase(S.tracks.begin() + i);
  }
}

}  // namespace

#define CC_API_BEGIN try {
#define CC_API_END   return 0; } catch (const std::exception& e) { cc::set_error(e.what()); return -1; }

extern "C" {

int cc_ocsort_create(cc_ocsort** h, int max_age, int min_hits, double iou_threshold, int delta_t, double inertia, int use_byte) {
  CC_API_BEGIN
  if (!h || delta_t < 1) throw std::invalid_argument("cc_ocsort_create: bad argument");
  std::unique_ptr<cc_ocsort> s(new cc_ocsort());
  s->max_age = max_age; s->min_hits = min_hits; s->iou_threshold = iou_threshold; s->delta_t = delta_t;
  s->inertia = inertia; s->use_byte = use_byte;
  *h = s.release();
  CC_API_END
}

int cc_ocsort_update(cc_ocsort* h, const float* dets, int n, double det_thresh, double* out, int cap, int* n_out) {
  CC_API_BEGIN
  if (!h || (n > 0 && !dets) || n < 0 || !n_out || (cap > 0 && !out)) throw std::invalid_argument("cc_ocsort_update: bad argument");
  std::vector<double> rows;
  ocsort_update(*h, dets, n, det_thresh, rows);
  const int m = (int)(rows.size() / 9);
  *n_out = m;
  if (m > cap) throw std::length_error("cc_ocsort_update: output capacity too small (state already advanced)");
  if (m) std::memcpy(out, rows.data(), rows.size() * sizeof(double));
  CC_API_END
}

// One frame for each of `count` cameras in one call: camera c reads rows_per rows at dets + c*rows_per*6 and writes
// at most cap_per rows at out + c*cap_per*9.  Cameras are independent, so they are spread over n_threads host threads.
int cc_ocsort_update_many(cc_ocsort* const* hs, int count, const float* dets, int rows_per, double det_thresh, double* out,
                          int cap_per, int* n_out, int n_threads) {
  CC_API_BEGIN
  if (!hs || count < 0 || rows_per < 0 || (rows_per > 0 && !dets) || !out || cap_per <= 0 || !n_out)
    throw std::invalid_argument("cc_ocsort_update_many: bad argument");
  for (int c = 0; c < count; ++c) if (!hs[c]) throw std::invalid_argument("cc_ocsort_update_many: null tracker");
  std::vector<std::string> errs(count);
  auto work = [&](int lo, int hi) {
    std::vector<double> rows;
    for (int c = lo; c < hi; ++c) {
      try {
        ocsort_update(*hs[c], dets + (size_t)c * rows_per * 6, rows_per, det_thresh, rows);
        const int m = (int)(rows.size() / 9);
        n_out[c] = m;
        if (m > cap_per) { errs[c] = "output capacity too small"; continue; }
        if (m) std::memcpy(out + (size_t)c * cap_per * 9, rows.data(), rows.size() * sizeof(double));
      } catch (const std::exception& e) { errs[c] = e.what(); }
    }
  };
  const int nt = std::max(1, std::min(n_threads, count));
  if (nt == 1) work(0, count);
  else {
    std::vector<std::thread> th;
    for (int t = 0; t < nt; ++t) th.emplace_back(work, (int)((long)count * t / nt), (int)((long)count * (t + 1) / nt));
    for (auto& t : th) t.join();
  }
  for (int c = 0; c < count; ++c) if (!errs[c].empty()) throw std::runtime_error("cc_ocsort_update_many: camera " + std::to_string(c) + ": " + errs[c]);
  CC_API_END
}

int cc_ocsort_num_tracks(cc_ocsort* h, int* n) {
  CC_API_BEGIN
  if (!h || !n) throw std::invalid_argument("cc_ocsort_num_tracks: bad argument");
  *n = (int)h->tracks.size();
  CC_API_END
}

void cc_ocsort_destroy(cc_ocsort* h) { delete h; }

}  // extern "C"
