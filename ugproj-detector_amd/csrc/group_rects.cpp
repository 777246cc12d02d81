// group_rects.cpp -- host rectangle grouping of detectMultiScale's raw candidates.
// Semantics: cv::groupRectangles(rects, minNeighbors, eps = 0.2) with cv::partition and
// SimilarRects, i.e. the post-processing inside the call at reference
// src/tracker/detector.cc:25-31 (minNeighbors = 2 there).
#include "group_rects.h"

#include <algorithm>
#include <cmath>
#include <numeric>

namespace hc {

namespace {
inline int rne(double v) { return (int)std::nearbyint(v); }
inline int rne(float v) { return (int)std::nearbyintf(v); }

inline bool similar(const hc_rect& p, const hc_rect& q, double eps) {
  const double delta = eps * (std::min(p.width, q.width) + std::min(p.height, q.height)) * 0.5;
  return std::abs(p.x - q.x) <= delta && std::abs(p.y - q.y) <= delta &&
         std::abs(p.x + p.width - q.x - q.width) <= delta &&
         std::abs(p.y + p.height - q.y - q.height) <= delta;
}
}  // namespace

void group_rectangles(std::vector<hc_rect>& rects, int group_threshold, double eps) {
  if (group_threshold <= 0 || rects.empty()) return;
  const int n = (int)rects.size();

  // equivalence classes = connected components of the similarity graph (disjoint sets by size).
  // Only pairs that can be similar are tested: similar(p, q) needs |p.x - q.x| <= delta and
  // delta <= eps/2 * (p.width + p.height), so in x-sorted order the partners of p that follow it
  // lie within that bound.  The set of edges found is exactly the set of similar pairs.
  std::vector<int> up(n), sz(n, 1), byx(n);
  std::iota(up.begin(), up.end(), 0);
  std::iota(byx.begin(), byx.end(), 0);
  std::sort(byx.begin(), byx.end(), [&](int a, int b) { return rects[a].x < rects[b].x; });
  auto root = [&](int v) {
    while (up[v] != v) { up[v] = up[up[v]]; v = up[v]; }
    return v;
  };
  for (int ia = 0; ia < n; ia++) {
    const int i = byx[ia];
    const hc_rect& p = rects[i];
    const double reach = eps * (p.width + p.height) * 0.5;
    for (int ib = ia + 1; ib < n; ib++) {
      const int j = byx[ib];
      if ((double)(rects[j].x - p.x) > reach) break;
      if (!similar(p, rects[j], eps)) continue;
      int a = root(i), b = root(j);
      if (a == b) continue;
      if (sz[a] < sz[b]) std::swap(a, b);
      up[b] = a;
      sz[a] += sz[b];
    }
  }
  // class ids in order of each class's first member (what cv::partition's final pass yields)
  std::vector<int> cls_of_root(n, -1), cls(n);
  int ncls = 0;
  for (int i = 0; i < n; i++) {
    const int r = root(i);
    if (cls_of_root[r] < 0) cls_of_root[r] = ncls++;
    cls[i] = cls_of_root[r];
  }

  struct Acc { long long x = 0, y = 0, w = 0, h = 0; int n = 0; };
  std::vector<Acc> acc(ncls);
  for (int i = 0; i < n; i++) {
    Acc& a = acc[cls[i]];
    a.x += rects[i].x; a.y += rects[i].y; a.w += rects[i].width; a.h += rects[i].height; a.n++;
  }
  std::vector<hc_rect> mean(ncls);
  for (int k = 0; k < ncls; k++) {
    const float inv = 1.f / (float)acc[k].n;               // f32 reciprocal, f32 products
    mean[k] = hc_rect{rne((float)(int)acc[k].x * inv), rne((float)(int)acc[k].y * inv),
                      rne((float)(int)acc[k].w * inv), rne((float)(int)acc[k].h * inv)};
  }

  rects.clear();
  for (int i = 0; i < ncls; i++) {
    const int n1 = acc[i].n;
    if (n1 <= group_threshold) continue;                   // too few neighbours
    const hc_rect& r1 = mean[i];
    bool nested = false;
    for (int j = 0; j < ncls && !nested; j++) {
      const int n2 = acc[j].n;
      if (j == i || n2 <= group_threshold) continue;
      const hc_rect& r2 = mean[j];
      const int dx = rne(r2.width * eps), dy = rne(r2.height * eps);
      nested = r1.x >= r2.x - dx && r1.y >= r2.y - dy && r1.x + r1.width <= r2.x + r2.width + dx &&
               r1.y + r1.height <= r2.y + r2.height + dy && (n2 > std::max(3, n1) || n1 < 3);
    }
    if (!nested) rects.push_back(r1);
  }
}

}  // namespace hc
