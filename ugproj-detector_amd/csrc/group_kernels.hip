// group_kernels.hip -- rectangle grouping of a batch's raw candidates ON the device:
//   k_group_bin      every raw candidate key into its frame's bin (frame counters privatised in LDS)
//   k_group_frames   one 256-thread workgroup per frame: keys sorted to upstream's single-thread order (rank by counting, LDS),
//                    decoded to rectangles (Rect(cvRound(x * scale), ...), A.6c), cv::partition by SimilarRects(0.2) as
//                    connected components (hook-under-the-smaller-root union-find on LDS atomics), class order = order of each
//                    class's first member, f32 means, the minNeighbors / nested-rectangle filter; the grouped boxes go to
//                    coherent page-locked host memory (the batch's k_publish, a later launch of the same stream, then
//                    publishes the sequence word the host polls).
// The arithmetic is cv::groupRectangles' (what hc::group_rectangles in group_rects.cpp restates for the host): integer
// sums, one f32 reciprocal per class (formed as (float)(1.0 / n): f64 -> f32 rounding of a quotient of two f32 values
// is innocuous, 53 >= 2 * 24 + 2, so it IS the f32 quotient), f32 products, round-half-even; the similarity bound and the
// nested-rectangle margins in f64.  Compiled with -ffp-contract=off like everything else.  It replaces the host's sort +
// decode + grouping of reference src/tracker/detector.cc:25-31's post-processing; a frame with more candidates than a bin
// holds, or more boxes than its output slot, is flagged and grouped on the host from the raw keys as before.

#include <hip/hip_runtime.h>

#include "device_types.h"
#include "kernels.h"
#include "device_common.h"

namespace hc {

#define HC_LAUNCH_CHECK() do { hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) return (int)e_; } while (0)

// Frame counters are privatised: a workgroup counts its contiguous slice of the candidate list per frame in LDS, reserves a
// range of every frame's bin with ONE global atomic per (workgroup, frame), and places its keys by LDS atomics -- 64 atomics
// per frame counter instead of one per candidate (210 at the bench shape: same-address atomics serialise in L2; 0.086 ->
// 0.008 ms per 512 frames).
constexpr int GB_FRAMES = 4096;      // frames the LDS histogram covers; larger batches take one global atomic per key
__global__ __launch_bounds__(256) void k_group_bin(const unsigned int* __restrict__ count, const unsigned long long* __restrict__ cand,
                                                   unsigned int cap, unsigned int* __restrict__ frame_count,
                                                   unsigned long long* __restrict__ bins, int nframes, int bin_cap) {
  __shared__ unsigned int hist[GB_FRAMES], base[GB_FRAMES];
  const unsigned int n = min(*count, cap);
  if (nframes > GB_FRAMES) {
    for (unsigned int i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u) {
      const unsigned long long k = cand[i];
      const unsigned int f = (unsigned int)(k >> 40);
      if (f >= (unsigned int)nframes) continue;
      const unsigned int pos = atomicAdd(&frame_count[f], 1u);
      if (pos < (unsigned int)bin_cap) bins[(size_t)f * HC_GROUP_BIN + pos] = k;
    }
    return;
  }
  const unsigned int per = (n + gridDim.x - 1) / gridDim.x, lo = min(n, blockIdx.x * per), hi = min(n, lo + per);
  for (int f = threadIdx.x; f < nframes; f += 256) hist[f] = 0u;
  __syncthreads();
  for (unsigned int i = lo + threadIdx.x; i < hi; i += 256u) {
    const unsigned int f = (unsigned int)(cand[i] >> 40);
    if (f < (unsigned int)nframes) atomicAdd(&hist[f], 1u);
  }
  __syncthreads();
  for (int f = threadIdx.x; f < nframes; f += 256) {
    const unsigned int c = hist[f];
    base[f] = c ? atomicAdd(&frame_count[f], c) : 0u;
    hist[f] = 0u;
  }
  __syncthreads();
  for (unsigned int i = lo + threadIdx.x; i < hi; i += 256u) {
    const unsigned long long k = cand[i];
    const unsigned int f = (unsigned int)(k >> 40);
    if (f >= (unsigned int)nframes) continue;
    const unsigned int pos = base[f] + atomicAdd(&hist[f], 1u);
    if (pos < (unsigned int)bin_cap) bins[(size_t)f * HC_GROUP_BIN + pos] = k;
  }
}

namespace {

// exclusive prefix sums of v[0 .. n) in place (n <= HC_GROUP_BIN), total returned to every thread; tmp: HC_GROUP_BIN words
__device__ int block_exclusive_scan(int* v, int* tmp, int n) {
  const int t = threadIdx.x;
  // four consecutive entries per thread, then a Hillis-Steele scan of the 256 thread totals
  int a[4], s = 0;
  for (int q = 0; q < 4; q++) {
    const int i = t * 4 + q;
    a[q] = i < n ? v[i] : 0;
    s += a[q];
  }
  tmp[t] = s;
  __syncthreads();
  for (int d = 1; d < 256; d <<= 1) {
    const int add = t >= d ? tmp[t - d] : 0;
    __syncthreads();
    tmp[t] += add;
    __syncthreads();
  }
  const int total = tmp[255];
  int run = tmp[t] - s;
  for (int q = 0; q < 4; q++) {
    const int i = t * 4 + q;
    if (i < n) v[i] = run;
    run += a[q];
  }
  __syncthreads();
  return total;
}

__device__ __forceinline__ int find_root(const volatile int* parent, int v) {
  int p = parent[v];
  while (p != v) { v = p; p = parent[v]; }
  return v;
}

}  // namespace

__global__ __launch_bounds__(256) void k_group_frames(const unsigned int* __restrict__ frame_count, const unsigned long long* __restrict__ bins,
                                                      const GroupLevel* __restrict__ levels, int semantics, int threshold,
                                                      GroupBox* __restrict__ h_boxes, unsigned int* __restrict__ h_nboxes, int bin_cap,
                                                      int out_cap) {
  __shared__ unsigned long long keys[HC_GROUP_BIN], sorted[HC_GROUP_BIN];
  __shared__ int4 rc[HC_GROUP_BIN];                 // x, y, width, height: the raw rectangles, later the class means
  __shared__ int parent[HC_GROUP_BIN], cls[HC_GROUP_BIN], tmp[256];
  __shared__ int sw[HC_GROUP_BIN], sh[HC_GROUP_BIN], sn[HC_GROUP_BIN];
  int* const sx = reinterpret_cast<int*>(keys);     // class sums of x and y: over the unsorted keys, dead after the sort
  int* const sy = sx + HC_GROUP_BIN;
  const int f = blockIdx.x, t = threadIdx.x;
  const unsigned int nraw = frame_count[f];
  if (nraw > (unsigned int)bin_cap) {                 // the bin kept the first bin_cap only: the host groups this frame
    if (t == 0) { h_nboxes[f] = 0xFFFFFFFFu; __threadfence_system(); }
    return;
  }
  const int n = (int)nraw;
  if (n == 0) {
    if (t == 0) { h_nboxes[f] = 0u; __threadfence_system(); }
    return;
  }
  // ---- upstream's order: frame -> scale -> y -> x == ascending key.  Every key's rank by counting (all lanes read the same key
  // of the list at a time: a broadcast, no bank conflict, no barrier per step -- a bitonic network of 256 keys is 36 barriers);
  // equal keys (the debug entry may hold duplicates) are ordered by their position in the bin
  for (int i = t; i < n; i += 256) keys[i] = bins[(size_t)f * HC_GROUP_BIN + i];
  __syncthreads();
  for (int i = t; i < n; i += 256) {
    const unsigned long long k = keys[i];
    int rank = 0;
    for (int j = 0; j < n; j++) {
      const unsigned long long q = keys[j];
      rank += (q < k || (q == k && j < i)) ? 1 : 0;
    }
    sorted[rank] = k;
  }
  __syncthreads();
  // ---- decode
  for (int i = t; i < n; i += 256) {
    const unsigned long long k = sorted[i];
    const GroupLevel L = levels[(k >> 32) & 0xFF];
    const int gy = (int)((k >> 16) & 0xFFFF), gx = (int)(k & 0xFFFF);
    int x, y;
    if (semantics == 1) {     // 2.4 (B.9): cvRound(x * factor), f64
      x = (int)rint((double)(gx * L.ystep) * L.factor);
      y = (int)rint((double)(gy * L.ystep) * L.factor);
    } else {                  // Rect(cvRound(x * scale), cvRound(y * scale), winSize): int -> f32 times the f32 scale
      x = (int)rintf((float)(gx * L.ystep) * L.scale);
      y = (int)rintf((float)(gy * L.ystep) * L.scale);
    }
    rc[i] = make_int4(x, y, L.win_w, L.win_h);
    parent[i] = i;
    sx[i] = 0; sy[i] = 0; sw[i] = 0; sh[i] = 0; sn[i] = 0;
  }
  __syncthreads();
  // ---- cv::partition: components of the SimilarRects graph.  A root is only ever hooked under a SMALLER root, so
  // parent[v] <= v throughout and a component's final root is its smallest member, its first in upstream's order.
  // Every lane walks the whole list (entry j is one 16-byte broadcast read), four entries per trip so that the reads of a
  // trip are in flight together; a pair is taken once (j > i).
  const double eps = 0.2;
  auto similar = [&](const int4& p, const int4& q) {
    const double delta = eps * (double)(min(p.z, q.z) + min(p.w, q.w)) * 0.5;
    return (double)abs(p.x - q.x) <= delta && (double)abs(p.y - q.y) <= delta &&
           (double)abs(p.x + p.z - q.x - q.z) <= delta && (double)abs(p.y + p.w - q.y - q.w) <= delta;
  };
  auto unite = [&](int a, int b) {
    for (;;) {
      a = find_root(parent, a);
      b = find_root(parent, b);
      if (a == b) break;
      const int hi = max(a, b), lo = min(a, b);
      if (atomicCAS(&parent[hi], hi, lo) == hi) break;
      a = hi; b = lo;      // somebody hooked `hi` meanwhile: walk on from there
    }
  };
  for (int i = t; i < n; i += 256) {
    const int4 p = rc[i];
    int j = 0;
    for (; j + 4 <= n; j += 4) {
      const int4 q0 = rc[j], q1 = rc[j + 1], q2 = rc[j + 2], q3 = rc[j + 3];
      const bool s0 = j > i && similar(p, q0), s1 = j + 1 > i && similar(p, q1), s2 = j + 2 > i && similar(p, q2),
                 s3 = j + 3 > i && similar(p, q3);
      if (s0 | s1 | s2 | s3) {
        if (s0) unite(i, j);
        if (s1) unite(i, j + 1);
        if (s2) unite(i, j + 2);
        if (s3) unite(i, j + 3);
      }
    }
    for (; j < n; j++)
      if (j > i && similar(p, rc[j])) unite(i, j);
  }
  __syncthreads();
  for (int i = t; i < n; i += 256) cls[i] = find_root(parent, i) == i ? 1 : 0;
  __syncthreads();
  const int ncls = block_exclusive_scan(cls, tmp, n);      // cls[root] = class id of that root's class
  // ---- class sums
  for (int i = t; i < n; i += 256) {
    const int c = cls[find_root(parent, i)];
    const int4 p = rc[i];
    atomicAdd(&sx[c], p.x); atomicAdd(&sy[c], p.y); atomicAdd(&sw[c], p.z); atomicAdd(&sh[c], p.w); atomicAdd(&sn[c], 1);
  }
  __syncthreads();
  // ---- class means over the rect array (dead now): f32 reciprocal, f32 products, round-half-even
  for (int c = t; c < ncls; c += 256) {
    const float inv = (float)(1.0 / (double)(float)sn[c]);
    rc[c] = make_int4((int)rintf((float)sx[c] * inv), (int)rintf((float)sy[c] * inv), (int)rintf((float)sw[c] * inv),
                      (int)rintf((float)sh[c] * inv));
  }
  __syncthreads();
  // ---- too few neighbours, or a small rectangle inside a stronger one
  for (int i = t; i < ncls; i += 256) {
    const int n1 = sn[i];
    bool keep = n1 > threshold;
    if (keep) {
      const int4 r1 = rc[i];
      for (int j = 0; j < ncls; j++) {
        const int n2 = sn[j];
        if (j == i || n2 <= threshold) continue;
        const int4 r2 = rc[j];
        const int dx = (int)rint((double)r2.z * eps), dy = (int)rint((double)r2.w * eps);
        if (r1.x >= r2.x - dx && r1.y >= r2.y - dy && r1.x + r1.z <= r2.x + r2.z + dx && r1.y + r1.w <= r2.y + r2.w + dy &&
            (n2 > max(3, n1) || n1 < 3)) {
          keep = false;
          break;
        }
      }
    }
    cls[i] = keep ? 1 : 0;
    parent[i] = keep ? 1 : 0;
  }
  __syncthreads();
  const int nout = block_exclusive_scan(cls, tmp, ncls);
  if (nout <= out_cap) {
    for (int i = t; i < ncls; i += 256)
      if (parent[i]) { const int4 r = rc[i]; h_boxes[(size_t)f * HC_GROUP_OUT + cls[i]] = GroupBox{r.x, r.y, r.z, r.w}; }
  }
  __threadfence_system();       // this thread's boxes are performed for the host ...
  __syncthreads();              // ... and every thread's of the frame
  if (t == 0) {
    h_nboxes[f] = nout <= out_cap ? (unsigned int)nout : 0xFFFFFFFFu;
    __threadfence_system();
  }
}

int launch_group(hipStream_t st, int nframes, const unsigned int* count, const unsigned long long* cand, unsigned int cap,
                 unsigned int* frame_count, unsigned long long* bins, const GroupLevel* levels, int semantics, int threshold,
                 GroupBox* h_boxes, unsigned int* h_nboxes, int bin_cap, int out_cap) {
  if (nframes <= 0) return 0;
  bin_cap = min(max(bin_cap, 1), HC_GROUP_BIN);
  out_cap = min(max(out_cap, 1), HC_GROUP_OUT);
  hipLaunchKernelGGL(k_group_bin, dim3(64), dim3(256), 0, st, count, cand, cap, frame_count, bins, nframes, bin_cap);
  HC_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_group_frames, dim3(nframes), dim3(256), 0, st, frame_count, bins, levels, semantics, threshold, h_boxes, h_nboxes,
                     bin_cap, out_cap);
  HC_LAUNCH_CHECK();
  return 0;
}

}  // namespace hc
