// group_kernels.hip -- rectangle grouping of a batch's raw candidates ON the device:
//   k_group_bin      every raw candidate key into its frame's bin (one atomic per key)
//   k_group_frames   one 256-thread workgroup per frame: keys sorted to upstream's single-thread order (bitonic, LDS),
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

__global__ __launch_bounds__(256) void k_group_bin(const unsigned int* __restrict__ count, const unsigned long long* __restrict__ cand,
                                                   unsigned int cap, unsigned int* __restrict__ frame_count,
                                                   unsigned long long* __restrict__ bins, int nframes, int bin_cap) {
  const unsigned int n = min(*count, cap);
  for (unsigned int i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u) {
    const unsigned long long k = cand[i];
    const unsigned int f = (unsigned int)(k >> 40);
    if (f >= (unsigned int)nframes) continue;
    const unsigned int pos = atomicAdd(&frame_count[f], 1u);
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
  __shared__ unsigned long long keys[HC_GROUP_BIN];
  __shared__ int rx[HC_GROUP_BIN], ry[HC_GROUP_BIN], rw[HC_GROUP_BIN], rh[HC_GROUP_BIN];
  __shared__ int parent[HC_GROUP_BIN], cls[HC_GROUP_BIN], tmp[256];
  __shared__ int sx[HC_GROUP_BIN], sy[HC_GROUP_BIN], sw[HC_GROUP_BIN], sh[HC_GROUP_BIN], sn[HC_GROUP_BIN];
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
  int npad = 1;
  while (npad < n) npad <<= 1;
  for (int i = t; i < npad; i += 256) keys[i] = i < n ? bins[(size_t)f * HC_GROUP_BIN + i] : ~0ull;
  __syncthreads();
  // ---- upstream's order: frame -> scale -> y -> x == ascending key
  for (int k = 2; k <= npad; k <<= 1)
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = t; i < npad; i += 256) {
        const int p = i ^ j;
        if (p > i) {
          const unsigned long long a = keys[i], b = keys[p];
          const bool up = (i & k) == 0;
          if ((a > b) == up) { keys[i] = b; keys[p] = a; }
        }
      }
      __syncthreads();
    }
  // ---- decode
  for (int i = t; i < n; i += 256) {
    const unsigned long long k = keys[i];
    const GroupLevel L = levels[(k >> 32) & 0xFF];
    const int gy = (int)((k >> 16) & 0xFFFF), gx = (int)(k & 0xFFFF);
    if (semantics == 1) {     // 2.4 (B.9): cvRound(x * factor), f64
      rx[i] = (int)rint((double)(gx * L.ystep) * L.factor);
      ry[i] = (int)rint((double)(gy * L.ystep) * L.factor);
    } else {                  // Rect(cvRound(x * scale), cvRound(y * scale), winSize): int -> f32 times the f32 scale
      rx[i] = (int)rintf((float)(gx * L.ystep) * L.scale);
      ry[i] = (int)rintf((float)(gy * L.ystep) * L.scale);
    }
    rw[i] = L.win_w; rh[i] = L.win_h;
    parent[i] = i;
    sx[i] = 0; sy[i] = 0; sw[i] = 0; sh[i] = 0; sn[i] = 0;
  }
  __syncthreads();
  // ---- cv::partition: components of the SimilarRects graph.  A root is only ever hooked under a SMALLER root, so
  // parent[v] <= v throughout and a component's final root is its smallest member, its first in upstream's order.
  const double eps = 0.2;
  for (int i = t; i < n; i += 256) {
    const int px = rx[i], py = ry[i], pw = rw[i], ph = rh[i];
    for (int j = i + 1; j < n; j++) {
      const int qx = rx[j], qy = ry[j], qw = rw[j], qh = rh[j];
      const double delta = eps * (double)(min(pw, qw) + min(ph, qh)) * 0.5;
      const bool sim = (double)abs(px - qx) <= delta && (double)abs(py - qy) <= delta &&
                       (double)abs(px + pw - qx - qw) <= delta && (double)abs(py + ph - qy - qh) <= delta;
      if (!sim) continue;
      int a = i, b = j;
      for (;;) {
        a = find_root(parent, a);
        b = find_root(parent, b);
        if (a == b) break;
        const int hi = max(a, b), lo = min(a, b);
        if (atomicCAS(&parent[hi], hi, lo) == hi) break;
        a = hi; b = lo;      // somebody hooked `hi` meanwhile: walk on from there
      }
    }
  }
  __syncthreads();
  for (int i = t; i < n; i += 256) cls[i] = find_root(parent, i) == i ? 1 : 0;
  __syncthreads();
  const int ncls = block_exclusive_scan(cls, tmp, n);      // cls[root] = class id of that root's class
  // ---- class sums
  for (int i = t; i < n; i += 256) {
    const int c = cls[find_root(parent, i)];
    atomicAdd(&sx[c], rx[i]); atomicAdd(&sy[c], ry[i]); atomicAdd(&sw[c], rw[i]); atomicAdd(&sh[c], rh[i]); atomicAdd(&sn[c], 1);
  }
  __syncthreads();
  // ---- class means over the rect arrays (dead now): f32 reciprocal, f32 products, round-half-even
  for (int c = t; c < ncls; c += 256) {
    const float inv = (float)(1.0 / (double)(float)sn[c]);
    rx[c] = (int)rintf((float)sx[c] * inv); ry[c] = (int)rintf((float)sy[c] * inv);
    rw[c] = (int)rintf((float)sw[c] * inv); rh[c] = (int)rintf((float)sh[c] * inv);
  }
  __syncthreads();
  // ---- too few neighbours, or a small rectangle inside a stronger one
  for (int i = t; i < ncls; i += 256) {
    const int n1 = sn[i];
    bool keep = n1 > threshold;
    if (keep) {
      const int x1 = rx[i], y1 = ry[i], w1 = rw[i], h1 = rh[i];
      for (int j = 0; j < ncls; j++) {
        const int n2 = sn[j];
        if (j == i || n2 <= threshold) continue;
        const int dx = (int)rint((double)rw[j] * eps), dy = (int)rint((double)rh[j] * eps);
        if (x1 >= rx[j] - dx && y1 >= ry[j] - dy && x1 + w1 <= rx[j] + rw[j] + dx && y1 + h1 <= ry[j] + rh[j] + dy &&
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
      if (parent[i]) h_boxes[(size_t)f * HC_GROUP_OUT + cls[i]] = GroupBox{rx[i], ry[i], rw[i], rh[i]};
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
