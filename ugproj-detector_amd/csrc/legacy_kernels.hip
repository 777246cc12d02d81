// legacy_kernels.hip -- T2: OpenCV 2.4.x semantics of the same call, the generation the
// reference literally builds against (src/tracker/associator.cc:3) and the branch its CASCADE_SCALE_IMAGE flag
// selects (src/tracker/detector.cc:30).  Behind an explicit switch (hc_detect_multiscale_cv24); the throughput path
// and every default stay on the OpenCV >= 4 semantics.  The scalar C path of 2.4's haar.cpp is the definition:
//   k_cv24_resize    cvResize(CV_INTER_LINEAR): f32 source coordinate, 11-bit coefficients, >>4 / >>16 / +2 >>2 rounding
//   (integrals)      k_integral_rows / k_integral_cols of kernels.hip: sum int32; the f64 sqsum of upstream holds exact
//                    integers and only window sums (< 2^32) are ever used, so the modulo-2^32 words give the same value
//   k_cv24_cascade   one thread per scanned window: f64 mean / variance, area-normalised f32 rect weights, f32 products
//                    accumulated in f64, f64 stage sums, thresholds scaled by the variance factor, the STAGE TREE of the
//                    cascade file (parent / next / child) -- no x-skip, no variance gate, last row / column not scanned
// Not tuned: one thread per window reading corners from global memory (L2).  -ffp-contract=off as everywhere.
#include <hip/hip_runtime.h>

#include "legacy_cv24.h"

namespace hc {

__global__ __launch_bounds__(256) void k_cv24_resize(const uint8_t* __restrict__ src, size_t frame_stride, size_t row_step, int rows,
                                                     int cols, const Cv24DevLevel* __restrict__ levels,
                                                     const int* __restrict__ tab_ofs, const short* __restrict__ tab_a0,
                                                     const short* __restrict__ tab_a1, uint8_t* __restrict__ pyr, size_t pyr_stride) {
  const Cv24DevLevel L = levels[blockIdx.y];
  src += (size_t)blockIdx.z * frame_stride;                    // one frame of the batch per grid z
  pyr += (size_t)blockIdx.z * pyr_stride;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= L.w * L.h) return;
  const int y = i / L.w, x = i - y * L.w;
  uint8_t out;
  if (L.resize_mode == 0) {                       // same size: copy
    out = src[(size_t)y * row_step + x];
  } else if (L.resize_mode == 2) {                // exactly half in both axes: area kernel
    const uint8_t* p = src + (size_t)(2 * y) * row_step + 2 * x;
    out = (uint8_t)((p[0] + p[1] + p[row_step] + p[row_step + 1] + 2) >> 2);
  } else {
    const int sx = tab_ofs[L.xtab + x], sx1 = min(sx + 1, cols - 1), sy = tab_ofs[L.ytab + y], sy1 = min(sy + 1, rows - 1);
    const int ax0 = tab_a0[L.xtab + x], ax1 = tab_a1[L.xtab + x], b0 = tab_a0[L.ytab + y], b1 = tab_a1[L.ytab + y];
    const uint8_t *r0 = src + (size_t)sy * row_step, *r1 = src + (size_t)sy1 * row_step;
    const int h0 = r0[sx] * ax0 + r0[sx1] * ax1, h1 = r1[sx] * ax0 + r1[sx1] * ax1;
    out = (uint8_t)((((b0 * (h0 >> 4)) >> 16) + ((b1 * (h1 >> 4)) >> 16) + 2) >> 2);
  }
  pyr[L.pyr_ofs + i] = out;
}

__device__ __forceinline__ int cv24_box(const uint32_t* __restrict__ I, int pitch, int x, int y, const int8_t* r) {
  const uint32_t* a = I + (size_t)(y + r[1]) * pitch + x + r[0];
  return (int)(a[0] - a[r[2]] - a[(size_t)r[3] * pitch] + a[(size_t)r[3] * pitch + r[2]]);
}

__global__ __launch_bounds__(256) void k_cv24_cascade(Cv24Args A) {
  const Cv24DevLevel L = A.levels[blockIdx.y];
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= L.nx * L.ny) return;
  const int gy = i / L.nx, gx = i - gy * L.nx;
  const int x = gx * L.ystep, y = gy * L.ystep;
  const int frame = blockIdx.z;
  const uint32_t* S = A.sum + (size_t)frame * A.int_stride + L.int_ofs;
  const uint32_t* Q = A.sq + (size_t)frame * A.int_stride + L.int_ofs;
  const int p = L.int_pitch;
  const int8_t eq[4] = {1, 1, (int8_t)(A.win_w - 2), (int8_t)(A.win_h - 2)};
  const double mean = (double)cv24_box(S, p, x, y, eq) * A.inv_area;
  const double sq = (double)(uint32_t)cv24_box(Q, p, x, y, eq);
  double vnf = sq * A.inv_area - mean * mean;
  vnf = vnf >= 0. ? sqrt(vnf) : 1.;
  int si = 0, passed = 0;
  bool accept = false;
  for (;;) {
    const Cv24Stage st = A.stages[si];
    double stage_sum = 0.0;
    for (int t = 0; t < st.ntrees; t++) {
      const Cv24Tree T = A.trees[st.first_tree + t];
      int idx = 0;
      do {
        const Cv24Node& N = A.nodes[T.first_node + idx];
        const double thr = (double)N.thr * vnf;
        double sum = (double)((float)cv24_box(S, p, x, y, N.rect[0]) * N.w[0]);
        sum += (double)((float)cv24_box(S, p, x, y, N.rect[1]) * N.w[1]);
        if (N.has3) sum += (double)((float)cv24_box(S, p, x, y, N.rect[2]) * N.w[2]);
        idx = sum < thr ? N.left : N.right;
      } while (idx > 0);
      stage_sum += (double)A.leaves[T.first_leaf - idx];
    }
    if (!(stage_sum < (double)st.thr)) {
      ++passed;
      if (st.child < 0) { accept = true; break; }
      si = st.child;
    } else {
      while (si >= 0 && A.stages[si].next < 0) si = A.stages[si].parent;
      if (si < 0) break;
      si = A.stages[si].next;
    }
  }
  if (A.taps) A.taps[(size_t)frame * A.grid_stride + L.grid_ofs + i] = (int8_t)(min(passed, 63) | (accept ? 64 : 0));
  if (accept) {
    const unsigned int slot = atomicAdd(A.cand_count, 1u);
    if (slot < A.cand_cap)
      A.cand[slot] = ((unsigned long long)frame << 40) | ((unsigned long long)blockIdx.y << 32) | ((unsigned long long)gy << 16) | (unsigned long long)gx;
  }
}

int launch_cv24_resize(hipStream_t st, int nframes, int nlevels, int max_pixels, const uint8_t* src, size_t frame_stride, size_t row_step,
                       int rows, int cols, const Cv24DevLevel* levels, const int* tab_ofs, const short* tab_a0, const short* tab_a1,
                       uint8_t* pyr, size_t pyr_stride) {
  if (nlevels == 0 || nframes == 0) return 0;
  hipLaunchKernelGGL(k_cv24_resize, dim3((max_pixels + 255) / 256, nlevels, nframes), dim3(256), 0, st, src, frame_stride, row_step, rows,
                     cols, levels, tab_ofs, tab_a0, tab_a1, pyr, pyr_stride);
  return (int)hipGetLastError();
}

int launch_cv24_cascade(hipStream_t st, int nframes, int nlevels, int max_positions, const Cv24Args& args) {
  if (nlevels == 0 || max_positions == 0 || nframes == 0) return 0;
  hipLaunchKernelGGL(k_cv24_cascade, dim3((max_positions + 255) / 256, nlevels, nframes), dim3(256), 0, st, args);
  return (int)hipGetLastError();
}

}  // namespace hc
