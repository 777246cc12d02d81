// cv24_cascade.hip -- the LDS-tiled band kernel (cascade_body.inc) built for the OpenCV-2.4 evaluation of the call
// ("T2", SURVEY.md Appendix B: the generation the reference compiles against, src/tracker/associator.cc:3, and the branch
// its CASCADE_SCALE_IMAGE flag selects, src/tracker/detector.cc:30), plus that generation's f64 variance factor.
//
// Same kernel text as the OpenCV >= 4 builds in kernels.hip -- tile fill, row-form first stages, bank-ordered survivor
// lists, cooperative stages, sliced finish -- with the node arithmetic of 2.4's scalar C path (HC_CV24 in cascade_body.inc):
// area-normalised f32 rect weights, per-rect f32 products accumulated in f64, thresholds scaled by the f64 variance
// factor, no variance gate and no x-skip (every grid position is evaluated; with no stage-0 failure ever reported the
// closed-form skip chain marks every position visited).  Cascades whose stages form a linear chain (alt, alt2, default)
// run here; a real stage tree (alt_tree) keeps the window-at-a-time kernel of legacy_kernels.hip.
#include <hip/hip_runtime.h>
#include <cstdio>

#include "device_types.h"
#include "kernels.h"
#include "device_common.h"
#include "legacy_cv24.h"

namespace hc {

namespace casc_cv24 {
#define HC_INTW 0
#define HC_CV24 1
#define HC_BODY_NO_LAUNCHERS
#include "cascade_body.inc"
#undef HC_INTW
}  // namespace casc_cv24

// The production band shapes only: 16 grid rows, 4 wavefronts at scan step 1 and 8 at scan step 2, whole bands.
int launch_cascade_cv24(hipStream_t st, int nframes, bool tree, int rows, int nwaves, size_t lds_bytes, const CascadeArgs& args) {
  if (args.nbands == 0 || nframes == 0) return 0;
  if (rows != 16 || (nwaves != 4 && nwaves != 8)) return (int)hipErrorInvalidValue;
  const dim3 grid(nframes, args.nbands), block(nwaves * 64);
#define HC_GO(TREE, NW, TAPS) hipLaunchKernelGGL((casc_cv24::k_cascade<TREE, 16, NW, false, TAPS>), grid, block, lds_bytes, st, args)
  if (tree) {
    if (nwaves == 4) { if (args.taps) HC_GO(true, 4, true); else HC_GO(true, 4, false); }
    else { if (args.taps) HC_GO(true, 8, true); else HC_GO(true, 8, false); }
  } else {
    if (nwaves == 4) { if (args.taps) HC_GO(false, 4, true); else HC_GO(false, 4, false); }
    else { if (args.taps) HC_GO(false, 8, true); else HC_GO(false, 8, false); }
  }
#undef HC_GO
  return (int)hipGetLastError();
}

// B.5: variance factor of every scanned window, f64 all the way (no low-variance gate):
//   mean = sum / area, var = sqsum / area - mean^2, factor = var >= 0 ? sqrt(var) : 1
__global__ __launch_bounds__(WG_THREADS) void k_cv24_variance(const DevLevel* __restrict__ levels, const WorkItem* __restrict__ items,
                                                              const uint32_t* __restrict__ sum, const uint32_t* __restrict__ sq,
                                                              size_t int_stride, int norm_w, int norm_h, double inv_area,
                                                              double* __restrict__ vnf, size_t grid_stride) {
  const WorkItem it = load_item(items, blockIdx.x);
  const HC_CONST DevLevel& L = as_const(levels)[it.level];
  const int frame = blockIdx.y;
  const int i = it.a + threadIdx.x;
  if (i >= L.nx * L.ny) return;
  const int gy = i / L.nx, gx = i - gy * L.nx;
  const int x = gx * L.ystep + 1, y = gy * L.ystep + 1;       // equ_rect = (1, 1, win_w - 2, win_h - 2)
  const size_t base = (size_t)frame * int_stride + L.int_ofs;
  const size_t p0 = base + (size_t)y * L.int_pitch + x, p2 = p0 + (size_t)norm_h * L.int_pitch;
  const int valsum = (int)(sum[p0] - sum[p0 + norm_w] - sum[p2] + sum[p2 + norm_w]);
  const uint32_t valsq = sq[p0] - sq[p0 + norm_w] - sq[p2] + sq[p2 + norm_w];
  const double mean = (double)valsum * inv_area;
  double v = (double)valsq * inv_area - mean * mean;
  v = v >= 0. ? sqrt(v) : 1.;
  vnf[(size_t)frame * grid_stride + L.grid_ofs + (size_t)gy * L.nx + gx] = v;
}

int launch_cv24_variance(hipStream_t st, int nframes, int nitems, const DevLevel* levels, const WorkItem* items, const uint32_t* sum,
                         const uint32_t* sq, size_t int_stride, int norm_w, int norm_h, double inv_area, double* vnf, size_t grid_stride) {
  if (nitems == 0 || nframes == 0) return 0;
  hipLaunchKernelGGL(k_cv24_variance, dim3(nitems, nframes), dim3(WG_THREADS), 0, st, levels, items, sum, sq, int_stride, norm_w, norm_h,
                     inv_area, vnf, grid_stride);
  return (int)hipGetLastError();
}

}  // namespace hc
