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

// ---- a real stage tree (haarcascade_frontalface_alt_tree: two branches below stage 4, the second tried when the first
// fails) on the same LDS tiles.  Every window carries the index of the stage it is at.  A passing window goes to the
// stage's first child, a failing one to the next sibling of the nearest ancestor-or-self that has one: in PREORDER of the
// stage tree (a stage, its child subtree, then its next sibling -- tree_order, from the host) both targets come later, so
// ONE sweep over that order visits every window at every stage of its path: the windows at stage s are gathered into a
// list, evaluated lane = window with the 2.4 node arithmetic of eval_stage, and sent on to tree_pass[s] or tree_fail[s].
// LDS: [ vnf f64 x TILE_WIN ][ list u16 x TILE_WIN ][ stage u8 x TILE_WIN ][ passed u8 x TILE_WIN ][ counter ][ tile ]
constexpr int TREE_ACCEPT = 254, TREE_REJECT = 255;
template <int NW>
__global__ __launch_bounds__(NW * 64, 8) void k_cv24_tree(CascadeArgs A) {
  constexpr int ROWS = 16, TILE_WIN = ROWS * TILE_POS, NTHREADS = NW * 64;
  extern __shared__ int32_t smem[];
  double* vnf_t = (double*)smem;
  uint16_t* list = (uint16_t*)(vnf_t + TILE_WIN);
  uint8_t* at_stage = (uint8_t*)(list + TILE_WIN);
  uint8_t* passed = at_stage + TILE_WIN;
  int* counter = (int*)(passed + TILE_WIN);
  int32_t* tile = (int32_t*)(counter + 2);
  const WorkItem it = load_item(A.bands, blockIdx.y);
  const HC_CONST DevLevel& L = as_const(A.levels)[it.level];
  const HC_CONST DevStage* stages = as_const(A.stages);
  const HC_CONST DevTree* trees = as_const(A.trees);
  const HC_CONST DevNode* nodes = as_const(A.nodes);
  const HC_CONST DevStageKinds* kinds = as_const(A.kinds);
  const int frame = blockIdx.x;
  const int ys = L.ystep;
  const TileGeom G = A.g;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
  const int gy0 = it.a, y0 = gy0 * ys, nx = L.nx, ny = L.ny;
  const uint32_t* __restrict__ I = A.sum + (size_t)frame * A.int_stride + L.int_ofs;
  const size_t grid_base = (size_t)frame * A.grid_stride + L.grid_ofs;
  const double* __restrict__ vnf_g = A.vnf64 + grid_base;
  const unsigned long long key_hi = ((unsigned long long)frame << 40) | ((unsigned long long)it.level << 32);
  const unsigned long long lt_mask = lane ? ((1ull << lane) - 1ull) : 0ull;
  for (int gx0 = 0; gx0 < nx; gx0 += TILE_POS) {
    __syncthreads();
    fill_tile<NW>(tile, I, L, G, ys, gx0 * ys, y0, wave, lane);
    for (int pos = threadIdx.x; pos < TILE_WIN; pos += NTHREADS) {
      const int gy = gy0 + (pos >> 6), gx = gx0 + (pos & 63);
      const bool valid = gy < ny && gx < nx;
      vnf_t[pos] = valid ? vnf_g[(size_t)gy * nx + gx] : 0.0;
      at_stage[pos] = valid ? 0 : (uint8_t)TREE_REJECT;
      passed[pos] = 0;
    }
    for (int k = 0; k < A.nstages; k++) {
      const int s = as_const(A.tree_order)[k];
      if (threadIdx.x == 0) counter[k & 1] = 0;
      __syncthreads();
      // the windows that are at stage s -> list (wave-aggregated append)
      for (int p0 = wave * 64; p0 < TILE_WIN; p0 += NTHREADS) {
        const int pos = p0 + lane;
        const bool mine = at_stage[pos] == s;
        const unsigned long long m = __ballot(mine);
        if (m) {
          int off = 0;
          if (lane == 0) off = atomicAdd(&counter[k & 1], __popcll(m));
          off = __builtin_amdgcn_readfirstlane(off);
          if (mine) list[off + __popcll(m & lt_mask)] = (uint16_t)pos;
        }
      }
      __syncthreads();
      const int n = counter[k & 1];
      if (n == 0) continue;                                        // uniform
      const int to_pass = as_const(A.tree_pass)[s], to_fail = as_const(A.tree_fail)[s];
      for (int i0 = wave * 64; i0 < n; i0 += NTHREADS) {
        const int i = i0 + lane;
        const bool active = i < n;
        const int pos = active ? list[i] : list[0];
        const int base = (pos >> 6) * ys * G.pitch + (pos & 63);
        const bool ok = eval_stage<false, NoSpec, 0>(tile, base, vnf_t[pos], s, stages, kinds, nodes, trees, A.leaves);
        if (active) {
          const int to = ok ? to_pass : to_fail;
          at_stage[pos] = (uint8_t)to;
          if (ok) passed[pos]++;
          if (to == TREE_ACCEPT) {
            const int gy = gy0 + (pos >> 6), gx = gx0 + (pos & 63);
            if (A.cand_cap) {
              const unsigned int slot = atomicAdd(A.cand_count, 1u);
              if (slot < A.cand_cap) A.cand[slot] = key_hi | ((unsigned long long)gy << 16) | (unsigned long long)gx;
            }
          }
        }
      }
    }
    if (A.taps) {
      __syncthreads();
      for (int pos = threadIdx.x; pos < TILE_WIN; pos += NTHREADS) {
        const int gy = gy0 + (pos >> 6), gx = gx0 + (pos & 63);
        if (gy < ny && gx < nx)
          A.taps[grid_base + (size_t)gy * nx + gx] = (int8_t)(min((int)passed[pos], 63) | (at_stage[pos] == TREE_ACCEPT ? 64 : 0));
      }
    }
  }
}
}  // namespace casc_cv24

// stage tree of a stump cascade on the tiles (k_cv24_tree); lds_bytes: tile words * 4 + TILE_WIN * 12 + 8
int launch_cv24_tree(hipStream_t st, int nframes, int nwaves, size_t lds_bytes, const CascadeArgs& args) {
  if (args.nbands == 0 || nframes == 0) return 0;
  const dim3 grid(nframes, args.nbands), block(nwaves * 64);
  if (nwaves == 4) hipLaunchKernelGGL((casc_cv24::k_cv24_tree<4>), grid, block, lds_bytes, st, args);
  else if (nwaves == 8) hipLaunchKernelGGL((casc_cv24::k_cv24_tree<8>), grid, block, lds_bytes, st, args);
  else return (int)hipErrorInvalidValue;
  return (int)hipGetLastError();
}

// The production band shapes only: 16 grid rows, 4 wavefronts at scan step 1 and 8 at scan step 2, whole bands.
int launch_cascade_cv24(hipStream_t st, int nframes, bool tree, int rows, int nwaves, size_t lds_bytes, const CascadeArgs& args, int* compiled) {
  if (compiled) *compiled = 0;
  if (args.nbands == 0 || nframes == 0) return 0;
  if (rows != 16 || (nwaves != 4 && nwaves != 8)) return (int)hipErrorInvalidValue;
  // compiled early stages with the 2.4 node arithmetic (csrc/spec/spec_*24.hip; args.spec from spec_lookup(.., cv24 = true))
  if (const SpecUnit* unit = args.spec > 0 ? spec_unit(args.spec) : nullptr) {
    if (unit->cv24 && unit->tree == tree) {
      const int rc = unit->launch(st, nframes, false, rows, nwaves, lds_bytes, args);
      if (rc >= 0) { if (compiled) *compiled = unit->nstages; return rc; }
    }
  }
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
