// kernels.hip -- hand-written gfx950 (CDNA4, wave64) kernels of the Haar-cascade path.
//
// The path replaces what cv::CascadeClassifier::detectMultiScale does per frame for the
// reference's call at src/tracker/detector.cc:25-31 (semantics: SURVEY.md Appendix A):
//   k_resize         A.4  exact bilinear pyramid, every level from the original frame
//   k_integral_rows  A.5  row prefix sums (sum, sqsum) -- wave64 __shfl_up scan + carry
//   k_integral_cols  A.5  column prefix sums, coalesced across lanes
//   k_variance       A.6a setWindow: variance normalisation factor + evaluable gate
//   k_cascade        A.6b/A.7/A.8/A.9 stage evaluation from LDS-staged integral tiles,
//                    x-skip chain resolved in closed form with wave ballots
// Integer work is modulo 2^32; float work is IEEE single ops in the upstream order and
// the file is compiled with -ffp-contract=off (upstream x86-64 baseline has no FMA).
// No MFMA: there is no dense contraction on this path.

#include <hip/hip_runtime.h>

#include "device_types.h"
#include "kernels.h"

namespace hc {

// Read-only tables (levels, work lists, cascade nodes/stages) are addressed through the
// constant address space: with a wave-uniform index the compiler then emits scalar
// s_load_* (SGPR operands, scalar cache) instead of per-lane global loads.  The memory is
// ordinary hipMalloc memory that no kernel writes.
#define HC_CONST __attribute__((address_space(4)))
template <class T>
__device__ __forceinline__ const HC_CONST T* as_const(const T* p) {
  return (const HC_CONST T*)p;
}

__device__ __forceinline__ WorkItem load_item(const WorkItem* items, int i) {
  const HC_CONST WorkItem* c = as_const(items) + i;
  WorkItem w;
  w.level = c->level; w.a = c->a; w.b = c->b;
  return w;
}

// ------------------------------------------------------------------------------------------
// A.4  pyramid: dst(level, dy, dx) from the ORIGINAL frame, 8.8 x 8.8 fixed point
// block = 64 x 4 destination pixels
__global__ __launch_bounds__(WG_THREADS) void k_resize(
    const uint8_t* __restrict__ frames, size_t frame_stride, size_t row_step, int rows, int cols,
    const DevLevel* __restrict__ levels, const uint32_t* __restrict__ xtab,
    const uint32_t* __restrict__ ytab, const WorkItem* __restrict__ items,
    uint8_t* __restrict__ pyr, size_t pyr_stride) {
  const WorkItem it = load_item(items, blockIdx.x);
  const HC_CONST DevLevel& L = as_const(levels)[it.level];
  const int dx = it.a + (threadIdx.x & 63);
  const int dy = it.b + (threadIdx.x >> 6);
  if (dx >= L.w || dy >= L.h) return;
  const uint32_t xt = xtab[L.xtab_ofs + dx], yt = ytab[L.ytab_ofs + dy];
  const int ox = xt >> 9, oy = yt >> 9;
  const uint32_t cx1 = xt & 511u, cy1 = yt & 511u, cx0 = 256u - cx1, cy0 = 256u - cy1;
  const int ox1 = min(ox + 1, cols - 1), oy1 = min(oy + 1, rows - 1);
  const uint8_t* src = frames + (size_t)blockIdx.y * frame_stride;
  const uint8_t* r0 = src + (size_t)oy * row_step;
  const uint8_t* r1 = src + (size_t)oy1 * row_step;
  const uint32_t h0 = cx0 * r0[ox] + cx1 * r0[ox1];   // 8.8, <= 65280
  const uint32_t h1 = cx0 * r1[ox] + cx1 * r1[ox1];
  const uint32_t v = h0 * cy0 + h1 * cy1;             // 16.16
  pyr[(size_t)blockIdx.y * pyr_stride + L.pyr_ofs + (size_t)dy * L.w + dx] = (uint8_t)min((v + 32768u) >> 16, 255u);
}

// ------------------------------------------------------------------------------------------
// A.5 pass 1: one wavefront per pyramid row; inclusive scan over 64 lanes by __shfl_up,
// carry in a register across 64-pixel chunks.  Writes I[y+1][x+1] = sum_{j<=x} p[y][j].
__global__ __launch_bounds__(WG_THREADS) void k_integral_rows(
    const uint8_t* __restrict__ pyr, size_t pyr_stride, const DevLevel* __restrict__ levels,
    const WorkItem* __restrict__ items, uint32_t* __restrict__ sum, uint32_t* __restrict__ sq,
    size_t int_stride) {
  const WorkItem it = load_item(items, blockIdx.x);
  const HC_CONST DevLevel& L = as_const(levels)[it.level];
  const int lane = threadIdx.x & 63;
  const int row = it.a + (threadIdx.x >> 6);
  if (row >= L.h) return;                       // whole wavefront leaves together
  const int w = L.w;
  const uint8_t* src = pyr + (size_t)blockIdx.y * pyr_stride + L.pyr_ofs + (size_t)row * w;
  const size_t o = (size_t)blockIdx.y * int_stride + L.int_ofs + (size_t)(row + 1) * L.int_pitch + 1;
  uint32_t* S = sum + o;
  uint32_t* Q = sq + o;
  uint32_t carry_s = 0, carry_q = 0;
  for (int x0 = 0; x0 < w; x0 += 64) {
    const int x = x0 + lane;
    const uint32_t p = x < w ? src[x] : 0u;
    uint32_t s = p, q = p * p;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const uint32_t ts = __shfl_up(s, d, 64), tq = __shfl_up(q, d, 64);
      if (lane >= d) { s += ts; q += tq; }
    }
    s += carry_s;
    q += carry_q;
    if (x < w) { S[x] = s; Q[x] = q; }
    carry_s = __shfl(s, 63, 64);
    carry_q = __shfl(q, 63, 64);
  }
}

// A.5 pass 2: one thread per column, running sum down the rows (lanes = adjacent columns,
// so every load/store is a coalesced row segment).  Loads are batched ahead of the
// dependent adds to keep 2 x 8 requests in flight per lane.
__global__ __launch_bounds__(WG_THREADS) void k_integral_cols(
    const DevLevel* __restrict__ levels, const WorkItem* __restrict__ items,
    uint32_t* __restrict__ sum, uint32_t* __restrict__ sq, size_t int_stride) {
  const WorkItem it = load_item(items, blockIdx.x);
  const HC_CONST DevLevel& L = as_const(levels)[it.level];
  const int x = it.a + threadIdx.x;
  if (x >= L.w) return;
  const int pitch = L.int_pitch, h = L.h;
  const size_t o = (size_t)blockIdx.y * int_stride + L.int_ofs + (size_t)pitch + x + 1;
  uint32_t* S = sum + o;
  uint32_t* Q = sq + o;
  uint32_t as = 0, aq = 0;
  int y = 0;
  for (; y + 8 <= h; y += 8) {
    uint32_t vs[8], vq[8];
#pragma unroll
    for (int k = 0; k < 8; k++) { vs[k] = S[(size_t)(y + k) * pitch]; vq[k] = Q[(size_t)(y + k) * pitch]; }
#pragma unroll
    for (int k = 0; k < 8; k++) {
      as += vs[k]; aq += vq[k];
      S[(size_t)(y + k) * pitch] = as; Q[(size_t)(y + k) * pitch] = aq;
    }
  }
  for (; y < h; y++) {
    as += S[(size_t)y * pitch]; aq += Q[(size_t)y * pitch];
    S[(size_t)y * pitch] = as; Q[(size_t)y * pitch] = aq;
  }
}

// ------------------------------------------------------------------------------------------
// A.6a setWindow for every scanned grid position: vnf = (float)(1/sqrt(area*sqsum - sum^2)),
// 0 marks "not evaluable" (nf <= 0 or area*vnf >= 0.1).
__global__ __launch_bounds__(WG_THREADS) void k_variance(
    const DevLevel* __restrict__ levels, const WorkItem* __restrict__ items,
    const uint32_t* __restrict__ sum, const uint32_t* __restrict__ sq, size_t int_stride,
    int norm_w, int norm_h, float* __restrict__ vnf, size_t grid_stride) {
  const WorkItem it = load_item(items, blockIdx.x);
  const HC_CONST DevLevel& L = as_const(levels)[it.level];
  const int i = it.a + threadIdx.x;
  if (i >= L.nx * L.ny) return;
  const int gy = i / L.nx, gx = i - gy * L.nx;
  const int x = gx * L.ystep + 1, y = gy * L.ystep + 1;       // normrect = (1, 1, win_w-2, win_h-2)
  const size_t base = (size_t)blockIdx.y * int_stride + L.int_ofs;
  const size_t p0 = base + (size_t)y * L.int_pitch + x, p2 = p0 + (size_t)norm_h * L.int_pitch;
  const int valsum = (int)(sum[p0] - sum[p0 + norm_w] - sum[p2] + sum[p2 + norm_w]);
  const uint32_t valsq = sq[p0] - sq[p0 + norm_w] - sq[p2] + sq[p2 + norm_w];
  const double area = (double)(norm_w * norm_h);
  double nf = area * (double)valsq - (double)valsum * (double)valsum;
  float out = 0.f;
  if (nf > 0.) {
    nf = sqrt(nf);
    const float v = (float)(1. / nf);
    if (area * (double)v < 1e-1) out = v;
  }
  vnf[(size_t)blockIdx.y * grid_stride + L.grid_ofs + (size_t)gy * L.nx + gx] = out;
}

// ------------------------------------------------------------------------------------------
// Cascade evaluation.
//
// Workgroup = 4 wavefronts = one band of BAND_ROWS grid rows of one level of one frame.
// It walks the band left to right in steps of TILE_POS positions; per step it stages the
// (BAND_ROWS-1)*ystep+win_h+1 by (TILE_POS-1)*ystep+win_w+1 window of the `sum` integral
// in LDS (coalesced row loads; for ystep 2 the even and odd columns go to separate planes so
// that lane k's corner reads are stride-1 in LDS).  Wave w owns grid rows w and w+4 of the
// band for the whole walk, so the x-skip carry (A.7) stays in its registers.
//
// Rect corner offsets are pre-resolved on the host into LDS word offsets (DevNode::off) and
// are wave-uniform -> scalar loads; a lane's corner address is base + off.
__device__ __forceinline__ float feature_value(const int32_t* tile, int base, const HC_CONST DevNode& n, float vnf) {
  const int s0 = tile[base + n.off[0]] - tile[base + n.off[1]] - tile[base + n.off[2]] + tile[base + n.off[3]];
  const int s1 = tile[base + n.off[4]] - tile[base + n.off[5]] - tile[base + n.off[6]] + tile[base + n.off[7]];
  float ret = n.w[0] * (float)s0 + n.w[1] * (float)s1;
  if (n.w[2] != 0.0f) {   // wave-uniform branch
    const int s2 = tile[base + n.off[8]] - tile[base + n.off[9]] - tile[base + n.off[10]] + tile[base + n.off[11]];
    ret += n.w[2] * (float)s2;
  }
  return ret * vnf;
}

template <bool TREE>
__device__ __forceinline__ bool stage_passes(const int32_t* tile, int base, float vnf, const HC_CONST DevStage& st,
                                             const HC_CONST DevNode* nodes, const HC_CONST DevTree* trees,
                                             const float* __restrict__ leaves) {
  double acc = 0.0;
  if (!TREE) {
    for (int t = 0; t < st.count; t++) {                      // A.8 predictOrderedStump
      const HC_CONST DevNode& n = nodes[st.first + t];
      const float v = feature_value(tile, base, n, vnf);
      acc += (double)(v < n.thr ? __uint_as_float(n.a) : __uint_as_float(n.b));
    }
  } else {
    for (int t = 0; t < st.count; t++) {                      // A.9 predictOrdered
      const HC_CONST DevTree& T = trees[st.first + t];
      int cur = 0;                                            // children always follow their parent
      for (int k = 0; k < T.nnodes; k++) {
        const HC_CONST DevNode& n = nodes[T.first_node + k];
        const float v = feature_value(tile, base, n, vnf);
        const int nxt = v < n.thr ? (int)n.a : (int)n.b;
        cur = (cur == k) ? nxt : cur;
      }
      acc += (double)leaves[T.first_leaf - cur];
    }
  }
  return !(acc < (double)st.thr);
}

template <bool TREE>
__global__ __launch_bounds__(WG_THREADS) void k_cascade(CascadeArgs A) {
  extern __shared__ int32_t tile[];
  // XCD-aware order: workgroups b, b+8, b+16.. share an XCD/L2 -> give them adjacent bands,
  // whose staged rows overlap by win_h.
  const int nb8 = (A.nbands + 7) >> 3;
  const int band = (blockIdx.x & 7) * nb8 + (blockIdx.x >> 3);
  if (band >= A.nbands) return;
  const WorkItem it = load_item(A.bands, band);
  const HC_CONST DevLevel& L = as_const(A.levels)[it.level];
  const HC_CONST DevStage* stages = as_const(A.stages);
  const HC_CONST DevTree* trees = as_const(A.trees);
  const int frame = blockIdx.y;
  const int ys = L.ystep;
  const TileGeom G = ys == 1 ? A.g1 : A.g2;
  const HC_CONST DevNode* nodes = as_const(ys == 1 ? A.nodes1 : A.nodes2);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int gy0 = it.a;
  const int y0 = gy0 * ys;
  const int nx = L.nx, ny = L.ny;
  const uint32_t* __restrict__ I = A.sum + (size_t)frame * A.int_stride + L.int_ofs;
  const size_t grid_base = (size_t)frame * A.grid_stride + L.grid_ofs;
  bool carry[2] = {true, true};   // is position 0 of the next tile visited (per owned row)

  for (int gx0 = 0; gx0 < nx; gx0 += TILE_POS) {
    const int x0 = gx0 * ys;
    __syncthreads();              // everybody is done with the previous tile
    for (int r = wave; r < G.rows; r += WG_THREADS / 64) {
      const int sy = y0 + r;
      const uint32_t* srow = I + (size_t)min(sy, L.h) * L.int_pitch;
      int32_t* drow = tile + r * G.pitch;
      for (int c = lane; c < G.cols; c += 64) {
        const int sx = x0 + c;
        const uint32_t v = (sy <= L.h && sx <= L.w) ? srow[sx] : 0u;
        const int d = ys == 1 ? c : ((c & 1) * G.half + (c >> 1));
        drow[d] = (int32_t)v;
      }
    }
    __syncthreads();

#pragma unroll
    for (int j = 0; j < 2; j++) {
      const int rr = wave + 4 * j;
      const int gy = gy0 + rr, gx = gx0 + lane;
      if (gy >= ny) continue;                                   // wave-uniform
      const bool valid = gx < nx;
      const float vnf = valid ? A.vnf[grid_base + (size_t)gy * nx + gx] : 0.f;
      const bool evaluable = vnf > 0.f;
      const int base = rr * ys * G.pitch + lane;

      // stage 0 on every position of the row segment
      const bool pass0 = stage_passes<TREE>(tile, base, vnf, stages[0], nodes, trees, A.leaves);
      const bool f = evaluable && !pass0;                      // "fails stage 0 if evaluated"

      // A.7 in closed form: position k is visited iff its distance to the nearest anchor
      // (k == row start, or f[k-1] == 0) at or left of it is even.
      const unsigned long long fm = __ballot(f);
      const unsigned long long below = lane ? (~fm & ((1ull << lane) - 1ull)) : 0ull;
      const int anchor = below ? 64 - __clzll((long long)below) : (carry[j] ? 0 : 1);
      const bool visited = ((lane - anchor) & 1) == 0;
      const unsigned long long vm = __ballot(visited);
      carry[j] = !(((vm & fm) >> 63) & 1ull);

      bool alive = valid && visited && evaluable && pass0;
      int passed = pass0 ? 1 : 0;
      for (int s = 1; s < A.nstages; s++) {
        if (!__any(alive)) break;
        const bool ok = stage_passes<TREE>(tile, base, vnf, stages[s], nodes, trees, A.leaves);
        alive = alive && ok;
        passed += alive ? 1 : 0;
      }
      if (alive) {                                              // A.6c: raw candidate
        const unsigned int slot = atomicAdd(A.cand_count, 1u);
        if (slot < A.cand_cap)
          A.cand[slot] = ((unsigned long long)frame << 40) | ((unsigned long long)it.level << 32) |
                         ((unsigned long long)gy << 16) | (unsigned long long)gx;
      }
      if (A.taps && valid && visited)
        A.taps[grid_base + (size_t)gy * nx + gx] = (int8_t)(evaluable ? passed : -1);
    }
  }
}

// ------------------------------------------------------------------------------------------
// launchers
#define HC_LAUNCH_CHECK() do { hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) return (int)e_; } while (0)

int launch_resize(hipStream_t st, int nframes, int nitems, const uint8_t* frames, size_t frame_stride,
                  size_t row_step, int rows, int cols, const DevLevel* levels, const uint32_t* xtab,
                  const uint32_t* ytab, const WorkItem* items, uint8_t* pyr, size_t pyr_stride) {
  if (nitems == 0 || nframes == 0) return 0;
  hipLaunchKernelGGL(k_resize, dim3(nitems, nframes), dim3(WG_THREADS), 0, st, frames, frame_stride, row_step,
                     rows, cols, levels, xtab, ytab, items, pyr, pyr_stride);
  HC_LAUNCH_CHECK();
  return 0;
}

int launch_integral_rows(hipStream_t st, int nframes, int nitems, const uint8_t* pyr, size_t pyr_stride,
                         const DevLevel* levels, const WorkItem* items, uint32_t* sum, uint32_t* sq,
                         size_t int_stride) {
  if (nitems == 0 || nframes == 0) return 0;
  hipLaunchKernelGGL(k_integral_rows, dim3(nitems, nframes), dim3(WG_THREADS), 0, st, pyr, pyr_stride, levels,
                     items, sum, sq, int_stride);
  HC_LAUNCH_CHECK();
  return 0;
}

int launch_integral_cols(hipStream_t st, int nframes, int nitems, const DevLevel* levels, const WorkItem* items,
                         uint32_t* sum, uint32_t* sq, size_t int_stride) {
  if (nitems == 0 || nframes == 0) return 0;
  hipLaunchKernelGGL(k_integral_cols, dim3(nitems, nframes), dim3(WG_THREADS), 0, st, levels, items, sum, sq,
                     int_stride);
  HC_LAUNCH_CHECK();
  return 0;
}

int launch_variance(hipStream_t st, int nframes, int nitems, const DevLevel* levels, const WorkItem* items,
                    const uint32_t* sum, const uint32_t* sq, size_t int_stride, int norm_w, int norm_h,
                    float* vnf, size_t grid_stride) {
  if (nitems == 0 || nframes == 0) return 0;
  hipLaunchKernelGGL(k_variance, dim3(nitems, nframes), dim3(WG_THREADS), 0, st, levels, items, sum, sq,
                     int_stride, norm_w, norm_h, vnf, grid_stride);
  HC_LAUNCH_CHECK();
  return 0;
}

int launch_cascade(hipStream_t st, int nframes, bool tree, size_t lds_bytes, const CascadeArgs& args) {
  if (args.nbands == 0 || nframes == 0) return 0;
  const int gx = ((args.nbands + 7) / 8) * 8;
  if (tree)
    hipLaunchKernelGGL(k_cascade<true>, dim3(gx, nframes), dim3(WG_THREADS), lds_bytes, st, args);
  else
    hipLaunchKernelGGL(k_cascade<false>, dim3(gx, nframes), dim3(WG_THREADS), lds_bytes, st, args);
  HC_LAUNCH_CHECK();
  return 0;
}

}  // namespace hc
