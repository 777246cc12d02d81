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
#include <cstdio>

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

// Inclusive prefix sum over the 64 lanes of a wavefront with DPP (no LDS traffic, no bpermute
// latency): Hillis-Steele inside each row of 16 lanes (row_shr 1/2/4/8, out-of-row lanes read 0),
// then row_bcast:15 carries row totals into rows 1 and 3, row_bcast:31 carries the half-wave
// total into rows 2 and 3.  u32 modular, exact.
__device__ __forceinline__ uint32_t wave_scan_u32(uint32_t v) {
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, false);   // row_shr:1
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, false);   // row_shr:2
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, false);   // row_shr:4
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, false);   // row_shr:8
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xA, 0xF, false);   // row_bcast:15 -> rows 1,3
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xC, 0xF, false);   // row_bcast:31 -> rows 2,3
  return v;
}

// A.6a: variance normalisation factor of one window from the sums of its inner rect (normrect):
// nf = area*valsq - valsum^2 (f64, exact); nf > 0: vnf = (float)(1/sqrt(nf)) kept iff area*(double)vnf < 0.1; else 0
//
// Upstream rounds three times: s = RN64(sqrt(nf)), r = RN64(1/s), v = RN32(r).  The correctly rounded f64 sqrt and
// divide cost ~100 instructions per window, so v is first derived from y ~ 1/sqrt(nf) (v_rsq_f64 + two Newton
// steps) and accepted only when that provably gives the same float: the residual e = 1 - nf*y^2, formed with an FMA,
// bounds y's relative error by |e|/2 + 2^-53, r is within 2^-52 of the true value, and if y's mantissa bits below
// f32 precision stay further than both errors from the rounding midpoint, no float boundary separates y from r, so
// RN32(y) == RN32(r).  Any window failing either test (about 3 in 10^7) takes the three-rounding route literally.
__device__ __forceinline__ float vnf_exact_slow(double nf) {
  nf = sqrt(nf);
  return (float)(1. / nf);
}

__device__ __forceinline__ float vnf_from_sums(int valsum, uint32_t valsq, double area) {
  const double nf = area * (double)valsq - (double)valsum * (double)valsum;
  float out = 0.f;
  if (nf > 0.) {
    double y = __builtin_amdgcn_rsq(nf);
#pragma unroll
    for (int it = 0; it < 2; it++) {
      const double e = __builtin_fma(-(nf * y), y, 1.0);
      y = __builtin_fma(y * 0.5, e, y);
    }
    const double e = __builtin_fma(-(nf * y), y, 1.0);
    const unsigned int low = (unsigned int)__double2loint(y) & 0x1fffffffu;                  // 29 mantissa bits below f32 precision
    const unsigned int dist = low > 0x10000000u ? low - 0x10000000u : 0x10000000u - low;     // ulps(f64) from the f32 rounding midpoint
    const bool sure = fabs(e) < 0x1p-46 && dist > 160u;                                       // |dy/y| < 2^-46 = 64 ulps; r adds < 2
    const float v = sure ? (float)y : vnf_exact_slow(nf);
    if (area * (double)v < 1e-1) out = v;
  }
  return out;
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
// A.5 pass 1: one wavefront per pyramid row; inclusive scan over 64 lanes by DPP (wave_scan_u32),
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
  uint32_t* Q = sq ? sq + o : nullptr;          // sq == nullptr: sum only (mask integral of the detectFaces path)
  uint32_t carry_s = 0, carry_q = 0;
  for (int x0 = 0; x0 < w; x0 += 64) {
    const int x = x0 + lane;
    const uint32_t p = x < w ? src[x] : 0u;
    const uint32_t s = wave_scan_u32(p) + carry_s;
    const uint32_t q = wave_scan_u32(p * p) + carry_q;
    if (x < w) { S[x] = s; if (Q) Q[x] = q; }
    carry_s = (uint32_t)__builtin_amdgcn_readlane((int)s, 63);
    carry_q = (uint32_t)__builtin_amdgcn_readlane((int)q, 63);
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
  if (!sq) {                                    // sum only
    uint32_t as = 0;
    for (int y = 0; y < h; y++) { as += S[(size_t)y * pitch]; S[(size_t)y * pitch] = as; }
    return;
  }
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
// A.4 + A.5 fused: one workgroup owns one pyramid level of one frame and marches down it in chunks
// of R rows.  Phase 1 (wave per row): exact-bilinear pixels straight from the original frame
// (L2-resident), 64-lane DPP inclusive scan of p and p*p with a register carry along the
// row -> row prefixes in LDS.  Phase 2 (thread per column): running column sums kept in
// registers across chunks, final integral rows stored coalesced.  The pyramid never touches
// HBM (unless the debug tap asks for it) and `sum`/`sqsum` are written exactly once:
// HBM traffic = the algorithmic 8 B per integral element.
#ifndef HC_PI_THREADS
#define HC_PI_THREADS 512
#endif
constexpr int PI_THREADS = HC_PI_THREADS;   // 8 (16) wavefronts: one pyramid row each per chunk of 8 (16) rows
constexpr int PI_UNROLL = 4;      // 64-pixel chunks whose source loads are in flight together

template <int CMAX>   // columns per thread: level width <= PI_THREADS * CMAX
__global__ __launch_bounds__(PI_THREADS) void k_pyramid_integral(
    const uint8_t* __restrict__ frames, size_t frame_stride, size_t row_step, int rows, int cols,
    const DevLevel* __restrict__ levels, const uint32_t* __restrict__ xtab, const uint32_t* __restrict__ ytab,
    uint32_t* __restrict__ sum, uint32_t* __restrict__ sq, size_t int_stride, uint8_t* __restrict__ pyr,
    size_t pyr_stride, int R, int lds_pitch, int src_pitch, float* __restrict__ vnf, size_t grid_stride, int norm_w,
    int norm_h) {
  extern __shared__ uint32_t rowbuf[];   // [2][R][lds_pitch] row prefixes | [lds_pitch] x table | per wave 2 source rows
  uint32_t* rs = rowbuf;
  uint32_t* rq = rowbuf + (size_t)R * lds_pitch;
  uint32_t* xt_s = rowbuf + (size_t)2 * R * lds_pitch;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  // the two source rows a destination row interpolates between are staged per wavefront with
  // coalesced dword loads; the 4 bytes per pixel are then gathered from LDS instead of through
  // the (much slower) per-lane byte path of the vector memory unit
  uint8_t* sr0 = (uint8_t*)(xt_s + lds_pitch) + (size_t)wave * 2 * src_pitch;
  uint8_t* sr1 = sr0 + src_pitch;
  const HC_CONST DevLevel& L = as_const(levels)[blockIdx.y];
  const int frame = blockIdx.x;
  const int w = L.w, h = L.h, pitch = L.int_pitch;
  const uint8_t* src = frames + (size_t)frame * frame_stride;
  const uint32_t* yt_l = ytab + L.ytab_ofs;
  uint32_t* S = sum + (size_t)frame * int_stride + L.int_ofs;
  uint32_t* Q = sq ? sq + (size_t)frame * int_stride + L.int_ofs : nullptr;   // null: sum only (mask integral)
  uint8_t* P = pyr ? pyr + (size_t)frame * pyr_stride + L.pyr_ofs : nullptr;
  const bool dwords = ((((size_t)src) | row_step) & 3) == 0;               // uniform: rows are 4-byte aligned
  uint32_t col_s[CMAX], col_q[CMAX];
#pragma unroll
  for (int k = 0; k < CMAX; k++) col_s[k] = col_q[k] = 0;
  for (int x = threadIdx.x; x < lds_pitch; x += PI_THREADS) xt_s[x] = x < w ? xtab[L.xtab_ofs + x] : 0u;
  __syncthreads();

  for (int y0 = 0; y0 < h; y0 += R) {
    const int nr = min(R, h - y0);
    for (int r = wave; r < nr; r += PI_THREADS / 64) {
      const int dy = y0 + r;
      const uint32_t yt = yt_l[dy];
      const int oy = yt >> 9, oy1 = min(oy + 1, rows - 1);
      const uint32_t cy1 = yt & 511u, cy0 = 256u - cy1;
      const uint8_t* g0 = src + (size_t)oy * row_step;
      const uint8_t* g1 = src + (size_t)oy1 * row_step;
      if (dwords) {
        const int nd = (cols + 3) >> 2;                 // src_pitch is a multiple of 4 >= cols; row_step >= cols
        for (int i = lane; i < nd; i += 64) {
          ((uint32_t*)sr0)[i] = ((const uint32_t*)g0)[i];
          ((uint32_t*)sr1)[i] = ((const uint32_t*)g1)[i];
        }
      } else {
        for (int i = lane; i < cols; i += 64) { sr0[i] = g0[i]; sr1[i] = g1[i]; }
      }
      // the same wavefront wrote and now reads the staging rows: LDS operations of one wave execute in
      // order, only the compiler has to be kept from moving the reads up
      __builtin_amdgcn_wave_barrier();
      uint32_t carry_s = 0, carry_q = 0;
      for (int x0 = 0; x0 < w; x0 += 64 * PI_UNROLL) {
        uint32_t px[PI_UNROLL];
#pragma unroll
        for (int u = 0; u < PI_UNROLL; u++) {
          const int x = x0 + 64 * u + lane;
          const uint32_t xt = xt_s[min(x, lds_pitch - 1)];
          const int ox = xt >> 9, ox1 = min(ox + 1, cols - 1);
          const uint32_t cx1 = xt & 511u, cx0 = 256u - cx1;
          const uint32_t h0 = cx0 * sr0[ox] + cx1 * sr0[ox1];
          const uint32_t h1 = cx0 * sr1[ox] + cx1 * sr1[ox1];
          px[u] = x < w ? min((h0 * cy0 + h1 * cy1 + 32768u) >> 16, 255u) : 0u;
        }
#pragma unroll
        for (int u = 0; u < PI_UNROLL; u++) {
          const int x = x0 + 64 * u + lane;
          if (x0 + 64 * u >= w) break;                   // wave-uniform
          if (P && x < w) P[(size_t)dy * w + x] = (uint8_t)px[u];
          const uint32_t s = wave_scan_u32(px[u]) + carry_s;
          const uint32_t q = wave_scan_u32(px[u] * px[u]) + carry_q;
          if (x < w) { rs[r * lds_pitch + x] = s; rq[r * lds_pitch + x] = q; }
          carry_s = (uint32_t)__builtin_amdgcn_readlane((int)s, 63);
          carry_q = (uint32_t)__builtin_amdgcn_readlane((int)q, 63);
        }
      }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < CMAX; k++) {
      const int x = threadIdx.x + k * PI_THREADS;
      if (x < w) {
        uint32_t cs = col_s[k], cq = col_q[k];
        uint32_t* so = S + (size_t)(y0 + 1) * pitch + x + 1;
        uint32_t* qo = Q + (size_t)(y0 + 1) * pitch + x + 1;
        for (int r = 0; r < nr; r++) {
          cs += rs[r * lds_pitch + x];
          cq += rq[r * lds_pitch + x];
          so[(size_t)r * pitch] = cs;
          if (Q) qo[(size_t)r * pitch] = cq;
        }
        col_s[k] = cs; col_q[k] = cq;
      }
    }
    __syncthreads();
    // Phase 3 (A.6a fused in): the variance factor of every window whose inner rect ENDS in this chunk.  Its 8 integral
    // words are this workgroup's own stores -- the bottom row from this chunk, the top row norm_h rows up -- drained by
    // the barrier above (s_waitcnt vmcnt(0) precedes it) and never read by this CU before their final value was
    // written, so they come back from L2 (or L1) current; no other kernel has to re-read the integrals for this.
    // Uses no LDS: the wavefronts that finish first go on to the next chunk's phase 1.
    if (vnf && Q && L.nx > 0) {
      const int ys = L.ystep, gnx = L.nx, gny = L.ny;
      // window row y = gy*ys has its bottom integral row at y + 1 + norm_h; rows y0+1 .. y0+nr were finished just now
      int gy_lo = max(y0 - norm_h, 0);
      gy_lo = (gy_lo + ys - 1) / ys;
      const int gy_hi = min((y0 + nr - 1 - norm_h) >= 0 ? (y0 + nr - 1 - norm_h) / ys : -1, gny - 1);
      const int nrows = gy_hi - gy_lo + 1;
      const double area = (double)(norm_w * norm_h);
      float* vout = vnf + (size_t)frame * grid_stride + L.grid_ofs;
      // four windows per thread and trip: all 32 integral words are requested before the first is used, so a chunk
      // pays the L2 round trip once instead of once per window (the march down the largest level is the kernel's
      // critical path)
      constexpr int PV = 4;
      const int total = nrows * gnx;
      for (int i0 = threadIdx.x; i0 < total; i0 += PV * PI_THREADS) {
        uint32_t s00[PV], s01[PV], s10[PV], s11[PV], q00[PV], q01[PV], q10[PV], q11[PV];
        size_t dst[PV];
#pragma unroll
        for (int u = 0; u < PV; u++) {
          const int i = min(i0 + u * PI_THREADS, total - 1);     // clamped: the surplus lanes recompute the last window
          const int gy = gy_lo + i / gnx, gx = i - (i / gnx) * gnx;
          const size_t p0 = (size_t)(gy * ys + 1) * pitch + gx * ys + 1, p2 = p0 + (size_t)norm_h * pitch;
          s00[u] = S[p0]; s01[u] = S[p0 + norm_w]; s10[u] = S[p2]; s11[u] = S[p2 + norm_w];
          q00[u] = Q[p0]; q01[u] = Q[p0 + norm_w]; q10[u] = Q[p2]; q11[u] = Q[p2 + norm_w];
          dst[u] = (size_t)gy * gnx + gx;
        }
#pragma unroll
        for (int u = 0; u < PV; u++) {
          const int valsum = (int)(s00[u] - s01[u] - s10[u] + s11[u]);
          const uint32_t valsq = q00[u] - q01[u] - q10[u] + q11[u];
          const float v = vnf_from_sums(valsum, valsq, area);
          if (i0 + u * PI_THREADS < total) vout[dst[u]] = v;
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
// A.6a setWindow for every scanned grid position: vnf = (float)(1/sqrt(area*sqsum - sum^2)),
// 0 marks "not evaluable" (nf <= 0 or area*vnf >= 0.1).
__global__ __launch_bounds__(WG_THREADS) void k_variance(
    const DevLevel* __restrict__ levels, const WorkItem* __restrict__ items,
    const uint32_t* __restrict__ sum, const uint32_t* __restrict__ sq, size_t int_stride,
    int norm_w, int norm_h, float* __restrict__ vnf, size_t grid_stride, int nitems, int nframes) {
  // Workgroups are dealt round-robin over the 8 XCDs, each with its own L2.  A window reads integral rows 18 (22)
  // apart, so the rows one workgroup brings in are wanted again by the workgroups a few grid rows further down the
  // same frame: all workgroups of a frame are therefore placed on ONE XCD (frame = 8 * group + id % 8) instead of
  // being sprayed over all eight, which made every XCD fetch the same lines (1.6 x the algorithmic read).
  int item, frame;
  if (nframes % 8 == 0) {
    const int id = blockIdx.x, xcd = id & 7, k = id >> 3;
    item = k % nitems;
    frame = (k / nitems) * 8 + xcd;
  } else {
    item = blockIdx.x % nitems;
    frame = blockIdx.x / nitems;
  }
  const WorkItem it = load_item(items, item);
  const HC_CONST DevLevel& L = as_const(levels)[it.level];
  const int i = it.a + threadIdx.x;
  if (i >= L.nx * L.ny) return;
  const int gy = i / L.nx, gx = i - gy * L.nx;
  const int x = gx * L.ystep + 1, y = gy * L.ystep + 1;       // normrect = (1, 1, win_w-2, win_h-2)
  const size_t base = (size_t)frame * int_stride + L.int_ofs;
  const size_t p0 = base + (size_t)y * L.int_pitch + x, p2 = p0 + (size_t)norm_h * L.int_pitch;
  const int valsum = (int)(sum[p0] - sum[p0 + norm_w] - sum[p2] + sum[p2 + norm_w]);
  const uint32_t valsq = sq[p0] - sq[p0 + norm_w] - sq[p2] + sq[p2 + norm_w];
  const float out = vnf_from_sums(valsum, valsq, (double)(norm_w * norm_h));
  vnf[(size_t)frame * grid_stride + L.grid_ofs + (size_t)gy * L.nx + gx] = out;
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
// A node is fetched as three 16-byte scalar loads into SGPRs (values, never addresses: the
// compiler must not turn the leaf select into a per-lane memory load) and the next node is
// requested before the current one is evaluated, so the scalar-load latency hides behind the
// LDS reads and the arithmetic of the current node.
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
struct NodeRegs { u32x4 q0, q1, q2; };   // q0,q1: off[0..7] + off[8..11] | w0 w1 ; q2: w2 thr a b

__device__ __forceinline__ NodeRegs load_node(const HC_CONST DevNode* nodes, int idx) {
  const HC_CONST u32x4* p = (const HC_CONST u32x4*)(nodes + idx);
  NodeRegs r;
  r.q0 = p[0]; r.q1 = p[1]; r.q2 = p[2];
  return r;
}

#define HC_OFF(q, k) (int)(((q)[(k) >> 1] >> (((k)&1) * 16)) & 0xffffu)

// One corner of the LDS tile for the window whose origin word is `base`; corner k of the node's offset table q.
#ifndef HC_OFF_WORDS
// DevNode::off holds BYTE offsets (hc_api.cpp build_nodes).  The 16-bit half of the scalar register is selected by the
// add itself (SDWA source select on an SGPR operand), so forming a corner address costs one VALU instruction and no
// scalar unpacking (the s_and / s_bfe pair per corner was 12 SALU instructions per stump pair).
__device__ __forceinline__ int hc_corner_addr(uint32_t packed, int half, int base_bytes) {
  int r;
  if (half) asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD" : "=v"(r) : "s"(packed), "v"(base_bytes));
  else asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:DWORD" : "=v"(r) : "s"(packed), "v"(base_bytes));
  return r;
}
// absolute LDS byte address of the window origin: the tile's own LDS address folded in once per window, so the SDWA
// add yields the final address and the load needs no further arithmetic
typedef const int32_t __attribute__((address_space(3))) * hc_lds_cptr;
#define HC_LDS_ADDR(tile, base) ((int)(uint32_t)(uintptr_t)(hc_lds_cptr)(tile) + ((base) << 2))
#define HC_TILE_Q(tile, base, q, k) (*(hc_lds_cptr)(uintptr_t)(uint32_t)hc_corner_addr((q)[(k) >> 1], (k)&1, HC_LDS_ADDR(tile, base)))
#define HC_TILE_O(tile, base, off) (*(hc_lds_cptr)(uintptr_t)(uint32_t)(HC_LDS_ADDR(tile, base) + (int)(off)))
#else
#define HC_TILE_Q(tile, base, q, k) ((tile)[(base) + HC_OFF(q, k)])
#define HC_TILE_O(tile, base, off) ((tile)[(base) + (off)])
#endif

__device__ __forceinline__ float feature_value(const int32_t* tile, int base, const NodeRegs& n, float vnf) {
  const int s0 = HC_TILE_Q(tile, base, n.q0, 0) - HC_TILE_Q(tile, base, n.q0, 1) - HC_TILE_Q(tile, base, n.q0, 2) +
                 HC_TILE_Q(tile, base, n.q0, 3);
  const int s1 = HC_TILE_Q(tile, base, n.q0, 4) - HC_TILE_Q(tile, base, n.q0, 5) - HC_TILE_Q(tile, base, n.q0, 6) +
                 HC_TILE_Q(tile, base, n.q0, 7);
  const float w0 = __uint_as_float(n.q1[2]), w1 = __uint_as_float(n.q1[3]), w2 = __uint_as_float(n.q2[0]);
  float ret = w0 * (float)s0 + w1 * (float)s1;
  if (w2 != 0.0f) {   // wave-uniform branch
    const int s2 = HC_TILE_Q(tile, base, n.q1, 0) - HC_TILE_Q(tile, base, n.q1, 1) - HC_TILE_Q(tile, base, n.q1, 2) +
                   HC_TILE_Q(tile, base, n.q1, 3);
    ret += w2 * (float)s2;
  }
  return ret * vnf;
}

#ifdef HC_PHASE
__device__ unsigned long long hc_phase[10];
static void dump_phase(hipStream_t st, int nwaves) {
  (void)hipStreamSynchronize(st);
  unsigned long long h[10];
  (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(hc_phase), sizeof(h));
  unsigned long long z[10] = {0};
  (void)hipMemcpyToSymbol(HIP_SYMBOL(hc_phase), z, sizeof(z));
  static const char* nm[10] = {"fill_load_store", "stage0", "stage_barriers", "dense", "sparse_requeue", "sparse_finish", "emit", "total",
                               "wait_prev_tile", "fill_barrier"};
  fprintf(stderr, "[phase nw=%d]", nwaves);
  for (int k = 0; k < 10; k++) fprintf(stderr, " %s=%.1f%%", nm[k], 100.0 * (double)h[k] / (double)h[7]);
  fprintf(stderr, "\n");
}
#endif
template <bool TREE>
__device__ __forceinline__ bool stage_passes(const int32_t* tile, int base, float vnf, const HC_CONST DevStage& st,
                                             const HC_CONST DevNode* nodes, const HC_CONST DevTree* trees,
                                             const float* __restrict__ leaves) {
  double acc = 0.0;
  const int first = st.first, count = st.count;
  const float sthr = st.thr;
  if (!TREE) {
    NodeRegs cur = load_node(nodes, first);
    for (int t = 0; t < count; t++) {                         // A.8 predictOrderedStump
      const NodeRegs nxt = load_node(nodes, first + min(t + 1, count - 1));
      const float v = feature_value(tile, base, cur, vnf);
      const float thr = __uint_as_float(cur.q2[1]);
      const float leaf = v < thr ? __uint_as_float(cur.q2[2]) : __uint_as_float(cur.q2[3]);
      acc += (double)leaf;
      cur = nxt;
    }
  } else {
    for (int t = 0; t < count; t++) {                         // A.9 predictOrdered
      const HC_CONST DevTree& T = trees[first + t];
      const int fn = T.first_node, nn = T.nnodes, fl = T.first_leaf;
      int at = 0;                                             // children always follow their parent
      NodeRegs cur = load_node(nodes, fn);
      for (int k = 0; k < nn; k++) {
        const NodeRegs nxt = load_node(nodes, fn + min(k + 1, nn - 1));
        const float v = feature_value(tile, base, cur, vnf);
        const float thr = __uint_as_float(cur.q2[1]);
        const int go = v < thr ? (int)cur.q2[2] : (int)cur.q2[3];
        at = (at == k) ? go : at;
        cur = nxt;
      }
      acc += (double)leaves[fl - at];
    }
  }
  return !(acc < (double)sthr);
}

// ---- stump stages with the stumps grouped by corner-sharing kind (device_types.h NodeKind) ----
// The distinct corners of one stump of kind KIND, read from the LDS tile (6, 8, 9 or 12 values).
template <int KIND>
struct Corners { int a0, a1, a2, a3, b0, b1, b2, b3, c0, c1, c2, c3; };

template <int KIND>
__device__ __forceinline__ Corners<KIND> read_corners(const int32_t* tile, int base, const NodeRegs& n) {
  Corners<KIND> c;
  c.a0 = HC_TILE_Q(tile, base, n.q0, 0); c.a1 = HC_TILE_Q(tile, base, n.q0, 1);
  c.a2 = HC_TILE_Q(tile, base, n.q0, 2); c.a3 = HC_TILE_Q(tile, base, n.q0, 3);
  c.b0 = (KIND == NK_LEFT || KIND == NK_TOP || KIND == NK_DIAG) ? c.a0 : HC_TILE_Q(tile, base, n.q0, 4);
  c.b1 = (KIND == NK_RIGHT || KIND == NK_TOP || KIND == NK_ANTIDIAG) ? c.a1 : HC_TILE_Q(tile, base, n.q0, 5);
  c.b2 = (KIND == NK_BOTTOM || KIND == NK_LEFT) ? c.a2 : HC_TILE_Q(tile, base, n.q0, 6);
  c.b3 = (KIND == NK_BOTTOM || KIND == NK_RIGHT) ? c.a3 : HC_TILE_Q(tile, base, n.q0, 7);
  if (KIND == NK_ANTIDIAG || KIND == NK_DIAG || KIND == NK_NONE3) {
    c.c0 = KIND == NK_DIAG ? c.b3 : HC_TILE_Q(tile, base, n.q1, 0);
    c.c1 = KIND == NK_ANTIDIAG ? c.b2 : HC_TILE_Q(tile, base, n.q1, 1);
    c.c2 = KIND == NK_ANTIDIAG ? c.a2 : HC_TILE_Q(tile, base, n.q1, 2);
    c.c3 = KIND == NK_DIAG ? c.a3 : HC_TILE_Q(tile, base, n.q1, 3);
  } else {
    c.c0 = c.c1 = c.c2 = c.c3 = 0;
  }
  return c;
}

template <int KIND>
__device__ __forceinline__ float feature_from_corners(const Corners<KIND>& c, const NodeRegs& n, float vnf) {
  const float w0 = __uint_as_float(n.q1[2]), w1 = __uint_as_float(n.q1[3]);
  float ret = w0 * (float)(c.a0 - c.a1 - c.a2 + c.a3) + w1 * (float)(c.b0 - c.b1 - c.b2 + c.b3);
  if (KIND == NK_ANTIDIAG || KIND == NK_DIAG || KIND == NK_NONE3)
    ret += __uint_as_float(n.q2[0]) * (float)(c.c0 - c.c1 - c.c2 + c.c3);
  return ret * vnf;
}

// Leaf accumulation of a stump in a stage whose f64 sums are exact in any order (DevStage::exact).  The host stores
// d = (double)left - (double)right in the node (a = low word, b = high word) and starts every stage sum at
// bsum = sum of the right leaves, so sum(leaf) = bsum + sum([v < thr] * d): the f32 select, its two scalar-to-vector
// moves and the f32 -> f64 conversion become one select of the high word of 0.0 / 1.0 and one exact FMA
// (d * c is exact for c in {0, 1}, so fma(d, c, acc) rounds exactly like acc + leaf: not at all, by the proof).
__device__ __forceinline__ void stump_acc(const NodeRegs& n, float v, double& acc) {
  const double d = __hiloint2double((int)n.q2[3], (int)n.q2[2]);
  const double c = __hiloint2double(v < __uint_as_float(n.q2[1]) ? 0x3ff00000 : 0, 0);
  acc = __builtin_fma(d, c, acc);
}

template <int KIND>
__device__ __forceinline__ void stump_run(const int32_t* tile, int base, float vnf, const HC_CONST DevNode* nodes,
                                          int first, int count, double& acc) {
  // Two stumps per iteration: both nodes arrive in one scalar load and the LDS reads of both are issued
  // before either feature is computed, so a wavefront pays the SMEM and the LDS latency once per pair.
  int t = 0;
  for (; t + 2 <= count; t += 2) {                            // wave-uniform trip count
    const NodeRegs n0 = load_node(nodes, first + t), n1 = load_node(nodes, first + t + 1);
    const Corners<KIND> c0 = read_corners<KIND>(tile, base, n0);
    const Corners<KIND> c1 = read_corners<KIND>(tile, base, n1);
    __builtin_amdgcn_sched_barrier(0);                        // keep all reads ahead of the arithmetic
    stump_acc(n0, feature_from_corners<KIND>(c0, n0, vnf), acc);
    stump_acc(n1, feature_from_corners<KIND>(c1, n1, vnf), acc);
  }
  if (t < count) {
    const NodeRegs n0 = load_node(nodes, first + t);
    const Corners<KIND> c0 = read_corners<KIND>(tile, base, n0);
    stump_acc(n0, feature_from_corners<KIND>(c0, n0, vnf), acc);
  }
}

__device__ __forceinline__ bool stage_passes_kinds(const int32_t* tile, int base, float vnf, const HC_CONST DevStage& st,
                                                   const HC_CONST DevStageKinds& kd, const HC_CONST DevNode* nodes) {
  double acc = st.bsum;                                    // the right leaves of every stump; stump_acc adds [v < thr] * (left - right)
  const int first = st.first;
  const float sthr = st.thr;
  const int e0 = kd.end[0], e1 = kd.end[1], e2 = kd.end[2], e3 = kd.end[3], e4 = kd.end[4], e5 = kd.end[5],
            e6 = kd.end[6], e7 = kd.end[7];
  stump_run<NK_NONE2>(tile, base, vnf, nodes, first, e0, acc);
  stump_run<NK_BOTTOM>(tile, base, vnf, nodes, first + e0, e1 - e0, acc);
  stump_run<NK_RIGHT>(tile, base, vnf, nodes, first + e1, e2 - e1, acc);
  stump_run<NK_LEFT>(tile, base, vnf, nodes, first + e2, e3 - e2, acc);
  stump_run<NK_TOP>(tile, base, vnf, nodes, first + e3, e4 - e3, acc);
  stump_run<NK_ANTIDIAG>(tile, base, vnf, nodes, first + e4, e5 - e4, acc);
  stump_run<NK_DIAG>(tile, base, vnf, nodes, first + e5, e6 - e5, acc);
  stump_run<NK_NONE3>(tile, base, vnf, nodes, first + e6, e7 - e6, acc);
  return !(acc < (double)sthr);
}

// Sum of the leaves of stumps [lo, hi) of a kind-grouped (exact) stage, in kind-sorted order: the share of one
// wavefront when several wavefronts split a stage between them (k_cascade, cooperative mode).
__device__ __forceinline__ double stage_partial_kinds(const int32_t* tile, int base, float vnf, const HC_CONST DevStage& st,
                                                      const HC_CONST DevStageKinds& kd, const HC_CONST DevNode* nodes,
                                                      int lo, int hi) {
  double acc = 0.0;
  const int first = st.first;
  const int e0 = kd.end[0], e1 = kd.end[1], e2 = kd.end[2], e3 = kd.end[3], e4 = kd.end[4], e5 = kd.end[5],
            e6 = kd.end[6], e7 = kd.end[7];
#define HC_RUN(KIND, A_, B_) { const int a_ = max(A_, lo), b_ = min(B_, hi); stump_run<KIND>(tile, base, vnf, nodes, first + a_, b_ - a_, acc); }
  HC_RUN(NK_NONE2, 0, e0)
  HC_RUN(NK_BOTTOM, e0, e1)
  HC_RUN(NK_RIGHT, e1, e2)
  HC_RUN(NK_LEFT, e2, e3)
  HC_RUN(NK_TOP, e3, e4)
  HC_RUN(NK_ANTIDIAG, e4, e5)
  HC_RUN(NK_DIAG, e5, e6)
  HC_RUN(NK_NONE3, e6, e7)
#undef HC_RUN
  return acc;
}

// One stage for one window per lane: trees -> ordered tree walk; stump stage with an order-free
// (exact) sum -> kind-grouped runs; any other stump stage -> ordered generic evaluation in file order.
template <bool TREE>
__device__ __forceinline__ bool eval_stage(const int32_t* tile, int base, float vnf, int s, const HC_CONST DevStage* stages,
                                           const HC_CONST DevStageKinds* kinds, const HC_CONST DevNode* nodes,
                                           const HC_CONST DevTree* trees, const float* __restrict__ leaves) {
  if (TREE) return stage_passes<true>(tile, base, vnf, stages[s], nodes, trees, leaves);
  if (stages[s].exact) return stage_passes_kinds(tile, base, vnf, stages[s], kinds[s], nodes);
  return stage_passes<false>(tile, base, vnf, stages[s], nodes, trees, leaves);
}

// ---- sparse mode: ONE window per wavefront, the stage's weak classifiers spread over the
// 64 lanes.  Legal only for stages whose f64 sum is exact in every order (DevStage::exact,
// proven on the host from the leaf values): then the order-free wave reduction gives the very
// same double as upstream's sequential accumulation.  Removes the long serial stump chains of
// the few windows that reach the deep stages.
__device__ __forceinline__ float feature_value_lane(const int32_t* tile, int base, const DevNode& n, float vnf) {
  const int s0 = HC_TILE_O(tile, base, n.off[0]) - HC_TILE_O(tile, base, n.off[1]) - HC_TILE_O(tile, base, n.off[2]) + HC_TILE_O(tile, base, n.off[3]);
  const int s1 = HC_TILE_O(tile, base, n.off[4]) - HC_TILE_O(tile, base, n.off[5]) - HC_TILE_O(tile, base, n.off[6]) + HC_TILE_O(tile, base, n.off[7]);
  float ret = n.w[0] * (float)s0 + n.w[1] * (float)s1;
  if (n.w[2] != 0.0f) {
    const int s2 = HC_TILE_O(tile, base, n.off[8]) - HC_TILE_O(tile, base, n.off[9]) - HC_TILE_O(tile, base, n.off[10]) + HC_TILE_O(tile, base, n.off[11]);
    ret += n.w[2] * (float)s2;
  }
  return ret * vnf;
}

// Sum of one f64 per lane over the wavefront, with DPP instead of LDS bpermutes: the inclusive-scan
// pattern of wave_scan_u32 on the two halves of the double; lane 63 ends up with the total, which
// is then broadcast.  Only used for stages whose sum is exact in any order, so the association
// chosen here cannot change the result.
#define HC_DPP_F64(v, CTRL, ROWMASK, BC)                                                               \
  __hiloint2double(__builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROWMASK, 0xF, BC),         \
                   __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROWMASK, 0xF, BC))
__device__ __forceinline__ double wave_sum_f64(double v) {
  // row_shr with every row enabled: bound_ctrl makes the lanes shifted in from outside the row read +0.0 by
  // themselves, so the destination needs no zero-initialisation (two v_mov fewer per step)
  v += HC_DPP_F64(v, 0x111, 0xF, true);    // row_shr:1
  v += HC_DPP_F64(v, 0x112, 0xF, true);    // row_shr:2
  v += HC_DPP_F64(v, 0x114, 0xF, true);    // row_shr:4
  v += HC_DPP_F64(v, 0x118, 0xF, true);    // row_shr:8
  v += HC_DPP_F64(v, 0x142, 0xA, false);   // row_bcast:15 -> rows 1,3 (rows 0,2 keep the 0 of `old`)
  v += HC_DPP_F64(v, 0x143, 0xC, false);   // row_bcast:31 -> rows 2,3
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 63), __builtin_amdgcn_readlane(__double2loint(v), 63));
}

template <bool TREE>
__device__ __forceinline__ bool stage_passes_sparse(const int32_t* tile, int base, float vnf, int first, int count,
                                                    float sthr, double bsum, int lane, const DevNode* __restrict__ nodes,
                                                    const DevTree* __restrict__ trees,
                                                    const float* __restrict__ leaves) {
  double part = 0.0;
  for (int t = lane; t < count; t += 64) {
    if (!TREE) {
      const DevNode n = nodes[first + t];
      const float v = feature_value_lane(tile, base, n, vnf);
      // exact stage: a/b hold d = left - right as an f64, the stage sum starts at bsum (see stump_acc)
      part = __builtin_fma(__hiloint2double((int)n.b, (int)n.a), __hiloint2double(v < n.thr ? 0x3ff00000 : 0, 0), part);
    } else {
      const DevTree T = trees[first + t];
      int at = 0;
      do {
        const DevNode n = nodes[T.first_node + at];
        const float v = feature_value_lane(tile, base, n, vnf);
        at = v < n.thr ? (int)n.a : (int)n.b;
      } while (at > 0);
      part += (double)leaves[T.first_leaf - at];
    }
  }
  return !(wave_sum_f64(part) + bsum < (double)sthr);   // bsum is 0 for tree stages
}

// LDS layout of k_cascade: [ tile: tile_words x i32 ][ vnf: ROWS*64 x f32 ][ bank matrix: ROWS*64 x u16 ][ survivor list: ROWS*64 x u16 ]
//                           [ 32 per-bank counter words + sparse-finish ticket ]   (-DHC_PLAIN_QUEUES: two plain queues + 3 counters, the round-1 form)
//
// Workgroup = 4 wavefronts = one band of ROWS grid rows of one level of one frame.  It walks
// the band left to right in steps of TILE_POS positions; per step it stages the
// (ROWS-1)*ystep+win_h+1 by (TILE_POS-1)*ystep+win_w+1 window of the `sum` integral in LDS
// (coalesced row loads; for ystep 2 the even and odd columns go to separate planes so that
// lane k's corner reads are stride-1 in LDS).  Wave w owns grid rows w, w+4, .. of the band
// for the whole walk, so the x-skip carry (A.7) stays in its registers.  Bigger ROWS = bigger
// survivor pool per workgroup = fuller wavefronts in the middle stages, and less halo per window.
// Stage the (G.rows x G.cols) window of the `sum` integral whose top-left corner is (x0, y0) into the LDS tile;
// wave w loads rows w, w + NW, ..  Out-of-image entries read as 0 (only windows that are never evaluated use them).
template <int NW>
__device__ __forceinline__ void fill_tile(int32_t* tile, const uint32_t* __restrict__ I, const HC_CONST DevLevel& L,
                                          const TileGeom& G, int ys, int x0, int y0, int wave, int lane) {
  for (int r = wave; r < G.rows; r += NW) {
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
}

#ifndef HC_CASCADE_WAVES
#define HC_CASCADE_WAVES 8   // waves per SIMD the register allocator must leave room for
#endif
#ifdef HC_PHASE
#define PH_MARK(k) do { const unsigned long long now_ = __builtin_readcyclecounter(); ph[k] += now_ - tlast; tlast = now_; } while (0)
#else
#define PH_MARK(k) do {} while (0)
#endif
// TAPS: the per-window stage-count tap is compiled in only for the debug instantiation (tests); the production kernel
// carries neither its branches nor its address arithmetic
template <bool TREE, int ROWS, int NW, bool SEG, bool TAPS>   // NW wavefronts per workgroup (4 or 8); SEG: work items are band segments
__global__ __launch_bounds__(NW * 64, HC_CASCADE_WAVES) void k_cascade(CascadeArgs A) {
  int8_t* const taps = TAPS ? A.taps : nullptr;
  constexpr int TILE_WIN = ROWS * TILE_POS;
  constexpr int NTHREADS = NW * 64;
  extern __shared__ int32_t smem[];
  // The fixed-size regions come first and the integral tile (whose size depends on the window and the scan step)
  // last: every region then starts at a compile-time constant, which the DS instructions carry in their offset field
  // instead of each (region, lane pattern) pair occupying an address VGPR.
  float* vnf_t = (float*)smem;                                  // variance factor of every window of the tile
#ifndef HC_PLAIN_QUEUES
  int32_t* tile = smem + (TILE_WIN * 8 + 136) / 4;
#else
  int32_t* tile = smem + (TILE_WIN * 8 + 16) / 4;
#endif
#ifndef HC_PLAIN_QUEUES
  // Survivor lists ordered by LDS bank.  Every corner read of a window is tile[base + off] with `off` uniform over
  // the wavefront, so which lanes of a 32-lane group collide on a bank is decided by base & 31 alone -- for every
  // read of every weak classifier of the stage.  A window that passes a stage is appended to column r = base & 31
  // of a [layer][32] matrix (one LDS atomic on counter r; a tile has exactly 2 * ROWS windows per residue, so
  // TILE_WIN entries always suffice).  Before the next stage the matrix is copied, layer by layer and skipping the
  // holes, into a dense list: every complete layer becomes one aligned run of 32 entries with 32 different banks,
  // i.e. a half-wavefront whose LDS reads are conflict-free; only the ragged last layers still collide.
  uint16_t* bankm = (uint16_t*)(vnf_t + TILE_WIN);              // [layer][32] survivors of the running stage
  uint16_t* list = bankm + TILE_WIN;                            // dense, bank-ordered input list of the running stage
  // 32 per-bank counters; the two alternating sets share a word (low / high 16 bits); word 32: sparse-finish ticket
  unsigned int* cres = (unsigned int*)(list + TILE_WIN);
  int* const ticket = (int*)(cres + 32);
  uint16_t* const queue = list;                                 // cooperative-stage scratch lives behind the list's first 256 entries
#define HC_APPEND(POS, BASE, SHOUT)                                                                              \
  do {                                                                                                           \
    const int r_ = (BASE) & 31;                                                                                  \
    bankm[((atomicAdd(&cres[r_], 1u << (SHOUT)) >> (SHOUT)) & 0xffffu) * 32 + r_] = (uint16_t)(POS);            \
  } while (0)
#else
  uint16_t* queue = (uint16_t*)(vnf_t + TILE_WIN);              // queue[2][TILE_WIN]
  int* cnt = (int*)(queue + 2 * TILE_WIN);                      // three rotating counters
#endif
  // Dispatch order: frame is the FAST grid dimension, bands (largest levels first) the slow one,
  // so the long-running bands of every frame start first and the tail of the launch is made of
  // the smallest bands.  With a batch that is a multiple of 8, workgroups b and b + nframes
  // (adjacent bands of one frame, whose staged rows overlap by win_h) still share an XCD/L2.
  const int band = blockIdx.y;
  const WorkItem it = load_item(A.bands, band);
  const HC_CONST DevLevel& L = as_const(A.levels)[it.level];
  const HC_CONST DevStage* stages = as_const(A.stages);
  const HC_CONST DevTree* trees = as_const(A.trees);
  const HC_CONST DevNode* nodes = as_const(A.nodes);
  const HC_CONST DevStageKinds* kinds = as_const(A.kinds);
  const int frame = blockIdx.x;
  const int ys = L.ystep;
  const TileGeom G = A.g;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int gy0 = it.a;
  const int y0 = gy0 * ys;
  const int nx = L.nx, ny = L.ny;
  const uint32_t* __restrict__ I = A.sum + (size_t)frame * A.int_stride + L.int_ofs;
  const size_t grid_base = (size_t)frame * A.grid_stride + L.grid_ofs;
  const float* __restrict__ vnf_g = A.vnf + grid_base;
  const unsigned long long key_hi = ((unsigned long long)frame << 40) | ((unsigned long long)it.level << 32);
  const unsigned long long lt_mask = lane ? ((1ull << lane) - 1ull) : 0ull;
  const int row_stages = min(A.row_stages, A.nstages);   // stages evaluated in row form before the first compaction
  unsigned int carry = 0xffffffffu;   // bit j: is position 0 of the next tile visited, for owned row wave + 4j
#ifdef HC_PHASE
  unsigned long long ph[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long tlast = __builtin_readcyclecounter();
  const unsigned long long tstart = tlast;
#endif

  // A work item is a whole band (b == 0) or a segment of it: b = first grid column | tiles << 16.  Segments let a
  // small batch use the whole chip (a band walk is a serial chain of tiles); the only state a walk carries from
  // tile to tile is the x-skip bit of every row (A.7), which a segment re-derives by looking back: it evaluates
  // stage 0 on the tile to its left and applies the closed form below to position 63.  A row whose 64 positions
  // there all fail stage 0 passes its incoming bit through unchanged (64 is even), so only such rows look one
  // more tile to the left, down to the row start, whose bit is 1.
  // (Whole-band launches use the SEG = false instantiation: the full batches' kernel carries none of this.)
  const int gx_begin = SEG ? (it.b & 0xffff) : 0;
  const int gx_end = SEG && (it.b >> 16) ? min(nx, gx_begin + (it.b >> 16) * TILE_POS) : nx;
  if (SEG && gx_begin > 0) {
    unsigned int need = 0;
    for (int j = 0; j < ROWS / NW; j++)
      if (gy0 + wave + NW * j < ny) need |= 1u << j;
    for (int gp = gx_begin - TILE_POS;; gp -= TILE_POS) {
      if (!__syncthreads_or(need != 0 && gp >= 0)) break;       // uniform; also fences the previous tile
      fill_tile<NW>(tile, I, L, G, ys, gp * ys, y0, wave, lane);
      __syncthreads();
#pragma unroll 1
      for (int j = 0; j < ROWS / NW; j++) {
        if (!((need >> j) & 1u)) continue;                        // wave-uniform
        const int rr = wave + NW * j;
        const float vnf = vnf_g[(size_t)(gy0 + rr) * nx + gp + lane];   // gp + 63 < gx_begin <= nx: all valid
        const bool pass0 = eval_stage<TREE>(tile, rr * ys * G.pitch + lane, vnf, 0, stages, kinds, nodes, trees, A.leaves);
        const unsigned long long fm = __ballot(vnf > 0.f && !pass0);
        if (!(fm >> 63)) { need &= ~(1u << j); continue; }        // position 63 does not fail: next bit is 1
        const unsigned long long below = ~fm & 0x7fffffffffffffffull;
        if (!below) continue;                                     // all 64 fail: pass-through, look further left
        const int anchor = 64 - __clzll((long long)below);
        if (((63 - anchor) & 1) == 0) carry &= ~(1u << j);        // position 63 visited and failing: skip position 0
        need &= ~(1u << j);
      }
    }
  }

  for (int gx0 = gx_begin; gx0 < gx_end; gx0 += TILE_POS) {
    const int x0 = gx0 * ys;
    __syncthreads();              // everybody is done with the previous tile and its queues
    PH_MARK(8);
    fill_tile<NW>(tile, I, L, G, ys, x0, y0, wave, lane);
#ifndef HC_PLAIN_QUEUES
    if (threadIdx.x < 33) cres[threadIdx.x] = 0;
#else
    if (threadIdx.x < 3) cnt[threadIdx.x] = 0;
#endif
    PH_MARK(0);
    __syncthreads();
    PH_MARK(9);

    // ---- stage 0 on every position; x-skip chain; survivors -> queue 0
#pragma unroll 1
    for (int j = 0; j < ROWS / NW; j++) {
      const int rr = wave + NW * j;
      const int gy = gy0 + rr, gx = gx0 + lane;
      if (gy >= ny) continue;                                   // wave-uniform
      const bool valid = gx < nx;
      const float vnf = valid ? vnf_g[(size_t)gy * nx + gx] : 0.f;
      const bool evaluable = vnf > 0.f;
      const int base = rr * ys * G.pitch + lane;
      vnf_t[rr * TILE_POS + lane] = vnf;
      const bool pass0 = eval_stage<TREE>(tile, base, vnf, 0, stages, kinds, nodes, trees, A.leaves);
      const bool f = evaluable && !pass0;                      // "fails stage 0 if evaluated"

      // A.7 in closed form: position k is visited iff its distance to the nearest anchor
      // (k == row start, or f[k-1] == 0) at or left of it is even.
      const unsigned long long fm = __ballot(f);
      const unsigned long long below = ~fm & lt_mask;
      const int anchor = below ? 64 - __clzll((long long)below) : (((carry >> j) & 1u) ? 0 : 1);
      const bool visited = ((lane - anchor) & 1) == 0;
      const unsigned long long vm = __ballot(visited);
      if (((vm & fm) >> 63) & 1ull) carry &= ~(1u << j); else carry |= (1u << j);

      bool surv = valid && visited && evaluable && pass0;
      if (taps && valid && visited && !surv)
        taps[grid_base + (size_t)gy * nx + gx] = (int8_t)(evaluable ? 0 : -1);
      // the next stage(s) stay in row form: lanes = adjacent windows -> conflict-free LDS reads
      // and no queue traffic while most windows are still alive
      for (int s = 1; s < row_stages; s++) {
        if (!__any(surv)) break;
        const bool ok = eval_stage<TREE>(tile, base, vnf, s, stages, kinds, nodes, trees, A.leaves);
        if (taps && surv && !ok) taps[grid_base + (size_t)gy * nx + gx] = (int8_t)s;
        surv = surv && ok;
      }
#ifndef HC_PLAIN_QUEUES
      if (surv) HC_APPEND(rr * TILE_POS + lane, base, 0);
#else
      const unsigned long long sm = __ballot(surv);
      if (sm) {
        int off = 0;
        if (lane == 0) off = atomicAdd(&cnt[0], __popcll(sm));
        off = __builtin_amdgcn_readfirstlane(off);
        if (surv) queue[off + __popcll(sm & lt_mask)] = (uint16_t)(rr * TILE_POS + lane);
      }
#endif
    }
    PH_MARK(1);
    __syncthreads();
    PH_MARK(2);

    // ---- remaining stages on the compacted survivor list, re-compacted after every stage.
    // With t = s - row_stages: queue t&1 is read, queue (t+1)&1 is written; counters rotate over
    // three slots so that the slot zeroed during a stage is neither the one read nor the one
    // being appended to.
    int s = row_stages;
#ifndef HC_PLAIN_QUEUES
    // Step t = s - row_stages: the windows that passed the previous stage sit in bankm, counted per bank in counter
    // half t & 1; they are first copied into `list`, then stage s runs over the list and appends its own survivors
    // to bankm / half (t + 1) & 1.
    for (;; s++) {
      const int t = s - row_stages;
      const int sh_in = (t & 1) * 16, sh_out = 16 - sh_in;
      // `lo` is the lane id made opaque to the optimiser: address expressions of this once-per-stage bookkeeping are
      // then formed here instead of being hoisted out of the tile loop, where they sat in VGPRs across the stump loops
      // (64-VGPR budget) and were spilled to scratch
      int lo = lane;
      asm volatile("" : "+v"(lo));
      const int kr = (int)((cres[lo & 31] >> sh_in) & 0xffffu);   // entries in this lane's bank column
      // every wavefront derives the same layer prefix: lane j ends up with s_mine = entries in layers < j
      int n = 0, s_mine = 0, nlayers = 0;
      for (;; nlayers++) {                                          // uniform
        const unsigned int lm = (unsigned int)__ballot(kr > nlayers);   // lanes 0..31: one per bank
        if (!lm) break;
        if (lo == nlayers) s_mine = n;
        n += __popc(lm);
      }
      if (n == 0) break;                                        // uniform
      for (int j0 = wave * 2; j0 < nlayers; j0 += NW * 2) {     // one layer per half-wavefront
        const int s0 = __builtin_amdgcn_readlane(s_mine, j0), s1 = __builtin_amdgcn_readlane(s_mine, min(j0 + 1, 63));
        const int jl = j0 + (lo >> 5), r = lo & 31;
        const bool have = jl < kr;
        const unsigned long long hm = __ballot(have);
        const unsigned int half = (unsigned int)(lo < 32 ? hm : (hm >> 32));
        if (have) list[(lo < 32 ? s0 : s1) + __popc(half & ((1u << r) - 1u))] = bankm[jl * 32 + r];
      }
      PH_MARK(4);
      __syncthreads();
      PH_MARK(2);
      const uint16_t* qin = list;
      if (s == A.nstages) {                                     // the list holds the accepted windows
        for (int i = threadIdx.x; i < n; i += NTHREADS) {       // A.6c: raw candidates
          const int pos = qin[i];
          const int gy = gy0 + (pos >> 6), gx = gx0 + (pos & 63);
          if (A.cand_cap) {
            const unsigned int slot = atomicAdd(A.cand_count, 1u);
            if (slot < A.cand_cap) A.cand[slot] = key_hi | ((unsigned long long)gy << 16) | (unsigned long long)gx;
          }
          if (taps) taps[grid_base + (size_t)gy * nx + gx] = (int8_t)A.nstages;
        }
        break;
      }
      if (threadIdx.x < 32) atomicAnd(&cres[threadIdx.x], 0xffffu << sh_out);   // every wave has read this half
      int* const cout = ticket;
#else
    for (; s < A.nstages; s++) {
      const int t = s - row_stages;
      const int n = cnt[t % 3];
      if (n == 0) break;                                        // uniform: read after the barrier
      const uint16_t* qin = queue + (t & 1) * TILE_WIN;
      uint16_t* qout = queue + ((t + 1) & 1) * TILE_WIN;
      int* cout = &cnt[(t + 1) % 3];
      if (threadIdx.x == 0) cnt[(t + 2) % 3] = 0;
#endif
      const int sfirst = stages[s].first, scount = stages[s].count;
      const float sthr = stages[s].thr;
      if (n <= A.sparse_max && stages[s].exact >= 2) {
        // Few survivors and every remaining stage has an order-free sum (exact == 2): finish them
        // here without any further queue or barrier.  Each wavefront pulls the next window off the list
        // (the still unused output counter of this stage serves as the work index: windows live from one
        // more stage to all of them, a static split leaves wavefronts idle behind the unlucky one); a window
        // runs through all remaining stages one stage at a time, the stage's weak classifiers spread over
        // the 64 lanes.
        for (;;) {
          // all 64 lanes add 1: the counter moves in steps of 64 per grab, any lane's ticket >> 6 is the index
          const int i = __builtin_amdgcn_readfirstlane(atomicAdd(cout, 1)) >> 6;
          if (i >= n) break;
          const int pos = qin[i];
          const int wy = gy0 + (pos >> 6), wx = gx0 + (pos & 63);
          const int base = (pos >> 6) * ys * G.pitch + (pos & 63);
          const float v = vnf_t[pos];
          int ss = s;
          for (; ss < A.nstages; ss++)
            if (!stage_passes_sparse<TREE>(tile, base, v, stages[ss].first, stages[ss].count, stages[ss].thr, stages[ss].bsum,
                                           lane, A.nodes, A.trees, A.leaves))
              break;
          if (lane == 0) {
            if (ss == A.nstages && A.cand_cap) {                  // A.6c: raw candidate (cap 0: truncated-cascade timing runs)
              const unsigned int slot = atomicAdd(A.cand_count, 1u);
              if (slot < A.cand_cap) A.cand[slot] = key_hi | ((unsigned long long)wy << 16) | (unsigned long long)wx;
            }
            if (taps) taps[grid_base + (size_t)wy * nx + wx] = (int8_t)ss;
          }
        }
        s = -1;                                                   // tile finished (uniform)
        PH_MARK(5);
        break;
      } else if (n <= A.sparse_max && stages[s].exact) {
        // few survivors: one window per wavefront, lanes across the stage's weak classifiers
        for (int i = wave; i < n; i += NW) {
          const int pos = qin[i];
          const int wy = gy0 + (pos >> 6), wx = gx0 + (pos & 63);
          const int base = (pos >> 6) * ys * G.pitch + (pos & 63);
          const bool ok = stage_passes_sparse<TREE>(tile, base, vnf_t[pos], sfirst, scount, sthr, stages[s].bsum,
                                                    lane, A.nodes, A.trees, A.leaves);
          if (lane == 0) {
#ifndef HC_PLAIN_QUEUES
            if (ok) HC_APPEND(pos, base, sh_out);
#else
            if (ok) qout[atomicAdd(cout, 1)] = (uint16_t)pos;
#endif
            else if (taps) taps[grid_base + (size_t)wy * nx + wx] = (int8_t)s;
          }
        }
#ifndef HC_PLAIN_QUEUES
      } else if (!TREE && ROWS >= 16 && stages[s].exact && ((n + 63) >> 6) * 2 <= NW && ((n + 63) >> 6) <= 3) {   // three scratch slots
#else
      } else if (!TREE && ROWS >= 16 && stages[s].exact && ((n + 63) >> 6) * 2 <= NW) {
#endif
        // Cooperative mode: the survivors fill at most NW/2 wavefronts, so the stage's critical path -- every
        // stump in sequence on one wavefront while the others wait at the barrier -- is cut by giving each
        // 64-window chunk to k = 2 or 4 wavefronts that evaluate 1/k of the stumps each.  Partial sums meet
        // in LDS (the unused tails of the two queues); legal because the stage's f64 sum is order-free.
        const int chunks = (n + 63) >> 6;
#ifndef HC_PLAIN_QUEUES
        const int k = chunks == 1 && NW >= 4 ? 4 : 2;             // chunks * (k - 1) <= 3 scratch slots behind the list
#else
        const int k = chunks * 4 <= NW && chunks <= 2 ? 4 : 2;
#endif
        const int chunk = wave / k, slice = wave - chunk * k;
        // scratch slot q = chunk * (k - 1) + slice - 1 (at most 6 of 64 doubles): three behind the first 256
        // entries of each queue (chunks <= 4, so neither queue holds more than 256 entries in this stage;
        // ROWS >= 16 makes a queue 2 KB; the host keeps tile_words even, so the doubles are 8-byte aligned)
        double* const scr0 = (double*)(queue + 256);
#ifndef HC_PLAIN_QUEUES
        double* const scr1 = scr0;                                // never used: at most three slots in this build
#else
        double* const scr1 = (double*)(queue + TILE_WIN + 256);
#endif
        int lc = lane;
        asm volatile("" : "+v"(lc));                              // see `lo` above: keeps the scratch addresses out of long-lived VGPRs
        const int i = chunk * 64 + lc;
        const bool active = chunk < chunks && i < n;
        const int pos = active ? qin[i] : qin[0];
        const int base = (pos >> 6) * ys * G.pitch + (pos & 63);
        double part = 0.0;
        if (chunk < chunks) {                                     // wave-uniform
          part = stage_partial_kinds(tile, base, vnf_t[pos], stages[s], kinds[s], nodes, scount * slice / k,
                                     scount * (slice + 1) / k);
          if (slice) {
            const int q = chunk * (k - 1) + slice - 1;
            (q < 3 ? scr0 + q * 64 : scr1 + (q - 3) * 64)[lc] = part;
          }
        }
        PH_MARK(3);
        __syncthreads();
        PH_MARK(2);
        if (chunk < chunks && slice == 0) {
          for (int j = 1; j < k; j++) {
            const int q = chunk * (k - 1) + j - 1;
            part += (q < 3 ? scr0 + q * 64 : scr1 + (q - 3) * 64)[lc];
          }
          part += stages[s].bsum;                                 // stage_partial_kinds sums [v < thr] * (left - right) only
          const bool ok = !(part < (double)sthr);
          const bool surv = active && ok;
#ifndef HC_PLAIN_QUEUES
          if (surv) HC_APPEND(pos, base, sh_out);
#else
          const unsigned long long sm = __ballot(surv);
          if (sm) {
            int off = 0;
            if (lane == 0) off = atomicAdd(cout, __popcll(sm));
            off = __builtin_amdgcn_readfirstlane(off);
            if (surv) qout[off + __popcll(sm & lt_mask)] = (uint16_t)pos;
          }
#endif
          if (taps && active && !ok) taps[grid_base + (size_t)(gy0 + (pos >> 6)) * nx + gx0 + (pos & 63)] = (int8_t)s;
        }
      } else {
        for (int i0 = wave * 64; i0 < n; i0 += NTHREADS) {
          const int i = i0 + lane;
          const bool active = i < n;
          const int pos = active ? qin[i] : qin[0];
          const int wy = gy0 + (pos >> 6), wx = gx0 + (pos & 63);
          const int base = (pos >> 6) * ys * G.pitch + (pos & 63);
          const bool ok = eval_stage<TREE>(tile, base, vnf_t[pos], s, stages, kinds, nodes, trees, A.leaves);
          const bool surv = active && ok;
#ifndef HC_PLAIN_QUEUES
          if (surv) HC_APPEND(pos, base, sh_out);
#else
          const unsigned long long sm = __ballot(surv);
          if (sm) {
            int off = 0;
            if (lane == 0) off = atomicAdd(cout, __popcll(sm));
            off = __builtin_amdgcn_readfirstlane(off);
            if (surv) qout[off + __popcll(sm & lt_mask)] = (uint16_t)pos;
          }
#endif
          if (taps && active && !ok) taps[grid_base + (size_t)wy * nx + wx] = (int8_t)s;
        }
        PH_MARK(3);
      }
      PH_MARK(4);
      __syncthreads();
      PH_MARK(2);
    }
#ifdef HC_PLAIN_QUEUES
    if (s == A.nstages) {                                       // the last written queue holds the accepted windows
      const int t = s - row_stages;
      const int n = cnt[t % 3];
      const uint16_t* qin = queue + (t & 1) * TILE_WIN;
      for (int i = threadIdx.x; i < n; i += NTHREADS) {       // A.6c: raw candidates
        const int pos = qin[i];
        const int gy = gy0 + (pos >> 6), gx = gx0 + (pos & 63);
        if (A.cand_cap) {
          const unsigned int slot = atomicAdd(A.cand_count, 1u);
          if (slot < A.cand_cap) A.cand[slot] = key_hi | ((unsigned long long)gy << 16) | (unsigned long long)gx;
        }
        if (taps) taps[grid_base + (size_t)gy * nx + gx] = (int8_t)A.nstages;
      }
    }
#endif
    PH_MARK(6);
  }
#ifdef HC_PHASE
  if (lane == 0) {
    ph[7] = __builtin_readcyclecounter() - tstart;
    for (int k = 0; k < 10; k++) atomicAdd(&hc_phase[k], ph[k]);
  }
#endif
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

int launch_pyramid_integral(hipStream_t st, int nframes, int nlevels, int max_w, const uint8_t* frames,
                            size_t frame_stride, size_t row_step, int rows, int cols, const DevLevel* levels,
                            const uint32_t* xtab, const uint32_t* ytab, uint32_t* sum, uint32_t* sq, size_t int_stride,
                            uint8_t* pyr, size_t pyr_stride, float* vnf, size_t grid_stride, int norm_w, int norm_h) {
  if (nlevels == 0 || nframes == 0) return 0;
  const int lds_pitch = (max_w + 63) & ~63;
  const int src_pitch = (cols + 3 + 3) & ~3;     // staged source row: dword granular, >= cols
  const size_t stage = (size_t)(PI_THREADS / 64) * 2 * src_pitch;
  int R = PI_THREADS / 64;                       // rows per chunk: (2 R + 1) x pitch words + staging rows of LDS
  while (R > 1 && (size_t)(2 * R + 1) * lds_pitch * 4 + stage > 81920u) R >>= 1;      // 2 workgroups per CU
  if (R < 4) {                                                                          // very wide frames: 1 per CU
    R = PI_THREADS / 64;
    while (R > 1 && (size_t)(2 * R + 1) * lds_pitch * 4 + stage > 147456u) R >>= 1;
  }
  const size_t lds = (size_t)(2 * R + 1) * lds_pitch * 4 + stage;
  const dim3 grid(nframes, nlevels), block(PI_THREADS);
#define HC_PI(C)                                                                                                  \
  hipLaunchKernelGGL((k_pyramid_integral<C>), grid, block, lds, st, frames, frame_stride, row_step, rows, cols,   \
                     levels, xtab, ytab, sum, sq, int_stride, pyr, pyr_stride, R, lds_pitch, src_pitch, vnf, grid_stride,    \
                     norm_w, norm_h)
  if (max_w <= 2 * PI_THREADS) HC_PI(2);
  else if (max_w <= 4 * PI_THREADS) HC_PI(4);
  else if (max_w <= 8 * PI_THREADS) HC_PI(8);
  else return (int)hipErrorInvalidValue;
#undef HC_PI
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
  hipLaunchKernelGGL(k_variance, dim3((unsigned)nitems * (unsigned)nframes), dim3(WG_THREADS), 0, st, levels, items, sum, sq,
                     int_stride, norm_w, norm_h, vnf, grid_stride, nitems, nframes);
  HC_LAUNCH_CHECK();
  return 0;
}

template <bool TREE, bool SEG, bool TAPS>
static int launch_cascade_rows(hipStream_t st, int nframes, int rows, int nwaves, size_t lds_bytes, const CascadeArgs& args) {
  const dim3 grid(nframes, args.nbands), block(nwaves * 64);
  if (nwaves == 4) {
    switch (rows) {
      case 8: hipLaunchKernelGGL((k_cascade<TREE, 8, 4, SEG, TAPS>), grid, block, lds_bytes, st, args); break;
      case 16: hipLaunchKernelGGL((k_cascade<TREE, 16, 4, SEG, TAPS>), grid, block, lds_bytes, st, args); break;
      case 32: hipLaunchKernelGGL((k_cascade<TREE, 32, 4, SEG, TAPS>), grid, block, lds_bytes, st, args); break;
      default: return (int)hipErrorInvalidValue;
    }
  } else if (nwaves == 8) {
    switch (rows) {
      case 8: hipLaunchKernelGGL((k_cascade<TREE, 8, 8, SEG, TAPS>), grid, block, lds_bytes, st, args); break;
      case 16: hipLaunchKernelGGL((k_cascade<TREE, 16, 8, SEG, TAPS>), grid, block, lds_bytes, st, args); break;
      case 32: hipLaunchKernelGGL((k_cascade<TREE, 32, 8, SEG, TAPS>), grid, block, lds_bytes, st, args); break;
      default: return (int)hipErrorInvalidValue;
    }
  } else {
    return (int)hipErrorInvalidValue;
  }
  HC_LAUNCH_CHECK();
  return 0;
}

template <bool TREE, bool SEG>
static int launch_cascade_taps(hipStream_t st, int nframes, int rows, int nwaves, size_t lds_bytes, const CascadeArgs& args) {
  return args.taps ? launch_cascade_rows<TREE, SEG, true>(st, nframes, rows, nwaves, lds_bytes, args)
                   : launch_cascade_rows<TREE, SEG, false>(st, nframes, rows, nwaves, lds_bytes, args);
}

int launch_cascade(hipStream_t st, int nframes, bool tree, bool segments, int rows, int nwaves, size_t lds_bytes,
                   const CascadeArgs& args) {
  if (args.nbands == 0 || nframes == 0) return 0;
  const int rc_ = tree ? (segments ? launch_cascade_taps<true, true>(st, nframes, rows, nwaves, lds_bytes, args)
                                   : launch_cascade_taps<true, false>(st, nframes, rows, nwaves, lds_bytes, args))
                       : (segments ? launch_cascade_taps<false, true>(st, nframes, rows, nwaves, lds_bytes, args)
                                   : launch_cascade_taps<false, false>(st, nframes, rows, nwaves, lds_bytes, args));
#ifdef HC_PHASE
  dump_phase(st, nwaves);
#endif
  return rc_;
}

}  // namespace hc
