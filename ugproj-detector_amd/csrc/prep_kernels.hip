// prep_kernels.hip -- hand-written gfx950 (CDNA4, wave64) kernels of the PREP half of the Haar-cascade path: everything
// cv::CascadeClassifier::detectMultiScale does per frame before the first weak classifier is evaluated, for the
// reference's call at src/tracker/detector.cc:25-31 (semantics: SURVEY.md Appendix A):
//   k_resize           A.4  exact bilinear pyramid, every level from the original frame
//   k_integral_rows    A.5  row prefix sums (sum, sqsum) -- wave64 DPP scan (row_shr / row_bcast) + register carry
//   k_integral_cols    A.5  column prefix sums, coalesced across lanes
//   k_pyramid_integral A.4 + A.5 (+ A.6a) fused: one workgroup marching down each (level, frame)
//   k_pyramid_strips   the same march for levels of ANY width: a level is cut into column strips, one workgroup each,
//                      the column-integrated carry handed from strip to strip through global memory
//   k_variance         A.6a setWindow: variance normalisation factor + evaluable gate
// Integer work is modulo 2^32; float work is IEEE single / double ops in the upstream order and the file is compiled with
// -ffp-contract=off (upstream x86-64 baseline has no FMA).  No MFMA: there is no dense contraction on this path.

#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "device_types.h"
#include "kernels.h"
#include "device_common.h"

namespace hc {

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
#ifndef HC_VNF_NEWTON
#define HC_VNF_NEWTON 2
#endif
#pragma unroll
    for (int it = 0; it < HC_VNF_NEWTON; it++) {
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

// B.5 (OpenCV 2.4): the f64 factor of a window from the same two sums, no gate (cv24_cascade.hip k_cv24_variance, same expression)
__device__ __forceinline__ double vnf24_from_sums(int valsum, uint32_t valsq, double inv_area) {
  const double mean = (double)valsum * inv_area;
  const double v = (double)valsq * inv_area - mean * mean;
  return v >= 0. ? sqrt(v) : 1.;
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
  // four 64-pixel pieces per trip: their loads are in flight together (one piece per trip means one exposed memory
  // round trip per 64 pixels of the row)
  for (int x0 = 0; x0 < w; x0 += 64 * 4) {
    uint32_t px[4];
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const int x = x0 + 64 * u + lane;
      px[u] = x < w ? src[x] : 0u;
    }
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const int x = x0 + 64 * u + lane;
      if (x0 + 64 * u >= w) break;                         // wave-uniform
      const uint32_t s = wave_scan_u32(px[u]) + carry_s;
      const uint32_t q = wave_scan_u32(px[u] * px[u]) + carry_q;
      if (x < w) { S[x] = s; if (Q) Q[x] = q; }
      carry_s = (uint32_t)__builtin_amdgcn_readlane((int)s, 63);
      carry_q = (uint32_t)__builtin_amdgcn_readlane((int)q, 63);
    }
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
// A wave-uniform base pointer plus a 32-bit BYTE offset per lane: the form the global memory instructions take directly
// (scalar base + 32-bit vector offset), so the access costs no 64-bit address arithmetic on the vector unit.
__device__ __forceinline__ uint32_t ld_u32_at(const uint32_t* base, uint32_t byte_off) {
  return *(const uint32_t*)((const char*)base + byte_off);
}
__device__ __forceinline__ void st_u32_at(uint32_t* base, uint32_t byte_off, uint32_t v) {
  *(uint32_t*)((char*)base + byte_off) = v;
}
__device__ __forceinline__ void st_f32_at(float* base, uint32_t byte_off, float v) {
  *(float*)((char*)base + byte_off) = v;
}

#ifndef HC_PI_THREADS
#define HC_PI_THREADS 512
#endif
constexpr int PI_THREADS = HC_PI_THREADS;   // 8 (16) wavefronts: one pyramid row each per chunk of 8 (16) rows
constexpr int PI_UNROLL = 4;      // consecutive pixels per lane in phase 1 (one 16-byte LDS access per quantity)
static_assert(PI_UNROLL == 4, "phase 1 moves four pixels per lane as one uint4");

// ---- variance factors from row prefixes that are still in LDS (k_pyramid_integral, NH > 0) ----
// The inner rect of a window spans norm_h pixel rows; its sums are  S = sum over those rows of B(r),  B(r) = the
// norm_w-pixel box sum of row r = rp(r, x + norm_w) - rp(r, x)  with rp the row prefix the chunk's phase 1 left in LDS.
// Every thread owns up to two grid columns and slides S down the level: S += B(r) - B(r - norm_h).  The norm_h old
// box sums live in a register ring of VR_RING entries, indexed with compile-time constants (the chunk loop is
// dispatched on y0 mod VR_RING), one packed word per row: B of the pixels (<= 32 * 255 < 2^13) in the low bits and
// 255 * B - Bsq = sum p (255 - p) (<= 32 * 16256 < 2^19) above them.  Nothing is read back from the integrals in
// global memory (the old phase 3 fetched 8 words per window from rows this workgroup had written norm_h rows earlier:
// 8.3 GB per 256 frames, none of it found in L2 any more).
constexpr int VR_RING = 24;       // > norm_h (18 or 22), a multiple of the chunk height 8
constexpr int VR_COLS = 1;        // grid columns per thread: the first PI_THREADS columns of a level (the few beyond them,
                                  // two levels of a 720p ladder, take the read-back route below: 24 registers per column)
struct VarRing {
  uint32_t pk[VR_COLS][VR_RING];
  uint32_t ws[VR_COLS], wq[VR_COLS];
};

template <int NH, int PH, int R0, int SEM>   // PH = (y0 / 8) mod 3: the ring slot of the chunk's row r is PH * 8 + r; rows R0 .. 7
__device__ __forceinline__ void variance_rows(VarRing& V, const uint32_t* rs, const uint32_t* rq, int lds_pitch, int nr, int y0,
                                              int ys, int gnx, int gny, int nw, double area, float* __restrict__ vout) {
  static_assert(NH > 0 && NH < VR_RING, "ring must be longer than the inner rect is tall");
  if constexpr (R0 < 8) {
    if (R0 >= nr) return;                                    // uniform
    const int yt = y0 + R0 - NH;                             // origin row of the windows whose inner rect ends in row y0 + R0
    const int gy = ys == 1 ? yt : (yt >> 1);
    const bool emit = yt >= 0 && (ys == 1 || !(yt & 1)) && gy < gny;   // uniform
    constexpr int slot_w = (PH * 8 + R0) % VR_RING, slot_r = (PH * 8 + R0 + VR_RING - NH) % VR_RING;
#pragma unroll
    for (int k = 0; k < VR_COLS; k++) {
      const int gx = (int)threadIdx.x + k * PI_THREADS;
      if (gx < gnx) {
        const int xo = R0 * lds_pitch + gx * ys;
        const uint32_t bs = rs[xo + nw] - rs[xo], bq = rq[xo + nw] - rq[xo];
        const uint32_t old = V.pk[k][slot_r];
        const uint32_t obs = old & 0x1fffu, obq = 255u * obs - (old >> 13);
        V.ws[k] += bs - obs;
        V.wq[k] += bq - obq;
        V.pk[k][slot_w] = bs | ((255u * bs - bq) << 13);
        if (emit) {
          if (SEM == 1) ((double*)vout)[(size_t)gy * gnx + gx] = vnf24_from_sums((int)V.ws[k], V.wq[k], area);   // area = 1 / area here
          else vout[(size_t)gy * gnx + gx] = vnf_from_sums((int)V.ws[k], V.wq[k], area);
        }
      }
    }
    variance_rows<NH, PH, R0 + 1, SEM>(V, rs, rq, lds_pitch, nr, y0, ys, gnx, gny, nw, area, vout);
  }
}

// SEM 1: the OpenCV 2.4 generation of the same pass (cvResize's 11-bit coefficients and rounding; f64 variance factors without
// a gate, 8 bytes per window) -- same march, same integrals, other numbers.  Its tables pack, per destination column / row,
// source index (13 bits) | first coefficient a0 (12 bits) << 13 | (a0 + a1 - 2047) << 25; the host builds them so that the
// general formula also reproduces 2.4's copy and exact-half special cases (legacy_cv24.cpp).  Ring route only (NH > 0).
template <int CMAX, int NH, int SEM = 0>   // CMAX columns per thread: level width <= PI_THREADS * CMAX; NH: 0, or norm_h with the variance ring
__global__ __launch_bounds__(PI_THREADS, (NH > 0 ? 4 : 1)) void k_pyramid_integral(
    const uint8_t* __restrict__ frames, size_t frame_stride, size_t row_step, int rows, int cols,
    const DevLevel* __restrict__ levels, const uint32_t* __restrict__ xtab, const uint32_t* __restrict__ ytab,
    uint32_t* __restrict__ sum, uint32_t* __restrict__ sq, size_t int_stride, uint8_t* __restrict__ pyr,
    size_t pyr_stride, int R, int lds_pitch, int src_pitch, float* __restrict__ vnf, size_t grid_stride, int norm_w,
    int norm_h, int xr_pitch, int store_sq, int frame_chunk) {
  extern __shared__ uint32_t rowbuf[];   // [2][R][lds_pitch] row prefixes | [lds_pitch] x table | per wave 2 source rows | [NH][xr_pitch] ring words
  uint32_t* rs = rowbuf;
  uint32_t* rq = rowbuf + (size_t)R * lds_pitch;
  uint32_t* xt_s = rowbuf + (size_t)2 * R * lds_pitch;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;   // uniform by construction: row addresses go scalar
  // the two source rows a destination row interpolates between are staged per wavefront with
  // coalesced dword loads; the 4 bytes per pixel are then gathered from LDS instead of through
  // the (much slower) per-lane byte path of the vector memory unit
  uint8_t* sr0 = (uint8_t*)(xt_s + lds_pitch) + (size_t)wave * 2 * src_pitch;
  uint8_t* sr1 = sr0 + src_pitch;
  // frames in chunks (see cascade_body.inc): all levels of a chunk of frames before the next chunk, so that several levels of
  // one frame are in flight together on its XCD and find each other's source rows in L2
  int level = blockIdx.y, frame = blockIdx.x;
  if (frame_chunk > 0 && (int)gridDim.x > frame_chunk) {
    const int n = gridDim.x, nl = gridDim.y, C = frame_chunk;
    const int id = blockIdx.y * n + blockIdx.x;
    const int full = n / C, per = C * nl;
    if (id < full * per) {
      const int chunk = id / per, rem = id - chunk * per;
      level = rem / C;
      frame = chunk * C + (rem - level * C);
    } else {
      const int r = n - full * C, rem = id - full * per;
      level = rem / r;
      frame = full * C + (rem - level * r);
    }
  }
  const HC_CONST DevLevel& L = as_const(levels)[level];
  const int w = L.w, h = L.h, pitch = L.int_pitch;
  const uint8_t* src = frames + (size_t)frame * frame_stride;
  const uint32_t* yt_l = ytab + L.ytab_ofs;
  uint32_t* S = sum + (size_t)frame * int_stride + L.int_ofs;
  const bool squares = sq != nullptr;                                          // false: sum only (mask integral)
  // the squares' integral is stored when somebody reads it: a debug tap, a k_variance launch of a shared pass, or this
  // kernel's own read-back route for grid columns that no ring covers.  With the rings it stays in LDS (a third of the
  // kernel's HBM writes on the bench route).
  const bool xring_on = NH > 0 && xr_pitch > 0;
  uint32_t* Q = squares && (store_sq || (NH > 0 && !xring_on && L.nx > VR_COLS * PI_THREADS) || NH == 0)
                    ? sq + (size_t)frame * int_stride + L.int_ofs : nullptr;
  uint8_t* P = pyr ? pyr + (size_t)frame * pyr_stride + L.pyr_ofs : nullptr;
  const bool dwords = ((((size_t)src) | row_step) & 3) == 0;               // uniform: rows are 4-byte aligned
  uint32_t col_s[CMAX], col_q[CMAX];
#pragma unroll
  for (int k = 0; k < CMAX; k++) col_s[k] = col_q[k] = 0;
  VarRing V;
  if (NH > 0) {
#pragma unroll
    for (int k = 0; k < VR_COLS; k++) {
      V.ws[k] = V.wq[k] = 0;
#pragma unroll
      for (int i = 0; i < VR_RING; i++) V.pk[k][i] = 0;
    }
  }
  uint32_t* const xring = xring_on ? (uint32_t*)((uint8_t*)(xt_s + lds_pitch) + (size_t)(PI_THREADS / 64) * 2 * src_pitch) : nullptr;
  if (xring_on)
    for (int i = threadIdx.x; i < ((NH > 0 ? NH : 1) + 2) * xr_pitch; i += PI_THREADS) xring[i] = 0u;   // NH ring slots + the two running sums
  int ph = 0, xslot = 0;
  for (int x = threadIdx.x; x < lds_pitch; x += PI_THREADS) xt_s[x] = x < w ? xtab[L.xtab_ofs + x] : 0u;
  __syncthreads();
  // The two source rows of a wavefront's NEXT destination row are requested at the end of its phase 1 and wait in
  // registers through phases 2 and 3 (HC_PI_PREFETCH pieces of 256 bytes per row; rows of up to 1280 pixels are covered
  // entirely): the chunk no longer starts with an exposed round trip to L2 / HBM.
#ifndef HC_PI_PREFETCH
#define HC_PI_PREFETCH 5
#endif
  constexpr int NPF = (NH > 0 && CMAX == 2) ? HC_PI_PREFETCH : 0;
  uint32_t pfa[NPF > 0 ? NPF : 1], pfb[NPF > 0 ? NPF : 1];
  int pf_dy = -1;                                                  // destination row the registers hold the sources of
  auto prefetch = [&](int dy) {
    if (NPF == 0 || !dwords || dy >= h) return;
    const uint32_t yt = yt_l[dy];
    const int oy = SEM == 1 ? (int)(yt & 0x1fffu) : (int)(yt >> 9), oy1 = min(oy + 1, rows - 1);
    const uint32_t* g0 = (const uint32_t*)(src + (size_t)oy * row_step);
    const uint32_t* g1 = (const uint32_t*)(src + (size_t)oy1 * row_step);
    const int nd = (cols + 3) >> 2;
#pragma unroll
    for (int u = 0; u < NPF; u++) {
      const int i = lane + 64 * u;
      pfa[u] = i < nd ? g0[i] : 0u;
      pfb[u] = i < nd ? g1[i] : 0u;
    }
    pf_dy = dy;
  };
  prefetch(wave);

  for (int y0 = 0; y0 < h; y0 += R) {
    const int nr = min(R, h - y0);
    for (int r = wave; r < nr; r += PI_THREADS / 64) {
      const int dy = y0 + r;
      const uint32_t yt = yt_l[dy];
      const int oy = SEM == 1 ? (int)(yt & 0x1fffu) : (int)(yt >> 9), oy1 = min(oy + 1, rows - 1);
      // SEM 0: 8.8 weights cy0 + cy1 = 256;  SEM 1: 2.4's 11-bit pair b0, b1 (their sum is 2047 .. 2049)
      const uint32_t cy1 = SEM == 1 ? ((yt >> 13) & 0xfffu) + 0u : (yt & 511u);
      const uint32_t cy0 = SEM == 1 ? 2047u + (yt >> 25) - cy1 : 256u - cy1;
      const uint8_t* g0 = src + (size_t)oy * row_step;
      const uint8_t* g1 = src + (size_t)oy1 * row_step;
      if (dwords) {
        const int nd = (cols + 3) >> 2;                 // src_pitch is a multiple of 4 >= cols; row_step >= cols
        int i_first = lane;
        if (NPF > 0 && pf_dy == dy) {                   // the first NPF pieces were requested a chunk ago
#pragma unroll
          for (int u = 0; u < NPF; u++) {
            const int i = lane + 64 * u;
            if (i < nd) { ((uint32_t*)sr0)[i] = pfa[u]; ((uint32_t*)sr1)[i] = pfb[u]; }
          }
          i_first = lane + 64 * NPF;
        }
        // four pieces of both rows per trip, all eight loads requested before the first LDS store (written as
        // load -> store per piece the compiler waits for every load separately: ten serial round trips per 1280-pixel row)
        for (int i0 = i_first; i0 < nd; i0 += 64 * 4) {
          uint32_t a[4], b[4];
#pragma unroll
          for (int u = 0; u < 4; u++) {
            const int i = i0 + 64 * u;
            a[u] = i < nd ? ((const uint32_t*)g0)[i] : 0u;
            b[u] = i < nd ? ((const uint32_t*)g1)[i] : 0u;
          }
#pragma unroll
          for (int u = 0; u < 4; u++) {
            const int i = i0 + 64 * u;
            if (i < nd) { ((uint32_t*)sr0)[i] = a[u]; ((uint32_t*)sr1)[i] = b[u]; }
          }
        }
      } else {
        for (int i = lane; i < cols; i += 64) { sr0[i] = g0[i]; sr1[i] = g1[i]; }
      }
      // the same wavefront wrote and now reads the staging rows: LDS operations of one wave execute in
      // order, only the compiler has to be kept from moving the reads up
      __builtin_amdgcn_wave_barrier();
      // Lane l owns the four consecutive pixels x0 + 4 l .. + 3 of every 256-pixel piece of the row: their prefix is
      // formed in registers (3 adds, 1 multiply + 3 multiply-adds for the squares), ONE wavefront scan per piece and
      // quantity spreads the lane totals (a scan per 64 pixels before), and the four row prefixes leave as one 16-byte
      // LDS store.
      uint32_t carry_s = 0, carry_q = 0;
      uint32_t* const rs_r = rs + r * lds_pitch;
      uint32_t* const rq_r = rq + r * lds_pitch;
      for (int x0 = 0; x0 < w; x0 += 64 * PI_UNROLL) {
        const int x = x0 + 4 * lane;
        uint32_t px[PI_UNROLL];
        if (x < lds_pitch) {                                // lds_pitch is a multiple of 64 >= w: whole quads
          const uint4 xq = *(const uint4*)(xt_s + x);
          const uint32_t xts[4] = {xq.x, xq.y, xq.z, xq.w};
#pragma unroll
          for (int u = 0; u < PI_UNROLL; u++) {
            const uint32_t xt = xts[u];
            if (SEM == 1) {
              // cvResize(CV_INTER_LINEAR) of 2.4: (((b0 * (h0 >> 4)) >> 16) + ((b1 * (h1 >> 4)) >> 16) + 2) >> 2 with
              // h = p[sx] * a0 + p[sx + 1] * a1; here cy1 holds b0 and cy0 holds b1 (decoded above in that order)
              const int ox = xt & 0x1fffu, ox1 = min(ox + 1, cols - 1);
              const uint32_t a0 = (xt >> 13) & 0xfffu, a1 = 2047u + (xt >> 25) - a0;
              const uint32_t h0 = sr0[ox] * a0 + sr0[ox1] * a1;
              const uint32_t h1 = sr1[ox] * a0 + sr1[ox1] * a1;
              px[u] = x + u < w ? (((cy1 * (h0 >> 4)) >> 16) + ((cy0 * (h1 >> 4)) >> 16) + 2u) >> 2 : 0u;
            } else {
              const int ox = xt >> 9, ox1 = min(ox + 1, cols - 1);
              const uint32_t cx1 = xt & 511u, cx0 = 256u - cx1;
              const uint32_t h0 = cx0 * sr0[ox] + cx1 * sr0[ox1];
              const uint32_t h1 = cx0 * sr1[ox] + cx1 * sr1[ox1];
              px[u] = x + u < w ? (h0 * cy0 + h1 * cy1 + 32768u) >> 16 : 0u;   // weights sum to 2^16: the value is <= 255
            }
          }
        } else {
#pragma unroll
          for (int u = 0; u < PI_UNROLL; u++) px[u] = 0u;
        }
        if (P) {
#pragma unroll
          for (int u = 0; u < PI_UNROLL; u++)
            if (x + u < w) P[(size_t)dy * w + x + u] = (uint8_t)px[u];
        }
        uint32_t ps[PI_UNROLL], pq[PI_UNROLL];
        ps[0] = px[0]; pq[0] = px[0] * px[0];
#pragma unroll
        for (int u = 1; u < PI_UNROLL; u++) { ps[u] = ps[u - 1] + px[u]; pq[u] = pq[u - 1] + px[u] * px[u]; }
        const uint32_t ts = wave_scan_u32(ps[PI_UNROLL - 1]), tq = wave_scan_u32(pq[PI_UNROLL - 1]);   // inclusive over lane totals
        const uint32_t bs = ts - ps[PI_UNROLL - 1] + carry_s, bq = tq - pq[PI_UNROLL - 1] + carry_q;    // what precedes this lane
        if (x < lds_pitch) {
          *(uint4*)(rs_r + x) = make_uint4(ps[0] + bs, ps[1] + bs, ps[2] + bs, ps[3] + bs);
          *(uint4*)(rq_r + x) = make_uint4(pq[0] + bq, pq[1] + bq, pq[2] + bq, pq[3] + bq);
        }
        carry_s += (uint32_t)__builtin_amdgcn_readlane((int)ts, 63);
        carry_q += (uint32_t)__builtin_amdgcn_readlane((int)tq, 63);
      }
      if (r + PI_THREADS / 64 >= nr) prefetch(dy + R);       // this wavefront's row of the next chunk (R rows per chunk, one per wavefront)
    }
    __syncthreads();
    {
      // row bases are wave-uniform pointers, the column a 32-bit lane offset: the stores need no 64-bit address arithmetic
      uint32_t* const so = S + (size_t)(y0 + 1) * pitch + 1;
      uint32_t* const qo = Q ? Q + (size_t)(y0 + 1) * pitch + 1 : nullptr;
#pragma unroll
      for (int k = 0; k < CMAX; k++) {
        const uint32_t x = threadIdx.x + k * PI_THREADS;
        if ((int)x < w) {
          uint32_t cs = col_s[k], cq = col_q[k];
          int r = 0;
          for (; r + 4 <= nr; r += 4) {                     // four rows' prefixes requested together
            uint32_t a[4], b[4];
#pragma unroll
            for (int u = 0; u < 4; u++) { a[u] = rs[(r + u) * lds_pitch + x]; b[u] = qo ? rq[(r + u) * lds_pitch + x] : 0u; }   // squares only when stored
#pragma unroll
            for (int u = 0; u < 4; u++) {
              cs += a[u]; cq += b[u];
              st_u32_at(so + (size_t)(r + u) * pitch, x * 4u, cs);
              if (qo) st_u32_at(qo + (size_t)(r + u) * pitch, x * 4u, cq);
            }
          }
          for (; r < nr; r++) {
            cs += rs[r * lds_pitch + x];
            st_u32_at(so + (size_t)r * pitch, x * 4u, cs);
            if (qo) {
              cq += rq[r * lds_pitch + x];
              st_u32_at(qo + (size_t)r * pitch, x * 4u, cq);
            }
          }
          col_s[k] = cs; col_q[k] = cq;
        }
      }
    }
    if (NH > 0) {
      // A.6a fused in, from the row prefixes of this chunk (see variance_rows); R == 8 on this route
      if (vnf && L.nx > 0) {
        // SEM 1: 8-byte factors (the buffer is the context's vnf64), and the helper wants 1 / area
        float* vout = SEM == 1 ? (float*)((double*)vnf + (size_t)frame * grid_stride + L.grid_ofs) : vnf + (size_t)frame * grid_stride + L.grid_ofs;
        const double area = SEM == 1 ? 1. / (double)(norm_w * norm_h) : (double)(norm_w * norm_h);
        if (ph == 0) variance_rows<NH ? NH : 1, 0, 0, SEM>(V, rs, rq, lds_pitch, nr, y0, L.ystep, L.nx, L.ny, norm_w, area, vout);
        else if (ph == 1) variance_rows<NH ? NH : 1, 1, 0, SEM>(V, rs, rq, lds_pitch, nr, y0, L.ystep, L.nx, L.ny, norm_w, area, vout);
        else variance_rows<NH ? NH : 1, 2, 0, SEM>(V, rs, rq, lds_pitch, nr, y0, L.ystep, L.nx, L.ny, norm_w, area, vout);
        // Grid columns beyond VR_COLS * PI_THREADS (two levels of a 720p ladder: 66 and 12 columns): the same slide with
        // the ring words in LDS, exactly NH slots per column (slot = row mod NH: the word read is the one replaced).
        // A plain loop on the first wavefronts only.
        const int gx = (int)threadIdx.x + VR_COLS * PI_THREADS;
        if (xring_on && gx < L.nx) {
          const int ys = L.ystep, gnx = L.nx, gny = L.ny;
          uint32_t* const acc = xring + (NH ? NH : 1) * xr_pitch + (int)threadIdx.x;
          uint32_t xs = acc[0], xq = acc[xr_pitch];
          int slot = xslot;
#pragma unroll 1
          for (int r = 0; r < nr; r++) {
            const int xo = r * lds_pitch + gx * ys;
            const uint32_t bs = rs[xo + norm_w] - rs[xo], bq = rq[xo + norm_w] - rq[xo];
            uint32_t* const sp = xring + slot * xr_pitch + (int)threadIdx.x;
            const uint32_t old = *sp;
            const uint32_t obs = old & 0x1fffu, obq = 255u * obs - (old >> 13);
            xs += bs - obs;
            xq += bq - obq;
            *sp = bs | ((255u * bs - bq) << 13);
            const int yt = y0 + r - (NH ? NH : 1);
            const int gy = ys == 1 ? yt : (yt >> 1);
            if (yt >= 0 && (ys == 1 || !(yt & 1)) && gy < gny) {
              if (SEM == 1) ((double*)vout)[(size_t)gy * gnx + gx] = vnf24_from_sums((int)xs, xq, area);
              else vout[(size_t)gy * gnx + gx] = vnf_from_sums((int)xs, xq, area);
            }
            slot = slot + 1 == (NH ? NH : 1) ? 0 : slot + 1;
          }
          acc[0] = xs; acc[xr_pitch] = xq;
        }
      }
      ph = ph == 2 ? 0 : ph + 1;
      xslot = (xslot + nr) % (NH ? NH : 1);
    }
    __syncthreads();
    // Phase 3 (A.6a fused in): the variance factor of every window whose inner rect ENDS in this chunk.  Its 8 integral
    // words are this workgroup's own stores -- the bottom row from this chunk, the top row norm_h rows up -- drained by
    // the barrier above (s_waitcnt vmcnt(0) precedes it) and never read by this CU before their final value was
    // written, so they come back from L2 (or L1) current; no other kernel has to re-read the integrals for this.
    // Uses no LDS: the wavefronts that finish first go on to the next chunk's phase 1.
    if (vnf && Q && !xring_on && L.nx > (NH > 0 ? VR_COLS * PI_THREADS : 0)) {   // NH > 0: only the grid columns beyond the ring's
      const int ys = L.ystep, gnx = L.nx, gny = L.ny;
      // window row y = gy*ys has its bottom integral row at y + 1 + norm_h; rows y0+1 .. y0+nr were finished just now
      int gy_lo = max(y0 - norm_h, 0);
      gy_lo = (gy_lo + ys - 1) / ys;
      const int gy_hi = min((y0 + nr - 1 - norm_h) >= 0 ? (y0 + nr - 1 - norm_h) / ys : -1, gny - 1);
      const double area = (double)(norm_w * norm_h);
      float* vout = vnf + (size_t)frame * grid_stride + L.grid_ofs;
      // Four grid rows per trip: all 32 integral words of a thread's four windows are requested before the first is
      // used, so a chunk pays the L2 round trip once instead of once per window (the march down the largest level is
      // the kernel's critical path).  Row bases are wave-uniform pointers and the column a 32-bit lane offset: no
      // division, no 64-bit address arithmetic per window.
      constexpr int PV = 4;
      for (int g0 = gy_lo; g0 <= gy_hi; g0 += PV) {
        const uint32_t *st[PV], *sb[PV], *qt[PV], *qb[PV];
        float* vo[PV];
#pragma unroll
        for (int u = 0; u < PV; u++) {
          const int gy = min(g0 + u, gy_hi);                   // clamped: a surplus row recomputes the last one
          const size_t top = (size_t)(gy * ys + 1) * pitch + 1;
          st[u] = S + top; sb[u] = st[u] + (size_t)norm_h * pitch;
          qt[u] = Q + top; qb[u] = qt[u] + (size_t)norm_h * pitch;
          vo[u] = vout + (size_t)gy * gnx;
        }
        for (uint32_t gx = threadIdx.x + (NH > 0 ? VR_COLS * PI_THREADS : 0); (int)gx < gnx; gx += PI_THREADS) {
          const uint32_t lo = gx * (uint32_t)ys * 4u, hi = lo + (uint32_t)norm_w * 4u;   // byte offsets
          uint32_t s00[PV], s01[PV], s10[PV], s11[PV], q00[PV], q01[PV], q10[PV], q11[PV];
#pragma unroll
          for (int u = 0; u < PV; u++) {
            s00[u] = ld_u32_at(st[u], lo); s01[u] = ld_u32_at(st[u], hi); s10[u] = ld_u32_at(sb[u], lo); s11[u] = ld_u32_at(sb[u], hi);
            q00[u] = ld_u32_at(qt[u], lo); q01[u] = ld_u32_at(qt[u], hi); q10[u] = ld_u32_at(qb[u], lo); q11[u] = ld_u32_at(qb[u], hi);
          }
#pragma unroll
          for (int u = 0; u < PV; u++) {
            const int valsum = (int)(s00[u] - s01[u] - s10[u] + s11[u]);
            const uint32_t valsq = q00[u] - q01[u] - q10[u] + q11[u];
            const float v = vnf_from_sums(valsum, valsq, area);
            if (g0 + u <= gy_hi) st_f32_at(vo[u], gx * 4u, v);
          }
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
// A.4 + A.5 + A.6a fused for levels of ANY width: k_pyramid_integral's march, one workgroup per column STRIP of a level
// (device_types.h DevStrip).  What a strip needs from outside itself is one number per row and quantity: the finished
// integral value of the column just left of it, I(y, c0 - 1) -- every integral word of the strip is its own local
// two-dimensional prefix plus that.  The left neighbour publishes it at the end of its phase 2 (plain stores, then a
// release store of the row count); the strip fetches the chunk's eight rows at the end of its phase 1 (acquire poll by
// one wavefront, values into LDS) and adds them as it stores.  Row prefixes, box sums and the variance rings are local
// differences and never see the carry.  Workgroup i waits only on a workgroup with a smaller linear id (the strip list
// runs left to right, the frame-chunk remap keeps the order), and every XCD starts its workgroups in id order: the
// smallest unfinished workgroup never waits, so the grid drains.  A wait that exceeds its bound (never seen; it would
// mean that premise is false) sets an error word that the batch's publication carries to the host, and goes on.
// What the strips buy: rows of eight per chunk and two workgroups per CU at any frame size (the one-workgroup-per-level
// form staged 2 x 8 whole source rows and 17 whole level rows in LDS: 3840x2160 left it two rows per chunk), and every
// grid column on a register ring (at most PI_THREADS per strip), so no level falls back to k_variance or to the
// three parallel kernels for its width.
template <int NH, int PH, int R0, int SEM>
__device__ __forceinline__ void strip_variance_rows_ph(VarRing& V, const uint32_t* rs, const uint32_t* rq, int lds_pitch, int nr, int y0,
                                                       int ys, int gnx, int gny, int nw, double area, float* __restrict__ vout,
                                                       bool active, int lx, int gcol) {
  if constexpr (R0 < 8) {
    if (R0 >= nr) return;                                    // uniform
    const int yt = y0 + R0 - NH;                             // origin row of the windows whose inner rect ends in row y0 + R0
    const int gy = ys == 1 ? yt : (yt >> 1);
    const bool emit = yt >= 0 && (ys == 1 || !(yt & 1)) && gy < gny;   // uniform
    constexpr int slot_w = (PH * 8 + R0) % VR_RING, slot_r = (PH * 8 + R0 + VR_RING - NH) % VR_RING;
    if (active) {
      const int xo = R0 * lds_pitch + lx;
      const uint32_t bs = rs[xo + nw] - rs[xo], bq = rq[xo + nw] - rq[xo];
      const uint32_t old = V.pk[0][slot_r];
      const uint32_t obs = old & 0x1fffu, obq = 255u * obs - (old >> 13);
      V.ws[0] += bs - obs;
      V.wq[0] += bq - obq;
      V.pk[0][slot_w] = bs | ((255u * bs - bq) << 13);
      if (emit) {
        if (SEM == 1) ((double*)vout)[(size_t)gy * gnx + gcol] = vnf24_from_sums((int)V.ws[0], V.wq[0], area);   // area = 1 / area here
        else vout[(size_t)gy * gnx + gcol] = vnf_from_sums((int)V.ws[0], V.wq[0], area);
      }
    }
    strip_variance_rows_ph<NH, PH, R0 + 1, SEM>(V, rs, rq, lds_pitch, nr, y0, ys, gnx, gny, nw, area, vout, active, lx, gcol);
  }
}

constexpr unsigned int STRIP_SPIN_LIMIT = 1u << 24;   // polls of ~0.1 us: seconds.  Reaching it is a bug, not a slow neighbour

template <int NH, int SEM>
__global__ __launch_bounds__(PI_THREADS, 4) void k_pyramid_strips(
    const uint8_t* __restrict__ frames, size_t frame_stride, size_t row_step, int rows, int cols,
    const DevLevel* __restrict__ levels, const DevStrip* __restrict__ strips, const uint32_t* __restrict__ xtab,
    const uint32_t* __restrict__ ytab, uint32_t* __restrict__ sum, uint32_t* __restrict__ sq, size_t int_stride,
    uint8_t* __restrict__ pyr, size_t pyr_stride, int lds_pitch, int src_pitch, float* __restrict__ vnf, size_t grid_stride,
    int norm_w, int norm_h, int store_sq, int frame_chunk, uint32_t* carry, size_t carry_stride, unsigned int* progress,
    unsigned int* err_word) {
  static_assert(PI_THREADS == 512 && NH > 0, "eight wavefronts, one row each per chunk; ring route only");
  constexpr int R = PI_THREADS / 64;
  extern __shared__ uint32_t rowbuf[];   // [2][R][lds_pitch] row prefixes | [lds_pitch] x table | [2 R] carries of the chunk | per wave 2 source spans
  uint32_t* rs = rowbuf;
  uint32_t* rq = rowbuf + (size_t)R * lds_pitch;
  uint32_t* xt_s = rowbuf + (size_t)2 * R * lds_pitch;
  uint32_t* cum = xt_s + lds_pitch;      // cum[2 r], cum[2 r + 1]: I(y0 + r, c0 - 1) of sum / sqsum
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
  uint8_t* sr0 = (uint8_t*)(cum + 2 * R) + (size_t)wave * 2 * src_pitch;
  uint8_t* sr1 = sr0 + src_pitch;
  // frames in chunks, every strip of a chunk of frames before the next chunk (see k_pyramid_integral); strip index
  // nondecreasing in the linear workgroup id inside a chunk, chunks in order: a left neighbour always has the smaller id
  int si = blockIdx.y, frame = blockIdx.x;
  if (frame_chunk > 0 && (int)gridDim.x > frame_chunk) {
    const int n = gridDim.x, nl = gridDim.y, C = frame_chunk;
    const int id = blockIdx.y * n + blockIdx.x;
    const int full = n / C, per = C * nl;
    if (id < full * per) {
      const int chunk = id / per, rem = id - chunk * per;
      si = rem / C;
      frame = chunk * C + (rem - si * C);
    } else {
      const int r = n - full * C, rem = id - full * per;
      si = rem / r;
      frame = full * C + (rem - si * r);
    }
  }
  const int nstrips = gridDim.y;
  const HC_CONST DevStrip& T = as_const(strips)[si];
  const HC_CONST DevLevel& L = as_const(levels)[T.level];
  const int w = L.w, h = L.h, pitch = L.int_pitch;
  const int c0 = T.c0, nst = T.c1 - T.c0, ncomp = T.ce - T.c0, sx0 = T.sx0, sxn = T.sxn;
  const uint8_t* src = frames + (size_t)frame * frame_stride;
  const uint32_t* yt_l = ytab + L.ytab_ofs;
  uint32_t* S = sum + (size_t)frame * int_stride + L.int_ofs;
  uint32_t* Q = (sq != nullptr && store_sq) ? sq + (size_t)frame * int_stride + L.int_ofs : nullptr;
  uint8_t* P = pyr ? pyr + (size_t)frame * pyr_stride + L.pyr_ofs : nullptr;
  const bool dwords = ((((size_t)src) | row_step) & 3) == 0;               // uniform: rows are 4-byte aligned (sx0 is)
  uint32_t col_s[2] = {0u, 0u}, col_q[2] = {0u, 0u};
  VarRing V;
  V.ws[0] = V.wq[0] = 0;
#pragma unroll
  for (int i = 0; i < VR_RING; i++) V.pk[0][i] = 0;
  int ph = 0;
  for (int x = threadIdx.x; x < lds_pitch; x += PI_THREADS) xt_s[x] = x < ncomp ? xtab[L.xtab_ofs + c0 + x] : 0u;
  if (threadIdx.x < 2 * R) cum[threadIdx.x] = 0u;
  __syncthreads();
  const unsigned int* left_progress = T.left >= 0 ? progress + (size_t)frame * nstrips + T.left : nullptr;
  // one 8-byte word per row: {sum, sqsum} of the finished integral at the strip's last stored column
  const unsigned long long* left_carry =
      T.left >= 0 ? (const unsigned long long*)(carry + (size_t)frame * carry_stride + as_const(strips)[T.left].carry_ofs) : nullptr;
  unsigned long long* my_carry = T.has_right ? (unsigned long long*)(carry + (size_t)frame * carry_stride + T.carry_ofs) : nullptr;
  unsigned int* my_progress = progress + (size_t)frame * nstrips + si;
  // the thread that owns the strip's last stored column publishes (phase 2)
  const int pub_k = (nst - 1) / PI_THREADS, pub_t = (nst - 1) - pub_k * PI_THREADS;

  constexpr int NPF = HC_PI_PREFETCH;
  uint32_t pfa[NPF], pfb[NPF];
  int pf_dy = -1;                                                  // destination row the registers hold the sources of
  const int nd = sxn >> 2;
  auto prefetch = [&](int dy) {
    if (!dwords || dy >= h) return;
    const uint32_t yt = yt_l[dy];
    const int oy = SEM == 1 ? (int)(yt & 0x1fffu) : (int)(yt >> 9), oy1 = min(oy + 1, rows - 1);
    const uint32_t* g0 = (const uint32_t*)(src + (size_t)oy * row_step + sx0);
    const uint32_t* g1 = (const uint32_t*)(src + (size_t)oy1 * row_step + sx0);
#pragma unroll
    for (int u = 0; u < NPF; u++) {
      const int i = lane + 64 * u;
      pfa[u] = i < nd ? g0[i] : 0u;
      pfb[u] = i < nd ? g1[i] : 0u;
    }
    pf_dy = dy;
  };
  prefetch(wave);

  for (int y0 = 0; y0 < h; y0 += R) {
    const int nr = min(R, h - y0);
    if (wave < nr) {
      const int r = wave, dy = y0 + r;
      const uint32_t yt = yt_l[dy];
      const int oy = SEM == 1 ? (int)(yt & 0x1fffu) : (int)(yt >> 9), oy1 = min(oy + 1, rows - 1);
      const uint32_t cy1 = SEM == 1 ? ((yt >> 13) & 0xfffu) + 0u : (yt & 511u);
      const uint32_t cy0 = SEM == 1 ? 2047u + (yt >> 25) - cy1 : 256u - cy1;
      const uint8_t* g0 = src + (size_t)oy * row_step + sx0;
      const uint8_t* g1 = src + (size_t)oy1 * row_step + sx0;
      if (dwords) {
        int i_first = lane;
        if (pf_dy == dy) {
#pragma unroll
          for (int u = 0; u < NPF; u++) {
            const int i = lane + 64 * u;
            if (i < nd) { ((uint32_t*)sr0)[i] = pfa[u]; ((uint32_t*)sr1)[i] = pfb[u]; }
          }
          i_first = lane + 64 * NPF;
        }
        for (int i0 = i_first; i0 < nd; i0 += 64 * 4) {
          uint32_t a[4], b[4];
#pragma unroll
          for (int u = 0; u < 4; u++) {
            const int i = i0 + 64 * u;
            a[u] = i < nd ? ((const uint32_t*)g0)[i] : 0u;
            b[u] = i < nd ? ((const uint32_t*)g1)[i] : 0u;
          }
#pragma unroll
          for (int u = 0; u < 4; u++) {
            const int i = i0 + 64 * u;
            if (i < nd) { ((uint32_t*)sr0)[i] = a[u]; ((uint32_t*)sr1)[i] = b[u]; }
          }
        }
      } else {
        for (int i = lane; i < sxn && sx0 + i < cols; i += 64) { sr0[i] = g0[i]; sr1[i] = g1[i]; }
      }
      __builtin_amdgcn_wave_barrier();
      uint32_t carry_s = 0, carry_q = 0;                      // strip-local row prefix: the row's own left part never enters
      uint32_t* const rs_r = rs + r * lds_pitch;
      uint32_t* const rq_r = rq + r * lds_pitch;
      for (int x0 = 0; x0 < ncomp; x0 += 64 * PI_UNROLL) {
        const int x = x0 + 4 * lane;
        uint32_t px[PI_UNROLL];
        if (x < lds_pitch) {
          const uint4 xq = *(const uint4*)(xt_s + x);
          const uint32_t xts[4] = {xq.x, xq.y, xq.z, xq.w};
#pragma unroll
          for (int u = 0; u < PI_UNROLL; u++) {
            const uint32_t xt = xts[u];
            if (SEM == 1) {
              const int oxa = xt & 0x1fffu, ox = oxa - sx0, ox1 = min(oxa + 1, cols - 1) - sx0;
              const uint32_t a0 = (xt >> 13) & 0xfffu, a1 = 2047u + (xt >> 25) - a0;
              const uint32_t h0 = sr0[ox] * a0 + sr0[ox1] * a1;
              const uint32_t h1 = sr1[ox] * a0 + sr1[ox1] * a1;
              px[u] = x + u < ncomp ? (((cy1 * (h0 >> 4)) >> 16) + ((cy0 * (h1 >> 4)) >> 16) + 2u) >> 2 : 0u;
            } else {
              const int oxa = xt >> 9, ox = oxa - sx0, ox1 = min(oxa + 1, cols - 1) - sx0;
              const uint32_t cx1 = xt & 511u, cx0 = 256u - cx1;
              const uint32_t h0 = cx0 * sr0[ox] + cx1 * sr0[ox1];
              const uint32_t h1 = cx0 * sr1[ox] + cx1 * sr1[ox1];
              px[u] = x + u < ncomp ? (h0 * cy0 + h1 * cy1 + 32768u) >> 16 : 0u;
            }
          }
        } else {
#pragma unroll
          for (int u = 0; u < PI_UNROLL; u++) px[u] = 0u;
        }
        if (P) {
#pragma unroll
          for (int u = 0; u < PI_UNROLL; u++)
            if (x + u < nst) P[(size_t)dy * w + c0 + x + u] = (uint8_t)px[u];
        }
        uint32_t ps[PI_UNROLL], pq[PI_UNROLL];
        ps[0] = px[0]; pq[0] = px[0] * px[0];
#pragma unroll
        for (int u = 1; u < PI_UNROLL; u++) { ps[u] = ps[u - 1] + px[u]; pq[u] = pq[u - 1] + px[u] * px[u]; }
        const uint32_t ts = wave_scan_u32(ps[PI_UNROLL - 1]), tq = wave_scan_u32(pq[PI_UNROLL - 1]);
        const uint32_t bs = ts - ps[PI_UNROLL - 1] + carry_s, bq = tq - pq[PI_UNROLL - 1] + carry_q;
        if (x < lds_pitch) {
          *(uint4*)(rs_r + x) = make_uint4(ps[0] + bs, ps[1] + bs, ps[2] + bs, ps[3] + bs);
          *(uint4*)(rq_r + x) = make_uint4(pq[0] + bq, pq[1] + bq, pq[2] + bq, pq[3] + bq);
        }
        carry_s += (uint32_t)__builtin_amdgcn_readlane((int)ts, 63);
        carry_q += (uint32_t)__builtin_amdgcn_readlane((int)tq, 63);
      }
      prefetch(dy + R);                                      // this wavefront's row of the next chunk
    }
    if (left_carry && wave == R - 1) {
      // The chunk's carries: rows y0 .. y0 + nr - 1 of the left neighbour's last column.  The hand-off is the measured
      // sc1 form (MI355X_MICROARCH.md, hand-offs with sc1 loads in place of the acquire): the producer stores every
      // carry word write-through (relaxed agent-scope atomic stores = sc1), waits vmcnt(0), then one lane stores the row
      // count sc1; here one lane polls the count with sc1 loads and, only after the poll has matched, this same
      // wavefront fetches the words with sc1 loads into LDS; the other wavefronts read them behind the barrier below.
      // No release / acquire fence on either side (an agent release writes back the XCD's L2: ~1.7 us, a chunk takes less).
      // The last wavefront does it: when the level's height is not a multiple of eight it is idle in the last chunk,
      // and everywhere it is the one whose row starts last.
      const unsigned int need = (unsigned int)(y0 + nr);
      if (lane == 0) {
        unsigned int spins = 0;
        while (__hip_atomic_load(left_progress, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < need) {
          __builtin_amdgcn_s_sleep(8);
          if (++spins > STRIP_SPIN_LIMIT) { __hip_atomic_store(err_word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
        }
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the matching poll has returned; nothing below moves above it
      __builtin_amdgcn_wave_barrier();
      if (lane < nr) {
        const unsigned long long cw = __hip_atomic_load(left_carry + (y0 + lane), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        cum[2 * lane] = (uint32_t)cw;
        cum[2 * lane + 1] = (uint32_t)(cw >> 32);
      }
    }
    __syncthreads();
    {
      uint32_t* const so = S + (size_t)(y0 + 1) * pitch + 1 + c0;
      uint32_t* const qo = Q ? Q + (size_t)(y0 + 1) * pitch + 1 + c0 : nullptr;
#pragma unroll
      for (int k = 0; k < 2; k++) {
        const uint32_t x = threadIdx.x + k * PI_THREADS;
        if ((int)x < nst) {
          uint32_t cs = col_s[k], cq = col_q[k];
          const bool pub = my_carry != nullptr && k == pub_k && (int)threadIdx.x == pub_t;
          int r = 0;
          for (; r + 4 <= nr; r += 4) {
            uint32_t a[4], b[4];
            const uint4 c01 = *(const uint4*)(cum + 2 * r), c23 = *(const uint4*)(cum + 2 * r + 4);
            const uint32_t cu[8] = {c01.x, c01.y, c01.z, c01.w, c23.x, c23.y, c23.z, c23.w};
#pragma unroll
            for (int u = 0; u < 4; u++) { a[u] = rs[(r + u) * lds_pitch + x]; b[u] = rq[(r + u) * lds_pitch + x]; }
#pragma unroll
            for (int u = 0; u < 4; u++) {
              cs += a[u]; cq += b[u];
              st_u32_at(so + (size_t)(r + u) * pitch, x * 4u, cs + cu[2 * u]);
              if (qo) st_u32_at(qo + (size_t)(r + u) * pitch, x * 4u, cq + cu[2 * u + 1]);
              if (pub) __hip_atomic_store(my_carry + (y0 + r + u), (unsigned long long)(cs + cu[2 * u]) | ((unsigned long long)(cq + cu[2 * u + 1]) << 32),
                                          __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
          }
          for (; r < nr; r++) {
            cs += rs[r * lds_pitch + x]; cq += rq[r * lds_pitch + x];
            st_u32_at(so + (size_t)r * pitch, x * 4u, cs + cum[2 * r]);
            if (qo) st_u32_at(qo + (size_t)r * pitch, x * 4u, cq + cum[2 * r + 1]);
            if (pub) __hip_atomic_store(my_carry + (y0 + r), (unsigned long long)(cs + cum[2 * r]) | ((unsigned long long)(cq + cum[2 * r + 1]) << 32),
                                        __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
          col_s[k] = cs; col_q[k] = cq;
          if (pub) {    // the carry words have left this wavefront before the count that announces them is stored
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_store(my_progress, (unsigned int)(y0 + nr), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
        }
      }
    }
    if (vnf && T.g1 > T.g0) {
      float* vout = SEM == 1 ? (float*)((double*)vnf + (size_t)frame * grid_stride + L.grid_ofs) : vnf + (size_t)frame * grid_stride + L.grid_ofs;
      const double area = SEM == 1 ? 1. / (double)(norm_w * norm_h) : (double)(norm_w * norm_h);
      const int gcol = T.g0 + (int)threadIdx.x;
      const bool active = gcol < T.g1;
      const int lx = gcol * L.ystep - c0;
      if (ph == 0) strip_variance_rows_ph<NH, 0, 0, SEM>(V, rs, rq, lds_pitch, nr, y0, L.ystep, L.nx, L.ny, norm_w, area, vout, active, lx, gcol);
      else if (ph == 1) strip_variance_rows_ph<NH, 1, 0, SEM>(V, rs, rq, lds_pitch, nr, y0, L.ystep, L.nx, L.ny, norm_w, area, vout, active, lx, gcol);
      else strip_variance_rows_ph<NH, 2, 0, SEM>(V, rs, rq, lds_pitch, nr, y0, L.ystep, L.nx, L.ny, norm_w, area, vout, active, lx, gcol);
    }
    ph = ph == 2 ? 0 : ph + 1;
    __syncthreads();
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

// Whether the fused kernel would produce the variance factors from its register rings for this geometry (levels of at most
// 1024 columns, eight rows per chunk at two workgroups per CU, a stock inner rect).  Where it cannot, its read-back
// route costs more than a separate k_variance launch (1920x1080, 96 frames: 4.54 ms against 2.5 + 1.27), so the caller
// then leaves the factors to k_variance.
bool pyramid_integral_has_ring(const PrepKnobs& knobs, int max_w, int cols, int norm_w, int norm_h) {
  if (PI_THREADS != 512 || max_w > 2 * PI_THREADS || norm_w > 32 || (norm_h != 18 && norm_h != 22)) return false;
  if (!knobs.var_ring) return false;
  const int src_pitch = (cols + 3 + 3) & ~3;
  const size_t stage = (size_t)(PI_THREADS / 64) * 2 * src_pitch;
  const int lds_pitch = (max_w + 63) & ~63;
  return (size_t)(2 * (PI_THREADS / 64) + 1) * lds_pitch * 4 + stage <= 81920u;
}

// Whether launch_pyramid_integral can run the OpenCV 2.4 generation for this geometry (ring route, every grid column on a ring,
// two workgroups per CU): mirrors the decisions below.
bool pyramid_integral_takes_cv24(const PrepKnobs& knobs, int max_w, int max_nx, int cols, int norm_w, int norm_h) {
  if (PI_THREADS != 512 || max_w > 2 * PI_THREADS || norm_w > 32 || (norm_h != 18 && norm_h != 22)) return false;
  if (!knobs.var_ring) return false;
  const int src_pitch = (cols + 3 + 3) & ~3;
  const size_t stage = (size_t)(PI_THREADS / 64) * 2 * src_pitch;
  const int R = PI_THREADS / 64;
  const int lds_pitch = (max_w + 63) & ~63, pitch4 = (max_w + 3) & ~3;
  if ((size_t)(2 * R + 1) * lds_pitch * 4 + stage > 81920u) return false;
  const int extra = max_nx - VR_COLS * PI_THREADS;
  if (extra <= 0) return true;
  if (!knobs.var_xring || extra > PI_THREADS) return false;
  const size_t ring_bytes = (size_t)(norm_h + 2) * ((extra + 3) & ~3) * 4;
  return (size_t)(2 * R + 1) * lds_pitch * 4 + stage + ring_bytes <= 81920u || (size_t)(2 * R + 1) * pitch4 * 4 + stage + ring_bytes <= 81920u;
}

int launch_pyramid_integral(hipStream_t st, int nframes, int nlevels, int max_w, const uint8_t* frames,
                            size_t frame_stride, size_t row_step, int rows, int cols, const DevLevel* levels,
                            const uint32_t* xtab, const uint32_t* ytab, uint32_t* sum, uint32_t* sq, size_t int_stride,
                            uint8_t* pyr, size_t pyr_stride, const PrepKnobs& knobs, float* vnf, size_t grid_stride, int norm_w,
                            int norm_h, int max_nx, bool store_sq, PrepLaunchInfo* info, int semantics) {
  if (nlevels == 0 || nframes == 0) return 0;
  int lds_pitch = (max_w + 63) & ~63;
  const int src_pitch = (cols + 3 + 3) & ~3;     // staged source row: dword granular, >= cols
  const size_t stage = (size_t)(PI_THREADS / 64) * 2 * src_pitch;
  int R = PI_THREADS / 64;                       // rows per chunk: (2 R + 1) x pitch words + staging rows of LDS
  while (R > 1 && (size_t)(2 * R + 1) * lds_pitch * 4 + stage > 81920u) R >>= 1;      // 2 workgroups per CU
  if (R < 4) {                                                                          // very wide frames: 1 per CU
    R = PI_THREADS / 64;
    while (R > 1 && (size_t)(2 * R + 1) * lds_pitch * 4 + stage > 147456u) R >>= 1;
  }
  // 32 frames per chunk: 512 resident workgroups = 16 levels of 32 frames (HC_PI_FRAME_CHUNK; 0 = level-major over the whole batch).
  // Source-row fetch per 512 frames 7.18 -> 4.14 GB at the same kernel time (16: 3.71 GB, +2.7 % time; profiles/r03_experiments.txt).
  const int frame_chunk = knobs.frame_chunk;      // per context (hc::Knobs), never a process-wide static
  const dim3 grid(nframes, nlevels), block(PI_THREADS);
  size_t lds = (size_t)(2 * R + 1) * lds_pitch * 4 + stage;
  int xr_pitch = 0;
#define HC_PI(C, NH) do {                                                                                         \
  if (semantics == 1 && NH > 0)                                                                                   \
    hipLaunchKernelGGL((k_pyramid_integral<C, NH, (NH > 0 ? 1 : 0)>), grid, block, lds, st, frames, frame_stride, row_step, rows, cols,   \
                       levels, xtab, ytab, sum, sq, int_stride, pyr, pyr_stride, R, lds_pitch, src_pitch, vnf, grid_stride,    \
                       norm_w, norm_h, xr_pitch, store_sq ? 1 : 0, frame_chunk);                                   \
  else                                                                                                            \
  hipLaunchKernelGGL((k_pyramid_integral<C, NH>), grid, block, lds, st, frames, frame_stride, row_step, rows, cols,   \
                     levels, xtab, ytab, sum, sq, int_stride, pyr, pyr_stride, R, lds_pitch, src_pitch, vnf, grid_stride,    \
                     norm_w, norm_h, xr_pitch, store_sq ? 1 : 0, frame_chunk); } while (0)
  // variance factors from the row prefixes in LDS (register ring, variance_rows): chunks of 8 rows, at most two grid
  // columns per thread, the stock inner-rect heights, box sums that fit the packed ring word.  Anything else reads the
  // integrals back (NH = 0).  HC_VAR_RING=0 pins the old route (A/B).
  const bool ring_ok = knobs.var_ring;
  const bool ring = ring_ok && vnf && sq && R == 8 && PI_THREADS == 512 && norm_w <= 32 &&
                    (norm_h == 18 || norm_h == 22);
  // the 2.4 generation exists on the ring route only, with every grid column covered by a ring (the caller asks first:
  // pyramid_integral_takes_cv24)
  if (semantics == 1 && !(ring && max_w <= 2 * PI_THREADS)) return (int)hipErrorInvalidValue;
  bool stored = sq != nullptr;
  if (ring) {
    // The grid columns beyond the register rings (max_nx - 512 of them) get ring words in LDS, norm_h slots each, when
    // that still leaves room for two workgroups per CU -- if need be with the row pitch rounded to 4 instead of 64
    // words.  Then nothing reads the squares' integral and it is not stored (store_sq: a debug tap or another kernel
    // wants it); otherwise those columns keep the read-back route and their levels store it.
    const bool xring_ok = knobs.var_xring;
    const int extra = max_nx - VR_COLS * PI_THREADS;
    stored = store_sq;                     // without the LDS rings only the levels with read-back columns store it
    if (extra <= 0) {}
    else if (xring_ok && extra <= PI_THREADS) {
      const int xp = (extra + 3) & ~3;
      const size_t ring_bytes = (size_t)(norm_h + 2) * xp * 4;       // + the two running sums per column
      const int pitch4 = (max_w + 3) & ~3;
      if (lds + ring_bytes <= 81920u) { xr_pitch = xp; lds += ring_bytes; stored = store_sq; }
      else if ((size_t)(2 * R + 1) * pitch4 * 4 + stage + ring_bytes <= 81920u) {
        lds_pitch = pitch4;
        lds = (size_t)(2 * R + 1) * lds_pitch * 4 + stage + ring_bytes;
        xr_pitch = xp;
        stored = store_sq;
      }
    }
  }
  if (info) {
    info->frame_chunk = frame_chunk;
    info->remapped = frame_chunk > 0 && nframes > frame_chunk ? 1 : 0;
    info->tail_frames = info->remapped ? nframes % frame_chunk : 0;
    info->rows_per_chunk = R;
    info->sq_stored = stored;
    info->route = !vnf ? 0 : ring ? 2 /* HC_VAR_RING */ : 3 /* HC_VAR_READBACK */;
  }
  if (semantics == 1 && max_nx > VR_COLS * PI_THREADS && xr_pitch == 0) return (int)hipErrorInvalidValue;   // no read-back route in this generation
  if (max_w <= 2 * PI_THREADS) {
    if (ring && norm_h == 18) HC_PI(2, 18);
    else if (ring) HC_PI(2, 22);
    else HC_PI(2, 0);
  }
  else if (max_w <= 4 * PI_THREADS) HC_PI(4, 0);
  else if (max_w <= 8 * PI_THREADS) HC_PI(8, 0);
  else return (int)hipErrorInvalidValue;
#undef HC_PI
  HC_LAUNCH_CHECK();
  return 0;
}

// ---- column strips (k_pyramid_strips) ---------------------------------------------------------------------------
// A level is cut into the fewest strips of about equal width such that every strip (i) computes at most STRIP_COMP_MAX
// pixel columns (stored columns + the norm_w halo: the LDS row pitch), (ii) owns at most PI_THREADS grid columns (one
// register ring per thread) and (iii) samples at most STRIP_SPAN_MAX bytes of a source row (the staging rows of a
// wavefront; what made whole-row staging cost 61 KB at 3840 columns).
constexpr int STRIP_COMP_MAX = 1024;  // hard cap (two stored columns per thread); the default comes with the knobs (640)
constexpr int STRIP_SPAN_MAX = 2048;

static void plan_strips_at_width(const std::vector<DevLevel>& levels, const std::vector<uint32_t>& xtab, int cols, int norm_w, int norm_h,
                                 int semantics, int comp_max_wanted, StripPlan& out) {
  out = StripPlan{};
  out.norm_w = norm_w; out.norm_h = norm_h;
  if (PI_THREADS != 512 || norm_w > 32 || norm_w < 1 || (norm_h != 18 && norm_h != 22) || levels.empty()) return;
  const auto src_col = [&](const DevLevel& L, int x) { const uint32_t t = xtab[L.xtab_ofs + x]; return semantics == 1 ? (int)(t & 0x1fffu) : (int)(t >> 9); };
  const int comp_max = std::min(STRIP_COMP_MAX, std::max(comp_max_wanted, norm_w + 2 + 8));   // never below what one window's box sums need
  int max_comp = 0, max_span = 0;
  size_t carry_words = 0;
  for (size_t li = 0; li < levels.size(); li++) {
    const DevLevel& L = levels[li];
    if (L.w <= 0 || L.h <= 0) continue;
    std::vector<DevStrip> mine;
    for (int n = 1;; n++) {
      if (n > L.w) return;                                             // cannot happen for sane geometry
      mine.clear();
      bool fits = true;
      for (int k = 0; k < n && fits; k++) {
        DevStrip T{};
        T.level = (int32_t)li;
        T.c0 = (int32_t)((long long)L.w * k / n);
        T.c1 = (int32_t)((long long)L.w * (k + 1) / n);
        T.g0 = std::min(L.nx, (T.c0 + L.ystep - 1) / L.ystep);
        T.g1 = k + 1 == n ? L.nx : std::min(L.nx, (T.c1 + L.ystep - 1) / L.ystep);
        T.ce = T.g1 > T.g0 ? std::max(T.c1, std::min(L.w, (T.g1 - 1) * L.ystep + norm_w + 1)) : T.c1;
        const int s0 = src_col(L, T.c0) & ~3, s1 = std::min(src_col(L, T.ce - 1) + 1, cols - 1);
        T.sx0 = s0;
        T.sxn = ((s1 - s0 + 1) + 3) & ~3;
        T.left = -1; T.has_right = 0;
        fits = T.ce - T.c0 <= comp_max && T.g1 - T.g0 <= PI_THREADS && T.sxn <= STRIP_SPAN_MAX && T.c1 > T.c0;
        mine.push_back(T);
      }
      if (fits) break;
    }
    for (size_t k = 0; k < mine.size(); k++) {
      DevStrip& T = mine[k];
      if (k > 0) T.left = (int32_t)(out.strips.size() - 1);
      if (k + 1 < mine.size()) {
        T.has_right = 1;
        T.carry_ofs = (int32_t)carry_words;
        carry_words += (size_t)2 * L.h;                              // one 8-byte word per row
      }
      max_comp = std::max(max_comp, T.ce - T.c0);
      max_span = std::max(max_span, T.sxn);
      out.strips.push_back(T);
    }
  }
  if (out.strips.empty() || out.strips.size() > 65535) return;
  out.lds_pitch = (max_comp + 63) & ~63;
  out.src_pitch = max_span + 4;                                       // (ox1 - sx0 can equal the span when ce - 1 maps to the last column)
  out.src_pitch = (out.src_pitch + 3) & ~3;
  out.lds_bytes = ((size_t)(2 * (PI_THREADS / 64) + 1) * out.lds_pitch + 2 * (PI_THREADS / 64)) * 4 + (size_t)(PI_THREADS / 64) * 2 * out.src_pitch;
  out.carry_words = (carry_words + 1) & ~(size_t)1;
  out.ok = out.lds_bytes <= 81920u;                                   // two workgroups per CU
}

// The widest strips whose workgroup still fits two to a CU: 832 computed columns take a 1280x720 ladder as single strips
// but for its two widest scan-step-1 levels (and run it faster than the one-workgroup-per-level kernel: 4.43 against 4.53
// ms per 512 frames, no register spills); 1920- and 3840-column frames need more LDS for their source spans and get 640.
// HC_STRIP_COMP_MAX pins a width (tests: narrow strips on small frames).
void plan_pyramid_strips(const std::vector<DevLevel>& levels, const std::vector<uint32_t>& xtab, int cols, int norm_w, int norm_h,
                         int semantics, const PrepKnobs& knobs, StripPlan& out) {
  if (knobs.strip_comp_max > 0) { plan_strips_at_width(levels, xtab, cols, norm_w, norm_h, semantics, knobs.strip_comp_max, out); return; }
  for (int wdt : {832, 640, 512, 384}) {
    plan_strips_at_width(levels, xtab, cols, norm_w, norm_h, semantics, wdt, out);
    if (out.ok) return;
  }
}

int launch_pyramid_strips(hipStream_t st, int nframes, const StripPlan& plan, const DevStrip* d_strips, const uint8_t* frames,
                          size_t frame_stride, size_t row_step, int rows, int cols, const DevLevel* levels, const uint32_t* xtab,
                          const uint32_t* ytab, uint32_t* sum, uint32_t* sq, size_t int_stride, uint8_t* pyr, size_t pyr_stride,
                          const PrepKnobs& knobs, float* vnf, size_t grid_stride, bool store_sq, uint32_t* carry,
                          unsigned int* progress, PrepLaunchInfo* info, int semantics) {
  if (!plan.ok || nframes == 0) return plan.ok ? 0 : (int)hipErrorInvalidValue;
  const int nstrips = (int)plan.strips.size();
  // row counts of every (frame, strip) and the error word behind them start at zero
  hipError_t e = hipMemsetAsync(progress, 0, ((size_t)nframes * nstrips + 1) * sizeof(unsigned int), st);
  if (e != hipSuccess) return (int)e;
  unsigned int* err_word = progress + (size_t)nframes * nstrips;
  const dim3 grid(nframes, nstrips), block(PI_THREADS);
#define HC_PS(NH, SEM)                                                                                                          \
  hipLaunchKernelGGL((k_pyramid_strips<NH, SEM>), grid, block, plan.lds_bytes, st, frames, frame_stride, row_step, rows, cols, levels, \
                     d_strips, xtab, ytab, sum, sq, int_stride, pyr, pyr_stride, plan.lds_pitch, plan.src_pitch, vnf, grid_stride,  \
                     plan.norm_w, plan.norm_h, store_sq ? 1 : 0, knobs.frame_chunk, carry, plan.carry_words, progress, err_word)
  if (semantics == 1) { if (plan.norm_h == 18) HC_PS(18, 1); else HC_PS(22, 1); }
  else { if (plan.norm_h == 18) HC_PS(18, 0); else HC_PS(22, 0); }
#undef HC_PS
  HC_LAUNCH_CHECK();
  if (info) {
    info->frame_chunk = knobs.frame_chunk;
    info->remapped = knobs.frame_chunk > 0 && nframes > knobs.frame_chunk ? 1 : 0;
    info->tail_frames = info->remapped ? nframes % knobs.frame_chunk : 0;
    info->rows_per_chunk = PI_THREADS / 64;
    info->sq_stored = sq != nullptr && store_sq;
    info->route = vnf ? 5 /* HC_VAR_RING_STRIPS */ : 0;
  }
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

}  // namespace hc
