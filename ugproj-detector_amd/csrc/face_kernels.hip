// face_kernels.hip -- the steps that bracket the cascade call inside
// FaceDetector::detectFaces (reference src/tracker/detector.cc:17-22 and :33-42), on device:
//   k_skin_gray        cvtColor(BGR2YCrCb) + inRange -> mask ; cvtColor(BGR2GRAY) -> gray ; histogram
//   k_equalize_lut     equalizeHist's lookup table from the histogram (one workgroup per frame)
//   k_apply_lut_mask   gray = lut[gray] & mask        (equalizeHist; gray &= mask)
//   k_box_mask_sums    sum of the mask inside every detected rect, from the mask's integral image
//   k_bgr2gray        cvtColor(BGR2GRAY / BGRA2GRAY) alone: what detectMultiScale applies to a colour image
//   k_area_decimate / k_area_resize   cv::resize(INTER_AREA) of the tracker's scan-size step
// 8-bit OpenCV semantics per SURVEY.md A.13 (colour constants revised in round 2): BGR2YCrCb uses the
// 14-bit coefficient set (yuv_shift = 14), BGR2GRAY the 15-bit set RY15/GY15/BY15 (gray_shift = 15) of
// OpenCV >= 4, both with round-to-nearest descale; equalizeHist scale = 255.f / (total - hist[first]) in
// f32, lut = saturate(cvRound(sum * scale)).  Integer work is exact; the f32 operations are IEEE single
// ops in upstream's order (file compiled with -ffp-contract=off).  Parity of these constants is unpinned.
#include <hip/hip_runtime.h>

#include "device_types.h"
#include "kernels.h"

namespace hc {

// COLOR_BGR2GRAY, 8-bit: RY15 = 9798, GY15 = 19235, BY15 = 3735, gray_shift = 15
__device__ __forceinline__ int gray15(int b, int g, int r) { return (b * 3735 + g * 19235 + r * 9798 + (1 << 14)) >> 15; }

// gray plane of a 3- or 4-channel 8-bit image (BGR / BGRA; alpha ignored)
__global__ __launch_bounds__(WG_THREADS) void k_bgr2gray(const uint8_t* __restrict__ src, size_t frame_stride, size_t row_step,
                                                         int channels, int rows, int cols, uint8_t* __restrict__ gray,
                                                         size_t plane_stride) {
  const int i = blockIdx.x * WG_THREADS + threadIdx.x;
  if (i >= rows * cols) return;
  const int y = i / cols, x = i - y * cols;
  const uint8_t* p = src + (size_t)blockIdx.y * frame_stride + (size_t)y * row_step + (size_t)channels * x;
  gray[(size_t)blockIdx.y * plane_stride + i] = (uint8_t)gray15(p[0], p[1], p[2]);
}

__global__ __launch_bounds__(WG_THREADS) void k_skin_gray(const uint8_t* __restrict__ bgr, size_t frame_stride,
                                                          size_t row_step, int rows, int cols, SkinBounds sb,
                                                          uint8_t* __restrict__ gray, uint8_t* __restrict__ mask,
                                                          size_t plane_stride, uint32_t* __restrict__ hist) {
  __shared__ uint32_t h[256];
  h[threadIdx.x] = 0;
  __syncthreads();
  const int frame = blockIdx.y;
  const int i = blockIdx.x * WG_THREADS + threadIdx.x;
  if (i < rows * cols) {
    const int y = i / cols, x = i - y * cols;
    const uint8_t* p = bgr + (size_t)frame * frame_stride + (size_t)y * row_step + 3 * (size_t)x;
    const int b = p[0], g = p[1], r = p[2];
    const int Y = (b * 1868 + g * 9617 + r * 4899 + 8192) >> 14;                  // Y of cvtColor(BGR2YCrCb)
    const int G = gray15(b, g, r);                                                 // cvtColor(BGR2GRAY)
    const int Cr = min(max(((r - Y) * 11682 + (128 << 14) + 8192) >> 14, 0), 255);
    const int Cb = min(max(((b - Y) * 9241 + (128 << 14) + 8192) >> 14, 0), 255);
    const bool in = Y >= sb.lo[0] && Y <= sb.hi[0] && Cr >= sb.lo[1] && Cr <= sb.hi[1] && Cb >= sb.lo[2] && Cb <= sb.hi[2];
    gray[(size_t)frame * plane_stride + i] = (uint8_t)G;
    mask[(size_t)frame * plane_stride + i] = in ? 255 : 0;
    atomicAdd(&h[G], 1u);
  }
  __syncthreads();
  if (h[threadIdx.x]) atomicAdd(&hist[(size_t)frame * 256 + threadIdx.x], h[threadIdx.x]);
}

__global__ __launch_bounds__(256) void k_equalize_lut(const uint32_t* __restrict__ hist, int total,
                                                      uint8_t* __restrict__ lut) {
  __shared__ uint32_t scan[256];
  __shared__ int first;
  const int t = threadIdx.x, frame = blockIdx.x;
  const uint32_t hv = hist[(size_t)frame * 256 + t];
  if (t == 0) first = 256;
  scan[t] = hv;
  __syncthreads();
  if (hv) atomicMin(&first, t);
  for (int d = 1; d < 256; d <<= 1) {           // inclusive scan of the histogram
    const uint32_t v = t >= d ? scan[t - d] : 0u;
    __syncthreads();
    scan[t] += v;
    __syncthreads();
  }
  const int i0 = first;
  const uint32_t h0 = hist[(size_t)frame * 256 + i0];
  uint8_t out;
  if ((int)h0 == total) {
    out = (uint8_t)i0;                            // constant image: dst.setTo(i0)
  } else if (t <= i0) {
    out = 0;
  } else {
    const float scale = 255.f / (float)(total - (int)h0);
    const int sum = (int)(scan[t] - scan[i0]);   // hist[i0+1] + .. + hist[t]
    out = (uint8_t)min(max(__float2int_rn((float)sum * scale), 0), 255);
  }
  lut[(size_t)frame * 256 + t] = out;
}

__global__ __launch_bounds__(WG_THREADS) void k_apply_lut_mask(uint8_t* __restrict__ gray, const uint8_t* __restrict__ mask,
                                                               size_t plane_stride, int npix,
                                                               const uint8_t* __restrict__ lut) {
  __shared__ uint8_t l[256];
  const int frame = blockIdx.y;
  l[threadIdx.x] = lut[(size_t)frame * 256 + threadIdx.x];
  __syncthreads();
  const int i = (blockIdx.x * WG_THREADS + threadIdx.x) * 4;
  uint8_t* g = gray + (size_t)frame * plane_stride;
  const uint8_t* m = mask + (size_t)frame * plane_stride;
  if (i + 4 <= npix && ((plane_stride & 3) == 0)) {
    const uint32_t gv = *(const uint32_t*)(g + i), mv = *(const uint32_t*)(m + i);
    const uint32_t o = (uint32_t)l[gv & 255] | ((uint32_t)l[(gv >> 8) & 255] << 8) | ((uint32_t)l[(gv >> 16) & 255] << 16) |
                       ((uint32_t)l[gv >> 24] << 24);
    *(uint32_t*)(g + i) = o & mv;
  } else {
    for (int k = i; k < min(i + 4, npix); k++) g[k] = l[g[k]] & m[k];
  }
}

// cv::resize(INTER_AREA) with integer decimation factors (reference src/tracker/face_tracker.cc:112-114):
// one thread per destination byte (pixel x channel).  2x2: (a+b+c+d+2)>>2; else cvRound(sum * (1.f/area)).
__global__ __launch_bounds__(WG_THREADS) void k_area_decimate(const uint8_t* __restrict__ src, size_t src_frame_stride,
                                                              size_t src_row_step, int channels, int fx, int fy,
                                                              uint8_t* __restrict__ dst, size_t dst_frame_stride,
                                                              int drows, int dcols) {
  const int i = blockIdx.x * WG_THREADS + threadIdx.x;
  const int rowb = dcols * channels;
  if (i >= drows * rowb) return;
  const int y = i / rowb, xb = i - y * rowb, x = xb / channels, ch = xb - x * channels;
  const uint8_t* s = src + (size_t)blockIdx.y * src_frame_stride + (size_t)(y * fy) * src_row_step + (size_t)(x * fx) * channels + ch;
  int sum = 0;
  for (int j = 0; j < fy; j++)
    for (int k = 0; k < fx; k++) sum += s[(size_t)j * src_row_step + (size_t)k * channels];
  const int v = (fx == 2 && fy == 2) ? (sum + 2) >> 2 : __float2int_rn((float)sum * (1.f / (float)(fx * fy)));
  dst[(size_t)blockIdx.y * dst_frame_stride + i] = (uint8_t)min(v, 255);
}

// cv::resize(INTER_AREA), general shrinking ratio (resizeArea_<uchar, float>): one thread per destination byte.
// Taps of destination column x: xofs[x] .. xofs[x+1] into (xsi, xal); rows likewise.  f32 accumulation in tap
// order, exactly upstream's operation sequence: buf = ((0 + S0*a0) + S1*a1) + ..; sum = b0*buf0, += b1*buf1, ..
__global__ __launch_bounds__(WG_THREADS) void k_area_resize(const uint8_t* __restrict__ src, size_t src_frame_stride,
                                                            size_t src_row_step, int channels, const int* __restrict__ xofs,
                                                            const int* __restrict__ xsi, const float* __restrict__ xal,
                                                            const int* __restrict__ yofs, const int* __restrict__ ysi,
                                                            const float* __restrict__ yal, uint8_t* __restrict__ dst,
                                                            size_t dst_frame_stride, int drows, int dcols) {
  const int i = blockIdx.x * WG_THREADS + threadIdx.x;
  const int rowb = dcols * channels;
  if (i >= drows * rowb) return;
  const int y = i / rowb, xb = i - y * rowb, x = xb / channels, ch = xb - x * channels;
  const uint8_t* base = src + (size_t)blockIdx.y * src_frame_stride + ch;
  const int k0 = xofs[x], k1 = xofs[x + 1], j0 = yofs[y], j1 = yofs[y + 1];
  float sum = 0.f;
  for (int j = j0; j < j1; j++) {
    const uint8_t* S = base + (size_t)ysi[j] * src_row_step;
    float buf = 0.f;
    for (int k = k0; k < k1; k++) {
      const float t = (float)S[(size_t)xsi[k] * channels] * xal[k];
      buf = buf + t;
    }
    const float t = yal[j] * buf;
    sum = (j == j0) ? t : sum + t;
  }
  dst[(size_t)blockIdx.y * dst_frame_stride + i] = (uint8_t)min(max(__float2int_rn(sum), 0), 255);
}

// sums[i] = sum of mask over rect i, from the (rows+1) x pitch integral of the mask
__global__ void k_box_mask_sums(const uint32_t* __restrict__ integ, size_t int_stride, int pitch, const int* __restrict__ boxes,
                                int nboxes, uint32_t* __restrict__ sums) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nboxes) return;
  const int f = boxes[5 * i], x = boxes[5 * i + 1], y = boxes[5 * i + 2], w = boxes[5 * i + 3], h = boxes[5 * i + 4];
  const uint32_t* I = integ + (size_t)f * int_stride;
  sums[i] = I[(size_t)(y + h) * pitch + x + w] - I[(size_t)y * pitch + x + w] - I[(size_t)(y + h) * pitch + x] +
            I[(size_t)y * pitch + x];
}

#define HC_LAUNCH_CHECK() do { hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) return (int)e_; } while (0)

int launch_face_prep(hipStream_t st, int nframes, const uint8_t* bgr, size_t frame_stride, size_t row_step, int rows,
                     int cols, const SkinBounds& sb, uint8_t* gray, uint8_t* mask, size_t plane_stride, uint32_t* hist,
                     uint8_t* lut) {
  if (nframes == 0) return 0;
  const int npix = rows * cols;
  hipError_t e = hipMemsetAsync(hist, 0, (size_t)nframes * 256 * sizeof(uint32_t), st);
  if (e != hipSuccess) return (int)e;
  hipLaunchKernelGGL(k_skin_gray, dim3((npix + WG_THREADS - 1) / WG_THREADS, nframes), dim3(WG_THREADS), 0, st, bgr,
                     frame_stride, row_step, rows, cols, sb, gray, mask, plane_stride, hist);
  HC_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_equalize_lut, dim3(nframes), dim3(256), 0, st, hist, npix, lut);
  HC_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_apply_lut_mask, dim3((npix + 4 * WG_THREADS - 1) / (4 * WG_THREADS), nframes), dim3(WG_THREADS), 0,
                     st, gray, mask, plane_stride, npix, lut);
  HC_LAUNCH_CHECK();
  return 0;
}

int launch_area_decimate(hipStream_t st, int nframes, const uint8_t* src, size_t src_frame_stride, size_t src_row_step,
                         int channels, int fx, int fy, uint8_t* dst, size_t dst_frame_stride, int drows, int dcols) {
  if (nframes == 0) return 0;
  const int n = drows * dcols * channels;
  hipLaunchKernelGGL(k_area_decimate, dim3((n + WG_THREADS - 1) / WG_THREADS, nframes), dim3(WG_THREADS), 0, st, src,
                     src_frame_stride, src_row_step, channels, fx, fy, dst, dst_frame_stride, drows, dcols);
  HC_LAUNCH_CHECK();
  return 0;
}

int launch_area_resize(hipStream_t st, int nframes, const uint8_t* src, size_t src_frame_stride, size_t src_row_step,
                       int channels, const int* xofs, const int* xsi, const float* xal, const int* yofs, const int* ysi,
                       const float* yal, uint8_t* dst, size_t dst_frame_stride, int drows, int dcols) {
  if (nframes == 0) return 0;
  const int n = drows * dcols * channels;
  hipLaunchKernelGGL(k_area_resize, dim3((n + WG_THREADS - 1) / WG_THREADS, nframes), dim3(WG_THREADS), 0, st, src,
                     src_frame_stride, src_row_step, channels, xofs, xsi, xal, yofs, ysi, yal, dst, dst_frame_stride,
                     drows, dcols);
  HC_LAUNCH_CHECK();
  return 0;
}

int launch_bgr2gray(hipStream_t st, int nframes, const uint8_t* src, size_t frame_stride, size_t row_step, int channels,
                    int rows, int cols, uint8_t* gray, size_t plane_stride) {
  if (nframes == 0) return 0;
  const int npix = rows * cols;
  hipLaunchKernelGGL(k_bgr2gray, dim3((npix + WG_THREADS - 1) / WG_THREADS, nframes), dim3(WG_THREADS), 0, st, src,
                     frame_stride, row_step, channels, rows, cols, gray, plane_stride);
  HC_LAUNCH_CHECK();
  return 0;
}

int launch_box_mask_sums(hipStream_t st, const uint32_t* integ, size_t int_stride, int pitch, const int* boxes, int nboxes,
                         uint32_t* sums) {
  if (nboxes == 0) return 0;
  hipLaunchKernelGGL(k_box_mask_sums, dim3((nboxes + 255) / 256), dim3(256), 0, st, integ, int_stride, pitch, boxes,
                     nboxes, sums);
  HC_LAUNCH_CHECK();
  return 0;
}

}  // namespace hc
