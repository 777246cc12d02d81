// api_boxes.cpp -- rectangles outside a detection call: cv::groupRectangles as a function, and box files: frame
// id -> rect list, the hand-over format between a batched / multi-GPU detection run and the sequential tracker (reference
// src/tracker/face_tracker.cc:123-148; the reference only has a TODO for tracker metadata files, src/file_io.h:41-48).
// Layout in include/hip_cascade.h; identical to ugproj-detector_amd/shard.py.  Pure host code.

#include <algorithm>
#include <cstdio>
#include <cstring>

#include "group_rects.h"
#include "hc_context.h"

extern "C" {

int hc_group_rectangles(const hc_rect* in, int n, int group_threshold, double eps, hc_rect* out, int cap, int* nout) {
  if ((!in && n > 0) || !nout || n < 0) return hc::set_err(HC_ERR_ARG, "null argument");
  std::vector<hc_rect> v(in, in + n);
  hc::group_rectangles(v, group_threshold, eps);
  if ((int)v.size() > cap) return hc::set_err(HC_ERR_CAPACITY, "output rectangle buffer too small");
  for (size_t i = 0; i < v.size(); i++) out[i] = v[i];
  *nout = (int)v.size();
  return HC_OK;
}
// ---- box files: frame id -> rect list, the hand-over format between a batched / multi-GPU detection
// run and the sequential tracker.  Layout in include/hip_cascade.h; identical to ugproj-detector_amd/shard.py.
int hc_boxes_save(const char* path, int nframes, const int32_t* frame_ids, const int32_t* offsets, const hc_rect* rects) {
  if (!path || nframes < 0 || (nframes > 0 && (!frame_ids || !offsets)) || (nframes > 0 && offsets[nframes] > 0 && !rects))
    return hc::set_err(HC_ERR_ARG, "hc_boxes_save: null argument");
  std::vector<int> order(nframes);
  for (int i = 0; i < nframes; i++) {
    order[i] = i;
    if (offsets[i + 1] < offsets[i]) return hc::set_err(HC_ERR_ARG, "hc_boxes_save: offsets must not decrease");
  }
  std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return frame_ids[a] < frame_ids[b]; });   // ascending id
  for (int i = 1; i < nframes; i++)
    if (frame_ids[order[i]] == frame_ids[order[i - 1]]) return hc::set_err(HC_ERR_ARG, "hc_boxes_save: duplicate frame id");
  FILE* f = fopen(path, "wb");
  if (!f) return hc::set_err(HC_ERR_IO, std::string("cannot create ") + path);
  const int32_t hdr[2] = {1, nframes};
  bool ok = fwrite("HCBX", 1, 4, f) == 4 && fwrite(hdr, 4, 2, f) == 2;
  for (int k = 0; ok && k < nframes; k++) {
    const int i = order[k];
    const int32_t rec[2] = {frame_ids[i], offsets[i + 1] - offsets[i]};
    ok = fwrite(rec, 4, 2, f) == 2;
    if (ok && rec[1] > 0) ok = fwrite(rects + offsets[i], sizeof(hc_rect), (size_t)rec[1], f) == (size_t)rec[1];
  }
  ok = (fclose(f) == 0) && ok;
  return ok ? HC_OK : hc::set_err(HC_ERR_IO, std::string("short write to ") + path);
}

int hc_boxes_load(const char* path, int32_t* frame_ids, int32_t* offsets, int frame_capacity, hc_rect* rects, int rect_capacity,
                  int* nframes, int* nrects) {
  if (!path || !nframes || !nrects) return hc::set_err(HC_ERR_ARG, "hc_boxes_load: null argument");
  *nframes = *nrects = 0;
  FILE* f = fopen(path, "rb");
  if (!f) return hc::set_err(HC_ERR_IO, std::string("cannot open ") + path);
  std::string buf;
  char tmp[1 << 16];
  size_t n;
  while ((n = fread(tmp, 1, sizeof tmp, f)) > 0) buf.append(tmp, n);
  fclose(f);
  if (buf.size() < 12 || memcmp(buf.data(), "HCBX", 4) != 0) return hc::set_err(HC_ERR_FORMAT, "not an HCBX file");
  int32_t hdr[2];
  memcpy(hdr, buf.data() + 4, 8);
  if (hdr[0] != 1 || hdr[1] < 0) return hc::set_err(HC_ERR_FORMAT, "unsupported HCBX version / frame count");
  // first pass: validate and count, so that a too-small buffer learns the sizes it needs
  size_t p = 12;
  long long total = 0;
  for (int i = 0; i < hdr[1]; i++) {
    if (p + 8 > buf.size()) return hc::set_err(HC_ERR_FORMAT, "truncated HCBX file");
    int32_t rec[2];
    memcpy(rec, buf.data() + p, 8);
    p += 8;
    if (rec[1] < 0 || p + (size_t)rec[1] * 16 > buf.size()) return hc::set_err(HC_ERR_FORMAT, "truncated HCBX file");
    p += (size_t)rec[1] * 16;
    total += rec[1];
  }
  *nframes = hdr[1];
  *nrects = (int)total;
  if (!frame_ids && !offsets && !rects) return HC_OK;                       // size query
  if (hdr[1] > frame_capacity || total > rect_capacity || !frame_ids || !offsets || (total > 0 && !rects))
    return hc::set_err(HC_ERR_CAPACITY, "hc_boxes_load: output buffers too small (sizes returned)");
  p = 12;
  int at = 0;
  for (int i = 0; i < hdr[1]; i++) {
    int32_t rec[2];
    memcpy(rec, buf.data() + p, 8);
    p += 8;
    frame_ids[i] = rec[0];
    offsets[i] = at;
    if (rec[1] > 0) memcpy(rects + at, buf.data() + p, (size_t)rec[1] * 16);
    p += (size_t)rec[1] * 16;
    at += rec[1];
  }
  offsets[hdr[1]] = at;
  return HC_OK;
}

}  // extern "C"
