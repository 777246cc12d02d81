// box_records.hpp -- fixed-capacity box records + overflow blocks: the payload of the multi-GPU box gather
// (SURVEY.md 8e), C++ side.  Same layout and rules as ugproj-detector_amd/shard.py, so a Python rank and a C++
// rank could sit in one gather:
//   record (int32 x RECORD_INTS): frame_id, count, then the first min(count, MAX_BOXES) rects as x, y, w, h
//   overflow block of a rank: rects MAX_BOXES.. of its frames with count > MAX_BOXES, in record order
// Every rank derives the size of every rank's overflow block from the gathered counts, so the second
// (variable-length) gather needs no size exchange and only happens for a batch in which some frame overflowed.
// The gather is lossless: the tracker consumes every box of every frame in frame order
// (reference src/tracker/face_tracker.cc:123-148, :294-302).  Pure host code.
#pragma once

#include <algorithm>
#include <cstdint>
#include <cstring>
#include <map>
#include <stdexcept>
#include <vector>

#include "../../include/hip_cascade.h"

namespace hip_cascade {

constexpr int MAX_BOXES = 64;
constexpr int RECORD_INTS = 2 + 4 * MAX_BOXES;

// rects of frame i: rects[offsets[i] .. offsets[i+1]) -- the layout hc_detect_batch_wait returns
inline void pack_records(const int32_t* frame_ids, int nframes, const int32_t* offsets, const hc_rect* rects,
                         std::vector<int32_t>& records, std::vector<hc_rect>& overflow) {
  records.assign((size_t)nframes * RECORD_INTS, 0);
  overflow.clear();
  for (int i = 0; i < nframes; i++) {
    int32_t* r = records.data() + (size_t)i * RECORD_INTS;
    const int count = offsets[i + 1] - offsets[i];
    r[0] = frame_ids[i];
    r[1] = count;
    const int k = std::min(count, MAX_BOXES);
    if (k > 0) memcpy(r + 2, rects + offsets[i], (size_t)k * sizeof(hc_rect));
    for (int j = MAX_BOXES; j < count; j++) overflow.push_back(rects[offsets[i] + j]);
  }
}

// boxes beyond MAX_BOXES of every rank, from the gathered records (rank order, per_rank records each)
inline std::vector<int64_t> overflow_counts(const int32_t* gathered, int world, int per_rank) {
  std::vector<int64_t> n(world, 0);
  for (int r = 0; r < world; r++)
    for (int i = 0; i < per_rank; i++) {
      const int32_t count = gathered[((size_t)r * per_rank + i) * RECORD_INTS + 1];
      if (count > MAX_BOXES) n[r] += count - MAX_BOXES;
    }
  return n;
}

// gathered records + one overflow block per rank (block r holds overflow_counts()[r] rects, possibly followed by
// padding up to `block_stride` rects) -> frame id -> complete box list.  Records with a negative frame id are padding.
inline std::map<int32_t, std::vector<hc_rect>> unpack_records(const int32_t* gathered, int world, int per_rank,
                                                              const hc_rect* overflow, size_t block_stride) {
  std::map<int32_t, std::vector<hc_rect>> out;
  const std::vector<int64_t> need = overflow_counts(gathered, world, per_rank);
  for (int r = 0; r < world; r++) {
    if ((size_t)need[r] > block_stride) throw std::runtime_error("overflow block shorter than the records say");
    size_t at = 0;
    for (int i = 0; i < per_rank; i++) {
      const int32_t* rec = gathered + ((size_t)r * per_rank + i) * RECORD_INTS;
      if (rec[0] < 0) continue;
      const int count = rec[1], k = std::min(count, MAX_BOXES);
      std::vector<hc_rect> boxes((size_t)count);
      if (k > 0) memcpy(boxes.data(), rec + 2, (size_t)k * sizeof(hc_rect));
      for (int j = MAX_BOXES; j < count; j++) boxes[j] = overflow[(size_t)r * block_stride + at++];
      out[rec[0]] = std::move(boxes);
    }
  }
  return out;
}

}  // namespace hip_cascade
