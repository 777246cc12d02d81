#pragma once

#include <vector>

#include "../../include/hip_cascade.h"

namespace hc {
// In-place cv::groupRectangles(rects, group_threshold, eps) semantics; output order = class order.
void group_rectangles(std::vector<hc_rect>& rects, int group_threshold, double eps);
}  // namespace hc
