// synth.hpp -- the deterministic synthetic frames of SURVEY.md 8d in C++, byte for byte what ugproj-detector_amd/synth.py
// writes (tests/test_host.py compares the two on the CPU), so that the C++ drivers (throughput_demo, multi_gpu_demo)
// need no frame file and a C++ run can be checked against the Python-side oracle tests frame by frame.
//
// All arithmetic is integer.  PRNG: SplitMix64, frame i seeded with 0x9E3779B97F4A7C15 * (i + 1).
//   noise    iid uniform bytes
//   texture  4 octaves of box-blurred noise, stretched to mean 128 / sigma ~40
//   planted  texture + face patches (data/face64.u8) stamped at SplitMix-chosen places and sizes
// The reference has no synthetic input of its own (it reads a video file, src/tracker/face_tracker.cc:93-111).
#pragma once

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

namespace hip_cascade {
namespace synth {

constexpr uint64_t GAMMA = 0x9E3779B97F4A7C15ull;

// the k-th (k >= 1) SplitMix64 output for `seed`
inline uint64_t splitmix64_at(uint64_t seed, uint64_t k) {
  uint64_t z = seed + k * GAMMA;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

inline uint64_t frame_seed(uint64_t i, uint64_t salt = 0) { return GAMMA * (i + 1) + salt * 0xD1B54A32D192ED03ull; }

// rows x cols iid bytes: the little-endian bytes of successive SplitMix64 words
inline void noise(int rows, int cols, uint64_t seed, uint8_t* out) {
  const size_t n = (size_t)rows * cols;
  for (size_t w = 0; w * 8 < n; w++) {
    const uint64_t z = splitmix64_at(seed, w + 1);
    for (size_t b = 0; b < 8 && w * 8 + b < n; b++) out[w * 8 + b] = (uint8_t)(z >> (8 * b));
  }
}

inline int64_t floor_div(int64_t a, int64_t b) {   // Python's // for b > 0
  int64_t q = a / b;
  if ((a % b) != 0 && (a < 0)) q--;
  return q;
}

// integer box mean with edge replication, window (2r+1)^2, accumulated into acc with weight wgt
inline void box_mean_acc(const uint8_t* img, int rows, int cols, int r, int64_t wgt, std::vector<int64_t>& acc) {
  const int pr = rows + 2 * r, pc = cols + 2 * r, k = 2 * r + 1;
  std::vector<int64_t> ii((size_t)(pr + 1) * (pc + 1), 0);
  for (int y = 0; y < pr; y++) {
    const int sy = y < r ? 0 : (y >= rows + r ? rows - 1 : y - r);
    int64_t run = 0;
    for (int x = 0; x < pc; x++) {
      const int sx = x < r ? 0 : (x >= cols + r ? cols - 1 : x - r);
      run += img[(size_t)sy * cols + sx];
      ii[(size_t)(y + 1) * (pc + 1) + x + 1] = ii[(size_t)y * (pc + 1) + x + 1] + run;
    }
  }
  const int64_t area = (int64_t)k * k;
  for (int y = 0; y < rows; y++)
    for (int x = 0; x < cols; x++) {
      const int64_t s = ii[(size_t)(y + k) * (pc + 1) + x + k] - ii[(size_t)y * (pc + 1) + x + k] - ii[(size_t)(y + k) * (pc + 1) + x] +
                        ii[(size_t)y * (pc + 1) + x];
      acc[(size_t)y * cols + x] += (s / area) * wgt;
    }
}

inline uint64_t isqrt_u128(unsigned __int128 n) {   // floor(sqrt(n)), n < 2^126
  if (n == 0) return 0;
  unsigned __int128 x = n, y = (x + 1) / 2;
  while (y < x) { x = y; y = (x + n / x) / 2; }
  return (uint64_t)x;
}

inline void texture(int rows, int cols, uint64_t seed, uint8_t* out) {
  const size_t n = (size_t)rows * cols;
  std::vector<int64_t> t(n, 0);
  std::vector<uint8_t> nz(n);
  const int radii[4] = {2, 4, 8, 16};
  for (int o = 0; o < 4; o++) {
    noise(rows, cols, seed + 0x1000ull * (uint64_t)(o + 1), nz.data());
    box_mean_acc(nz.data(), rows, cols, radii[o], o + 1, t);
  }
  __int128 s1 = 0, s2 = 0;
  for (size_t i = 0; i < n; i++) { s1 += t[i]; s2 += (__int128)t[i] * t[i]; }
  const __int128 nn = (__int128)n;
  const int64_t mean = (int64_t)(s1 / nn);                                  // s1 >= 0
  const unsigned __int128 var_fx = (unsigned __int128)((s2 * nn - s1 * s1) * 65536) / (unsigned __int128)(nn * nn);   // variance, 16.16
  int64_t std_fx = (int64_t)isqrt_u128(var_fx);                             // sigma, 8.8
  if (std_fx < 1) std_fx = 1;
  for (size_t i = 0; i < n; i++) {
    const int64_t v = 128 + floor_div((t[i] - mean) * (40 * 256), std_fx);
    out[i] = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
  }
}

// the 64x64 face patch that ships with the package (data/face64.u8)
inline std::vector<uint8_t> load_face_patch(const std::string& path) {
  std::vector<uint8_t> p(64 * 64);
  FILE* f = fopen(path.c_str(), "rb");
  if (!f || fread(p.data(), 1, p.size(), f) != p.size()) {
    if (f) fclose(f);
    throw std::runtime_error("cannot read the face patch " + path);
  }
  fclose(f);
  return p;
}

// integer nearest / box resample of the square n x n patch to size x size
inline std::vector<uint8_t> scale_patch(const std::vector<uint8_t>& patch, int n, int size) {
  std::vector<uint8_t> out((size_t)size * size);
  if (size >= n) {
    for (int y = 0; y < size; y++)
      for (int x = 0; x < size; x++) out[(size_t)y * size + x] = patch[(size_t)((int64_t)y * n / size) * n + (int64_t)x * n / size];
    return out;
  }
  std::vector<int64_t> ii((size_t)(n + 1) * (n + 1), 0);
  for (int y = 0; y < n; y++) {
    int64_t run = 0;
    for (int x = 0; x < n; x++) {
      run += patch[(size_t)y * n + x];
      ii[(size_t)(y + 1) * (n + 1) + x + 1] = ii[(size_t)y * (n + 1) + x + 1] + run;
    }
  }
  for (int y = 0; y < size; y++) {
    const int y0 = (int)((int64_t)y * n / size), y1 = (int)((int64_t)(y + 1) * n / size);
    for (int x = 0; x < size; x++) {
      const int x0 = (int)((int64_t)x * n / size), x1 = (int)((int64_t)(x + 1) * n / size);
      const int64_t s = ii[(size_t)y1 * (n + 1) + x1] - ii[(size_t)y0 * (n + 1) + x1] - ii[(size_t)y1 * (n + 1) + x0] + ii[(size_t)y0 * (n + 1) + x0];
      const int64_t area = (int64_t)(y1 - y0) * (x1 - x0);
      out[(size_t)y * size + x] = (uint8_t)((s + area / 2) / area);
    }
  }
  return out;
}

// texture + nfaces stamped patches; max_size 0 = max(min(rows, cols) / 3, min_size + 1) (synth.py planted)
inline void planted(int rows, int cols, uint64_t seed, const std::vector<uint8_t>& face, uint8_t* out, int nfaces = 8, int min_size = 40,
                    int max_size = 0) {
  texture(rows, cols, seed, out);
  if (max_size <= 0) {
    max_size = (rows < cols ? rows : cols) / 3;
    if (max_size < min_size + 1) max_size = min_size + 1;
  }
  for (int k = 0; k < nfaces; k++) {
    const uint64_t r0 = splitmix64_at(seed ^ 0xFACEull, 3 * k + 1), r1 = splitmix64_at(seed ^ 0xFACEull, 3 * k + 2),
                   r2 = splitmix64_at(seed ^ 0xFACEull, 3 * k + 3);
    const int size = min_size + (int)(r0 % (uint64_t)(max_size - min_size));
    if (size > rows || size > cols) continue;
    const int x = (int)(r1 % (uint64_t)(cols - size + 1)), y = (int)(r2 % (uint64_t)(rows - size + 1));
    const std::vector<uint8_t> p = scale_patch(face, 64, size);
    for (int yy = 0; yy < size; yy++) memcpy(out + (size_t)(y + yy) * cols + x, p.data() + (size_t)yy * size, (size_t)size);
  }
}

// frame `index` of a family ("noise", "texture", "planted"), as synth.py make_frames(kind, 1, rows, cols, first=index)
inline void make_frame(const std::string& kind, int rows, int cols, uint64_t index, const std::vector<uint8_t>& face, uint8_t* out) {
  const uint64_t seed = frame_seed(index);
  if (kind == "noise") noise(rows, cols, seed, out);
  else if (kind == "texture") texture(rows, cols, seed, out);
  else if (kind == "planted") planted(rows, cols, seed, face, out);
  else throw std::invalid_argument("unknown synthetic frame family " + kind);
}

}  // namespace synth
}  // namespace hip_cascade
