// device_types.h -- plain structs shared by host planning code and the HIP kernels.
#pragma once

#include <cstdint>

namespace hc {

// Cascade-kernel tiling: one workgroup (4 wavefronts of 64) owns a band of 8/16/32 grid rows
// of a pyramid level (chosen per ystep class at configure time) and walks it left to right in
// steps of TILE_POS grid positions.
constexpr int TILE_POS = 64;
constexpr int WG_THREADS = 256;

struct DevLevel {
  float scale;             // f32 scale factor of the level
  int32_t w, h;            // level image size
  int32_t ystep;           // 1 or 2
  int32_t szw_w, szw_h;    // valid window origins: x < szw_w, y < szw_h
  int32_t y_end;           // scanned rows: y < y_end
  int32_t nx, ny, ny_all;  // grid: nx = ceil(szw_w/ystep), ny = ceil(y_end/ystep), ny_all = ceil(szw_h/ystep)
  int32_t win_w, win_h;    // size of emitted rectangles
  int32_t int_pitch;       // integral row pitch (words)
  int32_t xtab_ofs, ytab_ofs;
  int64_t pyr_ofs;         // bytes, inside one frame's pyramid
  int64_t int_ofs;         // words, inside one frame's integral array
  int64_t grid_ofs;        // elements, inside one frame's grid arrays (vnf, taps): [ny_all][nx]
};

struct WorkItem { int32_t level, a, b; };

// One workgroup's share of a pyramid level in k_pyramid_strips: a column strip.  The strip STORES the integrals of
// pixel columns [c0, c1), COMPUTES pixels of [c0, ce) -- norm_w columns of halo, so that the inner-rect box sums of
// its own windows need nothing from the neighbour -- and emits the variance factors of grid columns [g0, g1).
// Strips of a level are listed left to right; every strip but the last publishes, per row, the finished integral
// values of its last stored column (sum, sqsum) for its right neighbour to add (the only cross-strip dependency).
struct DevStrip {
  int32_t level;
  int32_t c0, c1, ce;
  int32_t g0, g1;
  int32_t sx0, sxn;        // staged source bytes of a row: [sx0, sx0 + sxn), both multiples of 4
  int32_t left;            // index of the left neighbour strip, -1: none
  int32_t has_right;       // publishes carries for a right neighbour
  int32_t carry_ofs;       // word offset of this strip's published rows inside one frame's carry block ((h + 1) x 2 words)
  int32_t pad;
};

// LDS tile geometry of the cascade kernel for one ystep.
struct TileGeom {
  int32_t pitch;   // words per tile row
  int32_t half;    // ystep 2: word offset of the odd-column plane inside a row (0 for ystep 1)
  int32_t rows;    // integral rows staged
  int32_t cols;    // integral columns staged
};

// Corner-sharing kind of a stump.  Haar features are nested rects, so corners of rect 1 / rect 2
// often coincide with corners already loaded for rect 0 / rect 1; six patterns cover every shared
// corner in the four reference cascades.  Inside a stage whose f64 leaf sum is exact in any order
// (DevStage::exact) the host groups the stumps by kind, and the device runs one straight-line loop
// per kind that reads each distinct corner once (7.5 instead of 8.7 LDS reads per stump).
enum NodeKind : int32_t {
  NK_NONE2 = 0,     // 2 rects, 8 reads (also the safe default for any 2-rect pattern not listed)
  NK_BOTTOM = 1,    // 2 rects: r1.c2 == r0.c2, r1.c3 == r0.c3            6 reads
  NK_RIGHT = 2,     // 2 rects: r1.c1 == r0.c1, r1.c3 == r0.c3            6 reads
  NK_LEFT = 3,      // 2 rects: r1.c0 == r0.c0, r1.c2 == r0.c2            6 reads
  NK_TOP = 4,       // 2 rects: r1.c0 == r0.c0, r1.c1 == r0.c1            6 reads
  NK_ANTIDIAG = 5,  // 3 rects: r1.c1 == r0.c1, r2.c1 == r1.c2, r2.c2 == r0.c2   9 reads
  NK_DIAG = 6,      // 3 rects: r1.c0 == r0.c0, r2.c0 == r1.c3, r2.c3 == r0.c3   9 reads
  NK_NONE3 = 7,     // 3 rects, 12 reads
  NK_COUNT = 8
};
// end[k] = index (relative to DevStage::first) one past the last stump of kind k; end[-1] = 0
struct alignas(16) DevStageKinds { uint16_t end[NK_COUNT]; };   // 16-byte aligned: one scalar load (a 2-byte-aligned record is fetched with a vector load)

// One weak-classifier node, offsets pre-resolved for a TileGeom.
struct DevNode {
  uint16_t off[12];   // LDS byte offsets of 3 rects x 4 corners relative to the window origin
  float w[3];         // raw rect weights (w[2] == 0 -> two-rect feature)
  float thr;
  uint32_t a, b;      // stump in a stage with DevStage::exact == 0: f32 bits of the left / right leaf;
                      // stump in an exact stage: low / high word of the f64 (double)left - (double)right;
                      // tree: int32 child codes (<= 0 leaf, > 0 node)
};
static_assert(sizeof(DevNode) == 48, "DevNode layout");

// first/count in nodes (stump) or trees (tree); exact != 0: the f64 sum of any subset of the
// stage's leaf values is exact in any order (sparse stump-parallel mode is then bit-identical);
// exact == 2: the same holds for every later stage as well
// bsum: exact stump stages only -- the f64 sum of the right leaves, the value every stage sum starts from (0 otherwise)
// pair: tree stage in PAIR FORM -- exact, every tree a root with one leaf branch and one second node (all of
//   haarcascade_frontalface_alt2): a tree adds  other + go * ((lb - other) + [v1 < thr1] * (la - lb)),  go = the root
//   sends the window to its second node.  psum = sum of the `other` leaves; the two f64 differences sit in the a / b
//   words of the tree's two node records (dense node table only), DevTree::pad bit 0 says which root branch is the node.
struct DevStage { int32_t first, count; float thr; int16_t exact, pair; double bsum; double psum; };
static_assert(sizeof(DevStage) == 32, "DevStage layout");
struct DevTree { int32_t first_node, nnodes, first_leaf, pad; };   // pad bit 0 (pair form): the root's LEFT branch is the second node

// On-device rectangle grouping (group_kernels.hip): a frame's bin holds HC_GROUP_BIN raw candidate keys, its output slot
// HC_GROUP_OUT grouped boxes; a frame that exceeds either is flagged and grouped on the host.
#define HC_GROUP_BIN 1024
#define HC_GROUP_OUT 128
struct GroupLevel { float scale; int32_t ystep, win_w, win_h; double factor; };   // factor: the 2.4 generation's f64 scale
struct GroupBox { int32_t x, y, width, height; };                                 // == hc_rect

}  // namespace hc
