// spec_altp.hip -- the 16-bit-tile build of the cascade kernel (cascade_body.inc HC_P16) for haarcascade_frontalface_alt:
// scan-step-1 launch, whole bands, 16 grid rows and 4 wavefronts per workgroup.  The tile holds the integral modulo 2^16
// in two copies (one shifted by an entry), the row-form stages 0-1 evaluate two adjacent windows per lane from aligned
// 4-byte reads of entry pairs (spec_altp.inc pair0 / pair1, tools/gen_spec.py --p16), every later stage reads single
// 2-byte entries.  Exact: every rect of the cascade has at most 257 pixels after hc_api.cpp p16_model has cut the four
// larger ones in two, so a rect sum modulo 2^16 is the rect sum.  Anything this unit does not take (segments, the
// scan-step-2 launch, other cascades) runs the standard build with the standard tables.
#include <hip/hip_runtime.h>
#include <cstdio>

#include "../device_types.h"
#include "../kernels.h"
#include "../device_common.h"

#if !defined(HC_PHASE) && !defined(HC_OFF_WORDS)
namespace hc {
namespace casc_p16 {
#define HC_INTW 1
#define HC_P16 1
#define HC_BODY_NO_LAUNCHERS
#include "../cascade_body.inc"
#undef HC_INTW

// T16: one 2-byte entry of the tile, zero-extended; T2: an aligned pair of entries (two adjacent windows' corners);
// K / A as in spec_unit.inc
#define T16(a, o) ((int)*(hc_lds_c16)(uintptr_t)(uint32_t)((a) + (o)))
#define T2(a, o) (*(hc_lds_c16x2)(uintptr_t)(uint32_t)((a) + (o)))
#define K(lo, hi, dlo, dhi) asm volatile("s_mov_b32 %0, " #dlo "\n\ts_mov_b32 %1, " #dhi : "=s"(lo), "=s"(hi))
#define A(acc, vnf, f, thrbits, lo, hi)                                                                                   \
  do {                                                                                                                    \
    const float v_ = (float)(f) * (vnf);                                                                                  \
    acc = __builtin_fma(__hiloint2double(hi, lo), __hiloint2double(v_ < __uint_as_float(thrbits) ? 0x3ff00000 : 0, 0), acc); \
  } while (0)
#include "spec_altp.inc"
#undef T16
#undef T2
#undef K
#undef A

static int launch_unit(hipStream_t st, int nframes, bool segments, int rows, int nwaves, size_t lds_bytes, const CascadeArgs& args) {
  if (segments || args.g.half != 0 || args.g.pitch != Spec_altp::pitch1 || args.row_stages > Spec_altp::pair_stages) return -1;
  const dim3 grid(nframes, args.nbands);
  // the production shape (16 grid rows, 4 wavefronts: 1024 windows per tile) and the double tile (32 rows, 8 wavefronts: the
  // generated offsets depend on the tile's pitch only, not on its height)
  if (rows == 16 && nwaves == 4) {
    if (args.taps) hipLaunchKernelGGL((k_cascade<false, 16, 4, false, true, Spec_altp, 1>), grid, dim3(256), lds_bytes, st, args);
    else hipLaunchKernelGGL((k_cascade<false, 16, 4, false, false, Spec_altp, 1>), grid, dim3(256), lds_bytes, st, args);
  } else if (rows == 32 && nwaves == 8) {
    if (args.taps) hipLaunchKernelGGL((k_cascade<false, 32, 8, false, true, Spec_altp, 1>), grid, dim3(512), lds_bytes, st, args);
    else hipLaunchKernelGGL((k_cascade<false, 32, 8, false, false, Spec_altp, 1>), grid, dim3(512), lds_bytes, st, args);
  } else {
    return -1;
  }
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : (int)e;
}
}  // namespace casc_p16

const SpecUnit* spec_unit_altp() {
  static const SpecUnit u = {casc_p16::Spec_altp::hash, casc_p16::Spec_altp::win_w, casc_p16::Spec_altp::win_h,
                             casc_p16::Spec_altp::nstages, false, casc_p16::launch_unit, true};
  return &u;
}
}  // namespace hc
#endif
