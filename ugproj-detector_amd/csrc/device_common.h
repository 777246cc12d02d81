// device_common.h -- device-side helpers shared by the translation units that hold kernels (kernels.hip and the
// compiled-stage units csrc/spec/*.hip).  Include after <hip/hip_runtime.h> and device_types.h.
#pragma once

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

}  // namespace hc
