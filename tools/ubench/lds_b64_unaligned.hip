// Does ds_read_b64 at a 4-byte-aligned (not 8-byte-aligned) LDS address return the right two dwords on gfx950 under ROCm,
// and at what rate?  One wavefront-instruction reads lane l's dwords (2 l + o, 2 l + o + 1) for o = 0 (aligned) or 1.
// Bounded loops only; prints values-correct flags and cycles per wave-instruction (s_memtime, one workgroup per CU).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

template <int MODE>   // 0: b64 aligned, 1: b64 at odd dword, 2: ds_read2_b32 (o, o+1) at odd dword, 3: two ds_read_b32
__global__ __launch_bounds__(256) void k(uint32_t* out, unsigned long long* cyc, int iters) {
  __shared__ uint32_t buf[4096];
  for (int i = threadIdx.x; i < 4096; i += 256) buf[i] = (uint32_t)i * 2654435761u;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int o = MODE == 0 ? 0 : 1;
  uint32_t acc0 = 0, acc1 = 0;
  uint32_t addr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t*)buf + (uint32_t)(wave * 512 + 2 * lane + o) * 4u;
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; it++) {
    uint32_t a, b;
    if (MODE <= 1) {
      unsigned long long v;
      asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
      a = (uint32_t)v; b = (uint32_t)(v >> 32);
    } else if (MODE == 2) {
      unsigned long long v;
      asm volatile("ds_read2_b32 %0, %1 offset1:1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
      a = (uint32_t)v; b = (uint32_t)(v >> 32);
    } else {
      asm volatile("ds_read_b32 %0, %2\n\tds_read_b32 %1, %2 offset:4\n\ts_waitcnt lgkmcnt(0)" : "=v"(a), "=v"(b) : "v"(addr) : "memory");
    }
    acc0 += a; acc1 ^= b;
    addr ^= 64u;                                            // alternate between two rows: keeps the loop from being folded
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  out[(size_t)blockIdx.x * 256 + threadIdx.x] = acc0 + acc1;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
  // correctness of one read, checked on the host: lane l must see buf[512 w + 2 l + o] and the next word
  if (blockIdx.x == 0) {
    uint32_t a, b;
    const uint32_t ad = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t*)buf + (uint32_t)(wave * 512 + 2 * lane + o) * 4u;
    if (MODE <= 1) {
      unsigned long long v;
      asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(ad) : "memory");
      a = (uint32_t)v; b = (uint32_t)(v >> 32);
    } else { a = buf[wave * 512 + 2 * lane + o]; b = buf[wave * 512 + 2 * lane + o + 1]; }
    out[(size_t)gridDim.x * 256 + 2 * threadIdx.x] = a;
    out[(size_t)gridDim.x * 256 + 2 * threadIdx.x + 1] = b;
  }
}

template <int MODE>
static void run(const char* name) {
  const int blocks = 256, iters = 2000;
  uint32_t* d; unsigned long long* c;
  hipMalloc(&d, ((size_t)blocks * 256 + 512) * 4); hipMalloc(&c, blocks * 8);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, c, iters);
  hipDeviceSynchronize();
  std::vector<uint32_t> h((size_t)blocks * 256 + 512); std::vector<unsigned long long> hc(blocks);
  hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost); hipMemcpy(hc.data(), c, blocks * 8, hipMemcpyDeviceToHost);
  const int o = MODE == 0 ? 0 : 1;
  int bad = 0;
  for (int t = 0; t < 256; t++) {
    const int w = t >> 6, l = t & 63;
    const uint32_t ea = (uint32_t)(w * 512 + 2 * l + o) * 2654435761u, eb = (uint32_t)(w * 512 + 2 * l + o + 1) * 2654435761u;
    if (h[(size_t)blocks * 256 + 2 * t] != ea || h[(size_t)blocks * 256 + 2 * t + 1] != eb) bad++;
  }
  double s = 0; for (auto v : hc) s += (double)v;
  printf("%-28s values %s   %.1f cycles per iteration per wavefront (4 waves per CU reading)\n", name, bad ? "WRONG" : "ok", s / blocks / iters);
  hipFree(d); hipFree(c);
}

int main() {
  run<0>("ds_read_b64 aligned");
  run<1>("ds_read_b64 odd dword");
  run<2>("ds_read2_b32 odd dword");
  run<3>("2 x ds_read_b32 odd dword");
  return 0;
}
