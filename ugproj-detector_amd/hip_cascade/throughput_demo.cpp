// throughput_demo.cpp -- the batched C ABI driven from plain C++ (no Python, no torch, no HIP headers):
// two contexts software-pipelined, frames handed over as HOST buffers (so the rate printed here is the
// PCIe-inclusive one: H2D of every frame + kernels + D2H of candidates + host grouping).
//   throughput_demo <cascade> <frames.u8 | synth:kind[:first]> <nframes> <rows> <cols> [batch=32] [rounds=8] [pool=pageable|pinned|registered] [cycles=1]
// cycles: how often a round streams the frame pool through (every pass copies the frames up again): with a few batches
// per round the pipeline is mostly fill and drain, with cycles >= 4 the rate is the steady-state one.
// pool: where the frames live on the host -- plain pageable memory (what cv::VideoCapture hands the reference),
// a page-locked pool from hc_host_alloc, or the pageable buffer pinned in place with hc_host_register.
// frames.u8: nframes tight 8-bit gray frames.  Prints one line per round and a summary:
//   frames/s, total boxes (identical every round: the path is deterministic).
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include <algorithm>
#include <thread>
#include <unistd.h>

#include "../../include/hip_cascade.h"
#include "synth.hpp"

// frames.u8 argument: a file of tight 8-bit gray frames, or "synth:<noise|texture|planted>[:first]" -- frames first ..
// first + nframes - 1 of the deterministic generator (hip_cascade/synth.hpp == ugproj-detector_amd/synth.py), made here
// on a few threads; the face patch is data/face64.u8 next to this executable.
static bool load_or_make_frames(const char* arg, unsigned char* frames, int nframes, int rows, int cols) {
  const size_t fbytes = (size_t)rows * cols;
  if (strncmp(arg, "synth:", 6) != 0) {
    FILE* f = fopen(arg, "rb");
    const bool ok = f && fread(frames, 1, (size_t)nframes * fbytes, f) == (size_t)nframes * fbytes;
    if (f) fclose(f);
    return ok;
  }
  std::string kind(arg + 6);
  long first = 0;
  const size_t colon = kind.find(':');
  if (colon != std::string::npos) { first = atol(kind.c_str() + colon + 1); kind.resize(colon); }
  char exe[4096];
  const ssize_t len = readlink("/proc/self/exe", exe, sizeof exe - 1);
  if (len <= 0) return false;
  exe[len] = 0;
  std::string dir(exe);
  dir.resize(dir.rfind('/'));
  try {
    const std::vector<uint8_t> face = hip_cascade::synth::load_face_patch(dir + "/data/face64.u8");
    const int nt = (int)std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
    std::vector<std::thread> th;
    for (int t = 0; t < nt; t++)
      th.emplace_back([&, t] {
        for (int i = t; i < nframes; i += nt) hip_cascade::synth::make_frame(kind, rows, cols, (uint64_t)(first + i), face, frames + (size_t)i * fbytes);
      });
    for (auto& x : th) x.join();
  } catch (const std::exception& e) {
    fprintf(stderr, "%s\n", e.what());
    return false;
  }
  return true;
}

#define CHECK(x) do { int rc_ = (x); if (rc_ != HC_OK) { fprintf(stderr, "%s -> %d: %s\n", #x, rc_, hc_last_error()); return 2; } } while (0)

int main(int argc, char** argv) {
  if (argc < 6) { fprintf(stderr, "usage: %s cascade frames.u8 nframes rows cols [batch] [rounds]\n", argv[0]); return 64; }
  const int nframes = atoi(argv[3]), rows = atoi(argv[4]), cols = atoi(argv[5]);
  const int batch = argc > 6 ? atoi(argv[6]) : 32, rounds = argc > 7 ? atoi(argv[7]) : 8;
  const std::string pool = argc > 8 ? argv[8] : "pageable";
  const int cycles = argc > 9 ? std::max(1, atoi(argv[9])) : 1;
  const size_t total_bytes = (size_t)nframes * rows * cols;
  std::vector<unsigned char> pageable;
  unsigned char* frames = nullptr;
  if (pool == "pinned") {
    void* p = nullptr;
    CHECK(hc_host_alloc(total_bytes, &p));
    frames = (unsigned char*)p;
  } else {
    pageable.resize(total_bytes);
    frames = pageable.data();
    if (pool == "registered") CHECK(hc_host_register(frames, total_bytes));
    else if (pool != "pageable") { fprintf(stderr, "unknown pool %s\n", pool.c_str()); return 64; }
  }
  if (!load_or_make_frames(argv[2], frames, nframes, rows, cols)) { fprintf(stderr, "cannot read / make frames\n"); return 1; }
  hc_cascade* casc = nullptr;
  if (hc_cascade_load(argv[1], &casc) != HC_OK) { fprintf(stderr, "cannot load cascade: %s\n", hc_last_error()); return 1; }
  hc_context* ctx[2] = {nullptr, nullptr};
  for (auto& c : ctx) {
    // both pipeline slots on ONE stream (the first context's): kernels of consecutive batches back to back; the host
    // frames of the next batch still go up meanwhile, on the context's own upload stream (include/hip_cascade.h)
    CHECK(hc_context_create(casc, 0, ctx[0] ? hc_context_stream(ctx[0]) : nullptr, &c));
    // the reference's call: scaleFactor 1.1, minNeighbors 2, minSize 30x30 (src/tracker/detector.cc:25-31)
    CHECK(hc_context_configure(c, rows, cols, 1.1, 2, 30, 30, 0, 0, batch, 0));
  }
  const int pool_batches = (nframes + batch - 1) / batch, nbatches = pool_batches * cycles;
  std::vector<hc_rect> out(1 << 16);
  std::vector<int32_t> ofs(batch + 1);
  const size_t fbytes = (size_t)rows * cols;
  auto enqueue = [&](int b) {
    const int pb = b % pool_batches, n = std::min(batch, nframes - pb * batch);
    return hc_detect_batch_enqueue_host(ctx[b & 1], frames + (size_t)pb * batch * fbytes, fbytes, cols, n);
  };
  long first_total = -1;
  double best = 0;
  for (int r = 0; r < rounds; r++) {
    long total = 0;
    const auto t0 = std::chrono::steady_clock::now();
    CHECK(enqueue(0));
    for (int b = 0; b < nbatches; b++) {
      if (b + 1 < nbatches) CHECK(enqueue(b + 1));
      CHECK(hc_detect_batch_wait(ctx[b & 1], out.data(), (int)out.size(), ofs.data()));
      total += ofs[std::min(batch, nframes - (b % pool_batches) * batch)];
    }
    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    printf("round %d: %.1f frames/s (host frames in, boxes out), %ld boxes\n", r, (double)nframes * cycles / dt, total);
    if (first_total < 0) first_total = total;
    if (total != first_total) { fprintf(stderr, "non-deterministic box count\n"); return 3; }
    if (r > 0 && (double)nframes * cycles / dt > best) best = (double)nframes * cycles / dt;
  }
  printf("best %.1f frames/s PCIe-inclusive (%s host frames, batch %d, %d batches per round), %ld boxes per pass\n", best, pool.c_str(), batch,
         nbatches, first_total / cycles);
  hc_context_destroy(ctx[1]);            // the stream's owner goes last
  hc_context_destroy(ctx[0]);
  if (pool == "pinned") hc_host_free(frames);
  else if (pool == "registered") CHECK(hc_host_unregister(frames));
  hc_cascade_free(casc);
  return 0;
}
