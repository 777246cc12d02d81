// multi_gpu_demo.cpp -- the frame-sharded path from plain C++: ONE process,
// N contexts on N devices, one feeder thread per GPU, RCCL (ncclCommInitAll) only for the gather of box records.
//
//   multi_gpu_demo <cascade> <frames.u8 | synth:kind[:first]> <nframes> <rows> <cols> [ngpus=all] [batch=32] [rounds=3] [out.hcbx] [minNeighbors=2]
//
// frames.u8: nframes tight 8-bit gray frames (the "video").  Every round runs the whole video: the global batch
// g covers frames [g*N*B, (g+1)*N*B); inside it device d owns the contiguous block of B frames starting at
// (g*N + d)*B -- detection of a frame depends on that frame only (reference src/tracker/face_tracker.cc:290-304),
// so there is no data-path collective.  After each global batch the feeder threads hand their fixed-capacity
// records to the main thread, which runs ONE grouped ncclAllGather over all devices (xGMI), plus a second one for
// the overflow blocks only when some frame carries more than 64 boxes (box_records.hpp).  Device 0's copy is
// unpacked into frame order -- what the sequential tracker consumes (face_tracker.cc:123-148) -- and, if asked,
// written as an HCBX box file with hc_boxes_save.  Prints frames/s and the number of boxes per pass.
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include <algorithm>
#include <cstring>
#include <unistd.h>

#include "../../include/hip_cascade.h"
#include "box_records.hpp"
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

using hip_cascade::RECORD_INTS;

#define HC(x) do { int rc_ = (x); if (rc_ != HC_OK) { fprintf(stderr, "%s -> %d: %s\n", #x, rc_, hc_last_error()); exit(2); } } while (0)
#define HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s -> %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)
#define NCCL(x) do { ncclResult_t r_ = (x); if (r_ != ncclSuccess) { fprintf(stderr, "%s -> %s\n", #x, ncclGetErrorString(r_)); exit(2); } } while (0)

namespace {

// what one feeder thread hands over per global batch
struct Slot {
  std::vector<int32_t> records;          // B x RECORD_INTS
  std::vector<hc_rect> overflow;
  bool ready = false;
};

struct Device {
  int id = 0;
  hc_context* ctx[2] = {nullptr, nullptr};
  hipStream_t comm = nullptr;
  int32_t *d_send = nullptr, *d_recv = nullptr;      // records: B / N*B
  int32_t* h_send = nullptr;                         // pinned
  hc_rect *d_osend = nullptr, *d_orecv = nullptr;    // overflow blocks, grown on demand
  size_t ocap = 0;
  Slot slot[2];
  std::mutex mu;
  std::condition_variable cv;
};

}  // namespace

int main(int argc, char** argv) {
  if (argc < 6) { fprintf(stderr, "usage: %s cascade frames.u8 nframes rows cols [ngpus] [batch] [rounds] [out.hcbx]\n", argv[0]); return 64; }
  const int nframes = atoi(argv[3]), rows = atoi(argv[4]), cols = atoi(argv[5]);
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { fprintf(stderr, "no HIP device (there is no CPU fallback)\n"); return 1; }
  const int N = argc > 6 && atoi(argv[6]) > 0 ? std::min(atoi(argv[6]), ndev) : ndev;
  const int B = argc > 7 ? atoi(argv[7]) : 32, rounds = argc > 8 ? atoi(argv[8]) : 3;
  const char* out_path = argc > 9 && argv[9][0] ? argv[9] : nullptr;
  const int min_neighbors = argc > 10 ? atoi(argv[10]) : 2;      // 0 returns the raw candidates: frames with > 64 boxes
  const size_t fbytes = (size_t)rows * cols;
  void* pool = nullptr;
  HC(hc_host_alloc((size_t)nframes * fbytes, &pool));            // page-locked frame pool: async H2D
  unsigned char* frames = (unsigned char*)pool;
  if (!load_or_make_frames(argv[2], frames, nframes, rows, cols)) { fprintf(stderr, "cannot read / make frames\n"); return 1; }
  hc_cascade* casc = nullptr;
  HC(hc_cascade_load(argv[1], &casc));

  std::vector<int> devs(N);
  for (int d = 0; d < N; d++) devs[d] = d;
  std::vector<ncclComm_t> comms(N);
  NCCL(ncclCommInitAll(comms.data(), N, devs.data()));
  std::vector<Device> dev(N);
  for (int d = 0; d < N; d++) {
    Device& D = dev[d];
    D.id = d;
    HIP(hipSetDevice(d));
    HIP(hipStreamCreateWithFlags(&D.comm, hipStreamNonBlocking));
    HIP(hipMalloc((void**)&D.d_send, (size_t)B * RECORD_INTS * 4));
    HIP(hipMalloc((void**)&D.d_recv, (size_t)N * B * RECORD_INTS * 4));
    HIP(hipHostMalloc((void**)&D.h_send, (size_t)B * RECORD_INTS * 4, hipHostMallocDefault));
    for (auto& c : D.ctx) {
      // the pipeline slots of a GPU share the first one's stream: consecutive batches back to back (include/hip_cascade.h)
      HC(hc_context_create(casc, d, D.ctx[0] ? hc_context_stream(D.ctx[0]) : nullptr, &c));
      // the reference's call: scaleFactor 1.1, minNeighbors 2, minSize 30x30 (src/tracker/detector.cc:25-31)
      HC(hc_context_configure(c, rows, cols, 1.1, min_neighbors, 30, 30, 0, 0, B, 0));
    }
  }
  std::vector<int32_t> h_recv((size_t)N * B * RECORD_INTS);
  std::vector<hc_rect> h_orecv;
  const int per_global = N * B;
  const int nglobal = (nframes + per_global - 1) / per_global;

  long first_total = -1;
  double best = 0;
  std::map<int32_t, std::vector<hc_rect>> video;                 // frame id -> boxes, frame order
  for (int r = 0; r < rounds; r++) {
    video.clear();
    const auto t0 = std::chrono::steady_clock::now();
    // ---- feeder threads: device d runs its block of every global batch, two batches in flight
    std::vector<std::thread> feeders;
    for (int d = 0; d < N; d++) {
      feeders.emplace_back([&, d] {
        Device& D = dev[d];
        HIP(hipSetDevice(d));
        std::vector<hc_rect> out(1 << 18);
        std::vector<int32_t> ofs(B + 1), ids(B);
        auto block = [&](int g, int& lo, int& n) { lo = (g * N + d) * B; n = std::max(0, std::min(B, nframes - lo)); };
        auto enqueue = [&](int g) {
          int lo, n;
          block(g, lo, n);
          if (n > 0) HC(hc_detect_batch_enqueue_host(D.ctx[g & 1], frames + (size_t)lo * fbytes, fbytes, cols, n));
        };
        enqueue(0);
        for (int g = 0; g < nglobal; g++) {
          if (g + 1 < nglobal) enqueue(g + 1);
          int lo, n;
          block(g, lo, n);
          std::fill(ofs.begin(), ofs.end(), 0);
          if (n > 0) HC(hc_detect_batch_wait(D.ctx[g & 1], out.data(), (int)out.size(), ofs.data()));
          for (int i = 0; i < B; i++) { ids[i] = i < n ? lo + i : -1; if (i >= n) ofs[i + 1] = ofs[n]; }   // padding records
          Slot& S = D.slot[g & 1];
          std::unique_lock<std::mutex> l(D.mu);
          D.cv.wait(l, [&] { return !S.ready; });                // the gather of batch g-2 has consumed the slot
          hip_cascade::pack_records(ids.data(), B, ofs.data(), out.data(), S.records, S.overflow);
          S.ready = true;
          D.cv.notify_all();
        }
      });
    }
    // ---- main thread: one grouped all-gather of records per global batch (+ the overflow round when needed)
    for (int g = 0; g < nglobal; g++) {
      for (int d = 0; d < N; d++) {
        Device& D = dev[d];
        Slot& S = D.slot[g & 1];
        std::unique_lock<std::mutex> l(D.mu);
        D.cv.wait(l, [&] { return S.ready; });
        memcpy(D.h_send, S.records.data(), (size_t)B * RECORD_INTS * 4);
        HIP(hipSetDevice(d));
        HIP(hipMemcpyAsync(D.d_send, D.h_send, (size_t)B * RECORD_INTS * 4, hipMemcpyHostToDevice, D.comm));
      }
      NCCL(ncclGroupStart());
      for (int d = 0; d < N; d++) NCCL(ncclAllGather(dev[d].d_send, dev[d].d_recv, (size_t)B * RECORD_INTS, ncclInt32, comms[d], dev[d].comm));
      NCCL(ncclGroupEnd());
      HIP(hipSetDevice(0));
      HIP(hipMemcpyAsync(h_recv.data(), dev[0].d_recv, h_recv.size() * 4, hipMemcpyDeviceToHost, dev[0].comm));
      for (int d = 0; d < N; d++) { HIP(hipSetDevice(d)); HIP(hipStreamSynchronize(dev[d].comm)); }
      const std::vector<int64_t> over = hip_cascade::overflow_counts(h_recv.data(), N, B);
      size_t stride = 0;
      for (int64_t v : over) stride = std::max(stride, (size_t)v);
      if (stride > 0) {                                          // rare: some frame has more than 64 boxes
        for (int d = 0; d < N; d++) {
          Device& D = dev[d];
          HIP(hipSetDevice(d));
          if (D.ocap < stride) {
            if (D.d_osend) { HIP(hipFree(D.d_osend)); HIP(hipFree(D.d_orecv)); }
            HIP(hipMalloc((void**)&D.d_osend, stride * sizeof(hc_rect)));
            HIP(hipMalloc((void**)&D.d_orecv, (size_t)N * stride * sizeof(hc_rect)));
            D.ocap = stride;
          }
          const std::vector<hc_rect>& mine = D.slot[g & 1].overflow;
          if (!mine.empty()) HIP(hipMemcpyAsync(D.d_osend, mine.data(), mine.size() * sizeof(hc_rect), hipMemcpyHostToDevice, D.comm));
        }
        NCCL(ncclGroupStart());
        for (int d = 0; d < N; d++) NCCL(ncclAllGather(dev[d].d_osend, dev[d].d_orecv, stride * 4, ncclInt32, comms[d], dev[d].comm));
        NCCL(ncclGroupEnd());
        h_orecv.resize((size_t)N * stride);
        HIP(hipSetDevice(0));
        HIP(hipMemcpyAsync(h_orecv.data(), dev[0].d_orecv, h_orecv.size() * sizeof(hc_rect), hipMemcpyDeviceToHost, dev[0].comm));
        for (int d = 0; d < N; d++) { HIP(hipSetDevice(d)); HIP(hipStreamSynchronize(dev[d].comm)); }
      }
      for (int d = 0; d < N; d++) {                              // hand the slots back to the feeders
        Device& D = dev[d];
        std::lock_guard<std::mutex> l(D.mu);
        D.slot[g & 1].ready = false;
        D.cv.notify_all();
      }
      for (auto& kv : hip_cascade::unpack_records(h_recv.data(), N, B, h_orecv.data(), stride)) video[kv.first] = std::move(kv.second);
    }
    for (auto& t : feeders) t.join();
    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    long total = 0;
    for (auto& kv : video) total += (long)kv.second.size();
    if ((int)video.size() != nframes) { fprintf(stderr, "gather lost frames: %zu of %d\n", video.size(), nframes); return 3; }
    printf("round %d: %.1f frames/s on %d GPU(s), %ld boxes\n", r, nframes / dt, N, total);
    if (first_total < 0) first_total = total;
    if (total != first_total) { fprintf(stderr, "non-deterministic box count\n"); return 3; }
    if (r > 0 || rounds == 1) best = std::max(best, nframes / dt);
  }
  printf("best %.1f frames/s on %d GPU(s) (host frames in, gathered boxes out), %ld boxes per pass\n", best, N, first_total);
  if (out_path) {
    std::vector<int32_t> ids, ofs(1, 0);
    std::vector<hc_rect> rects;
    for (auto& kv : video) {
      ids.push_back(kv.first);
      rects.insert(rects.end(), kv.second.begin(), kv.second.end());
      ofs.push_back((int32_t)rects.size());
    }
    HC(hc_boxes_save(out_path, (int)ids.size(), ids.data(), ofs.data(), rects.data()));
    printf("wrote %s: %zu frames, %zu boxes\n", out_path, ids.size(), rects.size());
  }
  for (int d = 0; d < N; d++) {
    HIP(hipSetDevice(d));
    for (size_t k = sizeof(dev[d].ctx) / sizeof(dev[d].ctx[0]); k-- > 0;) hc_context_destroy(dev[d].ctx[k]);   // the stream's owner last
    NCCL(ncclCommDestroy(comms[d]));
  }
  hc_cascade_free(casc);
  hc_host_free(pool);
  return 0;
}
