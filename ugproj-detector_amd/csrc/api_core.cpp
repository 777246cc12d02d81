// api_core.cpp -- what every translation unit behind the C ABI (include/hip_cascade.h) leans on: the thread's error text,
// counted allocations, the post-processing worker pool, page-locked host memory for the caller's frame pool.
//
// The library replaces, behind a plain-C boundary, the two OpenCV calls the reference makes on its cv::CascadeClassifier:
// load (src/file_io.cc:11) and detectMultiScale (src/tracker/detector.cc:25-31).  No CPU fallback exists: every compute
// entry point needs a HIP device and reports HC_ERR_NO_DEVICE / HC_ERR_HIP otherwise.

#include <atomic>
#include <cstdlib>

#include "hc_context.h"

namespace {
thread_local std::string g_err;
std::atomic<long long> g_alloc_count{0};
}  // namespace

namespace hc {

int set_err(int code, const std::string& what) { g_err = what; return code; }
const std::string& last_err() { return g_err; }
int set_error(int code, const std::string& what) { return set_err(code, what); }   // hc_internal.h spelling
void count_alloc() { g_alloc_count++; }

int alloc_bytes(void** p, size_t bytes) {
  *p = nullptr;
  if (bytes == 0) bytes = 1;
  g_alloc_count++;
  hipError_t e = hipMalloc(p, bytes);
  if (e != hipSuccess) return set_err(HC_ERR_HIP, std::string("hipMalloc: ") + hipGetErrorString(e));
  return HC_OK;
}

int upload_bytes(void** p, const void* src, size_t bytes) {
  int rc = alloc_bytes(p, bytes);
  if (rc) return rc;
  if (bytes) HIPCHK(hipMemcpy(*p, src, bytes, hipMemcpyHostToDevice));
  return HC_OK;
}

int host_alloc_coherent(void** p, size_t bytes) {
  *p = nullptr;
  g_alloc_count++;
  HIPCHK(hipHostMalloc(p, bytes ? bytes : 1, hipHostMallocCoherent | hipHostMallocMapped));
  return HC_OK;
}

int deliver_sync_result(hc_cascade* c, const std::vector<hc_rect>& boxes, hc_rect* out, int out_capacity, int* out_count) {
  c->pending_results.erase(std::this_thread::get_id());
  *out_count = (int)boxes.size();
  if ((int)boxes.size() > out_capacity || (!out && !boxes.empty())) {
    c->pending_results[std::this_thread::get_id()] = boxes;   // the detection is done: the caller re-reads it instead of running it again
    return set_err(HC_ERR_CAPACITY, "output rectangle buffer too small (*out_count holds the size needed; "
                                    "hc_detect_multiscale_reread returns the rectangles without running the detection again)");
  }
  for (size_t i = 0; i < boxes.size(); i++) out[i] = boxes[i];
  return HC_OK;
}

}  // namespace hc

WorkerPool::WorkerPool(int nthreads) {
  for (int i = 0; i < nthreads; i++) workers_.emplace_back([this] { loop(); });
}

WorkerPool::~WorkerPool() {
  { std::lock_guard<std::mutex> l(mu_); stop_ = true; ++gen_; }
  cv_.notify_all();
  for (auto& t : workers_) t.join();
}

void WorkerPool::parallel_for(int n, const std::function<void(int)>& fn) {
  if (n <= 0) return;
  { std::lock_guard<std::mutex> l(mu_); fn_ = &fn; n_ = n; next_ = 0; active_ = (int)workers_.size(); ++gen_; }
  cv_.notify_all();
  drain();
  std::unique_lock<std::mutex> l(mu_);
  done_.wait(l, [this] { return active_ == 0; });
  fn_ = nullptr;
}

void WorkerPool::drain() {
  for (;;) {
    const int i = next_.fetch_add(1);
    if (i >= n_) break;
    (*fn_)(i);
  }
}

void WorkerPool::loop() {
  unsigned long long seen = 0;
  for (;;) {
    {
      std::unique_lock<std::mutex> l(mu_);
      cv_.wait(l, [&] { return gen_ != seen; });
      seen = gen_;
      if (stop_) return;
    }
    drain();
    { std::lock_guard<std::mutex> l(mu_); if (--active_ == 0) done_.notify_all(); }
  }
}

extern "C" {

const char* hc_last_error(void) { return g_err.c_str(); }
const char* hc_version(void) { return "hip_cascade 0.1 (gfx950)"; }
long long hc_debug_alloc_count(void) { return g_alloc_count.load(); }

int hc_host_alloc(size_t bytes, void** out) {
  if (!out || bytes == 0) return hc::set_err(HC_ERR_ARG, "hc_host_alloc: bad arguments");
  *out = nullptr;
  HIPCHK(hipHostMalloc(out, bytes, hipHostMallocDefault));
  return HC_OK;
}

void hc_host_free(void* p) {
  if (p) (void)hipHostFree(p);
}

int hc_host_register(void* p, size_t bytes) {
  if (!p || bytes == 0) return hc::set_err(HC_ERR_ARG, "hc_host_register: bad arguments");
  HIPCHK(hipHostRegister(p, bytes, hipHostRegisterDefault));
  return HC_OK;
}

int hc_host_unregister(void* p) {
  if (!p) return hc::set_err(HC_ERR_ARG, "hc_host_unregister: null pointer");
  HIPCHK(hipHostUnregister(p));
  return HC_OK;
}

}  // extern "C"
