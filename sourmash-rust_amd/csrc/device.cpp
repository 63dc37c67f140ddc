#include "device.hpp"

#include <cstdlib>
#include <cstring>

namespace smh {

void throw_hip(hipError_t e, const char* what, const char* file, int line) {
  std::string m = std::string("HIP error ") + hipGetErrorName(e) + " (" + hipGetErrorString(e) +
                  ") in " + what + " at " + file + ":" + std::to_string(line);
  throw_internal(m);
}

// ---- block pool -----------------------------------------------------------------------------
// hipMalloc / hipFree of the buffers a sketch is made of cost hundreds of microseconds each at
// the sizes of the benchmark configurations (a 10 GB batch leaves 80 MB of hashes, the protein
// share 270 MB) -- more than the sort that fills them.  Freed blocks are kept per DEVICE and size
// class (eight classes per power of two: at most 12.5 % slack) and handed out again.
//  * The pool is invisible to any other allocator in the process (PyTorch's caching allocator in
//    bench.py / distributed.py): memory parked here is memory torch cannot use.  It is therefore
//    capped -- 1 GiB by default, SOURMASH_AMD_POOL_MB=<MiB> (0 = no pooling) or smh_pool_set_limit()
//    change it -- and smh_release_workspace() hands everything back.
//  * A block freed with sync=false may be handed out again at once.  That is safe for the one caller
//    that does it (DeviceMirror) because every entry point that reads a mirror returns with its device
//    work complete (include/sourmash_amd.h; the two entry points that leave work open -- a single owner's
//    collection dictionary, Device::leave_open -- read no mirror): no kernel of a finished call can still
//    be reading it.
namespace {
std::mutex g_pool_mu;
std::map<std::pair<int, size_t>, std::vector<void*>> g_pool;
size_t g_pool_bytes = 0;
size_t g_pool_limit = [] {
  if (const char* e = std::getenv("SOURMASH_AMD_POOL_MB")) return (size_t)std::strtoull(e, nullptr, 10) << 20;
  return (size_t)1 << 30;
}();

size_t size_class(size_t need) {
  if (need <= 4096) return 4096;
  const int lg = 63 - __builtin_clzll((unsigned long long)(need - 1));   // need in (2^lg, 2^(lg+1)]
  const size_t step = (size_t)1 << (lg - 3);
  return (need + step - 1) / step * step;
}
int current_device() {
  int d = 0;
  (void)hipGetDevice(&d);
  return d;
}
}  // namespace

void* device_pool_alloc(size_t need, size_t* cap) {
  const size_t c = size_class(need);
  {
    std::lock_guard<std::mutex> lock(g_pool_mu);
    auto it = g_pool.find({current_device(), c});
    if (it != g_pool.end() && !it->second.empty()) {
      void* p = it->second.back();
      it->second.pop_back();
      g_pool_bytes -= c;
      *cap = c;
      return p;
    }
  }
  void* p = nullptr;
  hipError_t e = hipMalloc(&p, c);
  if (e == hipErrorOutOfMemory) {                         // give the pooled blocks back and try once more
    (void)hipGetLastError();
    device_pool_trim();
    e = hipMalloc(&p, c);
  }
  HIP_CHECK(e);
  *cap = c;
  return p;
}

void device_pool_free(void* ptr, size_t cap, bool sync) {
  if (!ptr) return;
  // hipFree waits for the device; a block that goes back to the pool instead may be handed to work on
  // another stream at once, so the callers whose block may still be in use keep that wait
  if (sync) (void)hipDeviceSynchronize();
  if (cap >= 4096 && size_class(cap) == cap) {
    std::lock_guard<std::mutex> lock(g_pool_mu);
    if (g_pool_bytes + cap <= g_pool_limit) {
      g_pool[{current_device(), cap}].push_back(ptr);
      g_pool_bytes += cap;
      return;
    }
  }
  (void)hipFree(ptr);
}

void device_pool_trim() {
  std::lock_guard<std::mutex> lock(g_pool_mu);
  for (auto& kv : g_pool) for (void* p : kv.second) (void)hipFree(p);
  g_pool.clear();
  g_pool_bytes = 0;
}

void device_pool_set_limit(size_t bytes) {
  {
    std::lock_guard<std::mutex> lock(g_pool_mu);
    g_pool_limit = bytes;
    if (g_pool_bytes <= bytes) return;
  }
  device_pool_trim();
}
size_t device_pool_bytes() {
  std::lock_guard<std::mutex> lock(g_pool_mu);
  return g_pool_bytes;
}

void DeviceBuffer::ensure(size_t need) {
  if (need <= bytes) return;
  const size_t want = need + need / 4 + 4096;
  release();
  ptr = device_pool_alloc(want, &bytes);
}
void DeviceBuffer::release() {
  if (ptr) device_pool_free(ptr, bytes, true);
  ptr = nullptr;
  bytes = 0;
}
void DeviceBuffer::release_after_sync() {
  if (ptr) device_pool_free(ptr, bytes, false);
  ptr = nullptr;
  bytes = 0;
}
DeviceBuffer::~DeviceBuffer() { release(); }

void PinnedBuffer::ensure(size_t need) {
  if (need <= bytes) return;
  size_t want = need + need / 4 + 4096;
  if (ptr) { HIP_CHECK(hipHostFree(ptr)); ptr = nullptr; bytes = 0; }
  HIP_CHECK(hipHostMalloc(&ptr, want, hipHostMallocDefault));
  bytes = want;
}
PinnedBuffer::~PinnedBuffer() { if (ptr) (void)hipHostFree(ptr); }

static int pick_device() {
  // SOURMASH_AMD_DEVICE selects explicitly; otherwise keep the caller's current device (a
  // torch.distributed rank has already called hipSetDevice(LOCAL_RANK) through torch).
  if (const char* e = std::getenv("SOURMASH_AMD_DEVICE")) return std::atoi(e);
  int cur = 0;
  if (hipGetDevice(&cur) == hipSuccess) return cur;
  return 0;
}

Device::Device() {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0)
    throw_internal("no HIP device available: the sourmash MI355X library needs a GPU (" +
                   std::string(e == hipSuccess ? "device count is 0" : hipGetErrorString(e)) + ")");
  device_ = pick_device();
  if (device_ < 0 || device_ >= n) device_ = 0;
  HIP_CHECK(hipSetDevice(device_));
  hipDeviceProp_t p;
  HIP_CHECK(hipGetDeviceProperties(&p, device_));
  cus_ = p.multiProcessorCount > 0 ? p.multiProcessorCount : 256;
  HIP_CHECK(hipStreamCreateWithFlags(&stream_, hipStreamNonBlocking));
  HIP_CHECK(hipStreamCreateWithFlags(&copy_stream_, hipStreamNonBlocking));
  HIP_CHECK(hipEventCreateWithFlags(&fence_, hipEventDisableTiming));
  HIP_CHECK(hipEventCreateWithFlags(&ev_fork_, hipEventDisableTiming));
  HIP_CHECK(hipEventCreateWithFlags(&ev_join_, hipEventDisableTiming));
  HIP_CHECK(hipEventCreateWithFlags(&open_event_, hipEventDisableTiming));
}

void Device::leave_open(hipStream_t s) {
  std::lock_guard<std::recursive_mutex> lock(mu_);
  HIP_CHECK(hipEventRecord(open_event_, s));
  open_stream_ = s;
  open_ = true;
}

void Device::order_after_open(hipStream_t s) {
  std::lock_guard<std::recursive_mutex> lock(mu_);
  if (!open_) return;
  if (hipEventQuery(open_event_) == hipSuccess) { open_ = false; return; }
  if (s != open_stream_) HIP_CHECK(hipStreamWaitEvent(s, open_event_, 0));
}

hipStream_t Device::user_stream(void* given) {
  std::lock_guard<std::recursive_mutex> lock(mu_);
  if (given) { order_after_open((hipStream_t)given); return (hipStream_t)given; }
  HIP_CHECK(hipEventRecord(fence_, nullptr));
  HIP_CHECK(hipStreamWaitEvent(stream_, fence_, 0));
  order_after_open(stream_);
  return stream_;
}

Device& Device::get() {
  static Device* d = new Device();  // leaked on purpose: no teardown-order games with the HIP runtime
  // a different host thread may have another current device
  (void)hipSetDevice(d->device_);
  return *d;
}

bool Device::available() {
  try {
    (void)get();
    return true;
  } catch (const Error&) {
    return false;
  }
}

void Device::profile_enable(bool on) { profiling_ = on; }

void Device::prof_begin(hipStream_t s) {
  if (!profiling_) return;
  HIP_CHECK(hipEventCreate(&cur_start_));
  HIP_CHECK(hipEventRecord(cur_start_, s));
}

void Device::prof_end(const char* name, hipStream_t s) {
  if (!profiling_ || !cur_start_) return;
  hipEvent_t b;
  HIP_CHECK(hipEventCreate(&b));
  HIP_CHECK(hipEventRecord(b, s));
  pending_.push_back({name, cur_start_, b});
  cur_start_ = nullptr;
  if (pending_.size() > 4096) drain();
}

void Device::drain() {
  for (auto& p : pending_) {
    float ms = 0.f;
    if (hipEventSynchronize(p.b) == hipSuccess && hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) {
      auto& t = times_[p.name];
      t.ms += ms;
      t.launches += 1;
    }
    (void)hipEventDestroy(p.a);
    (void)hipEventDestroy(p.b);
  }
  pending_.clear();
}

void Device::count(const char* name) {
  std::lock_guard<std::recursive_mutex> lock(mu_);
  times_[name].launches += 1;
}

KernelTimes Device::prof_get(const std::string& name) {
  drain();
  auto it = times_.find(name);
  return it == times_.end() ? KernelTimes{} : it->second;
}

void Device::prof_reset() {
  drain();
  times_.clear();
}

}  // namespace smh
