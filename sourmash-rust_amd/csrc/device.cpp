#include "device.hpp"

#include <cstdlib>
#include <cstring>

namespace smh {

void throw_hip(hipError_t e, const char* what, const char* file, int line) {
  std::string m = std::string("HIP error ") + hipGetErrorName(e) + " (" + hipGetErrorString(e) +
                  ") in " + what + " at " + file + ":" + std::to_string(line);
  throw_internal(m);
}

void DeviceBuffer::ensure(size_t need) {
  if (need <= bytes) return;
  size_t want = need + need / 4 + 4096;
  if (ptr) { HIP_CHECK(hipFree(ptr)); ptr = nullptr; bytes = 0; }
  HIP_CHECK(hipMalloc(&ptr, want));
  bytes = want;
}
void DeviceBuffer::release() {
  if (ptr) (void)hipFree(ptr);
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
}

hipStream_t Device::user_stream(void* given) {
  if (given) return (hipStream_t)given;
  std::lock_guard<std::recursive_mutex> lock(mu_);
  HIP_CHECK(hipEventRecord(fence_, nullptr));
  HIP_CHECK(hipStreamWaitEvent(stream_, fence_, 0));
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
