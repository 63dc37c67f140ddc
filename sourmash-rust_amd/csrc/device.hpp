// device.hpp -- HIP device context of the host library: lazy initialisation (so that the
// shared object loads and every C-ABI symbol resolves on a machine without a GPU), the
// library's own stream, grow-only device buffers and per-kernel HIP-event timers.
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "common.hpp"

namespace smh {

[[noreturn]] void throw_hip(hipError_t e, const char* what, const char* file, int line);

#define HIP_CHECK(expr)                                                  \
  do {                                                                   \
    hipError_t e_ = (expr);                                              \
    if (e_ != hipSuccess) ::smh::throw_hip(e_, #expr, __FILE__, __LINE__); \
  } while (0)

// Device blocks come from a pool of freed blocks kept by size class (device.cpp).  `cap` is the
// block's real size; give it back with the same value.  sync = wait for the device first, as
// hipFree would (for blocks that work still in flight may be using).
void* device_pool_alloc(size_t need, size_t* cap);
void device_pool_free(void* ptr, size_t cap, bool sync);
void device_pool_trim();   // hipFree everything the pool holds
void device_pool_set_limit(size_t bytes);   // cap on the bytes parked in the pool (default 1 GiB, SOURMASH_AMD_POOL_MB)
size_t device_pool_bytes();

// grow-only device allocation (never shrinks; released with the context or explicitly)
struct DeviceBuffer {
  void* ptr = nullptr;
  size_t bytes = 0;
  void ensure(size_t need);
  void release();
  void release_after_sync();   // the caller has just waited for the device: no second wait
  template <class T> T* as() const { return reinterpret_cast<T*>(ptr); }
  DeviceBuffer() = default;
  DeviceBuffer(const DeviceBuffer&) = delete;
  DeviceBuffer& operator=(const DeviceBuffer&) = delete;
  ~DeviceBuffer();
};

// grow-only page-locked host staging area: D2H at link speed instead of through a bounce buffer
struct PinnedBuffer {
  void* ptr = nullptr;
  size_t bytes = 0;
  void ensure(size_t need);
  template <class T> T* as() const { return reinterpret_cast<T*>(ptr); }
  PinnedBuffer() = default;
  PinnedBuffer(const PinnedBuffer&) = delete;
  PinnedBuffer& operator=(const PinnedBuffer&) = delete;
  ~PinnedBuffer();
};

struct KernelTimes {
  double ms = 0.0;
  uint64_t launches = 0;
};

class Device {
 public:
  // Throws Error(kInternal) when no HIP device is usable: the product path never falls
  // back to the CPU.
  static Device& get();
  static bool available();  // true when a GPU can be initialised (no throw)

  int id() const { return device_; }
  int cu_count() const { return cus_; }
  // the library's own stream (ordered after work an earlier entry point left open on another stream, see leave_open)
  hipStream_t stream() { order_after_open(stream_); return stream_; }
  hipStream_t copy_stream() const { return copy_stream_; }   // host->device chunks that overlap the hashing; the block compare's fill
  // the pair of events with which a call forks work onto copy_stream() and joins it again (under mutex())
  hipEvent_t fork_event() const { return ev_fork_; }
  hipEvent_t join_event() const { return ev_join_; }
  // Every entry point returns with its device work complete -- except the few that say so (smh_collection_begin with
  // world == 1, smh_collection_finish without a gathered buffer: a single owner's dictionary is only ever used by later
  // calls of this library).  Those call leave_open(s) last: an event is recorded behind their work, and EVERY later entry
  // point, whatever stream it works on, first makes its stream wait for that event (order_after_open, called by stream()
  // and user_stream()).  So the library's shared scratch buffers (tiled_scratch(), scratch) are never rewritten by a call
  // on stream B while the open work on stream A still reads them; calls on the same stream are ordered anyway.
  void leave_open(hipStream_t s);
  void order_after_open(hipStream_t s);
  // The stream an entry point with a `void *stream` argument works on.  Non-null: the caller's.
  // Null: the library's own (non-blocking) stream, first ordered after everything already queued
  // on the legacy default stream -- a caller whose "current stream" is the default one (torch's
  // usual state, handle 0) may have produced the inputs there, e.g. an all-gather it just waited on.
  hipStream_t user_stream(void* given);
  std::recursive_mutex& mutex() { return mu_; }

  // scratch shared by the fold / sort primitives (guarded by mutex())
  DeviceBuffer scratch;
  // per-tile record numbers of the running sketch launch (guarded by mutex(); see k_tile_records)
  DeviceBuffer tile_rec;

  // HIP-event timing of named kernels (enabled by smh_profile_enable)
  void profile_enable(bool on);
  bool profiling() const { return profiling_; }
  void prof_begin(hipStream_t s);
  void prof_end(const char* name, hipStream_t s);
  void count(const char* name);                   // an event counter under the same names (launches += 1, always on)
  KernelTimes prof_get(const std::string& name);  // synchronises pending events
  void prof_reset();

 private:
  Device();
  int device_ = 0;
  int cus_ = 256;
  hipStream_t stream_ = nullptr;
  hipStream_t copy_stream_ = nullptr;
  hipEvent_t fence_ = nullptr;
  hipEvent_t ev_fork_ = nullptr, ev_join_ = nullptr;
  hipEvent_t open_event_ = nullptr;
  hipStream_t open_stream_ = nullptr;
  bool open_ = false;
  std::recursive_mutex mu_;
  bool profiling_ = false;
  struct Pending { std::string name; hipEvent_t a, b; };
  std::vector<Pending> pending_;
  hipEvent_t cur_start_ = nullptr;
  std::map<std::string, KernelTimes> times_;
  void drain();
};

}  // namespace smh
