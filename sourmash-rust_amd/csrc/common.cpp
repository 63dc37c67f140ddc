#include "common.hpp"

namespace smh {

LastError& last_error() {
  static thread_local LastError slot;
  return slot;
}

// display strings: reference src/errors.rs:9-19
void throw_mismatch(uint32_t code) {
  switch (code) {
    case kMismatchKSizes: throw Error(code, "different ksizes cannot be compared");
    case kMismatchDNAProt: throw Error(code, "DNA/prot minhashes cannot be compared");
    case kMismatchMaxHash: throw Error(code, "mismatch in max_hash; comparison fail");
    default: throw Error(kMismatchSeed, "mismatch in seed; comparison fail");
  }
}

// reference src/utils.rs:47-50 ("sourmash panicked: {}")
void throw_panic(const std::string& what) { throw Error(kPanic, "sourmash panicked: " + what); }

// reference src/errors.rs:6-7 ("internal error: {}")
void throw_internal(const std::string& what) { throw Error(kInternal, "internal error: " + what); }

}  // namespace smh
