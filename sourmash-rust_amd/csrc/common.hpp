// common.hpp -- error model, small utilities shared by the host library.
//
// Error codes and messages restate the reference's src/errors.rs:3-50; the thread-local
// "last error" slot restates src/utils.rs:14-16,52-124.
#pragma once
#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

namespace smh {

enum ErrorCode : uint32_t {
  kNoError = 0,
  kPanic = 1,
  kInternal = 2,
  kMsg = 3,
  kUnknown = 4,
  kMismatchKSizes = 101,
  kMismatchDNAProt = 102,
  kMismatchMaxHash = 103,
  kMismatchSeed = 104,
  kInvalidDNA = 1101,
  kInvalidProt = 1102,
  kIo = 100001,
  kUtf8Error = 100002,
  kParseInt = 100003,
  kSerdeError = 100004,
};

// What a Rust `Err(e)` / panic is in the reference becomes a thrown Error here; the FFI
// landing pad (ffi.cpp) stores it in the thread-local slot exactly like utils.rs:154-166.
struct Error : std::exception {
  uint32_t code;
  std::string message;
  Error(uint32_t c, std::string m) : code(c), message(std::move(m)) {}
  const char* what() const noexcept override { return message.c_str(); }
};

[[noreturn]] void throw_mismatch(uint32_t code);          // 101..104 with the reference's text
[[noreturn]] void throw_panic(const std::string& what);    // "sourmash panicked: ..."
[[noreturn]] void throw_internal(const std::string& what); // "internal error: ..."

struct LastError {
  bool set = false;
  uint32_t code = 0;
  std::string message;
};
LastError& last_error();  // thread-local

}  // namespace smh
