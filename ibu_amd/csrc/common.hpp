// common.hpp — status/detail plumbing shared by the host and device halves of libibu_hip.so.
#pragma once
#include <errno.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <new>
#include <system_error>
#include <thread>
#include <vector>

#include "../../include/ibu_hip.h"

namespace ibu {

// Detail of the last failing call on this thread (the ABI's analogue of the Err(..) payload).
inline ibu_error_detail_t& tls_error() {
  static thread_local ibu_error_detail_t d;
  return d;
}

inline int32_t set_error(int32_t code, uint64_t a, uint64_t b, int os_errno, const char* fmt, ...) {
  ibu_error_detail_t& d = tls_error();
  d.code = code;
  d.a = a;
  d.b = b;
  d.os_errno = os_errno;
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(d.message, sizeof d.message, fmt, ap);
  va_end(ap);
  return code;
}

// Messages follow the reference's Display strings (src/error.rs:57-127).
inline int32_t err_io(int e, const char* what) {
  return set_error(IBU_ERR_IO, 0, 0, e, "I/O error: %s: %s", what, e ? strerror(e) : "unexpected end of file");
}
inline int32_t err_magic(uint32_t actual) {
  return set_error(IBU_ERR_INVALID_MAGIC, IBU_MAGIC, actual, 0,
                   "Invalid magic number, expected (%#x), found (%#x)", IBU_MAGIC, actual);
}
inline int32_t err_version(uint32_t actual) {
  return set_error(IBU_ERR_INVALID_VERSION, IBU_VERSION, actual, 0,
                   "Invalid version found, expected (%u), found (%u)", IBU_VERSION, actual);
}
inline int32_t err_bc_len(uint32_t len) {
  return set_error(IBU_ERR_INVALID_BC_LEN, len, 0, 0, "Invalid barcode length: %u (must be 1-32)", len);
}
inline int32_t err_umi_len(uint32_t len) {
  return set_error(IBU_ERR_INVALID_UMI_LEN, len, 0, 0, "Invalid UMI length: %u (must be 1-32)", len);
}
inline int32_t err_truncated(uint64_t pos) {
  return set_error(IBU_ERR_TRUNCATED_RECORD, pos, 0, 0, "Truncated record at position %llu",
                   (unsigned long long)pos);
}
inline int32_t err_map_size() {
  return set_error(IBU_ERR_INVALID_MAP_SIZE, 0, 0, 0, "Invalid map size - not a multiple of record size");
}
inline int32_t err_index(uint64_t idx, uint64_t max) {
  return set_error(IBU_ERR_INVALID_INDEX, idx, max, 0, "Invalid index (%llu) - Must be less than %llu",
                   (unsigned long long)idx, (unsigned long long)max);
}
inline int32_t err_process(uint64_t user_code) {
  return set_error(IBU_ERR_PROCESS, user_code, 0, 0, "Processing error: processor returned %llu",
                   (unsigned long long)user_code);
}
inline int32_t err_arg(const char* what) {
  return set_error(IBU_ERR_INVALID_ARG, 0, 0, 0, "Invalid argument: %s", what);
}
inline int32_t err_seq_len(uint32_t len) {
  return set_error(IBU_ERR_SEQ_LEN, len, 0, 0, "Invalid sequence length: %u (must be 1-32)", len);
}
inline int32_t err_niffler(const char* what) {
  return set_error(IBU_ERR_NIFFLER, 0, 0, 0, "Niffler error: %s", what);
}

// No C++ exception may cross the C ABI (the caller is Rust, C or ctypes: an escaping exception is std::terminate).
// Thread creation is where they come from on this path (std::system_error EAGAIN under a pids cgroup, bad_alloc):
// run_pieces runs fn(0..n-1) on up to n threads and NEVER throws — a piece whose thread cannot be started runs on the
// calling thread instead, so the call degrades to the single-threaded path.  fn itself must not throw.
template <class F>
inline void run_pieces(unsigned n, F&& fn) noexcept {
  if (n <= 1) { if (n) fn(0u); return; }
  std::vector<std::thread> th;
  unsigned started = 0;
  try {
    th.reserve(n - 1);
    for (; started + 1 < n; ++started) th.emplace_back(fn, started);
  } catch (...) {
    // `started` threads run; the rest is ours
  }
  for (unsigned i = started; i < n; ++i) fn(i);
  for (auto& t : th) t.join();
}
// Map whatever a C++ library call threw to a status; used as `catch (...) { return caught_io("what"); }` at the ABI.
inline int32_t caught_io(const char* what) {
  try { throw; }
  catch (const std::bad_alloc&) { return err_io(ENOMEM, what); }
  catch (const std::system_error& e) { return err_io(e.code().value() ? e.code().value() : EAGAIN, what); }
  catch (...) { return err_io(EIO, what); }
}

}  // namespace ibu
