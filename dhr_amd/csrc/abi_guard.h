// The exception barrier of the C ABI ("no C++ exceptions or exit() cross the boundary").
//
// Every function with C linkage that this library defines is a function-try-block
//
//     extern "C" int dhr_xxx(...) try {
//       ...
//     } DHR_CATCH_STATUS
//
// so that a std::bad_alloc from `new dhr_index()`, a std::vector / std::string that cannot grow, a std::system_error from
// std::thread, or anything a header-only dependency (hipCUB in select_global.hip) throws comes back as a negative dhr_status with
// dhr_last_error() set, instead of unwinding through ctypes / cgo / JNI frames (which ends in std::terminate).  The handler is one
// shared function (dhr::on_exception: rethrow-and-classify), the error record is a fixed thread-local buffer (abi.cpp), so the
// failure path itself does not allocate.  tests/test_abi_guard.py checks that every extern "C" definition in csrc/ carries the
// block, and drives real allocation failures through the entry points with the test hook below.
//
// Test hook: the library's own translation units see a library-PRIVATE (hidden visibility) replacement of operator new (abi.cpp)
// that counts allocations down and throws std::bad_alloc when an armed counter reaches zero -- dhr_debug_fail_alloc(n), or the
// environment variable DHR_TEST_FAIL_ALLOC=n read when the library is loaded.  Disarmed (the default) it is malloc.
#pragma once
#include <stdint.h>

namespace dhr {
// Inside a catch (...) handler only: classifies the exception in flight, records the message, returns the status
// (std::bad_alloc -> DHR_ERR_NOMEM, everything else -> DHR_ERR_INTERNAL).  Never throws.
int on_exception() noexcept;
// Counts as one host allocation of the armed failure counter (abi.cpp); entry points call it before they touch the device so
// that the injection also reaches calls that fail early on a host without a GPU.
void alloc_checkpoint();
}  // namespace dhr

// dhr_mem_kind arguments: any value but the two enumerators is refused (until round 6 an unknown kind was taken for "device": a host pointer
// handed to a kernel is a GPU fault that aborts the process, not a status).
#define DHR_MEM_KIND_OK(k) ((k) == DHR_MEM_HOST || (k) == DHR_MEM_DEVICE)

#define DHR_CATCH_STATUS catch (...) { return dhr::on_exception(); }
#define DHR_CATCH_VALUE(v) catch (...) { (void)dhr::on_exception(); return (v); }
#define DHR_CATCH_VOID catch (...) { (void)dhr::on_exception(); }
