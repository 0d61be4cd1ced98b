// ABI housekeeping of libdhr_hip.so: version / struct sizes, the per-thread error record, the exception classifier behind every
// extern "C" entry point (abi_guard.h) and the allocation-failure test hook.  Host code only.
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <new>

#include "../../include/dhr_hip.h"
#include "abi_guard.h"

namespace {
// The error record is a fixed buffer: recording a failure must not need the allocator that may just have failed.
thread_local char g_last_error[1024] = {0};

// Armed failure counter of the allocation hook: 0 = disarmed; n > 0 = the n-th host allocation from now throws std::bad_alloc.
std::atomic<long long> g_fail_in{0};
std::atomic<long long> g_allocs{0};        // host allocations of the library since it was loaded (dhr_debug_fail_alloc reports it)
struct EnvArm {
  EnvArm() {
    if (const char* e = getenv("DHR_TEST_FAIL_ALLOC")) g_fail_in.store(atoll(e));
  }
} g_env_arm;
}  // namespace

extern "C" int dhr_set_error_message(int code, const char* msg) {
  snprintf(g_last_error, sizeof(g_last_error), "%s", msg ? msg : "");
  return code;
}
extern "C" const char* dhr_last_error(void) { return g_last_error; }
extern "C" int dhr_version(void) { return DHR_VERSION; }
extern "C" void dhr_abi_sizes(int32_t out[4]) {
  if (!out) return;
  out[0] = (int32_t)sizeof(dhr_index_desc); out[1] = (int32_t)sizeof(dhr_query_batch);
  out[2] = (int32_t)sizeof(dhr_search_stats); out[3] = (int32_t)sizeof(dhr_file_info);
}
extern "C" int32_t dhr_abi_size(int32_t which) {
  switch (which) {
    case DHR_ABI_INDEX_DESC: return (int32_t)sizeof(dhr_index_desc);
    case DHR_ABI_QUERY_BATCH: return (int32_t)sizeof(dhr_query_batch);
    case DHR_ABI_SEARCH_STATS: return (int32_t)sizeof(dhr_search_stats);
    case DHR_ABI_FILE_INFO: return (int32_t)sizeof(dhr_file_info);
    case DHR_ABI_HOST_SHARD: return (int32_t)sizeof(dhr_host_shard);
  }
  return dhr_set_error_message(DHR_ERR_INVALID, "unknown struct id");
}

int dhr::on_exception() noexcept {
  try {
    throw;
  } catch (const std::bad_alloc&) {
    return dhr_set_error_message(DHR_ERR_NOMEM, "out of host memory (std::bad_alloc)");
  } catch (const std::exception& e) {
    char buf[900];
    snprintf(buf, sizeof(buf), "internal error: %s", e.what());
    return dhr_set_error_message(DHR_ERR_INTERNAL, buf);
  } catch (...) {
    return dhr_set_error_message(DHR_ERR_INTERNAL, "internal error: unknown C++ exception");
  }
}

void dhr::alloc_checkpoint() {
  g_allocs.fetch_add(1, std::memory_order_relaxed);
  long long v = g_fail_in.load(std::memory_order_relaxed);
  while (v > 0) {
    if (g_fail_in.compare_exchange_weak(v, v - 1, std::memory_order_relaxed)) {
      if (v == 1) throw std::bad_alloc();
      return;
    }
  }
}

// Test hook: arm the counter (n > 0: the n-th host allocation of the library from now fails once; 0: disarm).  Returns the number
// of host allocations the library has made since it was loaded, so that a test can count the allocations of a call and then fail
// each of them in turn.
extern "C" int64_t dhr_debug_fail_alloc(int64_t n) {
  g_fail_in.store(n > 0 ? n : 0);
  return (int64_t)g_allocs.load();
}

// Library-private operator new / delete: the linker's version script (libdhr.map) makes these six symbols LOCAL to libdhr_hip.so, so
// only the objects linked into this shared library bind to them (the process' global operator new is not interposed; <new> declares
// them with default visibility, which is why the attribute cannot sit here); plain malloc / free underneath, so memory may cross over
// to libstdc++'s own operators in either direction.
void* operator new(std::size_t n) {
  dhr::alloc_checkpoint();
  void* p = malloc(n ? n : 1);
  if (!p) throw std::bad_alloc();
  return p;
}
void* operator new[](std::size_t n) { return operator new(n); }
void operator delete(void* p) noexcept { free(p); }
void operator delete[](void* p) noexcept { free(p); }
void operator delete(void* p, std::size_t) noexcept { free(p); }
void operator delete[](void* p, std::size_t) noexcept { free(p); }
