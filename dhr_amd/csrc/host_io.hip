// Host-side text output of the path: the TREC run file (reference: retrieval/gip_retrieval.py:333-342 -- one Python
// '{} Q0 {} {} {} {}\n'.format(...) per result line on one thread: 14 s for 6 980 x 1 000 lines, ninety times the search itself).
// No device code in this translation unit; it lives in the library so that a binding gets the writer with the search.
#include "../../include/dhr_hip.h"
#include "abi_guard.h"
#include <charconv>
#include <cstdio>
#include <cstring>
#include <cmath>
#include <string>
#include <thread>
#include <vector>
#include <algorithm>

extern "C" int dhr_set_error_message(int code, const char* msg);

namespace {

// Python's repr(float) / '{}'.format(float) (float_repr_style 'short': PyOS_double_to_string(x, 'r', 0, Py_DTSF_ADD_DOT_0)):
// the shortest digit string that round-trips, fixed notation while -4 < decpt <= 16, else d[.ddd]e+XX with at least two exponent digits.
inline char* put_py_float(char* p, double x) {
  if (std::isnan(x)) { memcpy(p, "nan", 3); return p + 3; }
  if (std::isinf(x)) { if (x < 0) *p++ = '-'; memcpy(p, "inf", 3); return p + 3; }
  if (std::signbit(x)) { *p++ = '-'; x = -x; }
  if (x == 0.0) { memcpy(p, "0.0", 3); return p + 3; }
  char buf[48];
  auto res = std::to_chars(buf, buf + sizeof(buf), x, std::chars_format::scientific);      // d[.ddd]e[+-]XX, shortest round trip
  char* e = buf;
  while (e < res.ptr && *e != 'e') ++e;
  char digits[24];
  int nd = 0;
  for (char* c = buf; c < e; ++c)
    if (*c != '.') digits[nd++] = *c;
  int ex = 0;
  { const char* c = e + 1; const bool neg = *c == '-'; if (*c == '-' || *c == '+') ++c; while (c < res.ptr) ex = ex * 10 + (*c++ - '0'); if (neg) ex = -ex; }
  const int decpt = ex + 1;                        // value = 0.d1d2... x 10^decpt
  if (decpt > -4 && decpt <= 16) {
    if (decpt <= 0) {
      *p++ = '0'; *p++ = '.';
      for (int i = 0; i < -decpt; ++i) *p++ = '0';
      memcpy(p, digits, nd); p += nd;
    } else if (decpt >= nd) {
      memcpy(p, digits, nd); p += nd;
      for (int i = nd; i < decpt; ++i) *p++ = '0';
      *p++ = '.'; *p++ = '0';
    } else {
      memcpy(p, digits, decpt); p += decpt;
      *p++ = '.';
      memcpy(p, digits + decpt, nd - decpt); p += nd - decpt;
    }
  } else {
    *p++ = digits[0];
    if (nd > 1) { *p++ = '.'; memcpy(p, digits + 1, nd - 1); p += nd - 1; }
    *p++ = 'e';
    int x10 = decpt - 1;
    *p++ = x10 < 0 ? '-' : '+';
    if (x10 < 0) x10 = -x10;
    char t[8]; int nt = 0;
    do { t[nt++] = (char)('0' + x10 % 10); x10 /= 10; } while (x10);
    if (nt < 2) t[nt++] = '0';
    while (nt) *p++ = t[--nt];
  }
  return p;
}
inline char* put_uint(char* p, uint64_t v) {
  char t[24]; int n = 0;
  do { t[n++] = (char)('0' + v % 10); v /= 10; } while (v);
  while (n) *p++ = t[--n];
  return p;
}

}  // namespace

extern "C" int dhr_format_float(double x, char* out, int32_t cap) try {
  char buf[64];
  const int n = (int)(put_py_float(buf, x) - buf);
  if (!out || cap <= n) return dhr_set_error_message(DHR_ERR_INVALID, "buffer too small");
  memcpy(out, buf, n); out[n] = 0;
  return n;
} DHR_CATCH_STATUS

extern "C" int dhr_write_trec(const char* path, int32_t append, int64_t n_queries, int64_t k, const char* qid_blob, const int64_t* qid_off,
                              const char* docid_blob, const int64_t* docid_off, int64_t n_docs, const int64_t* rows, int64_t row_base,
                              const float* scores, const char* run_name, int32_t id_sep_bytes, int32_t n_threads, int64_t* lines_out) try {
  if (!path || !qid_blob || !qid_off || !docid_blob || !docid_off || !rows || !scores || !run_name || n_queries < 0 || k < 0 || n_docs < 0 || id_sep_bytes < 0)
    return dhr_set_error_message(DHR_ERR_INVALID, "null argument / negative size");
  const size_t run_len = strlen(run_name);
  int T = n_threads > 0 ? n_threads : (int)std::min<unsigned>(std::max(1u, std::thread::hardware_concurrency()), 64u);
  T = (int)std::max<int64_t>(1, std::min<int64_t>(T, n_queries));
  std::vector<std::string> part((size_t)T);
  std::vector<int64_t> lines((size_t)T, 0);
  std::vector<int> bad((size_t)T, 0);
  auto work_body = [&](int t) {
    const int64_t lo = n_queries * t / T, hi = n_queries * (t + 1) / T;
    std::string& out = part[t];
    out.reserve((size_t)((hi - lo) * k) * 48 + 64);
    char line[128];
    for (int64_t q = lo; q < hi; ++q) {
      const char* qs = qid_blob + qid_off[q];
      const size_t ql = (size_t)(qid_off[q + 1] - qid_off[q] - id_sep_bytes);
      int64_t rank = 0;                                       // position in the compacted list (padding dropped), gip_retrieval.py:336
      for (int64_t j = 0; j < k; ++j) {
        const int64_t r = rows[q * k + j];
        if (r < 0) continue;                                  // padding of a short list
        const int64_t d = r - row_base;
        if (d < 0 || d >= n_docs) { bad[t] = 1; return; }
        ++rank;
        const char* ds = docid_blob + docid_off[d];
        const size_t dl = (size_t)(docid_off[d + 1] - docid_off[d] - id_sep_bytes);
        if (dl == ql && memcmp(ds, qs, ql) == 0) continue;    // docid == query_id: skipped, the rank keeps its gap (:340)
        out.append(qs, ql);
        out.append(" Q0 ", 4);
        out.append(ds, dl);
        char* p = line;
        *p++ = ' ';
        p = put_uint(p, (uint64_t)rank);
        *p++ = ' ';
        p = put_py_float(p, (double)scores[q * k + j]);
        *p++ = ' ';
        out.append(line, (size_t)(p - line));
        out.append(run_name, run_len);
        out.push_back('\n');
        ++lines[t];
      }
    }
  };
  // (an exception must not leave a thread function -- std::terminate --: a worker that runs out of memory marks its part and returns)
  auto work = [&](int t) noexcept {
    try { work_body(t); } catch (const std::bad_alloc&) { bad[t] = 2; } catch (...) { bad[t] = 3; }
  };
  {
    std::vector<std::thread> th;
    th.reserve((size_t)T);
    try {
      for (int t = 1; t < T; ++t) th.emplace_back(work, t);
    } catch (...) {              // thread creation failed (std::system_error) or the vector could not grow: join what runs, then report
      for (auto& x : th) x.join();
      throw;
    }
    work(0);
    for (auto& x : th) x.join();
  }
  for (int t = 0; t < T; ++t) {
    if (bad[t] == 1) return dhr_set_error_message(DHR_ERR_INVALID, "a result row lies outside the docid list");
    if (bad[t] == 2) return dhr_set_error_message(DHR_ERR_NOMEM, "out of host memory while formatting the run file");
    if (bad[t]) return dhr_set_error_message(DHR_ERR_INTERNAL, "a formatting thread failed");
  }
  FILE* f = fopen(path, append ? "ab" : "wb");
  if (!f) return dhr_set_error_message(DHR_ERR_INVALID, "cannot open the output file");
  int64_t total = 0;
  for (int t = 0; t < T; ++t) {
    if (!part[t].empty() && fwrite(part[t].data(), 1, part[t].size(), f) != part[t].size()) { fclose(f); return dhr_set_error_message(DHR_ERR_INVALID, "short write"); }
    total += lines[t];
  }
  if (fclose(f) != 0) return dhr_set_error_message(DHR_ERR_INVALID, "close failed");
  if (lines_out) *lines_out = total;
  return DHR_OK;
} DHR_CATCH_STATUS
