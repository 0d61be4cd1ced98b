// Product-quantised first stage of --PQIP as an ADC scan (reference: retrieval/gip_retrieval.py:167-231
// -- faiss IndexPQ(d, M, nbits, METRIC_INNER_PRODUCT).search(queries, agip_topk); retrieval/quantize_index.py:27-37 builds it).
// faiss is a third-party dependency that is not in /root/reference: this restates its published algorithm (asymmetric distance
// computation, Jegou et al.): score(q, x) = sum_m <q_m, c_m[code_m(x)]>.  PARITY WITH FAISS IS UNPINNED (no golden vectors).
//
// What lives on the device: the codes, ONE BYTE PER SUB-QUANTISER AND ROW ([N][M] uint8: 64 B per row at the reference's M = 64 --
// 566 MB for 8.84 M rows instead of the 27 GB of a decoded fp16 copy) and the codebooks.  A search
//   1. builds the per-query lookup tables LUT[q][m][j] = <q_m, c_m[j]> in fp32 (adc_lut_kernel), stored PAIR-interleaved
//      ([pair][m][j][2]) so that one 8-byte LDS read serves two queries;
//   2. scans the codes (adc_scan_kernel): a workgroup holds the tables of one query pair in LDS (2 x M x 2^nbits x 4 B = 128 KiB
//      at M = 64) and streams a block of rows -- 64 B of codes per row, coalesced, 64 table reads per row for the two queries --
//      fused with the per-query threshold filter: a (row, score) pair reaches HBM only when it beats the query's running k-th
//      best score.  The kernel is bound by the LDS gather (2^nbits-entry tables indexed by data: ~2-way bank conflicts on
//      average) when many queries share the pass, by the code stream from HBM otherwise; bench.py --workload beir --pq reports
//      the achieved code bytes / s.
//   3. keeps a running top-k per query (select_kernel of the exact search, scores as keys), raising the thresholds after
//      every block of rows; blocks double in size (the number of rows that beat a running k-th best is ~k ln(N/N0)), a block
//      whose candidate lists overflow is re-run in halves.
// Scores are plain fp32 sums in table order m = 0..M-1: deterministic and independent of the blocking.
#include <algorithm>
#include <string>
#include <vector>

#include "dhr_internal.h"
#include <mutex>

using namespace dhr;

struct dhr_pq {
  int device = 0;
  int64_t n = 0, row_offset = 0;
  int d = 0, M = 0, nbits = 8, ksub = 256, dsub = 0;
  float* cb = nullptr;        // [M][ksub][dsub]
  uint8_t* codes = nullptr;   // [n][M]
  // search workspace (grow-only)
  int q_cap = 0, kp = 0;
  float* lut = nullptr;       // [q_cap / 2][M][ksub][2]
  float* q32 = nullptr;       // [q_cap][d]
  uint2* cand = nullptr;      // [q_cap][CAP]
  uint32_t* cnt = nullptr;
  uint64_t* keys = nullptr;   // [q_cap][CAP]
  uint64_t* topk = nullptr;   // [q_cap][kp]
  float *thr = nullptr, *tau = nullptr, *margin = nullptr;
  uint32_t* d_max = nullptr;
  int64_t bytes = 0;
  double last_scan_ms = 0, last_code_bytes = 0;
};

namespace {

constexpr int ADC_CAP = 32768;          // candidate list depth per query
constexpr int ADC_Q_CHUNK = 1024;       // queries per pass over the codes (bounds the workspace: 0.77 GB)
constexpr int ADC_THREADS = 1024;        // 16 waves share one pair's tables: with 256 threads the 128 KiB of LDS left ONE wave per SIMD to hide the gather latency (0.27 of the HBM rate)
constexpr int ADC_ROWS_PER_WG = 32768;

#define PQ_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return dhr_set_error_message(DHR_ERR_HIP, (std::string(#x) + ": " + hipGetErrorString(e_)).c_str()); } while (0)

// queries (fp16 or fp32, any leading dimension) -> dense fp32 [Q][d]
__global__ void adc_q32_kernel(const void* __restrict__ src, int is_f32, int64_t ld, int n_queries, int d, float* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)n_queries * d) return;
  const int q = (int)(i / d), c = (int)(i - (int64_t)q * d);
  out[i] = is_f32 ? ((const float*)src)[(int64_t)q * ld + c] : __half2float(((const __half*)src)[(int64_t)q * ld + c]);
}
// LUT[pair][m][j][q & 1] = sum_t q[m*dsub + t] * cb[m][j][t]     (fp32, t ascending)
__global__ void __launch_bounds__(256) adc_lut_kernel(const float* __restrict__ q32, int n_queries, int d, int M, int ksub, int dsub,
                                                      const float* __restrict__ cb, float* __restrict__ lut) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;      // (q, m, j)
  const int64_t per_q = (int64_t)M * ksub;
  if (i >= (int64_t)n_queries * per_q) return;
  const int q = (int)(i / per_q);
  const int mj = (int)(i - (int64_t)q * per_q);
  const int m = mj / ksub;
  const float* qs = q32 + (int64_t)q * d + m * dsub;
  const float* c = cb + (int64_t)mj * dsub;
  float s = 0.f;
  for (int t = 0; t < dsub; ++t) s = fmaf(qs[t], c[t], s);
  lut[((int64_t)(q >> 1) * per_q + mj) * 2 + (q & 1)] = s;
}
// grid (query pairs, row blocks): consecutive workgroups share the row block (L2), each with another pair's tables.
// thr == nullptr: dump mode (scores of rows [row_lo, row_hi) to out[q][row - row_lo]).
__global__ void __launch_bounds__(ADC_THREADS) adc_scan_kernel(const uint8_t* __restrict__ codes, int M, int ksub, int64_t row_lo, int64_t row_hi,
                                                       const float* __restrict__ lut, int n_queries, const float* __restrict__ thr,
                                                       uint2* __restrict__ cand, uint32_t* __restrict__ cnt, uint32_t cap,
                                                       float* __restrict__ dump, int64_t dump_ld) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float2* t = (float2*)smem;                                 // [M][ksub] (query 2p, query 2p+1)
  const int pair = blockIdx.x;
  const int64_t per_q = (int64_t)M * ksub;
  const float4* src = (const float4*)(lut + (int64_t)pair * per_q * 2);
  for (int i = threadIdx.x; i < per_q / 2; i += ADC_THREADS) ((float4*)t)[i] = src[i];
  __syncthreads();
  const int q0 = 2 * pair, q1 = q0 + 1;
  const bool has1 = q1 < n_queries;
  const float t0 = thr ? thr[q0] : 0.f, t1 = (thr && has1) ? thr[q1] : 0.f;
  const int64_t b_lo = row_lo + (int64_t)blockIdx.y * ADC_ROWS_PER_WG;
  const int64_t b_hi = b_lo + ADC_ROWS_PER_WG < row_hi ? b_lo + ADC_ROWS_PER_WG : row_hi;
  for (int64_t row = b_lo + threadIdx.x; row < b_hi; row += ADC_THREADS) {
    const uint8_t* c = codes + row * M;
    const bool vec16 = (M & 15) == 0;
    float a0 = 0.f, a1 = 0.f;
    int m = 0;
    for (; vec16 && m + 16 <= M; m += 16) {                    // 16 codes per 16-byte load (rows are 16-byte aligned only when M % 16 == 0)
      const uint4 w = *(const uint4*)(c + m);
      const uint32_t ws[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const uint32_t code = (ws[i >> 2] >> (8 * (i & 3))) & 0xffu;
        const float2 v = t[(m + i) * ksub + code];
        a0 += v.x; a1 += v.y;
      }
    }
    for (; m < M; ++m) { const float2 v = t[m * ksub + c[m]]; a0 += v.x; a1 += v.y; }
    if (!thr) {
      dump[(int64_t)q0 * dump_ld + (row - row_lo)] = a0;
      if (has1) dump[(int64_t)q1 * dump_ld + (row - row_lo)] = a1;
      continue;
    }
    if (a0 >= t0) { const uint32_t s = atomicAdd(cnt + q0, 1u); if (s < cap) cand[(int64_t)q0 * cap + s] = make_uint2((uint32_t)row, __float_as_uint(a0)); }
    if (has1 && a1 >= t1) { const uint32_t s = atomicAdd(cnt + q1, 1u); if (s < cap) cand[(int64_t)q1 * cap + s] = make_uint2((uint32_t)row, __float_as_uint(a1)); }
  }
}
__global__ void adc_keys_kernel(const uint2* __restrict__ cand, const uint32_t* __restrict__ cnt, uint32_t cap, int n_queries, uint64_t* __restrict__ keys) {
  const int q = blockIdx.y;
  const uint32_t n = cnt[q] < cap ? cnt[q] : cap;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const uint2 e = cand[(int64_t)q * cap + i];
    keys[(int64_t)q * cap + i] = make_key(__uint_as_float(e.y), e.x);
  }
}
__global__ void adc_init_kernel(float* thr, float* tau, float* margin, uint32_t* cnt, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) { thr[i] = -INFINITY; tau[i] = -INFINITY; margin[i] = 0.f; cnt[i] = 0u; }
}

template <typename T>
int grow(T*& p, size_t bytes, int64_t& total) {
  if (p) (void)hipFree(p);
  p = nullptr;
  if (hipMalloc((void**)&p, bytes) != hipSuccess) return dhr_set_error_message(DHR_ERR_HIP, ("hipMalloc of " + std::to_string(bytes) + " bytes failed (ADC workspace)").c_str());
  total += (int64_t)bytes;
  return DHR_OK;
}

int ensure_ws(dhr_pq* pq, int nq, int k) {
  int kp = 64;
  while (kp < k) kp <<= 1;
  const int q_cap = (std::min(nq, ADC_Q_CHUNK) + 1) & ~1;
  if (pq->q_cap >= q_cap && pq->kp == kp) return DHR_OK;
  const int qc = std::max(q_cap, pq->q_cap);
  int64_t tot = 0;
  int rc;
  const size_t per_q = (size_t)pq->M * pq->ksub;
  if ((rc = grow(pq->lut, (size_t)qc * per_q * 4, tot))) return rc;
  if ((rc = grow(pq->q32, (size_t)qc * pq->d * 4, tot))) return rc;
  if ((rc = grow(pq->cand, (size_t)qc * ADC_CAP * 8, tot))) return rc;
  if ((rc = grow(pq->cnt, (size_t)qc * 4, tot))) return rc;
  if ((rc = grow(pq->keys, (size_t)qc * ADC_CAP * 8, tot))) return rc;
  if ((rc = grow(pq->topk, (size_t)qc * kp * 8, tot))) return rc;
  if ((rc = grow(pq->thr, (size_t)qc * 4, tot))) return rc;
  if ((rc = grow(pq->tau, (size_t)qc * 4, tot))) return rc;
  if ((rc = grow(pq->margin, (size_t)qc * 4, tot))) return rc;
  if ((rc = grow(pq->d_max, 16, tot))) return rc;
  pq->q_cap = qc; pq->kp = kp;
  pq->bytes = (int64_t)pq->n * pq->M + (int64_t)pq->M * pq->ksub * pq->dsub * 4 + tot;
  return DHR_OK;
}

int build_lut(dhr_pq* pq, const dhr_query_batch* qb, int q_lo, int nq, hipStream_t s) {
  const void* src = qb->value;
  const int vsz = qb->value_dtype == DHR_VAL_F32 ? 4 : 2;
  const char* base = (const char*)src + (size_t)q_lo * qb->ld_value * vsz;
  void* staged = nullptr;
  if (qb->mem_kind == DHR_MEM_HOST) {                         // stage the chunk's rows
    const size_t bytes = (size_t)nq * qb->ld_value * vsz;
    PQ_HIP(hipMalloc(&staged, bytes));
    PQ_HIP(hipMemcpyAsync(staged, base, bytes, hipMemcpyHostToDevice, s));
    base = (const char*)staged;
  }
  const int64_t n1 = (int64_t)nq * pq->d;
  hipLaunchKernelGGL(adc_q32_kernel, dim3((unsigned)((n1 + 255) / 256)), dim3(256), 0, s, (const void*)base, qb->value_dtype == DHR_VAL_F32 ? 1 : 0,
                     qb->ld_value, nq, pq->d, pq->q32);
  const int64_t n2 = (int64_t)nq * pq->M * pq->ksub;
  if (nq & 1) PQ_HIP(hipMemsetAsync(pq->lut + (size_t)(nq >> 1) * pq->M * pq->ksub * 2, 0, (size_t)pq->M * pq->ksub * 8, s));   // the odd pair's second table
  hipLaunchKernelGGL(adc_lut_kernel, dim3((unsigned)((n2 + 255) / 256)), dim3(256), 0, s, pq->q32, nq, pq->d, pq->M, pq->ksub, pq->dsub, pq->cb, pq->lut);
  if (staged) { PQ_HIP(hipStreamSynchronize(s)); PQ_HIP(hipFree(staged)); }
  return DHR_OK;
}

int scan(dhr_pq* pq, int nq, int64_t lo, int64_t hi, bool filter, float* dump, int64_t dump_ld, hipStream_t s) {
  const int lds = pq->M * pq->ksub * 8;
  {      // per device and under a lock: handles on different devices may be used from different host threads (dhr_hip.h)
    static std::mutex attr_mu;
    static int attr_dev[64] = {};
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> lock(attr_mu);
    if (lds > attr_dev[dev & 63]) { PQ_HIP(hipFuncSetAttribute((const void*)adc_scan_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds)); attr_dev[dev & 63] = lds; }
  }
  const unsigned blocks = (unsigned)((hi - lo + ADC_ROWS_PER_WG - 1) / ADC_ROWS_PER_WG);
  hipLaunchKernelGGL(adc_scan_kernel, dim3((unsigned)((nq + 1) / 2), blocks), dim3(ADC_THREADS), lds, s, pq->codes, pq->M, pq->ksub, lo, hi, pq->lut, nq,
                     filter ? pq->thr : nullptr, pq->cand, pq->cnt, (uint32_t)ADC_CAP, dump, dump_ld);
  PQ_HIP(hipGetLastError());
  return DHR_OK;
}

}  // namespace

extern "C" int dhr_pq_create(int32_t device, int32_t mem_kind, int64_t n, int32_t d, int32_t M, int32_t nbits, const float* codebooks,
                             const uint8_t* codes, int64_t row_offset, dhr_pq** out) try {
  if (!out || !codebooks || !codes || n <= 0 || d <= 0 || M <= 0 || d % M || nbits < 1 || nbits > 8)
    return dhr_set_error_message(DHR_ERR_INVALID, "bad argument (1 <= nbits <= 8, d a multiple of M)");
  if (!DHR_MEM_KIND_OK(mem_kind)) return dhr_set_error_message(DHR_ERR_INVALID, "bad mem_kind");
  if ((M << nbits) * 8 > 160 * 1024 - 1024) return dhr_set_error_message(DHR_ERR_UNSUPPORTED, "the lookup tables of a query pair (M * 2^nbits * 8 B) do not fit the LDS");
  PQ_HIP(hipSetDevice(device));
  dhr_pq* pq = new dhr_pq();
  pq->device = device; pq->n = n; pq->row_offset = row_offset; pq->d = d; pq->M = M; pq->nbits = nbits; pq->ksub = 1 << nbits; pq->dsub = d / M;
  const size_t cb_bytes = (size_t)M * pq->ksub * pq->dsub * 4, code_bytes = (size_t)n * M;
  if (hipMalloc((void**)&pq->cb, cb_bytes) != hipSuccess || hipMalloc((void**)&pq->codes, code_bytes) != hipSuccess) {
    (void)hipFree(pq->cb); delete pq;
    return dhr_set_error_message(DHR_ERR_HIP, "hipMalloc of the PQ index failed");
  }
  const hipMemcpyKind kind = mem_kind == DHR_MEM_HOST ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice;
  PQ_HIP(hipMemcpy(pq->cb, codebooks, cb_bytes, kind));
  PQ_HIP(hipMemcpy(pq->codes, codes, code_bytes, kind));
  pq->bytes = (int64_t)(cb_bytes + code_bytes);
  *out = pq;
  return DHR_OK;
} DHR_CATCH_STATUS
extern "C" void dhr_pq_destroy(dhr_pq* pq) try {
  if (!pq) return;
  (void)hipSetDevice(pq->device);
  void* ps[] = {pq->cb, pq->codes, pq->lut, pq->q32, pq->cand, pq->cnt, pq->keys, pq->topk, pq->thr, pq->tau, pq->margin, pq->d_max};
  for (void* p : ps) (void)hipFree(p);
  delete pq;
} DHR_CATCH_VOID
extern "C" int64_t dhr_pq_device_bytes(const dhr_pq* pq) try { return pq ? pq->bytes : 0; } DHR_CATCH_VALUE(0)
extern "C" int dhr_pq_last_scan(const dhr_pq* pq, double* ms, double* code_bytes) try {
  if (!pq) return dhr_set_error_message(DHR_ERR_INVALID, "null handle");
  if (ms) *ms = pq->last_scan_ms;
  if (code_bytes) *code_bytes = pq->last_code_bytes;
  return DHR_OK;
} DHR_CATCH_STATUS

// Raw ADC scores of rows [row_lo, row_hi) for every query: out [n_queries][row_hi - row_lo] fp32, device memory (tests).
extern "C" int dhr_pq_adc_scores(dhr_pq* pq, const dhr_query_batch* qb, int64_t row_lo, int64_t row_hi, float* out_dev, void* stream) try {
  if (!pq || !qb || !qb->value || !out_dev || row_lo < 0 || row_hi > pq->n || row_lo >= row_hi) return dhr_set_error_message(DHR_ERR_INVALID, "bad argument");
  PQ_HIP(hipSetDevice(pq->device));
  hipStream_t s = (hipStream_t)stream;
  int rc;
  for (int q_lo = 0; q_lo < qb->n_queries; q_lo += ADC_Q_CHUNK) {
    const int nq = std::min(ADC_Q_CHUNK, qb->n_queries - q_lo);
    if ((rc = ensure_ws(pq, nq, 64))) return rc;
    if ((rc = build_lut(pq, qb, q_lo, nq, s))) return rc;
    if ((rc = scan(pq, nq, row_lo, row_hi, false, out_dev + (int64_t)q_lo * (row_hi - row_lo), row_hi - row_lo, s))) return rc;
  }
  PQ_HIP(hipStreamSynchronize(s));
  return DHR_OK;
} DHR_CATCH_STATUS

// IndexPQ.search(queries, k): per query the k rows with the largest ADC score, best first (score desc, row asc on exact ties);
// out_scores [n_queries][k] fp32, out_rows [n_queries][k] int64 global rows; (-inf, -1) beyond the corpus size.
extern "C" int dhr_pq_search(dhr_pq* pq, const dhr_query_batch* qb, int32_t k, float* out_scores, int64_t* out_rows, int32_t out_mem_kind,
                             void* stream) try {
  if (!pq || !qb || !qb->value || !out_scores || !out_rows || qb->n_queries <= 0) return dhr_set_error_message(DHR_ERR_INVALID, "bad argument");
  if (k <= 0 || k > (1 << 20)) return dhr_set_error_message(DHR_ERR_INVALID, "k must be in [1, 1048576]");      // k > 16384: the global-memory merge (select_global.hip)
  if (!DHR_MEM_KIND_OK(out_mem_kind) || !DHR_MEM_KIND_OK(qb->mem_kind)) return dhr_set_error_message(DHR_ERR_INVALID, "bad mem_kind");
  PQ_HIP(hipSetDevice(pq->device));
  hipStream_t s = (hipStream_t)stream;
  hipEvent_t e0, e1;
  PQ_HIP(hipEventCreate(&e0)); PQ_HIP(hipEventCreate(&e1));
  double scan_ms = 0, code_bytes = 0;
  int rc = DHR_OK;
  float* d_scores = nullptr;
  int64_t* d_rows = nullptr;
  void* stage = nullptr;
  auto done = [&](int code) { (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); (void)hipFree(stage); return code; };
  for (int q_lo = 0; q_lo < qb->n_queries; q_lo += ADC_Q_CHUNK) {
    const int nq = std::min(ADC_Q_CHUNK, qb->n_queries - q_lo);
    if ((rc = ensure_ws(pq, nq, k))) return done(rc);
    const int kp = pq->kp;
    if ((rc = build_lut(pq, qb, q_lo, nq, s))) return done(rc);
    hipLaunchKernelGGL(adc_init_kernel, dim3((nq + 255) / 256), dim3(256), 0, s, pq->thr, pq->tau, pq->margin, pq->cnt, nq);
    if (hipMemsetAsync(pq->topk, 0, (size_t)nq * kp * 8, s) != hipSuccess) return done(dhr_set_error_message(DHR_ERR_HIP, "memset failed"));
    SelectArgs sel{};
    sel.topk_keys = pq->topk; sel.in_keys = pq->keys; sel.ld_keys = ADC_CAP; sel.cnt = pq->cnt; sel.count_all = 0; sel.cap = ADC_CAP;
    sel.k = k; sel.kp = kp; sel.sort_n = 4 * kp; sel.kps = kp; sel.margin = pq->margin; sel.tau = pq->tau; sel.thr = pq->thr; sel.n_queries = nq;
    // blocks of rows: the first one is scored exhaustively (thresholds -inf: every row is a candidate, so it must fit a list);
    // afterwards each block is as large as everything before it
    int64_t lo = 0;
    int64_t step = std::min<int64_t>(pq->n, ADC_CAP / 2);
    while (lo < pq->n) {
      int64_t hi = std::min(pq->n, lo + step);
      for (;;) {
        if (hipMemsetAsync(pq->cnt, 0, (size_t)nq * 4, s) != hipSuccess) return done(dhr_set_error_message(DHR_ERR_HIP, "memset failed"));
        (void)hipEventRecord(e0, s);
        if ((rc = scan(pq, nq, lo, hi, true, nullptr, 0, s))) return done(rc);
        (void)hipEventRecord(e1, s);
        if (hipMemsetAsync(pq->d_max, 0, 16, s) != hipSuccess || launch_max_u32(pq->cnt, nq, pq->d_max, (unsigned long long*)(pq->d_max + 2), s) != hipSuccess)
          return done(dhr_set_error_message(DHR_ERR_HIP, "max launch failed"));
        uint32_t mx = 0;
        if (hipMemcpyAsync(&mx, pq->d_max, 4, hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess)
          return done(dhr_set_error_message(DHR_ERR_HIP, "count read-back failed"));
        float ms = 0;
        (void)hipEventElapsedTime(&ms, e0, e1);
        scan_ms += ms; code_bytes += (double)(hi - lo) * pq->M * ((nq + 1) / 2);
        if (mx <= (uint32_t)ADC_CAP) break;
        hi = lo + std::max<int64_t>((hi - lo) / 2, 1);            // a list overflowed: the block again, half as long (<= ADC_CAP rows cannot overflow)
      }
      hipLaunchKernelGGL(adc_keys_kernel, dim3(8, (unsigned)nq), dim3(256), 0, s, pq->cand, pq->cnt, (uint32_t)ADC_CAP, nq, pq->keys);
      if (launch_select(sel, s) != hipSuccess) return done(dhr_set_error_message(DHR_ERR_HIP, "select launch failed"));
      step = std::max<int64_t>(hi, step);
      lo = hi;
    }
    // deliver this chunk's lists
    float* os = out_scores + (int64_t)q_lo * k;
    int64_t* orow = out_rows + (int64_t)q_lo * k;
    if (out_mem_kind == DHR_MEM_HOST) {
      if (!stage && hipMalloc(&stage, (size_t)ADC_Q_CHUNK * k * 12) != hipSuccess) return done(dhr_set_error_message(DHR_ERR_HIP, "hipMalloc failed"));
      d_rows = (int64_t*)stage; d_scores = (float*)((char*)stage + (size_t)ADC_Q_CHUNK * k * 8);
    } else { d_scores = os; d_rows = orow; }
    if (launch_emit(pq->topk, kp, nq, k, pq->row_offset, d_scores, d_rows, s) != hipSuccess) return done(dhr_set_error_message(DHR_ERR_HIP, "emit launch failed"));
    if (out_mem_kind == DHR_MEM_HOST) {
      if (hipMemcpyAsync(os, d_scores, (size_t)nq * k * 4, hipMemcpyDeviceToHost, s) != hipSuccess ||
          hipMemcpyAsync(orow, d_rows, (size_t)nq * k * 8, hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess)
        return done(dhr_set_error_message(DHR_ERR_HIP, "result copy failed"));
    }
  }
  if (hipStreamSynchronize(s) != hipSuccess) return done(dhr_set_error_message(DHR_ERR_HIP, "stream synchronize failed"));
  pq->last_scan_ms = scan_ms; pq->last_code_bytes = code_bytes;
  return done(DHR_OK);
} DHR_CATCH_STATUS
