// Search entry points of the C ABI (include/dhr_hip.h): dhr_search, dhr_search_rerank, the staged calls of the sharded search, dhr_score_rows;
// densify, PQ training / encoding, debug hooks, shard reduce.  The handle and its build: index_build.hip; the controller (search_core): search_core.hip;
// the sharded control flow: sharded.hip; error record, exception classifier: abi.cpp.
#include "dhr_state.h"

extern "C" int dhr_search(dhr_index* ix, const dhr_query_batch* qb, int32_t k, float* out_scores, int64_t* out_rows,
                          int32_t out_mem_kind, void* stream) try {
  dhr::alloc_checkpoint();
  int rc = check_queries(ix, qb);
  if (rc) return rc;
  if (k <= 0) return set_error(DHR_ERR_INVALID, "k must be > 0");
  if (k > (1 << 20)) return set_error(DHR_ERR_UNSUPPORTED, "k > 1048576 is not supported");      // k > 16384: global-memory merge (select_global.hip)
  if (!out_scores || !out_rows) return set_error(DHR_ERR_INVALID, "null output pointer");
  if (!DHR_MEM_KIND_OK(out_mem_kind)) return set_error(DHR_ERR_INVALID, "bad out_mem_kind");
  HIP_TRY(hipSetDevice(ix->device));
  hipStream_t s = (hipStream_t)stream;
  const int Q = qb->n_queries;
  ix->pend.valid = false;          // a plain search overwrites the workspace of any staged search left open on this handle
  Events evs;
  hipEvent_t ev0, ev1;
  HIP_TRY(evs.add(&ev0)); HIP_TRY(evs.add(&ev1));
  HIP_TRY(hipEventRecord(ev0, s));
  Timer tm{ix->profile != 0, s, {}, {}};
  dhr_search_stats st{};
  st.n_rows = ix->n_rows; st.n_queries = Q; st.k = k;
  Workspace& w = ix->ws;
  Drain drain{ix, s};
  if ((rc = search_core(ix, w, qb, k, 0, tm, st, s)) != DHR_OK) return rc;

  // ---- results
  float* d_scores = out_scores;
  int64_t* d_rows = out_rows;
  if (out_mem_kind == DHR_MEM_HOST) {
    const size_t need = (size_t)Q * k * 12;
    if ((rc = grow(w.out_stage, w.out_stage_bytes, need, w.bytes)) != DHR_OK) return rc;
    d_rows = (int64_t*)w.out_stage;
    d_scores = (float*)((char*)w.out_stage + (size_t)Q * k * 8);
  }
  HIP_TRY(launch_emit(w.topk_keys, w.kp, Q, k, ix->row_offset, d_scores, d_rows, s));
  if (out_mem_kind == DHR_MEM_HOST) {
    HIP_TRY(hipMemcpyAsync(out_rows, d_rows, (size_t)Q * k * 8, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync(out_scores, d_scores, (size_t)Q * k * 4, hipMemcpyDeviceToHost, s));
  }
  HIP_TRY(hipEventRecord(ev1, s));
  HIP_TRY(hipStreamSynchronize(s));
  float total = 0.f;
  hipEventElapsedTime(&total, ev0, ev1);
  st.total_ms = total;
  double ms[5] = {0, 0, 0, 0, 0};
  tm.collect(ms);
  st.gemm_ms = ms[T_GEMM]; st.refine_ms = ms[T_REFINE]; st.rescore_ms = ms[T_RESCORE]; st.select_ms = ms[T_SELECT];
  st.prep_ms = ms[T_PREP];
  ix->stats = st;
  drain.armed = false;
  return DHR_OK;
} DHR_CATCH_STATUS

// Two-stage approximate GIP on the device (gip_retrieval.py:128-156): stage 1 is an ordinary search of the
// restricted batch for k1 rows, stage 2 the exact gated inner product of the full batch on exactly those rows
// and the top-k of that; the k1 rows never leave the device.
extern "C" int dhr_search_rerank(dhr_index* ix, const dhr_query_batch* qb1, const dhr_query_batch* qb2, int32_t k1, int32_t k,
                                 float* out_scores, int64_t* out_rows, int32_t out_mem_kind, void* stream) try {
  dhr::alloc_checkpoint();
  int rc = check_queries(ix, qb1);
  if (rc) return rc;
  if ((rc = check_queries(ix, qb2)) != DHR_OK) return rc;
  if (qb1->n_queries != qb2->n_queries) return set_error(DHR_ERR_INVALID, "the two query batches differ in n_queries");
  if (k <= 0 || k1 < k) return set_error(DHR_ERR_INVALID, "need 0 < k <= k1");
  if (k1 > (1 << 20)) return set_error(DHR_ERR_UNSUPPORTED, "k1 > 1048576 is not supported");
  if (!out_scores || !out_rows) return set_error(DHR_ERR_INVALID, "null output pointer");
  if (!DHR_MEM_KIND_OK(out_mem_kind)) return set_error(DHR_ERR_INVALID, "bad out_mem_kind");
  HIP_TRY(hipSetDevice(ix->device));
  hipStream_t s = (hipStream_t)stream;
  const int Q = qb1->n_queries;
  Events evs;
  hipEvent_t ev0, ev1;
  HIP_TRY(evs.add(&ev0)); HIP_TRY(evs.add(&ev1));
  HIP_TRY(hipEventRecord(ev0, s));
  Timer tm{ix->profile != 0, s, {}, {}};
  dhr_search_stats st{};
  st.n_rows = ix->n_rows; st.n_queries = Q; st.k = k;
  Workspace& w = ix->ws;
  Drain drain{ix, s};
  // ---- stage 1
  if ((rc = search_core(ix, w, qb1, k1, 0, tm, st, s)) != DHR_OK) return rc;
  // ---- stage 2: exact scores of the stage-1 rows under the full batch, top-k of those
  DevMem rows_mem;
  uint32_t*& d_rows32 = (uint32_t*&)rows_mem.p;
  HIP_TRY(hipMalloc((void**)&d_rows32, (size_t)Q * k1 * 4));
  auto done = [&](int code) { return code; };       // (rows_mem releases the scratch)
  if (launch_keys_to_rows(w.topk_keys, w.kp, Q, k1, d_rows32, s) != hipSuccess) return done(set_error(DHR_ERR_HIP, "keys_to_rows launch failed"));
  if (w.keys_ld < k1) return done(set_error(DHR_ERR_INTERNAL, "key buffer smaller than k1"));
  const bool gate2 = ix->d_dlr > 0 && qb2->index != nullptr && qb2->index_dtype != DHR_IDX_NONE;
  tm.begin(T_PREP);
  if ((rc = prep_queries(ix, w, qb2, s)) != DHR_OK) return done(rc);
  if (hipMemsetAsync(w.topk_keys, 0, (size_t)w.q_pad * w.kp * 8, s) != hipSuccess) return done(set_error(DHR_ERR_HIP, "memset failed"));
  tm.end();
  RescoreArgs r = base_rescore_args(ix, w, Q, gate2);
  r.rows32 = d_rows32; r.ld_rows = k1; r.count_all = (uint32_t)k1; r.max_count = (uint32_t)k1;
  r.out_keys = w.rs_keys; r.ld_keys = w.keys_ld;
  tm.begin(T_RESCORE);
  if (launch_rescore(r, s) != hipSuccess) return done(set_error(DHR_ERR_HIP, "rescore launch failed"));
  tm.end();
  st.candidates_exact += (int64_t)Q * k1;
  SelectArgs sel{};
  sel.topk_keys = w.topk_keys; sel.in_keys = w.rs_keys; sel.ld_keys = w.keys_ld; sel.cap = (uint32_t)w.cap;
  sel.cnt = nullptr; sel.count_all = (uint32_t)k1;
  sel.k = k; sel.kp = w.kp; sel.sort_n = select_sort_n(w.kp);
  sel.kps = 64; while (sel.kps < k) sel.kps <<= 1;
  sel.margin = w.margin; sel.tau = w.tau; sel.thr = w.thr; sel.n_queries = Q;
  tm.begin(T_SELECT);
  if (launch_select(sel, s) != hipSuccess) return done(set_error(DHR_ERR_HIP, "select launch failed"));
  tm.end();
  // ---- results
  float* d_scores = out_scores;
  int64_t* d_rows = out_rows;
  if (out_mem_kind == DHR_MEM_HOST) {
    const size_t need = (size_t)Q * k * 12;
    if ((rc = grow(w.out_stage, w.out_stage_bytes, need, w.bytes)) != DHR_OK) return done(rc);
    d_rows = (int64_t*)w.out_stage;
    d_scores = (float*)((char*)w.out_stage + (size_t)Q * k * 8);
  }
  if (launch_emit(w.topk_keys, w.kp, Q, k, ix->row_offset, d_scores, d_rows, s) != hipSuccess) return done(set_error(DHR_ERR_HIP, "emit launch failed"));
  if (out_mem_kind == DHR_MEM_HOST) {
    if (hipMemcpyAsync(out_rows, d_rows, (size_t)Q * k * 8, hipMemcpyDeviceToHost, s) != hipSuccess ||
        hipMemcpyAsync(out_scores, d_scores, (size_t)Q * k * 4, hipMemcpyDeviceToHost, s) != hipSuccess)
      return done(set_error(DHR_ERR_HIP, "D2H failed"));
  }
  if (hipEventRecord(ev1, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) return done(set_error(DHR_ERR_HIP, "two-stage search failed on the device"));
  float total = 0.f;
  hipEventElapsedTime(&total, ev0, ev1);
  st.total_ms = total;
  double ms[5] = {0, 0, 0, 0, 0};
  tm.collect(ms);
  st.gemm_ms = ms[T_GEMM]; st.refine_ms = ms[T_REFINE]; st.rescore_ms = ms[T_RESCORE]; st.select_ms = ms[T_SELECT];
  st.prep_ms = ms[T_PREP];
  ix->stats = st;
  drain.armed = false;
  return done(DHR_OK);
} DHR_CATCH_STATUS

// ---- staged search for the row-sharded path (dhr_amd/dist.py): the shards agree on ONE threshold per
// query after their sampled runs, so each shard collects only its share of the global top-k.
extern "C" int32_t dhr_search_sample_rank(const dhr_index* ix, int32_t k) try {
  if (!ix || k <= 0) return 0;
  int S = 0, r = k;
  plan_sampling(ix, k, S, r);
  return S >= 2 ? local_sample_rank(ix, r) : 0;
} DHR_CATCH_VALUE(0)
extern "C" int32_t dhr_search_union_rank(const dhr_index* ix, int32_t k) try {
  if (!ix || k <= 0) return 0;
  int S = 0, r = k;
  plan_sampling(ix, k, S, r);
  return S >= 2 ? r : 0;
} DHR_CATCH_VALUE(0)

// Ranks of the second agreement (dhr_search_mid): after the head, the sample and the first slice of the main pass a shard has seen the
// fraction f of its rows, scattered; the union of what the shards have seen holds k f +- sqrt(k f (1 - f)) of the final top-k, so its
// (k f + 6 sigma + 4)-th best score lies below the final k-th best (the counts verify it; a failure is repaired like any other).  A shard
// reports its share of that rank (local_sample_rank's rule).
extern "C" int32_t dhr_search_mid_ranks(const dhr_index* ix, int32_t k, int32_t* out_local, int32_t* out_union) try {
  if (out_local) *out_local = 0;
  if (out_union) *out_union = 0;
  if (!ix || k <= 0) return 0;
  int S = 0, r = k;
  plan_sampling(ix, k, S, r);
  if (S < 2) return 0;
  const int r_eff = local_sample_rank(ix, r);
  int64_t first = head_rows(ix, S, r_eff);
  first = std::min(round_up(first, TILE_ROWS), round_up(ix->n_rows, TILE_ROWS));
  const int64_t head = first / TILE_ROWS, rest = ix->n_tiles - head;
  if (rest <= 0) return 0;
  const int64_t n_sample = (rest + S - 1) / S, n_main = rest - n_sample;
  if (n_main < 64) return 0;
  const int64_t off = std::min<int64_t>(n_main, round_up(n_main * mid_share16() / 16, DOC_GROUP));
  const double f = (double)(head + n_sample + off) / (double)ix->n_tiles;
  const int ru = (int)std::min<double>(k, std::ceil((double)k * f + 6.0 * std::sqrt((double)k * f * (1.0 - f)) + 4.0));
  const double m = (double)ru / std::max(1, ix->sample_share);
  const int rl = ix->sample_share <= 1 ? ru : std::min(ru, (int)std::ceil(m + 5.0 * std::sqrt(m) + 4.0));
  if (out_local) *out_local = rl;
  if (out_union) *out_union = ru;
  return rl;
} DHR_CATCH_VALUE(0)
// The first slice of the main pass with the thresholds of the first agreement; leaves the shard's r_local best scores seen so far in
// out_scores_dev [Q, r_local] (r_local: dhr_search_mid_ranks, or what the shards agreed on).  dhr_search_finish then takes the thresholds of the
// second agreement.
static int search_mid_impl(dhr_index* ix, const float* tau_hat_dev, int32_t r_local, float* out_scores_dev, void* stream) {
  if (!ix || !ix->pend.valid) return set_error(DHR_ERR_INVALID, "dhr_search_mid without a matching dhr_search_begin");
  if (ix->pend.done) return DHR_OK;                   // the shard was not sampled: its search is complete already
  if (ix->pend.mid) return set_error(DHR_ERR_INVALID, "dhr_search_mid called twice");
  if (!tau_hat_dev || !out_scores_dev) return set_error(DHR_ERR_INVALID, "null pointer");
  HIP_TRY(hipSetDevice(ix->device));
  hipStream_t s = (hipStream_t)stream;
  if (dhr_search_mid_ranks(ix, ix->pend.k, nullptr, nullptr) <= 0) return set_error(DHR_ERR_INVALID, "this index has no mid step (dhr_search_mid_ranks returned 0)");
  if (r_local <= 0 || r_local > ix->pend.k) return set_error(DHR_ERR_INVALID, "r_local must be in [1, k]");
  const int32_t rl = r_local;
  Timer tm{ix->profile != 0, s, {}, {}};
  dhr_search_stats st = ix->stats;
  int rc;
  Drain drain{ix, s};
  if ((rc = search_core(ix, ix->ws, nullptr, ix->pend.k, 0, tm, st, s, 3, tau_hat_dev)) != DHR_OK) return rc;
  HIP_TRY(launch_emit_scores(ix->ws.topk_keys, ix->ws.kp, ix->pend.Q, rl, out_scores_dev, s));
  if (ix->profile) {
    HIP_TRY(hipStreamSynchronize(s));
    double ms[5] = {0, 0, 0, 0, 0};
    tm.collect(ms);
    st.gemm_ms += ms[T_GEMM]; st.refine_ms += ms[T_REFINE]; st.rescore_ms += ms[T_RESCORE]; st.select_ms += ms[T_SELECT];
  }
  ix->stats = st;
  drain.armed = false;
  return DHR_OK;
}
extern "C" int dhr_search_mid(dhr_index* ix, const float* tau_hat_dev, int32_t r_local, float* out_scores_dev, void* stream) try {
  int rc = search_mid_impl(ix, tau_hat_dev, r_local, out_scores_dev, stream);
  if (rc == DHR_OK && ix) HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
  return rc;
} DHR_CATCH_STATUS
extern "C" int dhr_internal_search_mid_async(dhr_index* ix, const float* tau_hat_dev, int32_t r_local, float* out_scores_dev, void* stream) try {
  return search_mid_impl(ix, tau_hat_dev, r_local, out_scores_dev, stream);
} DHR_CATCH_STATUS

// ---- first agreement in TWO rounds (round 5).  The shards of a sharded search each ran their whole sampled run from nothing, chasing their
// share of the union's rank on their own: eight runs together rescored 3.2 k rows per query where the unsharded search's one run rescores
// 0.7 k.  dhr_search_pre streams the first part of the shard's sample (pre_share16 / 16 of it) and reports its best scores seen so far; the
// union of the parts is a fraction phi of the union sample, so its (r phi + 6 sigma + 4)-th best score lies below the union sample's final
// r-th best (adaptive_rank's argument), and dhr_search_begin_rest streams the rest of the sample filtering at that COMMON threshold.  Whatever
// the threshold is worth, the lists a shard reports afterwards are complete above it, so the union threshold computed from them can only
// come out lower than the true one -- still valid; the count check at the end of the step verifies everything as before.
extern "C" int32_t dhr_search_pre_ranks(const dhr_index* ix, int32_t k, int32_t* out_local, int32_t* out_union) try {
  if (out_local) *out_local = 0;
  if (out_union) *out_union = 0;
  if (!ix || k <= 0) return 0;
  int S = 0, r = k;
  plan_sampling(ix, k, S, r);
  if (S < 2) return 0;
  const int r_eff = local_sample_rank(ix, r);
  int64_t first = head_rows(ix, S, r_eff);
  first = std::min(round_up(first, TILE_ROWS), round_up(ix->n_rows, TILE_ROWS));
  const int64_t first_valid = std::min(first, ix->n_rows);
  const int64_t rest = ix->n_tiles - first / TILE_ROWS;
  if (rest <= 0) return 0;
  const int64_t n_sample = (rest + S - 1) / S;
  const int64_t n_a = pre_positions(n_sample);
  if (n_a <= 0) return 0;
  const double phi = (double)(first_valid + n_a * TILE_ROWS) / (double)(first_valid + n_sample * TILE_ROWS);
  const int ru = adaptive_rank(r, phi);
  const double m = (double)ru / std::max(1, ix->sample_share);
  const int rl = ix->sample_share <= 1 ? ru : std::min(ru, (int)std::ceil(m + 5.0 * std::sqrt(m) + 4.0));
  if (out_local) *out_local = rl;
  if (out_union) *out_union = ru;
  return rl;
} DHR_CATCH_VALUE(0)
static int search_pre_impl(dhr_index* ix, const dhr_query_batch* qb, int32_t k, int32_t r_local, float* out_scores_dev, void* stream, bool sync) {
  dhr::alloc_checkpoint();
  int rc = check_queries(ix, qb);
  if (rc) return rc;
  if (k <= 0 || k > (1 << 20)) return set_error(DHR_ERR_INVALID, "k must be in [1, 1048576]");
  if (!out_scores_dev) return set_error(DHR_ERR_INVALID, "null pointer");
  if (dhr_search_pre_ranks(ix, k, nullptr, nullptr) <= 0) return set_error(DHR_ERR_INVALID, "this index has no pre step (dhr_search_pre_ranks returned 0)");
  if (r_local <= 0 || r_local > k) return set_error(DHR_ERR_INVALID, "r_local must be in [1, k]");
  HIP_TRY(hipSetDevice(ix->device));
  hipStream_t s = (hipStream_t)stream;
  ix->pend.valid = false;
  Timer tm{ix->profile != 0, s, {}, {}};
  dhr_search_stats st{};
  st.n_rows = ix->n_rows; st.n_queries = qb->n_queries; st.k = k;
  Drain drain{ix, s};
  if ((rc = search_core(ix, ix->ws, qb, k, 0, tm, st, s, 4)) != DHR_OK) return rc;
  HIP_TRY(launch_emit_scores(ix->ws.topk_keys, ix->ws.kp, qb->n_queries, r_local, out_scores_dev, s));
  ix->pend.dev_bound = ix->pend.dev_exact = 0;
  if (sync || ix->profile) {
    HIP_TRY(hipStreamSynchronize(s));
    double ms[5] = {0, 0, 0, 0, 0};
    tm.collect(ms);
    st.gemm_ms = ms[T_GEMM]; st.refine_ms = ms[T_REFINE]; st.rescore_ms = ms[T_RESCORE]; st.select_ms = ms[T_SELECT]; st.prep_ms = ms[T_PREP];
  }
  ix->stats = st;
  drain.armed = false;
  return DHR_OK;
}
extern "C" int dhr_search_pre(dhr_index* ix, const dhr_query_batch* qb, int32_t k, int32_t r_local, float* out_scores_dev, void* stream) try {
  return search_pre_impl(ix, qb, k, r_local, out_scores_dev, stream, true);
} DHR_CATCH_STATUS
extern "C" int dhr_internal_search_pre_async(dhr_index* ix, const dhr_query_batch* qb, int32_t k, int32_t r_local, float* out_scores_dev, void* stream) try {
  return search_pre_impl(ix, qb, k, r_local, out_scores_dev, stream, false);
} DHR_CATCH_STATUS
// the rest of the sampled run behind dhr_search_pre; leaves the handle where dhr_search_begin leaves it (out_sample_scores_dev: [Q, dhr_search_sample_rank])
static int search_begin_rest_impl(dhr_index* ix, const float* tau_dev, float* out_sample_scores_dev, void* stream, bool sync) {
  if (!ix || !ix->pend.valid || !ix->pend.pre) return set_error(DHR_ERR_INVALID, "dhr_search_begin_rest without a matching dhr_search_pre");
  if (!tau_dev || !out_sample_scores_dev) return set_error(DHR_ERR_INVALID, "null pointer");
  HIP_TRY(hipSetDevice(ix->device));
  hipStream_t s = (hipStream_t)stream;
  const int k = ix->pend.k, Q = ix->pend.Q;
  Timer tm{ix->profile != 0, s, {}, {}};
  dhr_search_stats st = ix->stats;                     // continue the counters of the pre call
  int rc;
  Drain drain{ix, s};
  if ((rc = search_core(ix, ix->ws, nullptr, k, 0, tm, st, s, 5, tau_dev)) != DHR_OK) return rc;
  const int r = dhr_search_sample_rank(ix, k);
  if (r > 0 && !ix->pend.done) HIP_TRY(launch_emit_scores(ix->ws.topk_keys, ix->ws.kp, Q, r, out_sample_scores_dev, s));
  ix->pend.dev_bound = ix->pend.dev_exact = 0;
  if (sync || ix->profile) {
    HIP_TRY(hipMemcpyAsync(ix->ws.h_stats, ix->ws.d_stats, 32, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    ix->pend.dev_bound = (int64_t)((unsigned long long*)ix->ws.h_stats)[0];
    ix->pend.dev_exact = (int64_t)((unsigned long long*)ix->ws.h_stats)[1];
    st.candidates_bound += ix->pend.dev_bound;
    st.candidates_exact += ix->pend.dev_exact;
    double ms[5] = {0, 0, 0, 0, 0};
    tm.collect(ms);
    st.gemm_ms += ms[T_GEMM]; st.refine_ms += ms[T_REFINE]; st.rescore_ms += ms[T_RESCORE]; st.select_ms += ms[T_SELECT]; st.prep_ms += ms[T_PREP];
  }
  ix->stats = st;
  drain.armed = false;
  return DHR_OK;
}
extern "C" int dhr_search_begin_rest(dhr_index* ix, const float* tau_dev, float* out_sample_scores_dev, void* stream) try {
  return search_begin_rest_impl(ix, tau_dev, out_sample_scores_dev, stream, true);
} DHR_CATCH_STATUS
extern "C" int dhr_internal_search_begin_rest_async(dhr_index* ix, const float* tau_dev, float* out_sample_scores_dev, void* stream) try {
  return search_begin_rest_impl(ix, tau_dev, out_sample_scores_dev, stream, false);
} DHR_CATCH_STATUS

// sync = false (dhr_search_sharded*): only enqueues when the controller runs without read-backs -- the shards of a one-process search then
// work concurrently until the collective layer's own synchronisation; statistics and timers are then not collected
static int search_begin_impl(dhr_index* ix, const dhr_query_batch* qb, int32_t k, float* out_sample_scores_dev, void* stream, bool sync) {
  dhr::alloc_checkpoint();
  int rc = check_queries(ix, qb);
  if (rc) return rc;
  if (k <= 0 || k > (1 << 20)) return set_error(DHR_ERR_INVALID, "k must be in [1, 1048576]");
  HIP_TRY(hipSetDevice(ix->device));
  hipStream_t s = (hipStream_t)stream;
  ix->pend.valid = false;
  Timer tm{ix->profile != 0, s, {}, {}};
  dhr_search_stats st{};
  st.n_rows = ix->n_rows; st.n_queries = qb->n_queries; st.k = k;
  Drain drain{ix, s};
  if ((rc = search_core(ix, ix->ws, qb, k, 0, tm, st, s, 1)) != DHR_OK) return rc;
  const int r = dhr_search_sample_rank(ix, k);
  if (r > 0 && !ix->pend.done) {
    if (!out_sample_scores_dev) return set_error(DHR_ERR_INVALID, "null sample score buffer");
    HIP_TRY(launch_emit_scores(ix->ws.topk_keys, ix->ws.kp, qb->n_queries, r, out_sample_scores_dev, s));
  }
  ix->pend.dev_bound = ix->pend.dev_exact = 0;
  if (sync || ix->profile) {
    HIP_TRY(hipMemcpyAsync(ix->ws.h_stats, ix->ws.d_stats, 32, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    // controller without read-backs: the counters live on the device (zero otherwise); remember what has been folded in
    ix->pend.dev_bound = (int64_t)((unsigned long long*)ix->ws.h_stats)[0];
    ix->pend.dev_exact = (int64_t)((unsigned long long*)ix->ws.h_stats)[1];
    st.candidates_bound += ix->pend.dev_bound;
    st.candidates_exact += ix->pend.dev_exact;
    double ms[5] = {0, 0, 0, 0, 0};
    tm.collect(ms);
    st.gemm_ms = ms[T_GEMM]; st.refine_ms = ms[T_REFINE]; st.rescore_ms = ms[T_RESCORE]; st.select_ms = ms[T_SELECT]; st.prep_ms = ms[T_PREP];
  }
  ix->stats = st;
  drain.armed = false;
  return DHR_OK;
}
extern "C" void dhr_internal_search_abort(dhr_index* ix) try {
  if (ix) ix->pend.valid = false;
} DHR_CATCH_VOID
extern "C" int dhr_search_begin(dhr_index* ix, const dhr_query_batch* qb, int32_t k, float* out_sample_scores_dev, void* stream) try {
  return search_begin_impl(ix, qb, k, out_sample_scores_dev, stream, true);
} DHR_CATCH_STATUS
extern "C" int dhr_internal_search_begin_async(dhr_index* ix, const dhr_query_batch* qb, int32_t k, float* out_sample_scores_dev, void* stream) try {
  return search_begin_impl(ix, qb, k, out_sample_scores_dev, stream, false);
} DHR_CATCH_STATUS

static int search_finish_impl(dhr_index* ix, const float* tau_hat_dev, float* out_scores, int64_t* out_rows,
                              int32_t* out_count_dev, int32_t out_mem_kind, void* stream, bool sync) {
  if (!ix || !ix->pend.valid) return set_error(DHR_ERR_INVALID, "dhr_search_finish without a matching dhr_search_begin");
  if (!out_scores || !out_rows || !out_count_dev) return set_error(DHR_ERR_INVALID, "null output pointer");
  if (!DHR_MEM_KIND_OK(out_mem_kind)) return set_error(DHR_ERR_INVALID, "bad out_mem_kind");
  if (!ix->pend.done && !tau_hat_dev) return set_error(DHR_ERR_INVALID, "thresholds are required (the shard ran a sampled pass)");
  HIP_TRY(hipSetDevice(ix->device));
  hipStream_t s = (hipStream_t)stream;
  const int Q = ix->pend.Q, k = ix->pend.k;
  Workspace& w = ix->ws;
  Timer tm{ix->profile != 0, s, {}, {}};
  dhr_search_stats st = ix->stats;                     // continue the counters of the begin call
  int rc;
  Drain drain{ix, s};
  if ((rc = search_core(ix, w, nullptr, k, 0, tm, st, s, 2, tau_hat_dev)) != DHR_OK) return rc;
  HIP_TRY(launch_count_ge(w.topk_keys, w.kp, k, ix->pend.done ? nullptr : w.tau_hat, ix->pend.done ? nullptr : w.fail_flags, Q,
                          out_count_dev, s));
  float* d_scores = out_scores;
  int64_t* d_rows = out_rows;
  if (out_mem_kind == DHR_MEM_HOST) {
    const size_t need = (size_t)Q * k * 12;
    if ((rc = grow(w.out_stage, w.out_stage_bytes, need, w.bytes)) != DHR_OK) return rc;
    d_rows = (int64_t*)w.out_stage;
    d_scores = (float*)((char*)w.out_stage + (size_t)Q * k * 8);
  }
  HIP_TRY(launch_emit(w.topk_keys, w.kp, Q, k, ix->row_offset, d_scores, d_rows, s));
  if (out_mem_kind == DHR_MEM_HOST) {
    HIP_TRY(hipMemcpyAsync(out_rows, d_rows, (size_t)Q * k * 8, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync(out_scores, d_scores, (size_t)Q * k * 4, hipMemcpyDeviceToHost, s));
  }
  if (sync || ix->profile || out_mem_kind == DHR_MEM_HOST) {
    HIP_TRY(hipMemcpyAsync(w.h_stats, w.d_stats, 32, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    // device counters of a controller without read-backs: they run on from the begin call
    st.candidates_bound += (int64_t)((unsigned long long*)w.h_stats)[0] - ix->pend.dev_bound;
    st.candidates_exact += (int64_t)((unsigned long long*)w.h_stats)[1] - ix->pend.dev_exact;
    double ms[5] = {0, 0, 0, 0, 0};
    tm.collect(ms);
    st.gemm_ms += ms[T_GEMM]; st.refine_ms += ms[T_REFINE]; st.rescore_ms += ms[T_RESCORE]; st.select_ms += ms[T_SELECT];
  }
  ix->stats = st;
  ix->pend.valid = false;
  drain.armed = false;
  return DHR_OK;
}
extern "C" int dhr_search_finish(dhr_index* ix, const float* tau_hat_dev, float* out_scores, int64_t* out_rows,
                                 int32_t* out_count_dev, int32_t out_mem_kind, void* stream) try {
  return search_finish_impl(ix, tau_hat_dev, out_scores, out_rows, out_count_dev, out_mem_kind, stream, true);
} DHR_CATCH_STATUS
extern "C" int dhr_internal_search_finish_async(dhr_index* ix, const float* tau_hat_dev, float* out_scores, int64_t* out_rows,
                                                int32_t* out_count_dev, int32_t out_mem_kind, void* stream) try {
  return search_finish_impl(ix, tau_hat_dev, out_scores, out_rows, out_count_dev, out_mem_kind, stream, false);
} DHR_CATCH_STATUS

extern "C" int dhr_score_rows(dhr_index* ix, const dhr_query_batch* qb, int32_t m, const int64_t* rows, float* out_scores,
                              int32_t mem_kind, void* stream) try {
  dhr::alloc_checkpoint();
  int rc = check_queries(ix, qb);
  if (rc) return rc;
  if (m <= 0 || !rows || !out_scores) return set_error(DHR_ERR_INVALID, "bad m / null pointer");
  if (!DHR_MEM_KIND_OK(mem_kind)) return set_error(DHR_ERR_INVALID, "bad mem_kind");
  if (ix->pend.valid && !ix->pend.done)      // the staged search keeps its query batch in the workspace this call would overwrite
    return set_error(DHR_ERR_INVALID, "dhr_score_rows between dhr_search_begin and dhr_search_finish on the same handle");
  HIP_TRY(hipSetDevice(ix->device));
  hipStream_t s = (hipStream_t)stream;
  const int Q = qb->n_queries;
  const bool gate = ix->d_dlr > 0 && qb->index != nullptr && qb->index_dtype != DHR_IDX_NONE;
  Workspace& w = ix->ws;
  if ((rc = ensure_ws(ix, w, Q, 1, 0, 1, true, true)) != DHR_OK) return rc;
  if ((rc = prep_queries(ix, w, qb, s)) != DHR_OK) return rc;
  const size_t n = (size_t)Q * m;
  DevMem tmp_mem;
  void*& tmp = tmp_mem.p;                  // [rows64 (host input only)] [rows32] [scores]
  HIP_TRY(hipMalloc(&tmp, n * 16));
  int64_t* d_rows64 = (int64_t*)tmp;
  uint32_t* d_rows32 = (uint32_t*)((char*)tmp + n * 8);
  float* d_sc = (float*)((char*)tmp + n * 12);
  const int64_t* src_rows = rows;
  auto done = [&](int code) { return code; };       // (tmp_mem releases the scratch)
  if (mem_kind == DHR_MEM_HOST) {
    if (hipMemcpyAsync(d_rows64, rows, n * 8, hipMemcpyHostToDevice, s) != hipSuccess) return done(set_error(DHR_ERR_HIP, "H2D failed"));
    src_rows = d_rows64;
  }
  if (launch_rows_to_local(src_rows, (int64_t)n, ix->row_offset, ix->n_rows, d_rows32, s) != hipSuccess)
    return done(set_error(DHR_ERR_HIP, "rows_to_local launch failed"));
  RescoreArgs r = base_rescore_args(ix, w, Q, gate);
  r.rows32 = d_rows32; r.ld_rows = m; r.count_all = (uint32_t)m; r.max_count = (uint32_t)m;
  r.out_scores = (mem_kind == DHR_MEM_HOST) ? d_sc : out_scores; r.ld_scores = m;
  if (launch_rescore(r, s) != hipSuccess) return done(set_error(DHR_ERR_HIP, "rescore launch failed"));
  if (mem_kind == DHR_MEM_HOST && hipMemcpyAsync(out_scores, d_sc, n * 4, hipMemcpyDeviceToHost, s) != hipSuccess)
    return done(set_error(DHR_ERR_HIP, "D2H failed"));
  if (hipStreamSynchronize(s) != hipSuccess) return done(set_error(DHR_ERR_HIP, "score_rows failed on the device"));
  return done(DHR_OK);
} DHR_CATCH_STATUS

extern "C" int dhr_densify(int32_t device, int32_t mem_kind, const void* lexical, int32_t value_dtype, int64_t ld, int64_t batch, int32_t vocab,
                           int32_t remove_dims, int32_t dims, void* out_value, int32_t out_value_dtype, int64_t ld_value, void* out_index,
                           int32_t index_dtype, int64_t ld_index, void* stream) try {
  if (!lexical || !out_value || !out_index) return set_error(DHR_ERR_INVALID, "null pointer");
  if (!DHR_MEM_KIND_OK(mem_kind)) return set_error(DHR_ERR_INVALID, "bad mem_kind");
  if (batch < 0 || vocab <= 0 || dims <= 0 || remove_dims < 0 || remove_dims >= vocab || ld < vocab || ld_value < dims || ld_index < dims)
    return set_error(DHR_ERR_INVALID, "bad sizes / strides");
  if ((vocab - remove_dims) % dims != 0)
    return set_error(DHR_ERR_INVALID, "Input lexical representation cannot be densified, please fix dims or remove_dims");
  if ((value_dtype != DHR_VAL_F16 && value_dtype != DHR_VAL_F32) || (out_value_dtype != DHR_VAL_F16 && out_value_dtype != DHR_VAL_F32))
    return set_error(DHR_ERR_INVALID, "bad value dtype");
  const int n_groups = (vocab - remove_dims) / dims;
  if (index_dtype != DHR_IDX_U8 && index_dtype != DHR_IDX_I16) return set_error(DHR_ERR_INVALID, "index dtype must be uint8 or int16");
  if (index_dtype == DHR_IDX_U8 && n_groups > 256) return set_error(DHR_ERR_UNSUPPORTED, "more than 256 groups need the int16 index dtype");
  if (n_groups > 32767) return set_error(DHR_ERR_UNSUPPORTED, "more than 32767 groups");
  if (batch == 0) return DHR_OK;
  HIP_TRY(hipSetDevice(device));
  hipStream_t s = (hipStream_t)stream;
  const int ies = value_dtype == DHR_VAL_F32 ? 4 : 2, oes = out_value_dtype == DHR_VAL_F32 ? 4 : 2, xes = index_dtype == DHR_IDX_I16 ? 2 : 1;
  if (mem_kind == DHR_MEM_DEVICE) {
    HIP_TRY(launch_densify(lexical, value_dtype == DHR_VAL_F32, ld, batch, remove_dims, dims, n_groups, out_value, out_value_dtype == DHR_VAL_F32,
                           ld_value, out_index, index_dtype == DHR_IDX_I16, ld_index, s));
    HIP_TRY(hipStreamSynchronize(s));
    return DHR_OK;
  }
  // host arrays: stage blocks of rows through the device
  const int64_t block = std::max<int64_t>(1, std::min<int64_t>(batch, ((int64_t)256 << 20) / ((int64_t)vocab * ies)));
  DevMem m_in, m_val, m_idx;
  void *&d_in = m_in.p, *&d_val = m_val.p, *&d_idx = m_idx.p;
  auto done = [&](int code) { return code; };       // (the three DevMem release the staging buffers)
  if (hipMalloc(&d_in, (size_t)block * vocab * ies) != hipSuccess || hipMalloc(&d_val, (size_t)block * dims * oes) != hipSuccess ||
      hipMalloc(&d_idx, (size_t)block * dims * xes) != hipSuccess)
    return done(set_error(DHR_ERR_HIP, "hipMalloc failed"));
  for (int64_t lo = 0; lo < batch; lo += block) {
    const int64_t rows = std::min(block, batch - lo);
    if (hipMemcpy2DAsync(d_in, (size_t)vocab * ies, (const char*)lexical + lo * ld * ies, (size_t)ld * ies, (size_t)vocab * ies, (size_t)rows,
                         hipMemcpyHostToDevice, s) != hipSuccess)
      return done(set_error(DHR_ERR_HIP, "H2D failed"));
    if (launch_densify(d_in, value_dtype == DHR_VAL_F32, vocab, rows, remove_dims, dims, n_groups, d_val, out_value_dtype == DHR_VAL_F32, dims, d_idx,
                       index_dtype == DHR_IDX_I16, dims, s) != hipSuccess)
      return done(set_error(DHR_ERR_HIP, "densify launch failed"));
    if (hipMemcpy2DAsync((char*)out_value + lo * ld_value * oes, (size_t)ld_value * oes, d_val, (size_t)dims * oes, (size_t)dims * oes, (size_t)rows,
                         hipMemcpyDeviceToHost, s) != hipSuccess ||
        hipMemcpy2DAsync((char*)out_index + lo * ld_index * xes, (size_t)ld_index * xes, d_idx, (size_t)dims * xes, (size_t)dims * xes, (size_t)rows,
                         hipMemcpyDeviceToHost, s) != hipSuccess)
      return done(set_error(DHR_ERR_HIP, "D2H failed"));
    if (hipStreamSynchronize(s) != hipSuccess) return done(set_error(DHR_ERR_HIP, "densify failed on the device"));
  }
  return done(DHR_OK);
} DHR_CATCH_STATUS

// ------------------------------------------------------------------------------------------ product quantiser
namespace {
int pq_check(const void* a, const void* b, int64_t n, int d, int M, int64_t ld, int mem_kind) {
  if (!a || !b) return set_error(DHR_ERR_INVALID, "null pointer");
  if (!DHR_MEM_KIND_OK(mem_kind)) return set_error(DHR_ERR_INVALID, "bad mem_kind");
  if (n < 0 || d <= 0 || M <= 0 || d % M != 0 || ld < d) return set_error(DHR_ERR_INVALID, "bad sizes (d must be a multiple of M, ld >= d)");
  if (d / M > 64) return set_error(DHR_ERR_UNSUPPORTED, "sub-vectors wider than 64 columns are not supported");
  return DHR_OK;
}
// host arrays are staged whole (PQ inputs are at most the corpus, which has to fit the device anyway)
struct Staged {
  void* dev = nullptr; bool owned = false;
  int in(const void* p, size_t bytes, int mem_kind, hipStream_t s) {
    if (mem_kind == DHR_MEM_DEVICE) { dev = const_cast<void*>(p); return DHR_OK; }
    if (hipMalloc(&dev, bytes ? bytes : 16) != hipSuccess) return set_error(DHR_ERR_HIP, "hipMalloc failed");
    owned = true;
    if (p && hipMemcpyAsync(dev, p, bytes, hipMemcpyHostToDevice, s) != hipSuccess) return set_error(DHR_ERR_HIP, "H2D failed");
    return DHR_OK;
  }
  int out(void* p, size_t bytes, hipStream_t s) {
    if (!owned) return DHR_OK;
    if (hipMemcpyAsync(p, dev, bytes, hipMemcpyDeviceToHost, s) != hipSuccess) return set_error(DHR_ERR_HIP, "D2H failed");
    return DHR_OK;
  }
  ~Staged() { if (owned) hipFree(dev); }
};
}  // namespace

extern "C" int dhr_pq_train(int32_t device, int32_t mem_kind, const void* values, int64_t ld, int64_t n, int32_t d, int32_t M, int32_t iters,
                            int64_t max_points, float* codebooks, double* out_error, void* stream) try {
  return dhr_pq_train_nbits(device, mem_kind, values, ld, n, d, M, 8, iters, max_points, codebooks, out_error, stream);
} DHR_CATCH_STATUS
extern "C" int dhr_pq_train_nbits(int32_t device, int32_t mem_kind, const void* values, int64_t ld, int64_t n, int32_t d, int32_t M, int32_t nbits,
                                  int32_t iters, int64_t max_points, float* codebooks, double* out_error, void* stream) try {
  int rc = pq_check(values, codebooks, n, d, M, ld, mem_kind);
  if (rc) return rc;
  if (nbits < 1 || nbits > 8) return set_error(DHR_ERR_UNSUPPORTED, "nbits must be in [1, 8] (one code byte per sub-quantiser on the device; faiss' bit-packed rows are a file format matter)");
  const int ksub = 1 << nbits;
  if (n < 1 || iters < 0 || max_points < ksub) return set_error(DHR_ERR_INVALID, "need n >= 1, iters >= 0, max_points >= 2^nbits");
  HIP_TRY(hipSetDevice(device));
  hipStream_t s = (hipStream_t)stream;
  const int dsub = d / M;
  // training rows: every `stride`-th row (host arrays: only those rows are staged)
  const int64_t stride = std::max<int64_t>(1, n / max_points);
  const int64_t np = (n + stride - 1) / stride;
  Staged v, cb;
  int64_t v_ld = ld, v_stride = stride;
  if (mem_kind == DHR_MEM_HOST) {
    if (hipMalloc(&v.dev, (size_t)np * d * 2) != hipSuccess) return set_error(DHR_ERR_HIP, "hipMalloc failed");
    v.owned = true;
    HIP_TRY(hipMemcpy2DAsync(v.dev, (size_t)d * 2, values, (size_t)ld * stride * 2, (size_t)d * 2, (size_t)np, hipMemcpyHostToDevice, s));
    v_ld = d; v_stride = 1;
  } else {
    v.dev = const_cast<void*>(values);
  }
  const size_t cb_bytes = (size_t)M * ksub * dsub * 4;
  if ((rc = cb.in(nullptr, cb_bytes, mem_kind == DHR_MEM_HOST ? DHR_MEM_HOST : DHR_MEM_DEVICE, s)) != DHR_OK) return rc;
  if (mem_kind == DHR_MEM_DEVICE) cb.dev = codebooks;
  float* d_cb = (float*)cb.dev;
  DevMem m_sums, m_counts, m_err;
  float*& sums = (float*&)m_sums.p; uint32_t*& counts = (uint32_t*&)m_counts.p; float*& err = (float*&)m_err.p;
  auto done = [&](int code) { return code; };       // (the three DevMem release the scratch)
  if (hipMalloc((void**)&sums, cb_bytes) != hipSuccess || hipMalloc((void**)&counts, (size_t)M * ksub * 4) != hipSuccess ||
      hipMalloc((void**)&err, (size_t)M * 4) != hipSuccess)
    return done(set_error(DHR_ERR_HIP, "hipMalloc failed"));
  if (launch_pq_init((const __half*)v.dev, v_ld, np, v_stride, dsub, M, d_cb, ksub, s) != hipSuccess) return done(set_error(DHR_ERR_HIP, "pq_init launch failed"));
  for (int it = 0; it <= iters; ++it) {
    if (hipMemsetAsync(sums, 0, cb_bytes, s) != hipSuccess || hipMemsetAsync(counts, 0, (size_t)M * ksub * 4, s) != hipSuccess ||
        hipMemsetAsync(err, 0, (size_t)M * 4, s) != hipSuccess)
      return done(set_error(DHR_ERR_HIP, "memset failed"));
    if (launch_pq_assign((const __half*)v.dev, v_ld, np, v_stride, dsub, M, d_cb, nullptr, 0, sums, counts, err, ksub, s) != hipSuccess)
      return done(set_error(DHR_ERR_HIP, "pq_assign launch failed"));
    if (it == iters) break;                               // the last pass only measures the error
    if (launch_pq_update(d_cb, sums, counts, dsub, M, ksub, s) != hipSuccess) return done(set_error(DHR_ERR_HIP, "pq_update launch failed"));
  }
  if (out_error) {
    std::vector<float> e(M);
    if (hipMemcpyAsync(e.data(), err, (size_t)M * 4, hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess)
      return done(set_error(DHR_ERR_HIP, "D2H failed"));
    double t = 0;
    for (float x : e) t += x;
    *out_error = t / (double)np;
  }
  if ((rc = cb.out(codebooks, cb_bytes, s)) != DHR_OK) return done(rc);
  if (hipStreamSynchronize(s) != hipSuccess) return done(set_error(DHR_ERR_HIP, "PQ training failed on the device"));
  return done(DHR_OK);
} DHR_CATCH_STATUS

extern "C" int dhr_pq_encode(int32_t device, int32_t mem_kind, const void* values, int64_t ld, int64_t n, int32_t d, int32_t M,
                             const float* codebooks, uint8_t* codes, void* stream) try {
  return dhr_pq_encode_nbits(device, mem_kind, values, ld, n, d, M, 8, codebooks, codes, stream);
} DHR_CATCH_STATUS
extern "C" int dhr_pq_encode_nbits(int32_t device, int32_t mem_kind, const void* values, int64_t ld, int64_t n, int32_t d, int32_t M, int32_t nbits,
                                   const float* codebooks, uint8_t* codes, void* stream) try {
  int rc = pq_check(values, codebooks, n, d, M, ld, mem_kind);
  if (rc) return rc;
  if (nbits < 1 || nbits > 8) return set_error(DHR_ERR_UNSUPPORTED, "nbits must be in [1, 8]");
  const int ksub = 1 << nbits;
  if (!codes) return set_error(DHR_ERR_INVALID, "null pointer");
  if (n == 0) return DHR_OK;
  HIP_TRY(hipSetDevice(device));
  hipStream_t s = (hipStream_t)stream;
  const int dsub = d / M;
  Staged cb, cd;
  if ((rc = cb.in(codebooks, (size_t)M * ksub * dsub * 4, mem_kind, s)) != DHR_OK) return rc;
  if ((rc = cd.in(nullptr, (size_t)n * M, mem_kind == DHR_MEM_HOST ? DHR_MEM_HOST : DHR_MEM_DEVICE, s)) != DHR_OK) return rc;
  if (mem_kind == DHR_MEM_DEVICE) cd.dev = codes;
  if (mem_kind == DHR_MEM_DEVICE) {
    HIP_TRY(launch_pq_assign((const __half*)values, ld, n, 1, dsub, M, (const float*)cb.dev, (uint8_t*)cd.dev, M, nullptr, nullptr, nullptr, ksub, s));
  } else {
    const int64_t block = 1 << 18;                          // rows per staged block
    DevMem stage_mem;
    void*& stage = stage_mem.p;
    if (hipMalloc(&stage, (size_t)std::min<int64_t>(block, n) * d * 2) != hipSuccess) return set_error(DHR_ERR_HIP, "hipMalloc failed");
    for (int64_t lo = 0; lo < n; lo += block) {
      const int64_t rows = std::min(block, n - lo);
      if (hipMemcpy2DAsync(stage, (size_t)d * 2, (const char*)values + lo * ld * 2, (size_t)ld * 2, (size_t)d * 2, (size_t)rows, hipMemcpyHostToDevice, s) != hipSuccess ||
          launch_pq_assign((const __half*)stage, d, rows, 1, dsub, M, (const float*)cb.dev, (uint8_t*)cd.dev + lo * M, M, nullptr, nullptr, nullptr, ksub, s) != hipSuccess ||
          hipStreamSynchronize(s) != hipSuccess)
        return set_error(DHR_ERR_HIP, "PQ encoding failed on the device");
    }
  }
  if ((rc = cd.out(codes, (size_t)n * M, s)) != DHR_OK) return rc;
  HIP_TRY(hipStreamSynchronize(s));
  return DHR_OK;
} DHR_CATCH_STATUS

extern "C" int dhr_pq_decode(int32_t device, int32_t mem_kind, const uint8_t* codes, int64_t n, int32_t d, int32_t M, const float* codebooks,
                             void* out_values, int64_t ld_out, void* stream) try {
  return dhr_pq_decode_nbits(device, mem_kind, codes, n, d, M, 8, codebooks, out_values, ld_out, stream);
} DHR_CATCH_STATUS
extern "C" int dhr_pq_decode_nbits(int32_t device, int32_t mem_kind, const uint8_t* codes, int64_t n, int32_t d, int32_t M, int32_t nbits,
                                   const float* codebooks, void* out_values, int64_t ld_out, void* stream) try {
  int rc = pq_check(codes, codebooks, n, d, M, ld_out, mem_kind);
  if (rc) return rc;
  if (nbits < 1 || nbits > 8) return set_error(DHR_ERR_UNSUPPORTED, "nbits must be in [1, 8]");
  const int ksub = 1 << nbits;
  if (!out_values) return set_error(DHR_ERR_INVALID, "null pointer");
  if (n == 0) return DHR_OK;
  HIP_TRY(hipSetDevice(device));
  hipStream_t s = (hipStream_t)stream;
  const int dsub = d / M;
  Staged cb, cd, ov;
  if ((rc = cb.in(codebooks, (size_t)M * ksub * dsub * 4, mem_kind, s)) != DHR_OK) return rc;
  if ((rc = cd.in(codes, (size_t)n * M, mem_kind, s)) != DHR_OK) return rc;
  int64_t ld_dev = ld_out;
  if (mem_kind == DHR_MEM_HOST) {
    if ((rc = ov.in(nullptr, (size_t)n * d * 2, DHR_MEM_HOST, s)) != DHR_OK) return rc;
    ld_dev = d;
  } else {
    ov.dev = out_values;
  }
  HIP_TRY(launch_pq_decode((const uint8_t*)cd.dev, M, n, M, dsub, (const float*)cb.dev, (__half*)ov.dev, ld_dev, ksub, s));
  if (mem_kind == DHR_MEM_HOST)
    HIP_TRY(hipMemcpy2DAsync(out_values, (size_t)ld_out * 2, ov.dev, (size_t)d * 2, (size_t)d * 2, (size_t)n, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  return DHR_OK;
} DHR_CATCH_STATUS

extern "C" int dhr_debug_bound_scores(dhr_index* ix, const dhr_query_batch* qb, int64_t row_lo, int64_t row_hi,
                                      float* out_dev, void* stream) try {
  int rc = check_queries(ix, qb);
  if (rc) return rc;
  if (row_lo < 0 || row_hi > ix->n_rows || row_lo >= row_hi || !out_dev) return set_error(DHR_ERR_INVALID, "bad row range");
  HIP_TRY(hipSetDevice(ix->device));
  hipStream_t s = (hipStream_t)stream;
  Workspace& w = ix->ws;
  if ((rc = ensure_ws(ix, w, qb->n_queries, 1, 0)) != DHR_OK) return rc;
  if ((rc = prep_queries(ix, w, qb, s)) != DHR_OK) return rc;
  GemmArgs g{};
  g.a_tiles = ix->tiles; g.b_tiles = w.q_tiles; g.ksteps = ix->ksteps; g.ts = ix->ts; g.td = ix->td; g.ts_q = w.ts_q; g.variant = ix->gemm_variant; g.i8_mul = (ix->dense_i8 || ix->gated_i8) ? w.i8_mul : nullptr; g.g8_shift = ix->gated_i8 ? w.g8_shift : nullptr; g.g8_rsum = ix->g8_rsum; g.seq_lo = row_lo / TILE_ROWS;
  g.seq_hi = (row_hi + TILE_ROWS - 1) / TILE_ROWS; g.map_mode = 0; g.period = 1; g.head = 0; g.n_tiles = ix->n_tiles; g.n_qtiles = w.q_pad / TILE_ROWS; g.n_rows = ix->n_rows;
  g.thr = w.thr; g.cand = w.cand; g.cnt = w.cnt; g.cap = (uint32_t)w.cap; g.n_queries = qb->n_queries;
  g.dump = out_dev; g.dump_ld = row_hi - row_lo; g.dump_row0 = row_lo;
  HIP_TRY(launch_gemm_filter(g, s));
  HIP_TRY(hipStreamSynchronize(s));
  return DHR_OK;
} DHR_CATCH_STATUS

extern "C" int dhr_debug_query_margins(dhr_index* ix, const dhr_query_batch* qb, float* out_host, void* stream) try {
  int rc = check_queries(ix, qb);
  if (rc) return rc;
  if (!out_host) return set_error(DHR_ERR_INVALID, "null output pointer");
  HIP_TRY(hipSetDevice(ix->device));
  hipStream_t s = (hipStream_t)stream;
  Workspace& w = ix->ws;
  if ((rc = ensure_ws(ix, w, qb->n_queries, 1, 0)) != DHR_OK) return rc;
  if ((rc = prep_queries(ix, w, qb, s)) != DHR_OK) return rc;
  HIP_TRY(hipMemcpyAsync(out_host, w.margin, (size_t)qb->n_queries * 4, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  return DHR_OK;
} DHR_CATCH_STATUS

// Kernel-tuning hook: the bound GEMM alone over the whole shard with the filter closed (thr = +inf),
// `iters` launches, average milliseconds per launch (hipEvents on the stream).
extern "C" void dhr_debug_seq_to_tile(int64_t seq, int32_t map_mode, int32_t period, int64_t head, int64_t perm_mul, int64_t perm_n, int64_t out[2]) try {
  out[0] = seq_to_tile_fast(seq, map_mode, period, head, perm_mul, perm_n, 1.0 / (double)(perm_n > 0 ? perm_n : 1), 1.0 / (double)(period > 1 ? period - 1 : 1));
  out[1] = seq_to_tile(seq, map_mode, period, head, perm_mul, perm_n);
} DHR_CATCH_VOID
extern "C" int dhr_debug_gemm_time(dhr_index* ix, const dhr_query_batch* qb, int32_t iters, double* ms_out, double* flops_out,
                                   void* stream) try {
  int rc = check_queries(ix, qb);
  if (rc) return rc;
  if (iters <= 0 || !ms_out) return set_error(DHR_ERR_INVALID, "bad iters / null pointer");
  HIP_TRY(hipSetDevice(ix->device));
  hipStream_t s = (hipStream_t)stream;
  Workspace& w = ix->ws;
  // DHR_GEMM_TIME_OPEN=1 (tuning): filter with the final thresholds of the previous dhr_search on this handle (same queries),
  // i.e. a realistic hit rate in the epilogue, instead of the closed filter.  (The workspace must be the one that search left:
  // same k, or ensure_ws would re-allocate it and the thresholds would be uninitialised memory -- as they were for a while.)
  const bool open = getenv("DHR_GEMM_TIME_OPEN") && atoi(getenv("DHR_GEMM_TIME_OPEN")) != 0 && w.thr_hat && ix->stats.k > 0 && ix->stats.n_queries == qb->n_queries;
  if ((rc = ensure_ws(ix, w, qb->n_queries, open ? (int)ix->stats.k : 1, 0)) != DHR_OK) return rc;
  if ((rc = prep_queries(ix, w, qb, s)) != DHR_OK) return rc;
  if (open) {
    HIP_TRY(hipMemcpyAsync(w.thr, w.thr_hat, (size_t)w.q_pad * 4, hipMemcpyDeviceToDevice, s));
  } else {
    std::vector<float> inf((size_t)w.q_pad, INFINITY);
    HIP_TRY(hipMemcpyAsync(w.thr, inf.data(), (size_t)w.q_pad * 4, hipMemcpyHostToDevice, s));
  }
  const bool opened = open;
  HIP_TRY(hipMemsetAsync(w.cnt, 0, (size_t)w.q_pad * 4, s));
  GemmArgs g{};
  g.a_tiles = ix->tiles; g.b_tiles = w.q_tiles; g.ksteps = ix->ksteps; g.ts = ix->ts; g.td = ix->td; g.ts_q = w.ts_q; g.variant = ix->gemm_variant; g.i8_mul = (ix->dense_i8 || ix->gated_i8) ? w.i8_mul : nullptr; g.g8_shift = ix->gated_i8 ? w.g8_shift : nullptr; g.g8_rsum = ix->g8_rsum; g.seq_lo = 0; g.seq_hi = ix->n_tiles; g.map_mode = 0;
  g.period = 1; g.head = 0; g.n_tiles = ix->n_tiles; g.n_qtiles = w.q_pad / TILE_ROWS; g.n_rows = ix->n_rows;
  g.thr = w.thr; g.cand = w.cand; g.cnt = w.cnt; g.cap = (uint32_t)w.cap; g.n_queries = qb->n_queries;
  HIP_TRY(launch_gemm_filter(g, s));                      // warm-up
  Events evs;
  hipEvent_t e0, e1;
  HIP_TRY(evs.add(&e0)); HIP_TRY(evs.add(&e1));
  HIP_TRY(hipEventRecord(e0, s));
  for (int i = 0; i < iters; ++i) {
    if (opened) HIP_TRY(hipMemsetAsync(w.cnt, 0, (size_t)w.q_pad * 4, s));       // the lists fill as in a search (a full list takes the cold surplus path)
    HIP_TRY(launch_gemm_filter(g, s));
  }
  HIP_TRY(hipEventRecord(e1, s));
  HIP_TRY(hipStreamSynchronize(s));
  float ms = 0.f;
  HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
  *ms_out = ms / iters;
  if (flops_out) *flops_out = 2.0 * (double)w.q_pad * (double)ix->n_tiles * TILE_ROWS * (double)ix->kt;
  return DHR_OK;
} DHR_CATCH_STATUS

// ------------------------------------------------------------------------------------------ shard reduce
extern "C" int dhr_merge_topk(int32_t device, int32_t n_queries, int32_t n_in, const float* in_scores,
                              const int64_t* in_rows, int32_t k_out, float* out_scores, int64_t* out_rows, void* stream) try {
  if (n_queries <= 0 || n_in <= 0 || k_out <= 0 || !in_scores || !in_rows || !out_scores || !out_rows)
    return set_error(DHR_ERR_INVALID, "bad argument");
  HIP_TRY(hipSetDevice(device));
  if (n_in > 16384) {       // beyond one workgroup's LDS: two stable segmented sorts through global memory (select_global.hip)
    if ((int64_t)n_queries * n_in > (int64_t)0x7fffffff) return set_error(DHR_ERR_UNSUPPORTED, "more than 2^31 entries in one device reduce");
    HIP_TRY(launch_merge_topk_global(n_queries, n_in, in_scores, in_rows, k_out, out_scores, out_rows, (hipStream_t)stream));
    return DHR_OK;
  }
  HIP_TRY(launch_merge_topk(n_queries, n_in, in_scores, in_rows, k_out, out_scores, out_rows, (hipStream_t)stream));
  return DHR_OK;
} DHR_CATCH_STATUS

extern "C" int dhr_merge_topk_lists(int32_t device, int32_t n_queries, int32_t n_lists, int32_t list_len, const float* in_scores,
                                    const int64_t* in_rows, int32_t k_out, float* out_scores, int64_t* out_rows, void* stream) try {
  if (n_queries <= 0 || n_lists <= 0 || list_len <= 0 || k_out <= 0 || !in_scores || !out_scores || (in_rows && !out_rows))
    return set_error(DHR_ERR_INVALID, "bad argument");
  if (((int64_t)n_lists * list_len + k_out) * (in_rows ? 12 : 4) > 160 * 1024 || n_lists > 64)
    return set_error(DHR_ERR_UNSUPPORTED, "the lists of one query do not fit the LDS ((n_lists*list_len + k_out)*12 B > 160 KiB) or n_lists > 64");
  HIP_TRY(hipSetDevice(device));
  HIP_TRY(launch_merge_lists(n_queries, n_lists, list_len, in_scores, in_rows, k_out, out_scores, in_rows ? out_rows : nullptr,
                             (hipStream_t)stream));
  return DHR_OK;
} DHR_CATCH_STATUS

extern "C" int dhr_merge_topk_lists_host(int32_t n_queries, int32_t n_lists, int32_t list_len, const float* in_scores,
                                         const int64_t* in_rows, int32_t k_out, float* out_scores, int64_t* out_rows) try {
  if (n_queries <= 0 || n_lists <= 0 || list_len <= 0 || k_out <= 0 || !in_scores || !out_scores || (in_rows && !out_rows))
    return set_error(DHR_ERR_INVALID, "bad argument");
  std::vector<int64_t> order;
  for (int q = 0; q < n_queries; ++q) {
    order.clear();
    for (int l = 0; l < n_lists; ++l)
      for (int j = 0; j < list_len; ++j) {
        const int64_t src = ((int64_t)l * n_queries + q) * list_len + j;
        if (!in_rows || in_rows[src] >= 0) order.push_back(src);
      }
    const int take = std::min<int>(k_out, (int)order.size());
    std::partial_sort(order.begin(), order.begin() + take, order.end(), [&](int64_t a, int64_t b) {
      const uint32_t ka = f32_ordered(in_scores[a]), kb = f32_ordered(in_scores[b]);
      if (ka != kb) return ka > kb;
      if (in_rows && in_rows[a] != in_rows[b]) return in_rows[a] < in_rows[b];
      return a < b;                                        // list order, then position
    });
    for (int j = 0; j < k_out; ++j) {
      out_scores[(size_t)q * k_out + j] = j < take ? in_scores[order[j]] : -INFINITY;
      if (in_rows) out_rows[(size_t)q * k_out + j] = j < take ? in_rows[order[j]] : -1;
    }
  }
  return DHR_OK;
} DHR_CATCH_STATUS

extern "C" int dhr_merge_topk_host(int32_t n_queries, int32_t n_in, const float* in_scores, const int64_t* in_rows,
                                   int32_t k_out, float* out_scores, int64_t* out_rows) try {
  if (n_queries <= 0 || n_in <= 0 || k_out <= 0 || !in_scores || !in_rows || !out_scores || !out_rows)
    return set_error(DHR_ERR_INVALID, "bad argument");
  std::vector<int> order;
  for (int q = 0; q < n_queries; ++q) {
    const float* s = in_scores + (size_t)q * n_in;
    const int64_t* r = in_rows + (size_t)q * n_in;
    order.clear();
    for (int j = 0; j < n_in; ++j)
      if (r[j] >= 0) order.push_back(j);
    const int take = std::min<int>(k_out, (int)order.size());
    std::partial_sort(order.begin(), order.begin() + take, order.end(), [&](int a, int b) {
      const uint32_t ka = f32_ordered(s[a]), kb = f32_ordered(s[b]);
      if (ka != kb) return ka > kb;
      return r[a] < r[b];
    });
    for (int j = 0; j < k_out; ++j) {
      out_scores[(size_t)q * k_out + j] = j < take ? s[order[j]] : -INFINITY;
      out_rows[(size_t)q * k_out + j] = j < take ? r[order[j]] : -1;
    }
  }
  return DHR_OK;
} DHR_CATCH_STATUS
