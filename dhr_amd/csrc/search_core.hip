// The phase controller of the search (round 6: split out of api.hip): workspace, query preparation, sampled thresholds, the main pass.
//
// Search = phases over growing corpus chunks:
//   phase 0   : threshold bootstrap (sampled searches) or the first rows scored exhaustively (exact), to seed every query's tau
//   phase p>0 : bound GEMM over the next chunk with the fused filter  U >= tau - margin  -> candidate
//               lists; refine; exact rescoring of the candidates; per-query top-k merge -> new tau.
// tau (exact k-th best so far) never exceeds the final k-th best and U >= exact score, so no row of
// the true top-k is ever dropped; chunk sizes adapt to the observed candidate counts, and a phase
// whose candidate list overflowed is re-run in halves (a chunk of <= cap rows cannot overflow).
#include "dhr_state.h"

// ------------------------------------------------------------------------------------------ workspace
template <typename T>
static hipError_t re_malloc(T*& p, size_t bytes, int64_t& total) {
  if (p) hipFree(p);
  p = nullptr;
  total += (int64_t)bytes;
  return hipMalloc((void**)&p, bytes ? bytes : 16);
}

// use_refine: the batch goes through the refine step (gated batch on an index with heavy lists).  An UNGATED batch on such an index
// (plain inner product: --IP stage 1) rescores its bound lists directly, so they must not be deeper than the key buffer: it gets the
// list depths of an index without heavy lists.  (Until round 3 it kept the 262 144-entry bound lists over a 32 768-entry key buffer:
// a query with more than 32 768 bound candidates in one chunk wrote its keys over the next queries' -- found by the verification
// failures of the --IP mode at full size, 62 of 6 980 queries per step.)
// queries_only: the caller needs the per-query buffers only (dhr_score_rows: no lists, no running top-k) -- ANY workspace of this index
// with enough query rows serves, so that stage 2 of a composed --rerank / --PQIP step between two searches does not free and re-allocate
// the multi-GB lists every time (hipFree synchronises the device).
int ensure_ws(dhr_index* ix, Workspace& w, int n_queries, int k, int64_t keys_ld_min, int64_t cap_mult, bool use_refine, bool queries_only) {
  const int q_pad = (int)round_up(n_queries, TILE_ROWS);
  // (q_pad is the ACTIVE padded query count: query_prep_kernel prepares rows < q_pad only, so the re-use must set it -- a batch that
  // followed a smaller one through this return kept the smaller count and scored its later queries against stale operand rows)
  if (queries_only && w.q_alloc >= q_pad && w.kt == ix->kt && w.q32 != nullptr) { w.q_pad = q_pad; return DHR_OK; }
  int kp = 1;
  while (kp < k) kp <<= 1;
  if (kp < 64) kp = 64;
  const bool refine = (ix->heavy_key != nullptr || ix->resid8 != nullptr) && use_refine;
  // default list depth: 262144 (refine) / 65536, but never more than ~32 GiB for the two bound-list sets of a big batch
  int64_t cap = 0, cap_r = 0, keys_ld = 0, cap_deep = 0, arena = 0;
  auto plan_depths = [&](int q_for_cap) {
    int64_t base_cap = refine ? 262144 : 65536;
    // ... sized by the SHARD: a list cannot hold more rows than the shard has, and the chunk planner of the main pass cuts the pass
    // so that the fullest list fits, so a 1/8 shard gets 1/8 of the depth (floor: 32768 / 16384) instead of the full-corpus workspace
    // (measured at config 3: depth 262 144 / 131 072 / 65 536 = 195.2 / 196.5 / 196.2 ms per step, 100.2 / 85.2 / 77.7 GB)
    // (gated_i8 indexes: the int8 bound passes ~1.5x the rows of the fp16 one, and far more for the few queries with two or three
    // dominant terms -- the fullest list decides the chunk count of the main pass, so these get 4x the depth: 32 -> 8 chunks at config 3)
    // Round 3: n_rows / 8, capped at 262 144 -- a 1/8 shard of config 4 planned 22 chunks at n_rows / 32 (its fullest list is as long as
    // the whole corpus's in proportion, but the floor of the depth is not), each with its own host round trips.
    const int64_t by_rows = ix->gated_i8 ? std::min<int64_t>(std::max<int64_t>(ix->n_rows / 8, 32768), 262144)
                                         : std::max<int64_t>(ix->n_rows / 128, refine ? 32768 : 16384);
    while (base_cap > by_rows && base_cap > 4096) base_cap >>= 1;          // power-of-two floor of n_rows / 128 (65 536 at 8.84 M rows)
    while (base_cap > 4096 && (int64_t)q_for_cap * base_cap * 16 > ((int64_t)32 << 30)) base_cap >>= 1;
    if (ix->cand_cap > 0) base_cap = ix->cand_cap;
    // fallback depths serve a handful of queries: 16x deeper lists there cost little memory
    cap = std::min<int64_t>(base_cap * cap_mult, (int64_t)1 << 22);
    cap_deep = cap;
    // Two-tier lists (round 5).  The depth above is what the HOTTEST query of a batch needs (a few per cent of the queries pass 10-100 x the
    // average through the filter); as the stride of [q_pad][cap] arrays it cost config 3 two 15 GB list sets of which a step fills 0.1 GB.
    // Now every query owns `stride` slots and a hot one gets the rest of its depth from an arena shared by the batch, planned on the device
    // from the previous launch's list lengths (plan_overflow_kernel).  Only where the bound lists are read by a refine level (the rescoring
    // kernel and the key buffer keep their uniform stride), for the first attempt (the fallback depths serve a handful of queries), and not
    // when the caller fixed the depth (DHR_PARAM_CAND_CAP).
    const int64_t stride = ix->list_stride > 0 ? ix->list_stride : 32768;
    arena = 0;
    const int variant = ix->gemm_variant ? ix->gemm_variant : g_gemm_variant;
    const bool kernel_writes_tier = ix->gated_i8 || variant != 4;      // (the 4-wave kernel writes the uniform part only: dhr_internal.h cand_store)
    if (refine && cap_mult == 1 && ix->cand_cap <= 0 && cap > stride && kernel_writes_tier) {
      cap = stride;
      arena = std::max<int64_t>((int64_t)4 << 20, std::min<int64_t>((int64_t)q_for_cap * 8192, (int64_t)128 << 20));
      arena = std::max(arena, 2 * (cap_deep - cap));
    }
    // survivor lists: 32 768 entries, and at least 4 x the padded k (agip_topk 10 000: a chunk of the main pass must be able to bring
    // a hot query's share of its 10 000 best -- 12 queries per step overflowed 32 768 and were redone)
    cap_r = refine ? std::min<int64_t>(cap_deep, std::max<int64_t>(32768, 4 * (int64_t)kp) * cap_mult) : cap;
    keys_ld = std::max<int64_t>(cap_r, keys_ld_min);
  };
  // A SMALLER batch re-uses the buffers of a larger one (same list depths and strides; q_pad is the ACTIVE padded query count).  Until round 4
  // any other batch size freed and re-allocated the whole workspace -- tens of GB, and hipFree synchronises the device: the repair of ONE
  // failed query of a sharded step (dhr_search on a sub-batch, then the next full batch) cost 0.7 s.
  // (the list depths of the LARGER batch: beyond ~8 000 queries they are halved to bound the memory)
  if (w.q_alloc >= q_pad && w.kt == ix->kt) {
    plan_depths(w.q_alloc);
    if (w.kp == kp && w.cap == cap && w.cap_deep == cap_deep && w.arena == arena && w.cap_r == cap_r && w.keys_ld >= keys_ld) { w.q_pad = q_pad; return DHR_OK; }
  }
  plan_depths(q_pad);
  free_ws(w);
  int64_t tot = 0;
  HIP_TRY(re_malloc(w.q_tiles, (size_t)q_pad * ix->kt * 2, tot));
  HIP_TRY(re_malloc(w.q32, (size_t)q_pad * ix->k_rm * 4, tot));
  HIP_TRY(re_malloc(w.q_idx, (size_t)q_pad * std::max(ix->d_dlr, 8) * 2, tot));
  HIP_TRY(re_malloc(w.q16, (size_t)q_pad * ix->k_rm * 2, tot));
  HIP_TRY(re_malloc(w.q_idx8, (size_t)q_pad * std::max(ix->d_dlr, 8), tot));
  HIP_TRY(re_malloc(w.q_inexact, 16, tot));
  HIP_TRY(re_malloc(w.margin, (size_t)q_pad * 4, tot));
  HIP_TRY(re_malloc(w.i8_mul, (size_t)q_pad * 4, tot));
  if (ix->gated_i8) {
    HIP_TRY(re_malloc(w.g8_q8, (size_t)q_pad * ix->d_dlr, tot));
    HIP_TRY(re_malloc(w.g8_shift, (size_t)q_pad * 4, tot));
    HIP_TRY(re_malloc(w.g8_unit, (size_t)q_pad * 4, tot));
  }
  HIP_TRY(re_malloc(w.tau, (size_t)q_pad * 4, tot));
  HIP_TRY(re_malloc(w.thr, (size_t)q_pad * 4, tot));
  HIP_TRY(re_malloc(w.cnt, (size_t)q_pad * 4, tot));
  HIP_TRY(re_malloc(w.cand, (size_t)q_pad * cap * 8, tot));
  // The keys of the exact rescoring (score bits | row) overwrite the survivor entries they were computed from: entry i of a query is read
  // (its row) and written (its key) by the same wave of rescore_kernel, nothing reads the survivor lists afterwards, and both are 8 bytes
  // -- with a refine level and equal strides the key buffer IS the survivor array (1.9 GB of a config-3 workspace).
  const bool alias = refine && keys_ld == cap_r;
  if (!alias) HIP_TRY(re_malloc(w.rs_keys, (size_t)q_pad * keys_ld * 8, tot));
  HIP_TRY(re_malloc(w.topk_keys, (size_t)q_pad * kp * 8, tot));
  HIP_TRY(re_malloc(w.d_max, 16, tot));
  HIP_TRY(re_malloc(w.tau_hat, (size_t)q_pad * 4, tot));
  HIP_TRY(re_malloc(w.fail_flags, (size_t)q_pad * 4, tot));
  HIP_TRY(re_malloc(w.thr_hat, (size_t)q_pad * 4, tot));
  HIP_TRY(re_malloc(w.blk_off, (size_t)2 * (q_pad + 1) * 4, tot));
  HIP_TRY(re_malloc(w.boot_rows, (size_t)q_pad * BOOT_M * 4, tot));
  HIP_TRY(re_malloc(w.boot_bound, (size_t)q_pad * TILE_ROWS * 4, tot));
  if (ix->resid8) HIP_TRY(re_malloc(w.thr_raise, (size_t)q_pad * 4, tot));
  if (arena > 0) {
    HIP_TRY(re_malloc(w.ovf, (size_t)arena * 8, tot));
    HIP_TRY(re_malloc(w.ovf_off, (size_t)q_pad * 4, tot));
    HIP_TRY(re_malloc(w.ovf_cap, (size_t)q_pad * 4, tot));
    HIP_TRY(re_malloc(w.cnt_plan, (size_t)q_pad * 4, tot));
    HIP_TRY(re_malloc(w.tier_dev, 2 * sizeof(ListTier), tot));
    { const ListTier t0{w.ovf, w.ovf_off, w.ovf_cap}; HIP_TRY(hipMemcpy(w.tier_dev, &t0, sizeof t0, hipMemcpyHostToDevice)); }
    HIP_TRY(hipMemset(w.cnt_plan, 0, (size_t)q_pad * 4));
    HIP_TRY(hipMemset(w.ovf_off, 0, (size_t)q_pad * 4));
    HIP_TRY(hipMemset(w.ovf_cap, 0, (size_t)q_pad * 4));
  }
  if (refine) {
    HIP_TRY(re_malloc(w.q_pack, (size_t)q_pad * std::max(ix->d_dlr, 8) * 4, tot));
    HIP_TRY(re_malloc(w.cand_r, (size_t)q_pad * cap_r * 8, tot));
    HIP_TRY(re_malloc(w.cnt_r, (size_t)q_pad * 4, tot));
    if (alias) { w.rs_keys = (uint64_t*)w.cand_r; w.keys_alias = true; }
  }
  HIP_TRY(hipHostMalloc(&w.h_pinned, 16, hipHostMallocDefault));
  HIP_TRY(hipHostMalloc((void**)&w.h_pinned2, 32, hipHostMallocDefault));
  HIP_TRY(re_malloc(w.d_max2, 32, tot));
  HIP_TRY(re_malloc(w.d_ref, 16, tot));
  HIP_TRY(hipHostMalloc(&w.h_ref, 16, hipHostMallocDefault));
  HIP_TRY(re_malloc(w.d_stats, 32, tot));
  HIP_TRY(hipHostMalloc(&w.h_stats, 32, hipHostMallocDefault));
  w.q_pad = w.q_alloc = q_pad; w.kp = kp; w.cap = cap; w.cap_deep = cap_deep; w.arena = arena; w.cap_r = cap_r; w.keys_ld = keys_ld; w.kt = ix->kt; w.d_dlr = ix->d_dlr;
  w.bytes = tot;
  return DHR_OK;
}

int check_queries(const dhr_index* ix, const dhr_query_batch* qb) {
  if (!ix || !qb) return set_error(DHR_ERR_INVALID, "null argument");
  if (qb->n_queries <= 0) return set_error(DHR_ERR_INVALID, "n_queries must be > 0");
  // the bound GEMM's grid carries DOC_GROUP x (padded queries / 256) in one 16-bit dimension (launch_gemm_filter)
  if ((int64_t)DOC_GROUP * ((qb->n_queries + TILE_ROWS - 1) / TILE_ROWS) > 65535)
    return set_error(DHR_ERR_UNSUPPORTED, "more than 4 194 048 queries in one call: split the batch (the Python mirror hands over 8 192 at a time)");
  if (!qb->value || qb->ld_value < ix->k - ix->dlr_pad) return set_error(DHR_ERR_INVALID, "bad query value pointer / ld_value");
  if (qb->value_dtype != DHR_VAL_F16 && qb->value_dtype != DHR_VAL_F32) return set_error(DHR_ERR_INVALID, "bad value_dtype");
  const bool has_idx = qb->index != nullptr && qb->index_dtype != DHR_IDX_NONE;
  if (has_idx && ix->d_dlr == 0)
    return set_error(DHR_ERR_INVALID, "the query batch has an index array but the corpus index was built without one");
  if (has_idx && (qb->index_dtype < DHR_IDX_U8 || qb->index_dtype > DHR_IDX_I16)) return set_error(DHR_ERR_INVALID, "bad index_dtype");
  if (has_idx && qb->ld_index < ix->d_dlr - ix->dlr_pad) return set_error(DHR_ERR_INVALID, "bad query ld_index");
  if (qb->mem_kind != DHR_MEM_HOST && qb->mem_kind != DHR_MEM_DEVICE) return set_error(DHR_ERR_INVALID, "bad mem_kind");
  return DHR_OK;
}

// internal sub-batches (the queries a fallback redoes) are gathered from the library's own padded copies: their records already have the
// padded width, unlike a caller's batch (dhr_index::dlr_pad)
int grow(void*& p, size_t& have, size_t need, int64_t& total) {
  if (have >= need) return DHR_OK;
  if (p) hipFree(p);
  p = nullptr;
  HIP_TRY(hipMalloc(&p, need));
  total += (int64_t)(need - have);
  have = need;
  return DHR_OK;
}

// queries -> device operand tiles / fp32 copy / idx / margins (all inside the workspace)
int prep_queries(dhr_index* ix, Workspace& w, const dhr_query_batch* qb, hipStream_t s) {
  const void* v = qb->value;
  const void* qi = qb->index;
  int64_t ldv = qb->ld_value, ldi = qb->ld_index;
  const int es = qb->value_dtype == DHR_VAL_F32 ? 4 : 2;
  if (ix->dlr_pad > 0 && qb->mem_kind != MEM_DEVICE_PADDED) {          // the caller's [gated | ungated] records -> [gated | zero slices | ungated], index -> [index | zeros]
    const hipMemcpyKind kind = qb->mem_kind == DHR_MEM_HOST ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice;
    const int d_in = ix->d_dlr - ix->dlr_pad;
    int rc = grow(w.q_stage, w.q_stage_bytes, (size_t)qb->n_queries * ix->k * es, w.bytes);
    if (rc) return rc;
    HIP_TRY(hipMemsetAsync(w.q_stage, 0, (size_t)qb->n_queries * ix->k * es, s));
    HIP_TRY(hipMemcpy2DAsync(w.q_stage, (size_t)ix->k * es, qb->value, (size_t)qb->ld_value * es, (size_t)d_in * es, (size_t)qb->n_queries, kind, s));
    if (ix->d_cls > 0)
      HIP_TRY(hipMemcpy2DAsync((char*)w.q_stage + (size_t)ix->d_dlr * es, (size_t)ix->k * es, (const char*)qb->value + (size_t)d_in * es,
                               (size_t)qb->ld_value * es, (size_t)ix->d_cls * es, (size_t)qb->n_queries, kind, s));
    v = w.q_stage; ldv = ix->k;
    if (qb->index && qb->index_dtype != DHR_IDX_NONE) {
      const int ies = idx_esize(qb->index_dtype);
      rc = grow(w.qi_stage, w.qi_stage_bytes, (size_t)qb->n_queries * ix->d_dlr * ies, w.bytes);
      if (rc) return rc;
      HIP_TRY(hipMemsetAsync(w.qi_stage, 0, (size_t)qb->n_queries * ix->d_dlr * ies, s));
      HIP_TRY(hipMemcpy2DAsync(w.qi_stage, (size_t)ix->d_dlr * ies, qb->index, (size_t)qb->ld_index * ies, (size_t)d_in * ies,
                               (size_t)qb->n_queries, kind, s));
      qi = w.qi_stage; ldi = ix->d_dlr;
    }
  } else if (qb->mem_kind == DHR_MEM_HOST) {
    int rc = grow(w.q_stage, w.q_stage_bytes, (size_t)qb->n_queries * ix->k * es, w.bytes);
    if (rc) return rc;
    HIP_TRY(hipMemcpy2DAsync(w.q_stage, (size_t)ix->k * es, qb->value, (size_t)qb->ld_value * es, (size_t)ix->k * es,
                             (size_t)qb->n_queries, hipMemcpyHostToDevice, s));
    v = w.q_stage; ldv = ix->k;
    if (ix->d_dlr > 0 && qb->index && qb->index_dtype != DHR_IDX_NONE) {
      const int ies = idx_esize(qb->index_dtype);
      rc = grow(w.qi_stage, w.qi_stage_bytes, (size_t)qb->n_queries * ix->d_dlr * ies, w.bytes);
      if (rc) return rc;
      HIP_TRY(hipMemcpy2DAsync(w.qi_stage, (size_t)ix->d_dlr * ies, qb->index, (size_t)qb->ld_index * ies,
                               (size_t)ix->d_dlr * ies, (size_t)qb->n_queries, hipMemcpyHostToDevice, s));
      qi = w.qi_stage; ldi = ix->d_dlr;
    }
  }
  w.ts_q = sparse_query_stages(ix->ts, ix->d_dlr > 0 && qb->index, ix->gated_i8);
  G8Prep g8{};
  if (ix->gated_i8) { g8.inv_cs = ix->g8_inv_cs; g8.w = ix->g8_w; g8.s_ref = ix->g8_sref; g8.max_shift = ix->g8_max_shift; g8.q8 = w.g8_q8; g8.shift = w.g8_shift; g8.unit = w.g8_unit; }
  if (ix->resid8) { g8.thr_raise = w.thr_raise; g8.resid_ec2 = ix->resid_ec2; }
  HIP_TRY(hipMemsetAsync(w.q_inexact, 0, 8, s));      // [0] some query is not fp16-representable, [1] some query has an all-zero chunk
  HIP_TRY(launch_query_prep(v, qb->value_dtype == DHR_VAL_F32, ldv, (ix->d_dlr > 0 && qb->index) ? qi : nullptr, qb->index_dtype, ldi,
                            qb->n_queries, w.q_pad, ix->d_dlr, ix->d_cls, ix->k_rm, ix->n_buckets, ix->kt, ix->bucket_map,
                            ix->abs_mode, ix->dmax, w.q_tiles, w.q32,
                            w.q_idx, w.margin, w.tau, w.thr, ix->ts, ix->td, w.q_pack, w.q16, w.q_idx8, w.q_inexact, ix->idx_dtype,
                            ix->dense_i8 ? ix->i8_scale : 0.f, ix->i8_ec, ix->i8_nc, w.i8_mul, ix->i8_col_scale, g8, s));
  return DHR_OK;
}


// The refine step serves gated batches, and -- on gated_i8 indexes -- ungated ones too (--IP stage 1): there it takes the int8
// products of a row's listed entries off the bound and puts their real products back, whatever the index values (RefineArgs::ungated).
static inline bool uses_refine(const dhr_index* ix, bool gate) { return (ix->heavy_key != nullptr && (gate || ix->gated_i8)) || ix->resid8 != nullptr; }
RescoreArgs base_rescore_args(const dhr_index* ix, const Workspace& w, int n_queries, bool gate) {
  RescoreArgs r{};
  r.vals_rm = ix->vals_rm; r.c_idx = ix->c_idx; r.c_idx_dtype = ix->idx_dtype;
  r.q32 = w.q32; r.q_idx = w.q_idx; r.d_dlr = ix->d_dlr; r.k_rm = ix->k_rm;
  r.q16 = w.q16; r.q_idx8 = w.q_idx8; r.q_inexact = w.q_inexact;
  r.n_rows = ix->n_rows; r.n_queries = n_queries; r.gate = gate ? 1 : 0;
  return r;
}

// One bound-GEMM launch over sequence positions [lo,hi) + candidate statistics read back.
static int gemm_phase(dhr_index* ix, Workspace& w, int Q, int64_t lo, int64_t hi, int map_mode, int period, int64_t head,
                      Timer& tm, dhr_search_stats& st, hipStream_t s, uint32_t* maxc, unsigned long long* sumc) {
  GemmArgs g{};
  g.a_tiles = ix->tiles; g.b_tiles = w.q_tiles; g.ksteps = ix->ksteps; g.k_split = ix->d_dlr / TILE_K; g.ts = ix->ts; g.td = ix->td; g.ts_q = w.ts_q; g.variant = ix->gemm_variant; g.i8_mul = (ix->dense_i8 || ix->gated_i8) ? w.i8_mul : nullptr; g.g8_shift = ix->gated_i8 ? w.g8_shift : nullptr; g.g8_rsum = ix->g8_rsum;
  g.seq_lo = lo; g.seq_hi = hi; g.map_mode = map_mode; g.period = period; g.head = head; g.n_tiles = ix->n_tiles;
  g.n_qtiles = w.q_pad / TILE_ROWS; g.n_rows = ix->n_rows; g.thr = w.thr; g.cand = w.cand; g.cnt = w.cnt;
  g.cap = (uint32_t)w.cap; g.n_queries = Q;
  HIP_TRY(hipMemsetAsync(w.cnt, 0, (size_t)w.q_pad * 4, s));
  HIP_TRY(hipMemsetAsync(w.d_max, 0, 16, s));
  tm.begin(T_GEMM); HIP_TRY(launch_gemm_filter(g, s)); tm.end();
  ix->last_gemm_kernel = g_last_gemm_kernel;
  HIP_TRY(launch_max_u32(w.cnt, Q, w.d_max, (unsigned long long*)(w.d_max + 2), s));
  HIP_TRY(hipMemcpyAsync(w.h_pinned, w.d_max, 16, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  *maxc = ((uint32_t*)w.h_pinned)[0];
  memcpy(sumc, (uint32_t*)w.h_pinned + 2, 8);
  const double rows = (double)(hi - lo) * TILE_ROWS;
  st.phases++;
  st.gemm_rows += (int64_t)rows;
  st.gemm_flops += 2.0 * (double)w.q_pad * rows * (double)ix->kt;
  st.gemm_flops_alg += 2.0 * (double)Q * rows * (double)ix->k;
  return DHR_OK;
}

// The same without a host read-back (controller of the first attempt of a sampled search): list lengths stay in device memory -- the
// statistics are accumulated there (w.d_stats), a list that overflowed flags its query (fail_flags: redone by the fallback), and the
// per-candidate kernels are launched with a fixed grid that walks the block list by grid stride.
static uint32_t async_grid() {
  return FLAT_GRID_ASYNC;
}
static int gemm_phase_async(dhr_index* ix, Workspace& w, int Q, int64_t lo, int64_t hi, int map_mode, int period, int64_t head,
                            Timer& tm, dhr_search_stats& st, hipStream_t s) {
  GemmArgs g{};
  if (w.arena > 0) g.tier = w.tier_dev;      // two-tier lists: planned by the caller (stream_phases)
  g.a_tiles = ix->tiles; g.b_tiles = w.q_tiles; g.ksteps = ix->ksteps; g.k_split = ix->d_dlr / TILE_K; g.ts = ix->ts; g.td = ix->td; g.ts_q = w.ts_q; g.variant = ix->gemm_variant; g.i8_mul = (ix->dense_i8 || ix->gated_i8) ? w.i8_mul : nullptr; g.g8_shift = ix->gated_i8 ? w.g8_shift : nullptr; g.g8_rsum = ix->g8_rsum;
  g.seq_lo = lo; g.seq_hi = hi; g.map_mode = map_mode; g.period = period; g.head = head; g.n_tiles = ix->n_tiles;
  g.n_qtiles = w.q_pad / TILE_ROWS; g.n_rows = ix->n_rows; g.thr = w.thr; g.cand = w.cand; g.cnt = w.cnt;
  g.cap = (uint32_t)w.cap; g.n_queries = Q;
  HIP_TRY(hipMemsetAsync(w.cnt, 0, (size_t)w.q_pad * 4, s));
  tm.begin(T_GEMM); HIP_TRY(launch_gemm_filter(g, s)); tm.end();      // (list statistics + overflow marks: rescore_select_async, one launch)
  ix->last_gemm_kernel = g_last_gemm_kernel;
  const double rows = (double)(hi - lo) * TILE_ROWS;
  st.phases++;
  st.gemm_rows += (int64_t)rows;
  st.gemm_flops += 2.0 * (double)w.q_pad * rows * (double)ix->kt;
  st.gemm_flops_alg += 2.0 * (double)Q * rows * (double)ix->k;
  return DHR_OK;
}
// (d_fullest_bound / d_fullest: where the length of the fullest bound / survivor list of this phase is stored, or nullptr)
static int rescore_select_async(dhr_index* ix, Workspace& w, int Q, bool gate, SelectArgs& sel, const uint2* cand, const uint32_t* cnt,
                                const float* thr, Timer& tm, hipStream_t s, uint32_t* d_fullest_bound = nullptr, uint32_t* d_fullest = nullptr,
                                const uint2* ovf = nullptr) {
  uint32_t list_cap = (uint32_t)w.cap;
  const bool refine = uses_refine(ix, gate);
  if (!refine) ovf = nullptr;                    // (the second tier exists for lists a refine level reads, ensure_ws)
  const uint32_t* ovf_cap = ovf ? w.ovf_cap : nullptr;
  // the bound lists: statistics, overflow marks, block offsets of the kernel that walks them and (refine) the survivor counters cleared
  HIP_TRY(launch_lists_ready(cnt, (uint32_t)w.cap, Q, refine ? (uint32_t)REFINE_PER_WG : (uint32_t)RESCORE_CANDS_PER_WG, refine ? w.blk_off : w.blk_off + w.q_pad + 1,
                             d_fullest_bound, refine ? nullptr : d_fullest, w.d_stats + 0, refine ? nullptr : w.d_stats + 1, w.fail_flags,
                             refine ? w.cnt_r : nullptr, refine ? (int)w.q_pad : 0, s, ovf_cap));
  if (refine) {
    RefineArgs f{};
    f.ovf = ovf; f.ovf_off = ovf ? w.ovf_off : nullptr; f.ovf_cap = ovf_cap;
    f.cand = cand; f.cnt = cnt; f.cap = (uint32_t)w.cap; f.heavy_key = ix->heavy_key; f.heavy_val = ix->heavy_val;
    f.q_pack = w.q_pack; f.d_dlr = ix->d_dlr; f.thr = thr; f.out = w.cand_r; f.out_cnt = w.cnt_r; f.out_cap = (uint32_t)w.cap_r;
    f.n_queries = Q; f.max_count = 1;
    if (ix->gated_i8) { f.g8_q8 = w.g8_q8; f.g8_inv_cs = ix->g8_inv_cs; f.g8_unit = w.g8_unit; f.abs_mode = ix->abs_mode ? 1 : 0; f.ungated = gate ? 0 : 1; }
    if (ix->resid8) { f.resid8 = ix->resid8; f.resid_ld = ix->resid_ld; f.q32 = w.q32; f.q32_ld = ix->k_rm; f.col_scale = ix->i8_col_scale; f.d_cls = ix->d_cls; f.thr_raise = w.thr_raise; }
    f.blk_off = w.blk_off; f.flat_blocks = async_grid();
    tm.begin_on(T_REFINE, s); HIP_TRY(launch_refine(f, s)); tm.end_on(s);
    list_cap = (uint32_t)w.cap_r;
    cand = w.cand_r; cnt = w.cnt_r;
    HIP_TRY(launch_lists_ready(cnt, list_cap, Q, (uint32_t)RESCORE_CANDS_PER_WG, w.blk_off + w.q_pad + 1, d_fullest, nullptr, w.d_stats + 1,
                               nullptr, w.fail_flags, nullptr, 0, s));
  }
  RescoreArgs r = base_rescore_args(ix, w, Q, gate);
  r.cand = cand; r.cnt = cnt; r.cap = list_cap; r.max_count = 1;
  r.out_keys = w.rs_keys; r.ld_keys = w.keys_ld;
  r.blk_off = w.blk_off + w.q_pad + 1; r.flat_blocks = async_grid();
  tm.begin_on(T_RESCORE, s); HIP_TRY(launch_rescore(r, s)); tm.end_on(s);
  sel.cnt = cnt; sel.count_all = 0; sel.cap = list_cap;
  tm.begin_on(T_SELECT, s); HIP_TRY(launch_select(sel, s)); tm.end_on(s);
  return DHR_OK;
}

// Candidates of one phase -> [refine on the heavy lists] -> exact rescoring -> top-k merge, all on `s`.
// The refine step needs one host read-back (size of the surviving lists) to size the rescoring grid.
static int rescore_select(dhr_index* ix, Workspace& w, int Q, bool gate, SelectArgs& sel, const uint2* cand,
                          const uint32_t* cnt, const float* thr, uint32_t maxc, Timer& tm, dhr_search_stats& st,
                          hipStream_t s, int64_t bound_sum, uint32_t* fail_flags) {
  uint32_t maxr = std::min<uint32_t>(maxc, (uint32_t)w.cap);
  if (maxr == 0) return DHR_OK;
  int64_t exact = bound_sum;
  uint32_t list_cap = (uint32_t)w.cap;
  if (uses_refine(ix, gate)) {
    RefineArgs f{};
    f.cand = cand; f.cnt = cnt; f.cap = (uint32_t)w.cap; f.heavy_key = ix->heavy_key; f.heavy_val = ix->heavy_val;
    f.q_pack = w.q_pack; f.d_dlr = ix->d_dlr; f.thr = thr; f.out = w.cand_r; f.out_cnt = w.cnt_r; f.out_cap = (uint32_t)w.cap_r;
    f.n_queries = Q; f.max_count = maxr;
    if (ix->gated_i8) { f.g8_q8 = w.g8_q8; f.g8_inv_cs = ix->g8_inv_cs; f.g8_unit = w.g8_unit; f.abs_mode = ix->abs_mode ? 1 : 0; f.ungated = gate ? 0 : 1; }
    if (ix->resid8) { f.resid8 = ix->resid8; f.resid_ld = ix->resid_ld; f.q32 = w.q32; f.q32_ld = ix->k_rm; f.col_scale = ix->i8_col_scale; f.d_cls = ix->d_cls; f.thr_raise = w.thr_raise; }
    // flat launch: one workgroup per REAL block of 256 candidates (bound_sum / 256 + Q is an upper bound of their number)
    HIP_TRY(launch_block_offsets(cnt, (uint32_t)w.cap, Q, REFINE_PER_WG, w.blk_off, s));
    f.blk_off = w.blk_off; f.flat_blocks = (uint32_t)std::min<int64_t>(bound_sum / REFINE_PER_WG + Q, (int64_t)0x7fffffff);
    HIP_TRY(hipMemsetAsync(w.cnt_r, 0, (size_t)w.q_pad * 4, s));
    HIP_TRY(hipMemsetAsync(w.d_ref, 0, 16, s));
    tm.begin_on(T_REFINE, s); HIP_TRY(launch_refine(f, s)); tm.end_on(s);
    HIP_TRY(launch_max_u32(w.cnt_r, Q, w.d_ref, (unsigned long long*)(w.d_ref + 2), s));
    HIP_TRY(hipMemcpyAsync(w.h_ref, w.d_ref, 16, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    w.last_maxr = ((uint32_t*)w.h_ref)[0];
    maxr = std::min<uint32_t>(((uint32_t*)w.h_ref)[0], (uint32_t)w.cap_r);
    if (((uint32_t*)w.h_ref)[0] > (uint32_t)w.cap_r) {          // survivors list overflowed: those queries are redone
      if (fail_flags) HIP_TRY(launch_mark_overflow(w.cnt_r, (uint32_t)w.cap_r, Q, fail_flags, s));
      else return 1;                                           // streaming controller: redo this chunk in halves
    }
    list_cap = (uint32_t)w.cap_r;
    unsigned long long sum;
    memcpy(&sum, (uint32_t*)w.h_ref + 2, 8);
    exact = (int64_t)sum;
    cand = w.cand_r; cnt = w.cnt_r;
  }
  st.candidates_exact += exact;
  if (getenv("DHR_DEBUG_PLAN"))
    fprintf(stderr, "[dhr]   lists: bound %.0f per query (fullest %u) -> exact %.0f per query (fullest %u)\n", (double)bound_sum / Q, maxc, (double)exact / Q, maxr);
  if (maxr == 0) return DHR_OK;
  RescoreArgs r = base_rescore_args(ix, w, Q, gate);
  r.cand = cand; r.cnt = cnt; r.cap = list_cap; r.max_count = maxr;
  r.out_keys = w.rs_keys; r.ld_keys = w.keys_ld;
  HIP_TRY(launch_block_offsets(cnt, list_cap, Q, RESCORE_CANDS_PER_WG, w.blk_off + w.q_pad + 1, s));
  r.blk_off = w.blk_off + w.q_pad + 1; r.flat_blocks = (uint32_t)std::min<int64_t>(exact / RESCORE_CANDS_PER_WG + Q, (int64_t)0x7fffffff);
  tm.begin_on(T_RESCORE, s); HIP_TRY(launch_rescore(r, s)); tm.end_on(s);
  sel.cnt = cnt; sel.count_all = 0; sel.cap = list_cap;
  tm.begin_on(T_SELECT, s); HIP_TRY(launch_select(sel, s)); tm.end_on(s);
  return DHR_OK;
}

// Streaming phases over a tile sequence with growing chunks (exact for any input: tau only ever
// comes from exact scores already seen, overflowing chunks are re-run in halves).
// Rank that defines the threshold of a sampled run after a fraction phi of the sample has been seen.  The run's goal is the r-th best
// score of the WHOLE sample; of the sample's r best rows a scattered fraction phi holds r phi +- sqrt(r phi (1 - phi)), so the
// (r phi + 6 sigma + 4)-th best seen lies below the sample's final r-th best except with negligible probability (Poisson tail < 1e-8 per
// check at every phi; a query for which it does not fails the final verification -- thresholds only ever rise, tau_hat is their maximum --
// and is redone).  With the fixed rank r of rounds 1-3 every phase of the sampled run let ~r x (rows of the phase / rows seen) x the
// bound's amplification through: 1 400 of the 4 340 exact rescorings per query of a config-3 step were spent finding the 64 best of
// the 1/32 sample.
int adaptive_rank(int r, double phi) {
  if (!(phi < 1.0)) return r;
  if (phi < 0.0) phi = 0.0;
  const double m = (double)r * phi;
  return std::max(1, std::min(r, (int)std::ceil(m + 6.0 * std::sqrt(m * (1.0 - phi)) + 4.0)));
}

static int stream_phases(dhr_index* ix, Workspace& w, int Q, bool gate, SelectArgs& sel, int64_t n_seq, int map_mode,
                         int period, int64_t head, int64_t first_chunk, int64_t seen_rows, Timer& tm, dhr_search_stats& st,
                         hipStream_t s, double* last_rate = nullptr, double* last_rate_r = nullptr, bool async_ctl = false,
                         int64_t* last_rows = nullptr, int rank_target = 0, int64_t rank_rows = 0, int64_t pos0 = 0, int growth16 = 0, int64_t prev_rows0 = 0) {
  if (growth16 <= 0) growth16 = ix->max_growth16;
  int64_t pos = pos0;          // (pos0 > 0: the run resumes behind a part another call streamed -- dhr_search_begin_rest)
  int64_t prev_rows = prev_rows0;      // rows of the phase whose list lengths w.cnt still holds (the second list tier of the next phase is planned from them)
  int64_t chunk = std::max<int64_t>(DOC_GROUP, first_chunk);
  while (pos < n_seq) {
    chunk = std::min(chunk, round_up(n_seq - pos, DOC_GROUP));
    const int64_t hi = std::min(pos + chunk, n_seq);
    if (rank_target > 0) {        // sampled run: the rank grows with the fraction of the sample seen once this phase is merged
      sel.k = adaptive_rank(rank_target, (double)(seen_rows + (hi - pos) * TILE_ROWS) / (double)std::max<int64_t>(rank_rows, 1));
      sel.monotone = 1;
    }
    if (async_ctl) {          // enqueue only: an overflowing list flags its query instead of halving the chunk
      // two-tier lists: the second tier of this phase from the list lengths of the previous one (w.cnt still holds them; none for the first)
      if (w.arena > 0)
        HIP_TRY(launch_plan_overflow(prev_rows > 0 ? w.cnt : nullptr, prev_rows > 0 ? (double)((hi - pos) * TILE_ROWS) / (double)prev_rows : 0.0, (uint32_t)w.cap,
                                     (uint32_t)(w.cap_deep - w.cap), (uint32_t)w.arena, Q, w.ovf_off, w.ovf_cap, s));
      // (d_max2: {fullest bound list, -, -, -, fullest survivor list} of the latest phase -- what the chunk plan of the main pass reads)
      int rc = gemm_phase_async(ix, w, Q, pos, hi, map_mode, period, head, tm, st, s);
      if (rc) return rc;
      if ((rc = rescore_select_async(ix, w, Q, gate, sel, w.cand, w.cnt, w.thr, tm, s, w.d_max2, w.d_max2 + 4, w.arena > 0 ? w.ovf : nullptr)) != DHR_OK) return rc;
      prev_rows = (hi - pos) * TILE_ROWS;
      if (last_rows) *last_rows = (hi - pos) * TILE_ROWS;
      seen_rows += (hi - pos) * TILE_ROWS;
      pos = hi;
      chunk = std::max<int64_t>(DOC_GROUP, round_up(seen_rows * growth16 / 16 / TILE_ROWS, DOC_GROUP));
      continue;
    }
    uint32_t maxc; unsigned long long sumc;
    int rc = gemm_phase(ix, w, Q, pos, hi, map_mode, period, head, tm, st, s, &maxc, &sumc);
    if (rc) return rc;
    if (getenv("DHR_DEBUG_PLAN")) fprintf(stderr, "[dhr] stream phase: tiles [%lld, %lld) of %lld (period %d)\n", (long long)pos, (long long)hi, (long long)n_seq, period);
    const int64_t chunk_rows = (hi - pos) * TILE_ROWS;
    if (maxc > w.cap && chunk > DOC_GROUP) {               // overflow: redo this chunk in halves
      st.overflow_retries++;
      chunk = std::max<int64_t>(DOC_GROUP, round_up(chunk / 2, DOC_GROUP));
      continue;
    }
    // (cannot happen: every list depth the workspace plans holds the rows of a minimum chunk -- ensure_ws, DHR_PARAM_CAND_CAP / _LIST_STRIDE >= 1024)
    if (maxc > w.cap) return set_error(DHR_ERR_INTERNAL, "bound list overflow at the minimum chunk size");
    st.candidates_bound += (int64_t)sumc;
    if (last_rate) *last_rate = (double)maxc / (double)chunk_rows;      // fullest list per corpus row, at the latest thresholds
    rc = rescore_select(ix, w, Q, gate, sel, w.cand, w.cnt, w.thr, maxc, tm, st, s, (int64_t)sumc, nullptr);
    if (rc == 1 && chunk > DOC_GROUP) {                    // survivor lists overflowed: same cure as a bound-list overflow
      st.overflow_retries++;
      chunk = std::max<int64_t>(DOC_GROUP, round_up(chunk / 2, DOC_GROUP));
      continue;
    }
    if (rc != DHR_OK) return rc < 0 ? rc : set_error(DHR_ERR_INTERNAL, "survivor list overflow at the minimum chunk size");
    if (last_rate_r) *last_rate_r = (uses_refine(ix, gate) && maxc > 0) ? (double)w.last_maxr / (double)chunk_rows : 0.0;   // fullest SURVIVOR list per corpus row
    pos = hi;
    seen_rows += chunk_rows;
    // next chunk: aim at cap/4 candidates for the fullest query, never more than growth * rows seen
    const double target = (double)w.cap / 2.0;
    double next_rows = (maxc == 0) ? (double)chunk_rows * 4.0 : (double)chunk_rows * target / (double)maxc;
    next_rows = std::min(next_rows, (double)seen_rows * ix->max_growth16 / 16.0);
    chunk = std::max<int64_t>(DOC_GROUP, (int64_t)(next_rows / (DOC_GROUP * TILE_ROWS)) * DOC_GROUP);
  }
  return DHR_OK;
}

// Leaves the sorted top-k keys of every query in w.topk_keys.  qb must already be validated.
// stage 3 (dhr_search_mid): the first slice of the main pass with the caller's thresholds, then stop -- dhr_search_finish resumes behind it with
// the thresholds the shards agree on from what they have seen by then.
// stage 0: whole search.  stage 1 (dhr_search_begin): stop after the sampled run.  stage 2 (dhr_search_finish):
// resume at the main pass with the caller's thresholds tau_ext (device [Q]); no local verification.
// Conservative rank of the sampled threshold: the k/S top rows a 1/S sample holds on average + 5 sigma + 4 (4 sigma until round 2:
// one query in ~50 000 then saw its sample hold 58 rows above a score that fewer than k rows of the corpus reach, and a failed
// query costs extra passes over the corpus for its whole query tile; the extrapolated thresholds make the looser start cheap).
static int sample_rank_of(double mean) { return (int)std::ceil(mean + 5.0 * std::sqrt(mean) + 4.0); }

// Sample period and rank for this index and k: the configured period, halved (32 -> 16 -> 8 -> 4) while the corpus is too small for it
// -- the sample must span >= 32 tiles and hold >= 16 r rows, and r must stay below k; S = 0 (r = k): no sampling, plain streaming.
// (Until round 2 a corpus below ~270 k rows was never sampled: the 100 k-row config 1 rescored 8.7 k rows per query, most of them while
// the streaming thresholds were still warming up.)
void plan_sampling(const dhr_index* ix, int k, int& S, int& r) {
  for (S = ix->sample_period; S >= 2; S = S >= 8 ? S / 2 : 0) {
    const int rr = sample_rank_of((double)k / S);
    const int64_t head_guess = round_up(std::max<int64_t>(256, 2 * (int64_t)rr), TILE_ROWS) / TILE_ROWS;
    const int64_t rest_guess = ix->n_tiles - head_guess;
    if (!(rr >= k || rest_guess < 32 * (int64_t)S || (rest_guess / S) * TILE_ROWS < 16 * (int64_t)rr)) { r = rr; return; }
  }
  S = 0; r = k;
}

// Sharded search: the common threshold is the r-th best sample score of the UNION of the shards' samples.  A shard's share of those r
// scores is ~Binomial(r, 1 / shards), so it only has to report (and, during its sampled run, to chase) its
// r / shards + 5 sqrt(r / shards) + 4 best: 26 instead of 64 at 8 shards.  A longer share than that only makes the union's r-th best
// come out LOWER (still a valid threshold, the verification of the counts catches what it costs).
int local_sample_rank(const dhr_index* ix, int r) {
  if (ix->sample_share <= 1 || r <= 0) return r;
  const double m = (double)r / ix->sample_share;
  return std::min(r, (int)std::ceil(m + 5.0 * std::sqrt(m) + 4.0));
}

// Share of the main pass that dhr_search_mid runs before the shards agree on thresholds a second time, in 1/16ths (default 2 = 1/8: with the
// 1/32 sample the shards have then seen ~15 % of their rows)
// Share of the SAMPLE that dhr_search_pre streams before the shards agree on a first common threshold, in 1/16ths (default 2 = 1/8), and the
// sample positions that is (whole tile groups; 0: the sample is too small to split)
// (round 6 swept both shares on the emulated 8-shard step -- pre 2 / 4, mid 1 / 2 / 3 sixteenths, one phase or two for the first part, sample
// period 16: 18.1-18.6 ms against 18.3, inside the box-to-box noise (profiles/r06_shard_sim.txt) -- and made them constants)
static int pre_share16() { return 2; }
int64_t pre_positions(int64_t n_sample) {
  if (n_sample < 64) return 0;
  const int64_t a = round_up(std::max<int64_t>(DOC_GROUP, n_sample * pre_share16() / 16), DOC_GROUP);
  return a < n_sample ? a : 0;
}
int mid_share16() { return 2; }
// Rows scored exhaustively in phase 0 (>= the rank that defines tau, so that tau exists afterwards), whole tiles -- 512 rows until round 3, 256
// since; 0 = none: a sampled search (S >= 2) bootstraps its first thresholds from the bound GEMM instead (search_core), unless the caller
// fixed the head (DHR_PARAM_FIRST_ROWS).
int64_t head_rows(const dhr_index* ix, int S, int r_eff) {
  if (S >= 2 && ix->first_rows <= 0 && ix->n_tiles >= 8) {
    // the bootstrap takes the r0-th best of at most BOOT_M rows of tile 0: a sampling rule under which r0 outgrows that (today r0 <= 44) must not
    // publish the threshold of a lower rank -- it falls back to the exhaustive head instead (search_core computes r0 from the same expression)
    const int64_t sample_rows = ((ix->n_tiles + S - 1) / S) * TILE_ROWS;
    if (adaptive_rank(r_eff, (double)TILE_ROWS / (double)sample_rows) <= BOOT_M) return 0;
  }
  return ix->first_rows > 0 ? std::max<int64_t>(ix->first_rows, 2 * (int64_t)r_eff) : std::max<int64_t>(S >= 2 ? 256 : 512, 2 * (int64_t)r_eff);
}
int search_core(dhr_index* ix, Workspace& w, const dhr_query_batch* qb, int k, int depth, Timer& tm,
                       dhr_search_stats& st, hipStream_t s, int stage, const float* tau_ext) {
  int rc;
  // stage 4 (dhr_search_pre): query preparation, phase 0 and the FIRST part of the sampled run, then stop -- the shards exchange their best sample
  // scores seen so far; stage 5 (dhr_search_begin_rest): the rest of the sampled run from the threshold they agreed on, then as stage 1.
  const bool fresh = stage == 0 || stage == 1 || stage == 4;       // the call brings the query batch (else: resumed from ix->pend)
  const int Q = !fresh ? ix->pend.Q : qb->n_queries;
  const bool gate = !fresh ? ix->pend.gate
                           : (ix->d_dlr > 0 && qb->index != nullptr && qb->index_dtype != DHR_IDX_NONE);   // else plain IP
  const int64_t n = ix->n_rows;
  // depth 0: sampled thresholds; depth 1 (queries that failed at depth 0): the same with 16x list capacity;
  // depth 2: plain streaming, exact for any input
  const bool allow_sampling = depth < 2;
  // sampled threshold: period S, conservative rank r (DESIGN.md "controller")
  int S = 0, r_eff = k;
  if (allow_sampling) plan_sampling(ix, k, S, r_eff);
  if (S >= 2 && stage != 0) r_eff = local_sample_rank(ix, r_eff);          // staged (sharded) search: this shard's share of the union's rank
  // rows scored exhaustively in phase 0 (>= the rank that defines tau, so that tau exists afterwards), whole tile groups
  // (512 rows until round 3; with sampled thresholds the head only has to hold 2 r rows, and it is a fixed cost of every rank of the
  // sharded search: 256 rows x 6 980 queries are 1.1 ms of exhaustive rescoring)
  int64_t first = head_rows(ix, S, r_eff);
  // Round 5: a SAMPLED search seeds its thresholds without an exhaustive head (use_bootstrap: first == 0).  The head cost every search
  // 256 rows x all queries of exact rescoring, bound by instruction issue (0.9-1.1 ms; for a 1/8 shard a quarter of its sampled run), only
  // to know the ~6th best score of 256 rows.  Instead: ONE corpus tile through the bound GEMM with an open filter, the BOOT_M-or-fewer best
  // rows of every query BY BOUND rescored exactly, and the r0-th best of those exact scores is the first threshold -- a lower bound of the
  // r0-th best of the tile whatever the bound's ranking is worth (adaptive_rank's argument with phi = 256 rows of the sample).  The running
  // list is cleared again: tile 0 is the first tile of the sample and comes back through the ordinary filtered phases.
  const bool bootstrap = first == 0;
  first = std::min(round_up(first, TILE_ROWS), round_up(n, TILE_ROWS));
  const int64_t first_valid = std::min(first, n);
  if ((rc = ensure_ws(ix, w, Q, k, first_valid, depth == 0 ? 1 : 16, gate || ix->gated_i8 || ix->resid8 != nullptr)) != DHR_OK) return rc;

  // first attempt of a sampled search: the controller only enqueues (no host read-backs); DHR_PARAM_ASYNC_CONTROLLER 0 keeps the
  // host-driven controller (and the fallback depths always use it: it is the one that is exact for any input)
  const bool async_ctl = depth == 0 && S >= 2 && ix->async_ctl != 0 && !getenv("DHR_DEBUG_PLAN");
  const bool plan_read = ix->async_ctl >= 2;      // 2: the chunk plan of the main pass reads the sampled run's list lengths back
  if (fresh) {
    tm.begin(T_PREP);
    if ((rc = prep_queries(ix, w, qb, s)) != DHR_OK) return rc;
    HIP_TRY(hipMemsetAsync(w.topk_keys, 0, (size_t)w.q_pad * w.kp * 8, s));
    HIP_TRY(hipMemsetAsync(w.fail_flags, 0, (size_t)w.q_pad * 4, s));
    HIP_TRY(hipMemsetAsync(w.d_stats, 0, 32, s));
    tm.end();
  }

  const int64_t head = first / TILE_ROWS;                       // tiles scored exhaustively
  const int64_t rest = ix->n_tiles - head;

  SelectArgs sel{};
  sel.topk_keys = w.topk_keys; sel.in_keys = w.rs_keys; sel.ld_keys = w.keys_ld; sel.cap = (uint32_t)w.cap;
  sel.k = r_eff; sel.kp = w.kp; sel.sort_n = select_sort_n(w.kp);
  sel.k_keep = k;                          // the threshold is the r-th best seen, the list keeps the k best (ties with the final k-th score survive the sampled run)
  sel.kps = 64; while (sel.kps < k) sel.kps <<= 1; sel.margin = w.margin; sel.tau = w.tau; sel.thr = w.thr;
  sel.n_queries = Q;

  // ---- phase 0: threshold bootstrap (sampled searches), or exhaustive exact scoring of rows [0, first_valid)
  if (fresh && bootstrap) {
    const int64_t sample_rows = ((rest + S - 1) / S) * TILE_ROWS;
    const int r0 = adaptive_rank(r_eff, (double)TILE_ROWS / (double)sample_rows);
    const int m = std::min(BOOT_M, std::max(16, 2 * r0));
    {      // tile 0 through the bound GEMM's dump variant: [Q][256] bound scores, no lists
      GemmArgs g{};
      g.a_tiles = ix->tiles; g.b_tiles = w.q_tiles; g.ksteps = ix->ksteps; g.k_split = ix->d_dlr / TILE_K; g.ts = ix->ts; g.td = ix->td; g.ts_q = w.ts_q; g.variant = ix->gemm_variant; g.i8_mul = (ix->dense_i8 || ix->gated_i8) ? w.i8_mul : nullptr; g.g8_shift = ix->gated_i8 ? w.g8_shift : nullptr; g.g8_rsum = ix->g8_rsum;
      g.seq_lo = 0; g.seq_hi = 1; g.map_mode = 0; g.period = 1; g.head = 0; g.n_tiles = ix->n_tiles;
      g.n_qtiles = w.q_pad / TILE_ROWS; g.n_rows = ix->n_rows; g.thr = w.thr; g.cand = w.cand; g.cnt = w.cnt; g.cap = (uint32_t)w.cap; g.n_queries = Q;
      g.dump = w.boot_bound; g.dump_ld = TILE_ROWS; g.dump_row0 = 0;
      tm.begin(T_GEMM); HIP_TRY(launch_gemm_filter(g, s)); tm.end();
      st.phases++;
    }
    HIP_TRY(launch_bound_topm(w.boot_bound, (int)std::min<int64_t>(TILE_ROWS, n), Q, m, w.boot_rows, s));
    RescoreArgs r = base_rescore_args(ix, w, Q, gate);
    r.rows32 = w.boot_rows; r.ld_rows = m; r.count_all = (uint32_t)m; r.max_count = (uint32_t)m;
    r.out_keys = w.rs_keys; r.ld_keys = w.keys_ld;
    tm.begin(T_RESCORE); HIP_TRY(launch_rescore(r, s)); tm.end();
    sel.cnt = nullptr; sel.count_all = (uint32_t)m;
    sel.k = std::min(r0, m); sel.monotone = 1;
    tm.begin(T_SELECT); HIP_TRY(launch_select(sel, s)); tm.end();
    HIP_TRY(hipMemsetAsync(w.topk_keys, 0, (size_t)w.q_pad * w.kp * 8, s));       // thresholds stay (tau, thr); the rows come back with the sample
    st.candidates_exact += (int64_t)m * Q;
  } else if (fresh) {
    RescoreArgs r = base_rescore_args(ix, w, Q, gate);
    r.row0 = 0; r.count_all = (uint32_t)first_valid; r.max_count = (uint32_t)first_valid;
    r.out_keys = w.rs_keys; r.ld_keys = w.keys_ld;
    tm.begin(T_RESCORE); HIP_TRY(launch_rescore(r, s)); tm.end();
    sel.cnt = nullptr; sel.count_all = (uint32_t)first_valid;
    if (S >= 2 && rest > 0) {       // the head is the first part of the sample (adaptive_rank)
      const int64_t sample_rows = first_valid + ((rest + S - 1) / S) * TILE_ROWS;
      sel.k = adaptive_rank(r_eff, (double)first_valid / (double)sample_rows);
      sel.monotone = 1;
    }
    tm.begin(T_SELECT); HIP_TRY(launch_select(sel, s)); tm.end();
    st.candidates_exact += (int64_t)first_valid * Q;
  }
  if (!fresh && ix->pend.done) return DHR_OK;              // the begin call already finished the search
  if (rest <= 0 || S < 2) {
    // plain streaming over all remaining tiles
    if (rest > 0 && (rc = stream_phases(ix, w, Q, gate, sel, rest, 1, 1, head, head, first_valid, tm, st, s)) != DHR_OK) return rc;
    if (stage == 1 || stage == 4) { ix->pend.valid = true; ix->pend.done = true; ix->pend.pre = false; ix->pend.gate = gate; ix->pend.Q = Q; ix->pend.k = k; }
    return DHR_OK;
  }

  // ---- sampled run: top-r_eff of {head rows} + {every S-th tile}  ->  tau_hat
  const int64_t n_sample = (rest + S - 1) / S;
  double rate = 0.0, rate_r = 0.0;
  if (fresh || stage == 5) {
    int64_t last_rows = 0;
    int64_t pos0 = 0, seen0 = first_valid, n_hi = n_sample, chunk0 = std::max<int64_t>(head, DOC_GROUP);
    if (stage == 4) {
      n_hi = pre_positions(n_sample);
      if (n_hi <= 0) return set_error(DHR_ERR_INVALID, "the sample of this index is too small for a pre step");
    }
    if (stage == 5) {
      // the threshold the shards agreed on after the first part (never below this shard's own: thresholds only rise), and on with the growth rule
      pos0 = ix->pend.pre_pos; seen0 = ix->pend.pre_seen;
      HIP_TRY(launch_raise_thr(w.tau, tau_ext, Q, s));
      HIP_TRY(launch_make_thr(w.tau, w.margin, Q, w.q_pad, w.thr, s));
      // ONE phase for the rest: the agreed threshold is the union's (r phi + 6 sigma + 4)-th best of 1/8 of
      // the union sample -- 8 x the rows this shard has seen -- and a phase of a shard's sampled run is bound by its launches, not by its rows
      chunk0 = round_up(n_sample - pos0, DOC_GROUP);
    }
    // (the first part of a shard's sample is 17 tiles of a 1/8 shard of the benchmark: phases of 4 + 13 tiles instead of 4 + 8 + 5 -- a phase there
    // is bound by its ~8 dependent launches, 0.4-0.5 ms, not by its rows: growth 4 x there)
    constexpr int pre_growth = 64;
    if ((rc = stream_phases(ix, w, Q, gate, sel, n_hi, 1, S, head, chunk0, seen0, tm, st, s, &rate, &rate_r, async_ctl, &last_rows,
                            r_eff, first_valid + n_sample * TILE_ROWS, pos0, stage == 4 ? std::max(ix->max_growth16, pre_growth) : 0,
                            stage == 5 ? ix->pend.pre_last_rows : 0)) != DHR_OK) return rc;      // (stage 5: w.cnt still holds the lists of the pre call's last phase)
    if (stage == 4) {
      ix->pend.valid = true; ix->pend.done = false; ix->pend.mid = false; ix->pend.pre = true; ix->pend.gate = gate; ix->pend.Q = Q; ix->pend.k = k;
      ix->pend.pre_pos = n_hi; ix->pend.pre_seen = first_valid + n_hi * TILE_ROWS; ix->pend.pre_last_rows = last_rows;
      return DHR_OK;
    }
    HIP_TRY(hipMemcpyAsync(w.tau_hat, w.tau, (size_t)w.q_pad * 4, hipMemcpyDeviceToDevice, s));
    if (async_ctl && w.arena > 0) {          // the main pass plans its second list tier from these (possibly in a later call: staged search)
      HIP_TRY(hipMemcpyAsync(w.cnt_plan, w.cnt, (size_t)w.q_pad * 4, hipMemcpyDeviceToDevice, s));
      w.plan_rows = last_rows;
    } else w.plan_rows = 0;                  // nothing to plan from: a main pass that finds the arena in use (a controller switched between the calls) plans no segments
    if (async_ctl && plan_read && stage == 0 && last_rows > 0) {
      // the ONE read-back besides the final one: 32 bytes, the fullest bound / survivor list of the last sampled phase -> how many chunks
      // the main pass needs for the hottest query's lists to fit (a list that overflows costs its query tile an extra pass over the corpus:
      // on the 5 M-row BEIR corpora 4 of 7 405 queries per step did, 85.5 ms instead of 77.7)
      HIP_TRY(hipMemcpyAsync(w.h_pinned2, w.d_max2, 32, hipMemcpyDeviceToHost, s));
      HIP_TRY(hipStreamSynchronize(s));
      const uint32_t* hp = (const uint32_t*)w.h_pinned2;
      rate = (double)hp[0] / (double)last_rows;
      rate_r = uses_refine(ix, gate) ? (double)hp[4] / (double)last_rows : 0.0;
    }
    if (stage == 1 || stage == 5) {
      ix->pend.valid = true; ix->pend.done = false; ix->pend.mid = false; ix->pend.pre = false; ix->pend.gate = gate; ix->pend.Q = Q; ix->pend.k = k; ix->pend.rate = rate; ix->pend.rate_r = rate_r;
      return DHR_OK;
    }
  } else {
    rate = ix->pend.rate; rate_r = ix->pend.rate_r;
    if (stage == 2 && ix->pend.mid) {
      // second agreement (after dhr_search_mid): thresholds only ever rise
      HIP_TRY(launch_raise_thr(w.tau_hat, tau_ext, Q, s));
      HIP_TRY(launch_make_thr(w.tau_hat, w.margin, Q, w.q_pad, w.thr, s));
      HIP_TRY(launch_raise_thr(w.thr_hat, w.thr, Q, s));
    } else {
      // thresholds agreed between the shards: tau_ext >= this shard's own tau_hat in general
      HIP_TRY(hipMemcpyAsync(w.tau_hat, tau_ext, (size_t)Q * 4, hipMemcpyDeviceToDevice, s));
      HIP_TRY(launch_flag_tau_above(w.tau, w.tau_hat, Q, w.fail_flags, s));      // sample rows this shard dropped below its own (higher) threshold
      HIP_TRY(launch_make_thr(w.tau_hat, w.margin, Q, w.q_pad, w.thr, s));
    }
  }

  // ---- main pass: all other tiles with the FROZEN threshold tau_hat - margin, in a few chunks; the
  // bound GEMM of chunk i+1 (stream s) overlaps the exact rescoring + top-k merge of chunk i (aux stream)
  sel.k = k; sel.kps = w.kp; sel.monotone = 0;
  const int64_t n_main = rest - n_sample;
  // progressive_thr 2 (default, first attempt of an unsharded search only): the main pass visits the non-sample tiles in a scattered
  // order (i -> i * perm_mul mod n_main, perm_mul ~ 0.618 n_main and coprime), so that what has been seen after any chunk is a
  // scattered fraction of the corpus whatever the order of the rows, and the thresholds are extrapolated from it (raise_thr_rank_kernel)
  const bool extrapolate = ix->progressive_thr >= 2 && stage == 0 && depth == 0 && n_main >= 64 && k >= 16;
  const bool mid_proto = stage == 3 || (stage == 2 && ix->pend.mid);       // the shards agree a second time after a first slice: it must be a scattered one
  const bool scatter = extrapolate || (mid_proto && n_main >= 64);
  int64_t perm_mul = 1;
  if (scatter) {
    perm_mul = (int64_t)(0.6180339887 * (double)n_main) | 1;
    auto gcd = [](int64_t a, int64_t b) { while (b) { const int64_t t = a % b; a = b; b = t; } return a; };
    while (gcd(perm_mul, n_main) != 1) perm_mul += 2;
  }
  {
    if (!(stage == 2 && ix->pend.mid)) HIP_TRY(hipMemcpyAsync(w.thr_hat, w.thr, (size_t)w.q_pad * 4, hipMemcpyDeviceToDevice, s));
    if (!w.cand2) {
      int64_t tot = 0;
      HIP_TRY(re_malloc(w.cand2, (size_t)w.q_alloc * w.cap * 8, tot));
      HIP_TRY(re_malloc(w.cnt2, (size_t)w.q_alloc * 4, tot));
      if (w.arena > 0) {
        HIP_TRY(re_malloc(w.ovf2, (size_t)w.arena * 8, tot));
        const ListTier t1{w.ovf2, w.ovf_off, w.ovf_cap};
        HIP_TRY(hipMemcpy(w.tier_dev + 1, &t1, sizeof t1, hipMemcpyHostToDevice));
      }
      w.bytes += tot;
    }
    // Streams of the main pass.  Default: the bound GEMM on the caller's stream, refine/rescoring/select on a
    // non-blocking aux stream.  With aux_cus = N the aux stream is confined to N CUs (the low N bits of the CU
    // mask are spread evenly over the 8 XCDs) so that the memory-bound aux kernels take only the CUs they need
    // from the GEMM; gemm_exclusive additionally keeps the GEMM (on an internal stream) off those CUs.
    // (measured on a 1/8 shard of config 4, round 3: unmasked 17.0 ms per finish, 128 CUs 18.2, no overlap 18.1; dense-only indexes
    // measured best with 128 CUs in round 1)
    const int aux_cus = ix->aux_cus >= 0 ? ix->aux_cus : (ix->d_dlr == 0 ? 128 : 0);
    if (ix->aux_cus_made != aux_cus || ix->gemm_excl_made != ix->gemm_exclusive) {
      if (ix->s_aux) { hipStreamDestroy(ix->s_aux); ix->s_aux = nullptr; }
      if (ix->s_gemm) { hipStreamDestroy(ix->s_gemm); ix->s_gemm = nullptr; }
      if (aux_cus > 0) {
        uint32_t m_aux[8], m_gemm[8];
        for (int i = 0; i < 8; ++i) {
          const int lo = i * 32;
          const int n = std::max(0, std::min(32, aux_cus - lo));
          m_aux[i] = n >= 32 ? 0xffffffffu : ((1u << n) - 1u);
          m_gemm[i] = ix->gemm_exclusive ? ~m_aux[i] : 0xffffffffu;
        }
        // (a runtime that refuses the mask -- other CU count, masking disabled -- just gets the unmasked streams)
        if (hipExtStreamCreateWithCUMask(&ix->s_aux, 8, m_aux) != hipSuccess ||
            hipExtStreamCreateWithCUMask(&ix->s_gemm, 8, m_gemm) != hipSuccess) {
          (void)hipGetLastError();
          if (ix->s_aux) { hipStreamDestroy(ix->s_aux); ix->s_aux = nullptr; }
          if (ix->s_gemm) { hipStreamDestroy(ix->s_gemm); ix->s_gemm = nullptr; }
        }
      }
      if (!ix->s_aux) {
        // (round 5 measured the aux stream at the lowest queue priority: no gain -- a CU between two GEMM workgroups fills with gather waves whatever the priority)
        HIP_TRY(hipStreamCreateWithFlags(&ix->s_aux, hipStreamNonBlocking));
      }
      ix->aux_cus_made = aux_cus; ix->gemm_excl_made = ix->gemm_exclusive;
    }
    hipStream_t sg = ix->s_gemm ? ix->s_gemm : s;
    // default: dense-only indexes, and the main pass of a SHARD (staged search: its refine / rescoring / select are a larger share of a
    // shorter step -- 18.1 -> 17.0 ms per finish on a 1/8 shard of config 4; the unsharded gated search gains 2 % and its GEMM launches
    // would be timed under contention, so it keeps them serial)
    // Round 4: the default everywhere.  The unsharded gated search gains 2 ms per config-3 step (124.5 vs 126.2 ms; 1.5 % in round 3); its GEMM
    // launches then share CUs and the memory system with the gathers (92 -> 116 ms of launch durations per step), so the kernel's own rate
    // is profiled with DHR_PARAM_OVERLAP_AUX = 0 (bench.py reports both).
    const bool overlap = ix->overlap_aux < 0 ? true : ix->overlap_aux != 0;
    hipStream_t sb = overlap ? ix->s_aux : sg;
    Events evs;                 // every event of the pass (destroyed on every way out)
    // chunk count: at least main_chunks, more when the sampled run predicts that the fullest list would not fit
    // (rate = bound candidates per corpus row of the fullest query at the final sample thresholds, 1.5x headroom)
    // ... and the same for the survivor lists of the refine step, which are shallower (cap_r): a query whose bound the heavy lists
    // do not tighten fills them first
    // (two-tier lists: a hot query's list may grow to cap_deep; the host-driven controller only has the uniform stride)
    const bool two_tier = async_ctl && w.arena > 0;
    const int64_t plan_cap = two_tier ? w.cap_deep : w.cap;
    const int64_t need = std::max((int64_t)std::ceil(1.5 * rate * (double)n_main * TILE_ROWS / (double)plan_cap),
                                  (int64_t)std::ceil(1.5 * rate_r * (double)n_main * TILE_ROWS / (double)w.cap_r));
    // (without read-backs the sampled run's rates are not known here: a fixed 8 chunks (12 where the thresholds are extrapolated, below), which the deep lists of round 3 cover at config 3 --
    // 203 k entries in the fullest list of the first chunk against 262 144 slots; a list that overflows anyway flags its query)
    // ... scaled with the shard: one chunk per ~4 200 tiles, 2 to 8 (a 1/8 shard of config 4: 2 chunks; 8 cost it 7 ms of launches)
    // ... and, where that takes at most 12 chunks, so many that the FIRST (largest: 3 / (2 M) of the pass) chunk has no more rows than a list has
    // slots: a small corpus then cannot overflow a list whatever its scores are (queries with fewer than k matching rows filter at 0)
    int64_t by_size = std::min<int64_t>(8, std::max<int64_t>(2, (n_main + 4199) / 4200));
    const int64_t no_overflow = (3 * n_main * TILE_ROWS + 2 * std::min(plan_cap, w.cap_r) - 1) / (2 * std::min(plan_cap, w.cap_r));
    if (no_overflow <= 12) by_size = std::max(by_size, no_overflow);
    // ... and so many that the HOTTEST queries fit: on the benchmark's data a query passes ~20 k rows per 1 000 results through the bound
    // filter and ~4 k through the refine step, the hottest ten times that, whatever the corpus size -- on a 0.5 M-row corpus (BEIR quora,
    // 10 000 queries) that is a fifth of the rows of a chunk, and with 2 chunks 195 queries per step overflowed their 65 536-entry lists and
    // were redone (60 ms per step instead of 28).  The first chunk is 3 / (2 M) of the pass.  (A shard chases its share of k.)
    {
      const double k_eff = (double)k / (double)std::max(1, stage >= 2 ? ix->sample_share : 1);
      const int64_t by_hot = (int64_t)std::ceil(300.0 * k_eff / (double)plan_cap);
      const int64_t by_hot_r = (int64_t)std::ceil(65.0 * k_eff / (double)w.cap_r);
      by_size = std::max(by_size, std::min<int64_t>(24, std::max(by_hot, by_hot_r)));
    }
    // ... and, where the thresholds are extrapolated between the chunks (unsharded search), one chunk per ~2 800 tiles up to 12: every chunk
    // boundary is a chance to raise them.  Config 3: 8 -> 12 chunks rescores 3.29 k instead of 3.37 k rows per query, -0.4 ms; config 2 (whose
    // refine level is the larger share of its step): 78.0 -> 75.5 ms.  16 measure the same, 24 / 32 / 48 lose it again to launch boundaries
    // (config 3: 121.9 / 123.2 / 126.6 ms against 121.2-121.6 at 16 and 121.9 at 8 on one box).
    if (extrapolate) by_size = std::max(by_size, std::min<int64_t>(12, (n_main + 2799) / 2800));
    const int64_t want = async_ctl ? std::max<int64_t>(std::max<int64_t>(ix->main_chunks, by_size), (plan_read && stage == 0) ? need : 0) : std::max<int64_t>(ix->main_chunks, need);
    // (k > 4096: every chunk boundary costs a merge of the 16 384-slot running list of every query, 1.0-2.3 ms whatever the chunk brought; capping
    // the plan at 4 / 6 / 8 chunks there was measured in round 5 -- 242 -> 250-275 ms for --theta 0.3 --rerank with agip_topk 10 000: the
    // lists of the hottest queries overflow and their queries are redone.  The plan stays.)
    const int M_plain = (int)std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>(want, 64), n_main / (16 * DOC_GROUP)));
    // mid protocol: chunk 0 is the slice dhr_search_mid runs (mid_share16 / 16 of the pass), the plain plan covers the rest
    const int M = mid_proto ? M_plain + 1 : M_plain;
    if (getenv("DHR_DEBUG_PLAN"))
      fprintf(stderr, "[dhr] main pass: rate %.3e (x n_main rows = %.0f of cap %lld), rate_r %.3e (%.0f of cap_r %lld), need %lld, chunks %d, n_main %lld tiles\n", rate,
              rate * (double)n_main * TILE_ROWS, (long long)w.cap, rate_r, rate_r * (double)n_main * TILE_ROWS, (long long)w.cap_r, (long long)need, M, (long long)n_main);
    // chunk i covers [bound[i], bound[i+1]): sizes fall off linearly (weights M, M-1, ..., 1 on top of an equal
    // share) so that the refine/rescoring tail that cannot overlap a GEMM (the last chunk's) is short
    std::vector<int64_t> bound(M + 1, 0);
    {
      const int first = mid_proto ? 1 : 0;
      const int64_t off = mid_proto ? std::min<int64_t>(n_main, round_up(n_main * mid_share16() / 16, DOC_GROUP)) : 0;
      bound[first] = off;
      double acc = 0.0, tot = 0.0;
      for (int i = 0; i < M_plain; ++i) tot += 1.0 + 2.0 * (M_plain - 1 - i) / std::max(1, M_plain - 1);
      for (int i = 0; i < M_plain; ++i) {
        acc += 1.0 + 2.0 * (M_plain - 1 - i) / std::max(1, M_plain - 1);
        bound[first + i + 1] = std::min<int64_t>(n_main, off + round_up((int64_t)((n_main - off) * acc / tot), DOC_GROUP));
      }
      bound[M] = n_main;
    }
    const int c_lo = (stage == 2 && ix->pend.mid) ? 1 : 0, c_hi = stage == 3 ? 1 : M;      // the chunks THIS call runs
    if (two_tier) {
      // second tier of the bound lists, ONE plan for every chunk of the pass (both list sets share it): from the lists of the last sampled
      // phase, scaled to the largest chunk -- the pass filters with thresholds at least as high as that phase did, and they only rise
      int64_t big = 0;
      for (int i = 0; i < M; ++i) big = std::max(big, bound[i + 1] - bound[i]);
      HIP_TRY(launch_plan_overflow(w.plan_rows > 0 ? w.cnt_plan : nullptr, w.plan_rows > 0 ? (double)(big * TILE_ROWS) / (double)w.plan_rows : 0.0, (uint32_t)w.cap,
                                   (uint32_t)(w.cap_deep - w.cap), (uint32_t)w.arena, Q, w.ovf_off, w.ovf_cap, s));
    }
    // The GEMM / aux streams enter the pass behind everything the caller's stream holds so far -- INCLUDING the plan above: with CU masks
    // (DHR_PARAM_AUX_CUS / GEMM_EXCLUSIVE) the bound GEMM runs on s_gemm and refine on s_aux, and until round 5 they waited on an event
    // recorded BEFORE the plan kernel, so a GEMM that spilled past `cap` could pair a new ovf_cap with an old ovf_off.
    if (sg != s || sb != s) {
      hipEvent_t ev_enter = nullptr;
      HIP_TRY(evs.add(&ev_enter, hipEventDisableTiming));
      HIP_TRY(hipEventRecord(ev_enter, s));
      if (sg != s) HIP_TRY(hipStreamWaitEvent(sg, ev_enter, 0));
      if (sb != s) HIP_TRY(hipStreamWaitEvent(sb, ev_enter, 0));
    }
    std::vector<hipEvent_t> ev_gemm(M), ev_done(M);
    for (int i = 0; i < M; ++i) {
      HIP_TRY(evs.add(&ev_gemm[i], hipEventDisableTiming));
      HIP_TRY(evs.add(&ev_done[i], hipEventDisableTiming));
    }
    uint32_t* h = (uint32_t*)w.h_pinned;          // 16 bytes per set: {max, pad, sum64}; two sets live in 32 bytes
    auto enqueue_gemm = [&](int i) -> int {
      uint2* cand = (i & 1) ? w.cand2 : w.cand;
      uint32_t* cnt = (i & 1) ? w.cnt2 : w.cnt;
      const int64_t lo = bound[i], hi = bound[i + 1];
      if (i >= c_lo + 2) HIP_TRY(hipStreamWaitEvent(sg, ev_done[i - 2], 0));      // list set is free again
      GemmArgs g{};
      g.a_tiles = ix->tiles; g.b_tiles = w.q_tiles; g.ksteps = ix->ksteps; g.k_split = ix->d_dlr / TILE_K; g.ts = ix->ts; g.td = ix->td; g.ts_q = w.ts_q; g.variant = ix->gemm_variant; g.i8_mul = (ix->dense_i8 || ix->gated_i8) ? w.i8_mul : nullptr; g.g8_shift = ix->gated_i8 ? w.g8_shift : nullptr; g.g8_rsum = ix->g8_rsum;
      g.seq_lo = lo; g.seq_hi = hi; g.map_mode = scatter ? 3 : 2; g.period = S; g.head = head; g.n_tiles = ix->n_tiles; g.perm_mul = perm_mul; g.perm_n = n_main;
      g.n_qtiles = w.q_pad / TILE_ROWS; g.n_rows = ix->n_rows; g.thr = w.thr_hat; g.cand = cand; g.cnt = cnt;
      g.cap = (uint32_t)w.cap; g.n_queries = Q;
      if (two_tier) g.tier = w.tier_dev + (i & 1);
      HIP_TRY(hipMemsetAsync(cnt, 0, (size_t)w.q_pad * 4, sg));
      if (!async_ctl) HIP_TRY(hipMemsetAsync(w.d_max2 + 4 * (i & 1), 0, 16, sg));
      tm.begin_on(T_GEMM, sg); HIP_TRY(launch_gemm_filter(g, sg)); tm.end_on(sg);
      ix->last_gemm_kernel = g_last_gemm_kernel;
      if (!async_ctl) {       // the host-driven controller sizes the per-candidate launches from the list lengths; the enqueue-only one leaves them on the device
        HIP_TRY(launch_max_u32(cnt, Q, w.d_max2 + 4 * (i & 1), (unsigned long long*)(w.d_max2 + 4 * (i & 1) + 2), sg));
        HIP_TRY(hipMemcpyAsync(w.h_pinned2 + 16 * (i & 1), w.d_max2 + 4 * (i & 1), 16, hipMemcpyDeviceToHost, sg));
      }
      HIP_TRY(hipEventRecord(ev_gemm[i], sg));
      const double rows = (double)(hi - lo) * TILE_ROWS;
      st.phases++;
      st.gemm_rows += (int64_t)rows;
      st.gemm_flops += 2.0 * (double)w.q_pad * rows * (double)ix->kt;
      st.gemm_flops_alg += 2.0 * (double)Q * rows * (double)ix->k;
      return DHR_OK;
    };
    (void)h;
    if ((rc = enqueue_gemm(c_lo)) != DHR_OK) return rc;
    for (int i = c_lo; i < c_hi; ++i) {
      if (i + 1 < c_hi && (rc = enqueue_gemm(i + 1)) != DHR_OK) return rc;
      if (async_ctl) {
        uint2* cand_a = (i & 1) ? w.cand2 : w.cand;
        uint32_t* cnt_a = (i & 1) ? w.cnt2 : w.cnt;
        if (sb != sg) HIP_TRY(hipStreamWaitEvent(sb, ev_gemm[i], 0));
        if ((rc = rescore_select_async(ix, w, Q, gate, sel, cand_a, cnt_a, w.thr_hat, tm, sb, nullptr, nullptr, two_tier ? ((i & 1) ? w.ovf2 : w.ovf) : nullptr)) != DHR_OK) return rc;
        if (ix->progressive_thr) HIP_TRY(launch_raise_thr(w.thr_hat, sel.thr, Q, sb));
        if (extrapolate && i + 1 < M) {
          const double f = (double)(head + n_sample + bound[i + 1]) / (double)ix->n_tiles;
          const int r = (int)std::ceil((double)k * f + 6.0 * std::sqrt((double)k * f * (1.0 - f)) + 4.0);
          if (r < k) HIP_TRY(launch_raise_thr_rank(w.thr_hat, w.tau_hat, w.topk_keys, w.kp, r, w.margin, Q, sb));
        }
        HIP_TRY(hipEventRecord(ev_done[i], sb));
        continue;
      }
      HIP_TRY(hipEventSynchronize(ev_gemm[i]));
      const uint32_t maxc = *(const uint32_t*)(w.h_pinned2 + 16 * (i & 1));
      unsigned long long sumc;
      memcpy(&sumc, w.h_pinned2 + 16 * (i & 1) + 8, 8);
      st.candidates_bound += (int64_t)sumc;
      uint2* cand = (i & 1) ? w.cand2 : w.cand;
      uint32_t* cnt = (i & 1) ? w.cnt2 : w.cnt;
      if (getenv("DHR_DEBUG_PLAN")) fprintf(stderr, "[dhr] main chunk %d: tiles [%lld, %lld)\n", i, (long long)bound[i], (long long)bound[i + 1]);
      HIP_TRY(launch_mark_overflow(cnt, (uint32_t)w.cap, Q, w.fail_flags, sb));
      if ((rc = rescore_select(ix, w, Q, gate, sel, cand, cnt, w.thr_hat, maxc, tm, st, sb, (int64_t)sumc, w.fail_flags)) != DHR_OK) return rc;
      if (ix->progressive_thr) HIP_TRY(launch_raise_thr(w.thr_hat, sel.thr, Q, sb));   // later chunks filter with the running exact thresholds
      if (extrapolate && i + 1 < M) {
        const double f = (double)(head + n_sample + bound[i + 1]) / (double)ix->n_tiles;
        const int r = (int)std::ceil((double)k * f + 6.0 * std::sqrt((double)k * f * (1.0 - f)) + 4.0);      // 6 sigma: a failure costs a whole extra pass for its query tile
        if (r < k) HIP_TRY(launch_raise_thr_rank(w.thr_hat, w.tau_hat, w.topk_keys, w.kp, r, w.margin, Q, sb));
      }
      HIP_TRY(hipEventRecord(ev_done[i], sb));
    }
    HIP_TRY(hipStreamWaitEvent(s, ev_done[c_hi - 1], 0));
  }
  if (stage == 3) ix->pend.mid = true;
  if (stage >= 2) return DHR_OK;                                // the caller verifies across shards
  // ---- verify; queries whose list overflowed or that found < k rows above tau_hat are redone exactly
  if (getenv("DHR_DEBUG_FAIL")) {        // diagnostics: which queries are about to be redone, and why
    std::vector<uint32_t> ff(Q); std::vector<float> th(Q);
    (void)hipStreamSynchronize(s);
    (void)hipMemcpy(ff.data(), w.fail_flags, (size_t)Q * 4, hipMemcpyDeviceToHost);
    (void)hipMemcpy(th.data(), w.tau_hat, (size_t)Q * 4, hipMemcpyDeviceToHost);
    for (int q = 0; q < Q; ++q) {
      uint64_t key = 0;
      (void)hipMemcpy(&key, w.topk_keys + (size_t)q * w.kp + (k - 1), 8, hipMemcpyDeviceToHost);
      const float kth = key ? ordered_f32((uint32_t)(key >> 32)) : -INFINITY;
      if (ff[q] || !(kth >= th[q])) fprintf(stderr, "[dhr] depth %d query %d will be redone: list overflow %u, k-th best %.6f, threshold %.6f\n", depth, q, ff[q], kth, th[q]);
    }
  }
  HIP_TRY(hipMemsetAsync(w.d_max, 0, 16, s));
  HIP_TRY(launch_max_u32(w.fail_flags, Q, w.d_max + 1, (unsigned long long*)(w.d_max + 2), s));      // overflow marks so far (the verification adds its own below)
  HIP_TRY(launch_verify(w.topk_keys, w.kp, k, w.tau_hat, Q, w.fail_flags, w.d_max, s));
  HIP_TRY(hipMemcpyAsync(w.h_pinned, w.d_max, 16, hipMemcpyDeviceToHost, s));
  if (async_ctl) HIP_TRY(hipMemcpyAsync(w.h_stats, w.d_stats, 32, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));            // the ONE host read of a search whose first attempt succeeds: how many queries must be redone
  if (async_ctl) {
    st.candidates_bound += (int64_t)((unsigned long long*)w.h_stats)[0];
    st.candidates_exact += (int64_t)((unsigned long long*)w.h_stats)[1];
  }
  const uint32_t n_fail = ((uint32_t*)w.h_pinned)[0];
  const uint32_t n_overflow = ((uint32_t*)w.h_pinned)[1];
  st.sample_fallback_queries += n_fail;
  if (n_fail == 0) return DHR_OK;

  std::vector<uint32_t> flags(Q);
  HIP_TRY(hipMemcpy(flags.data(), w.fail_flags, (size_t)Q * 4, hipMemcpyDeviceToHost));
  std::vector<int32_t> ids;
  for (int q = 0; q < Q; ++q)
    if (flags[q]) ids.push_back(q);
  // depth 1 (the same sampling scheme with 16x deeper lists) only cures overflowed lists; a threshold that came out too high would
  // come out too high again from the same sample: those queries go straight to the plain streaming pass
  int next_depth = depth + 1;
  if (depth == 0 && (int)n_overflow == 0) next_depth = 2;
  const int nf = (int)ids.size();
  DevMem tmp_mem;
  void*& tmp = tmp_mem.p;
  const size_t b32 = (size_t)nf * ix->k_rm * 4, bidx = (size_t)nf * std::max(ix->d_dlr, 8) * 2, bids = (size_t)nf * 4;
  HIP_TRY(hipMalloc(&tmp, b32 + bidx + bids + 64));
  float* f32 = (float*)tmp;
  int16_t* fidx = (int16_t*)((char*)tmp + b32);
  int32_t* d_ids = (int32_t*)((char*)tmp + b32 + bidx);
  auto done = [&](int code) { return code; };       // (tmp_mem releases the scratch)
  if (hipMemcpyAsync(d_ids, ids.data(), bids, hipMemcpyHostToDevice, s) != hipSuccess) return done(set_error(DHR_ERR_HIP, "H2D failed"));
  if (launch_gather_queries(w.q32, w.q_idx, ix->k_rm, ix->d_dlr, d_ids, nf, f32, fidx, s) != hipSuccess)
    return done(set_error(DHR_ERR_HIP, "gather_queries launch failed"));
  dhr_query_batch sub{};
  sub.n_queries = nf; sub.mem_kind = ix->dlr_pad > 0 ? MEM_DEVICE_PADDED : DHR_MEM_DEVICE; sub.value = f32; sub.value_dtype = DHR_VAL_F32; sub.ld_value = ix->k_rm;
  sub.index = gate ? fidx : nullptr; sub.index_dtype = gate ? DHR_IDX_I16 : DHR_IDX_NONE; sub.ld_index = ix->d_dlr;
  Workspace& w2 = ix->ws_fb[next_depth - 1];
  if ((rc = search_core(ix, w2, &sub, k, next_depth, tm, st, s)) != DHR_OK) return done(rc);
  if (w2.kp != w.kp) return done(set_error(DHR_ERR_INTERNAL, "fallback workspace mismatch"));
  if (launch_scatter_keys(w2.topk_keys, w.topk_keys, w.kp, d_ids, nf, s) != hipSuccess)
    return done(set_error(DHR_ERR_HIP, "scatter_keys launch failed"));
  if (hipStreamSynchronize(s) != hipSuccess) return done(set_error(DHR_ERR_HIP, "fallback search failed on the device"));
  return done(DHR_OK);
}
