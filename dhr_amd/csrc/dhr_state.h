// State shared by the host-side translation units of the library (round 6 split of api.hip, 2 583 lines):
//   index_build.hip   dhr_index_create / destroy / parameters / info, the index file
//   search_core.hip   workspace, query preparation, the phase controller (search_core)
//   api.hip           the search entry points (dhr_search, dhr_search_rerank, the staged calls), dhr_score_rows, densify, PQ training, debug hooks, shard reduce
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "dhr_internal.h"
#include <hip/hip_ext.h>

using namespace dhr;

// the calling thread's error record lives in abi.cpp (a fixed buffer: recording a failure does not allocate)
inline int set_error(int code, const std::string& msg) { return dhr_set_error_message(code, msg.c_str()); }
#define HIP_TRY(expr)                                                                                          \
  do {                                                                                                         \
    hipError_t _e = (expr);                                                                                    \
    if (_e != hipSuccess)                                                                                      \
      return set_error(DHR_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e) + " (" __FILE__ ":" +    \
                                        std::to_string(__LINE__) + ")");                                       \
  } while (0)

static inline int64_t round_up(int64_t v, int64_t m) { return (v + m - 1) / m * m; }
// Scratch that must not outlive a call whichever way it ends -- an early `return set_error(...)`, or an exception on its way to the barrier
// of the entry point (abi_guard.h)
struct DevMem {
  void* p = nullptr;
  DevMem() = default;
  DevMem(const DevMem&) = delete;
  DevMem& operator=(const DevMem&) = delete;
  ~DevMem() { if (p) (void)hipFree(p); }
};
struct Events {
  std::vector<hipEvent_t> v;
  Events() = default;
  Events(const Events&) = delete;
  Events& operator=(const Events&) = delete;
  ~Events() { for (hipEvent_t e : v) if (e) (void)hipEventDestroy(e); }
  hipError_t add(hipEvent_t* out, unsigned flags = hipEventDefault) {
    v.reserve(v.size() + 1);                 // (grow first: an event that exists is always in the list)
    hipEvent_t e = nullptr;
    const hipError_t rc = hipEventCreateWithFlags(&e, flags);
    if (rc == hipSuccess) v.push_back(e);
    *out = e;
    return rc;
  }
};
static inline int idx_esize(int dt) { return dt == DHR_IDX_I16 ? 2 : 1; }

struct Workspace {
  int q_pad = 0, kp = 0;
  int q_alloc = 0;       // query rows the buffers were allocated for (>= q_pad: a smaller batch re-uses them)
  int64_t cap = 0;       // entries every query owns in the bound-candidate lists (cand, cand2): the stride of the uniform arrays
  int64_t cap_deep = 0;  // what ONE query's list may grow to: cap, or (two-tier lists) cap + its segment of the arena, at most this -- the depth the chunk plans assume
  int64_t arena = 0;     // two-tier lists: entries of each overflow arena (0: uniform lists only)
  uint2 *ovf = nullptr, *ovf2 = nullptr;          // the arenas of the two list sets
  uint32_t *ovf_off = nullptr, *ovf_cap = nullptr;   // [q_pad] a query's segment (plan_overflow_kernel); planned before every sampled phase and once for the main pass
  uint32_t* boot_rows = nullptr;                  // [q_pad][BOOT_M] rows of the threshold bootstrap (search_core phase 0)
  float* boot_bound = nullptr;                    // [q_pad][256] bound scores of corpus tile 0 (the GEMM's dump variant)
  bool keys_alias = false;                        // rs_keys IS cand_r (see ensure_ws)
  ListTier* tier_dev = nullptr;                   // device copy of {ovf, ovf_off, ovf_cap} and {ovf2, ovf_off, ovf_cap}: what GemmArgs::tier points at
  uint32_t* cnt_plan = nullptr;                   // [q_pad] bound-list lengths of the last sampled phase, kept for the plan of the main pass (a staged search resumes in another call)
  int64_t plan_rows = 0;                          // ... and the rows that phase covered
  uint32_t last_maxr = 0;   // fullest survivor list of the latest refine step (before clamping to cap_r): the chunk planner sizes the main pass by it too
  int64_t cap_r = 0;     // capacity of the lists that reach the exact rescoring (refine survivors; == cap without refine)
  int64_t keys_ld = 0, kt = 0, d_dlr = 0;
  int ts_q = 0;          // sparse stages of the current query operand (2:4 layout)
  __half* q_tiles = nullptr;
  float* q32 = nullptr;
  int16_t* q_idx = nullptr;
  __half* q16 = nullptr;                 // fp16 copy of the queries + index bytes + 'not fp16-representable' flag (fast rescoring path)
  uint8_t* q_idx8 = nullptr;
  uint32_t* q_inexact = nullptr;
  float *margin = nullptr, *tau = nullptr, *thr = nullptr;
  float* i8_mul = nullptr;               // dense_i8 indexes: per-query factor (corpus scale x query scale) of the int8 stages
  uint8_t* g8_q8 = nullptr;              // gated_i8 indexes: [q_pad][d_dlr] gated int8 operand values, [q_pad] shift, [q_pad] unit of a gated product
  int32_t* g8_shift = nullptr;
  float* g8_unit = nullptr;
  uint32_t* cnt = nullptr;
  uint2* cand = nullptr;
  uint64_t *rs_keys = nullptr, *topk_keys = nullptr;
  uint32_t* d_max = nullptr;             // {max, pad} + u64 sum live in one 16-byte device block
  float* tau_hat = nullptr;
  float* thr_hat = nullptr;              // frozen main-pass threshold (tau_hat - margin)
  uint32_t* fail_flags = nullptr;
  uint32_t* q_pack = nullptr;            // [q_pad][d_dlr] refine operand words
  float* thr_raise = nullptr;            // [q_pad] dense-only int8 index with a residual image: what its refine level adds to the filter threshold
  uint2* cand_r = nullptr;               // refine survivors
  uint32_t* cnt_r = nullptr;
  uint32_t* blk_off = nullptr;           // 2 x (q_pad + 1): block offsets of the flat refine / rescoring launches
  uint2* cand2 = nullptr;                // second candidate list set: chunk i+1's GEMM overlaps chunk i's rescoring
  uint32_t* cnt2 = nullptr;
  void* h_pinned = nullptr;              // 16 bytes pinned mirror
  char* h_pinned2 = nullptr;             // 2 x 16 bytes pinned (main-pass chunk statistics)
  uint32_t* d_max2 = nullptr;            // 2 x 16 bytes device
  unsigned long long* d_stats = nullptr; // 4 x u64 device: {bound candidates, exact rescorings, -, -} of a search whose controller runs without host read-backs
  void* h_stats = nullptr;               // pinned mirror
  uint32_t* d_ref = nullptr;             // 16 bytes device: refine survivors {max, pad, sum64}
  void* h_ref = nullptr;                 // pinned mirror
  void* q_stage = nullptr;  size_t q_stage_bytes = 0;
  void* qi_stage = nullptr; size_t qi_stage_bytes = 0;
  void* out_stage = nullptr; size_t out_stage_bytes = 0;
  int64_t bytes = 0;
};

struct dhr_index {
  int device = 0;
  int64_t n_rows = 0, n_tiles = 0, row_offset = 0;
  int d_dlr = 0, d_cls = 0, k = 0, idx_dtype = DHR_IDX_NONE;
  int dlr_pad = 0;     // zero slices appended to the caller's gated half so that d_dlr is a multiple of 8 (16-byte operand chunks): the caller's
                       // records are [d_dlr - dlr_pad gated | d_cls ungated] wide, the library's [d_dlr | d_cls]; a padded slice holds value 0 and index 0
                       // on both sides and adds 0 * 0 to every score (gip_retrieval.py:238 takes any --emb_dim)
  int k_rm = 0;        // row-major padded width (k rounded up to 64): q32 rows, vals_rm rows
  int n_buckets = 1;   // index buckets per gated slice in the bound operands
  int idx_buckets_req = 0;   // what the caller asked for (dhr_index_desc.idx_buckets), kept for dhr_index_save
  int kt = 0;          // operand-tile columns = n_buckets*d_dlr + d_cls rounded up to 64
  int ksteps = 0;      // kt / 64
  int ts = 0, td = 0;  // 2:4 sparse layout (two buckets): ts 32-slice stages + td dense stages; ts == 0 -> dense layouts
  __half* tiles = nullptr;
  __half* vals_rm = nullptr;
  void* c_idx = nullptr;
  uint8_t* bucket_map = nullptr;   // [d_dlr][256] for 8-bit index dtypes, else null (value % n_buckets)
  uint32_t* heavy_key = nullptr;   // [n_rows][HEAVY] refine lists (largest gated entries of every row), or null
  __half* heavy_val = nullptr;
  bool abs_mode = false;
  float dmax = 0.f;
  // dense_i8: the ungated stages of the bound operands are int8 images (64 columns per stage) -- scale of the corpus image, corpus-wide
  // maxima of ||d - scale*d8|| and ||scale*d8|| over the ungated part of a row (the filter margin pays for them, query_prep_kernel)
  bool dense_i8 = false;
  float i8_scale = 0.f, i8_ec = 0.f, i8_nc = 0.f;
  uint8_t* resid8 = nullptr;               // dense-only int8 index: [n_rows][resid_ld] residual image (what the int8 image lost, four bits per value in 1/14 steps, + 8): the refine level
  int resid_ld = 0;
  float resid_ec2 = 0.f;                   //   >= the norm of what the residual image itself loses (weighted space of i8_ec)
  float* i8_col_scale = nullptr;           // [d_cls] int8 step of every ungated column (its largest |value| / 127): outlier columns do not cost the others their resolution
  // gated_i8: the gated stages are int8 2:4 images too (gemm_g8.hip): column j in units of its own step, rounded up; the query
  // side carries w_j = step_j / g8_sref as a weight (query_prep_kernel)
  bool gated_i8 = false;
  float g8_sref = 0.f;
  int g8_max_shift = 0;
  float* g8_inv_cs = nullptr;              // [d_dlr] 1 / step_j (with 1e-6 of head room)
  float* g8_w = nullptr;                   // [d_dlr] step_j / g8_sref (rounded up)
  int32_t* g8_rsum = nullptr;              // [n_tiles * 256] 128 x sum of the row's gated int8 values (accumulator start of gemm_g8.hip)
  int64_t index_bytes = 0;
  // params
  int64_t cand_cap = 0, first_rows = 0;   // 0 = default (262144 with refine lists, else 65536)
  int64_t list_stride = 0;                // DHR_PARAM_LIST_STRIDE (0 = 32768)
  int profile = 0, max_growth16 = 32;
  int sample_period = 32;
  int async_ctl = 2;                       // (2: + one 32-byte read after the sampled run for the chunk plan of the main pass) first attempt of a sampled search: the controller only ENQUEUES (no host read-backs between the phases; list
                                           // overflows are flagged on the device and cured by the fallback); 0 = the host-driven controller of rounds 1-2
  int sample_share = 1;                    // shards the sampled threshold is agreed between (dhr_search_sharded sets it): a shard then keeps only the part of
                                           // the union's r best sample scores it can plausibly hold (local_sample_rank)
  int main_chunks = 2;
  int progressive_thr = 2;
  int n_cu = 256;
  int gemm_variant = 0;                    // 2:4 layout kernel of THIS handle (0 = library default)
  int last_gemm_kernel = 0;                // DHR_INFO_GEMM_KERNEL: what the latest search's bound-GEMM launches ran (kernels.hip g_last_gemm_kernel)
  int overlap_aux = -1;                    // 0: refine / rescoring / select run on the GEMM's stream (every kernel gets the whole chip); 1: beside the next chunk's GEMM on the aux stream; -1 (default) = 1 (round 4; until then gated unsharded searches ran serially)
  int aux_cus = -1, gemm_exclusive = 0;    // CU-masked streams of the main pass (0 = no mask; -1 = default: 128 CUs for dense-only indexes, no mask for gated ones)
  int aux_cus_made = -1, gemm_excl_made = -1;
  hipStream_t s_gemm = nullptr;         // main-pass GEMM stream when CU masks are in use
  hipStream_t s_aux = nullptr;          // non-blocking stream for rescoring/select overlapped with the main-pass GEMM
  // staged search (dhr_search_begin / dhr_search_finish): state carried between the two calls
  struct { bool valid = false, done = false, gate = false, mid = false, pre = false; int Q = 0, k = 0; double rate = 0.0, rate_r = 0.0; int64_t dev_bound = 0, dev_exact = 0;
           int64_t pre_pos = 0, pre_seen = 0, pre_last_rows = 0; } pend;   // mid: dhr_search_mid ran the first slice of the main pass; pre: dhr_search_pre ran the first part of the sampled run (sample positions [0, pre_pos), pre_seen rows)
  Workspace ws, ws_fb[2];              // ws_fb[d]: workspace of fallback depth d+1 (16x candidate capacity)
  void* sh_arena = nullptr;            // grow-only scratch of dhr_search_sharded_local (sharded.hip): the gathered blocks of a step, kept between steps
  size_t sh_arena_bytes = 0;
  dhr_search_stats stats{};
};

inline void free_ws(Workspace& w) {
  hipFree(w.q_tiles); hipFree(w.q32); hipFree(w.q_idx); hipFree(w.q16); hipFree(w.q_idx8); hipFree(w.q_inexact); hipFree(w.margin); hipFree(w.i8_mul); hipFree(w.g8_q8); hipFree(w.g8_shift); hipFree(w.g8_unit); hipFree(w.tau); hipFree(w.thr);
  hipFree(w.cnt); hipFree(w.cand); if (!w.keys_alias) hipFree(w.rs_keys); hipFree(w.topk_keys); hipFree(w.d_max); hipFree(w.tau_hat); hipFree(w.fail_flags); hipFree(w.thr_hat); hipFree(w.cand2); hipFree(w.cnt2); hipFree(w.q_pack); hipFree(w.cand_r); hipFree(w.cnt_r); hipFree(w.thr_raise); hipFree(w.blk_off); hipFree(w.ovf); hipFree(w.ovf2); hipFree(w.ovf_off); hipFree(w.ovf_cap); hipFree(w.cnt_plan); hipFree(w.tier_dev); hipFree(w.boot_rows); hipFree(w.boot_bound);
  if (w.h_pinned) hipHostFree(w.h_pinned);
  if (w.h_pinned2) hipHostFree(w.h_pinned2);
  hipFree(w.d_max2); hipFree(w.d_ref); hipFree(w.d_stats);
  if (w.h_ref) hipHostFree(w.h_ref);
  if (w.h_stats) hipHostFree(w.h_stats);
  hipFree(w.q_stage); hipFree(w.qi_stage); hipFree(w.out_stage);
  w = Workspace();
}


// ---- search_core.hip
constexpr int BOOT_M = 64;       // most rows the threshold bootstrap rescores per query
constexpr int32_t MEM_DEVICE_PADDED = 2;      // internal sub-batches (the queries a fallback redoes) come from the library's own padded copies (dhr_index::dlr_pad)
#ifndef SELECT_SORT_Q
#define SELECT_SORT_Q 8       // LDS keys of select_kernel in quarters of kp: the list + one round of up to kp new keys (16 until round 5: 32 KB for top-1000 held a CU at 5 workgroups)
#endif
inline int select_sort_n(int kp) { return kp * SELECT_SORT_Q / 4; }
enum { T_GEMM = 0, T_REFINE = 1, T_RESCORE = 2, T_SELECT = 3, T_PREP = 4 };
// A search that fails half-way (a HIP error, an exception on its way to the barrier) has kernels in flight on the caller's stream and on the
// handle's aux / GEMM streams, all working on the handle's workspace: the streams are drained before the call returns, so that the next call on
// the handle starts from idle streams.  Disarmed on the successful way out (which synchronises, or hands the stream back, by its own rules).
struct Drain {
  dhr_index* ix; hipStream_t s; bool armed = true;
  ~Drain() {
    if (!armed) return;
    (void)hipStreamSynchronize(s);
    if (ix->s_aux) (void)hipStreamSynchronize(ix->s_aux);
    if (ix->s_gemm) (void)hipStreamSynchronize(ix->s_gemm);
    (void)hipGetLastError();
  }
};
struct Timer {
  bool on; hipStream_t s; std::vector<std::pair<hipEvent_t, hipEvent_t>> ev; std::vector<int> kind;
  ~Timer() { for (auto& e : ev) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); } }      // a call that failed before collect()
  void begin(int k) { begin_on(k, s); }
  void end() { end_on(s); }
  void begin_on(int k, hipStream_t st) { if (!on) return; ev.reserve(ev.size() + 1); kind.reserve(kind.size() + 1); hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b); hipEventRecord(a, st); ev.push_back({a, b}); kind.push_back(k); }
  void end_on(hipStream_t st) { if (!on) return; hipEventRecord(ev.back().second, st); }
  void collect(double* ms /*[5]*/) {
    for (size_t i = 0; i < ev.size(); ++i) {
      float t = 0.f;
      if (hipEventElapsedTime(&t, ev[i].first, ev[i].second) == hipSuccess) ms[kind[i]] += t;
      hipEventDestroy(ev[i].first); hipEventDestroy(ev[i].second);
    }
    ev.clear(); kind.clear();
  }
};

int ensure_ws(dhr_index* ix, Workspace& w, int n_queries, int k, int64_t keys_ld_min, int64_t cap_mult = 1, bool use_refine = true, bool queries_only = false);
int check_queries(const dhr_index* ix, const dhr_query_batch* qb);
int grow(void*& p, size_t& have, size_t need, int64_t& total);
int prep_queries(dhr_index* ix, Workspace& w, const dhr_query_batch* qb, hipStream_t s);
dhr::RescoreArgs base_rescore_args(const dhr_index* ix, const Workspace& w, int n_queries, bool gate);
int adaptive_rank(int r, double phi);
void plan_sampling(const dhr_index* ix, int k, int& S, int& r);
int local_sample_rank(const dhr_index* ix, int r);
int64_t pre_positions(int64_t n_sample);
int mid_share16();
int64_t head_rows(const dhr_index* ix, int S, int r_eff);
// Leaves the sorted top-k keys of every query in w.topk_keys.  stage 0: whole search; 1 dhr_search_begin; 2 dhr_search_finish; 3 dhr_search_mid; 4 dhr_search_pre; 5 dhr_search_begin_rest
int search_core(dhr_index* ix, Workspace& w, const dhr_query_batch* qb, int k, int depth, Timer& tm, dhr_search_stats& st, hipStream_t s, int stage = 0,
                const float* tau_ext = nullptr);
