// Host controller + C ABI of libdhr_hip.so (include/dhr_hip.h).
//
// Search = phases over growing corpus chunks:
//   phase 0   : the first rows are scored exhaustively (exact) to seed every query's top-k / tau
//   phase p>0 : bound GEMM over the next chunk with the fused filter  U >= tau - margin  -> candidate
//               lists; exact rescoring of the candidates; per-query top-k merge -> new tau.
// tau (exact k-th best so far) never exceeds the final k-th best and U >= exact score, so no row of
// the true top-k is ever dropped; chunk sizes adapt to the observed candidate counts, and a phase
// whose candidate list overflowed is re-run in halves (a chunk of <= cap rows cannot overflow).
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include "dhr_internal.h"
#include <hip/hip_ext.h>

using namespace dhr;

// the calling thread's error record lives in abi.cpp (a fixed buffer: recording a failure does not allocate)
static int set_error(int code, const std::string& msg) { return dhr_set_error_message(code, msg.c_str()); }
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

static void free_ws(Workspace& w) {
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

extern "C" void dhr_index_destroy(dhr_index* ix) try {
  if (!ix) return;
  hipSetDevice(ix->device);
  free_ws(ix->ws);
  free_ws(ix->ws_fb[0]);
  free_ws(ix->ws_fb[1]);
  if (ix->s_aux) hipStreamDestroy(ix->s_aux);
  if (ix->s_gemm) hipStreamDestroy(ix->s_gemm);
  hipFree(ix->sh_arena);
  hipFree(ix->resid8); hipFree(ix->i8_col_scale); hipFree(ix->g8_inv_cs); hipFree(ix->g8_w); hipFree(ix->g8_rsum); hipFree(ix->tiles); hipFree(ix->c_idx); hipFree(ix->vals_rm); hipFree(ix->bucket_map); hipFree(ix->heavy_key);      // (heavy_val points into the heavy_key records)
  delete ix;
} DHR_CATCH_VOID

static int g_opt_dense_i8 = -1;      // -1: gated indexes with ungated columns only; 0: never; 1: dense-only indexes too
static int g_opt_gated_i8 = -1;      // int8 image of the gated half: -1 by corpus size (>= GATED_I8_MIN_ROWS rows; _NARROW where the ungated half is narrower than half the gated one), 0 never, 1 wherever the layout allows it
// The int8 gated image takes ~30 % off the bound GEMM and lets ~1.5-2x the rows through the filter (its values are rounded UP): the GEMM scales
// with the rows of the shard, the extra refine / rescoring work with the queries only.  Measured (exact search, ms per step, fp16 / int8 image):
// 8.84 M x (768+768) 169.7 / 137.1; a 1.1 M-row shard of it 3.5 / 2.9 per eighth of the step; 5.4 M x (768+128) 80.0 / 71.9 and 84.9 / 85.6;
// 2.7 M 28.5 / 35.7; 0.52 M 27.8 / 44.5; 58 k 2.9 / 3.7.
// Break-even: ~2 M rows where the ungated half is as wide as the gated one (0.66 ps saved per (query, row) pair against ~1.4 us of extra
// refine / rescoring per query), ~5 M rows with a narrow ungated half (0.35 ps per pair); a SHARD of a sharded search collects only its share
// of the candidates, so it breaks even 8x earlier -- hence 1 M / 4 M.
constexpr int64_t GATED_I8_MIN_ROWS = 1000000, GATED_I8_MIN_ROWS_NARROW = 4000000;
constexpr int64_t DENSE_ONLY_I8_MIN_ROWS = 1000000;      // dense-only indexes: the int8 image by default from this many rows (if its margin is small enough, dhr_index_create)
extern "C" int dhr_set_option(int32_t option, int64_t value) try {
  if (option == DHR_OPT_DENSE_I8) { g_opt_dense_i8 = value < 0 ? -1 : (value != 0); return DHR_OK; }
  if (option == DHR_OPT_GATED_I8) { g_opt_gated_i8 = value < 0 ? -1 : (value != 0); return DHR_OK; }
  return set_error(DHR_ERR_INVALID, "unknown option");
} DHR_CATCH_STATUS
extern "C" int dhr_index_get_info(const dhr_index* ix, int32_t what, double* out) try {
  if (!ix || !out) return set_error(DHR_ERR_INVALID, "null argument");
  switch (what) {
    case DHR_INFO_DENSE_I8: *out = ix->dense_i8 ? 1.0 : 0.0; return DHR_OK;
    case DHR_INFO_I8_SCALE: *out = ix->i8_scale; return DHR_OK;
    case DHR_INFO_I8_ROW_ERR: *out = ix->i8_ec; return DHR_OK;
    case DHR_INFO_I8_ROW_NORM: *out = ix->i8_nc; return DHR_OK;
    case DHR_INFO_ROW_NORM_MAX: *out = ix->dmax; return DHR_OK;
    case DHR_INFO_GATED_I8: *out = ix->gated_i8 ? 1.0 : 0.0; return DHR_OK;
    case DHR_INFO_GEMM_KERNEL: *out = (double)ix->last_gemm_kernel; return DHR_OK;
    case DHR_INFO_TILE_BYTES: *out = (double)((size_t)ix->n_tiles * ((size_t)ix->ts * (ix->gated_i8 ? S8_STAGE_A : SP_STAGE_A) + (size_t)ix->td * SP_DENSE)); return DHR_OK;
  }
  return set_error(DHR_ERR_INVALID, "unknown info id");
} DHR_CATCH_STATUS

extern "C" int dhr_index_set_param(dhr_index* ix, int32_t param, int64_t value) try {
  if (!ix) return set_error(DHR_ERR_INVALID, "null index");
  switch (param) {
    case DHR_PARAM_CAND_CAP:
      if (value < 1024 || value > (1 << 22)) return set_error(DHR_ERR_INVALID, "cand_cap must be in [1024, 4194304]");
      ix->cand_cap = value; return DHR_OK;
    case DHR_PARAM_FIRST_ROWS:
      if (value < 0) return set_error(DHR_ERR_INVALID, "first_rows must be >= 0");
      ix->first_rows = value; return DHR_OK;
    case DHR_PARAM_PROFILE: ix->profile = value != 0; return DHR_OK;
    case DHR_PARAM_SAMPLE_PERIOD:
      if (value < 0 || value > 256) return set_error(DHR_ERR_INVALID, "sample_period must be in [0,256] (0/1 = off)");
      ix->sample_period = (int)value; return DHR_OK;
    case DHR_PARAM_ASYNC_CONTROLLER: ix->async_ctl = value < 0 ? 0 : (value > 2 ? 2 : (int)value); return DHR_OK;
    case DHR_PARAM_LIST_STRIDE:
      if (value != 0 && (value < 256 || value > (1 << 22) || value % 256)) return set_error(DHR_ERR_INVALID, "list_stride must be 0 (default) or a multiple of 256 in [256, 4194304]");
      ix->list_stride = value; return DHR_OK;
    case DHR_PARAM_SAMPLE_SHARE:
      if (value < 1 || value > 4096) return set_error(DHR_ERR_INVALID, "sample_share must be in [1,4096]");
      ix->sample_share = (int)value; return DHR_OK;
    case DHR_PARAM_MAIN_CHUNKS:
      if (value < 1 || value > 64) return set_error(DHR_ERR_INVALID, "main_chunks must be in [1,64]");
      ix->main_chunks = (int)value; return DHR_OK;
    case DHR_PARAM_PROGRESSIVE_THR: ix->progressive_thr = value < 0 ? 0 : (value > 2 ? 2 : (int)value); return DHR_OK;
    case DHR_PARAM_AUX_CUS:
      if (value < 0 || value > 192 || value % 8) return set_error(DHR_ERR_INVALID, "aux_cus must be a multiple of 8 in [0,192]");
      ix->aux_cus = (int)value; return DHR_OK;
    case DHR_PARAM_GEMM_EXCLUSIVE: ix->gemm_exclusive = value != 0; return DHR_OK;
    case DHR_PARAM_OVERLAP_AUX: ix->overlap_aux = value < 0 ? -1 : value != 0; return DHR_OK;
    case DHR_PARAM_GEMM_VARIANT:
#ifdef DHR_AB_VARIANTS
      if (value == 6) { ix->gemm_variant = 6; return DHR_OK; }      // A/B builds: persistent workgroups on gated_i8 indexes (tools/ab/gemm_g8p.hip)
#endif
      if (value != 4 && value != 5) return set_error(DHR_ERR_INVALID, "gemm_variant: 4 (4 waves, 128 x 128 wave tiles) or 5 (8 waves, 128 x 64 wave tiles; default) -- the fp16-gated kernel; integer (gated_i8) indexes have one kernel");
      ix->gemm_variant = (int)value; return DHR_OK;
    case DHR_PARAM_MAX_GROWTH:
      if (value < 1 || value > 1024) return set_error(DHR_ERR_INVALID, "max_growth must be in [1,1024] sixteenths");
      ix->max_growth16 = (int)value; return DHR_OK;
  }
  return set_error(DHR_ERR_INVALID, "unknown parameter");
} DHR_CATCH_STATUS

extern "C" int dhr_index_device(const dhr_index* ix) try { return ix ? ix->device : -1; } DHR_CATCH_VALUE(-1)
extern "C" void dhr_internal_index_arena(dhr_index* ix, void*** base, size_t** bytes) try { *base = &ix->sh_arena; *bytes = &ix->sh_arena_bytes; } DHR_CATCH_VOID
extern "C" int64_t dhr_index_device_bytes(const dhr_index* ix) try { return ix ? ix->index_bytes + ix->ws.bytes + ix->ws_fb[0].bytes + ix->ws_fb[1].bytes + (int64_t)ix->sh_arena_bytes : 0; } DHR_CATCH_VALUE(0)
extern "C" int dhr_get_stats(const dhr_index* ix, dhr_search_stats* out) try {
  if (!ix || !out) return set_error(DHR_ERR_INVALID, "null argument");
  *out = ix->stats;
  return DHR_OK;
} DHR_CATCH_STATUS

// ------------------------------------------------------------------------------------------ index build
// Pass 1 of the index build: the caller's rows (host rows staged block by block) -> row-major device copy vals_rm, norms and
// sign scan.  Everything else (bucket maps, operand tiles, refine lists) is derived from vals_rm / c_idx on the device.
static int ingest(dhr_index* ix, const dhr_index_desc* d, uint32_t* d_flags /* {max_sq, neg} */, void* stage, int64_t block_rows,
                  hipStream_t s) {
  const int64_t n = ix->n_rows;
  for (int64_t lo = 0; lo < n; lo += block_rows) {
    const int64_t rows = std::min(block_rows, n - lo);
    const __half* src;
    int64_t ld;
    if (ix->dlr_pad > 0) {        // [gated | ungated] of the caller -> [gated | zero slices | ungated] (the staging buffer was zeroed once)
      const hipMemcpyKind kind = d->mem_kind == DHR_MEM_HOST ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice;
      const int d_in = ix->d_dlr - ix->dlr_pad;
      const char* base = (const char*)d->value + lo * d->ld_value * 2;
      HIP_TRY(hipMemcpy2DAsync(stage, (size_t)ix->k * 2, base, (size_t)d->ld_value * 2, (size_t)d_in * 2, (size_t)rows, kind, s));
      if (ix->d_cls > 0)
        HIP_TRY(hipMemcpy2DAsync((char*)stage + (size_t)ix->d_dlr * 2, (size_t)ix->k * 2, base + (size_t)d_in * 2, (size_t)d->ld_value * 2,
                                 (size_t)ix->d_cls * 2, (size_t)rows, kind, s));
      src = (const __half*)stage;
      ld = ix->k;
    } else if (d->mem_kind == DHR_MEM_HOST) {
      HIP_TRY(hipMemcpy2DAsync(stage, (size_t)ix->k * 2, (const char*)d->value + lo * d->ld_value * 2,
                               (size_t)d->ld_value * 2, (size_t)ix->k * 2, (size_t)rows, hipMemcpyHostToDevice, s));
      src = (const __half*)stage;
      ld = ix->k;
    } else {
      src = (const __half*)d->value + lo * d->ld_value;
      ld = d->ld_value;
    }
    HIP_TRY(launch_scan_rows(src, ld, rows, ix->d_dlr, ix->k, d_flags, d_flags + 1, s));
    HIP_TRY(launch_copy_rows(src, ld, rows, ix->k, ix->k_rm, ix->vals_rm + lo * ix->k_rm, s));
    if (d->mem_kind == DHR_MEM_HOST || ix->dlr_pad > 0) HIP_TRY(hipStreamSynchronize(s));   // the staging buffer is reused
  }
  return DHR_OK;
}
// Pass 2: the bound-GEMM operand tiles from the device copy (needs the bucket map and abs_mode).
static int build_tiles(dhr_index* ix, hipStream_t s) {
  const int64_t n = ix->n_rows, fill = ix->n_tiles * TILE_ROWS;
  // stage images: 2:4 sparse stages of 32 gated slices, then stages of 32 (fp16) / 64 (int8) ungated columns
  HIP_TRY(launch_tile_rows_sparse(ix->vals_rm, ix->k_rm, 0, n, fill, ix->d_dlr, ix->d_cls, ix->ts, ix->td, ix->c_idx, ix->idx_dtype,
                                  ix->bucket_map, ix->abs_mode, (char*)ix->tiles, ix->dense_i8 ? 1.f / ix->i8_scale : 0.f, ix->i8_col_scale,
                                  ix->gated_i8 ? ix->g8_inv_cs : nullptr, s));
  return DHR_OK;
}

// Per-slice index-value -> bucket table, balanced by value MASS (greedy: heaviest value first into the lightest bucket;
// values that never occur with a non-zero entry are dealt round-robin).
static void build_bucket_map(const std::vector<float>& hist, int d_dlr, int nb, std::vector<uint8_t>& map) {
  map.assign((size_t)d_dlr * 256, 0);
  std::vector<int> order(256);
  std::vector<double> load(nb);
  for (int j = 0; j < d_dlr; ++j) {
    const float* h = &hist[(size_t)j * 256];
    for (int v = 0; v < 256; ++v) order[v] = v;
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return h[a] > h[b]; });
    std::fill(load.begin(), load.end(), 0);
    int rr = 0;
    for (int v : order) {
      int best = 0;
      if (h[v] == 0) best = rr++ % nb;                   // unseen values: round robin
      else
        for (int b = 1; b < nb; ++b)
          if (load[b] < load[best]) best = b;
      map[(size_t)j * 256 + v] = (uint8_t)best;
      load[best] += h[v];
    }
  }
}

extern "C" int dhr_index_create(const dhr_index_desc* d_user, dhr_index** out) try {
  if (!d_user || !out) return set_error(DHR_ERR_INVALID, "null argument");
  // --emb_dim that is not a multiple of 8: the library appends zero slices (dhr_index::dlr_pad); below, `d` is the descriptor with
  // the padded width -- only the two places that READ the caller's arrays (the index copy, ingest) use the caller's widths
  dhr_index_desc d_padded = *d_user;
  const int dlr_pad = (d_user->d_dlr > 0 && d_user->index != nullptr && d_user->d_dlr % 8) ? 8 - d_user->d_dlr % 8 : 0;
  if (d_user->ld_value < (int64_t)d_user->d_dlr + d_user->d_cls) return set_error(DHR_ERR_INVALID, "bad value pointer / ld_value");
  if (d_user->index && d_user->index_dtype != DHR_IDX_NONE && d_user->ld_index < d_user->d_dlr) return set_error(DHR_ERR_INVALID, "bad ld_index");
  d_padded.d_dlr += dlr_pad;
  if (dlr_pad) { d_padded.ld_value = std::max<int64_t>(d_padded.ld_value, (int64_t)d_padded.d_dlr + d_padded.d_cls); d_padded.ld_index = std::max<int64_t>(d_padded.ld_index, d_padded.d_dlr); }
  const dhr_index_desc* d = &d_padded;
  *out = nullptr;
  if (d->n_rows <= 0 || d->n_rows >= (int64_t)0xFFFFFF00ll) return set_error(DHR_ERR_INVALID, "n_rows must be in [1, 2^32-256)");
  if (d->d_dlr < 0 || d->d_cls < 0 || d->d_dlr + d->d_cls <= 0) return set_error(DHR_ERR_INVALID, "bad d_dlr/d_cls");
  if (!d->value || d->ld_value < d->d_dlr + d->d_cls) return set_error(DHR_ERR_INVALID, "bad value pointer / ld_value");
  const bool has_idx = d->index != nullptr && d->index_dtype != DHR_IDX_NONE;
  if (has_idx != (d->d_dlr > 0))
    return set_error(DHR_ERR_INVALID, "an index array is required iff d_dlr > 0 (dense-only: index=NULL, d_dlr=0)");
  if (has_idx && (d->index_dtype < DHR_IDX_U8 || d->index_dtype > DHR_IDX_I16)) return set_error(DHR_ERR_INVALID, "bad index_dtype");
  if (has_idx && d->ld_index < d->d_dlr) return set_error(DHR_ERR_INVALID, "bad ld_index");
  if (d->d_dlr + d->d_cls > 8192) return set_error(DHR_ERR_UNSUPPORTED, "more than 8192 columns");
  if (d->idx_buckets < 0 || d->idx_buckets > 2)
    return set_error(d->idx_buckets > 2 ? DHR_ERR_UNSUPPORTED : DHR_ERR_INVALID, "idx_buckets must be 0 (default: 2), 1 (ungated bound) or 2: the bucket-split operands of more than two buckets went with the K-step tile layout (round 6)");
  if (d->mem_kind != DHR_MEM_HOST && d->mem_kind != DHR_MEM_DEVICE) return set_error(DHR_ERR_INVALID, "bad mem_kind");
  dhr::alloc_checkpoint();
  HIP_TRY(hipSetDevice(d->device));

  // everything below is released by this guard unless the build reaches its end (early returns and exceptions alike)
  struct Build {
    dhr_index* ix = nullptr; void* stage = nullptr; uint32_t* d_flags = nullptr; uint32_t* d_hist = nullptr;
    ~Build() { (void)hipFree(stage); (void)hipFree(d_flags); (void)hipFree(d_hist); if (ix) dhr_index_destroy(ix); }
  } build;
  dhr_index* ix = build.ix = new dhr_index();
  ix->device = d->device;
  ix->idx_buckets_req = d->idx_buckets;
  { hipDeviceProp_t pr; HIP_TRY(hipGetDeviceProperties(&pr, d->device)); ix->n_cu = pr.multiProcessorCount; }
  ix->n_rows = d->n_rows;
  ix->row_offset = d->row_offset;
  ix->d_dlr = d->d_dlr;
  ix->d_cls = d->d_cls;
  ix->dlr_pad = dlr_pad;
  ix->k = d->d_dlr + d->d_cls;
  ix->k_rm = (int)round_up(ix->k, TILE_K);
  // ONE layout (round 6): stage images -- 2:4 sparse stages of 32 gated slices (two index buckets per slice; idx_buckets = 1: every index
  // value in bucket 0, the ungated bound) and stages of 32 (fp16) / 64 (int8) ungated columns, an EVEN number of either kind: gated widths that
  // are no multiple of 64 and odd ungated stage counts are rounded up with all-zero stages (the tile builder and query_prep_kernel write zeros
  // behind d_dlr / d_cls), so that every index runs on the two 8-wave kernels.  The K-step tile layout (bucket counts above two, their own
  // kernel) and the 12-wave kernel of odd stage counts were retired.
  auto even_up = [](int v) { return v + (v & 1); };
  // int8 image of the ungated columns (process-wide option / DHR_DENSE_I8; default: gated indexes only, where the gated part
  // dominates the spread of the scores and the int8 margin costs few extra candidates -- DESIGN.md section 6b)
  int want_i8 = g_opt_dense_i8;
  if (const char* e = getenv("DHR_DENSE_I8")) want_i8 = atoi(e);
  bool dense_only_trial = false;
  if (has_idx) {
    ix->n_buckets = d->idx_buckets == 1 ? 1 : 2;
    ix->ts = even_up((d->d_dlr + 31) / 32);
    ix->dense_i8 = d->d_cls > 0 && (want_i8 < 0 || want_i8 > 0);
    ix->td = ix->dense_i8 ? 2 * ((d->d_cls + 127) / 128) : even_up((d->d_cls + 31) / 32);     // ungated columns in 32-column fp16 stages or PAIRS of 64-column int8 stages
    ix->kt = ix->ts * TILE_K + ix->td * 32;            // operand bytes / 2 per row (fp16: logical columns, two bucket columns per gated slice)
    // gated half as int8 on the 2:4 int8 instruction (gemm_g8.hip): default for large shards whose ungated half (if any)
    // is the int8 image too; DHR_GATED_I8=0 / dhr_set_option(DHR_OPT_GATED_I8, 0) keeps the fp16 image
    int want_g8 = g_opt_gated_i8;
    if (const char* e = getenv("DHR_GATED_I8")) want_g8 = atoi(e);
    if (want_g8 < 0) want_g8 = d->n_rows >= (2 * d->d_cls >= d->d_dlr ? GATED_I8_MIN_ROWS : GATED_I8_MIN_ROWS_NARROW) ? 1 : 0;
    ix->gated_i8 = want_g8 != 0 && ix->n_buckets == 2 && (d->d_cls == 0 || ix->dense_i8) && d->d_dlr <= 4096;
  } else {
    // dense-only index: the same stage images with no gated stage (ts = 0)
    ix->n_buckets = 1;
    ix->ts = 0;
    // int8 image for a dense-only index: explicitly (option = 1), or -- default, large shards -- on trial: the margin it needs is measured
    // below (i8_row_err pass) and the fp16 image is kept where it would be too large a share of the spread of the scores
    dense_only_trial = want_i8 < 0 && d->n_rows >= DENSE_ONLY_I8_MIN_ROWS && d->d_cls >= 128;
    ix->dense_i8 = want_i8 > 0 || dense_only_trial;
    ix->td = ix->dense_i8 ? 2 * ((d->d_cls + 127) / 128) : even_up((d->d_cls + 31) / 32);
    ix->kt = ix->td * 32;
  }
  ix->ksteps = (ix->kt + TILE_K - 1) / TILE_K;
  ix->idx_dtype = has_idx ? d->index_dtype : DHR_IDX_NONE;
  ix->n_tiles = (d->n_rows + TILE_ROWS - 1) / TILE_ROWS;
  hipStream_t s = nullptr;
  void*& stage = build.stage;
  uint32_t*& d_flags = build.d_flags;
  uint32_t*& d_hist = build.d_hist;
  int rc = DHR_OK;
  auto fail = [&](int code) { return code; };        // (the guard above releases the handle and the scratch)

  const size_t rm_bytes = (size_t)ix->n_rows * ix->k_rm * 2;
  if (hipMalloc((void**)&ix->vals_rm, rm_bytes) != hipSuccess)
    return fail(set_error(DHR_ERR_HIP, "hipMalloc of " + std::to_string(rm_bytes) + " bytes for the row-major corpus copy failed"));
  ix->index_bytes = (int64_t)rm_bytes;
  if (hipMalloc((void**)&d_flags, 16) != hipSuccess) return fail(set_error(DHR_ERR_HIP, "hipMalloc failed"));
  if (hipMemsetAsync(d_flags, 0, 16, s) != hipSuccess) return fail(set_error(DHR_ERR_HIP, "hipMemsetAsync failed"));
  // the index array
  if (has_idx) {
    const int es = idx_esize(d->index_dtype);
    const size_t ib = (size_t)d->n_rows * d->d_dlr * es;
    if (hipMalloc(&ix->c_idx, ib) != hipSuccess) return fail(set_error(DHR_ERR_HIP, "hipMalloc of the index array failed"));
    ix->index_bytes += (int64_t)ib;
    if (dlr_pad && hipMemsetAsync(ix->c_idx, 0, ib, s) != hipSuccess) return fail(set_error(DHR_ERR_HIP, "hipMemsetAsync failed"));
    if (hipMemcpy2DAsync(ix->c_idx, (size_t)d->d_dlr * es, d_user->index, (size_t)d_user->ld_index * es, (size_t)d_user->d_dlr * es,
                         (size_t)d->n_rows, d->mem_kind == DHR_MEM_HOST ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice,
                         s) != hipSuccess)
      return fail(set_error(DHR_ERR_HIP, "copy of the index array failed"));
  }
  // pass 1: row-major device copy of the values (+ norms, sign scan)
  const int64_t block_rows = 65536;
  if ((d->mem_kind == DHR_MEM_HOST || dlr_pad > 0) &&
      hipMalloc(&stage, (size_t)std::min<int64_t>(block_rows, d->n_rows) * ix->k * 2) != hipSuccess)
    return fail(set_error(DHR_ERR_HIP, "hipMalloc of the staging buffer failed"));
  if (dlr_pad > 0 && hipMemsetAsync(stage, 0, (size_t)std::min<int64_t>(block_rows, d->n_rows) * ix->k * 2, s) != hipSuccess)
    return fail(set_error(DHR_ERR_HIP, "hipMemsetAsync failed"));
  if ((rc = ingest(ix, d_user, d_flags, stage, block_rows, s)) != DHR_OK) return fail(rc);
  uint32_t flags[4] = {0, 0, 0, 0};
  if (hipMemcpy(flags, d_flags, 16, hipMemcpyDeviceToHost) != hipSuccess) return fail(set_error(DHR_ERR_HIP, "hipMemcpy failed"));
  float max_sq;
  memcpy(&max_sq, &flags[0], 4);
  ix->dmax = std::sqrt(max_sq) * 1.0005f + 1e-30f;
  ix->abs_mode = flags[1] != 0;     // negative gated values: the bound needs |q|.|d| on the gated half
  if (ix->dense_i8) {
    float amax;
    memcpy(&amax, &flags[2], 4);
    float gmax;
    memcpy(&gmax, &flags[3], 4);
    ix->i8_scale = std::max(amax > 0.f ? amax / 127.f : 1.f, gmax / 60000.f);      // gated values must fit fp16 in units of the scale
    // per-column steps: column j is quantised in its own step cs_j <= scale, the query side carries cs_j / scale as a weight
    // (query_prep_kernel) -- a few large columns (outlier dimensions of encoder outputs) then do not push
    // every other column into a handful of int8 levels
    std::vector<uint32_t> cm((size_t)ix->d_cls, 0u);
    {
      DevMem cmd;
      uint32_t*& d_cm = (uint32_t*&)cmd.p;
      if (hipMalloc((void**)&d_cm, cm.size() * 4) != hipSuccess || hipMalloc((void**)&ix->i8_col_scale, cm.size() * 4) != hipSuccess)
        return fail(set_error(DHR_ERR_HIP, "hipMalloc failed"));
      if (hipMemsetAsync(d_cm, 0, cm.size() * 4, s) != hipSuccess ||
          launch_col_absmax(ix->vals_rm, ix->k_rm, ix->n_rows, ix->d_dlr, ix->d_cls, d_cm, s) != hipSuccess ||
          hipMemcpy(cm.data(), d_cm, cm.size() * 4, hipMemcpyDeviceToHost) != hipSuccess)
        return fail(set_error(DHR_ERR_HIP, "column scan failed"));
    }
    std::vector<float> cs(cm.size());
    for (size_t j = 0; j < cm.size(); ++j) {
      float m;
      memcpy(&m, &cm[j], 4);
      // step of column j = scale * (its largest |value| / the largest of all)^(3/4): the exponent splits a column's dynamic range
      // between the corpus image (finer steps for small columns) and the query weights (which then stay within ~two orders of
      // magnitude) -- measured on anisotropic columns the margin is 1.8x smaller than with exponent 1 and 4-6x smaller than with
      // one step for all columns; on iid columns all exponents are equal (tests/test_i8_bound.py)
      const float ratio = m > 0.f ? std::min(m / (127.f * ix->i8_scale), 1.f) : 1.f;
      cs[j] = ix->i8_scale * std::max(std::pow(ratio, 0.75f), 1.f / 1024.f);
    }
    if (hipMemcpy(ix->i8_col_scale, cs.data(), cs.size() * 4, hipMemcpyHostToDevice) != hipSuccess)
      return fail(set_error(DHR_ERR_HIP, "hipMemcpy failed"));
    if (hipMemsetAsync(d_flags, 0, 16, s) != hipSuccess ||
        launch_i8_row_err(ix->vals_rm, ix->k_rm, ix->n_rows, ix->d_dlr, ix->d_cls, ix->i8_scale, ix->i8_col_scale, d_flags, s) != hipSuccess ||
        hipMemcpy(flags, d_flags, 8, hipMemcpyDeviceToHost) != hipSuccess)
      return fail(set_error(DHR_ERR_HIP, "int8 row-error pass failed"));
    float e2, n2;
    memcpy(&e2, &flags[0], 4); memcpy(&n2, &flags[1], 4);
    ix->i8_ec = std::sqrt(e2) * 1.001f;
    ix->i8_nc = std::sqrt(n2) * 1.001f;
    if (dense_only_trial) {
      // The filter margin of the int8 image is ~ ||q|| ec (+ ~60 % for the query's own rounding) while the scores of a dense-only index
      // spread by ~ ||q|| ||d|| / sqrt(d_cls): with sqrt(d_cls) ec / ||d|| = 0.36 (the benchmark's N(0, 0.1) columns: margin 0.6 sigma) the
      // int8 search lets ~8x the rows through the filter and is still faster (config 2: 85.1 vs 89.1 ms per step: the int8 GEMM takes half
      // the fp16 one's time and is not held by the package power cap, the extra rescoring of 1.5 KB rows costs less than that); the
      // candidates grow exponentially with the ratio, so anything much coarser keeps the fp16 image.
      const float ratio = std::sqrt((float)ix->d_cls) * ix->i8_ec / std::max(ix->i8_nc, 1e-30f);
      if (!(ratio <= 0.45f)) {
        ix->dense_i8 = false;
        hipFree(ix->i8_col_scale); ix->i8_col_scale = nullptr;
        ix->i8_scale = ix->i8_ec = ix->i8_nc = 0.f;
        ix->td = ((d->d_cls + 31) / 32 + 1) & ~1;
        ix->kt = ix->td * 32;
        ix->ksteps = (ix->kt + TILE_K - 1) / TILE_K;
      }
    }
  }
  // dense-only int8 index: residual image = the refine level between the filter and the exact rescoring.
  // The int8 margin is  ||q'|| ec (corpus rounding) + ||q' - q8'|| nc (query rounding); with the residuals the first term is MEASURED per
  // candidate from 384 bytes (four bits per value) instead of bounded, and the candidates that only the corpus half of the margin let through
  // never reach the 1.5 KB rows of the exact rescoring.
  {
    const int ld = (int)round_up(ix->d_cls, 256) / 2;           // four bits per value
    if (ix->dense_i8 && ix->d_dlr == 0 && ld <= 512 && ix->i8_ec > 0.f) {
      const size_t rb = (size_t)ix->n_rows * ld;
      if (hipMalloc((void**)&ix->resid8, rb) != hipSuccess) return fail(set_error(DHR_ERR_HIP, "hipMalloc of the residual image failed"));
      ix->index_bytes += (int64_t)rb;
      ix->resid_ld = ld;
      if (launch_resid_build(ix->vals_rm, ix->k_rm, ix->n_rows, ix->d_dlr, ix->d_cls, ix->i8_col_scale, ix->resid8, ld, s) != hipSuccess)
        return fail(set_error(DHR_ERR_HIP, "residual image launch failed"));
      ix->resid_ec2 = std::sqrt((float)ix->d_cls) * ix->i8_scale / 28.f;       // every residual is within half of 1/14 of its column's step: scale / 28 in the weighted space
    }
  }
  const size_t tile_bytes = (size_t)ix->n_tiles * ((size_t)ix->ts * (ix->gated_i8 ? S8_STAGE_A : SP_STAGE_A) + (size_t)ix->td * SP_DENSE);
  if (hipMalloc((void**)&ix->tiles, tile_bytes) != hipSuccess)
    return fail(set_error(DHR_ERR_HIP, "hipMalloc of " + std::to_string(tile_bytes) + " bytes for the corpus tiles failed"));
  ix->index_bytes += (int64_t)tile_bytes;
  if (ix->gated_i8) {
    // steps of the gated columns: s_ref = (largest gated |value|) / 127, column j in s_ref * (its own largest / the largest)^(3/4)
    // (the exponent splits a small column's range between a finer corpus step and a smaller query weight, as for the ungated columns)
    float gmax;
    memcpy(&gmax, &flags[3], 4);
    std::vector<uint32_t> cm((size_t)ix->d_dlr, 0u);
    {
      DevMem cmd;
      uint32_t*& d_cm = (uint32_t*&)cmd.p;
      if (hipMalloc((void**)&d_cm, cm.size() * 4) != hipSuccess || hipMalloc((void**)&ix->g8_inv_cs, cm.size() * 4) != hipSuccess ||
          hipMalloc((void**)&ix->g8_w, cm.size() * 4) != hipSuccess)
        return fail(set_error(DHR_ERR_HIP, "hipMalloc failed"));
      if (hipMemsetAsync(d_cm, 0, cm.size() * 4, s) != hipSuccess ||
          launch_col_absmax(ix->vals_rm, ix->k_rm, ix->n_rows, 0, ix->d_dlr, d_cm, s) != hipSuccess ||
          hipMemcpy(cm.data(), d_cm, cm.size() * 4, hipMemcpyDeviceToHost) != hipSuccess)
        return fail(set_error(DHR_ERR_HIP, "column scan failed"));
    }
    ix->g8_sref = (gmax > 0.f ? gmax : 1.f) * (1.00001f / 127.f);
    std::vector<float> inv(cm.size()), wj(cm.size());
    for (size_t j = 0; j < cm.size(); ++j) {
      float m;
      memcpy(&m, &cm[j], 4);
      const float ratio = m > 0.f ? std::min(m / gmax, 1.f) : 1.f;
      const float f = std::max(std::pow(ratio, 0.75f), 1.f / 1024.f);      // m / (s_ref f) = 127 ratio^(1/4) / 1.00001 <= 127
      inv[j] = 1.000001f / (ix->g8_sref * f);
      wj[j] = f * 1.000001f;
    }
    if (hipMemcpy(ix->g8_inv_cs, inv.data(), inv.size() * 4, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(ix->g8_w, wj.data(), wj.size() * 4, hipMemcpyHostToDevice) != hipSuccess)
      return fail(set_error(DHR_ERR_HIP, "hipMemcpy failed"));
    ix->g8_max_shift = 0;
    while (ix->g8_max_shift < 7 && 255.0 * 127.0 * (double)(ix->ts * 32) * (double)(2 << ix->g8_max_shift) <= 1073741824.0) ++ix->g8_max_shift;
    const size_t rb = (size_t)ix->n_tiles * TILE_ROWS * 4;
    if (hipMalloc((void**)&ix->g8_rsum, rb) != hipSuccess) return fail(set_error(DHR_ERR_HIP, "hipMalloc failed"));
    ix->index_bytes += (int64_t)rb;
    if (launch_g8_row_sum(ix->vals_rm, ix->k_rm, ix->n_rows, ix->n_tiles * TILE_ROWS, ix->d_dlr, ix->abs_mode, ix->g8_inv_cs, ix->g8_rsum, s) != hipSuccess)
      return fail(set_error(DHR_ERR_HIP, "g8_row_sum launch failed"));
  }
  // bucket maps from the value mass per (slice, index value)
  if (has_idx && ix->n_buckets == 1) {
    // idx_buckets = 1: every index value in bucket 0 (an all-zero map serves 8- and 16-bit index dtypes alike: bucket_of reads map[j][value & 255])
    const size_t mb = (size_t)d->d_dlr * 256;
    if (hipMalloc((void**)&ix->bucket_map, mb) != hipSuccess || hipMemsetAsync(ix->bucket_map, 0, mb, s) != hipSuccess)
      return fail(set_error(DHR_ERR_HIP, "bucket map allocation failed"));
  }
  if (has_idx && ix->n_buckets > 1 && idx_esize(d->index_dtype) == 1) {
    const size_t hb = (size_t)d->d_dlr * 256 * 4;
    if (hipMalloc((void**)&d_hist, hb) != hipSuccess || hipMemsetAsync(d_hist, 0, hb, s) != hipSuccess)
      return fail(set_error(DHR_ERR_HIP, "hipMalloc (index histogram) failed"));
    if (launch_idx_hist((const uint8_t*)ix->c_idx, ix->vals_rm, ix->k_rm, d->n_rows, d->d_dlr, (float*)d_hist, s) != hipSuccess)
      return fail(set_error(DHR_ERR_HIP, "idx_hist launch failed"));
    std::vector<float> hist((size_t)d->d_dlr * 256);
    if (hipMemcpy(hist.data(), d_hist, hb, hipMemcpyDeviceToHost) != hipSuccess) return fail(set_error(DHR_ERR_HIP, "hipMemcpy failed"));
    std::vector<uint8_t> map;
    build_bucket_map(hist, d->d_dlr, ix->n_buckets, map);
    if (hipMalloc((void**)&ix->bucket_map, map.size()) != hipSuccess ||
        hipMemcpy(ix->bucket_map, map.data(), map.size(), hipMemcpyHostToDevice) != hipSuccess)
      return fail(set_error(DHR_ERR_HIP, "bucket map upload failed"));
  }
  // pass 2: operand tiles
  if ((rc = build_tiles(ix, s)) != DHR_OK) return fail(rc);
  if (has_idx && d->d_dlr <= 4096) {
    const size_t hb = (size_t)d->n_rows * HEAVY_KEY_STRIDE * 4;       // one 6 x HEAVY-byte record per row: the keys, then the values
    if (hipMalloc((void**)&ix->heavy_key, hb) != hipSuccess)
      return fail(set_error(DHR_ERR_HIP, "hipMalloc of the refine lists failed"));
    ix->heavy_val = (__half*)((char*)ix->heavy_key + HEAVY * 4);
    ix->index_bytes += (int64_t)hb;
    if (launch_heavy_build(ix->vals_rm, ix->k_rm, ix->c_idx, ix->idx_dtype, d->n_rows, d->d_dlr, ix->bucket_map, ix->n_buckets,
                           ix->heavy_key, ix->heavy_val, ix->gated_i8 ? ix->g8_inv_cs : nullptr, ix->abs_mode ? 1 : 0, s) != hipSuccess)
      return fail(set_error(DHR_ERR_HIP, "heavy_build launch failed"));
  }
  if (hipStreamSynchronize(s) != hipSuccess) return fail(set_error(DHR_ERR_HIP, "index build failed on the device"));
  build.ix = nullptr;          // the caller's from here on
  *out = ix;
  return DHR_OK;
} DHR_CATCH_STATUS

// ------------------------------------------------------------------------------------------ index file
// [4096-byte header][values: n_rows x k_rm fp16][slice indices: n_rows x d_dlr][caller blob], sections page-aligned.
// The file holds the corpus in the reference's own record layout (row-major fp16 values, row-major indices), NOT
// the device images: measured, re-tiling from device memory costs 0.06 s per 2 M rows while the images would
// double the file (15 vs 7.7 GB per 2 M rows) -- reading the extra bytes is slower than recomputing them.
// What the file removes is the monolithic pickle: no unpickling, no host copies, no fp32 cast; the mapping
// is streamed to the device block by block by the ordinary ingest path.
namespace {
struct FileHeader {
  char magic[8];
  uint32_t version, header_bytes;
  int64_t n_rows, row_offset;
  int32_t d_dlr, d_cls, k_rm, idx_dtype, idx_buckets, pad0;
  uint64_t val_offset, val_bytes, idx_offset, idx_bytes, blob_offset, blob_bytes;
};
static_assert(sizeof(FileHeader) <= 4096, "header must fit one page");
const char FILE_MAGIC[8] = {'D', 'H', 'R', 'I', 'D', 'X', '1', 0};
constexpr uint32_t FILE_VERSION = 1;

bool write_all(int fd, const void* p, size_t n) {
  const char* c = (const char*)p;
  while (n) {
    const ssize_t w = write(fd, c, n);
    if (w <= 0) return false;
    c += w; n -= (size_t)w;
  }
  return true;
}
int read_header(const char* path, FileHeader& h, int* fd_out) {
  if (!path) return set_error(DHR_ERR_INVALID, "null path");
  const int fd = open(path, O_RDONLY);
  if (fd < 0) return set_error(DHR_ERR_INVALID, std::string("cannot open ") + path);
  char page[4096];
  const ssize_t got = pread(fd, page, sizeof(page), 0);
  if (got != (ssize_t)sizeof(page)) { close(fd); return set_error(DHR_ERR_INVALID, std::string(path) + " is not a device-ready index file (short header)"); }
  memcpy(&h, page, sizeof(h));
  if (memcmp(h.magic, FILE_MAGIC, 8) != 0 || h.header_bytes != 4096) { close(fd); return set_error(DHR_ERR_INVALID, std::string(path) + " is not a device-ready index file"); }
  if (h.version != FILE_VERSION) {
    close(fd);
    return set_error(DHR_ERR_UNSUPPORTED, std::string(path) + " has file format version " + std::to_string(h.version) + ", this library reads version " +
                                          std::to_string(FILE_VERSION));
  }
  if (fd_out) *fd_out = fd; else close(fd);
  return DHR_OK;
}
}  // namespace

extern "C" int dhr_index_save(const dhr_index* ix, const char* path, const void* blob, int64_t blob_bytes) try {
  if (!ix || !path || blob_bytes < 0 || (blob_bytes > 0 && !blob)) return set_error(DHR_ERR_INVALID, "null index / path or bad blob");
  HIP_TRY(hipSetDevice(ix->device));
  FileHeader h{};
  memcpy(h.magic, FILE_MAGIC, 8);
  h.version = FILE_VERSION; h.header_bytes = 4096;
  h.n_rows = ix->n_rows; h.row_offset = ix->row_offset;
  h.d_dlr = ix->d_dlr; h.d_cls = ix->d_cls; h.k_rm = ix->k_rm; h.idx_dtype = ix->idx_dtype; h.idx_buckets = ix->idx_buckets_req;
  h.pad0 = ix->dlr_pad;            // the file holds the padded records; a loaded index takes the caller's unpadded queries again
  h.val_offset = 4096; h.val_bytes = (uint64_t)ix->n_rows * ix->k_rm * 2;
  h.idx_offset = (h.val_offset + h.val_bytes + 4095) / 4096 * 4096;
  h.idx_bytes = ix->c_idx ? (uint64_t)ix->n_rows * ix->d_dlr * idx_esize(ix->idx_dtype) : 0;
  h.blob_offset = (h.idx_offset + h.idx_bytes + 4095) / 4096 * 4096;
  h.blob_bytes = (uint64_t)blob_bytes;
  const int fd = open(path, O_WRONLY | O_CREAT | O_TRUNC, 0644);
  if (fd < 0) return set_error(DHR_ERR_INVALID, std::string("cannot create ") + path);
  const size_t CH = (size_t)64 << 20;
  void* pin = nullptr;
  if (hipHostMalloc(&pin, CH, hipHostMallocDefault) != hipSuccess) { close(fd); return set_error(DHR_ERR_HIP, "hipHostMalloc failed"); }
  auto fail = [&](int code) { hipHostFree(pin); close(fd); unlink(path); return code; };
  char page[4096] = {0};
  memcpy(page, &h, sizeof(h));
  if (!write_all(fd, page, 4096)) return fail(set_error(DHR_ERR_INVALID, "write failed (header)"));
  const void* src[2] = {ix->vals_rm, ix->c_idx};
  const uint64_t off[2] = {h.val_offset, h.idx_offset}, bytes[2] = {h.val_bytes, h.idx_bytes};
  for (int i = 0; i < 2; ++i) {
    if (lseek(fd, (off_t)off[i], SEEK_SET) < 0) return fail(set_error(DHR_ERR_INVALID, "seek failed"));
    for (uint64_t done = 0; done < bytes[i]; done += CH) {
      const size_t n = (size_t)std::min<uint64_t>(CH, bytes[i] - done);
      if (hipMemcpy(pin, (const char*)src[i] + done, n, hipMemcpyDeviceToHost) != hipSuccess) return fail(set_error(DHR_ERR_HIP, "D2H copy failed while saving"));
      if (!write_all(fd, pin, n)) return fail(set_error(DHR_ERR_INVALID, "write failed (disk full?)"));
    }
  }
  if (lseek(fd, (off_t)h.blob_offset, SEEK_SET) < 0) return fail(set_error(DHR_ERR_INVALID, "seek failed"));
  if (blob_bytes > 0 && !write_all(fd, blob, (size_t)blob_bytes)) return fail(set_error(DHR_ERR_INVALID, "write failed (blob)"));
  if (blob_bytes == 0 && ftruncate(fd, (off_t)h.blob_offset) != 0) return fail(set_error(DHR_ERR_INVALID, "truncate failed"));
  hipHostFree(pin);
  if (close(fd) != 0) { unlink(path); return set_error(DHR_ERR_INVALID, "close failed"); }
  return DHR_OK;
} DHR_CATCH_STATUS

extern "C" int dhr_index_file_info(const char* path, dhr_file_info* out) try {
  if (!out) return set_error(DHR_ERR_INVALID, "null output");
  FileHeader h;
  int rc = read_header(path, h, nullptr);
  if (rc) return rc;
  out->n_rows = h.n_rows; out->row_offset = h.row_offset; out->d_dlr = h.d_dlr - ((h.pad0 > 0 && h.pad0 < 8) ? h.pad0 : 0); out->d_cls = h.d_cls;
  out->index_dtype = h.idx_dtype; out->idx_buckets = h.idx_buckets; out->file_version = h.version; out->reserved = 0;
  out->payload_bytes = (int64_t)(h.val_bytes + h.idx_bytes);
  out->blob_offset = (int64_t)h.blob_offset; out->blob_bytes = (int64_t)h.blob_bytes;
  return DHR_OK;
} DHR_CATCH_STATUS

extern "C" int dhr_index_load(const char* path, int32_t device, int64_t row_offset, dhr_index** out) try {
  if (!out) return set_error(DHR_ERR_INVALID, "null output");
  *out = nullptr;
  FileHeader h;
  int fd = -1;
  int rc = read_header(path, h, &fd);
  if (rc) return rc;
  struct stat sb;
  if (fstat(fd, &sb) != 0) { close(fd); return set_error(DHR_ERR_INVALID, "fstat failed"); }
  const int es = h.idx_bytes ? idx_esize(h.idx_dtype) : 0;
  if (h.val_offset + h.val_bytes > (uint64_t)sb.st_size || h.idx_offset + h.idx_bytes > (uint64_t)sb.st_size ||
      h.n_rows <= 0 || h.k_rm < h.d_dlr + h.d_cls || h.val_bytes != (uint64_t)h.n_rows * h.k_rm * 2 ||
      h.idx_bytes != (uint64_t)(es ? h.n_rows * h.d_dlr * es : 0)) {
    close(fd);
    return set_error(DHR_ERR_INVALID, std::string(path) + " is truncated or inconsistent");
  }
  void* map = mmap(nullptr, (size_t)sb.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
  close(fd);
  if (map == MAP_FAILED) return set_error(DHR_ERR_INVALID, "mmap failed");
  (void)madvise(map, (size_t)sb.st_size, MADV_SEQUENTIAL);
  dhr_index_desc d{};
  d.device = device; d.mem_kind = DHR_MEM_HOST; d.n_rows = h.n_rows; d.d_dlr = h.d_dlr; d.d_cls = h.d_cls;
  d.value = (const char*)map + h.val_offset; d.ld_value = h.k_rm;
  d.index = h.idx_bytes ? (const char*)map + h.idx_offset : nullptr; d.index_dtype = h.idx_bytes ? h.idx_dtype : DHR_IDX_NONE;
  d.idx_buckets = h.idx_buckets; d.ld_index = h.d_dlr;
  d.row_offset = row_offset >= 0 ? row_offset : h.row_offset;
  rc = dhr_index_create(&d, out);
  munmap(map, (size_t)sb.st_size);
  if (rc == DHR_OK && h.pad0 > 0 && h.pad0 < 8) (*out)->dlr_pad = h.pad0;
  return rc;
} DHR_CATCH_STATUS

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
constexpr int BOOT_M = 64;       // most rows the threshold bootstrap rescores per query
static int ensure_ws(dhr_index* ix, Workspace& w, int n_queries, int k, int64_t keys_ld_min, int64_t cap_mult = 1, bool use_refine = true,
                     bool queries_only = false) {
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

static int check_queries(const dhr_index* ix, const dhr_query_batch* qb) {
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
constexpr int32_t MEM_DEVICE_PADDED = 2;
static int grow(void*& p, size_t& have, size_t need, int64_t& total) {
  if (have >= need) return DHR_OK;
  if (p) hipFree(p);
  p = nullptr;
  HIP_TRY(hipMalloc(&p, need));
  total += (int64_t)(need - have);
  have = need;
  return DHR_OK;
}

// queries -> device operand tiles / fp32 copy / idx / margins (all inside the workspace)
static int prep_queries(dhr_index* ix, Workspace& w, const dhr_query_batch* qb, hipStream_t s) {
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
enum { T_GEMM = 0, T_REFINE = 1, T_RESCORE = 2, T_SELECT = 3, T_PREP = 4 };

// The refine step serves gated batches, and -- on gated_i8 indexes -- ungated ones too (--IP stage 1): there it takes the int8
// products of a row's listed entries off the bound and puts their real products back, whatever the index values (RefineArgs::ungated).
#ifndef SELECT_SORT_Q
#define SELECT_SORT_Q 8       // LDS keys of select_kernel in quarters of kp: the list + one round of up to kp new keys (16 until round 5: 32 KB for top-1000 held a CU at 5 workgroups)
#endif
static inline int select_sort_n(int kp) { return kp * SELECT_SORT_Q / 4; }
static inline bool uses_refine(const dhr_index* ix, bool gate) { return (ix->heavy_key != nullptr && (gate || ix->gated_i8)) || ix->resid8 != nullptr; }
static RescoreArgs base_rescore_args(const dhr_index* ix, const Workspace& w, int n_queries, bool gate) {
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
static int adaptive_rank(int r, double phi) {
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
static void plan_sampling(const dhr_index* ix, int k, int& S, int& r) {
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
static int local_sample_rank(const dhr_index* ix, int r) {
  if (ix->sample_share <= 1 || r <= 0) return r;
  const double m = (double)r / ix->sample_share;
  return std::min(r, (int)std::ceil(m + 5.0 * std::sqrt(m) + 4.0));
}

// Share of the main pass that dhr_search_mid runs before the shards agree on thresholds a second time, in 1/16ths (default 2 = 1/8: with the
// 1/32 sample the shards have then seen ~15 % of their rows)
// Share of the SAMPLE that dhr_search_pre streams before the shards agree on a first common threshold, in 1/16ths (default 2 = 1/8), and the
// sample positions that is (whole tile groups; 0: the sample is too small to split)
static int pre_share16() {
  static const int v = getenv("DHR_PRE_SHARE16") ? std::max(1, std::min(12, atoi(getenv("DHR_PRE_SHARE16")))) : 2;
  return v;
}
static int64_t pre_positions(int64_t n_sample) {
  if (n_sample < 64) return 0;
  const int64_t a = round_up(std::max<int64_t>(DOC_GROUP, n_sample * pre_share16() / 16), DOC_GROUP);
  return a < n_sample ? a : 0;
}
static int mid_share16() {
  static const int v = getenv("DHR_MID_SHARE16") ? std::max(1, std::min(12, atoi(getenv("DHR_MID_SHARE16")))) : 2;
  return v;
}
// Rows scored exhaustively in phase 0 (>= the rank that defines tau, so that tau exists afterwards), whole tiles -- 512 rows until round 3, 256
// since; 0 = none: a sampled search (S >= 2) bootstraps its first thresholds from the bound GEMM instead (search_core), unless the caller
// fixed the head (DHR_PARAM_FIRST_ROWS).
static int64_t head_rows(const dhr_index* ix, int S, int r_eff) {
  if (S >= 2 && ix->first_rows <= 0 && ix->n_tiles >= 8) {
    // the bootstrap takes the r0-th best of at most BOOT_M rows of tile 0: a sampling rule under which r0 outgrows that (today r0 <= 44) must not
    // publish the threshold of a lower rank -- it falls back to the exhaustive head instead (search_core computes r0 from the same expression)
    const int64_t sample_rows = ((ix->n_tiles + S - 1) / S) * TILE_ROWS;
    if (adaptive_rank(r_eff, (double)TILE_ROWS / (double)sample_rows) <= BOOT_M) return 0;
  }
  return ix->first_rows > 0 ? std::max<int64_t>(ix->first_rows, 2 * (int64_t)r_eff) : std::max<int64_t>(S >= 2 ? 256 : 512, 2 * (int64_t)r_eff);
}
static int search_core(dhr_index* ix, Workspace& w, const dhr_query_batch* qb, int k, int depth, Timer& tm,
                       dhr_search_stats& st, hipStream_t s, int stage = 0, const float* tau_ext = nullptr) {
  int rc;
  // stage 4 (dhr_search_pre): query preparation, phase 0 and the FIRST part of the sampled run, then stop -- the shards exchange their best sample
  // scores seen so far; stage 5 (dhr_search_begin_rest): the rest of the sampled run from the threshold they agreed on, then as stage 1.
  const bool fresh = stage == 0 || stage == 1 || stage == 4;       // the call brings the query batch (else: resumed from ix->pend)
  const int Q = !fresh ? ix->pend.Q : qb->n_queries;
  const bool gate = !fresh ? ix->pend.gate
                           : (ix->d_dlr > 0 && qb->index != nullptr && qb->index_dtype != DHR_IDX_NONE);   // else plain IP
  const int64_t n = ix->n_rows;
  const int64_t group_rows = (int64_t)DOC_GROUP * TILE_ROWS;
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
      static const int pre_one = getenv("DHR_PRE_ONE") ? atoi(getenv("DHR_PRE_ONE")) : 0;      // A/B: the first part of the sample as ONE phase behind the bootstrap
      if (pre_one) chunk0 = round_up(n_hi, DOC_GROUP);
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

extern "C" int dhr_search(dhr_index* ix, const dhr_query_batch* qb, int32_t k, float* out_scores, int64_t* out_rows,
                          int32_t out_mem_kind, void* stream) try {
  dhr::alloc_checkpoint();
  int rc = check_queries(ix, qb);
  if (rc) return rc;
  if (k <= 0) return set_error(DHR_ERR_INVALID, "k must be > 0");
  if (k > (1 << 20)) return set_error(DHR_ERR_UNSUPPORTED, "k > 1048576 is not supported");      // k > 16384: global-memory merge (select_global.hip)
  if (!out_scores || !out_rows) return set_error(DHR_ERR_INVALID, "null output pointer");
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
int pq_check(const void* a, const void* b, int64_t n, int d, int M, int64_t ld) {
  if (!a || !b) return set_error(DHR_ERR_INVALID, "null pointer");
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
  int rc = pq_check(values, codebooks, n, d, M, ld);
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
  int rc = pq_check(values, codebooks, n, d, M, ld);
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
  int rc = pq_check(codes, codebooks, n, d, M, ld_out);
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
