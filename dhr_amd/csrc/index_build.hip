// Index handles of libdhr_hip.so (round 6: split out of api.hip): build (dhr_index_create), parameters and info, the index file.
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include "dhr_state.h"

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
      if (value < DOC_GROUP * TILE_ROWS || value > (1 << 22)) return set_error(DHR_ERR_INVALID, "cand_cap must be in [1024, 4194304]");      // (>= the rows of a minimum chunk, as list_stride below)
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
      // (>= the rows of the smallest chunk the host-driven controller can fall back to, DOC_GROUP tiles: a list that cannot hold every row of it could
      // overflow with nothing left to halve -- until round 6 the range began at 256, and such a chunk lost its entries beyond the stride SILENTLY
      // on corpora too small for the sampled controller; found by tools/stress_modes.py)
      if (value != 0 && (value < DOC_GROUP * TILE_ROWS || value > (1 << 22) || value % 256))
        return set_error(DHR_ERR_INVALID, "list_stride must be 0 (default) or a multiple of 256 in [1024, 4194304]");
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

