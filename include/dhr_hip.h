/*
 * dhr_hip.h -- C ABI of libdhr_hip.so: MI355X (gfx950) brute-force dense-hybrid retrieval.
 *
 * The reference (castorini/dhr, /root/reference) has NO native/FFI interface for this path: the
 * hot path is Python calling torch ops (retrieval/gip_retrieval.py:60-165).  This header is the
 * boundary a maintainer would bind instead of those torch calls (ctypes stub: INTEGRATION.md).
 * Each entry point names the reference code it replaces.  Plain pointers and sizes only; no C++
 * exceptions, no exit() cross this boundary; every function returns 0 or a negative dhr_status.
 *
 * Threading: one handle is used by one host thread at a time; distinct handles (e.g. one per
 * device / per rank) may be used concurrently.
 *
 * Streams: every entry point works on the stream it is given.  dhr_search ENQUEUES its first attempt (sampled run, main pass, refine /
 * rescoring / select of every chunk: no host read-backs between the phases since round 3) with two small host reads on that stream: 32
 * bytes after the sampled run (the fullest candidate lists, which size the chunks of the main pass; DHR_PARAM_ASYNC_CONTROLLER = 1 drops
 * it for a fixed plan) and, at the end, the number of queries whose verification failed (a list overflowed, or the sampled threshold came
 * out too high -- adversarial row orders), because those are redone by a host-driven controller that is exact for any input.  It
 * returns with the stream idle.  Corpora too small to sample (below ~33 k rows) use the host-driven controller throughout.
 * dhr_search_begin / dhr_search_finish and dhr_search_sharded over RCCL enqueue everything up to ONE host read (the count of queries to repair).
 */
#ifndef DHR_HIP_H
#define DHR_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DHR_VERSION 105 /* 0.1.5: exception barrier on every entry point (DHR_ERR_NOMEM), a failed sharded step is collective (DHR_ERR_PEER), dhr_abi_size, dhr_host_shard::struct_size, dhr_comm_abort no longer frees the handle, counts + list prefixes in ONE all-gather; 0.1.4: dhr_search_pre / dhr_search_pre_ranks / dhr_search_begin_rest (first agreement of the sharded search in two rounds), dhr_host_shard::pre_ranks / pre / begin_rest, DHR_PARAM_LIST_STRIDE; 0.1.3: dhr_comm_info / dhr_comm_abort */

typedef enum dhr_status {
  DHR_OK = 0,
  DHR_ERR_INVALID = -1,     /* bad argument (NULL, negative size, unsupported dtype ...) */
  DHR_ERR_UNSUPPORTED = -2, /* legal input the kernels do not cover (see dhr_last_error) */
  DHR_ERR_HIP = -3,         /* a HIP runtime call failed (no device, OOM, launch failure) */
  DHR_ERR_INTERNAL = -4,
  DHR_ERR_NOMEM = -5,       /* a host allocation failed (std::bad_alloc caught at the boundary); the handle stays usable */
  DHR_ERR_PEER = -6         /* sharded search: ANOTHER rank failed in this step (its status travelled with the step's all-gathers); every rank
                               returns from the same step with an error -- the failing rank with its own status, the others with this one */
} dhr_status;

typedef enum dhr_idx_dtype { DHR_IDX_NONE = 0, DHR_IDX_U8 = 1, DHR_IDX_I8 = 2, DHR_IDX_I16 = 3 } dhr_idx_dtype;
typedef enum dhr_val_dtype { DHR_VAL_F16 = 0, DHR_VAL_F32 = 1 } dhr_val_dtype;
typedef enum dhr_mem_kind { DHR_MEM_HOST = 0, DHR_MEM_DEVICE = 1 } dhr_mem_kind;

typedef struct dhr_index dhr_index; /* opaque */

/* One corpus shard in the reference's record layout (3-list pickle [value, index, ids]:
 * tevatron/driver/encode.py:165-170,203-204; merged by retrieval/index.py:26-47):
 *   value  fp16 [n_rows, d_dlr + d_cls] row-major, first d_dlr columns = densified lexical (gated),
 *          the rest = dense [CLS] (ungated);
 *   index  uint8 / int8 / int16 [n_rows, d_dlr], or NULL for dense-only models (then d_dlr = 0).
 * The library copies and re-lays the data into its own device layout (MFMA-fragment tiles); the
 * caller keeps ownership of the inputs and may free them after dhr_index_create returns. */
typedef struct dhr_index_desc {
  int32_t device;      /* HIP device ordinal */
  int32_t mem_kind;    /* dhr_mem_kind of value/index */
  int64_t n_rows;      /* rows of THIS shard (gip_retrieval.py:292-306 slice) */
  int32_t d_dlr;       /* gated columns == --emb_dim when index != NULL, else 0.  Any width: where it is not a multiple of 8 the library
                          appends zero slices to its own copies (of the corpus and of every query batch) */
  int32_t d_cls;       /* ungated columns */
  const void* value;   /* fp16 */
  int64_t ld_value;    /* elements between consecutive rows (>= d_dlr + d_cls) */
  const void* index;   /* or NULL */
  int32_t index_dtype; /* dhr_idx_dtype */
  int32_t idx_buckets; /* index buckets per gated slice in the bound GEMM operands (0 = default: 2, on the 2:4 sparse matrix cores; 1 = ungated bound) */
  int64_t ld_index;    /* elements between consecutive rows (>= d_dlr) */
  int64_t row_offset;  /* global row id of local row 0; added to every returned row */
} dhr_index_desc;

/* A batch of queries, same record layout as the corpus (value may also be fp32: the reference's
 * CPU path converts queries to fp32 and scales the CLS tail by --lamda in fp32,
 * gip_retrieval.py:275-283 -- hand that fp32 array over unchanged). */
typedef struct dhr_query_batch {
  int32_t n_queries;
  int32_t mem_kind;
  const void* value;   /* [n_queries, d_dlr + d_cls] */
  int32_t value_dtype; /* dhr_val_dtype */
  int32_t index_dtype; /* dhr_idx_dtype; NONE iff the index was built without an index array */
  int64_t ld_value;
  const void* index;   /* [n_queries, d_dlr] or NULL */
  int64_t ld_index;
} dhr_query_batch;

/* Counters of the last dhr_search on a handle (what the phase controller did and how long the
 * kernels ran, measured with hipEvents on the search stream). */
typedef struct dhr_search_stats {
  int64_t n_rows, n_queries, k;
  int32_t phases;            /* bound-GEMM launches */
  int32_t overflow_retries;  /* phases re-run in halves because a candidate list filled */
  int64_t candidates_bound;  /* pairs that passed the bound filter (sum over queries) */
  int64_t candidates_exact;  /* pairs rescored exactly (incl. the dense first phase) */
  int64_t gemm_rows;         /* corpus rows pushed through the bound GEMM */
  int64_t sample_fallback_queries; /* queries redone exactly because the sampled threshold was too high */
  double gemm_ms, refine_ms, rescore_ms, select_ms, prep_ms, total_ms;
  double gemm_flops;         /* 2 * Q_pad * rows * K_tile actually issued by the bound GEMM (padding, bucket split) */
  double gemm_flops_alg;     /* algorithmic: 2 * Q * rows * (d_dlr + d_cls), summed over the launches */
} dhr_search_stats;

/* Tunables (dhr_index_set_param). */
typedef enum dhr_param {
  DHR_PARAM_CAND_CAP = 1,     /* per-query candidate list capacity (entries) */
  DHR_PARAM_FIRST_ROWS = 2,   /* rows scored exhaustively to seed the thresholds (>= k enforced) */
  DHR_PARAM_PROFILE = 3,      /* 1: record per-kernel hipEvent timings into dhr_search_stats */
  DHR_PARAM_MAX_GROWTH = 4,   /* max (next chunk rows) / (rows seen), in 1/16ths (default 32 = 2x) */
  DHR_PARAM_MAIN_CHUNKS = 7,  /* lower limit of the number of main-pass chunks (the controller adds chunks so that the candidate lists fit) */
  DHR_PARAM_AUX_CUS = 9,      /* main pass: confine refine / rescoring / select to this many CUs (multiple of 8, spread over the XCDs; 0 = no CU mask; default: 128 for dense-only indexes, no mask for gated ones) */
  DHR_PARAM_GEMM_EXCLUSIVE = 10, /* with AUX_CUS: 1 = run the bound GEMM on the other CUs only */
  DHR_PARAM_OVERLAP_AUX = 11,  /* 0: refine / rescoring / select of a chunk run after its GEMM on the same stream; 1: beside the GEMM of the next chunk on a CU-masked stream; -1 (default): 1 */
  DHR_PARAM_PROGRESSIVE_THR = 8, /* later main-pass chunks filter with 1: the running exact k-th best; 2 (default): additionally the rank extrapolated from the scattered fraction of the corpus seen so far (main pass in a scattered tile order; a query whose extrapolation was too high fails the final verification and is redone); 0: the sampled threshold only */
  DHR_PARAM_GEMM_VARIANT = 6, /* bound-GEMM kernel of the 2:4 layout: 3 = 12 waves (producer / consumer), 4 = 4 waves with 128 x 128 wave tiles, 5 = 8 waves with 128 x 64 wave tiles (default), 6 = 5, and on gated_i8 indexes persistent workgroups (gemm_g8p.hip: the same results and, the kernel being held by the package power cap, the same time in fewer cycles -- DESIGN.md section 4b) */
  DHR_PARAM_ASYNC_CONTROLLER = 13, /* the first attempt of a sampled search ENQUEUES its phases without reading list lengths back between them.
                                     2 (default): one 32-byte read after the sampled run (the fullest lists: how many chunks the main pass
                                     needs) + ONE at the end (the number of queries whose verification failed); 1: the final read only (fixed
                                     chunk plan; what the staged / sharded entry points always do); 0: the host-driven controller (reads the list
                                     lengths back after every phase).  DHR_ASYNC overrides. */
  DHR_PARAM_SAMPLE_SHARE = 12, /* staged search only (dhr_search_begin / finish): the number of shards the sampled threshold is agreed between; a shard then
                                 reports its r / shards + 5 sqrt(r / shards) + 4 best sample scores (dhr_search_sample_rank) instead of all r
                                 (dhr_search_union_rank), and plans its lists for its SHARE of k: a shard that turns out to hold far more of the top-k
                                 than k / shards reports count = -1 for those queries in dhr_search_finish (the caller's repair step, which
                                 dhr_search_sharded* run themselves, redoes them with local thresholds).  dhr_search_sharded[_local] set the
                                 parameter themselves.  Default 1. */
  DHR_PARAM_LIST_STRIDE = 14, /* entries every query owns in the uniform part of the bound-candidate lists (default 0 = 32 768); what a hot query needs
                                 beyond that comes from a shared arena planned on the device (two-tier lists: the workspace of an 8.8 M-row index is
                                 ~7 GB instead of 34).  Takes effect for indexes with a refine level whose planned list depth exceeds it; tests
                                 force a small stride to exercise the second tier on small inputs.  0 or a multiple of 256 in [1024, 4194304] */
  DHR_PARAM_SAMPLE_PERIOD = 5 /* every S-th corpus tile seeds the thresholds (default 32; 0 = plain streaming) */
} dhr_param;

/* Process-wide options read by dhr_index_create (dhr_set_option). */
typedef enum dhr_option {
  DHR_OPT_DENSE_I8 = 1, /* int8 image of the UNGATED columns in the bound GEMM (v_mfma_i32_32x32x32_i8, twice the fp16 instruction's
                           columns per issue; the quantisation error is paid by the filter margin, results stay exact):
                           -1 (default) = gated indexes that also carry ungated columns, and dense-only shards of at least 1 000 000 rows whose
                           measured quantisation error is small against the spread of their scores (sqrt(d_cls) x max row error / max row norm
                           <= 0.45); 0 = never, 1 = every dense-only index too.
                           The environment variable DHR_DENSE_I8 overrides it. */
  DHR_OPT_GATED_I8 = 2  /* int8 image of the GATED columns too (v_smfmac_i32_32x32x64_i8: the 2:4 instruction on int8 operands, 32 slices
                           per issue; values rounded UP per column step, so the bound stays a bound and results stay exact):
                           -1 (default) = shards of at least 1 000 000 rows (4 000 000 where the ungated half is narrower than half the
                           gated one), where the cheaper GEMM outweighs the extra candidates the looser bound lets through (below
                           that the refine / rescoring work dominates the step); 1 = wherever the layout
                           allows it (two buckets, d_dlr a multiple of 64, the ungated half an int8 image or absent); 0 = keep the fp16
                           image.  The environment variable DHR_GATED_I8 (-1 / 0 / 1) overrides it. */
} dhr_option;
int dhr_set_option(int32_t option, int64_t value);
/* Read-only facts about a built index (dhr_index_get_info). */
typedef enum dhr_info {
  DHR_INFO_DENSE_I8 = 1,      /* 1 if the ungated stages are int8 images */
  DHR_INFO_I8_SCALE = 2,      /* corpus-side scale: max |ungated value| / 127 */
  DHR_INFO_I8_ROW_ERR = 3,    /* max over rows of || d - scale * q8(d) ||   (ungated part) */
  DHR_INFO_I8_ROW_NORM = 4,   /* max over rows of || scale * q8(d) || */
  DHR_INFO_ROW_NORM_MAX = 5,  /* max over rows of || row || */
  DHR_INFO_TILE_BYTES = 6,    /* bytes of the bound-GEMM operand images */
  DHR_INFO_GATED_I8 = 7,      /* 1 if the gated stages are int8 2:4 images */
  DHR_INFO_GEMM_KERNEL = 8    /* the bound-GEMM kernel the handle's latest search launched (0: none yet): 3 / 4 gemm_filter_wx_kernel<2> / <4>
                                 (fp16 2:4 image, 8 / 4 waves), 5 gemm_filter_g8_kernel (integer operands, 8 waves), 6 the same with persistent workgroups (A/B
                                 builds only); 1 and 2 named the two kernels retired in version 105 and are never reported */
} dhr_info;
int dhr_index_get_info(const dhr_index* index, int32_t what, double* out);

int dhr_version(void);
/* sizeof of the four public structs as this library was compiled: { dhr_index_desc, dhr_query_batch, dhr_search_stats,
 * dhr_file_info }.  A binding compares them with its own declarations at load time (a stale library whose struct layout
 * differs would otherwise corrupt memory silently). */
void dhr_abi_sizes(int32_t out[4]);
/* The same, one struct at a time (new structs get an id here instead of a longer array): returns sizeof as compiled, or a negative status. */
enum { DHR_ABI_INDEX_DESC = 0, DHR_ABI_QUERY_BATCH = 1, DHR_ABI_SEARCH_STATS = 2, DHR_ABI_FILE_INFO = 3, DHR_ABI_HOST_SHARD = 4 };
int32_t dhr_abi_size(int32_t which);
/* Message of the last failure on the calling thread ("" if none). */
const char* dhr_last_error(void);

/* Replaces the corpus half of main() (gip_retrieval.py:289-315: slice, astype, .cuda). */
int dhr_index_create(const dhr_index_desc* desc, dhr_index** out);
void dhr_index_destroy(dhr_index* index);
int dhr_index_set_param(dhr_index* index, int32_t param, int64_t value);
/* Device bytes held by the handle (index + workspaces). */
int64_t dhr_index_device_bytes(const dhr_index* index);

/* Index file.  The reference keeps ONE monolithic pickle that every run unpickles,
 * casts to fp32 and copies to the GPU (gip_retrieval.py:289-315).  dhr_index_save writes the corpus of a built
 * index as page-aligned raw sections of one file -- row-major fp16 values (rows padded to a multiple of 64
 * columns), row-major slice indices -- plus an opaque caller blob (the docid list); dhr_index_load mmaps the
 * file and streams it to the device through the ordinary ingest path: no unpickling, no host copies.  The
 * device images (operand tiles, refine lists) are NOT stored: re-tiling from device memory costs 0.06 s per
 * 2 M rows, the images would double the file.  row_offset < 0 keeps the stored value. */
typedef struct dhr_file_info {
  int64_t n_rows, row_offset;
  int32_t d_dlr, d_cls, index_dtype, idx_buckets;
  uint32_t file_version, reserved;
  int64_t payload_bytes;           /* value + index bytes in the file */
  int64_t blob_offset, blob_bytes; /* the caller blob: plain file bytes [blob_offset, blob_offset + blob_bytes) */
} dhr_file_info;
int dhr_index_save(const dhr_index* index, const char* path, const void* blob, int64_t blob_bytes);
int dhr_index_file_info(const char* path, dhr_file_info* out);
int dhr_index_load(const char* path, int32_t device, int64_t row_offset, dhr_index** out);

/* Replaces the query loop of GIP_retrieval (brute force, gip_retrieval.py:115-126) and of
 * IP_retrieval (:70-79) for ALL queries of the batch at once:
 *   score[q][n] = sum_{j<d_dlr} [c_idx[n][j]==q_idx[j]] * c_val[n][j]*q_val[j] + sum_{c} c_val*q_val
 * and returns per query the k best rows, best first (score desc, row asc on exact ties).
 *   out_scores [n_queries, k] fp32, out_rows [n_queries, k] int64 GLOBAL rows (row_offset added);
 *   if k > n_rows the tail is (-inf, -1)   (the reference raises / truncates: SURVEY appendix).
 * Scores are the exactly-rounded value of the gated inner product (fp64 accumulation of exact
 * products), i.e. within fp32 summation noise of the reference's einsum.
 * out_mem_kind says where out_scores/out_rows live.  stream is a hipStream_t (NULL = default). */
int dhr_search(dhr_index* index, const dhr_query_batch* queries, int32_t k, float* out_scores,
               int64_t* out_rows, int32_t out_mem_kind, void* stream);

/* Two-stage approximate GIP, both stages on the device (gip_retrieval.py:128-156):
 *   stage 1 = dhr_search(stage1, k1): the caller has restricted the batch the way the reference does -- values
 *             <= theta zeroed (:130-136), or index = NULL for the ungated --IP first stage (:139);
 *   stage 2 = exact gated inner product of the FULL batch on exactly those k1 rows (:144-146), top-k of that
 *             (score desc, row asc; fewer than k valid rows -> (-inf, -1) padding).
 * Same n_queries in both batches; 0 < k <= k1 <= 1048576 (above 16384 the global-memory merge, slower). */
int dhr_search_rerank(dhr_index* index, const dhr_query_batch* stage1, const dhr_query_batch* full, int32_t k1, int32_t k,
                      float* out_scores, int64_t* out_rows, int32_t out_mem_kind, void* stream);

/* Staged form of dhr_search for the row-sharded path (one handle per shard / rank).  After
 * dhr_search_begin every shard holds the r best exact scores of its SAMPLE per query (r =
 * dhr_search_sample_rank, the same on equally sized shards; 0 = the shard is too small to sample and
 * begin already ran the whole search).  The caller gathers the [n_queries, r] score blocks of all
 * shards, takes the r-th best of the union per query as the common threshold tau_hat and hands it to
 * dhr_search_finish, which runs the main pass with it and returns the shard's top-k plus, per query,
 * how many of its rows reach tau_hat (-1: a candidate list overflowed).  If the counts of all shards
 * sum to >= k (and none is -1) the union of the shard lists contains the global top-k; otherwise
 * those queries are redone with dhr_search.  Replaces one process per shard + merge.result.py.
 * out_sample_scores_dev / tau_hat_dev / out_count_dev are DEVICE pointers. */
int32_t dhr_search_sample_rank(const dhr_index* index, int32_t k);
/* The rank of the union of the shards' samples that defines the common threshold (== dhr_search_sample_rank unless
 * DHR_PARAM_SAMPLE_SHARE > 1): the caller merges the gathered [shards, Q, sample_rank] lists and takes the
 * min(union_rank, shards * sample_rank)-th best per query. */
int32_t dhr_search_union_rank(const dhr_index* index, int32_t k);
int dhr_search_begin(dhr_index* index, const dhr_query_batch* queries, int32_t k, float* out_sample_scores_dev, void* stream);
int dhr_search_finish(dhr_index* index, const float* tau_hat_dev, float* out_scores, int64_t* out_rows,
                      int32_t* out_count_dev, int32_t out_mem_kind, void* stream);
/* Optional step between the two: a SECOND agreement on the thresholds.  dhr_search_mid runs the first slice of the main pass (1/8 of it,
 * in scattered order) with the thresholds of the first agreement and leaves the shard's r_local best scores seen so far -- head, sample and
 * slice -- in out_scores_dev [n_queries, r_local] (r_local: dhr_search_mid_ranks, or the largest over the shards where their sizes differ).
 * The shards have then seen the fraction f of their rows; the caller gathers the blocks and
 * takes the r_union-th best of the union per query (r_union = k f + 6 sigma + 4: it lies below the final k-th best), and hands
 * max(first threshold, that) to dhr_search_finish, which runs the rest of the pass with it and counts against it.  A 1/8 shard of the 8.8 M-row
 * benchmark then rescores ~360 instead of ~490 rows per query in its main pass.  dhr_search_mid_ranks returns r_local (also through
 * out_local; 0: this index has no such step -- it is too small -- and dhr_search_finish follows dhr_search_begin directly). */
int32_t dhr_search_mid_ranks(const dhr_index* index, int32_t k, int32_t* out_local, int32_t* out_union);
int dhr_search_mid(dhr_index* index, const float* tau_hat_dev, int32_t r_local, float* out_scores_dev, void* stream);
/* Optional: dhr_search_begin in TWO calls with an agreement in between (round 5).  dhr_search_pre prepares the batch and streams the first
 * part of the shard's sample (1/8 of it), leaving the shard's r_local best scores seen so far in out_scores_dev [n_queries, r_local]
 * (dhr_search_pre_ranks; 0: the sample is too small to split -- use dhr_search_begin).  The caller gathers the blocks, takes the r_union-th
 * best of the union per query (it lies below the union sample's final rank-r score: the same argument as inside a sampled run) and hands it to
 * dhr_search_begin_rest, which streams the rest of the sample filtering at it and leaves the handle exactly where dhr_search_begin does
 * (out_sample_scores_dev [n_queries, dhr_search_sample_rank]).  Eight shards that each ran their sampled run from nothing rescored 3.2 k rows per
 * query between them where the unsharded search's one run rescores 0.7 k. */
int32_t dhr_search_pre_ranks(const dhr_index* index, int32_t k, int32_t* out_local, int32_t* out_union);
int dhr_search_pre(dhr_index* index, const dhr_query_batch* queries, int32_t k, int32_t r_local, float* out_scores_dev, void* stream);
int dhr_search_begin_rest(dhr_index* index, const float* tau_dev, float* out_sample_scores_dev, void* stream);

/* Exact gated inner product of each query against m given rows (stage 2 of --rerank,
 * gip_retrieval.py:144-146 / :207-208).  rows [n_queries, m] int64 GLOBAL rows (row < 0 -> -inf).
 * out_scores [n_queries, m].  Pointers live in mem_kind memory. */
int dhr_score_rows(dhr_index* index, const dhr_query_batch* queries, int32_t m, const int64_t* rows,
                   float* out_scores, int32_t mem_kind, void* stream);

/* The multi-shard reduce (retrieval/merge.result.py:22-42 without the text round trip): per query,
 * the k_out best of n_lists*k_in (score, row) pairs, best first (score desc, row asc); entries with
 * row < 0 are padding.  in_scores/in_rows are [n_queries, n_lists*k_in] (each query's lists
 * concatenated).  All four pointers are DEVICE pointers on `device`.  Any n_in (up to 16 384 entries per query in one workgroup's LDS,
 * beyond that two stable segmented sorts through global memory; n_queries * n_in < 2^31). */
int dhr_merge_topk(int32_t device, int32_t n_queries, int32_t n_in, const float* in_scores,
                   const int64_t* in_rows, int32_t k_out, float* out_scores, int64_t* out_rows, void* stream);
/* Same reduce on HOST pointers (no GPU needed; used by the CPU/gloo tests of the sharded path). */
int dhr_merge_topk_host(int32_t n_queries, int32_t n_in, const float* in_scores, const int64_t* in_rows,
                        int32_t k_out, float* out_scores, int64_t* out_rows);

/* The same reduce for lists that are already SORTED, in the layout an all-gather of the per-shard results
 * leaves them in: in_scores / in_rows are [n_lists, n_queries, list_len]; every list is in output order
 * (score desc, row asc) with its padding (row < 0) at the tail -- what dhr_search / dhr_search_finish write.
 * No sort is run: every entry finds its output rank by binary searches in the other lists.  in_rows may be
 * NULL (scores only, e.g. the shards' sample scores of dhr_search_begin; ties keep list order; out_rows is
 * ignored).  (n_lists*list_len + k_out)*12 B (4 B without rows) must fit 160 KiB, n_lists <= 64.  DEVICE pointers
 * on `device`. */
int dhr_merge_topk_lists(int32_t device, int32_t n_queries, int32_t n_lists, int32_t list_len, const float* in_scores,
                         const int64_t* in_rows, int32_t k_out, float* out_scores, int64_t* out_rows, void* stream);
/* Host twin (no sortedness needed: it sorts). */
int dhr_merge_topk_lists_host(int32_t n_queries, int32_t n_lists, int32_t list_len, const float* in_scores,
                              const int64_t* in_rows, int32_t k_out, float* out_scores, int64_t* out_rows);
/* Fused densify, the step immediately upstream of the index (tevatron/DHR/utils.py:5-22 and the
 * casts of tevatron/driver/encode.py:155-158,180-183).  lexical is [batch, vocab] (value_dtype DHR_VAL_F32 or DHR_VAL_F16,
 * row stride ld); the columns [remove_dims, vocab) are viewed as [n_groups, dims] and
 *   out_value[b][j] = max_g lexical[b][remove_dims + g*dims + j]     (out_value_dtype: DHR_VAL_F16 rounds to fp16, DHR_VAL_F32 exact)
 *   out_index[b][j] = the FIRST g attaining it                        (index_dtype DHR_IDX_U8, needs n_groups <= 256, or DHR_IDX_I16)
 * written with row strides ld_value / ld_index, i.e. directly into the value / index arrays of an index record (the
 * CLS columns follow at out_value + dims).  DHR_ERR_INVALID when (vocab - remove_dims) is not a multiple of dims
 * (the reference raises ValueError).  All three arrays live in mem_kind memory. */
int dhr_densify(int32_t device, int32_t mem_kind, const void* lexical, int32_t value_dtype, int64_t ld, int64_t batch, int32_t vocab,
                int32_t remove_dims, int32_t dims, void* out_value, int32_t out_value_dtype, int64_t ld_value, void* out_index,
                int32_t index_dtype, int64_t ld_index, void* stream);

/* Product quantiser for the first stage of --PQIP.  The reference calls faiss
 * IndexPQ(d, M = 64, nbits = 8, METRIC_INNER_PRODUCT) (retrieval/quantize_index.py:27-37, gip_retrieval.py:167-231); faiss is not
 * part of the reference tree, so these restate its published algorithm (per-subspace Lloyd k-means, nearest-centroid codes, ADC
 * inner-product scores) -- codebooks differ from faiss's (different initialisation and sub-sampling): PARITY UNPINNED, judged
 * by recall against the exact search.  All arrays live in mem_kind memory; values are the fp16 rows of an index record
 * (row stride ld, d = M * dsub columns used), codebooks fp32 [M][256][dsub], codes uint8 [n][M].
 *   dhr_pq_train : k-means on at most max_points evenly spaced rows, `iters` Lloyd iterations; out_error (host, may be NULL)
 *                  receives the mean squared quantisation error per training row after the last assignment.
 *   dhr_pq_encode: nearest centroid per subspace (first minimum).
 *   dhr_pq_decode: fp16 reconstruction [n][ld_out]; an exact inner-product search over it (dhr_index_create with index = NULL,
 *                  then dhr_search) returns exactly the ADC ranking, on the matrix cores. */
int dhr_pq_train(int32_t device, int32_t mem_kind, const void* values_f16, int64_t ld, int64_t n, int32_t d, int32_t M, int32_t iters,
                 int64_t max_points, float* codebooks, double* out_error, void* stream);
int dhr_pq_encode(int32_t device, int32_t mem_kind, const void* values_f16, int64_t ld, int64_t n, int32_t d, int32_t M,
                  const float* codebooks, uint8_t* codes, void* stream);
int dhr_pq_decode(int32_t device, int32_t mem_kind, const uint8_t* codes, int64_t n, int32_t d, int32_t M, const float* codebooks,
                  void* out_values_f16, int64_t ld_out, void* stream);
/* The same three with `--n_bits` (quantize_index.py:22,29: faiss.IndexPQ(d, M, nbits)): 2^nbits centroids per sub-quantiser, 1 <= nbits <= 8,
 * codebooks [M][2^nbits][d / M]; codes stay one byte per sub-quantiser on the device and in these calls (faiss' bit-packed rows are
 * written / read by the index file code, dhr_amd/retrieval/quantize_index.py).  The functions above are nbits = 8. */
int dhr_pq_train_nbits(int32_t device, int32_t mem_kind, const void* values_f16, int64_t ld, int64_t n, int32_t d, int32_t M, int32_t nbits,
                       int32_t iters, int64_t max_points, float* codebooks, double* out_error, void* stream);
int dhr_pq_encode_nbits(int32_t device, int32_t mem_kind, const void* values_f16, int64_t ld, int64_t n, int32_t d, int32_t M, int32_t nbits,
                        const float* codebooks, uint8_t* codes, void* stream);
int dhr_pq_decode_nbits(int32_t device, int32_t mem_kind, const uint8_t* codes, int64_t n, int32_t d, int32_t M, int32_t nbits,
                        const float* codebooks, void* out_values_f16, int64_t ld_out, void* stream);

/* The TREC run file (gip_retrieval.py:333-342): for every query, in order, one line "qid Q0 docid rank score run_name" per result row,
 * rank = 1-based position in the query's list (rows < 0 are padding of a short list and take no rank), lines whose docid equals the
 * query id skipped (their rank stays unused, as in the reference), score printed as Python prints float(score): the shortest digit
 * string that round-trips the DOUBLE, fixed notation for 1e-4 <= |x| < 1e16.  Ids are handed over as one byte blob each with
 * n + 1 offsets (id i = blob[off[i], off[i+1] - id_sep_bytes): id_sep_bytes = 1 for ids joined with one separator byte each); rows are [n_queries][k], row - row_base indexes the docid list; formatted on
 * n_threads host threads (0 = all, at most 64), written in query order.  dhr_format_float: the score formatting alone (returns the
 * length).  Host code only -- no device is touched. */
int dhr_write_trec(const char* path, int32_t append, int64_t n_queries, int64_t k, const char* qid_blob, const int64_t* qid_off,
                   const char* docid_blob, const int64_t* docid_off, int64_t n_docs, const int64_t* rows, int64_t row_base,
                   const float* scores, const char* run_name, int32_t id_sep_bytes, int32_t n_threads, int64_t* lines_out);
int dhr_format_float(double x, char* out, int32_t cap);

int dhr_get_stats(const dhr_index* index, dhr_search_stats* out);

/* Debug/test hook: the bound-GEMM scores U[q][row] for rows [row_lo,row_hi) of the shard, written
 * to a DEVICE buffer [n_queries, row_hi-row_lo] fp32 (U >= exact score; equal for dense-only). */
int dhr_debug_bound_scores(dhr_index* index, const dhr_query_batch* queries, int64_t row_lo, int64_t row_hi,
                           float* out_dev, void* stream);

/* Debug/test hook: the filter margins of a query batch (HOST buffer [n_queries] fp32): the bound GEMM guarantees
 * U[q][row] >= exact score - margin[q] for every row, which is what lets the filter drop rows without losing a top-k row. */
int dhr_debug_query_margins(dhr_index* index, const dhr_query_batch* queries, float* out_host, void* stream);

/* Test hook (host code only, no device needed): the corpus tile that position `seq` of a bound-GEMM launch maps to, computed with
 * the kernels' division-free arithmetic (out[0]) and with plain integer division (out[1]); the two must agree for every input. */
void dhr_debug_seq_to_tile(int64_t seq, int32_t map_mode, int32_t period, int64_t head, int64_t perm_mul, int64_t perm_n, int64_t out[2]);
/* Test hook of the exception barrier: arms a failure of the library's n-th HOST allocation from now (operator new inside the library only --
 * the process' allocator is not interposed; 0 disarms; the environment variable DHR_TEST_FAIL_ALLOC=n arms it at load time).  The entry
 * point in which it fires returns DHR_ERR_NOMEM.  Returns the number of host allocations the library has made so far. */
int64_t dhr_debug_fail_alloc(int64_t n);
/* Test hook: how many queries the calling thread's last dhr_search_sharded* call redid with local thresholds (its repair path: a
 * correct result either way, but each repaired query costs the step an extra pass over its query tile). */
int32_t dhr_debug_sharded_repairs(void);

/* Kernel-tuning hook: the bound GEMM alone over the whole shard with the filter closed; average
 * milliseconds per launch over `iters` launches and the flops one launch issues (padded sizes). */
int dhr_debug_gemm_time(dhr_index* index, const dhr_query_batch* queries, int32_t iters, double* ms_out,
                        double* flops_out, void* stream);

/* ---- Product-quantised first stage of --PQIP as an ADC scan.  Replaces
 * faiss.read_index(...) + IndexPQ.search(queries, agip_topk) (retrieval/gip_retrieval.py:170,202; built by
 * retrieval/quantize_index.py:27-37 as IndexPQ(d, M, nbits, METRIC_INNER_PRODUCT)).  faiss is not part of the reference tree: the
 * algorithm is restated from its publication, PARITY WITH FAISS IS UNPINNED.  The handle keeps the codes (ONE BYTE per
 * sub-quantiser and row: 64 B per row at M = 64) and the codebooks ([M][2^nbits][d/M] fp32, faiss' centroid order) on the
 * device; a search builds per-query lookup tables <q_m, c_m[j]> and scans the codes (HBM / LDS-gather bound), fused with the
 * running top-k threshold filter.  score(q, x) = sum_m LUT[q][m][code_m(x)] in fp32, m ascending.
 *   codes     [n][M] uint8, one code per byte (unpack files with nbits < 8 first)
 *   queries   values only ([n_queries][d], fp16 or fp32; index arrays are ignored: the first stage is ungated)
 *   dhr_pq_search    -> [n_queries][k] (score desc, row asc on exact ties; global rows; (-inf, -1) beyond the corpus), k <= 1048576 (above 16384 the global-memory merge)
 *   dhr_pq_adc_scores-> raw scores of rows [row_lo, row_hi), device memory [n_queries][row_hi - row_lo] (tests)
 *   dhr_pq_last_scan -> duration (ms, hipEvents) and code bytes read by the scan kernel launches of the last search */
typedef struct dhr_pq dhr_pq; /* opaque */
int dhr_pq_create(int32_t device, int32_t mem_kind, int64_t n, int32_t d, int32_t M, int32_t nbits, const float* codebooks,
                  const uint8_t* codes, int64_t row_offset, dhr_pq** out);
void dhr_pq_destroy(dhr_pq* pq);
int64_t dhr_pq_device_bytes(const dhr_pq* pq);
int dhr_pq_search(dhr_pq* pq, const dhr_query_batch* queries, int32_t k, float* out_scores, int64_t* out_rows, int32_t out_mem_kind,
                  void* stream);
int dhr_pq_adc_scores(dhr_pq* pq, const dhr_query_batch* queries, int64_t row_lo, int64_t row_hi, float* out_dev, void* stream);
int dhr_pq_last_scan(const dhr_pq* pq, double* ms, double* code_bytes);

/* ---- Row-sharded search: replaces the reference's --total_shrad / --shrad runs plus
 * retrieval/merge.result.py:22-42.  The corpus rows are split like gip_retrieval.py:292-306 (per = N // S, the last shard takes
 * the remainder; dhr_index_desc.row_offset = the shard's first global row), queries are replicated, and the result is the global
 * [n_queries, k] (scores, rows) -- bit-identical to the unsharded dhr_search of the whole corpus -- on EVERY rank.
 *
 * dhr_comm: an RCCL communicator (librccl is linked directly; all-gathers run on the caller's stream over xGMI).
 *   dhr_comm_unique_id   rank 0: a 128-byte id to hand to every rank (any out-of-band channel)
 *   dhr_comm_create      every rank: ncclCommInitRank on `device` (collective over the ranks)
 *   dhr_comm_wrap        adopt an existing ncclComm_t (not destroyed by dhr_comm_destroy)
 * dhr_search_sharded: one process per GPU; collective -- every rank calls it with the same batch and k.  Sequence: sampled
 *   pass -> all-gather of [Q, r] sample scores -> common per-query thresholds -> main pass -> all-gather of the per-query
 *   counts -> all-gather of the list prefixes [Q, kk] -> rank merge of the sorted lists; one host read at the end (number of
 *   queries whose lists were incomplete; those are redone with local thresholds).
 * dhr_search_sharded_local: one process, several shard handles (on one or more devices), no communicator -- the same control
 *   flow with the gathers done by device copies.  A device-resident query batch must be readable from every shard's device
 *   (same device, or peer access); host batches always work. */
typedef struct dhr_comm dhr_comm; /* opaque */
int dhr_comm_unique_id(void* out128, int32_t out_bytes);
int dhr_comm_create(const void* unique_id128, int32_t world, int32_t rank, int32_t device, dhr_comm** out);
int dhr_comm_wrap(void* nccl_comm, int32_t world, int32_t rank, int32_t device, dhr_comm** out);
/* A communicator whose all-gather is done by the CALLER over any transport (torch.distributed on gloo, MPI, pipes ...):
 *   allgather(user, send, recv, bytes) gathers `bytes` from every rank into recv = [world][bytes], rank order; HOST buffers
 *   (the library stages device blocks through pinned memory around the call); returns 0 on success.  Collective: every rank
 *   calls it the same number of times with the same sizes.  dhr_search_sharded then runs the same control flow as over RCCL.
 * Used where RCCL is not available between the ranks (several ranks on one GPU, CPU-only process groups). */
typedef int (*dhr_allgather_fn)(void* user, const void* send, void* recv, int64_t bytes);
int dhr_comm_create_callback(int32_t world, int32_t rank, int32_t device, dhr_allgather_fn allgather, void* user, dhr_comm** out);
void dhr_comm_destroy(dhr_comm* comm);
/* What the communicator really is, asked of the transport itself (a scaling record can then show that RCCL spanned N ranks):
 *   DHR_COMM_TRANSPORT 0 = RCCL, 1 = caller-supplied host transport;  DHR_COMM_WORLD / DHR_COMM_RANK / DHR_COMM_DEVICE: for an RCCL
 *   communicator ncclCommCount / ncclCommUserRank / ncclCommCuDevice of the ncclComm_t, else the values given at creation.
 * Returns the value, or a negative status. */
enum { DHR_COMM_TRANSPORT = 0, DHR_COMM_WORLD = 1, DHR_COMM_RANK = 2, DHR_COMM_DEVICE = 3 };
int dhr_comm_info(const dhr_comm* comm, int32_t what);
/* Way out of a bring-up that hangs (a peer never reached ncclCommInitRank / a collective): ncclCommAbort instead of ncclCommDestroy --
 * may be called from another host thread than the one blocked in the collective.  The handle is only marked dead (every later call on it
 * fails with DHR_ERR_INVALID): the thread that was blocked unwinds through code that still uses it, so the caller frees it with
 * dhr_comm_destroy once that thread has returned. */
void dhr_comm_abort(dhr_comm* comm);
/* The sharded control flow over a shard the CALLER implements in host memory (no device is touched): test / bring-up hook.  The
 * callbacks mirror the staged C ABI: sample_rank / union_rank (dhr_search_sample_rank with DHR_PARAM_SAMPLE_SHARE = share /
 * dhr_search_union_rank), begin (writes [n_queries, sample_rank] best sample scores, best first), finish (tau [n_queries] ->
 * [n_queries, k] sorted lists with (-inf, -1) tails + per-query counts of rows >= tau, -1 = incomplete), search (plain top-k).
 * All return 0 on success; all arrays are host memory; rows are global.  The query batch must be a host batch. */
typedef struct dhr_host_shard {
  uint32_t struct_size; /* sizeof(dhr_host_shard) as the CALLER compiled it: dhr_search_sharded_host rejects any other value (the struct has grown
                           between versions; a shorter caller struct would otherwise be read past its end) */
  uint32_t reserved;
  void* user;
  int32_t (*sample_rank)(void* user, int32_t k, int32_t share);
  int32_t (*union_rank)(void* user, int32_t k);
  int32_t (*begin)(void* user, const dhr_query_batch* queries, int32_t k, int32_t share, float* out_sample);
  int32_t (*finish)(void* user, const float* tau, float* out_scores, int64_t* out_rows, int32_t* out_count);
  int32_t (*search)(void* user, const dhr_query_batch* queries, int32_t k, float* out_scores, int64_t* out_rows);
  /* optional (both NULL: no second agreement): dhr_search_mid_ranks / dhr_search_mid of the shard */
  int32_t (*mid_ranks)(void* user, int32_t k, int32_t share, int32_t* out_local, int32_t* out_union);
  int32_t (*mid)(void* user, const float* tau, int32_t r_local, float* out_scores);
  /* optional (all NULL: begin in one call): dhr_search_pre_ranks / dhr_search_pre / dhr_search_begin_rest of the shard */
  int32_t (*pre_ranks)(void* user, int32_t k, int32_t share, int32_t* out_local, int32_t* out_union);
  int32_t (*pre)(void* user, const dhr_query_batch* queries, int32_t k, int32_t share, int32_t r_local, float* out_scores);
  int32_t (*begin_rest)(void* user, const float* tau, float* out_sample);
} dhr_host_shard;
int dhr_search_sharded_host(const dhr_host_shard* shard, int32_t world, int32_t rank, dhr_allgather_fn allgather, void* user,
                            const dhr_query_batch* queries, int32_t k, float* out_scores, int64_t* out_rows);
int dhr_search_sharded(dhr_index* shard, dhr_comm* comm, const dhr_query_batch* queries, int32_t k, float* out_scores,
                       int64_t* out_rows, int32_t out_mem_kind, void* stream);
int dhr_search_sharded_local(dhr_index** shards, int32_t n_shards, const dhr_query_batch* queries, int32_t k, float* out_scores,
                             int64_t* out_rows, int32_t out_mem_kind, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DHR_HIP_H */
