// Row-sharded search behind the C ABI: the corpus is split over ranks exactly like the reference's
// --total_shrad / --shrad (retrieval/gip_retrieval.py:292-306), queries are replicated, every shard searches its rows and the
// per-shard lists are reduced to the global top-k -- retrieval/merge.result.py:22-42 without the text files.
//
//   dhr_search_sharded        one process per GPU: this rank's shard + a dhr_comm (an RCCL communicator: librccl linked directly,
//                             all-gathers over xGMI)
//   dhr_search_sharded_local  one process, several shards (handles on one or more devices): the same control flow with the
//                             all-gather done by device copies -- also what the single-GPU tests drive
//
// Both run ONE implementation (sharded_core) over a "gather" callback.  Sequence (collectives on the caller's stream):
//   1. every shard runs its sampled pass (dhr_search_begin) and holds the r best exact sample scores per query;
//   2. all-gather of [Q, r] fp32; the r-th best of the union is the common threshold tau_q, so a shard collects only ITS SHARE
//      of the global top-k (rank merge of the sorted lists in place, dhr_merge_topk_lists without rows);
//   3. main pass with tau (dhr_search_finish) -> sorted per-shard lists + the count of rows reaching tau (-1: list overflow);
//   4. all-gather of the counts [Q] int32: a query is complete iff the union holds >= k rows, no shard overflowed and no shard's
//      share exceeds the gathered prefix; failures are flagged ON THE DEVICE (identically on every rank);
//   5. all-gather of the list prefixes [Q, kk] (kk = a fixed fraction of k by world size: no host read decides it) and the
//      rank merge of the sorted lists (dhr_merge_topk_lists) -> the global [Q, k];
//   6. ONE host read (the number of failed queries); failed queries (unrepresentative sample, skewed shards) are redone with
//      purely local thresholds (dhr_search on the sub-batch), gathered at full length and scattered into the result.
// Shards that cannot be sampled uniformly (tiny or unequal shards, k > rows of a shard) take the local-threshold path for
// the whole batch.  Results are bit-identical to the unsharded search (exact scores, ties by global row).
#include <rccl/rccl.h>

#include <algorithm>
#include <cstring>
#include <string>
#include <vector>

#include "dhr_internal.h"

struct dhr_comm {
  ncclComm_t comm = nullptr;
  bool owned = false;
  int world = 1, rank = 0, device = 0;
  void* arena = nullptr;          // grow-only device scratch
  size_t arena_bytes = 0;
};


namespace {

using namespace dhr;

#define SH_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return dhr_set_error_message(DHR_ERR_HIP, (std::string(#x) + ": " + hipGetErrorString(e_)).c_str()); } while (0)
#define SH_NCCL(x) do { ncclResult_t r_ = (x); if (r_ != ncclSuccess) return dhr_set_error_message(DHR_ERR_HIP, (std::string(#x) + ": " + ncclGetErrorString(r_)).c_str()); } while (0)
#define SH_TRY(x) do { int rc_ = (x); if (rc_ != DHR_OK) return rc_; } while (0)

__global__ void column_kernel(const float* __restrict__ in, int ld, int col, int n, float* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = in[(int64_t)i * ld + col];
}
// counts [world][Q] -> fail flag per query, compact list of failed query ids, their number
__global__ void fail_kernel(const int32_t* __restrict__ counts, int world, int n_queries, int k, int kk, int32_t* __restrict__ fail_ids,
                            int32_t* __restrict__ n_failed) {
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= n_queries) return;
  int64_t tot = 0;
  bool bad = false;
  for (int w = 0; w < world; ++w) {
    const int32_t c = counts[(int64_t)w * n_queries + q];
    if (c < 0 || c > kk) bad = true;
    tot += c > 0 ? c : 0;
  }
  if (bad || tot < k) fail_ids[atomicAdd(n_failed, 1)] = q;
}
__global__ void prefix_kernel(const float* __restrict__ s, const int64_t* __restrict__ r, int k, int kk, int n_queries, float* __restrict__ os,
                              int64_t* __restrict__ orow) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)n_queries * kk) return;
  const int q = (int)(i / kk), j = (int)(i - (int64_t)q * kk);
  os[i] = s[(int64_t)q * k + j];
  orow[i] = r[(int64_t)q * k + j];
}
__global__ void gather_rows_bytes_kernel(const char* __restrict__ src, int64_t stride, int row_bytes, const int32_t* __restrict__ ids, int n,
                                         char* __restrict__ dst) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)n * row_bytes) return;
  const int r = (int)(i / row_bytes), b = (int)(i - (int64_t)r * row_bytes);
  dst[i] = src[(int64_t)ids[r] * stride + b];
}
__global__ void scatter_result_kernel(const float* __restrict__ s, const int64_t* __restrict__ r, const int32_t* __restrict__ ids, int n, int k,
                                      float* __restrict__ os, int64_t* __restrict__ orow) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)n * k) return;
  const int f = (int)(i / k), j = (int)(i - (int64_t)f * k);
  os[(int64_t)ids[f] * k + j] = s[i];
  orow[(int64_t)ids[f] * k + j] = r[i];
}

struct Arena {                    // bump allocator over a grow-only device buffer (one per shard context)
  void** base; size_t* cap; size_t used = 0; int device; bool persist = true;
  std::vector<void*> spill;       // allocations beyond the arena's current size (freed at the end of the call; the arena grows for the next)
  size_t want = 0;
  void* get(size_t bytes) {
    bytes = (bytes + 255) & ~(size_t)255;
    want += bytes;
    if (used + bytes <= *cap) { void* p = (char*)*base + used; used += bytes; return p; }
    void* p = nullptr;
    if (hipMalloc(&p, bytes) != hipSuccess) return nullptr;
    spill.push_back(p);
    return p;
  }
  void finish() {
    for (void* p : spill) (void)hipFree(p);
    spill.clear();
    if (persist && want > *cap) { (void)hipFree(*base); *base = nullptr; *cap = 0; if (hipMalloc(base, want) == hipSuccess) *cap = want; }
  }
};

struct ShardCtx {
  dhr_index* ix;
  int device;
  hipStream_t stream;
  Arena* arena;
};

// all-gather of `bytes` per shard: send[i] (local shard i's block) -> recv[i] = [world][bytes] on local shard i's device
struct Gather {
  int world, n_local, first;      // local shards are ranks [first, first + n_local)
  dhr_comm* comm;                 // SPMD: n_local == 1
  int run(const std::vector<ShardCtx>& sh, const std::vector<const void*>& send, const std::vector<void*>& recv, size_t bytes) const {
    if (comm) {
      SH_NCCL(ncclAllGather(send[0], recv[0], bytes, ncclInt8, comm->comm, sh[0].stream));
      return DHR_OK;
    }
    // one process: make every block visible to every local shard (plain device copies; peer copies across devices)
    for (int i = 0; i < n_local; ++i) SH_HIP(hipStreamSynchronize(sh[i].stream));
    for (int d = 0; d < n_local; ++d) {
      SH_HIP(hipSetDevice(sh[d].device));
      for (int s = 0; s < n_local; ++s)
        SH_HIP(hipMemcpyAsync((char*)recv[d] + (size_t)s * bytes, send[s], bytes, hipMemcpyDefault, sh[d].stream));
    }
    for (int i = 0; i < n_local; ++i) SH_HIP(hipStreamSynchronize(sh[i].stream));
    return DHR_OK;
  }
};

int merge_gathered(const ShardCtx& c, int Q, int world, int L, const float* gs, const int64_t* gr, int k, float* os, int64_t* orow) {
  // sorted per-shard lists in [world, Q, L] layout: rank merge in the LDS when the lists of a query fit, else the general reduce
  if (world <= 64 && ((int64_t)world * L + k) * 12 <= 160 * 1024)
    return dhr_merge_topk_lists(c.device, Q, world, L, gs, gr, k, os, orow, c.stream);
  // [world, Q, L] -> [Q, world * L]
  float* ts = (float*)c.arena->get((size_t)Q * world * L * 4);
  int64_t* tr = (int64_t*)c.arena->get((size_t)Q * world * L * 8);
  if (!ts || !tr) return dhr_set_error_message(DHR_ERR_HIP, "out of device memory in the shard reduce");
  for (int w = 0; w < world; ++w) {
    SH_HIP(hipMemcpy2DAsync(ts + (size_t)w * L, (size_t)world * L * 4, gs + (size_t)w * Q * L, (size_t)L * 4, (size_t)L * 4, Q, hipMemcpyDeviceToDevice, c.stream));
    SH_HIP(hipMemcpy2DAsync(tr + (size_t)w * L, (size_t)world * L * 8, gr + (size_t)w * Q * L, (size_t)L * 8, (size_t)L * 8, Q, hipMemcpyDeviceToDevice, c.stream));
  }
  return dhr_merge_topk(c.device, Q, world * L, ts, tr, k, os, orow, c.stream);
}

// local thresholds for a (sub-)batch: every shard searches with k, full lists gathered and merged.  out_* [n_q, k] per local shard.
int local_path(const std::vector<ShardCtx>& sh, const Gather& g, const std::vector<dhr_query_batch>& qb, int k,
               const std::vector<float*>& out_s, const std::vector<int64_t*>& out_r) {
  const int nl = g.n_local, world = g.world, Q = qb[0].n_queries;
  std::vector<const void*> send_s(nl), send_r(nl);
  std::vector<void*> recv_s(nl), recv_r(nl);
  for (int i = 0; i < nl; ++i) {
    SH_HIP(hipSetDevice(sh[i].device));
    float* ls = (float*)sh[i].arena->get((size_t)Q * k * 4);
    int64_t* lr = (int64_t*)sh[i].arena->get((size_t)Q * k * 8);
    recv_s[i] = sh[i].arena->get((size_t)world * Q * k * 4);
    recv_r[i] = sh[i].arena->get((size_t)world * Q * k * 8);
    if (!ls || !lr || !recv_s[i] || !recv_r[i]) return dhr_set_error_message(DHR_ERR_HIP, "out of device memory in the sharded search");
    SH_TRY(dhr_search(sh[i].ix, &qb[i], k, ls, lr, DHR_MEM_DEVICE, sh[i].stream));
    send_s[i] = ls; send_r[i] = lr;
  }
  SH_TRY(g.run(sh, send_s, recv_s, (size_t)Q * k * 4));
  SH_TRY(g.run(sh, send_r, recv_r, (size_t)Q * k * 8));
  for (int i = 0; i < nl; ++i) {
    SH_HIP(hipSetDevice(sh[i].device));
    SH_TRY(merge_gathered(sh[i], Q, world, k, (const float*)recv_s[i], (const int64_t*)recv_r[i], k, out_s[i], out_r[i]));
  }
  return DHR_OK;
}

int prefix_len(int k, int world) {
  if (world <= 2) return k;
  const int kk = ((3 * k + world - 1) / world + 64 + 63) / 64 * 64;
  return std::min(k, kk);
}

// agree on the sample rank: all shards must report the same r > 0 (SPMD: one tiny all-reduce + host read, cached per k)
int agree_rank(const std::vector<ShardCtx>& sh, const Gather& g, int k, int* r_out) {
  int r = dhr_search_sample_rank(sh[0].ix, k);
  for (int i = 1; i < g.n_local; ++i) if (dhr_search_sample_rank(sh[i].ix, k) != r) r = 0;
  if (g.comm && g.world > 1) {
    dhr_comm* c = g.comm;
    // Not cached: the answer depends on every rank's shard size and sample period, a communicator outlives the index handles
    // (a new index can reuse a freed handle's address), and a rank that hit a cache the others missed would skip a collective they
    // issue.  The agreement costs one 8-byte all-reduce and one host read per search.
    int32_t* d = (int32_t*)sh[0].arena->get(16);
    if (!d) return dhr_set_error_message(DHR_ERR_HIP, "out of device memory");
    const int32_t h[2] = {r, -r};
    SH_HIP(hipMemcpyAsync(d, h, 8, hipMemcpyHostToDevice, sh[0].stream));
    SH_NCCL(ncclAllReduce(d, d + 2, 2, ncclInt32, ncclMin, c->comm, sh[0].stream));
    int32_t o[2];
    SH_HIP(hipMemcpyAsync(o, d + 2, 8, hipMemcpyDeviceToHost, sh[0].stream));
    SH_HIP(hipStreamSynchronize(sh[0].stream));
    r = (o[0] == r && -o[1] == r) ? r : 0;
  }
  *r_out = r;
  return DHR_OK;
}

int sharded_core(std::vector<ShardCtx>& sh, const Gather& g, const dhr_query_batch* qb_in, int k, const std::vector<float*>& out_s,
                 const std::vector<int64_t*>& out_r) {
  const int nl = g.n_local, world = g.world, Q = qb_in->n_queries;
  std::vector<dhr_query_batch> qb(nl, *qb_in);
  for (int i = 0; i < nl; ++i) SH_TRY(dhr_index_set_param(sh[i].ix, DHR_PARAM_SAMPLE_SHARE, world));      // a shard chases only its share of the union's rank
  int r = 0;
  SH_TRY(agree_rank(sh, g, k, &r));
  if (r <= 0) return local_path(sh, g, qb, k, out_s, out_r);
  const int ru = std::min<int>(dhr_search_union_rank(sh[0].ix, k), world * r);      // rank of the union that defines the threshold; the lists are r long

  // 1-2: sampled passes, common thresholds
  std::vector<const void*> send(nl);
  std::vector<void*> recv(nl);
  std::vector<float*> tau(nl);
  for (int i = 0; i < nl; ++i) {
    SH_HIP(hipSetDevice(sh[i].device));
    float* sample = (float*)sh[i].arena->get((size_t)Q * r * 4);
    recv[i] = sh[i].arena->get((size_t)world * Q * r * 4);
    tau[i] = (float*)sh[i].arena->get((size_t)Q * 4);
    if (!sample || !recv[i] || !tau[i]) return dhr_set_error_message(DHR_ERR_HIP, "out of device memory in the sharded search");
    SH_TRY(dhr_internal_search_begin_async(sh[i].ix, &qb[i], k, sample, sh[i].stream));
    send[i] = sample;
  }
  SH_TRY(g.run(sh, send, recv, (size_t)Q * r * 4));
  for (int i = 0; i < nl; ++i) {
    SH_HIP(hipSetDevice(sh[i].device));
    float* merged = (float*)sh[i].arena->get((size_t)Q * ru * 4);
    if (!merged) return dhr_set_error_message(DHR_ERR_HIP, "out of device memory in the sharded search");
    SH_TRY(dhr_merge_topk_lists(sh[i].device, Q, world, r, (const float*)recv[i], nullptr, ru, merged, nullptr, sh[i].stream));
    hipLaunchKernelGGL(column_kernel, dim3((Q + 255) / 256), dim3(256), 0, sh[i].stream, merged, ru, ru - 1, Q, tau[i]);
  }
  // 3-4: main passes, counts, failure flags
  const int kk = prefix_len(k, world);
  std::vector<float*> ls(nl);
  std::vector<int64_t*> lr(nl);
  std::vector<int32_t*> fail_ids(nl), n_failed(nl);
  std::vector<const void*> send_c(nl);
  std::vector<void*> recv_c(nl);
  for (int i = 0; i < nl; ++i) {
    SH_HIP(hipSetDevice(sh[i].device));
    ls[i] = (float*)sh[i].arena->get((size_t)Q * k * 4);
    lr[i] = (int64_t*)sh[i].arena->get((size_t)Q * k * 8);
    int32_t* cnt = (int32_t*)sh[i].arena->get((size_t)Q * 4);
    recv_c[i] = sh[i].arena->get((size_t)world * Q * 4);
    fail_ids[i] = (int32_t*)sh[i].arena->get((size_t)Q * 4);
    n_failed[i] = (int32_t*)sh[i].arena->get(256);
    if (!ls[i] || !lr[i] || !cnt || !recv_c[i] || !fail_ids[i] || !n_failed[i]) return dhr_set_error_message(DHR_ERR_HIP, "out of device memory in the sharded search");
    SH_TRY(dhr_internal_search_finish_async(sh[i].ix, tau[i], ls[i], lr[i], cnt, DHR_MEM_DEVICE, sh[i].stream));
    send_c[i] = cnt;
  }
  SH_TRY(g.run(sh, send_c, recv_c, (size_t)Q * 4));
  // 5: list prefixes, reduce
  std::vector<const void*> send_s(nl), send_r(nl);
  std::vector<void*> recv_s(nl), recv_r(nl);
  for (int i = 0; i < nl; ++i) {
    SH_HIP(hipSetDevice(sh[i].device));
    SH_HIP(hipMemsetAsync(n_failed[i], 0, 4, sh[i].stream));
    hipLaunchKernelGGL(fail_kernel, dim3((Q + 255) / 256), dim3(256), 0, sh[i].stream, (const int32_t*)recv_c[i], world, Q, k, kk, fail_ids[i], n_failed[i]);
    float* ps = ls[i];
    int64_t* pr = lr[i];
    if (kk < k) {
      ps = (float*)sh[i].arena->get((size_t)Q * kk * 4);
      pr = (int64_t*)sh[i].arena->get((size_t)Q * kk * 8);
      if (!ps || !pr) return dhr_set_error_message(DHR_ERR_HIP, "out of device memory in the sharded search");
      const int64_t n = (int64_t)Q * kk;
      hipLaunchKernelGGL(prefix_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, sh[i].stream, ls[i], lr[i], k, kk, Q, ps, pr);
    }
    recv_s[i] = sh[i].arena->get((size_t)world * Q * kk * 4);
    recv_r[i] = sh[i].arena->get((size_t)world * Q * kk * 8);
    if (!recv_s[i] || !recv_r[i]) return dhr_set_error_message(DHR_ERR_HIP, "out of device memory in the sharded search");
    send_s[i] = ps; send_r[i] = pr;
  }
  SH_TRY(g.run(sh, send_s, recv_s, (size_t)Q * kk * 4));
  SH_TRY(g.run(sh, send_r, recv_r, (size_t)Q * kk * 8));
  std::vector<int32_t> nf(nl, 0);
  for (int i = 0; i < nl; ++i) {
    SH_HIP(hipSetDevice(sh[i].device));
    SH_TRY(merge_gathered(sh[i], Q, world, kk, (const float*)recv_s[i], (const int64_t*)recv_r[i], k, out_s[i], out_r[i]));
    SH_HIP(hipMemcpyAsync(&nf[i], n_failed[i], 4, hipMemcpyDeviceToHost, sh[i].stream));
  }
  // 6: the one host read; identical on every rank (it is a function of the gathered counts)
  for (int i = 0; i < nl; ++i) SH_HIP(hipStreamSynchronize(sh[i].stream));
  const int F = nf[0];
  if (F == 0) return DHR_OK;
  // failed queries: sub-batch with local thresholds.  The ids come out of an atomic append: sort them so that every rank
  // (and every local shard) holds the same order.
  std::vector<int32_t> ids(F);
  SH_HIP(hipSetDevice(sh[0].device));
  SH_HIP(hipMemcpy(ids.data(), fail_ids[0], (size_t)F * 4, hipMemcpyDeviceToHost));
  std::sort(ids.begin(), ids.end());
  const int vsz = qb_in->value_dtype == DHR_VAL_F32 ? 4 : 2;
  const int isz = qb_in->index_dtype == DHR_IDX_I16 ? 2 : 1;
  std::vector<dhr_query_batch> sub(nl, *qb_in);
  std::vector<std::vector<char>> host_v(1), host_i(1);
  std::vector<float*> fs(nl);
  std::vector<int64_t*> fr(nl);
  std::vector<int32_t*> d_ids(nl);
  const int64_t vrow = qb_in->ld_value * vsz, irow = qb_in->index ? qb_in->ld_index * isz : 0;
  if (qb_in->mem_kind == DHR_MEM_HOST) {          // host batch: gather on the host once, shared by the local shards
    host_v[0].resize((size_t)F * vrow);
    for (int f = 0; f < F; ++f) memcpy(host_v[0].data() + (size_t)f * vrow, (const char*)qb_in->value + (size_t)ids[f] * vrow, vrow);
    if (qb_in->index) {
      host_i[0].resize((size_t)F * irow);
      for (int f = 0; f < F; ++f) memcpy(host_i[0].data() + (size_t)f * irow, (const char*)qb_in->index + (size_t)ids[f] * irow, irow);
    }
  }
  for (int i = 0; i < nl; ++i) {
    SH_HIP(hipSetDevice(sh[i].device));
    d_ids[i] = (int32_t*)sh[i].arena->get((size_t)F * 4);
    fs[i] = (float*)sh[i].arena->get((size_t)F * k * 4);
    fr[i] = (int64_t*)sh[i].arena->get((size_t)F * k * 8);
    if (!d_ids[i] || !fs[i] || !fr[i]) return dhr_set_error_message(DHR_ERR_HIP, "out of device memory in the sharded search");
    SH_HIP(hipMemcpyAsync(d_ids[i], ids.data(), (size_t)F * 4, hipMemcpyHostToDevice, sh[i].stream));
    sub[i].n_queries = F;
    if (qb_in->mem_kind == DHR_MEM_HOST) {
      sub[i].value = host_v[0].data();
      sub[i].index = qb_in->index ? host_i[0].data() : nullptr;
    } else {
      // device batch: the caller's arrays live on ONE device; gather there (shard 0's context must be that device for SPMD,
      // and in the one-process form every shard on another device reads through peer access / managed mapping)
      char* gv = (char*)sh[i].arena->get((size_t)F * vrow);
      char* gi = qb_in->index ? (char*)sh[i].arena->get((size_t)F * irow) : nullptr;
      if (!gv || (qb_in->index && !gi)) return dhr_set_error_message(DHR_ERR_HIP, "out of device memory in the sharded search");
      int64_t n = (int64_t)F * vrow;
      hipLaunchKernelGGL(gather_rows_bytes_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, sh[i].stream, (const char*)qb_in->value, vrow, (int)vrow, d_ids[i], F, gv);
      if (gi) {
        n = (int64_t)F * irow;
        hipLaunchKernelGGL(gather_rows_bytes_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, sh[i].stream, (const char*)qb_in->index, irow, (int)irow, d_ids[i], F, gi);
      }
      sub[i].value = gv;
      sub[i].index = gi;
    }
  }
  SH_TRY(local_path(sh, g, sub, k, fs, fr));
  for (int i = 0; i < nl; ++i) {
    SH_HIP(hipSetDevice(sh[i].device));
    const int64_t n = (int64_t)F * k;
    hipLaunchKernelGGL(scatter_result_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, sh[i].stream, fs[i], fr[i], d_ids[i], F, k, out_s[i], out_r[i]);
    SH_HIP(hipStreamSynchronize(sh[i].stream));
  }
  return DHR_OK;
}

int deliver(const ShardCtx& c, int Q, int k, const float* ds, const int64_t* dr, float* out_scores, int64_t* out_rows, int mem_kind) {
  if (mem_kind == DHR_MEM_HOST) {
    SH_HIP(hipMemcpyAsync(out_scores, ds, (size_t)Q * k * 4, hipMemcpyDeviceToHost, c.stream));
    SH_HIP(hipMemcpyAsync(out_rows, dr, (size_t)Q * k * 8, hipMemcpyDeviceToHost, c.stream));
  }
  SH_HIP(hipStreamSynchronize(c.stream));
  return DHR_OK;
}

}  // namespace


extern "C" int dhr_comm_unique_id(void* out, int32_t out_bytes) {
  if (!out || out_bytes < (int32_t)NCCL_UNIQUE_ID_BYTES) return dhr_set_error_message(DHR_ERR_INVALID, "the unique id needs a 128-byte buffer");
  ncclUniqueId id;
  SH_NCCL(ncclGetUniqueId(&id));
  memcpy(out, id.internal, NCCL_UNIQUE_ID_BYTES);
  return DHR_OK;
}
extern "C" int dhr_comm_create(const void* unique_id, int32_t world, int32_t rank, int32_t device, dhr_comm** out) {
  if (!unique_id || !out || world < 1 || rank < 0 || rank >= world) return dhr_set_error_message(DHR_ERR_INVALID, "bad argument");
  SH_HIP(hipSetDevice(device));
  ncclUniqueId id;
  memcpy(id.internal, unique_id, NCCL_UNIQUE_ID_BYTES);
  dhr_comm* c = new dhr_comm();
  c->world = world; c->rank = rank; c->device = device; c->owned = true;
  ncclResult_t r = ncclCommInitRank(&c->comm, world, id, rank);
  if (r != ncclSuccess) { delete c; return dhr_set_error_message(DHR_ERR_HIP, (std::string("ncclCommInitRank: ") + ncclGetErrorString(r)).c_str()); }
  *out = c;
  return DHR_OK;
}
extern "C" int dhr_comm_wrap(void* nccl_comm, int32_t world, int32_t rank, int32_t device, dhr_comm** out) {
  if (!nccl_comm || !out || world < 1 || rank < 0 || rank >= world) return dhr_set_error_message(DHR_ERR_INVALID, "bad argument");
  dhr_comm* c = new dhr_comm();
  c->comm = (ncclComm_t)nccl_comm; c->world = world; c->rank = rank; c->device = device; c->owned = false;
  *out = c;
  return DHR_OK;
}
extern "C" void dhr_comm_destroy(dhr_comm* c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  if (c->owned && c->comm) (void)ncclCommDestroy(c->comm);
  (void)hipFree(c->arena);
  delete c;
}

extern "C" int dhr_search_sharded(dhr_index* shard, dhr_comm* comm, const dhr_query_batch* qb, int32_t k, float* out_scores, int64_t* out_rows,
                                  int32_t out_mem_kind, void* stream) {
  if (!shard || !comm || !qb || !out_scores || !out_rows || k <= 0) return dhr_set_error_message(DHR_ERR_INVALID, "bad argument");
  const int device = dhr_index_device(shard);
  if (device != comm->device) return dhr_set_error_message(DHR_ERR_INVALID, "the shard and the communicator live on different devices");
  SH_HIP(hipSetDevice(device));
  Arena arena{&comm->arena, &comm->arena_bytes, 0, device};
  std::vector<ShardCtx> sh{{shard, device, (hipStream_t)stream, &arena}};
  Gather g{comm->world, 1, comm->rank, comm->world > 1 ? comm : nullptr};
  if (comm->world == 1) g.comm = nullptr;          // a single rank gathers by copy (n_local == world == 1)
  const int Q = qb->n_queries;
  float* ds = out_scores;
  int64_t* dr = out_rows;
  if (out_mem_kind == DHR_MEM_HOST) {
    ds = (float*)arena.get((size_t)Q * k * 4);
    dr = (int64_t*)arena.get((size_t)Q * k * 8);
    if (!ds || !dr) { arena.finish(); return dhr_set_error_message(DHR_ERR_HIP, "out of device memory"); }
  }
  int rc = sharded_core(sh, g, qb, k, {ds}, {dr});
  if (rc == DHR_OK) rc = deliver(sh[0], Q, k, ds, dr, out_scores, out_rows, out_mem_kind);
  else (void)hipStreamSynchronize((hipStream_t)stream);
  arena.finish();
  return rc;
}

namespace { struct LocalScratch { void* base = nullptr; size_t cap = 0; }; }

extern "C" int dhr_search_sharded_local(dhr_index** shards, int32_t n_shards, const dhr_query_batch* qb, int32_t k, float* out_scores,
                                        int64_t* out_rows, int32_t out_mem_kind, void* stream) {
  if (!shards || n_shards < 1 || n_shards > 64 || !qb || !out_scores || !out_rows || k <= 0) return dhr_set_error_message(DHR_ERR_INVALID, "bad argument (1 <= n_shards <= 64)");
  std::vector<LocalScratch> scratch(n_shards);
  std::vector<Arena> arenas;
  arenas.reserve(n_shards);
  std::vector<ShardCtx> sh;
  std::vector<hipStream_t> own(n_shards, nullptr);
  const int dev0 = dhr_index_device(shards[0]);
  for (int i = 0; i < n_shards; ++i) {
    if (!shards[i]) return dhr_set_error_message(DHR_ERR_INVALID, "null shard handle");
    const int dev = dhr_index_device(shards[i]);
    arenas.push_back(Arena{&scratch[i].base, &scratch[i].cap, 0, dev, false});
    hipStream_t s = (hipStream_t)stream;
    if (dev != dev0) { SH_HIP(hipSetDevice(dev)); SH_HIP(hipStreamCreateWithFlags(&own[i], hipStreamNonBlocking)); s = own[i]; }
    sh.push_back({shards[i], dev, s, &arenas[i]});
  }
  Gather g{n_shards, n_shards, 0, nullptr};
  const int Q = qb->n_queries;
  std::vector<float*> os(n_shards);
  std::vector<int64_t*> orow(n_shards);
  int rc = DHR_OK;
  for (int i = 0; i < n_shards && rc == DHR_OK; ++i) {
    if (hipSetDevice(sh[i].device) != hipSuccess) rc = dhr_set_error_message(DHR_ERR_HIP, "hipSetDevice failed");
    if (i == 0 && out_mem_kind == DHR_MEM_DEVICE) { os[i] = out_scores; orow[i] = out_rows; continue; }
    os[i] = (float*)arenas[i].get((size_t)Q * k * 4);
    orow[i] = (int64_t*)arenas[i].get((size_t)Q * k * 8);
    if (!os[i] || !orow[i]) rc = dhr_set_error_message(DHR_ERR_HIP, "out of device memory");
  }
  if (rc == DHR_OK) rc = sharded_core(sh, g, qb, k, os, orow);
  if (rc == DHR_OK) { (void)hipSetDevice(sh[0].device); rc = deliver(sh[0], Q, k, os[0], orow[0], out_scores, out_rows, out_mem_kind); }
  for (int i = 0; i < n_shards; ++i) {
    (void)hipSetDevice(sh[i].device);
    (void)hipStreamSynchronize(sh[i].stream);
    arenas[i].finish();
    (void)hipFree(scratch[i].base);
    if (own[i]) (void)hipStreamDestroy(own[i]);
  }
  return rc;
}
