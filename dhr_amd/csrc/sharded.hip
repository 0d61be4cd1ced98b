// Row-sharded search behind the C ABI: the corpus is split over ranks exactly like the reference's
// --total_shrad / --shrad (retrieval/gip_retrieval.py:292-306), queries are replicated, every shard searches its rows and the
// per-shard lists are reduced to the global top-k -- retrieval/merge.result.py:22-42 without the text files.
//
//   dhr_search_sharded        one process per GPU: this rank's shard + a dhr_comm (an RCCL communicator: librccl linked directly,
//                             all-gathers over xGMI)
//   dhr_search_sharded_local  one process, several shards (handles on one or more devices): the same control flow with the
//                             all-gather done by device copies -- also what the single-GPU tests drive
//
//   dhr_search_sharded_host   the same control flow over caller-supplied host shards and a caller-supplied all-gather (no device):
//                             what the CPU test-suite drives with world size 2 / 3 over gloo
//
// All run ONE implementation (sharded_core) over a Backend.  Sequence (collectives on the caller's stream):
//   1. every shard runs its sampled pass (dhr_search_begin) and holds the r best exact sample scores per query;
//   2. all-gather of [Q, r] fp32; the r-th best of the union is the common threshold tau_q, so a shard collects only ITS SHARE
//      of the global top-k (rank merge of the sorted lists in place, dhr_merge_topk_lists without rows);
//   2b. second agreement (dhr_search_mid; skipped where a shard is too small for it): every shard runs the first slice of
//      its main pass with tau, all-gather of its best scores seen so far [Q, r2] fp32, tau_q = max(tau_q, the (k f + 6 sigma + 4)-th best of
//      the union) -- f = the scattered fraction of the corpus the shards have seen by then;
//   3. rest of the main pass with tau (dhr_search_finish) -> sorted per-shard lists + the count of rows reaching tau (-1: list overflow);
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
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "dhr_internal.h"

struct dhr_comm {
  ncclComm_t comm = nullptr;      // RCCL (dhr_comm_create / dhr_comm_wrap), or
  dhr_allgather_fn cb = nullptr;  // ... a host transport supplied by the caller (dhr_comm_create_callback): HOST buffers
  void* cb_user = nullptr;
  bool owned = false;
  int world = 1, rank = 0, device = 0;
  void* arena = nullptr;          // grow-only device scratch
  size_t arena_bytes = 0;
  void* h_stage = nullptr;        // pinned staging of the callback transport: [send | recv]
  size_t h_stage_bytes = 0;
  bool dead = false;              // dhr_comm_abort ran: the handle only waits for dhr_comm_destroy
};


namespace {

using namespace dhr;

#define SH_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return dhr_set_error_message(DHR_ERR_HIP, (std::string(#x) + ": " + hipGetErrorString(e_)).c_str()); } while (0)
#define SH_NCCL(x) do { ncclResult_t r_ = (x); if (r_ != ncclSuccess) return dhr_set_error_message(DHR_ERR_HIP, (std::string(#x) + ": " + ncclGetErrorString(r_)).c_str()); } while (0)
#define SH_TRY(x) do { int rc_ = (x); if (rc_ != DHR_OK) return rc_; } while (0)

// ------------------------------------------------------------------------------------------------------------------------------------
// The sharded search is ONE control flow (sharded_core below) over a Backend: where the shards live, how their blocks are
// gathered, and the handful of per-query reductions between the collectives.  Two backends:
//   HipBackend   shards = dhr_index handles on GPUs; gathers = RCCL all-gathers (one process per GPU), a caller-supplied host
//                transport (dhr_comm_create_callback: torch.distributed over gloo, MPI, ...), or device copies (one process,
//                several shards); reductions = the kernels of this file and kernels.hip.  The product.
//   HostBackend  shards = caller-supplied callbacks on host memory, gathers = a caller-supplied callback, reductions = plain
//                loops (dhr_merge_topk_lists_host).  No device is touched: it exists so that the CPU test-suite can drive
//                THIS control flow with world size 2 / 3 over gloo (tests/test_dist_gloo.py) -- until round 4 those tests ran
//                a torch restatement of it (dhr_amd/dist.py sharded_search_torch, now deleted).
// ------------------------------------------------------------------------------------------------------------------------------------
// One rank's share of an all-gather: [payload | status record (16 B)], `stride` bytes apart in the receive buffer.  The status record is what
// makes a failed step COLLECTIVE (round 6): a rank whose local work failed (a HIP error, out of memory, a failing shard callback) does not leave
// the sequence of collectives -- it skips its remaining local work, keeps issuing the step's all-gathers with the agreed sizes, and its status
// travels in the record of every block it sends.  A transport whose blocks pass through host memory (the caller's host transport, host shards)
// reads the records right behind each gather and every rank returns from the SAME gather; over RCCL (and in the one-process form) the records are
// folded on the device into the word that the step's one host read fetches anyway, and every rank returns at the end of the step.  The failing
// rank returns its own status and message, the others DHR_ERR_PEER naming it.  (Until round 5 the failing rank returned at once and the others
// waited in their next collective for the transport's timeout.)
struct Block { size_t payload, stride; };
inline Block block_of(size_t payload) {
  payload = (payload + 15) & ~(size_t)15;
  return {payload, (payload + 16 + 255) & ~(size_t)255};
}
// first failing rank of a gathered buffer in HOST memory (0: none)
inline int host_block_status(const void* blocks, int world, const Block& b, int* who) {
  for (int w = 0; w < world; ++w) {
    int32_t st;
    memcpy(&st, (const char*)blocks + (size_t)w * b.stride + b.payload, 4);
    if (st != 0) { *who = w; return st; }
  }
  return 0;
}
inline int peer_error(int who, int status) {
  char buf[160];
  snprintf(buf, sizeof(buf), "rank %d failed in this sharded step (its status: %d); the step was abandoned on every rank", who, status);
  return dhr_set_error_message(DHR_ERR_PEER, buf);
}

struct Backend {
  int world = 1, n_local = 1;
  virtual ~Backend() {}
  virtual void* alloc(int i, size_t bytes) = 0;                 // scratch in local shard i's memory, released when the call ends
  virtual int zero(int i, void* p, size_t bytes) = 0;
  virtual int set_share(int i, int share) = 0;                  // DHR_PARAM_SAMPLE_SHARE
  virtual void abort(int) {}                                    // forget a staged search that did not reach its finish call
  virtual int sample_rank(int i, int k) = 0;
  virtual int union_rank(int i, int k) = 0;
  virtual int begin(int i, const dhr_query_batch* qb, int k, float* sample) = 0;
  virtual int finish(int i, const float* tau, float* ls, int64_t* lr, int32_t* cnt) = 0;
  virtual int mid_ranks(int i, int k, int* r_local, int* r_union) = 0;   // second agreement (dhr_search_mid_ranks); r_local 0: the shard has no such step
  virtual int mid(int i, const float* tau, int r_local, float* scores) = 0;
  // first agreement in two rounds (dhr_search_pre_ranks / dhr_search_pre / dhr_search_begin_rest); r_local 0: the shard has no such step
  virtual int pre_ranks(int i, int k, int* r_local, int* r_union) = 0;
  virtual int pre(int i, const dhr_query_batch* qb, int k, int r_local, float* scores) = 0;
  virtual int begin_rest(int i, const float* tau, float* sample) = 0;
  virtual int search(int i, const dhr_query_batch* qb, int k, float* s, int64_t* r) = 0;
  // the status record of local shard i's outgoing block
  virtual int put_status(int i, void* block, const Block& b, int status) = 0;
  // all-gather of one block per shard: send[i] (local shard i's block, b.stride bytes) -> recv[i] = [world][b.stride] in local shard i's memory.
  // A transport whose blocks pass through host memory returns DHR_ERR_PEER here when some rank's record carries a failure.
  virtual int gather(const std::vector<const void*>& send, const std::vector<void*>& recv, const Block& b) = 0;
  // ... the others fold the records of a gathered buffer into rec[1] (status of the first failing rank) / rec[2] (that rank), on the device
  virtual int fold_status(int i, const void* gathered, const Block& b, int32_t* rec) = 0;
  virtual int min_over_ranks(int32_t v[12]) = 0;                // element-wise minimum over all processes; one host read
  // gathered [world] blocks of [Q, r] sorted sample scores -> tau[q] = the ru-th best of the union
  virtual int union_threshold(int i, const float* gathered, const Block& b, int Q, int r, int ru, float* tau) = 0;
  virtual int union_threshold2(int i, const float* gathered, const Block& b, int Q, int r, int ru, float* tau) = 0;   // the same, tau[q] = max(tau[q], that)
  // counts [world] blocks of [Q] -> compact list of failed query ids + their number in rec[0] (a query is complete iff the union holds >= k rows,
  // no shard overflowed (-1) and no shard's share exceeds the gathered prefix kk)
  virtual int flag_failures(int i, const int32_t* counts, const Block& b, int Q, int k, int kk, int32_t* fail_ids, int32_t* rec) = 0;
  virtual int prefix(int i, const float* s, const int64_t* r, int k, int kk, int Q, float* os, int64_t* orow) = 0;
  // [world] blocks holding a query-major [Q, L] score section (gs) and row section (gr) each, sorted lists -> [Q, k]
  virtual int merge(int i, int Q, int L, const float* gs, const int64_t* gr, const Block& b, int k, float* os, int64_t* orow) = 0;
  // THE host read: rec[0] failed queries (+ their ids), rec[1] / rec[2] the first failing rank's status / rank (0: none)
  virtual int read_failed(const std::vector<int32_t*>& rec, const std::vector<int32_t*>& fail_ids, std::vector<int32_t>& ids, int* peer_status, int* peer_rank) = 0;
  virtual int sub_batch(int i, const dhr_query_batch* in, const std::vector<int32_t>& ids, dhr_query_batch* out, int32_t** ids_mem) = 0;
  virtual int scatter(int i, const float* s, const int64_t* r, const int32_t* ids_mem, int F, int k, float* os, int64_t* orow) = 0;
  virtual int sync(int i) = 0;
};

int prefix_len(int k, int world) {
  if (world <= 2) return k;
  const int kk = ((3 * k + world - 1) / world + 64 + 63) / 64 * 64;
  return std::min(k, kk);
}

// host-side pieces shared by both backends ------------------------------------------------------------------------------------------
void host_flag_failures(const int32_t* counts, const Block& b, int world, int Q, int k, int kk, int32_t* fail_ids, int32_t* n_failed) {
  int nf = 0;
  for (int q = 0; q < Q; ++q) {
    int64_t tot = 0;
    bool bad = false;
    for (int w = 0; w < world; ++w) {
      const int32_t c = ((const int32_t*)((const char*)counts + (size_t)w * b.stride))[q];
      if (c < 0 || c > kk) bad = true;
      tot += c > 0 ? c : 0;
    }
    if (bad || tot < k) fail_ids[nf++] = q;
  }
  *n_failed = nf;
}
// rows `ids` of a HOST query batch, gathered into `v` / `x`
void host_sub_batch(const dhr_query_batch* in, const std::vector<int32_t>& ids, std::vector<char>& v, std::vector<char>& x, dhr_query_batch* out) {
  const int vsz = in->value_dtype == DHR_VAL_F32 ? 4 : 2, isz = in->index_dtype == DHR_IDX_I16 ? 2 : 1;
  const int64_t vrow = in->ld_value * vsz, irow = in->index ? in->ld_index * isz : 0;
  const int F = (int)ids.size();
  v.resize((size_t)F * vrow);
  for (int f = 0; f < F; ++f) memcpy(v.data() + (size_t)f * vrow, (const char*)in->value + (size_t)ids[f] * vrow, vrow);
  if (in->index) {
    x.resize((size_t)F * irow);
    for (int f = 0; f < F; ++f) memcpy(x.data() + (size_t)f * irow, (const char*)in->index + (size_t)ids[f] * irow, irow);
  }
  *out = *in;
  out->n_queries = F;
  out->value = v.data();
  out->index = in->index ? x.data() : nullptr;
}

// ------------------------------------------------------------------------------------------------------------------------------------
// The state of ONE sharded step: which local shard has failed (and with what message), and the device record the status words of the
// gathered blocks are folded into.
struct Step {
  Backend& B;
  const bool spmd;                 // other processes take part in the collectives: a local failure must not leave their sequence
  std::vector<int> st;             // first failure of local shard i (DHR_OK: none); its later local work is skipped
  std::vector<int32_t*> rec;       // per local shard: {failed queries, first failing rank's status, that rank, -} in the shard's memory
  char msg[1024];                  // dhr_last_error() of the first local failure (later calls overwrite the thread's record)
  // What this rank still OWES the other ranks (round 6, second half: a failure OUTSIDE the local work -- a std::bad_alloc in the control flow itself, a block
  // that cannot be allocated -- used to return at once and leave the others in their next collective until the transport's timeout): the rank agreement
  // is one exchange of a fixed size; once the ranks are agreed the blocks of the step's all-gathers follow from them (owe(), fixed storage: planning
  // must not allocate).  leave() pays the debt with this rank's status in every block.
  Block plan[8];
  int n_plan = 0, done = 0;        // planned all-gathers; how many of them have been issued
  bool agreed = false;             // the rank agreement has been issued (successfully or not: it is never issued twice)
  bool over = false;               // the step is over on EVERY rank at this point of the sequence (a failure all of them have just read in the same
                                   // exchange) or the transport itself failed: no further collective may be issued
  explicit Step(Backend& b) : B(b), spmd(b.world > b.n_local), st(b.n_local, DHR_OK), rec(b.n_local, nullptr) { msg[0] = 0; }
  void owe(const Block& b) { if (n_plan < 8) plan[n_plan++] = b; }
  // The way out of a step that failed with `status` somewhere the sequence above does not cover: the agreement with the status in v[10] if it is still
  // owed, else every planned all-gather not yet issued, with the status in its record (host transports end the step on every rank at the first such
  // gather; with device-side records the others run the plan to its end and read the status there).  Never throws; keeps the thread's error message.
  void leave(int status) noexcept {
    if (!spmd || status == DHR_OK || over) return;
    char keep[1024];
    snprintf(keep, sizeof(keep), "%s", dhr_last_error());
    try {
      if (!agreed) {
        agreed = true;
        int32_t v[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, status, 0};
        (void)B.min_over_ranks(v);
      } else {
        std::vector<const void*> send(B.n_local);
        std::vector<void*> recv(B.n_local);
        while (done < n_plan) {
          const Block& b = plan[done];
          bool ok = true;
          for (int i = 0; i < B.n_local && ok; ++i) {
            void* blk = B.alloc(i, b.stride);
            recv[i] = B.alloc(i, (size_t)B.world * b.stride);
            ok = blk && recv[i] && B.zero(i, blk, b.stride) == DHR_OK && B.put_status(i, blk, b, st[i] != DHR_OK ? st[i] : status) == DHR_OK;
            send[i] = blk;
          }
          if (!ok) break;
          ++done;
          if (B.gather(send, recv, b) != DHR_OK) { over = true; break; }
        }
      }
    } catch (...) {
    }
    (void)dhr_set_error_message(status, keep);
  }
  int own() const { for (int v : st) if (v != DHR_OK) return v; return DHR_OK; }
  int own_error() const { return dhr_set_error_message(own(), msg); }
  // local work of shard i.  A failure is recorded; in a one-process search (nobody else is waiting) it ends the step at once.
  template <class F> int local(int i, F&& op) {
    if (st[i] != DHR_OK) return DHR_OK;
    int rc;
    try { rc = op(); } catch (...) { rc = dhr::on_exception(); }
    if (rc == DHR_OK) return DHR_OK;
    if (!spmd) return rc;
    if (own() == DHR_OK) snprintf(msg, sizeof(msg), "%s", dhr_last_error());
    st[i] = rc;
    return DHR_OK;
  }
  int init() {
    for (int i = 0; i < B.n_local; ++i) {
      rec[i] = (int32_t*)B.alloc(i, 256);
      if (!rec[i]) return dhr_set_error_message(DHR_ERR_HIP, "out of memory in the sharded search");
      SH_TRY(B.zero(i, rec[i], 16));
    }
    return DHR_OK;
  }
  // the all-gather of one block per shard, status records included
  int gather(const std::vector<const void*>& send, const std::vector<void*>& recv, const Block& b) {
    if (done < n_plan && (plan[done].stride != b.stride || plan[done].payload != b.payload))
      return dhr_set_error_message(DHR_ERR_INTERNAL, "sharded step: an all-gather does not match the step's plan");
    for (int i = 0; i < B.n_local; ++i) SH_TRY(B.put_status(i, const_cast<void*>(send[i]), b, st[i]));
    ++done;
    const int rc = B.gather(send, recv, b);
    if (rc != DHR_OK) over = true;                                      // a failure every rank has read in this gather, or a broken transport
    if (rc == DHR_ERR_PEER && own() != DHR_OK) return own_error();      // this rank is the one that failed
    SH_TRY(rc);
    for (int i = 0; i < B.n_local; ++i) SH_TRY(B.fold_status(i, recv[i], b, rec[i]));
    return DHR_OK;
  }
  // the verdict of a step whose status words travelled on the device (peer_status / peer_rank from the host read)
  int verdict(int peer_status, int peer_rank) const {
    if (own() != DHR_OK) return own_error();
    if (peer_status != 0) return peer_error(peer_rank, peer_status);
    return DHR_OK;
  }
};
#define SH_LOCAL(i, expr) SH_TRY(S.local(i, [&]() -> int { return (expr); }))

// local thresholds for a (sub-)batch: every shard searches with k, full lists gathered (ONE all-gather: scores and rows in one block) and
// merged.  out_* [n_q, k] per local shard.
// the block of the step's last all-gather: [Q] counts | [Q, kk] scores | [Q, kk] rows
inline size_t final_off_s(int Q) { return ((size_t)Q * 4 + 15) & ~(size_t)15; }
inline size_t final_off_r(int Q, int kk) { return (final_off_s(Q) + (size_t)Q * kk * 4 + 15) & ~(size_t)15; }
inline Block final_block(int Q, int kk) { return block_of(final_off_r(Q, kk) + (size_t)Q * kk * 8); }
inline Block local_block(int Q, int k) { return block_of((((size_t)Q * k * 4 + 15) & ~(size_t)15) + (size_t)Q * k * 8); }
int local_path(Step& S, const std::vector<dhr_query_batch>& qb, int k, const std::vector<float*>& out_s, const std::vector<int64_t*>& out_r) {
  Backend& B = S.B;
  const int nl = B.n_local, world = B.world, Q = qb[0].n_queries;
  const size_t off_r = ((size_t)Q * k * 4 + 15) & ~(size_t)15;
  const Block b = local_block(Q, k);
  std::vector<const void*> send(nl);
  std::vector<void*> recv(nl);
  for (int i = 0; i < nl; ++i) {
    char* blk = (char*)B.alloc(i, b.stride);
    recv[i] = B.alloc(i, (size_t)world * b.stride);
    if (!blk || !recv[i]) return dhr_set_error_message(DHR_ERR_HIP, "out of memory in the sharded search");
    SH_LOCAL(i, B.search(i, &qb[i], k, (float*)blk, (int64_t*)(blk + off_r)));
    send[i] = blk;
  }
  SH_TRY(S.gather(send, recv, b));
  for (int i = 0; i < nl; ++i)
    SH_LOCAL(i, B.merge(i, Q, k, (const float*)recv[i], (const int64_t*)((const char*)recv[i] + off_r), b, k, out_s[i], out_r[i]));
  return DHR_OK;
}

// agree on the sample ranks: all shards must report the same local share r > 0 AND the same union rank ru (local_sample_rank is many-to-one:
// unequal shard sizes or per-handle sample periods can give equal shares of different union ranks, and the count check of step 4 assumes
// ONE common threshold column).  Not cached: the answer depends on every rank's shard size and sample period, a communicator outlives the
// index handles, and a rank that hit a cache the others missed would skip a collective they issue.
// The ranks of the SECOND agreement (Backend::mid_ranks) ride along: shards whose sizes differ by a tile may compute ranks that differ by one;
// every shard then uses the LARGEST (a lower threshold: still valid), and the step is skipped when some shard has none.
// The ranks of the two-round FIRST agreement (Backend::pre_ranks) travel the same way: the largest over the shards, none if some shard has none.
// v[10] carries the ranks' status so far (a rank whose set-up failed says so here instead of staying away from the exchange).
int agree_rank(Step& S, int k, int* r_out, int* ru_out, int* rl_mid_out, int* ru_mid_out, int* rl_pre_out, int* ru_pre_out) {
  Backend& B = S.B;
  int r = B.sample_rank(0, k);
  int ru = B.union_rank(0, k);
  int ml_min = 1 << 30, ml_max = 0, mu_max = 0;
  int pl_min = 1 << 30, pl_max = 0, pu_max = 0;
  for (int i = 0; i < B.n_local; ++i) {
    if (B.sample_rank(i, k) != r || B.union_rank(i, k) != ru) r = 0;
    int ml = 0, mu = 0;
    SH_LOCAL(i, B.mid_ranks(i, k, &ml, &mu));
    if (ml <= 0 || mu <= 0) ml = mu = 0;
    ml_min = std::min(ml_min, ml); ml_max = std::max(ml_max, ml); mu_max = std::max(mu_max, mu);
    int pl = 0, pu = 0;
    SH_LOCAL(i, B.pre_ranks(i, k, &pl, &pu));
    if (pl <= 0 || pu <= 0) pl = pu = 0;
    pl_min = std::min(pl_min, pl); pl_max = std::max(pl_max, pl); pu_max = std::max(pu_max, pu);
  }
  int32_t v[12] = {r, -r, ru, -ru, ml_min, -ml_max, -mu_max, pl_min, -pl_max, -pu_max, S.own(), 0};
  S.agreed = true;
  { const int rc_x = B.min_over_ranks(v); if (rc_x != DHR_OK) { S.over = true; return rc_x; } }
  if (v[10] != DHR_OK) S.over = true;                 // every rank reads the same v[10] and leaves here
  if (v[10] != DHR_OK) return S.own() != DHR_OK ? S.own_error() : dhr_set_error_message(DHR_ERR_PEER, "another rank failed while setting up this sharded step; the step was abandoned on every rank");
  *r_out = (v[0] == r && -v[1] == r && v[2] == ru && -v[3] == ru) ? r : 0;
  *ru_out = ru;
  *rl_mid_out = v[4] > 0 ? -v[5] : 0;
  *ru_mid_out = v[4] > 0 ? -v[6] : 0;
  *rl_pre_out = v[7] > 0 ? -v[8] : 0;
  *ru_pre_out = v[7] > 0 ? -v[9] : 0;
  return DHR_OK;
}
// DHR_PARAM_SAMPLE_SHARE is handle state: the sharded entry points set it for their own staged calls and put 1 back on every way out, so
// that a later dhr_search_begin / finish on the same handle (another shard count, or none) does not inherit a stale share
struct ShareGuard {
  Backend& B;
  // (and a step that failed between begin and finish -- arena out of memory, a transport error, a failed mid step -- must not leave the
  // handle "pending": dhr_score_rows refuses a pending handle; after a completed step this is a no-op)
  ~ShareGuard() { for (int i = 0; i < B.n_local; ++i) { (void)B.set_share(i, 1); B.abort(i); } }
};

thread_local int g_last_repairs = 0;      // queries the calling thread's last sharded search redid with local thresholds (dhr_debug_sharded_repairs)

static int sharded_body(Step& S, const dhr_query_batch* qb_in, int k, float* const* out_s_in, int64_t* const* out_r_in);
// The step and its way out: whatever the body returns or throws, a rank that fails does not leave the others waiting (Step::leave).  out_s / out_r: one
// [Q, k] device (host shards: host) array per LOCAL shard.
int sharded_core(Backend& B, const dhr_query_batch* qb_in, int k, float* const* out_s, int64_t* const* out_r) {
  g_last_repairs = 0;
  const bool spmd = B.world > B.n_local;
  int rc;
  try {
    Step S(B);
    try {
      dhr::alloc_checkpoint();
      rc = sharded_body(S, qb_in, k, out_s, out_r);
    } catch (...) {
      rc = dhr::on_exception();
    }
    S.leave(rc);
    return rc;
  } catch (...) {          // the step's own state could not be built: nothing has been exchanged yet
    rc = dhr::on_exception();
  }
  if (spmd) {
    char keep[1024];
    snprintf(keep, sizeof(keep), "%s", dhr_last_error());
    try {
      int32_t v[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, rc, 0};
      (void)B.min_over_ranks(v);
    } catch (...) {
    }
    (void)dhr_set_error_message(rc, keep);
  }
  return rc;
}
static int sharded_body(Step& S, const dhr_query_batch* qb_in, int k, float* const* out_s_in, int64_t* const* out_r_in) {
  Backend& B = S.B;
  const int nl = B.n_local, world = B.world, Q = qb_in->n_queries;
  ShareGuard share_guard{B};
  const std::vector<float*> out_s(out_s_in, out_s_in + nl);
  const std::vector<int64_t*> out_r(out_r_in, out_r_in + nl);
  std::vector<dhr_query_batch> qb(nl, *qb_in);
  for (int i = 0; i < nl; ++i) SH_LOCAL(i, B.set_share(i, world));      // a shard chases only its share of the union's rank
  int r = 0, ru_all = 0, rl_mid = 0, ru_mid = 0, rl_pre = 0, ru_pre = 0;
  SH_TRY(agree_rank(S, k, &r, &ru_all, &rl_mid, &ru_mid, &rl_pre, &ru_pre));
  // the all-gathers this step owes from here on (Step::leave): nothing below may fail before they are on record
  const int kk_plan = prefix_len(k, world);
  if (r <= 0) S.owe(local_block(Q, k));
  else {
    if (rl_pre > 0 && ru_pre > 0) S.owe(block_of((size_t)Q * rl_pre * 4));
    S.owe(block_of((size_t)Q * r * 4));
    if (rl_mid > 0 && ru_mid > 0) S.owe(block_of((size_t)Q * rl_mid * 4));
    S.owe(final_block(Q, kk_plan));
  }
  SH_TRY(S.init());
  std::vector<int32_t*> fail_ids(nl, nullptr);
  std::vector<int32_t> ids;
  ids.reserve((size_t)Q);          // (the read behind the last planned all-gather must not allocate: a rank that failed THERE would not know whether the
                                   // others go on to a repair step -- until then it was the one gap a random failure could hit: tests/test_dist_gloo.py)
  int peer_status = 0, peer_rank = 0;
  if (r <= 0) {
    // shards that cannot be sampled alike: local thresholds for the whole batch
    SH_TRY(local_path(S, qb, k, out_s, out_r));
    SH_TRY(B.read_failed(S.rec, fail_ids, ids, &peer_status, &peer_rank));
    return S.verdict(peer_status, peer_rank);
  }
  const int ru = std::min<int>(ru_all, world * r);      // rank of the union that defines the threshold; the lists are r long

  // 1-2: sampled passes, common thresholds
  const Block b_sample = block_of((size_t)Q * r * 4);
  std::vector<const void*> send(nl);
  std::vector<void*> recv(nl);
  std::vector<float*> tau(nl);
  // 1a (round 5): the first part of every shard's sample, a first common threshold from the union of the parts, the rest of the samples
  // filtered at it -- eight sampled runs that each started from nothing rescored 3.2 k rows per query between them, the unsharded search's one
  // run 0.7 k.  One more all-gather of [Q, ~17] scores.
  std::vector<float*> sample_v(nl);
  for (int i = 0; i < nl; ++i) {
    sample_v[i] = (float*)B.alloc(i, b_sample.stride);
    recv[i] = B.alloc(i, (size_t)world * b_sample.stride);
    tau[i] = (float*)B.alloc(i, (size_t)Q * 4);
    if (!sample_v[i] || !recv[i] || !tau[i]) return dhr_set_error_message(DHR_ERR_HIP, "out of memory in the sharded search");
  }
  if (rl_pre > 0 && ru_pre > 0) {
    const int ru0 = std::min<int>(ru_pre, world * rl_pre);
    const Block b0 = block_of((size_t)Q * rl_pre * 4);
    std::vector<const void*> send0(nl);
    std::vector<void*> recv0(nl);
    std::vector<float*> tau0(nl);
    for (int i = 0; i < nl; ++i) {
      float* seen = (float*)B.alloc(i, b0.stride);
      recv0[i] = B.alloc(i, (size_t)world * b0.stride);
      tau0[i] = (float*)B.alloc(i, (size_t)Q * 4);
      if (!seen || !recv0[i] || !tau0[i]) return dhr_set_error_message(DHR_ERR_HIP, "out of memory in the sharded search");
      SH_LOCAL(i, B.pre(i, &qb[i], k, rl_pre, seen));
      send0[i] = seen;
    }
    SH_TRY(S.gather(send0, recv0, b0));
    for (int i = 0; i < nl; ++i) {
      SH_LOCAL(i, B.union_threshold(i, (const float*)recv0[i], b0, Q, rl_pre, ru0, tau0[i]));
      SH_LOCAL(i, B.begin_rest(i, tau0[i], sample_v[i]));
      send[i] = sample_v[i];
    }
  } else
    for (int i = 0; i < nl; ++i) {
      SH_LOCAL(i, B.begin(i, &qb[i], k, sample_v[i]));
      send[i] = sample_v[i];
    }
  SH_TRY(S.gather(send, recv, b_sample));
  for (int i = 0; i < nl; ++i) SH_LOCAL(i, B.union_threshold(i, (const float*)recv[i], b_sample, Q, r, ru, tau[i]));
  // 2b: second agreement.  Every shard runs the first slice of its main pass with tau and reports its best scores seen so far; the union of
  // what the shards have seen is a scattered fraction f of the corpus, and its (k f + 6 sigma + 4)-th best score is the threshold of the rest
  // of the pass (the shard keeps the larger of the two; the counts below are taken against it).  A 1/8 shard of the 8.8 M-row benchmark
  // rescores ~360 instead of ~490 rows per query in its main pass for one more all-gather of [Q, ~60] scores.
  if (rl_mid > 0 && ru_mid > 0) {
    const int ru2 = std::min<int>(ru_mid, world * rl_mid);
    const Block b2 = block_of((size_t)Q * rl_mid * 4);
    std::vector<const void*> send2(nl);
    std::vector<void*> recv2(nl);
    for (int i = 0; i < nl; ++i) {
      float* seen = (float*)B.alloc(i, b2.stride);
      recv2[i] = B.alloc(i, (size_t)world * b2.stride);
      if (!seen || !recv2[i]) return dhr_set_error_message(DHR_ERR_HIP, "out of memory in the sharded search");
      SH_LOCAL(i, B.mid(i, tau[i], rl_mid, seen));
      send2[i] = seen;
    }
    SH_TRY(S.gather(send2, recv2, b2));
    for (int i = 0; i < nl; ++i) SH_LOCAL(i, B.union_threshold2(i, (const float*)recv2[i], b2, Q, rl_mid, ru2, tau[i]));
  }
  // 3-5: main passes; ONE all-gather of [counts | list prefixes: scores | rows] per shard (round 6; until then the counts travelled in a
  // collective of their own, and scores and rows in one each), failure flags, reduce
  const int kk = prefix_len(k, world);
  const size_t off_s = final_off_s(Q), off_r = final_off_r(Q, kk);
  const Block bl = final_block(Q, kk);
  std::vector<const void*> send_l(nl);
  std::vector<void*> recv_l(nl);
  for (int i = 0; i < nl; ++i) {
    char* blk = (char*)B.alloc(i, bl.stride);
    recv_l[i] = B.alloc(i, (size_t)world * bl.stride);
    fail_ids[i] = (int32_t*)B.alloc(i, (size_t)Q * 4);
    if (!blk || !recv_l[i] || !fail_ids[i]) return dhr_set_error_message(DHR_ERR_HIP, "out of memory in the sharded search");
    float* ps = (float*)(blk + off_s);
    int64_t* pr = (int64_t*)(blk + off_r);
    if (kk == k) {
      SH_LOCAL(i, B.finish(i, tau[i], ps, pr, (int32_t*)blk));          // the full lists ARE the prefixes: straight into the block
    } else {
      float* ls = (float*)B.alloc(i, (size_t)Q * k * 4);
      int64_t* lr = (int64_t*)B.alloc(i, (size_t)Q * k * 8);
      if (!ls || !lr) return dhr_set_error_message(DHR_ERR_HIP, "out of memory in the sharded search");
      SH_LOCAL(i, B.finish(i, tau[i], ls, lr, (int32_t*)blk));
      SH_LOCAL(i, B.prefix(i, ls, lr, k, kk, Q, ps, pr));
    }
    send_l[i] = blk;
  }
  SH_TRY(S.gather(send_l, recv_l, bl));
  for (int i = 0; i < nl; ++i) {
    SH_LOCAL(i, B.flag_failures(i, (const int32_t*)recv_l[i], bl, Q, k, kk, fail_ids[i], S.rec[i]));
    SH_LOCAL(i, B.merge(i, Q, kk, (const float*)((const char*)recv_l[i] + off_s), (const int64_t*)((const char*)recv_l[i] + off_r), bl, k, out_s[i], out_r[i]));
  }
  // 6: the one host read; identical on every rank (it is a function of the gathered counts and status records).  The ids come out of an
  // atomic append on the device: sorted, so that every rank (and every local shard) holds the same order.
  SH_TRY(B.read_failed(S.rec, fail_ids, ids, &peer_status, &peer_rank));
  if (peer_status != 0) return S.verdict(peer_status, peer_rank);      // a failure from before the last gather: every rank reads the same record
  const int F = (int)ids.size();
  g_last_repairs = F;
  // (a failure of THIS rank behind the last gather -- its reduce -- is unknown to the others: without a repair step there is no collective left
  // and it just reports it; with one it keeps to the sequence below and its status travels in that gather)
  if (F == 0) return S.verdict(0, 0);
  S.owe(local_block(F, k));                              // the repair step's all-gather (every rank read the same F)
  std::sort(ids.begin(), ids.end());
  // failed queries (unrepresentative sample, skewed shards): sub-batch with local thresholds, gathered at full length, scattered into the result
  std::vector<dhr_query_batch> sub(nl, *qb_in);
  std::vector<float*> fs(nl);
  std::vector<int64_t*> fr(nl);
  std::vector<int32_t*> ids_mem(nl);
  for (int i = 0; i < nl; ++i) {
    SH_LOCAL(i, B.sub_batch(i, qb_in, ids, &sub[i], &ids_mem[i]));
    fs[i] = (float*)B.alloc(i, (size_t)F * k * 4);
    fr[i] = (int64_t*)B.alloc(i, (size_t)F * k * 8);
    if (!fs[i] || !fr[i]) return dhr_set_error_message(DHR_ERR_HIP, "out of memory in the sharded search");
    if (S.st[i] != DHR_OK) sub[i].n_queries = F;            // (a shard that failed before it built its sub-batch still sizes its blocks by F)
  }
  for (int i = 0; i < nl; ++i) SH_TRY(B.zero(i, S.rec[i], 16));
  SH_TRY(local_path(S, sub, k, fs, fr));
  for (int i = 0; i < nl; ++i) SH_LOCAL(i, B.scatter(i, fs[i], fr[i], ids_mem[i], F, k, out_s[i], out_r[i]));
  std::vector<int32_t*> no_ids(nl, nullptr);
  std::vector<int32_t> none;
  SH_TRY(B.read_failed(S.rec, no_ids, none, &peer_status, &peer_rank));      // (synchronises every local shard)
  return S.verdict(peer_status, peer_rank);
}

// ------------------------------------------------------------------------------------------------------------------------------------
// HipBackend
__global__ void column_kernel(const float* __restrict__ in, int ld, int col, int n, float* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = in[(int64_t)i * ld + col];
}
__global__ void column_max_kernel(const float* __restrict__ in, int ld, int col, int n, float* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = fmaxf(out[i], in[(int64_t)i * ld + col]);
}
// counts [world][Q] -> compact list of failed query ids, their number
// the status records of a gathered buffer -> rec[1] / rec[2] = status / rank of the first failing rank (kept once set)
__global__ void fold_status_kernel(const char* __restrict__ gathered, int world, size_t stride, size_t payload, int32_t* __restrict__ rec) {
  if (threadIdx.x != 0 || rec[1] != 0) return;
  for (int w = 0; w < world; ++w) {
    const int32_t st = *(const int32_t*)(gathered + (size_t)w * stride + payload);
    if (st != 0) { rec[1] = st; rec[2] = w; return; }
  }
}
__global__ void fail_kernel(const int32_t* __restrict__ counts, size_t stride, int world, int n_queries, int k, int kk, int32_t* __restrict__ fail_ids,
                            int32_t* __restrict__ n_failed) {
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= n_queries) return;
  int64_t tot = 0;
  bool bad = false;
  for (int w = 0; w < world; ++w) {
    const int32_t c = ((const int32_t*)((const char*)counts + (size_t)w * stride))[q];
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

struct HipBackend : Backend {
  std::vector<ShardCtx> sh;
  dhr_comm* comm = nullptr;       // SPMD (n_local == 1, world > 1): RCCL or the caller's host transport; null: one process, gathers are device copies
  std::vector<std::vector<char>> host_v, host_i;      // host sub-batches of the repair step (shared by the local shards)

  void* alloc(int i, size_t bytes) override { (void)hipSetDevice(sh[i].device); return sh[i].arena->get(bytes); }
  int zero(int i, void* p, size_t bytes) override {
    SH_HIP(hipSetDevice(sh[i].device));
    SH_HIP(hipMemsetAsync(p, 0, bytes, sh[i].stream));
    return DHR_OK;
  }
  int put_status(int i, void* block, const Block& b, int status) override {
    SH_HIP(hipSetDevice(sh[i].device));
    SH_HIP(hipMemsetD32Async((hipDeviceptr_t)((char*)block + b.payload), status, 4, sh[i].stream));
    return DHR_OK;
  }
  int fold_status(int i, const void* gathered, const Block& b, int32_t* rec) override {
    if (comm && comm->cb) return DHR_OK;             // the host transport read the records behind the gather already
    SH_HIP(hipSetDevice(sh[i].device));
    hipLaunchKernelGGL(fold_status_kernel, dim3(1), dim3(64), 0, sh[i].stream, (const char*)gathered, world, b.stride, b.payload, rec);
    return DHR_OK;
  }
  int set_share(int i, int share) override { return dhr_index_set_param(sh[i].ix, DHR_PARAM_SAMPLE_SHARE, share); }
  void abort(int i) override { dhr_internal_search_abort(sh[i].ix); }
  int sample_rank(int i, int k) override { return dhr_search_sample_rank(sh[i].ix, k); }
  int union_rank(int i, int k) override { return dhr_search_union_rank(sh[i].ix, k); }
  int begin(int i, const dhr_query_batch* qb, int k, float* sample) override {
    SH_HIP(hipSetDevice(sh[i].device));
    return dhr_internal_search_begin_async(sh[i].ix, qb, k, sample, sh[i].stream);
  }
  int finish(int i, const float* tau, float* ls, int64_t* lr, int32_t* cnt) override {
    SH_HIP(hipSetDevice(sh[i].device));
    return dhr_internal_search_finish_async(sh[i].ix, tau, ls, lr, cnt, DHR_MEM_DEVICE, sh[i].stream);
  }
  int mid_ranks(int i, int k, int* r_local, int* r_union) override {
    int32_t a = 0, b = 0;
    (void)dhr_search_mid_ranks(sh[i].ix, k, &a, &b);
    *r_local = a; *r_union = b;
    return DHR_OK;
  }
  int mid(int i, const float* tau, int r_local, float* scores) override {
    SH_HIP(hipSetDevice(sh[i].device));
    return dhr_internal_search_mid_async(sh[i].ix, tau, r_local, scores, sh[i].stream);
  }
  int pre_ranks(int i, int k, int* r_local, int* r_union) override {
    int32_t a = 0, b = 0;
    (void)dhr_search_pre_ranks(sh[i].ix, k, &a, &b);
    *r_local = a; *r_union = b;
    return DHR_OK;
  }
  int pre(int i, const dhr_query_batch* qb, int k, int r_local, float* scores) override {
    SH_HIP(hipSetDevice(sh[i].device));
    return dhr_internal_search_pre_async(sh[i].ix, qb, k, r_local, scores, sh[i].stream);
  }
  int begin_rest(int i, const float* tau, float* sample) override {
    SH_HIP(hipSetDevice(sh[i].device));
    return dhr_internal_search_begin_rest_async(sh[i].ix, tau, sample, sh[i].stream);
  }
  int search(int i, const dhr_query_batch* qb, int k, float* s, int64_t* r) override {
    SH_HIP(hipSetDevice(sh[i].device));
    return dhr_search(sh[i].ix, qb, k, s, r, DHR_MEM_DEVICE, sh[i].stream);
  }
  int gather(const std::vector<const void*>& send, const std::vector<void*>& recv, const Block& b) override {
    const size_t bytes = b.stride;
    if (comm && comm->dead) return dhr_set_error_message(DHR_ERR_INVALID, "the communicator was aborted (dhr_comm_abort)");
    if (comm && comm->comm) {
      SH_NCCL(ncclAllGather(send[0], recv[0], bytes, ncclInt8, comm->comm, sh[0].stream));
      return DHR_OK;
    }
    if (comm && comm->cb) {       // the caller's transport moves HOST buffers: stage through pinned memory on this stream
      const size_t need = bytes * (size_t)(world + 1);
      if (comm->h_stage_bytes < need) {
        if (comm->h_stage) (void)hipHostFree(comm->h_stage);
        comm->h_stage = nullptr; comm->h_stage_bytes = 0;
        SH_HIP(hipHostMalloc(&comm->h_stage, need, hipHostMallocDefault));
        comm->h_stage_bytes = need;
      }
      char* hs = (char*)comm->h_stage;
      char* hr = hs + bytes;
      SH_HIP(hipMemcpyAsync(hs, send[0], bytes, hipMemcpyDeviceToHost, sh[0].stream));
      SH_HIP(hipStreamSynchronize(sh[0].stream));
      if (comm->cb(comm->cb_user, hs, hr, (int64_t)bytes) != 0) return dhr_set_error_message(DHR_ERR_INTERNAL, "the caller's all-gather callback failed");
      int who = 0;
      if (const int st = host_block_status(hr, world, b, &who)) return peer_error(who, st);      // the blocks are in host memory: every rank leaves HERE
      SH_HIP(hipMemcpyAsync(recv[0], hr, bytes * (size_t)world, hipMemcpyHostToDevice, sh[0].stream));
      return DHR_OK;
    }
    // one process: make every block visible to every local shard (plain device copies; peer copies across devices).  Shards that share ONE
    // stream (the single-GPU emulation: every shard on the caller's stream) need no host synchronisation: the copies are ordered by the stream
    bool one_stream = true;
    for (int i = 1; i < n_local; ++i) one_stream = one_stream && sh[i].stream == sh[0].stream && sh[i].device == sh[0].device;
    if (!one_stream) for (int i = 0; i < n_local; ++i) SH_HIP(hipStreamSynchronize(sh[i].stream));
    for (int d = 0; d < n_local; ++d) {
      SH_HIP(hipSetDevice(sh[d].device));
      for (int s = 0; s < n_local; ++s)
        SH_HIP(hipMemcpyAsync((char*)recv[d] + (size_t)s * bytes, send[s], bytes, hipMemcpyDefault, sh[d].stream));
    }
    if (!one_stream) for (int i = 0; i < n_local; ++i) SH_HIP(hipStreamSynchronize(sh[i].stream));
    return DHR_OK;
  }
  int min_over_ranks(int32_t v[12]) override {
    if (!comm || world <= 1) return DHR_OK;
    if (comm->dead) return dhr_set_error_message(DHR_ERR_INVALID, "the communicator was aborted (dhr_comm_abort)");
    int32_t few[64 * 12];                            // (no allocation in front of the exchange for up to 64 ranks: Step::leave)
    std::vector<int32_t> many;
    if (world > 64) many.resize((size_t)world * 12);
    int32_t* all = world > 64 ? many.data() : few;
    if (comm->cb) {                                  // host transport: the values are host memory already
      if (comm->cb(comm->cb_user, v, all, 48) != 0) return dhr_set_error_message(DHR_ERR_INTERNAL, "the caller's all-gather callback failed");
    } else {                                         // (no status record on this exchange: v[10] is the status)
      SH_HIP(hipSetDevice(sh[0].device));
      int32_t* d = (int32_t*)sh[0].arena->get(64 + (size_t)world * 48);
      if (!d) return dhr_set_error_message(DHR_ERR_HIP, "out of device memory");
      SH_HIP(hipMemcpyAsync(d, v, 48, hipMemcpyHostToDevice, sh[0].stream));
      SH_NCCL(ncclAllGather(d, d + 16, 48, ncclInt8, comm->comm, sh[0].stream));
      SH_HIP(hipMemcpyAsync(all, d + 16, (size_t)world * 48, hipMemcpyDeviceToHost, sh[0].stream));
      SH_HIP(hipStreamSynchronize(sh[0].stream));
    }
    for (int w = 0; w < world; ++w)
      for (int j = 0; j < 12; ++j) v[j] = std::min(v[j], all[(size_t)w * 12 + j]);
    return DHR_OK;
  }
  // the r best of [world] sorted score lists per query, lists b.stride bytes apart
  int merge_scores(int i, const float* gathered, const Block& b, int Q, int r, int ru, float** merged_out) {
    SH_HIP(hipSetDevice(sh[i].device));
    float* merged = (float*)sh[i].arena->get((size_t)Q * ru * 4);
    if (!merged) return dhr_set_error_message(DHR_ERR_HIP, "out of device memory in the sharded search");
    if (((int64_t)world * r + ru) * 4 > 160 * 1024 || world > 64)
      return dhr_set_error_message(DHR_ERR_UNSUPPORTED, "the sample lists of one query do not fit the LDS");
    SH_HIP(launch_merge_lists(Q, world, r, gathered, nullptr, ru, merged, nullptr, sh[i].stream, (int64_t)(b.stride / 4), 0));
    *merged_out = merged;
    return DHR_OK;
  }
  int union_threshold(int i, const float* gathered, const Block& b, int Q, int r, int ru, float* tau) override {
    float* merged = nullptr;
    SH_TRY(merge_scores(i, gathered, b, Q, r, ru, &merged));
    hipLaunchKernelGGL(column_kernel, dim3((Q + 255) / 256), dim3(256), 0, sh[i].stream, merged, ru, ru - 1, Q, tau);
    return DHR_OK;
  }
  int union_threshold2(int i, const float* gathered, const Block& b, int Q, int r, int ru, float* tau) override {
    float* merged = nullptr;
    SH_TRY(merge_scores(i, gathered, b, Q, r, ru, &merged));
    hipLaunchKernelGGL(column_max_kernel, dim3((Q + 255) / 256), dim3(256), 0, sh[i].stream, merged, ru, ru - 1, Q, tau);
    return DHR_OK;
  }
  int flag_failures(int i, const int32_t* counts, const Block& b, int Q, int k, int kk, int32_t* fail_ids, int32_t* rec) override {
    SH_HIP(hipSetDevice(sh[i].device));
    hipLaunchKernelGGL(fail_kernel, dim3((Q + 255) / 256), dim3(256), 0, sh[i].stream, counts, b.stride, world, Q, k, kk, fail_ids, rec);
    return DHR_OK;
  }
  int prefix(int i, const float* s, const int64_t* r, int k, int kk, int Q, float* os, int64_t* orow) override {
    SH_HIP(hipSetDevice(sh[i].device));
    const int64_t n = (int64_t)Q * kk;
    hipLaunchKernelGGL(prefix_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, sh[i].stream, s, r, k, kk, Q, os, orow);
    return DHR_OK;
  }
  int merge(int i, int Q, int L, const float* gs, const int64_t* gr, const Block& b, int k, float* os, int64_t* orow) override {
    const ShardCtx& c = sh[i];
    SH_HIP(hipSetDevice(c.device));
    // sorted per-shard lists, one [Q, L] section per rank: rank merge in the LDS when the lists of a query fit, else the general reduce
    if (world <= 64 && ((int64_t)world * L + k) * 12 <= 160 * 1024) {
      SH_HIP(launch_merge_lists(Q, world, L, gs, gr, k, os, orow, c.stream, (int64_t)(b.stride / 4), (int64_t)(b.stride / 8)));
      return DHR_OK;
    }
    // [world, Q, L] -> [Q, world * L]
    float* ts = (float*)c.arena->get((size_t)Q * world * L * 4);
    int64_t* tr = (int64_t*)c.arena->get((size_t)Q * world * L * 8);
    if (!ts || !tr) return dhr_set_error_message(DHR_ERR_HIP, "out of device memory in the shard reduce");
    for (int w = 0; w < world; ++w) {
      SH_HIP(hipMemcpy2DAsync(ts + (size_t)w * L, (size_t)world * L * 4, (const char*)gs + (size_t)w * b.stride, (size_t)L * 4, (size_t)L * 4, Q, hipMemcpyDeviceToDevice, c.stream));
      SH_HIP(hipMemcpy2DAsync(tr + (size_t)w * L, (size_t)world * L * 8, (const char*)gr + (size_t)w * b.stride, (size_t)L * 8, (size_t)L * 8, Q, hipMemcpyDeviceToDevice, c.stream));
    }
    return dhr_merge_topk(c.device, Q, world * L, ts, tr, k, os, orow, c.stream);
  }
  int read_failed(const std::vector<int32_t*>& rec, const std::vector<int32_t*>& fail_ids, std::vector<int32_t>& ids, int* peer_status, int* peer_rank) override {
    int32_t h[4] = {0, 0, 0, 0};
    SH_HIP(hipSetDevice(sh[0].device));
    SH_HIP(hipMemcpyAsync(h, rec[0], 16, hipMemcpyDeviceToHost, sh[0].stream));
    for (int i = 0; i < n_local; ++i) SH_HIP(hipStreamSynchronize(sh[i].stream));
    *peer_status = h[1]; *peer_rank = h[2];
    const int32_t nf = fail_ids[0] ? h[0] : 0;
    ids.resize((size_t)nf);
    if (nf > 0) SH_HIP(hipMemcpy(ids.data(), fail_ids[0], (size_t)nf * 4, hipMemcpyDeviceToHost));
    return DHR_OK;
  }
  int sub_batch(int i, const dhr_query_batch* in, const std::vector<int32_t>& ids, dhr_query_batch* out, int32_t** ids_mem) override {
    const int F = (int)ids.size();
    SH_HIP(hipSetDevice(sh[i].device));
    int32_t* d_ids = (int32_t*)sh[i].arena->get((size_t)F * 4);
    if (!d_ids) return dhr_set_error_message(DHR_ERR_HIP, "out of device memory in the sharded search");
    SH_HIP(hipMemcpyAsync(d_ids, ids.data(), (size_t)F * 4, hipMemcpyHostToDevice, sh[i].stream));
    *ids_mem = d_ids;
    if (in->mem_kind == DHR_MEM_HOST) {          // host batch: gathered on the host once, shared by the local shards
      if (host_v.empty()) { host_v.resize(1); host_i.resize(1); dhr_query_batch tmp; host_sub_batch(in, ids, host_v[0], host_i[0], &tmp); }
      *out = *in;
      out->n_queries = F;
      out->value = host_v[0].data();
      out->index = in->index ? host_i[0].data() : nullptr;
      return DHR_OK;
    }
    // device batch: the caller's arrays live on ONE device; gather there (shard 0's context must be that device for SPMD, and in the
    // one-process form every shard on another device reads through peer access / managed mapping)
    const int vsz = in->value_dtype == DHR_VAL_F32 ? 4 : 2, isz = in->index_dtype == DHR_IDX_I16 ? 2 : 1;
    const int64_t vrow = in->ld_value * vsz, irow = in->index ? in->ld_index * isz : 0;
    char* gv = (char*)sh[i].arena->get((size_t)F * vrow);
    char* gi = in->index ? (char*)sh[i].arena->get((size_t)F * irow) : nullptr;
    if (!gv || (in->index && !gi)) return dhr_set_error_message(DHR_ERR_HIP, "out of device memory in the sharded search");
    int64_t n = (int64_t)F * vrow;
    hipLaunchKernelGGL(gather_rows_bytes_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, sh[i].stream, (const char*)in->value, vrow, (int)vrow, d_ids, F, gv);
    if (gi) {
      n = (int64_t)F * irow;
      hipLaunchKernelGGL(gather_rows_bytes_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, sh[i].stream, (const char*)in->index, irow, (int)irow, d_ids, F, gi);
    }
    *out = *in;
    out->n_queries = F;
    out->value = gv;
    out->index = gi;
    return DHR_OK;
  }
  int scatter(int i, const float* s, const int64_t* r, const int32_t* ids_mem, int F, int k, float* os, int64_t* orow) override {
    SH_HIP(hipSetDevice(sh[i].device));
    const int64_t n = (int64_t)F * k;
    hipLaunchKernelGGL(scatter_result_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, sh[i].stream, s, r, ids_mem, F, k, os, orow);
    return DHR_OK;
  }
  int sync(int i) override { SH_HIP(hipSetDevice(sh[i].device)); SH_HIP(hipStreamSynchronize(sh[i].stream)); return DHR_OK; }
};

// ------------------------------------------------------------------------------------------------------------------------------------
// HostBackend: one caller-supplied shard in host memory, caller-supplied all-gather.  Test / bring-up hook (dhr_search_sharded_host).
struct HostBackend : Backend {
  const dhr_host_shard* shard = nullptr;
  dhr_allgather_fn cb = nullptr;
  void* cb_user = nullptr;
  int share = 1;
  std::vector<std::vector<char>> mem;
  std::vector<char> sub_v, sub_i;
  std::vector<int32_t> sub_ids;

  void* alloc(int, size_t bytes) override { mem.emplace_back(bytes ? bytes : 16); return mem.back().data(); }
  int zero(int, void* p, size_t bytes) override { memset(p, 0, bytes); return DHR_OK; }
  int put_status(int, void* block, const Block& b, int status) override { const int32_t st = status; memcpy((char*)block + b.payload, &st, 4); return DHR_OK; }
  int fold_status(int, const void*, const Block&, int32_t*) override { return DHR_OK; }      // (gather read the records)
  int set_share(int, int s) override { share = s; return DHR_OK; }
  int sample_rank(int, int k) override { return shard->sample_rank(shard->user, k, share); }
  int union_rank(int, int k) override { return shard->union_rank(shard->user, k); }
  int begin(int, const dhr_query_batch* qb, int k, float* sample) override { return shard->begin(shard->user, qb, k, share, sample) == 0 ? DHR_OK : dhr_set_error_message(DHR_ERR_INTERNAL, "host shard: begin failed"); }
  int finish(int, const float* tau, float* ls, int64_t* lr, int32_t* cnt) override { return shard->finish(shard->user, tau, ls, lr, cnt) == 0 ? DHR_OK : dhr_set_error_message(DHR_ERR_INTERNAL, "host shard: finish failed"); }
  int mid_ranks(int, int k, int* r_local, int* r_union) override {
    int32_t a = 0, b = 0;
    if (shard->mid_ranks && shard->mid) (void)shard->mid_ranks(shard->user, k, share, &a, &b);
    *r_local = a; *r_union = b;
    return DHR_OK;
  }
  int mid(int, const float* tau, int r_local, float* scores) override { return shard->mid(shard->user, tau, r_local, scores) == 0 ? DHR_OK : dhr_set_error_message(DHR_ERR_INTERNAL, "host shard: mid failed"); }
  int pre_ranks(int, int k, int* r_local, int* r_union) override {
    int32_t a = 0, b = 0;
    if (shard->pre_ranks && shard->pre && shard->begin_rest) (void)shard->pre_ranks(shard->user, k, share, &a, &b);
    *r_local = a; *r_union = b;
    return DHR_OK;
  }
  int pre(int, const dhr_query_batch* qb, int k, int r_local, float* scores) override { return shard->pre(shard->user, qb, k, share, r_local, scores) == 0 ? DHR_OK : dhr_set_error_message(DHR_ERR_INTERNAL, "host shard: pre failed"); }
  int begin_rest(int, const float* tau, float* sample) override { return shard->begin_rest(shard->user, tau, sample) == 0 ? DHR_OK : dhr_set_error_message(DHR_ERR_INTERNAL, "host shard: begin_rest failed"); }
  int search(int, const dhr_query_batch* qb, int k, float* s, int64_t* r) override { return shard->search(shard->user, qb, k, s, r) == 0 ? DHR_OK : dhr_set_error_message(DHR_ERR_INTERNAL, "host shard: search failed"); }
  int gather(const std::vector<const void*>& send, const std::vector<void*>& recv, const Block& b) override {
    if (world == 1) memcpy(recv[0], send[0], b.stride);
    else if (cb(cb_user, send[0], recv[0], (int64_t)b.stride) != 0) return dhr_set_error_message(DHR_ERR_INTERNAL, "the caller's all-gather callback failed");
    int who = 0;
    if (const int st = host_block_status(recv[0], world, b, &who)) return peer_error(who, st);
    return DHR_OK;
  }
  int min_over_ranks(int32_t v[12]) override {
    if (world <= 1) return DHR_OK;
    int32_t few[64 * 12];                            // (no allocation in front of the exchange for up to 64 ranks: Step::leave)
    std::vector<int32_t> many;
    if (world > 64) many.resize((size_t)world * 12);
    int32_t* all = world > 64 ? many.data() : few;
    if (cb(cb_user, v, all, 48) != 0) return dhr_set_error_message(DHR_ERR_INTERNAL, "the caller's all-gather callback failed");
    for (int w = 0; w < world; ++w)
      for (int j = 0; j < 12; ++j) v[j] = std::min(v[j], all[(size_t)w * 12 + j]);
    return DHR_OK;
  }
  // [world] sections of [Q, L] floats / int64, b.stride bytes apart -> the dense [world, Q, L] arrays dhr_merge_topk_lists_host takes
  void densify(const void* g, const Block& b, size_t section_bytes, std::vector<char>& out) const {
    out.resize((size_t)world * section_bytes);
    for (int w = 0; w < world; ++w) memcpy(out.data() + (size_t)w * section_bytes, (const char*)g + (size_t)w * b.stride, section_bytes);
  }
  int union_threshold(int, const float* gathered, const Block& b, int Q, int r, int ru, float* tau) override {
    std::vector<float> merged((size_t)Q * ru);
    std::vector<char> dense;
    densify(gathered, b, (size_t)Q * r * 4, dense);
    SH_TRY(dhr_merge_topk_lists_host(Q, world, r, (const float*)dense.data(), nullptr, ru, merged.data(), nullptr));
    for (int q = 0; q < Q; ++q) tau[q] = merged[(size_t)q * ru + (ru - 1)];
    return DHR_OK;
  }
  int union_threshold2(int, const float* gathered, const Block& b, int Q, int r, int ru, float* tau) override {
    std::vector<float> merged((size_t)Q * ru);
    std::vector<char> dense;
    densify(gathered, b, (size_t)Q * r * 4, dense);
    SH_TRY(dhr_merge_topk_lists_host(Q, world, r, (const float*)dense.data(), nullptr, ru, merged.data(), nullptr));
    for (int q = 0; q < Q; ++q) tau[q] = std::max(tau[q], merged[(size_t)q * ru + (ru - 1)]);
    return DHR_OK;
  }
  int flag_failures(int, const int32_t* counts, const Block& b, int Q, int k, int kk, int32_t* fail_ids, int32_t* rec) override {
    host_flag_failures(counts, b, world, Q, k, kk, fail_ids, rec);
    return DHR_OK;
  }
  int prefix(int, const float* s, const int64_t* r, int k, int kk, int Q, float* os, int64_t* orow) override {
    for (int q = 0; q < Q; ++q) {
      memcpy(os + (size_t)q * kk, s + (size_t)q * k, (size_t)kk * 4);
      memcpy(orow + (size_t)q * kk, r + (size_t)q * k, (size_t)kk * 8);
    }
    return DHR_OK;
  }
  int merge(int, int Q, int L, const float* gs, const int64_t* gr, const Block& b, int k, float* os, int64_t* orow) override {
    std::vector<char> ds, dr;
    densify(gs, b, (size_t)Q * L * 4, ds);
    densify(gr, b, (size_t)Q * L * 8, dr);
    return dhr_merge_topk_lists_host(Q, world, L, (const float*)ds.data(), (const int64_t*)dr.data(), k, os, orow);
  }
  int read_failed(const std::vector<int32_t*>& rec, const std::vector<int32_t*>& fail_ids, std::vector<int32_t>& ids, int* peer_status, int* peer_rank) override {
    *peer_status = 0; *peer_rank = 0;                 // (host-visible blocks: a failure ended the step at its gather)
    if (fail_ids[0]) ids.assign(fail_ids[0], fail_ids[0] + rec[0][0]); else ids.clear();
    return DHR_OK;
  }
  int sub_batch(int, const dhr_query_batch* in, const std::vector<int32_t>& ids, dhr_query_batch* out, int32_t** ids_mem) override {
    if (in->mem_kind != DHR_MEM_HOST) return dhr_set_error_message(DHR_ERR_INVALID, "host shards take host query batches");
    host_sub_batch(in, ids, sub_v, sub_i, out);
    sub_ids = ids;
    *ids_mem = sub_ids.data();
    return DHR_OK;
  }
  int scatter(int, const float* s, const int64_t* r, const int32_t* ids_mem, int F, int k, float* os, int64_t* orow) override {
    for (int f = 0; f < F; ++f) {
      memcpy(os + (size_t)ids_mem[f] * k, s + (size_t)f * k, (size_t)k * 4);
      memcpy(orow + (size_t)ids_mem[f] * k, r + (size_t)f * k, (size_t)k * 8);
    }
    return DHR_OK;
  }
  int sync(int) override { return DHR_OK; }
};

int deliver(const ShardCtx& c, int Q, int k, const float* ds, const int64_t* dr, float* out_scores, int64_t* out_rows, int mem_kind) {
  if (mem_kind == DHR_MEM_HOST) {
    SH_HIP(hipMemcpyAsync(out_scores, ds, (size_t)Q * k * 4, hipMemcpyDeviceToHost, c.stream));
    SH_HIP(hipMemcpyAsync(out_rows, dr, (size_t)Q * k * 8, hipMemcpyDeviceToHost, c.stream));
  }
  SH_HIP(hipStreamSynchronize(c.stream));
  return DHR_OK;
}

}  // namespace


extern "C" int dhr_comm_unique_id(void* out, int32_t out_bytes) try {
  if (!out || out_bytes < (int32_t)NCCL_UNIQUE_ID_BYTES) return dhr_set_error_message(DHR_ERR_INVALID, "the unique id needs a 128-byte buffer");
  ncclUniqueId id;
  SH_NCCL(ncclGetUniqueId(&id));
  memcpy(out, id.internal, NCCL_UNIQUE_ID_BYTES);
  return DHR_OK;
} DHR_CATCH_STATUS
extern "C" int dhr_comm_create(const void* unique_id, int32_t world, int32_t rank, int32_t device, dhr_comm** out) try {
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
} DHR_CATCH_STATUS
extern "C" int dhr_comm_wrap(void* nccl_comm, int32_t world, int32_t rank, int32_t device, dhr_comm** out) try {
  if (!nccl_comm || !out || world < 1 || rank < 0 || rank >= world) return dhr_set_error_message(DHR_ERR_INVALID, "bad argument");
  dhr_comm* c = new dhr_comm();
  c->comm = (ncclComm_t)nccl_comm; c->world = world; c->rank = rank; c->device = device; c->owned = false;
  *out = c;
  return DHR_OK;
} DHR_CATCH_STATUS
extern "C" int dhr_comm_create_callback(int32_t world, int32_t rank, int32_t device, dhr_allgather_fn allgather, void* user, dhr_comm** out) try {
  if (!allgather || !out || world < 1 || rank < 0 || rank >= world) return dhr_set_error_message(DHR_ERR_INVALID, "bad argument");
  dhr_comm* c = new dhr_comm();
  c->cb = allgather; c->cb_user = user; c->world = world; c->rank = rank; c->device = device; c->owned = false;
  *out = c;
  return DHR_OK;
} DHR_CATCH_STATUS
extern "C" void dhr_comm_destroy(dhr_comm* c) try {
  if (!c) return;
  (void)hipSetDevice(c->device);
  if (c->owned && c->comm && !c->dead) (void)ncclCommDestroy(c->comm);      // (an aborted communicator is gone already)
  (void)hipFree(c->arena);
  if (c->h_stage) (void)hipHostFree(c->h_stage);
  delete c;
} DHR_CATCH_VOID

extern "C" int32_t dhr_debug_sharded_repairs(void) try { return g_last_repairs; } DHR_CATCH_VALUE(0)
extern "C" int dhr_comm_info(const dhr_comm* c, int32_t what) try {
  if (!c) return dhr_set_error_message(DHR_ERR_INVALID, "null communicator");
  if (c->dead) return dhr_set_error_message(DHR_ERR_INVALID, "the communicator was aborted (dhr_comm_abort)");
  int v = 0;
  switch (what) {
    case DHR_COMM_TRANSPORT: return c->comm ? 0 : 1;
    case DHR_COMM_WORLD: if (!c->comm) return c->world; SH_NCCL(ncclCommCount(c->comm, &v)); return v;
    case DHR_COMM_RANK: if (!c->comm) return c->rank; SH_NCCL(ncclCommUserRank(c->comm, &v)); return v;
    case DHR_COMM_DEVICE: if (!c->comm) return c->device; SH_NCCL(ncclCommCuDevice(c->comm, &v)); return v;
    default: return dhr_set_error_message(DHR_ERR_INVALID, "unknown dhr_comm_info field");
  }
} DHR_CATCH_STATUS
// Only aborts and marks the handle: the thread that was blocked in one of the communicator's collectives returns through sharded_core /
// HipBackend, which still use the struct and its arena (until round 5 this freed both under that thread: use after free).  The caller
// frees the handle with dhr_comm_destroy once that thread is back.
extern "C" void dhr_comm_abort(dhr_comm* c) try {
  if (!c || c->dead) return;
  c->dead = true;
  if (c->owned && c->comm) (void)ncclCommAbort(c->comm);
} DHR_CATCH_VOID

extern "C" int dhr_search_sharded_host(const dhr_host_shard* shard, int32_t world, int32_t rank, dhr_allgather_fn allgather, void* user,
                                       const dhr_query_batch* qb, int32_t k, float* out_scores, int64_t* out_rows) try {
  if (!shard) return dhr_set_error_message(DHR_ERR_INVALID, "bad argument");
  if (shard->struct_size != (uint32_t)sizeof(dhr_host_shard))
    return dhr_set_error_message(DHR_ERR_INVALID, "dhr_host_shard::struct_size does not match this library (the caller was built against another version of dhr_hip.h)");
  if (!shard->sample_rank || !shard->union_rank || !shard->begin || !shard->finish || !shard->search || !qb || !out_scores || !out_rows ||
      k <= 0 || world < 1 || rank < 0 || rank >= world || (world > 1 && !allgather))
    return dhr_set_error_message(DHR_ERR_INVALID, "bad argument");
  if (qb->mem_kind != DHR_MEM_HOST) return dhr_set_error_message(DHR_ERR_INVALID, "host shards take host query batches");
  HostBackend B;
  B.world = world; B.n_local = 1;
  B.shard = shard; B.cb = allgather; B.cb_user = user;
  return sharded_core(B, qb, k, &out_scores, &out_rows);
} DHR_CATCH_STATUS

extern "C" int dhr_search_sharded(dhr_index* shard, dhr_comm* comm, const dhr_query_batch* qb, int32_t k, float* out_scores, int64_t* out_rows,
                                  int32_t out_mem_kind, void* stream) try {
  if (!shard || !comm || !qb || !out_scores || !out_rows || k <= 0 || !DHR_MEM_KIND_OK(out_mem_kind)) return dhr_set_error_message(DHR_ERR_INVALID, "bad argument");
  if (comm->dead) return dhr_set_error_message(DHR_ERR_INVALID, "the communicator was aborted (dhr_comm_abort)");
  const int device = dhr_index_device(shard);
  if (device != comm->device) return dhr_set_error_message(DHR_ERR_INVALID, "the shard and the communicator live on different devices");
  SH_HIP(hipSetDevice(device));
  Arena arena{&comm->arena, &comm->arena_bytes, 0, device};
  HipBackend B;
  B.world = comm->world; B.n_local = 1;
  B.comm = comm->world > 1 ? comm : nullptr;       // a single rank gathers by copy (n_local == world == 1)
  std::vector<ShardCtx>& sh = B.sh;
  const int Q = qb->n_queries;
  float* ds = out_scores;
  int64_t* dr = out_rows;
  // this rank's set-up: a failure here is answered in the rank agreement the others are about to enter (the step's own failures: Step::leave)
  int rc = DHR_OK;
  try {
    sh.push_back({shard, device, (hipStream_t)stream, &arena});
    if (out_mem_kind == DHR_MEM_HOST) {
      ds = (float*)arena.get((size_t)Q * k * 4);
      dr = (int64_t*)arena.get((size_t)Q * k * 8);
      if (!ds || !dr) rc = dhr_set_error_message(DHR_ERR_HIP, "out of device memory");
    }
  } catch (...) {
    rc = dhr::on_exception();
  }
  if (rc != DHR_OK) {
    if (sh.size() == 1 && B.comm) {
      char keep[1024];
      snprintf(keep, sizeof(keep), "%s", dhr_last_error());
      try {
        int32_t v[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, rc, 0};
        (void)B.min_over_ranks(v);
      } catch (...) {
      }
      (void)hipStreamSynchronize((hipStream_t)stream);
      (void)dhr_set_error_message(rc, keep);
    }
    arena.finish();
    return rc;
  }
  rc = sharded_core(B, qb, k, &ds, &dr);
  if (rc == DHR_OK) rc = deliver(sh[0], Q, k, ds, dr, out_scores, out_rows, out_mem_kind);
  else (void)hipStreamSynchronize((hipStream_t)stream);
  arena.finish();
  return rc;
} DHR_CATCH_STATUS

extern "C" int dhr_search_sharded_local(dhr_index** shards, int32_t n_shards, const dhr_query_batch* qb, int32_t k, float* out_scores,
                                        int64_t* out_rows, int32_t out_mem_kind, void* stream) try {
  if (!shards || n_shards < 1 || n_shards > 64 || !qb || !out_scores || !out_rows || k <= 0 || !DHR_MEM_KIND_OK(out_mem_kind))
    return dhr_set_error_message(DHR_ERR_INVALID, "bad argument (1 <= n_shards <= 64)");
  std::vector<Arena> arenas;
  arenas.reserve(n_shards);
  HipBackend B;
  B.world = n_shards; B.n_local = n_shards;
  std::vector<ShardCtx>& sh = B.sh;
  std::vector<hipStream_t> own(n_shards, nullptr);
  const int dev0 = dhr_index_device(shards[0]);
  for (int i = 0; i < n_shards; ++i) {
    if (!shards[i]) return dhr_set_error_message(DHR_ERR_INVALID, "null shard handle");
    const int dev = dhr_index_device(shards[i]);
    // (the scratch of a step lives in the shard's handle and is kept between steps: until round 6 every block of every step was a hipMalloc /
    // hipFree pair -- 24 instead of 18 ms per shard of the 8-shard benchmark step)
    void** a_base = nullptr; size_t* a_cap = nullptr;
    dhr_internal_index_arena(shards[i], &a_base, &a_cap);
    arenas.push_back(Arena{a_base, a_cap, 0, dev});
    hipStream_t s = (hipStream_t)stream;
    if (dev != dev0) { SH_HIP(hipSetDevice(dev)); SH_HIP(hipStreamCreateWithFlags(&own[i], hipStreamNonBlocking)); s = own[i]; }
    sh.push_back({shards[i], dev, s, &arenas[i]});
  }
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
  if (rc == DHR_OK) rc = sharded_core(B, qb, k, os.data(), orow.data());
  if (rc == DHR_OK) { (void)hipSetDevice(sh[0].device); rc = deliver(sh[0], Q, k, os[0], orow[0], out_scores, out_rows, out_mem_kind); }
  for (int i = 0; i < n_shards; ++i) {
    (void)hipSetDevice(sh[i].device);
    (void)hipStreamSynchronize(sh[i].stream);
    arenas[i].finish();
    if (own[i]) (void)hipStreamDestroy(own[i]);
  }
  return rc;
} DHR_CATCH_STATUS
