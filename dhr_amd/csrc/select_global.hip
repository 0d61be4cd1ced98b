// Running top-k for k > 16384 (beyond what the LDS merge kernels hold): the same contract as select_kernel / select_big_kernel --
// merge a query's new keys into its sorted running list, publish the exact k-th best score as the new threshold -- through global
// memory: concatenate [running list | new keys], one segmented radix sort (hipCUB, descending u64 keys: score desc, row asc), keep
// the first kp.  The reference's torch.topk has no limit on k (gip_retrieval.py:123,142); this path is slow (a full sort per merge)
// and only taken for k > 16384.
#include <hipcub/hipcub.hpp>

#include "dhr_internal.h"

namespace dhr {

namespace {
__global__ void sg_concat_kernel(SelectArgs p, int64_t L, uint64_t* __restrict__ c) {
  const int q = blockIdx.y;
  uint32_t count = p.cnt ? p.cnt[q] : p.count_all;
  if (p.cnt && count > p.cap) count = p.cap;
  const uint64_t* topk = p.topk_keys + (int64_t)q * p.kp;
  const uint64_t* in = p.in_keys + (int64_t)q * p.ld_keys;
  for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < L; j += (int64_t)gridDim.x * blockDim.x)
    c[(int64_t)q * L + j] = j < p.kp ? topk[j] : (j - p.kp < (int64_t)count ? in[j - p.kp] : 0ull);
}
__global__ void sg_take_kernel(SelectArgs p, int64_t L, const uint64_t* __restrict__ sorted) {
  const int q = blockIdx.y;
  uint64_t* topk = p.topk_keys + (int64_t)q * p.kp;
  for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < p.kp; j += (int64_t)gridDim.x * blockDim.x) {
    const uint64_t key = sorted[(int64_t)q * L + j];
    topk[j] = key;
    if (j == p.k - 1 && key != 0ull) {
      float t = ordered_f32((uint32_t)(key >> 32));
      if (p.monotone) t = fmaxf(t, p.tau[q]);
      p.tau[q] = t;
      p.thr[q] = t - p.margin[q];
    }
  }
}
struct SegOffset {
  int64_t L;
  __host__ __device__ int64_t operator()(int64_t i) const { return i * L; }
};
}  // namespace

hipError_t launch_select_global(const SelectArgs& a, hipStream_t s) {
  const int64_t ld_in = a.cnt ? std::min<int64_t>(a.ld_keys, a.cap) : std::min<int64_t>(a.ld_keys, a.count_all);
  const int64_t L = (int64_t)a.kp + std::max<int64_t>(ld_in, 1);
  const int64_t items = L * a.n_queries;
  if (items > (int64_t)0x7fffffff) return hipErrorInvalidValue;         // hipCUB's item count is an int
  uint64_t *c = nullptr, *c2 = nullptr;
  void* tmp = nullptr;
  size_t tmp_bytes = 0;
  hipError_t e;
  hipcub::CountingInputIterator<int64_t> cnt_it(0);
  hipcub::TransformInputIterator<int64_t, SegOffset, hipcub::CountingInputIterator<int64_t>> begin_it(cnt_it, SegOffset{L});
  hipcub::TransformInputIterator<int64_t, SegOffset, hipcub::CountingInputIterator<int64_t>> end_it(cnt_it + 1, SegOffset{L});
  if ((e = hipMallocAsync((void**)&c, (size_t)items * 8, s)) != hipSuccess) return e;
  if ((e = hipMallocAsync((void**)&c2, (size_t)items * 8, s)) != hipSuccess) { (void)hipFreeAsync(c, s); return e; }
  e = hipcub::DeviceSegmentedRadixSort::SortKeysDescending(nullptr, tmp_bytes, c, c2, (int)items, a.n_queries, begin_it, end_it, 0, 64, s);
  if (e == hipSuccess) e = hipMallocAsync(&tmp, tmp_bytes ? tmp_bytes : 16, s);
  if (e == hipSuccess) {
    const unsigned bx = (unsigned)std::min<int64_t>((L + 255) / 256, 1024);
    hipLaunchKernelGGL(sg_concat_kernel, dim3(bx, (unsigned)a.n_queries), dim3(256), 0, s, a, L, c);
    e = hipcub::DeviceSegmentedRadixSort::SortKeysDescending(tmp, tmp_bytes, c, c2, (int)items, a.n_queries, begin_it, end_it, 0, 64, s);
    if (e == hipSuccess) {
      const unsigned bt = (unsigned)std::min<int64_t>(((int64_t)a.kp + 255) / 256, 1024);
      hipLaunchKernelGGL(sg_take_kernel, dim3(bt, (unsigned)a.n_queries), dim3(256), 0, s, a, L, (const uint64_t*)c2);
      e = hipGetLastError();
    }
  }
  (void)hipFreeAsync(tmp, s);
  (void)hipFreeAsync(c2, s);
  (void)hipFreeAsync(c, s);
  return e;
}

// dhr_merge_topk beyond what one workgroup's LDS holds (n_in > 16384 entries per query, e.g. full-k lists of more than 16 shards):
// order = score descending, global row ascending (merge.result.py:22-42 with the library's tie rule).  Rows are 64-bit, so the order is
// produced by two STABLE segmented radix sorts: by row ascending, then by the order-preserving score bits descending; padding entries
// (row < 0) get the lowest score key and come out as (-inf, -1).
namespace {
__global__ void mg_init_kernel(int64_t total, int n_in, const int64_t* __restrict__ rows, uint64_t* __restrict__ rkey, uint32_t* __restrict__ idx) {
  for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = rows[g];
    rkey[g] = r < 0 ? ~0ull : (uint64_t)r;                  // padding last
    idx[g] = (uint32_t)(g % n_in);
  }
}
__global__ void mg_score_kernel(int64_t total, int n_in, const float* __restrict__ scores, const int64_t* __restrict__ rows,
                                const uint32_t* __restrict__ idx_by_row, uint32_t* __restrict__ skey) {
  for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (int64_t)gridDim.x * blockDim.x) {
    const int64_t src = (g / n_in) * n_in + idx_by_row[g];
    skey[g] = rows[src] < 0 ? 0u : f32_ordered(scores[src]);
  }
}
__global__ void mg_emit_kernel(int n_queries, int n_in, int k_out, const float* __restrict__ scores, const int64_t* __restrict__ rows,
                               const uint32_t* __restrict__ idx_sorted, float* __restrict__ out_scores, int64_t* __restrict__ out_rows) {
  const int q = blockIdx.y;
  for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < k_out; j += gridDim.x * blockDim.x) {
    float sc = -INFINITY;
    int64_t row = -1;
    if (j < n_in) {
      const int64_t src = (int64_t)q * n_in + idx_sorted[(int64_t)q * n_in + j];
      if (rows[src] >= 0) { sc = scores[src]; row = rows[src]; }
    }
    out_scores[(int64_t)q * k_out + j] = sc;
    out_rows[(int64_t)q * k_out + j] = row;
  }
}
}  // namespace

hipError_t launch_merge_topk_global(int n_queries, int n_in, const float* in_scores, const int64_t* in_rows, int k_out, float* out_scores,
                                    int64_t* out_rows, hipStream_t s) {
  const int64_t total = (int64_t)n_queries * n_in;
  if (total > (int64_t)0x7fffffff) return hipErrorInvalidValue;         // hipCUB's item count is an int
  uint64_t *rk = nullptr, *rk2 = nullptr;
  uint32_t *ix = nullptr, *ix2 = nullptr, *sk = nullptr, *sk2 = nullptr;
  void* tmp = nullptr;
  size_t t1 = 0, t2 = 0;
  hipcub::CountingInputIterator<int64_t> cnt_it(0);
  hipcub::TransformInputIterator<int64_t, SegOffset, hipcub::CountingInputIterator<int64_t>> b_it(cnt_it, SegOffset{n_in});
  hipcub::TransformInputIterator<int64_t, SegOffset, hipcub::CountingInputIterator<int64_t>> e_it(cnt_it + 1, SegOffset{n_in});
  hipError_t e = hipSuccess;
  auto alloc = [&](void** p, size_t b) { if (e == hipSuccess) e = hipMallocAsync(p, b, s); };
  alloc((void**)&rk, (size_t)total * 8); alloc((void**)&rk2, (size_t)total * 8);
  alloc((void**)&ix, (size_t)total * 4); alloc((void**)&ix2, (size_t)total * 4);
  alloc((void**)&sk, (size_t)total * 4); alloc((void**)&sk2, (size_t)total * 4);
  if (e == hipSuccess) e = hipcub::DeviceSegmentedRadixSort::SortPairs(nullptr, t1, rk, rk2, ix, ix2, (int)total, n_queries, b_it, e_it, 0, 64, s);
  if (e == hipSuccess) e = hipcub::DeviceSegmentedRadixSort::SortPairsDescending(nullptr, t2, sk, sk2, ix2, ix, (int)total, n_queries, b_it, e_it, 0, 32, s);
  alloc(&tmp, std::max<size_t>(std::max(t1, t2), 16));
  if (e == hipSuccess) {
    const unsigned blocks = (unsigned)std::min<int64_t>((total + 255) / 256, 65536);
    hipLaunchKernelGGL(mg_init_kernel, dim3(blocks), dim3(256), 0, s, total, n_in, in_rows, rk, ix);
    e = hipcub::DeviceSegmentedRadixSort::SortPairs(tmp, t1, rk, rk2, ix, ix2, (int)total, n_queries, b_it, e_it, 0, 64, s);       // ix2: positions by row ascending
    if (e == hipSuccess) {
      hipLaunchKernelGGL(mg_score_kernel, dim3(blocks), dim3(256), 0, s, total, n_in, in_scores, in_rows, (const uint32_t*)ix2, sk);
      e = hipcub::DeviceSegmentedRadixSort::SortPairsDescending(tmp, t2, sk, sk2, ix2, ix, (int)total, n_queries, b_it, e_it, 0, 32, s);   // stable: ties keep the row order
    }
    if (e == hipSuccess) {
      hipLaunchKernelGGL(mg_emit_kernel, dim3((unsigned)std::min((k_out + 255) / 256, 1024), (unsigned)n_queries), dim3(256), 0, s, n_queries, n_in, k_out,
                         in_scores, in_rows, (const uint32_t*)ix, out_scores, out_rows);
      e = hipGetLastError();
    }
  }
  (void)hipFreeAsync(tmp, s); (void)hipFreeAsync(sk2, s); (void)hipFreeAsync(sk, s); (void)hipFreeAsync(ix2, s); (void)hipFreeAsync(ix, s);
  (void)hipFreeAsync(rk2, s); (void)hipFreeAsync(rk, s);
  return e;
}

}  // namespace dhr
