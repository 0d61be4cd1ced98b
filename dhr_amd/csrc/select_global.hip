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
      const float t = ordered_f32((uint32_t)(key >> 32));
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

}  // namespace dhr
