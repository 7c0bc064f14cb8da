// launch_order.hpp - order in which a step launch visits the envs: most expensive first.
//
// Workgroups are dispatched in index order and a launch ends when its slowest waves end, the last ones running alone at
// low occupancy.  Envs differ in cost (active limits, contacts, solver iterations), and the cost of a control step predicts
// the next one's, so each step launch is followed by a counting sort of the envs by that cost (one workgroup, keys 0..255,
// descending, ties in arbitrary order); the step kernels read `env = order[blockIdx.x]` and leave their cost in a compact
// key array (`cost[env]`, read coalesced here).  Envs are independent: results do not depend on the order.
#pragma once
#include <hip/hip_runtime.h>

namespace ffe_order {
namespace {  // one copy per translation unit

__global__ __launch_bounds__(1024) void order_by_cost(const int *__restrict__ cost, int *__restrict__ order, int batch) {
  __shared__ int hist[256], start[256];
  const int t = threadIdx.x;
  if (t < 256) hist[t] = 0;
  __syncthreads();
  constexpr int KPT = 16;  // keys kept in registers between the two passes (batches up to 16 384; larger ones re-read)
  int key[KPT];
#pragma unroll
  for (int q = 0; q < KPT; q++) {
    const int e = t + q * 1024;
    key[q] = e < batch ? min(255, max(0, cost[e])) : -1;
    if (key[q] >= 0) atomicAdd(&hist[key[q]], 1);
  }
  for (int e = t + KPT * 1024; e < batch; e += 1024) atomicAdd(&hist[min(255, max(0, cost[e]))], 1);
  __syncthreads();
  // exclusive suffix sums (descending keys first): start[k] = number of envs with a key > k
  if (t < 256) start[t] = hist[t];
  __syncthreads();
  for (int off = 1; off < 256; off <<= 1) {
    int v = 0;
    if (t < 256 && t + off < 256) v = start[t + off];
    __syncthreads();
    if (t < 256) start[t] += v;
    __syncthreads();
  }
  if (t < 256) start[t] -= hist[t];
  __syncthreads();
#pragma unroll
  for (int q = 0; q < KPT; q++)
    if (key[q] >= 0) order[atomicAdd(&start[key[q]], 1)] = t + q * 1024;
  for (int e = t + KPT * 1024; e < batch; e += 1024) order[atomicAdd(&start[min(255, max(0, cost[e]))], 1)] = e;
}

}  // namespace
}  // namespace ffe_order
