// nstep.hip - bulk n-step transition writer on the device: the batched counterpart of the adder the reference's actors
// feed (acme.adders.reverb.NStepTransitionAdder(n_step=50, discount=gamma), built at agents/ray_distributed_dmpo.py:514-521
// and driven by actors.py:91-101 observe_first / observe).  acme is not in the reference tree or in this image: the
// semantics below restate its published behaviour and are checked against a numpy restatement only (parity unpinned).
//
// Per env and per observed step t (action a_t, next timestep (r, d, o_{t+1}, step_type)):
//   * the ring keeps the last n entries (o_s, a_s, r_{s+1}, d_{s+1});
//   * every observed step writes the transition from the OLDEST held entry to o_{t+1} - acme's _write runs on every add() and does
//     not wait for n entries, so an episode's first n - 1 steps yield the short transitions (o_0 -> o_1), (o_0 -> o_2), ...:
//       (o_s, a_s, R, D, o_{t+1}),  R = r_0 + g d_0 r_1 + g^2 d_0 d_1 r_2 + ...,  D = g^(m-1) d_0 d_1 ... d_(m-1)   (m entries spanned)
//     (acme's _compute_cumulative_quantities: the env discounts multiply in, the learner applies one more g);
//   * on LAST the remaining, shorter, transitions are flushed as well (acme's _write_last), all ending in o_{t+1}: an episode of
//     T < n steps leaves 2 T - 1 transitions, one of T >= n steps leaves T + n - 1;
//   * FIRST starts a new episode: the ring is cleared and o_0 stored.
// One wavefront per env: lanes move the observation / action rows (coalesced), lane-parallel products give R and D.
// Transitions land in a device-resident replay ring (slot = running counter mod capacity); nothing touches the host.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <memory>
#include <stdexcept>
#include <string>

#include "../../include/flybody_env.h"

namespace ffn {

struct Dev {
  int batch, obs_dim, act_dim, n_step;
  float gamma;
  long long capacity;
  // per-env rings
  float *r_obs, *r_act, *r_rew, *r_disc;  // [B][n][O], [B][n][A], [B][n], [B][n]
  float *last_obs;                        // [B][O] observation the next action will be taken from
  int *head, *count;                      // ring write position / entries held
  // replay ring
  float *t_obs, *t_act, *t_ret, *t_disc, *t_next;
  unsigned long long *written;            // transitions written so far (monotone)
};

// Writes the `total` transitions that start at ring entries 0 .. total - 1 (0 = oldest of `len` held entries) and all end in
// `next_obs`: total = 1 on an ordinary step, = len on LAST (the shorter tails).  `rew_l` / `disc_l`: lane i holds the reward /
// discount of the i-th oldest entry (n_step <= 64), else they are read from the ring.
// acme's order of operations (left to right from each transition's own start) is kept so that the float32 result matches a scalar
// restatement bit for bit - but the chains of the different starts are independent: lane j runs the chain of start j, all of them
// in lock step over one pass of the entries (values broadcast from the lanes, no memory access inside the dependent chain).  A LAST
// step used to run its up to n chains one after the other, n (n + 1) / 2 dependent steps in one wave, and the few envs that end an
// episode in a given call set the duration of the whole launch.
__device__ __forceinline__ void emit_all(const Dev &D, int env, int lane, int first, int total, int len, const float *next_obs, unsigned long long base,
                                         float rew_l, float disc_l, bool in_lanes) {
  const int n = D.n_step, O = D.obs_dim, A = D.act_dim;
  const float *rr = D.r_rew + (size_t)env * n, *rd = D.r_disc + (size_t)env * n;
  const unsigned long long cap = (unsigned long long)D.capacity;
  if (total <= 64) {
    float ret = 0.f, td = 1.f;
    {
#pragma clang fp contract(off)
      for (int i = 0; i < len; i++) {
        float r_, d_;
        if (in_lanes) { r_ = __shfl(rew_l, i); d_ = __shfl(disc_l, i); }
        else { const int e = (first + i) % n; r_ = rr[e]; d_ = rd[e]; }
        if (lane == i) { ret = r_; td = d_; }
        else if (lane < i) { td *= D.gamma; ret += r_ * td; td *= d_; }
      }
    }
    if (lane < total) { const unsigned long long slot = (base + lane) % cap; D.t_ret[slot] = ret; D.t_disc[slot] = td; }
  } else {  // (n_step > 64: one chain after the other)
    for (int s = 0; s < total; s++) {
      float ret = 0.f, td = 1.f;
      {
#pragma clang fp contract(off)
        for (int i = s; i < len; i++) {
          const int e = (first + i) % n;
          const float r_ = rr[e], d_ = rd[e];
          if (i == s) { ret = r_; td = d_; }
          else { td *= D.gamma; ret += r_ * td; td *= d_; }
        }
      }
      if (lane == 0) { const unsigned long long slot = (base + s) % cap; D.t_ret[slot] = ret; D.t_disc[slot] = td; }
    }
  }
  // the rows: for each column block, the entries of kBatch transitions are loaded back to back and then stored, so that a LAST step's
  // up to n copies cost n / kBatch memory round trips instead of n (the pointers may alias as far as the compiler knows: written as
  // a plain loop it keeps every load behind the previous iteration's stores)
  constexpr int kBatch = 16;
  const float *ro = D.r_obs + (size_t)env * n * O, *ra = D.r_act + (size_t)env * n * A;
  for (int k = lane; k < O; k += 64) {
    const float nxt = next_obs[k];
    for (int s0 = 0; s0 < total; s0 += kBatch) {
      float v[kBatch];
#pragma unroll
      for (int u = 0; u < kBatch; u++) v[u] = s0 + u < total ? ro[(size_t)((first + s0 + u) % n) * O + k] : 0.f;
#pragma unroll
      for (int u = 0; u < kBatch; u++)
        if (s0 + u < total) { const unsigned long long slot = (base + s0 + u) % cap; D.t_obs[slot * O + k] = v[u]; D.t_next[slot * O + k] = nxt; }
    }
  }
  for (int k = lane; k < A; k += 64) {
    for (int s0 = 0; s0 < total; s0 += kBatch) {
      float v[kBatch];
#pragma unroll
      for (int u = 0; u < kBatch; u++) v[u] = s0 + u < total ? ra[(size_t)((first + s0 + u) % n) * A + k] : 0.f;
#pragma unroll
      for (int u = 0; u < kBatch; u++)
        if (s0 + u < total) D.t_act[((base + s0 + u) % cap) * A + k] = v[u];
    }
  }
}

// kEnvsPerBlock envs per workgroup, one wavefront each.  The slots of the replay ring are claimed with ONE atomic per workgroup
// (the waves' counts are summed through LDS): a device-scope atomic on a single address is served by the memory side, one after
// the other across all eight XCDs, and one per env (8 192 per call) cost more than everything else in this kernel together.
constexpr int kEnvsPerBlock = 16;
__global__ __launch_bounds__(64 * kEnvsPerBlock) void nstep_observe_kernel(Dev D, const float *__restrict__ action, const int *__restrict__ step_type,
                                                                           const float *__restrict__ reward, const float *__restrict__ discount,
                                                                           const float *__restrict__ obs) {
  __shared__ int s_total[kEnvsPerBlock];
  __shared__ unsigned long long s_base;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int env = blockIdx.x * kEnvsPerBlock + wave;
  const bool live = env < D.batch;
  const int n = D.n_step, O = D.obs_dim, A = D.act_dim;
  const float *o_next = obs + (size_t)(live ? env : 0) * O;
  float *lo = D.last_obs + (size_t)(live ? env : 0) * O;
  const int st = live ? step_type[env] : FFE_STEP_FIRST;
  const bool first_step = st == FFE_STEP_FIRST;
  int head = 0, cnt = 0, head_new = 0, cnt_new = 0, first = 0, total = 0;
  if (live && !first_step) {
    head = D.head[env]; cnt = D.count[env];
    head_new = (head + 1) % n; cnt_new = cnt < n ? cnt + 1 : n;
    first = (head_new - cnt_new + n) % n;  // ring index of the oldest held entry after this append
    // how many transitions this call writes: the one from the oldest held entry (n steps long once the ring is full, shorter during
    // an episode's first steps) and, on LAST, every shorter tail
    total = 1 + (st == FFE_STEP_LAST ? cnt_new - 1 : 0);
  }
  if (lane == 0) s_total[wave] = total;
  __syncthreads();
  if (threadIdx.x == 0) {
    int sum = 0;
    for (int w = 0; w < kEnvsPerBlock; w++) sum += s_total[w];
    s_base = sum > 0 ? atomicAdd(D.written, (unsigned long long)sum) : 0ull;
  }
  // (the atomic's round trip overlaps the loads and row copies below; the base is read after the next barrier)
  // rewards / discounts of the held entries, one per lane (the entry appended by this call comes from the arguments)
  const bool in_lanes = n <= 64;
  float rew_l = 0.f, disc_l = 0.f;
  if (live && !first_step) {
    if (in_lanes && lane < cnt_new) {
      const int e = (first + lane) % n;
      if (e == head) { rew_l = reward[env]; disc_l = discount[env]; }
      else { rew_l = D.r_rew[(size_t)env * n + e]; disc_l = D.r_disc[(size_t)env * n + e]; }
    }
    // append (o_t, a_t, r_{t+1}, d_{t+1}); when the ring is full its oldest entry (already written out) is overwritten
    float *ro = D.r_obs + ((size_t)env * n + head) * O, *ra = D.r_act + ((size_t)env * n + head) * A;
    for (int k = lane; k < O; k += 64) ro[k] = lo[k];
    for (int k = lane; k < A; k += 64) ra[k] = action[(size_t)env * A + k];
    if (lane == 0) { D.r_rew[(size_t)env * n + head] = reward[env]; D.r_disc[(size_t)env * n + head] = discount[env]; }
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __syncthreads();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  if (!live) return;
  if (first_step) {  // observe_first: new episode
    for (int k = lane; k < O; k += 64) lo[k] = o_next[k];
    if (lane == 0) { D.head[env] = 0; D.count[env] = 0; }
    return;
  }
  if (total > 0) {
    unsigned long long base = s_base;
    for (int w = 0; w < wave; w++) base += (unsigned long long)s_total[w];
    // transition 0 starts at the oldest held entry (the n-step transition once the ring is full), the rest (LAST only) start later
    emit_all(D, env, lane, first, total, cnt_new, o_next, base, rew_l, disc_l, in_lanes);
  }
  for (int k = lane; k < O; k += 64) lo[k] = o_next[k];
  if (lane == 0) { D.head[env] = head_new; D.count[env] = cnt_new; }
}

// one [B][O + 3] row per env: observation | reward | discount | step_type (as float): the unit the per-step gather moves
__global__ void pack_timestep_kernel(const float *__restrict__ obs, const float *__restrict__ rew, const float *__restrict__ disc,
                                     const int *__restrict__ st, float *__restrict__ out, int batch, int obs_dim) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x, w = obs_dim + 3;
  if (i >= (long long)batch * w) return;
  const int env = (int)(i / w), k = (int)(i - (long long)env * w);
  out[i] = k < obs_dim ? obs[(long long)env * obs_dim + k] : (k == obs_dim ? rew[env] : (k == obs_dim + 1 ? disc[env] : (float)st[env]));
}

// per-env running episode return / length and batch totals of finished episodes, in one launch: the statistics the reference's
// EnvironmentLoop logs (episode_return, episode_length; agents/ray_distributed_dmpo.py:401-440).  A FIRST row adds nothing; a LAST
// row's episode is added to the totals {episodes, sum of lengths} (int64) and {sum of returns} (float64) and its counters restart.
__global__ void episode_stats_kernel(const int *__restrict__ st, const float *__restrict__ rew, float *__restrict__ ep_ret, long long *__restrict__ ep_len,
                                     long long *__restrict__ tot_i, double *__restrict__ tot_ret, int batch) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  int done = 0;
  long long len = 0;
  double ret = 0.0;
  if (i < batch) {
    const int s = st[i];
    float r = ep_ret[i];
    long long l = ep_len[i];
    if (s != FFE_STEP_FIRST) { r += rew[i]; l += 1; }
    if (s == FFE_STEP_LAST) { done = 1; len = l; ret = (double)r; r = 0.f; l = 0; }
    ep_ret[i] = r; ep_len[i] = l;
  }
  // wave totals by cross-lane sums, then one atomic per wave and quantity (LAST rows are rare)
  for (int o = 32; o > 0; o >>= 1) { done += __shfl_xor(done, o); len += __shfl_xor(len, o); ret += __shfl_xor(ret, o); }
  if ((threadIdx.x & 63) == 0 && done) {
    atomicAdd((unsigned long long *)&tot_i[0], (unsigned long long)done);
    atomicAdd((unsigned long long *)&tot_i[1], (unsigned long long)len);
    atomicAdd(tot_ret, ret);
  }
}

struct Handle {
  Dev d{};
  int device = 0;
  void *allocs[16] = {nullptr};
  int nalloc = 0;
  std::string err;
};

}  // namespace ffn

using ffn::Handle;

struct ffe_nstep {
  Handle h;
};

static thread_local std::string g_nerr;

extern "C" {

int ffe_nstep_create(int batch, int obs_dim, int act_dim, int n_step, float discount, long long capacity, int device, ffe_nstep_handle *out) {
  if (!out) return -1;
  *out = nullptr;
  if (batch <= 0 || obs_dim <= 0 || act_dim <= 0 || n_step <= 0 || capacity <= 0) { g_nerr = "ffe_nstep_create: bad arguments"; return -1; }
  int ndev = 0, prev = -1;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev) { g_nerr = "no such HIP device: the MI355X path has no CPU fallback"; return -1; }
  (void)hipGetDevice(&prev);
  (void)hipSetDevice(device);
  std::unique_ptr<ffe_nstep> p(new ffe_nstep());
  Handle &H = p->h;
  H.device = device;
  ffn::Dev &D = H.d;
  D.batch = batch; D.obs_dim = obs_dim; D.act_dim = act_dim; D.n_step = n_step; D.gamma = discount; D.capacity = capacity;
  bool ok = true;
  auto alloc = [&](size_t bytes) -> void * {
    void *q = nullptr;
    if (hipMalloc(&q, bytes) != hipSuccess) { ok = false; return nullptr; }
    (void)hipMemset(q, 0, bytes);
    H.allocs[H.nalloc++] = q;
    return q;
  };
  const size_t B = (size_t)batch, n = (size_t)n_step, O = (size_t)obs_dim, A = (size_t)act_dim, C = (size_t)capacity;
  D.r_obs = (float *)alloc(B * n * O * 4); D.r_act = (float *)alloc(B * n * A * 4); D.r_rew = (float *)alloc(B * n * 4); D.r_disc = (float *)alloc(B * n * 4);
  D.last_obs = (float *)alloc(B * O * 4); D.head = (int *)alloc(B * 4); D.count = (int *)alloc(B * 4);
  D.t_obs = (float *)alloc(C * O * 4); D.t_act = (float *)alloc(C * A * 4); D.t_ret = (float *)alloc(C * 4); D.t_disc = (float *)alloc(C * 4);
  D.t_next = (float *)alloc(C * O * 4); D.written = (unsigned long long *)alloc(8);
  (void)hipDeviceSynchronize();
  if (prev >= 0 && prev != device) (void)hipSetDevice(prev);
  if (!ok) {
    for (int k = 0; k < H.nalloc; k++) (void)hipFree(H.allocs[k]);
    g_nerr = "ffe_nstep_create: out of device memory";
    return -1;
  }
  *out = p.release();
  return 0;
}

int ffe_nstep_destroy(ffe_nstep_handle p) {
  if (!p) return -1;
  int prev = -1;
  (void)hipGetDevice(&prev);
  (void)hipSetDevice(p->h.device);
  for (int k = 0; k < p->h.nalloc; k++) (void)hipFree(p->h.allocs[k]);
  if (prev >= 0 && prev != p->h.device) (void)hipSetDevice(prev);
  delete p;
  return 0;
}

int ffe_nstep_observe(ffe_nstep_handle p, const float *action_dev, const int32_t *step_type_dev, const float *reward_dev, const float *discount_dev,
                      const float *obs_dev, void *stream) {
  if (!p || !step_type_dev || !reward_dev || !discount_dev || !obs_dev || !action_dev) return -1;
  int prev = -1;
  (void)hipGetDevice(&prev);
  if (prev != p->h.device) (void)hipSetDevice(p->h.device);
  hipLaunchKernelGGL(ffn::nstep_observe_kernel, dim3((p->h.d.batch + ffn::kEnvsPerBlock - 1) / ffn::kEnvsPerBlock), dim3(64 * ffn::kEnvsPerBlock), 0, static_cast<hipStream_t>(stream), p->h.d, action_dev, step_type_dev,
                     reward_dev, discount_dev, obs_dev);
  const hipError_t e = hipGetLastError();
  if (prev >= 0 && prev != p->h.device) (void)hipSetDevice(prev);
  if (e != hipSuccess) { p->h.err = hipGetErrorString(e); return -2; }
  return 0;
}

int ffe_nstep_buffers(ffe_nstep_handle p, float **obs, float **act, float **ret, float **disc, float **next_obs, unsigned long long **written_dev) {
  if (!p) return -1;
  const ffn::Dev &D = p->h.d;
  if (obs) *obs = D.t_obs;
  if (act) *act = D.t_act;
  if (ret) *ret = D.t_ret;
  if (disc) *disc = D.t_disc;
  if (next_obs) *next_obs = D.t_next;
  if (written_dev) *written_dev = D.written;
  return 0;
}

const char *ffe_nstep_last_error(ffe_nstep_handle p) { return p ? p->h.err.c_str() : g_nerr.c_str(); }

int ffe_pack_timestep(const float *obs_dev, const float *reward_dev, const float *discount_dev, const int32_t *step_type_dev, float *packed_dev,
                      int batch, int obs_dim, void *stream) {
  if (!obs_dev || !reward_dev || !discount_dev || !step_type_dev || !packed_dev || batch <= 0 || obs_dim <= 0) return -1;
  const long long total = (long long)batch * (obs_dim + 3);
  hipLaunchKernelGGL(ffn::pack_timestep_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), obs_dev, reward_dev,
                     discount_dev, step_type_dev, packed_dev, batch, obs_dim);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}

int ffe_episode_stats(const int32_t *step_type_dev, const float *reward_dev, float *episode_return_dev, long long *episode_length_dev,
                      long long *totals_i64_dev, double *total_return_dev, int batch, void *stream) {
  if (!step_type_dev || !reward_dev || !episode_return_dev || !episode_length_dev || !totals_i64_dev || !total_return_dev || batch <= 0) return -1;
  hipLaunchKernelGGL(ffn::episode_stats_kernel, dim3((batch + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream), step_type_dev, reward_dev,
                     episode_return_dev, episode_length_dev, totals_i64_dev, total_return_dev, batch);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}

}  // extern "C"
