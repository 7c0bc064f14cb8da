// nstep.hip - bulk n-step transition writer on the device: the batched counterpart of the adder the reference's actors
// feed (acme.adders.reverb.NStepTransitionAdder(n_step=50, discount=gamma), built at agents/ray_distributed_dmpo.py:514-521
// and driven by actors.py:91-101 observe_first / observe).  acme is not in the reference tree or in this image: the
// semantics below restate its published behaviour and are checked against a numpy restatement only (parity unpinned).
//
// Per env and per observed step t (action a_t, next timestep (r, d, o_{t+1}, step_type)):
//   * the ring keeps the last n entries (o_s, a_s, r_{s+1}, d_{s+1});
//   * once n entries are held, the transition starting n steps back is written:
//       (o_s, a_s, R, D, o_{t+1}),  R = r_0 + g d_0 r_1 + g^2 d_0 d_1 r_2 + ...,  D = g^(n-1) d_0 d_1 ... d_(n-1)
//     (acme's _compute_cumulative_quantities: the env discounts multiply in, the learner applies one more g);
//   * on LAST the remaining, shorter, transitions are flushed as well (acme's _write_last), all ending in o_{t+1};
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

// writes the transition that starts at ring entry `s` (0 = oldest of `len` entries), spanning entries s .. len-1
__device__ __forceinline__ void emit(const Dev &D, int env, int lane, int first, int s, int len, const float *next_obs, unsigned long long slot) {
  const int n = D.n_step, O = D.obs_dim, A = D.act_dim;
  const float *rr = D.r_rew + (size_t)env * n, *rd = D.r_disc + (size_t)env * n;
  // serial over <= n terms in lane 0's order would be simplest; keep acme's order of operations (left to right) so the
  // float32 result matches a scalar restatement bit for bit
  float ret = 0.f, td = 1.f;
  if (lane == 0) {
#pragma clang fp contract(off)
    for (int i = s; i < len; i++) {
      const int e = (first + i) % n;
      if (i == s) { ret = rr[e]; td = rd[e]; }
      else { td *= D.gamma; ret += rr[e] * td; td *= rd[e]; }
    }
    D.t_ret[slot] = ret; D.t_disc[slot] = td;
  }
  const int e0 = (first + s) % n;
  const float *so = D.r_obs + ((size_t)env * n + e0) * O, *sa = D.r_act + ((size_t)env * n + e0) * A;
  for (int k = lane; k < O; k += 64) { D.t_obs[slot * O + k] = so[k]; D.t_next[slot * O + k] = next_obs[k]; }
  for (int k = lane; k < A; k += 64) D.t_act[slot * A + k] = sa[k];
}

__global__ __launch_bounds__(64) void nstep_observe_kernel(Dev D, const float *__restrict__ action, const int *__restrict__ step_type,
                                                           const float *__restrict__ reward, const float *__restrict__ discount,
                                                           const float *__restrict__ obs) {
  const int env = blockIdx.x, lane = threadIdx.x;
  if (env >= D.batch) return;
  const int n = D.n_step, O = D.obs_dim, A = D.act_dim;
  const float *o_next = obs + (size_t)env * O;
  float *lo = D.last_obs + (size_t)env * O;
  const int st = step_type[env];
  if (st == FFE_STEP_FIRST) {  // observe_first: new episode
    for (int k = lane; k < O; k += 64) lo[k] = o_next[k];
    if (lane == 0) { D.head[env] = 0; D.count[env] = 0; }
    return;
  }
  int head = D.head[env], cnt = D.count[env];
  // append (o_t, a_t, r_{t+1}, d_{t+1}); when the ring is full its oldest entry (already written out) is overwritten
  float *ro = D.r_obs + ((size_t)env * n + head) * O, *ra = D.r_act + ((size_t)env * n + head) * A;
  for (int k = lane; k < O; k += 64) ro[k] = lo[k];
  for (int k = lane; k < A; k += 64) ra[k] = action[(size_t)env * A + k];
  if (lane == 0) { D.r_rew[(size_t)env * n + head] = reward[env]; D.r_disc[(size_t)env * n + head] = discount[env]; }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  head = (head + 1) % n;
  cnt = cnt < n ? cnt + 1 : n;
  const int first = (head - cnt + n) % n;  // ring index of the oldest held entry
  // how many transitions this call writes: the full-length one (if n entries are held) and, on LAST, every shorter tail
  const int n_full = cnt == n ? 1 : 0;
  const int n_tail = st == FFE_STEP_LAST ? cnt - n_full : 0;
  const int total = n_full + n_tail;
  unsigned long long base = 0;
  if (total > 0) {
    if (lane == 0) base = atomicAdd(D.written, (unsigned long long)total);
    base = __shfl(base, 0);
    for (int j = 0; j < total; j++) {
      // j = 0 is the oldest start; with a full ring that is the n-step transition, the rest (LAST only) start later
      const int s = (n_full ? 0 : 0) + j;
      emit(D, env, lane, first, s, cnt, o_next, (base + j) % (unsigned long long)D.capacity);
    }
  }
  for (int k = lane; k < O; k += 64) lo[k] = o_next[k];
  if (lane == 0) { D.head[env] = head; D.count[env] = cnt; }
}

// one [B][O + 3] row per env: observation | reward | discount | step_type (as float): the unit the per-step gather moves
__global__ void pack_timestep_kernel(const float *__restrict__ obs, const float *__restrict__ rew, const float *__restrict__ disc,
                                     const int *__restrict__ st, float *__restrict__ out, int batch, int obs_dim) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x, w = obs_dim + 3;
  if (i >= (long long)batch * w) return;
  const int env = (int)(i / w), k = (int)(i - (long long)env * w);
  out[i] = k < obs_dim ? obs[(long long)env * obs_dim + k] : (k == obs_dim ? rew[env] : (k == obs_dim + 1 ? disc[env] : (float)st[env]));
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
  hipLaunchKernelGGL(ffn::nstep_observe_kernel, dim3(p->h.d.batch), dim3(64), 0, static_cast<hipStream_t>(stream), p->h.d, action_dev, step_type_dev,
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

}  // extern "C"
