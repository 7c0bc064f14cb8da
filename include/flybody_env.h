/*
 * flybody_env.h - C ABI of the MI355X-native batched fruit-fly environment.
 *
 * The reference has no FFI on this path: its boundary is the Python dm_env.Environment protocol
 * (SURVEY.md section 8b).  Each entry point below therefore names the reference *Python* interface
 * it stands in for; the ctypes binding a maintainer adds on the reference side is shown in
 * INTEGRATION.md, and flybody_amd/batched_env.py is that binding in this repo.
 *
 * Conventions
 *   - every function returns 0 on success, <0 on error (ffe_last_error() has the text); nothing throws;
 *   - all "dev" pointers are device (HBM) buffers owned by the caller; the library owns env state;
 *   - `stream` is a hipStream_t passed as void*; calls are asynchronous on it and never synchronise;
 *   - one handle per (device, stream); handles are not thread-safe; every entry point runs on the handle's device
 *     whatever the caller's current device is, and leaves the caller's current device as it found it;
 *   - layouts are row-major with the env index leading: act[B][A], obs[B][O].
 */
#ifndef FLYBODY_ENV_H_
#define FLYBODY_ENV_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ffe_env *ffe_handle;

enum { FFE_STEP_FIRST = 0, FFE_STEP_MID = 1, FFE_STEP_LAST = 2 }; /* dm_env.StepType */

/* physics switches (tests / BASELINE config 2 "constraints off"); 0 = everything on */
enum {
  FFE_NO_FLUID = 1, FFE_NO_LIMIT = 2, FFE_NO_DAMPER = 4, FFE_NO_SPRING = 8, FFE_NO_GRAVITY = 16, FFE_NO_ACTUATION = 32,
  FFE_NO_CONTACT = 64, FFE_NO_NOSLIP = 128, FFE_NO_ADHESION = 256 /* walk_on_ball only */
};

/* Task inputs of fly_envs.flight_imitation (vnl_ray/fly_envs.py:29-72).  All host pointers, float64,
 * copied during ffe_create. */
typedef struct {
  /* WingBeatPatternGenerator tables (vnl_ray/tasks/pattern_generators.py:18-119), built by the host */
  int32_t wb_nfreq;            /* number of beat frequencies (201) */
  const double *wb_beat_freqs; /* [nfreq] */
  const int32_t *wb_tab_off;   /* [nfreq+1] row offsets */
  const double *wb_traj;       /* [rows][6] wing angles */
  const double *wb_phase;      /* [rows] */
  double wb_base_freq, wb_rel_range, wb_rate, wb_dt_ctrl;
  /* reference trajectories after the per-episode preprocessing of flight_imitation.py:97-104.  The reference serves
   * trajectories of different lengths (trajectory_loaders.py:98-100,124-129): rows of all trajectories are concatenated
   * and `traj_off[ntraj + 1]` holds the first row of each; traj_off == NULL means every trajectory has `traj_len` rows. */
  int32_t ntraj, traj_len;
  const double *ref_qpos; /* [rows][7] ghost root pose */
  const double *ref_qvel; /* [rows][6] */
  const int32_t *traj_off; /* [ntraj + 1] or NULL */
  int32_t future_steps;     /* fly_envs.py:65 (5) */
  int32_t time_limit_steps; /* round(time_limit / control_timestep), fly_envs.py:54 (3000): caps `_traj_timesteps`
                               (flight_imitation.py:107-108: min(len(trajectory), this) - (future_steps + 1)) */
  int32_t episode_limit_steps; /* control steps after which composer.Environment's `physics.time() >= time_limit` fires;
                                  MuJoCo's time is a float64 running sum of the physics timestep, so this is 3001, not
                                  3000, for 0.6 s at 5e-5 s (flybody_amd/batched_env.py:time_limit_control_steps);
                                  <= 0 means time_limit_steps */
  double terminal_com_dist; /* fly_envs.py:33 (2.0) */
  double ghost_accel_z;     /* gravity felt by the armature-1 ghost, cm/s^2 */
  int32_t pad_first_obs;    /* 0 = dm_control zero-padded sensor buffers at reset */
  int32_t physics_flags;    /* FFE_NO_* */
  /* acme.wrappers.CanonicalSpecWrapper folded in (train_dmpo_ray.py:128-129; same map as tasks/task_utils.py:53-76
   * canonical2real): actions arrive in [-1,1] and are mapped to lo + (a+1)/2 (hi-lo); optional clip to [-1,1] first */
  int32_t canonical_actions;
  int32_t clip_actions;
} ffe_flight_task;

/* Task inputs of fly_envs.walk_on_ball (vnl_ray/fly_envs.py:125-157).  The arena (ball position / radius / density,
 * tasks/arenas/ball.py:61-69), the actuator filters (joint_filter 0.01, adhesion_filter 0.007) and the claw friction
 * (tasks/walk_on_ball.py:21,41-42) are compiled into the model blob (flybody_amd/assets/fly_ball.ffmb). */
typedef struct {
  double control_timestep;  /* tasks/constants.py:17 (2e-3 s; the model carries the 2e-4 s physics step) */
  int32_t time_limit_steps; /* control steps after which `physics.time() >= time_limit` fires (fly_envs.py:144, 2.0 s):
                               1001 - ten thousand float64 additions of 2e-4 give 1.9999999999998 - see
                               flybody_amd/batched_env.py:time_limit_control_steps */
  int32_t pad_first_obs;    /* 0 = dm_control zero-padded sensor buffers at reset */
  int32_t physics_flags;    /* FFE_NO_* */
  int32_t canonical_actions, clip_actions; /* acme.wrappers.CanonicalSpecWrapper folded in, as in ffe_flight_task */
} ffe_ball_task;

typedef struct {
  int32_t batch, nq, nv, nu, action_dim, obs_dim, nsub;
  double physics_timestep, control_timestep;
  /* observation layout: offsets into the obs row, in the order dm_control emits the walker observables */
  int32_t off_accelerometer, off_gyro, off_joints_pos, off_joints_vel, off_velocimeter, off_world_zaxis,
      off_ref_displacement, off_ref_root_quat, n_obs_joints, n_ref;
} ffe_spec_t;

/* fly_envs.flight_imitation(...) -> composer.Environment (fly_envs.py:29-72); environment_factory() call
 * site agents/ray_distributed_dmpo.py:314.  `model_blob` is the compiled model (flybody_amd/assets). */
int ffe_create_flight(const void *model_blob, size_t blob_size, const ffe_flight_task *task, int batch, int device,
                      uint64_t seed, uint64_t env_id_base, ffe_handle *out);
/* fly_envs.walk_on_ball(...) -> composer.Environment (fly_envs.py:125-157): tethered fly walking on a floating ball
 * (tasks/walk_on_ball.py:16-95; physics via MuJoCo mj_step with contacts, elliptic cones, noslip).  The returned handle
 * works with ffe_spec / ffe_action_bounds / ffe_reset / ffe_step / ffe_physics_step / ffe_get_state / ffe_set_state /
 * ffe_get_task_state / ffe_time_steps / ffe_destroy.  Observation row (289 floats): accelerometer 3 |
 * actuator_activation 59 | appendages_pos 21 | ball_qvel 3 | force 18 | gyro 3 | joints_pos 85 | joints_vel 85 |
 * touch 6 | velocimeter 3 | world_zaxis 3.  State layout: qpos[106] = ball quaternion, then the 102 hinges;
 * qvel[105] = ball angular velocity (body frame), then the hinges. */
int ffe_create_walk_on_ball(const void *model_blob, size_t blob_size, const ffe_ball_task *task, int batch, int device,
                            ffe_handle *out);
/* physics.data.act (actuator activations, [B][nu] float64 device buffers) of a walk_on_ball handle */
int ffe_get_act(ffe_handle h, double *act_dev, void *stream);
int ffe_set_act(ffe_handle h, const double *act_dev, void *stream);
int ffe_destroy(ffe_handle h);

/* observation_spec()/action_spec()/reward_spec()/discount_spec() (ray_distributed_dmpo.py:315) */
int ffe_spec(ffe_handle h, ffe_spec_t *spec);
/* FruitFly.get_action_spec bounds (fruitfly/fruitfly.py:496-526); host arrays of action_dim floats */
int ffe_action_bounds(ffe_handle h, float *minimum, float *maximum);

/* Environment.reset() (acme loop, ray_distributed_dmpo.py:404): every env starts a new episode and reports
 * FIRST.  Outputs as in ffe_step. */
int ffe_reset(ffe_handle h, float *obs_dev, float *reward_dev, float *discount_dev, int32_t *step_type_dev,
              void *stream);
/* Environment.reset() of a subset (one composer.Environment per actor in the reference, each reset on its own:
 * ray_distributed_dmpo.py:401-404; evaluators restart episodes at will): envs with mask_dev[i] != 0 start a new
 * episode and report FIRST; the state and the output rows of the other envs are left untouched. */
int ffe_reset_envs(ffe_handle h, const uint8_t *mask_dev, float *obs_dev, float *reward_dev, float *discount_dev,
                   int32_t *step_type_dev, void *stream);
/* Environment.step(action) (ray_distributed_dmpo.py:404 via acme.EnvironmentLoop.run_episode): one control step of
 * every env = before_step, nsub physics substeps, reward, discount, termination, observation
 * (tasks/flight_imitation.py:149-220, tasks/base.py:190-217).  An env that returned LAST performs its reset on
 * this call instead and returns FIRST (reward 0, discount 1), exactly as dm_control auto-resets.
 * act_dev[B][action_dim] is in the raw action spec and is not modified. */
int ffe_step(ffe_handle h, const float *act_dev, float *obs_dev, float *reward_dev, float *discount_dev,
             int32_t *step_type_dev, void *stream);

/* physics.set_control(ctrl) + nsteps x physics.step() with no task layer (fruitfly/fruitfly.py:492 reaching MuJoCo's mj_step):
 * advances every env's (qpos, qvel) by `nsteps` physics steps under ctrl_dev[B][nu] (clamped to ctrlrange).  BASELINE config 2
 * ("free-flight, dynamics only"); combine with FFE_NO_LIMIT for "constraints off".  Produces no observation. */
int ffe_physics_step(ffe_handle h, const float *ctrl_dev, int nsteps, void *stream);

/* FlightImitationWBPG.set_next_trajectory_index (flight_imitation.py:87-91), plus the initial wing-beat phase
 * the reference draws from its RandomState (flight_imitation.py:137).  Host arrays [B]; traj_idx<0 keeps the
 * counter-based draw.  Applies to each env's next reset only.  The host arrays are staged on `stream` and may be reused
 * as soon as the call returns. */
int ffe_force_next_episode(ffe_handle h, const int32_t *traj_idx_host, const double *phase_host, void *stream);

/* physics.get_state()/set_state() analogue for parity tests: qpos[B][nq] (root position first), qvel[B][nv],
 * float64 device buffers.  set_state leaves task counters untouched. */
int ffe_get_state(ffe_handle h, double *qpos_dev, double *qvel_dev, void *stream);
int ffe_set_state(ffe_handle h, const double *qpos_dev, const double *qvel_dev, void *stream);
/* task-side state per env: {wbpg_step, wbpg_freq_idx, step_counter, traj_idx, needs_reset, n_active_limits,
 * solver_iters, contacts} int32[B][8] and {wbpg_ctrl_freq, ghost_pos[3], ghost_quat[4]} float64[B][8].
 * flight handles, int 7 (the fly's own contacts, for the parity tests): bits 0-7 contacts of the current position stage; bits
 * 8-15 non-zero when a position stage of the last control step met more contacts than the solver carries (6; the deepest are
 * kept); bits 16-31 the contacts each of the last step's four substeps used, 4 bits each.
 * walk_on_ball handles: {contact history lo, hi, step_counter, fly-fly contacts, needs_reset, contacts, solver_iters, overflow
 * (sticky over the episode; bit 0 more than 16 contacts, bit 1 more than 48 constraint rows, bit 2 more than 24 columns in one
 * block of M)} - the contact history holds, 4 bits per substep for the first 16 substeps of the last control step, the number of
 * contacts that received constraint rows; real 0 holds, 5 bits per substep for the first 12 substeps, the number of contacts
 * DETECTED inside their margin (adhesion is shared over those); the other reals are zero */
int ffe_get_task_state(ffe_handle h, int32_t *ints_dev, double *reals_dev, void *stream);

/* name and duration helper for bench.py's roofline: launches `iters` steps bracketed by HIP events on `stream`
 * and returns the mean milliseconds per ffe_step launch (synchronises the stream). */
int ffe_time_steps(ffe_handle h, const float *act_dev, float *obs_dev, float *reward_dev, float *discount_dev,
                   int32_t *step_type_dev, int iters, void *stream, float *ms_per_step);
/* the same for the step kernel alone (bench.py's roofline.kernel_ms; what rocprofv3 --kernel-trace reports for it): HIP events
 * immediately around each of `iters` launches of the step kernel, without the launch-order kernel that follows it; synchronises
 * after every launch */
int ffe_time_kernel(ffe_handle h, const float *act_dev, float *obs_dev, float *reward_dev, float *discount_dev,
                    int32_t *step_type_dev, int iters, void *stream, float *ms_per_kernel);

/* device unit test of the in-kernel quaternion helpers against vnl_ray/quaternions.py goldens:
 * op 0 mult_quat(a,b) 1 reciprocal_quat(a) 2 rotate_vec_with_quat(a.xyz,b) 3 quat_dist_short_arc(a,b)
 * 4 get_dquat_local(a,b); a,b,out are device float[n][4] */
int ffe_test_quat(int op, const float *a_dev, const float *b_dev, float *out_dev, int n, void *stream);

const char *ffe_last_error(ffe_handle h);
const char *ffe_version(void);

/* ---- bulk n-step transition writer: the adder the reference's actors feed, batched and device-resident.
 * Replaces acme.adders.reverb.NStepTransitionAdder(n_step, discount) as built at agents/ray_distributed_dmpo.py:514-521
 * (n_step = 50, train_dmpo_ray.py:236) and driven through actor.observe_first / actor.observe (agents/actors.py:91-101).
 * One call per env step with the action that was applied and the timestep ffe_step returned; FIRST rows start an episode
 * (their action is ignored).  Per env each call writes the transition from the oldest held entry (at most n steps back) to the
 * new observation, (o_s, a_s, R, D, o_t+1) with R = r_0 + g d_0 r_1 + g^2 d_0 d_1 r_2 + ... and D = g^(m-1) d_0 ... d_(m-1) over the
 * m <= n entries spanned - like acme's adder it does not wait for n entries, so an episode's first n - 1 steps yield the short
 * transitions (o_0 -> o_1), (o_0 -> o_2), ...; LAST also flushes the shorter tails (an episode of T steps leaves T + min(T, n) - 1).
 * Transitions go to a device replay ring of `capacity` slots (slot = count mod capacity).  acme is not in the reference tree:
 * these semantics restate its published behaviour (parity unpinned, tests/test_nstep.py). */
typedef struct ffe_nstep *ffe_nstep_handle;
int ffe_nstep_create(int batch, int obs_dim, int act_dim, int n_step, float discount, long long capacity, int device, ffe_nstep_handle *out);
int ffe_nstep_observe(ffe_nstep_handle h, const float *action_dev, const int32_t *step_type_dev, const float *reward_dev,
                      const float *discount_dev, const float *obs_dev, void *stream);
/* device pointers of the replay ring: obs[capacity][O], act[capacity][A], n-step return[capacity], discount[capacity],
 * next_obs[capacity][O], and the running count of transitions written */
int ffe_nstep_buffers(ffe_nstep_handle h, float **obs, float **act, float **ret, float **disc, float **next_obs, unsigned long long **written_dev);
int ffe_nstep_destroy(ffe_nstep_handle h);
/* The unit the per-step gather to a central learner moves (the reference has no collective - its actors
 * push transitions through Reverb, agents/ray_distributed_dmpo.py:106-115): one fused launch packs (obs, reward, discount,
 * step_type) of the current device into packed_dev[B][obs_dim + 3] (flybody_amd/distributed.py:TimestepGather). */
int ffe_pack_timestep(const float *obs_dev, const float *reward_dev, const float *discount_dev, const int32_t *step_type_dev,
                      float *packed_dev, int batch, int obs_dim, void *stream);

/* episode statistics of a batched actor loop, the ones the reference's EnvironmentLoop logs (agents/ray_distributed_dmpo.py:401-440:
 * episode_return, episode_length): per env the running return [B] float32 and length [B] int64 (a FIRST row adds nothing), and for
 * rows reporting LAST the batch totals {finished episodes, sum of their lengths} (int64[2]) and the sum of their returns (float64[1]);
 * the finished env's counters restart.  One launch per step, nothing read back. */
int ffe_episode_stats(const int32_t *step_type_dev, const float *reward_dev, float *episode_return_dev, long long *episode_length_dev,
                      long long *totals_i64_dev, double *total_return_dev, int batch, void *stream);
const char *ffe_nstep_last_error(ffe_nstep_handle h);

#ifdef __cplusplus
}
#endif
#endif /* FLYBODY_ENV_H_ */
