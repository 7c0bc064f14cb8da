// ball_env.hpp - internal C++ interface between the C ABI (fly_env.hip) and the walk_on_ball kernels (ball_env.hip).
#pragma once
#include <cstddef>
#include <cstdint>
#include <string>

namespace ffb {

struct BallEnv;  // opaque

struct BallTaskHost {
  int time_limit_steps, pad_first_obs, physics_flags, canonical_actions, clip_actions;
  double control_timestep;
};

// All functions throw std::runtime_error on failure; the C ABI wrappers translate that into error codes.
BallEnv *ball_create(const void *blob, size_t blob_size, const BallTaskHost &task, int batch, int device);
void ball_destroy(BallEnv *e);
void ball_spec(const BallEnv *e, int *nq, int *nv, int *nu, int *action_dim, int *obs_dim, int *nsub, double *h, double *ctrl_dt);
void ball_action_bounds(const BallEnv *e, float *mn, float *mx);
// mode 0 step, 1 reset all, 2 bare physics (nphys steps), 3 reset the envs whose mask byte is set
void ball_launch(BallEnv *e, const float *act, float *obs, float *rew, float *disc, int32_t *st, void *stream, int mode, int nphys,
                 const uint8_t *mask);
void ball_get_state(BallEnv *e, double *qpos, double *qvel, void *stream);
void ball_set_state(BallEnv *e, const double *qpos, const double *qvel, void *stream);
void ball_get_act(BallEnv *e, double *act, void *stream);
void ball_set_act(BallEnv *e, const double *act, void *stream);
void ball_get_task_state(BallEnv *e, int32_t *ints, double *reals, void *stream);
float ball_time_steps(BallEnv *e, const float *act, float *obs, float *rew, float *disc, int32_t *st, int iters, void *stream);
float ball_time_kernel(BallEnv *e, const float *act, float *obs, float *rew, float *disc, int32_t *st, int iters, void *stream);

}  // namespace ffb
