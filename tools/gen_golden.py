"""Generate golden vectors by importing the reference's own numpy-only modules.

Runs only in the build container (needs /root/reference).  Outputs small fixtures under
tests/golden/ that travel to the GPU box; the reference itself never does.

    PYTHONDONTWRITEBYTECODE=1 python tools/gen_golden.py

Covered: vnl_ray.quaternions (all functions on the fly path),
vnl_ray.tasks.pattern_generators.WingBeatPatternGenerator (tables + a 5000-step reset/step trace
with table switching), and root2com/com2root re-typed on top of the imported
quaternions.rotate_vec_with_quat (task_utils itself pulls TensorFlow and cannot be imported).
"""
import hashlib
import os
import sys
import tempfile

import numpy as np

sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference")
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))

from vnl_ray import quaternions as RQ  # noqa: E402
from vnl_ray.tasks import rewards as RR  # noqa: E402
from vnl_ray.tasks.pattern_generators import WingBeatPatternGenerator  # noqa: E402

from flybody_amd.tasks.synthetic import base_wing_pattern  # noqa: E402

OUT = os.path.join(os.path.dirname(__file__), "..", "tests", "golden")


def quaternion_goldens():
    rng = np.random.RandomState(1234)
    n = 64
    q1 = rng.randn(n, 4)
    q2 = rng.randn(n, 4)
    u1 = q1 / np.linalg.norm(q1, axis=-1, keepdims=True)
    u2 = q2 / np.linalg.norm(q2, axis=-1, keepdims=True)
    v = rng.randn(n, 3)
    root = rng.randn(n, 3)
    ang = rng.uniform(-np.pi, np.pi, n)
    zedge = np.array([[0.0, 0, 0], [0, 0, 1], [0, 0, -1], [0, 0, 2.5], [1, 0, 0], [0.3, -0.2, 0.9]])
    out = dict(
        q1=q1, q2=q2, u1=u1, u2=u2, v=v, root=root, ang=ang, zedge=zedge,
        mult_quat=RQ.mult_quat(q1, q2),
        conj_quat=RQ.conj_quat(q1),
        reciprocal_quat=RQ.reciprocal_quat(q1),
        rotate_vec_with_quat=RQ.rotate_vec_with_quat(v, u1),
        rotate_vec_nonunit=RQ.rotate_vec_with_quat(v, q1),
        get_dquat=RQ.get_dquat(u1, u2),
        get_dquat_local=RQ.get_dquat_local(u1, u2),
        get_dquat_local_bcast=RQ.get_dquat_local(u1[0], u2[:6]),
        quat_dist_short_arc=RQ.quat_dist_short_arc(u1, u2),
        quat_dist_identity=RQ.quat_dist_short_arc(np.array([1.0, 0, 0, 0]), u2),
        quat_dist_same=RQ.quat_dist_short_arc(u1, u1),
        get_egocentric_vec=RQ.get_egocentric_vec(root, v, u1),
        vec_world_to_local=RQ.vec_world_to_local(v, u1),
        vec_world_to_local_hover=RQ.vec_world_to_local(v, u1, np.array([0.915, 0, 0.403, 0])),
        quat_z2vec=RQ.quat_z2vec(v),
        quat_z2vec_edge=RQ.quat_z2vec(zedge),
        axis_angle_to_quat=RQ.axis_angle_to_quat(v, ang),
        joint_orientation_quat=RQ.joint_orientation_quat(v, ang),
        quat_to_angvel=RQ.quat_to_angvel(u1, dt=2e-4),
        log_quat=RQ.log_quat(q1),
    )
    # re-typed task_utils.root2com / com2root (task_utils.py:174-213) over the imported rotate
    offset = np.array([-0.03697732, 0.00029205, -0.0142447])
    root_qpos = np.concatenate((root, u1), axis=1)
    out["root2com"] = np.stack([root_qpos[i, :3] + RQ.rotate_vec_with_quat(offset, root_qpos[i, 3:]) for i in range(n)])
    out["com2root"] = root + RQ.rotate_vec_with_quat(-offset, u1)
    np.savez_compressed(os.path.join(OUT, "quaternions.npz"), **out)
    print("quaternions.npz:", len(out), "arrays")


def wbpg_goldens():
    pattern = base_wing_pattern()
    with tempfile.NamedTemporaryFile(suffix=".npy", delete=False) as f:
        np.save(f, pattern)
        path = f.name
    gen = WingBeatPatternGenerator(base_pattern_path=path)
    os.unlink(path)
    lens = np.array([t["traj"].shape[0] for t in gen.traj_ctrl])
    traj = np.concatenate([t["traj"] for t in gen.traj_ctrl], 0)
    phase = np.concatenate([t["phase"] for t in gen.traj_ctrl], 0)
    keep = [0, 9, 100, 200]
    out = dict(
        pattern=pattern, beat_freqs=gen.beat_freqs, n_repeats=np.array(gen._n_repeats), rel_errors=np.array(gen._rel_errors),
        table_len=lens, rate=np.array(gen._rate), keep=np.array(keep),
        traj_sha256=np.frombuffer(hashlib.sha256(np.ascontiguousarray(traj).tobytes()).digest(), dtype=np.uint8),
        phase_sha256=np.frombuffer(hashlib.sha256(np.ascontiguousarray(phase).tobytes()).digest(), dtype=np.uint8),
        traj_colsum=traj.sum(0), phase_sum=np.array(phase.sum()),
    )
    for k in keep:
        out[f"traj_{k}"] = gen.traj_ctrl[k]["traj"]
        out[f"phase_{k}"] = gen.traj_ctrl[k]["phase"]
    # reset at several phases (avoid the reference's unguarded [step+1] on a last row)
    phases = np.array([0.0, 0.1, 0.25, 0.5, 0.77, 0.93])
    rq, rv, rs = [], [], []
    for p in phases:
        qp, qv = gen.reset(initial_phase=p, return_qvel=True)
        rq.append(qp.copy()); rv.append(qv.copy()); rs.append(gen._step)
    out.update(reset_phases=phases, reset_qpos=np.array(rq), reset_qvel=np.array(rv), reset_step=np.array(rs))
    # 5000-step trace under U(-1,1) user actions (table switches on most steps), then slow sweeps
    rng = np.random.RandomState(7)
    act = np.concatenate([rng.uniform(-1, 1, 3000), np.sin(np.arange(1000) * 0.01), np.full(500, 1.0), np.full(500, -1.0)])
    gen.reset(initial_phase=0.3)
    angles, st, fi, cf = [], [], [], []
    for a in act:
        ang = gen.step(ctrl_freq=gen.base_beat_freq * (1 + gen.rel_freq_range * a))
        angles.append(ang.copy()); st.append(gen._step); fi.append(gen._freq_idx); cf.append(gen._ctrl_freq)
    out.update(trace_act=act, trace_angles=np.array(angles), trace_step=np.array(st), trace_freq_idx=np.array(fi),
               trace_ctrl_freq=np.array(cf), trace_phase0=np.array(0.3))
    np.savez_compressed(os.path.join(OUT, "wbpg.npz"), **out)
    print("wbpg.npz: rows", traj.shape[0], "switches", int((np.diff(fi) != 0).sum()))


def rewards_goldens():
    """vnl_ray.tasks.rewards (walking imitation): compute_diffs, reward_factors_deep_mimic and
    get_reference_features on seeded features shaped like the fly's (108 dofs, 6 end-effector sites, root + 102 joint quaternions)."""
    rng = np.random.RandomState(4321)
    ncase, nv, nsite, nj = 12, 108, 6, 102
    out = dict(ncase=np.array(ncase))

    def unit(x):
        return x / np.linalg.norm(x, axis=-1, keepdims=True)

    T = 9
    ref = dict(qpos=np.concatenate((rng.randn(T, 3) * 0.1, unit(rng.randn(T, 4)), rng.uniform(-1, 1, (T, nj))), axis=1),
               qvel=rng.randn(T, nv) * 20.0, root2site=rng.randn(T, nsite, 3) * 0.1, joint_quat=unit(rng.randn(T, nj, 4)))
    for k, v in ref.items():
        out["ref_" + k] = v
    for c in range(ncase):
        step = c % T
        rf = RR.get_reference_features(ref, step)
        for k, v in rf.items():
            out[f"reffeat{c}_{k}"] = np.asarray(v)
        scale = [0.0, 1e-3, 1e-2, 0.05, 0.2, 1.0][c % 6]   # from identical to unrelated
        wf = {
            "com": rf["com"] + scale * rng.randn(3) * 0.1,
            "qvel": rf["qvel"] + scale * rng.randn(nv) * 30.0,
            "root2site": rf["root2site"] + scale * rng.randn(nsite, 3) * 0.1,
            "joint_quat": unit(rf["joint_quat"] + scale * rng.randn(nj + 1, 4)) * (1.0 if c % 2 else rng.uniform(0.5, 2.0)),  # scale-invariant
        }
        for k, v in wf.items():
            out[f"walker{c}_{k}"] = v
        for n in (1, 2):
            d = RR.compute_diffs(wf, rf, n=n)
            out[f"diffs{c}_n{n}"] = np.array([d[k] for k in ("com", "qvel", "root2site", "joint_quat")])
        out[f"factors{c}"] = RR.reward_factors_deep_mimic(wf, rf)
        std = {"com": 0.05, "qvel": 30.0, "root2site": 0.1, "joint_quat": 0.8}
        out[f"factors{c}_custom"] = RR.reward_factors_deep_mimic(wf, rf, std=std, weights=(1.0, 0.5, 2.0, 0.25))
    np.savez_compressed(os.path.join(OUT, "rewards.npz"), **out)
    print("rewards.npz:", len(out), "arrays")


def walker_feature_goldens():
    """The quaternion helpers behind vnl_ray.tasks.rewards.get_walker_features (walk_imitation's reward): quat_z2vec (with its
    edge cases), axis_angle_to_quat, joint_orientation_quat, get_egocentric_vec; and the exact sequence get_walker_features applies
    to joint axes (rewards.py:45-52) on seeded stand-ins for physics.bind(...).xaxis / qpos."""
    rng = np.random.RandomState(97)
    n = 48
    vec = rng.randn(n, 3)
    vec[0] = [0.0, 0.0, 2.0]; vec[1] = [0.0, 0.0, -0.5]; vec[2] = [0.0, 0.0, 0.0]
    ang = rng.uniform(-np.pi, np.pi, n)
    root_quat = rng.randn(4); root_quat /= np.linalg.norm(root_quat)
    root_pos, sites = rng.randn(3), rng.randn(6, 3)
    xaxis = rng.randn(n, 3); xaxis /= np.linalg.norm(xaxis, axis=-1, keepdims=True)
    local = RQ.rotate_vec_with_quat(xaxis, RQ.reciprocal_quat(root_quat))
    out = {"vec": vec, "ang": ang, "z2vec": RQ.quat_z2vec(vec), "axis_angle": RQ.axis_angle_to_quat(xaxis, ang),
           "joint_orientation": RQ.joint_orientation_quat(xaxis, ang), "root_quat": root_quat, "root_pos": root_pos, "sites": sites,
           "egocentric": RQ.get_egocentric_vec(root_pos, sites, root_quat), "xaxis": xaxis,
           "joint_quat_local": RQ.joint_orientation_quat(local, ang)}
    np.savez_compressed(os.path.join(OUT, "walker_features.npz"), **out)
    print("walker_features.npz:", len(out), "arrays")


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    quaternion_goldens()
    wbpg_goldens()
    rewards_goldens()
    walker_feature_goldens()
