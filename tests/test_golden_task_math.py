"""Oracle + host task maths against golden vectors generated from the reference's own numpy modules
(tools/gen_golden.py imports vnl_ray.quaternions and vnl_ray.tasks.pattern_generators)."""
import hashlib

import numpy as np

from oracle import oracle as O


def test_wbpg_tables_bit_exact(golden_wbpg, wb_tables):
    g, T = golden_wbpg, wb_tables
    assert np.array_equal(T.beat_freqs, g["beat_freqs"])
    assert np.array_equal(T.n_repeats, g["n_repeats"])
    assert np.array_equal(T.rel_errors, g["rel_errors"])
    assert np.array_equal(np.diff(T.tab_off), g["table_len"])
    assert T.rate == float(g["rate"])
    assert hashlib.sha256(np.ascontiguousarray(T.traj).tobytes()).digest() == g["traj_sha256"].tobytes()
    assert hashlib.sha256(np.ascontiguousarray(T.phase).tobytes()).digest() == g["phase_sha256"].tobytes()
    for k in g["keep"]:
        tr, ph = T.table(int(k))
        assert np.array_equal(tr, g[f"traj_{k}"]) and np.array_equal(ph, g[f"phase_{k}"])


def _env(oracle_model, wb_tables, ref_traj):
    return O.OracleFlightEnv(oracle_model, wb_tables, *ref_traj)


def test_wbpg_reset_matches_reference(golden_wbpg, oracle_model, wb_tables, ref_traj):
    g = golden_wbpg
    env = _env(oracle_model, wb_tables, ref_traj)
    for p, q, v, s in zip(g["reset_phases"], g["reset_qpos"], g["reset_qvel"], g["reset_step"]):
        qq, vv = env.wbpg_reset(p)
        assert np.array_equal(qq, q) and np.array_equal(vv, v)
        assert env.wbpg_state()[0] == s


def test_wbpg_trace_matches_reference(golden_wbpg, oracle_model, wb_tables, ref_traj):
    """5000 steps incl. 3536 table switches: internal state and returned angles are bit-identical."""
    g = golden_wbpg
    env = _env(oracle_model, wb_tables, ref_traj)
    env.wbpg_reset(float(g["trace_phase0"]))
    for i, a in enumerate(g["trace_act"]):
        ang = env.wbpg_step(wb_tables.base_freq * (1 + wb_tables.rel_range * a))
        st, fi, cf = env.wbpg_state()
        assert st == g["trace_step"][i] and fi == g["trace_freq_idx"][i], i
        assert cf == g["trace_ctrl_freq"][i], i
        assert np.array_equal(ang, g["trace_angles"][i]), i


def test_oracle_quaternion_helpers(golden_quat):
    g = golden_quat
    n = len(g["q1"])
    for i in range(n):
        np.testing.assert_allclose(O.test_quat(0, g["q1"][i], g["q2"][i]), g["mult_quat"][i], rtol=0, atol=1e-15)
        np.testing.assert_allclose(O.test_quat(1, g["q1"][i]), g["reciprocal_quat"][i], rtol=1e-15, atol=1e-16)
        np.testing.assert_allclose(O.test_quat(2, g["v"][i], g["u1"][i], 3), g["rotate_vec_with_quat"][i], rtol=0, atol=2e-15)
        np.testing.assert_allclose(O.test_quat(2, g["v"][i], g["q1"][i], 3), g["rotate_vec_nonunit"][i], rtol=0, atol=4e-15)
        np.testing.assert_allclose(O.test_quat(3, g["u1"][i], g["u2"][i], 1)[0], g["quat_dist_short_arc"][i], rtol=0, atol=1e-14)
        np.testing.assert_allclose(O.test_quat(4, g["u1"][i], g["u2"][i]), g["get_dquat_local"][i], rtol=0, atol=2e-15)
        a = np.concatenate((g["root"][i], g["u1"][i]))
        np.testing.assert_allclose(O.test_quat(5, a, None, 3), g["root2com"][i], rtol=0, atol=2e-15)
    ident = np.array([1.0, 0, 0, 0])
    for i in range(n):
        np.testing.assert_allclose(O.test_quat(3, ident, g["u2"][i], 1)[0], g["quat_dist_identity"][i], rtol=0, atol=1e-14)
        # identical quaternions: arccos argument must be clamped to 1 exactly as the reference does
        assert np.isfinite(O.test_quat(3, g["u1"][i], g["u1"][i], 1)[0])


def test_host_com_root_transforms(golden_quat):
    from flybody_amd.tasks.trajectories import com2root, root2com

    g = golden_quat
    np.testing.assert_allclose(com2root(g["root"], g["u1"]), g["com2root"], rtol=0, atol=2e-15)
    rq = np.concatenate((g["root"], g["u1"]), axis=1)
    np.testing.assert_allclose(root2com(rq), g["root2com"], rtol=0, atol=2e-15)
    np.testing.assert_allclose(root2com(np.concatenate((com2root(g["root"], g["u1"]), g["u1"]), 1)), g["root"], atol=1e-14)
