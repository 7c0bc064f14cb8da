"""Convert a reference flight-trajectory HDF5 file to the `.npz` layout `fly_envs.flight_imitation(ref_path=...)` reads.

The reference stores `trajectories/<zero-padded idx>/{com_qpos (T_i,7), com_qvel (T_i,6)}` plus `timestep_seconds`
(`vnl_ray/tasks/trajectory_loaders.py:33-35,90-96`).  h5py is not available in the build image, so this runs wherever
the dataset lives:

    python tools/convert_hdf5_to_npz.py flight-dataset.hdf5 flight-dataset.npz [--min-len 8]

Every trajectory keeps its own length (the reference serves them ragged, `trajectory_loaders.py:98-100,124-129`, and
`flight_imitation.py:107-108` ends an episode by the length of the trajectory it drew): rows are concatenated and
`traj_off (N+1,)` records where each starts (`flybody_amd/tasks/trajectories.py:save_npz`).
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def convert(read_group, keys, dt, dst, min_len=8):
    """`read_group(key) -> (com_qpos, com_qvel)`; split out so the layout logic is testable without h5py."""
    import numpy as np

    from flybody_amd.tasks.trajectories import save_npz

    qpos, qvel, keep = [], [], []
    for i, k in enumerate(keys):
        q, v = read_group(k)
        q, v = np.asarray(q, dtype=np.float64), np.asarray(v, dtype=np.float64)
        assert q.ndim == 2 and q.shape[1] == 7 and v.shape == (len(q), 6), (k, q.shape, v.shape)
        if len(q) >= min_len:
            qpos.append(q); qvel.append(v); keep.append(i)
    if not keep:
        raise SystemExit("no trajectory is long enough")
    save_npz(dst, qpos, qvel, dt, source_index=np.array(keep))
    lens = [len(q) for q in qpos]
    return len(keep), min(lens), max(lens)


def convert_walking(read_snippet, n, lengths, joint_names, site_names, dst, dt=2e-3, min_len=66):
    """The walking dataset (`trajectory_loaders.py:135-223`): per snippet `root_qpos (T,7)`, `qpos (T,J)`, `root_qvel (T,6)`,
    `qvel (T,J)`, `root2site (T,S,3)`, `joint_quat (T,J,4)`, cut to `trajectory_lengths[i]` rows, x / y of the root re-based to the
    first row (`trajectory_loaders.py:206`).  `read_snippet(i) -> dict` of those arrays; kept apart from h5py for testing.  An
    episode needs future_steps + 2 = 66 rows (`walk_imitation.py:99-100`): shorter snippets are dropped."""
    import numpy as np

    from flybody_amd.tasks.walking import save_npz

    out, keep = [], []
    for i in range(n):
        g, L = read_snippet(i), int(lengths[i])
        if L < min_len:
            continue
        qpos = np.concatenate((np.asarray(g["root_qpos"], dtype=np.float64)[:L], np.asarray(g["qpos"], dtype=np.float64)[:L]), axis=1)
        qvel = np.concatenate((np.asarray(g["root_qvel"], dtype=np.float64)[:L], np.asarray(g["qvel"], dtype=np.float64)[:L]), axis=1)
        qpos[:, :2] -= qpos[0, :2]
        out.append({"qpos": qpos, "qvel": qvel, "root2site": np.asarray(g["root2site"], dtype=np.float64)[:L],
                    "joint_quat": np.asarray(g["joint_quat"], dtype=np.float64)[:L]})
        keep.append(i)
    if not out:
        raise SystemExit("no snippet is long enough")
    save_npz(dst, out, joint_names, site_names, dt)
    return len(keep), min(len(s["qpos"]) for s in out), max(len(s["qpos"]) for s in out)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("src")
    ap.add_argument("dst")
    ap.add_argument("--walking", action="store_true", help="the source is the walking imitation dataset (snippets with tracked joints and sites)")
    ap.add_argument("--min-len", type=int, default=8, help="drop trajectories shorter than this many steps (future_steps + 2 is the minimum an episode needs)")
    args = ap.parse_args()
    import h5py  # noqa: deferred, not installed in the build image

    if args.walking:
        with h5py.File(args.src, "r") as f:
            lens = f["trajectory_lengths"][()]
            n = len(lens)
            nz = len(str(n))
            jn = [s.decode("utf-8") for s in f["id2name"]["joints"]]
            sn = [s.decode("utf-8") for s in f["id2name"]["sites"]]
            rd = lambda i: {k: f["trajectories"][str(i).zfill(nz)][k][()] for k in ("root_qpos", "qpos", "root_qvel", "qvel", "root2site", "joint_quat")}
            n_kept, lo, hi = convert_walking(rd, n, lens, jn, sn, args.dst, min_len=max(args.min_len, 66))
        print(f"wrote {args.dst}: {n_kept} snippets, {lo}..{hi} rows")
        return
    with h5py.File(args.src, "r") as f:
        dt = float(f["timestep_seconds"][()])
        n = len(f["trajectories"])
        keys = [str(i).zfill(len(str(n))) for i in range(n)]  # trajectory_loaders.py:91-93
        n_kept, lo, hi = convert(lambda k: (f["trajectories"][k]["com_qpos"][()], f["trajectories"][k]["com_qvel"][()]), keys, dt, args.dst, args.min_len)
    print(f"wrote {args.dst}: {n_kept} trajectories, {lo}..{hi} steps @ {dt} s")


if __name__ == "__main__":
    main()
