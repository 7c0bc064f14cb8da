"""Convert a reference flight-trajectory HDF5 file to the `.npz` layout `fly_envs.flight_imitation(ref_path=...)` reads.

The reference stores `trajectories/<zero-padded idx>/{com_qpos (T,7), com_qvel (T,6)}` plus `timestep_seconds`
(`vnl_ray/tasks/trajectory_loaders.py:33-35,90-96`).  h5py is not available in the build image, so this runs wherever
the dataset lives:

    python tools/convert_hdf5_to_npz.py flight-dataset.hdf5 flight-dataset.npz [--min-len 3006]

Trajectories are truncated to the shortest kept length so they stack into (N, T, 7) / (N, T, 6).
"""
import argparse

import numpy as np


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("src")
    ap.add_argument("dst")
    ap.add_argument("--min-len", type=int, default=0, help="drop trajectories shorter than this many steps")
    args = ap.parse_args()
    import h5py  # noqa: deferred, not installed in the build image

    with h5py.File(args.src, "r") as f:
        dt = float(f["timestep_seconds"][()])
        keys = sorted(f["trajectories"].keys())
        qpos = [f["trajectories"][k]["com_qpos"][()] for k in keys]
        qvel = [f["trajectories"][k]["com_qvel"][()] for k in keys]
    keep = [i for i, q in enumerate(qpos) if len(q) >= max(args.min_len, 8)]
    if not keep:
        raise SystemExit("no trajectory is long enough")
    t = min(len(qpos[i]) for i in keep)
    np.savez_compressed(args.dst, com_qpos=np.stack([qpos[i][:t] for i in keep]).astype(np.float64),
                        com_qvel=np.stack([qvel[i][:t] for i in keep]).astype(np.float64), timestep_seconds=np.float64(dt),
                        source_index=np.array(keep))
    print(f"wrote {args.dst}: {len(keep)} trajectories x {t} steps @ {dt} s")


if __name__ == "__main__":
    main()
