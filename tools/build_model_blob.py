"""Compile the reference's fruit-fly MJCF into the committed model blob.

Runs only where `/root/reference` exists (this container); the GPU box uses the committed
`flybody_amd/assets/fly_flight.ffmb` + `fly_flight.json`.

    python tools/build_model_blob.py
"""
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))

from flybody_amd.model.blob import model_tensors, write_blob
from flybody_amd.model.compiler import build_ball_model, build_flight_model, build_walk_model

OUT = os.path.join(os.path.dirname(__file__), "..", "flybody_amd", "assets")


def main():
    m, L = build_flight_model()
    # with its collision geoms: the reference leaves fly - fly collisions on in flight (tasks/base.py:299-302 disables the floor only)
    write_blob(os.path.join(OUT, "fly_flight.ffmb"), model_tensors(m, L, with_collision=True))
    meta = {
        "body_name": m.body_name,
        "geom_name": m.geom_name,
        "jnt_name": m.jnt_name,
        "act_name": m.act_name,
        "ten_name": m.ten_name,
        "observable_joints": m.walker["observable_joints"],
        "action_names": [m.act_name[i] for c in ("adhesion", "head", "mouth", "antennae", "wings", "abdomen", "legs")
                         for i in (m.walker["ctrl_indices"][c] or [])] + ["user_0"],
        "notes": {k: (float(v) if not isinstance(v, (list, str)) else v) for k, v in m.notes.items()},
    }
    with open(os.path.join(OUT, "fly_flight.json"), "w") as f:
        json.dump(meta, f, indent=1)
    print("wrote", OUT, "nbody", m.nbody, "nv", m.nv, "nu", m.nu)
    # the flight model WITHOUT its collision geoms: the oracle twin of `ffe_physics_step`'s contact-free BASELINE config 2 and of the
    # known-answer physics tests (tests/test_oracle_physics.py)
    ocol = os.path.join(os.path.dirname(__file__), "..", "oracle", "assets")
    os.makedirs(ocol, exist_ok=True)
    write_blob(os.path.join(ocol, "fly_flight_nocollision.ffmb"), model_tensors(m, L))
    b = build_ball_model()
    write_blob(os.path.join(OUT, "fly_ball.ffmb"), model_tensors(b, with_collision=True))
    meta = {
        "body_name": b.body_name, "jnt_name": b.jnt_name, "act_name": b.act_name, "ten_name": b.ten_name,
        "geom_name": b.geom_name, "site_name": b.sites_name,
        "observable_joints": b.walker["observable_joints"],
        "action_names": [b.act_name[i] for c in ("adhesion", "head", "mouth", "antennae", "wings", "abdomen", "legs")
                         for i in (b.walker["ctrl_indices"][c] or [])],
    }
    with open(os.path.join(OUT, "fly_ball.json"), "w") as f:
        json.dump(meta, f, indent=1)
    print("wrote ball model: nbody", b.nbody, "nq", b.nq, "nv", b.nv, "nu", b.nu, "ngeom", len(b.geom_bodyid))
    # fly_envs.walk_imitation: free-root walking fly on a floor plane.  The reference's walking dataset (absent) names the mocap
    # joints and sites it tracks (trajectory_loaders.py:217-223); restated here as every leg joint and the six claw sites.
    w = build_walk_model()
    write_blob(os.path.join(OUT, "fly_walk.ffmb"), model_tensors(w, with_collision=True))
    leg = [n for n in w.jnt_name if any(t in n for t in ("_T1_", "_T2_", "_T3_"))]
    meta = {
        "body_name": w.body_name, "jnt_name": w.jnt_name, "act_name": w.act_name, "ten_name": w.ten_name,
        "geom_name": w.geom_name, "site_name": w.sites_name,
        "observable_joints": w.walker["observable_joints"],
        "action_names": [w.act_name[i] for c in ("adhesion", "head", "mouth", "antennae", "wings", "abdomen", "legs")
                         for i in (w.walker["ctrl_indices"][c] or [])],
        "mocap_joints": leg,
        "mocap_sites": [n for n in w.sites_name if n.startswith("claw_")],
    }
    with open(os.path.join(OUT, "fly_walk.json"), "w") as f:
        json.dump(meta, f, indent=1)
    print("wrote walk model: nbody", w.nbody, "nq", w.nq, "nv", w.nv, "nu", w.nu, "ngeom", len(w.geom_bodyid), "mocap joints", len(leg))


if __name__ == "__main__":
    main()
