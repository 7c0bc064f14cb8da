"""How close do the fly's own collision geoms come to each other under random actions?  

Runs the float64 oracle (test infrastructure; never the product path) with the collision geoms attached and records, at
every control step of full-range random-action rollouts:

  * supported pairs (sphere / capsule, what `ball_step_kernel` collides): number of fly-fly contacts MuJoCo's primitive
    colliders would create and the smallest `dist - margin` seen;
  * unsupported pairs (an ellipsoid or a cylinder on either side: MuJoCo's general convex routine, not restated): a
    rigorous lower bound of their separation (Gilbert's algorithm on the Minkowski difference, `convex_separation` in
    oracle/fly_oracle.c), the smallest over the rollout and the pair that set it.

    python tools/self_collision_stats.py flight --steps 100000 [--procs 8]
    python tools/self_collision_stats.py ball   --steps 20000  [--amp 0.2]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)


def _worker(args):
    kind, idx, steps, amp = args
    import numpy as np

    from oracle import oracle as O

    rng = np.random.RandomState(1000 + idx)
    if kind == "flight":
        from flybody_amd.tasks.synthetic import base_wing_pattern, flight_trajectories
        from flybody_amd.tasks.trajectories import preprocess
        from flybody_amd.tasks.wbpg import build_tables

        blob = os.path.join(ROOT, "flybody_amd", "assets", "fly_flight.ffmb")
        names = json.load(open(blob.replace(".ffmb", ".json")))["geom_name"]
        m = O.OracleModel(blob)
        tables = build_tables(base_wing_pattern())
        rq, rv = preprocess(*flight_trajectories(16, 3006))
        env = O.OracleFlightEnv(m, tables, rq, rv, seed=5, env_id=idx)
        lo = np.array([-0.2, -3, -0.5, -1, -1, -1, -1, -1, -1, -0.7, -1.05, -1.0])
        hi = np.array([0.2, 3, 0.3, 1, 1, 1, 1, 1, 1, 0.7, 0.7, 1.0])
        draw = lambda: lo + (hi - lo) * rng.uniform(0, 1, 12)
    else:
        blob = os.path.join(ROOT, "flybody_amd", "assets", "fly_ball.ffmb")
        names = json.load(open(blob.replace(".ffmb", ".json")))["geom_name"]
        m = O.OracleModel(blob)
        env = O.OracleBallEnv(m)
        draw = lambda: rng.uniform(-amp, amp, 59)
    m.L.fo_set_measure_unsupported(m.ptr, 1)
    env.reset()
    d = env.data
    out = dict(steps=0, selfcon_steps=0, selfcon_max=0, self_min=1e30, self_pair=None, unsup_min=1e30, unsup_pair=None, near_unsup_steps=0,
               episodes=0, unsup_touch_steps=0, unsup_pairs={}, sup_pairs={}, first_obs_touch=0)
    for _ in range(steps):
        st = env.step(draw())[0]
        out["steps"] += 1
        out["episodes"] += int(st == 0)
        # fly-fly contacts among the supported pairs at the last position stage of this control step
        c = d.contacts()
        nself = 0
        for row in c:
            if "ball" not in names[int(row[0])] and "ball" not in names[int(row[1])]:
                nself += 1
                key = names[int(row[0])] + " | " + names[int(row[1])]
                out["sup_pairs"][key] = out["sup_pairs"].get(key, 0) + 1
        out["selfcon_steps"] += int(nself > 0)
        out["selfcon_max"] = max(out["selfcon_max"], nself)
        s, g1, g2 = d.self_min_clear()
        if s < out["self_min"]:
            out["self_min"], out["self_pair"] = s, (names[g1], names[g2])
        u, g1, g2 = d.unsupported_min_sep()
        if u < out["unsup_min"]:
            out["unsup_min"], out["unsup_pair"] = u, (names[g1], names[g2])
        out["near_unsup_steps"] += int(d.near_unsupported > 0)
        nt, pairs = d.unsupported_touching()
        out["unsup_touch_steps"] += int(nt > 0)
        out["first_obs_touch"] += int(nt > 0 and st == 0)
        for g1, g2 in pairs:
            key = names[g1] + " | " + names[g2]
            out["unsup_pairs"][key] = out["unsup_pairs"].get(key, 0) + 1
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("kind", choices=("flight", "ball"))
    ap.add_argument("--steps", type=int, default=100000, help="control steps in total")
    ap.add_argument("--procs", type=int, default=8)
    ap.add_argument("--amp", type=float, default=0.2, help="ball: raw action amplitude")
    args = ap.parse_args()
    import multiprocessing as mp

    from oracle import oracle as O

    O.build()
    per = -(-args.steps // args.procs)
    t0 = time.perf_counter()
    with mp.get_context("fork").Pool(args.procs) as pool:
        res = pool.map(_worker, [(args.kind, i, per, args.amp) for i in range(args.procs)])
    tot = dict(kind=args.kind, control_steps=sum(r["steps"] for r in res), episodes=sum(r["episodes"] for r in res),
               steps_with_supported_fly_fly_contact=sum(r["selfcon_steps"] for r in res), max_fly_fly_contacts=max(r["selfcon_max"] for r in res),
               steps_with_unsupported_pair_in_bounding_range=sum(r["near_unsup_steps"] for r in res))
    b = min(res, key=lambda r: r["self_min"])
    tot["supported_min_dist_minus_margin_cm"], tot["supported_min_pair"] = b["self_min"], b["self_pair"]
    b = min(res, key=lambda r: r["unsup_min"])
    tot["unsupported_min_separation_lower_bound_cm"], tot["unsupported_min_pair"] = b["unsup_min"], b["unsup_pair"]
    tot["steps_with_unsupported_pair_within_margin"] = sum(r["unsup_touch_steps"] for r in res)
    tot["episode_starts_with_unsupported_pair_within_margin"] = sum(r["first_obs_touch"] for r in res)
    for k in ("unsup_pairs", "sup_pairs"):
        agg = {}
        for r in res:
            for kk, v in r[k].items():
                agg[kk] = agg.get(kk, 0) + v
        tot[k + "_steps"] = dict(sorted(agg.items(), key=lambda kv: -kv[1])[:12])
    tot["amp"] = args.amp if args.kind == "ball" else "full action spec"
    tot["wall_s"] = round(time.perf_counter() - t0, 1)
    print(json.dumps(tot))


if __name__ == "__main__":
    main()
