"""Throughput of the walk_on_ball kernel (BASELINE.json configs[2]: contacts + solver, batch 4096 on one MI355X).

    python tools/bench_ball.py [--batch 4096] [--steps 50] [--amp 0.2]
"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))


def main():
    import torch

    from flybody_amd import fly_envs

    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=4096)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--amp", type=float, default=0.2)
    ap.add_argument("--flags", type=int, default=0)
    args = ap.parse_args()
    env = fly_envs.walk_on_ball(batch_size=args.batch, physics_flags=args.flags)
    env.reset()
    g = torch.Generator(device="cuda").manual_seed(0)
    acts = [(torch.rand(args.batch, 59, device="cuda", generator=g) * 2 * args.amp - args.amp).contiguous() for _ in range(8)]
    for k in range(args.warmup):
        env.step(acts[k % 8])
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    for k in range(args.steps):
        env.step(acts[k % 8])
    ev1.record()
    torch.cuda.synchronize()
    ms = ev0.elapsed_time(ev1) / args.steps
    ints, _ = env.get_task_state()
    print(json.dumps({"workload": "walk_on_ball", "batch": args.batch, "ms_per_step": ms, "env_steps_per_s": args.batch / ms * 1e3,
                      "mean_contacts": float(ints[:, 5].float().mean()), "mean_newton_iters_per_step": float(ints[:, 6].float().mean()),
                      "flags": args.flags}))


if __name__ == "__main__":
    main()
