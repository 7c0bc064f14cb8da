#!/usr/bin/env python3
"""bench.py - env-steps/s of the batched flight-imitation environment on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N>1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

A "step" is one control step (4 physics substeps + WBPG + observation/reward/termination/auto-reset) of every
env = one launch of `flight_step_kernel`.  Workload = BASELINE.json configs[3]: flight-imitation, B = 8192 envs
per GPU (config 5 = the same on 8 GPUs), synthetic wing-beat pattern and reference trajectories, canonical
U(-1,1) actions already resident in HBM.  For N>1 each rank owns 8192 envs (weak scaling, no data-path
collective inside the physics) and every step ends with the RCCL gather of (obs, reward, discount, step_type)
to rank 0 that a central learner needs.

One JSON line on rank 0; `roofline` is measured live with HIP events on the launch stream; `cpu_baseline` times
the float64 CPU oracle (oracle/, the checker - never the product) on the host cores, at N=1 only.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ENVS_PER_GPU = 8192
ALGO_BYTES_PER_ENV_STEP = 1408  # reads 624 + writes 784 (fp32 state, minimal I/O)
BALL_ENVS_PER_GPU = 4096        # BASELINE.json configs[2]: walk_on_ball, contacts + solver, batch 4096 on one MI355X
BALL_ALGO_BYTES_PER_ENV_STEP = 4412  # reads 1740 + writes 2672
HBM_PEAK_GBS = 8000.0           # MI355X_MICROARCH.md: 8.0 TB/s spec


def _cpu_worker(args):
    """One oracle env per process, `steps` control steps of the same workload; returns (steps, seconds)."""
    idx, steps = args
    import numpy as np

    from flybody_amd.tasks.synthetic import base_wing_pattern, flight_trajectories
    from flybody_amd.tasks.trajectories import preprocess
    from flybody_amd.tasks.wbpg import build_tables
    from oracle import oracle as O

    tables = build_tables(base_wing_pattern())
    rq, rv = preprocess(*flight_trajectories())  # the GPU leg's own 64-trajectory reference set (fly_envs.flight_imitation's default)
    m = O.OracleModel(os.path.join(ROOT, "flybody_amd", "assets", "fly_flight.ffmb"))
    env = O.OracleFlightEnv(m, tables, rq, rv, seed=0, env_id=idx)
    rng = np.random.RandomState(idx)
    amin = np.array([-0.2, -3, -0.5, -1, -1, -1, -1, -1, -1, -0.7, -1.05, -1.0])
    amax = np.array([0.2, 3, 0.3, 1, 1, 1, 1, 1, 1, 0.7, 0.7, 1.0])
    acts = amin + (amax - amin) * (0.5 + 0.5 * rng.uniform(-1, 1, (steps, 12)))
    for k in range(50):
        env.step(acts[k])
    t0 = time.perf_counter()
    for k in range(steps):
        env.step(acts[k])
    return steps, time.perf_counter() - t0


BALL_AMP = 0.2  # raw action amplitude of the walk_on_ball workload (--action-amplitude; read by the forked CPU workers too)


def _cpu_worker_ball(args):
    """walk_on_ball twin of `_cpu_worker` (raw actions U(-0.2, 0.2), BASELINE configs[0] / [2])."""
    idx, steps = args
    import numpy as np

    from oracle import oracle as O

    m = O.OracleModel(os.path.join(ROOT, "flybody_amd", "assets", "fly_ball.ffmb"))
    env = O.OracleBallEnv(m)
    env.reset()
    rng = np.random.RandomState(idx)
    acts = rng.uniform(-BALL_AMP, BALL_AMP, (steps, 59))
    for k in range(10):
        env.step(acts[k])
    t0 = time.perf_counter()
    for k in range(steps):
        env.step(acts[k])
    return steps, time.perf_counter() - t0


def cpu_baseline(target_seconds=20.0, workload="flight_imitation"):
    """Oracle ("port") on all host cores, one env per process as the reference runs its actors
    (train_dmpo_ray.py:432-452).  Must run before this process touches the GPU (fork)."""
    import multiprocessing as mp

    from oracle import oracle as O

    O.build()
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    # a one-GPU box shares its host: stay within the 16-core share unless told otherwise
    cores = max(1, min(cores, int(os.environ.get("FLYBODY_BENCH_CORES", "16"))))
    worker = _cpu_worker_ball if workload == "walk_on_ball" else _cpu_worker
    steps, dt = worker((0, 100 if workload == "walk_on_ball" else 1500))
    per_core = steps / dt
    n = int(max(50 if workload == "walk_on_ball" else 500, 0.7 * per_core * target_seconds))
    ctx = mp.get_context("fork")
    t0 = time.perf_counter()
    with ctx.Pool(cores) as pool:
        res = pool.map(worker, [(i, n) for i in range(cores)])
    wall = max(r[1] for r in res)
    value = sum(r[0] for r in res) / wall
    return {"value": round(value, 1), "unit": "env-steps/s", "cores": cores, "kind": "port",
            "sample": f"{cores} processes x 1 float64 oracle env x {n} control steps of the same {workload} workload "
                      f"(single-core {per_core:.0f} env-steps/s; pool wall {time.perf_counter() - t0:.1f}s)"}


class _FakeEnv:
    """Stand-in with the BatchedFlyEnv surface on CPU tensors: lets the tests rehearse the multi-process control flow
    (sharding, double-buffered gather, barriers, max-over-ranks timing) with gloo.  Never used for a reported number."""

    def __init__(self, B):
        import types

        import numpy as np
        import torch

        self._t, self.B = torch, B
        self.spec = types.SimpleNamespace(obs_dim=104, nsub=4, action_dim=12)
        self._spec = types.SimpleNamespace(minimum=-np.ones(12, np.float32), maximum=np.ones(12, np.float32), shape=(12,))
        self.flat_observation = torch.zeros(B, 104)

    def action_spec(self):
        return self._spec

    def reset(self):
        pass

    def step(self, a):
        import types

        t = self._t
        self.flat_observation = a.sum(1, keepdim=True).expand(self.B, 104).contiguous()
        return types.SimpleNamespace(reward=t.ones(self.B), discount=t.ones(self.B), step_type=t.ones(self.B, dtype=t.int32))

    def time_steps(self, a, n):
        return 1.0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--envs-per-gpu", type=int, default=None)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-async-groups", action="store_true", help="skip the extra measurement of the same envs as two asynchronous groups")
    ap.add_argument("--action-amplitude", type=float, default=0.2,
                    help="walk_on_ball: raw actions U(-a, a)^59; 0.2 = BASELINE configs[2] (task_utils.py:13-24), 1.0 = every actuator saturating")
    ap.add_argument("--rehearse-gather", action="store_true",
                    help="N=1 only: also time the steps with the packing half of the per-step gather (TimestepGather without the "
                         "collective) to show what the N>1 path adds per step on the env's own GPU")
    ap.add_argument("--workload", choices=("flight_imitation", "walk_on_ball"), default="flight_imitation",
                    help="flight_imitation = the headline metric (BASELINE configs[3]/[4]); walk_on_ball = configs[2]")
    args = ap.parse_args()
    ball = args.workload == "walk_on_ball"
    if args.envs_per_gpu is None:
        args.envs_per_gpu = BALL_ENVS_PER_GPU if ball else ENVS_PER_GPU

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run for --gpus > 1")
        args.gpus = world

    global BALL_AMP
    BALL_AMP = float(args.action_amplitude)
    # FLYBODY_BENCH_FORCE_MULTI=1 (rehearsal on one GPU): take the N > 1 path - process group, action scatter, timestep gather as real RCCL
    # collectives on a one-rank group, double buffering, barriers - with the real env.  Everything but the wire.
    multi = world > 1 or os.environ.get("FLYBODY_BENCH_FORCE_MULTI") == "1"
    forced = multi and world == 1
    cpu = None
    if rank == 0 and not multi and not args.no_cpu_baseline:
        cpu = cpu_baseline(workload=args.workload)  # before any GPU initialisation (uses fork)

    import torch
    import torch.distributed as dist

    fake = os.environ.get("FLYBODY_BENCH_FAKE") == "1"  # CPU rehearsal of the N>1 control flow (tests only)
    if fake:
        dev = torch.device("cpu")
        backend = "gloo"
    else:
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)
        backend = "nccl"
    if multi:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if forced:
            os.environ.setdefault("MASTER_PORT", "29533"); os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group(backend, **({} if fake else {"device_id": dev}))

    from flybody_amd.distributed import ActionScatter, TimestepGather, shard

    B = args.envs_per_gpu
    env_id_base, _ = shard(rank, world, B)
    if fake:
        env = _FakeEnv(B)
    else:
        from flybody_amd import fly_envs

        if ball:
            env = fly_envs.walk_on_ball(batch_size=B, device=local_rank)
        else:
            env = fly_envs.flight_imitation(batch_size=B, device=local_rank, random_state=0, env_id_base=env_id_base)
    spec = env.action_spec()
    lo = torch.tensor(spec.minimum, device=dev)
    hi = torch.tensor(spec.maximum, device=dev)
    if ball:  # BASELINE configs[2]: raw actions U(-0.2, 0.2)^59 (`task_utils.py:13-24`)
        lo, hi = torch.full_like(lo, -BALL_AMP), torch.full_like(hi, BALL_AMP)
    g = torch.Generator(device=dev).manual_seed(1234 + rank)
    npool = 16
    # N = 1: the actions are this rank's own.  N > 1: the policy side sits on rank 0 - rank 0 draws the actions of
    # all N * B envs and every rank receives its block through one scatter per step, issued one step ahead (double-buffered) so that
    # it travels while the current step is simulated
    nact = spec.shape[0]
    rows = B * world if (multi and rank == 0) else B
    acts = [(lo + (hi - lo) * torch.rand(rows, nact, device=dev, generator=g)).contiguous() for _ in range(npool)] if (not multi or rank == 0) else None
    scatters = [ActionScatter(B, nact, dev, world, rank, force_collective=forced) for _ in range(2)] if multi else None
    swork = [None, None]

    def issue_scatter(k):
        swork[k & 1] = scatters[k & 1](acts[k % npool] if rank == 0 else None, async_op=True)

    # The per-step gather of everything a central learner consumes: one RCCL call per step,
    # double-buffered so that the gather of step k travels over xGMI while step k+1 is being simulated.
    gathers = [TimestepGather(B, env.spec.obs_dim, dev, world, rank, force_collective=forced) for _ in range(2)]
    works = [None, None]

    def one_step(k):
        if multi:
            if swork[k & 1] is None:
                issue_scatter(k)
            swork[k & 1].wait()
            swork[k & 1] = None
            a = scatters[k & 1].local
            issue_scatter(k + 1)
        else:
            a = acts[k % npool]
        ts = env.step(a)
        if multi:
            i = k & 1
            if works[i] is not None:
                works[i].wait()
            works[i] = gathers[i](env.flat_observation, ts.reward, ts.discount, ts.step_type, async_op=True)
        return ts

    def drain():
        for i in range(2):
            if works[i] is not None:
                works[i].wait()
                works[i] = None
            if swork[i] is not None:
                swork[i].wait()
                swork[i] = None

    def sync():
        if not fake:
            torch.cuda.synchronize(dev)

    env.reset()
    for k in range(args.warmup):
        one_step(k)

    def fence():
        drain()
        sync()
        if multi:
            dist.barrier()
        sync()

    fence()
    # HIP events on the launch stream (the env launches on torch's current stream) bracket the same K launches the wall
    # clock times: the step kernel is the only thing queued on that stream, back to back, so elapsed / K is its mean duration
    ev0 = ev1 = None
    if not fake:
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    if ev0 is not None:
        ev0.record()
    for k in range(args.steps):
        one_step(k)
    if ev1 is not None:
        ev1.record()
    fence()
    elapsed = time.perf_counter() - t0
    if multi:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    rehearsal = None
    if args.rehearse_gather and not multi and not fake:
        # the N>1 step = env.step + pack into one [B, O + 3] buffer + one RCCL gather; at N=1 only the collective is missing
        fence()
        t1 = time.perf_counter()
        for k in range(args.steps):
            ts = env.step(acts[k % npool])
            gathers[k & 1](env.flat_observation, ts.reward, ts.discount, ts.step_type)
        fence()
        with_pack = time.perf_counter() - t1
        nbytes = B * (env.spec.obs_dim + 3) * 4
        rehearsal = {"ms_per_step_with_pack": round(1e3 * with_pack / args.steps, 4), "pack_overhead_ms": round(1e3 * (with_pack - elapsed) / args.steps, 4),
                     "bytes_per_rank_per_step": nbytes,
                     "xgmi_wire_us_per_rank_at_153GBps": round(nbytes / 153e9 * 1e6, 1),
                     "note": "the gather itself is asynchronous on RCCL's stream and double-buffered (bench.py one_step): it overlaps the next "
                             "step's kernel; 7 ranks send to rank 0 over 7 distinct xGMI links"}

    # dominant kernel: mean launch duration by HIP events on the launch stream, same workload
    # kernel_ms = HIP events immediately around each launch of the step kernel alone (ffe_time_kernel: the duration rocprofv3's
    # kernel trace reports for it); the timed region's own events also span the 6-12 us launch-order kernel that follows every step
    # (one launch per call, cycling through the same action pool as the timed region: same workload)
    nk = min(args.steps, 100)
    kacts = [acts[k % npool] for k in range(nk)] if not multi else [scatters[0].local] * nk  # (N > 1: the block last received)
    k_ms = (sum(env.time_kernel(kacts[k], 1) for k in range(nk)) / nk) if hasattr(env, "time_kernel") else env.time_steps(kacts[0], nk)
    sync()
    k_ms_region = ev0.elapsed_time(ev1) / args.steps if ev0 is not None else k_ms

    # What an asynchronous consumer gets from the same envs on this GPU (reported beside the headline, never as `value`): the batch as two
    # groups of B / 2 envs, each stepped on its own HIP stream, ordered only against its own previous step - the reference's actors are
    # separate processes that never wait for each other.  One group's drain overlaps the other's start.
    async_groups = None
    if not multi and not fake and not args.no_async_groups and B % 2 == 0:
        from flybody_amd.groups import EnvGroups

        fence()
        G = 2
        if ball:
            grp = EnvGroups(fly_envs.walk_on_ball, B, groups=G, device=local_rank)
        else:
            grp = EnvGroups(fly_envs.flight_imitation, B, groups=G, device=local_rank, random_state=0, env_id_base=env_id_base)
        gacts = [[acts[k][grp.rows(gi)].contiguous() for k in range(npool)] for gi in range(G)]
        grp.reset()
        sync()
        for k in range(args.warmup + args.steps):  # (as deep into the episodes as the headline region started its last step)
            grp.step([gacts[gi][k % npool] for gi in range(G)])
        grp.synchronize(); sync()
        t1 = time.perf_counter()
        for k in range(args.steps):
            grp.step([gacts[gi][k % npool] for gi in range(G)])
        grp.synchronize(); sync()
        el = time.perf_counter() - t1
        async_groups = {"groups": G, "envs_per_group": B // G, "value": round(B * args.steps / el, 1), "unit": "env-steps/s", "ms_per_step_of_all_envs": round(1e3 * el / args.steps, 4),
                        "note": "same envs, same actions, same results (tests/test_gpu_groups.py); needs a consumer that works per group on the group's stream (flybody_amd/groups.py)"}
        grp.close()


    if rank == 0:
        # HBM traffic of the dominant kernel from the committed rocprofv3 PMC passes of this same command (separate
        # FETCH_SIZE / WRITE_SIZE runs; KiB units; read side doubled per the gfx950 note in MI355X_MICROARCH.md - an
        # upper bound here, since these are 4-byte-per-lane reads, not the wide streams the x2 was calibrated on)
        traffic, traffic_src = None, None
        pj = os.path.join(ROOT, "profiles", "r03_pmc_ball_kernel.json" if ball else "r03_pmc_flight_kernel.json")
        # (counters of an earlier round's kernel are not this kernel's: without the file, traffic is null)
        valu_frac = None
        flop = None
        if os.path.exists(pj) and B == (BALL_ENVS_PER_GPU if ball else ENVS_PER_GPU) and not fake and (not ball or BALL_AMP == 0.2):
            with open(pj) as f:
                pm = json.load(f)
            if "FETCH_SIZE" in pm and "WRITE_SIZE" in pm:
                traffic = int((2 * pm["FETCH_SIZE"]["median"] + pm["WRITE_SIZE"]["median"]) * 1024)
                traffic_src = "profiles/" + os.path.basename(pj) + " (2*FETCH_SIZE + WRITE_SIZE, per launch)"
            if "SQ_INSTS_VALU" in pm and "GRBM_GUI_ACTIVE" in pm:
                # secondary roofline: VALU issue.  A wave64 VALU instruction occupies its SIMD-32 for 2 cycles (MI355X_MICROARCH.md);
                # kernel cycles = GRBM_GUI_ACTIVE / 8 XCDs; 1024 SIMDs.  From the committed PMC passes of this same command.
                valu_frac = round(pm["SQ_INSTS_VALU"]["median"] * 2.0 / (pm["GRBM_GUI_ACTIVE"]["median"] / 8.0 * 1024.0), 4)
        fj = os.path.join(ROOT, "profiles", "r02_oracle_flop_count.json")
        if os.path.exists(fj) and not fake:
            with open(fj) as f:
                fc = json.load(f)["walk_on_ball" if ball else "flight_imitation"]
            per = fc["total_without_constraint_solve"] if ball else fc["total"]
        total_env_steps = world * B * args.steps
        value = total_env_steps / elapsed
        if os.path.exists(fj) and not fake:
            flop = {"flop_per_env_step": round(per), "achieved_tflops": round(value / world * per / 1e12, 3), "peak_fp32_tflops": 157.3,
                    "frac": round(value / world * per / 157.3e12, 5),
                    "source": "profiles/r02_oracle_flop_count.json (instrumented float64 oracle, tools/count_flops.py" +
                              ("; smooth dynamics only - the oracle's dense constraint solver is not the kernel's algorithm)" if ball else "; counted before the fly's own contacts existed: collision and contact rows excluded)")}
        algo = BALL_ALGO_BYTES_PER_ENV_STEP if ball else ALGO_BYTES_PER_ENV_STEP
        achieved = algo * B / (k_ms * 1e-3) / 1e9
        out = {
            "metric": "env-steps/sec (whole node), fruitfly " + ("walk_on_ball task" if ball else "flight-imitation task"),
            "value": round(value, 1), "unit": "env-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": ("walk_on_ball (BASELINE configs[2]): 10 substeps @2e-4 s, ball contacts (elliptic cones, Newton + noslip), the fly's own contacts (every geom pair MuJoCo collides: capsules, ellipsoids, cylinders), adhesion, "
                                    "filtered actuators, touch/force sensors, obs/reward/termination/auto-reset") if ball else
                                   "flight_imitation (BASELINE configs[3]): 4 substeps @5e-5 s (joint limits, fluid forces, the fly's own contacts) + WBPG + obs/reward/termination/auto-reset",
                       "envs_per_gpu": B, "global_batch": world * B, "parallelism": f"env-sharded x{world}" + (" + RCCL action scatter from / timestep gather to rank 0" if multi else "") + (" (one-rank rehearsal: FLYBODY_BENCH_FORCE_MULTI)" if forced else ""),
                       "actions": f"raw U(-{BALL_AMP:g}, {BALL_AMP:g})^59, resident in HBM" if ball else "uniform over the raw action spec (canonical U(-1,1)), resident in HBM"},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 4), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 8), "traffic": traffic, "traffic_source": traffic_src,
                         "kernel": "ball_step_kernel" if ball else "flight_step_kernel", "kernel_ms": round(k_ms, 4), "timed_region_ms_per_launch_incl_order_kernel": round(k_ms_region, 4),
                         "algorithmic_bytes_per_launch": algo * B, "valu_frac": valu_frac, "flop": flop,
                         "note": "fused wave-per-env step keeps state on chip; VALU/LDS-latency bound, not HBM bound (DESIGN.md)"},
            "physics_substeps_per_s": round(value * env.spec.nsub, 1),
        }
        if hasattr(env, "get_task_state") and not fake:
            # envs that met more simultaneous contacts / constraint rows than the kernel carries (the deepest are kept):
            # walk_on_ball: the flag is sticky over the episode; flight: a position stage of the last control step
            ints = env.get_task_state()[0]
            flagged = int((ints[:, 7] != 0).sum()) if ball else int((((ints[:, 7] >> 8) & 255) != 0).sum())
            out["capacity"] = {"envs_flagged_at_end": flagged, "of": B,
                               "limits": "16 contacts / 48 constraint rows / 24 columns per block of M" if ball else "6 simultaneous contacts of the fly with itself"}
        if async_groups is not None:
            out["async_groups"] = async_groups
        if cpu is not None:
            out["cpu_baseline"] = cpu
        if rehearsal is not None:
            out["gather_rehearsal"] = rehearsal
        print(json.dumps(out), flush=True)
    if multi:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
