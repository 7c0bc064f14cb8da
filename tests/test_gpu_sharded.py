"""The env-sharded N > 1 path with REAL envs, rehearsed on one GPU: two OS processes (ranks) each own half of the global envs
(`env_id_base` = rank x B) on the same card, step them with the same per-env actions a single full-size handle gets, pack every
timestep with the fused device kernel and gather it to rank 0 in one collective per step - over gloo here, because two RCCL ranks
cannot share a device; everything before the collective (sharding, episode draws keyed by global env id, packing, buffer layout) is
the production path of bench.py's N > 1 mode.  Rank 0 compares every gathered row, bit for bit, with the single handle."""
import os
import socket

import pytest
import torch

pytestmark = pytest.mark.gpu

STEPS, B_RANK, WORLD = 200, 96, 2


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _make_env(batch, base):
    from flybody_amd.batched_env import BatchedFlyEnv
    from flybody_amd.tasks.synthetic import base_wing_pattern, flight_trajectories
    from flybody_amd.tasks.trajectories import preprocess
    from flybody_amd.tasks.wbpg import build_tables

    tables = build_tables(base_wing_pattern())
    rq, rv = preprocess(*flight_trajectories(8, 140))  # short trajectories: several episode ends and resets inside the run
    return BatchedFlyEnv(tables, rq, rv, batch_size=batch, seed=7, env_id_base=base)


def _actions(env, total):
    spec = env.action_spec()
    lo, hi = torch.tensor(spec.minimum), torch.tensor(spec.maximum)
    g = torch.Generator().manual_seed(3)
    return (lo + (hi - lo) * torch.rand(STEPS, total, len(lo), generator=g)).float()  # [step][global env][action]


def _worker(rank, world, port, q):
    import torch.distributed as dist

    from flybody_amd.distributed import TimestepGather, shard

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    base, n = shard(rank, world, B_RANK)
    env = _make_env(n, base)
    acts = _actions(env, world * B_RANK)[:, base : base + n].cuda().contiguous()
    O = env.spec.obs_dim
    gather = TimestepGather(n, O, "cuda", world, rank)
    rows = []
    ts = env.reset()
    for k in range(STEPS + 1):
        out = gather(env.flat_observation, ts.reward, ts.discount, ts.step_type)
        if rank == 0:
            rows.append(torch.cat([o.clone() for o in out], 0))  # rank-major = global env order
        if k < STEPS:
            ts = env.step(acts[k])
    ok, detail = True, ""
    if rank == 0:
        full = _make_env(world * B_RANK, 0)
        a_full = _actions(full, world * B_RANK).cuda().contiguous()
        pk = TimestepGather(world * B_RANK, O, "cuda", 1, 0)
        ts = full.reset()
        nlast = 0
        for k in range(STEPS + 1):
            pk(full.flat_observation, ts.reward, ts.discount, ts.step_type)
            want = pk.pack.cpu()
            if not torch.equal(want, rows[k]):
                ok, detail = False, f"step {k}: {(want != rows[k]).sum().item()} differing values"
                break
            nlast += int((want[:, -1] == 2).sum())
            if k < STEPS:
                ts = full.step(a_full[k])
        if ok and nlast < 10:
            ok, detail = False, f"only {nlast} episode ends inside the run"
    q.put((rank, ok, detail))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_ranks_on_one_gpu_reproduce_the_single_handle_through_the_gather():
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, WORLD, port, q)) for r in range(WORLD)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=280) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    assert all(ok for _, ok, _ in res), res
