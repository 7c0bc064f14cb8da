"""World-size-2 gloo rehearsal of the N>1 path: env-id sharding and the one-collective timestep gather."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    from flybody_amd.distributed import ActionScatter, TimestepGather, shard

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    B, O = 32, 104
    base, n = shard(rank, world, B)
    assert (base, n) == (rank * B, B)
    g = torch.Generator().manual_seed(100 + rank)
    obs = torch.rand(B, O, generator=g)
    rew, disc = torch.rand(B, generator=g), (torch.rand(B, generator=g) > 0.1).float()
    st = torch.randint(0, 3, (B,), generator=g, dtype=torch.int32)
    gather = TimestepGather(B, O, "cpu", world, rank)
    out = gather(obs, rew, disc, st)
    ok = True
    if rank == 0:
        for r in range(world):
            gr = torch.Generator().manual_seed(100 + r)
            o = torch.rand(B, O, generator=gr)
            rw, dc = torch.rand(B, generator=gr), (torch.rand(B, generator=gr) > 0.1).float()
            s = torch.randint(0, 3, (B,), generator=gr, dtype=torch.int32)
            uo, ur, ud, us = TimestepGather.unpack(out[r], O)
            ok &= torch.equal(uo, o) and torch.equal(ur, rw) and torch.equal(ud, dc) and torch.equal(us, s)
    else:
        ok = out is None
    # the action path: rank 0 holds every env's action, each rank receives its contiguous block (blocking and asynchronous)
    A = 12
    ga = torch.Generator().manual_seed(7)
    all_actions = torch.rand(world * B, A, generator=ga)
    sc = ActionScatter(B, A, "cpu", world, rank)
    mine = sc(all_actions if rank == 0 else None)
    ok &= torch.equal(mine, all_actions[rank * B:(rank + 1) * B])
    w = sc((2 * all_actions) if rank == 0 else None, async_op=True)
    w.wait()
    ok &= torch.equal(sc.local, 2 * all_actions[rank * B:(rank + 1) * B])
    q.put((rank, bool(ok)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_gather_world_size_2():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=100) for _ in range(world))
    for p in procs:
        p.join(30)
    assert res == {0: True, 1: True}


def test_episode_draws_do_not_depend_on_the_sharding():
    """The counter-based generator is keyed by the *global* env id, so env 40 draws the same trajectory whether it is
    env 40 of one handle or env 8 of the rank whose env_id_base is 32."""
    from oracle import oracle as O

    a = [O.rng_u64(7, 40, ep, 0) for ep in range(4)]
    b = [O.rng_u64(7, 32 + 8, ep, 0) for ep in range(4)]
    assert a == b and len(set(a)) == 4


@pytest.mark.timeout(300)
def test_bench_multiprocess_control_flow_on_gloo():
    """bench.py's N>1 path (env-id sharding, double-buffered async gather, barriers, max-over-ranks timing, one JSON
    line from rank 0) rehearsed with 2 gloo ranks and a stand-in env on CPU tensors."""
    import json
    import subprocess
    import sys

    from conftest import ROOT

    env = dict(os.environ, FLYBODY_BENCH_FAKE="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "2",
           "--envs-per-gpu", "64"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=280)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 128 and d["scaling"] == "weak" and d["steps"] == 6
    assert "roofline" in d and "cpu_baseline" not in d
