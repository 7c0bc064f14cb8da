"""The production collective backend on hardware: RCCL (torch.distributed backend "nccl") at world size 1.

One GPU cannot show a scaling curve, but it can run everything of the N > 1 step except the wire: communicator init, the
asynchronous one-collective timestep gather on RCCL's own stream behind the real env's launch stream, the double-buffer protocol
of bench.py's loop, the action scatter in the other direction, and teardown.  Runs in a child process so that the RCCL
communicator is created before any other GPU work of that process and never shares one with the rest of the test session."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

CHILD = r'''
import os, sys
sys.path.insert(0, %(root)r)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29531")
import torch, torch.distributed as dist
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
assert dist.get_backend() == "nccl"
from flybody_amd import fly_envs
from flybody_amd.distributed import ActionScatter, TimestepGather
B = 512
env = fly_envs.flight_imitation(batch_size=B, random_state=0)
spec = env.action_spec()
lo, hi = torch.tensor(spec.minimum, device=dev), torch.tensor(spec.maximum, device=dev)
g = torch.Generator(device=dev).manual_seed(3)
gathers = [TimestepGather(B, env.spec.obs_dim, dev, 1, 0, force_collective=True) for _ in range(2)]
scatters = [ActionScatter(B, spec.shape[0], dev, 1, 0, force_collective=True) for _ in range(2)]
works, kept = [None, None], [None, None]
env.reset()
nok = 0
for k in range(40):
    a_all = (lo + (hi - lo) * torch.rand(B, spec.shape[0], device=dev, generator=g)).contiguous()
    w = scatters[k & 1](a_all, async_op=True)          # rank 0 -> ranks over RCCL
    w.wait()
    a = scatters[k & 1].local
    assert torch.equal(a, a_all)
    ts = env.step(a)
    i = k & 1
    if works[i] is not None:                            # the gather issued two steps ago must have landed intact
        works[i].wait()
        uo, ur, ud, us = TimestepGather.unpack(gathers[i].out[0], env.spec.obs_dim)
        ko, kr, kd, ks = kept[i]
        assert torch.equal(uo, ko) and torch.equal(ur, kr) and torch.equal(ud, kd) and torch.equal(us, ks)
        nok += 1
    kept[i] = (env.flat_observation.clone(), ts.reward.clone(), ts.discount.clone(), ts.step_type.clone())
    works[i] = gathers[i](env.flat_observation, ts.reward, ts.discount, ts.step_type, async_op=True)
for i in range(2):
    works[i].wait()
torch.cuda.synchronize()
dist.barrier()
dist.destroy_process_group()
env.close()
print("RCCL_OK", nok)
'''


@pytest.mark.timeout(600)
def test_rccl_gather_and_scatter_at_world_size_1():
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, "-c", CHILD % {"root": ROOT}], env=env, capture_output=True, text=True, timeout=550)
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-3000:])
    assert "RCCL_OK 38" in out.stdout, out.stdout[-500:]


@pytest.mark.timeout(600)
def test_bench_multi_rank_path_on_one_gpu():
    """bench.py's N > 1 loop - process group on RCCL, double-buffered action scatter and timestep gather as real collectives, barriers,
    max-over-ranks timing - with the real env on a one-rank group (FLYBODY_BENCH_FORCE_MULTI): everything but the wire."""
    import json
    import subprocess
    import sys

    import torch

    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
    env = dict(os.environ, FLYBODY_BENCH_FORCE_MULTI="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29541")
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--no-cpu-baseline", "--steps", "40", "--warmup", "10", "--envs-per-gpu", "2048"],
                         env=env, capture_output=True, text=True, timeout=500)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["config"]["global_batch"] == 2048 and "rehearsal" in d["config"]["parallelism"] and d["value"] > 1e6
