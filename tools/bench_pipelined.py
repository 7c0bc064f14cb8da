"""What asynchronous sub-batches buy: the same 8 192 (4 096) envs as G handles of B / G envs, each stepped on its own HIP stream, the
host enqueueing their steps round-robin - the way the reference runs its actors (one process per env, nobody waits for anybody).  A launch
of the whole batch ends with a drain in which most wave slots idle; with several launches in flight one group's
drain overlaps another group's start.  Not bench.py's configuration (one batched env, one launch per step): reported beside it.
    python tools/bench_pipelined.py [flight|ball] [groups ...]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
from flybody_amd import fly_envs

kind = sys.argv[1] if len(sys.argv) > 1 else "flight"
groups = [int(x) for x in sys.argv[2:]] or [1, 2, 4]
Btot = 8192 if kind == "flight" else 4096
for G in groups:
    B = Btot // G
    envs, streams, acts = [], [], []
    for g in range(G):
        if kind == "flight":
            e = fly_envs.flight_imitation(batch_size=B, random_state=0, env_id_base=g * B)
        else:
            e = fly_envs.walk_on_ball(batch_size=B)
        spec = e.action_spec()
        lo, hi = torch.tensor(spec.minimum, device="cuda"), torch.tensor(spec.maximum, device="cuda")
        if kind != "flight":
            lo, hi = torch.full_like(lo, -0.2), torch.full_like(hi, 0.2)
        gen = torch.Generator(device="cuda").manual_seed(1234 + g)
        acts.append([(lo + (hi - lo) * torch.rand(B, spec.shape[0], device="cuda", generator=gen)).contiguous() for _ in range(16)])
        envs.append(e); streams.append(torch.cuda.Stream())
        e.reset()
    torch.cuda.synchronize()
    def run(n):
        for k in range(n):
            for g in range(G):
                with torch.cuda.stream(streams[g]):
                    envs[g].step(acts[g][k % 16])
        torch.cuda.synchronize()
    run(60 if kind == "flight" else 330)
    n = 200 if kind == "flight" else 100
    t0 = time.perf_counter(); run(n); dt = (time.perf_counter() - t0) / n
    print(f"{kind}: {G} group(s) of {B} envs, each on its own stream: {dt * 1e3:.4f} ms per step of all {Btot} envs, {Btot / dt / 1e6:.3f} M env-steps/s", flush=True)
    for e in envs: e.close()
