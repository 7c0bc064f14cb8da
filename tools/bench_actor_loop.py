"""End-to-end actor throughput on one GPU: env.step + policy network + n-step transition writer, all device-resident
(`flybody_amd/actor_loop.py`; what the reference does with one OS process per env: agents/ray_distributed_dmpo.py:401-440,514-521).
The policy is the reference's DMPO policy shape (LayerNormMLP 512-512-256 + a diagonal Gaussian head, train_dmpo_ray.py),
random weights, sampled actions in the canonical [-1, 1] spec (clipped).    python tools/bench_actor_loop.py [steps]"""
import json, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
from flybody_amd.actor_loop import BatchedActorLoop, NStepTransitionWriter
from flybody_amd.batched_env import BatchedFlyEnv
from flybody_amd.tasks.synthetic import base_wing_pattern, flight_trajectories
from flybody_amd.tasks.trajectories import preprocess
from flybody_amd.tasks.wbpg import build_tables

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
B = 8192
tables = build_tables(base_wing_pattern()); rq, rv = preprocess(*flight_trajectories(64, 3006))


class Policy(torch.nn.Module):
    def __init__(self, obs, act):
        super().__init__()
        self.l0 = torch.nn.Linear(obs, 512); self.ln = torch.nn.LayerNorm(512)
        self.l1 = torch.nn.Linear(512, 512); self.l2 = torch.nn.Linear(512, 256)
        self.mean = torch.nn.Linear(256, act); self.std = torch.nn.Linear(256, act)

    def forward(self, o):
        h = torch.tanh(self.ln(self.l0(o)))
        h = torch.nn.functional.elu(self.l1(h)); h = torch.nn.functional.elu(self.l2(h))
        mu, sd = self.mean(h), torch.nn.functional.softplus(self.std(h)) + 1e-4
        return torch.clamp(mu + sd * torch.randn_like(mu), -1.0, 1.0)


torch.manual_seed(0)
out = {}
for name in ("env_only_fixed_action", "env_plus_policy", "env_plus_policy_plus_nstep_writer", "env_plus_bf16_policy_plus_nstep_writer"):
    env = BatchedFlyEnv(tables, rq, rv, batch_size=B, seed=0, canonical_actions=True, clip_actions=True)
    pol = Policy(env.spec.obs_dim, env.spec.action_dim).cuda()
    fixed = (torch.rand(B, env.spec.action_dim, device="cuda") * 2 - 1).contiguous()
    if name == "env_only_fixed_action":
        policy = lambda o: fixed
    elif "bf16" in name:  # the same network under bf16 autocast (MFMA GEMMs); actions come back as float32
        def policy(o, pol=pol):
            with torch.autocast(device_type="cuda", dtype=torch.bfloat16):
                return pol(o).float()
    else:
        policy = pol
    adder = NStepTransitionWriter(B, env.spec.obs_dim, env.spec.action_dim, n_step=50, discount=0.99, capacity=1 << 20) if name.endswith("writer") else None
    loop = BatchedActorLoop(env, policy, adder)
    loop.run(30)
    r = loop.run(steps)
    out[name] = {"env_steps_per_s": round(r["steps_per_second"], 1), "ms_per_step": round(1e3 * B / r["steps_per_second"], 4), "episodes": r["episodes"],
                 "mean_episode_length": round(r["episode_length"], 1)}
    rg = loop.run(steps, graph=True)  # the same iteration captured once into a HIP graph and replayed
    out[name]["hip_graph"] = {"env_steps_per_s": round(rg["steps_per_second"], 1), "ms_per_step": round(1e3 * B / rg["steps_per_second"], 4),
                              "episodes": rg["episodes"]}
    if adder is not None:
        out[name]["transitions_written"] = adder.num_written(); adder.close()
    env.close()
# the same actor as two asynchronous groups of B / 2 envs (flybody_amd/groups.py, GroupedActorLoop): policy, env step and writer of a group
# on that group's stream, nobody waits for the other group
from flybody_amd import fly_envs
from flybody_amd.actor_loop import GroupedActorLoop
from flybody_amd.groups import EnvGroups

for G in (2,):
    grp = EnvGroups(fly_envs.flight_imitation, B, groups=G, random_state=0, canonical_actions=True, clip_actions=True)
    pol = Policy(grp.envs[0].spec.obs_dim, grp.envs[0].spec.action_dim).cuda()
    adders = [NStepTransitionWriter(B // G, e.spec.obs_dim, e.spec.action_dim, n_step=50, discount=0.99, capacity=1 << 19) for e in grp.envs]
    loop = GroupedActorLoop(grp, pol, adders)
    loop.run(30)
    r = loop.run(steps)
    out[f"env_plus_policy_plus_nstep_writer_{G}_async_groups"] = {"env_steps_per_s": round(r["steps_per_second"], 1), "ms_per_step": round(1e3 * B / r["steps_per_second"], 4),
                                                                "episodes": r["episodes"], "mean_episode_length": round(r["episode_length"], 1),
                                                                "transitions_written": sum(a.num_written() for a in adders)}
    rg = loop.run(steps, graph=True)
    out[f"env_plus_policy_plus_nstep_writer_{G}_async_groups"]["hip_graph"] = {"env_steps_per_s": round(rg["steps_per_second"], 1), "ms_per_step": round(1e3 * B / rg["steps_per_second"], 4), "episodes": rg["episodes"]}
    for a in adders: a.close()
    grp.close()
print(json.dumps(out))
