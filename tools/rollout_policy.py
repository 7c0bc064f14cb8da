"""Example: a random-weight MLP policy driving 8192 flight-imitation envs entirely on the GPU."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
from flybody_amd import fly_envs
from flybody_amd.actor_loop import BatchedActorLoop

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
env = fly_envs.flight_imitation(batch_size=B, random_state=0, canonical_actions=True, clip_actions=True)
torch.manual_seed(0)
policy = torch.nn.Sequential(torch.nn.Linear(env.spec.obs_dim, 256), torch.nn.ELU(), torch.nn.Linear(256, 256), torch.nn.ELU(),
                             torch.nn.Linear(256, env.spec.action_dim), torch.nn.Tanh()).to(env.device)
loop = BatchedActorLoop(env, policy)
print(loop.run(300))
