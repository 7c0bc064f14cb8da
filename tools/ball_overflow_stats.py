"""How often walk_on_ball envs exceed the kernel's contact / constraint-row capacities (16 contacts, 48 rows, 24 columns per block:
the extra ones are dropped and the env is flagged, DESIGN.md known gaps) as a function of the action amplitude.
    python tools/ball_overflow_stats.py [batch] [steps]"""
import os, sys, json
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
from flybody_amd import fly_envs

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 400
for amp in (0.2, 0.5, 1.0):
    env = fly_envs.walk_on_ball(batch_size=B)
    env.reset()
    g = torch.Generator(device="cuda").manual_seed(0)
    by_reason = [0, 0, 0]
    prev_flag = torch.zeros(B, dtype=torch.bool, device="cuda"); new_total = 0; exposure = 0
    flagged_steps = 0; ncon_max = 0; ncon_sum = 0; it_sum = 0; last = 0
    for k in range(steps):
        a = ((torch.rand(B, 59, device="cuda", generator=g) * 2 - 1) * amp).contiguous()
        ts = env.step(a)
        ints, _ = env.get_task_state()
        flagged_steps += int((ints[:, 7] != 0).sum())
        for b in range(3):
            by_reason[b] += int(((ints[:, 7] >> b) & 1).sum())
        fl = ints[:, 7] != 0
        if k >= 50:  # the flag is sticky over an episode: rate of FIRST overflows among the envs not yet flagged
            new_total += int((fl & ~prev_flag).sum()); exposure += int((~prev_flag).sum())
        prev_flag = fl
        ncon_max = max(ncon_max, int(ints[:, 5].max())); ncon_sum += int(ints[:, 5].sum()); it_sum += int(ints[:, 6].sum())
        last += int((ts.step_type == 2).sum())
    print(json.dumps({"action_amplitude": amp, "envs": B, "steps": steps, "env_steps_flagged_overflow": flagged_steps,
                      "flagged_fraction": flagged_steps / (B * steps), "env_steps_in_episodes_that_exceeded [contacts>16, rows>48, block columns>24]": by_reason, "first_overflow_rate_per_env_step": new_total / max(1, exposure), "max_contacts": ncon_max, "mean_contacts": ncon_sum / (B * steps),
                      "mean_newton_iters": it_sum / (B * steps), "episode_ends": last}), flush=True)
    env.close()
