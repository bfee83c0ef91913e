#!/usr/bin/env python3
"""The experiment of Algorithms/MCTS/Agent_RandInt.py (run_experiment, :12-64) for a whole batch of episodes at
once: every env plans with MCTS(root).best_action(no_simulations, search_depth) every 5th step and holds the
action in between; reports what the reference prints — NMAC probability, goal probability, average conflicts per
episode — next to two planner-free baselines on the same seeds.

  python examples/mcts_agent.py --envs 2048 --intruders 20 --steps 1500
"""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import uamenv


def run(policy, B, N, steps, sims, depth, seed, update_frame=5, agent="agent_randint"):
    # agent_randint: Agent_RandInt.py = RandIntruder env + nodes_single_randintru.py (6-wide state);
    # agent: Agent.py = SingleAircraftMCTSEnv + nodes_single.py (4-wide state, upstream quirks included)
    variant, model = ("randintruder", "randintru") if agent == "agent_randint" else ("mcts", "straight4")
    env = uamenv.BatchedSingleAircraftEnv(B, N, variant=variant, seed=seed, monitor=True)
    planner = uamenv.BatchedMCTS(N, seed=seed, simulations=sims, search_depth=depth, model=model) if policy == "mcts" else None
    env.reset()
    counts = torch.zeros(5, dtype=torch.int64, device=env.device)        # info codes of finished episodes
    conflicts = torch.zeros((), dtype=torch.int64, device=env.device)    # env.no_conflict summed over finished episodes
    last_nc = env.no_conflict.clone()
    t0 = time.perf_counter()
    acts = None
    for t in range(steps):
        if policy == "mcts":
            if t % update_frame == 0:                                  # Agent_RandInt.py:33-42
                acts = planner.best_action(planner.roots_from_env(env), root_id0=t * B)
        elif policy == "goal":
            acts = env.policy_actions("goal")
        else:
            acts = env.policy_actions("random", seed=t)
        nc_before = env.no_conflict                                    # counter before the step (auto-reset clears it)
        obs, rew, done, infos = env.step(acts)
        d = done.view(torch.uint8).bool()
        codes = infos.codes[d].long()
        counts += torch.bincount(codes, minlength=5)
        # conflicts of an episode = the counter at its last step: the value before the step plus this step's news is not
        # visible after an auto-reset, so use the pre-step value as a (slightly low) estimate, like env.no_conflict read
        # right before the terminal step
        conflicts += nc_before[d].long().sum()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    episodes = int(counts.sum().item())
    c = counts.cpu().tolist()
    return {"agent": agent, "policy": policy, "envs": B, "intruders": N, "steps": steps, "episodes": episodes,
            "nmac_prob": round(c[3] / max(1, episodes), 4), "goal_prob": round(c[2] / max(1, episodes), 4),
            "avg_conflicts_per_episode": round(float(conflicts.item()) / max(1, episodes), 3),
            "env_steps_per_s": round(B * steps / wall), "wall_s": round(wall, 2)}


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, default=2048)
    ap.add_argument("--intruders", type=int, default=20)       # Algorithms/MCTS/config_single.py: intruder_size = 20
    ap.add_argument("--steps", type=int, default=1500)
    ap.add_argument("--no_simulations", "-s", type=int, default=100)
    ap.add_argument("--search_depth", "-d", type=int, default=3)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--policies", default="mcts,goal,random")
    ap.add_argument("--agent", default="agent_randint", choices=["agent_randint", "agent"])
    a = ap.parse_args()
    for pol in a.policies.split(","):
        print(json.dumps(run(pol, a.envs, a.intruders, a.steps, a.no_simulations, a.search_depth, a.seed, agent=a.agent)))
