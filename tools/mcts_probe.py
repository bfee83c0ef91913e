import sys, time, json
sys.path.insert(0, "/root/repo")
import torch, uamenv
B, N = 8192, 64
env = uamenv.BatchedSingleAircraftEnv(B, N, variant="randintruder", seed=0); env.reset()
for _ in range(20): env.step(env.policy_actions("goal"))
for name, over in (("default", {}), ("no_headchg", dict(heading_change_prob=0.0)), ("sims50", dict(simulations=50)), ("depth2", dict(search_depth=2)), ("frames5", dict(simulate_frame=5))):
    pl = uamenv.BatchedMCTS(N, seed=0, **over)
    roots = pl.roots_from_env(env)
    pl.best_action(roots); torch.cuda.synchronize()
    ts = []
    for r in range(3):
        t0 = time.perf_counter(); pl.best_action(roots, root_id0=r * B); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    print(name, round(sorted(ts)[1] * 1e3, 2), "ms")
