#!/usr/bin/env python3
"""Development tool: randomized uam_step_many (single-launch loop or queued launches) vs K separate steps, plus a
get_state/set_state round trip into a fresh handle in the middle."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import uamenv
import parity_util as pu

rs = np.random.RandomState(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
budget = float(sys.argv[2]) if len(sys.argv) > 2 else 120.0
t0, n, bad = time.time(), 0, 0
while time.time() - t0 < budget:
    variant = rs.choice(["discrete9", "continuous", "continuous2", "mcts", "randintruder", "random_start", "her9", "her3"])
    rng = os.environ.get("FUZZ_RNG") or rs.choice(["philox", "philox", "philox", "mt19937"])
    prec = rs.choice(["f32", "f32", "exact"])
    N = int(rs.choice([1, 2, 4, 5, 8, 9, 16, 31, 32, 33, 64, 100, 256]))
    if variant in ("her9", "her3") and N < 6:
        continue
    B = int(rs.choice([1, 3, 32, 33, 257, 1000, 5000, 30000]))
    K = int(rs.choice([1, 2, 5, 16, 17, 40]))
    kw = dict(variant=variant, rng=rng, precision=prec, seed=int(rs.randint(1 << 30)), monitor=bool(rs.randint(2)))
    fuse = str(rs.choice(["auto", "auto", "always", "never"]))
    a = uamenv.BatchedSingleAircraftEnv(B, N, fuse_rollout=fuse, **kw)
    b = uamenv.BatchedSingleAircraftEnv(B, N, graph_safe=int(rng == "philox" and rs.randint(3) == 0), **kw)
    a.reset(); b.reset()
    ok = True
    for rep in range(2):
        mode = int(rs.randint(3))       # 0 step_many, 1 rollout with R slots, 2 rollout_policy (closed loop in the launch)
        if mode == 2:
            kind = "goal" if rs.randint(2) else "random"
            pseed = int(rs.randint(1 << 20))
            R = int(rs.choice([1, 2, K]))
            obs = torch.full((R, B, a.obs_dim), float("nan"), device=a.device)
            _, rew_m, done_m, info_m, acts = a.rollout_policy(K, kind=kind, seed=pseed, obs_out=obs)
            ab = None
            for k in range(K):
                # b's policy call numbers must follow a's: one call per step on both
                ab = b.policy_actions(kind, seed=pseed, out=ab)
                ok &= torch.equal(acts[k], ab)
                o, r, d, i = b.step(ab)
                ok &= torch.equal(rew_m[k], r) and torch.equal(done_m[k], d.view(torch.uint8)) and torch.equal(info_m[k], i.codes)
                if k >= K - R:
                    ok &= torch.equal(obs[k % R], o)
        else:
            tape = torch.stack([a.policy_actions("goal" if rs.randint(2) else "random", seed=int(rs.randint(1 << 20))) for _ in range(K)]).contiguous()
            for _ in range(K):
                b.policy_actions("random")      # keep the two handles' policy call counters level
            if mode == 1:
                R = int(rs.choice([1, 2, 3, K]))
                if (B * a.obs_dim * 4) % 16 and a.obs_dim % 4 == 0:
                    R = 1
                obs = torch.full((R, B, a.obs_dim), float("nan"), device=a.device)
                rew_m = torch.empty((K, B), device=a.device); done_m = torch.empty((K, B), dtype=torch.uint8, device=a.device)
                info_m = torch.empty((K, B), dtype=torch.uint8, device=a.device)
                a.rollout_into(tape, rew_m, done_m, info_m, obs)
            else:
                obs_m, rew_m, done_m, info_m = a.step_many(tape)
            for k in range(K):
                o, r, d, i = b.step(tape[k])
                ok &= torch.equal(rew_m[k], r) and torch.equal(done_m[k], d.view(torch.uint8)) and torch.equal(info_m[k], i.codes)
                if mode == 1 and k >= K - R:
                    ok &= torch.equal(obs[k % R], o)
            if mode == 0:
                ok &= torch.equal(obs_m, o)
        if rep == 0:      # checkpoint a, restore into a fresh handle, continue with that one
            st = a.get_state()
            c = uamenv.BatchedSingleAircraftEnv(B, N, fuse_rollout=fuse, **kw)
            c.reset()
            c.set_state(st)
            for _ in range(K):                  # the policy call counter is the handle's, not part of the state
                c.policy_actions("random")
            a.close()
            a = c
    try:
        pu.compare_states(a.get_state(), b.get_state(), "end", hold_sh=variant == "randintruder")
    except AssertionError as ex:
        ok = False
        print(ex)
    if not ok:
        bad += 1
        print("FAIL B=%d N=%d K=%d fuse=%s %s" % (B, N, K, fuse, kw))
    a.close(); b.close()
    n += 1
print("fuzz step_many: %d configurations, %d failures, %.0f s" % (n, bad, time.time() - t0))
sys.exit(1 if bad else 0)
