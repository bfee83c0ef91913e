#!/usr/bin/env python3
"""Development probe (with a -DUAM_STAMP=1 build, UAMENV_LIB=build_ab/stamp.so): MT19937 single-step launches with a
goal radius that ends EVERY episode at EVERY step (argument 5000) against the normal 20 — the in-step reset path when
every wave runs it (hot in the instruction cache) and when one wave in a few hundred does (cold).
usage: hot_reset_probe.py [goal_radius]"""
import os, sys
sys.path.insert(0, os.getcwd())
import torch, uamenv
class Cfg(uamenv.Config):
    goal_radius = float(sys.argv[1]) if len(sys.argv) > 1 else 20.0
env = uamenv.BatchedSingleAircraftEnv(65536, 32, rng="mt19937", precision="exact", seed=0, config=Cfg)
env.reset()
a = env.policy_actions("random")
nd = 0
for i in range(40):
    _, _, d, _ = env.step(a)
torch.cuda.synchronize()
print("goal_radius", Cfg.goal_radius, "dones in last step", int(d.sum()))
