#!/usr/bin/env python3
"""Development tool: time variants of libuamenv_hip.so (built with -DUAM_ABLATE=<mask> or other
flags into csrc/build/variants/) on the roofline configuration, interleaved in one process per variant.
Usage: python tools/ablate.py [variant.so ...]"""
import glob, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import sys, os, json, time
sys.path.insert(0, %r)
import torch, uamenv
B, N = int(os.environ.get("AB_B", 65536)), int(os.environ.get("AB_N", 32))
env = uamenv.BatchedSingleAircraftEnv(B, N, seed=0)
env.reset()
tape = torch.randint(0, 9, (64, B), dtype=torch.int32, device=env.device)
pol = os.environ.get("AB_POLICY", "random")
def run(n):
    for k in range(n):
        a = tape[k %% 64] if pol == "random" else env.policy_actions("goal")
        env.step(a)
run(300); torch.cuda.synchronize()
res = []
for rep in range(5):
    env.timing_begin(); run(500); res.append(env.timing_end() * 1e3)
print(json.dumps({"lib": os.environ.get("UAMENV_LIB", "default"), "us_per_step": sorted(res)[len(res)//2], "all": [round(r, 2) for r in res]}))
''' % ROOT

def main():
    libs = sys.argv[1:] or sorted(glob.glob(os.path.join(ROOT, "gym-guidance-collision-avoidance-single_amd/csrc/build/variants/*.so")))
    for lib in libs:
        env = dict(os.environ, UAMENV_LIB=lib)
        out = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True)
        print(out.stdout.strip() or out.stderr[-2000:])

if __name__ == "__main__":
    main()
