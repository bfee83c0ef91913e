#!/usr/bin/env python3
"""Golden fixtures for the k-nearest observation / HER env, recorded from the REFERENCE
(Simulators/SingleAircraftDiscrete9HEREnv.py) in the build container; same rules as make_golden.py.

  g13_her9_*.npz: np.random.seed(seed); env = SingleAircraftDiscrete9HEREnv()  (its __init__ resets once);
                  obs0 = env.reset(); then DummyVecEnv-order stepping.  The dict observation is stored
                  flattened: observation (4 + 5n) | achieved_goal (2) | desired_goal (2).
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402


def flat(ob):
    return np.concatenate([np.asarray(ob["observation"], dtype=np.float64), np.asarray(ob["achieved_goal"], dtype=np.float64),
                           np.asarray(ob["desired_goal"], dtype=np.float64)])


def run(Config, her, name, N, seed, T, policy, sparse):
    keep = Config.sparse_reward
    Config.sparse_reward = sparse
    Config.intruder_size = N
    try:
        np.random.seed(seed)
        env = her.SingleAircraftDiscrete9HEREnv()
        obs0 = flat(env.reset())
        st0 = mg.snapshot(env)
        rec = mg.Recorder(N, 25)
        acts = []
        for t in range(T):
            if policy == "goal":
                a0, a1 = mg.goal_seeking_action(env, Config.d_heading)
                a = a0 * 3 + a1
            else:
                a = 4 if t % 7 else (t // 7) % 9
            acts.append(a)
            ob, r, d, info = env.step(a)
            rec.rows["obs"].append(flat(ob))
            rec.rows["reward"].append(float(r))
            rec.rows["done"].append(bool(d))
            rec.rows["info"].append({"": 0, "c": 1, "g": 2, "n": 3, "w": 4}[info["result"]])
            rec.rows["no_conflict"].append(int(env.no_conflict))
            if d:
                ro = env.reset()
                rec.reset_steps.append(t)
                rec.reset_obs.append(flat(ro))
            st = mg.snapshot(env)
            rec.rows["conflict"].append(st["int_conflict"])
            rec.rows["promoted"].append(st["int_promoted"])
            if t % 25 == 0 or d or t == T - 1 or t < 8:
                rec.add_state(t, st)
        data = rec.finish()
        data["actions"] = np.asarray(acts, dtype=np.int32)
        data["obs0"] = obs0
        mg.add_prefixed(data, "st0_", st0)
        mg.save(name, dict(seed=seed, N=N, variant="SingleAircraftDiscrete9HEREnv", obs_mode=3, action_mode=0, dynamics=0,
                           knn=int(Config.n), sparse_reward=int(sparse)), data)
    finally:
        Config.sparse_reward = keep


def main():
    sys.dont_write_bytecode = True
    Config, m_env, m_env2, m_mcts, m_ri = mg.load_reference("/root/reference")
    import SingleAircraftDiscrete9HEREnv as her
    run(Config, her, "g13_her9_n8_seed41", 8, 41, 1500, "goal", False)
    run(Config, her, "g13_her9_n32_seed42", 32, 42, 800, "goal", False)
    run(Config, her, "g13_her9_sparse_n8_seed43", 8, 43, 1200, "straight", True)


if __name__ == "__main__":
    main()
