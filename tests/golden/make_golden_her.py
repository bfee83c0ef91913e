#!/usr/bin/env python3
"""Golden fixtures for the k-nearest observation / HER env, recorded from the REFERENCE
(Simulators/SingleAircraftDiscrete9HEREnv.py) in the build container; same rules as make_golden.py.

  g13_her9_*.npz: np.random.seed(seed); env = SingleAircraftDiscrete9HEREnv()  (its __init__ resets once);
                  obs0 = env.reset(); then DummyVecEnv-order stepping.  The dict observation is stored
                  flattened: observation (4 + 5n) | achieved_goal (2) | desired_goal (2).
  g14_her3_*.npz: the same for SingleAircraftDiscrete3HEREnv (3 heading actions, goal drawn 100 px inside the
                  map, reward shaped by the nearest swept intruder).  Its step() returns the nearest
                  distance in place of the info dict (:178), stored as `nearest` (+ `nearest_f64`: the
                  value was a float64, i.e. from a promoted intruder); the result code is captured from
                  the env's own _terminal_reward() return value.
  g15_her_relabel.npz: compute_reward / compute_input_reward of both HER envs on float32-representable rows
                  (trajectory observations with goals swapped, plus rows on both sides of every threshold).
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


def run3(Config, her3, name, N, seed, T, policy, sparse):
    keep = Config.sparse_reward
    Config.sparse_reward = sparse
    Config.intruder_size = N
    try:
        np.random.seed(seed)
        env = her3.SingleAircraftDiscrete3HEREnv()
        last = {}
        inner = env._terminal_reward

        def spy():
            out = inner()
            last["code"] = out[2]
            return out
        env._terminal_reward = spy
        obs0 = flat(env.reset())
        st0 = mg.snapshot(env)
        rec = mg.Recorder(N, 25)
        acts, nearest, nearest_f64 = [], [], []
        for t in range(T):
            if policy == "goal":
                a = mg.goal_seeking_action(env, Config.d_heading)[0]
            else:
                a = 1 if t % 5 else (t // 5) % 3
            acts.append(a)
            ob, r, d, near = env.step(a)
            rec.rows["obs"].append(flat(ob))
            rec.rows["reward"].append(float(r))
            rec.rows["done"].append(bool(d))
            rec.rows["info"].append({"": 0, "c": 1, "g": 2, "n": 3, "w": 4}[last["code"]])
            rec.rows["no_conflict"].append(int(env.no_conflict))
            nearest.append(float(near))
            nearest_f64.append(isinstance(near, np.float64))
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
        data["nearest"] = np.asarray(nearest, dtype=np.float64)
        data["nearest_f64"] = np.asarray(nearest_f64, dtype=np.uint8)
        mg.add_prefixed(data, "st0_", st0)
        mg.save(name, dict(seed=seed, N=N, variant="SingleAircraftDiscrete3HEREnv", obs_mode=3, action_mode=3, dynamics=0,
                           knn=int(Config.n), sparse_reward=int(sparse), shaped_reward=1, goal_margin=100,
                           conflict_coeff=float(Config.conflict_coeff)), data)
    finally:
        Config.sparse_reward = keep


def relabel(Config, her, her3):
    """compute_reward / compute_input_reward on float32-representable rows."""
    rng = np.random.RandomState(77)
    Config.intruder_size = 8
    np.random.seed(5)
    env9 = her.SingleAircraftDiscrete9HEREnv()
    env3 = her3.SingleAircraftDiscrete3HEREnv()
    n = int(Config.n)
    rows = []
    ob = env9.reset()
    for t in range(600):
        ob, r, d, _ = env9.step(int(rng.randint(9)))
        goal = ob["desired_goal"] if t % 3 else ob["achieved_goal"] + rng.uniform(-0.02, 0.02, 2)   # relabelled goals, some hits
        rows.append(np.concatenate([ob["observation"], goal]))
        if d:
            ob = env9.reset()
    # rows on both sides of each threshold: entries idx*4+4, idx*4+5 are what the reference reads (stride 4 on a
    # 5-wide record: it mixes fields; preserved)
    W = float(Config.window_width)
    for thr in (Config.NMAC_dist, Config.minimum_separation, Config.goal_radius):
        for k in range(40):
            row = rng.uniform(0.05, 0.95, 4 + 5 * n + 2)
            d = thr * (1 + rng.uniform(-2e-3, 2e-3))
            ang = rng.uniform(0, 2 * np.pi)
            col = rng.randint(n)
            tx, ty = row[0] + d * np.cos(ang) / W, row[1] + d * np.sin(ang) / W
            if thr == Config.goal_radius:
                row[-2], row[-1] = tx, ty
            else:
                row[col * 4 + 4], row[col * 4 + 5] = tx, ty
            rows.append(row)
    X = np.asarray(rows, dtype=np.float32).astype(np.float64)      # the C-ABI takes float32 rows
    data = {"inputs": X.astype(np.float32)}
    keep = Config.sparse_reward
    for sparse in (False, True):
        Config.sparse_reward = sparse
        tag = "sparse" if sparse else "dense"
        data["input_reward9_" + tag] = np.asarray([env9.compute_input_reward(x.copy()) for x in X], dtype=np.float64)
        data["input_reward3_" + tag] = np.asarray([env3.compute_input_reward(x.copy()) for x in X], dtype=np.float64)
        data["goal_reward9_" + tag] = np.asarray([env9.compute_reward(x[:2].copy(), x[-2:].copy(), None) for x in X], dtype=np.float64)
        data["goal_reward3_" + tag] = np.asarray([env3.compute_reward(x[:2].copy(), x[-2:].copy(), None) for x in X], dtype=np.float64)
    Config.sparse_reward = keep
    assert np.array_equal(data["input_reward9_dense"], data["input_reward3_dense"])
    mg.save("g15_her_relabel", dict(seed=77, N=8, knn=n), data)


def main():
    sys.dont_write_bytecode = True
    Config, m_env, m_env2, m_mcts, m_ri = mg.load_reference("/root/reference")
    import SingleAircraftDiscrete9HEREnv as her
    run(Config, her, "g13_her9_n8_seed41", 8, 41, 1500, "goal", False)
    run(Config, her, "g13_her9_n32_seed42", 32, 42, 800, "goal", False)
    run(Config, her, "g13_her9_sparse_n8_seed43", 8, 43, 1200, "straight", True)
    import SingleAircraftDiscrete3HEREnv as her3
    run3(Config, her3, "g14_her3_n8_seed51", 8, 51, 1500, "goal", False)
    run3(Config, her3, "g14_her3_n32_seed52", 32, 52, 800, "goal", False)
    run3(Config, her3, "g14_her3_sparse_n8_seed53", 8, 53, 1200, "straight", True)
    run3(Config, her3, "g14_her3_straight_n80_seed54", 80, 54, 500, "straight", False)
    relabel(Config, her, her3)


if __name__ == "__main__":
    main()
