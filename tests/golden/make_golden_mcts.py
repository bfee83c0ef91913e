#!/usr/bin/env python3
"""Golden fixtures for the MCTS forward model / search, recorded from the REFERENCE
(Algorithms/MCTS/nodes_single_randintru.py, search_single.py, common.py) in the build container.
Same rules as make_golden.py: data only, reference imported unmodified under the gym stub.

  g11_mcts_move_n<N>.npz        SingleAircraftState.move(): for each case np.random.seed(seed) then
                                SingleAircraftState(state).move(action)
      states [C, 6N+8], actions [C,2], seeds [C]
      next_states [C, 6N+8], flags [C,3] (hit_wall, conflict, reach_goal), rewards [C], rng_pos [C]
  g12_mcts_search_n<N>.npz      MCTS(root).best_action(sims, depth) after np.random.seed(seed)
      states [C, 6N+8], seeds [C], sims, depth
      best_action [C,2], child_action [C,9,2], child_q [C,9], child_n [C,9], root_n [C], rng_pos [C],
      n_nodes [C]
  g16_mcts4_move_*, g17_mcts4_search_*: the same for the 4-wide model Algorithms/MCTS/nodes_single.py (what
      Agent.py plans with on SingleAircraftMCTSEnv observations): states [C, 4N+8].  Upstream quirks recorded
      as they are: the last intruder is ignored ((len - 9) // 4) and the speed is re-derived from the ownship's
      previous vy (state[-5]) every frame.
  g18_mcts4_move_zero_*: the 4-wide model with EVERY sigma zero (heading_sigma too): move() is then a deterministic
      function of (state, action) — whatever the random stream — so the GPU's uam_mcts_move (Philox) can be compared
      with the reference's next state and flags DIRECTLY, without the oracle in between.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402  (gym stub + helpers)


def load_mcts(ref):
    Config, m_env, m_env2, m_mcts, m_ri = mg.load_reference(ref)
    sys.path.insert(0, os.path.join(ref, "Algorithms", "MCTS"))
    import nodes_single_randintru as nodes
    import search_single as search
    import config_single
    return Config, m_ri, nodes, search, config_single.Config


def collect_roots(Config, m_ri, N, seed, count, stride, pair_action=False):
    """Root observations from a RandIntruder (or MCTSEnv) env driven by the goal-seeking heuristic."""
    Config.intruder_size = N
    np.random.seed(seed)
    env = m_ri.SingleAircraftEnv()
    obs = env.reset()
    roots = [np.array(obs, dtype=np.float64)]
    t = 0
    while len(roots) < count:
        a = mg.goal_seeking_action(env, Config.d_heading)
        obs, r, d, info = env.step(a)
        t += 1
        if d:
            obs = env.reset()
        if t % stride == 0:
            roots.append(np.array(obs, dtype=np.float64))
    return np.stack(roots)


def mt_pos():
    return int(np.random.get_state()[2])


def gen_moves(nodes, roots, name, width=6):
    states, actions, seeds, nxt, flags, rewards, pos = [], [], [], [], [], [], []
    k = 0
    for r in roots:
        for a0 in range(3):
            for a1 in range(3):
                seed = 1000 + k
                k += 1
                np.random.seed(seed)
                st = nodes.SingleAircraftState(state=r.copy())
                out = st.move((a0, a1))
                states.append(r)
                actions.append((a0, a1))
                seeds.append(seed)
                nxt.append(np.array(out.state, dtype=np.float64))
                flags.append((out.hit_wall, out.conflict, out.reach_goal))
                rewards.append(float(out.reward()))
                pos.append(mt_pos())
    mg.save(name, dict(N=(roots.shape[1] - 8) // width, width=width),
            dict(states=np.stack(states), actions=np.asarray(actions, np.int32), seeds=np.asarray(seeds, np.uint32),
                 next_states=np.stack(nxt), flags=np.asarray(flags, np.uint8), rewards=np.asarray(rewards),
                 rng_pos=np.asarray(pos, np.int32)))


def gen_search(nodes, search, roots, name, sims, depth, width=6):
    best, ca, cq, cn, rn, pos, nn, seeds = [], [], [], [], [], [], [], []

    def count_nodes(n):
        return 1 + sum(count_nodes(c) for c in n.children)

    for k, r in enumerate(roots):
        seed = 5000 + k
        np.random.seed(seed)
        root = nodes.SingleAircraftNode(state=nodes.SingleAircraftState(state=r.copy()))
        m = search.MCTS(root)
        b = m.best_action(sims, depth)
        best.append(b.state.prev_action)
        acts = np.full((9, 2), -1, np.int32)
        q = np.zeros(9)
        n = np.zeros(9)
        for i, c in enumerate(root.children):
            acts[i] = c.state.prev_action
            q[i], n[i] = c.q, c.n
        ca.append(acts); cq.append(q); cn.append(n)
        rn.append(root.n)
        pos.append(mt_pos())
        nn.append(count_nodes(root))
        seeds.append(seed)
    mg.save(name, dict(N=(roots.shape[1] - 8) // width, sims=sims, depth=depth, width=width),
            dict(states=roots, seeds=np.asarray(seeds, np.uint32), best_action=np.asarray(best, np.int32),
                 child_action=np.stack(ca), child_q=np.stack(cq), child_n=np.stack(cn), root_n=np.asarray(rn),
                 rng_pos=np.asarray(pos, np.int32), n_nodes=np.asarray(nn, np.int32)))


def main():
    ref = "/root/reference"
    sys.dont_write_bytecode = True
    Config, m_ri, nodes, search, mcfg = load_mcts(ref)
    for N, seed in ((8, 31), (20, 32)):
        roots = collect_roots(Config, m_ri, N, seed, 12, 37)
        gen_moves(nodes, roots, "g11_mcts_move_n%d" % N)
        gen_search(nodes, search, roots[:8], "g12_mcts_search_n%d" % N, mcfg.no_simulation, mcfg.search_depth)
    # hand-placed roots so that every flag of move() occurs: intruder ahead (conflict), ownship about to
    # leave the map (hit_wall, checked before conflict), goal a few frames ahead (reach_goal)
    base = collect_roots(Config, m_ri, 8, 34, 3, 29)
    stress = []
    for r in base:
        own = r[-8:].copy()
        hx, hy = np.cos(own[5]), np.sin(own[5])
        a = r.copy(); a[0] = own[0] + 30 * hx; a[1] = own[1] + 30 * hy; a[2] = -hx; a[3] = -hy; stress.append(a)
        b = r.copy(); b[-8] = 796.5; b[-7] = 400.0; b[-3] = 0.0; b[0:48:6] = np.linspace(100, 500, 8); b[1:48:6] = 100.0; stress.append(b)
        c = r.copy(); c[-2] = own[0] + 25 * hx; c[-1] = own[1] + 25 * hy; stress.append(c)
        d = r.copy(); d[6] = own[0] + 17 * hx; d[7] = own[1] + 17 * hy; d[-2] = own[0] + 3 * hx; d[-1] = own[1] + 3 * hy
        stress.append(d)   # conflict masks goal in the same frame
    gen_moves(nodes, np.stack(stress), "g11_mcts_move_n8_flags")
    gen_search(nodes, search, np.stack(stress)[:8], "g12_mcts_search_n8_flags", mcfg.no_simulation, mcfg.search_depth)
    # position/speed noise switched on (the shipped MCTS config has both at 0): exercises the gauss cache order
    keep = (mcfg.position_sigma, mcfg.speed_sigma)
    mcfg.position_sigma, mcfg.speed_sigma = 10 / 30, 2 / 30
    try:
        roots = collect_roots(Config, m_ri, 8, 33, 6, 41)
        gen_moves(nodes, roots, "g11_mcts_move_n8_noisy")
    finally:
        mcfg.position_sigma, mcfg.speed_sigma = keep
    main4(Config, mcfg, search)


def main4(Config, mcfg, search):
    """The 4-wide model (nodes_single.py) on SingleAircraftMCTSEnv observations."""
    import nodes_single as nodes4
    import SingleAircraftMCTSEnv as m_mcts
    for N, seed in ((8, 41), (20, 42)):
        roots = collect_roots(Config, m_mcts, N, seed, 10, 37)
        gen_moves(nodes4, roots, "g16_mcts4_move_n%d" % N, width=4)
        gen_search(nodes4, search, roots[:6], "g17_mcts4_search_n%d" % N, mcfg.no_simulation, mcfg.search_depth, width=4)
    base = collect_roots(Config, m_mcts, 8, 44, 3, 29)
    stress = []
    for r in base:
        own = r[-8:].copy()
        hx, hy = np.cos(own[5]), np.sin(own[5])
        a = r.copy(); a[0] = own[0] + 30 * hx; a[1] = own[1] + 30 * hy; a[2] = -hx; a[3] = -hy; stress.append(a)
        b = r.copy(); b[-8] = 796.5; b[-7] = 400.0; b[-3] = 0.0; b[0:32:4] = np.linspace(100, 500, 8); b[1:32:4] = 100.0; stress.append(b)
        c = r.copy(); c[-2] = own[0] + 25 * hx; c[-1] = own[1] + 25 * hy; stress.append(c)
        # the LAST intruder sits on the ownship's path: the model ignores it ((len - 9) // 4), no conflict
        d = r.copy(); d[28] = own[0] + 20 * hx; d[29] = own[1] + 20 * hy; d[30] = -hx; d[31] = -hy; stress.append(d)
    gen_moves(nodes4, np.stack(stress), "g16_mcts4_move_n8_flags", width=4)
    keep = (mcfg.position_sigma, mcfg.speed_sigma)
    mcfg.position_sigma, mcfg.speed_sigma = 10 / 30, 2 / 30
    try:
        roots = collect_roots(Config, m_mcts, 8, 43, 5, 41)
        gen_moves(nodes4, roots, "g16_mcts4_move_n8_noisy", width=4)
    finally:
        mcfg.position_sigma, mcfg.speed_sigma = keep
    main_zero(Config, mcfg)


def main_zero(Config, mcfg):
    """Zero-noise move() cases of the 4-wide model: deterministic, RNG-independent."""
    import nodes_single as nodes4
    import SingleAircraftMCTSEnv as m_mcts
    keep = (mcfg.position_sigma, mcfg.speed_sigma, mcfg.heading_sigma)
    mcfg.position_sigma, mcfg.speed_sigma, mcfg.heading_sigma = 0.0, 0.0, 0.0
    try:
        for N, seed in ((8, 61), (20, 62)):
            roots = collect_roots(Config, m_mcts, N, seed, 8, 31)
            gen_moves(nodes4, roots, "g18_mcts4_move_zero_n%d" % N, width=4)
        base = collect_roots(Config, m_mcts, 8, 64, 3, 29)
        stress = []
        for r in base:
            own = r[-8:].copy()
            hx, hy = np.cos(own[5]), np.sin(own[5])
            a = r.copy(); a[0] = own[0] + 30 * hx; a[1] = own[1] + 30 * hy; a[2] = -hx; a[3] = -hy; stress.append(a)
            b = r.copy(); b[-8] = 796.5; b[-7] = 400.0; b[-3] = 0.0; b[0:32:4] = np.linspace(100, 500, 8); b[1:32:4] = 100.0; stress.append(b)
            c = r.copy(); c[-2] = own[0] + 25 * hx; c[-1] = own[1] + 25 * hy; stress.append(c)
            d = r.copy(); d[28] = own[0] + 20 * hx; d[29] = own[1] + 20 * hy; d[30] = -hx; d[31] = -hy; stress.append(d)
        gen_moves(nodes4, np.stack(stress), "g18_mcts4_move_zero_n8_flags", width=4)
    finally:
        mcfg.position_sigma, mcfg.speed_sigma, mcfg.heading_sigma = keep


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "zero":      # only the g18 fixtures
        sys.dont_write_bytecode = True
        _Config, _m_ri, _nodes, _search, _mcfg = load_mcts("/root/reference")
        main_zero(_Config, _mcfg)
    else:
        main()
