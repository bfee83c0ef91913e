"""graph_safe handles: policy + step captured once into a HIP graph and replayed must be bit-identical to the eager
loop on an ordinary handle (closed loop: every step's action comes from the observation / state of the step before)."""
import numpy as np
import pytest

import parity_util as pu

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def uamenv():
    import torch
    assert torch.cuda.is_available()
    import uamenv as m
    m._capi.lib()
    return m


@pytest.mark.parametrize("variant,B,N,kind", [("discrete9", 4096, 8, "goal"), ("randintruder", 2048, 64, "goal"),
                                              ("discrete9", 600, 32, "random"), ("continuous", 8192, 32, "goal")])
def test_graph_replay_equals_eager_closed_loop(uamenv, variant, B, N, kind):
    import torch
    steps, replays = 4, 12
    g_env = uamenv.BatchedSingleAircraftEnv(B, N, variant=variant, seed=31, graph_safe=1)
    e_env = uamenv.BatchedSingleAircraftEnv(B, N, variant=variant, seed=31)
    assert g_env.steps_per_launch(8) == 1                       # graph-safe handles never fuse chunks
    og, oe = g_env.reset(), e_env.reset()
    assert torch.equal(og, oe)
    acts_g = g_env.policy_actions(kind)        # allocates the static action buffer (a graph-safe handle numbers its
    acts_e = torch.empty_like(acts_g)          # policy calls by the device step counter: this call consumed nothing)

    loop = uamenv.GraphedLoop(g_env, lambda obs: g_env.policy_actions(kind, out=acts_g), steps=steps)
    hold_sh = variant == "randintruder"
    n_done = 0
    for r in range(replays):
        obs, rew, done, info = loop.replay()
        for j in range(steps):
            a = e_env.policy_actions(kind, out=acts_e)
            o, rr, dd, ii = e_env.step(a)
            assert torch.equal(rew[j], rr) and torch.equal(done[j], dd.view(torch.uint8)) and torch.equal(info[j], ii.codes), (r, j)
            n_done += int(dd.sum().item())
        assert torch.equal(obs, o), r
        if r % 4 == 3:
            pu.compare_states(g_env.get_state(), e_env.get_state(), "after replay %d" % r, hold_sh=hold_sh)
    st = g_env.get_state()
    assert int(st["step_index"]) == steps * replays             # read back from the device counters
    if kind == "goal":
        assert n_done > 0, "the loop should have finished (and auto-reset) some episodes"
    g_env.close(); e_env.close()


def test_graph_safe_is_a_plain_env_too_and_rejects_mt19937(uamenv, oracle):
    """Eager stepping of a graph_safe handle (per-env device counters, one-step noise launch) against the oracle."""
    pu.run_lockstep(uamenv, oracle, 300, 32, 60, policy="goal", rng="philox", precision="f32", graph_safe=1, check_state_every=10)
    with pytest.raises(uamenv.UamError):
        uamenv.BatchedSingleAircraftEnv(8, 8, rng="mt19937", precision="exact", graph_safe=1)
    with pytest.raises(ValueError):
        uamenv.GraphedLoop(uamenv.BatchedSingleAircraftEnv(8, 8), lambda o: None)
