"""uam_rollout (every step's observation kept, chunk of K steps per C call) and bench.py's own multi-rank path
on the GPU box: two gloo ranks sharing cuda:0, started by bench.py itself."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import parity_util as pu  # noqa: E402


@pytest.fixture(scope="module")
def uamenv():
    import uamenv as u
    u.build()
    return u


@pytest.mark.parametrize("variant,B,N,K,R,looped", [
    ("discrete9", 4096, 8, 7, 7, True),        # looped kernel, full rollout slab
    ("continuous", 8192, 32, 12, 2, True),     # BASELINE config 4's per-GPU shard at N=8: looped kernel, double buffer
    ("randintruder", 512, 64, 5, 1, True),     # one buffer overwritten every step
    ("her9", 700, 20, 9, 4, True),
    ("discrete9", 32768, 32, 5, 3, True),      # every wave slot taken by sweeping waves: the flavour without ownship wave
    ("discrete9", 65536, 32, 4, 2, True),      # grid larger than one resident round: the looped kernel in two rounds
    ("randintruder", 65536, 16, 3, 1, False),  # ... for the plain step only: this one queues K single-step launches
    ("discrete9", 300, 5, 6, 6, True),         # D = 28: 16-byte aligned rows only every 4th env
    # the three-wave form (sweeping + ownship + copy wave, uam_sweep_many3_kernel) in its other geometries
    ("mcts", 2048, 16, 21, 3, True),           # raw 4-wide rows, pair actions; 4 envs per block, 16 steps per ownship pass
    ("continuous", 16384, 32, 11, 2, True),    # two sweeping waves per block, 4 steps per ownship pass
    ("discrete9", 65536, 8, 5, 1, True),       # 64 envs per block: one step per ownship pass
    ("continuous2", 1000, 64, 19, 19, True),   # partial last block, K not a multiple of the pass length
    ("randintruder", 8192, 64, 9, 2, True),    # BASELINE config 5's shape: held speed / heading, six-wide rows, two sweeping waves
    ("randintruder", 4096, 32, 12, 3, True),   # ... with 8 lanes per env
])
def test_rollout_keeps_every_observation(uamenv, variant, B, N, K, R, looped):
    """Slot k % R of the rollout's obs buffer must hold exactly what uam_step would have returned at step k, and
    rewards / dones / infos / state must equal K separate steps bit for bit."""
    import torch
    a = uamenv.BatchedSingleAircraftEnv(B, N, variant=variant, seed=21)
    b = uamenv.BatchedSingleAircraftEnv(B, N, variant=variant, seed=21)
    a.reset(); b.reset()
    dev = a.device
    assert (a.steps_per_launch(K) == K) == looped, "unexpected launch form"
    if variant in ("mcts", "continuous2", "randintruder") and B != 65536 or (variant, B) in (("continuous", 16384), ("discrete9", 65536), ("continuous", 8192), ("discrete9", 4096)):
        want_kernel = "uam_sweep_many4_kernel" if (variant, B, N) == ("discrete9", 65536, 32) else "uam_sweep_many3_kernel"
        assert a.rollout_kernel(K) == want_kernel, a.rollout_kernel(K)
    for rep in range(2):
        tape = torch.stack([a.policy_actions("goal" if rep else "random", seed=rep * 50 + k) for k in range(K)]).contiguous()
        obs = torch.full((R, B, a.obs_dim), float("nan"), dtype=torch.float32, device=dev)
        rew = torch.empty((K, B), dtype=torch.float32, device=dev)
        done = torch.empty((K, B), dtype=torch.uint8, device=dev)
        info = torch.empty((K, B), dtype=torch.uint8, device=dev)
        a.rollout_into(tape, rew, done, info, obs)
        want = {}
        for k in range(K):
            o, r, d, i = b.step(tape[k])
            assert torch.equal(rew[k], r) and torch.equal(done[k], d.view(torch.uint8)) and torch.equal(info[k], i.codes), (rep, k)
            want[k % R] = o.clone()
        for slot, o in want.items():
            assert torch.equal(obs[slot], o), (rep, slot)
        pu.compare_states(a.get_state(), b.get_state(), "after rollout rep %d" % rep, hold_sh=variant == "randintruder")
    a.close(); b.close()


@pytest.mark.parametrize("variant,B,N,K", [
    ("continuous", 8192, 32, 420),      # BASELINE config 4's per-GPU shard: 8 envs per block, spawn wave, 8 steps per ownship pass
    ("discrete9", 4096, 8, 400),        # 4 envs per block, 16 steps per ownship pass
    ("mcts", 2048, 16, 300),            # raw rows, pair actions
    ("continuous", 16384, 32, 260),     # two sweeping waves per block: no room for a spawn wave
    ("discrete9", 1000, 64, 250),       # partial last block
])
def test_long_closed_loop_tape_through_the_looped_kernel(uamenv, variant, B, N, K):
    """Hundreds of steps in ONE launch from a tape that a goal-seeking closed loop recorded on a twin handle: goals, NMACs,
    conflicts, resets and the respawns around them all fall inside the launch — the paths where the looped kernel's
    helper waves run ahead of the sweep on records that a reset replaces (ownship wave) or draw respawns for a step that
    a reset makes obsolete (spawn wave).  Everything must equal the twin's K separate steps bit for bit."""
    import torch
    a = uamenv.BatchedSingleAircraftEnv(B, N, variant=variant, seed=77)
    b = uamenv.BatchedSingleAircraftEnv(B, N, variant=variant, seed=77)
    a.reset(); b.reset()
    dev = a.device
    assert a.rollout_kernel(K) == "uam_sweep_many3_kernel" and a.steps_per_launch(K) == K
    tape, want_r, want_d, want_i = [], [], [], []
    for k in range(K):
        act = b.policy_actions("goal", seed=k)
        o, r, d, i = b.step(act)
        tape.append(act.clone()); want_r.append(r.clone()); want_d.append(d.view(torch.uint8).clone()); want_i.append(i.codes.clone())
    tape = torch.stack(tape).contiguous()
    obs = torch.full((2, B, a.obs_dim), float("nan"), dtype=torch.float32, device=dev)
    rew = torch.empty((K, B), dtype=torch.float32, device=dev)
    done = torch.empty((K, B), dtype=torch.uint8, device=dev)
    info = torch.empty((K, B), dtype=torch.uint8, device=dev)
    a.rollout_into(tape, rew, done, info, obs)
    assert torch.equal(rew, torch.stack(want_r)) and torch.equal(done, torch.stack(want_d)) and torch.equal(info, torch.stack(want_i))
    assert torch.equal(obs[(K - 1) % 2], o)
    pu.compare_states(a.get_state(), b.get_state(), "after the long rollout")
    assert int(done.sum()) > B // 8, "too few episode ends inside the launch: %d" % int(done.sum())
    a.close(); b.close()


@pytest.mark.parametrize("variant,precision,B,N,K,R,one_launch", [
    ("discrete9", "exact", 4096, 32, 40, 3, True),      # resets inside the launch (goal-seeking tape: episodes end)
    ("continuous", "f32", 1000, 8, 9, 9, True),
    ("randintruder", "exact", 600, 64, 6, 2, True),     # heading-change draws: the window is refilled several times a step
    ("random_start", "exact", 300, 80, 30, 1, True),    # N = 80: a reset draws more than one round of the stream
    ("her9", "exact", 500, 20, 5, 5, False),            # the HER variants loop over single-step launches
    ("discrete9", "exact", 65536, 32, 8, 1, True),      # the benchmarked shape: two rounds of waves, resets in every launch
])
def test_mt19937_rollout_in_one_launch_equals_single_steps(uamenv, variant, precision, B, N, K, R, one_launch):
    """uam_sweep_mt_many_kernel: every wave runs its envs' K steps back to back (the MT19937 step is wave-local).  Rewards,
    dones, infos, the observation slots, the whole state and the position of every env's NumPy stream must equal K
    separate uam_step launches, across episode ends inside the launch."""
    import torch
    kw = dict(variant=variant, rng="mt19937", precision=precision, seed=77)
    a = uamenv.BatchedSingleAircraftEnv(B, N, **kw)
    b = uamenv.BatchedSingleAircraftEnv(B, N, **kw)
    a.reset(); b.reset()
    dev = a.device
    assert (a.steps_per_launch(K) == K) == one_launch, "unexpected launch form"
    ended = 0
    for rep in range(3):
        tape = torch.stack([a.policy_actions("goal" if (rep + k) % 4 else "random", seed=rep * 50 + k) for k in range(K)]).contiguous()
        obs = torch.full((R, B, a.obs_dim), float("nan"), dtype=torch.float32, device=dev)
        rew = torch.empty((K, B), dtype=torch.float32, device=dev)
        done = torch.empty((K, B), dtype=torch.uint8, device=dev)
        info = torch.empty((K, B), dtype=torch.uint8, device=dev)
        a.rollout_into(tape, rew, done, info, obs)
        want = {}
        for k in range(K):
            o, r, d, i = b.step(tape[k])
            assert torch.equal(rew[k], r) and torch.equal(done[k], d.view(torch.uint8)) and torch.equal(info[k], i.codes), (rep, k)
            want[k % R] = o.clone()
        for slot, o in want.items():
            assert torch.equal(obs[slot], o), (rep, slot)
        ended += int(done.sum())
        ga, gb = a.get_state(), b.get_state()
        pu.compare_states(ga, gb, "after rollout rep %d" % rep, hold_sh=variant == "randintruder")
        for f in ("mt_pos", "mt_key"):
            if f in ga:
                assert (ga[f] == gb[f]).all(), f
    if variant in ("discrete9", "random_start"):
        assert ended > 0, "the tape must end episodes inside a launch"
    a.close(); b.close()


@pytest.mark.parametrize("variant,B,N,K,R", [
    ("discrete9", 4096, 32, 24, 2),
    ("randintruder", 600, 64, 6, 6),
    ("random_start", 300, 80, 12, 1),
    ("discrete9", 65536, 32, 6, 1),
])
def test_exact_philox_rollout_in_one_launch_equals_single_steps(uamenv, variant, B, N, K, R):
    """uam_sweep_heavy_many_kernel (float64 positions, Philox): every block runs its envs' K steps back to back."""
    import torch
    kw = dict(variant=variant, rng="philox", precision="exact", seed=78)
    a = uamenv.BatchedSingleAircraftEnv(B, N, **kw)
    b = uamenv.BatchedSingleAircraftEnv(B, N, **kw)
    a.reset(); b.reset()
    dev = a.device
    assert a.steps_per_launch(K) == K
    for rep in range(3):
        tape = torch.stack([a.policy_actions("goal" if (rep + k) % 4 else "random", seed=rep * 50 + k) for k in range(K)]).contiguous()
        obs = torch.full((R, B, a.obs_dim), float("nan"), dtype=torch.float32, device=dev)
        rew = torch.empty((K, B), dtype=torch.float32, device=dev)
        done = torch.empty((K, B), dtype=torch.uint8, device=dev)
        info = torch.empty((K, B), dtype=torch.uint8, device=dev)
        a.rollout_into(tape, rew, done, info, obs)
        want = {}
        for k in range(K):
            o, r, d, i = b.step(tape[k])
            assert torch.equal(rew[k], r) and torch.equal(done[k], d.view(torch.uint8)) and torch.equal(info[k], i.codes), (rep, k)
            want[k % R] = o.clone()
        for slot, o in want.items():
            assert torch.equal(obs[slot], o), (rep, slot)
        pu.compare_states(a.get_state(), b.get_state(), "after rollout rep %d" % rep, hold_sh=variant == "randintruder")
    a.close(); b.close()


def test_rollout_rejects_bad_arguments(uamenv):
    import torch
    from uamenv import _capi as capi
    env = uamenv.BatchedSingleAircraftEnv(64, 8)
    env.reset()
    dev = env.device
    tape = torch.zeros((4, 64), dtype=torch.int32, device=dev)
    rew = torch.empty((4, 64), dtype=torch.float32, device=dev)
    done = torch.empty((4, 64), dtype=torch.uint8, device=dev)
    obs = torch.empty((2, 64, env.obs_dim), dtype=torch.float32, device=dev)
    L = capi.lib()
    assert L.uam_rollout(env._h, 4, tape.data_ptr(), obs.data_ptr(), 0, rew.data_ptr(), done.data_ptr(), None, None) == -22
    assert L.uam_rollout(env._h, 4, tape.data_ptr(), None, 2, rew.data_ptr(), done.data_ptr(), None, None) == -22
    assert L.uam_rollout(env._h, 0, tape.data_ptr(), obs.data_ptr(), 2, rew.data_ptr(), done.data_ptr(), None, None) == -22
    assert L.uam_steps_per_launch(env._h, 0) == 0 and L.uam_steps_per_launch(env._h, 1) == 1
    with pytest.raises(ValueError):
        env.rollout_into(tape, rew, done, None, obs[:, :32])
    env.close()


def _run_bench(extra, timeout=300):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + extra, capture_output=True, text=True,
                         timeout=timeout, cwd=ROOT)
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-3000:])
    lines = [l for l in out.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_two_rank_rehearsal_of_its_own_exchange_path(tmp_path):
    """`bench.py --gpus 2` with no rendezvous in the environment: it must start its two ranks itself (gloo, both on
    cuda:0 — the box has one GPU), step each rank's shard in chunks through uam_rollout, exchange one slab per
    chunk and print ONE line whose strong-scaling field partitions the batch.  Its `--dump-exchange` pass then steps
    ONE partitioned batch from a known tape: what rank 0's collectives gathered from the two HIP-filled slabs must
    equal, byte for byte, the rewards and dones of one unsharded handle on the same tape."""
    import numpy as np
    import torch
    import uamenv
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    dump = str(tmp_path / "gathered.npz")
    d = _run_bench(["--gpus", "2", "--backend", "gloo", "--same-device", "--steps", "70", "--warmup", "10", "--reps", "2",
                    "--batch", "8192", "--intruders", "8", "--chunk", "32", "--no-cpu-baseline", "--sustain-seconds", "0",
                    "--dump-exchange", dump])
    assert d["n_gpus"] == 2 and d["steps"] == 70 and d["scaling"] == "weak"
    assert d["config"]["envs_per_gpu"] == 8192 and d["config"]["envs_total"] == 16384
    assert len(d["reps_ms"]) == 2 and d["ms_per_step"] > 0
    assert abs(d["value"] - 16384 * 70 / (d["ms_per_step"] * 70e-3)) / d["value"] < 1e-9
    assert "gather" in d["config"]["exchange"] and "3 per rank" in d["config"]["exchange"]     # ceil(70 / 32)
    s = d["strong"]
    assert s["scaling"] == "strong" and s["variant"] == "continuous" and s["envs_per_gpu"] == 4096 and s["envs_total"] == 8192
    assert s["value"] > 0 and "cpu_baseline" not in d
    g = np.load(dump)
    assert int(g["world"]) == 2 and str(g["mode"]) == "gather" and int(g["collectives"]) == 2
    tape = g["tape"]
    K2, Btot = tape.shape
    assert Btot == 8192 and g["rew"].shape == (K2, Btot) and g["done"].shape == (K2, Btot)
    one = uamenv.BatchedSingleAircraftEnv(Btot, int(g["intruders"]), variant="continuous", seed=0, auto_reset=True)
    one.reset()
    t = torch.from_numpy(tape).to(one.device)
    rew = torch.empty((K2, Btot), dtype=torch.float32, device=one.device)
    done = torch.empty((K2, Btot), dtype=torch.uint8, device=one.device)
    info = torch.empty((K2, Btot), dtype=torch.uint8, device=one.device)
    half = K2 // 2
    one.rollout_into(t[:half].contiguous(), rew[:half], done[:half], info[:half])
    one.rollout_into(t[half:].contiguous(), rew[half:], done[half:], info[half:])
    torch.cuda.synchronize()
    assert np.array_equal(g["rew"].view(np.uint32), rew.cpu().numpy().view(np.uint32)), "gathered rewards differ from the unsharded batch"
    assert np.array_equal(g["done"], done.cpu().numpy()), "gathered dones differ from the unsharded batch"
    assert float(np.abs(g["rew"]).sum()) > 0
    one.close()


@pytest.mark.parametrize("variant,B,N,K,kind", [
    ("discrete9", 4096, 8, 24, "goal"),         # looped kernel, the ownship wave evaluates the policy
    ("continuous", 8192, 32, 16, "goal"),       # BASELINE config 4's shard
    ("randintruder", 2048, 64, 10, "goal"),
    ("discrete9", 1000, 32, 12, "random"),
    ("mcts", 3000, 20, 9, "random"),            # pair actions
    ("continuous2", 2048, 16, 14, "goal"),      # float pair actions (the packaged 2Env's semantics)
    ("continuous2", 777, 8, 10, "random"),
    ("discrete9", 32768, 32, 6, "goal"),        # the flavour without ownship wave
    ("discrete9", 65536, 32, 3, "goal"),        # not fused: a policy launch and a step launch per step
])
def test_rollout_policy_closes_the_loop_like_policy_plus_step(uamenv, variant, B, N, K, kind):
    """uam_rollout_policy (a fresh action from the device-side policy before EVERY step, inside the launch where the
    chunk is fused) must equal K x (uam_policy, uam_step) bit for bit — rewards, dones, infos, the actions taken, every
    observation and the final state — including across episode ends (goal seeking finishes episodes)."""
    import torch
    a = uamenv.BatchedSingleAircraftEnv(B, N, variant=variant, seed=77)
    b = uamenv.BatchedSingleAircraftEnv(B, N, variant=variant, seed=77)
    a.reset(); b.reset()
    acts_b = None
    dones = 0
    for rep in range(3 if kind == "goal" and B <= 8192 else 2):
        reps_K = K if rep < 2 else 200                     # a long chunk so that goal seeking ends episodes
        obs = torch.full((reps_K, B, a.obs_dim), float("nan"), dtype=torch.float32, device=a.device)
        o_, rew, done, info, acts = a.rollout_policy(reps_K, kind=kind, seed=5, obs_out=obs)
        for k in range(reps_K):
            acts_b = b.policy_actions(kind, seed=5, out=acts_b)
            assert torch.equal(acts[k], acts_b), (rep, k)
            o, r, d, i = b.step(acts_b)
            assert torch.equal(rew[k], r) and torch.equal(done[k], d.view(torch.uint8)) and torch.equal(info[k], i.codes), (rep, k)
            assert torch.equal(obs[k], o), (rep, k)
            dones += int(d.sum().item())
        pu.compare_states(a.get_state(), b.get_state(), "after policy rollout %d" % rep, hold_sh=variant == "randintruder")
    if kind == "goal" and B <= 8192:
        assert dones > 0, "no episode ended: the reset hand-over to the policy was not exercised"
    a.close(); b.close()
