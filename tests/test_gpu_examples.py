"""The example callers either side of the path (examples/): they must run on the HIP env and keep the arithmetic of the
reference code they stand in for."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "examples"))

pytestmark = pytest.mark.gpu


def test_ppo_rollout_collector_matches_the_runner_arithmetic():
    """examples/ppo_rollout.py stands in for baselines/ppo2/runner.py:20-60.  The slabs it fills must be what the loop
    saw (obs before the step, the env's float32 rewards, the done flag reported BEFORE each step), and its device-side
    GAE(lambda) must equal the runner's reverse loop (runner.py:46-58) restated here in NumPy on the same slabs."""
    import torch
    assert torch.cuda.is_available()
    import uamenv
    import ppo_rollout as pr
    torch.manual_seed(3)
    B, N, T = 512, 8, 24
    env = uamenv.BatchedSingleAircraftEnv(B, N, seed=11)
    twin = uamenv.BatchedSingleAircraftEnv(B, N, seed=11)
    model = pr.MlpPolicy(env.obs_dim).to(env.device)
    runner = pr.DeviceRunner(env, model, T, gamma=0.99, lam=0.95)
    first_obs = runner.obs.clone()
    obs, returns, masks, actions, values, neglogpacs, advs = runner.run()
    assert obs.shape == (B * T, env.obs_dim) and returns.shape == (B * T,) and actions.dtype == torch.int32
    # env-major flattening (sf01): row e * T + t is env e at step t
    assert torch.equal(obs.view(B, T, -1)[:, 0], first_obs)
    # replay the recorded actions on a twin handle: same observations, rewards and dones
    o = twin.reset()
    a_tb = actions.view(B, T).t().contiguous()
    d_prev = torch.zeros(B, dtype=torch.bool, device=env.device)
    for t in range(T):
        assert torch.equal(obs.view(B, T, -1)[:, t], o), t
        assert torch.equal(masks.view(B, T)[:, t], d_prev), t
        o, r, d, _ = twin.step(a_tb[t])
        assert torch.equal(runner.mb_rewards[t], r), t
        o = o.clone(); d_prev = d.clone()
    # the runner's GAE loop, in float32 NumPy
    rew = runner.mb_rewards.cpu().numpy(); val = runner.mb_values.cpu().numpy(); dn = runner.mb_dones.cpu().numpy()
    last_v = model.value(runner.obs).cpu().numpy(); last_d = runner.dones.cpu().numpy()
    adv = np.zeros_like(rew); lastgaelam = np.zeros(B, dtype=np.float32)
    for t in reversed(range(T)):
        nnt = 1.0 - (last_d if t == T - 1 else dn[t + 1]).astype(np.float32)
        nv = last_v if t == T - 1 else val[t + 1]
        delta = rew[t] + np.float32(0.99) * nv * nnt - val[t]
        adv[t] = lastgaelam = delta + np.float32(0.99 * 0.95) * nnt * lastgaelam
    want = (adv + val).swapaxes(0, 1).reshape(-1)
    np.testing.assert_allclose(returns.cpu().numpy(), want, rtol=1e-5, atol=1e-6)
    # and one clipped-surrogate update runs on the batch
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    loss = pr.ppo_update(model, opt, (obs, returns, masks, actions, values, neglogpacs, advs), epochs=1, minibatches=2)
    assert np.isfinite(loss)
    env.close(); twin.close()
