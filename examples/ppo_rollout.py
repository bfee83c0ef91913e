#!/usr/bin/env python3
"""The minibatch collector of baselines' PPO (Algorithms/baselines-master/baselines/ppo2/runner.py:20-60) on top of the
batched env, with nothing leaving the device: `nsteps` closed-loop steps of B envs — policy forward, `env.step` — into
preallocated [nsteps, B, ...] slabs, then the bootstrapped GAE(lambda) returns of runner.py:46-58 as one reverse sweep
of device tensors.  What `Runner.run()` hands to the learner (obs, returns, masks, actions, values, neglogpacs, flattened
env-major like its `sf01`) is returned as torch tensors; the VecEnv semantics it relies on are the env's own: float32
rewards, auto-reset with the reset observation on `done` (dummy_vec_env.py:46-57).

  python examples/ppo_rollout.py --envs 4096 --intruders 8 --nsteps 128 --updates 3
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import uamenv  # noqa: E402


class MlpPolicy(torch.nn.Module):
    """A small actor-critic (two tanh layers, a categorical head over the 9 discrete actions and a value head): the shape
    of baselines' `mlp` policy, enough to close the loop with a learned policy."""

    def __init__(self, obs_dim, n_actions=9, hidden=64):
        super().__init__()
        self.body = torch.nn.Sequential(torch.nn.Linear(obs_dim, hidden), torch.nn.Tanh(),
                                        torch.nn.Linear(hidden, hidden), torch.nn.Tanh())
        self.pi = torch.nn.Linear(hidden, n_actions)
        self.vf = torch.nn.Linear(hidden, 1)

    def forward(self, obs):
        h = self.body(obs)
        return torch.distributions.Categorical(logits=self.pi(h)), self.vf(h).squeeze(-1)

    @torch.no_grad()
    def step(self, obs):
        """(actions int32[B], values[B], neglogpacs[B]) — model.step of runner.py:29"""
        dist, v = self(obs)
        a = dist.sample()
        return a.to(torch.int32), v, -dist.log_prob(a)

    @torch.no_grad()
    def value(self, obs):
        return self(obs)[1]


class DeviceRunner:
    """`Runner` (ppo2/runner.py) for a `BatchedSingleAircraftEnv`: same fields, same arithmetic, device tensors."""

    def __init__(self, env, model, nsteps, gamma=0.99, lam=0.95):
        self.env, self.model, self.nsteps, self.gamma, self.lam = env, model, int(nsteps), float(gamma), float(lam)
        B, D, dev = env.num_envs, env.obs_dim, env.device
        self.obs = env.reset().clone()                                   # AbstractEnvRunner: self.obs[:] = env.reset()
        self.dones = torch.zeros(B, dtype=torch.bool, device=dev)
        T = self.nsteps
        self.mb_obs = torch.empty((T, B, D), dtype=torch.float32, device=dev)
        self.mb_rewards = torch.empty((T, B), dtype=torch.float32, device=dev)
        self.mb_actions = torch.empty((T, B), dtype=torch.int32, device=dev)
        self.mb_values = torch.empty((T, B), dtype=torch.float32, device=dev)
        self.mb_neglogpacs = torch.empty((T, B), dtype=torch.float32, device=dev)
        self.mb_dones = torch.empty((T, B), dtype=torch.bool, device=dev)
        self._done_u8 = torch.empty((T, B), dtype=torch.uint8, device=dev)

    def run(self):
        env, T = self.env, self.nsteps
        for t in range(T):                                               # runner.py:26-39
            actions, values, neglogpacs = self.model.step(self.obs)
            self.mb_obs[t].copy_(self.obs)
            self.mb_actions[t].copy_(actions)
            self.mb_values[t].copy_(values)
            self.mb_neglogpacs[t].copy_(neglogpacs)
            self.mb_dones[t].copy_(self.dones)
            obs = env.step_into(actions, self.mb_rewards[t], self._done_u8[t])   # rewards / dones straight into the slabs
            self.obs.copy_(obs)
            self.dones = self._done_u8[t].bool()
        last_values = self.model.value(self.obs)                         # runner.py:44
        returns, advs = gae_returns(self.mb_rewards, self.mb_values, self.mb_dones, self.dones, last_values,
                                    self.gamma, self.lam)
        return tuple(sf01(x) for x in (self.mb_obs, returns, self.mb_dones, self.mb_actions, self.mb_values,
                                       self.mb_neglogpacs)) + (sf01(advs),)


def gae_returns(rewards, values, dones, last_dones, last_values, gamma, lam):
    """runner.py:46-58: delta_t = r_t + gamma V_{t+1} (1 - done_{t+1}) - V_t, A_t = delta_t + gamma lam (1 - done_{t+1})
    A_{t+1}; returns = A + V.  `dones[t]` is the done flag the env reported BEFORE step t (mb_dones)."""
    T = rewards.shape[0]
    advs = torch.zeros_like(rewards)
    lastgaelam = torch.zeros_like(rewards[0])
    for t in reversed(range(T)):
        if t == T - 1:
            nextnonterminal = 1.0 - last_dones.to(rewards.dtype)
            nextvalues = last_values
        else:
            nextnonterminal = 1.0 - dones[t + 1].to(rewards.dtype)
            nextvalues = values[t + 1]
        delta = rewards[t] + gamma * nextvalues * nextnonterminal - values[t]
        lastgaelam = delta + gamma * lam * nextnonterminal * lastgaelam
        advs[t] = lastgaelam
    return advs + values, advs


def sf01(x):
    """swap axes 0 and 1, then flatten them (runner.py:62-67)"""
    return x.transpose(0, 1).reshape((x.shape[0] * x.shape[1],) + tuple(x.shape[2:]))


def ppo_update(model, opt, batch, clip=0.2, vf_coef=0.5, ent_coef=0.0, epochs=2, minibatches=4):
    """The clipped-surrogate update of ppo2 on one collected batch (model.train of ppo2/model.py, in torch)."""
    obs, returns, _, actions, values, neglogpacs, advs = batch
    advs = (advs - advs.mean()) / (advs.std() + 1e-8)
    n = obs.shape[0]
    for _ in range(epochs):
        perm = torch.randperm(n, device=obs.device)
        for idx in perm.chunk(minibatches):
            dist, v = model(obs[idx])
            nlp = -dist.log_prob(actions[idx].long())
            ratio = torch.exp(neglogpacs[idx] - nlp)
            pg = torch.max(-advs[idx] * ratio, -advs[idx] * torch.clamp(ratio, 1 - clip, 1 + clip)).mean()
            vclip = values[idx] + torch.clamp(v - values[idx], -clip, clip)
            vf = 0.5 * torch.max((v - returns[idx]) ** 2, (vclip - returns[idx]) ** 2).mean()
            loss = pg + vf_coef * vf - ent_coef * dist.entropy().mean()
            opt.zero_grad(set_to_none=True)
            loss.backward()
            opt.step()
    return float(loss.detach())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, default=4096)
    ap.add_argument("--intruders", type=int, default=8)
    ap.add_argument("--nsteps", type=int, default=128)
    ap.add_argument("--updates", type=int, default=3)
    ap.add_argument("--seed", type=int, default=0)
    args = ap.parse_args()
    torch.manual_seed(args.seed)
    env = uamenv.BatchedSingleAircraftEnv(args.envs, args.intruders, seed=args.seed)
    model = MlpPolicy(env.obs_dim).to(env.device)
    opt = torch.optim.Adam(model.parameters(), lr=3e-4)
    runner = DeviceRunner(env, model, args.nsteps)
    for u in range(args.updates):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        batch = runner.run()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        loss = ppo_update(model, opt, batch)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print(json.dumps({"update": u, "transitions": int(batch[0].shape[0]), "collect_s": t1 - t0, "learn_s": t2 - t1,
                          "env_steps_per_sec": batch[0].shape[0] / (t1 - t0), "mean_return": float(batch[1].mean()),
                          "loss": loss}))
    env.close()


if __name__ == "__main__":
    main()
