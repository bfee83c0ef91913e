"""NumPy-facing VecEnv adapter: the contract of
Algorithms/baselines-master/baselines/common/vec_env/__init__.py:26-135 (`num_envs`, `observation_space`,
`action_space`, `reset`, `step_async`, `step_wait`, `step`, `close`, `get_images`, `render`, `unwrapped`) as
`DummyVecEnv` fulfils it (dummy_vec_env.py:18-57: float32 rewards, bool dones, list of info dicts, obs of a done
env = its reset obs), with `bench.Monitor`'s `info['episode']` (bench/monitor.py:58-75) coming from the device
accumulators — so a baselines-style learner that wants host arrays plugs in without any per-env Python.

Everything still runs in the one HIP launch per step of `BatchedSingleAircraftEnv`; this class only adds the
PCIe leg: actions host->device and obs/reward/done/info device->host through page-locked staging buffers, one
stream synchronisation per step.  That leg, not the kernel, bounds this interface.
"""
import time

import numpy as np

from . import _capi as capi
from .envs import BatchedSingleAircraftEnv, Config


class HostInfos:
    """`infos` of one step: a sequence of `{'result': code}` dicts built on demand from host arrays, with
    `'episode': {'r', 'l', 't'}` on the envs whose episode just ended."""

    def __init__(self, codes, done, ep_ret, ep_len, t):
        self.codes, self._done, self._ret, self._len, self._t = codes, done, ep_ret, ep_len, t

    def __len__(self):
        return len(self.codes)

    def __getitem__(self, i):
        info = {"result": capi.INFO_CODES[int(self.codes[i])]}
        if self._ret is not None and self._done[i]:
            info["episode"] = {"r": round(float(self._ret[i]), 6), "l": int(self._len[i]), "t": self._t}
        return info

    def __iter__(self):
        for i in range(len(self)):
            yield self[i]

    def episodes(self):
        """[(env index, return, length)] of the episodes that ended at this step (no per-env dict building)."""
        if self._ret is None:
            return []
        idx = np.nonzero(self._done)[0]
        return [(int(i), float(self._ret[i]), int(self._len[i])) for i in idx]


class UamVecEnv:
    """baselines `VecEnv` duck type over `BatchedSingleAircraftEnv` (NumPy in, NumPy out)."""

    def __init__(self, num_envs, num_intruders=None, variant="discrete9", monitor=True, config=Config, **kw):
        self.venv = BatchedSingleAircraftEnv(num_envs, num_intruders, variant, monitor=monitor, config=config, **kw)
        torch = self.venv._torch
        self._torch = torch
        v = self.venv
        self.num_envs, self.observation_space, self.action_space = v.num_envs, v.observation_space, v.action_space
        B, D = v.num_envs, v.obs_dim
        am = v.cfg.action_mode
        act_shape = (B, 2) if am in (capi.ACT_PAIR, capi.ACT_CONTINUOUS2) else (B,)
        act_dt = torch.float32 if am in (capi.ACT_CONTINUOUS1, capi.ACT_CONTINUOUS2) else torch.int32
        pin = dict(pin_memory=True)
        self._h_act = torch.empty(act_shape, dtype=act_dt, **pin)
        self._d_act = torch.empty(act_shape, dtype=act_dt, device=v.device)
        # two alternating observation buffers: the array a call returns stays untouched until the call after the
        # next one (a learner keeps obs_t while it asks for obs_t+1), so no defensive 4*B*D-byte host copy per step
        self._h_obs2 = [torch.empty((B, D), dtype=torch.float32, **pin) for _ in range(2)]
        self._flip = 0
        # every per-env scalar output lives in ONE device slab (episode return f64 | reward f32 | episode length i32 |
        # done u8 | info u8) so that it crosses PCIe in a single copy: small batches are copy-latency bound
        self._d_slab = torch.empty((B * 18,), dtype=torch.uint8, device=v.device)
        self._h_slab = torch.empty((B * 18,), dtype=torch.uint8, **pin)

        def views(raw):
            return (raw[:8 * B].view(torch.float64), raw[8 * B:12 * B].view(torch.float32),
                    raw[12 * B:16 * B].view(torch.int32), raw[16 * B:17 * B], raw[17 * B:])
        self._d_ret, d_rew, self._d_len, d_done, d_info = views(self._d_slab)
        self._h_ret, self._h_rew, self._h_len, self._h_done, self._h_info = views(self._h_slab)
        v.bind_outputs(d_rew, d_done, d_info)
        self._tstart = time.time()
        self.closed = False

    @property
    def unwrapped(self):
        return self

    def seed(self, seeds=None):
        return self.venv.seed(seeds)

    def _next_obs_buf(self):
        self._flip ^= 1
        return self._h_obs2[self._flip]

    def reset(self):
        obs = self.venv.reset()
        h = self._next_obs_buf()
        h.copy_(obs, non_blocking=True)
        self._torch.cuda.current_stream(self.venv.device).synchronize()
        return h.numpy()

    def step_async(self, actions):
        a = np.asarray(actions)
        self._h_act.numpy()[...] = a.reshape(self._h_act.shape)      # casts to the action dtype like DummyVecEnv's env.step(a)
        self._d_act.copy_(self._h_act, non_blocking=True)
        self.venv.step_async(self._d_act)

    def step_wait(self):
        v = self.venv
        obs, rew, done, infos = v.step_wait()
        h_obs = self._next_obs_buf()
        h_obs.copy_(obs, non_blocking=True)
        if v.monitor:
            v.episode_stats(self._d_ret, self._d_len)
        self._h_slab.copy_(self._d_slab, non_blocking=True)
        self._torch.cuda.current_stream(v.device).synchronize()
        done_np = self._h_done.numpy().astype(bool)
        infos = HostInfos(self._h_info.numpy().copy(), done_np,
                          self._h_ret.numpy().copy() if v.monitor else None,
                          self._h_len.numpy().copy() if v.monitor else None, round(time.time() - self._tstart, 6))
        return h_obs.numpy(), self._h_rew.numpy().copy(), done_np, infos

    def step(self, actions):
        self.step_async(actions)
        return self.step_wait()

    def get_images(self):
        return []          # no renderer on the GPU box

    def render(self, mode="human"):
        return None

    def close(self):
        if not self.closed:
            self.venv.close()
        self.closed = True
