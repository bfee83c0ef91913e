"""Batched MCTS planner over the HIP C-ABI: `MCTS(root).best_action(no_simulations, search_depth)` of
Algorithms/MCTS/search_single.py:8-13 for B root states in one launch (one wavefront per root).

A root state is what the reference builds with `SingleAircraftState(state=last_observation)` from the
env's raw observation: float64 `[B, 6N+8]` for the RandIntruder env and nodes_single_randintru.py
(Algorithms/MCTS/Agent_RandInt.py:38-41, `model="randintru"`), `[B, 4N+8]` for SingleAircraftMCTSEnv and
nodes_single.py (Algorithms/MCTS/Agent.py:36-40, `model="straight4"`, upstream quirks included).
"""
import ctypes as C

from . import _capi as capi


MODELS = {"randintru": 0, "straight4": 1}


def mcts_config(num_intruders, **over):
    cfg = capi.UamMctsConfig()
    capi.check(capi.lib().uam_mcts_config_default(C.byref(cfg), int(num_intruders)))
    if isinstance(over.get("model"), str):
        over["model"] = MODELS[over["model"]]
    for k, v in over.items():
        if not hasattr(cfg, k):
            raise AttributeError("uam_mcts_config has no field %r" % k)
        setattr(cfg, k, v)
    return cfg


class BatchedMCTS:
    def __init__(self, num_intruders, seed=0, device=None, cfg=None, **over):
        import torch
        self._torch = torch
        if not torch.cuda.is_available():
            raise RuntimeError("uamenv needs a HIP GPU; there is no CPU backend")
        self._lib = capi.lib()
        self.device = torch.device("cuda", torch.cuda.current_device() if device is None else
                                   (device if isinstance(device, int) else torch.device(device).index or 0))
        self.cfg = cfg if cfg is not None else mcts_config(num_intruders, seed=int(seed) & 0xFFFFFFFFFFFFFFFF, **over)
        self.N = int(self.cfg.num_intruders)
        self.state_dim = (4 if int(self.cfg.model) == 1 else 6) * self.N + 8
        self._ws = None
        self.decisions = 0

    def _stream(self):
        return C.c_void_p(self._torch.cuda.current_stream(self.device).cuda_stream)

    def _roots(self, roots):
        torch = self._torch
        r = torch.as_tensor(roots).to(device=self.device, dtype=torch.float64).contiguous()
        if r.dim() != 2 or r.shape[1] != self.state_dim:
            raise ValueError("roots must be [B, %d], got %s" % (self.state_dim, tuple(r.shape)))
        return r

    def roots_from_env(self, env):
        """float64 root states straight from a `BatchedSingleAircraftEnv`'s device state (RandIntruder env for
        the 6-wide model, any straight-line env for the 4-wide one)."""
        torch = self._torch
        held = int(env.cfg.dynamics) == capi.DYN_RANDINTRUDER or int(env.cfg.obs_mode) == capi.OBS_RAW6
        if (6 if held else 4) * self.N + 8 != self.state_dim or int(env.cfg.num_intruders) != self.N:
            raise ValueError("env variant / intruder count does not match this planner's state layout")
        out = torch.empty((env.num_envs, self.state_dim), dtype=torch.float64, device=self.device)
        capi.check(self._lib.uam_mcts_roots_from_env(env._h, out.data_ptr(), self._stream()))
        return out

    def best_action(self, roots, root_id0=0, want_children=False):
        """-> actions int32 [B, 2] (and the root children's (q, n) [B, 9] in expansion order)."""
        torch = self._torch
        r = self._roots(roots)
        B = int(r.shape[0])
        need = int(self._lib.uam_mcts_workspace_bytes(C.byref(self.cfg), B))
        if need < 0:
            capi.check(-22)
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty((need,), dtype=torch.uint8, device=self.device)
        acts = torch.empty((B, 2), dtype=torch.int32, device=self.device)
        cq = torch.empty((B, 9), dtype=torch.float64, device=self.device) if want_children else None
        cn = torch.empty((B, 9), dtype=torch.float64, device=self.device) if want_children else None
        capi.check(self._lib.uam_mcts_best_action(
            C.byref(self.cfg), self.device.index, r.data_ptr(), B, int(root_id0) & 0xFFFFFFFF, self._ws.data_ptr(),
            acts.data_ptr(), None if cq is None else cq.data_ptr(), None if cn is None else cn.data_ptr(), self._stream()))
        self.decisions += B
        return (acts, cq, cn) if want_children else acts

    def move(self, states, actions, root_id0=0, sim=0, move=0):
        """SingleAircraftState(state).move(action) per row -> (next [B, 6N+8], flags uint8 [B,3], reward [B])."""
        torch = self._torch
        s = self._roots(states)
        B = int(s.shape[0])
        a = torch.as_tensor(actions).to(device=self.device, dtype=torch.int32).contiguous().view(B, 2)
        nxt = torch.empty_like(s)
        flags = torch.empty((B, 3), dtype=torch.uint8, device=self.device)
        rew = torch.empty((B,), dtype=torch.float64, device=self.device)
        capi.check(self._lib.uam_mcts_move(C.byref(self.cfg), self.device.index, s.data_ptr(), a.data_ptr(), B,
                                           int(root_id0) & 0xFFFFFFFF, int(sim), int(move), nxt.data_ptr(),
                                           flags.data_ptr(), rew.data_ptr(), self._stream()))
        return nxt, flags, rew
