"""MI355X-native batched single-aircraft guidance / collision-avoidance env.

The directory name carries the upstream project's name; import it as `uamenv`
(the sibling `uamenv/` package aliases this directory).  Only the hot path of
the reference lives here: HIP kernels + C-ABI (`csrc/`) and the host-side mirror
of the env interface (`envs.py`).
"""
from . import _capi
from ._capi import UamError, build, default_config
from .envs import (BatchedSingleAircraftEnv, Config, GraphedLoop, LazyInfos, PackagedConfig, PackagedSingleAircraft2Env, SingleAircraft2Env, SingleAircraftEnv,
                   SingleAircraftMCTSEnv, SingleAircraftMCTSRandIntruderEnv, SingleAircraftRandomEnv,
                   SingleAircraftDiscrete9HEREnv, SingleAircraftDiscrete3HEREnv, her_goal_reward,
                   her_input_reward, make_config)
from . import spaces
from .sharded import RolloutExchange, ShardedSingleAircraftEnv, shard_bounds
from .mcts import BatchedMCTS, mcts_config
from .vec_env import UamVecEnv, HostInfos

__all__ = [
    "GraphedLoop", "RolloutExchange", "PackagedConfig", "PackagedSingleAircraft2Env",
    "BatchedSingleAircraftEnv", "Config", "LazyInfos", "SingleAircraft2Env", "SingleAircraftEnv",
    "SingleAircraftMCTSEnv", "SingleAircraftMCTSRandIntruderEnv", "SingleAircraftRandomEnv",
    "SingleAircraftDiscrete9HEREnv", "SingleAircraftDiscrete3HEREnv", "her_goal_reward", "her_input_reward",
    "make_config",
    "UamError", "build", "default_config", "spaces", "ShardedSingleAircraftEnv", "shard_bounds", "BatchedMCTS", "mcts_config",
    "UamVecEnv", "HostInfos",
]
