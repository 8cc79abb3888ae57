"""evomotion_amd — MI355X-native vectorised robot_walk environment + PPO rollout path.

Product code: HIP kernels and the C ABI in csrc/ (libevomotion_hip.so), host mirrors of the reference's
Environment / Agent interfaces in env.py / agent.py.  The CPU oracle under /oracle is test infrastructure
and is never imported from here."""
from ._lib import DEFAULT_SKELETON, EvmError, LIB_PATH  # noqa: F401
from .env import RolloutStep, Step, VecRobotJump, VecRobotWalk, get_environment  # noqa: F401
from .agent import (ActorModule, CriticModule, FusedActorCritic, PpoGaeAgent, RandomAgent, TrajectoryReplayBuffer, VecPpoGaeAgent,  # noqa: F401,E402
                    truncated_normal_log_pdf, truncated_normal_entropy, truncated_normal_sample)
from .checkpoint import load_into, load_th, save_th  # noqa: F401,E402
from .metrics import LossMeter  # noqa: F401,E402
from .replay import ReplayRing  # noqa: F401,E402
from .sac import EntropyParameter, QNetworkModule, ReplayBuffer, SoftActorCriticAgent, VecSacAgent  # noqa: F401,E402
