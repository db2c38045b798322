from .misc import temporal_difference, value_update  # noqa: F401
from .ddpg import ddpg_update  # noqa: F401
from .td3 import td3_update  # noqa: F401
from .reinforce import ChooseREINFORCE, reinforce_update  # noqa: F401
from .bcq import bcq_update  # noqa: F401
from .sac import soft_q_update  # noqa: F401

__all__ = ["temporal_difference", "value_update", "ddpg_update", "td3_update", "ChooseREINFORCE", "reinforce_update",
           "bcq_update", "soft_q_update"]
