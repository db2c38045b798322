from . import utils, env, store  # noqa: F401
from .utils import *  # noqa: F401,F403
from .env import *  # noqa: F401,F403
from .dataset_functions import *  # noqa: F401,F403
from .pandas_backend import pd  # noqa: F401
