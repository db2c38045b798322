from . import misc, plot  # noqa: F401
from .misc import *  # noqa: F401,F403
from .plot import *  # noqa: F401,F403
