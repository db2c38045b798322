from . import algo, models, update, engine, fused  # noqa: F401
from .models import *  # noqa: F401,F403
from .algo import *  # noqa: F401,F403
from .update import *  # noqa: F401,F403
