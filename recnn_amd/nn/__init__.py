from . import algo, models, update, engine, fused, graphed  # noqa: F401
from .models import *  # noqa: F401,F403
from .algo import *  # noqa: F401,F403
from .update import *  # noqa: F401,F403
from .graphed import GraphedUpdate  # noqa: F401
