"""recnn_amd -- MI355X-native implementation of RecNN's DDPG/TD3 inner training step.

Same import surface as the reference package for this path (`recnn/__init__.py:1-2`):
`recnn_amd.nn.{Actor, Critic, Algo, DDPG, TD3, ddpg_update, td3_update, ...}`,
`recnn_amd.data.env.{FrameEnv, DataPath, ...}`, `recnn_amd.data.get_base_batch`,
`recnn_amd.utils.{soft_update, DummyWriter, write_losses, Plotter}`.

All arithmetic of the path runs in `recnn_amd/csrc/librecnn_hip.so` (hand-written HIP for gfx950)
through the C ABI of `include/recnn_hip.h`; torch is used for device memory, streams and
`torch.distributed` only.  There is no CPU or eager-PyTorch fallback: without the library, or on a
CPU device, the update functions raise.

`import recnn_amd; recnn_amd.install_as("recnn")` registers the package under the reference's
name so that existing notebooks (`import recnn`) run unchanged.
"""
import sys as _sys

from . import _lib  # noqa: F401
from . import data, utils, nn  # noqa: F401
from .data import pd  # noqa: F401
from . import optim, parallel  # noqa: F401

__all__ = ["data", "utils", "nn", "optim", "parallel", "pd", "install_as"]


def install_as(name: str = "recnn") -> None:
    """Alias this package (and its sub-modules) as `name` in sys.modules: `import recnn` then resolves here."""
    prefix = __name__ + "."
    for mod_name, mod in list(_sys.modules.items()):
        if mod_name == __name__:
            _sys.modules[name] = mod
        elif mod_name.startswith(prefix):
            _sys.modules[name + "." + mod_name[len(prefix):]] = mod
