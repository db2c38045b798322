"""Import-time stand-in for the third-party ``torch_optimizer`` package.

TEST INFRASTRUCTURE ONLY.  The reference (`/root/reference/recnn/nn/algo.py:6`) does
``import torch_optimizer as optim`` and builds ``optim.Ranger`` optimizers
(`algo.py:84-89,139-147`); the package is not installed in this image and cannot be
installed (no network).  This stub only lets ``import recnn`` of the *reference* succeed
inside ``oracle/ref_loader.py``; the arithmetic of the real Ranger/RAdam is NOT
reproduced here.  Every golden vector is generated with ``torch.optim.Adam`` injected
through the reference's public ``algo.optimizers[...]`` dict, so these classes never
contribute numbers to a fixture ("parity unpinned" at the Ranger boundary).
"""
import torch


class Ranger(torch.optim.Adam):
    """Placeholder so that `optim.Ranger(params, lr=..., weight_decay=...)` constructs."""


class RAdam(torch.optim.RAdam):
    """Placeholder for `optim.RAdam` used by `.circleci/tests/learning.py:51-52`."""
