"""CPU checks of the native boundary: the shared library loads without a GPU, exports every symbol that
include/recnn_hip.h declares, the ctypes struct declarations match sizeof() on the C side, and the product
path refuses to run without a GPU (no fallback)."""
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "recnn_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(recnn_[a-z0-9_]+)\s*\(", src)))


def test_library_loads_and_exports_every_declared_symbol():
    from recnn_amd import _lib as L
    lib = L.load()
    assert lib.recnn_abi_version() == 2
    names = _declared()
    assert len(names) >= 35
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/recnn_hip.h but not exported"
    for n in names:
        assert n in L.SIGNATURES, f"{n} has no ctypes signature in recnn_amd/_lib.py"


def test_bad_arguments_return_error_codes_not_crashes():
    from recnn_amd import _lib as L
    lib = L.load()
    assert lib.recnn_engine_query(None, None) == -1
    assert b"null" in lib.recnn_last_error()
    cfg = L.EngineConfig(0, 0, 1290, 126, 256, 2048, 1, 0, 0)      # action_dim not a multiple of 8
    sz = L.EngineSizes()
    import ctypes as C
    assert lib.recnn_engine_query(C.byref(cfg), C.byref(sz)) == -1
    cfg.action_dim = 128
    assert lib.recnn_engine_query(C.byref(cfg), C.byref(sz)) == 0
    assert sz.ld_x == 1536 and sz.master_floats_actor == 429_184 and sz.master_floats_critic == 429_313
    assert sz.workspace_bytes > 0 and sz.workspace_bytes % 256 == 0


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-GPU behaviour")
def test_no_gpu_means_loud_failure_not_fallback():
    from recnn_amd import _lib as L
    from recnn_amd.nn.engine import StepEngine
    with pytest.raises(L.RecnnHipError):
        StepEngine("ddpg", 1290, 128, 256, 32, device=torch.device("cpu"))
    import recnn_amd
    with pytest.raises(L.RecnnHipError):
        recnn_amd.utils.soft_update(recnn_amd.nn.Actor(27, 8, 16), recnn_amd.nn.Actor(27, 8, 16), 1.0)


def test_product_never_imports_the_oracle():
    bad = []
    for dp, _, files in os.walk(os.path.join(ROOT, "recnn_amd")):
        for f in files:
            if f.endswith(".py"):
                txt = open(os.path.join(dp, f)).read()
                if re.search(r"^\s*(from|import)\s+oracle\b", txt, flags=re.M):
                    bad.append(os.path.join(dp, f))
    assert not bad, bad


def test_import_recnn_resolves_to_this_implementation():
    """With the repository on sys.path, `import recnn` (what the reference's notebooks do) IS recnn_amd -- no install_as call,
    no changed line; sub-module imports and from-imports work."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys; sys.path.insert(0, %r)\n"
            "import recnn\n"
            "import recnn.nn, recnn.data.env, recnn.utils\n"
            "from recnn.nn import DDPG, TD3, Actor, Critic, ddpg_update\n"
            "from recnn.data.env import FrameEnv, DataPath\n"
            "import recnn_amd\n"
            "assert recnn is recnn_amd and recnn.nn.DDPG is recnn_amd.nn.DDPG and FrameEnv is recnn_amd.data.env.FrameEnv\n"
            "assert recnn.data.get_base_batch is recnn_amd.data.get_base_batch and recnn.utils.soft_update is recnn_amd.utils.soft_update\n"
            "print('ok')\n") % root
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and out.stdout.strip().endswith("ok"), out.stderr[-2000:]


def test_csr_workspace_query_is_host_only_and_validates():
    """recnn_csr_workspace_bytes is pure host arithmetic (no GPU needed): grows with the row count, refuses row counts whose
    indices would not fit the 32-bit payload of the radix sort; recnn_frame_plan_rows refuses null pointers with an error code
    and a message instead of launching."""
    import ctypes as C
    from recnn_amd import _lib as L
    lib = L.load()
    a, b = C.c_int64(0), C.c_int64(0)
    assert lib.recnn_csr_workspace_bytes(1000, C.byref(a)) == 0 and lib.recnn_csr_workspace_bytes(20_000_263, C.byref(b)) == 0
    assert 0 < a.value < b.value and b.value >= 20_000_263 * (8 + 8 + 4 + 4 + 4)
    assert lib.recnn_csr_workspace_bytes(1 << 33, C.byref(a)) != 0
    assert b"rows" in lib.recnn_last_error()
    assert lib.recnn_frame_plan_rows(None, None, 4, 1, 10, 8, None, None) != 0
    assert b"frame_plan_rows" in lib.recnn_last_error()


def test_new_rows_fail_loudly_without_a_gpu():
    """No CPU fallback anywhere: the device ETL, the VAE kernels, the BCQ modules and the candidate scoring raise RecnnHipError on
    CPU inputs instead of computing something else."""
    import numpy as np
    import recnn_amd
    from recnn_amd import _lib as L
    from recnn_amd.data import dataset_functions as F
    from recnn_amd.nn import functional as F_hip
    if torch.cuda.is_available():
        pytest.skip("CPU-only behaviour")
    with pytest.raises(L.RecnnHipError):
        F.csr_from_ratings_device(np.arange(5), np.arange(5), np.ones(5), np.arange(5))
    with pytest.raises(L.RecnnHipError):
        F_hip.vae_latent(torch.zeros(2, 4), torch.zeros(2, 2))
    with pytest.raises(L.RecnnHipError):
        F_hip.vae_loss(torch.zeros(2, 3), torch.zeros(2, 3), torch.zeros(2, 2), torch.ones(2, 2))
    gen = recnn_amd.nn.bcqGenerator(6, 2, 3)
    with pytest.raises(L.RecnnHipError):
        gen(torch.zeros(4, 6), torch.zeros(4, 2))
    with pytest.raises(L.RecnnHipError):
        recnn_amd.nn.bcqPerturbator(6, 2, 8)(torch.zeros(4, 6), torch.zeros(4, 2))
    with pytest.raises(L.RecnnHipError):
        recnn_amd.nn.Critic(6, 2, 8).candidates(torch.zeros(4, 6), torch.zeros(8, 2), 2)
