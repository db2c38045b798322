"""recnn_dp_allreduce_flat (csrc/comm.hip) and the data-parallel steps built on it, with 2 and 3 ranks sharing this GPU: hipIpc
maps a buffer of the same device like a peer's, so flags, epochs, peer reads / writes and graph replay are all exercised."""
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _launch(world, tmp_path, *args, fused="0"):
    port = 29850 + (os.getpid() % 100)
    # ranks sharing ONE GPU spin on each other inside their collective launches: all of them must be resident at once, so each
    # takes 32 workgroups here (128 by default: with 3 ranks the third found no free CU until a time slice ended, measured)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""),
               RECNN_COMM_WORKGROUPS="32", RECNN_COMM_FUSED=fused)
    # (RECNN_COMM_FUSED=0: with the exchange inside the critics' optimizer launch every one of its 420 workgroups waits for its
    # counterpart on the other rank -- on separate GPUs they all run at once, on a shared one the second rank's launch finds no
    # free CU until a time slice ends.  The fused form is covered at world 1 below, against both the single-GPU step and the
    # unfused collective.)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "tests", "comm2_worker.py"), str(tmp_path), *args]
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + "\n" + r.stderr[-3000:]
    return json.load(open(os.path.join(tmp_path, "comm2.json")))


@pytest.mark.parametrize("world", [2, 3])
def test_peer_allreduce_matches_gloo(cuda, world, tmp_path):
    """54 collectives of 1 .. 500,000 floats (odd sizes, tails) + 75 replayed from a captured graph; bit-equal to gloo at world
    2 (asserted in the worker), every rank bit-identical at any world, fp32 summation-order distance at world 3."""
    js = _launch(world, tmp_path, "allreduce")["allreduce"]
    assert js["world"] == world and js["collectives"] == 54
    assert js["worst_rel"] <= (0.0 if world == 2 else 1e-6)


@pytest.mark.parametrize("dtype,mode", [("fp32", "graphs"), ("bf16", "graphs"), ("bf16", "eager")])
def test_device_collective_steps_equal_host_collective_steps(cuda, dtype, mode, tmp_path):
    """DataParallelStepper(comm=PeerComm): 8 DDPG steps (3 policy steps) of 2 x 1024 rows with the collectives as launches
    inside the step / run graph == the same stepper on dist.all_reduce between phase graphs: bit for bit."""
    js = _launch(2, tmp_path, "stepper", dtype, mode)["stepper"]
    assert js["replica_gap"] == 0.0
    assert js["host_losses"] == js["dev_losses"], (js["host_losses"], js["dev_losses"])
    assert all(v == 0.0 for v in js["param_diff"].values()), js["param_diff"]


def test_fused_in_optimizer_exchange_with_two_ranks(cuda, tmp_path):
    """The critics' gradient exchange INSIDE their optimizer launch (optim.hip exchange_grads: workgroup b publishes its elements'
    slab sums, meets workgroup b of the other rank, reduces its share, reads the sums back) with TWO ranks -- so far it had only
    ever run at world 1.  Networks of 34 / 16 / 32 units make that launch a few dozen workgroups, so both ranks' launches are
    resident on the one GPU at once.  Against the same steps on dist.all_reduce between phases: bit for bit."""
    js = _launch(2, tmp_path, "stepper_small", "fp32", "graphs", fused="1")["stepper"]
    assert js["comm_fused"] == 1
    assert js["replica_gap"] == 0.0
    assert js["host_losses"] == js["dev_losses"], (js["host_losses"], js["dev_losses"])
    assert all(v == 0.0 for v in js["param_diff"].values()), js["param_diff"]


def test_world_one_comm_and_bounded_wait(cuda):
    """World 1: the collective is the identity (through all three phases) and an attached engine steps as before."""
    from recnn_amd.parallel import PeerComm
    comm = PeerComm(10_000)
    x = torch.randn(9_999, device="cuda")
    want = x.clone()
    for _ in range(5):
        comm.all_reduce(x)
    torch.cuda.synchronize()
    assert torch.equal(x, want)
    comm.check()
    with pytest.raises(Exception):
        comm.all_reduce(torch.zeros(10_065, device="cuda"))       # does not fit the communicator
    comm.close()


def _ddpg_engine(rows, seed=3):
    from recnn_amd import _lib as L
    from recnn_amd.nn.engine import StepEngine
    from tests.dp2_worker import init_nets
    S, A, H = 1290, 128, 256
    actor, critic = init_nets(0, S, A, H)
    eng = StepEngine("ddpg", S, A, H, rows, dtype="bf16", mask_mode="hash", seed=seed, device=torch.device("cuda"))
    for ni, p in ((L.NET_POLICY, actor), (L.NET_TARGET_POLICY, actor), (L.NET_VALUE1, critic), (L.NET_TARGET_VALUE1, critic)):
        eng.load_params(ni, p)
    eng.set_hyper(policy_opt=dict(lr=1e-4, weight_decay=1e-2), value_opt=dict(lr=1e-4, weight_decay=1e-2), policy_every=3)
    eng.set_counters()
    gen = torch.Generator().manual_seed(5)
    eng.pack_batch(torch.randn(rows, S, generator=gen), torch.randn(rows, A, generator=gen), torch.randn(rows, generator=gen) * 3,
                   torch.randn(rows, S, generator=gen), (torch.rand(rows, generator=gen) < 0.1).float())
    return eng


@pytest.mark.parametrize("fused", [1, 0])
def test_world_one_collective_steps_equal_single_gpu_steps(cuda, fused):
    """An engine with a world-1 communicator attached -- exchange inside the critic's optimizer launch (fused) or as launches of
    its own -- against the plain single-GPU engine: 14 steps (eager + run graphs), parameters and Adam state bit for bit.
    (World 1 runs every phase of the protocol against itself: publish, flags, reduce-scatter share, read-back.)"""
    from recnn_amd import _lib as L
    from recnn_amd.parallel import PeerComm
    rows = 1024
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        ref = _ddpg_engine(rows)
        ref.graph_build(rows)
        for t in range(4):
            ref.step(rows, True, t)
        ref.graph_run(4, 10)
        try:
            eng = _ddpg_engine(rows)
            eng.set_tuning(comm_fused=fused)
            comm = PeerComm(PeerComm.floats_for(eng))
            eng.set_comm(comm)
            eng.graph_build(rows)
            for t in range(4):
                eng.step(rows, True, t)
            eng.graph_run(4, 10)
            side.synchronize()
            comm.check()
        finally:
            pass
    side.synchronize()
    assert ref.counters() == eng.counters()
    for ni in (L.NET_POLICY, L.NET_VALUE1, L.NET_TARGET_POLICY, L.NET_TARGET_VALUE1):
        assert torch.equal(ref.params[ni], eng.params[ni]), ni
    for ni in (L.NET_POLICY, L.NET_VALUE1):
        assert torch.equal(ref.adam_m[ni], eng.adam_m[ni]) and torch.equal(ref.adam_v[ni], eng.adam_v[ni]), ni
        assert torch.equal(ref.grads[ni], eng.grads[ni]), ni      # the bound arena holds the (summed) gradient either way
    eng.set_comm(None)
    comm.close()
