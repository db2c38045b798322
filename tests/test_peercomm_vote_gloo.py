"""World-size-2 gloo test (CPU) of PeerComm.create's agreement protocol (recnn_amd/parallel.py).

`create` must return None on EVERY rank when ANY rank fails, wherever that rank fails -- before the handle exchange (create /
export), after it (connect), in the self-test -- and every rank must run exactly ONE exchange gather and ONE vote gather: a rank
that gathers a third time pairs with nobody and sits there until the process-group timeout.  The three library steps of the
constructor are replaced by stand-ins (no GPU here); each scenario is followed by an all_reduce that only completes if the ranks
are still in step, under a 20 s process-group timeout so that a mismatch fails instead of hanging the suite.
"""
import datetime
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

SCENARIOS = ("healthy", "open_fails", "export_fails", "connect_fails", "self_test_fails", "python_error_in_connect")


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=20))
    from recnn_amd import _lib as L
    from recnn_amd.parallel import PeerComm

    out = {}
    for sc in SCENARIOS:
        bad = rank == 1            # rank 1 is the one that fails

        class Fake(PeerComm):
            gathers = 0

            def _open(self):
                self.lib = None
                if bad and sc == "open_fails":
                    raise L.RecnnHipError("no fine-grained memory")
                self.handle = None

            def _export(self):
                if bad and sc == "export_fails":
                    raise L.RecnnHipError("hipIpcGetMemHandle: invalid argument")
                return bytes([self.rank]) * 8

            def _connect(self, every):
                assert [h[0] for h in every] == list(range(self.world))
                if bad and sc == "connect_fails":
                    raise L.RecnnHipError("hipIpcOpenMemHandle failed")
                if bad and sc == "python_error_in_connect":
                    raise ValueError("not a library failure")

            def self_test(self):
                return not (bad and sc == "self_test_fails")

            def close(self):
                self.handle = None

        got, raised = None, None
        try:
            got = Fake.create(1024)
        except ValueError as ex:
            raised = str(ex)
        # still in step?  (a rank with a surplus or missing gather would pair this all_reduce with a gather and time out / throw)
        t = torch.tensor([float(rank + 1)])
        dist.all_reduce(t)
        assert float(t) == 3.0
        out[sc] = (got is not None, raised)
    q.put((rank, out))
    dist.destroy_process_group()


def test_create_agrees_wherever_a_rank_fails():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=150) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for sc in SCENARIOS:
        for rank in (0, 1):
            have, raised = res[rank][sc]
            assert have == (sc == "healthy"), (sc, rank, res)
            # a non-library exception is re-raised on the rank it happened on -- AFTER the agreement
            assert (raised is not None) == (sc == "python_error_in_connect" and rank == 1), (sc, rank, res)
