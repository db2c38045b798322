"""Host logic of the reference-shaped loop on planned batches (recnn_amd/nn/algo.py: Algo.batches / update / flush / lazy losses)
against a stand-in for the fused context: what is queued, when it is executed, which step a lazy loss resolves to."""
import pytest

from recnn_amd.nn.algo import Algo, PlannedBatch


class _Engine:
    def __init__(self):
        self.steps = 0          # engine steps executed so far
        self.reads = 0

    def counters(self):
        return (self.steps, 0, 0, 0)

    def loss_history(self, n):
        self.reads += 1
        assert 0 <= n <= 1024
        return [{"value": 100.0 + s, "policy": -float(s)} for s in range(self.steps - n, self.steps)]

    def losses(self):
        return {"value": 100.0 + self.steps - 1, "policy": -float(self.steps - 1)}


class _Ctx:
    def __init__(self):
        self.engine = _Engine()
        self.sampler = {"rows": 8, "pos": 0, "cursor": 0, "n_batches": 1000, "epoch": 0, "perms": {}, "upb": 2}
        self.modules = {}
        self.runs = []

    def ensure(self, nets, rows): pass
    def set_hyper(self, params, a, b): pass
    def apply_external(self, rows): pass
    def bump(self, opt, ni, n): pass
    def mark_stepped(self, nis): pass

    def run_steps(self, first, n, every=None, prepare=False):
        self.runs.append((first, n))
        self.engine.steps += n
        self.sampler["pos"] += n
        self.sampler["cursor"] = (self.sampler["cursor"] + n) % self.sampler["n_batches"]


def _algo():
    a = Algo()
    a.params = {"policy_step": 10}
    a.optimizers = {"policy_optimizer": None, "value_optimizer": None}
    a._fused_ctx, a._fused_keys = _Ctx(), ("policy_optimizer", "value_optimizer")
    a._fused_adam_cfgs = lambda keys: [None, None]
    return a


def test_updates_are_queued_and_replayed_sixty_at_a_time():
    a = _algo()
    got = []
    for batch in a.batches(130):
        got.append(a.update(batch, learn=True))
        a.step()
    assert a._fused_ctx.runs == [(0, 60), (60, 60)] and a._step == 130      # ten steps still queued
    assert a._fused_ctx.engine.reads == 0                                    # nothing read back yet
    assert float(got[125]["value"]) == 225.0                                 # resolves: flushes the rest, one ring read
    assert a._fused_ctx.runs[-1] == (120, 10) and a._fused_ctx.engine.reads == 1
    assert [float(l["policy"]) for l in got[:3]] == [0.0, -1.0, -2.0] and a._fused_ctx.engine.reads == 1
    assert all(l["step"] == i for i, l in enumerate(got))


def test_losses_are_banked_before_the_device_ring_wraps():
    a = _algo()
    got = []
    for batch in a.batches(2000):
        got.append(a.update(batch))
        a.step()
    a.flush()
    assert a._fused_ctx.engine.reads == 2                                    # after 960 and 1920 steps (> 900 each)
    assert float(got[0]["value"]) == 100.0 and float(got[1999]["value"]) == 2099.0 and float(got[961]["policy"]) == -961.0


def test_run_and_the_loop_share_the_step_count():
    a = _algo()
    out = a.run(25)
    assert out["step"] == 24 and a._step == 25 and a._fused_ctx.runs == [(0, 25)]
    it = a.batches()
    l = a.update(next(it)); a.step()
    assert l["step"] == 25
    out = a.run(5)                                                           # flushes the queued step first
    assert a._fused_ctx.runs == [(0, 25), (25, 1), (26, 5)] and a._step == 31
    assert float(l["value"]) == 125.0


def test_handles_must_be_used_in_order_once():
    a = _algo()
    it = a.batches()
    first = next(it)
    a.update(first); a.step()
    with pytest.raises(RuntimeError, match="out of order"):
        a.update(first)
    with pytest.raises(RuntimeError, match="out of order"):
        a.update(PlannedBatch(_algo(), a._step))                             # a handle of another Algo
    with pytest.raises(ValueError):
        a.update(next(a.batches()), learn=False)


def test_handles_carry_their_sampler_position():
    a = _algo()
    it = a.batches()
    h0 = next(it); a.update(h0); a.step()
    h1 = next(it)
    assert (h0.pos, h1.pos) == (0, 1) and a.batches_left_in_epoch() == 999          # one queued, none executed
    a.run(10)
    assert next(a.batches()).pos == 11 and a.batches_left_in_epoch() == 989
    assert set(h1.keys()) >= {"state", "action", "reward", "next_state", "done"} and "state" in h1


def test_run_banks_unread_losses_before_it_overwrites_the_ring():
    """~900 queued-and-flushed steps whose lazy losses nobody read, then run(500): the device ring holds 1024 steps, so the
    oldest unread ones must be banked BEFORE the run executes -- they are still readable afterwards."""
    a = _algo()
    lazies = []
    for batch in a.batches(840):
        lazies.append(a.update(batch, learn=True))
        a.step()
    a.flush()
    assert a._since_ring_read == 840
    a.run(500)
    assert a._fused_ctx.engine.steps == 1340
    assert float(lazies[0]["value"]) == 100.0 and float(lazies[839]["value"]) == 939.0      # steps 0 and 839, not ring garbage
    assert float(lazies[3]["policy"]) == -3.0
