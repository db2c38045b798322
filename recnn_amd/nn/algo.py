"""Algo / DDPG / TD3 facades (reference: recnn/nn/algo.py:15-179).

Same public attributes (nets, optimizers, params, _step, debug, writer, device, loss_layout, algorithm) and
methods (update, to, step).  Targets are deep copies put in eval mode and hard-synced (soft_tau = 1.0) exactly
as the reference does; the sync itself happens when the networks reach the GPU if they were built on the CPU.

Default optimizers: like the reference (`torch_optimizer.Ranger(lr=1e-5, weight_decay=1e-2)`, algo.py:84-89,139-147)
the facades build `recnn_amd.optim.Ranger(lr=1e-5, weight_decay=1e-2)` -- RAdam + Lookahead, executed by the fused HIP
optimizer pass of the step engine.  `torch_optimizer` is a third-party package the reference neither vendors nor pins, so
that arithmetic is restated from its published algorithm and flagged "parity unpinned" (recnn_amd/optim.py).
The optimizer of record for parity and for bench.py is Adam (north_star; the substitution the reference's own docs
show): `algo.optimizers[...] = recnn_amd.optim.Adam(...)` / `torch.optim.Adam(...)`, or
`recnn_amd.nn.algo.set_default_optimizer("adam")` before constructing the facade.  Any other torch optimizer can be
assigned too (it is then stepped by torch between the engine's gradient phases).
"""
import copy

import torch

from .. import optim, utils
from . import update

__all__ = ["Algo", "DDPG", "TD3", "Reinforce", "set_default_optimizer"]

_DEFAULT_OPTIMIZER = "ranger"


def set_default_optimizer(kind: str):
    """'ranger' (the reference's default shape) or 'adam' (the optimizer of record for parity runs)."""
    global _DEFAULT_OPTIMIZER
    if kind not in ("ranger", "adam"):
        raise ValueError(kind)
    _DEFAULT_OPTIMIZER = kind


def _default_opt(params, lr=1e-5, weight_decay=1e-2):
    cls = optim.Ranger if _DEFAULT_OPTIMIZER == "ranger" else optim.Adam
    return cls(params, lr=lr, weight_decay=weight_decay)


def _hard_sync(net, target):
    """soft_update with tau = 1.0 (algo.py:80-81).  CPU-resident freshly built nets are synced by value copy:
    tau = 1 makes the lerp an exact copy, no arithmetic is involved."""
    if next(net.parameters()).is_cuda:
        utils.soft_update(net, target, soft_tau=1.0)
    else:
        with torch.no_grad():
            for tp, p in zip(target.parameters(), net.parameters()):
                tp.data.copy_(p.data)


class PlannedBatch:
    """Handle of the fixed-size batch the engine's sampler draws for update step `index` (Algo.batches): `algo.update(batch)`
    queues it without ever building it on the host side; reading it like the reference's batch dict (`batch["state"]`,
    `.keys()`, `data.get_base_batch(batch)`) materialises the same rows through the gather kernel, once."""
    __slots__ = ("algo", "index", "pos", "_rows")
    KEYS = ("state", "action", "reward", "next_state", "done", "meta")

    def __init__(self, algo, index, pos=None):
        self.algo, self.index, self.pos, self._rows = algo, index, pos, None

    def _materialise(self):
        if self._rows is None:
            if self.pos is None:
                raise TypeError("this planned batch carries no sampler position: it cannot be materialised")
            self._rows = self.algo._planned_rows(self.pos)
        return self._rows

    def __getitem__(self, key):
        return self._materialise()[key]

    def keys(self):
        return self.KEYS

    def __contains__(self, key):
        return key in self.KEYS

    def __iter__(self):
        return iter(self.KEYS)

    def items(self):
        return self._materialise().items()

    def __repr__(self):
        return f"PlannedBatch(step={self.index})"


class LazyScalar:
    """A loss value of a queued / replayed step: behaves like the float it becomes when first looked at."""
    __slots__ = ("_src", "_key", "_val")

    def __init__(self, src, key):
        self._src, self._key, self._val = src, key, None

    def item(self):
        if self._val is None:
            self._val = float(self._src._values()[self._key])
        return self._val

    __float__ = item

    def __format__(self, spec):
        return format(self.item(), spec)

    def __repr__(self):
        return repr(self.item())

    def __bool__(self):
        return bool(self.item())

    def _bin(op):
        return lambda self, other: op(self.item(), float(other))

    def _rbin(op):
        return lambda self, other: op(float(other), self.item())

    import operator as _o
    __add__, __sub__, __mul__, __truediv__ = _bin(_o.add), _bin(_o.sub), _bin(_o.mul), _bin(_o.truediv)
    __radd__, __rsub__, __rmul__, __rtruediv__ = _rbin(_o.add), _rbin(_o.sub), _rbin(_o.mul), _rbin(_o.truediv)
    __lt__, __le__, __gt__, __ge__, __eq__, __ne__ = _bin(_o.lt), _bin(_o.le), _bin(_o.gt), _bin(_o.ge), _bin(_o.eq), _bin(_o.ne)
    __neg__ = lambda self: -self.item()
    __abs__ = lambda self: abs(self.item())
    __hash__ = object.__hash__
    del _bin, _rbin, _o


class LazyLosses(dict):
    """The dict `update()` returns ({'value': .., 'policy': .., 'step': ..}, the reference's losses) for a queued step: the
    values are LazyScalars, resolved -- queue flushed, the device's loss history read once -- when one is looked at."""

    def __init__(self, algo, step, keys):
        super().__init__({k: LazyScalar(self, k) for k in keys})
        self["step"] = step
        self._algo, self._step = algo, step

    def _values(self):
        return self._algo._loss_of(self._step)


class Algo:
    def __init__(self):
        self.nets = {"value_net": None, "policy_net": None}
        self.optimizers = {"policy_optimizer": None, "value_optimizer": None}
        self.params = {"Some parameters here": None}
        self._step = 0
        self.debug = {}
        self.writer = utils.misc.DummyWriter()
        self.device = torch.device("cpu")
        self.loss_layout = {"test": {"value": [], "policy": [], "step": []},
                            "train": {"value": [], "policy": [], "step": []}}
        self.algorithm = None

    def update(self, batch, learn=True):
        if isinstance(batch, PlannedBatch):
            return self._update_planned(batch, learn)
        if getattr(self, "_queue", (0, 0))[1] or getattr(self, "_since_ring_read", 0):
            self.flush()
            self._bank_losses()     # (step-by-step updates that follow move the device's loss history on)
        return self.algorithm(batch, self.params, self.nets, self.optimizers, device=self.device, debug=self.debug,
                              writer=self.writer, learn=learn, step=self._step)

    def to(self, device):
        self.nets = {k: v.to(device) for k, v in self.nets.items()}
        self.device = device
        return self

    def step(self):
        self._step += 1

    # ---- extension: fused training on a device-resident FrameEnv (no reference counterpart) --------------------
    def attach_env(self, env, rows_per_batch: int, users_per_batch: int = None, shard=(0, 1), dtype: str = None,
                   drive_loader: bool = False):
        """Let the engine sample its own batches from `env`'s TRAIN users: `run(n)` then executes n update steps
        (sampler, gather, update, step()) as hipGraph replays with no Python or host work per step.  Equivalent to
        `for batch in env.train_dataloader: self.update(batch); self.step()` with fixed `rows_per_batch`-row batches.
        Needs optimizers the engine runs itself (recnn_amd.optim.Adam / Ranger, torch.optim.Adam) in `self.optimizers`.
        `shard=(rank, world)` restricts the sampler to this data-parallel rank's share of the train users.
        `dtype`: 'fp32' | 'bf16' compute type of the engine (default: fused.DEFAULTS['dtype']); only honoured before the
        networks' first update.

        Batches are FIXED-SIZE (`rows_per_batch` rows; the run graphs are captured for one size).  `users_per_batch=None` (default):
        DENSE epochs -- the windows of the epoch's shuffled users are concatenated in that order and cut into `rows_per_batch`-row
        batches; a user's windows continue in the next batch, `done` marks each user's last window, and what an epoch leaves over
        (< one batch) opens the next one: one epoch visits every (user, window) exactly once, like the reference's whole-user
        batches (`recnn/data/utils.py:161-187`), and `env.collate_rows(...)` materialises any of these batches.
        `users_per_batch=k`: every batch draws k users and keeps the first `rows_per_batch` rows of their windows (rounds 1-3;
        long histories are then mostly not visited)."""
        from . import fused
        algo = "td3" if "value_net1" in self.nets else "ddpg"
        keys = ("policy_optimizer", "value_optimizer1", "value_optimizer2") if algo == "td3" else ("policy_optimizer", "value_optimizer")
        cfgs = self._fused_adam_cfgs(keys)
        ctx = fused.context_for(algo, self.nets)
        if dtype is not None and ctx.engine is None:
            ctx.dtype = dtype
        ctx.ensure(self.nets, rows_per_batch)
        ctx.set_hyper(self.params, cfgs[0], cfgs[1])
        ctx.apply_external(rows_per_batch)
        for k, ni in zip(keys, (fused.L.NET_POLICY, fused.L.NET_VALUE1, fused.L.NET_VALUE2)):
            ctx.mirror_optimizer_state(self.optimizers[k], ni)
        ctx.attach_sampler(env, rows_per_batch, users_per_batch, shard)
        self._fused_ctx, self._fused_keys = ctx, keys
        if drive_loader:
            # `for batch in env.train_dataloader: algo.update(batch); algo.step()` -- the reference's loop, verbatim -- then
            # iterates handles of the engine's own batches (FrameLoader.__iter__) and runs at the speed of `run`
            env.train_dataloader.planner = self
        return self

    def _fused_adam_cfgs(self, keys):
        from ..optim import fused_config
        cfgs = [fused_config(self.optimizers[k]) for k in keys]
        if any(c is None for c in cfgs) or (len(cfgs) == 3 and cfgs[1] != cfgs[2]):
            raise ValueError("attach_env / run need optimizers the engine runs itself: recnn_amd.optim.Adam / Ranger or "
                             "torch.optim.Adam (both critics of TD3 with equal settings)")
        return cfgs

    def prepare_run(self, n_steps: int, first_step: int = None):
        """Build, ahead of time, everything `run(n_steps)` called at step number `first_step` (default: the current one)
        replays: the graph family and -- for n_steps <= 64 -- a run graph made to order, so that the call is ONE graph launch.
        (`run` does this by itself the second time it sees the same request shape.)"""
        ctx = getattr(self, "_fused_ctx", None)
        if ctx is None:
            raise RuntimeError("call attach_env(env, rows_per_batch) first")
        every = self.params["policy_update" if "value_net1" in self.nets else "policy_step"]
        cfgs = self._fused_adam_cfgs(self._fused_keys)
        ctx.ensure(self.nets, ctx.sampler["rows"])
        ctx.set_hyper(self.params, cfgs[0], cfgs[1])
        ctx.apply_external(ctx.sampler["rows"])
        ctx.run_steps(self._step if first_step is None else first_step, n_steps, every=every, prepare=True)

    # ---- the reference's loop shape on the fused path -------------------------------------------------------------
    def batches(self, n: int = None):
        """`for batch in algo.batches(): loss = algo.update(batch); algo.step()` -- the reference's training loop
        (examples: `for batch in tqdm(env.train_dataloader): loss = ddpg.update(batch, learn=True); ddpg.step()`) at the speed
        of `run`: the handles yielded here stand for the fixed-size batches the engine's sampler draws (see attach_env),
        `update` only QUEUES the step and returns lazy losses, and the queue is executed as run graphs of up to 60 steps --
        when it is full, when a loss value is read (float(), format, comparison ...) or at `flush()`.  Handles must be consumed
        in order, one `update` + `step` each."""
        if getattr(self, "_fused_ctx", None) is None:
            raise RuntimeError("call attach_env(env, rows_per_batch) first")
        i = 0
        while n is None or i < n:
            sm = self._fused_ctx.sampler
            yield PlannedBatch(self, self._step, pos=sm["pos"] + getattr(self, "_queue", (0, 0))[1])
            i += 1

    def batches_left_in_epoch(self) -> int:
        """Planned batches until the engine's sampler finishes its current epoch permutation (queued steps counted)."""
        sm = self._fused_ctx.sampler
        left = sm["n_batches"] - (sm["cursor"] + getattr(self, "_queue", (0, 0))[1]) % sm["n_batches"]
        return left

    def _planned_rows(self, pos):
        """The batch at sampler position `pos` as the reference's dict (for inspection): every earlier step is executed first,
        so that the epoch permutation the position falls into has been drawn."""
        self.flush()
        sm = self._fused_ctx.sampler
        if sm.get("dense"):
            ep = max((e for e, p0 in sm["epoch_pos"].items() if p0 <= pos), default=None)
            if ep is None:
                raise KeyError("the user sequence of that epoch is no longer kept (planned batches can be read up to two epochs back)")
            seq, skip0, n_e = sm["seqs"][ep]
            idx = pos - sm["epoch_pos"][ep]
            if idx >= n_e:
                raise RuntimeError("this batch belongs to an epoch whose permutation is drawn when the previous epoch's last step runs")
            return sm["env"].collate_rows(seq, skip0, idx * sm["rows"], sm["rows"])
        epoch, idx = divmod(pos, sm["n_batches"])
        if pos > sm["pos"] and epoch > sm["epoch"]:
            raise RuntimeError("this batch belongs to an epoch whose permutation is drawn when the previous epoch's last step runs")
        perm = sm["perms"].get(epoch)
        if perm is None:
            raise KeyError("the permutation of that epoch is no longer kept (planned batches can be read up to two epochs back)")
        slots = perm[idx * sm["upb"]:(idx + 1) * sm["upb"]]
        return sm["env"].collate_slots(slots, rows_per_batch=sm["rows"])

    def _update_planned(self, batch, learn):
        if not learn:
            raise ValueError("planned batches are training steps (learn=True); evaluate test batches with env.collate_users / the test loader")
        first, queued = getattr(self, "_queue", (self._step, 0))
        if queued == 0:
            first = self._step
        if batch.algo is not self or batch.index != self._step or self._step != first + queued:
            self.flush()
            raise RuntimeError(f"planned batch {batch.index} used out of order (the next step is {self._step}): each handle of "
                               "algo.batches() takes exactly one update() followed by one step()")
        self._queue = (first, queued + 1)
        lazy = LazyLosses(self, self._step, ("value1", "value2", "policy") if "value_net1" in self.nets else ("value", "policy"))
        if queued + 1 >= self.queue_limit:
            self.flush()
        return lazy

    queue_limit = 60      # steps per flush: whole policy cycles, one run graph

    def flush(self):
        """Execute the queued update steps now (asynchronously: graph launches, no host sync)."""
        first, queued = getattr(self, "_queue", (0, 0))
        if queued:
            self._queue = (first + queued, 0)
            self._execute(first, queued)
            done = self.__dict__.setdefault("_since_ring_read", 0) + queued
            self._since_ring_read = done
            if done > 900:          # the device keeps the last 1024 steps' losses: bank them before they are overwritten
                self._bank_losses()

    def _bank_losses(self):
        eng = self._fused_ctx.engine
        end = eng.counters()[0]
        n = min(1024, getattr(self, "_since_ring_read", 0), end)
        bank = self.__dict__.setdefault("_loss_bank", {})
        last = getattr(self, "_last_exec_step", end - 1)      # Algo step number of the engine's most recent step
        for i, h in enumerate(eng.loss_history(n)):
            bank[last - (n - 1 - i)] = h
        self._since_ring_read = 0
        while len(bank) > 65536:     # bounded: the oldest banked steps go first
            bank.pop(next(iter(bank)))

    def _loss_of(self, step):
        bank = self.__dict__.setdefault("_loss_bank", {})
        if step not in bank:
            self.flush()
            self._bank_losses()
        if step not in bank:
            raise KeyError(f"the losses of step {step} are no longer kept (read them within 65536 steps)")
        return bank[step]

    def _execute(self, first: int, n_steps: int):
        ctx = self._fused_ctx
        every = self.params["policy_update" if "value_net1" in self.nets else "policy_step"]
        # hyper-parameters and optimizer settings are frozen into the graphs: re-read them (lr schedules, edits of
        # self.params) -- a change rebuilds the graphs
        cfgs = self._fused_adam_cfgs(self._fused_keys)
        ctx.ensure(self.nets, ctx.sampler["rows"])
        ctx.set_hyper(self.params, cfgs[0], cfgs[1])
        ctx.apply_external(ctx.sampler["rows"])
        ctx.run_steps(first, n_steps, every=every)
        self._last_exec_step = first + n_steps - 1
        n_policy = len(range(first + (-first) % every, first + n_steps, every))
        from . import fused
        for k, ni in zip(self._fused_keys, (fused.L.NET_POLICY, fused.L.NET_VALUE1, fused.L.NET_VALUE2)):
            ctx.bump(self.optimizers[k], ni, n_policy if ni == fused.L.NET_POLICY else n_steps)
        ctx.mark_stepped(list(ctx.modules))     # the graphs wrote every network's parameters in place

    def run(self, n_steps: int, history: bool = False):
        """n_steps fused update steps (see attach_env); returns the losses of the last one (one device sync).
        history=True: returns (last losses, [losses of each of the n_steps steps]) -- what the reference's loop would
        have collected from `update()` step by step (kept on the device, up to 1024 steps back)."""
        ctx = getattr(self, "_fused_ctx", None)
        if ctx is None:
            raise RuntimeError("call attach_env(env, rows_per_batch) first")
        self.flush()
        # the device keeps the losses of the last 1024 steps: steps queued-and-flushed earlier whose lazy losses nobody read yet
        # are banked BEFORE this call's steps overwrite them (~900 unread steps + run(500) lost the oldest).  (Of a single
        # run(n) with n > 1024 the device keeps the last 1024 steps' losses; `history=True` returns those.)
        pending = getattr(self, "_since_ring_read", 0)
        if pending and pending + n_steps > 900:
            self._bank_losses()
        self._execute(self._step, n_steps)
        self._step += n_steps
        self._since_ring_read = getattr(self, "_since_ring_read", 0) + n_steps
        losses = ctx.engine.losses()
        losses["step"] = self._step - 1
        if history:
            hist = ctx.engine.loss_history(min(n_steps, 1024))
            for i, h in enumerate(hist):
                h["step"] = self._step - len(hist) + i
            return losses, hist
        return losses


class DDPG(Algo):
    def __init__(self, policy_net, value_net):
        super().__init__()
        self.algorithm = update.ddpg_update
        target_policy_net = copy.deepcopy(policy_net)
        target_value_net = copy.deepcopy(value_net)
        target_policy_net.eval()
        target_value_net.eval()
        _hard_sync(value_net, target_value_net)
        _hard_sync(policy_net, target_policy_net)
        value_optimizer = _default_opt(value_net.parameters())
        policy_optimizer = _default_opt(policy_net.parameters())
        self.nets = {"value_net": value_net, "target_value_net": target_value_net, "policy_net": policy_net,
                     "target_policy_net": target_policy_net}
        self.optimizers = {"policy_optimizer": policy_optimizer, "value_optimizer": value_optimizer}
        self.params = {"gamma": 0.99, "min_value": -10, "max_value": 10, "policy_step": 10, "soft_tau": 0.001}
        self.loss_layout = {"test": {"value": [], "policy": [], "step": []},
                            "train": {"value": [], "policy": [], "step": []}}


class TD3(Algo):
    def __init__(self, policy_net, value_net1, value_net2):
        super().__init__()
        self.algorithm = update.td3_update
        target_policy_net = copy.deepcopy(policy_net)
        target_value_net1 = copy.deepcopy(value_net1)
        target_value_net2 = copy.deepcopy(value_net2)
        for t in (target_policy_net, target_value_net1, target_value_net2):
            t.eval()
        _hard_sync(value_net1, target_value_net1)
        _hard_sync(value_net2, target_value_net2)
        _hard_sync(policy_net, target_policy_net)
        value_optimizer1 = _default_opt(value_net1.parameters())
        value_optimizer2 = _default_opt(value_net2.parameters())
        policy_optimizer = _default_opt(policy_net.parameters())
        self.nets = {"value_net1": value_net1, "target_value_net1": target_value_net1, "value_net2": value_net2,
                     "target_value_net2": target_value_net2, "policy_net": policy_net,
                     "target_policy_net": target_policy_net}
        self.optimizers = {"policy_optimizer": policy_optimizer, "value_optimizer1": value_optimizer1,
                           "value_optimizer2": value_optimizer2}
        self.params = {"gamma": 0.99, "noise_std": 0.5, "noise_clip": 3, "soft_tau": 0.001, "policy_update": 10,
                       "policy_lr": 1e-5, "value_lr": 1e-5, "actor_weight_init": 25e-2, "critic_weight_init": 6e-1}
        self.loss_layout = {"test": {"value1": [], "value2": [], "policy": [], "step": []},
                            "train": {"value1": [], "value2": [], "policy": [], "step": []}}


class Reinforce(Algo):
    """REINFORCE with a learned critic as reward model (algo.py:182-233): `policy_net` a DiscreteActor over the catalogue,
    `value_net` a Critic over [state | action distribution].  `params["reinforce"]` selects the estimator
    (`ChooseREINFORCE.basic_reinforce` / `reinforce_with_correction` / `reinforce_with_TopK_correction`), `params["K"]`
    the top-K size of the latter."""

    def __init__(self, policy_net, value_net):
        super().__init__()
        self.algorithm = update.reinforce_update
        target_policy_net = copy.deepcopy(policy_net)
        target_value_net = copy.deepcopy(value_net)
        target_policy_net.eval()
        target_value_net.eval()
        _hard_sync(value_net, target_value_net)
        _hard_sync(policy_net, target_policy_net)
        value_optimizer = _default_opt(value_net.parameters())
        policy_optimizer = _default_opt(policy_net.parameters())
        self.nets = {"value_net": value_net, "target_value_net": target_value_net, "policy_net": policy_net,
                     "target_policy_net": target_policy_net}
        self.optimizers = {"policy_optimizer": policy_optimizer, "value_optimizer": value_optimizer}
        self.params = {"reinforce": update.ChooseREINFORCE(update.ChooseREINFORCE.basic_reinforce), "K": 10, "gamma": 0.99,
                       "min_value": -10, "max_value": 10, "policy_step": 10, "soft_tau": 0.001}
        self.loss_layout = {"test": {"value": [], "policy": [], "step": []},
                            "train": {"value": [], "policy": [], "step": []}}
