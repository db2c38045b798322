"""Binding of the reference-shaped objects (nets / optimizer dicts of nn.Modules and torch optimizers) to the
fused HIP step engine.

`ddpg_update(batch, params, nets, optimizer, ...)` keeps the reference's signature and in-place semantics
(recnn/nn/update/ddpg.py:8-18): the modules in `nets` stay the owners of the weights.  On first use their
parameter tensors are *adopted*: copied into the engine's flat fp32 arenas and re-pointed (`param.data`) at
views of those arenas, so that optimizers, `state_dict()`, `torch.save` and user code keep seeing live values
while the kernels read and write the same memory.  External in-place changes (load_state_dict, a user
optimizer) are detected through the tensors' version counters and the compute-layout shadows are refreshed.
"""
from __future__ import annotations

import weakref
from typing import Dict, Optional

import numpy as np
import contextlib
import os

import torch

from .. import _lib as L
from ..optim import fused_config
from .engine import NET_NAMES_DDPG, NET_NAMES_TD3, PARAM_NAMES, StepEngine

_MODULE_PARAMS = (("linear1", "weight"), ("linear1", "bias"), ("linear2", "weight"), ("linear2", "bias"),
                  ("linear3", "weight"), ("linear3", "bias"))

# process-wide defaults for engines created by the update functions
# dtype: the update functions are the reference's drop-in API, so they default to the reference's arithmetic (fp32,
# exact-fp32 MFMA: matches the CPU reference within 1e-4).  bf16 compute (fp32 master weights and accumulation) is the
# throughput mode: opt in with set_defaults(dtype="bf16") -- bench.py and Algo.attach_env(..., dtype="bf16") do.
DEFAULTS = {"dtype": "fp32", "mask_mode": "hash", "seed": None, "min_capacity": 256}

_contexts: "weakref.WeakKeyDictionary" = weakref.WeakKeyDictionary()
_by_module: "weakref.WeakKeyDictionary" = weakref.WeakKeyDictionary()


def set_defaults(**kw):
    """dtype='bf16'|'fp32', mask_mode='hash'|'none', seed=int|None, min_capacity=rows."""
    for k, v in kw.items():
        if k not in DEFAULTS:
            raise KeyError(k)
        DEFAULTS[k] = v


def notify_params_changed(module):
    ctx = _by_module.get(module)
    if ctx is not None:
        ctx.dirty.add(id(module))
    from . import functional
    functional.mark_written(module.parameters())         # cached derived layouts of the module-level forward are stale too


def _module_params(m):
    """The six parameters of an Actor / Critic in engine order.  Read through nn.Module's own dicts: `m.linear1.weight` goes
    through two `nn.Module.__getattr__` calls (~1 us each), and every update()/run() call looks at all 24 parameters of the
    four networks twice (adoption check, version check) -- that was most of the ~65 us of Python in front of a graph launch."""
    try:
        mods = m._modules
        return [mods[a]._parameters[b] for a, b in _MODULE_PARAMS]
    except (AttributeError, KeyError):       # not a plain nn.Module layout: the generic path
        return [getattr(getattr(m, a), b) for a, b in _MODULE_PARAMS]


class FusedContext:
    def __init__(self, algo: str, nets: Dict[str, torch.nn.Module]):
        self.algo = algo
        self.names = NET_NAMES_TD3 if algo == "td3" else NET_NAMES_DDPG
        missing = [k for k in self.names if k not in nets]
        if missing:
            raise KeyError(f"{algo}_update: nets is missing {missing}")
        pol = nets["policy_net"]
        self.S = pol.linear1.in_features
        self.A = pol.linear3.out_features
        self.H = pol.linear1.out_features
        self.engine: Optional[StepEngine] = None
        self.modules: Dict[int, torch.nn.Module] = {}
        self.versions: Dict[int, list] = {}
        self.dirty = set()
        self.opt_t = {L.NET_POLICY: 0, L.NET_VALUE1: 0, L.NET_VALUE2: 0}
        self.hyper_key = None
        self.external = None        # (masks, noise) for the next update (parity runs)
        self.dtype = DEFAULTS["dtype"]
        self.mask_mode = DEFAULTS["mask_mode"]
        self.seed = DEFAULTS["seed"]
        self.sampler_env = None
        self._view_ptrs, self._view_ptrs_engine = {}, None

    # ------------------------------------------------------------------ engine lifetime
    def _check_modules(self, nets):
        for name, ni in self.names.items():
            m = nets[name]
            p = m.linear1.weight
            if not p.is_cuda:
                raise L.RecnnHipError(
                    f"{self.algo}_update: network '{name}' lives on {p.device}; recnn_amd runs on the GPU only "
                    "(no CPU fallback) -- call algo.to(torch.device('cuda')) / net.to('cuda') first")
            want_in = self.S + self.A if ni >= L.NET_VALUE1 else self.S
            if m.linear1.in_features != want_in or m.linear1.out_features != self.H or m.linear2.in_features != self.H:
                raise L.RecnnHipError(f"network '{name}' does not have the Actor/Critic shape the engine was built for")
        # Dropout as the engine implements it: Dropout(0.5) active on the learning nets, none on the targets -- what the
        # reference's Algo classes set up (models.py:60,205: p=0.5; algo.py:76-77: targets .eval()).  Anything else is
        # refused instead of silently computing something different from what the modules say.
        learners = [nets[n] for n, ni in self.names.items() if ni in (L.NET_POLICY, L.NET_VALUE1, L.NET_VALUE2)]
        targets = [nets[n] for n, ni in self.names.items() if ni not in (L.NET_POLICY, L.NET_VALUE1, L.NET_VALUE2)]
        train = {bool(m.training) for m in learners}
        if len(train) != 1 or any(t.training for t in targets):
            raise L.RecnnHipError(f"{self.algo}_update: unsupported train/eval mix -- the learning nets must share one mode "
                                  "and the target nets must be in eval mode (as recnn.nn.DDPG / TD3 construct them)")
        self.nets_train = train.pop()
        for m in learners:
            dl = getattr(m, "drop_layer", None)
            p_drop = float(getattr(dl, "p", 0.5)) if dl is not None else 0.0
            if self.nets_train and self.mask_mode == "hash" and p_drop != 0.5:
                raise L.RecnnHipError(f"{self.algo}_update: drop_layer.p = {p_drop}: the fused step implements Dropout(p=0.5) "
                                      "(the reference's value) or none (put the nets in eval mode / set_defaults(mask_mode='none'))")

    def ensure(self, nets, rows: int):
        # the module checks (devices, shapes, train / eval mix, dropout p) walk ~40 nn.Module attributes: repeat them only when
        # something they look at may have changed -- another module object, a mode flip, another dropout p, a parameter that
        # moved (the adoption test below sees that), another mask mode
        key = tuple((id(nets[n]), nets[n].training, getattr(nets[n]._modules.get("drop_layer"), "p", None)) for n in self.names)
        key = key + (self.mask_mode,)
        stale = [ni for name, ni in self.names.items()
                 if self.engine is None or self.modules.get(ni) is not nets[name] or not self._is_adopted(ni, nets[name])]
        if stale or key != getattr(self, "_checked_key", None):
            self._checked_key = None
            self._check_modules(nets)
            self._checked_key = key
        dev = _module_params(nets["policy_net"])[0].device
        if self.engine is None or rows > self.engine.max_rows or self.engine.device != dev:
            cap = max(DEFAULTS["min_capacity"], 1 << (max(rows, 1) - 1).bit_length())
            old = self.engine
            seed = self.seed if self.seed is not None else (torch.initial_seed() & 0x7FFFFFFF)
            eng = StepEngine(self.algo, self.S, self.A, self.H, cap, dtype=self.dtype,
                             mask_mode="hash" if self.mask_mode == "hash" else "none", seed=seed, device=dev)
            if old is not None and old.device == dev:       # carry optimizer state over to the bigger engine
                for ni in eng.adam_m:
                    eng.adam_m[ni].copy_(old.adam_m[ni])
                    eng.adam_v[ni].copy_(old.adam_v[ni])
            self.engine = eng
            self.modules = {}
            self.hyper_key = None
            eng.set_counters(self.opt_t[L.NET_POLICY], self.opt_t[L.NET_VALUE1], self.opt_t[L.NET_VALUE2], 0)
            stale = list(self.names.values())
        for name, ni in self.names.items():
            if ni in stale:
                self._adopt(ni, nets[name])
        self._sync_versions()

    def _is_adopted(self, ni, m):
        ptrs = self._view_ptrs.get(ni)
        if ptrs is None or self._view_ptrs_engine is not self.engine:
            if self._view_ptrs_engine is not self.engine:
                self._view_ptrs, self._view_ptrs_engine = {}, self.engine
            views = self.engine.param_views(ni)
            ptrs = self._view_ptrs[ni] = [views[k].data_ptr() for k in PARAM_NAMES]
        return all(p.data_ptr() == q for p, q in zip(_module_params(m), ptrs))

    def _adopt(self, ni, m):
        eng = self.engine
        views = eng.param_views(ni)
        gviews = eng.param_views(ni, eng.grads[ni]) if ni in eng.grads else None
        with torch.no_grad():
            for p, k in zip(_module_params(m), PARAM_NAMES):
                views[k].copy_(p.data.to(torch.float32))
                p.data = views[k]
                if gviews is not None:
                    p.grad = gviews[k]
        self.modules[ni] = m
        _by_module[m] = self
        self.versions[ni] = [p._version for p in _module_params(m)]
        eng.refresh(ni)

    def _sync_versions(self):
        """Refresh the shadows of every net whose parameters were modified behind the engine's back."""
        for ni, m in self.modules.items():
            cur = [p._version for p in _module_params(m)]
            if cur != self.versions.get(ni) or id(m) in self.dirty:
                self.engine.refresh(ni)
                self.versions[ni] = cur
                self.dirty.discard(id(m))

    def mark_stepped(self, nis):
        """The engine just wrote the parameters of networks `nis` from its kernels (optimizer step / soft update): tell the
        module-level forward path, whose derived weight layouts (`functional._derived_of`) are keyed on version counters the
        kernels never touch (a bf16 `policy_net(state)` after fused training used the pre-training weights)."""
        from . import functional
        for ni in nis:
            m = self.modules.get(ni)
            if m is not None:
                functional.mark_written(_module_params(m))

    def attach_grads(self, ni):
        m = self.modules[ni]
        gviews = self.engine.param_views(ni, self.engine.grads[ni])
        for p, k in zip(_module_params(m), PARAM_NAMES):
            if p.grad is None or p.grad.data_ptr() != gviews[k].data_ptr():
                p.grad = gviews[k]

    # ------------------------------------------------------------------ per-step inputs
    def load_batch(self, batch) -> int:
        eng = self.engine
        packed = getattr(batch, "packed", None)
        if packed is not None and packed[5] == eng.ld_x and packed[0].device == eng.device \
                and batch["state"].data_ptr() == packed[0].data_ptr() + 4 * self.A:
            xs, xn, reward, done, rows, _ = packed
            eng.bind_batch(xs, xn, reward, done)
            self._bound_external_batch = True
            self.graph_rows = None          # binding a batch drops the engine's graphs
            return rows
        self._own_batch()
        for k in ("state", "action", "reward", "next_state", "done"):
            if k not in batch:
                raise KeyError(f"batch has no '{k}'")
        return eng.pack_batch(batch["state"], batch["action"], batch["reward"], batch["next_state"], batch["done"])

    def _own_batch(self):
        """Back to the engine's own packed-row buffers after a FrameEnv batch was bound in place."""
        if getattr(self, "_bound_external_batch", None):
            self.engine.bind_batch()
            self._bound_external_batch = None
            self.graph_rows = None

    def set_hyper(self, algo_params: dict, pol_cfg: Optional[dict], val_cfg: Optional[dict]):
        P = algo_params
        if self.algo == "ddpg":
            key = ("ddpg", P["gamma"], P["min_value"], P["max_value"], P["soft_tau"], P["policy_step"])
            kw = dict(gamma=P["gamma"], min_value=P["min_value"], max_value=P["max_value"], soft_tau=P["soft_tau"],
                      policy_every=P["policy_step"])
        else:
            key = ("td3", P["gamma"], P["noise_std"], P["noise_clip"], P["soft_tau"], P["policy_update"])
            kw = dict(gamma=P["gamma"], soft_tau=P["soft_tau"], policy_every=P["policy_update"], noise_std=P["noise_std"],
                      noise_clip=P["noise_clip"])
        key = key + (tuple(sorted((pol_cfg or {}).items())), tuple(sorted((val_cfg or {}).items())))
        if key != self.hyper_key:
            self.engine.set_hyper(policy_opt=pol_cfg, value_opt=val_cfg, **kw)
            self.hyper_key = key
            self.graph_rows = None          # the engine dropped its graphs (they freeze the hyper-parameters)

    def apply_external(self, rows):
        """Parity runs: dropout masks / TD3 noise supplied by the caller for exactly one update."""
        eng = self.engine
        if self.external is None:
            want = L.MASK_HASH if (self.mask_mode == "hash" and getattr(self, "nets_train", True)) else L.MASK_NONE
            if eng.mask_mode != want:
                L.call("recnn_engine_set_mask_mode", eng.handle, want)
                eng.mask_mode = want
                self.graph_rows = None      # the engine dropped its graphs
            return
        masks, noise = self.external
        self.external = None
        if masks is not None:
            if eng.ext_masks is None:
                eng.ext_masks = torch.ones(eng.n_masks, eng.max_rows, eng.H, dtype=torch.uint8, device=eng.device)
                eng._bind_external()
            if eng.mask_mode != L.MASK_EXTERNAL:
                L.call("recnn_engine_set_mask_mode", eng.handle, L.MASK_EXTERNAL)
                eng.mask_mode = L.MASK_EXTERNAL
                self.graph_rows = None
        eng.set_external(masks=masks, noise=noise)

    # ------------------------------------------------------------------ fused training loop on a device-resident env
    def attach_sampler(self, env, rows: int, users_per_batch: int = None, shard=(0, 1)):
        """Bind `env`'s replay store to the engine: every step then builds its own `rows`-row batch on the GPU from an
        epoch permutation of the env's TRAIN users.
        users_per_batch = None (default): DENSE epochs -- the windows of the shuffled users are concatenated and cut into
        `rows`-row batches (a user's windows continue in the next batch; the epoch's leftover is carried into the next epoch), so
        one epoch visits every (user, window) exactly once, like the reference's whole-user batches (recnn/data/utils.py:161-187).
        users_per_batch = k: every batch draws k users and keeps the first `rows` rows of their windows (rounds 1-3)."""
        eng = self.engine
        st = env.store
        train = np.asarray(st.slots(env.base.train_user_dataset.users), dtype=np.int64)
        train = train[shard[0]::shard[1]]          # data parallel: rank r of W owns every W-th train user
        lens = st.lengths[train] - env.frame_size
        if users_per_batch is None and os.environ.get("RECNN_SAMPLER_DENSE", "1") != "0":
            wins = np.maximum(lens, 0).astype(np.int64)
            total = int(wins.sum())
            if total < rows:
                raise ValueError(f"attach_env: the train users hold {total} windows, fewer than one batch of {rows} rows")
            if total + rows >= 2 ** 31:
                raise ValueError("attach_env: more than 2^31 windows per epoch (shard the users over more ranks)")
            nb_max = (total + rows) // rows + 1
            self.sampler = dict(env=env, rows=rows, upb=0, dense=True, train_host=train.astype(np.int32), win_of=None,
                                n_batches=0, cursor=0, pos=0, epoch=-1, perms={}, seqs={}, epoch_pos={}, carry=(np.zeros(0, np.int32), 0),
                                lengths=st.lengths, frame=env.frame_size)
            self.perm = None
            eng.bind_sampler_dense(st.items, st.ratings, st.user_off, rows, env.frame_size, self.A, env.table, 2 * len(train) + 1, nb_max)
            self._reshuffle()
            self.graph_rows = None
            return
        if users_per_batch is None:       # enough users that even the shortest histories fill `rows` rows
            k = np.sort(lens)
            users_per_batch = int(np.searchsorted(np.cumsum(k), rows) + 1)
        users_per_batch = min(users_per_batch, len(train))
        self.sampler = dict(env=env, rows=rows, upb=users_per_batch, train=torch.from_numpy(train).to(eng.device),
                            train_host=train, n_batches=len(train) // users_per_batch, cursor=0, pos=0, epoch=-1, perms={})
        self.perm = torch.empty(self.sampler["n_batches"] * users_per_batch, dtype=torch.int32, device=eng.device)
        self._reshuffle()
        eng.bind_sampler(st.items, st.ratings, st.user_off, self.perm, users_per_batch, env.frame_size, self.A, env.table,
                         plan_rows=rows if os.environ.get("RECNN_SAMPLER_PLAN", "1") != "0" else 0)
        self.graph_rows = None

    def _reshuffle_dense(self):
        """Next dense epoch: [the previous epoch's leftover] + a fresh permutation of the train users; plan table on the device."""
        sm = self.sampler
        order = torch.randperm(len(sm["train_host"])).numpy()                                      # CPU generator, as RandomSampler
        seq, skip0, n_e, sm["carry"] = dense_epoch(sm["carry"], sm["train_host"][order], sm["lengths"], sm["frame"], sm["rows"])
        sm["epoch"] += 1
        sm["n_batches"] = n_e
        sm["seqs"][sm["epoch"]] = (seq, skip0, n_e)
        sm["epoch_pos"][sm["epoch"]] = sm["pos"]
        for old in (sm["epoch"] - 3,):
            sm["seqs"].pop(old, None)
            sm["epoch_pos"].pop(old, None)
        self.engine.cursor.zero_()                     # the device cursor counts batches of the CURRENT epoch (stream-ordered)
        self.engine.plan_dense(seq, skip0)

    def _reshuffle(self):
        sm = self.sampler
        if sm.get("dense"):
            return self._reshuffle_dense()
        order = torch.randperm(sm["train"].numel())[: self.perm.numel()]                            # CPU generator, as RandomSampler
        # the host keeps the last epochs' permutations (store slots): a planned batch can be materialised for inspection
        sm["epoch"] += 1
        sm["perms"][sm["epoch"]] = sm["train_host"][order.numpy()]
        sm["perms"].pop(sm["epoch"] - 3, None)
        order = order.to(sm["train"].device)
        self.perm.copy_(sm["train"][order].to(torch.int32))
        if getattr(self.engine, "has_sampler", False):
            self.engine.plan_sampler()          # the plan table follows the permutation (same stream as the copy)

    def run_steps(self, first_step: int, n_steps: int, every: int = None, prepare: bool = False):
        """`n_steps` consecutive learn steps by hipGraph replay; the permutation is redrawn at every epoch boundary.
        `every` (the policy-update period) lets the context recognise repeated requests: the second time the same
        (first_step mod every, n_steps <= 64) comes in, a run graph is made to order for it and such requests are one graph
        launch from then on.  prepare=True only builds what the request needs (graphs, the made-to-order graph) and runs nothing."""
        eng, sm = self.engine, self.sampler
        # hipGraph capture/replay needs a real (non-null) stream.  A caller that already is on one (bench.py, training loops
        # that own a stream) gets its graphs launched right there; from the default stream the work hops to a private stream,
        # ordered after and before the caller's (two event waits and a stream switch: ~20 us of host time per call)
        cur = torch.cuda.current_stream(eng.device)
        direct = cur.cuda_stream != 0
        if not direct:
            if getattr(self, "_side", None) is None:
                self._side = torch.cuda.Stream(device=eng.device)
            self._side.wait_stream(cur)
        self._own_batch()                    # the sampler writes the engine's own rows
        with (contextlib.nullcontext() if direct else torch.cuda.stream(self._side)):
            if getattr(self, "graph_rows", None) != sm["rows"]:
                eng.graph_build(sm["rows"])
                self.graph_rows = sm["rows"]
                self._run_seen = {}
            if every and 2 <= n_steps <= 64:
                key = (first_step % every, n_steps)
                seen = self.__dict__.setdefault("_run_seen", {}).get(key, 0)
                if seen == 1 or (prepare and seen < 2):
                    eng.graph_prepare(first_step, n_steps)
                    seen = 2
                self._run_seen[key] = max(seen, 1)
            if prepare:
                n_steps = 0
            done = 0
            while done < n_steps:
                chunk = min(n_steps - done, sm["n_batches"] - sm["cursor"])
                eng.graph_run(first_step + done, chunk)
                done += chunk
                sm["cursor"] += chunk
                sm["pos"] += chunk
                if sm["cursor"] >= sm["n_batches"]:      # epoch finished (the device cursor wrapped to 0 by itself)
                    sm["cursor"] = 0
                    self._reshuffle()
        if not direct:
            cur.wait_stream(self._side)

    # ------------------------------------------------------------------ optimizer state mirrors
    def mirror_optimizer_state(self, opt, ni):
        """Expose the engine's optimizer state (Adam / RAdam moments, Lookahead slow weights) through `opt.state`
        (state_dict compatibility); state that existed before the adoption is carried over."""
        eng = self.engine
        mv = eng.param_views(ni, eng.adam_m[ni])
        vv = eng.param_views(ni, eng.adam_v[ni])
        cfg = fused_config(opt)
        ranger = cfg is not None and cfg["kind"] == "ranger"
        sv = None
        if ranger:
            fresh = ni not in eng.slow
            sv = eng.param_views(ni, eng.ensure_slow(ni))
        t = self.opt_t[ni]
        is_torch = type(opt) is torch.optim.Adam
        for p, k in zip(_module_params(self.modules[ni]), PARAM_NAMES):
            st = opt.state[p]
            if st.get("exp_avg") is not None and st["exp_avg"].data_ptr() != mv[k].data_ptr():
                mv[k].copy_(st["exp_avg"])            # state that existed before adoption
                vv[k].copy_(st["exp_avg_sq"])
                t = max(t, int(st.get("step", 0)))
            st["exp_avg"], st["exp_avg_sq"] = mv[k], vv[k]
            if ranger:
                if st.get("slow_buffer") is not None and st["slow_buffer"].data_ptr() != sv[k].data_ptr():
                    sv[k].copy_(st["slow_buffer"])
                elif st.get("slow_buffer") is None and (fresh or self.opt_t[ni] == 0):
                    sv[k].copy_(p.data)               # Lookahead starts from the parameters as they are at the first step
                st["slow_buffer"] = sv[k]
            st["step"] = torch.tensor(float(t)) if is_torch else t
        if t != self.opt_t[ni]:
            self.opt_t[ni] = t
            eng.set_counters(self.opt_t[L.NET_POLICY], self.opt_t[L.NET_VALUE1], self.opt_t[L.NET_VALUE2], 0)

    def bump(self, opt, ni, inc: int = 1):
        self.opt_t[ni] += inc
        t = self.opt_t[ni]
        is_torch = type(opt) is torch.optim.Adam
        for p in _module_params(self.modules[ni]):
            st = opt.state[p]
            if "exp_avg" in st:
                if is_torch and isinstance(st.get("step"), torch.Tensor):
                    st["step"].fill_(float(t))
                else:
                    st["step"] = torch.tensor(float(t)) if is_torch else t


def dense_epoch(carry, perm_slots, lengths, frame: int, rows: int):
    """Host bookkeeping of one DENSE epoch (no device work): the epoch's user sequence is the previous epoch's leftover -- `carry` =
    (slots, windows of slots[0] already consumed) -- followed by the new permutation; its windows, concatenated, give
    n_e = total // rows whole batches; what is left (the tail of the sequence from the user that holds global row n_e * rows) is
    the next carry.  Returns (seq int32, skip0, n_e, new_carry)."""
    carry_slots, skip0 = carry
    seq = np.concatenate([np.asarray(carry_slots, dtype=np.int32), np.asarray(perm_slots, dtype=np.int32)])
    wins = np.maximum(np.asarray(lengths)[seq].astype(np.int64) - frame, 0)
    wins[0] = max(int(wins[0]) - int(skip0), 0)
    cum = np.cumsum(wins)
    total = int(cum[-1])
    n_e = total // rows
    end = n_e * rows
    if end == total:
        new_carry = (np.zeros(0, np.int32), 0)
    else:
        idx = int(np.searchsorted(cum, end, side="right"))            # the entry that holds global row `end`
        before = int(cum[idx - 1]) if idx > 0 else 0
        new_carry = (seq[idx:].copy(), (end - before) + (int(skip0) if idx == 0 else 0))
    return seq, int(skip0), n_e, new_carry


def context_for(algo: str, nets) -> FusedContext:
    anchor = nets["policy_net"]
    ctx = _contexts.get(anchor)
    if ctx is None or ctx.algo != algo:
        ctx = FusedContext(algo, nets)
        _contexts[anchor] = ctx
    return ctx


class external_randomness:
    """Context manager for parity tests: the next update of `algo_or_nets` uses these dropout keep-masks
    (uint8 [B, H] each, reference consumption order) and, for TD3, this unclipped noise draw."""

    def __init__(self, nets, masks=None, noise=None, algo="ddpg"):
        self.ctx = context_for(algo, nets)
        self.payload = (masks, noise)

    def __enter__(self):
        self.ctx.external = self.payload
        return self

    def __exit__(self, *exc):
        self.ctx.external = None
        return False


def fused_adam_configs(optimizer: dict, keys):
    """Per-optimizer configurations when ALL of them run inside the engine (Adam or Ranger arithmetic), else None."""
    cfgs = [fused_config(optimizer.get(k)) for k in keys]
    return cfgs if all(c is not None for c in cfgs) else None
