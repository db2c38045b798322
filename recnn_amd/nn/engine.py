"""StepEngine: Python owner of the device buffers behind the fused HIP DDPG/TD3 step.

Host plumbing only (torch allocates device memory, ctypes calls the C ABI); all arithmetic of
the step runs in librecnn_hip.so.  Mirrors the data the reference's update functions touch
(recnn/nn/update/ddpg.py:8-104, td3.py:8-150): four (DDPG) or six (TD3) networks, their
optimizer state, one batch.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Optional, Sequence

import torch

from .. import _lib as L

PARAM_NAMES = ("w1", "b1", "w2", "b2", "w3", "b3")
NET_NAMES_DDPG = {"policy_net": L.NET_POLICY, "target_policy_net": L.NET_TARGET_POLICY,
                  "value_net": L.NET_VALUE1, "target_value_net": L.NET_TARGET_VALUE1}
NET_NAMES_TD3 = {"policy_net": L.NET_POLICY, "target_policy_net": L.NET_TARGET_POLICY,
                 "value_net1": L.NET_VALUE1, "target_value_net1": L.NET_TARGET_VALUE1,
                 "value_net2": L.NET_VALUE2, "target_value_net2": L.NET_TARGET_VALUE2}
LEARNING = (L.NET_POLICY, L.NET_VALUE1, L.NET_VALUE2)


def _require_gpu(device: torch.device):
    if device.type != "cuda" or not torch.cuda.is_available():
        raise L.RecnnHipError(
            "recnn_amd runs on an AMD GPU through librecnn_hip.so only; got device "
            f"{device} (torch.cuda.is_available()={torch.cuda.is_available()}). There is no CPU fallback.")


class StepEngine:
    def __init__(self, algo: str, state_dim: int, action_dim: int, hidden: int, max_rows: int,
                 dtype: str = "fp32", mask_mode: str = "hash", seed: int = 0,
                 device: Optional[torch.device] = None):
        device = torch.device("cuda") if device is None else torch.device(device)
        _require_gpu(device)
        self.lib = L.load()
        self.device = device
        self.algo = algo
        self.td3 = algo == "td3"
        self.S, self.A, self.H = state_dim, action_dim, hidden
        self.max_rows = max_rows
        self.dtype = dtype
        self.mask_mode = {"none": L.MASK_NONE, "hash": L.MASK_HASH, "external": L.MASK_EXTERNAL}[mask_mode]
        cfg = L.EngineConfig(L.ALGO_TD3 if self.td3 else L.ALGO_DDPG, L.DTYPES[dtype],
                             state_dim, action_dim, hidden, max_rows, self.mask_mode, seed & 0xFFFFFFFF,
                             device.index or 0)
        self.cfg = cfg
        sz = L.EngineSizes()
        L.call("recnn_engine_query", C.byref(cfg), C.byref(sz))
        self.sizes = sz
        self._view_cache = {}
        self.ld_x = int(sz.ld_x)
        with torch.cuda.device(device):
            self.workspace = torch.zeros(int(sz.workspace_bytes), dtype=torch.uint8, device=device)
            self.xs = torch.zeros(int(sz.x_rows), self.ld_x, dtype=torch.float32, device=device)
            self.xn = torch.zeros(int(sz.x_rows), self.ld_x, dtype=torch.float32, device=device)
            self.reward = torch.zeros(max_rows, dtype=torch.float32, device=device)
            self.done = torch.zeros(max_rows, dtype=torch.float32, device=device)
            h = C.c_void_p()
            L.call("recnn_engine_create", C.byref(cfg), L.ptr(self.workspace), C.byref(h))
            self.handle = h
            self.set_tuning()                # library defaults, the environment, _tune.set_default_tuning(...)
            self.nets = sorted((NET_NAMES_TD3 if self.td3 else NET_NAMES_DDPG).values())
            self.params: Dict[int, torch.Tensor] = {}
            self.grads: Dict[int, torch.Tensor] = {}
            self.adam_m: Dict[int, torch.Tensor] = {}
            self.adam_v: Dict[int, torch.Tensor] = {}
            self.slow: Dict[int, torch.Tensor] = {}
            for ni in self.nets:
                n = int(sz.master_floats_critic if ni >= L.NET_VALUE1 else sz.master_floats_actor)
                self.params[ni] = torch.zeros(n, dtype=torch.float32, device=device)
                if ni in LEARNING:
                    self.grads[ni] = torch.zeros(n, dtype=torch.float32, device=device)
                    self.adam_m[ni] = torch.zeros(n, dtype=torch.float32, device=device)
                    self.adam_v[ni] = torch.zeros(n, dtype=torch.float32, device=device)
                L.call("recnn_engine_bind_net", h, ni, L.ptr(self.params[ni]), L.ptr(self.grads.get(ni)),
                       L.ptr(self.adam_m.get(ni)), L.ptr(self.adam_v.get(ni)))
            L.call("recnn_engine_bind_batch", h, L.ptr(self.xs), L.ptr(self.xn), L.ptr(self.reward), L.ptr(self.done))
            self.has_sampler = False
            self.n_masks = 8 if self.td3 else 6
            self.ext_masks = None
            self.ext_noise = None
            if self.mask_mode == L.MASK_EXTERNAL:
                self.ext_masks = torch.ones(self.n_masks, max_rows, hidden, dtype=torch.uint8, device=device)
            self._bind_external()
        self._losses_host = (C.c_float * 4)()
        self._view_cache = {}

    # ------------------------------------------------------------------ plumbing
    def _bind_external(self):
        L.call("recnn_engine_bind_external", self.handle, L.ptr(self.ext_masks), L.ptr(self.ext_noise))

    def __del__(self):
        try:
            if getattr(self, "handle", None):
                self.lib.recnn_engine_destroy(self.handle)
                self.handle = None
        except Exception:
            pass

    def _stream(self):
        return L.current_stream()

    def set_tuning(self, **fields):
        """Schedule / tile choices of THIS engine (`recnn_engine_tuning`, all compute the same numbers); drops built graphs."""
        from .._tune import make_tuning
        cur = {f: getattr(self.tuning, f) for f in L.TUNING_FIELDS} if fields and getattr(self, "tuning", None) is not None else {}
        self.tuning = make_tuning(**{**cur, **fields}) if cur else make_tuning(**fields)
        L.call("recnn_engine_set_tuning", self.handle, C.byref(self.tuning))

    def in_dim(self, ni: int) -> int:
        return self.S + self.A if ni >= L.NET_VALUE1 else self.S

    def out_dim(self, ni: int) -> int:
        return 1 if ni >= L.NET_VALUE1 else self.A

    def param_views(self, ni: int, arena: Optional[torch.Tensor] = None) -> Dict[str, torch.Tensor]:
        """Views [out,in]/[out] into a flat canonical arena laid out [w1|b1|w2|b2|w3|b3]."""
        a = self.params[ni] if arena is None else arena
        key = (ni, a.data_ptr())
        hit = self._view_cache.get(key)
        if hit is not None:
            return hit
        H, i, o = self.H, self.in_dim(ni), self.out_dim(ni)
        shapes = [(H, i), (H,), (H, H), (H,), (o, H), (o,)]
        out, off = {}, 0
        for name, shp in zip(PARAM_NAMES, shapes):
            n = 1
            for d in shp:
                n *= d
            out[name] = a[off:off + n].view(*shp)
            off += n
        self._view_cache[key] = out
        return out

    def load_params(self, ni: int, p: Dict[str, torch.Tensor]):
        views = self.param_views(ni)
        with torch.no_grad():
            for k in PARAM_NAMES:
                views[k].copy_(p[k].to(self.device, torch.float32))
        self.refresh(ni)

    def refresh(self, ni: int):
        L.call("recnn_engine_refresh", self.handle, ni, self._stream())

    def set_hyper(self, gamma=0.99, min_value=-10.0, max_value=10.0, soft_tau=0.001, policy_every=10,
                  noise_std=0.5, noise_clip=3.0, policy_opt=None, value_opt=None):
        def opt(d):
            d = dict(d or {})
            if d.get("kind", "adam") == "ranger":       # torch_optimizer.Ranger's defaults (recnn_amd/optim.py)
                base = dict(kind="ranger", lr=1e-3, beta1=0.95, beta2=0.999, eps=1e-5, weight_decay=0.0, alpha=0.5, k=6,
                            nsma_threshold=5.0)
            else:
                base = dict(kind="adam", lr=1e-3, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=0.0, alpha=0.0, k=0,
                            nsma_threshold=5.0)
            return {**base, **d}
        po, vo = opt(policy_opt), opt(value_opt)
        h = L.Hyper()
        h.gamma, h.min_value, h.max_value = gamma, min_value, max_value
        h.soft_tau, h.policy_every = soft_tau, int(policy_every)
        h.noise_std, h.noise_clip = noise_std, noise_clip
        for i, o in enumerate((po, vo)):
            h.lr[i], h.beta1[i], h.beta2[i], h.eps[i], h.weight_decay[i] = o["lr"], o["beta1"], o["beta2"], o["eps"], o["weight_decay"]
            h.opt_kind[i] = L.OPT_RANGER if o["kind"] == "ranger" else L.OPT_ADAM
            h.la_alpha[i], h.la_k[i], h.nsma_threshold[i] = o["alpha"], int(o["k"]), o["nsma_threshold"]
        nets = ((L.NET_POLICY,), self.value_nets())
        for i, o in enumerate((po, vo)):
            if o["kind"] == "ranger":
                for ni in nets[i]:
                    self.ensure_slow(ni)
        self.hyper = h
        L.call("recnn_engine_set_hyper", self.handle, C.byref(h))

    def ensure_slow(self, ni: int) -> torch.Tensor:
        """Lookahead slow-weight arena of a learning network (Ranger), created as a copy of the current parameters --
        what torch_optimizer.Ranger does when it first sees a parameter."""
        if ni not in self.slow:
            self.slow[ni] = self.params[ni].detach().clone()
            L.call("recnn_engine_bind_slow", self.handle, ni, L.ptr(self.slow[ni]))
        return self.slow[ni]

    def set_counters(self, policy_t=0, value1_t=0, value2_t=0, step=0):
        L.call("recnn_engine_set_counters", self.handle, policy_t, value1_t, value2_t, step)

    # ------------------------------------------------------------------ batch
    def pack_batch(self, state, action, reward, next_state, done) -> int:
        """Copy a canonical (reference-layout) batch into the packed rows."""
        rows = state.shape[0]
        if rows > self.max_rows:
            raise ValueError(f"batch has {rows} rows, engine capacity is {self.max_rows}")
        f = lambda t: t.to(self.device, torch.float32)
        state, action, next_state = f(state), f(action), f(next_state)
        for t in (state, action, next_state):
            assert t.stride(-1) == 1
        L.call("recnn_pack_batch", L.ptr(state), state.stride(0), L.ptr(action), action.stride(0),
               L.ptr(next_state), next_state.stride(0), rows, self.S, self.A,
               L.ptr(self.xs), L.ptr(self.xn), self.ld_x, self._stream())
        self.reward[:rows].copy_(f(reward).reshape(-1))
        self.done[:rows].copy_(f(done).reshape(-1))
        return rows

    def bind_batch(self, xs=None, xn=None, reward=None, done=None):
        """Bind packed rows owned by someone else (a FrameEnv batch); no arguments = back to the engine's own."""
        if xs is None:
            xs, xn, reward, done = self.xs, self.xn, self.reward, self.done
        self._bound = (xs, xn, reward, done)          # keep alive while kernels may still read them
        L.call("recnn_engine_bind_batch", self.handle, L.ptr(xs), L.ptr(xn), L.ptr(reward), L.ptr(done))

    def set_external(self, masks: Optional[Sequence[torch.Tensor]] = None, noise: Optional[torch.Tensor] = None):
        if masks is not None:
            assert self.ext_masks is not None, "engine was not created with mask_mode='external'"
            for i, m in enumerate(masks):
                self.ext_masks[i, :m.shape[0]].copy_(m.to(self.device, torch.uint8))
        if noise is not None:
            if self.ext_noise is None:
                self.ext_noise = torch.zeros(self.max_rows, self.A, dtype=torch.float32, device=self.device)
                self._bind_external()
            self.ext_noise[:noise.shape[0]].copy_(noise.to(self.device, torch.float32))

    def bind_sampler(self, items, ratings, user_off, perm, users_per_batch: int, frame: int, emb_dim: int, table,
                     plan_rows: int = 0):
        """Attach a device-resident replay store: every step then builds its own batch on the GPU.
        plan_rows > 0 also makes the per-epoch plan table (`recnn_frame_plan_rows`: one int64 per batch row) that lets the
        gather reach a row's window with one load; whoever rewrites `perm` afterwards must call `plan_sampler()`."""
        dev = self.device
        assert items.dtype == torch.int32 and ratings.dtype == torch.float32 and user_off.dtype == torch.int64
        assert perm.dtype == torch.int32 and table.dtype == torch.float32
        n_batches = perm.numel() // users_per_batch
        assert n_batches >= 1, "permutation shorter than one batch of users"
        self._row_off = torch.zeros(users_per_batch + 1, dtype=torch.int32, device=dev)
        self.cursor = torch.zeros(1, dtype=torch.int32, device=dev)
        self._smp_keep = (items, ratings, user_off, perm, table)
        self._plan = None
        self._plan_args = None
        if plan_rows > 0 and users_per_batch <= 4096:
            self._plan = torch.empty(n_batches * plan_rows, dtype=torch.int64, device=dev)
            self._plan_args = (user_off, perm, users_per_batch, n_batches, frame, plan_rows)
            self.plan_sampler()
        m = L.Sampler(items.data_ptr(), ratings.data_ptr(), user_off.data_ptr(), perm.data_ptr(), users_per_batch,
                      n_batches, frame, emb_dim, table.data_ptr(), self._row_off.data_ptr(), self.cursor.data_ptr(),
                      None if self._plan is None else self._plan.data_ptr(), plan_rows if self._plan is not None else 0)
        L.call("recnn_engine_bind_sampler", self.handle, C.byref(m))
        self.n_batches = n_batches
        self.has_sampler = True

    def bind_sampler_dense(self, items, ratings, user_off, rows: int, frame: int, emb_dim: int, table, n_seq_max: int, n_batches_max: int):
        """Attach a device-resident replay store in DENSE mode: the batches of an epoch are consecutive `rows`-row cuts of the
        concatenated windows of the epoch's user sequence (every window of every user, once: `recnn_frame_plan_dense`).  The plan
        table (one int64 per batch row of up to `n_batches_max` batches) is (re)made per epoch by `plan_dense`."""
        dev = self.device
        assert items.dtype == torch.int32 and ratings.dtype == torch.float32 and user_off.dtype == torch.int64 and table.dtype == torch.float32
        self._seq = torch.zeros(n_seq_max, dtype=torch.int32, device=dev)
        self._row_off = torch.zeros(n_seq_max + 1, dtype=torch.int32, device=dev)
        self.cursor = torch.zeros(1, dtype=torch.int32, device=dev)
        self._plan = torch.full((n_batches_max * rows,), -1, dtype=torch.int64, device=dev)
        self._plan_args = None
        self._dense = (user_off, frame, rows, n_batches_max)
        self._smp_keep = (items, ratings, user_off, table)
        # (perm / users_per_batch are not read by a planned gather; the device cursor wraps at n_batches_max, which the host's epoch
        # logic never reaches: it resets the cursor at every epoch end)
        m = L.Sampler(items.data_ptr(), ratings.data_ptr(), user_off.data_ptr(), self._seq.data_ptr(), 1, n_batches_max, frame, emb_dim,
                      table.data_ptr(), self._row_off.data_ptr(), self.cursor.data_ptr(), self._plan.data_ptr(), rows)
        L.call("recnn_engine_bind_sampler", self.handle, C.byref(m))
        self.n_batches = n_batches_max
        self.has_sampler = True

    def plan_dense(self, seq_host, skip0: int):
        """Make the plan table of the epoch whose user sequence (store slots, numpy int32) is `seq_host` (stream-ordered)."""
        user_off, frame, rows, n_batches_max = self._dense
        n = int(len(seq_host))
        assert 0 < n <= self._seq.numel()
        self._seq[:n].copy_(torch.from_numpy(seq_host), non_blocking=False)
        L.call("recnn_frame_plan_dense", L.ptr(user_off), L.ptr(self._seq), n, int(skip0), frame, rows, L.ptr(self._row_off),
               n_batches_max * rows, L.ptr(self._plan), L.current_stream())

    def plan_sampler(self):
        """(Re)make the plan table for the current contents of the bound permutation (stream-ordered on the current stream)."""
        if getattr(self, "_plan", None) is None:
            return
        user_off, perm, upb, n_batches, frame, rows = self._plan_args
        L.call("recnn_frame_plan_rows", L.ptr(user_off), L.ptr(perm), upb, n_batches, frame, rows, L.ptr(self._plan),
               L.current_stream())

    def unbind_sampler(self):
        L.call("recnn_engine_bind_sampler", self.handle, None)
        self.has_sampler = False

    def sampler_eager(self, on: bool):
        """Eager calls (step / value_grads / finish) normally run on the BOUND batch even while a sampler is attached
        (only graph replays, the data-parallel phase graphs and profile() draw from it); on=True makes them sample too."""
        L.call("recnn_engine_sampler_eager", self.handle, int(on))

    def profile(self, rows: int, policy, n_steps: int = 20):
        """Per-launch average device time of an eager step: [(name, ms, flops)].  policy = False / True: an ordinary / a policy
        step; policy = 2: cycle mode (one policy cycle's gather + frozen-network launches + one step on the split forward)."""
        cap = 48
        ms = (C.c_float * cap)()
        fl = (C.c_double * cap)()
        names = (C.c_char_p * cap)()
        n = C.c_int(cap)
        L.call("recnn_engine_profile", self.handle, rows, int(policy), n_steps, self._stream(), ms, fl, names, C.byref(n))
        return [(names[i].decode(), float(ms[i]), float(fl[i])) for i in range(n.value)]

    # ------------------------------------------------------------------ stepping
    def step(self, rows: int, learn: bool, step: int):
        L.call("recnn_engine_step", self.handle, rows, int(learn), int(step), self._stream())

    # phase API (external optimizers, data parallel all-reduce between phases); see include/recnn_hip.h
    def value_grads(self, rows: int, learn: bool = True):
        L.call("recnn_engine_value_grads", self.handle, rows, int(learn), self._stream())

    def value_apply(self, soft: bool, grad_scale: float = 1.0):
        L.call("recnn_engine_value_apply", self.handle, int(soft), float(grad_scale), self._stream())

    def policy_grads(self, rows: int, backward: bool):
        L.call("recnn_engine_policy_grads", self.handle, rows, int(backward), self._stream())

    def policy_apply(self, soft: bool, grad_scale: float = 1.0):
        L.call("recnn_engine_policy_apply", self.handle, int(soft), float(grad_scale), self._stream())

    def finish(self, rows: int, value_stepped: bool, policy_stepped: bool):
        L.call("recnn_engine_finish", self.handle, rows, int(value_stepped), int(policy_stepped), self._stream())

    @property
    def policy_every(self) -> int:
        return int(self.hyper.policy_every)

    def value_nets(self):
        return (L.NET_VALUE1, L.NET_VALUE2) if self.td3 else (L.NET_VALUE1,)

    def grad_arena(self, ni: int) -> torch.Tensor:
        return self.grads[ni]

    def set_comm(self, comm, grad_scale: float = None):
        """Attach a connected recnn_amd.parallel.PeerComm (None detaches): every step -- eager or inside the run graphs, which
        must be rebuilt -- then all-reduces the flat gradient arenas in-stream and steps the optimizers on grad / world."""
        self._comm = comm     # keeps the communicator alive as long as the engine points at it
        L.call("recnn_engine_set_comm", self.handle, comm.handle if comm is not None else None,
               float(grad_scale if grad_scale is not None else (1.0 / comm.world if comm is not None else 1.0)))

    def dp_graph_build(self, rows: int, grad_scale: float, overlap_actor: bool = False):
        L.call("recnn_engine_dp_graph_build", self.handle, rows, float(grad_scale), int(overlap_actor), self._stream())
        self._dp_graphs = True

    def dp_sets(self) -> int:
        return int(self.lib.recnn_engine_dp_sets(self.handle))

    def dp_graph_launch(self, which: int):
        L.call("recnn_engine_dp_graph_launch", self.handle, which, self._stream())

    def graph_build(self, rows: int):
        L.call("recnn_engine_graph_build", self.handle, rows, self._stream())

    def graph_prepare(self, first_step: int, n_steps: int):
        L.call("recnn_engine_graph_prepare", self.handle, first_step, n_steps, self._stream())

    def graph_run(self, first_step: int, n_steps: int):
        L.call("recnn_engine_graph_run", self.handle, first_step, n_steps, self._stream())

    def counters(self):
        """(steps finalized, actor optimizer steps, critic 1 steps, critic 2 steps) as the device counts them."""
        h = (C.c_int32 * 4)()
        L.call("recnn_engine_read_counters", self.handle, h, self._stream())
        return tuple(int(x) for x in h)

    def loss_history(self, n_steps: int):
        """Losses of the last `n_steps` (<= 1024) steps, oldest first -- also of steps replayed inside run graphs."""
        end = self.counters()[0]
        ring = self.buffer("loss_ring", 1024)
        rows = ring[torch.arange(end - n_steps, end) % 1024].tolist()       # one conversion, not one per step
        if self.td3:
            return [{"value1": v[0], "value2": v[1], "policy": v[2]} for v in rows]
        return [{"value": v[0], "policy": v[1]} for v in rows]

    def losses(self):
        """Synchronises the stream and returns the last step's losses as python floats."""
        L.call("recnn_engine_read_losses", self.handle, self._losses_host, self._stream())
        v = list(self._losses_host)
        if self.td3:
            return {"value1": v[0], "value2": v[1], "policy": v[2]}
        return {"value": v[0], "policy": v[1]}

    def buffer(self, name: str, rows: Optional[int] = None) -> torch.Tensor:
        """Copy of an intermediate device buffer (debug / tests)."""
        r, c, ld, f = C.c_int64(), C.c_int64(), C.c_int64(), C.c_int()
        p = self.lib.recnn_engine_buffer(self.handle, name.encode(), C.byref(r), C.byref(c), C.byref(ld), C.byref(f))
        if not p:
            raise KeyError(name)
        rows = int(r.value) if rows is None else rows
        esz = 4 if f.value else (4 if self.dtype == "fp32" else 2)
        nbytes = rows * ld.value * esz
        raw = None
        for owner in (self.workspace, self.xn, self.xs) + tuple(getattr(self, "_bound", ())[:2]):
            base = owner.data_ptr()
            if base <= p < base + owner.numel() * owner.element_size():
                raw = owner.view(-1).view(torch.uint8)[p - base: p - base + nbytes]
                break
        if raw is None:
            raise KeyError(name)
        dt = torch.float32 if esz == 4 else torch.bfloat16
        t = raw.view(dt).view(rows, ld.value)
        if esz == 2 and self.dtype == "bf16x3":
            # split-bf16 rows (csrc/x3.h): logical column c = hi at 2 (c & ~31) + (c & 31), lo 32 elements further
            col = torch.arange(c.value, device=t.device)
            col = (col // 32) * 64 + (col % 32)
            t = t[:, col].float() + t[:, col + 32].float()
        else:
            t = t[:, :c.value].float().clone()
        if name in ("critic1_dz2", "critic1_dz1") and self.lib.recnn_engine_unit_backward(self.handle):
            # the fused bf16 path stores dz / d (unit backward tensors); the per-row seed d is applied inside the dW launch
            t *= self.buffer("delta1", rows).reshape(rows, 1)
        return t
