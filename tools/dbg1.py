import os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd())
from oracle import recnn_oracle as O
from recnn_amd import _lib as L
from recnn_amd.nn.engine import StepEngine
from tests.helpers import rel_err
from tests.test_gpu_engine import _init_nets, _rand_batch, _unpack

def grads_of(eng, ni):
    return eng.param_views(ni, eng.grads[ni])

def tiny():
    g = np.load("tests/golden/ddpg_tiny.npz")
    in_dim, act, hid, B, steps, _ = [int(x) for x in g["dims"]]
    eng = StepEngine("ddpg", in_dim, act, hid, B, dtype="fp32", mask_mode="external")
    pol, val = _unpack(g, "policy"), _unpack(g, "value")
    for ni, p in ((0, pol), (1, pol), (2, val), (3, val)): eng.load_params(ni, p)
    eng.set_hyper(policy_opt=dict(lr=1e-3), value_opt=dict(lr=1e-3)); eng.set_counters()
    b = {k: torch.from_numpy(g[f"batch0.{k}"]) for k in ("state", "action", "reward", "next_state", "done")}
    masks = [torch.from_numpy(m) for m in g["masks"][0]]
    ost = O.DDPGState.create(O.clone_params(pol), O.clone_params(val), O.AdamState(), O.AdamState())
    tr = {}
    O.ddpg_step(ost, b, masks, 0, True, trace=tr)
    eng.pack_batch(b["state"], b["action"], b["reward"], b["next_state"], b["done"]); eng.set_external(masks=masks)
    eng.step(B, True, 0); print("tiny losses", eng.losses())
    print(" oracle loss", float(((tr["value"]-tr["expected"])**2).mean()), float(-tr["q_pi"].mean()), " from eng buffers", float(((eng.buffer("q1",B)-eng.buffer("expected",B))**2).mean()), float(-eng.buffer("q_pi",B).mean()))
    for name, key in (("next_action","next_action"),("target_q","target_value"),("expected","expected"),("q1","value"),("gen_action","gen_action"),("q_pi","q_pi")):
        print(" tiny", name, rel_err(eng.buffer(name, B), tr[key]))
    # hidden of critic
    s, a = b["state"], b["action"]
    _, (x, h1, h2) = O.critic_forward(val, s, a, masks[0], masks[1])
    print(" tiny critic h1", rel_err(eng.buffer("critic1_h1", B), h1), "h2", rel_err(eng.buffer("critic1_h2", B), h2))
    print(" xs", rel_err(eng.xs[:B, :act], a), rel_err(eng.xs[:B, act:act+in_dim], s))

def big(algo, B, dtype="fp32", vlr=1e-3):
    S, A, H = 1290, 128, 256
    nc = 2 if algo == "td3" else 1
    actor, critics = _init_nets(2, S, A, H, nc)
    gen = torch.Generator().manual_seed(3)
    batch = _rand_batch(B, S, A, gen)
    nm = 8 if algo == "td3" else 6
    masks = [(torch.rand(B, H, generator=gen) < 0.5).to(torch.uint8) for _ in range(nm)]
    noise = torch.randn(B, A, generator=gen) * 0.5
    eng = StepEngine(algo, S, A, H, B, dtype=dtype, mask_mode="external")
    tr = {}
    if algo == "td3":
        ost = O.TD3State.create(O.clone_params(actor), O.clone_params(critics[0]), O.clone_params(critics[1]), O.AdamState(lr=1e-3), O.AdamState(lr=vlr), O.AdamState(lr=vlr))
        for ni, p in ((0, actor), (1, actor), (2, critics[0]), (3, critics[0]), (4, critics[1]), (5, critics[1])): eng.load_params(ni, p)
    else:
        ost = O.DDPGState.create(O.clone_params(actor), O.clone_params(critics[0]), O.AdamState(lr=1e-3), O.AdamState(lr=vlr))
        for ni, p in ((0, actor), (1, actor), (2, critics[0]), (3, critics[0])): eng.load_params(ni, p)
    eng.set_hyper(policy_opt=dict(lr=1e-3), value_opt=dict(lr=vlr)); eng.set_counters()
    eng.pack_batch(batch["state"], batch["action"], batch["reward"], batch["next_state"], batch["done"])
    eng.set_external(masks=masks, noise=noise if algo == "td3" else None)
    if algo == "td3":
        ref = O.td3_step(ost, batch, noise, masks, 0, True, trace=tr)
    else:
        ref = O.ddpg_step(ost, batch, masks, 0, True, trace=tr)
    eng.step(B, True, 0)
    print(algo, B, dtype, "losses", eng.losses(), ref)
    gv = grads_of(eng, 2); gp = grads_of(eng, 0)
    vg = tr["value_grads"] if algo == "ddpg" else tr["value1_grads"]
    for k in O.PARAM_ORDER:
        print("  value grad", k, rel_err(gv[k], vg[k]), "  policy grad", k, rel_err(gp[k], tr["policy_grads"][k]), float(tr["policy_grads"][k].abs().sum()), float(gp[k].abs().sum()))
    print("  coef", eng.buffer("clip_coef").item(), tr["clip_coef"])
    print("  dact", rel_err(eng.buffer("dact", B), tr["dact"]), "gen_action", rel_err(eng.buffer("gen_action", B), tr["gen_action"]))
    v1 = ost.value if algo == "ddpg" else ost.value1   # already updated
    mi = 4 if algo == "ddpg" else 6
    q, (x, h1, h2) = O.critic_forward(v1, batch["state"], tr["gen_action"], masks[mi], masks[mi+1])
    dq = torch.full_like(q, -1.0 / B)
    _, dxa, inter = O.mlp_backward(v1, (x, h1, h2), dq, True, need_dx=True, need_dw=False)
    print("  pc_h1", rel_err(eng.buffer("pc_h1", B), h1), "pc_h2", rel_err(eng.buffer("pc_h2", B), h2), "dze2", rel_err(eng.buffer("dze2", B), inter["dz2"]), "dze1", rel_err(eng.buffer("dze1", B), inter["dz1"]))
    d = (eng.buffer("dact", B).cpu() - tr["dact"]).abs()
    print("  dact err rows", (d.max(1).values > 1e-3 * tr["dact"].abs().max()).nonzero().flatten()[:20].tolist(), "cols", (d.max(0).values > 1e-3 * tr["dact"].abs().max()).nonzero().flatten()[:20].tolist())
    for tag, ni, refp in (("policy", 0, ost.policy), ("value", 2, ost.value if algo == "ddpg" else ost.value1)):
        got = eng.param_views(ni)
        print("  params", tag, [f"{rel_err(got[k], refp[k]):.1e}" for k in O.PARAM_ORDER])

big("td3", 4096, vlr=0.0)
big("td3", 4096, vlr=1e-3)
