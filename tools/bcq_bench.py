#!/usr/bin/env python3
"""BCQ step throughput at the DDPG bench's batch: 2048 transition rows, state 1290, action 128,
VAE latent 512 / hidden 750 (the notebook's Generator(1290, 128, 512)), perturbator / critics hidden 256, 10 candidates per
next state, perturbator_step 30.  GPU: recnn_amd.nn.bcq_update with the fused HIP Adam, HIP events around K steps.
CPU baseline: the oracle restatement of the reference's bcq_update (oracle/bcq_oracle.py) on the host cores, a bounded
number of steps.  Prints one JSON line.   usage: python tools/bcq_bench.py [--steps 60] [--rows 2048] [--no-cpu]"""
import argparse
import copy
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402


def flops_per_step(B, S, A, L, H, G, n, pstep):
    mac = lambda rows, *dims: rows * sum(a * b for a, b in zip(dims[:-1], dims[1:]))
    enc = mac(B, S + A, G, G, 2 * L)
    dec = mac(B, S + L, G, G, A)
    gen = 3 * (enc + dec)                                         # forward + dX + dW
    cand = (mac(B * n, S + L, G, G, A) + mac(B * n, S + A, H, H, A) + mac(B * n, S + A, H, H, 1))   # as the reference does it
    cand_split = (B * S * G + mac(B * n, L, G, G, A)) + (B * S * H + mac(B * n, A, H, H, A)) + (B * S * H + mac(B * n, A, H, H, 1))
    crit = 3 * mac(B, S + A, H, H, 1)
    pert_f = mac(B, S + L, G, G, A) + mac(B, S + A, H, H, A) + mac(B, S + A, H, H, 1)
    pert_b = (2 * mac(B, S + A, H, H, A) + mac(B, S + A, H, H, 1)) / pstep
    return 2.0 * (gen + cand + crit + pert_f + pert_b), 2.0 * (gen + cand_split + crit + pert_f + pert_b)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--rows", type=int, default=2048)
    ap.add_argument("--latent", type=int, default=512)
    ap.add_argument("--candidates", type=int, default=10)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--dtype", default="fp32", choices=["fp32", "bf16"], help="compute type of the GEMMs (functional.set_mlp_dtype)")
    ap.add_argument("--graphed", action="store_true", help="replay the step from hipGraphs (recnn_amd.nn.GraphedUpdate): one graph launch per step")
    ap.add_argument("--no-split", action="store_true", help="score candidates on materialised repeated states (the reference's formulation)")
    args = ap.parse_args()
    from recnn_amd import optim
    from recnn_amd.nn import functional as F_hip
    F_hip.set_mlp_dtype(args.dtype)
    from recnn_amd.nn import bcq_update
    from recnn_amd.nn import models as M
    S, A, L, H, B, n = 1290, 128, args.latent, 256, args.rows, args.candidates
    torch.manual_seed(0)
    gen, pert, v1, v2 = M.bcqGenerator(S, A, L), M.bcqPerturbator(S, A, H), M.Critic(S, A, H), M.Critic(S, A, H)
    tpert, tv1, tv2 = copy.deepcopy(pert).eval(), copy.deepcopy(v1).eval(), copy.deepcopy(v2).eval()
    params = {"gamma": 0.99, "soft_tau": 0.001, "n_generator_samples": n, "perturbator_step": 30}
    batches = [{"state": torch.randn(B, S), "action": torch.randn(B, A) * 0.5, "reward": torch.randn(B),
                "next_state": torch.randn(B, S), "done": (torch.rand(B) < 0.05).float()} for _ in range(2)]
    cpu = None
    if not args.no_cpu:
        from oracle import recnn_oracle as O
        from oracle import bcq_oracle as Q
        from oracle.reinforce_oracle import AdamDict
        P = O.params_from_module
        st = Q.BCQState(Q.generator_params_from_module(gen), P(pert), P(tpert), P(v1), P(tv1), P(v2), P(tv2),
                        AdamDict(Q.GEN_ORDER, lr=1e-5), AdamDict(O.PARAM_ORDER, lr=1e-5), AdamDict(O.PARAM_ORDER, lr=1e-5),
                        params=dict(params))
        torch.set_num_threads(min(os.cpu_count() or 1, 32))     # many-core hosts lose to oversubscription beyond that (bench.py)
        k, t0 = 0, time.perf_counter()
        while True:
            b = batches[k % 2]
            eps, zn, zc = torch.randn(B, L), torch.randn(B * n, L), torch.randn(B, L)
            mk = O.draw_dropout_masks(6, B, H)
            Q.bcq_step(st, b, eps, zn, zc, mk, step=k + 1)
            k += 1
            el = time.perf_counter() - t0
            if el > 15.0 and k >= 2:
                break
        cpu = {"value": k / el, "unit": "steps/s", "cores": torch.get_num_threads(), "kind": "port",
               "sample": f"{k} BCQ steps of B={B} x {n} candidates (draws + update, fp32, Adam) in {el:.1f}s"}
    for m in (gen, pert, tpert, v1, v2, tv1, tv2):
        m.cuda()
    if args.no_split:
        tpert.train(); tpert.drop_layer.p = 0.0      # off the candidate fast path: handled by the generic branch
    nets = {"generator_net": gen, "perturbator_net": pert, "target_perturbator_net": tpert, "value_net1": v1,
            "target_value_net1": tv1, "value_net2": v2, "target_value_net2": tv2}
    optimizer = {"generator_optimizer": optim.Adam(gen.parameters(), lr=1e-5), "value_optimizer1": optim.Adam(v1.parameters(), lr=1e-5),
                 "value_optimizer2": optim.Adam(v2.parameters(), lr=1e-5), "perturbator_optimizer": optim.Adam(pert.parameters(), lr=1e-5)}
    gb = [{k: v.cuda() for k, v in b.items()} for b in batches]
    step = 0
    if args.graphed:
        from recnn_amd.nn import GraphedUpdate
        optimizer = {k: optim.Adam(o.param_groups[0]["params"], lr=1e-5, capturable=True) for k, o in optimizer.items()}
        gu = GraphedUpdate(bcq_update, gb[0], params, nets, optimizer, period_key="perturbator_step", warmup=2)
        step = 2
        for _ in range(max(args.warmup, 32)):          # both step kinds get captured here (perturbator_step = 30)
            gu(gb[step % 2]); step += 1
    else:
        for _ in range(args.warmup):
            bcq_update(gb[step % 2], params, nets, optimizer, learn=True, step=step)
            step += 1
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(args.steps):
        if args.graphed:
            out = gu(gb[step % 2])
        else:
            out = bcq_update(gb[step % 2], params, nets, optimizer, learn=True, step=step)
        step += 1
    e1.record()
    torch.cuda.synchronize()
    if args.graphed:
        out = {k: (float(v) if k != "step" else v) for k, v in out.items()}
    wall = time.perf_counter() - t0
    ms = e0.elapsed_time(e1) / args.steps
    f_ref, f_split = flops_per_step(B, S, A, L, 256, 750, n, 30)
    line = {"metric": "BCQ update steps/sec (batch %d, %d candidates/state, latent %d)" % (B, n, L), "value": 1000.0 / ms,
            "unit": "steps/s", "ms_per_step": ms, "wall_ms_per_step": 1000.0 * wall / args.steps, "steps": args.steps,
            "dtype": "f32 (exact-fp32 MFMA)" if args.dtype == "fp32" else "bf16 MFMA (fp32 accumulation, master weights, optimizer)", "candidate_path": "materialised" if args.no_split else "shared state part", "issue": "hipGraph replay" if args.graphed else "eager",
            "gflop_per_step_reference_formulation": f_ref / 1e9, "gflop_per_step_executed": (f_ref if args.no_split else f_split) / 1e9,
            "tflops_executed": (f_ref if args.no_split else f_split) / (ms * 1e-3) / 1e12, "final_losses": out, "cpu_baseline": cpu}
    print(json.dumps(line))


if __name__ == "__main__":
    main()
