"""bench.py's own N > 1 code path on a one-GPU box: `RECNN_BENCH_SINGLE_DEVICE=1 python bench.py --gpus 2` starts its
two ranks itself (torch.distributed.run, gloo, both ranks on GPU 0), runs the multi-GPU preflight in front of the timed regions, steps the
data-parallel path and prints ONE JSON line that says what the collective was, what `value` counts and what the preflight found.
Functional only -- ranks sharing a GPU are time-sliced -- so nothing here asserts a rate.  The same for the catalogue-sharded
REINFORCE entry (`bench.py --algo reinforce --gpus 2`, tools/reinforce_bench.py)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(cmd, extra_env, timeout=600):
    env = dict(os.environ, RECNN_BENCH_SINGLE_DEVICE="1", HSA_ENABLE_IPC_MODE_LEGACY="0", **extra_env)
    p = subprocess.run([sys.executable] + cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
    assert p.returncode == 0, (p.stdout[-2000:], p.stderr[-3000:])
    lines = [ln for ln in p.stdout.strip().splitlines() if ln.startswith("{")]
    return json.loads(lines[-1]), lines


def test_bench_two_ranks_on_one_gpu_prints_the_multi_gpu_record(cuda):
    out, lines = _run(["bench.py", "--gpus", "2", "--steps", "4", "--warmup", "2", "--repeats", "2", "--no-extras", "--no-cpu-baseline",
                       "--no-traffic"], {})
    assert out["n_gpus"] == 2 and out["metric"].startswith("DDPG update steps/sec") and out["scaling"] == "weak"
    mg = out["multi_gpu"]
    assert mg["world_size"] == 2 and mg["backend"] == "gloo" and mg["ranks_share_one_gpu"] is True
    assert mg["collective"].split()[0] in ("peer", "rccl") and mg["collective_requested"] == "peer"
    # value = rank-steps/s, global_updates_per_s = synchronised updates/s: a factor of N apart under weak scaling
    assert out["value_is"].startswith("rank-steps/s")
    assert abs(out["value"] - 2 * out["global_updates_per_s"]) <= 1e-6 * out["value"]
    assert abs(mg["rank_steps_per_s"] - out["value"]) <= 1e-6 * out["value"]
    assert "rank-steps/s" in out["config"]["workload"] and out["config"]["parallelism"].startswith("dp2")
    # the preflight ran in front of the timed regions and its summary is embedded; its stage lines came first on stdout
    pf = mg["preflight"]
    assert pf is not None and set(pf["stages"]) >= {"devices", "process_group", "identity", "replicas"}, pf
    assert pf["stages"]["identity"] and pf["stages"]["replicas"], pf
    assert any('"stage": "summary"' in ln for ln in lines[:-1])
    # round 6: the same run times BOTH exchanges -- per-collective latency at the critic's / TD3's gradient sizes and the data-parallel
    # steps with the device collective (headline) and with host-issued all-reduces -- so that one SCALE run yields the decision
    ab = mg["collective_ab"]
    assert ab is not None and set(ab["latency_us"]) == {"429312", "858496"}, ab
    for row in ab["latency_us"].values():
        assert row["rccl"] > 0 and (mg["collective"].split()[0] != "peer" or row["peer"] > 0), ab
    assert ab["rccl"]["ms_per_step"] > 0 and ab["rccl"]["rank_steps_per_s"] > 0
    if mg["collective"].split()[0] == "peer":
        assert ab["peer"]["ms_per_step"] > 0 and ab["faster"] in ("peer", "rccl") and ab["steps"] == 6
    pf_votes = pf.get("peer_votes")
    assert pf_votes is not None and len(pf_votes) == 2, pf            # the collective was chosen by a vote over the ranks
    assert all(abs(v) < 1e6 for v in out["config"]["final_losses"].values())


def test_bench_reinforce_sharded_over_two_ranks_on_one_gpu(cuda):
    """`--algo reinforce --gpus 2`: the vocab-parallel actor / critic pair inside reinforce_update, batches from the discrete-action
    FrameEnv, learned Beta (small catalogue here: the entry point, not the 100k measurement)."""
    env = {"RECNN_REINFORCE_ITEMS": "4000", "RECNN_REINFORCE_HIDDEN": "128"}
    out, _ = _run(["bench.py", "--algo", "reinforce", "--gpus", "2", "--steps", "23", "--dtype", "fp32"], env)
    assert out["n_gpus"] == 2 and out["unit"] == "update iterations/s" and out["value"] > 0
    d = out["detail"]
    assert d["world"] == 2 and d["backend"] == "gloo" and d["n_items"] == 4000 and d["shard"] == [0, 2000]
    assert len(d["policy_step_ms"]) == 2 and d["beta"] == "learned" and "FrameEnv" in d["batches"]
    one, _ = _run(["bench.py", "--algo", "reinforce", "--gpus", "1", "--steps", "23", "--dtype", "fp32"], env)
    assert one["n_gpus"] == 1 and one["detail"]["world"] == 1 and len(one["detail"]["policy_step_ms"]) == 2
