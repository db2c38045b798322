"""bench.py's choice of the `roofline` kernel, on the launch tables the two schedules actually produce (numbers from
profiles/r05_bench_driver.json): `roofline` is the kernel with the largest us per step, all its launches summed, as a `rocprofv3 --stats`
listing of the same command ranks them (kernel_table); the launch with the largest share of the algorithmic flops is reported beside it
(pick_dominant -> `roofline_flop_dominant`)."""
import importlib.util
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


CYCLE = [("frame_gather_cycle", 0.0284, 0.0), ("frozen_actors", 0.0426, 28.09e9), ("frozen_target_critics", 0.0397, 24.58e9),
         ("l1_critic", 0.0091, 1.49e9), ("tail_critic", 0.0128, 0.54e9), ("dw_critic", 0.0102, 1.76e9), ("adam_critic", 0.0088, 0.0)]
FUSED = [("frame_gather", 0.0077, 0.0), ("mlp_fwd_nets", 0.0261, 7.29e9), ("dw_critic", 0.0129, 1.76e9), ("adam_critic+gather", 0.0116, 0.0)]
# an eager policy step of the fused schedule: the ordinary step's launches + the actor's chain
FUSED_POL = FUSED + [("bwd_chain_policy", 0.0138, 1.2e9), ("dw_actor", 0.0105, 1.75e9), ("grad_reduce_actor", 0.0053, 0.0), ("adam_actor", 0.0090, 0.0)]


def test_cycle_schedule_flop_dominant_launch_is_the_frozen_one():
    b = _bench()
    dom, dom_time, flops_per_step, share = b.pick_dominant(CYCLE, 10)
    assert dom[0] == "frozen_actors" and dom_time[0] == "tail_critic"
    assert abs(flops_per_step(dom) - 2.809e9) < 1e6 and abs(share(dom) - 0.00426) < 1e-9       # a cycle launch serves 10 steps
    assert flops_per_step(dom_time) == 0.54e9 and share(dom_time) == 0.0128


def test_roofline_kernel_is_the_one_with_the_most_time_per_step():
    b = _bench()
    tab = b.kernel_table(CYCLE, FUSED, FUSED_POL, 10)
    by = {t["kernel"]: t for t in tab}
    # the two launches of mlp_frozen_kernel count 1/10 per step each: (42.6 + 39.7) / 10 = 8.23 us/step
    assert abs(by["mlp_frozen_kernel"]["ms_per_step"] - 0.00823) < 1e-9 and abs(by["mlp_frozen_kernel"]["launches_per_step"] - 0.2) < 1e-12
    # the dW kernel carries the critic's launch every step and the actor's every 10th: 10.2 + 1.05 us/step, 1.1 launches
    dw = by["gemm_dw_dma_kernel"]
    assert abs(dw["ms_per_step"] - (0.0102 + 0.00105)) < 1e-9 and abs(dw["launches_per_step"] - 1.1) < 1e-12
    assert abs(dw["avg_ms"] - dw["ms_per_step"] / 1.1) < 1e-12 and abs(dw["flops_per_launch"] - (1.76e9 + 0.175e9) / 1.1) < 1.0
    assert tab[0]["kernel"] == "mlp_tail_kernel" and tab[0]["slots"] == ["tail_critic"]          # 12.8 us/step: the top of the list
    assert abs(sum(t["share_of_step_time"] for t in tab) - 1.0) < 1e-12
    assert [t["ms_per_step"] for t in tab] == sorted((t["ms_per_step"] for t in tab), reverse=True)


def test_fused_schedule_keeps_the_choice_of_rounds_1_to_4():
    b = _bench()
    dom, dom_time, _, _ = b.pick_dominant(FUSED, 10)
    assert dom[0] == dom_time[0] == "mlp_fwd_nets"
    assert b.kernel_table(FUSED, FUSED, FUSED_POL, 10)[0]["kernel"] == "mlps_fwd_kernel"


def test_no_mfma_launch_at_all_falls_back_to_the_longest_one():
    b = _bench()
    dom, dom_time, _, _ = b.pick_dominant([("frame_gather", 0.0077, 0.0), ("adam_critic", 0.0088, 0.0)], 10)
    assert dom[0] == dom_time[0] == "adam_critic"


def test_every_frozen_slot_has_a_kernel_name_for_the_counter_pass():
    b = _bench()
    for slot in ("frozen_actors", "frozen_target_critics"):
        assert b.KERNEL_OF_SLOT[slot] == "mlp_frozen_kernel" and slot in b.FROZEN_SLOTS
