"""bench.py's choice of the `roofline` kernel (pick_dominant), on the launch tables the two schedules actually produce (numbers from
profiles/r05_bench_driver.json): the launch with the largest share of the step's algorithmic flops; the longest per-step launch beside it."""
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


def test_cycle_schedule_names_the_frozen_launch_and_the_tail_beside_it():
    b = _bench()
    dom, dom_time, flops_per_step, share = b.pick_dominant(CYCLE, 10)
    assert dom[0] == "frozen_actors" and dom_time[0] == "tail_critic"
    assert abs(flops_per_step(dom) - 2.809e9) < 1e6 and abs(share(dom) - 0.00426) < 1e-9       # a cycle launch serves 10 steps
    assert flops_per_step(dom_time) == 0.54e9 and share(dom_time) == 0.0128


def test_fused_schedule_keeps_the_choice_of_rounds_1_to_4():
    b = _bench()
    dom, dom_time, _, _ = b.pick_dominant(FUSED, 10)
    assert dom[0] == dom_time[0] == "mlp_fwd_nets"


def test_no_mfma_launch_at_all_falls_back_to_the_longest_one():
    b = _bench()
    dom, dom_time, _, _ = b.pick_dominant([("frame_gather", 0.0077, 0.0), ("adam_critic", 0.0088, 0.0)], 10)
    assert dom[0] == dom_time[0] == "adam_critic"


def test_every_frozen_slot_has_a_kernel_name_for_the_counter_pass():
    b = _bench()
    for slot in ("frozen_actors", "frozen_target_critics"):
        assert b.KERNEL_OF_SLOT[slot] == "mlp_frozen_kernel" and slot in b.FROZEN_SLOTS
