"""CPU: the pieces of bench.py that turn raw measurements into the JSON line -- the marker-based split of a counter trace,
the roofline block of a K-step run and of the full-length kernel -- on synthetic inputs (no GPU)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_counter_trace_is_split_at_the_marker_dispatches():
    import bench
    k16, k20, kz, mark = "void fdtd::k_bulk_split<float, 16, 4>(p)", "void fdtd::k_bulk_split<float, 20, 4>(p)", \
        "void fdtd::k_zone<float, 20>(p)", "void fdtd::k_reduce<float>(a)"
    trace = [(k16, {"FETCH_SIZE": 9.0})] * 5 + [(mark, {})] * 3 + [(kz, {"FETCH_SIZE": 1.0}), (k20, {"FETCH_SIZE": 10.0})] * 6 + \
        [(mark, {})] * 3 + [(k16, {"FETCH_SIZE": 8.0})] * 8 + [(mark, {})]
    secs = bench._sections(trace)
    assert [len(s) for s in secs] == [5, 12, 8]
    assert bench._is_pass(k20) and bench._is_pass(kz) and not bench._is_pass(mark)
    assert bench._short(k20) == "fdtd::k_bulk_split<float, 20, 4>"


def test_roofline_block_of_a_run_and_of_the_full_length_kernel():
    import bench
    cells, steps = 16384 * 16384, 20
    r = dict(events_ms=[1.9, 1.8, 1.7, 1.75, 1.72], walls=[0.002] * 5, pass_launches=1, step_launches=0, bpc=24, launch_steps=16,
             run_shape=[300, 4, 150, 1, 1, 0, 0], run_last_nt=20, full_shape=[309, 4, 137, 1, 1, 0, 0], launch_ms=1.30,
             launch_ms_minmax=[1.28, 1.40])
    traffic = {"run": {"bytes": 7_300_000_000, "read": 4_080_000_000, "write": 3_220_000_000, "valu_insts": 1.2e9,
                       "kernels": [{"name": "fdtd::k_zone<float, 20>", "dispatches": 1.0}, {"name": "fdtd::k_bulk_split<float, 20>", "dispatches": 1.0}]},
               "full": {"bytes": 7_100_000_000, "read": 3_880_000_000, "write": 3_220_000_000, "valu_insts": 9.3e8,
                        "kernels": [{"name": "fdtd::k_bulk_split<float, 16>", "dispatches": 1.0}]},
               "passes_per_run": 1, "cycle": 16, "shapes": {}, "source": "test"}
    b = bench.roofline_block(cells, steps, r, traffic)
    ev = float(np.median(r["events_ms"]))
    assert b["bound"] == "hbm" and b["peak"] == 8000.0 and b["unit"] == "GB/s"
    assert abs(b["achieved"] - 7.3e9 / (ev * 1e-3) / 1e9) < 0.1 and abs(b["frac"] - b["achieved"] / 8000.0) < 1e-3
    assert b["traffic"] == 7_300_000_000 and b["traffic_per_run"] == 7_300_000_000 and abs(b["overfetch"] - 7.3e9 / (cells * 24)) < 1e-3
    assert abs(b["avg_launch_ms"] - ev) < 1e-4 and abs(b["traffic"] / (b["avg_launch_ms"] * 1e-3) / 1e9 - b["achieved"]) < 0.5
    assert b["kernel"] == "fdtd::k_zone<float, 20> + fdtd::k_bulk_split<float, 20>" and b["launch_shape"]["pass_steps"] == 20
    assert abs(b["algorithmic"]["x_peak"] - cells * steps * 24 / (ev * 1e-3) / 1e9 / 8000.0) < 1e-2
    ss = b["steady_state"]
    assert ss["steps_per_launch"] == 16 and abs(ss["value"] - cells * 16 / 1.30e-3 / 1e6) < 1.0
    assert abs(ss["frac"] - 7.1e9 / 1.30e-3 / 1e9 / 8000.0) < 1e-3 and ss["launch_shape"]["xcd_map"] == 1
    # a failed live measurement leaves the timing in place and says why there is no traffic figure
    b2 = bench.roofline_block(cells, steps, r, "live PMC failed (test)")
    assert b2["traffic"] is None and b2["frac"] is None and "traffic_note" in b2 and b2["steady_state"]["value"] == ss["value"]
