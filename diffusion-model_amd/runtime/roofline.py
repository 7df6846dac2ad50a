"""Live roofline measurement of the hot-path kernels.

One eager pass of the benchmarked step runs with a launch probe installed in runtime.ops: every launch of a
compute kernel is bracketed by two timing events recorded on the stream the kernel is launched on, and its
algorithmic cost (each operand read once, each result written once; 2 flops per multiply-accumulate) is computed
from the launch's shapes.  Per kernel name: achieved rate = sum of algorithmic bytes / sum of launch durations.
The kernel with the largest total duration is reported against its bound."""
from __future__ import annotations

import json
import os
from collections import defaultdict
from typing import Callable, Dict

import torch

from . import ops

SPIN_CYCLES = 60000
HBM_PEAK_GBS = 8000.0          # HBM3E peak of one MI355X
MFMA_BF16_PEAK_TFLOPS = 2500.0  # dense bf16 matrix-core peak


class _Probe:
    def __init__(self):
        self.rows = []

    def launch(self, kernel: str, cost, fn):
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        # keep the GPU busy for a few tens of microseconds first, so that both events and the launch are already queued
        # when it gets to them: otherwise the interval also contains the host's launch latency (several microseconds
        # per call from Python), which a captured graph replay does not pay
        torch.cuda._sleep(SPIN_CYCLES)
        e0.record()
        fn()
        e1.record()
        self.rows.append((kernel() if callable(kernel) else kernel, e0, e1) + tuple(cost()))


def measure(step: Callable[[], None], reps: int = 3, traffic_file: str = "", detail: list = None) -> Dict:
    """Run `step` (one full eager pass of the hot path) reps+1 times with the probe installed; the first is warm-up.
    `detail`, when a list, also receives one row per distinct launch shape."""
    probe = _Probe()
    ops.PROBE = probe
    try:
        step()
        torch.cuda.synchronize()
        probe.rows.clear()
        for _ in range(reps):
            step()
        torch.cuda.synchronize()
    finally:
        ops.PROBE = None
    acc = defaultdict(lambda: [0, 0.0, 0.0, 0.0])      # launches, seconds, bytes, flops
    shapes = defaultdict(lambda: [0, 0.0, 0.0, 0.0])
    for kernel, e0, e1, nbytes, flops, tag in probe.rows:
        if detail is not None:
            b = shapes[kernel + " " + tag]
            b[0] += 1
            b[1] += e0.elapsed_time(e1) * 1e-3
            b[2] += nbytes
            b[3] += flops
        a = acc[kernel]
        a[0] += 1
        a[1] += e0.elapsed_time(e1) * 1e-3
        a[2] += nbytes
        a[3] += flops
    if detail is not None:      # per launch shape: (label, launches per step, average us, GB/s, TFLOP/s)
        detail.extend((k, c // reps, 1e6 * sec / c, nb / sec * 1e-9, fl / sec * 1e-12) for k, (c, sec, nb, fl) in shapes.items())
    table = {}
    for k, (cnt, sec, nbytes, flops) in acc.items():
        table[k] = {"launches_per_step": cnt // reps, "avg_us": 1e6 * sec / cnt, "ms_per_step": 1e3 * sec / reps,
                    "GB/s": nbytes / sec * 1e-9, "TFLOP/s": flops / sec * 1e-12, "bytes_per_launch": nbytes / cnt}
    top = max(table, key=lambda k: table[k]["ms_per_step"])
    t = table[top]
    hbm_frac, mfma_frac = t["GB/s"] / HBM_PEAK_GBS, t["TFLOP/s"] / MFMA_BF16_PEAK_TFLOPS
    traffic, measured = None, {}
    if traffic_file and os.path.exists(traffic_file):      # HBM bytes per launch from a separate counter pass, when recorded
        with open(traffic_file) as f:
            measured = json.load(f)
        traffic = measured.get(top, {}).get("hbm_bytes_per_launch")
    if hbm_frac >= mfma_frac:
        out = {"bound": "hbm", "achieved": t["GB/s"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": hbm_frac}
    else:
        out = {"bound": "mfma", "achieved": t["TFLOP/s"], "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": mfma_frac}
    # whole step: algorithmic bytes and flops of every launch of one step over the time the launches took back to back
    tot_sec = sum(v[1] for v in acc.values()) / reps
    tot_b = sum(v[2] for v in acc.values()) / reps
    tot_f = sum(v[3] for v in acc.values()) / reps
    out["step"] = {"algorithmic_GB": tot_b * 1e-9, "GFLOP": tot_f * 1e-9, "kernel_ms": tot_sec * 1e3,
                   "hbm_frac": tot_b / tot_sec * 1e-9 / HBM_PEAK_GBS, "mfma_frac": tot_f / tot_sec * 1e-12 / MFMA_BF16_PEAK_TFLOPS}
    out.update({"traffic": traffic, "kernel": top, "launches_per_step": t["launches_per_step"], "avg_us": t["avg_us"],
                "algorithmic_bytes_per_launch": t["bytes_per_launch"],
                # per kernel family, over the same eager launches: algorithmic bytes per launch beside the HBM bytes per launch
                # the counter pass measured (below the algorithmic figure where operands are still in the last-level cache
                # from their producer, above it where a kernel re-reads)
                "kernels": {k: dict({kk: round(vv, 3) for kk, vv in v.items() if kk != "bytes_per_launch"},
                                    algorithmic_bytes_per_launch=round(v["bytes_per_launch"]),
                                    hbm_bytes_per_launch=(round(measured[k]["hbm_bytes_per_launch"]) if k in measured else None))
                            for k, v in table.items()}})
    return out
