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


# Probe name (what runtime.ops brackets with events: one host-side launch call) -> (family, stream of the pipelined step it
# runs on, device kernels one such call emits as the profiler names them).  A family merges the instantiations and the
# large-image variants of one kernel design; the grouped weight-gradient call emits a pointwise kernel and several tap-list
# kernels, the image-end call one of its few kernels (stem, head, the 32-channel 3x3 kernel).
FAMILIES = {
    "sir_fwd_kernel": ("sir_fwd", "main", ("sir_fwd_kernel",)),
    "sir_fwd_big_kernel": ("sir_fwd", "main", ("sir_fwd_big_kernel",)),
    "sir_bwd_kernel": ("sir_bwd", "main", ("sir_bwd_kernel",)),
    "sir_bwd_big_kernel": ("sir_bwd", "main", ("sir_bwd_big_kernel",)),
    "conv_wgrad_group_kernel": ("conv_wgrad_group", "side", ("conv_wgrad_group_kernel", "pw_wgrad_group_kernel")),
    "conv_wgrad_kernel": ("conv_wgrad", "side", ("conv_wgrad_kernel", "pw_wgrad_kernel", "conv_wgrad_pipe_kernel", "conv_wgrad_c32_kernel")),
    "conv_thin_kernel": ("conv_thin", "main", ("conv_thin_in_kernel", "conv_thin_out_kernel", "conv_thin_out_mfma_kernel", "conv3x3_c32_kernel")),
    "embed_bwd_kernel": ("embed_bwd", "side", ("embed_bwd_kernel",)),
    "dwconv3x3_wgrad_kernel": ("dwconv3x3_wgrad", "side", ("dwconv3x3_wgrad_kernel",)),
    "unet_lds_kernel": ("unet_lds", "main", ("unet_lds_kernel",)),
}


def _family(name: str):
    return FAMILIES.get(name, (name[:-7] if name.endswith("_kernel") else name, "main", (name,)))


def _pmc_bytes_per_step(measured: Dict, kernels) -> float:
    """HBM bytes one step moves in the given device kernels, from the counter-pass file: per kernel name it holds the bytes per
    launch and the launches sampled; the step counter's kernel runs once per step (the optimizer kernel may run in pieces), so its
    launch count is the number of steps."""
    steps = measured.get("bump_step_kernel", measured.get("adam_kernel", {})).get("launches_sampled", 0)
    if not steps:
        return None
    tot, seen = 0.0, False
    for k in kernels:
        if k in measured:
            tot += measured[k]["hbm_bytes_per_launch"] * measured[k]["launches_sampled"] / steps
            seen = True
    return tot if seen else None


def measure(step: Callable[[], None], reps: int = 3, traffic_file: str = "", detail: list = None) -> Dict:
    """Run `step` (one full eager pass of the hot path) reps+1 times with the probe installed; the first is warm-up.
    `detail`, when a list, also receives one row per distinct launch shape.

    Reported: per kernel family (see FAMILIES) launches, time, algorithmic bytes and flops per launch and, from the counter
    file, the HBM bytes per launch (sum over the device kernels of one host call); at the top level the family with the
    largest total time, and under `critical_path` the family with the largest time on the MAIN stream of the pipelined step
    (the side stream's leaf work runs beside it and is not what the step waits for)."""
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
    meta = {}
    for kernel, e0, e1, nbytes, flops, tag in probe.rows:
        sec = e0.elapsed_time(e1) * 1e-3
        if detail is not None:
            b = shapes[kernel + " " + tag]
            b[0] += 1
            b[1] += sec
            b[2] += nbytes
            b[3] += flops
        fam, stream, members = _family(kernel)
        if fam in meta:
            members = tuple(dict.fromkeys(meta[fam][1] + members))
        meta[fam] = (stream, members)
        a = acc[fam]
        a[0] += 1
        a[1] += sec
        a[2] += nbytes
        a[3] += flops
    if detail is not None:      # per launch shape: (label, launches per step, average us, GB/s, TFLOP/s)
        detail.extend((k, c // reps, 1e6 * sec / c, nb / sec * 1e-9, fl / sec * 1e-12) for k, (c, sec, nb, fl) in shapes.items())
    measured = {}
    if traffic_file and os.path.exists(traffic_file):      # HBM bytes per launch from a separate counter pass, when recorded
        with open(traffic_file) as f:
            measured = json.load(f)
    table = {}
    for k, (cnt, sec, nbytes, flops) in acc.items():
        per_step = _pmc_bytes_per_step(measured, meta[k][1])
        table[k] = {"stream": meta[k][0], "device_kernels": list(meta[k][1]), "launches_per_step": cnt // reps, "avg_us": 1e6 * sec / cnt,
                    "ms_per_step": 1e3 * sec / reps, "GB/s": nbytes / sec * 1e-9, "TFLOP/s": flops / sec * 1e-12,
                    "bytes_per_launch": nbytes / cnt,
                    "hbm_bytes_per_launch": None if per_step is None else per_step / max(1, cnt // reps)}

    def entry(name: str) -> Dict:
        t = table[name]
        hbm_frac, mfma_frac = t["GB/s"] / HBM_PEAK_GBS, t["TFLOP/s"] / MFMA_BF16_PEAK_TFLOPS
        if hbm_frac >= mfma_frac:
            o = {"bound": "hbm", "achieved": t["GB/s"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": hbm_frac}
        else:
            o = {"bound": "mfma", "achieved": t["TFLOP/s"], "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": mfma_frac}
        o.update({"traffic": t["hbm_bytes_per_launch"], "kernel": name, "device_kernels": t["device_kernels"], "stream": t["stream"],
                  "launches_per_step": t["launches_per_step"], "avg_us": t["avg_us"], "ms_per_step": t["ms_per_step"],
                  "algorithmic_bytes_per_launch": t["bytes_per_launch"]})
        return o

    out = entry(max(table, key=lambda k: table[k]["ms_per_step"]))
    main = [k for k in table if table[k]["stream"] == "main"]
    if main:
        out["critical_path"] = entry(max(main, key=lambda k: table[k]["ms_per_step"]))
    out["traffic_source"] = (f"profiles/{os.path.basename(traffic_file)}: separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over eager "
                             "steps of this code on an earlier box (read side x2, the gfx950 correction), summed over the device "
                             "kernels of one host call; not measured in this run") if measured else None
    # whole step: algorithmic bytes and flops of every launch of one step over the time the launches took back to back
    tot_sec = sum(v[1] for v in acc.values()) / reps
    tot_b = sum(v[2] for v in acc.values()) / reps
    tot_f = sum(v[3] for v in acc.values()) / reps
    out["step"] = {"algorithmic_GB": tot_b * 1e-9, "GFLOP": tot_f * 1e-9, "kernel_ms": tot_sec * 1e3,
                   "main_stream_kernel_ms": sum(table[k]["ms_per_step"] for k in main),
                   "hbm_frac": tot_b / tot_sec * 1e-9 / HBM_PEAK_GBS, "mfma_frac": tot_f / tot_sec * 1e-12 / MFMA_BF16_PEAK_TFLOPS}
    # per kernel family, over the same eager launches: algorithmic bytes per launch beside the HBM bytes per launch the
    # counter pass measured (below the algorithmic figure where operands are still in the last-level cache from their
    # producer, above it where a kernel re-reads)
    out["kernels"] = {k: dict({kk: (round(vv, 3) if isinstance(vv, float) else vv) for kk, vv in v.items()
                               if kk not in ("bytes_per_launch", "hbm_bytes_per_launch", "device_kernels")},
                              algorithmic_bytes_per_launch=round(v["bytes_per_launch"]),
                              hbm_bytes_per_launch=(None if v["hbm_bytes_per_launch"] is None else round(v["hbm_bytes_per_launch"])))
                      for k, v in table.items()}
    return out
