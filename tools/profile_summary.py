#!/usr/bin/env python3
"""Condense rocprofv3 output directories into the small files kept under profiles/.

  python tools/profile_summary.py stats  <dir with *_kernel_stats.csv>  profiles/rNN_kernel_stats.csv
  python tools/profile_summary.py traffic <FETCH_SIZE dir> <WRITE_SIZE dir> profiles/hbm_traffic.json
  python tools/profile_summary.py mfma <SQ counter dir> <kernel stats dir> profiles/rNN_mfma_util.json

traffic: HBM bytes per launch and kernel family = (2 x FETCH_SIZE + WRITE_SIZE) x 1024.  rocprofv3 reports both
counters in KiB; on gfx950 FETCH_SIZE tallies the 128-byte requests of wide coalesced reads at 64 bytes, hence the
factor 2 on the read side.  Kernel families merge template instances (conv_mfma_kernel<4,1,1> -> conv_mfma_kernel).

mfma: matrix-core evidence per kernel family from one counter pass (SQ_VALU_MFMA_BUSY_CYCLES, SQ_BUSY_CU_CYCLES,
SQ_INSTS_VALU_MFMA_MOPS_BF16, GRBM_GUI_ACTIVE):
  mfma_busy_frac_of_cu_busy = SQ_VALU_MFMA_BUSY_CYCLES / (4 x SQ_BUSY_CU_CYCLES)   (busy SIMD-cycles of the matrix pipes over
                              4 SIMDs x busy CU-cycles: how full the matrix pipes are while the kernel's CUs have work)
  mfma_busy_frac_of_chip    = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE x 256 CUs x 4 SIMDs)
  bf16_tflops               = SQ_INSTS_VALU_MFMA_MOPS_BF16 x 512 flops / average duration from the kernel statistics
                              (against the 2500 TFLOP/s dense bf16 peak)"""
import collections
import csv
import glob
import json
import os
import re
import shutil
import sys


def family(name: str) -> str:
    name = re.sub(r"^void ", "", name).replace("(anonymous namespace)::", "")
    m = re.match(r"_Z(\d+)", name)
    if m:
        n = int(m.group(1))
        return name[m.end():m.end() + n]
    return re.split(r"[<(]", name)[0].strip()


def instance(name: str) -> str:
    """template instance without the argument list: conv_img_kernel<8, 1, 1, 1, 512, 0>"""
    name = re.sub(r"^void ", "", name).replace("(anonymous namespace)::", "")
    depth = 0
    for i, ch in enumerate(name):
        depth += ch == "<"
        depth -= ch == ">"
        if ch == "(" and depth == 0:
            return name[:i].strip()
    return name.strip()


def counter(dirname: str, which: str, key=family):
    f = glob.glob(os.path.join(dirname, "**", "*_counter_collection.csv"), recursive=True)[0]
    acc = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != which:
            continue
        a = acc[key(r["Kernel_Name"])]
        a[0] += 1
        a[1] += float(r["Counter_Value"])
    return acc


def main():
    if sys.argv[1] == "stats":
        f = glob.glob(os.path.join(sys.argv[2], "**", "*_kernel_stats.csv"), recursive=True)[0]
        shutil.copy(f, sys.argv[3])
    elif sys.argv[1] == "traffic":
        rd, wr = counter(sys.argv[2], "FETCH_SIZE"), counter(sys.argv[3], "WRITE_SIZE")
        out = {}
        for k in sorted(rd):
            n, kib_r = rd[k]
            kib_w = wr.get(k, [1, 0.0])[1] / max(1, wr.get(k, [1, 0.0])[0])
            out[k] = {"launches_sampled": n, "fetch_kib_per_launch_raw": kib_r / n, "write_kib_per_launch": kib_w,
                      "hbm_bytes_per_launch": (2.0 * kib_r / n + kib_w) * 1024.0}
        json.dump(out, open(sys.argv[4], "w"), indent=1)
    elif sys.argv[1] == "mfma":
        names = ("SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CU_CYCLES", "SQ_INSTS_VALU_MFMA_MOPS_BF16", "GRBM_GUI_ACTIVE")
        out = {}
        for key, into in ((family, out), (instance, out.setdefault("by_template_instance", {}))):
            mfma_rows(names, key, into)
        json.dump(out, open(sys.argv[4], "w"), indent=1)
    else:
        raise SystemExit(__doc__)


def mfma_rows(names, key, out):
        cs = {n: counter(sys.argv[2], n, key) for n in names}
        dur = {}
        f = glob.glob(os.path.join(sys.argv[3], "**", "*_kernel_stats.csv"), recursive=True)
        if f:
            tot = collections.defaultdict(lambda: [0, 0.0])
            for r in csv.DictReader(open(f[0])):
                t = tot[key(r["Name"])]
                t[0] += int(r["Calls"]); t[1] += float(r["TotalDurationNs"])
            dur = {k: v[1] / max(1, v[0]) for k, v in tot.items()}
        for k in sorted(cs[names[0]]):
            n = cs[names[0]][k][0]
            busy, cu, mops, gui = (cs[q].get(k, [1, 0.0])[1] / max(1, cs[q].get(k, [1, 0.0])[0]) for q in names)
            if mops <= 0:
                continue
            row = {"launches_sampled": n, "mfma_busy_cycles_per_launch": busy, "cu_busy_cycles_per_launch": cu, "gui_active_per_launch": gui,
                   "mfma_mops_bf16_per_launch": mops, "bf16_gflop_per_launch": mops * 512 * 1e-9,
                   "mfma_busy_frac_of_cu_busy": busy / (4.0 * cu) if cu else None,
                   "mfma_busy_frac_of_chip": busy / (gui * 256 * 4) if gui else None}
            if k in dur:
                row["avg_duration_us"] = dur[k] * 1e-3
                row["bf16_tflops"] = mops * 512 / dur[k] * 1e-3
                row["frac_of_2500_tflops"] = row["bf16_tflops"] / 2500.0
            out[k] = row


if __name__ == "__main__":
    main()
