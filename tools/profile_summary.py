#!/usr/bin/env python3
"""Condense rocprofv3 output directories into the small files kept under profiles/.

  python tools/profile_summary.py stats  <dir with *_kernel_stats.csv>  profiles/rNN_kernel_stats.csv
  python tools/profile_summary.py traffic <FETCH_SIZE dir> <WRITE_SIZE dir> profiles/hbm_traffic.json

traffic: HBM bytes per launch and kernel family = (2 x FETCH_SIZE + WRITE_SIZE) x 1024.  rocprofv3 reports both
counters in KiB; on gfx950 FETCH_SIZE tallies the 128-byte requests of wide coalesced reads at 64 bytes, hence the
factor 2 on the read side.  Kernel families merge template instances (conv_mfma_kernel<4,1,1> -> conv_mfma_kernel)."""
import collections
import csv
import glob
import json
import os
import re
import shutil
import sys


def family(name: str) -> str:
    name = re.sub(r"^void ", "", name)
    m = re.match(r"_Z(\d+)", name)
    if m:
        n = int(m.group(1))
        return name[m.end():m.end() + n]
    return re.split(r"[<(]", name)[0].strip()


def counter(dirname: str, which: str):
    f = glob.glob(os.path.join(dirname, "**", "*_counter_collection.csv"), recursive=True)[0]
    acc = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != which:
            continue
        a = acc[family(r["Kernel_Name"])]
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
    else:
        raise SystemExit(__doc__)


if __name__ == "__main__":
    main()
