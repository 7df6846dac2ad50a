"""Compile every csrc/*.hip for gfx950 with -Rpass-analysis=kernel-resource-usage and print one line per kernel:
file, kernel, VGPRs, AGPRs, scratch bytes/lane, spilled VGPRs, waves/SIMD, LDS bytes.  CPU only (hipcc cross-compiles).
usage: python tools/kernel_resources.py [--spills-only] [file.hip ...]"""
import glob
import os
import re
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "diffusion-model_amd", "csrc")
PAT = re.compile(r"Function Name: (\S+).*?VGPRs: (\d+).*?AGPRs: (\d+).*?ScratchSize \[bytes/lane\]: (\d+).*?Occupancy \[waves/SIMD\]: (\d+).*?"
                 r"VGPRs Spill: (\d+).*?LDS Size \[bytes/block\]: (\d+)", re.S)


def one(src):
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I", CSRC, "-I", os.path.join(ROOT, "include"),
           "-c", src, "-o", os.devnull, "-Rpass-analysis=kernel-resource-usage"]
    err = subprocess.run(cmd, capture_output=True, text=True).stderr
    rows = []
    for m in PAT.finditer(err):
        name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
        rows.append((os.path.basename(src), re.sub(r"\(.*", "", name.replace("void ", "")),) + tuple(int(v) for v in m.groups()[1:]))
    return rows


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    srcs = [os.path.join(CSRC, a) for a in args] or sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    with ThreadPoolExecutor(4) as ex:
        rows = [r for rs in ex.map(one, srcs) for r in rs]
    print(f"{'file':20s} {'kernel':64s} VGPR AGPR scratch spill waves LDS")
    for r in rows:
        if "--spills-only" in sys.argv and r[4] == 0 and r[6] == 0:
            continue
        print(f"{r[0]:20s} {r[1][:64]:64s} {r[2]:4d} {r[3]:4d} {r[4]:7d} {r[6]:5d} {r[5]:5d} {r[7]}")
