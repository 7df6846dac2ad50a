"""Build libddpm_hip.so (gfx950 only) from csrc/*.hip with hipcc, in-tree.

    python diffusion-model_amd/build.py [--force] [-j N]

Each translation unit is compiled to build/<name>.o (skipped when up to date) and the objects are
linked into lib/libddpm_hip.so.  Nothing but hipcc and the HIP runtime is needed: the library has a
plain C ABI (include/ddpm_hip.h) and no dependency on torch.
"""
from __future__ import annotations

import argparse
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "build")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libddpm_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function", "-I", CSRC, "-I", os.path.join(os.path.dirname(HERE), "include")]


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, jobs: int = 8, verbose: bool = True) -> str:
    os.makedirs(OBJ, exist_ok=True)
    os.makedirs(LIBDIR, exist_ok=True)
    headers = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith(".h")]
    headers.append(os.path.join(os.path.dirname(HERE), "include", "ddpm_hip.h"))
    sources = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith(".hip")]
    objs, todo = [], []
    for s in sources:
        o = os.path.join(OBJ, os.path.basename(s)[:-4] + ".o")
        objs.append(o)
        if force or _stale(o, [s] + headers):
            todo.append((s, o))

    def cc(pair):
        s, o = pair
        cmd = [HIPCC] + FLAGS + ["-c", s, "-o", o]
        r = subprocess.run(cmd, capture_output=True, text=True)
        return s, r.returncode, r.stdout + r.stderr

    if todo:
        with ThreadPoolExecutor(max_workers=max(1, min(jobs, len(todo)))) as ex:
            for s, rc, out in ex.map(cc, todo):
                if verbose or rc:
                    print(f"[hipcc] {os.path.basename(s)} rc={rc}", file=sys.stderr)
                if out.strip() and (verbose or rc):
                    print(out, file=sys.stderr)
                if rc:
                    raise RuntimeError(f"hipcc failed on {s}")
    if force or todo or _stale(LIB, objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode:
            print(r.stdout + r.stderr, file=sys.stderr)
            raise RuntimeError("link failed")
        if verbose:
            print(f"[link] {LIB}", file=sys.stderr)
    return LIB


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--force", action="store_true")
    ap.add_argument("-j", type=int, default=8)
    a = ap.parse_args()
    print(build(a.force, a.j))
