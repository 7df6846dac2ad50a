"""64x64x3 training step with and without the fused backward of the 16x16 / 64-channel inverted-residual blocks (that
instantiation spills registers): python tools/micro/ab_sir_bwd_64_16.py [0|1]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from dm_amd.runtime import ops  # noqa: E402

if len(sys.argv) > 1 and sys.argv[1] == "1":
    orig = ops.sir_backward_supported
    ops.sir_backward_supported = lambda c, h, w, e, g, a: False if (c, w) == (64, 16) else orig(c, h, w, e, g, a)
sys.argv = ["bench.py", "--size", "64", "--channels", "3", "--batch", "128", "--no-sample", "--no-cpu", "--no-roofline", "--no-extra", "--steps", "30"]
bench.main()
