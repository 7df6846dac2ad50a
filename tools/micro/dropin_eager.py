"""The reference trainer's loop shape on the drop-in modules (bench.py's dropin_eager leg) alone, for profiling:
python tools/micro/dropin_eager.py [micro_steps]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
print(bench.dropin_eager(torch.device("cuda", 0), 1, 32, 256, micro_steps=n))
