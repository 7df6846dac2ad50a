"""The reference trainer's loop shape on the module surface, nothing else: python tools/micro/dropin_only.py [micro steps]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

dev = torch.device("cuda", 0)
print(bench.dropin_eager(dev, 1, 32, 256, micro_steps=int(sys.argv[1]) if len(sys.argv) > 1 else 30))
