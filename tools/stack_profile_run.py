"""workload for rocprofv3: a few forwards of one sibling reasoner.  python tools/stack_profile_run.py <model> <B> [engine]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from objectpermanence_amd import ModelsFactory  # noqa: E402
from synthdata import opnet as synth  # noqa: E402
from tools.stack_time import CFG  # noqa: E402

name, B = sys.argv[1], int(sys.argv[2])
engine = sys.argv[3] if len(sys.argv) > 3 else "auto"
boxes, _ = synth.make_batch(0, min(B, 32), 300)
x = torch.from_numpy(np.tile(synth.boxes5(boxes), ((B + 31) // 32, 1, 1, 1))[:B]).cuda()
m = ModelsFactory.get_model(name, CFG[name]).eval().cuda()
if engine == "chain":
    m._runner.use_xcd = "0"
with torch.no_grad():
    for _ in range(10):
        m(x)
torch.cuda.synchronize()
