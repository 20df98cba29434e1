import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from objectpermanence_amd import ModelsFactory, FusedAdam
from objectpermanence_amd.training import train_step
from oracle import synth
from tools.stack_time import CFG
name = "non_linear_lstm"; B = 32
m = ModelsFactory.get_model(name, CFG[name]).to("cuda:0").train(True)
opt = FusedAdam(m.parameters(), lr=1e-4)
b, l = synth.make_batch(0, 8, 300)
x = torch.from_numpy(np.tile(synth.boxes5(b), (4, 1, 1, 1))[:B].copy()).cuda()
y = torch.from_numpy(np.tile(l, (4, 1, 1))[:B].copy()).cuda()
for _ in range(7):
    train_step(name, m, opt, x, y)
torch.cuda.synchronize()
