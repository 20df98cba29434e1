import os, sys, torch, numpy as np
sys.path.insert(0, os.getcwd())
from objectpermanence_amd import ModelsFactory
from synthdata import opnet as synth
CFG = {"object_to_track_pred_dim": 15, "object_to_track_hidden_dim": 256, "videos_hidden_dim": 512}
m = ModelsFactory.get_model("opnet", CFG)
m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in synth.opnet_synth_params(CFG).items()})
m = m.to("cuda:0").train(True)
boxes, _ = synth.make_batch(0, 32, 300)
x = torch.from_numpy(boxes).cuda()
def timed(fn, reps=5):
    for _ in range(2): fn()
    torch.cuda.synchronize(); best = 1e9
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); best = min(best, e0.elapsed_time(e1))
    return best
from objectpermanence_amd import l1_mean
_, labels = synth.make_batch(0, 32, 300)
lab = torch.from_numpy(labels).cuda()
def step():
    m.zero_grad(set_to_none=True)
    y, _ = m(x)
    l1_mean(y, lab).backward()
base = None
for dbg, name in [(0, "full"), (1, "no gather"), (2, "no history stores"), (8, "no head"), (16, "no products"), (17, "no products, no gather"), (4, "no cells (and so no dfb parts: head off too)"), (29, "nothing (loop, barriers, prefetches)")]:
    os.environ["OPNET_X4_DEBUG_BWD"] = str(dbg | (8 if dbg & 4 else 0))
    t = timed(step)
    base = base or t
    print(f"backward debug {dbg:2d} {name:28s}: fwd+loss+bwd {t:.3f} ms ({t - base:+.3f})", flush=True)
os.environ["OPNET_X4_DEBUG_BWD"] = "0"
for dbg, name in [(0, "full"), (1, "no gather"), (2, "no history stores"), (3, "no gather, no history"), (4, "no cells"), (8, "no head"), (16, "no products"), (17, "no products, no gather"), (31, "nothing")]:
    os.environ["OPNET_X4_DEBUG"] = str(dbg)
    print(f"debug {dbg:2d} {name:28s}: forward {timed(lambda: m(x)):.3f} ms", flush=True)
