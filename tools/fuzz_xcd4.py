"""Randomised shape sweep of the 4-clip persistent kernels (reference hidden sizes) against the fp64 oracle / port: inference
y and logits, training loss and gradients, at odd batch sizes and sequence lengths (not part of the test suite; GPU box).
    SEED=0 CASES=40 python tools/fuzz_xcd4.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from objectpermanence_amd import ModelsFactory, l1_mean, _lib
from oracle import synth, opnet_oracle as oo, torch_port

CFG = {"object_to_track_pred_dim": 15, "object_to_track_hidden_dim": 256, "videos_hidden_dim": 512}
rng = np.random.default_rng(int(os.environ.get("SEED", "0")))
worst = {}


def note(k, v):
    worst[k] = max(worst.get(k, 0.0), float(v))


st = (_lib.ctypes.c_uint * 4)()
for case in range(int(os.environ.get("CASES", "40"))):
    B = int(rng.integers(1, 65))
    T = int(rng.choice([1, 2, 3, 5, 8, 13, 21, 34, 40]))
    p = synth.opnet_synth_params(CFG, salt=case)
    boxes, labels = synth.make_batch(5000 + case, B, T)
    m = ModelsFactory.get_model("opnet", CFG)
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in p.items()})
    m.eval().to("cuda:0")
    x = torch.from_numpy(boxes).cuda()
    with torch.no_grad():
        y, lg = m(x)
    torch.cuda.synchronize()
    assert m._wants_xcd4(B)
    _lib.load().opnet_xcd4_last_status(st)
    assert st[0] == 0, list(st)
    ry, rl = oo.opnet_forward(boxes, p, np.float64)
    note("inference_y", np.abs(y.cpu().numpy() - ry).max())
    note("inference_logits", np.abs(lg.cpu().numpy() - rl).max())
    if B <= 32 and B * T <= 1400:
        m.train(True)
        loss = l1_mean(m(x)[0], torch.from_numpy(labels).cuda())
        loss.backward()
        torch.cuda.synchronize()
        _lib.load().opnet_xcd4_last_status(st)
        assert st[0] == 0, list(st)
        rloss, rg = torch_port.loss_and_grads(boxes, labels, p, dtype=torch.float64)[:2]
        note("train_loss", abs(float(loss.detach()) - rloss))
        for k, prm in m.named_parameters():
            note("train_grad_rel", np.abs(prm.grad.cpu().numpy() - rg[k]).max() / max(1e-2, np.abs(rg[k]).max()))
    print(f"case {case}: B={B} T={T} ok", flush=True)
print("WORST", {k: f"{v:.3e}" for k, v in worst.items()})
