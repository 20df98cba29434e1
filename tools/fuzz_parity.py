"""Randomised shape sweep of the HIP reasoners against the numpy oracle (not part of the test suite; run on the GPU box):
OPNet inference + training gradients, the stacked-LSTM siblings, at odd batch sizes / sequence lengths / hidden sizes."""
import os, sys, itertools
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from objectpermanence_amd import ModelsFactory, l1_mean
from oracle import synth, opnet_oracle as oo, torch_port

rng = np.random.default_rng(int(os.environ.get("SEED", "0")))
worst = {}
def note(k, v):
    worst[k] = max(worst.get(k, 0.0), float(v))

n_cases = int(os.environ.get("CASES", "40"))
for case in range(n_cases):
    B = int(rng.choice([1, 2, 3, 15, 16, 17, 31, 32, 33, 47, 64, 65, 97, 129]))
    T = int(rng.choice([1, 2, 3, 4, 7, 16, 33]))
    H1 = int(rng.choice([16, 32, 48, 64, 112, 256]))
    H2 = int(rng.choice([16, 32, 80, 128, 512]))
    cfg = {"object_to_track_pred_dim": 15, "object_to_track_hidden_dim": H1, "videos_hidden_dim": H2}
    p = synth.opnet_synth_params(cfg, salt=case)
    boxes, labels = synth.make_batch(1000 + case, B, T)
    m = ModelsFactory.get_model("opnet", cfg)
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in p.items()})
    m.eval().to("cuda:0")
    x = torch.from_numpy(boxes).cuda()
    with torch.no_grad():
        y, lg = m(x)
    ry, rl = oo.opnet_forward(boxes, p, np.float64)
    note("opnet_y", np.abs(y.cpu().numpy() - ry).max())
    note("opnet_logits", np.abs(lg.cpu().numpy() - rl).max() / max(1.0, np.abs(rl).max()))
    if case % 3 == 0 and B * T <= 2200:
        m.train(True)
        loss = l1_mean(m(x)[0], torch.from_numpy(labels).cuda())
        loss.backward()
        rloss, rg = torch_port.loss_and_grads(boxes, labels, p, dtype=torch.float64)[:2]
        note("opnet_loss", abs(float(loss.detach()) - rloss))
        for k, prm in m.named_parameters():
            note("opnet_grad_rel", np.abs(prm.grad.cpu().numpy() - rg[k]).max() / max(1e-2, np.abs(rg[k]).max()))
    print(f"case {case}: B={B} T={T} H1={H1} H2={H2} ok", flush=True)
for case in range(n_cases // 2):
    B = int(rng.choice([1, 5, 16, 33, 70])); T = int(rng.choice([1, 3, 9, 20])); H = int(rng.choice([16, 48, 128]))
    for name, cfg, pf in (("baseline_lstm", {"videos_hidden_dim": H}, synth.baseline_lstm_synth_params),
                          ("non_linear_lstm", {"boxes_features_dim": int(rng.choice([16, 48])), "videos_hidden_dim": H}, synth.non_linear_lstm_synth_params)):
        p = pf(cfg, salt=case)
        boxes, labels = synth.make_batch(2000 + case, B, T)
        x5 = synth.boxes5(boxes)
        m = ModelsFactory.get_model(name, cfg)
        m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in p.items()})
        m.to("cuda:0").train(True)
        loss = l1_mean(m(torch.from_numpy(x5).cuda()), torch.from_numpy(labels).cuda())
        loss.backward()
        rloss, rg, ry = torch_port.sibling_loss_and_grads(name, x5, labels, p, dtype=torch.float64)
        note(name + "_loss", abs(float(loss.detach()) - rloss))
        for k, prm in m.named_parameters():
            note(name + "_grad_rel", np.abs(prm.grad.cpu().numpy() - rg[k]).max() / max(1e-2, np.abs(rg[k]).max()))
    print(f"sibling case {case}: B={B} T={T} H={H} ok", flush=True)
print("WORST", {k: f"{v:.3e}" for k, v in worst.items()})
