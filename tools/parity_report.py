"""Print parity statistics of the HIP path vs the reference-generated golden (real config)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from objectpermanence_amd import ModelsFactory
from oracle import synth, opnet_oracle as oo
cfg = {"object_to_track_pred_dim": 15, "object_to_track_hidden_dim": 256, "videos_hidden_dim": 512}
g = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "opnet_real.npz"))
params = synth.opnet_synth_params(cfg)
m = ModelsFactory.get_model("opnet", cfg)
m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in params.items()})
m.eval().to("cuda:0")
boxes, labels = synth.make_batch(0, 4, 300)
with torch.no_grad():
    y, lg = m(torch.from_numpy(boxes).cuda())
y = y.cpu().numpy(); lg = lg.cpu().numpy()
y64, lg64 = oo.opnet_forward(boxes, params, np.float64)
px, pxr = oo.postprocess_to_pixels(y), oo.postprocess_to_pixels(g["y"])
print("vs reference(torch fp32): max|dy| %.3e  last-5-frames %.3e  max|dlogits| %.3e  int-pixel flips %d / %d (max %d px)" % (
    np.abs(y - g["y"]).max(), np.abs(y[:, -5:] - g["y"][:, -5:]).max(), np.abs(lg - g["logits"]).max(),
    (px != pxr).sum(), px.size, np.abs(px - pxr).max()))
print("vs oracle fp64:           max|dy| %.3e ; reference fp32 vs oracle fp64: %.3e" % (np.abs(y - y64).max(), np.abs(g["y"] - y64).max()))
px64 = oo.postprocess_to_pixels(y64.astype(np.float32))
print("flips: hip vs fp64 %d ; reference vs fp64 %d" % ((px != px64).sum(), (pxr != px64).sum()))
