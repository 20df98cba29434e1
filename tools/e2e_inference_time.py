"""End-to-end rate of the inference driver from files: N synthetic <video>.pkl + <video>_bb.json ->
reasoning_inference_main -> predictions, with the native and with the numpy input encoder.
    python tools/e2e_inference_time.py [n_clips] [num_workers]"""
import json
import os
import pickle
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from objectpermanence_amd.inference_main import reasoning_inference_main  # noqa: E402
from synthdata import opnet as synth  # noqa: E402

CFG = {"object_to_track_pred_dim": 15, "object_to_track_hidden_dim": 256, "videos_hidden_dim": 512}
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8000
workers = int(sys.argv[2]) if len(sys.argv) > 2 else 8
with tempfile.TemporaryDirectory() as tmp:
    s, l = os.path.join(tmp, "s"), os.path.join(tmp, "l")
    os.mkdir(s); os.mkdir(l)
    raws = [synth.make_raw_video(i, "plain") for i in range(32)]
    for k in range(n):
        bb, lab, gt = raws[k % 32]
        with open(os.path.join(s, f"v{k:05d}.pkl"), "wb") as f:
            pickle.dump({"bb": bb, "labels": lab}, f, pickle.HIGHEST_PROTOCOL)
        with open(os.path.join(l, f"v{k:05d}_bb.json"), "w") as f:
            json.dump(gt, f)
    params = synth.opnet_synth_params(CFG)
    torch.save({k: torch.from_numpy(v) for k, v in params.items()}, os.path.join(tmp, "opnet.pth"))
    json.dump(CFG, open(os.path.join(tmp, "model.json"), "w"))
    json.dump({"batch_size": 16, "num_workers": workers, "device": "cuda:0", "model_path": os.path.join(tmp, "opnet.pth"),
               "videos_dir": "unused", "sample_dir": s, "labels_dir": l}, open(os.path.join(tmp, "infer.json"), "w"))
    res = {}
    # native file reader + encoder | pickle.load / json.load + native encoder | ... + numpy encoder | native again
    # ClipFileLoader (threads) | torch DataLoader (processes) over the native reader | ... over pickle.load | ... + numpy encoder
    for tag, pkl, enc, nat in (("native reader", "1", "1", "1"), ("DataLoader + native reader", "1", "1", "0"),
                               ("pickle + native encoder", "0", "1", "0"), ("pickle + numpy encoder", "0", "0", "0"),
                               ("native reader", "1", "1", "1")):
        os.environ["OPNET_NATIVE_PKL"], os.environ["OPNET_NATIVE_ENCODE"], os.environ["OPNET_NATIVE_LOADER"] = pkl, enc, nat
        t0 = time.perf_counter()
        out = reasoning_inference_main("opnet", os.path.join(tmp, "out"), os.path.join(tmp, "infer.json"), os.path.join(tmp, "model.json"),
                                       write_files=False)
        dt = time.perf_counter() - t0
        tm = out["timing"]
        res[tag] = tm["steady_clips_per_s"]
        print(f"{tag}: {n} clips, {workers} workers / threads, batch 16: {dt:.2f} s = {n / dt:.0f} clips/s; start-up {tm['startup_s']:.2f} s, "
              f"steady state {tm['steady_clips_per_s']:.0f} clips/s (mean IoU {out['mean_iou']:.4f})", flush=True)
    print(f"steady state: {res['native reader'] / res['pickle + native encoder']:.2f} x with the native file reader")
