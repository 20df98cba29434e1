import cProfile, pstats, json, os, pickle, sys, tempfile, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from objectpermanence_amd.inference_main import reasoning_inference_main
from synthdata import opnet as synth
CFG = {"object_to_track_pred_dim": 15, "object_to_track_hidden_dim": 256, "videos_hidden_dim": 512}
n, workers = 4000, 8
with tempfile.TemporaryDirectory() as tmp:
    s, l = os.path.join(tmp, "s"), os.path.join(tmp, "l")
    os.mkdir(s); os.mkdir(l)
    raws = [synth.make_raw_video(i, "plain") for i in range(32)]
    for k in range(n):
        bb, lab, gt = raws[k % 32]
        pickle.dump({"bb": bb, "labels": lab}, open(os.path.join(s, f"v{k:05d}.pkl"), "wb"), pickle.HIGHEST_PROTOCOL)
        json.dump(gt, open(os.path.join(l, f"v{k:05d}_bb.json"), "w"))
    params = synth.opnet_synth_params(CFG)
    torch.save({k: torch.from_numpy(v) for k, v in params.items()}, os.path.join(tmp, "opnet.pth"))
    json.dump(CFG, open(os.path.join(tmp, "model.json"), "w"))
    json.dump({"batch_size": 16, "num_workers": workers, "device": "cuda:0", "model_path": os.path.join(tmp, "opnet.pth"),
               "videos_dir": "unused", "sample_dir": s, "labels_dir": l}, open(os.path.join(tmp, "infer.json"), "w"))
    args = ("opnet", os.path.join(tmp, "out"), os.path.join(tmp, "infer.json"), os.path.join(tmp, "model.json"))
    reasoning_inference_main(*args, write_files=False)
    pr = cProfile.Profile(); pr.enable()
    out = reasoning_inference_main(*args, write_files=False)
    pr.disable()
    print(out["timing"])
    pstats.Stats(pr).sort_stats("cumulative").print_stats(35)
