#!/usr/bin/env python3
"""bench.py - OPNet inference throughput on MI355X (the BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W [--batch 32] [--engine xcd|chain] [--repeats 15]
    python bench.py --mode train | transformer | detect      (configs 2 / 5, 3, 4 - their own lines)

The timed region (exactly K steps between barrier + synchronize) is run --repeats times; `value` is the median.

One "step" = one pass of the hot path (OPNet.forward through libopnet_hip.so) over one batch of `--batch` synthetic CATER
clips (300 frames x 15 slots x 6 features, fp32) that is already resident in HBM, followed by the device-side
post-processing to int32 pixel boxes.  Steps are independent requests; how many are in flight is the engine's business:

  --engine xcd (default)  the steps are submitted to objectpermanence_amd.serving.ReasonerServer, which concatenates up to
                          `--inflight` pending batches (default: up to 32 = 1024 clips) into ONE per-XCD persistent forward (csrc/opnet_xcd_kernels.hip:
                          every XCD runs the 300-step recurrence of its own clips with all weights in registers);
  --engine chain          round 1's form: one hipGraph of T+3 step launches per batch, spread over S HIP streams.

With N > 1 every rank runs its own batches (clips are independent: weak scaling, no data-path collective inside the
forward) and the int32 predictions of every launch are all-gathered over RCCL.  `--gpus N` without a launcher re-executes
itself under torch.distributed.run (one rank per GPU) and fails loudly if the node has fewer GPUs.

Rank 0 prints ONE JSON line: BASELINE.json's metric (clips/s, whole job), plus
  roofline     - the dominant kernel against the roofline that bounds it: opnet_xcd_forward against the fp32 MFMA peak
                 (algorithmic FLOPs of the clips of a launch / the kernel's own duration, HIP events around every launch of
                 it on its stream); for --engine chain, opnet_step against the HBM streaming model of SURVEY.md 8-d4 with
                 the duration of ONE launch (single-stream pass), the multi-stream figure under `roofline_amortised`;
  roofline_hbm_model / whole_job_mfma_frac - the same run expressed in north_star's streaming-model bytes and as
                 clips/s x FLOPs/clip over the fp32 MFMA peak (wall clock, all overheads in);
  cpu_baseline - oracle/opnet_oracle.c (a C/OpenMP port of the reference algorithm) timed on this host's cores on a
                 bounded sample of the same workload, batch 8 / 16 / 32, N = 1 only.

Inputs and weights are seeded synthetic data from `synthdata/` (data only, shared with the tests).  `oracle/` is used
for exactly two things, both outside the timed region: the cpu_baseline leg and a parity assert of the HIP outputs
against it.  The measured path is libopnet_hip.so only.
"""
from __future__ import annotations

import argparse
import ctypes
import json
import os
import subprocess
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

CFG = {"object_to_track_pred_dim": 15, "object_to_track_hidden_dim": 256, "videos_hidden_dim": 512}
T_FRAMES = 300
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)
MFMA_F32_PEAK_TF = 157.3       # MI355X_MICROARCH.md: fp32-input MFMA = the fp32 vector rate
FLOP_PER_CLIP = 2 * 300 * 1_421_146   # 852.7 MFLOP per 300-frame clip (forward)
W_BYTES = 5_684_224            # all six fp32 weight tensors
STATE_BYTES_PER_CLIP = 12_672  # per clip per time step: x_t + h,c read + h,c write of both LSTMs (8-d4)


def _committed_pmc(stem):
    """the newest committed profiles/r<N>_<stem>.json (rocprofv3 --pmc passes reduced by tools/pmc_reduce.py / pmc_traffic.py)"""
    for rnd in (6, 5, 4, 3):
        path = os.path.join(REPO, "profiles", f"r{rnd}_{stem}.json")
        try:
            with open(path) as f:
                return json.load(f), f"profiles/r{rnd}_{stem}.json"
        except (OSError, ValueError):
            continue
    return None, None


def pmc_traffic(kernel, clips_per_launch):
    """`roofline.traffic`: HBM-side bytes per launch of `kernel`.  PMC counters cannot be sampled from inside the process
    that is being timed, so this is the figure of the committed rocprofv3 --pmc passes of this same command
    (profiles/r<N>_pmc_traffic.json: FETCH_SIZE / WRITE_SIZE corrected as MI355X_MICROARCH.md prescribes, per launch, keyed by
    kernel and clips per launch); null when no pass of this build and shape is committed."""
    data, src = _committed_pmc("pmc_traffic")
    rec = (data or {}).get(kernel, {}).get(str(int(clips_per_launch)))
    if not rec:
        return {"traffic": None}
    return {"traffic": rec["bytes_per_launch"], "traffic_source": f"{src} ({rec.get('note', 'rocprofv3 --pmc')})"}


def pmc_mfma_busy(kernel, stem="mfma_util_transformer"):
    """matrix-pipe busy fraction of `kernel` from the committed PMC pass of this command (SQ_VALU_MFMA_BUSY_CYCLES over
    (GRBM_GUI_ACTIVE / 8 XCDs) x 1024 SIMDs, tools/pmc_reduce.py mfma); nothing when no pass is committed"""
    data, src = _committed_pmc(stem)
    for name, rec in (data or {}).get("kernels", {}).items():
        if kernel in name and "mfma_util" in rec:
            return {"mfma_busy": rec["mfma_util"], "mfma_busy_source": src}
    return {}


def accuracy_block(dev):
    """mean-IoU / mAP@0.5 of the HIP path with TRAINED weights on 16 held-out synthetic clips, next to the
    values the reference's own model + ResultsAnalyzer produce for the same weights and clips
    (tests/golden/opnet_trained_*.npz; oracle/gen_golden.py).  Outside the timed region."""
    from objectpermanence_amd import ModelsFactory, metrics
    from synthdata import opnet as synth
    wpath = os.path.join(REPO, "tests", "golden", "opnet_trained_fp16.npz")
    epath = os.path.join(REPO, "tests", "golden", "opnet_trained_eval.npz")
    if not (os.path.exists(wpath) and os.path.exists(epath)):
        return {}
    w, g = np.load(wpath), np.load(epath)
    m = ModelsFactory.get_model("opnet", CFG)
    m.load_state_dict({k: torch.from_numpy(w[k].astype(np.float32)) for k in w.files})
    m.eval().to(dev)
    boxes, labels = synth.make_batch(int(g["first"]), int(g["n"]), T_FRAMES)
    with torch.no_grad():
        y, _ = m(torch.from_numpy(boxes).to(dev))
    pred_px, _, iou = metrics.postprocess_and_iou(y, torch.from_numpy(labels).to(dev))
    miou, map50 = metrics.mean_iou_and_map(iou)
    # the int32 pixel boxes (inference_main.py:219) against the reference's for the same weights and clips
    mism = int((pred_px.cpu().numpy() != g["pred_px"]).sum())
    return {"accuracy": {"weights": "OPNet trained on MI355X by this repo's training path (tools/train_synthetic.py), fp16-rounded",
                         "clips": int(g["n"]), "mean_iou": round(miou, 6), "map_0.5": round(map50, 6),
                         "int_pixel_mismatches": mism, "int_pixel_components": int(g["pred_px"].size),
                         "reference_mean_iou": round(float(g["video_mean_iou"].mean()), 6),
                         "reference_map_0.5": round(float(g["video_map50"].mean()), 6),
                         "max_abs_dy_vs_reference": float(np.abs(y.cpu().numpy() - g["y"]).max())}}


def inference_from_files_block(n_clips=4096, threads=12):
    """What a user of the inference driver gets (outside the timed region): `n_clips` synthetic <video>.pkl + <video>_bb.json on
    local disk -> reasoning_inference_main (native clip-file reader into pinned buffers, ReasonerServer, int32 post-process, IoU)
    at the reference's 12 workers (configs/inference_config.json: num_workers 12, batch_size 16).  cold: first pass after the page
    cache was dropped (null when this process may not drop it); warm: the best of the three passes after it (every pass listed)."""
    import pickle
    import tempfile
    from objectpermanence_amd.inference_main import reasoning_inference_main
    from synthdata import opnet as synth
    with tempfile.TemporaryDirectory() as tmp:
        sdir, ldir = os.path.join(tmp, "s"), os.path.join(tmp, "l")
        os.mkdir(sdir)
        os.mkdir(ldir)
        raws = [synth.make_raw_video(i, "plain") for i in range(32)]
        for k in range(n_clips):
            bb, lab, gt = raws[k % 32]
            with open(os.path.join(sdir, f"v{k:05d}.pkl"), "wb") as f:
                pickle.dump({"bb": bb, "labels": lab}, f, pickle.HIGHEST_PROTOCOL)
            with open(os.path.join(ldir, f"v{k:05d}_bb.json"), "w") as f:
                json.dump(gt, f)
        torch.save({k: torch.from_numpy(v) for k, v in synth.opnet_synth_params(CFG).items()}, os.path.join(tmp, "opnet.pth"))
        json.dump(CFG, open(os.path.join(tmp, "model.json"), "w"))
        json.dump({"batch_size": 16, "num_workers": threads, "device": "cuda:0", "model_path": os.path.join(tmp, "opnet.pth"),
                   "videos_dir": "unused", "sample_dir": sdir, "labels_dir": ldir}, open(os.path.join(tmp, "infer.json"), "w"))
        dropped = False
        try:
            os.sync()
            with open("/proc/sys/vm/drop_caches", "w") as f:
                f.write("3\n")
            dropped = True
        except OSError:
            pass
        rates = []
        for _ in range(4):
            t0 = time.perf_counter()
            res = reasoning_inference_main("opnet", os.path.join(tmp, "out"), os.path.join(tmp, "infer.json"),
                                           os.path.join(tmp, "model.json"), write_files=False)
            rates.append((n_clips / (time.perf_counter() - t0), res["timing"]["steady_clips_per_s"]))
    return {"inference_from_files": {"clips": n_clips, "threads": threads, "batch_size": 16,
                                     "clips_per_s_cold": round(rates[0][0], 1) if dropped else None,
                                     "clips_per_s_warm": round(max(r[0] for r in rates[1:]), 1),
                                     "steady_clips_per_s_warm": round(max(r[1] for r in rates[1:]), 1),
                                     "clips_per_s_by_pass": [round(r[0], 1) for r in rates],
                                     "what": "python -m objectpermanence_amd reasoning_inference from .pkl / _bb.json files on local disk, whole call "
                                             "(start-up included) and its steady state; the headline `value` is the model call alone with inputs in HBM"}}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=None,
                    help="clips per GPU per step (default 32, the BASELINE configs); frames per pass for --mode detect (default 16)")
    ap.add_argument("--global-batch", type=int, default=None,
                    help="--mode train: clips per step over ALL GPUs (BASELINE.json config 5 says 256): per-rank batch = global / N, "
                         "the line says \"scaling\": \"strong\" (default: --batch clips per GPU, weak scaling)")
    ap.add_argument("--engine", choices=["xcd", "chain"], default="xcd",
                    help="xcd = request batching into the per-XCD persistent forward (default); chain = one hipGraph of "
                         "step launches per batch on S streams (round 1)")
    ap.add_argument("--inflight", type=int, default=0,
                    help="--engine xcd: batches concatenated into one persistent launch (0 = all steps up to 1024 clips, "
                         "split evenly when there are more)")
    ap.add_argument("--launcher-selftest", action="store_true",
                    help="only exercise the rank launcher / rendezvous (gloo, no GPU work) and print the world size")
    ap.add_argument("--streams", type=int, default=0,
                    help="HIP streams the independent steps are spread over (step i runs on stream i %% S); "
                         "0 = calibrate S in {1,2,3,4} on untimed steps before the warm-up and keep the fastest")
    ap.add_argument("--mode", choices=["infer", "train", "detect", "transformer"], default="infer",
                    help="infer = the BASELINE metric (default); train = fwd + L1 + bwd + Adam step (configs 2/5); "
                         "transformer = transformer_lstm inference, one clip per step (config 3); detect = config 4's front-end")
    ap.add_argument("--exact", action="store_true", help="--mode transformer: serve on the lone request's kernels (at most 16 requests "
                                                       "per pass by default, results bit-identical to lone forwards) instead of the throughput form")
    ap.add_argument("--heads", type=int, default=4, help="--mode transformer: attention heads (BASELINE.json config 3 says 4; "
                                                         "configs/transformer_lstm_model_config.json ships 2)")
    ap.add_argument("--gather-every", type=int, default=16,
                    help="N > 1: steps of a stream whose predictions are all-gathered in ONE collective (fewer, larger messages)")
    ap.add_argument("--force-dist", action="store_true", help="initialise RCCL and run the N > 1 exchange path even with one rank")
    ap.add_argument("--loss", choices=["l1", "smooth_l1"], default="smooth_l1",
                    help="--mode train: BASELINE.json config 2 names SmoothL1; the reference's own training uses l1")
    ap.add_argument("--repeats", type=int, default=15,
                    help="the timed region (exactly --steps steps between barrier + synchronize) is run this many times; the line "
                         "reports the MEDIAN (value_min / value_max alongside): one 5 ms region is box-to-box noise")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="target wall time of the CPU baseline sample")
    return ap.parse_args()


def cpu_baseline(boxes_np, params, seconds):
    """Time the C/OpenMP port on this host at the batch sizes of SURVEY.md 8-d5 (8 / 16 / 32 clips, best of >= 5 after a
    warm-up).  Returns the dict for the JSON line (value = the bench's own batch) + the port's output for that batch."""
    from oracle import c_oracle
    threads = c_oracle.usable_cores()  # affinity mask capped by the cgroup CPU quota
    B = boxes_np.shape[0]
    sweep, y_full, t_used = {}, None, 0.0
    sizes = sorted({b for b in (8, 16, 32) if b <= B} | {B})
    for b in sizes:
        x = np.ascontiguousarray(boxes_np[:b])
        y, _ = c_oracle.opnet_forward(x, params, threads)           # warm-up (also page-in / build)
        best, reps, t_end = float("inf"), 0, time.perf_counter() + seconds / len(sizes)
        while reps < 5 or time.perf_counter() < t_end:
            t0 = time.perf_counter()
            y, _ = c_oracle.opnet_forward(x, params, threads)
            dt = time.perf_counter() - t0
            best, reps, t_used = min(best, dt), reps + 1, t_used + dt
        sweep[str(b)] = round(b / best, 2)
        if b == B:
            y_full = y
    return {
        "value": sweep[str(B)], "unit": "clips/s", "cores": threads, "kind": "port",
        "sample": f"best-of-n forward of {B} clips x {boxes_np.shape[1]} frames (oracle/opnet_oracle.c, gcc -O3 -march=native "
                  f"-fopenmp, {threads} threads, {t_used:.1f} s of CPU work in total)",
        "clips_per_s_by_batch": sweep,
    }, y_full


def collective_proof(dist, dev, world):
    """N > 1: what makes the line self-proving - the backend the group runs on and how many ranks one all-reduce of ones saw
    (every rank contributes 1.0 from ITS device).  Called by every rank, outside the timed region."""
    t = torch.ones(1, device=dev)
    dist.all_reduce(t)
    seen = int(round(float(t.item())))
    if seen != world or dist.get_world_size() != world:
        raise SystemExit(f"bench: --gpus {world} but the collective saw {seen} rank(s) (world size {dist.get_world_size()})")
    return {"backend": str(dist.get_backend()), "ranks_seen": seen, "world_size": dist.get_world_size()}


def bench_train(args, model, boxes, labels, world, rank, dev, dist, params):
    """Training throughput (BASELINE.json configs 2 / 5; not the headline metric): one step = forward + loss + backward +
    Adam on `--batch` clips per GPU, gradients all-reduced over RCCL when N > 1."""
    from objectpermanence_amd import FusedAdam
    from objectpermanence_amd.training import train_step
    model.train(True)
    opt = FusedAdam(model.parameters(), lr=1e-3)
    comm = torch.cuda.Stream(device=dev)
    B = int(boxes.shape[0])
    for _ in range(args.warmup):
        train_step("opnet", model, opt, boxes, labels, n_global=world * B, comm_stream=comm, loss_kind=args.loss)
    torch.cuda.synchronize(dev)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    comm_events, last = [], {}

    def region():
        ev0.record()
        for _ in range(args.steps):
            last["loss"] = train_step("opnet", model, opt, boxes, labels, n_global=world * B, comm_stream=comm,
                                      loss_kind=args.loss, comm_events=comm_events if world > 1 or args.force_dist else None)
        ev1.record()

    proof = collective_proof(dist, dev, world) if dist is not None else None
    elapsed, t_min, t_max, _all = timed_repeats(args, dev, dist, world, region)
    loss = last["loss"]
    gpu_ms = ev0.elapsed_time(ev1)             # the last repeat
    from objectpermanence_amd.training import step_aborted
    if step_aborted(model):
        raise SystemExit("bench: a persistent launch of a training step aborted; no line is printed for such a run")
    comm_ms = sum(a.elapsed_time(b) for a, b in comm_events) / max(len(comm_events), 1) if comm_events else None
    if rank == 0:
        from objectpermanence_amd import _lib
        x4_ok = (os.environ.get("OPNET_XCD4", "1") != "0"
                 and bool(_lib.load().opnet_xcd_supported(CFG["object_to_track_hidden_dim"], CFG["videos_hidden_dim"])))
        persistent = x4_ok and B <= int(os.environ.get("OPNET_XCD4_MAX_B", "32"))
        fwd_persistent = x4_ok and B <= int(os.environ.get("OPNET_XCD4_MAX_B", os.environ.get("OPNET_XCD4_FWD_MAX_B", "96")))
        fwd16 = (x4_ok and not fwd_persistent and os.environ.get("OPNET_XCD_TRAIN", "1") != "0" and 96 < B <= 1024
                 and B >= int(os.environ.get("OPNET_XCD_TRAIN_MIN_B", "97")))
        rbs = (B + 31) // 32
        nsl = int(os.environ.get("OPNET_BWD_SLICES", "-1"))
        nsl = min(4, rbs, (1 if rbs < 2 else 3 if rbs in (5, 6) else 4 if rbs > 12 else 2) if nsl < 0 else max(nsl, 1))
        bwd = "chain backward" if nsl < 2 or os.environ.get("OPNET_BWD_MODE") else f"backward as {nsl} chains over slices of the batch, side by side"
        engine = ("xcd4" if persistent else f"xcd4 forward + {bwd}" if fwd_persistent
                  else f"xcd (16-clip persistent) forward + {bwd}" if fwd16 else "chain")
        kernels = ("opnet_xcd4_forward + opnet_xcd4_backward (the whole recurrence as ONE persistent launch each, 4-clip groups "
                   "per XCD, weights resident in registers) + opnet_wgrad" if persistent else
                   "opnet_xcd_forward<.., true> (ONE persistent launch of 16-clip groups writing the histories) + opnet_bwd_fused "
                   "(one launch per reverse step and slice of the batch, the slices' chains on separate streams) + opnet_wgrad" if fwd16 else
                   "opnet_step / opnet_xcd4_forward + opnet_bwd_fused (one launch per reverse step and slice of the batch) + opnet_wgrad")
        # algorithmic bytes of one training step under the per-time-step streaming model: the
        # forward streams the weights once per step and moves each clip's state (+ the saved history: gates 4x, h, c per
        # unit), the backward streams W_hh^T / W_ih2^T / the heads once per reverse step and reads the history back
        hist = B * (256 + 512) * (4 + 1 + 1) * 4                     # gates, h, c of both LSTMs per clip per step
        per_t = 2 * (W_BYTES + B * STATE_BYTES_PER_CLIP) + 2 * hist
        alg = per_t * T_FRAMES + 2 * W_BYTES * 3                     # + weight gradients written, Adam read/write
        achieved = alg * args.steps / (gpu_ms * 1e-3) / 1e9
        cpu = None
        if world == 1 and not args.no_cpu_baseline:
            from oracle import torch_port
            torch.set_num_threads(max(1, min(os.cpu_count() or 1, 16)))
            cpu_model = torch_port.OPNetTorch(params)
            cpu_opt = torch.optim.Adam(cpu_model.parameters(), lr=1e-3)
            xb, lb = boxes.cpu(), labels.cpu()
            loss_fn = torch.nn.SmoothL1Loss() if args.loss == "smooth_l1" else torch.nn.L1Loss()

            def cpu_step():
                cpu_opt.zero_grad()
                loss_fn(cpu_model(xb), lb).backward()
                cpu_opt.step()

            cpu_step()                                                   # warm-up
            t1 = time.perf_counter()
            n_cpu = 0
            while time.perf_counter() - t1 < args.cpu_seconds:
                cpu_step()
                n_cpu += 1
            dt = time.perf_counter() - t1
            cpu = {"value": round(n_cpu * B / dt, 2), "unit": "clips/s", "cores": torch.get_num_threads(), "kind": "port",
                   "sample": f"{n_cpu} x (forward + {args.loss} + backward + Adam) of {B} clips x 300 frames, "
                             f"oracle/torch_port.OPNetTorch (the graph on torch's CPU LSTM op, fp32), {dt:.1f} s"}
        line = ({
            "metric": "CATER clips/sec OPNet training step (fwd + loss + bwd + Adam)",
            "value": round(world * B * args.steps / elapsed, 1), "unit": "clips/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "repeats": max(1, args.repeats), "value_is": "median over the repeats of the timed region",
            "value_min": round(world * B * args.steps / t_max, 1), "value_max": round(world * B * args.steps / t_min, 1),
            "higher_is_better": True, "scaling": "strong" if args.global_batch is not None else "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"opnet training (BASELINE.json config {5 if world > 1 or args.global_batch is not None else 2}), "
                                   + (f"global batch {args.global_batch} clips fixed over the GPUs, " if args.global_batch is not None else "")
                                   + f"batch={B} clips/GPU x 300 "
                                   f"frames x 15 slots (10 objects), {args.loss} bbox loss, Adam lr 1e-3"
                                   + (f", data parallel over {world} GPUs: one RCCL all-reduce of the flat {W_BYTES + 16} B gradient "
                                      "bucket per step on a side stream" if world > 1 else ""),
                       "global_batch": world * B, "parallelism": f"dp{world}", "loss": args.loss},
            # the gradient all-reduce alone: HIP events around the collective on the comm stream, mean per step
            "allreduce_ms_per_step": None if comm_ms is None else round(comm_ms, 4),
            "collective": None if proof is None else dict(proof, op="all_reduce (sum) of the flat gradient bucket, weighted n_local / n_global",
                                                          bytes_per_step=W_BYTES + 16, allreduce_ms_per_step=None if comm_ms is None else round(comm_ms, 4),
                                                          stream="side stream (comm), the optimiser step waits for it"),
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": None,
                         "kernel": kernels + "; algorithmic bytes of the whole step (north_star's per-time-step weight-streaming model) / GPU time",
                         "alg_bytes_per_step": alg},
            # forward 852.7 MFLOP per clip, backward twice that, over the fp32 MFMA peak
            "roofline_mfma": {"bound": "mfma", "achieved": round(3 * FLOP_PER_CLIP * B * args.steps / (gpu_ms * 1e-3) / 1e12, 2),
                              "peak": MFMA_F32_PEAK_TF, "unit": "TFLOP/s",
                              "frac": round(3 * FLOP_PER_CLIP * B * args.steps / (gpu_ms * 1e-3) / 1e12 / MFMA_F32_PEAK_TF, 4)},
            "engine": engine,
            "cpu_baseline": cpu,
            "final_loss": float(loss.item())})
    if dist is not None:
        dist.destroy_process_group()
    if rank == 0:
        emit(line)


def detector_flops_per_frame(h=800, w=1088, rois=1000, classes=193, frames_per_pass=None):
    """algorithmic FLOPs of one eval-mode fasterrcnn_resnet50_fpn call at the padded size: every conv / linear of the
    backbone, FPN, RPN head and box heads, 2 * M * N * K each.  frames_per_pass given: returns
    (algorithmic, executed, layers) - the stride-1 3 x 3 convs that the product runs as Winograd F(2 x 2, 3 x 3) at that pass size
    (objectpermanence_amd.detector._Conv._winograd) issue 1 / 2.25 of their direct-form MACs on the matrix pipe."""
    total, executed, wino_layers = 0, 0, 0
    if frames_per_pass is not None:
        from objectpermanence_amd.detector import _Conv
        wino_on = os.environ.get("OPDET_WINOGRAD", "1") != "0"

    def conv(hh, ww, cin, cout, k, s, p):
        nonlocal total, executed, wino_layers
        oh, ow = (hh + 2 * p - k) // s + 1, (ww + 2 * p - k) // s + 1
        fl = 2 * oh * ow * cout * k * k * cin
        total += fl
        if (frames_per_pass is not None and wino_on and k == 3 and s == 1 and cin % 16 == 0 and cin >= _Conv.WINO_MIN_CIN
                and frames_per_pass * ((hh + 1) // 2) * ((ww + 1) // 2) >= _Conv.WINO_MIN_TILES):
            executed += fl / 2.25
            wino_layers += 1
        else:
            executed += fl
        return oh, ow

    hh, ww = conv(h, w, 3, 64, 7, 2, 3)
    hh, ww = (hh - 1) // 2 + 1, (ww - 1) // 2 + 1
    inpl, feats = 64, []
    for li, (nb, pl) in enumerate(zip((3, 4, 6, 3), (64, 128, 256, 512)), 1):
        for b in range(nb):
            st = 2 if (b == 0 and li > 1) else 1
            if b == 0:
                conv(hh, ww, inpl, pl * 4, 1, st, 0)
            conv(hh, ww, inpl, pl, 1, 1, 0)
            h2, w2 = conv(hh, ww, pl, pl, 3, st, 1)
            conv(h2, w2, pl, pl * 4, 1, 1, 0)
            hh, ww, inpl = h2, w2, pl * 4
        feats.append((hh, ww, inpl))
    levels = []
    for fh, fw, c in feats:
        conv(fh, fw, c, 256, 1, 1, 0)
        conv(fh, fw, 256, 256, 3, 1, 1)
        levels.append((fh, fw))
    levels.append(((feats[3][0] - 1) // 2 + 1, (feats[3][1] - 1) // 2 + 1))
    for fh, fw in levels:
        conv(fh, fw, 256, 256, 3, 1, 1)
        conv(fh, fw, 256, 15, 1, 1, 0)
    head = 2 * rois * (12544 * 1024 + 1024 * 1024 + 1024 * classes * 5)
    total += head
    executed += head
    return total if frames_per_pass is None else (total, executed, wino_layers)


def _passes_in_flight():
    from objectpermanence_amd.detector import PASSES_IN_FLIGHT
    return max(1, int(os.environ.get("OPDET_IN_FLIGHT", str(PASSES_IN_FLIGHT))))


def _detector_passes(det, frames, dev, streams):
    """run(n): n passes over `frames`, len(streams) of them enqueued at a time on alternating streams (as
    preprocess_perception_main runs a video); returns the last pass's detections"""
    def run(n):
        waiting, out = [], None
        for k in range(n):
            with torch.cuda.stream(streams[k % len(streams)]):
                waiting.append(det.detect_batch_async(frames, dev))
            if len(waiting) >= len(streams):
                out = waiting.pop(0)()
        while waiting:
            out = waiting.pop(0)()
        return out
    return run


def detector_block(dev, nf=16, passes=6):
    """BASELINE.json config 4's front-end next to the headline (outside the timed region; `--mode detect` is the full
    measurement): `passes` calls of CaterObjectDetector.detect_batch on `nf` 240x320 frames, PASSES_IN_FLIGHT passes in flight.
    Parity of this stage is UNPINNED (no torchvision in the image: DESIGN.md section 11)."""
    from objectpermanence_amd.detector import CaterObjectDetector
    from synthdata import detector as sd
    t0 = time.perf_counter()
    params = {**sd.synth_backbone_params(), **sd.synth_head_params()}
    det = CaterObjectDetector(None)
    det.load_state_dict(params, dev)
    frames = [f for f in np.random.default_rng(0).integers(0, 256, size=(nf, 240, 320, 3), dtype=np.uint8)]
    streams = [torch.cuda.Stream(device=dev) for _ in range(_passes_in_flight())]

    run = _detector_passes(det, frames, dev, streams)

    run(len(streams) + 1)             # every stream has run a pass: its workspaces exist
    torch.cuda.synchronize(dev)
    t1 = time.perf_counter()
    out = run(passes)
    torch.cuda.synchronize(dev)
    dt = time.perf_counter() - t1
    one = [frames[0]]
    det.detect_batch(one, dev)
    torch.cuda.synchronize(dev)
    t2 = time.perf_counter()
    for _ in range(10):
        det.detect_batch(one, dev)
    torch.cuda.synchronize(dev)
    single = (time.perf_counter() - t2) / 10
    fl, fl_exec, n_wino = detector_flops_per_frame(frames_per_pass=nf)
    tf = fl * nf * passes / dt / 1e12
    return {"detector": {
        "workload": f"Faster-RCNN R50-FPN (193 classes) eval on 240x320 frames -> 800x1066, {nf} frames per pass, synthetic weights",
        "frames_per_s": round(nf * passes / dt, 1), "ms_per_pass": round(dt / passes * 1e3, 2),
        # config 4 end to end: a clip is 300 frames through this detector (the reasoner then costs ~7 us per clip: the headline)
        "clips_per_s_end_to_end_300_frames": round(nf * passes / dt / 300.0, 3),
        "single_frame_call": {"ms": round(single * 1e3, 2), "frames_per_s": round(1.0 / single, 1)},
        "kernel": "conv2d_nhwc_glds (dominant; all dense launches of a pass over the whole-pass time incl. selection stages)",
        "roofline": {"bound": "mfma", "achieved": round(tf, 2), "peak": MFMA_F32_PEAK_TF, "unit": "TFLOP/s",
                     "frac": round(tf / MFMA_F32_PEAK_TF, 4), "gflop_per_frame": round(fl / 1e9, 1),
                     # frac counts the DIRECT convolution's flops (the algorithm's); the matrix pipe issued fewer:
                     "executed_gflop_per_frame": round(fl_exec / 1e9, 1), "executed_frac": round(tf * fl_exec / fl / MFMA_F32_PEAK_TF, 4)},
        "winograd": {"layers_per_frame": n_wino, "form": "F(2 x 2, 3 x 3), fp32, stride-1 3 x 3 convs with >= 256 input channels",
                     "max_rel_error_vs_direct_conv": "2.2e-6 of max|y| (tools/probes/winograd_probe.hip); detections compared with the "
                                                     "oracle's in tests/test_detector_gpu.py as before", "switch": "OPDET_WINOGRAD=0"},
        "parity": "unpinned (checked against the build-authored oracle/detector_oracle.py only)",
        "detections_per_frame": [len(o["scores"]) for o in out][:4], "block_seconds": round(time.perf_counter() - t0, 1)}}


def bench_detect(args, world, rank, dev, dist):
    """Detector throughput (config 4's front-end, not the BASELINE headline): one step = CaterObjectDetector.detect_batch
    on `--batch` 240x320 frames per GPU (frames are independent: weak scaling, no collective)."""
    from objectpermanence_amd.detector import CaterObjectDetector
    from synthdata import detector as sd
    params = {**sd.synth_backbone_params(), **sd.synth_head_params()}
    det = CaterObjectDetector(None)
    det.load_state_dict(params, dev)
    nf = args.batch
    frames = [f for f in np.random.default_rng(rank).integers(0, 256, size=(nf, 240, 320, 3), dtype=np.uint8)]
    # PASSES_IN_FLIGHT passes enqueued at a time on alternating streams, as preprocess_perception_main runs a video: the
    # kernels of pass k overlap the conv GEMMs of pass k+1
    streams = [torch.cuda.Stream(device=dev) for _ in range(_passes_in_flight())]

    run = _detector_passes(det, frames, dev, streams)

    out = run(max(len(streams) + 1, args.warmup))
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    out = run(args.steps)
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize(dev)
    elapsed = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    if rank == 0:
        fl, fl_exec, n_wino = detector_flops_per_frame(frames_per_pass=nf)
        tf = fl * nf * args.steps / elapsed / 1e12
        cpu = None
        if world == 1 and not args.no_cpu_baseline:
            torch.set_num_threads(max(1, min(os.cpu_count() or 1, 16)))
            from oracle import detector_oracle as do          # cpu_baseline leg only
            t1 = time.perf_counter()
            do.detector_forward(frames[0], params, dtype=torch.float32)
            cpu = {"value": round(1.0 / (time.perf_counter() - t1), 3), "unit": "frames/s", "cores": torch.get_num_threads(),
                   "kind": "port", "sample": "1 frame through oracle/detector_oracle.py (torch fp32 convs + numpy selection stages)"}
        line = ({
            "metric": "detector frames/sec (Faster-RCNN R50-FPN 193 classes, 240x320 frame -> 800x1066, eval)",
            "value": round(world * nf * args.steps / elapsed, 1), "unit": "frames/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"detector, {nf} frames per pass per GPU, synthetic weights", "parallelism": f"dp{world}"},
            "roofline": {"bound": "mfma", "achieved": round(tf / world, 2), "peak": 157.3, "unit": "TFLOP/s",
                         "frac": round(tf / world / 157.3, 4), "traffic": None,
                         "kernel": "all dense launches of a pass (conv2d_nhwc_glds dominates); whole-step time incl. selection stages",
                         "gflop_per_frame": round(fl / 1e9, 1),
                         # frac counts the DIRECT convolution's flops (the algorithm's); with Winograd layers the pipe issued fewer:
                         "executed_gflop_per_frame": round(fl_exec / 1e9, 1), "executed_frac": round(tf / world * fl_exec / fl / 157.3, 4)},
            "winograd": {"layers_per_frame": n_wino, "form": "F(2 x 2, 3 x 3), fp32; results within 2.2e-6 of max|y| of the direct conv",
                         "switch": "OPDET_WINOGRAD=0"},
            "parity": "unpinned (checked against the build-authored oracle/detector_oracle.py only)",
            "cpu_baseline": cpu,
            "detections_per_frame": [len(o["scores"]) for o in out][:4]})
    if dist is not None:
        dist.destroy_process_group()
    if rank == 0:
        emit(line)


def _transformer_model(heads, dev):
    from objectpermanence_amd import ModelsFactory
    from synthdata import opnet as synth
    cfg = {"boxes_features_dim": 256, "num_attention_heads": heads, "num_attention_layers": 2, "num_lstm_layers": 2,
           "lstm_hidden_dim": 512}
    params = synth.transformer_lstm_synth_params(cfg)
    model = ModelsFactory.get_model("transformer_lstm", cfg)
    model.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in params.items()})
    return model.eval().to(dev), params


def _median_forward_ms(fn, dev, reps=11):
    fn()
    torch.cuda.synchronize(dev)
    times = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize(dev)
        times.append(e0.elapsed_time(e1))
    return sorted(times)[len(times) // 2]


STACK_KERNELS = {"x": ("seqx", "seqx_forward<64, 2> (4-clip groups: the latency form)"),
                 "t": ("seqt", "seqt_forward<2> (16-clip groups: the throughput form, layer-0 input product hoisted into one GEMM)")}


def _read_profile(lib):
    from objectpermanence_amd import _lib
    prof = {}
    for tag, name in ((1, "seqx"), (2, "attn"), (3, "seqt"), (4, "ffn")):
        kms, nl = ctypes.c_double(0.0), ctypes.c_int(0)
        _lib.check(lib.opnet_kernel_profile_read(tag, ctypes.byref(kms), ctypes.byref(nl)), "opnet_kernel_profile_read")
        prof[name] = (kms.value, nl.value)
    return prof


def transformer_serving(model, reqs, per_pass, dev, lib, rounds=12, exact=True):
    """`reqs` (independent transformer_lstm requests of one shape) through a ReasonerServer that merges up to `per_pass` of them
    into one pass (exact: on the lone request's kernels; else passes of 64 clips or more in the throughput form); returns
    (results of the last round, seconds per round by HIP events, persistent-stack / attention profile)."""
    from objectpermanence_amd.serving import ReasonerServer
    b = int(reqs[0].shape[0])
    server = ReasonerServer(model, "transformer_lstm", max_clips=per_pass * b, exact=exact)

    def serve(n_rounds):
        """every request of n_rounds rounds submitted before any result is asked for - requests are independent, and a result
        asked for makes the caller's stream (which the next pass's inputs come from) wait for its pass"""
        hs = [server.submit(x) for _ in range(n_rounds) for x in reqs]
        server.flush()
        return [h.result() for h in hs][-len(reqs):]

    with torch.no_grad():
        serve(min(4, rounds))        # (the server alternates two side streams: each packs its register image on first use)
        torch.cuda.synchronize(dev)
        lib.opnet_xcd_profile(1)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        outs = serve(rounds)
        e1.record()
        torch.cuda.synchronize(dev)
    prof = _read_profile(lib)
    lib.opnet_xcd_profile(0)
    return outs, e0.elapsed_time(e1) * 1e-3 / rounds, prof


E_TX, H_TX, FFN_TX = 256, 512, 2048
# algorithmic work per clip: the stacked LSTM 2 x T x (4H (E + H) + 4H (H + H)) flop; the live encoder (slot 0
# only) per layer S (E 3E + E E + 2 E FFN) MAC + attention 4 S^2 E flop
TX_STACK_FLOP = 2 * T_FRAMES * (4 * H_TX * (E_TX + H_TX) + 4 * H_TX * (H_TX + H_TX))
TX_ENC_FLOP = 2 * (2 * T_FRAMES * (E_TX * 3 * E_TX + E_TX * E_TX + 2 * E_TX * FFN_TX) + 4 * T_FRAMES * T_FRAMES * E_TX)


def _stack_roofline(prof, clips_per_launch):
    """the persistent stacked-LSTM launch of a pass: which form ran, its time by HIP events, its share of the fp32 MFMA peak.
    The throughput form's launch does not contain the layer-0 input product (hoisted into a GEMM before it): its flop are the
    recurrent ones, 2 T x 4H (H + 2H)."""
    form = "t" if prof["seqt"][1] else ("x" if prof["seqx"][1] else None)
    if form is None:
        return None
    ms = prof[STACK_KERNELS[form][0]][0] / prof[STACK_KERNELS[form][0]][1]
    flop = TX_STACK_FLOP if form == "x" else 2 * T_FRAMES * 4 * H_TX * 3 * H_TX
    tf = clips_per_launch * flop / (ms * 1e-3) / 1e12
    return {"bound": "mfma", "achieved": round(tf, 2), "peak": MFMA_F32_PEAK_TF, "unit": "TFLOP/s", "frac": round(tf / MFMA_F32_PEAK_TF, 4),
            "kernel": STACK_KERNELS[form][1], "launch_ms": round(ms, 4), "launches": prof[STACK_KERNELS[form][0]][1],
            "clips_per_launch": int(clips_per_launch), "alg_flop_per_launch": int(clips_per_launch * flop)}


def transformer_block(dev, heads=4, n_exact=16, n_tp=256):
    """BASELINE.json config 3 next to the headline (outside the timed region; `--mode transformer` is the full measurement): one
    one-clip request alone; `n_exact` independent one-clip requests served per pass on the lone request's kernels (bit-identical
    results); `n_tp` per pass in the throughput form (large GEMM tiles, 16-clip persistent stack launch).  And the stacked LSTM of
    baseline_lstm at 256 clips (the per-epoch evaluation's call pattern) in both of its persistent forms."""
    from objectpermanence_amd import ModelsFactory, _lib
    from synthdata import opnet as synth
    lib = _lib.load()
    model, _ = _transformer_model(heads, dev)
    base = torch.from_numpy(synth.boxes5(synth.make_batch(5000, 64, T_FRAMES)[0])).to(dev)
    reqs = [base[i:i + 1].contiguous() for i in range(64)]
    with torch.no_grad():
        lone_ms = _median_forward_ms(lambda: model(reqs[0]), dev)
        lone = model(reqs[0]).clone()
    outs, sec, prof = transformer_serving(model, reqs[:n_exact], n_exact, dev, lib, exact=True)
    out = {"workload": f"transformer_lstm (d_model 256, {heads} heads, 2 encoder layers, 2 x LSTM 512), seq_len 300 per request",
           "flop_per_clip": {"lstm_stack": TX_STACK_FLOP, "encoder_live": TX_ENC_FLOP},
           "mfma_bound_clips_per_s": round(MFMA_F32_PEAK_TF * 1e12 / (TX_STACK_FLOP + TX_ENC_FLOP), 1),
           "lone_request": {"ms_per_forward": round(lone_ms, 3), "clips_per_s": round(1e3 / lone_ms, 1)},
           "served": {"requests_per_pass": n_exact, "ms_per_pass": round(sec * 1e3, 3), "clips_per_s": round(n_exact / sec, 1),
                      "bit_identical_to_lone_forward": bool(torch.equal(outs[0], lone)), "roofline": _stack_roofline(prof, n_exact)}}
    if model.max_requests_per_pass(1, T_FRAMES) >= n_tp:
        big = reqs * (2 * n_tp // 64)                                                  # two passes per round
        outs, sec, prof = transformer_serving(model, big, n_tp, dev, lib, rounds=3, exact=False)
        cps = len(big) / sec
        out["served_throughput_form"] = {
            "requests_per_pass": n_tp, "ms_per_pass": round(sec * 1e3 / 2, 3), "clips_per_s": round(cps, 1),
            "max_abs_diff_vs_lone_forward": float((outs[0] - lone).abs().max()),
            "whole_job_mfma_frac": round(cps * (TX_STACK_FLOP + TX_ENC_FLOP) / (MFMA_F32_PEAK_TF * 1e12), 4),
            "roofline": _stack_roofline(prof, n_tp)}
    # baseline_lstm, one forward of 256 clips: throughput form (what the model picks) and the 4-clip latency form
    cfg = {"videos_hidden_dim": 512}
    bl = ModelsFactory.get_model("baseline_lstm", cfg)
    bl.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in synth.baseline_lstm_synth_params(cfg).items()})
    bl.eval().to(dev)
    x256 = torch.cat([base] * 4).contiguous()
    sib = {"workload": "baseline_lstm (LSTM 75 -> 512 + Linear 512 -> 4), one forward of 256 clips x 300 frames"}
    bflop = 2 * T_FRAMES * 4 * H_TX * (80 + H_TX)
    for form, tag, key in (("auto", 3, "throughput_form"), ("0", 1, "latency_form")):
        bl._runner.use_xcdt = form
        with torch.no_grad():
            ms = _median_forward_ms(lambda: bl(x256), dev, reps=7)
            lib.opnet_xcd_profile(1)
            bl(x256)
            torch.cuda.synchronize(dev)
        kms, nl = ctypes.c_double(0.0), ctypes.c_int(0)
        _lib.check(lib.opnet_kernel_profile_read(tag, ctypes.byref(kms), ctypes.byref(nl)), "opnet_kernel_profile_read")
        lib.opnet_xcd_profile(0)
        sib[key] = {"ms_per_forward": round(ms, 3), "clips_per_s": round(256e3 / ms, 1), "launches": nl.value,
                    "kernel_ms": round(kms.value, 4),
                    "mfma_frac": round(256 * bflop / (kms.value * 1e-3) / 1e12 / MFMA_F32_PEAK_TF, 4) if kms.value > 0 else None}
    out["baseline_lstm_256_clips"] = sib
    return {"transformer_step": out}


def _train_step_ms(name, model, x, y, dev, lib, warm=3, reps=9):
    """median ms of train_step (forward + L1 + backward + Adam) by HIP events + the profiled kernels' time of ONE step"""
    from objectpermanence_amd import FusedAdam
    from objectpermanence_amd.training import step_aborted, train_step
    opt = FusedAdam(model.parameters(), lr=1e-4)
    for _ in range(warm):
        train_step(name, model, opt, x, y)
    torch.cuda.synchronize(dev)
    times = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        loss = train_step(name, model, opt, x, y)
        e1.record()
        torch.cuda.synchronize(dev)
        times.append(e0.elapsed_time(e1))
    lib.opnet_xcd_profile(1)
    train_step(name, model, opt, x, y)
    torch.cuda.synchronize(dev)
    prof = {}
    for tag, key in ((1, "seqx_forward"), (7, "seqx_backward"), (5, "attention_forward"), (6, "attention_backward")):
        kms, nl = ctypes.c_double(0.0), ctypes.c_int(0)
        lib.opnet_kernel_profile_read(tag, ctypes.byref(kms), ctypes.byref(nl))
        prof[key] = (kms.value, nl.value)
    lib.opnet_xcd_profile(0)
    if step_aborted(model):
        raise SystemExit("bench: a persistent launch of a sibling training step aborted")
    return sorted(times)[len(times) // 2], float(loss), prof


def siblings_training_block(dev, heads=2):
    """Training steps of the stacked-LSTM reasoners next to the headline (outside the timed region): transformer_lstm (BASELINE.json
    config 3's model; `heads` = configs/transformer_lstm_config.json's 2) and baseline_lstm at the batches 1, the reference's 16
    (configs/training_config.json) and 32.  One step = forward + L1 + backward + Adam (training_main.py:183-217), dropout 0.1 live
    in the encoder.  FLOP per step counted from the model (forward: SURVEY.md 8-d4; backward: twice the forward, + the attention
    backward's recomputed score and dP tiles: 7 instead of 4 products of S^2 E MACs per layer)."""
    from objectpermanence_amd import ModelsFactory, _lib
    from synthdata import opnet as synth
    lib = _lib.load()
    base_b, base_l = synth.make_batch(7000, 32, T_FRAMES)
    xb = torch.from_numpy(synth.boxes5(base_b)).to(dev)
    yb = torch.from_numpy(base_l).to(dev)
    # ---- transformer_lstm ----
    tx = {"workload": f"transformer_lstm (d_model 256, {heads} heads, 2 encoder layers, 2 x LSTM 512) training step: forward + L1 + "
                      "backward + Adam, dropout 0.1; one minibatch = ONE sequence of B x 300 tokens (the reference's sequence-first encoder)",
          "batches": {}}
    for B in (1, 16, 32):
        S = B * T_FRAMES
        torch.cuda.synchronize(dev)
        torch.cuda.empty_cache()
        torch.cuda.reset_peak_memory_stats(dev)
        held = torch.cuda.memory_allocated(dev)          # (what the earlier blocks of this process still hold)
        model, _ = _transformer_model(heads, dev)        # a fresh model per batch size: its workspaces are this batch's
        model.train(True)
        ms, loss, prof = _train_step_ms("transformer_lstm", model, xb[:B].contiguous(), yb[:B].contiguous(), dev, lib)
        tok = 2 * S * (E_TX * 3 * E_TX + E_TX * E_TX + 2 * E_TX * FFN_TX) * 2           # token-wise products, both layers (flop)
        att_u = 2 * S * S * E_TX                                                        # one S x S x E product over all heads (flop)
        fwd = tok + 2 * 2 * att_u + B * TX_STACK_FLOP
        bwd = 2 * tok + 2 * 7 * att_u + 2 * B * TX_STACK_FLOP
        a_f, a_b = prof["attention_forward"], prof["attention_backward"]
        blk = {"ms_per_step": round(ms, 3), "clips_per_s": round(B / ms * 1e3, 1), "tokens": S, "loss": round(loss, 5),
               "flop_per_step": int(fwd + bwd),
               "mfma_frac": round((fwd + bwd) / (ms * 1e-3) / 1e12 / MFMA_F32_PEAK_TF, 4),
               "kernel_ms": {k: round(v[0], 4) for k, v in prof.items()}, "kernel_launches": {k: v[1] for k, v in prof.items()},
               "peak_device_bytes": int(torch.cuda.max_memory_allocated(dev) - held)}
        if a_b[0] > 0:      # the dominant kernels: the attention backward passes of the two layers (prep + dQ pass + dK / dV pass)
            tf = 2 * 7 * att_u / (a_b[0] * 1e-3) / 1e12
            blk["roofline"] = {"bound": "mfma", "achieved": round(tf, 2), "peak": MFMA_F32_PEAK_TF, "unit": "TFLOP/s",
                               "frac": round(tf / MFMA_F32_PEAK_TF, 4),
                               "kernel": "attention_bwd<128, false> (dQ) + attention_bwd<128, true> (dK, dV) of both layers, HIP events around "
                                         "each layer's launches", "alg_flop": int(2 * 7 * att_u)}
            blk["attention_forward_mfma_frac"] = round(2 * 2 * att_u / (a_f[0] * 1e-3) / 1e12 / MFMA_F32_PEAK_TF, 4) if a_f[0] > 0 else None
        tx["batches"][str(B)] = blk
        del model
    # ---- baseline_lstm ----
    cfg = {"videos_hidden_dim": 512}
    bl = ModelsFactory.get_model("baseline_lstm", cfg)
    bl.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in synth.baseline_lstm_synth_params(cfg).items()})
    bl.to(dev).train(True)
    bflop = 2 * T_FRAMES * 4 * H_TX * (80 + H_TX)
    bs = {"workload": "baseline_lstm (LSTM 75 -> 512 + Linear 512 -> 4) training step: forward + L1 + backward + Adam", "batches": {}}
    for B in (1, 16, 32):
        ms, loss, prof = _train_step_ms("baseline_lstm", bl, xb[:B].contiguous(), yb[:B].contiguous(), dev, lib)
        f, b_ = prof["seqx_forward"], prof["seqx_backward"]
        blk = {"ms_per_step": round(ms, 3), "clips_per_s": round(B / ms * 1e3, 1), "loss": round(loss, 5),
               "flop_per_step": int(3 * B * bflop), "mfma_frac": round(3 * B * bflop / (ms * 1e-3) / 1e12 / MFMA_F32_PEAK_TF, 4),
               "kernel_ms": {"seqx_forward": round(f[0], 4), "seqx_backward": round(b_[0], 4)}}
        if b_[0] > 0:
            # the reverse recurrence: W_hh^T da per step = 2 x 4H x H flop per clip; latency-bound by construction (one 4-clip group
            # per XCD, 300 dependent steps with an exchange between the XCD's 32 CUs each)
            tf = B * 2 * T_FRAMES * 4 * H_TX * H_TX / (b_[0] * 1e-3) / 1e12
            blk["roofline"] = {"bound": "mfma", "achieved": round(tf, 3), "peak": MFMA_F32_PEAK_TF, "unit": "TFLOP/s",
                               "frac": round(tf / MFMA_F32_PEAK_TF, 5), "kernel": "seqx_backward<1> (one persistent launch per backward)",
                               "us_per_reverse_step": round(b_[0] * 1e3 / T_FRAMES, 3)}
        bs["batches"][str(B)] = blk
    return {"transformer_training": tx, "baseline_lstm_training": bs}


def bench_transformer(args, world, rank, dev, dist):
    """BASELINE.json config 3: transformer_lstm (d_model 256, `--heads` heads, 2 encoder layers, 2 LSTM layers of 512).  One step
    = one REQUEST of `--batch` clips (default ONE clip: seq_len S = 300 - the reference's sequence-first encoder attends over
    all B x 300 frames of a call, so a request's clip count IS its sequence length).  The requests are independent; they are
    submitted to a serving.ReasonerServer, which merges up to `--inflight` pending requests into one pass - token-wise stages
    over all tokens, attention inside a request, ONE persistent stacked-LSTM launch over all clips.  Default (`--inflight` 256):
    the throughput form (large GEMM tiles, 16-clip column groups; every request agrees with its lone forward to rounding, checked
    below); `--exact`: at most 16 per pass on the lone request's kernels, every result bit-identical to the lone forward
    (checked).  Weak scaling over ranks, no collective."""
    from objectpermanence_amd import _lib
    from objectpermanence_amd.serving import ReasonerServer
    from synthdata import opnet as synth
    lib = _lib.load()
    B = args.batch or 1
    exact = bool(args.exact)
    model, params = _transformer_model(args.heads, dev)
    per_pass = max(1, args.inflight or (16 if exact else 256))
    per_pass = min(per_pass, model.max_requests_per_pass(B, T_FRAMES, exact))
    # DISTINCT clips in every request of the timed region (up to 256 requests; a longer region cycles through them)
    n_req = args.steps
    n_dist = min(n_req, 256)
    req_np = [synth.boxes5(synth.make_batch((rank * n_dist + i) * B, B, T_FRAMES)[0]) for i in range(n_dist)]
    req_dev = [torch.from_numpy(a).to(dev) for a in req_np]
    reqs = [req_dev[i % n_dist] for i in range(n_req)]
    server = ReasonerServer(model, "transformer_lstm", max_clips=per_pass * B, exact=exact)
    last = {}

    def region():
        with torch.no_grad():
            hs = [server.submit(x) for x in reqs]
            server.flush()
            last["y"] = [h.result() for h in hs]

    for _ in range(max(1, (args.warmup + n_req - 1) // n_req)):
        region()
    torch.cuda.synchronize(dev)
    lib.opnet_xcd_profile(1)
    f0 = server.forwards
    elapsed, t_min, t_max, _all = timed_repeats(args, dev, dist, world, region)
    prof = _read_profile(lib)
    lib.opnet_xcd_profile(0)
    from objectpermanence_amd.launch_monitor import verify_launches
    if verify_launches(model):
        raise SystemExit("bench: a persistent launch aborted; no line is printed for such a run")
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank != 0:
        return
    R = max(1, args.repeats)
    passes = (server.forwards - f0) // R
    S, E = B * T_FRAMES, E_TX
    clips_per_s = world * B * n_req / elapsed
    # every DISTINCT request's served result against its LONE forward (the reference's call pattern)
    with torch.no_grad():
        lone_ms = _median_forward_ms(lambda: model(reqs[0]), dev)
        diffs = [float((model(req_dev[i]) - last["y"][i]).abs().max()) for i in range(n_dist)]
    identical = max(diffs) == 0.0
    stack_flop = B * TX_STACK_FLOP                                                   # per request
    attn_flop = 4 * S * S * E                                                      # one attention call of ONE request
    enc_flop = 2 * (2 * S * (E * 3 * E + E * E + 2 * E * FFN_TX) + attn_flop)
    req_per_launch = n_req / max(passes, 1)
    form = "t" if prof["seqt"][1] else ("x" if prof["seqx"][1] else None)
    attn_ms = prof["attn"][0] / max(prof["attn"][1], 1)
    line = {
        "metric": "CATER clips/sec transformer_lstm inference (BASELINE.json config 3)",
        "value": round(clips_per_s, 1), "unit": "clips/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 4),
        "repeats": R, "value_is": "median over the repeats of the timed region",
        "value_min": round(world * B * n_req / t_max, 1), "value_max": round(world * B * n_req / t_min, 1),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"transformer_lstm inference, one step = one independent request of {B} clip(s) x 300 frames = one "
                               f"sequence of S = {S} tokens (seq_len 300 per clip), d_model 256, {args.heads} heads, 2 encoder layers "
                               "(FFN 2048), 2 LSTM layers of 512, slot-0 path (exact: slots 1..14 never reach the output), inputs "
                               f"resident in HBM; a ReasonerServer merges up to {per_pass} pending requests into one pass "
                               "(segmented attention, one persistent stacked-LSTM launch) - " +
                               ("on the lone request's kernels, every result bit-identical to the request's lone forward" if exact else
                                "in the throughput form (token-wise products on large tiles, 16-clip column groups): every result "
                                "agrees with the request's lone forward to rounding"),
                   "global_batch": world * B, "frames": T_FRAMES, "parallelism": f"dp{world}", "heads": args.heads,
                   "requests_per_pass": per_pass, "passes": passes, "form": "exact" if exact else "throughput",
                   "distinct_requests": n_dist,
                   "weights": "synthetic (synthdata/opnet.py counter RNG), fp32",
                   "engine": {"t": "stacked LSTM as one persistent launch of 16-clip groups (seqt_forward)",
                              "x": "stacked LSTM as one persistent launch of 4-clip groups (seqx_forward)",
                              None: "launch per time step"}[form]},
        "served_results_bit_identical_to_lone_forward": bool(identical),
        "served_vs_lone_max_abs_diff": max(diffs), "requests_checked": n_dist,
        "lone_request": {"ms_per_forward": round(lone_ms, 3), "clips_per_s": round(B * 1e3 / lone_ms, 1),
                         "note": "one request alone (the reference's call pattern): a latency, 2 of 8 XCDs busy"},
        "flop_per_request": {"lstm_stack": stack_flop, "encoder_live": enc_flop},
        "mfma_bound_clips_per_s": round(B * MFMA_F32_PEAK_TF * 1e12 / (stack_flop + enc_flop), 1),
        "whole_job_mfma_frac": round(clips_per_s / world / B * (stack_flop + enc_flop) / (MFMA_F32_PEAK_TF * 1e12), 4),
    }
    if exact and not identical:
        raise SystemExit("bench: a served request differs from its lone forward")
    if not max(diffs) < 2e-5:
        raise SystemExit(f"bench: a served request differs from its lone forward by {max(diffs)}")
    if form is not None:
        roof = _stack_roofline(prof, req_per_launch * B)
        kname = "seqt_forward" if form == "t" else "seqx_forward"
        roof.update(pmc_traffic(kname, int(req_per_launch * B)))
        roof.update(pmc_mfma_busy(kname))
        roof["us_per_time_step"] = round(roof["launch_ms"] * 1e3 / T_FRAMES, 3)
        roof["timing"] = ("HIP events around every launch of the kernel on its stream (opnet_xcd_profile); two passes are in flight "
                          "(serving.py), so a launch may share the chip with the other pass's encoder kernels")
        line["roofline"] = roof
        # north_star's per-time-step weight-streaming model of the same recurrence: all LSTM weights once per step per launch
        w_bytes = 4 * (4 * H_TX * (E + H_TX) + 4 * H_TX * (H_TX + H_TX))
        model_gbs = T_FRAMES * (w_bytes + req_per_launch * B * 4 * (E + 4 * H_TX * 2)) / (roof["launch_ms"] * 1e-3) / 1e9
        line["roofline_hbm_model"] = {"bound": "hbm", "achieved": round(model_gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                      "frac": round(model_gbs / HBM_PEAK_GBS, 4),
                                      "note": "SURVEY.md 8-d4 streaming-model bytes (14.7 MB of LSTM weights once per time step) "
                                              "over the kernel time; the kernel itself reads the weights once per launch"}
    if prof["attn"][1] > 0:
        tfa = req_per_launch * attn_flop / (attn_ms * 1e-3) / 1e12
        line["roofline_attention"] = {"bound": "mfma", "achieved": round(tfa, 3), "peak": MFMA_F32_PEAK_TF, "unit": "TFLOP/s",
                                      "frac": round(tfa / MFMA_F32_PEAK_TF, 4), "kernel": "attention_glds (one call = the segments of "
                                      "all requests of a pass)", "call_ms": round(attn_ms, 4), "calls": prof["attn"][1],
                                      "alg_flop_per_call": int(req_per_launch * attn_flop)}
    if prof.get("ffn", (0, 0))[1] > 0:
        # linear1 -> ReLU -> linear2 of an encoder layer as one kernel (csrc/ffn_kernels.hip): 2 x 2 x E x FFN flop per token
        ffn_ms = prof["ffn"][0] / prof["ffn"][1]
        ffn_flop = req_per_launch * B * T_FRAMES * 4 * E * FFN_TX
        tff = ffn_flop / (ffn_ms * 1e-3) / 1e12
        line["roofline_feed_forward"] = {"bound": "mfma", "achieved": round(tff, 3), "peak": MFMA_F32_PEAK_TF, "unit": "TFLOP/s",
                                         "frac": round(tff / MFMA_F32_PEAK_TF, 4), "kernel": "ffn_fused_w8 (one launch = the feed-forward block of one "
                                         "encoder layer over all token rows of a pass; the [rows][2048] activations stay in LDS)",
                                         "launch_ms": round(ffn_ms, 4), "launches": prof["ffn"][1], "alg_flop_per_launch": int(ffn_flop),
                                         **pmc_mfma_busy("ffn_fused_w8")}
    # coupled minibatches (the reference's training / evaluation call: ONE sequence of S = b x 300): one forward each
    extra = {}
    for b in (16, 32):
        bx, _ = synth.make_batch(0, b, T_FRAMES)
        xb = torch.from_numpy(synth.boxes5(bx)).to(dev)
        with torch.no_grad():
            ms = _median_forward_ms(lambda: model(xb), dev, reps=5)
        extra[str(b)] = {"ms_per_forward": round(ms, 3), "clips_per_s": round(b / ms * 1e3, 1), "S": b * T_FRAMES}
    line["coupled_minibatches"] = extra
    if world == 1 and not args.no_cpu_baseline:
        from oracle import torch_port                                   # cpu_baseline leg + parity of the last output
        threads = max(1, min(os.cpu_count() or 1, 16))
        torch.set_num_threads(threads)
        pt = {k: torch.from_numpy(v) for k, v in params.items()}
        xt = torch.from_numpy(req_np[(n_req - 1) % n_dist])
        with torch.no_grad():
            y_cpu = torch_port.transformer_lstm_forward(xt, pt, args.heads)
            t1, n_cpu = time.perf_counter(), 0
            while time.perf_counter() - t1 < args.cpu_seconds or n_cpu < 3:
                torch_port.transformer_lstm_forward(xt, pt, args.heads)
                n_cpu += 1
            dt = time.perf_counter() - t1
        line["cpu_baseline"] = {"value": round(n_cpu * B / dt, 2), "unit": "clips/s", "cores": threads, "kind": "port",
                                "sample": f"{n_cpu} forwards of one request ({B} clip(s) x 300 frames), oracle/torch_port.transformer_lstm_forward "
                                          f"(slot-0 path on torch CPU ops, fp32), {dt:.1f} s"}
        err = float((last["y"][-1].cpu() - y_cpu).abs().max())
        line["parity_max_abs_dy_vs_cpu_port"] = err
        if not err < 1e-4:
            raise SystemExit(f"bench: HIP output differs from the CPU port by {err}")
    emit(line)


def launch_ranks(args) -> int:
    """`python bench.py --gpus N` without a launcher: start N ranks of this script under torch.distributed.run (one process
    per GPU, rendezvous on 127.0.0.1) and return its exit code."""
    if not args.launcher_selftest:
        have = torch.cuda.device_count() if torch.cuda.is_available() else 0
        if have < args.gpus:
            raise SystemExit(f"bench.py --gpus {args.gpus}: this node exposes {have} GPU(s); refusing to print a line for "
                             f"fewer ranks than asked")
    port = os.environ.get("MASTER_PORT") or str(29500 + (os.getpid() % 2000))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", port, os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd)


def launcher_selftest(world, rank, mode="infer", batch=32, strong=False):
    """rendezvous + one collective on gloo: what `--gpus N` has to get right before any GPU work (CPU test).  --mode train
    additionally runs config 5's exchange - the flat OPNet gradient bucket (1 421 056 floats + the guard slot) through
    parallel.GradBucket.all_reduce, weighted n_local / n_global - on CPU tensors, and reports the line's config block."""
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    t = torch.ones(1)
    dist.all_reduce(t)
    assert int(t.item()) == dist.get_world_size() == world
    out = {"launcher_selftest": True, "n_gpus": world, "ranks_seen": int(t.item())}
    if mode == "train":
        from objectpermanence_amd import ModelsFactory, parallel
        model = ModelsFactory.get_model("opnet", CFG)
        bucket = parallel.GradBucket(model.parameters())
        for i in range(len(bucket.params)):
            bucket.view(i).fill_(float(rank + 1))
        bucket.guard.zero_()
        bucket.guard[0] = 1.0 if rank == world - 1 else 0.0             # "the last rank's persistent launch gave up"
        bucket.all_reduce(batch, world * batch)
        want = sum(r + 1 for r in range(world)) / world
        assert bucket.flat.numel() == W_BYTES // 4 and torch.allclose(bucket.flat, torch.full_like(bucket.flat, want))
        assert float(bucket.guard[0]) == 1.0 and float(bucket.guard[1]) == 0.0    # every rank sees that SOME rank aborted
        out.update({"mode": "train", "grad_bucket_floats": int(bucket.flat.numel()), "guard_after_allreduce": float(bucket.guard[0]),
                    "scaling": "strong" if strong else "weak", "batch_per_gpu": batch,
                    "config": {"global_batch": world * batch, "parallelism": f"dp{world}"}})
    if rank == 0:
        print(json.dumps(out), flush=True)
    dist.destroy_process_group()


def main():
    args = parse()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    if args.launcher_selftest:
        if args.global_batch is not None and args.global_batch % world:
            raise SystemExit(f"--global-batch {args.global_batch} does not divide over {world} GPUs")
        return launcher_selftest(world, rank, args.mode, args.global_batch // world if args.global_batch is not None else (args.batch or 32),
                                 strong=args.global_batch is not None)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a ROCm GPU (the HIP path has no CPU fallback)")
    quiet_stdout()
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)
    dist = None
    if world > 1 or args.force_dist:
        import torch.distributed as dist
        if args.force_dist:
            # the product's own switch (objectpermanence_amd/parallel.py): every data-parallel branch - the gradient bucket's
            # all-reduce on the comm stream, the guard slots, the gathers - runs even in a group of one rank
            os.environ["OPNET_FORCE_DIST"] = "1"
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        # (OPNET_DIST_BACKEND=gloo: several ranks on ONE device - how a single-GPU box rehearses `--gpus 2`, tests/test_dp_two_ranks_gpu.py)
        backend = os.environ.get("OPNET_DIST_BACKEND", "nccl")
        dist.init_process_group(backend=backend, rank=rank, world_size=world, **({"device_id": dev} if backend == "nccl" else {}))
        if dist.get_world_size() != args.gpus:
            raise SystemExit(f"--gpus {args.gpus} but the process group has {dist.get_world_size()} ranks")

    if args.mode == "transformer":
        # exact: sixteen passes of 16 one-clip requests; throughput: sixteen passes of 256 (a region starts with the GPU idle while the
        # host submits the first pass's 256 requests - ~1.1 ms - which four passes per region would charge at 0.28 ms a pass)
        if args.steps == 200:
            args.steps, args.warmup = (256, 32) if args.exact else (4096, 512)
        return bench_transformer(args, world, rank, dev, dist)
    if args.mode == "detect":
        args.batch = args.batch or 16     # frames per pass
        if args.steps == 200:
            args.steps, args.warmup = 10, 2
        return bench_detect(args, world, rank, dev, dist)

    from objectpermanence_amd import ModelsFactory
    from synthdata import opnet as synth

    if args.global_batch is not None:
        if args.mode != "train":
            raise SystemExit("--global-batch is a --mode train option (BASELINE.json config 5)")
        if args.global_batch <= 0 or args.global_batch % world:
            raise SystemExit(f"--global-batch {args.global_batch} does not divide over {world} GPUs")
        args.batch = args.global_batch // world
    args.batch = args.batch or 32
    B = args.batch
    params = synth.opnet_synth_params(CFG)
    model = ModelsFactory.get_model("opnet", CFG)
    model.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in params.items()})
    model.eval().to(dev)

    # rank r owns clips [r*B, (r+1)*B) of the synthetic set; distinct clips per rank (weak scaling)
    n_unique = min(B, 32)
    boxes_np, labels_np = synth.make_batch(rank * B, n_unique, T_FRAMES)
    if n_unique < B:
        reps = (B + n_unique - 1) // n_unique
        boxes_np = np.tile(boxes_np, (reps, 1, 1, 1))[:B]
        labels_np = np.tile(labels_np, (reps, 1, 1))[:B]
    boxes = torch.from_numpy(boxes_np).to(dev)
    labels = torch.from_numpy(labels_np).to(dev)

    if args.mode == "train":
        return bench_train(args, model, boxes, labels, world, rank, dev, dist, params)
    from objectpermanence_amd import _lib
    if args.engine == "xcd" and not _lib.load().opnet_xcd_supported(CFG["object_to_track_hidden_dim"], CFG["videos_hidden_dim"]):
        print("bench: this device does not expose all 8 XCDs x 32 CUs - falling back to --engine chain", file=sys.stderr)
        args.engine = "chain"
    if args.engine == "chain":
        out, y = bench_infer_chain(args, model, boxes, world, rank, dev, dist)
    else:
        # DISTINCT clips in every step (up to 32 batches = one full launch; a longer run cycles through them): rank r owns
        # clips [r * nd * B, (r + 1) * nd * B) of the synthetic set
        nd = max(1, min(args.steps, 32))
        all_np, _ = synth.make_batch(rank * nd * B, nd * B, T_FRAMES)
        all_dev = torch.from_numpy(all_np).to(dev)
        batches = [all_dev[i * B:(i + 1) * B] for i in range(nd)]
        out, y = bench_infer_xcd(args, model, batches, world, rank, dev, dist)
    # the ranks part here: everything below is rank 0's own (no collective), so that no rank waits in a process-group
    # shutdown while rank 0 is still measuring the extras
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        out.update(accuracy_block(dev))
        out.update(other_batches(model, boxes, dev))
        out.update(training_block(params, boxes, labels, dev))
        out.update(transformer_block(dev))
        out.update(siblings_training_block(dev))
        out.update(detector_block(dev))
        if world == 1:
            out.update(inference_from_files_block())
        if world == 1 and not args.no_cpu_baseline:
            cb, y_cpu = cpu_baseline(boxes_np, params, args.cpu_seconds)
            out["cpu_baseline"] = cb
            y_np = y.cpu().numpy()
            if args.engine == "chain":
                # every timed step is the same B clips
                err = max(float(np.abs(y_np[lo:lo + B] - y_cpu).max()) for lo in range(0, y_np.shape[0], B))
                out["parity_clips_checked"] = int(B)
            else:
                # the LAST launch of the timed region carried steps [steps - n_last, steps): its clips - distinct while the
                # launch holds at most 32 batches - through the C port, every one of them
                from oracle import c_oracle
                n_last = y_np.shape[0] // B
                ids = [(i % nd) for i in range(args.steps - n_last, args.steps)]
                x_last = np.concatenate([all_np[i * B:(i + 1) * B] for i in ids])
                y_port, _ = c_oracle.opnet_forward(np.ascontiguousarray(x_last), params, c_oracle.usable_cores())
                err = float(np.abs(y_np - y_port).max())
                out["parity_clips_checked"] = int(len(set(ids)) * B)
                out["parity_note"] = "distinct clips of the last timed launch, each held against oracle/opnet_oracle.c"
            out["parity_max_abs_dy_vs_cpu_port"] = err
            if not err < 1e-4 and not os.environ.get("OPNET_HIP_LIB"):
                raise SystemExit(f"bench: HIP output of the last timed step differs from the CPU port by {err}")
        emit(out)


def timed_repeats(args, dev, dist, world, region):
    """Run `region()` (enqueue exactly --steps steps) --repeats times, each bracketed by barrier + synchronize on both sides;
    per repeat the MAX over ranks; returns (median, min, max, all) in seconds."""
    import gc
    times = []
    gc.collect()
    gc.disable()            # a generation-2 collection of the interpreter (30-50 ms with torch imported) inside a 5 ms region is not the path's time
    for _ in range(max(1, args.repeats)):
        torch.cuda.synchronize(dev)
        if dist is not None and world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        region()
        torch.cuda.synchronize(dev)
        if dist is not None and world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)
        times.append(time.perf_counter() - t0)
    gc.enable()
    if dist is not None and world > 1:
        tt = torch.tensor(times, dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        times = [float(v) for v in tt.tolist()]
    srt = sorted(times)
    return srt[len(srt) // 2], srt[0], srt[-1], times


_REAL_STDOUT = None


def quiet_stdout():
    """file descriptor 1 -> stderr until the line is printed: RCCL writes a version banner through C stdio when the first
    communicator is made (five lines on rank 0's stdout), and the contract is ONE JSON line there"""
    global _REAL_STDOUT
    if _REAL_STDOUT is None:
        sys.stdout.flush()
        _REAL_STDOUT = os.dup(1)
        os.dup2(2, 1)


def emit(out):
    """the JSON line is the ONLY thing on stdout: what C stdio still buffers (the RCCL banner) is flushed to where fd 1 points
    now - stderr - before the real stdout comes back"""
    global _REAL_STDOUT
    sys.stdout.flush()
    try:
        ctypes.CDLL(None).fflush(None)
    except OSError:
        pass
    if _REAL_STDOUT is not None:
        os.dup2(_REAL_STDOUT, 1)
        os.close(_REAL_STDOUT)
        _REAL_STDOUT = None
    print(json.dumps(out), flush=True)


def _line(args, world, B, clips_per_s, elapsed, workload, extra_cfg, spread=None):
    rep = {}
    if spread is not None:      # (min time -> max rate)
        t_min, t_max = spread
        rep = {"repeats": max(1, args.repeats), "value_is": "median over the repeats of the timed region",
               "value_min": round(world * B * args.steps / t_max, 1), "value_max": round(world * B * args.steps / t_min, 1)}
    return {
        "metric": "CATER clips/sec (300f x 10obj) OPNet inference",
        "value": round(clips_per_s, 1), "unit": "clips/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 4), **rep,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": workload, "global_batch": world * B, "frames": T_FRAMES, "parallelism": f"dp{world}",
                   "weights": "synthetic (synthdata/opnet.py counter RNG), fp32", **extra_cfg},
        # the whole job against the machine: every overhead (input concatenation, pack, output head, post-process,
        # launch gaps) is inside this one
        "whole_job_mfma_frac": round(clips_per_s / world * FLOP_PER_CLIP / (MFMA_F32_PEAK_TF * 1e12), 4),
    }


def training_block(params, boxes, labels, dev):
    """BASELINE.json config 2 next to the headline (outside the timed region; `--mode train` is the full measurement): a few
    training steps - forward + L1 + backward + Adam - of this rank's batch on a fresh copy of the model."""
    from objectpermanence_amd import FusedAdam, ModelsFactory, _lib
    from objectpermanence_amd.training import train_step
    m = ModelsFactory.get_model("opnet", CFG)
    m.load_state_dict({k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in params.items()})
    m = m.to(dev).train(True)
    opt = FusedAdam(m.parameters(), lr=1e-3)
    B = int(boxes.shape[0])
    n_warm, n_timed = 5, 20          # (5 timed steps charged the host's run-up to the first launch at ~30 us per step)
    for _ in range(n_warm):
        train_step("opnet", m, opt, boxes, labels, n_global=B, comm_stream=None, loss_kind="l1")
    torch.cuda.synchronize(dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n_timed):
        train_step("opnet", m, opt, boxes, labels, n_global=B, comm_stream=None, loss_kind="l1")
    e1.record()
    torch.cuda.synchronize(dev)
    ms = e0.elapsed_time(e1) / n_timed
    persistent = (B <= int(os.environ.get("OPNET_XCD4_MAX_B", "32")) and os.environ.get("OPNET_XCD4", "1") != "0"
                  and bool(_lib.load().opnet_xcd_supported(CFG["object_to_track_hidden_dim"], CFG["videos_hidden_dim"])))
    return {"training_step": {"batch": B, "ms_per_step": round(ms, 3), "clips_per_s": round(B / ms * 1e3, 1), "loss": "l1", "steps": n_timed, "warmup": n_warm,
                              "engine": "xcd4 (forward and reverse recurrence as one persistent launch each)" if persistent
                              else "chain (one launch per time step)"}}


def other_batches(model, boxes, dev):
    """The reference's own shipped batch sizes next to the bench's: configs/inference_config.json batch_size 16 and
    configs/training_config.json inference_batch_size 400 (one forward in flight, whatever engine OPNet.forward picks).
    Outside the timed region."""
    res = {}
    forced, model.use_xcd = model.use_xcd, "auto"
    for b in (16, 32, 400):      # 32 = one request of the bench's own batch ALONE (a latency, not 1 / throughput)
        reps = (b + boxes.shape[0] - 1) // boxes.shape[0]
        x = boxes.repeat(reps, 1, 1, 1)[:b].contiguous()
        with torch.no_grad():
            model(x)
            torch.cuda.synchronize(dev)
            # eleven forwards, each timed on its own (HIP events): the median is the request's latency; the maximum is kept in
            # the line (one forward in several hundred was seen at 30-50 ms: the kernel trace shows the GPU idle between two
            # launches of that forward - the Python interpreter's garbage collector, not a kernel; tools/stall_probe.py)
            times = []
            for _ in range(11):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                model(x)
                e1.record()
                torch.cuda.synchronize(dev)
                times.append(e0.elapsed_time(e1))
        ms = sorted(times)[len(times) // 2]
        res[str(b)] = {"ms_per_forward": round(ms, 3), "clips_per_s": round(b / ms * 1e3, 1), "max_ms": round(max(times), 3),
                       "engine": "xcd4" if model._wants_xcd4(b) else "xcd" if model._wants_xcd(b) else "chain"}
    model.use_xcd = forced
    return {"reference_batch_sizes": res}


def shader_clock_probe(model, batches, n_clips, dev, lib):
    """The shader clock the persistent forward actually ran at: ONE extra launch of the timed region's shape with the kernel's
    in-kernel stamps on (opnet_xcd_set_trace: s_memtime of block 0 at the top of every phase; a tick is a shader cycle) between HIP
    events around the kernel (opnet_xcd_profile).  The chip clocks to its power budget (MI355X_MICROARCH.md, DVFS): a dense fp32 MFMA
    stream does not sustain the 2.4 GHz that `roofline.peak` is quoted at, and `frac` against the peak AT THE MEASURED CLOCK is what the
    kernel's own cycles achieve.  Outside the timed region; a traced launch is a few % slower than an untraced one and is only used
    for the clock."""
    from objectpermanence_amd import _lib
    try:
        x = torch.cat([batches[i % len(batches)] for i in range((n_clips + batches[0].shape[0] - 1) // batches[0].shape[0])])[:n_clips].contiguous()
        T = int(x.shape[1])
        ng = (n_clips + 127) // 128
        tr = torch.zeros((T + 4) * ng * 8, dtype=torch.int64, device=dev)
        with torch.no_grad():
            model(x)                                   # the shape's workspace exists, the clock has ramped
            torch.cuda.synchronize(dev)
            kms, nl = ctypes.c_double(0.0), ctypes.c_int(0)
            lib.opnet_xcd_profile(1)
            lib.opnet_xcd_profile_read(ctypes.byref(kms), ctypes.byref(nl))      # (drain)
            lib.opnet_xcd_set_trace(tr.data_ptr())
            model(x)
            torch.cuda.synchronize(dev)
            lib.opnet_xcd_set_trace(None)
            _lib.check(lib.opnet_xcd_profile_read(ctypes.byref(kms), ctypes.byref(nl)), "opnet_xcd_profile_read")
            lib.opnet_xcd_profile(0)
        if model.verify_launches() or nl.value != 1 or kms.value <= 0:
            return None
        t0 = tr.cpu().numpy().reshape(-1, 8)[:, 0]
        t0 = t0[t0 != 0]
        if len(t0) < 8:
            return None
        ticks = float(t0[-1] - t0[0]) * len(t0) / (len(t0) - 1)      # the stamps bracket all phases but the last
        ghz = ticks / (kms.value * 1e6)
        return {"effective_ghz": round(ghz, 3), "nominal_ghz": 2.4, "phases_stamped": int(len(t0)), "traced_launch_ms": round(kms.value, 4),
                "method": "s_memtime stamps of block 0 over one extra traced launch of this shape / its HIP-event time (a tick = a shader cycle; "
                          "the kernel's prologue, ~0.5 % of the launch, is counted as phases)"}
    except Exception as e:      # the probe is context, never a reason to lose the line
        return {"error": str(e)[:200]}


def bench_infer_xcd(args, model, batches, world, rank, dev, dist):
    """Request batching into the per-XCD persistent forward: every step submits its batch to a ReasonerServer, which runs
    `per_launch` pending batches as one launch; the post-process (and, N > 1, the all-gather of the int32 predictions) runs
    once per launch on the launch's whole output.  Step i submits batches[i % len(batches)]: DISTINCT clips in every step of
    the driver's --steps 20 (640 distinct clips per launch)."""
    from objectpermanence_amd import _lib, metrics
    from objectpermanence_amd.serving import ReasonerServer
    lib = _lib.load()
    B = int(batches[0].shape[0])
    nd = len(batches)
    cap = max(1, int(lib.opnet_xcd_max_batch()) // B)
    if args.inflight > 0:
        per_launch = min(args.inflight, cap)
    else:
        # all steps in one launch when they fit; otherwise full launches (1024 clips = 8 groups on every XCD - a launch
        # takes as long as its fullest XCD, so 29 batches would cost the time of 32) and one smaller launch for the rest
        per_launch = min(args.steps, cap)
    model.use_xcd = "1"
    server = ReasonerServer(model, "opnet", max_clips=per_launch * B)
    exchange = dist is not None
    state = {"y": None, "pred": None, "seen": 0, "gathered": []}
    comm = torch.cuda.Stream(device=dev) if exchange else None
    # OPNET_DP_OVERLAP=1: the persistent launch does NOT wait for the previous launch's all-gather (see below) - the ordering was
    # chosen blind on one GPU; the switch lets the first multi-GPU run A/B it from two driver lines
    waits = os.environ.get("OPNET_DP_OVERLAP", "0") != "1"
    gather_events = []
    if exchange and waits:
        # the next launch's input concatenation overlaps the collective, the persistent launch itself waits for it: it needs
        # every CU of the device, and an RCCL kernel that holds some while a slower rank's collective sits behind ITS 5-ms
        # launch would stall this rank's launch for as long (ranks would take turns waiting for each other)
        server.before_launch = lambda: torch.cuda.current_stream(dev).wait_stream(comm)

    def after_flush():
        # one post-process per launch on the stream the forward was enqueued on; the all-gather of its int32 predictions
        # (3 MB for 640 clips) runs on a SIDE stream behind an event, so the host side of the next launch (request queue,
        # input concatenation) does not wait for the collective (north_star: "overlapped ... on a side HIP stream")
        if server.forwards == state["seen"]:
            return
        state["seen"] = server.forwards
        y = server.last_output[0]
        pred_px, _, _ = metrics.postprocess_and_iou(y)
        if exchange:
            gathered = torch.empty((world * pred_px.shape[0],) + tuple(pred_px.shape[1:]), dtype=pred_px.dtype, device=dev)
            comm.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(comm):
                g0, g1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                g0.record(comm)
                dist.all_gather_into_tensor(gathered, pred_px)
                g1.record(comm)
                gather_events.append((g0, g1, int(pred_px.numel()) * pred_px.element_size()))
            pred_px.record_stream(comm)
            state["gathered"] = (state["gathered"] + [gathered])[-4:]       # kept alive until the collectives are done
        state["y"], state["pred"] = y, pred_px

    def run(n):
        for i in range(n):
            server.submit(batches[i % nd])
            after_flush()
        server.flush()
        after_flush()
        if exchange:
            torch.cuda.current_stream(dev).wait_stream(comm)       # the step is done when its predictions are everywhere

    # untimed: the W warm-up steps, then one launch of every shape the timed region will issue (the remainder launch, then
    # the full one), so that their history workspaces exist and are mapped - a steady-state server has them - and the
    # last thing the GPU did before the timed region is what it does in it (measured: a 640-clip launch takes 5.48 ms after
    # a launch of its own shape and 5.75 ms after the 160-clip launch five warm-up steps make)
    run(max(args.warmup, 1))
    for shape_steps in sorted({args.steps % per_launch, per_launch} - {0}):
        run(shape_steps)
    torch.cuda.synchronize(dev)
    proof = collective_proof(dist, dev, world) if exchange else None
    del gather_events[:]
    lib.opnet_xcd_profile(1)
    f0 = server.forwards
    elapsed, t_min, t_max, _all = timed_repeats(args, dev, dist, world, lambda: run(args.steps))
    kms, nl = ctypes.c_double(0.0), ctypes.c_int(0)
    _lib.check(lib.opnet_xcd_profile_read(ctypes.byref(kms), ctypes.byref(nl)), "opnet_xcd_profile_read")
    if proof is not None and gather_events:
        proof.update({"op": "all_gather_into_tensor of the launch's int32 pixel boxes, one per persistent launch, on a side stream",
                      "allgather_ms_per_launch": round(sum(a.elapsed_time(b) for a, b, _ in gather_events) / len(gather_events), 4),
                      "allgathers": len(gather_events), "bytes_per_rank_per_launch": gather_events[-1][2],
                      "launch_waits_for_gather": bool(waits),
                      "switch": "OPNET_DP_OVERLAP=1 lets the persistent launch start without waiting for the previous all-gather"})
    lib.opnet_xcd_profile(0)
    # an aborted persistent launch was re-run on the chain by the model (launch_monitor.py): a bench line must not be
    # quoted on such a run
    if model.verify_launches() or model._monitor.aborted:
        raise SystemExit(f"bench: {model._monitor.aborted} persistent launch(es) aborted: {model.xcd_status()}")
    if rank != 0:
        return None, None
    R = max(1, args.repeats)
    clips = B * args.steps
    clips_per_s = world * clips / elapsed
    launches = (server.forwards - f0) // R
    assert (server.forwards - f0) == nl.value, (server.forwards - f0, nl.value)
    kernel_s = kms.value * 1e-3 / R            # the kernel's mean duration per timed region (HIP events around every launch)
    tf = clips * FLOP_PER_CLIP / kernel_s / 1e12
    # north_star's streaming model (every time step streams all weights once per B-clip batch and moves
    # each clip's state): what the launch-per-step design had to move for these clips, over this kernel's time.  It is a
    # MODEL figure here - the persistent kernel reads the weights once per launch and keeps them in registers.
    model_bytes = clips * T_FRAMES * (W_BYTES / B + STATE_BYTES_PER_CLIP)
    out = _line(args, world, B, clips_per_s, elapsed,
                f"opnet (configs/opnet_model_config.json: H1=256, H2=512) inference, batch={B} clips/GPU/step x 300 frames x "
                "15 slots (10 objects) x 6 features, precomputed bbox input resident in HBM, int32 pixel-box post-process on "
                f"device; steps are requests to a ReasonerServer that runs up to {per_launch} pending batches "
                f"({per_launch * B} clips) as one per-XCD persistent forward",
                {"engine": "xcd", "batches_per_launch": per_launch, "launches": launches,
                 "distinct_clips_per_timed_region": min(nd, args.steps) * B}, spread=(t_min, t_max))
    out["value_is"] = (f"median over the repeats of the timed region; a repeat is {launches} persistent launch(es) of {clips // max(launches, 1)} "
                       "clips (about 5 ms of GPU time): the spread value_min..value_max is the spread over single launches")
    if proof is not None:
        out["collective"] = proof
    cpl = clips // max(launches, 1)
    traffic = pmc_traffic("opnet_xcd_forward", cpl)
    if traffic.get("traffic"):
        # the launch's companions and SURVEY.md 8-d4's compulsory model (130 800 B per clip + the 5.68 MB of weights once per
        # launch) next to the kernel's own bytes: what the whole forward moves per launch
        pack = pmc_traffic("opnet_xcd_pack_input", cpl).get("traffic")
        compulsory = cpl * 130_800 + W_BYTES
        # the server's torch.cat of the launch's requests (serving.ReasonerServer, concat = True) sits in the timed region too: it
        # reads and writes every input byte once more (108 000 B per clip each way; arithmetic, not a PMC figure)
        concat = 2 * cpl * 108_000 if per_launch > 1 else 0
        traffic["traffic_pack_kernel"] = pack
        traffic["traffic_request_concat"] = concat
        traffic["compulsory_bytes_8d4"] = compulsory
        traffic["traffic_over_compulsory_8d4"] = round(traffic["traffic"] / compulsory, 3)
        traffic["forward_total_over_compulsory_8d4"] = round((traffic["traffic"] + (pack or 0) + concat) / compulsory, 3)
    out["roofline"] = {"bound": "mfma", "achieved": round(tf, 2), "peak": MFMA_F32_PEAK_TF, "unit": "TFLOP/s",
                       "frac": round(tf / MFMA_F32_PEAK_TF, 4), **traffic,
                       "kernel": "opnet_xcd_forward", "launch_ms": round(kms.value / max(launches * R, 1), 4),
                       "launches": launches, "alg_flop_per_launch": int(clips * FLOP_PER_CLIP / max(launches, 1)),
                       "timing": "HIP events around every launch of the kernel on its stream (opnet_xcd_profile)"}
    clk = shader_clock_probe(model, batches, cpl, dev, lib)
    if clk:
        if clk.get("effective_ghz"):
            clk["frac_of_peak_at_this_clock"] = round(tf / (MFMA_F32_PEAK_TF * clk["effective_ghz"] / clk["nominal_ghz"]), 4)
        out["roofline"]["shader_clock"] = clk
    out["roofline_hbm_model"] = {"bound": "hbm", "achieved": round(model_bytes / kernel_s / 1e9, 1), "peak": HBM_PEAK_GBS,
                                 "unit": "GB/s", "frac": round(model_bytes / kernel_s / 1e9 / HBM_PEAK_GBS, 4),
                                 "note": "SURVEY.md 8-d4 streaming-model bytes of the same clips (weights once per time step per "
                                         f"{B}-clip batch) over the kernel time; not bytes this kernel moves"}
    return out, state["y"]


def bench_infer_chain(args, model, boxes, world, rank, dev, dist):
    """round 1's engine: one hipGraph of T+3 opnet_step launches per batch, independent batches spread over S streams"""
    from objectpermanence_amd import metrics
    B = int(boxes.shape[0])
    model.use_xcd = "0"
    exchange = dist is not None
    S = args.streams if args.streams > 0 else 4
    # HIP multiplexes streams onto 4 hardware queues by default and streams sharing a queue serialise
    # (measured: raising GPU_MAX_HW_QUEUES to 8 collapses the overlap), so the timed region uses exactly
    # S <= 4 created streams and keeps the null stream out of it: stream 0 doubles as the timing stream.
    # The stream -> hardware-queue mapping is decided by the runtime at stream creation and varies from
    # run to run (measured: the same S gives 38 k or 57 k clips/s), so a pool of 8 streams is created and
    # the calibration below also picks WHICH S of them to use.
    pool = [torch.cuda.Stream(device=dev) for _ in range(8)]
    streams = pool[:S]
    main_stream = streams[0]
    G = max(1, args.gather_every)
    local_acc = [torch.empty((G, B, T_FRAMES, 4), dtype=torch.int32, device=dev) for _ in range(4)] if exchange else None
    gathered = [torch.empty((world * G * B, T_FRAMES, 4), dtype=torch.int32, device=dev) for _ in range(4)] if exchange else None
    filled = [0, 0, 0, 0]

    def flush(k):
        n = filled[k]
        if n:
            with torch.cuda.stream(streams[k]):
                dist.all_gather_into_tensor(gathered[k][:world * n * B], local_acc[k][:n].view(n * B, T_FRAMES, 4))
            filled[k] = 0

    def step(i):
        # independent batches: step i is enqueued on stream i % S, so up to S forwards are in flight
        k = i % S
        with torch.cuda.stream(streams[k]):
            with torch.no_grad():
                y, _logits = model(boxes)
            pred_px, _, _ = metrics.postprocess_and_iou(y)
            if exchange:
                local_acc[k][filled[k]].copy_(pred_px)
                filled[k] += 1
        if exchange and filled[k] == G:
            flush(k)
        return y, pred_px

    def drain():
        if exchange:
            for k in range(len(streams)):
                flush(k)
        for st in pool:
            if st is not main_stream:
                main_stream.wait_stream(st)

    if args.streams <= 0:
        # untimed calibration: how many of the 4 streams to use (stream -> hardware-queue mapping varies)
        best = (0.0, 1, 0)
        for off in (0, 4):
            for cand in (1, 2, 3, 4):
                if cand == 1 and off:
                    continue
                S, streams = cand, pool[off:off + cand]
                main_stream = streams[0]
                for i in range(2 * cand):
                    step(i)
                drain(); torch.cuda.synchronize(dev)
                tc = time.perf_counter()
                for i in range(24):
                    step(i)
                drain(); torch.cuda.synchronize(dev)
                rate = 24 / (time.perf_counter() - tc)
                if rate > best[0]:
                    best = (rate, cand, off)
        S, off = best[1], best[2]
        if world > 1:   # every rank must use the same S for the collective order: take rank 0's choice
            sc = torch.tensor([S], device=dev)
            dist.broadcast(sc, 0)
            S = int(sc.item())
        streams = pool[off:off + S]
        main_stream = streams[0]
    for i in range(args.warmup):
        step(i)
    drain()
    torch.cuda.synchronize(dev)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    last = {}

    def region():
        ev0.record(main_stream)
        for st in streams:
            if st is not main_stream:
                st.wait_stream(main_stream)
        for i in range(args.steps):
            last["y"], last["pred"] = step(i)
        drain()
        ev1.record(main_stream)

    proof = collective_proof(dist, dev, world) if exchange else None
    elapsed, t_min, t_max, _all = timed_repeats(args, dev, dist, world, region)
    y = last["y"]
    gpu_ms = ev0.elapsed_time(ev1)  # HIP events bracketing all launch streams of the last repeat: the kernels only
    # ONE launch's own duration: a single-stream pass (nothing overlaps it), HIP events on that stream
    single = streams[:1]
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(single[0]), torch.no_grad():
        model(boxes)
        e0.record(single[0])
        for _ in range(10):
            model(boxes)
        e1.record(single[0])
    torch.cuda.synchronize(dev)
    single_launch_us = e0.elapsed_time(e1) * 1e3 / (10 * (T_FRAMES + 3))
    if rank != 0:
        return None, None
    clips_per_s = world * B * args.steps / elapsed
    # dominant kernel: opnet_step, T+3 launches per forward.  Algorithmic bytes under the per-time-step streaming model:
    # every step reads all weights once and moves each clip's state.
    n_launch = args.steps * (T_FRAMES + 3)
    alg_bytes_per_launch = (W_BYTES + B * STATE_BYTES_PER_CLIP) * T_FRAMES / (T_FRAMES + 3)
    achieved = alg_bytes_per_launch / (single_launch_us * 1e-6) / 1e9
    amort_us = gpu_ms * 1e3 / n_launch
    amort = alg_bytes_per_launch / (amort_us * 1e-6) / 1e9
    out = _line(args, world, B, clips_per_s, elapsed,
                f"opnet (configs/opnet_model_config.json: H1=256, H2=512) inference, batch={B} clips/GPU/step x 300 frames x "
                "15 slots (10 objects) x 6 features, precomputed bbox input resident in HBM, int32 pixel-box post-process on "
                f"device; independent steps spread over {S} HIP streams",
                {"engine": "chain", "streams": S}, spread=(t_min, t_max))
    if proof is not None:
        out["collective"] = dict(proof, op=f"all_gather_into_tensor of the int32 pixel boxes of {G} steps of a stream, on that stream",
                                 launch_waits_for_gather=False)
    out["roofline"] = {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                       "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": None, "kernel": "opnet_step",
                       "launch_us": round(single_launch_us, 3), "alg_bytes_per_launch": int(alg_bytes_per_launch),
                       "timing": "HIP events around 10 forwards (T+3 launches each) alone on one stream"}
    out["roofline_amortised"] = {"achieved": round(amort, 1), "frac": round(amort / HBM_PEAK_GBS, 4), "unit": "GB/s",
                                 "launches_in_flight": S, "amortised_launch_us": round(amort_us, 3),
                                 "note": "all algorithmic bytes of the timed region / GPU time with S forwards in flight: counts "
                                         "the weight stream once per concurrent forward; not a per-kernel figure"}
    return out, y


if __name__ == "__main__":
    main()
