"""Command line of the reference (main.py:13-140) on the HIP path:

    python -m objectpermanence_amd {inference, preprocess, training, analysis, cater_inference} ...

Same sub-commands and flags, so the reference's shell recipes and JSON configs carry over; only the learned reasoners
are served (`--model_type` of the programmed heuristics / DaSiamRPN tracker is refused - out of scope).

Multi-GPU: start it under torchrun, one rank per GPU -

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29500 \
        -m objectpermanence_amd training --model_type opnet --model_config ... --training_config ...

and every sub-command but `analysis` joins the RCCL group on `cuda:LOCAL_RANK` (parallel.init_from_env) and shards its work.
"""
from __future__ import annotations

import argparse
import json
import sys

from .supported_models import TRAINING_SUPPORTED_MODELS

_ANALYSIS_FILES = ["containment_annotations", "containment_only_static_annotations", "containment_with_movements_annotations",
                   "visibility_ratio_gt_0", "visibility_ratio_gt_30", "visibility_ratio_gt_99"]


def build_parser() -> argparse.ArgumentParser:
    parser = argparse.ArgumentParser(prog="python -m objectpermanence_amd",
                                     description="training and inference over the CATER data (MI355X / HIP path)")
    sub = parser.add_subparsers(dest="mode", required=True)

    def command(name, *flags):
        p = sub.add_parser(name)
        for flag, required, extra in flags:
            p.add_argument("--" + flag, type=str, required=required, **extra)
        return p

    models = {"choices": sorted(TRAINING_SUPPORTED_MODELS)}
    command("inference", ("model_type", True, models), ("results_dir", True, {}), ("inference_config", True, {}),
            ("model_config", False, {}))
    command("preprocess", ("results_dir", True, {}), ("config", True, {}))
    command("training", ("model_type", True, models), ("model_config", True, {}), ("training_config", True, {}))
    command("analysis", ("predictions_dir", True, {}), ("labels_dir", True, {}),
            *[(f, False, {}) for f in _ANALYSIS_FILES], ("iou_thresholds", True, {"default": "0.5,0.9"}),
            ("output_file", True, {}))
    command("cater_inference", ("results_dir", True, {}), ("inference_config", True, {}), ("model_config", False, {}))
    return parser


def main(argv=None) -> int:
    args = build_parser().parse_args(argv)
    if args.mode == "analysis":          # offline CSV over prediction files: no device, no ranks
        return _run(args)
    # Data parallelism (not in the reference, which takes ONE device from its JSON: training_main.py:144, inference_main.py:189):
    #   python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 -m objectpermanence_amd training ...
    # torchrun's environment (WORLD_SIZE / RANK / LOCAL_RANK) makes this process rank RANK of an RCCL group on cuda:LOCAL_RANK -
    # the JSON's "device" is overridden by it - and the drivers shard clips / minibatches over the ranks (parallel.py);
    # rank 0 writes the files.  Without that environment this is the reference's single-device run.
    from . import parallel
    import os
    if os.environ.get("OPNET_SEED"):     # a reproducible random initialisation (the reference leaves torch's seed alone)
        import torch
        torch.manual_seed(int(os.environ["OPNET_SEED"]))
    with parallel.init_from_env():
        return _run(args)


def _run(args) -> int:
    if args.mode == "inference":
        from .inference_main import reasoning_inference_main
        reasoning_inference_main(args.model_type, args.results_dir, args.inference_config, args.model_config)
    elif args.mode == "preprocess":
        from .preprocess_perception_main import preprocess_main
        preprocess_main(args.results_dir, args.config)
    elif args.mode == "training":
        from .training_main import training_main
        with open(args.model_config, "rb") as f:
            model_config = json.load(f)
        with open(args.training_config, "rb") as f:
            train_config = json.load(f)
        training_main(args.model_type, train_config, model_config)
    elif args.mode == "analysis":
        from .analysis import analyze_results
        analyze_results(args.predictions_dir, args.labels_dir, args.output_file,
                        *[getattr(args, f) for f in _ANALYSIS_FILES], [float(t) for t in args.iou_thresholds.split(",")])
    elif args.mode == "cater_inference":
        from .cater_setup_inference import cater_setup_inference
        cater_setup_inference("opnet", args.results_dir, args.inference_config, args.model_config)      # main.py:133
    return 0


if __name__ == "__main__":
    sys.exit(main())
