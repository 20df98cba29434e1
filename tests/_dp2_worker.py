"""One rank of tests/test_dp_two_ranks_gpu.py: the product's data-parallel training step with the HIP model, two ranks sharing the
box's ONE GPU (LOCAL_RANK 0 for both), collectives over gloo (RCCL refuses two ranks on one device).  argv: out_dir B T steps"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from objectpermanence_amd import FusedAdam, ModelsFactory, parallel          # noqa: E402
from objectpermanence_amd.training import global_loss, step_aborted, train_step          # noqa: E402
from oracle import synth          # noqa: E402  (test code: seeded clips and weights)

CFG = {"object_to_track_pred_dim": 15, "object_to_track_hidden_dim": 256, "videos_hidden_dim": 512}


def main():
    out_dir, B, T, steps = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    launch = parallel.init_from_env(backend="gloo")
    world, rank, active = parallel.world_rank()
    assert active and launch.owned and launch.device == torch.device("cuda", 0)
    dev = parallel.resolve_device("cuda:7")          # the JSON's device loses against cuda:LOCAL_RANK
    m = ModelsFactory.get_model("opnet", CFG)
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in synth.opnet_synth_params(CFG).items()})
    m = m.to(dev).train(True)
    opt = FusedAdam(m.parameters(), lr=1e-3)
    comm = torch.cuda.Stream(device=dev)
    losses, grads1 = [], None
    for k in range(steps):
        boxes_np, labels_np = synth.make_batch(10 * k, B, T)
        lo, hi = parallel.shard_range(B, world, rank)
        loss = train_step("opnet", m, opt, torch.from_numpy(boxes_np[lo:hi]).to(dev), torch.from_numpy(labels_np[lo:hi]).to(dev),
                          n_global=B, comm_stream=comm)
        assert not step_aborted(m)
        losses.append((float(loss), float(global_loss(m, loss))))
        if k == 0:                   # the all-reduced mean-loss gradients of the first step (the bucket the optimiser read)
            grads1 = {n: p.grad.detach().cpu().clone() for n, p in m.named_parameters()}
    torch.cuda.synchronize()
    torch.save({"params": {n: p.detach().cpu() for n, p in m.named_parameters()}, "losses": losses, "grads1": grads1,
                "guard": m._grad_bucket.guard.cpu()}, os.path.join(out_dir, f"rank{rank}.pt"))
    parallel.shutdown(launch)


if __name__ == "__main__":
    main()
