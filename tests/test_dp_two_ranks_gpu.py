"""BASELINE config 5 at world size TWO with the HIP model, on the one GPU a test box has: two processes (LOCAL_RANK 0 both, gloo for
the collectives - RCCL refuses two ranks on one device) each run the product's `train_step` on their contiguous half of every
minibatch: in-place gradient bucket, all-reduce weighted n_local / n_global, guard slots, guarded Adam.  The two ranks must end
with IDENTICAL weights, and those must be the single-process full-minibatch weights up to the order of the fp32 sums (the reference's
contract for a sharded run: N-GPU == 1-GPU, training_main.py:183-217).  The persistent engines are switched off in the workers (two
processes cannot both hold all 256 CUs): what runs is the launch chain, the same arithmetic behind the same exchange.
tests/test_dp_rccl_gpu.py has the RCCL transport (world of one, bit-identical); test_host_logic.py the gloo logic on CPU."""
import os
import subprocess
import sys

import pytest
import torch

from oracle import synth

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CFG = {"object_to_track_pred_dim": 15, "object_to_track_hidden_dim": 256, "videos_hidden_dim": 512}


def _single(B, T, steps):
    from objectpermanence_amd import FusedAdam, ModelsFactory
    from objectpermanence_amd.training import train_step
    m = ModelsFactory.get_model("opnet", CFG)
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in synth.opnet_synth_params(CFG).items()})
    m = m.to("cuda:0").train(True)
    opt = FusedAdam(m.parameters(), lr=1e-3)
    losses, grads1 = [], None
    for k in range(steps):
        boxes_np, labels_np = synth.make_batch(10 * k, B, T)
        losses.append(float(train_step("opnet", m, opt, torch.from_numpy(boxes_np).to("cuda:0"), torch.from_numpy(labels_np).to("cuda:0"),
                                       n_global=B)))
        if k == 0:
            grads1 = {n: p.grad.detach().cpu().clone() for n, p in m.named_parameters()}
    torch.cuda.synchronize()
    return {n: p.detach().cpu() for n, p in m.named_parameters()}, losses, grads1


@pytest.mark.parametrize("B,T", [(12, 10), (7, 25)])          # (7, 25): shards of 4 and 3 clips - unequal weights n_local / n_global
def test_two_ranks_on_one_gpu_train_to_the_single_process_weights(tmp_path, B, T):
    steps = 3
    env = dict(os.environ, WORLD_SIZE="2", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29700 + os.getpid() % 200),
               OPNET_XCD4="0", OPNET_XCD="0", HSA_ENABLE_IPC_MODE_LEGACY="0", PYTHONPATH=REPO)
    env.pop("OPNET_FORCE_DIST", None)
    procs = [subprocess.Popen([sys.executable, os.path.join(REPO, "tests", "_dp2_worker.py"), str(tmp_path), str(B), str(T), str(steps)],
                              env=dict(env, RANK=str(r)), cwd=REPO, stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(2)]
    outs = [p.communicate(timeout=600)[0].decode(errors="replace") for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(o[-3000:] for o in outs)
    r0, r1 = (torch.load(tmp_path / f"rank{r}.pt") for r in range(2))
    want, want_losses, want_grads = _single(B, T, steps)
    gmax = max(float(g.abs().max()) for g in want_grads.values())
    for n in want:
        assert torch.equal(r0["params"][n], r1["params"][n]), n                     # the ranks agree bit for bit
        assert torch.equal(r0["grads1"][n], r1["grads1"][n]), n
        # the exchanged gradient is the single-process mean-loss gradient up to the order of the fp32 sums ...
        assert float((r0["grads1"][n] - want_grads[n]).abs().max()) <= 2e-5 * gmax, n
        # ... and three Adam steps (lr 1e-3: an update of ~1e-3 per step whatever the gradient's size, so rounding noise in
        # near-zero gradients shows at the 1e-5 level) land within a percent of one step's size
        assert float((r0["params"][n] - want[n]).abs().max()) <= 3e-5, n
    for k in range(steps):
        # slot 2 of the guard carries the whole minibatch's loss to every rank: sum of the shards' n_local / n_global shares
        assert r0["losses"][k][1] == r1["losses"][k][1]
        assert abs(r0["losses"][k][1] - want_losses[k]) <= 1e-6 * max(1.0, abs(want_losses[k]))
        if B % 2 == 0:
            assert r0["losses"][k][0] != r1["losses"][k][0]                          # (their own shards' losses differ)
    assert float(r0["guard"][0]) == 0.0 and float(r0["guard"][1]) == 0.0
