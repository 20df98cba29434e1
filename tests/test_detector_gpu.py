"""GPU checks of the detector conv path against the build-authored torch restatement
(oracle/detector_oracle.py).  Parity with the reference's torchvision detector is UNPINNED."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import detector_oracle as do, synth

pytestmark = pytest.mark.gpu


def _nhwc(x_nchw):
    return x_nchw.permute(0, 2, 3, 1).contiguous()


@pytest.mark.parametrize("cin,cout,k,stride,pad,h,w", [
    (3, 64, 7, 2, 3, 37, 53),      # stem (3 -> 4 channels, 49 taps)
    (64, 64, 1, 1, 0, 20, 27), (64, 64, 3, 1, 1, 20, 27), (128, 128, 3, 2, 1, 21, 30),
    (256, 512, 1, 2, 0, 17, 19), (20, 70, 3, 1, 1, 9, 11), (2048, 256, 1, 1, 0, 4, 5),
])
def test_conv2d_matches_torch(cin, cout, k, stride, pad, h, w):
    from objectpermanence_amd.detector import _Conv
    sd = {"w": synth.synth_tensor(f"w{cin}{cout}{k}", (cout, cin, k, k), float(np.sqrt(6.0 / (cin * k * k)))),
          "b": synth.synth_tensor("b", (cout,), 0.2)}
    x = torch.from_numpy(synth.synth_tensor("x", (2, cin, h, w), 1.0))
    res = torch.from_numpy(synth.synth_tensor("r", (2, cout, (h + 2 * pad - k) // stride + 1, (w + 2 * pad - k) // stride + 1), 1.0))
    conv = _Conv(sd, "w", bias="b", stride=stride, pad=pad)
    xin = x
    if cin % 4:
        xin = torch.cat([x, torch.zeros(2, 4 - cin % 4, h, w)], dim=1)
    y = conv(_nhwc(xin).cuda(), relu=True, residual=_nhwc(res).cuda())
    torch.cuda.synchronize()
    ref = F.relu(F.conv2d(x.double(), torch.from_numpy(sd["w"]).double(), torch.from_numpy(sd["b"]).double(),
                          stride=stride, padding=pad) + res.double())
    got = y.cpu().permute(0, 3, 1, 2).double()
    assert got.shape == ref.shape
    assert (got - ref).abs().max() < 2e-5 * max(1.0, ref.abs().max())


def test_preprocess_matches_oracle():
    from objectpermanence_amd.detector import preprocess_frame
    rng = np.random.default_rng(0)
    frame = rng.integers(0, 256, size=(240, 320, 3), dtype=np.uint8)
    for min_size, max_size in ((800, 1333), (96, 120)):
        y = preprocess_frame(frame, "cuda:0", min_size, max_size)
        torch.cuda.synchronize()
        ref = do.preprocess(frame, min_size, max_size)
        assert tuple(y.shape) == (1, ref.shape[2], ref.shape[3], 4)
        got = y.cpu()[..., :3].permute(0, 3, 1, 2)
        assert (got - ref).abs().max() < 2e-5
        assert float(y[..., 3].abs().max()) == 0.0
    assert tuple(preprocess_frame(frame, "cuda:0").shape) == (1, 800, 1088, 4)     # 240x320 -> 800x1066 -> pad /32


def test_backbone_fpn_matches_oracle():
    from objectpermanence_amd.detector import ResNet50FPNBackbone, preprocess_frame
    params = do.synth_backbone_params()
    rng = np.random.default_rng(1)
    frame = rng.integers(0, 256, size=(60, 80, 3), dtype=np.uint8)
    bb = ResNet50FPNBackbone(params, "cuda:0")
    x = preprocess_frame(frame, "cuda:0", min_size=128, max_size=200)          # 128 x 170 -> 128 x 192
    feats = bb.forward_nhwc(x)
    torch.cuda.synchronize()
    ref = do.backbone_fpn_forward(do.preprocess(frame, 128, 200), params)
    assert list(feats.keys()) == list(ref.keys())
    for k in ref:
        got = feats[k].cpu().permute(0, 3, 1, 2).double()
        assert got.shape == ref[k].shape, k
        scale = float(ref[k].abs().max())
        assert scale > 1e-2, "degenerate oracle activations"
        assert float((got - ref[k]).abs().max()) < 5e-4 * max(1.0, scale), k


def test_remove_low_probability_object():
    from objectpermanence_amd.detector import CaterObjectDetector
    out = {"boxes": torch.arange(20.).view(5, 4), "labels": torch.tensor([3, 140, 7, 9, 1]),
           "scores": torch.tensor([0.99, 0.9, 0.8, 0.79, 0.95])}     # last one is out of order on purpose
    kept = CaterObjectDetector.remove_low_probability_object(out)
    # k = count(scores >= 0.8) = 4 -> the first 4 rows, exactly the reference's prefix behaviour
    assert kept["scores"].tolist() == pytest.approx([0.99, 0.9, 0.8, 0.79]) and kept["boxes"].shape == (4, 4)
    with pytest.raises(NotImplementedError):
        CaterObjectDetector("x.pth")(np.zeros((240, 320, 3), np.uint8), torch.device("cuda:0"))
