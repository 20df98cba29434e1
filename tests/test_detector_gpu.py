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


@pytest.mark.parametrize("cin,cout,n,h,w,relu,ws_mb", [
    (256, 256, 2, 50, 68, True, 8192),       # an FPN / RPN-head shape, even map
    (256, 256, 3, 25, 34, False, 8192),      # odd height: the last tile row is half outside the map
    (512, 512, 2, 13, 17, True, 8192),       # layer4's conv2, odd both ways
    (256, 128, 5, 20, 27, True, 4),          # workspace capped to 4 MB: the pass runs in chunks of images
    (64, 64, 2, 40, 36, True, 8192),         # (below the product's channel threshold: forced here)
])
def test_conv2d_winograd_matches_torch_and_the_direct_conv(monkeypatch, cin, cout, n, h, w, relu, ws_mb):
    """Winograd F(2 x 2, 3 x 3) form of the stride-1 3 x 3 convs (csrc/wino_kernels.hip): against torch's fp64 conv2d to the tolerance of
    the direct kernel's tests, and against the direct kernel itself to 1e-5 of max|y| (the bar the probe was held to)"""
    from objectpermanence_amd.detector import _Conv
    sd = {"w": synth.synth_tensor(f"ww{cin}{cout}", (cout, cin, 3, 3), float(np.sqrt(6.0 / (cin * 9)))),
          "b": synth.synth_tensor("wb", (cout,), 0.2)}
    x = torch.from_numpy(synth.synth_tensor("wx", (n, cin, h, w), 1.0))
    conv = _Conv(sd, "w", bias="b", stride=1, pad=1)
    monkeypatch.setenv("OPDET_WINO_WS_MB", str(ws_mb))
    monkeypatch.setattr(_Conv, "WINO_MIN_CIN", 16)
    monkeypatch.setattr(_Conv, "WINO_MIN_TILES", 0)
    assert conv._winograd(n, h, w, None)
    y = conv(_nhwc(x).cuda(), relu=relu)
    monkeypatch.setenv("OPDET_WINOGRAD", "0")
    assert not conv._winograd(n, h, w, None)
    y_direct = conv(_nhwc(x).cuda(), relu=relu)
    torch.cuda.synchronize()
    ref = F.conv2d(x.double(), torch.from_numpy(sd["w"]).double(), torch.from_numpy(sd["b"]).double(), stride=1, padding=1)
    if relu:
        ref = F.relu(ref)
    got = y.cpu().permute(0, 3, 1, 2).double()
    assert got.shape == ref.shape
    assert (got - ref).abs().max() < 2e-5 * max(1.0, ref.abs().max())
    assert (y - y_direct).abs().max().item() <= 1e-5 * max(1.0, y_direct.abs().max().item())


def test_winograd_weights_of_a_fresh_conv_are_complete_before_another_stream_uses_them():
    """the first Winograd call of a conv transforms its weights on the calling stream; a second pass in flight on ANOTHER stream must
    not multiply by them before that has happened (seen as differing detections of the second pass of a fresh detector)"""
    from objectpermanence_amd.detector import _Conv
    sd = {"w": synth.synth_tensor("wfw", (256, 256, 3, 3), 0.03), "b": synth.synth_tensor("wfb", (256,), 0.2)}
    x = _nhwc(torch.from_numpy(synth.synth_tensor("wfx", (2, 256, 40, 48), 1.0))).cuda()
    ref = _Conv(sd, "w", bias="b", stride=1, pad=1)(x, relu=True).clone()
    torch.cuda.synchronize()
    conv = _Conv(sd, "w", bias="b", stride=1, pad=1)
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    with torch.cuda.stream(sa):
        torch.cuda._sleep(200_000_000)              # ~0.1 s of queue ahead of the weight transform
        ya = conv(x, relu=True)
    with torch.cuda.stream(sb):
        yb = conv(x, relu=True)
    torch.cuda.synchronize()
    assert torch.equal(ya, ref) and torch.equal(yb, ref)


def test_winograd_workspaces_are_bounded_over_streams():
    """one workspace per stream the convs are enqueued on, at most OPDET_WINO_WS_STREAMS (4) of them kept: a session that
    goes through many streams does not hold a workspace for each"""
    from objectpermanence_amd import detector
    from objectpermanence_amd.detector import _Conv
    sd = {"w": synth.synth_tensor("wsw", (256, 256, 3, 3), 0.03), "b": synth.synth_tensor("wsb", (256,), 0.2)}
    x = _nhwc(torch.from_numpy(synth.synth_tensor("wsx", (1, 256, 40, 48), 1.0))).cuda()
    conv = _Conv(sd, "w", bias="b", stride=1, pad=1)
    ref = conv(x, relu=True)
    torch.cuda.synchronize()
    for _ in range(7):
        st = torch.cuda.Stream()
        with torch.cuda.stream(st):
            y = conv(x, relu=True)
        st.synchronize()
        assert torch.equal(y, ref)
    assert len(detector._WINO_WS) <= detector._WINO_WS_MAX


@pytest.mark.parametrize("cin,cout,k,stride,pad,h,w", [
    (32, 80, 3, 1, 1, 128, 130),     # >= 256 pixel tiles: the LDS-staged kernels (BN = 128), ragged M
    (64, 48, 1, 1, 0, 131, 127),     # BN = 64
    (16, 200, 3, 2, 1, 257, 255),    # stride 2, two channel tiles with a ragged second one
    (48, 77, 3, 1, 1, 128, 129),     # Cout % 4 != 0: scalar epilogue
    (4, 64, 7, 2, 3, 300, 260),      # stem-like (Cin = 4): the unaligned register-staged variant
])
def test_conv2d_tiled_matches_torch(cin, cout, k, stride, pad, h, w):
    from objectpermanence_amd.detector import _Conv
    sd = {"w": synth.synth_tensor(f"tw{cin}{cout}{k}", (cout, cin, k, k), float(np.sqrt(6.0 / (cin * k * k)))),
          "b": synth.synth_tensor("tb", (cout,), 0.2)}
    x = torch.from_numpy(synth.synth_tensor("tx", (2, cin, h, w), 1.0))
    oh, ow = (h + 2 * pad - k) // stride + 1, (w + 2 * pad - k) // stride + 1
    assert (2 * oh * ow + 127) // 128 >= 256
    res = torch.from_numpy(synth.synth_tensor("tr", (2, cout, oh, ow), 1.0))
    conv = _Conv(sd, "w", bias="b", stride=stride, pad=pad)
    y = conv(_nhwc(x).cuda(), relu=True, residual=_nhwc(res).cuda())
    y2 = conv(_nhwc(x).cuda(), relu=False)
    torch.cuda.synchronize()
    lin = F.conv2d(x.double(), torch.from_numpy(sd["w"]).double(), torch.from_numpy(sd["b"]).double(), stride=stride, padding=pad)
    for got, ref in ((y, F.relu(lin + res.double())), (y2, lin)):
        got = got.cpu().permute(0, 3, 1, 2).double()
        assert got.shape == ref.shape
        assert (got - ref).abs().max() < 2e-5 * max(1.0, ref.abs().max())


@pytest.mark.parametrize("cin,cout,k,stride,pad,n,h,w,slices", [
    (512, 512, 3, 1, 1, 1, 25, 34, 12),     # layer4 3x3 of one frame: 56 tiles of 128 x 64, 288 K steps
    (256, 256, 3, 1, 1, 1, 50, 68, 6),      # layer3 3x3: 108 tiles, 144 K steps
    (1024, 512, 1, 1, 0, 1, 25, 34, 4),     # 1x1, 64 K steps
    (256, 48, 3, 2, 1, 1, 50, 68, 9),       # BN = 64 tiles, stride 2, Cout % 16 != 0
    (64, 64, 1, 1, 0, 2, 20, 27, 1),        # K too short: not split (0 bytes of scratch), same entry point
])
def test_conv2d_split_k_matches_torch(cin, cout, k, stride, pad, n, h, w, slices):
    """one frame's deep layers: K split over blockIdx.z, partial sums added in slice order by conv_splitk_reduce"""
    from objectpermanence_amd import _lib
    from objectpermanence_amd.detector import _Conv
    sd = {"w": synth.synth_tensor(f"sw{cin}{cout}{k}", (cout, cin, k, k), float(np.sqrt(6.0 / (cin * k * k)))),
          "b": synth.synth_tensor("sb", (cout,), 0.2)}
    x = torch.from_numpy(synth.synth_tensor("sx", (n, cin, h, w), 1.0))
    oh, ow = (h + 2 * pad - k) // stride + 1, (w + 2 * pad - k) // stride + 1
    res = torch.from_numpy(synth.synth_tensor("sr", (n, cout, oh, ow), 1.0))
    conv = _Conv(sd, "w", bias="b", stride=stride, pad=pad)
    nws = _lib.load().opdet_conv2d_workspace_bytes(n, h, w, cin, cout, k, k, stride, pad, conv.kp)
    assert nws == (slices * n * oh * ow * cout * 4 if slices > 1 else 0)
    y = conv(_nhwc(x).cuda(), relu=True, residual=_nhwc(res).cuda())
    y2 = conv(_nhwc(x).cuda(), relu=False)
    y3 = conv(_nhwc(x).cuda(), relu=False)
    torch.cuda.synchronize()
    assert torch.equal(y2, y3)                                  # slice order is fixed: run-to-run identical
    lin = F.conv2d(x.double(), torch.from_numpy(sd["w"]).double(), torch.from_numpy(sd["b"]).double(), stride=stride, padding=pad)
    for got, ref in ((y, F.relu(lin + res.double())), (y2, lin)):
        got = got.cpu().permute(0, 3, 1, 2).double()
        assert got.shape == ref.shape
        assert (got - ref).abs().max() < 2e-5 * max(1.0, ref.abs().max())


@pytest.mark.parametrize("cin,cout,n,h,w,th,tw", [
    (256, 256, 2, 128, 130, 64, 65),      # the LDS-DMA kernel un-split: the addition rides in the conv's epilogue (narrow tile: prefetched)
    (512, 256, 1, 100, 136, 50, 68),      # one frame's P3 lateral: 107 tiles x 4
    (1024, 256, 1, 50, 68, 25, 34),       # one frame's P4 lateral: split K -> conv + in-place upsample_add
    (32, 40, 1, 23, 31, 12, 16),          # tiny: the un-staged kernel -> conv + in-place upsample_add; odd sizes (floor(dst * T / O))
])
def test_lateral_conv_plus_upsampled_top(cin, cout, n, h, w, th, tw):
    """FeaturePyramidNetwork's top-down step in one call = lateral conv, then F.interpolate(top, nearest) added: bit for bit the two
    launches it replaces (same fp32 operations in the same order), and torch's fp64 within rounding."""
    from objectpermanence_amd import _lib
    from objectpermanence_amd.detector import _Conv
    sd = {"w": synth.synth_tensor(f"lw{cin}{cout}", (cout, cin, 1, 1), float(np.sqrt(6.0 / cin))), "b": synth.synth_tensor("lb", (cout,), 0.2)}
    x = torch.from_numpy(synth.synth_tensor("lx", (n, cin, h, w), 1.0))
    top = torch.from_numpy(synth.synth_tensor("lt", (n, cout, th, tw), 1.0))
    conv = _Conv(sd, "w", bias="b")
    xd, td = _nhwc(x).cuda(), _nhwc(top).cuda().contiguous()
    got = conv.plus_upsampled(xd, td)
    lat = conv(xd, relu=False)
    two = torch.empty_like(lat)
    _lib.check(_lib.load().opdet_upsample_add_f32(lat.data_ptr(), td.data_ptr(), two.data_ptr(), n, h, w, cout, th, tw,
                                                  torch.cuda.current_stream().cuda_stream), "opdet_upsample_add_f32")
    torch.cuda.synchronize()
    assert torch.equal(got, two)
    ref = F.conv2d(x.double(), torch.from_numpy(sd["w"]).double(), torch.from_numpy(sd["b"]).double()) + \
        F.interpolate(top.double(), size=(h, w), mode="nearest")
    g = got.cpu().permute(0, 3, 1, 2).double()
    assert (g - ref).abs().max() < 2e-5 * max(1.0, ref.abs().max())


@pytest.mark.parametrize("c3in,dsin,cout,stride,n,h,w,path", [
    (64, 64, 256, 1, 2, 128, 130, "one"),        # layer1-like: both sources at stride 1, narrow tiles, ragged M
    (128, 256, 512, 2, 2, 100, 136, "one"),      # layer2-like: the second source every other pixel
    (256, 512, 1024, 2, 1, 50, 68, "split"),     # one frame's layer3: K split, the slice boundary inside / across the sources
    (512, 1024, 2048, 2, 1, 25, 34, "split"),    # one frame's layer4
    (64, 64, 256, 1, 1, 9, 11, "two"),           # too few tiles for the LDS-DMA kernel: the two convs
])
def test_conv3_plus_downsample_as_one_product(c3in, dsin, cout, stride, n, h, w, path):
    """relu(conv3(out) + downsample(x)) of a stage's first bottleneck as ONE product over the concatenated K (opdet_conv2d_dual_f32)
    against torch fp64 and against the two launches it replaces (same math, another summation order: rounding apart)."""
    from objectpermanence_amd import _lib
    from objectpermanence_amd.detector import _Conv, _DualConv
    sd = {"w3": synth.synth_tensor(f"d3{c3in}{cout}", (cout, c3in, 1, 1), float(np.sqrt(3.0 / c3in))), "b3": synth.synth_tensor("db3", (cout,), 0.2),
          "wd": synth.synth_tensor(f"dd{dsin}{cout}", (cout, dsin, 1, 1), float(np.sqrt(3.0 / dsin))), "bd": synth.synth_tensor("dbd", (cout,), 0.2)}
    out = torch.from_numpy(synth.synth_tensor("dout", (n, c3in, h, w), 1.0))
    hx, wx = (h - 1) * stride + 1 + (stride - 1), (w - 1) * stride + 1          # (an even height at stride 2, an odd width)
    x = torch.from_numpy(synth.synth_tensor("dx", (n, dsin, hx, wx), 1.0))
    assert (hx - 1) // stride + 1 == h and (wx - 1) // stride + 1 == w
    c3, ds = _Conv(sd, "w3", bias="b3"), _Conv(sd, "wd", bias="bd", stride=stride)
    dual = _DualConv(c3, ds)
    od, xd = _nhwc(out).cuda(), _nhwc(x).cuda()
    nws = _lib.load().opdet_conv2d_dual_workspace_bytes(n, h, w, c3in, hx, wx, dsin, stride, cout)
    assert {"one": nws == 0, "split": nws > 0, "two": nws < 0}[path]
    got = dual(od, xd)
    again = dual(od, xd)
    two = c3(od, relu=True, residual=ds(xd, relu=False))
    torch.cuda.synchronize()
    assert torch.equal(got, again)
    ref = F.relu(F.conv2d(out.double(), torch.from_numpy(sd["w3"]).double(), torch.from_numpy(sd["b3"]).double()) +
                 F.conv2d(x.double(), torch.from_numpy(sd["wd"]).double(), torch.from_numpy(sd["bd"]).double(), stride=stride))
    g = got.cpu().permute(0, 3, 1, 2).double()
    assert g.shape == ref.shape
    assert (g - ref).abs().max() < 2e-5 * max(1.0, ref.abs().max())
    assert (got - two).abs().max() < 2e-5 * max(1.0, float(two.abs().max()))
    if path == "two":
        assert torch.equal(got, two)


def test_linear_rows_split_k_and_refusals():
    """TwoMLPHead.fc6 of one frame (1000 x 12544 -> 1024): 128 tiles of 128 x 64, 784 K steps -> 5 slices"""
    from objectpermanence_amd import _lib
    from objectpermanence_amd.detector import _Linear
    lib = _lib.load()
    wt = torch.from_numpy(synth.synth_tensor("fc6w", (1024, 12544), 0.02))
    b = torch.from_numpy(synth.synth_tensor("fc6b", (1024,), 0.2))
    x = torch.from_numpy(synth.synth_tensor("fc6x", (1000, 12544), 1.0))
    fc = _Linear(wt, b, "cuda:0")
    assert lib.opdet_conv2d_workspace_bytes(1, 1, 1000, 12544, 1024, 1, 1, 1, 0, fc.kp) == 5 * 1000 * 1024 * 4
    y = fc.rows(x.cuda(), relu=True)
    torch.cuda.synchronize()
    ref = F.relu(x.double() @ wt.double().t() + b.double())
    assert (y.cpu().double() - ref).abs().max() < 2e-5 * max(1.0, ref.abs().max())
    # too little scratch for a split shape is refused, not silently run another way
    xs, ys = x.cuda(), torch.empty((1000, 1024), device="cuda:0")
    small = torch.empty(1024, dtype=torch.uint8, device="cuda:0")
    rc = lib.opdet_conv2d_ws_f32(xs.data_ptr(), fc.w.data_ptr(), fc.b.data_ptr(), None, ys.data_ptr(), 1, 1, 1000, 12544, 1024, 1, 1, 1, 0,
                                 fc.kp, 1, small.data_ptr(), small.numel(), None)
    assert rc != 0 and b"workspace" in lib.opnet_last_error()


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
    with pytest.raises(RuntimeError):
        CaterObjectDetector("x.pth")(np.zeros((240, 320, 3), np.uint8), torch.device("cuda:0"))     # load_model() first


# ---- RPN / RoIAlign / box heads / detections -------------------------------------------------------
MIN_SIZE, MAX_SIZE = 128, 200          # small resize target so the CPU restatement finishes in seconds


@pytest.fixture(scope="module")
def det_case():
    from objectpermanence_amd.detector import CaterObjectDetector
    params = {**do.synth_backbone_params(), **do.synth_head_params()}
    frame = np.random.default_rng(1).integers(0, 256, size=(60, 80, 3), dtype=np.uint8)
    det = CaterObjectDetector(None, min_size=MIN_SIZE, max_size=MAX_SIZE)
    det.load_state_dict(params, "cuda:0")
    ref_det, ref = do.detector_forward(frame, params, MIN_SIZE, MAX_SIZE)
    return det, params, frame, ref_det, ref


def _hip_stages(det, frame):
    from objectpermanence_amd.detector import preprocess_frame, resized_size
    x = preprocess_frame(frame, "cuda:0", MIN_SIZE, MAX_SIZE)
    feats = det.backbone.forward_nhwc(x)
    image_size = resized_size(frame.shape[0], frame.shape[1], MIN_SIZE, MAX_SIZE)
    head = det.heads.rpn_head(feats)
    props, pscores, count = det.heads.proposals(head, image_size, x.shape[1:3])          # batched over the images of `head`: here one
    return x, feats, image_size, head, props[0], pscores[0], count


def test_rpn_head_and_proposals_match_oracle(det_case):
    det, params, frame, _, ref = det_case
    x, feats, image_size, head, props, pscores, count = _hip_stages(det, frame)
    torch.cuda.synchronize()
    for got, want in zip(head, ref["rpn_head"]):
        assert got.shape[1:] == want.shape
        assert np.abs(got[0].cpu().numpy() - want).max() < 2e-3 * max(1.0, np.abs(want).max())
        assert float(got[..., 15].abs().max()) == 0.0
    # the discrete stage on IDENTICAL inputs (the HIP head outputs): same proposals in the same order
    want_b, want_s, _ = do.rpn_proposals([h[0].cpu().numpy() for h in head], image_size, x.shape[1:3])
    n = int(count.item())
    assert n == want_b.shape[0] and n > 100
    assert np.array_equal(pscores[:n].cpu().numpy(), want_s)
    assert np.abs(props[:n].cpu().numpy() - want_b).max() < 1e-3
    assert float(props[n:].abs().sum()) == 0.0


def test_rpn_proposals_tiny_and_capped():
    """fewer anchors than pre_nms_top_n on every level, post_nms_top_n smaller than the survivors"""
    from objectpermanence_amd.detector import FasterRCNNHeads
    heads = FasterRCNNHeads(do.synth_head_params(), "cuda:0", post_nms_top_n=7)
    rng = np.random.default_rng(3)
    outs = [rng.normal(0, 1.0, size=(1, h, w, 16)).astype(np.float32) for h, w in ((6, 9), (3, 5), (2, 3))]
    for o in outs:
        o[..., 3:15] *= 0.3
    props, scores, count = heads.proposals([torch.from_numpy(o).cuda() for o in outs], (40, 70), (48, 72))
    props, scores = props[0], scores[0]
    want_b, want_s, _ = do.rpn_proposals([o[0] for o in outs], (40, 70), (48, 72), post_nms_top_n=7)
    assert int(count.item()) == 7 == want_b.shape[0]
    assert np.array_equal(scores.cpu().numpy(), want_s)
    assert np.abs(props.cpu().numpy() - want_b).max() < 1e-4


def test_multiscale_roi_align_matches_oracle(det_case):
    det, _, frame, _, _ = det_case
    x, feats, image_size, *_ = _hip_stages(det, frame)
    rng = np.random.default_rng(5)
    n = 300
    # boxes from a few pixels to several times the image, so all four levels and the clamped / outside
    # sampling branches are exercised
    cx, cy = rng.uniform(-10, image_size[1] + 10, n), rng.uniform(-10, image_size[0] + 10, n)
    w, h = np.exp(rng.uniform(np.log(2), np.log(900), n)), np.exp(rng.uniform(np.log(2), np.log(900), n))
    rois = np.stack([cx - w / 2, cy - h / 2, cx + w / 2, cy + h / 2], axis=1).astype(np.float32)
    lv = do.map_levels(rois)
    assert set(lv.tolist()) == {0, 1, 2, 3}
    count = torch.tensor([n - 20], dtype=torch.int32, device="cuda:0")
    fl = list(feats.values())
    got = det.heads.roi_align(fl, torch.from_numpy(rois).cuda(), count, image_size).cpu().numpy()
    want = do.multiscale_roi_align([f[0].cpu().numpy() for f in fl[:4]], rois, image_size)
    scale = np.abs(want).max()
    assert scale > 0.1
    assert np.abs(got[:n - 20] - want[:n - 20]).max() < 1e-4 * scale
    assert np.abs(got[n - 20:]).max() == 0.0


def test_box_heads_and_detections_match_oracle(det_case):
    det, params, frame, _, ref = det_case
    x, feats, image_size, head, props, pscores, count = _hip_stages(det, frame)
    pooled = det.heads.roi_align(list(feats.values()), props, count, image_size)
    cls, reg = det.heads.box_heads(pooled)
    boxes, scores, labels, n_det = (t[0] for t in det.heads.detections(cls, reg, props, count, image_size, frame.shape[:2]))
    torch.cuda.synchronize()
    n = int(count.item())
    want_cls, want_reg = do.box_heads_forward(pooled.cpu().numpy()[:n], params)
    assert np.abs(cls[:n].cpu().numpy() - want_cls).max() < 2e-4 * np.abs(want_cls).max()
    assert np.abs(reg[:n].cpu().numpy() - want_reg).max() < 2e-4 * np.abs(want_reg).max()
    # discrete stage on identical inputs
    want = do.postprocess_detections(cls[:n].cpu().numpy(), reg[:n].cpu().numpy(), props[:n].cpu().numpy(), image_size,
                                     frame.shape[:2])
    k = int(n_det.item())
    assert k == want["boxes"].shape[0] == 100
    got_l, got_s, got_b = labels[:k].cpu().numpy(), scores[:k].cpu().numpy(), boxes[:k].cpu().numpy()
    assert np.all(np.diff(got_s) <= 0)
    same = got_l == want["labels"]
    assert same.mean() >= 0.97                       # a 1-ulp softmax difference may swap two near-tied scores
    assert np.abs(got_s - want["scores"]).max() < 1e-5
    assert np.abs(got_b[same] - want["boxes"][same]).max() < 1e-3


def _match(got, want, tol=0.05):
    """fraction of reference detections with a same-label detection within tol px"""
    hits = 0
    for b, l in zip(want["boxes"], want["labels"]):
        cand = got["boxes"][got["labels"] == l]
        hits += bool(len(cand) and np.abs(cand - b).max(axis=1).min() < tol)
    return hits / max(1, len(want["labels"]))


def test_detector_call_end_to_end(det_case):
    det, params, frame, ref_det, _ = det_case
    out = det(frame, torch.device("cuda:0"))
    assert isinstance(out, list) and len(out) == 1 and set(out[0]) == {"boxes", "labels", "scores"}
    got = {k: v.cpu().numpy() for k, v in out[0].items()}
    assert got["labels"].dtype == np.int64 and got["boxes"].shape == (len(got["scores"]), 4)
    assert np.all(np.diff(got["scores"]) <= 0)
    # end to end the continuous stages differ by ~1e-4 between fp32 MFMA and the fp64 restatement, which can flip
    # individual top-k / NMS decisions; most detections must still coincide
    assert _match(got, ref_det) >= 0.9
    from objectpermanence_amd.detector import CaterObjectDetector
    kept = CaterObjectDetector.remove_low_probability_object(out[0], 0.8)
    assert len(kept["scores"]) == int((got["scores"] >= 0.8).sum()) > 0
    # batched call vs frame-by-frame call: the dense stages may pick another tile shape (summation order), so the
    # comparison is the same "most detections coincide" as against the restatement
    frame2 = np.random.default_rng(2).integers(0, 256, size=(60, 80, 3), dtype=np.uint8)
    both = det.detect_batch([frame, frame2], torch.device("cuda:0"))
    single2 = det(frame2, torch.device("cuda:0"))[0]
    for b, s in ((both[0], out[0]), (both[1], single2)):
        b, s = ({k: v.cpu().numpy() for k, v in d.items()} for d in (b, s))
        assert _match(b, s) >= 0.95 and _match(s, b) >= 0.95


def test_selection_stages_batched_equal_per_image(det_case):
    """One launch per stage over the images of a pass (missing 4) gives each image exactly what its own one-image
    call gives, on identical dense inputs: proposals, RoIAlign rows, detections - bit for bit - for images whose candidate counts
    differ (one of them a constant frame: few survivors)."""
    det, params, frame, _, _ = det_case
    rng = np.random.default_rng(11)
    frames = [frame, rng.integers(0, 256, size=frame.shape, dtype=np.uint8), np.full(frame.shape, 37, dtype=np.uint8),
              rng.integers(0, 64, size=frame.shape, dtype=np.uint8), frame[::-1].copy()]
    dev = torch.device("cuda:0")
    feats = det.backbone_features_batch(frames, dev)
    from objectpermanence_amd.detector import preprocess_frame, resized_size
    image_size = resized_size(frame.shape[0], frame.shape[1], MIN_SIZE, MAX_SIZE)
    padded = tuple(preprocess_frame(frame, dev, MIN_SIZE, MAX_SIZE).shape[1:3])
    heads = det.heads
    head = heads.rpn_head(feats)
    maps = list(feats.values())
    n = len(frames)
    props, pscores, count = heads.proposals(head, image_size, padded)
    pooled = heads.roi_align(maps, props, count, image_size)
    cls, reg = heads.box_heads(pooled)
    boxes, scores, labels, n_det = heads.detections(cls, reg, props, count, image_size, frame.shape[:2])
    torch.cuda.synchronize()
    r = props.shape[1]
    assert len(set(count.tolist())) > 1 and len(set(n_det.tolist())) >= 1
    for i in range(n):
        p1, s1, c1 = heads.proposals([h[i:i + 1].contiguous() for h in head], image_size, padded)
        assert int(c1) == int(count[i]) and torch.equal(p1[0], props[i]) and torch.equal(s1[0], pscores[i])
        ra = heads.roi_align([m[i:i + 1].contiguous() for m in maps], p1, c1, image_size)
        assert torch.equal(ra, pooled[i * r:(i + 1) * r])
        b1, sc1, l1, nd1 = heads.detections(cls[i * r:(i + 1) * r], reg[i * r:(i + 1) * r], p1, c1, image_size, frame.shape[:2])
        assert int(nd1) == int(n_det[i])
        assert torch.equal(b1[0], boxes[i]) and torch.equal(sc1[0], scores[i]) and torch.equal(l1[0], labels[i])
    # and through the product entry point: a pass of frames = the same frames one by one where the dense stages agree bit for bit
    # (they may not - another tile shape - so only "most detections coincide" is asked there: test_detector_end_to_end)


def test_detector_full_size_properties():
    """240x320 frame -> 800x1066 (the reference's call): shapes, ordering, box bounds, determinism"""
    from objectpermanence_amd.detector import CaterObjectDetector
    params = {**do.synth_backbone_params(), **do.synth_head_params()}
    det = CaterObjectDetector(None)
    det.load_state_dict(params, "cuda:0")
    frame = np.random.default_rng(4).integers(0, 256, size=(240, 320, 3), dtype=np.uint8)
    a = det(frame, torch.device("cuda:0"))[0]
    b = det(frame, torch.device("cuda:0"))[0]
    for k in a:
        assert torch.equal(a[k], b[k])
    n = len(a["scores"])
    assert 0 < n <= 100 and a["boxes"].shape == (n, 4)
    bx = a["boxes"].cpu().numpy()
    assert bx.min() >= 0 and bx[:, [0, 2]].max() <= 320.0 + 1e-3 and bx[:, [1, 3]].max() <= 240.0 + 1e-3
    assert np.all(bx[:, 2] > bx[:, 0]) and np.all(bx[:, 3] > bx[:, 1])
    lab = a["labels"].cpu().numpy()
    assert lab.min() >= 1 and lab.max() <= 192
    assert np.all(np.diff(a["scores"].cpu().numpy()) <= 0) and float(a["scores"][-1]) > 0.05


def test_detector_no_detections_and_perception_lists():
    """a detector that calls everything background: empty outputs all the way to the per-frame lists of the pkl"""
    from objectpermanence_amd.detector import CaterObjectDetector
    from objectpermanence_amd.preprocess_perception_main import output_video_predictions
    params = {**do.synth_backbone_params(), **do.synth_head_params()}
    params["roi_heads.box_predictor.cls_score.bias"] = params["roi_heads.box_predictor.cls_score.bias"].copy()
    params["roi_heads.box_predictor.cls_score.bias"][0] = 80.0           # background wins every softmax
    det = CaterObjectDetector(None, min_size=MIN_SIZE, max_size=MAX_SIZE)
    det.load_state_dict(params, "cuda:0")
    frames = np.random.default_rng(9).integers(0, 256, size=(3, 60, 80, 3), dtype=np.uint8)
    out = det(frames[0], torch.device("cuda:0"))[0]
    assert out["boxes"].shape == (0, 4) and out["labels"].shape == (0,) and out["scores"].shape == (0,)
    assert out["labels"].dtype == torch.int64
    kept = det.remove_low_probability_object(out)
    assert kept["boxes"].shape == (0, 4)
    bb, lab = output_video_predictions(frames, det, torch.device("cuda:0"), frames_per_pass=2)
    assert len(bb) == len(lab) == 3 and all(b.shape == (0, 4) and l.shape == (0,) for b, l in zip(bb, lab))
