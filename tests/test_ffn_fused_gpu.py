"""The encoder's feed-forward block as one kernel (csrc/ffn_kernels.hip, opseq_ffn_fused_f32): linear2(relu(linear1(x))) of
nn.TransformerEncoderLayer in eval mode (reference baselines/learned_models.py:166-171 builds the layers, :184 runs them).

Checked (i) against the plain fp32 torch statement of the same block, (ii) bit for bit against the two token-wise products it
replaces inside opseq_encoder_layer_batched_f32 (OPSEQ_FFN_FUSED=0 keeps them), on tile plans of every kind - one short tile, full
rounds of 64-token tiles, balanced tail tiles of 16 / 32 / 48 tokens, ragged row counts."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
E, FFN = 256, 2048


def _lib():
    from objectpermanence_amd import _lib
    return _lib, _lib.load()


def _weights(seed, ffn=FFN):
    g = torch.Generator().manual_seed(seed)
    w1 = (torch.rand(ffn, E, generator=g) - 0.5) * (2.0 / E ** 0.5)
    b1 = (torch.rand(ffn, generator=g) - 0.5) * 0.2
    w2 = (torch.rand(E, ffn, generator=g) - 0.5) * (2.0 / ffn ** 0.5)
    b2 = (torch.rand(E, generator=g) - 0.5) * 0.2
    return [t.cuda().contiguous() for t in (w1, b1, w2, b2)]


def _fused(x, w1, b1, w2, b2):
    binding, lib = _lib()
    y = torch.full_like(x, float("nan"))
    rc = lib.opseq_ffn_fused_f32(x.data_ptr(), w1.data_ptr(), b1.data_ptr(), w2.data_ptr(), b2.data_ptr(), y.data_ptr(),
                                 x.shape[0], x.shape[1], w1.shape[0], torch.cuda.current_stream().cuda_stream)
    binding.check(rc, "opseq_ffn_fused_f32")
    torch.cuda.synchronize()
    return y


# 1 .. 63: one short tile; 64 * 256 = 16 384 rows are one full round of an MI355X; 16 384 + 256 * 16 k: tails of k fragments
@pytest.mark.parametrize("M", [1, 15, 16, 17, 63, 64, 65, 300, 4800, 16384, 16384 + 4096 - 5, 16384 + 8192 + 3, 16384 + 12288, 40000])
def test_fused_block_against_the_torch_statement(M):
    w1, b1, w2, b2 = _weights(M)
    x = torch.randn(M, E, generator=torch.Generator().manual_seed(M + 1)).cuda()
    y = _fused(x, w1, b1, w2, b2)
    ref = torch.relu(x.double() @ w1.double().T + b1.double()) @ w2.double().T + b2.double()
    err = (y.double() - ref).abs().max().item()
    assert torch.isfinite(y).all()
    assert err < 2e-5 * max(1.0, ref.abs().max().item()), err      # fp32 sums over 256 and 2 048 terms against fp64


@pytest.mark.parametrize("ffn,M", [(128, 200), (128, 20000), (384, 777), (1024, 16384 + 33), (4096, 5000)])
def test_other_hidden_widths(ffn, M):
    """one chunk of 128 hidden units (the ring's first and last chunk at once), three, eight, thirty-two"""
    w1, b1, w2, b2 = _weights(ffn + M, ffn)
    x = torch.randn(M, E, generator=torch.Generator().manual_seed(M)).cuda()
    y = _fused(x, w1, b1, w2, b2)
    ref = torch.relu(x.double() @ w1.double().T + b1.double()) @ w2.double().T + b2.double()
    assert torch.isfinite(y).all()
    assert (y.double() - ref).abs().max().item() < 2e-5 * max(1.0, ref.abs().max().item())


def test_rows_do_not_depend_on_where_they_sit():
    """a token row's result is the same bits alone, inside a short tile, inside a 64-token tile and inside a tail tile"""
    w1, b1, w2, b2 = _weights(3)
    x = torch.randn(16384 + 4096 + 11, E, generator=torch.Generator().manual_seed(5)).cuda()
    y = _fused(x, w1, b1, w2, b2)
    for lo, hi in ((0, 1), (16384, 16384 + 40), (20000, 20491), (7, 7 + 64 * 3)):
        assert torch.equal(_fused(x[lo:hi].contiguous(), w1, b1, w2, b2), y[lo:hi]), (lo, hi)


def test_two_streams_at_once_give_the_sequential_results():
    """two launches on two streams share the chip tile by tile (one 144-KB workgroup per CU): each still returns its own result"""
    binding, lib = _lib()
    w1, b1, w2, b2 = _weights(9)
    xs = [torch.randn(m, E, generator=torch.Generator().manual_seed(m)).cuda() for m in (30000, 20011)]
    ref = [_fused(x, w1, b1, w2, b2) for x in xs]
    streams = [torch.cuda.Stream() for _ in xs]
    for rep in range(3):
        ys = [torch.full_like(x, float("nan")) for x in xs]
        torch.cuda.synchronize()
        for x, y, st in zip(xs, ys, streams):
            rc = lib.opseq_ffn_fused_f32(x.data_ptr(), w1.data_ptr(), b1.data_ptr(), w2.data_ptr(), b2.data_ptr(), y.data_ptr(),
                                         x.shape[0], E, FFN, st.cuda_stream)
            binding.check(rc, "opseq_ffn_fused_f32")
        torch.cuda.synchronize()
        for y, r in zip(ys, ref):
            assert torch.equal(y, r)


def test_shapes_outside_the_kernel_are_refused():
    binding, lib = _lib()
    assert lib.opseq_ffn_fused_supported(300, 256, 2048) == 1
    assert lib.opseq_ffn_fused_supported(300, 128, 2048) == 0      # E != 256
    assert lib.opseq_ffn_fused_supported(300, 256, 2000) == 0      # ffn not a multiple of 128
    assert lib.opseq_ffn_fused_supported(1 << 22, 256, 2048) == 0  # x at 2 GiB and beyond
    w1, b1, w2, b2 = _weights(1)
    x = torch.zeros(4, 128).cuda()
    rc = lib.opseq_ffn_fused_f32(x.data_ptr(), w1.data_ptr(), b1.data_ptr(), w2.data_ptr(), b2.data_ptr(), x.data_ptr(), 4, 128, FFN, 0)
    assert rc == -2                                                # OPNET_ESHAPE


# (rows -> tail tiles of the single products: 12 000 -> 48 tokens, 21 000 -> 32, 19 400 -> 16 after one full round, 4 200 / 2 100 / 150: conv tiles)
@pytest.mark.parametrize("heads,n_seg,S", [(4, 40, 300), (2, 14, 300), (4, 70, 300), (4, 388, 50), (2, 7, 300), (4, 3, 50)])
def test_encoder_layer_with_the_fused_block_against_the_two_products(heads, n_seg, S):
    """opseq_encoder_layer_batched_f32 with and without the fused kernel: the same words in z wherever the two products run in the
    ascending-K order of conv2d_nhwc_glds / gemm_bias_act (4 033 rows and more); below that linear2 is a K-split product
    (gemm_bias_act_ks: four K quarters summed) and the two agree to rounding"""
    binding, lib = _lib()
    g = torch.Generator().manual_seed(heads * 1000 + n_seg)
    rnd = lambda *shape, scale=1.0: ((torch.rand(*shape, generator=g) - 0.5) * 2 * scale).cuda().contiguous()
    in_w, in_b = rnd(3 * E, E, scale=E ** -0.5), rnd(3 * E, scale=0.1)
    out_w, out_b = rnd(E, E, scale=E ** -0.5), rnd(E, scale=0.1)
    l1_w, l1_b, l2_w, l2_b = _weights(n_seg)
    n1_w, n1_b, n2_w, n2_b = 1 + rnd(E, scale=0.1), rnd(E, scale=0.1), 1 + rnd(E, scale=0.1), rnd(E, scale=0.1)
    z0 = rnd(n_seg * S, E)
    ws = torch.empty(lib.opseq_encoder_workspace_bytes(n_seg * S, E, heads, FFN) // 4, device="cuda")
    out = {}
    old = os.environ.get("OPSEQ_FFN_FUSED"), os.environ.get("OPSEQ_GEMM_W8")
    try:
        for flag in ("0", "1"):
            # ... and the layer's other K = 256 products (input / output projection) on gemm_k256_w8 from 8 192 rows on: the same bits again
            os.environ["OPSEQ_FFN_FUSED"] = os.environ["OPSEQ_GEMM_W8"] = flag
            z = z0.clone()
            rc = lib.opseq_encoder_layer_batched_f32(z.data_ptr(), *(t.data_ptr() for t in (in_w, in_b, out_w, out_b, l1_w, l1_b, l2_w, l2_b,
                                                                                          n1_w, n1_b, n2_w, n2_b)),
                                                     ws.data_ptr(), ws.numel() * 4, S, n_seg, E, heads, FFN, torch.cuda.current_stream().cuda_stream)
            binding.check(rc, "opseq_encoder_layer_batched_f32")
            torch.cuda.synchronize()
            out[flag] = z
    finally:
        for name, v in zip(("OPSEQ_FFN_FUSED", "OPSEQ_GEMM_W8"), old):
            if v is None:
                os.environ.pop(name, None)
            else:
                os.environ[name] = v
    assert torch.isfinite(out["1"]).all() and not torch.equal(out["1"], z0)
    if ((n_seg * S + 63) // 64) * 4 >= 256:
        assert torch.equal(out["0"], out["1"])
    else:
        assert (out["0"] - out["1"]).abs().max().item() < 1e-5


def test_a_served_throughput_pass_still_matches_its_lone_forwards():
    """transformer_lstm, 48 one-clip requests in one pass of the throughput form (the fused block inside): within 1e-5 of the lone forwards"""
    from objectpermanence_amd import ModelsFactory
    from oracle import synth
    cfg = {"boxes_features_dim": 256, "num_attention_heads": 4, "num_attention_layers": 2, "num_lstm_layers": 2, "lstm_hidden_dim": 512}
    m = ModelsFactory.get_model("transformer_lstm", cfg)
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in synth.transformer_lstm_synth_params(cfg).items()})
    m = m.eval().to("cuda:0")
    x = torch.from_numpy(synth.boxes5(synth.make_batch(11, 48, 60)[0])).cuda()
    with torch.no_grad():
        merged = m.forward_segments(x, 48, exact=False)
        alone = torch.cat([m(x[r:r + 1]) for r in (0, 17, 47)])
    torch.cuda.synchronize()
    assert np.abs((merged[[0, 17, 47]] - alone).cpu().numpy()).max() < 1e-5


def test_a_throughput_pass_is_the_same_bits_with_and_without_the_resident_token_kernels():
    """transformer_lstm, 48 one-clip requests of 300 frames in one throughput pass (14 400 token rows): the encoder's products and the
    hoisted layer-0 input product of the stacked LSTM on gemm_k256_w8 / ffn_fused_w8, or all of them on conv2d_nhwc_glds"""
    from objectpermanence_amd import ModelsFactory
    from oracle import synth
    cfg = {"boxes_features_dim": 256, "num_attention_heads": 4, "num_attention_layers": 2, "num_lstm_layers": 2, "lstm_hidden_dim": 512}
    m = ModelsFactory.get_model("transformer_lstm", cfg)
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in synth.transformer_lstm_synth_params(cfg).items()})
    m = m.eval().to("cuda:0")
    x = torch.from_numpy(synth.boxes5(synth.make_batch(3, 48, 300)[0])).cuda()
    old = os.environ.get("OPSEQ_FFN_FUSED"), os.environ.get("OPSEQ_GEMM_W8")
    out = {}
    try:
        for flag in ("0", "1"):
            os.environ["OPSEQ_FFN_FUSED"] = os.environ["OPSEQ_GEMM_W8"] = flag
            with torch.no_grad():
                out[flag] = m.forward_segments(x, 48, exact=False).clone()
            torch.cuda.synchronize()
    finally:
        for name, v in zip(("OPSEQ_FFN_FUSED", "OPSEQ_GEMM_W8"), old):
            if v is None:
                os.environ.pop(name, None)
            else:
                os.environ[name] = v
    assert m._runner._monitor.verify() == 0
    assert torch.isfinite(out["1"]).all() and torch.equal(out["0"], out["1"])
