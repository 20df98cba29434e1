"""The detector's own stable radix sort (csrc/det_sort_kernels.hip) against torch.sort(stable=True) on the host: it orders the RPN
candidates and the detections (torchvision 0.5.0 filter_proposals / postprocess_detections, reached from baselines/detector.py:84), so
bit-exact keys AND the order of equal keys (the values) are the bar."""
import ctypes

import numpy as np
import pytest
import torch

from objectpermanence_amd import _lib

pytestmark = pytest.mark.gpu


def _sort(keys: torch.Tensor, bits: int):
    lib = _lib.load()
    n = keys.numel()
    key64 = keys.dtype == torch.int64
    dev = torch.device("cuda:0")
    k_in = keys.to(dev).clone()
    v_in = torch.arange(n, dtype=torch.int32, device=dev)
    k_out = torch.empty_like(k_in)
    v_out = torch.empty_like(v_in)
    scratch = torch.empty(lib.opdet_test_sort_scratch_bytes(n), dtype=torch.uint8, device=dev)
    where = ctypes.c_int(-1)
    rc = lib.opdet_test_sort_pairs(k_in.data_ptr(), v_in.data_ptr(), k_out.data_ptr(), v_out.data_ptr(), n, bits, int(key64),
                                   scratch.data_ptr(), scratch.numel(), ctypes.byref(where), None)
    assert rc == 0, lib.opnet_last_error().decode()
    torch.cuda.synchronize()
    assert where.value in (0, 1)
    return (k_out, v_out) if where.value else (k_in, v_in)


def _check(keys: torch.Tensor, bits: int):
    k, v = _sort(keys, bits)
    # the expected order: stable argsort of the low `bits` bits
    if keys.dtype == torch.int64:
        low = keys & ((1 << bits) - 1) if bits < 63 else keys
    else:
        low = (keys.to(torch.int64) & 0xFFFFFFFF) & ((1 << bits) - 1)
    order = torch.sort(low, stable=True).indices
    assert torch.equal(v.cpu().to(torch.int64), order)
    assert torch.equal(k.cpu(), keys[order])


@pytest.mark.parametrize("n", [1, 63, 64, 65, 255, 256, 257, 1000, 2047, 2048, 2049, 4663, 8192])
def test_small_sort_u32(n):
    g = torch.Generator().manual_seed(n)
    keys = torch.randint(-2**31, 2**31, (n,), generator=g, dtype=torch.int64).to(torch.int32)
    _check(keys, 32)


@pytest.mark.parametrize("n", [8193, 10000, 192000, 2048 * 97, 2048 * 97 + 1, 2048 * 600 + 77])   # the last: more tiles than workgroups
def test_tiled_sort_u32(n):
    g = torch.Generator().manual_seed(n)
    keys = torch.randint(-2**31, 2**31, (n,), generator=g, dtype=torch.int64).to(torch.int32)
    _check(keys, 32)


@pytest.mark.parametrize("n", [5, 4096, 217413])
def test_sort_u64_35_bits(n):
    # the RPN keys: level in bits 32..34, the flipped objectness below; bits above 35 must be ignored
    g = torch.Generator().manual_seed(n)
    keys = torch.randint(0, 2**35, (n,), generator=g, dtype=torch.int64) | (torch.randint(0, 2, (n,), generator=g, dtype=torch.int64) << 40)
    _check(keys, 35)


@pytest.mark.parametrize("n", [300, 4663, 192000])
def test_ties_keep_input_order(n):
    # few distinct keys (the 0xffffffff padding of discarded candidates is the real case): values must stay in input order
    g = torch.Generator().manual_seed(7 + n)
    keys = torch.randint(0, 5, (n,), generator=g, dtype=torch.int64)
    keys = torch.where(keys == 4, torch.full_like(keys, -1), keys * 0x01010101).to(torch.int32)
    _check(keys, 32)
    _check(torch.full((n,), -1, dtype=torch.int32), 32)


def test_refusals():
    lib = _lib.load()
    where = ctypes.c_int(0)
    buf = torch.empty(64, dtype=torch.int32, device="cuda:0")
    assert lib.opdet_test_sort_pairs(buf.data_ptr(), buf.data_ptr(), buf.data_ptr(), buf.data_ptr(), 16, 33, 0, buf.data_ptr(), 256,
                                     ctypes.byref(where), None) != 0
    assert lib.opdet_test_sort_pairs(buf.data_ptr(), buf.data_ptr(), buf.data_ptr(), buf.data_ptr(), 100000, 32, 0, buf.data_ptr(), 256,
                                     ctypes.byref(where), None) != 0
    assert lib.opdet_test_sort_pairs(None, buf.data_ptr(), buf.data_ptr(), buf.data_ptr(), 16, 32, 0, buf.data_ptr(), 256,
                                     ctypes.byref(where), None) != 0


def test_many_sorts_in_a_row_agree():
    """300 sorts of fresh keys at the RPN's size, each checked (a race between the passes' launches or inside the wave-private
    ranking would show up as a wrong permutation)."""
    lib = _lib.load()
    dev = torch.device("cuda:0")
    n = 217413
    g = torch.Generator(device=dev).manual_seed(11)
    scratch = torch.empty(lib.opdet_test_sort_scratch_bytes(n), dtype=torch.uint8, device=dev)
    where = ctypes.c_int(-1)
    bad = 0
    for it in range(300):
        keys = torch.randint(0, 2**35, (n,), generator=g, dtype=torch.int64, device=dev)
        if it % 3 == 0:
            keys = keys & 0x7_0000_00FF          # heavy ties
        k_in, v_in = keys.clone(), torch.arange(n, dtype=torch.int32, device=dev)
        k_out, v_out = torch.empty_like(k_in), torch.empty_like(v_in)
        rc = lib.opdet_test_sort_pairs(k_in.data_ptr(), v_in.data_ptr(), k_out.data_ptr(), v_out.data_ptr(), n, 35, 1,
                                       scratch.data_ptr(), scratch.numel(), ctypes.byref(where), None)
        assert rc == 0
        k, v = (k_out, v_out) if where.value else (k_in, v_in)
        order = torch.sort(keys, stable=True).indices
        bad += int(not (torch.equal(v.to(torch.int64), order) and torch.equal(k, keys[order])))
    assert bad == 0
