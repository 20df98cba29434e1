"""Times the detector's radix sort by itself (the test-only entry) at the detector's two sizes; GPU box only."""
import ctypes
import torch
from objectpermanence_amd import _lib

lib = _lib.load()
dev = torch.device("cuda:0")
for n, bits, key64 in [(217413, 35, 1), (192000, 32, 0), (4663, 32, 0)]:
    g = torch.Generator().manual_seed(1)
    if key64:
        keys = torch.randint(0, 2**35, (n,), generator=g, dtype=torch.int64).to(dev)
    else:
        keys = torch.randint(-2**31, 2**31, (n,), generator=g, dtype=torch.int64).to(torch.int32).to(dev)
    k_in = keys.clone(); k_out = torch.empty_like(keys)
    v_in = torch.arange(n, dtype=torch.int32, device=dev); v_out = torch.empty_like(v_in)
    scratch = torch.empty(lib.opdet_test_sort_scratch_bytes(n), dtype=torch.uint8, device=dev)
    where = ctypes.c_int(0)
    def run():
        rc = lib.opdet_test_sort_pairs(k_in.data_ptr(), v_in.data_ptr(), k_out.data_ptr(), v_out.data_ptr(), n, bits, key64,
                                       scratch.data_ptr(), scratch.numel(), ctypes.byref(where), None)
        assert rc == 0
    for _ in range(5):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        run()
    e1.record(); torch.cuda.synchronize()
    print(f"n={n} bits={bits} key64={key64}: {e0.elapsed_time(e1) / 50 * 1e3:.1f} us per sort")
