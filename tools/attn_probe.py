"""Time opseq_attention_f32: python tools/attn_probe.py [S:E:heads ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from objectpermanence_amd import _lib
lib = _lib.load()
cases = [tuple(int(v) for v in a.split(":")) for a in sys.argv[1:]] or [(300, 256, 4), (2400, 256, 4), (9600, 256, 2), (9600, 256, 4), (19200, 256, 2), (19200, 256, 4)]
for S, E, nh in cases:
    qkv = torch.randn((S, 3 * E), device="cuda:0")
    out = torch.empty((S, E), device="cuda:0")
    st = torch.cuda.current_stream().cuda_stream
    ws = torch.empty(lib.opseq_attention_workspace_bytes(S, E, nh) if os.environ.get("NOSPLIT") is None else 16, dtype=torch.uint8, device="cuda:0")
    run = lambda: _lib.check(lib.opseq_attention_f32(qkv.data_ptr(), out.data_ptr(), S, E, nh, ws.data_ptr() if ws.numel() > 16 else None, ws.numel(), st), "att")
    for _ in range(3): run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 20 if S < 5000 else 5
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps): run()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    fl = 4.0 * S * S * E
    print(f"S={S:6d} E={E} heads={nh} hd={E // nh:3d}: {ms * 1e3:9.1f} us  {fl / ms / 1e9:7.1f} TFLOP/s  ({fl / ms / 1e9 / 157.3:.2f} of fp32 MFMA peak)", flush=True)
