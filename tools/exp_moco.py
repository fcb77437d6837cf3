#!/usr/bin/env python3
"""Back-to-back C-ABI launches of the MoCo forward / backward (no Python autograd in the loop):
kernel time without host gaps between launches."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torch.nn.functional as F
from image_classification_amd import _lib

b, K, D, T = 256, 65536, 128, 0.1
dev = torch.device("cuda:0")
L = _lib.lib()
q = F.normalize(torch.randn(b, D, device=dev), dim=1)
k = F.normalize(torch.randn(b, D, device=dev), dim=1)
mem = F.normalize(torch.rand(K, D, device=dev), dim=1)
ws_bytes = int(L.mi_moco_workspace_bytes(b, K, D))
ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
lse, lpos, loss = (torch.empty(n, device=dev) for n in (b, b, 1))
g = torch.ones(1, device=dev)
dq = torch.empty_like(q)
st = _lib.stream_ptr(dev)

def fwd():
    _lib.check(L.mi_moco_fwd(q.data_ptr(), k.data_ptr(), mem.data_ptr(), b, K, D, T, lse.data_ptr(), lpos.data_ptr(), loss.data_ptr(), ws.data_ptr(), ws_bytes, st), "fwd")
def bwd():
    _lib.check(L.mi_moco_bwd(q.data_ptr(), k.data_ptr(), mem.data_ptr(), b, K, D, T, lse.data_ptr(), lpos.data_ptr(), g.data_ptr(), dq.data_ptr(), None, None, 0, 0, ws.data_ptr(), ws_bytes, st), "bwd")

for name, fn in (("fwd", fwd), ("bwd", bwd)):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 2000
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    print(f"{name}: {ms * 1e3:.1f} us/call  {K * D * 4 / ms / 1e6:.0f} GB/s queue read  {(2 if name == 'fwd' else 4) * b * K * D / ms / 1e9:.1f} TFLOP/s")
