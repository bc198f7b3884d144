"""One direct conv layer (plain / masked), a few launches (for rocprofv3 --pmc): python tools/exp/one_layer_direct.py N H Cin Cout [masked]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import pggan_amd as pg
ops = pg.ops
N, H, ci, co = [int(v) for v in sys.argv[1:5]]
masked = len(sys.argv) > 5
x = torch.randn(N, H, H, ci, device='cuda'); w = torch.randn(3, 3, co, ci, device='cuda') * 0.05; b = torch.randn(co, device='cuda')
m = torch.randint(0, 16, (N, H, H, co // 4), device='cuda', dtype=torch.uint8) if masked else None
y = torch.empty(N, H, H, co, device='cuda')
for _ in range(5):
    if masked:
        ops.conv2d(x, w, None, N, H, H, 3, 1, 0.5, mask=m, out=y)
    else:
        ops.conv2d(x, w, b, N, H, H, 3, 1, 0.5, 0.2, out=y)
torch.cuda.synchronize()
