"""One Winograd conv layer, a few launches (for rocprofv3 --pmc): python tools/exp/one_layer.py N H Cin Cout [reps]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import pggan_amd as pg
ops = pg.ops
N, H, ci, co = [int(v) for v in sys.argv[1:5]]
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 5
x = torch.randn(N, H, H, ci, device='cuda'); w = torch.randn(3, 3, co, ci, device='cuda') * 0.05; b = torch.randn(co, device='cuda')
u = ops.wino_transform_weights(w)
y = ops.conv2d_wino(x, u, b, N, H, H, 0.5, 0.2)
for _ in range(reps):
    ops.conv2d_wino(x, u, b, N, H, H, 0.5, 0.2, out=y)
torch.cuda.synchronize()
