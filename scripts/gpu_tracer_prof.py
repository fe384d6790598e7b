"""Runs only the primary-ray closest-hit trace (x6) for PMC profiling of the queue tracer."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from sunray_amd import scenes, runtime as rt
W, H = 1920, 1080
desc = scenes.heightfield(708)
sc = rt.Scene(0).load(desc)
m = rt.camera_matrices(desc.camera_pos, desc.camera_target, desc.fov_y, W, H)
dev = "cuda:0"
vi = torch.tensor(list(m.view_inverse), device=dev).reshape(4, 4)
pi = torch.tensor(list(m.proj_inverse), device=dev).reshape(4, 4)
py, px = torch.meshgrid(torch.arange(H, device=dev), torch.arange(W, device=dev), indexing="ij")
tile = lambda a: a.reshape(H // 8, 8, W // 8, 8).permute(0, 2, 1, 3).reshape(-1)
px, py = tile(px), tile(py)
dx = (px + 0.5) / W * 2 - 1; dy = (py + 0.5) / H * 2 - 1
tgt = torch.stack([dx, dy, torch.ones_like(dx), torch.ones_like(dx)], 1) @ pi.T
t3 = torch.nn.functional.normalize(tgt[:, :3], dim=1)
d = (torch.cat([t3, torch.zeros(len(t3), 1, device=dev)], 1) @ vi.T)[:, :3]
r = torch.empty(W * H, 8, device=dev)
r[:, 0:3] = vi[:3, 3]; r[:, 3] = 0.001; r[:, 4:7] = d; r[:, 7] = 10000.0
for _ in range(6):
    sc.trace_closest(r, W * H)
torch.cuda.synchronize()
