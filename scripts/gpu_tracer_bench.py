"""Throughput of the stand-alone ray-queue tracers (sr_trace_closest / sr_trace_any) on the bench scene:
primary rays, cosine-distributed bounce rays from the primary hit points, shadow rays to the lights."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from sunray_amd import abi, scenes, runtime as rt
W, H = 1920, 1080
desc = scenes.heightfield(708)
sc = rt.Scene(0).load(desc)
m = rt.camera_matrices(desc.camera_pos, desc.camera_target, desc.fov_y, W, H)
dev = "cuda:0"
vi = torch.tensor(list(m.view_inverse), device=dev).reshape(4, 4)
pi = torch.tensor(list(m.proj_inverse), device=dev).reshape(4, 4)
py, px = torch.meshgrid(torch.arange(H, device=dev), torch.arange(W, device=dev), indexing="ij")
# 8x8-tiled order like the passes' waves
def tile_order(a):
    return a.reshape(H // 8, 8, W // 8, 8).permute(0, 2, 1, 3).reshape(-1)
px, py = tile_order(px), tile_order(py)
dx = (px + 0.5) / W * 2 - 1; dy = (py + 0.5) / H * 2 - 1
tgt = torch.stack([dx, dy, torch.ones_like(dx), torch.ones_like(dx)], 1) @ pi.T
t3 = torch.nn.functional.normalize(tgt[:, :3], dim=1)
d = (torch.cat([t3, torch.zeros(len(t3), 1, device=dev)], 1) @ vi.T)[:, :3]
o = vi[:3, 3].expand(len(d), 3)
def rays(o, d, tmin, tmax):
    n = len(o)
    r = torch.empty(n, 8, device=dev)
    r[:, 0:3] = o; r[:, 3] = tmin; r[:, 4:7] = d; r[:, 7] = tmax if not torch.is_tensor(tmax) else 0
    if torch.is_tensor(tmax): r[:, 7] = tmax
    return r.contiguous()
def timed(kind, fn, n, reps=5):
    fn(); torch.cuda.synchronize()
    sc.enable_timing(True)
    for _ in range(reps): fn()
    ms, k = sc.read_timing(kind); sc.enable_timing(False)
    return ms / k, n / (ms / k) / 1e3
N = W * H
prim = rays(o, d, 0.001, 10000.0)
hits = sc.trace_closest(prim, N)
ms, mr = timed(2, lambda: sc.trace_closest(prim, N), N); print("primary closest   %.3f ms  %.0f Mray/s" % (ms, mr))
t = hits[:, 0:1]
ok = (t[:, 0] > 0)
p = o + d * t
g = torch.Generator(device=dev); g.manual_seed(1)
r1 = torch.rand(N, generator=g, device=dev); r2 = torch.rand(N, generator=g, device=dev)
phi = 2 * np.pi * r1; rr = torch.sqrt(r2)
bd = torch.stack([rr * torch.cos(phi), torch.sqrt(1 - r2), rr * torch.sin(phi)], 1)  # cosine around +y
bo = p + torch.tensor([0.0, 0.002, 0.0], device=dev)
bounce = rays(bo[ok], bd[ok], 0.001, 10000.0); nb = len(bounce)
ms, mr = timed(2, lambda: sc.trace_closest(bounce, nb), nb); print("bounce closest    %.3f ms  %.0f Mray/s (%d rays)" % (ms, mr, nb))
ms, mr = timed(3, lambda: sc.trace_any(bounce, nb), nb); print("bounce any        %.3f ms  %.0f Mray/s" % (ms, mr))
lights = torch.tensor([[7.5 * np.cos(2 * np.pi * k / 8), 7.0, 7.5 * np.sin(2 * np.pi * k / 8)] for k in range(8)], device=dev, dtype=torch.float32)
li = torch.randint(0, 8, (N,), generator=g, device=dev)
lp = lights[li] + (torch.rand(N, 3, generator=g, device=dev) - 0.5) * torch.tensor([2.4, 0.0, 2.4], device=dev)
sd = lp - bo; dist = sd.norm(dim=1); sd = sd / dist[:, None]
shadow = rays(bo[ok], sd[ok], 0.001, (dist - 0.001)[ok])
ms, mr = timed(3, lambda: sc.trace_any(shadow, nb), nb); print("shadow any        %.3f ms  %.0f Mray/s" % (ms, mr))
perm = torch.randperm(nb, generator=g, device=dev)
shuf = shadow[perm].contiguous()
ms, mr = timed(3, lambda: sc.trace_any(shuf, nb), nb); print("shadow any (shuffled) %.3f ms  %.0f Mray/s" % (ms, mr))
sc.set_instrumented(True); sc.reset_counters(); sc.trace_closest(prim, N); c = sc.counters(); print("primary: boxes/ray %.1f tris/ray %.2f" % (c.boxes_tested / N, c.tris_tested / N))
sc.reset_counters(); sc.trace_closest(bounce, nb); c = sc.counters(); print("bounce : boxes/ray %.1f tris/ray %.2f" % (c.boxes_tested / nb, c.tris_tested / nb))
sc.reset_counters(); sc.trace_any(shadow, nb); c = sc.counters(); print("shadow : boxes/ray %.1f tris/ray %.2f" % (c.boxes_tested / nb, c.tris_tested / nb))
