"""What would sorting secondary rays buy? The GI bounce rays of the bench frame (cosine-distributed directions from the primary
hit points) traced by the stand-alone tracer in three orders: as the passes issue them (pixel order in 8x8 tiles),
sorted by direction octant + Morton code of the origin (the binning a global ray queue would do), and shuffled (the worst
case). Each set: 1 warm + 5 timed launches, in this order — under `rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum` the dispatches of
trace_rays_kernel<false,false,false> come in groups of six per set (pmc_by_group below reads the CSV back)."""
import sys, os, glob, csv
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from sunray_amd import abi, scenes, runtime as rt

if len(sys.argv) > 2 and sys.argv[1] == "--pmc-csv":      # post-process: mean L2 hit rate per group of six dispatches
    rows = {}
    for f in glob.glob(sys.argv[2] + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "trace_rays_kernel<false, false, false>" in r["Kernel_Name"]:
                rows.setdefault(int(r["Dispatch_Id"]), {})[r["Counter_Name"]] = float(r["Counter_Value"])
    ids = sorted(rows)
    names = ["primary (fills the hit points)", "bounce, pixel order", "bounce, sorted (octant + origin Morton)", "bounce, shuffled"]
    ids = ids[1:]                                         # the first closest launch is the primary pass that produces the hit points
    for g, name in enumerate(names[1:]):
        grp = ids[g * 6 + 1:(g + 1) * 6]
        hit = sum(rows[i]["TCC_HIT_sum"] for i in grp); miss = sum(rows[i]["TCC_MISS_sum"] for i in grp)
        print("%-42s L2 hit rate %.3f  (requests per launch %.1f M)" % (name, hit / (hit + miss), (hit + miss) / len(grp) / 1e6))
    sys.exit(0)

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
o = vi[:3, 3].expand(len(d), 3)


def rays(o, d, tmin, tmax):
    r = torch.empty(len(o), 8, device=dev)
    r[:, 0:3] = o; r[:, 3] = tmin; r[:, 4:7] = d; r[:, 7] = tmax
    return r.contiguous()


N = W * H
hits = sc.trace_closest(rays(o, d, 0.001, 10000.0), N)
ok = hits[:, 0] > 0
p = o + d * hits[:, 0:1]
g = torch.Generator(device=dev); g.manual_seed(1)
r1 = torch.rand(N, generator=g, device=dev); r2 = torch.rand(N, generator=g, device=dev)
phi = 2 * np.pi * r1; rr = torch.sqrt(r2)
bd = torch.stack([rr * torch.cos(phi), torch.sqrt(1 - r2), rr * torch.sin(phi)], 1)
bo = (p + torch.tensor([0.0, 0.002, 0.0], device=dev))[ok]; bd = bd[ok]
n = len(bo)


def morton3(q):      # 10 bits per axis
    def spread(v):
        v = v & 0x3FF
        v = (v | (v << 16)) & 0x030000FF; v = (v | (v << 8)) & 0x0300F00F; v = (v | (v << 4)) & 0x030C30C3; v = (v | (v << 2)) & 0x09249249
        return v
    return spread(q[:, 0]) | (spread(q[:, 1]) << 1) | (spread(q[:, 2]) << 2)


lo, hi = bo.min(0).values, bo.max(0).values
q = ((bo - lo) / (hi - lo + 1e-9) * 1023).long()
octant = ((bd[:, 0] < 0).long() | ((bd[:, 1] < 0).long() << 1) | ((bd[:, 2] < 0).long() << 2))
key = (octant << 30) | morton3(q)
order_sorted = torch.argsort(key)
order_shuf = torch.randperm(n, generator=g, device=dev)
sets = [("bounce, pixel order", rays(bo, bd, 0.001, 10000.0)),
        ("bounce, sorted (octant + origin Morton)", rays(bo[order_sorted], bd[order_sorted], 0.001, 10000.0)),
        ("bounce, shuffled", rays(bo[order_shuf], bd[order_shuf], 0.001, 10000.0))]
for name, r in sets:
    sc.trace_closest(r, n); torch.cuda.synchronize()
    sc.enable_timing(True)
    for _ in range(5):
        sc.trace_closest(r, n)
    ms, k = sc.read_timing(2); sc.enable_timing(False)
    print("%-42s %.3f ms  %.0f Mray/s  (%d rays)" % (name, ms / k, n / (ms / k) / 1e3, n), flush=True)
