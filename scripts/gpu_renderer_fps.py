"""Frame time of the Renderer facade on the bench scene at 1080p: waiting for every frame vs keeping two frames in flight."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from sunray_amd import scenes, runtime as rt
desc = scenes.heightfield(708)
r = rt.Renderer((1920, 1080))
for m in desc.meshes:
    r.load_mesh(m.key, m.vertices, m.indices, m.material)
cam = (desc.camera_pos, desc.camera_target, desc.fov_y)
for _ in range(6):
    r.wait_frame(r.render(cam, desc.instances))
t0 = time.perf_counter()
for _ in range(40):
    r.wait_frame(r.render(cam, desc.instances))
a = (time.perf_counter() - t0) / 40 * 1e3
t0 = time.perf_counter()
prev = None
for _ in range(40):
    f = r.render(cam, desc.instances)
    if prev is not None:
        r.wait_frame(prev)
    prev = f
r.wait_frame(prev)
b = (time.perf_counter() - t0) / 40 * 1e3
print("Renderer 1080p, 1M triangles (ris + final + temporal + 4x denoise + tonemap): %.3f ms/frame waiting for each frame, %.3f ms/frame with two frames in flight" % (a, b))
