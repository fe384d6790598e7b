"""Deterministic procedural scenes for the BASELINE.json configurations.

No Stanford Bunny / Sponza asset exists offline, so configs 2-4 use procedural stand-ins of the
stated triangle counts. Everything is generated with numpy from fixed seeds (a PCG-hash value
noise, the same pcg_hash as shaders/rt_utils.slang:38-45), so the oracle and the HIP path are fed
byte-identical vertex/index/material arrays.
"""
from dataclasses import dataclass, field
from typing import List, Tuple

import numpy as np

from . import abi


@dataclass
class MeshDesc:
    key: int
    vertices: np.ndarray  # abi.VERTEX
    indices: np.ndarray   # uint32
    material: np.ndarray  # abi.MATERIAL scalar


@dataclass
class SceneDesc:
    name: str
    meshes: List[MeshDesc] = field(default_factory=list)
    instances: List[Tuple[int, List[np.ndarray]]] = field(default_factory=list)  # (key, [3x4 row-major])
    camera_pos: Tuple[float, float, float] = (0.0, 0.0, 1.0)   # Camera::default (camera.rs:10-18)
    camera_target: Tuple[float, float, float] = (0.0, 0.0, 0.0)
    fov_y: float = 45.0

    def n_triangles(self):
        per_key = {m.key: len(m.indices) // 3 for m in self.meshes}
        return sum(per_key[k] * len(ts) for k, ts in self.instances)


def make_vertices(pos, nrm):
    v = np.zeros(len(pos), dtype=abi.VERTEX)
    v["position"] = np.asarray(pos, dtype=np.float32)
    v["normal"] = np.asarray(nrm, dtype=np.float32)
    return v


def translate(x, y, z, s=1.0):
    return np.array([s, 0, 0, x, 0, s, 0, y, 0, 0, s, z], dtype=np.float32)


def rotate_y(angle, x=0.0, y=0.0, z=0.0, s=1.0):
    c, sn = np.float32(np.cos(angle)), np.float32(np.sin(angle))
    return np.array([s * c, 0, s * sn, x, 0, s, 0, y, -s * sn, 0, s * c, z], dtype=np.float32)


def quad(p0, p1, p2, p3, normal):
    """Two triangles (p0,p1,p2), (p0,p2,p3); geometric normal of both = cross(p1-p0, p2-p0)."""
    pos = np.array([p0, p1, p2, p3], dtype=np.float32)
    nrm = np.tile(np.asarray(normal, dtype=np.float32), (4, 1))
    return make_vertices(pos, nrm), np.array([0, 1, 2, 0, 2, 3], dtype=np.uint32)


def uv_sphere(radius, segments=32, rings=16):
    """UV sphere, `segments` x `rings`: 2*segments*(rings-1) triangles (32x16 -> 960)."""
    pos, idx = [(0.0, radius, 0.0)], []
    for r in range(1, rings):
        phi = np.pi * r / rings
        for s in range(segments):
            th = 2 * np.pi * s / segments
            pos.append((radius * np.sin(phi) * np.cos(th), radius * np.cos(phi), radius * np.sin(phi) * np.sin(th)))
    pos.append((0.0, -radius, 0.0))
    south = len(pos) - 1
    ring = lambda r, s: 1 + (r - 1) * segments + (s % segments)
    for s in range(segments):
        idx += [0, ring(1, s + 1), ring(1, s)]
        idx += [south, ring(rings - 1, s), ring(rings - 1, s + 1)]
    for r in range(1, rings - 1):
        for s in range(segments):
            a, b, c, d = ring(r, s), ring(r, s + 1), ring(r + 1, s + 1), ring(r + 1, s)
            idx += [a, b, c, a, c, d]
    pos = np.array(pos, dtype=np.float32)
    nrm = pos / np.linalg.norm(pos, axis=1, keepdims=True)
    return make_vertices(pos, nrm.astype(np.float32)), np.array(idx, dtype=np.uint32)


# ---- value noise on the reference's pcg_hash ----------------------------------------------------
def _pcg_hash(x):
    x = x.astype(np.uint32)
    x ^= x >> np.uint32(16); x *= np.uint32(0x7feb352d); x ^= x >> np.uint32(15)
    x *= np.uint32(0x846ca68b); x ^= x >> np.uint32(16)
    return x


def _lattice(ix, iy, seed):
    with np.errstate(over="ignore"):
        h = _pcg_hash(ix.astype(np.uint32) * np.uint32(0x9E3779B1) ^ _pcg_hash(iy.astype(np.uint32) + np.uint32(seed)))
    return h.astype(np.float64) / 4294967296.0


def fbm(x, y, seed, octaves=5):
    """Smooth value-noise fBm in [0,1), float64 in, float64 out."""
    out = np.zeros_like(x, dtype=np.float64)
    amp, freq, norm = 0.5, 1.0, 0.0
    for o in range(octaves):
        fx, fy = x * freq, y * freq
        ix, iy = np.floor(fx).astype(np.int64), np.floor(fy).astype(np.int64)
        tx, ty = fx - ix, fy - iy
        tx, ty = tx * tx * (3 - 2 * tx), ty * ty * (3 - 2 * ty)
        s = seed + 1013 * o
        a, b = _lattice(ix, iy, s), _lattice(ix + 1, iy, s)
        c, d = _lattice(ix, iy + 1, s), _lattice(ix + 1, iy + 1, s)
        out += amp * ((a * (1 - tx) + b * tx) * (1 - ty) + (c * (1 - tx) + d * tx) * ty)
        norm += amp
        amp *= 0.5
        freq *= 2.0
    return out / norm


# ---- config 1: Cornell box (6 quads + 960-triangle sphere) -----------------------------------
def cornell_box(sphere_material=None):
    s = SceneDesc("cornell_box", camera_pos=(0.0, 1.0, 3.4), camera_target=(0.0, 1.0, 0.0), fov_y=45.0)
    grey = abi.material(base_color=(0.8, 0.8, 0.8, 1.0), roughness=0.5)
    red = abi.material(base_color=(0.80, 0.0, 0.003, 1.0), roughness=0.5)
    blue = abi.material(base_color=(0.02, 0.0, 0.80, 1.0), roughness=0.5)
    light = abi.material(base_color=(0.8, 0.8, 0.8, 1.0), roughness=0.5, emissive_factor=(1.0, 1.0, 1.0), emissive_strength=10.0)
    walls = [
        (1, quad((-1, 0, 1), (1, 0, 1), (1, 0, -1), (-1, 0, -1), (0, 1, 0)), grey),    # floor
        (2, quad((-1, 2, -1), (1, 2, -1), (1, 2, 1), (-1, 2, 1), (0, -1, 0)), grey),   # ceiling
        (3, quad((-1, 0, -1), (1, 0, -1), (1, 2, -1), (-1, 2, -1), (0, 0, 1)), grey),  # back
        (4, quad((-1, 0, 1), (-1, 0, -1), (-1, 2, -1), (-1, 2, 1), (1, 0, 0)), red),   # left
        (5, quad((1, 0, -1), (1, 0, 1), (1, 2, 1), (1, 2, -1), (-1, 0, 0)), blue),     # right
        # light quad just under the ceiling, geometric normal (0,-1,0)
        (6, quad((-0.3, 1.99, -0.3), (0.3, 1.99, -0.3), (0.3, 1.99, 0.3), (-0.3, 1.99, 0.3), (0, -1, 0)), light),
    ]
    for key, (v, i), mat in walls:
        s.meshes.append(MeshDesc(key, v, i, mat))
        s.instances.append((key, [abi.IDENTITY_TRANSFORM.copy()]))
    sv, si = uv_sphere(0.4, 32, 16)
    if sphere_material is None:
        sphere_material = abi.material(base_color=(0.8, 0.8, 0.8, 1.0), metallic=0.0, roughness=0.5)
    s.meshes.append(MeshDesc(7, sv, si, sphere_material))
    s.instances.append((7, [translate(0.35, 0.4, 0.1)]))
    return s


def cornell_glass_mirror():
    """Cornell box with a glass sphere, a mirror sphere and a second (rotated, scaled) instance:
    exercises the transmission / perfect-mirror branches (ray_gen_ris.slang:95-117,
    ray_gen_final.slang:106-133) and non-identity WorldToObject normals (closest_hit.slang:48-50)."""
    s = cornell_box(sphere_material=abi.material(base_color=(0.9, 0.95, 1.0, 1.0), roughness=0.05, transmission=1.0, ior=1.5))
    s.name = "cornell_glass_mirror"
    sv, si = uv_sphere(0.3, 24, 12)
    mirror = abi.material(base_color=(0.95, 0.95, 0.95, 1.0), metallic=1.0, roughness=0.02)
    s.meshes.append(MeshDesc(8, sv, si, mirror))
    s.instances.append((8, [translate(-0.45, 0.3, -0.3), rotate_y(0.7, -0.1, 1.3, -0.5, 0.6)]))
    return s


# ---- config 2: 70 000-triangle torus knot ("Bunny" stand-in) ----------------------------------
def torus_knot(segments=350, sides=100, seed=1234):
    s = SceneDesc("torus_knot_70k", camera_pos=(0.0, 4.5, 9.0), camera_target=(0.0, 1.6, 0.0), fov_y=45.0)
    t = np.linspace(0.0, 2 * np.pi, segments, endpoint=False)
    p, q = 2, 3
    def curve(t):
        r = 1.6 + 0.7 * np.cos(q * t)
        return np.stack([r * np.cos(p * t), 0.7 * np.sin(q * t) + 1.8, r * np.sin(p * t)], axis=1)
    c = curve(t)
    dt = 1e-4
    tan = curve(t + dt) - curve(t - dt)
    tan /= np.linalg.norm(tan, axis=1, keepdims=True)
    up = np.array([0.0, 1.0, 0.0])
    n1 = np.cross(tan, up)
    n1 /= np.linalg.norm(n1, axis=1, keepdims=True)
    n2 = np.cross(tan, n1)
    a = np.linspace(0.0, 2 * np.pi, sides, endpoint=False)
    ca, sa = np.cos(a)[None, :, None], np.sin(a)[None, :, None]
    radial = n1[:, None, :] * ca + n2[:, None, :] * sa
    ii, jj = np.meshgrid(np.arange(segments), np.arange(sides), indexing="ij")
    disp = 0.42 + 0.16 * (fbm(ii / 9.0, jj / 9.0, seed, 4) - 0.5)
    pos = c[:, None, :] + radial * disp[:, :, None]
    pos = pos.reshape(-1, 3)
    nrm = radial.reshape(-1, 3)
    vid = lambda i, j: (i % segments) * sides + (j % sides)
    i0, j0 = ii.ravel(), jj.ravel()
    a_, b_, c_, d_ = vid(i0, j0), vid(i0 + 1, j0), vid(i0 + 1, j0 + 1), vid(i0, j0 + 1)
    idx = np.stack([a_, b_, c_, a_, c_, d_], axis=1).ravel().astype(np.uint32)
    knot = abi.material(base_color=(0.7, 0.7, 0.7, 1.0), roughness=0.6)
    s.meshes.append(MeshDesc(1, make_vertices(pos, nrm), idx, knot))
    s.instances.append((1, [abi.IDENTITY_TRANSFORM.copy()]))
    gv, gi = quad((-12, 0, 12), (12, 0, 12), (12, 0, -12), (-12, 0, -12), (0, 1, 0))
    s.meshes.append(MeshDesc(2, gv, gi, abi.material(base_color=(0.55, 0.55, 0.6, 1.0), roughness=0.8)))
    s.instances.append((2, [abi.IDENTITY_TRANSFORM.copy()]))
    lv, li = quad((-2, 7, -2), (2, 7, -2), (2, 7, 2), (-2, 7, 2), (0, -1, 0))
    s.meshes.append(MeshDesc(3, lv, li, abi.material(roughness=0.5, emissive_factor=(1.0, 0.95, 0.9), emissive_strength=12.0)))
    s.instances.append((3, [abi.IDENTITY_TRANSFORM.copy()]))
    return s


# ---- config 3/5: procedural heightfield, 2*(n-1)^2 triangles (n=708 -> 999 698) -----------------
def heightfield(n=708, seed=42, extent=15.0, amplitude=3.0, n_lights=8):
    s = SceneDesc("heightfield_%dk" % (2 * (n - 1) ** 2 // 1000), camera_pos=(0.0, 11.0, 21.0),
                  camera_target=(0.0, 0.5, 0.0), fov_y=45.0)
    g = np.arange(n)
    ii, jj = np.meshgrid(g, g, indexing="ij")
    x = (ii / (n - 1) * 2 - 1) * extent
    z = (jj / (n - 1) * 2 - 1) * extent
    hfun = lambda xx, zz: amplitude * fbm(xx * 0.25 + 40.0, zz * 0.25 + 40.0, seed, 6)
    y = hfun(x, z)
    e = extent * 2 / (n - 1)
    dx = (hfun(x + e, z) - hfun(x - e, z)) / (2 * e)
    dz = (hfun(x, z + e) - hfun(x, z - e)) / (2 * e)
    nrm = np.stack([-dx, np.ones_like(dx), -dz], axis=-1)
    nrm /= np.linalg.norm(nrm, axis=-1, keepdims=True)
    pos = np.stack([x, y, z], axis=-1).reshape(-1, 3)
    vid = lambda i, j: i * n + j
    i0, j0 = np.meshgrid(np.arange(n - 1), np.arange(n - 1), indexing="ij")
    i0, j0 = i0.ravel(), j0.ravel()
    a_, b_, c_, d_ = vid(i0, j0), vid(i0, j0 + 1), vid(i0 + 1, j0 + 1), vid(i0 + 1, j0)
    idx = np.stack([a_, b_, c_, a_, c_, d_], axis=1).ravel().astype(np.uint32)
    s.meshes.append(MeshDesc(1, make_vertices(pos, nrm.reshape(-1, 3)), idx,
                             abi.material(base_color=(0.62, 0.55, 0.42, 1.0), roughness=0.7)))
    s.instances.append((1, [abi.IDENTITY_TRANSFORM.copy()]))
    lv, li = quad((-1.2, 0, -1.2), (1.2, 0, -1.2), (1.2, 0, 1.2), (-1.2, 0, 1.2), (0, -1, 0))
    s.meshes.append(MeshDesc(2, lv, li, abi.material(roughness=0.5, emissive_factor=(1.0, 0.96, 0.9), emissive_strength=14.0)))
    xs = []
    for k in range(n_lights):
        ang = 2 * np.pi * k / max(n_lights, 1)
        xs.append(translate(7.5 * np.cos(ang), 7.0, 7.5 * np.sin(ang)))
    s.instances.append((2, xs))
    return s


def white_noise_rgba8(w=128, h=128, seed=7):
    """Stand-in for the reference's blue-noise texture (lib.rs:281-309): RGBA8, grey replicated to rgb,
    alpha 255. It is an INPUT of the path; the asset itself is not copied."""
    ii, jj = np.meshgrid(np.arange(h), np.arange(w), indexing="ij")
    with np.errstate(over="ignore"):
        v = (_pcg_hash((ii * w + jj).astype(np.uint32) ^ np.uint32(seed * 2654435761 & 0xFFFFFFFF)) >> np.uint32(24)).astype(np.uint8)
    tex = np.stack([v, v, v, np.full_like(v, 255)], axis=-1)
    return np.ascontiguousarray(tex)
