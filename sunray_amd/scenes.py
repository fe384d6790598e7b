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
    images: List[np.ndarray] = field(default_factory=list)     # (h, w, channels) uint8, slot order
    samplers: List[Tuple[int, int, int, int]] = field(default_factory=list)  # (min, mag, address u, address v), slot order
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


# ---- config 4: textured atrium ("Sponza-scale" stand-in) ------------------------------
def _tex_rgba(h, w, fn):
    jj, ii = np.meshgrid((np.arange(w) + 0.5) / w, (np.arange(h) + 0.5) / h)
    return np.ascontiguousarray(np.clip(np.rint(np.stack(fn(jj, ii), axis=-1) * 255.0), 0, 255).astype(np.uint8))


def grid_patch(p0, du, dv, nu, nv, normal, tangent, uv_scale=(1.0, 1.0)):
    """(nu x nv)-quad tessellated parallelogram p0 + a*du + b*dv with uv = (a, b) * uv_scale on every uv set,
    vertex normal `normal`, tangent (`tangent`, +1)."""
    a, b = np.meshgrid(np.linspace(0, 1, nu + 1), np.linspace(0, 1, nv + 1), indexing="ij")
    pos = np.asarray(p0, dtype=np.float64) + a[..., None] * np.asarray(du, dtype=np.float64) + b[..., None] * np.asarray(dv, dtype=np.float64)
    v = make_vertices(pos.reshape(-1, 3), np.tile(np.asarray(normal, dtype=np.float32), ((nu + 1) * (nv + 1), 1)))
    v["tangent"] = tuple(tangent) + (1.0,)
    uv = np.stack([a * uv_scale[0], b * uv_scale[1]], axis=-1).reshape(-1, 2).astype(np.float32)
    for k in ("base_color", "metallic_roughness", "normal", "occlusion", "emissive"):
        v[k + "_tex_coord"] = uv
    i0, j0 = np.meshgrid(np.arange(nu), np.arange(nv), indexing="ij")
    vid = lambda i, j: (i * (nv + 1) + j).ravel()
    a_, b_, c_, d_ = vid(i0, j0), vid(i0 + 1, j0), vid(i0 + 1, j0 + 1), vid(i0, j0 + 1)
    return v, np.stack([a_, b_, c_, a_, c_, d_], axis=1).ravel().astype(np.uint32)


def cylinder(radius, height, segments, rings, u_repeat=4.0, flute=0.0):
    """Open cylinder around +y with a seam (duplicated column of vertices so u runs 0..u_repeat), optional fluting."""
    th, yy = np.meshgrid(np.linspace(0, 2 * np.pi, segments + 1), np.linspace(0, 1, rings + 1), indexing="ij")
    r = radius * (1.0 + flute * np.cos(th * 12.0)) * (1.0 - 0.12 * yy)   # slight taper
    pos = np.stack([r * np.cos(th), yy * height, r * np.sin(th)], axis=-1).reshape(-1, 3)
    nrm = np.stack([np.cos(th), np.full_like(th, 0.12 * radius / height), np.sin(th)], axis=-1).reshape(-1, 3)
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    v = make_vertices(pos, nrm.astype(np.float32))
    tan = np.stack([-np.sin(th), np.zeros_like(th), np.cos(th), -np.ones_like(th)], axis=-1).reshape(-1, 4)   # handedness -1
    v["tangent"] = tan.astype(np.float32)
    uv = np.stack([th / (2 * np.pi) * u_repeat, yy * 3.0], axis=-1).reshape(-1, 2).astype(np.float32)
    for k in ("base_color", "metallic_roughness", "normal", "occlusion", "emissive"):
        v[k + "_tex_coord"] = uv
    v["normal_tex_coord"] = uv * np.float32(2.0)          # the normal map uses its own uv set (closest_hit.slang:37)
    i0, j0 = np.meshgrid(np.arange(segments), np.arange(rings), indexing="ij")
    vid = lambda i, j: (i * (rings + 1) + j).ravel()
    a_, b_, c_, d_ = vid(i0, j0), vid(i0, j0 + 1), vid(i0 + 1, j0 + 1), vid(i0 + 1, j0)
    return v, np.stack([a_, b_, c_, a_, c_, d_], axis=1).ravel().astype(np.uint32)


def scale_rotate_y(angle, sx, sy, sz, x, y, z):
    c, sn = np.float32(np.cos(angle)), np.float32(np.sin(angle))
    return np.array([sx * c, 0, sz * sn, x, 0, sy, 0, y, -sx * sn, 0, sz * c, z], dtype=np.float32)


def atrium(columns_per_side=20, col_segments=96, col_rings=30, floor_div=64, tex=512, n_lamps=32, seed=99):
    """Colonnade around a tiled floor: ~250k triangles in ~120 instances at the defaults, 512^2 procedural
    base-colour / metallic-roughness / normal / emissive textures, 2*n_lamps emissive triangles, 5 % mirrors.
    Every sampler path is used: LINEAR+REPEAT (floor, columns), LINEAR+MIRRORED_REPEAT (walls),
    NEAREST+CLAMP_TO_EDGE (lamps), and 1/3/4-channel image uploads."""
    s = SceneDesc("atrium", camera_pos=(0.0, 3.2, 17.0), camera_target=(0.0, 2.4, 0.0), fov_y=55.0)
    F = lambda x, y, sd, sc=6.0, o=5: fbm(x * sc, y * sc, sd, o)
    # images -------------------------------------------------------------------------------------
    def marble(u, v):
        n = F(u, v, seed) * 0.6 + 0.4 * np.abs(np.sin((u * 4 + F(u, v, seed + 1) * 2.5) * np.pi))
        check = ((np.floor(u * 4) + np.floor(v * 4)) % 2) * 0.25 + 0.6
        return (0.2 + 0.7 * n * check, 0.2 + 0.65 * n * check, 0.22 + 0.6 * n, np.ones_like(u))
    def sandstone(u, v):
        n = F(u, v, seed + 7, 10.0)
        return (0.55 + 0.35 * n, 0.42 + 0.3 * n, 0.25 + 0.25 * n)                         # RGB: alpha widened to 0
    def plaster(u, v):
        n = F(u, v, seed + 13, 3.0)
        stripe = 0.85 + 0.15 * (np.floor(v * 8) % 2)
        return (0.3 + 0.6 * n * stripe, 0.3 + 0.55 * n * stripe, 0.35 + 0.5 * n, np.ones_like(u))
    def met_rough(u, v):                                                                    # G = roughness, B = metallic
        n = F(u, v, seed + 21, 8.0)
        return (np.zeros_like(u), 0.3 + 0.7 * n, (F(u, v, seed + 22, 2.0) > 0.62) * 1.0)   # RGB
    def bump_normal(u, v):
        e = 1.0 / tex
        hgt = lambda a, b: F(a, b, seed + 31, 12.0, 4)
        dx = (hgt(u + e, v) - hgt(u - e, v)) / (2 * e) * 0.02
        dy = (hgt(u, v + e) - hgt(u, v - e)) / (2 * e) * 0.02
        n = np.stack([-dx, -dy, np.ones_like(dx)], axis=-1)
        n /= np.linalg.norm(n, axis=-1, keepdims=True)
        return (n[..., 0] * 0.5 + 0.5, n[..., 1] * 0.5 + 0.5, n[..., 2] * 0.5 + 0.5, np.ones_like(u))
    def lamp_glow(u, v):                                                                    # single channel, coarse: NEAREST shows texels
        return (np.clip(1.2 - 2.0 * np.hypot(u - 0.5, v - 0.5), 0.0, 1.0),)
    s.images = [_tex_rgba(tex, tex, marble), _tex_rgba(tex, tex, sandstone), _tex_rgba(tex, tex, plaster), _tex_rgba(tex, tex, met_rough),
                _tex_rgba(tex, tex, bump_normal), _tex_rgba(max(tex // 32, 4), max(tex // 32, 4), lamp_glow)[..., 0]]
    IMG_MARBLE, IMG_SAND, IMG_PLASTER, IMG_MR, IMG_NORMAL, IMG_GLOW = range(6)
    s.samplers = [(abi.FILTER_LINEAR, abi.FILTER_LINEAR, abi.ADDRESS_REPEAT, abi.ADDRESS_REPEAT),
                  (abi.FILTER_LINEAR, abi.FILTER_LINEAR, abi.ADDRESS_MIRRORED_REPEAT, abi.ADDRESS_MIRRORED_REPEAT),
                  (abi.FILTER_NEAREST, abi.FILTER_NEAREST, abi.ADDRESS_CLAMP_TO_EDGE, abi.ADDRESS_CLAMP_TO_EDGE),
                  (abi.FILTER_LINEAR, abi.FILTER_LINEAR, abi.ADDRESS_CLAMP_TO_EDGE, abi.ADDRESS_REPEAT)]
    SMP_REPEAT, SMP_MIRROR, SMP_NEAREST_CLAMP, SMP_MIXED = range(4)
    key = [0]
    def add(v, i, mat, xforms):
        key[0] += 1
        s.meshes.append(MeshDesc(key[0], v, i, mat))
        s.instances.append((key[0], xforms))
    L, Wd, Hh = 18.0, 9.0, 7.0          # half length (z), half width (x), height
    # floor: tiled marble with its own mr + normal maps
    add(*grid_patch((-Wd, 0, L), (2 * Wd, 0, 0), (0, 0, -2 * L), floor_div, floor_div, (0, 1, 0), (1, 0, 0), uv_scale=(6.0, 12.0)),
        abi.material(base_color=(1, 1, 1, 1), roughness=1.0, metallic=1.0,
                     textures={"base_color": (IMG_MARBLE, SMP_REPEAT), "metallic_roughness": (IMG_MR, SMP_REPEAT), "normal": (IMG_NORMAL, SMP_REPEAT)}),
        [abi.IDENTITY_TRANSFORM.copy()])
    # ceiling + 4 walls: plaster, mirrored repeat; back wall uses the mixed sampler
    wall = lambda smp: abi.material(base_color=(0.9, 0.9, 0.9, 1), roughness=0.85, textures={"base_color": (IMG_PLASTER, smp), "normal": (IMG_NORMAL, SMP_REPEAT)})
    wd = max(floor_div // 4, 2)
    add(*grid_patch((-Wd, Hh, -L), (2 * Wd, 0, 0), (0, 0, 2 * L), wd, wd, (0, -1, 0), (1, 0, 0), (3.0, 6.0)), wall(SMP_MIRROR), [abi.IDENTITY_TRANSFORM.copy()])
    add(*grid_patch((-Wd, 0, -L), (0, 0, 2 * L), (0, Hh, 0), wd, wd, (1, 0, 0), (0, 0, 1), (5.0, 1.5)), wall(SMP_MIRROR), [abi.IDENTITY_TRANSFORM.copy()])
    add(*grid_patch((Wd, 0, L), (0, 0, -2 * L), (0, Hh, 0), wd, wd, (-1, 0, 0), (0, 0, -1), (5.0, 1.5)), wall(SMP_MIRROR), [abi.IDENTITY_TRANSFORM.copy()])
    add(*grid_patch((-Wd, 0, -L), (2 * Wd, 0, 0), (0, Hh, 0), wd, wd, (0, 0, 1), (1, 0, 0), (2.5, 1.5)), wall(SMP_MIXED), [abi.IDENTITY_TRANSFORM.copy()])
    # columns: one fluted cylinder mesh, 2 rows of instances with per-instance rotation and non-uniform scale
    cv, ci = cylinder(0.55, 6.2, col_segments, col_rings, flute=0.04)
    col_mat = abi.material(base_color=(1, 1, 1, 1), roughness=0.9, textures={"base_color": (IMG_SAND, SMP_REPEAT), "normal": (IMG_NORMAL, SMP_REPEAT),
                                                                              "metallic_roughness": (IMG_MR, SMP_MIRROR)})
    xs = []
    for side in (-1, 1):
        for k in range(columns_per_side):
            z = -L + 1.5 + (2 * L - 3.0) * k / max(columns_per_side - 1, 1)
            xs.append(scale_rotate_y(0.37 * k + side, 1.0 + 0.05 * (k % 3), 1.0 + 0.02 * (k % 5), 0.9 + 0.04 * (k % 4), side * 5.5, 0.0, z))
    add(cv, ci, col_mat, xs)
    # column bases: untextured boxes-as-patches are skipped; plinth = short wide cylinder, separate mesh, untextured
    pv, pi = cylinder(0.8, 0.35, max(col_segments // 4, 8), 1)
    add(pv, pi, abi.material(base_color=(0.5, 0.48, 0.45, 1), roughness=0.6),
        [translate(side * 5.5, 0.0, -L + 1.5 + (2 * L - 3.0) * k / max(columns_per_side - 1, 1)) for side in (-1, 1) for k in range(columns_per_side)])
    # mirrors and a glass sphere along the axis (5 % of instances are mirrors)
    sv, si = uv_sphere(0.9, 32, 16)
    add(sv, si, abi.material(base_color=(0.95, 0.95, 0.95, 1), metallic=1.0, roughness=0.03),
        [translate(-2.2, 0.9, 6.0), translate(2.4, 0.9, 1.0), translate(-1.0, 0.9, -5.0), translate(1.6, 0.9, -10.0), scale_rotate_y(0.4, 1.4, 0.7, 1.0, 0.0, 0.63, 9.5)])
    add(sv, si, abi.material(base_color=(0.9, 0.97, 0.95, 1), roughness=0.05, transmission=1.0, ior=1.45), [translate(0.3, 0.9, 12.0)])
    # lamps: n_lamps emissive quads under the ceiling, each its own mesh (distinct emission), glow texture NEAREST/clamp
    for k in range(n_lamps):
        cx = (-1 if k % 2 else 1) * (2.0 + 1.5 * ((k // 2) % 2))
        cz = -L + 2.0 + (2 * L - 4.0) * (k // 2) / max(n_lamps // 2 - 1, 1)
        hue = np.array([1.0, 0.85 + 0.1 * np.sin(k), 0.6 + 0.3 * np.cos(1.7 * k)])
        lv, li = grid_patch((cx - 0.45, Hh - 0.4, cz - 0.45), (0.9, 0, 0), (0, 0, 0.9), 1, 1, (0, -1, 0), (1, 0, 0))
        textures = {"emissive": (IMG_GLOW, SMP_NEAREST_CLAMP)} if k % 4 != 3 else None   # every 4th lamp untextured
        add(lv, li, abi.material(base_color=(0.8, 0.8, 0.8, 1), roughness=0.5, emissive_factor=tuple(hue), emissive_strength=9.0 + (k % 5), textures=textures),
            [abi.IDENTITY_TRANSFORM.copy()])
    return s


# ---- instanced field: many instances of few meshes (two-level instancing) ---------------
def instanced_field(n_instances=300, segments=24, rings=12, n_meshes=3, seed=5, extent=12.0, nonuniform=True, n_lamps=4):
    """`n_instances` rotated, (non-uniformly) scaled and translated instances of `n_meshes` bumpy blobs (2 * segments * (rings - 1)
    triangles each: 24 x 12 -> 528) over a ground quad, lit by a few emissive quads. Transforms are general affine maps
    (rotation about an arbitrary axis, per-axis scales within a factor of three, a small shear), which is what separates an
    object-space walk from a world-space one."""
    rng = np.random.default_rng(seed)
    s = SceneDesc("instanced_field_%d" % n_instances, camera_pos=(0.0, 9.0, 1.6 * extent), camera_target=(0.0, 0.8, 0.0), fov_y=50.0)
    key = 1
    blob_keys = []
    for m in range(n_meshes):
        v, idx = uv_sphere(1.0, segments, rings)
        p = v["position"].astype(np.float64)
        bump = 1.0 + 0.25 * np.sin(3.0 * p[:, 0] + m) * np.cos(2.0 * p[:, 1] - m) + 0.15 * np.sin(5.0 * p[:, 2])
        v["position"] = (p * bump[:, None]).astype(np.float32)
        mat = abi.material(base_color=(0.3 + 0.2 * m, 0.7 - 0.15 * m, 0.4, 1.0), metallic=0.0 if m else 0.95, roughness=0.6 if m else 0.05)
        s.meshes.append(MeshDesc(key, v, idx, mat))
        blob_keys.append(key)
        key += 1
    gv, gi = quad((-2 * extent, 0, -2 * extent), (-2 * extent, 0, 2 * extent), (2 * extent, 0, 2 * extent), (2 * extent, 0, -2 * extent), (0, 1, 0))
    s.meshes.append(MeshDesc(key, gv, gi, abi.material(base_color=(0.7, 0.7, 0.7, 1.0), roughness=0.8)))
    ground_key = key
    key += 1
    lv, li = quad((-1, 0, -1), (1, 0, -1), (1, 0, 1), (-1, 0, 1), (0, -1, 0))
    s.meshes.append(MeshDesc(key, lv, li, abi.material(base_color=(1, 1, 1, 1), emissive_factor=(1.0, 0.95, 0.85), emissive_strength=18.0)))
    lamp_key = key
    per_key = {k: [] for k in blob_keys}
    for i in range(n_instances):
        axis = rng.normal(size=3); axis /= np.linalg.norm(axis)
        ang = rng.uniform(0, 2 * np.pi)
        K = np.array([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]])
        R = np.eye(3) + np.sin(ang) * K + (1 - np.cos(ang)) * (K @ K)
        sc = rng.uniform(0.25, 0.75, size=3) if nonuniform else np.full(3, rng.uniform(0.3, 0.7))
        S = np.diag(sc)
        if nonuniform:
            S[0, 1] = 0.15 * sc[0]
        M = R @ S
        t = np.array([rng.uniform(-extent, extent), rng.uniform(0.3, 3.5), rng.uniform(-extent, extent)])
        xf = np.concatenate([M, t[:, None]], axis=1).astype(np.float32).reshape(12)
        per_key[blob_keys[i % n_meshes]].append(xf)
    for k in blob_keys:
        if per_key[k]:
            s.instances.append((k, per_key[k]))
    s.instances.append((ground_key, [translate(0, 0, 0)]))
    s.instances.append((lamp_key, [translate(extent * (0.6 * np.cos(2.4 * j)), 7.0 + 0.5 * j, extent * (0.6 * np.sin(2.4 * j)), 1.5) for j in range(n_lamps)]))
    return s
