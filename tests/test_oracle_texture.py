"""Texture fetch + the textured half of closest_hit in the oracle (SURVEY.md §8a K4/K11 sample_texture,
rt_utils.slang:121-133; closest_hit.slang:34-46,56-72,82-87). The reference holds no texture vectors
(parity unpinned): the KATs below are hand-derived from the Vulkan texel-filtering equations, and an
independent float64 numpy restatement of those equations cross-checks the C code on random coordinates."""
import numpy as np
import pytest

from sunray_amd import abi, scenes

N, L = abi.FILTER_NEAREST, abi.FILTER_LINEAR
REP, MIR, CLAMP = abi.ADDRESS_REPEAT, abi.ADDRESS_MIRRORED_REPEAT, abi.ADDRESS_CLAMP_TO_EDGE


def np_sample(img, mag, mode_u, mode_v, s, t):
    """Vulkan spec texel filtering at LOD 0, float64: (u,v) = (s*w, t*h); NEAREST floor; LINEAR -0.5 shift."""
    h, w = img.shape[:2]

    def wrap(i, n, mode):
        if mode == REP:
            return i % n
        if mode == MIR:
            m = i % (2 * n)
            return m if m < n else 2 * n - 1 - m
        return min(max(i, 0), n - 1)
    u, v = s * w, t * h
    px = img.astype(np.float64) / 255.0
    if mag == N:
        return px[wrap(int(np.floor(v)), h, mode_v), wrap(int(np.floor(u)), w, mode_u)]
    u, v = u - 0.5, v - 0.5
    i0, j0 = int(np.floor(u)), int(np.floor(v))
    a, b = u - i0, v - j0
    g = lambda i, j: px[wrap(j, h, mode_v), wrap(i, w, mode_u)]
    return (g(i0, j0) * (1 - a) + g(i0 + 1, j0) * a) * (1 - b) + (g(i0, j0 + 1) * (1 - a) + g(i0 + 1, j0 + 1) * a) * b


@pytest.fixture()
def tex_scene(oracle):
    s = oracle.OracleScene()
    img = np.array([[[0, 10, 20, 30], [255, 110, 120, 130]],
                    [[40, 50, 60, 70], [80, 90, 100, 200]]], dtype=np.uint8)   # 2x2, row 0 = t in [0, .5)
    s.add_image(img)
    for mag in (N, L):
        for mu in (REP, MIR, CLAMP):
            s.add_sampler(mag, mag, mu, mu)
    return s, img


def smp(mag, mode):
    return (0 if mag == N else 3) + mode


def test_sample_texture_hand_kats(tex_scene):
    s, img = tex_scene
    f = lambda *px: np.array(px, dtype=np.float32) / np.float32(255.0)
    # NEAREST picks the texel whose cell holds (s*w, t*h)
    assert (s.sample_texture(0, smp(N, REP), 0.25, 0.25) == f(0, 10, 20, 30)).all()
    assert (s.sample_texture(0, smp(N, REP), 0.75, 0.25) == f(255, 110, 120, 130)).all()
    assert (s.sample_texture(0, smp(N, REP), 0.25, 0.75) == f(40, 50, 60, 70)).all()
    assert (s.sample_texture(0, smp(N, REP), 1.25, -0.25) == f(40, 50, 60, 70)).all()          # repeat: (0.25, 0.75)
    assert (s.sample_texture(0, smp(N, MIR), 1.25, 0.25) == f(255, 110, 120, 130)).all()       # mirror: 1.25 -> 0.75
    assert (s.sample_texture(0, smp(N, MIR), -0.25, 0.25) == f(0, 10, 20, 30)).all()           # mirror: -0.25 -> 0.25
    assert (s.sample_texture(0, smp(N, CLAMP), 7.0, -3.0) == f(255, 110, 120, 130)).all()      # clamp: last column, first row
    # LINEAR at a texel centre returns that texel; at the image centre the mean of all four
    assert (s.sample_texture(0, smp(L, CLAMP), 0.25, 0.25) == f(0, 10, 20, 30)).all()
    mean = (f(0, 10, 20, 30) * np.float32(0.5) + f(255, 110, 120, 130) * np.float32(0.5)) * np.float32(0.5) + \
           (f(40, 50, 60, 70) * np.float32(0.5) + f(80, 90, 100, 200) * np.float32(0.5)) * np.float32(0.5)
    assert (s.sample_texture(0, smp(L, REP), 0.5, 0.5) == mean).all()
    # LINEAR on the edge: clamp repeats the edge texel, repeat blends with the opposite edge
    assert (s.sample_texture(0, smp(L, CLAMP), 0.0, 0.25) == f(0, 10, 20, 30)).all()
    edge = f(255, 110, 120, 130) * np.float32(0.5) + f(0, 10, 20, 30) * np.float32(0.5)       # i0 = -1 -> 1, i1 = 0, a = .5
    assert (s.sample_texture(0, smp(L, REP), 0.0, 0.25) == edge).all()
    assert (s.sample_texture(0, smp(L, MIR), 0.0, 0.25) == f(0, 10, 20, 30)).all()             # mirror: i0 = -1 -> 0
    # NULL_TEXTURE returns the fallback unchanged (rt_utils.slang:127-129); non-finite coordinates read as 0
    assert (s.sample_texture(abi.NULL_TEXTURE, 0, 0.3, 0.3, (1, 2, 3, 4)) == [1, 2, 3, 4]).all()
    assert (s.sample_texture(0, smp(N, REP), float("nan"), float("inf")) == f(0, 10, 20, 30)).all()


@pytest.mark.parametrize("mag", [N, L])
@pytest.mark.parametrize("mode", [REP, MIR, CLAMP])
def test_sample_texture_matches_float64_equations(oracle, mag, mode):
    rng = np.random.default_rng(5 + mag * 3 + mode)
    img = rng.integers(0, 256, size=(7, 5, 4), dtype=np.uint8)      # non-square, odd sizes
    s = oracle.OracleScene()
    s.add_image(img)
    s.add_sampler(mag, mag, mode, (mode + 1) % 3)                    # different modes per axis
    for u, v in rng.uniform(-3.0, 4.0, size=(400, 2)):
        u, v = float(np.float32(u)), float(np.float32(v))
        got = s.sample_texture(0, 0, u, v)
        want = np_sample(img, mag, mode, (mode + 1) % 3, u, v)
        if mag == N and (abs(u * 5 - round(u * 5)) < 1e-4 or abs(v * 7 - round(v * 7)) < 1e-4):
            continue   # on a texel boundary the fp32 wrap may pick the neighbour
        assert np.abs(got - want).max() < 3e-5, (u, v, got, want)


def test_images_are_widened_with_zero_channels(oracle):
    s = oracle.OracleScene()
    s.add_image(np.array([[200]], dtype=np.uint8))                       # R8 -> (R,0,0,0)   utils.rs:27-43
    s.add_image(np.array([[[10, 20, 30]]], dtype=np.uint8))              # RGB8 -> alpha 0
    s.add_sampler(N, N, REP, REP)
    assert (s.sample_texture(0, 0, 0.5, 0.5) == np.array([200, 0, 0, 0], dtype=np.float32) / np.float32(255)).all()
    assert (s.sample_texture(1, 0, 0.5, 0.5) == np.array([10, 20, 30, 0], dtype=np.float32) / np.float32(255)).all()


def test_textured_closest_hit_payload(oracle):
    """One textured quad facing +z: the payload's albedo / emission / roughness / metallic come from the
    textures at the hit's uv, and a flat normal map leaves the normal (almost) geometric."""
    s = oracle.OracleScene()
    rng = np.random.default_rng(3)
    base = rng.integers(0, 256, size=(8, 8, 4), dtype=np.uint8)
    mr = rng.integers(0, 256, size=(8, 8, 3), dtype=np.uint8)
    flat = np.tile(np.array([128, 128, 255, 255], dtype=np.uint8), (4, 4, 1))
    tilt = np.tile(np.array([255, 128, 128, 255], dtype=np.uint8), (4, 4, 1))   # tangent-space (+1, 0, ~0)
    for im in (base, mr, flat, tilt):
        s.add_image(im)
    s.add_sampler(N, N, CLAMP, CLAMP)
    v, i = scenes.grid_patch((-1, -1, 0), (2, 0, 0), (0, 2, 0), 1, 1, (0, 0, 1), (1, 0, 0))
    mat = abi.material(base_color=(1, 1, 1, 1), roughness=0.8, metallic=0.5, emissive_factor=(1, 1, 1), emissive_strength=2.0,
                       textures={"base_color": (0, 0), "metallic_roughness": (1, 0), "normal": (2, 0), "emissive": (0, 0)})
    s.add_mesh(1, v, i, mat)
    v2 = v.copy(); v2["position"][:, 2] -= 5.0
    s.add_mesh(2, v2, i, abi.material(textures={"normal": (3, 0)}))
    s.set_instances([(1, [abi.IDENTITY_TRANSFORM]), (2, [abi.IDENTITY_TRANSFORM])])
    rays = np.zeros(2, dtype=abi.RAY)
    rays["origin"] = [(0.3, -0.4, 2.0), (0.3, -0.4, -2.0)]
    rays["dir"] = (0, 0, -1); rays["tmin"] = 0.001; rays["tmax"] = 100.0
    hits = s.trace_closest(rays)
    pl = s.shade_closest_hit(hits)
    u, vv = (0.3 + 1) / 2, (-0.4 + 1) / 2            # uv of the hit on the patch
    tx, ty = int(u * 8), int(vv * 8)
    want_rgb = base[ty, tx, :3]
    assert pl["albedo_packed"][0] == int(want_rgb[0]) | int(want_rgb[1]) << 8 | int(want_rgb[2]) << 16 | 0xFF << 24
    assert np.allclose(pl["emission"][0], want_rgb.astype(np.float32) / 255 * 2.0, atol=1e-6)
    rough, metal = np.frombuffer(np.uint32(pl["material_info"][0]).tobytes(), dtype=np.float16)
    assert abs(rough - 0.8 * mr[ty, tx, 1] / 255) < 2e-3 and abs(metal - 0.5 * mr[ty, tx, 2] / 255) < 2e-3
    n0 = oracle.unpack_normal(int(pl["normal_packed"][0]))
    assert np.allclose(n0, [0, 0, 1], atol=0.02)
    # second quad: normal map says tangent-space +x -> world normal leans to the tangent (+x)
    n1 = oracle.unpack_normal(int(pl["normal_packed"][1]))
    assert n1[0] > 0.95
