"""Independent float64 checks of the oracle's lighting / sampling / reservoir helpers.

Parity is unpinned (the reference ships no vectors and cannot run here), and the oracle and the HIP helpers are
statement-by-statement twins of the Slang text, so a shared mis-reading would pass every GPU-vs-oracle test. These
tests come at the same functions from the other side: each is written here from its mathematical definition
(microfacet BRDF, Heitz' visible-normal sampling, weighted reservoir sampling) in vectorised float64 numpy, in a different
shape than the shader's statement order, and compared with the fp32 oracle on random inputs, next to the properties the
definition implies (unit half vectors in the visible hemisphere, reciprocity of the geometry term, exact selection
probabilities of a reservoir merge). Reference text: shaders/rt_utils.slang:150-274."""
import ctypes as C

import numpy as np
import pytest

from sunray_amd import abi

PI_REF = 3.14159        # the literal most helpers use (rt_utils.slang:172,223,230,260)
PI_VNDF = 3.14159265    # sample_ggx_vndf's (rt_utils.slang:192)


@pytest.fixture(scope="module")
def L(oracle):
    lib = oracle.lib()
    lib.orc_gi_target_pdf.restype = C.c_float
    return lib


def f(x):
    return C.c_float(float(x))


def v3(a):
    return (C.c_float * 3)(*[float(x) for x in a])


def unit(v):
    v = np.asarray(v, dtype=np.float64)
    return v / np.linalg.norm(v, axis=-1, keepdims=True)


def onb(n):
    """Duff et al. 2017 branchless orthonormal basis, the construction build_onb names (rt_utils.slang:150-156)."""
    s = np.where(n[..., 2] >= 0.0, 1.0, -1.0)
    a = -1.0 / (s + n[..., 2])
    b = n[..., 0] * n[..., 1] * a
    t = np.stack([1.0 + s * n[..., 0] ** 2 * a, s * b, -s * n[..., 0]], -1)
    bt = np.stack([b, s + n[..., 1] ** 2 * a, -n[..., 1]], -1)
    return t, bt


def brdf_light(P, N, V, albedo, rough, metal, Le, X, Nl):
    """Unshadowed radiance from a light sample X (normal Nl, emission Le) towards the viewer: GGX specular with the
    height-correlated Smith visibility + Lambert diffuse weighted by (1 - metallic)(1 - F), times cos cos / d^2."""
    d = X - P
    dist = np.maximum(np.linalg.norm(d, axis=-1), 1e-4)
    Ld = d / dist[:, None]
    cos_s = np.einsum("ij,ij->i", N, Ld)
    cos_l = -np.einsum("ij,ij->i", Nl, Ld)
    lit = (cos_s > 0) & (cos_l > 0)
    H = unit(V + Ld)
    nh = np.maximum(np.einsum("ij,ij->i", N, H), 0.0)
    vh = np.maximum(np.einsum("ij,ij->i", V, H), 0.0)
    nv = np.maximum(np.einsum("ij,ij->i", N, V), 1e-3)
    alpha = rough ** 2
    D = alpha ** 2 / (PI_REF * ((nh ** 2) * (alpha ** 2 - 1.0) + 1.0) ** 2)
    F0 = 0.04 + (albedo - 0.04) * metal[:, None]
    Fr = F0 + (1.0 - F0) * ((1.0 - vh) ** 5)[:, None]
    lam_v = cos_s * np.sqrt(nv ** 2 * (1 - alpha ** 2) + alpha ** 2)
    lam_l = nv * np.sqrt(cos_s ** 2 * (1 - alpha ** 2) + alpha ** 2)
    Vis = 0.5 / np.maximum(lam_v + lam_l, 1e-4)
    spec = (D * Vis)[:, None] * Fr
    diff = albedo * (1.0 - metal)[:, None] * (1.0 - Fr) / PI_REF
    G = cos_s * cos_l / np.maximum(dist ** 2, 1e-4)
    out = Le * (diff + spec) * G[:, None]
    out[~lit] = 0.0
    return out


def test_eval_unshadowed_light_against_float64_microfacet_model(L):
    rng = np.random.default_rng(11)
    n = 4000
    P = rng.uniform(-2, 2, (n, 3)); N = unit(rng.normal(size=(n, 3))); V = unit(N + 0.9 * unit(rng.normal(size=(n, 3))))
    X = P + unit(N + 0.8 * unit(rng.normal(size=(n, 3)))) * rng.uniform(0.2, 6.0, (n, 1))
    Nl = unit(P - X + 0.7 * rng.normal(size=(n, 3)))
    albedo = rng.uniform(0.05, 0.95, (n, 3)); rough = rng.uniform(0.05, 1.0, n); metal = rng.uniform(0, 1, n) * (rng.random(n) > 0.5)
    Le = rng.uniform(0.5, 12.0, (n, 3))
    want = brdf_light(P, N, V, albedo, rough, metal, Le, X, Nl)
    got = np.zeros((n, 3), np.float32)
    o = (C.c_float * 3)()
    for i in range(n):
        L.orc_eval_unshadowed_light(v3(P[i]), v3(N[i]), v3(V[i]), v3(albedo[i]), f(rough[i]), f(metal[i]), v3(Le[i]), v3(X[i]), v3(Nl[i]), o)
        got[i] = o[:]
    assert (want.max(axis=1) > 0).mean() > 0.3                         # plenty of lit configurations
    assert np.allclose(got, want, rtol=3e-4, atol=1e-6), np.abs(got - want).max()
    # properties: no light from behind either surface; scales linearly with emission; inverse-square in distance
    back = unit(-N)
    L.orc_eval_unshadowed_light(v3(P[0]), v3(N[0]), v3(V[0]), v3(albedo[0]), f(0.5), f(0.0), v3(Le[0]), v3(P[0] + back[0]), v3(N[0]), o)
    assert o[:] == [0.0, 0.0, 0.0]
    k = int(np.argmax(want.max(axis=1)))
    a = (C.c_float * 3)(); b = (C.c_float * 3)()
    L.orc_eval_unshadowed_light(v3(P[k]), v3(N[k]), v3(V[k]), v3(albedo[k]), f(rough[k]), f(metal[k]), v3(Le[k]), v3(X[k]), v3(Nl[k]), a)
    L.orc_eval_unshadowed_light(v3(P[k]), v3(N[k]), v3(V[k]), v3(albedo[k]), f(rough[k]), f(metal[k]), v3(2 * Le[k]), v3(X[k]), v3(Nl[k]), b)
    assert np.allclose(np.array(b[:]), 2 * np.array(a[:]), rtol=1e-6)
    L.orc_eval_unshadowed_light(v3(P[k]), v3(N[k]), v3(V[k]), v3(albedo[k]), f(rough[k]), f(metal[k]), v3(Le[k]), v3(P[k] + 2 * (X[k] - P[k])), v3(Nl[k]), b)
    assert np.allclose(np.array(b[:]), np.array(a[:]) / 4, rtol=1e-4)


def vndf_sample(N, V, rough, u1, u2):
    """Heitz 2018, "Sampling the GGX Distribution of Visible Normals", in the tangent frame of N: stretch the view
    vector, sample the projected disk (re-parameterised by the visible half), project onto the hemisphere, unstretch."""
    T, B = onb(N)
    Vl = np.stack([np.einsum("ij,ij->i", V, T), np.einsum("ij,ij->i", V, B), np.einsum("ij,ij->i", V, N)], -1)
    a = np.maximum(rough ** 2, 1e-3)[:, None]
    Vh = unit(Vl * np.concatenate([a, a, np.ones_like(a)], 1))
    lensq = Vh[:, 0] ** 2 + Vh[:, 1] ** 2
    T1 = np.where((lensq > 0)[:, None], np.stack([-Vh[:, 1], Vh[:, 0], np.zeros_like(lensq)], -1) / np.sqrt(np.where(lensq > 0, lensq, 1.0))[:, None],
                  np.array([1.0, 0.0, 0.0]))
    T2 = np.cross(Vh, T1)
    r = np.sqrt(u1); phi = 2.0 * PI_VNDF * u2
    t1 = r * np.cos(phi); t2 = r * np.sin(phi)
    s = 0.5 * (1.0 + Vh[:, 2])
    t2 = (1.0 - s) * np.sqrt(1.0 - t1 ** 2) + s * t2
    Nh = t1[:, None] * T1 + t2[:, None] * T2 + np.sqrt(np.maximum(0.0, 1.0 - t1 ** 2 - t2 ** 2))[:, None] * Vh
    Hl = unit(np.stack([a[:, 0] * Nh[:, 0], a[:, 0] * Nh[:, 1], np.maximum(0.0, Nh[:, 2])], -1))
    return T * Hl[:, 0:1] + B * Hl[:, 1:2] + N * Hl[:, 2:3]


def test_sample_ggx_vndf_against_float64_heitz(L):
    rng = np.random.default_rng(12)
    n = 4000
    N = unit(rng.normal(size=(n, 3))); V = unit(N + 0.95 * unit(rng.normal(size=(n, 3))))
    rough = rng.uniform(0.02, 1.0, n); u1 = rng.random(n); u2 = rng.random(n)
    want = vndf_sample(N, V, rough, u1, u2)
    got = np.zeros((n, 3), np.float32)
    o = (C.c_float * 3)()
    for i in range(n):
        L.orc_sample_ggx_vndf(v3(N[i]), v3(V[i]), f(rough[i]), f(u1[i]), f(u2[i]), o)
        got[i] = o[:]
    assert np.allclose(got, want, atol=2e-4), np.abs(got - want).max()
    # what visible-normal sampling guarantees: unit half vectors, in the upper hemisphere of N, facing the viewer
    assert np.allclose(np.linalg.norm(got, axis=1), 1.0, atol=1e-5)
    assert (np.einsum("ij,ij->i", got.astype(np.float64), N) >= -1e-6).all()
    assert (np.einsum("ij,ij->i", got.astype(np.float64), V) >= -1e-4).all()
    # and its distribution: for a smooth surface the half vectors concentrate around N, for a rough one they spread
    smooth = vndf_sample(N, V, np.full(n, 0.05), u1, u2); coarse = vndf_sample(N, V, np.full(n, 0.9), u1, u2)
    assert np.einsum("ij,ij->i", smooth, N).mean() > 0.999 > np.einsum("ij,ij->i", coarse, N).mean()


def test_gi_target_pdf_against_float64(L):
    rng = np.random.default_rng(13)
    n = 3000
    P = rng.uniform(-3, 3, (n, 3)); N = unit(rng.normal(size=(n, 3))); X = P + rng.normal(size=(n, 3)) * 2.0
    albedo = rng.uniform(0, 1, (n, 3)); metal = rng.uniform(0, 1, n); rad = rng.uniform(0, 5, (n, 3))
    w = X - P
    cosv = np.maximum(np.einsum("ij,ij->i", N, w / np.maximum(np.linalg.norm(w, axis=1), 1e-4)[:, None]), 0.0)
    want = (rad * albedo * ((1.0 - metal) / PI_REF * cosv)[:, None]).max(axis=1)
    got = np.array([L.orc_gi_target_pdf(v3(P[i]), v3(N[i]), v3(albedo[i]), f(metal[i]), v3(X[i]), v3(rad[i])) for i in range(n)])
    assert np.allclose(got, want, rtol=2e-5, atol=1e-7)
    assert (got[cosv == 0.0] == 0.0).all() and (cosv == 0.0).mean() > 0.3      # samples below the horizon have no target weight


def _res(dtype, **kw):
    r = np.zeros((), dtype=dtype)
    for k, v in kw.items():
        r[k] = v
    return r


def test_reservoir_merges_are_weighted_reservoir_sampling(L):
    """merge_reservoirs / merge_reservoirs_gi (rt_utils.slang:244-253,265-274): M accumulates, the weight p_hat * W * M
    (* jacobian) joins w_sum, and the incoming sample replaces the kept one iff rand < weight / w_sum — i.e. with exactly
    the probability weighted reservoir sampling prescribes. Checked value by value and by counting over a rand sweep."""
    rng = np.random.default_rng(14)
    p = lambda r: r.ctypes.data_as(C.c_void_p)
    for gi in (False, True):
        dt = abi.RESERVOIR_GI if gi else abi.RESERVOIR
        pos, other = ("sample_pos", "sample_radiance") if gi else ("light_pos", "light_normal")
        tag = "sample_normal_packed" if gi else "light_idx"
        for _ in range(200):
            w0, m0 = rng.uniform(0, 4), float(rng.integers(0, 12))
            new = _res(dt, **{pos: rng.normal(size=3), other: rng.normal(size=3), tag: 77, "W": rng.uniform(0, 3), "M": float(rng.integers(1, 10))})
            p_hat, jac = rng.uniform(0, 2), (rng.uniform(0, 10) if gi else 1.0)
            weight = np.float32(np.float32(np.float32(p_hat) * new["W"]) * new["M"]) * (np.float32(jac) if gi else np.float32(1))
            total = np.float32(w0) + weight
            thresh = float(weight) / max(float(total), 1e-4)
            taken = 0
            sweep = np.linspace(0.0, 1.0, 41)
            for u in sweep:
                r = _res(dt, **{pos: [1, 2, 3], other: [4, 5, 6], tag: 5, "w_sum": w0, "M": m0, "W": 0.25, "depth": 9.0, "hit_normal_packed": 123})
                if gi:
                    L.orc_merge_reservoirs_gi(p(r), p(new), f(p_hat), f(jac), f(u))
                else:
                    L.orc_merge_reservoirs(p(r), p(new), f(p_hat), f(u))
                assert r["M"] == np.float32(m0) + new["M"] and abs(float(r["w_sum"]) - float(total)) <= 1e-6 * max(1.0, float(total))
                assert r["W"] == np.float32(0.25) and r["depth"] == 9.0 and r["hit_normal_packed"] == 123   # untouched by a merge
                took = r[tag] == 77
                assert took == (np.float32(u) < np.float32(thresh)) or abs(u - thresh) < 1e-6
                if took:
                    assert np.array_equal(r[pos], new[pos]) and np.array_equal(r[other], new[other])
                else:
                    assert list(r[pos]) == [1, 2, 3] and list(r[other]) == [4, 5, 6] and r[tag] == 5
                taken += int(took)
            assert abs(taken / len(sweep) - min(thresh, 1.0)) <= 1.5 / len(sweep)      # selection probability = weight / w_sum
