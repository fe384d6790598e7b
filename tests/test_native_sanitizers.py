"""Host-only native code under AddressSanitizer + UndefinedBehaviorSanitizer (g++; GPU sanitizers are not available): the glTF /
PNG / JSON loader on mutated inputs, and the host BVH builder on random and degenerate triangle sets."""
import glob
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "sunray_amd", "csrc")
FLAGS = ["-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer", "-pthread"]
ENV = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")

pytestmark = pytest.mark.skipif(shutil.which("g++") is None, reason="g++ not available")


def test_gltf_loader_mutation_fuzz_under_sanitizers(tmp_path):
    exe = str(tmp_path / "fuzz_gltf")
    subprocess.check_call(["g++"] + FLAGS + ["-I", CSRC, os.path.join(ROOT, "tests", "native", "fuzz_gltf.cpp"), os.path.join(CSRC, "gltf_load.cpp"), "-lz", "-o", exe])
    seeds = [os.path.join(ROOT, "tests", "golden", "mini_scene.glb")] + sorted(glob.glob("/root/reference/examples/assets/Room*.glb"))[:1]
    for k, seed in enumerate(seeds):
        out = subprocess.run([exe, seed, "2500", str(17 + k), str(tmp_path / "m.glb")], capture_output=True, text=True, env=ENV)
        assert out.returncode == 0 and "runtime error" not in out.stderr and "AddressSanitizer" not in out.stderr, out.stderr[-3000:]
        assert " parsed" in out.stdout


def test_host_bvh_builder_under_sanitizers(tmp_path):
    exe = str(tmp_path / "bvh_asan")
    subprocess.check_call(["g++"] + FLAGS + ["-I", CSRC, os.path.join(ROOT, "tests", "native", "bvh_asan.cpp"), os.path.join(CSRC, "bvh_build.cpp"),
                                             os.path.join(CSRC, "host_prep.cpp"), "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True, env=ENV)
    assert out.returncode == 0 and "bvh ok" in out.stdout and "runtime error" not in out.stderr and "AddressSanitizer" not in out.stderr, out.stdout + out.stderr[-3000:]
