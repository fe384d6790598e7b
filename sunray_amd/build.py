"""Builds sunray_amd/libsunray_hip.so (HIP kernels for gfx950 + the C ABI) in-tree with hipcc.

hipcc cross-compiles gfx950 code objects without a GPU. Flags that matter for parity:
  -ffp-contract=off                         no implicit FMA fusion, host and device
  -fhip-fp32-correctly-rounded-divide-sqrt  IEEE divide/sqrt on the device
Code generation only (same IEEE operation per element, same bits):
  -fno-slp-vectorize                        no v_pk_*_f32: a packed fp32 instruction takes two issue slots on gfx950 and the
                                            register pairs it needs cost the final pass 10 spilled VGPRs and ~100 v_mov
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libsunray_hip.so")
SOURCES = ["kernels.hip", "post.hip", "bvh_gpu.hip", "api.cpp", "renderer.cpp", "multi_gpu.cpp", "gltf_load.cpp", "jpeg_decode.cpp", "host_prep.cpp", "bvh_build.cpp"]
HEADERS = ["rt_device.h", "traverse.h", "kernels.h", "host.h", "bvh_gpu.h", os.path.join("..", "..", "include", "sunray_hip.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
         "-fhip-fp32-correctly-rounded-divide-sqrt", "-fno-fast-math", "-fno-slp-vectorize", "-Wall", "-Wno-unused-function",
         "-pthread"]


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    if not force and not needs_build():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs = []
    os.makedirs(os.path.join(HERE, "_obj"), exist_ok=True)
    for src in SOURCES:
        obj = os.path.join(HERE, "_obj", src + ".o")
        cmd = [hipcc] + FLAGS + (["-x", "hip"] if src.endswith(".cpp") else []) + ["-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
        objs.append(obj)
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs + ["-pthread", "-lz", "-ldl"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose=True)
    print(LIB)
