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
HEADERS = ["rt_device.h", "traverse.h", "bvh_layout.h", "kernels.h", "host.h", "bvh_gpu.h", os.path.join("..", "..", "include", "sunray_hip.h")]
RESOURCES = os.path.join(HERE, "_obj", "kernels.hip.resources.txt")     # the compiler's per-kernel register / scratch report
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
        if src == "kernels.hip":        # keep the register / spill figures of the pass kernels next to the object (kernel_resources())
            r = subprocess.run(cmd + ["-Rpass-analysis=kernel-resource-usage"], stderr=subprocess.PIPE, text=True)
            remarks = [ln for ln in r.stderr.splitlines() if "-Rpass-analysis=kernel-resource-usage" in ln]
            sys.stderr.write("\n".join(ln for ln in r.stderr.splitlines() if "-Rpass-analysis=kernel-resource-usage" not in ln))
            if r.returncode != 0:
                raise subprocess.CalledProcessError(r.returncode, cmd)
            with open(RESOURCES, "w") as f:
                f.write("\n".join(remarks) + "\n")
        else:
            subprocess.check_call(cmd)
        objs.append(obj)
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs + ["-pthread", "-lz", "-ldl"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB


def kernel_resources():
    """{kernel name: {"vgprs", "sgprs", "scratch", "occupancy", "sgpr_spills", "vgpr_spills"}} of kernels.hip as the compiler
    reported them at the last build (builds if there is no report yet)."""
    import re
    if not os.path.exists(RESOURCES) or needs_build():
        build(force=True)
    keys = {"TotalSGPRs": "sgprs", "VGPRs": "vgprs", "ScratchSize [bytes/lane]": "scratch", "Occupancy [waves/SIMD]": "occupancy",
            "SGPRs Spill": "sgpr_spills", "VGPRs Spill": "vgpr_spills"}
    out, cur = {}, None
    for ln in open(RESOURCES):
        m = re.search(r"remark:\s+(.*?):\s+(\S+) \[-Rpass", ln)
        if not m:
            continue
        k, v = m.group(1).strip(), m.group(2)
        if k == "Function Name":
            cur = out.setdefault(v, {})
        elif k in keys and cur is not None:
            cur[keys[k]] = int(v)
    return out


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose=True)
    print(LIB)
