#!/usr/bin/env python3
"""Build libmi355x_vllm.so for gfx950 with hipcc (cross-compiles without a GPU).

    python vllm-neuron_amd/csrc/build.py [--force]

Objects go to csrc/build/, the library next to the sources (in-tree, git-ignored: it
travels to the GPU box with the snapshot).
"""
import concurrent.futures as cf
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SOURCES = ["linear_kernels.hip", "attn_kernels.hip", "misc_kernels.hip", "model.hip", "tp_group.hip"]
HEADERS = ["mi_common.h", "linear_kernels.h", "attn_kernels.h", "misc_kernels.h", "model_internal.h", "tp_group.h",
           os.path.join("..", "..", "include", "mi355x_vllm.h")]
LIB = os.path.join(HERE, "libmi355x_vllm.so")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function", "-Wno-unused-value",
         "-Wno-unused-result", "-I/opt/rocm/include"]
# extra flags for probe builds (e.g. MI_EXTRA_FLAGS="-DMI_TRACE"); part of the stamp below
FLAGS += os.environ.get("MI_EXTRA_FLAGS", "").split()
STAMP = os.path.join(HERE, "build", "flags.stamp")


def _mtime(p):
    return os.path.getmtime(p) if os.path.exists(p) else 0.0


def _flags_changed():
    """An object built with other flags is stale whatever its mtime says."""
    want = " ".join(FLAGS)
    try:
        with open(STAMP) as f:
            same = f.read() == want
    except OSError:
        same = False
    if not same:
        with open(STAMP, "w") as f:
            f.write(want)
    return not same


def _compile(src, force=False):
    obj = os.path.join(HERE, "build", src.replace(".hip", ".o"))
    deps = [os.path.join(HERE, src)] + [os.path.join(HERE, h) for h in HEADERS]
    if _mtime(obj) > max(_mtime(d) for d in deps) and not force:
        return obj, ""
    cmd = ["hipcc", *FLAGS, "-c", os.path.join(HERE, src), "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed for {src}:\n{r.stdout}\n{r.stderr}")
    return obj, r.stderr


def build(verbose=True):
    os.makedirs(os.path.join(HERE, "build"), exist_ok=True)
    force = "--force" in sys.argv or _flags_changed()
    with cf.ThreadPoolExecutor(max_workers=4) as ex:
        results = list(ex.map(lambda src: _compile(src, force), SOURCES))
    objs = [o for o, _ in results]
    for _, warn in results:
        if warn and verbose:
            sys.stderr.write(warn)
    if _mtime(LIB) < max(_mtime(o) for o in objs) or force:
        cmd = ["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", LIB,
               "-L/opt/rocm/lib", "-lrccl", "-Wl,-rpath,/opt/rocm/lib"]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    return LIB


if __name__ == "__main__":
    print(build())
