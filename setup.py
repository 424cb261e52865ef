# SPDX-License-Identifier: Apache-2.0
"""Packaging for the MI355X vLLM platform plugin (entry point mirrors /root/reference/setup.py:41-43).

The sources live in `vllm-neuron_amd/` (the directory name the build contract fixes; a hyphen is not
importable), so the distribution maps the importable package `vllm_neuron_amd` onto that directory
with `package_dir` -- an installed copy does not depend on the checkout (the in-tree
`vllm_neuron_amd/` stub is only for running from the repository root).  The HIP library is built
for gfx950 by the `build_py` step (hipcc; `csrc/build.py`) and shipped as package data next to its
sources, where `_native.py` looks for it.
"""
import os
import subprocess
import sys

from setuptools import setup
from setuptools.command.build_py import build_py

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = "vllm-neuron_amd"


class BuildWithHip(build_py):
    """Compile libmi355x_vllm.so before the package files are collected."""

    def run(self):
        lib = os.path.join(HERE, SRC, "csrc", "libmi355x_vllm.so")
        if os.environ.get("MI355X_SKIP_HIP_BUILD") != "1":
            subprocess.check_call([sys.executable, os.path.join(HERE, SRC, "csrc", "build.py")])
        if not os.path.exists(lib):
            raise RuntimeError(f"{lib} is missing: the plugin has no CPU fallback (hipcc --offload-arch=gfx950 needed)")
        super().run()


setup(
    name="vllm-neuron-amd",
    version="0.2.0",
    description="vLLM MI355X (gfx950) backend plugin: hand-written HIP hot path behind the vllm-neuron plugin boundary",
    license="Apache 2.0",
    package_dir={"vllm_neuron_amd": SRC},
    packages=["vllm_neuron_amd", "vllm_neuron_amd.core", "vllm_neuron_amd.worker"],
    package_data={"vllm_neuron_amd": ["csrc/libmi355x_vllm.so", "csrc/*.h", "csrc/*.hip", "csrc/build.py"]},
    data_files=[("include", ["include/mi355x_vllm.h"])],
    cmdclass={"build_py": BuildWithHip},
    python_requires=">=3.10",
    entry_points={"vllm.platform_plugins": ["mi355x = vllm_neuron_amd:register"]},
)
