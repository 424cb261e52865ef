# SPDX-License-Identifier: Apache-2.0
"""Packaging for the MI355X vLLM platform plugin (entry point mirrors /root/reference/setup.py:41-43).

The importable package is `vllm_neuron_amd` (a shim that points at the source directory
`vllm-neuron_amd/`); the HIP library is built in-tree by `python __graft_entry__.py`.
"""
from setuptools import setup

setup(
    name="vllm-neuron-amd",
    version="0.1.0",
    description="vLLM MI355X (gfx950) backend plugin: hand-written HIP hot path behind the vllm-neuron plugin boundary",
    license="Apache 2.0",
    packages=["vllm_neuron_amd"],
    python_requires=">=3.10",
    entry_points={"vllm.platform_plugins": ["mi355x = vllm_neuron_amd:register"]},
    include_package_data=True,
)
