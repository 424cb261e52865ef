# SPDX-License-Identifier: Apache-2.0
"""MI355X (gfx950) out-of-tree platform plugin for vLLM.

Entry point (setup.py): group ``vllm.platform_plugins``, ``mi355x = vllm_neuron_amd:register``
— the same mechanism the reference uses (/root/reference/setup.py:41-43,
/root/reference/vllm_neuron/__init__.py:14-23).
"""

import glob
import warnings

PLATFORM_QUALNAME = "vllm_neuron_amd.platform.MI355XPlatform"


def _is_mi355x_dev() -> bool:
    """An AMD GPU is visible to this process: the KFD node plus at least one render node."""
    return len(glob.glob("/dev/kfd")) > 0 and len(glob.glob("/dev/dri/renderD*")) > 0


def register():
    """Return the platform class qualname when a device is present, else None (+ UserWarning),
    exactly like the reference's ``register``."""
    if not _is_mi355x_dev():
        warnings.warn(
            "No AMD GPU devices found (/dev/kfd). "
            "Skipping MI355X plugin registration.",
            category=UserWarning,
        )
        return None
    return PLATFORM_QUALNAME
