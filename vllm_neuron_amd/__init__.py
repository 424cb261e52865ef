"""Import shim: the package sources live in ``vllm-neuron_amd/`` (the directory name the
build contract fixes); a hyphen is not importable, so this stub points the package path
there and runs its ``__init__``.  ``import vllm_neuron_amd`` is the supported spelling."""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "vllm-neuron_amd")
__path__ = [_real]
_init = _os.path.join(_real, "__init__.py")
with open(_init) as _f:
    exec(compile(_f.read(), _init, "exec"))
del _f, _init
