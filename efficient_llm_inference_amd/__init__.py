"""Import alias for the product package.

The product lives in ``efficient-llm-inference_amd/`` (the layout this repo is required to
use), a name Python cannot import. This stub makes ``import efficient_llm_inference_amd``
resolve to that directory: it points ``__path__`` there and executes the real ``__init__``.
"""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "efficient-llm-inference_amd")
__path__ = [_real]
__file__ = _os.path.join(_real, "__init__.py")
with open(__file__) as _f:
    exec(compile(_f.read(), __file__, "exec"))
del _f
