"""Import shim: the package directory is `clip-retrieval_amd/` (not a valid Python identifier), so
`import clip_retrieval_amd` resolves here and is re-pointed at that directory."""
import os as _os

__path__ = [_os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "clip-retrieval_amd")]
__package__ = "clip_retrieval_amd"
if __spec__ is not None:
    __spec__.submodule_search_locations = __path__
__file__ = _os.path.join(__path__[0], "__init__.py")
with open(__file__, "r", encoding="utf-8") as _f:
    exec(compile(_f.read(), __file__, "exec"))
