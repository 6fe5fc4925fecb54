"""Import shim: the product package lives in the directory `yolo-puncture_amd/` (the layout the build
contract names); a hyphen is not importable, so `import yolo_puncture_amd` resolves to that directory."""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "yolo-puncture_amd")
__path__ = [_real]
__file__ = _os.path.join(_real, "__init__.py")
with open(__file__) as _f:
    exec(compile(_f.read(), __file__, "exec"))
