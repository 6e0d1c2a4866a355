"""Importable alias of the package directory `video-coding_amd/` (a hyphen is
not a valid Python identifier).  `import video_coding_amd` loads the code that
lives in ../video-coding_amd/."""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "video-coding_amd")
__path__ = [_real]
with open(_os.path.join(_real, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_real, "__init__.py"), "exec"))
