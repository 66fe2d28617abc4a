"""Import shim: the package directory is named `godot-audio-spatializer_amd/` (not a valid Python
identifier), so `import godot_audio_spatializer_amd` resolves here and re-exports that directory
as a regular package."""
import os as _os

__path__ = [_os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "godot-audio-spatializer_amd")]
__file__ = _os.path.join(__path__[0], "__init__.py")
__package__ = __name__
with open(__file__) as _f:
    exec(compile(_f.read(), __file__, "exec"))
