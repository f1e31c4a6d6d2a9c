"""Import shim: the package directory is named `i3rc-monte-carlo-model_amd` (not a Python identifier);
`import i3rc_monte_carlo_model_amd` loads it from there."""
import os as _os

__path__ = [_os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "i3rc-monte-carlo-model_amd")]
__package__ = __name__
__file__ = _os.path.join(__path__[0], "__init__.py")
with open(__file__) as _f:
    exec(compile(_f.read(), __file__, "exec"))
