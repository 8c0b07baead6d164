"""Drop-in `utils` package: with `compat/` ahead of the reference checkout on sys.path,
`utils.utils` comes from here (the GPU front-end behind the reference's function names,
utils/utils.py) while `utils.plotters` and `utils.plot` are still the reference's own files."""
from pkgutil import extend_path

__path__ = extend_path(__path__, __name__)
