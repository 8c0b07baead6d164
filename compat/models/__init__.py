"""Drop-in `models` package: put this directory's parent (`compat/`) AHEAD of the reference
checkout on sys.path.  `models.Matcher` then comes from here (the GPU matcher behind the
reference's class, models/Matcher.py:13-275); every other submodule -- Constants, TrackDatabase,
BundleAdjustment, LoopClosure, TrajectorySolver -- is still found in the reference's own `models`
directory further down the path and runs unchanged."""
from pkgutil import extend_path

__path__ = extend_path(__path__, __name__)
