"""quick-mer2_amd — MI355X-native `quicKmer2 count` hot path.

The directory name is not an importable identifier; load it with
`__graft_entry__.load_package()` (importlib by path) or put this directory on
sys.path and `import pyqm2`.
"""
from . import pyqm2  # noqa: F401
from .pyqm2 import Counter, Stats, lib, tools  # noqa: F401
