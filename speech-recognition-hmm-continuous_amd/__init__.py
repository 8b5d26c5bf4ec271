"""MI355X-native GMM-HMM core: HIP kernels + C ABI (csrc/, include/ghmm.h) and the
ctypes face used by bench.py and tests.  The directory name is not a Python
identifier; load it with `ghmm_amd = load()` from tests/_load.py / bench.py."""
from . import em, ghmm, launch  # noqa: F401
