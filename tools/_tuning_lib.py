"""Imported first by the tools that need tuning hooks (in-kernel time stamps, MFMA probes, tile / schedule overrides through DINODET_*
variables): points the ctypes loader at the -DDINODET_TUNING build (lib/libdinodet_tuning.so, include/dinodet_tuning.h), building it when
it is missing.  The release library has none of those hooks."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
_LIB = os.path.join(ROOT, "dinov2_od_amd", "lib", "libdinodet_tuning.so")
if "DINODET_LIB" not in os.environ:
    from dinov2_od_amd._build import build
    build(verbose=False, tuning=True)
    os.environ["DINODET_LIB"] = _LIB
