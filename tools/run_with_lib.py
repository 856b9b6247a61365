#!/usr/bin/env python3
"""Developer aid: run a script of this tree (bench.py, tools/traffic_split.py ...) against ANOTHER build of libvolviz_hip.so -- e.g. last
round's, built from `git archive <commit>` -- to A/B two rounds on one box.   python tools/run_with_lib.py <lib.so> bench.py --config c5 ...
Entry points the other build lacks are stubbed (calling one raises); everything else goes through the normal binding."""
import ctypes, os, runpy, sys

lib_path, script = os.path.abspath(sys.argv[1]), sys.argv[2]
sys.argv = sys.argv[2:]
REPO = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(REPO, "volume-viz_amd", "python")); sys.path.insert(0, REPO)
import torch  # noqa: F401  (before the first vv_init: INTEGRATION.md)
import volviz_amd as vv

_real = ctypes.CDLL


class _Tolerant(_real):
    def __getattr__(self, name):
        try:
            return super().__getattr__(name)
        except AttributeError:
            if not name.startswith("vv_"):
                raise
            def missing(*a, **k):
                raise RuntimeError(f"{name} does not exist in {lib_path}")
            f = ctypes.CFUNCTYPE(ctypes.c_int)(lambda: -1)
            setattr(self, name, f)
            return f


vv.C.CDLL = _Tolerant
vv.LIB_PATH = lib_path
runpy.run_path(os.path.join(REPO, script) if not os.path.isabs(script) else script, run_name="__main__")
