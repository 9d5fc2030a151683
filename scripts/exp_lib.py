"""Run scripts/quick_solve.py against an alternative build of the engine library."""
import os, sys, runpy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from correlation_amd import _ffi
_ffi.LIB_PATH = os.path.join(os.path.dirname(_ffi.LIB_PATH), sys.argv[1])
orig = _ffi.load_library
_ffi.load_library = lambda path=None: orig(_ffi.LIB_PATH)
import correlation_amd
correlation_amd.engine._ffi.load_library = _ffi.load_library
sys.argv = ["quick_solve.py"] + sys.argv[2:]
runpy.run_path(os.path.join(os.path.dirname(os.path.abspath(__file__)), "quick_solve.py"), run_name="__main__")
