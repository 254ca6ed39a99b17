import sys, os
sys.path.insert(0, "/root/repo")
from covo_mpc_amd import _lib
sq, it = int(sys.argv[1]), int(sys.argv[2])
lib = _lib.load_library()
_lib.check(lib.covo_debug_set_ns_tail(sq, it))
sys.argv = ["bench.py", "--no-cpu-baseline"] + sys.argv[3:]
import bench
bench.main()
