"""bench.py against another build of the library (experiments): python tools/bench_lib.py <lib.so> [bench.py arguments]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dagnet_amd import _capi

_capi._LIB_PATH = os.path.abspath(sys.argv[1])
sys.argv = ["bench.py"] + sys.argv[2:]
import bench

bench.main()
