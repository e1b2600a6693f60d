"""bench.py on the EXPERIMENTS library (libmv2_hip_exp.so: environment switches are read by the library): in-graph A/B of a switch on one box.
usage: MV2_PW_DMA=0 python tools/bench_exp.py [bench.py flags]"""
import os
import runpy
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import magvit2_pytorch_amd.ops_hip as o          # noqa: E402

o.LIB_PATH = o.LIB_PATH.replace('libmv2_hip.so', 'libmv2_hip_exp.so')
sys.argv = ['bench.py'] + sys.argv[1:]
runpy.run_path(os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'bench.py'), run_name='__main__')
