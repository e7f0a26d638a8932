"""Same-box A/B helper: runs bench.py with a knob flipped by environment variable.
   NOBAL=1  disables the automatic longest-first launch order (ranking_amd._ops._BALANCE_MIN_LISTS)
   usage (through gpurun):  NOBAL=1 python tools/ab_bench.py --workload approx_ndcg ; python tools/ab_bench.py ...
"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ranking_amd._ops as o
if os.environ.get('NOBAL'):
    o._BALANCE_MIN_LISTS = 1 << 30
sys.argv = ['bench.py', '--no-cpu-baseline'] + sys.argv[1:]
import runpy
runpy.run_path(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'bench.py'), run_name='__main__')
