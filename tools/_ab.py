import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ranking_amd._lib as l
if os.environ.get('OLD'):
    l.LIB_PATH = os.path.join(l.CSRC, 'libtfr_hip_old.so')
    l._stale = lambda: False
sys.argv = ['bench.py', '--no-cpu-baseline'] + sys.argv[1:]
import runpy
runpy.run_path(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'bench.py'), run_name='__main__')
