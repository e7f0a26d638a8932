import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ranking_amd._ops as o
if os.environ.get('NOBAL'):
    o._BALANCE_MIN_LISTS = 1 << 30
sys.argv = ['bench.py', '--no-cpu-baseline'] + sys.argv[1:]
import runpy
runpy.run_path(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'bench.py'), run_name='__main__')
