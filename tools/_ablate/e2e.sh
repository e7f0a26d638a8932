python -m pytest tests/test_gpu_tower.py -x -q -m gpu 2>&1 | tail -4
for p in 0 1; do echo "WGRAD_256=$p"; 
for w in e2e_softmax e2e_groupwise_gumbel; do TFR_WGRAD_256=$p python bench.py --workload $w --steps 30 --warmup 5 --no-cpu-baseline --also none 2>/dev/null | python tools/bench_brief.py; done; done
cd /tmp; export TMPDIR=/tmp; rocprofv3 --kernel-trace --stats -d /tmp/prof -o x -- python $GRAFT_REPO_ROOT/bench.py --workload e2e_softmax --steps 30 --warmup 5 --no-cpu-baseline --also none > /dev/null 2>&1; python - <<'PY'
import csv,glob
f=glob.glob('/tmp/prof/**/*kernel_stats.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
for r in rows[:16]: print('%-70s %6s %10.1f %8.2f'%(r['Name'][:70],r['Calls'],float(r['TotalDurationNs'])/1e3,float(r['AverageNs'])/1e3))
PY
