cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pp; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pp -o p -- python $GRAFT_REPO_ROOT/tools/bench_dense.py > /tmp/pp.log 2>&1
python - <<'PY'
import csv
for r in csv.DictReader(open('/tmp/pp/p_kernel_stats.csv')):
    if any(k in r['Name'] for k in ('potrf_diag','apply_inv','tri_inv')):
        print(r['Name'][:70], r['Calls'], 'avg us', float(r['AverageNs'])/1e3, 'min', float(r['MinNs'])/1e3)
PY
grep gemm /tmp/pp.log | python -c "
import sys,json
for l in sys.stdin:
    r=json.loads(l); print(r['dtype'], r['n'], 'potrf %.2f ms | trsm %.2f ms' % (r['potrf_ms'], r['trsm_ms']))"
