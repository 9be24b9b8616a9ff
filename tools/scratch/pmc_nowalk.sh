cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
for f in 1073741827 1073741843; do
  rm -rf /tmp/pm; AIPT_TRACE_LANES=1 timeout 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d /tmp/pm -o p -- python tools/trace_bench.py --sizes 1280x720 --batch 10 --frames 4 --flags $f > /dev/null 2>&1
  python - <<PY
import csv,collections
rows=list(csv.DictReader(open('/tmp/pm/p_counter_collection.csv')))
acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
for r in rows:
    k=r['Kernel_Name'].split('(')[0].replace('void aipt::','')
    if 'trace_bounce' not in k: continue
    acc[k][r['Counter_Name']]+=float(r['Counter_Value'])
    if r['Counter_Name']=='SQ_WAVES': n[k]+=1
print('flags $f')
for k,v in acc.items(): print('  ',k, n[k],'launches', {c:round(x/n[k]/1e6,2) for c,x in v.items()}, 'M per launch')
PY
done
