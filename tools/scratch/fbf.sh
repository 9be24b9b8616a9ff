cd "${GRAFT_REPO_ROOT:-.}"
for f in 3 1073741827 536870915; do
 python bench.py --batch 1 --no-cpu-baseline --no-roofline-events --steps 40 --warmup 10 --trace-flags $f 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('flags $f', d['value'], d['frame'])"
done
python bench.py --batch 1 --prefetch --no-cpu-baseline --no-roofline-events --steps 40 --warmup 10 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('prefetch', d['value'])"
