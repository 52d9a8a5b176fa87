for s in 1 2 3 4; do PHD_SPLIT=$s timeout -k 10 100 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-extra --no-events 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('split $s', d['ms_per_step'])"; done
timeout -k 10 100 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --force-dist --no-extra 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('force-dist', d['ms_per_step'], d.get('rccl_ranks'))"
for c in S; do timeout -k 10 100 python bench.py --config $c --steps 10 --warmup 3 --no-cpu-baseline --no-extra 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('config $c', d['ms_per_step'], d['value'])"; done
