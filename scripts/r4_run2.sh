#!/bin/bash
# round-4 GPU session 2: the new tests, then bench lines (default, one-rank sharded rehearsals)
set -u
TAG=${1:-r4b}; O=gpurun_out/$TAG; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_round4.py tests/test_gpu_multiproc.py tests/test_full_size.py -m gpu -x -q --durations=8 > $O/tests.log 2>&1
echo "pytest rc=$?" >> $O/tests.log; tail -25 $O/tests.log
timeout -k 10 400 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
python - <<PY
import json
d = json.load(open("$O/bench_default.json"))
print("ms/step", d["ms_per_step"], "value", d["value"])
print("iso", d.get("kernel_ms_isolated"))
for k, v in d.get("other_modes", {}).items():
    print(k, {a: b for a, b in v.items() if a in ("ms_per_step", "ms_per_synchronous_update", "value_ms", "value_and_gradient_ms", "us_per_call", "error", "kernel_ms_isolated")})
PY
for w in steady survey; do
  timeout -k 10 200 python bench.py --force-dist --weights $w --steps 40 --warmup 3 --no-cpu-baseline --no-extra > $O/forcedist_$w.json 2> $O/forcedist_$w.err; echo "forcedist $w rc=$?"
  timeout -k 10 200 python bench.py --weights $w --steps 40 --warmup 3 --no-cpu-baseline --no-extra > $O/plain_$w.json 2> $O/plain_$w.err; echo "plain $w rc=$?"
done
timeout -k 10 200 python bench.py --force-dist --host-plan --weights steady --steps 40 --warmup 3 --no-cpu-baseline --no-extra > $O/forcedist_hostplan.json 2> $O/forcedist_hostplan.err; echo "hostplan rc=$?"
timeout -k 10 200 python bench.py --force-dist --collective allreduce --weights steady --steps 40 --warmup 3 --no-cpu-baseline --no-extra > $O/forcedist_allreduce.json 2> $O/forcedist_allreduce.err; echo "allreduce rc=$?"
python - <<PY
import json
for n in ("forcedist_steady", "plain_steady", "forcedist_survey", "plain_survey", "forcedist_hostplan", "forcedist_allreduce"):
    try:
        d = json.load(open("$O/%s.json" % n))
        print(n, "ms/step %.4f host_issue_us %.1f" % (d["ms_per_step"], d["host_issue_us_per_step"]), d.get("sharded_step", {}).get("phase_ms"), d.get("sharded_step", {}).get("rccl_probe_us"))
    except Exception as e:
        print(n, "failed", e)
PY
