#!/bin/bash
# round-4 closing run, part 1: the whole GPU suite, the default bench line, the one-rank rehearsals of the sharded step, the
# multi-device handle's rehearsal on one GPU
set -u
O=gpurun_out/${1:-r4final}; mkdir -p $O
timeout -k 10 1100 python -m pytest tests -m gpu -q --durations=10 > $O/tests.log 2>&1; echo "pytest rc=$?" | tee -a $O/tests.log; tail -16 $O/tests.log
timeout -k 10 400 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
for w in steady survey; do
  timeout -k 10 200 python bench.py --force-dist --weights $w --steps 40 --warmup 3 --no-cpu-baseline --no-extra > $O/forcedist_$w.json 2> $O/forcedist_$w.err; echo "forcedist $w rc=$?"
  timeout -k 10 200 python bench.py --weights $w --steps 40 --warmup 3 --no-cpu-baseline --no-extra > $O/plain_$w.json 2> $O/plain_$w.err; echo "plain $w rc=$?"
done
timeout -k 10 200 python bench.py --force-dist --host-plan --weights steady --steps 40 --warmup 3 --no-cpu-baseline --no-extra > $O/forcedist_hostplan.json 2> $O/forcedist_hostplan.err; echo "hostplan rc=$?"
timeout -k 10 200 python bench.py --force-dist --collective allreduce --weights steady --steps 40 --warmup 3 --no-cpu-baseline --no-extra > $O/forcedist_allreduce.json 2> $O/forcedist_allreduce.err; echo "allreduce rc=$?"
for cfg in A B512; do
  timeout -k 10 300 python bench.py --single-process --gpus 8 --devices 0,0,0,0,0,0,0,0 --config $cfg --weights steady --steps 40 --warmup 3 > $O/multi8_$cfg.json 2> $O/multi8_$cfg.err; echo "multi8 $cfg rc=$?"
done
python - <<PY
import json
d = json.load(open("$O/bench_default.json"))
print("default ms/step", d["ms_per_step"], "value", d["value"], "roofline", d["roofline"]["frac"], d["roofline"]["kernel"], "cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["csharp_runtime"][:20])
print("iso", {k: round(v * 1e3, 1) for k, v in d.get("kernel_ms_isolated", {}).items()})
for k, v in d.get("other_modes", {}).items():
    print(k, {a: b for a, b in v.items() if a in ("ms_per_step", "ms_per_synchronous_update", "value_ms", "value_and_gradient_ms", "us_per_call", "error")})
for n in ("forcedist_steady", "plain_steady", "forcedist_survey", "plain_survey", "forcedist_hostplan", "forcedist_allreduce", "multi8_A", "multi8_B512"):
    try:
        d = json.load(open("$O/%s.json" % n))
        print(n, "ms/step %.4f" % d["ms_per_step"], "host_issue_us %.1f" % d.get("host_issue_us_per_step", -1), d.get("sharded_step", {}).get("phase_ms"), d.get("single_handle_same_particles", {}).get("ms_per_step"))
    except Exception as e:
        print(n, "failed", e)
PY
