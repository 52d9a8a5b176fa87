#!/bin/bash
O=gpurun_out/${1:-r3e}; mkdir -p $O
for w in 8 1; do timeout -k 10 200 python scripts/nr_stamps.py $w 2>/dev/null | tail -1 | tee -a $O/nr_stamps.log; done
for q in 4 16; do for cm in 512 0; do
  GPU_MAX_HW_QUEUES=$q PHD_CHAIN_MAX=$cm timeout -k 10 300 python bench.py --single-process --gpus 8 --devices 0,0,0,0,0,0,0,0 --config A --weights steady --steps 40 --warmup 3 2>$O/m.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); m=d['multi_host']
print('A   hwq $q chain_max $cm: %.3f ms/step, single %.3f, ratio %.2f, issue %.0f us, post %.1f us' % (d['ms_per_step'], d['single_handle_same_particles']['ms_per_step'], d['single_handle_same_particles']['multi_over_single'], m['worker_issue_us_per_step'], m['post_us_per_step']), {k: round(v,3) for k,v in m['phase_ms_first_shard'].items()})" | tee -a $O/probe.log
  GPU_MAX_HW_QUEUES=$q PHD_CHAIN_MAX=$cm timeout -k 10 300 python bench.py --single-process --gpus 8 --devices 0,0,0,0,0,0,0,0 --config B512 --weights steady --steps 40 --warmup 3 2>$O/m.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); m=d['multi_host']
print('B512 hwq $q chain_max $cm: %.3f ms/step, single %.3f, ratio %.2f, issue %.0f us, post %.1f us' % (d['ms_per_step'], d['single_handle_same_particles']['ms_per_step'], d['single_handle_same_particles']['multi_over_single'], m['worker_issue_us_per_step'], m['post_us_per_step']), {k: round(v,3) for k,v in m['phase_ms_first_shard'].items()})" | tee -a $O/probe.log
done; done
