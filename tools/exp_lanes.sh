#!/bin/bash
cd $GRAFT_REPO_ROOT; OUT=gpurun_out/r3m; mkdir -p $OUT
for rnd in 1 2; do
for L in 3 4 5 6; do for Q in 4 8; do
  GPU_MAX_HW_QUEUES=$Q python bench.py --lanes $L --no-side-legs --no-cpu-baseline > $OUT/l${L}_q${Q}_$rnd.json 2>/dev/null
  python -c "
import json;d=json.loads(open('$OUT/l${L}_q${Q}_$rnd.json').read().strip().splitlines()[-1]);print('round $rnd lanes $L hwq $Q',round(d['value'],2),'fps', round(d['config']['hbm_peak_gb'],1),'GB')"
done; done; done
