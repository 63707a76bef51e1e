#!/bin/bash
cd $GRAFT_REPO_ROOT; OUT=gpurun_out/r3v; mkdir -p $OUT
for rnd in 1 2; do
for L in 4 6 8; do
  python bench.py --lanes $L --no-side-legs --no-cpu-baseline > $OUT/l${L}_$rnd.json 2>/dev/null
  python -c "
import json;d=json.loads(open('$OUT/l${L}_$rnd.json').read().strip().splitlines()[-1]);print('round $rnd lanes $L',round(d['value'],2),'fps peak',round(d['config']['hbm_peak_gb'],1),'GB steady',round(d['config']['hbm_steady_gb'],1))"
done; done
