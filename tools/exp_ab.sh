#!/bin/bash
# same-box A/B of the default bench line (headline leg only): tools/exp_ab.sh OUTDIR "name|ENV=.. ENV=.." ...   (3 interleaved rounds)
cd $GRAFT_REPO_ROOT; OUT=$1; shift; mkdir -p $OUT
for rnd in 1 2 3; do
  for spec in "$@"; do
    name=${spec%%|*}; envs=${spec#*|}
    env $envs python bench.py --no-side-legs --no-cpu-baseline ${BENCH_ARGS:-} > $OUT/${name}_$rnd.json 2>/dev/null
    python -c "
import json;d=json.loads(open('$OUT/${name}_$rnd.json').read().strip().splitlines()[-1]);print('round $rnd $name',round(d['value'],2),'fps')"
  done
done
