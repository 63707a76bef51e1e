#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r3h
run() { name=$1; shift; env "$@" python bench.py --lanes 1 --steps 12 --no-side-legs --no-cpu-baseline > gpurun_out/r3h/$name.json 2>/dev/null; python -c "
import json;d=json.loads(open('gpurun_out/r3h/$name.json').read().strip().splitlines()[-1]);print('$name',round(d['value'],2),'fps',round(d['ms_per_step'],2),'ms')"; }
run base A=1
run prio0 DSVC_MV_STREAM_PRIO=0
run grid248 DSVC_PP_GRID=248
run grid240 DSVC_PP_GRID=240
run grid248p0 DSVC_PP_GRID=248 DSVC_MV_STREAM_PRIO=0
run base2 A=1
run l4 A=1
