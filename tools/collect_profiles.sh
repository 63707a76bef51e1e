#!/bin/bash
# Runs on the GPU box (through gpurun): rocprofv3 summaries of bench.py for profiles/ (kernel trace + stats, and separate PMC passes).
#   gpurun --timeout 1100 -- 'bash tools/collect_profiles.sh r3p'
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-prof}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B=$GRAFT_REPO_ROOT/bench.py
run() {   # name, rocprof args..., -- bench args
  name=$1; shift
  echo "== $name"; 
  rocprofv3 "$@" > $OUT/$name.stdout 2> $OUT/$name.stderr
  echo "rc=$?"
}
run l1   --kernel-trace --stats --output-format csv -d $OUT/l1   -o l1   -- python3 $B --lanes 1 --steps 12 --no-side-legs --no-cpu-baseline
run l4   --kernel-trace --stats --output-format csv -d $OUT/l4   -o l4   -- python3 $B --steps 12 --no-side-legs --no-cpu-baseline
run f32  --kernel-trace --stats --output-format csv -d $OUT/f32  -o f32  -- python3 $B --lanes 1 --steps 6 --precision fp32 --no-side-legs --no-cpu-baseline
run pmc_fetch --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -o p -- python3 $B --lanes 1 --steps 4 --warmup 3 --no-side-legs --no-cpu-baseline
run pmc_write --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -o p -- python3 $B --lanes 1 --steps 4 --warmup 3 --no-side-legs --no-cpu-baseline
run pmc_fetch_f32 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch_f32 -o p -- python3 $B --precision fp32 --lanes 1 --steps 2 --warmup 2 --no-side-legs --no-cpu-baseline
run pmc_write_f32 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write_f32 -o p -- python3 $B --precision fp32 --lanes 1 --steps 2 --warmup 2 --no-side-legs --no-cpu-baseline
run pmc_sq --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $OUT/pmc_sq -o p -- python3 $B --lanes 1 --steps 3 --warmup 3 --no-side-legs --no-cpu-baseline
# keep the merged files small: stats + counter CSVs only (the raw traces are large)
find $OUT -name "*kernel_trace.csv" -size +20M -delete
ls -la $OUT/*/* | head -40
for n in l1 l4 f32; do tail -1 $OUT/$n.stdout; done
