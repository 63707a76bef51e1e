#!/bin/bash
# Run GPU steps one after the other on the gpurun box; a step that was killed by its timeout (exit 124 / 137) ends the sequence
# (never start another GPU step behind a hang), an ordinary failure (test assertion) does not.
#   usage: tools/gpu_steps.sh OUTDIR "name|seconds|command" ...
OUT=$1; shift
mkdir -p "$OUT"
rc_all=0
for spec in "$@"; do
  name=${spec%%|*}; rest=${spec#*|}; secs=${rest%%|*}; cmd=${rest#*|}
  echo "== $name (limit ${secs}s): $cmd"
  timeout -k 10 "$secs" bash -c "$cmd" > "$OUT/$name.log" 2>&1
  rc=$?
  echo "rc=$rc"; tail -n 6 "$OUT/$name.log"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "step $name hit its limit: stopping"; exit $rc; fi
  [ $rc -ne 0 ] && rc_all=$rc
done
exit $rc_all
