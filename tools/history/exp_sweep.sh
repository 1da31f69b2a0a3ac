#!/bin/bash
# usage: tools/exp_sweep.sh VAR v1 v2 ... -- [extra env assignments]   : bench step + phases for each value of an experiment knob
VAR=$1; shift
VALS=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do VALS+=("$1"); shift; done
[ "$1" == "--" ] && shift
for v in "${VALS[@]}"; do
  echo -n "$VAR=$v $* : "
  env $VAR=$v "$@" python bench.py --no-cpu-baseline --steps 10 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['phases_ms'])"
done
