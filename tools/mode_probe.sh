#!/bin/bash
# Run ON the GPU box: the per-process speed mode of k_force_fast (NOTEBOOK.md 3.3.1) under an environment variable of the HIP runtime
#   tools/mode_probe.sh <VAR=VALUE> [processes=4]   -- alternating processes without / with the variable; force scope per launch of each
var=$1; n=${2:-4}
for i in $(seq 1 $n); do
  echo "plain:   $(python3 tools/ab_exp.py C2 150 0 2>/dev/null | tail -1)"
  echo "$var: $(env "$var" python3 tools/ab_exp.py C2 150 0 2>/dev/null | tail -1)"
done
