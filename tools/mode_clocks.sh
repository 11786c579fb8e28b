#!/bin/bash
# Run ON the GPU box: clocks and power of the GPU (rocm-smi / amd-smi) sampled every 2 s while tools/ab_exp.py times the C2 step -- is the speed mode of
# k_force_fast (NOTEBOOK.md 3.3.1) a clock / power state?   tools/mode_clocks.sh [steps=4000]
steps=${1:-4000}
/opt/rocm/bin/rocm-smi --showclkfrq 2>/dev/null | grep -v "^$" | head -40
python3 tools/ab_exp.py C2 $steps 0 > /tmp/ab.log 2>/dev/null &
pid=$!
while kill -0 $pid 2>/dev/null; do
  /opt/rocm/bin/rocm-smi --showclocks --showpower --showuse 2>/dev/null | grep -i "sclk\|mclk\|fclk\|power (W)\|GPU use" | sed 's/.*: //' | tr '\n' ' '
  echo
  sleep 2
done
echo "$(tail -1 /tmp/ab.log)"
(command -v amd-smi >/dev/null && amd-smi metric -c -p 2>/dev/null | head -60) || true
