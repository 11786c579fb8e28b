#!/bin/bash
# Run ON the GPU box: kernel-trace statistics of tools/bench_scopes.py with and without one environment switch
#   tools/kt_ab.sh <tag> <NAME=VALUE> [label=C2] [steps=200]
tag=$1; sw=$2; label=${3:-C2}; steps=${4:-200}
out=$GRAFT_REPO_ROOT/gpurun_out/kt_$tag
rm -rf "$out"; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/a" -o run -- python3 tools/bench_scopes.py $label $steps > "$out/a.log" 2>&1
export "CHSPH_$sw"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/b" -o run -- python3 tools/bench_scopes.py $label $steps > "$out/b.log" 2>&1
for v in a b; do
  echo "== $v: $(tail -1 $out/$v.log)"
  f=$(find "$out/$v" -name '*kernel_stats.csv' | head -1)
  python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:14]:
    print("  %-28s calls %6s avg %9.2f us  %5s %%" % (r["Name"].split("(")[0].replace("void chsph::", "").replace("chsph::", "")[:28], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
done > "$out/summary.txt" 2>&1
cat "$out/summary.txt"
