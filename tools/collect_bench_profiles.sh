#!/bin/bash
# Run ON the GPU box (through gpurun): the profiles bench.py's roofline object refers to.
#   tools/collect_bench_profiles.sh <tag>
# 1. kernel-trace statistics of the DEFAULT bench command (the force kernel's average there must agree with roofline.avg_launch_ms)
# 2. HBM-traffic and issue-side counters on a shorter run of the same workload (PMC passes serialise the kernels), one pass per counter group
set -e
tag=$1
out=$GRAFT_REPO_ROOT/gpurun_out/prof_$tag
rm -rf "$out"; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/kt" -o run -- python3 bench.py --no-cpu-baseline > "$out/kt.log" 2>&1
SHORT="python3 bench.py --steps 20 --warmup 2 --exact-steps 0 --no-cpu-baseline"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$out/fetch" -o run -- $SHORT > "$out/fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$out/write" -o run -- $SHORT > "$out/write.log" 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d "$out/tcc" -o run -- $SHORT > "$out/tcc.log" 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE --output-format csv -d "$out/sq" -o run -- $SHORT > "$out/sq.log" 2>&1
rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum SQ_INSTS_VMEM_RD --output-format csv -d "$out/tcp" -o run -- $SHORT > "$out/tcp.log" 2>&1
{ echo "# per-kernel averages per launch (rocprofv3 --pmc, separate passes of: $SHORT)"; python3 tools/pmc_agg.py "$out/sq"; python3 tools/pmc_agg.py "$out/tcp"; } > "$out/counters.txt" 2>&1 || true
python3 tools/prof_report.py "$out" > "$out/report.md"
tail -1 "$out/kt.log" > "$out/bench_line.json" || true
grep -h '"metric"' "$out/kt.log" > "$out/bench_line.json" || true
cat "$out/report.md" | head -14
