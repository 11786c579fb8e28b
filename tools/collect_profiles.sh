#!/bin/bash
# Run ON the GPU box (through gpurun): kernel-trace statistics and the two HBM-traffic PMC passes of one command.
#   tools/collect_profiles.sh <tag> <program and args...>      e.g.  tools/collect_profiles.sh c2 python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline
# Counters go in their own runs, one TCC counter per pass (FETCH_SIZE and WRITE_SIZE do not fit together on gfx950).
set -e
tag=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/prof_$tag
rm -rf "$out"; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/kt" -o run -- "$@" > "$out/kt.log" 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$out/fetch" -o run -- "$@" > "$out/fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$out/write" -o run -- "$@" > "$out/write.log" 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d "$out/tcc" -o run -- "$@" > "$out/tcc.log" 2>&1
# issue-side and vector-cache counters of the same command (what bounds the pair loops: NOTEBOOK.md 3.3)
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE --output-format csv -d "$out/sq" -o run -- "$@" > "$out/sq.log" 2>&1
rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum SQ_INSTS_VMEM_RD --output-format csv -d "$out/tcp" -o run -- "$@" > "$out/tcp.log" 2>&1
{ echo "# per-kernel averages per launch (rocprofv3 --pmc, separate passes)"; python3 tools/pmc_agg.py "$out/sq"; python3 tools/pmc_agg.py "$out/tcp"; } > "$out/counters.txt" 2>&1 || true
python3 tools/prof_report.py "$out" > "$out/report.md"
cat "$out/report.md"
