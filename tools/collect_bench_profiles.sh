#!/bin/bash
# Run ON the GPU box (through gpurun): the profiles bench.py's roofline object refers to.
#   tools/collect_bench_profiles.sh <tag>
# 1. kernel-trace statistics of the DEFAULT bench command minus its two legs outside the timed region (the CPU baseline and the extra C3 run of
#    `single_gpu_same_workload`, whose 17 M-marker launches would be averaged into the same kernel names): the force kernel's average there must
#    agree with roofline.avg_launch_ms
# 2. HBM-traffic and issue-side counters on a shorter run of the same workload (PMC passes serialise the kernels), one pass per counter group
set -e
tag=$1
out=$GRAFT_REPO_ROOT/gpurun_out/prof_$tag
rm -rf "$out"; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
echo "kernel trace of the default bench command"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/kt" -o run -- python3 bench.py --no-cpu-baseline --same-workload-steps 0 > "$out/kt.log" 2>&1
SHORT="python3 bench.py --steps 20 --warmup 2 --exact-steps 0 --no-cpu-baseline --same-workload-steps 0"
echo "FETCH_SIZE / WRITE_SIZE / TCC / SQ / TCP passes"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$out/fetch" -o run -- $SHORT > "$out/fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$out/write" -o run -- $SHORT > "$out/write.log" 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d "$out/tcc" -o run -- $SHORT > "$out/tcc.log" 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE --output-format csv -d "$out/sq" -o run -- $SHORT > "$out/sq.log" 2>&1
rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum SQ_INSTS_VMEM_RD --output-format csv -d "$out/tcp" -o run -- $SHORT > "$out/tcp.log" 2>&1
# texture-address / vector-cache pipe of the gather-bound force kernel and LDS behaviour of the staged neighbour search.
# Few counters per pass: a pass that asked for all the TA / TCP stall counters below at once made rocprofv3 abort and hang in round 2 (more derived
# counters than the TA / TCP blocks of gfx950 have registers for; rocprofv3 fails while building the group, after the GPU is initialised).  Split like
# this, every pass has completed in every collection of rounds 3 and 4; the time limit stays as a guard.
pass() {  # pass <subdir> <counters...>
    local d=$1; shift
    echo "pmc pass $d: $*"
    timeout -k 10 180 rocprofv3 --pmc "$@" --output-format csv -d "$out/$d" -o run -- $SHORT > "$out/$d.log" 2>&1 || echo "  (pass $d failed or timed out)"
}
pass ta1 TA_BUSY_avr GRBM_GUI_ACTIVE
pass ta2 TA_TA_BUSY_sum
pass ta3 TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum
pass tcp2 TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum
pass tcp3 TCP_TCR_TCP_STALL_CYCLES_sum TCP_TCP_TA_ADDR_STALL_CYCLES_sum
pass lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS
{ echo "# per-kernel averages per launch (rocprofv3 --pmc, separate passes of: $SHORT)"; for g in sq tcp ta1 ta2 ta3 tcp2 tcp3 lds; do python3 tools/pmc_agg.py "$out/$g"; done; } > "$out/counters.txt" 2>&1 || true
python3 tools/prof_report.py "$out" > "$out/report.md"
tail -1 "$out/kt.log" > "$out/bench_line.json" || true
grep -h '"metric"' "$out/kt.log" > "$out/bench_line.json" || true
cat "$out/report.md" | head -14
