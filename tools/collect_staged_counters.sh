#!/bin/bash
# Run ON the GPU box (through gpurun): counters of the LDS-staged force kernel (variant build `staged`, CHSPH_FORCE_STAGED=1) next to the shipped
# gather kernel's, same workload (C2, 20 steps after 30 of warm-up through tools/bench_scopes.py), one small PMC pass per group.
#   tools/build_variant.sh staged -DCHSPH_WITH_STAGED_FORCE   (in the build container)
#   tools/collect_staged_counters.sh <tag>
tag=$1
out=$GRAFT_REPO_ROOT/gpurun_out/staged_$tag
rm -rf "$out"; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export CHSPH_LIB=$GRAFT_REPO_ROOT/chrono-wisc_amd/csrc/variants/libchsph_mi_staged.so
RUN="python3 tools/bench_scopes.py C2 20"
for mode in 0 1; do
    export CHSPH_FORCE_STAGED=$mode
    d=$out/m$mode; mkdir -p $d
    timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d "$d/kt" -o run -- $RUN > "$d/kt.log" 2>&1 || echo "kt failed"
    pass() { local g=$1; shift; timeout -k 10 180 rocprofv3 --pmc "$@" --output-format csv -d "$d/$g" -o run -- $RUN > "$d/$g.log" 2>&1 || echo "  (pass $g failed or timed out)"; }
    pass sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE
    pass lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS
    pass ta1 TA_BUSY_avr GRBM_GUI_ACTIVE
    pass tcp TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum
    pass waves SQ_WAVES SQ_LEVEL_WAVES SQ_BUSY_CU_CYCLES
    { echo "# CHSPH_FORCE_STAGED=$mode: per-kernel averages per launch (rocprofv3 --pmc, separate passes of: $RUN)"; for g in sq lds ta1 tcp waves; do python3 tools/pmc_agg.py "$d/$g" k_force; done; } > "$out/counters_m$mode.txt" 2>&1
    python3 - "$d/kt" > "$out/kernel_stats_m$mode.txt" <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + '/**/*kernel_stats.csv', recursive=True):
    for r in list(csv.DictReader(open(f)))[:12]:
        print(r.get('Name', '')[:90], r.get('Calls'), r.get('AverageNs'), r.get('Percentage'))
PY
done
cat "$out"/counters_m*.txt "$out"/kernel_stats_m*.txt | cut -c1-400
