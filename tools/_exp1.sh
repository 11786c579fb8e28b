set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
L=gpurun_out/exp1.log; : > $L
export CHSPH_ARITH=1
for d in 1 2 3; do echo "== DBG=$d" >> $L; CHSPH_FAST_DBG=$d python tools/bench_force.py C2 10 >> $L 2>&1; done
for p in 27000 40000 53000 80000; do echo "== LDSPAD=$p" >> $L; CHSPH_FAST_LDSPAD=$p python tools/bench_force.py C2 10 >> $L 2>&1; done
P=gpurun_out/exp1_pmc; rm -rf $P; mkdir -p $P
rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum --output-format csv -d $P/a -o run -- python3 tools/bench_force.py C2 3 > $P/a.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD --output-format csv -d $P/b -o run -- python3 tools/bench_force.py C2 3 > $P/b.log 2>&1
rocprofv3 --pmc TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_READ_sum --output-format csv -d $P/c -o run -- python3 tools/bench_force.py C2 3 > $P/c.log 2>&1
rocprofv3 --pmc TA_BUSY_avr TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TD_TD_BUSY_sum GRBM_GUI_ACTIVE --output-format csv -d $P/d -o run -- python3 tools/bench_force.py C2 3 > $P/d.log 2>&1 || echo "pass d failed" >> $L
for x in a b c d; do echo "== pmc $x" >> $L; python3 tools/pmc_agg.py $P/$x k_force >> $L 2>&1 || true; done
cat $L
