set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
L=gpurun_out/exp4.log; : > $L
export CHSPH_ARITH=1 CHSPH_NO_FORCE_QUAD=1 CHSPH_NO_PACKED=1
P=gpurun_out/exp4_pmc; rm -rf $P; mkdir -p $P
for d in 0 2 3; do
  export CHSPH_FAST_DBG=$d
  rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVE_CYCLES --output-format csv -d $P/d${d} -o run -- python3 tools/bench_force.py C2 3 > $P/d${d}.log 2>&1
  rocprofv3 --kernel-trace --stats --output-format csv -d $P/k${d} -o run -- python3 tools/bench_force.py C2 3 > $P/k${d}.log 2>&1
  echo "== DBG=$d" >> $L; python3 tools/pmc_agg.py $P/d${d} k_force >> $L 2>&1 || true
  grep "k_force" $P/k${d}/*kernel_stats.csv | cut -c1-200 >> $L || true
done
cat $L
