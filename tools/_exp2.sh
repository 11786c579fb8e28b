set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
L=gpurun_out/exp2.log; : > $L
export CHSPH_ARITH=1
P=gpurun_out/exp2_pmc; rm -rf $P; mkdir -p $P
for v in tile gather; do
  if [ $v = gather ]; then export CHSPH_NO_FORCE_TILE=1; fi
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU --output-format csv -d $P/${v}_a -o run -- python3 tools/bench_force.py C2 3 > $P/${v}_a.log 2>&1
  rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD --output-format csv -d $P/${v}_b -o run -- python3 tools/bench_force.py C2 3 > $P/${v}_b.log 2>&1
  rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum --output-format csv -d $P/${v}_c -o run -- python3 tools/bench_force.py C2 3 > $P/${v}_c.log 2>&1
  for x in a b c; do echo "== $v $x" >> $L; python3 tools/pmc_agg.py $P/${v}_$x k_force >> $L 2>&1 || true; done
done
cat $L
