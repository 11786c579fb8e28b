set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
L=gpurun_out/exp3.log; : > $L
export CHSPH_ARITH=1
P=gpurun_out/exp3_pmc; rm -rf $P; mkdir -p $P
for v in "$@"; do
  unset CHSPH_NO_FORCE_QUAD CHSPH_NO_PACKED CHSPH_FORCE_TILE
  if [ $v = packed ]; then export CHSPH_NO_FORCE_QUAD=1; fi
  if [ $v = scalar ]; then export CHSPH_NO_FORCE_QUAD=1 CHSPH_NO_PACKED=1; fi
  if [ $v = tile ]; then export CHSPH_FORCE_TILE=1; fi
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU --output-format csv -d $P/${v}_a -o run -- python3 tools/bench_force.py C2 3 > $P/${v}_a.log 2>&1
  rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum SQ_INSTS_VMEM_RD SQ_INSTS_LDS --output-format csv -d $P/${v}_c -o run -- python3 tools/bench_force.py C2 3 > $P/${v}_c.log 2>&1
  for x in a c; do echo "== $v $x" >> $L; python3 tools/pmc_agg.py $P/${v}_$x k_force >> $L 2>&1 || true; done
  grep "force kernel" $P/${v}_a.log | sed 's/; per-phase.*//' >> $L
done
cat $L
