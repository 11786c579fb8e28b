set -e
run() { python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port $1 bench.py --gpus 2 --steps 100 --warmup 20 --same-gpu --backend gloo --config C2 --no-cpu-baseline; }
timeout -k 10 280 bash -c "$(declare -f run); run 29611" > gpurun_out/reh_one.log 2>&1
CHSPH_SLAB_TWO_ROUND=1 timeout -k 10 280 bash -c "$(declare -f run); run 29612" > gpurun_out/reh_two.log 2>&1
tail -1 gpurun_out/reh_one.log; tail -1 gpurun_out/reh_two.log
