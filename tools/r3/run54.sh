#!/bin/bash
# headline round through the distributed code paths on ONE rank (RCCL process group of size 1, BLADES_FORCE_COLLECTIVES=1) vs plain, same box
set -e
mkdir -p gpurun_out; rm -f gpurun_out/run54.log
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1
echo "plain" | tee -a gpurun_out/run54.log
timeout -k 10 300 python3 bench.py --steps 5 --warmup 2 2>/dev/null | tail -1 | cut -c1-160 | tee -a gpurun_out/run54.log
echo "BLADES_FORCE_COLLECTIVES=1 (alltoall layout)" | tee -a gpurun_out/run54.log
BLADES_FORCE_COLLECTIVES=1 timeout -k 10 300 python3 bench.py --steps 5 --warmup 2 2>gpurun_out/run54.err | tail -1 | cut -c1-160 | tee -a gpurun_out/run54.log
echo "BLADES_FORCE_COLLECTIVES=1 --layout allgather" | tee -a gpurun_out/run54.log
BLADES_FORCE_COLLECTIVES=1 timeout -k 10 300 python3 bench.py --steps 5 --warmup 2 --layout allgather 2>>gpurun_out/run54.err | tail -1 | cut -c1-160 | tee -a gpurun_out/run54.log
