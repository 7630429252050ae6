#!/bin/bash
# round-3 GPU call 4: determinism test, serial kernel table of the headline round, PMC table of the shipping kernels,
# aggregator / adversary timings, small-config kernel tables
set -o pipefail
O=gpurun_out/r3; mkdir -p $O
timeout -k 10 300 python -u -m pytest tests/test_ops_gpu.py -m gpu -q -k "bitwise_deterministic" --timeout 120 --timeout-method thread > $O/pytest4.log 2>&1 || tail -30 $O/pytest4.log
tail -2 $O/pytest4.log
BLADES_IGEMM_PROF=2 timeout -k 10 200 python3 tools/r3/igemm_trace.py --n 12800 > $O/igemm_trace.log 2>&1 || { tail -20 $O/igemm_trace.log; exit 1; }
grep -v amdgpu.ids $O/igemm_trace.log
bash tools/profile_serial_round.sh $O/serial || exit 1
head -30 $O/serial/steady_kernel_stats.txt
bash tools/r3/profile_pmc3.sh || exit 1
timeout -k 10 400 python3 tools/bench_aggregators.py 1000 resnet18 > $O/aggregators.log 2>&1 || { tail $O/aggregators.log; exit 1; }
cat $O/aggregators.log
bash tools/profile_small_configs.sh || exit 1
cat gpurun_out/small/bench.log
