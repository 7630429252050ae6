#!/bin/bash
# The north-star configuration through the CLI with the round-3 build: 1000 clients, 200 ALIE, six aggregators, 40 rounds each
set -o pipefail
mkdir -p gpurun_out/r3/e2e18
( while true; do sleep 50; echo "[alive $(date +%T)]"; done ) &
HB=$!
timeout -k 10 1050 python -m blades_amd.train file blades_amd/tuned_examples/fedsgd_resnet18_cifar10_1000clients.yaml --synthetic-data --local-mode --stop '{"training_iteration": 40}' --local-dir gpurun_out/r3/e2e18 > gpurun_out/r3/e2e18.log 2>&1
rc=$?
kill $HB
python - <<'PY'
import json, glob, statistics
for d in sorted(glob.glob('gpurun_out/r3/e2e18/*/FEDAVG_*')):
    rows=[json.loads(l) for l in open(d+'/result.json')]
    ts=[r['time_this_iter_s'] for r in rows if r['training_iteration']>3 and r.get('acc_top_1') is None]
    ev=[r for r in rows if r.get('acc_top_1') is not None]
    med=statistics.median(ts)
    print(f"{d.split('aggregator=')[1]:18s} rounds {rows[-1]['training_iteration']:3d}  median round {med*1e3:7.1f} ms = {1/med:.3f} rounds/s   train_loss {rows[-1]['train_loss']:.4f}  acc@{ev[-1]['training_iteration'] if ev else '-'} {ev[-1]['acc_top_1'] if ev else '-'}  phases {rows[-1].get('timers',{}).get('attack_ms')} / {rows[-1].get('timers',{}).get('aggregate_and_step_ms')} ms attack / aggregate")
PY
exit $rc
