#!/bin/bash
# A/B of the round-2 switches on the small configurations and the bf16 headline, one box
b() { timeout -k 10 200 python3 bench.py "$@" 2>/dev/null | tail -1 | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"])'; }
small="--clients 60 --batch 64 --steps 50 --warmup 5 --aggregator Median"
echo "mlp default            $(b --dataset mnist --model mlp $small)"
echo "mlp again              $(b --dataset mnist --model mlp $small)"
echo "cct default            $(b --dataset cifar10 --model cct $small)"
echo "cct STEM_IM2COL=0      $(BLADES_STEM_IM2COL=0 b --dataset cifar10 --model cct $small)"
echo "cct BN_STATS=0         $(BLADES_BN_STATS_IN_CONV=0 b --dataset cifar10 --model cct $small)"
echo "cct SEGMENTS=0         $(BLADES_IGEMM_SEGMENTS=0 b --dataset cifar10 --model cct $small)"
echo "cnn default            $(b --dataset fashionmnist --model cnn $small)"
echo "cnn STEM_IM2COL=0      $(BLADES_STEM_IM2COL=0 b --dataset fashionmnist --model cnn $small)"
echo "bf16 default           $(b --amp bf16 --steps 8 --warmup 2)"
echo "bf16 SEGMENTS=0        $(BLADES_IGEMM_SEGMENTS=0 b --amp bf16 --steps 8 --warmup 2)"
echo "bf16 STREAMS=2         $(BLADES_CHUNK_STREAMS=2 b --amp bf16 --steps 8 --warmup 2)"
