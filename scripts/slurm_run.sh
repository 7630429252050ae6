#!/bin/bash
# One 8 x MI355X node, one experiment file — the counterpart of the reference's SLURM wrapper
# (uppmax_run.sh:1-9: 16 cores, 2 GPUs, 20 h).  Trials of the grid are spread over the node's GPUs by the
# trial scheduler (one trial per GPU by default); a single large trial can instead take the whole node with
#   python -m blades_amd.parallel.launcher --nproc 8 -- -m blades_amd.train file <yaml>
#SBATCH -t 20:00:00
#SBATCH -N 1
#SBATCH -n 64
#SBATCH --gpus=8

export HSA_ENABLE_IPC_MODE_LEGACY=0
cd "$(dirname "$0")/.."
python -m blades_amd.train file "${1:-blades_amd/tuned_examples/fedavg_60_mnist.yaml}" "${@:2}"
