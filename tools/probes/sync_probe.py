"""Lists every host<->device synchronisation a steady-state FL round performs (torch's sync debug mode)."""
import os, sys, tempfile, warnings, traceback
sys.path.insert(0, os.getcwd())
os.environ.setdefault("BLADES_SYNTHETIC_DATA", "1")
import torch
from blades_amd.algorithms.fedavg import FedavgConfig

ds, model = (sys.argv[1], sys.argv[2]) if len(sys.argv) > 2 else ("cifar10", "resnet")
agg = sys.argv[3] if len(sys.argv) > 3 else "Median"
adv = sys.argv[4] if len(sys.argv) > 4 else "ALIEAdversary"
cfg = (FedavgConfig().resources(num_gpus_for_driver=1)
       .data(dataset_config={"type": ds, "num_clients": 60, "num_classes": 10, "train_batch_size": 64,
                             "n_train": 60 * 128, "n_test": 60 * 16, "data_root": tempfile.mkdtemp(), "use_cache": False})
       .training(global_model=model, server_config={"aggregator": {"type": agg}, "optimizer": {"lr": 0.1}})
       .client(client_config={"lr": 1.0})
       .adversary(num_malicious_clients=12, adversary_config={"type": f"blades.adversaries.{adv}"})
       .evaluation(evaluation_interval=0))
algo = cfg.build()
for _ in range(3):
    algo.train()
torch.cuda.synchronize()
seen = {}
def showwarning(message, category, filename, lineno, file=None, line=None):
    st = [f for f in traceback.extract_stack() if "/blades_amd/" in f.filename]
    key = tuple((os.path.relpath(f.filename), f.lineno) for f in st[-3:])
    seen[key] = seen.get(key, 0) + 1
warnings.showwarning = showwarning
warnings.simplefilter("always")
torch.cuda.set_sync_debug_mode("warn")
algo.train()
torch.cuda.set_sync_debug_mode("default")
print(f"{ds}/{model} {agg} {adv}: {sum(seen.values())} synchronising calls in one round")
for k, n in sorted(seen.items(), key=lambda kv: -kv[1]):
    print(f"  {n:3d} x  " + "  <-  ".join(f"{f}:{l}" for f, l in reversed(k)))
