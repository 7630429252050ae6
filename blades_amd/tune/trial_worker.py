"""Subprocess entry point of one trial: ``python -m blades_amd.tune.trial_worker spec.json``."""
import json
import sys


def main():
    with open(sys.argv[1]) as f:
        spec = json.load(f)
    from blades_amd.parallel.runtime import init_runtime, shutdown_runtime
    from blades_amd.tune.experiment import run_trial

    init_runtime()
    run_trial(spec["trial"], resume=spec.get("resume", False), verbose=spec.get("verbose", 1))
    shutdown_runtime()


if __name__ == "__main__":
    main()
