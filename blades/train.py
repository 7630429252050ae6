from blades_amd.train import *  # noqa: F401,F403
from blades_amd.train import main, train_app, load_experiments_from_file, run_fllib_experiments  # noqa: F401

if __name__ == "__main__":
    main()
