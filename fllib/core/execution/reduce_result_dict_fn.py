from blades_amd.parallel.reduce_result_dict_fn import _reduce_mean_results  # noqa: F401
