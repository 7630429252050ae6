from blades_amd.models.catalog import ModelCatalog  # noqa: F401
