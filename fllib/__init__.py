"""Import-compatibility shim: ``fllib.*`` of the reference resolves to ``blades_amd``."""
