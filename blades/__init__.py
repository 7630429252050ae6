"""Import-compatibility shim: ``blades.*`` of the reference resolves to ``blades_amd``."""
