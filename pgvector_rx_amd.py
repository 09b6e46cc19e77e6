"""Import alias: the package directory is named `pgvector-rx_amd` (not a Python identifier)."""
import importlib
import sys

sys.modules[__name__] = importlib.import_module("pgvector-rx_amd")
