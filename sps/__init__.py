"""Drop-in alias: ``import sps.models.models`` / ``sps.datasets.util`` / ``sps.datasets.blt_dataset``
resolve to the MI355X-native implementations in ``sps_amd`` (same module paths as the reference
package ``src/sps``, reference setup.py:3-19)."""
import importlib
import sys

_ALIASES = {
    "sps.models": "sps_amd.models",
    "sps.models.models": "sps_amd.models.models",
    "sps.models.minkunet": "sps_amd.models.minkunet",
    # the reference imports the backbone class from here (src/sps/models/models.py:10, c_ws/src/mos4d/scripts/mos4d.py:9)
    "sps.models.MinkowskiEngine": "sps_amd.models",
    "sps.models.MinkowskiEngine.customminkunet": "sps_amd.models.minkunet",
    "sps.models.MinkowskiEngine.minkunet": "sps_amd.models.minkunet",
    "sps.datasets": "sps_amd.datasets",
    "sps.datasets.util": "sps_amd.datasets.util",
    "sps.datasets.blt_dataset": "sps_amd.datasets.blt_dataset",
    "sps.datasets.augmentation": "sps_amd.datasets.augmentation",
}
for _alias, _target in _ALIASES.items():
    sys.modules[_alias] = importlib.import_module(_target)
models = sys.modules["sps.models"]
datasets = sys.modules["sps.datasets"]
