"""Import alias: `import pp_amd` == the package directory
`3d-object-detection-for-autonomous-navigation_amd/` (whose name is not an identifier)."""
import importlib
import os
import sys

_root = os.path.dirname(os.path.abspath(__file__))
if _root not in sys.path:
    sys.path.insert(0, _root)
sys.modules[__name__] = importlib.import_module("3d-object-detection-for-autonomous-navigation_amd")
