"""Import alias: ``import action_detection_amd`` -> the package in ./action-detection_amd/."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "action-detection_amd")
_spec = importlib.util.spec_from_file_location(
    "action_detection_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["action_detection_amd"] = _mod
_spec.loader.exec_module(_mod)
