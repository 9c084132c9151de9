"""Importable alias for the package directory ``diffusion-handwriting-generation.pytorch_amd/``.

The directory name required by the project layout is not a valid Python
identifier, so ``import dhg_amd`` loads that directory as the package
``dhg_amd`` (sub-modules resolve normally: ``dhg_amd.model`` ...).
"""
import importlib.util
import os
import sys

_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "diffusion-handwriting-generation.pytorch_amd")
_spec = importlib.util.spec_from_file_location(
    "dhg_amd", os.path.join(_DIR, "__init__.py"), submodule_search_locations=[_DIR])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["dhg_amd"] = _mod
_spec.loader.exec_module(_mod)
