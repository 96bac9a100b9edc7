"""Import shim: the package directory is named `etx-tracer_amd/` (not a valid Python identifier), this module makes
`import etx_tracer_amd` resolve to it."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "etx-tracer_amd")
_spec = importlib.util.spec_from_file_location("etx_tracer_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_module = importlib.util.module_from_spec(_spec)
sys.modules["etx_tracer_amd"] = _module
_spec.loader.exec_module(_module)
