"""Import alias: `import tell_amd` loads the package that lives in the
(non-identifier) directory `transform-and-tell_amd/`."""
import importlib.util
import os
import sys

_pkg_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'transform-and-tell_amd')
_spec = importlib.util.spec_from_file_location(
    'tell_amd', os.path.join(_pkg_dir, '__init__.py'), submodule_search_locations=[_pkg_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules['tell_amd'] = _mod
_spec.loader.exec_module(_mod)
