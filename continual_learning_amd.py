"""Import shim: ``import continual_learning_amd`` loads the package that lives in ``continual-learning_amd/``
(a hyphen is not importable).  The module replaces itself in sys.modules with the real package."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'continual-learning_amd')
_spec = importlib.util.spec_from_file_location('continual_learning_amd', os.path.join(_dir, '__init__.py'),
                                               submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules['continual_learning_amd'] = _mod
_spec.loader.exec_module(_mod)
