"""Import shim: `import zs_amd` loads the package that lives in ./zerospeech-tts-without-t_amd/
(the directory name the build contract prescribes is not a valid Python identifier)."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'zerospeech-tts-without-t_amd')
_spec = importlib.util.spec_from_file_location('zs_amd', os.path.join(_dir, '__init__.py'),
                                               submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules['zs_amd'] = _mod
_spec.loader.exec_module(_mod)
