"""Import shim: exposes the package directory
``truely-real-time-ai-generated-video-detection-framework-for-social-platforms_amd/`` (a name the
build contract fixes but Python cannot import) as the module ``truely_amd``."""
import importlib.util
import os
import sys

_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)),
                    "truely-real-time-ai-generated-video-detection-framework-for-social-platforms_amd")
_spec = importlib.util.spec_from_file_location("truely_amd", os.path.join(_DIR, "__init__.py"),
                                               submodule_search_locations=[_DIR])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["truely_amd"] = _mod
_spec.loader.exec_module(_mod)
