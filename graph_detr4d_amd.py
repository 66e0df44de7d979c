"""Import alias: `import graph_detr4d_amd` loads the package that lives in `graph-detr4d_amd/`.

The package directory keeps the project's name (with a hyphen, which Python cannot import
directly); this one-file shim replaces itself in sys.modules with the real package, so
`import graph_detr4d_amd` and `from graph_detr4d_amd.x import y` work from the repo root.
"""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'graph-detr4d_amd')
_spec = importlib.util.spec_from_file_location(
    'graph_detr4d_amd', os.path.join(_dir, '__init__.py'), submodule_search_locations=[_dir])
_pkg = importlib.util.module_from_spec(_spec)
sys.modules['graph_detr4d_amd'] = _pkg
_spec.loader.exec_module(_pkg)
