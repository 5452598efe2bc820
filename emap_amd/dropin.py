"""Make EMAP's unmodified ``src/runner`` + ``main.py`` use emap_amd.

    import emap_amd.dropin; emap_amd.dropin.install()      # before `from src.runner... import`

registers ``src.models.udf_model``, ``src.models.udf_renderer_blending``, ``src.models.embedder`` and
``src.models.loss`` in ``sys.modules`` as aliases of the emap_amd modules of the same names, so
``runner_base.py:9-13``'s imports resolve to the HIP-backed classes.  See INTEGRATION.md.
"""
import importlib
import sys
import types

_ALIASES = {"src.models.udf_model": "emap_amd.udf_model",
            "src.models.udf_renderer_blending": "emap_amd.udf_renderer_blending",
            "src.models.embedder": "emap_amd.embedder",
            "src.models.loss": "emap_amd.loss"}


def install(force: bool = True):
    for pkg in ("src", "src.models"):
        if pkg not in sys.modules:
            try:
                importlib.import_module(pkg)
            except ImportError:
                m = types.ModuleType(pkg)
                m.__path__ = []
                sys.modules[pkg] = m
    for alias, real in _ALIASES.items():
        if force or alias not in sys.modules:
            mod = importlib.import_module(real)
            sys.modules[alias] = mod
            setattr(sys.modules["src.models"], alias.rsplit(".", 1)[1], mod)
    # extraction queries (SURVEY par. 8 f2): the reference module keeps its other functions; only the two query routines
    # are replaced, and only if the module can be imported at all (it needs nothing but torch)
    try:
        ep = importlib.import_module("src.edge_extraction.extract_pointcloud")
    except Exception:
        ep = None
    if ep is not None:
        from . import extraction
        ep.get_udf_normals_grid = extraction.get_udf_normals_grid
        ep.get_udf_normals_slow = extraction.get_udf_normals_slow
    return sorted(_ALIASES)
