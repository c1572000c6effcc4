"""The Python seam of the drop-in boundary.

The reference resolves its classes by dotted name -- ``train.model_class = model.network.HoloSceneNetwork``,
``train.loss_class = model.loss.HoloSceneLoss`` through ``utils.general.get_class`` (utils/general.py:188-194,
training/holoscene_train.py:137-151) -- and imports ``from model.network import HoloSceneNetwork``,
``from hashencoder.hashgrid import HashEncoder`` (model/network.py:10) directly.  ``install()`` makes those names resolve to this
package:

  * ``hashencoder``, ``hashencoder.hashgrid``, ``hashencoder.backend`` are replaced wholesale (importing the reference's
    ``hashencoder.backend`` would JIT-compile its CUDA sources, hashencoder/backend.py:12-24);
  * for ``model.network / ray_sampler / density / embedder / loss``: if the reference's own module is importable (we are running
    inside its checkout), the mirrored CLASSES are overlaid on it, so every other symbol the trainers import from those modules
    (e.g. ``model.loss.compute_scale_and_shift`` used by the evaluation code, holoscene_train.py:27) keeps working; otherwise
    the name is bound to this package's module.
"""
import importlib
import sys
import types

_WHOLESALE = ("hashencoder", "hashencoder.hashgrid", "hashencoder.backend")
_OVERLAY = {
    "model.network": ("ObjectImplicitNetworkGrid", "RenderingNetwork", "HoloSceneNetwork"),
    "model.ray_sampler": ("RaySampler", "UniformSampler", "ErrorBoundSampler"),
    "model.density": ("LaplaceDensity",),
    "model.embedder": ("Embedder", "get_embedder"),
    "model.loss": ("MonoSDFLoss", "HoloSceneLoss", "compute_scale_and_shift_batch"),
}


# classes the reference keeps in model/network.py (:1835-2209) that live in their own module here
_EXTRA = {"model.network": ("holoscene_amd.model.object_network", ("SingleObjectImplicitNetworkGrid", "SingleObjectRenderingNetwork", "ObjectSDFNetwork"))}


def _bind(name, module):
    """sys.modules entry + attribute on the parent package (``__import__('a.b')`` returns ``a`` and the caller walks attributes)."""
    sys.modules[name] = module
    parent, _, leaf = name.rpartition(".")
    if parent:
        setattr(sys.modules[parent], leaf, module)


def install():
    """Returns {dotted module name: "replaced" | "overlaid"}."""
    done = {}
    for name in _WHOLESALE:
        _bind(name, importlib.import_module("holoscene_amd." + name))
        done[name] = "replaced"
    for name, symbols in _OVERLAY.items():
        ours = importlib.import_module("holoscene_amd." + name)
        if name in _EXTRA:       # make the dotted names of the reference resolve on this package's module too
            extra_mod, extra_syms = _EXTRA[name]
            for sym in extra_syms:
                setattr(ours, sym, getattr(importlib.import_module(extra_mod), sym))
            symbols = tuple(symbols) + tuple(extra_syms)
        try:
            theirs = importlib.import_module(name)          # the reference's own module, when we run inside its checkout
        except ImportError:
            theirs = None
        if theirs is None or theirs is ours:
            pkg = name.partition(".")[0]
            if pkg not in sys.modules:                      # no reference package of that name: a namespace for the aliases
                sys.modules[pkg] = types.ModuleType(pkg)
                sys.modules[pkg].__path__ = []
            _bind(name, ours)
            done[name] = "replaced"
        else:
            for sym in symbols:
                if hasattr(ours, sym):
                    setattr(theirs, sym, getattr(ours, sym))
            done[name] = "overlaid"
    return done
