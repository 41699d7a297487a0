"""The VQGAN loss stack behind the reference's surface (reference losses/__init__.py:1-2): same import paths and class names, so
``conf/*.yaml`` (``_target_: losses.loss_img.VQLPIPSWithDiscriminator``, ``losses.VQVAEWithBCELoss``) resolve unchanged.

``losses.loss_seg`` (the BCE losses of the VQ-SEG stage, reference losses/loss_seg.py) is NOT restated here: it is a few lines of
elementwise torch off the hot path.  This package shadows the reference's ``losses`` package, so it falls
through to the reference's own file instead: ``__path__`` is extended with the ``losses/`` directory of the reference checkout --
``$MAS_REFERENCE_ROOT/losses`` or any later ``sys.path`` entry that holds a ``losses/loss_seg.py`` -- and ``losses.loss_seg``,
``losses.BCELossWithQuant`` and ``losses.VQVAEWithBCELoss`` resolve to it, unmodified.  The extension happens when this package is
imported (so ``import losses.loss_seg`` / ``from losses.loss_seg import X`` work: submodule imports consult ``__path__``, never the
module ``__getattr__``) and again, lazily, on attribute access (``sys.path`` is often completed after the first import)."""
import os
import sys

from .loss_img import VQLPIPSWithDiscriminator
from .lpips import LPIPS
from .lpips_with_object import LPIPSWithObject

_HERE = os.path.dirname(os.path.abspath(__file__))


def _reference_dirs():
    cands = []
    root = os.environ.get("MAS_REFERENCE_ROOT")
    if root:
        cands.append(os.path.join(root, "losses"))
    cands += [os.path.join(p or ".", "losses") for p in sys.path]
    out = []
    for d in cands:
        d = os.path.abspath(d)
        if d != _HERE and d not in out and os.path.isfile(os.path.join(d, "loss_seg.py")):
            out.append(d)
    return out


def _extend_path():
    for d in _reference_dirs():
        if d not in __path__:
            __path__.append(d)                          # behind this package's own directory: its modules keep winning


_extend_path()                                          # eager: the plain ``import losses.loss_seg`` form sees the reference's file


def __getattr__(name):
    if name in ("BCELossWithQuant", "VQVAEWithBCELoss", "loss_seg"):
        _extend_path()                                  # lazy retry: sys.path / MAS_REFERENCE_ROOT may have been set after the import
        try:
            import importlib
            mod = importlib.import_module(__name__ + ".loss_seg")
        except ImportError as e:
            raise AttributeError(
                f"losses.{name} is the reference's own losses/loss_seg.py (off the hot path, not restated in this package): put the "
                "reference checkout on sys.path behind this package or set MAS_REFERENCE_ROOT to it") from e
        return mod if name == "loss_seg" else getattr(mod, name)
    raise AttributeError(f"module {__name__!r} has no attribute {name!r}")
