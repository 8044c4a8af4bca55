"""Import-free answer to "which audio model is this?" (mirrors ``mlx_audio/registry.py:34-154``).

Model families are the directories ``mlx_audio_amd/<kind>/models/<family>`` and each ``<kind>/utils.py`` may carry
a literal ``MODEL_REMAPPING`` dict of aliases; both are read from the file system / with ``ast`` so that nothing
heavy (torch, the HIP library) is imported.  The set of families is what THIS package ships (the hot path:
Kokoro), so ``SUPPORTED_MODEL_TYPES`` is a subset of the reference's.
"""
from __future__ import annotations

import ast
import re
from functools import lru_cache
from pathlib import Path
from typing import Dict, FrozenSet, List, Optional, Tuple

_PKG = Path(__file__).resolve().parent
_VOICE_KINDS_FIRST = ("tts", "stt", "sts")
# backbones that are far more often plain LLMs: only an explicit audio alias classifies them
_AMBIGUOUS_FAMILIES = frozenset({"llama", "qwen3", "dense"})


def _is_pkg_dir(p: Path) -> bool:
    return p.is_dir() and not p.name.startswith(("__", "."))


@lru_cache(maxsize=None)
def kinds() -> Tuple[str, ...]:
    """A kind is loadable only when ``<kind>/utils.py`` (its ``load_model``) exists: ``codec`` ships decoder engines but no loader."""
    found = sorted(p.name for p in _PKG.iterdir() if _is_pkg_dir(p) and (p / "models").is_dir() and (p / "utils.py").is_file())
    head = [k for k in _VOICE_KINDS_FIRST if k in found]
    return tuple(head + [k for k in found if k not in head])


def _exports_model(pkg: Path) -> bool:
    """True when the family's ``__init__.py`` binds the name ``Model`` (import, class or assignment), read with ``ast``: a family directory
    without a ``Model`` cannot be loaded, so it must not be advertised."""
    try:
        tree = ast.parse((pkg / "__init__.py").read_text())
    except (OSError, SyntaxError):
        return False
    for node in tree.body:
        if isinstance(node, (ast.Import, ast.ImportFrom)) and any((a.asname or a.name) == "Model" for a in node.names):
            return True
        if isinstance(node, ast.ClassDef) and node.name == "Model":
            return True
        if isinstance(node, ast.Assign) and any(isinstance(t, ast.Name) and t.id == "Model" for t in node.targets):
            return True
    return False


@lru_cache(maxsize=None)
def _families(kind: str) -> FrozenSet[str]:
    root = _PKG / kind / "models"
    return frozenset(p.name for p in root.iterdir() if _is_pkg_dir(p) and _exports_model(p)) if root.is_dir() else frozenset()


@lru_cache(maxsize=None)
def _remapping(kind: str) -> Dict[str, str]:
    src = _PKG / kind / "utils.py"
    try:
        tree = ast.parse(src.read_text())
    except (OSError, SyntaxError):
        return {}
    for node in tree.body:
        if isinstance(node, ast.Assign) and any(isinstance(t, ast.Name) and t.id == "MODEL_REMAPPING" for t in node.targets):
            try:
                return {str(k).lower(): str(v).lower() for k, v in ast.literal_eval(node.value).items()}
            except (ValueError, SyntaxError):
                return {}
    return {}


@lru_cache(maxsize=None)
def supported_model_types(kind: str) -> FrozenSet[str]:
    fams = _families(kind)
    remap = {k: v for k, v in _remapping(kind).items() if v in fams}  # an alias of a family that cannot load is not supported either
    return fams | frozenset(remap) | frozenset(remap.values())


def _name_parts(name: str) -> List[str]:
    return [s for s in re.split(r"[/\\\-_. ]+", name.lower()) if s]


def classify_model(model_type: str, model_name: str = "") -> Optional[str]:
    """config ``model_type`` (+ optional repo name) -> kind (``"tts"`` ...) or None; resolution order as in the
    reference loader: explicit alias, unique family match, then repo-name hints."""
    mt = (model_type or "").strip().lower()
    ks = kinds()
    for k in ks:
        if mt and mt in _remapping(k):
            return k
    hits = [k for k in ks if mt and mt in (_families(k) - _AMBIGUOUS_FAMILIES)]
    if len(hits) == 1:
        return hits[0]
    parts = _name_parts(model_name)
    for k in ks:
        fams, remap = _families(k) - _AMBIGUOUS_FAMILIES, _remapping(k)
        if mt and remap.get(mt, mt) in fams:
            return k
        if any(remap.get(s, s) in fams or s in remap for s in parts):
            return k
    return None


def is_supported_model(model_type: str, model_name: str = "") -> bool:
    return classify_model(model_type, model_name) is not None


SUPPORTED_MODEL_TYPES: Dict[str, FrozenSet[str]] = {k: supported_model_types(k) for k in kinds()}
