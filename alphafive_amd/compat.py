"""Unchanged reference imports: `install()` registers this package's mirrors under the module names the
reference's scripts import, so `main.py:2-15`, `self_play.py:2-8` and `choose_best_player.py:2-9` keep their import
lines (`from genData.player import Player`, `from genData.network import ResNet as model`, `import utils`,
`import config`, `from utils import RandomStack`) and get the MI355X engine's classes:

    genData.player      -> alphafive_amd.player      (genData/player.py:23   Player)
    genData.network     -> alphafive_amd.network     (genData/network.py:10  ResNet)
    genData.networkAPI  -> alphafive_amd.networkAPI  (genData/networkAPI.py:10 NetworkAPI)
    utils               -> alphafive_amd.utils       (utils.py: RandomStack, board helpers, BLACK_WIN/WHITE_WIN/DRAW)
    config              -> alphafive_amd.config      (config.py:2-27) unless the caller passes its own config module

    import alphafive_amd.compat as compat; compat.install()        # first line of the launcher, nothing else changes

`install(config=my_config_module)` keeps a caller-owned config (a run started inside the reference checkout wants its own
`config.py`).  `uninstall()` restores whatever was registered before.  Nothing here touches TensorFlow or pygame: the
trainer-side `import tensorflow as tf` lines of `main.py` are the caller's business (out of scope).
"""
import importlib
import sys
import types

_NAMES = {"genData.player": "alphafive_amd.player", "genData.network": "alphafive_amd.network",
          "genData.networkAPI": "alphafive_amd.networkAPI", "utils": "alphafive_amd.utils",
          "config": "alphafive_amd.config"}
_saved = None


def install(config=None, force=True):
    """Register the mirrors in sys.modules.  force=False leaves names alone that are already imported (e.g. the
    reference's own `utils` in a process that imported it first).  Returns the dict name -> module it registered."""
    global _saved
    if _saved is None:
        _saved = {k: sys.modules.get(k) for k in list(_NAMES) + ["genData"]}
    done = {}
    pkg = types.ModuleType("genData")
    pkg.__doc__ = "alphafive_amd.compat: stand-in for the reference's genData package"
    pkg.__path__ = []                                  # a package, with nothing to find on disk
    if force or "genData" not in sys.modules:
        sys.modules["genData"] = pkg
    else:
        pkg = sys.modules["genData"]
    for name, target in _NAMES.items():
        if not force and name in sys.modules:
            continue
        mod = config if (name == "config" and config is not None) else importlib.import_module(target)
        sys.modules[name] = mod
        if name.startswith("genData."):
            setattr(pkg, name.split(".", 1)[1], mod)
        done[name] = mod
    return done


def uninstall():
    """Undo install(): put back what the names pointed at before (or remove them)."""
    global _saved
    if _saved is None:
        return
    for k, v in _saved.items():
        if v is None:
            sys.modules.pop(k, None)
        else:
            sys.modules[k] = v
    _saved = None
