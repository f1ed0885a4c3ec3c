"""Import shims under the dotted module names the unchanged reference imports at module level
(/root/reference/generate_illusion.py:1-21, fitness_calculator.py:1-5).

``shims/`` holds from-scratch packages named like the reference's three git submodules -- ``chainer_prednet``,
``optical_flow``, ``pytorch_neat`` -- whose entry points run on the HIP engine; the names the reference imports but
never calls are inert placeholders.  ``optional/`` holds stand-ins for third-party packages a user's host normally
has (``neat`` = neat-python, ``cv2``, ``google.colab``); they are only put on the path on request or when the real
package is not importable.

    from evolutionary_illusion_generator_amd import compat
    compat.install()                       # sys.path: shims (+ optional stand-ins for what is missing)
    import generate_illusion as gi         # the reference's file, unedited, from the user's checkout
    compat.use_fast_path(gi)               # optional: one batched device pass per generation instead of PNG files
    gi.neat_illusion(out_dir, "prednet.npz", cfg_path, gi.StructureType.Circles, 160, 120, [3, 48, 96, 192])

Without ``use_fast_path`` the reference's own glue runs (PNG files under ./temp, per-genome Python loops) and only
the three submodule calls go to the GPU: test_prednet evaluates the whole population in one batched roll-out,
lucas_kanade and the create_cppn node calls run per genome.  There is no CPU fallback in either mode.
"""
import importlib.util
import os
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
SHIM_DIR = os.path.join(_HERE, "shims")
OPTIONAL_DIR = os.path.join(_HERE, "optional")
OPTIONAL = {"neat": "_neat", "cv2": "_cv2", "google.colab": "_colab"}  # importable name -> sys.path entry under optional/


def _missing(name):
    try:
        return importlib.util.find_spec(name) is None
    except (ImportError, ValueError):
        return True


def install(optional="missing"):
    """Put the shim packages on sys.path (in front, so that they win over half-initialised submodule directories of
    a reference checkout).  optional: "missing" (default) adds a stand-in only for packages that are not importable,
    "all" / "none", or an iterable of names out of ("neat", "cv2", "google.colab").  Returns the stand-ins added."""
    if SHIM_DIR not in sys.path:
        sys.path.insert(0, SHIM_DIR)
    if optional == "none":
        wanted = []
    elif optional == "all":
        wanted = list(OPTIONAL)
    elif optional == "missing":
        wanted = [n for n in OPTIONAL if _missing(n)]
    else:
        wanted = list(optional)
    added = []
    for name in wanted:
        d = os.path.join(OPTIONAL_DIR, OPTIONAL[name])
        if d not in sys.path:
            sys.path.append(d)  # behind everything else: a real installation always wins
        added.append(name)
    return added


def use_fast_path(module):
    """Rebind the reference module's fitness functions to the batched device path (same signatures):
    get_fitnesses_neat (generate_illusion.py:478), get_vectors / calculate_fitness (fitness_calculator.py:468,505)."""
    from .. import fitness
    for name in ("get_fitnesses_neat", "get_vectors", "calculate_fitness"):
        if hasattr(module, name):
            setattr(module, name, getattr(fitness, name))
    return module
