"""Stand-in for neat-python on hosts where it is not installed: re-exports neat_lite under the name `neat`
(generate_illusion.py:9 `import neat`).  A real neat-python installation always wins (compat.install appends this
directory behind site-packages)."""
from evolutionary_illusion_generator_amd.neat_lite import *  # noqa: F401,F403
from evolutionary_illusion_generator_amd import neat_lite as _nl

__all__ = [n for n in dir(_nl) if not n.startswith("_")]
