"""pymra_amd - MI355X (gfx950) implementation of pyMRA's per-node prior/posterior inference path,
behind pyMRA's own Python surface (MRATree / MRATools).  See DESIGN.md."""
from . import MRATools            # noqa: F401
from .MRATree import MRATree      # noqa: F401
from . import DataLoader           # noqa: F401

__all__ = ["MRATree", "MRATools", "DataLoader"]
