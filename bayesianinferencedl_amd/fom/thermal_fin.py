"""Drop-in for fom/thermal_fin.py::get_space (reference fom/thermal_fin.py:4-20).

The reference meshes the fin with mshr (CGAL, not reproducible -- SURVEY S3).  This
build uses a conforming lattice triangulation: ``resolution`` is mapped to the lattice
divisor m (40 -> m = 12 -> 1597 DoFs, in place of the reference's 1446)."""
from ..fem import FinMesh, FunctionSpace, Function, resolution_to_m  # noqa: F401

_spaces = {}


def get_space(resolution, m=None):
    m = resolution_to_m(resolution) if m is None else int(m)
    if m not in _spaces:
        _spaces[m] = FunctionSpace(FinMesh(m))
    return _spaces[m]
