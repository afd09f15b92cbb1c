"""Directions of the integrated positional encoding's basis (the reference's `internal/geopoly.py:78-123`,
`generate_basis`): the vertices of a tesselated octahedron / icosahedron with antipodal pairs removed.

Written from the definition, not from the reference's code; what has to agree is the RESULT -- values and ORDER, because
column 16 j' + ... of `spatial_net.0.weight` belongs to direction b of degree j (coord.py:129-133, 102-126), so a
checkpoint only loads into the right columns when the directions come out in the reference's order.  That order is
fixed by three conventions, kept here: the seed polyhedron's vertex / face enumeration, barycentric subdivision points
enumerated (i, j) -> weights (i, j, v - i - j) / v over the faces in order with first occurrences kept, and antipodes
dropped by keeping the vertex that appears first.  tests/golden/geopoly.npz holds the reference's own output for
('octahedron', 1..2) and ('icosahedron', 1..3); tests/test_basis.py compares bit for bit.
"""
import numpy as np

_PHI = (np.sqrt(5.0) + 1.0) / 2.0

# seed solids: vertices (unit length after the common scale) and triangular faces
_ICO_VERTS = np.array([(-1, 0, _PHI), (1, 0, _PHI), (-1, 0, -_PHI), (1, 0, -_PHI), (0, _PHI, 1), (0, _PHI, -1),
                       (0, -_PHI, 1), (0, -_PHI, -1), (_PHI, 1, 0), (-_PHI, 1, 0), (_PHI, -1, 0), (-_PHI, -1, 0)]) / np.sqrt(_PHI + 2.0)
_ICO_FACES = np.array([(0, 4, 1), (0, 9, 4), (9, 5, 4), (4, 5, 8), (4, 8, 1), (8, 10, 1), (8, 3, 10), (5, 3, 8), (5, 2, 3), (2, 7, 3),
                       (7, 10, 3), (7, 6, 10), (7, 11, 6), (11, 0, 6), (0, 1, 6), (6, 1, 10), (9, 0, 11), (9, 11, 2), (9, 2, 5), (7, 2, 11)])
_OCT_VERTS = np.array([(0, 0, -1), (0, 0, 1), (0, -1, 0), (0, 1, 0), (-1, 0, 0), (1, 0, 0)], dtype=np.float64)


def _pairwise_sq_dist(a, b):
    """squared distances between the rows of a and the rows of b, through |x|^2 + |y|^2 - 2 x.y (the expansion matters:
    the near-zero entries decide which vertices count as equal)"""
    d = (a * a).sum(1)[:, None] + (b * b).sum(1)[None, :] - 2.0 * (a @ b.T)
    return np.maximum(d, 0.0)


def _octahedron_faces():
    """the eight faces = for every cube corner (sign triple) the three vertices at squared distance 2 from it"""
    corners = np.array([(sx, sy, sz) for sx in (-1, 1) for sy in (-1, 1) for sz in (-1, 1)], dtype=np.float64)
    hits = np.argwhere(_pairwise_sq_dist(corners, _OCT_VERTS) == 2.0)     # sorted by corner, then vertex: 3 per corner
    return np.sort(hits[:, 1].reshape(3, -1).T, axis=1)


def _subdivide(verts, faces, v, eps):
    if not isinstance(v, int):
        raise ValueError(f"v {v} must an integer")
    if v < 1:
        raise ValueError(f"v {v} must be >= 1")
    bary = np.array([(i, j, v - i - j) for i in range(v + 1) for j in range(v + 1 - i)]) / v
    pts = []
    for f in faces:
        p = bary @ verts[f, :]
        pts.append(p / np.sqrt((p * p).sum(1, keepdims=True)))
    pts = np.concatenate(pts, 0)
    d = _pairwise_sq_dist(pts, pts)
    first = np.array([int(np.flatnonzero(row <= eps)[0]) for row in d])    # first vertex each one coincides with
    return pts[np.unique(first), :]


def generate_basis(base_shape, angular_tesselation, remove_symmetries=True, eps=1e-4):
    """-> float32 [n, 3]: the basis directions, components in the reference's (reversed: z, y, x) order; MLP uses its
    transpose [3, n] (internal/models.py:483-484)."""
    if base_shape == "icosahedron":
        verts = _subdivide(_ICO_VERTS, _ICO_FACES, angular_tesselation, eps)
    elif base_shape == "octahedron":
        verts = _subdivide(_OCT_VERTS, _octahedron_faces(), angular_tesselation, eps)
    else:
        raise ValueError(f"base_shape {base_shape} not supported")
    if remove_symmetries:
        anti = _pairwise_sq_dist(verts, -verts) < eps                       # anti[i, k]: vertex k is the antipode of vertex i
        verts = verts[np.triu(anti).any(1), :]                             # keep i when its antipode comes at or after it
    return verts[:, ::-1].astype(np.float32)
