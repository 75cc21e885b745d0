"""ORACLE (test infrastructure, never shipped, never measured as the product).

Literal CPU restatement of the skeleton-graph builder of the reference,
`/root/reference/net/utils/graph.py` (citations below are file:line of that
file).  It deliberately follows the reference's *sequential* numpy
semantics (dense matrix powers, in-place scans) so that the product's
independent builder (`ist-gcn_amd/net/utils/graph.py`, BFS based) can be
checked against it, and both are pinned by the golden fixtures in
`tests/golden/graph_*.npz` that were produced by importing the reference
itself (`tests/golden/make_golden.py`).

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline`
leg may import this module.
"""
import numpy as np

# ---------------------------------------------------------------------------
# skeleton tables (graph.py:47-143).  `sym` is None where the reference never
# sets `self.spatial_symmetric` -- constructing such a layout raises
# AttributeError at graph.py:37-38, which we reproduce.
# ---------------------------------------------------------------------------
_OPENPOSE_BONES = [(4, 3), (3, 2), (7, 6), (6, 5), (13, 12), (12, 11), (10, 9),
                   (9, 8), (11, 5), (8, 2), (5, 1), (2, 1), (0, 1), (15, 0),
                   (14, 0), (17, 15), (16, 14)]
_OPENPOSE_SYM = [(14, 15), (16, 17), (2, 5), (3, 6), (4, 7), (8, 11), (9, 12), (10, 13)]
_NTU_BONES_1B = [(1, 2), (2, 21), (3, 21), (4, 3), (5, 21), (6, 5), (7, 6),
                 (8, 7), (9, 21), (10, 9), (11, 10), (12, 11), (13, 1),
                 (14, 13), (15, 14), (16, 15), (17, 1), (18, 17), (19, 18),
                 (20, 19), (22, 23), (23, 8), (24, 25), (25, 12)]
_NTU_HALF_1B = [(1, 2), (2, 13), (3, 13), (4, 3), (5, 13), (6, 5), (7, 6),
                (8, 7), (9, 1), (10, 9), (11, 10), (12, 11), (14, 15), (15, 8)]
_NTU_SYM_1B = [(23, 25), (24, 22), (11, 7), (10, 6), (9, 5), (8, 12), (16, 20),
               (17, 13), (18, 14), (19, 15)]
_NTU_EDGE_1B = [(1, 2), (3, 2), (4, 3), (5, 2), (6, 5), (7, 6), (8, 7), (9, 2),
                (10, 9), (11, 10), (12, 11), (13, 1), (14, 13), (15, 14),
                (16, 15), (17, 1), (18, 17), (19, 18), (20, 19), (21, 22),
                (22, 8), (23, 24), (24, 12)]


def _zero_based(pairs):
    return [(a - 1, b - 1) for a, b in pairs]


def _layout_table(layout):
    """-> (num_node, bones, sym_or_None, center)   graph.py:47-143"""
    if layout == 'openpose':
        return 18, list(_OPENPOSE_BONES), list(_OPENPOSE_SYM), 1
    if layout == 'openpose_gravity':
        return 19, list(_OPENPOSE_BONES) + [(18, j) for j in range(18)], None, 1
    if layout == 'openpose_sym':
        return 18, list(_OPENPOSE_BONES), list(_OPENPOSE_SYM), 1
    if layout == 'ntu-rgb+d':
        return 25, _zero_based(_NTU_BONES_1B), [], 20
    if layout == 'ntu-rgb+d_half':
        return 15, _zero_based(_NTU_HALF_1B), [], 12
    if layout == 'ntu-rgb+d_gravity':
        return 26, _zero_based(_NTU_BONES_1B + [(26, j) for j in range(1, 26)]), None, 20
    if layout == 'ntu-rgb+d_sym':
        return 25, _zero_based(_NTU_BONES_1B), _zero_based(_NTU_SYM_1B), 20
    if layout == 'ntu_edge':
        return 24, _zero_based(_NTU_EDGE_1B), None, 2
    raise ValueError("Do Not Exist This Layout.")


def _hops_by_matrix_power(adj, upto):
    """graph.py:396-420: hop[i,j] = smallest d<=upto with (adj^d)[i,j] > 0, else inf."""
    n = adj.shape[0]
    reach = np.stack([np.linalg.matrix_power(adj, d) for d in range(upto + 1)]) > 0
    hop = np.full((n, n), np.inf)
    for d in range(upto, -1, -1):
        hop[reach[d]] = d
    return hop


def hop_distances(num_node, edge, sym, max_hop):
    """graph.py:364-445 -> (adjacency_no_sym, hop_all, hop_with_sym, hop_upto_max_hop)"""
    adj = np.zeros((num_node, num_node))
    for i, j in edge:
        adj[j, i] = 1
        adj[i, j] = 1
    plain = adj.copy()
    with_sym = adj.copy()
    for i, j in sym:
        with_sym[j, i] = 1
        with_sym[i, j] = 1
    hop_sym = _hops_by_matrix_power(with_sym, num_node)
    hop23 = _hops_by_matrix_power(plain, max_hop)
    hop_all = _hops_by_matrix_power(plain, num_node)
    return plain, hop_all, hop_sym, hop23


def column_normalise(a):
    """graph.py:453-461  A . D^-1 with D = diag(column sums)."""
    col = a.sum(0)
    d = np.zeros_like(a)
    for i in range(a.shape[0]):
        if col[i] > 0:
            d[i, i] = col[i] ** (-1)
    return a @ d


def ring_norm(hop_k, hop, num_node):
    """graph.py:498-505: ones where hop in {0, hop_k}, column-normalised."""
    a = np.zeros((num_node, num_node))
    for h in (0, hop_k):
        a[hop == h] = 1
    return column_normalise(a)


def _spatial_partitions(hop, norm1, center, valid_hop, nodes):
    """graph.py:164-187 (same loop in every 'spatial*' strategy).
    `nodes` is the index range scanned (num_node, or num_node-1 for *_gravity)."""
    n = hop.shape[0]
    parts = []
    for h in valid_hop:
        root = np.zeros((n, n))
        close = np.zeros((n, n))
        further = np.zeros((n, n))
        for i in range(nodes):
            for j in range(nodes):
                if hop[j, i] != h:
                    continue
                if hop[j, center] == hop[i, center]:
                    root[j, i] = norm1[j, i]
                elif hop[j, center] > hop[i, center]:
                    close[j, i] = norm1[j, i]
                else:
                    further[j, i] = norm1[j, i]
        if h == 0:
            parts.append(root)
        else:
            parts.append(root + close)
            parts.append(further)
    return np.stack(parts)


def widen_one_hop(adj, a, norm, num_node, kernel_size):
    """graph.py:508-525 (`add_one_distance` via `get_A`).  The scan mutates the
    array it is scanning, and tests partition 1 (not `kernel`) for emptiness;
    both are reproduced."""
    res = a.copy()
    for kernel in range(1, kernel_size):
        for i in range(num_node):
            for j in range(num_node):
                if res[kernel][j, i] != 0:
                    res[kernel][j, i] = norm[j, i]
                    for k in range(num_node):
                        if adj[j][k] == 1 and res[1][k, i] == 0 and k != i:
                            res[kernel][k, i] = norm[k, i]
    return res


def append_symmetric(a, norm, num_node, sym):
    """graph.py:528-536: one-directional (i,j) entries as an extra partition."""
    s = np.zeros((num_node, num_node))
    for i, j in sym:
        s[i, j] = norm[i, j]
    return np.append(a, s[None], axis=0)


class GraphRef:
    """graph.py:5-42.  Attributes A (and A2, A3 for 'spatial_3*'), edge, center,
    num_node, hop_dis, as the reference exposes them."""

    def __init__(self, layout='openpose', strategy='uniform', max_hop=3,
                 dilation=1, kernel_size=3):
        self.max_hop, self.dilation, self.kernel_size = max_hop, dilation, kernel_size
        n, bones, sym, center = _layout_table(layout)
        self.num_node, self.center = n, center
        self.edge = [(i, i) for i in range(n)] + bones
        if sym is None:
            # graph.py:37-38 reads self.spatial_symmetric, never set for these layouts
            raise AttributeError("'Graph' object has no attribute 'spatial_symmetric'")
        self.spatial_symmetric = sym
        (self.adjacency_matrix, self.hop_dis, self.hop_dis_sym,
         self.hop_dis23) = hop_distances(n, self.edge, sym, max_hop)
        self._build(strategy)

    def _build(self, strategy):
        n = self.num_node
        valid_hop = range(0, 1 + 1, self.dilation)
        norm1 = ring_norm(1, self.hop_dis_sym, n)       # graph.py:151
        norm2 = ring_norm(2, self.hop_dis, n)           # graph.py:152
        norm3 = ring_norm(3, self.hop_dis, n)           # graph.py:153
        if strategy == 'uniform':
            self.A = norm1[None].copy()
        elif strategy == 'distance':
            a = np.zeros((len(valid_hop), n, n))
            for idx, h in enumerate(valid_hop):
                a[idx][self.hop_dis == h] = norm1[self.hop_dis == h]
            self.A = a
        elif strategy in ('spatial', 'spatial_half'):
            self.A = _spatial_partitions(self.hop_dis, norm1, self.center, valid_hop, n)
        elif strategy in ('openpose_gravity', 'ntu-rgb+d_gravity'):
            g = 18 if strategy == 'openpose_gravity' else 25
            a = _spatial_partitions(self.hop_dis, norm1, self.center, valid_hop, n - 1)
            grav = np.zeros((n, n))
            for i in range(n):                           # IndexError when g >= n, as upstream
                grav[g, i] = norm1[g, i]
                grav[i, g] = norm1[i, g]
            self.A = np.concatenate([a, grav[None]], 0)
        elif strategy in ('spatial_3', 'spatial_3_sym'):
            a = _spatial_partitions(self.hop_dis, norm1, self.center, valid_hop, n)
            a2 = widen_one_hop(self.adjacency_matrix, a, norm2, n, self.kernel_size)
            a3 = widen_one_hop(self.adjacency_matrix, a2, norm3, n, self.kernel_size)
            if strategy == 'spatial_3_sym':
                a = append_symmetric(a, norm1, n, self.spatial_symmetric)
                zero = np.zeros((1, n, n))
                a2 = np.append(a2, zero, axis=0)
                a3 = np.append(a3, zero, axis=0)
            self.A, self.A2, self.A3 = a, a2, a3
        elif strategy == 'spatial_sym':
            a = _spatial_partitions(self.hop_dis, norm1, self.center, valid_hop, n)
            self.A = append_symmetric(a, norm2, n, self.spatial_symmetric)   # graph.py:322 uses norm2
        else:
            raise ValueError("Do Not Exist This Strategy")
