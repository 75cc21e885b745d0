"""Skeleton graph -> K-partition adjacency stacks A (and A2, A3) -- host side, runs once per Model.

Drop-in for the reference's `net.utils.graph.Graph` (/root/reference/net/utils/graph.py:5-536):
same constructor, same attributes (`A`, `A2`, `A3`, `edge`, `center`, `num_node`, `hop_dis`,
`spatial_symmetric`), same values to the last bit (float64) and the same failures for the layout /
strategy pairs that fail upstream (tests/test_host_graph.py checks all 72 pairs against fixtures
generated from the reference).  Written independently of the oracle: hop distances come from a
breadth-first search instead of dense matrix powers, partitions are built with masks, and the
2-/3-hop widening walks neighbour lists.

Extras the HIP engine uses: `pattern()` = union sparsity pattern of the stacks, which bounds the
in-LDS adjacency lists of the graph-conv kernel.
"""
from collections import deque

import numpy as np

_OP_BONES = ((4, 3), (3, 2), (7, 6), (6, 5), (13, 12), (12, 11), (10, 9), (9, 8), (11, 5), (8, 2),
             (5, 1), (2, 1), (0, 1), (15, 0), (14, 0), (17, 15), (16, 14))
_OP_MIRROR = ((14, 15), (16, 17), (2, 5), (3, 6), (4, 7), (8, 11), (9, 12), (10, 13))
_NTU_BONES = ((1, 2), (2, 21), (3, 21), (4, 3), (5, 21), (6, 5), (7, 6), (8, 7), (9, 21), (10, 9),
              (11, 10), (12, 11), (13, 1), (14, 13), (15, 14), (16, 15), (17, 1), (18, 17), (19, 18),
              (20, 19), (22, 23), (23, 8), (24, 25), (25, 12))
_NTU_MIRROR = ((23, 25), (24, 22), (11, 7), (10, 6), (9, 5), (8, 12), (16, 20), (17, 13), (18, 14), (19, 15))
_NTU_HALF = ((1, 2), (2, 13), (3, 13), (4, 3), (5, 13), (6, 5), (7, 6), (8, 7), (9, 1), (10, 9), (11, 10),
             (12, 11), (14, 15), (15, 8))
_NTU_EDGE = ((1, 2), (3, 2), (4, 3), (5, 2), (6, 5), (7, 6), (8, 7), (9, 2), (10, 9), (11, 10), (12, 11),
             (13, 1), (14, 13), (15, 14), (16, 15), (17, 1), (18, 17), (19, 18), (20, 19), (21, 22), (22, 8),
             (23, 24), (24, 12))


def _shift(pairs):
    return [(a - 1, b - 1) for a, b in pairs]


# layout -> (num_node, bones (0-based), mirror pairs or None, centre joint)      graph.py:47-143
# None = the reference never defines spatial_symmetric for that layout and cannot construct it.
LAYOUTS = {
    'openpose': (18, list(_OP_BONES), list(_OP_MIRROR), 1),
    'openpose_gravity': (19, list(_OP_BONES) + [(18, j) for j in range(18)], None, 1),
    'openpose_sym': (18, list(_OP_BONES), list(_OP_MIRROR), 1),
    'ntu-rgb+d': (25, _shift(_NTU_BONES), [], 20),
    'ntu-rgb+d_half': (15, _shift(_NTU_HALF), [], 12),
    'ntu-rgb+d_gravity': (26, _shift(_NTU_BONES) + [(25, j) for j in range(25)], None, 20),
    'ntu-rgb+d_sym': (25, _shift(_NTU_BONES), _shift(_NTU_MIRROR), 20),
    'ntu_edge': (24, _shift(_NTU_EDGE), None, 2),
}


def _bfs_hops(n, links, limit):
    """All-pairs hop count over undirected `links`; pairs farther than `limit` (or unreachable) -> inf."""
    nbr = [[] for _ in range(n)]
    for a, b in links:
        if a != b:
            nbr[a].append(b)
            nbr[b].append(a)
    hop = np.full((n, n), np.inf)
    for src in range(n):
        hop[src, src] = 0
        todo = deque([src])
        while todo:
            u = todo.popleft()
            if hop[src, u] >= limit:
                continue
            for w in nbr[u]:
                if hop[src, w] == np.inf:
                    hop[src, w] = hop[src, u] + 1
                    todo.append(w)
    return hop


def _ring(hop, k):
    """Column-normalised indicator of {hop == 0} U {hop == k}   (get_norm + normalize_digraph, graph.py:453-505)."""
    ind = ((hop == 0) | (hop == k)).astype(np.float64)
    deg = ind.sum(0)
    inv = np.zeros_like(deg)
    nz = deg > 0
    inv[nz] = deg[nz] ** (-1)
    return ind * inv[None, :]


def _partition(hop, norm, centre, hops, nodes):
    """Spatial-configuration partitioning (graph.py:164-187): self / centripetal+same / centrifugal."""
    n = hop.shape[0]
    live = np.zeros((n, n), dtype=bool)
    live[:nodes, :nodes] = True
    dj = hop[:, centre][:, None]          # distance of the row joint j to the centre
    di = hop[:, centre][None, :]          # distance of the column joint i
    out = []
    for h in hops:
        at = (hop == h) & live
        same = np.where(at & (dj == di), norm, 0.0)
        closer = np.where(at & (dj > di), norm, 0.0)
        further = np.where(at & (dj < di), norm, 0.0)
        if h == 0:
            out.append(same)
        else:
            out.append(same + closer)
            out.append(further)
    return np.stack(out)


def _widen(bones_adj, stack, norm, kernel_size):
    """`add_one_distance` (graph.py:508-518): push every non-zero of partitions 1.. one bone outwards, taking
    the values of `norm`.  Order matters upstream (the scan sees its own insertions, and tests partition 1
    whatever partition it is filling); a per-column ascending sweep over a live mask reproduces it."""
    n = bones_adj.shape[0]
    nbrs = [np.flatnonzero(bones_adj[j] == 1) for j in range(n)]
    res = stack.copy()
    for part in range(1, kernel_size):
        cur = res[part]
        for i in range(n):
            j = 0
            while j < n:
                if cur[j, i] != 0:
                    cur[j, i] = norm[j, i]
                    for k in nbrs[j]:
                        if k != i and res[1][k, i] == 0:
                            cur[k, i] = norm[k, i]
                j += 1
    return res


class Graph:
    def __init__(self, layout='openpose', strategy='uniform', max_hop=3, dilation=1, kernel_size=3):
        self.max_hop, self.dilation, self.kernel_size = max_hop, dilation, kernel_size
        if layout not in LAYOUTS:
            raise ValueError("Do Not Exist This Layout.")
        n, bones, mirror, centre = LAYOUTS[layout]
        self.num_node, self.center = n, centre
        self.edge = [(i, i) for i in range(n)] + list(bones)
        if mirror is None:
            raise AttributeError("'Graph' object has no attribute 'spatial_symmetric'")
        self.spatial_symmetric = list(mirror)
        adj = np.zeros((n, n))
        for a, b in self.edge:
            adj[a, b] = adj[b, a] = 1
        self.adjacency_matrix = adj
        self.hop_dis = _bfs_hops(n, bones, n)                       # full shortest paths
        self.hop_dis_sym = _bfs_hops(n, list(bones) + self.spatial_symmetric, n)
        self.hop_dis23 = _bfs_hops(n, bones, max_hop)
        self._assemble(strategy)

    def _assemble(self, strategy):
        n = self.num_node
        hops = range(0, 2, self.dilation)
        n1 = _ring(self.hop_dis_sym, 1)
        if strategy == 'uniform':
            self.A = n1[None].copy()
        elif strategy == 'distance':
            self.A = np.stack([np.where(self.hop_dis == h, n1, 0.0) for h in hops])
        elif strategy in ('spatial', 'spatial_half'):
            self.A = _partition(self.hop_dis, n1, self.center, hops, n)
        elif strategy in ('openpose_gravity', 'ntu-rgb+d_gravity'):
            g = 18 if strategy == 'openpose_gravity' else 25
            if g >= n:
                raise IndexError('index %d is out of bounds for axis 0 with size %d' % (g, n))
            base = _partition(self.hop_dis, n1, self.center, hops, n - 1)
            extra = np.zeros((n, n))
            extra[g, :] = n1[g, :]
            extra[:, g] = n1[:, g]
            self.A = np.concatenate([base, extra[None]], 0)
        elif strategy in ('spatial_3', 'spatial_3_sym'):
            a1 = _partition(self.hop_dis, n1, self.center, hops, n)
            a2 = _widen(self.adjacency_matrix, a1, _ring(self.hop_dis, 2), self.kernel_size)
            a3 = _widen(self.adjacency_matrix, a2, _ring(self.hop_dis, 3), self.kernel_size)
            if strategy == 'spatial_3_sym':
                a1 = np.concatenate([a1, self._mirror_part(n1)[None]], 0)
                blank = np.zeros((1, n, n))
                a2, a3 = np.concatenate([a2, blank], 0), np.concatenate([a3, blank], 0)
            self.A, self.A2, self.A3 = a1, a2, a3
        elif strategy == 'spatial_sym':
            a1 = _partition(self.hop_dis, n1, self.center, hops, n)
            self.A = np.concatenate([a1, self._mirror_part(_ring(self.hop_dis, 2))[None]], 0)
        else:
            raise ValueError("Do Not Exist This Strategy")

    def _mirror_part(self, norm):
        s = np.zeros((self.num_node, self.num_node))
        for i, j in self.spatial_symmetric:       # one direction only, as upstream (graph.py:530-531)
            s[i, j] = norm[i, j]
        return s

    def __str__(self):
        return 'Graph(num_node=%d, K=%d)' % (self.num_node, self.A.shape[0])

    def pattern(self):
        """(K,V,V) bool: positions that can ever be non-zero in an importance-weighted combination."""
        p = self.A != 0
        for name in ('A2', 'A3'):
            if hasattr(self, name):
                p = p | (getattr(self, name) != 0)
        return p
