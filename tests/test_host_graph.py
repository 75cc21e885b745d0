"""CPU: the product's graph builder (ist-gcn_amd/net/utils/graph.py) against the reference-generated
fixtures (bit exact, float64) for all 72 layout x strategy pairs, incl. the pairs that fail upstream."""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN
from istgcn_amd.net.utils.graph import Graph

MAN = json.load(open(os.path.join(GOLDEN, 'graph_g1.json')))


@pytest.mark.parametrize('tag', sorted(MAN))
def test_graph_matches_reference(tag, golden):
    g1 = golden('graph_g1.npz')
    lay, st = tag.split('|')
    rec = MAN[tag]
    if 'error' in rec:
        with pytest.raises(Exception) as ei:
            Graph(layout=lay, strategy=st)
        assert type(ei.value).__name__ == rec['error']
        return
    g = Graph(layout=lay, strategy=st)
    assert (g.num_node, g.center, g.A.shape[0]) == (rec['num_node'], rec['center'], rec['K'])
    assert np.array_equal(np.asarray(g.edge), g1[tag + '|edge'])
    assert g.A.dtype == np.float64 and np.array_equal(g.A, g1[tag + '|A'])
    assert hasattr(g, 'A2') == rec['has_A23']
    if rec['has_A23']:
        assert np.array_equal(g.A2, g1[tag + '|A2']) and np.array_equal(g.A3, g1[tag + '|A3'])
        pat = g.pattern()
        assert pat.shape == g.A.shape and pat.sum() >= (g.A != 0).sum()


def test_ctor_args_and_unknowns(golden):
    assert np.array_equal(Graph('ntu-rgb+d', 'spatial', max_hop=2).A, golden('graph_g1.npz')['ntu-rgb+d|spatial|max_hop2|A'])
    with pytest.raises(ValueError):
        Graph('no-such-layout', 'spatial')
    with pytest.raises(ValueError):
        Graph('ntu-rgb+d', 'no-such-strategy')


def test_nnz_counts_of_survey():
    """nnz(A)=73/79/85 for NTU spatial_3, 52 for openpose spatial."""
    g = Graph('ntu-rgb+d', 'spatial_3')
    assert [(a != 0).sum() for a in (g.A, g.A2, g.A3)] == [73, 79, 85]
    assert (Graph('openpose', 'spatial').A != 0).sum() == 52
    assert g.pattern().sum() == 187
