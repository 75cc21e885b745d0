"""CPU: the C-ABI library builds for gfx950, loads, exports every symbol of include/istgcn.h, and the host-side
geometry queries (no device work) agree with the packing code; the drop-in Models honour the state_dict contract
and refuse to compute without the GPU (no fallback)."""
import ctypes
import importlib
import json
import os

import pytest
import torch

from conftest import GOLDEN

SD = json.load(open(os.path.join(GOLDEN, 'state_dict_g5.json')))
CFG = {
    'st_gcnold': (dict(layout='ntu-rgb+d', strategy='spatial'), 60),
    'st_gcn_msgcn': (dict(layout='ntu-rgb+d', strategy='spatial_3'), 60),
    'st_gcn_mstcn_1x1': (dict(layout='openpose', strategy='spatial'), 400),
    'st_gcn_multi3_fix_3A_mstcn': (dict(layout='ntu-rgb+d', strategy='spatial_3'), 60),
    'st_gcn_mstcn_1x1_deep': (dict(layout='ntu-rgb+d', strategy='spatial'), 60),
    'st_gcn_mstcn': (dict(layout='ntu-rgb+d', strategy='spatial'), 60),
    'st_gcn_msgcn_new': (dict(layout='ntu-rgb+d', strategy='spatial_3'), 60),
    'st_gcn_deep_msgcn': (dict(layout='ntu-rgb+d', strategy='spatial_3'), 60),
}


@pytest.fixture(scope='module')
def lib():
    from istgcn_amd import _lib
    _lib.build()
    return _lib.load()


def test_library_exports_every_declared_symbol(lib):
    from istgcn_amd import _lib
    names = _lib.declared_symbols()
    assert len(names) >= 13
    for n in names:
        assert hasattr(lib, n), n


def test_shipped_library_has_no_experiment_hooks_and_knows_its_sources(lib):
    """ablation masks (`*_ABL`: results wrong), debug buffers (`*_DBG`: hipMalloc inside an entry point) and
    the experiment-only variant switches exist behind -DISTGCN_EXPERIMENT only; the default library reads dispatch
    overrides (documented in INTEGRATION.md), once.  #7: the library carries the hash of the sources it was built from."""
    import re
    from istgcn_amd import _lib
    blob = open(_lib.LIB_PATH, 'rb').read()
    names = set(m.group(0).decode() for m in re.finditer(rb'ISTGCN_[A-Z0-9_]{3,}', blob))
    bad = [n for n in names if n.endswith('_ABL') or '_DBG' in n or n in ('ISTGCN_RC_NCT', 'ISTGCN_GWG_OT', 'ISTGCN_RC_SPLIT', 'ISTGCN_DEBUG')]
    assert not bad, bad
    # (round 5: three of round 4's nine overrides are left -- one per kernel family that still has two generations in the
    #  library; each is exercised against the reference fixtures by tests/test_gpu_overrides.py)
    assert names <= {'ISTGCN_GCN_RC', 'ISTGCN_TCONV_LEAN', 'ISTGCN_TWG_LEAN'}, names
    assert _lib.build_id() == _lib.csrc_hash() and len(_lib.csrc_hash()) == 16


def test_geometry_and_packing_agree(lib):
    from istgcn_amd import ops
    for dt, epl in ((torch.float32, 4), (torch.bfloat16, 8)):
        for cin, cout, K in ((3, 64, 3), (64, 64, 3), (64, 128, 3), (256, 256, 3), (40, 24, 2), (64, 64, 4)):
            cce, nch, kkp, mttot, e = ops.gcn_geometry(cin, cout, K, ops._DT[dt])
            assert e == epl and cce % epl == 0 and nch * cce >= cin and kkp % (2 * epl) == 0 and kkp >= K * cce
            wr = torch.arange(cout * K * cin, dtype=torch.float32).view(cout, K, cin) % 251
            wp = ops.pack_gcn_weight(wr, dt)
            n_old = nch * mttot * kkp * 32
            rc = lib.istgcn_gcn_rc_layout(cin, cout, K, ops._DT[dt])
            assert wp.numel() == n_old + (K * cout * ((cin + 15) // 16 * 16) if rc else 0)
            assert lib.istgcn_gcn_rc_offset(cin, cout, K, ops._DT[dt]) == (n_old if rc else -1)
            if rc and dt == torch.float32:
                # float32 section (csrc/gcn_rc_f32.hip): [jt][k][q][s4][h][c][e] = Wr[32 jt + c][k][64 q + 32 h + 4 s4 + e]
                q = wp.reshape(-1)[n_old:].view(cout // 32, K, cin // 64, 8, 2, 32, 4)
                for (c, k, i) in ((0, 0, 0), (cout - 1, K - 1, cin - 1), (cout // 2, K // 2, cin // 3)):
                    assert float(q[c // 32, k, i // 64, (i % 32) // 4, (i % 64) // 32, c % 32, i % 4]) == float(wr[c, k, i])
            elif rc:
                # register-chained section (csrc/gcn_rc.hip): [jt][k][s][h][c][e] = Wr[32 jt + c][k][16 s + 8 h + e]
                q = wp.reshape(-1)[n_old:].view(cout // 32, K, (cin + 15) // 16, 2, 32, 8)
                for (c, k, i) in ((0, 0, 0), (cout - 1, K - 1, cin - 1), (cout // 2, K // 2, cin // 3)):
                    assert float(q[c // 32, k, i // 16, (i % 16) // 8, c % 32, i % 8]) == float(wr[c, k, i].to(dt))
            wp = wp.reshape(-1)[:n_old].view(nch, mttot, kkp // (2 * epl), 2, 32, epl)
            # spot-check the documented index map of include/istgcn.h
            for (c, k, i) in ((0, 0, 0), (cout - 1, K - 1, cin - 1), (cout // 2, K // 2, cin // 3)):
                ch, il = divmod(i, cce)
                kk = k * cce + il
                kg, rem = divmod(kk, 2 * epl)
                h, j = divmod(rem, epl)
                assert float(wp[ch, c // 32, kg, h, c % 32, j]) == float(wr[c, k, i].to(dt))
    taps = list(range(-4, 5))
    cc, nch, mttot, epl = ops.tconv_geometry(25, 64, 128, taps, 1, 0)
    wf = torch.randn(9, 128, 64)
    wp = ops.pack_tconv_weight(wf, 25, taps, 1, torch.float32)
    assert wp.shape == (nch, 9, cc // (2 * epl), mttot, 2, 32, epl)
    assert float(wp[1, 3, 1, 2, 1, 5, 2]) == float(wf[3, 2 * 32 + 5, 1 * cc + 1 * 2 * epl + 1 * epl + 2])


def test_invalid_arguments_are_rejected_before_launch(lib):
    # NULL pointers / bad sizes -> ISTGCN_EINVAL (1), nothing touches a device
    assert lib.istgcn_gcn_fwd(None, None, None, None, None, None, None, 0, None, 1, 1, 1, 1, 25, 3, 8, 3, 1, 1, 10, 0, 0, None) == 1
    assert lib.istgcn_tconv(None, None, None, None, 0, None, None, None, None, 0, 0, 1, 1, 1, 1, 25, 8, 8, 1, None, 1, 1, 0, 0, 0, None) == 1
    assert lib.istgcn_bn_finalize(None, 0, 0, ctypes.c_double(1.0), None, None, None, None, ctypes.c_float(0.1), ctypes.c_float(1e-5), 1, None, 4, None) == 1
    vals = [ctypes.c_int() for _ in range(5)]
    assert lib.istgcn_gcn_geometry(64, 64, 3, 7, *[ctypes.byref(v) for v in vals]) == 1


@pytest.mark.parametrize('tag', sorted(CFG))
def test_dropin_state_dict_contract(tag):
    gargs, nc = CFG[tag]
    m = importlib.import_module('istgcn_amd.net.' + tag).Model(3, nc, gargs, True, dropout=0.5)
    assert {k: list(v.shape) for k, v in m.state_dict().items()} == SD[tag]
    assert [k for k, _ in m.named_parameters()] == SD[tag + '#param_names']
    assert sum(p.numel() for p in m.parameters()) == SD[tag + '#nparam']
    assert hasattr(m.graph, 'edge') and m.graph.A.shape[1] == m.A.shape[1]
    # weights_init (recognition.py:31-44) finds real Conv2d / BatchNorm modules
    from istgcn_amd import harness
    m.apply(harness.weights_init)
    assert float(m.fcn.bias.abs().sum()) == 0.0
    # loads a checkpoint produced by the oracle/reference key-for-key, and round-trips
    m2 = importlib.import_module('istgcn_amd.net.' + tag).Model(3, nc, gargs, True)
    m2.load_state_dict(m.state_dict(), strict=True)


def test_no_cpu_fallback():
    from istgcn_amd.net import st_gcnold
    m = st_gcnold.Model(3, 60, dict(layout='ntu-rgb+d', strategy='spatial'), True)
    with pytest.raises(RuntimeError, match='MI355X'):
        m(torch.zeros(1, 3, 8, 25, 2))
    from istgcn_amd.net.utils.tgcn import ConvTemporalGraphical
    with pytest.raises(RuntimeError, match='MI355X'):
        ConvTemporalGraphical(3, 8, 3)(torch.zeros(1, 3, 4, 25), torch.zeros(3, 25, 25))


def test_edge_importance_weighting_off_and_aliases():
    from istgcn_amd.net import st_gcn, st_gcn_multi3_fix_3A_mstcn as full
    m = st_gcn.Model(3, 60, dict(layout='ntu-rgb+d', strategy='spatial'), False)
    assert m.edge_importance == [1] * 10
    m = full.Model(3, 60, dict(layout='ntu-rgb+d', strategy='spatial_3'), False, dropout=0.5)
    assert not any('edge_importance' in k for k in m.state_dict())
    assert sum('mstcn_importance' in k for k in m.state_dict()) == 10
