"""Host logic of rel2shape's launch grouping (sdfusion_txt2shape_model.py:493-511 slices the objects into mini-batches of 7;
whole mini-batches are coalesced into launches, dealt evenly -- ADVICE r4): no GPU needed."""
import types

from commonscenes_amd.sdfusion import SDFusionText2ShapeModel


def _slices(n, mini_B=7, launch_B=32, eta=0.0, lo=0):
    stub = types.SimpleNamespace(mini_B=7, launch_B=32)
    out = SDFusionText2ShapeModel._launch_slices(stub, lo, lo + n, mini_B, launch_B, eta)
    return [s.stop - s.start for s in out], out


def test_launches_are_whole_minibatches_dealt_evenly():
    assert _slices(32)[0] == [32]                          # five mini-batches, one launch (launch_B rounds UP to 35)
    assert _slices(35)[0] == [35]
    assert _slices(40)[0] == [21, 19]                      # six mini-batches over two launches: 3 + 3, not 35 + 5
    assert _slices(36)[0] == [21, 15]
    assert _slices(256)[0] == [35] * 5 + [28] * 2 + [25]   # 37 mini-batches over 8 launches
    assert _slices(5)[0] == [5]
    assert _slices(0)[0] == []
    assert _slices(9, launch_B=0)[0] == [7, 2]             # the reference's own loop
    assert _slices(40, launch_B=0)[0] == [7, 7, 7, 7, 7, 5]
    assert _slices(40, eta=0.5)[0] == [7, 7, 7, 7, 7, 5]   # noise drawn per slice and step: no coalescing
    assert _slices(40, mini_B=8, launch_B=16)[0] == [16, 16, 8]
    assert _slices(20, mini_B=32, launch_B=32)[0] == [20]


def test_slices_cover_the_range_in_order():
    for n in (1, 6, 7, 8, 13, 14, 33, 64, 100, 255):
        for lo in (0, 11):
            sizes, sl = _slices(n, lo=lo)
            assert sl[0].start == lo and sl[-1].stop == lo + n and sum(sizes) == n
            for a, b in zip(sl, sl[1:]):
                assert a.stop == b.start
            assert all(s % 7 == 0 for s in sizes[:-1]), (n, sizes)   # only the last launch may hold a ragged mini-batch
            assert max(sizes) <= 35
