"""The hardware-queue probe behind the overlapped schedule's stream choice (pdgn_amd/streams.py)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_probe_detects_serialisation_and_plan_keeps_issuing_queue_free():
    from pdgn_amd import streams
    dev = torch.device("cuda", 0)
    main = torch.cuda.current_stream(dev)
    s = torch.cuda.Stream(device=dev)
    assert streams.shares_queue(s, s, dev), "two kernels on ONE stream must be seen as serialised"
    streams.reset()
    p = streams.plan(dev)
    assert p.probed
    assert streams.plan(dev) is p, "cached per (device, issuing stream)"
    roles = p.d + [p.lp, p.knn]
    assert len({r.cuda_stream for r in roles}) == 6, "six distinct streams"
    if p.n_queues >= 1:
        for r in roles:
            assert not streams.shares_queue(main, r, dev), "a side stream shares the issuing stream's hardware queue"
    if p.n_queues >= 3:
        assert streams.shares_queue(p.lp, p.knn, dev)
        # D1-D3 share a queue, D4 has its own (the backward of D4(G(z2)) is what the generator's backward waits for)
        assert streams.shares_queue(p.d[0], p.d[1], dev) and streams.shares_queue(p.d[0], p.d[2], dev)
        assert not streams.shares_queue(p.d[0], p.d[3], dev)
        assert not streams.shares_queue(p.d[0], p.knn, dev) and not streams.shares_queue(p.d[3], p.knn, dev)


def test_spin_rejects_long_waits():
    from pdgn_amd import _lib
    import ctypes
    assert _lib.lib().pdgn_spin(ctypes.c_uint(200000), ctypes.c_void_p(0)) == -1
