"""mirt_trace_static (include/mirt.h): the allocation-free variant of mirt_trace -- device buffers in and out, every
intermediate in a static device array.  Checked against mirt_trace itself (bit-exact), two calls in a row, the second with
a smaller input; collected last (the file name sorts after the other tests)."""
import numpy as np
import pytest

from gpu_common import random_rays
from raytracer_amd import scenes

pytestmark = pytest.mark.gpu


def test_trace_static_equals_trace(mirt):
    torch = pytest.importorskip("torch")
    sc = scenes.conference_scene(20000)
    s = mirt.Scene(sc)
    big = random_rays(sc, 50000, 31, True)
    small = random_rays(sc, 1777, 32, True)
    dev = "cuda:0"
    for anyhit in (False, True):
        for trav in (0, 1, 2):
            idx = torch.full((len(big),), -77, dtype=torch.int32, device=dev)
            tb = torch.full((len(big),), 0x7777, dtype=torch.int32, device=dev)
            for rays in (big, small):                                   # the second call reuses the array with a smaller input
                n = len(rays)
                r_dev = torch.from_numpy(rays).to(dev)
                idx.fill_(-77)
                tb.fill_(0x7777)
                mirt.trace_static(s, r_dev.data_ptr(), n, idx.data_ptr(), tb.data_ptr(), anyhit=anyhit, traversal=trav)
                want_i, want_t = s.trace(rays, anyhit=anyhit, traversal=trav)
                got_i = idx.cpu().numpy()
                got_t = tb.cpu().numpy().view(np.uint32)
                assert (got_i[:n] == want_i).all(), (anyhit, trav, n, int((got_i[:n] != want_i).sum()))
                assert (got_t[:n] == want_t).all(), (anyhit, trav, n)
                assert (got_i[n:] == -77).all() and (got_t[n:] == 0x7777).all()      # nothing written beyond n
            assert (want_i >= 0).any() and (want_i < 0).any()
    # limits and the empty call
    r_dev = torch.zeros((8,), dtype=torch.float32, device=dev)
    mirt.trace_static(s, r_dev.data_ptr(), 0, idx.data_ptr(), tb.data_ptr())
    with pytest.raises(mirt.MirtError) as e:
        mirt.trace_static(s, r_dev.data_ptr(), mirt.TRACE_STATIC_MAX_RAYS + 1, idx.data_ptr(), tb.data_ptr())
    assert e.value.code == -4 and "MIRT_TRACE_STATIC_MAX_RAYS" in e.value.msg
    # rays_dev is read as float4: a pointer that is not 16-byte aligned is refused, not dereferenced
    r_dev = torch.zeros((64,), dtype=torch.float32, device=dev)
    with pytest.raises(mirt.MirtError) as e:
        mirt.trace_static(s, r_dev.data_ptr() + 4, 4, idx.data_ptr(), tb.data_ptr())
    assert e.value.code == -1 and "16-byte aligned" in e.value.msg
    s.close()


def test_trace_static_allocates_nothing(mirt):
    """The point of the entry point: device memory in use does not move across repeated calls (hipMemGetInfo before / after;
    torch's caching allocator is quiescent: every buffer is made before the first reading), and the scratch address the calls
    use is the one hipGetSymbolAddress returned for this device - results land where mirt_trace's do."""
    torch = pytest.importorskip("torch")
    sc = scenes.conference_scene(20000)
    s = mirt.Scene(sc)
    dev = "cuda:0"
    rays = random_rays(sc, 20000, 33, True)
    r_dev = torch.from_numpy(rays).to(dev)
    idx = torch.zeros((len(rays),), dtype=torch.int32, device=dev)
    tb = torch.zeros((len(rays),), dtype=torch.int32, device=dev)
    for anyhit in (False, True):                                     # warm: first-use costs (code objects) are not allocations of the call
        mirt.trace_static(s, r_dev.data_ptr(), len(rays), idx.data_ptr(), tb.data_ptr(), anyhit=anyhit)
    torch.cuda.synchronize()
    free0, _ = torch.cuda.mem_get_info()
    for k in range(20):
        mirt.trace_static(s, r_dev.data_ptr(), len(rays) - 37 * k, idx.data_ptr(), tb.data_ptr(), anyhit=bool(k & 1), traversal=k % 3)
    torch.cuda.synchronize()
    free1, _ = torch.cuda.mem_get_info()
    assert free1 == free0, f"device memory in use moved by {free0 - free1} bytes over 20 calls"
    s.close()
