"""ShardedRenderer's control flow without a GPU: torch's stream objects are replaced by stand-ins and the context by a
recorder, so that the order of the calls a frame queues — render tiles, (quantise,) gather, de-interleave — and the
buffers they are handed are checked on the CPU.  (The real thing runs in tests/test_gpu_*.py.)"""
import contextlib

import numpy as np
import pytest
import torch

import raytracinginoneweekend_amd as rt
from raytracinginoneweekend_amd import distributed as rd


LOG = []


class FakeStream:
    count = 0

    def __init__(self):
        FakeStream.count += 1
        self.cuda_stream = 0x1000 + FakeStream.count

    def wait_stream(self, other):
        pass

    def wait_event(self, ev):
        LOG.append(("wait", self.cuda_stream, ev.tag))

    def synchronize(self):
        pass


class FakeEvent:
    count = 0

    def __init__(self):
        FakeEvent.count += 1
        self.tag = FakeEvent.count

    def record(self, stream):
        LOG.append(("record", stream.cuda_stream, self.tag))


class Recorder:
    def __init__(self):
        self.calls = []
        self.stream = None

    def set_stream(self, handle):
        self.stream = handle
        LOG.append(("set_stream", id(self), handle))

    def render_tiles_device(self, cam, params, ptr, want_stats=False):
        LOG.append(("render", id(self), self.stream))
        self.calls.append(("render", params.rank, params.world_size, params.spp, ptr))
        return {"rays": 1} if want_stats else None

    def quantise_device(self, src, n_pixels, spp, dst):
        self.calls.append(("quantise", src, n_pixels, spp, dst))

    def deinterleave_device(self, src, tprp, params, dst):
        LOG.append(("deinterleave", id(self), self.stream))
        self.calls.append(("deinterleave_f32", src, tprp, dst))

    def deinterleave_u8_device(self, src, tprp, params, dst):
        self.calls.append(("deinterleave_u8", src, tprp, dst))


@pytest.fixture()
def fake_cuda(monkeypatch):
    monkeypatch.setattr(torch.cuda, "Stream", lambda device=None: FakeStream())
    monkeypatch.setattr(torch.cuda, "current_stream", lambda device=None: FakeStream())
    monkeypatch.setattr(torch.cuda, "stream", lambda s: contextlib.nullcontext())
    monkeypatch.setattr(torch.cuda, "Event", FakeEvent)
    LOG.clear()


def test_frame_queues_render_gather_deinterleave_on_its_own_stream(fake_cuda):
    W, H = 100, 60
    ctx = Recorder()
    sr = rd.ShardedRenderer(ctx, W, H, 0, 1, torch.device("cpu"))
    assert ctx.stream == sr.stream.cuda_stream != 0                  # the context works on the renderer's stream, not on handle 0
    tprp = rt.local_tile_count(W, H, 0, 1)
    assert tuple(sr.local.shape) == (tprp, 64, 3) and tuple(sr.image.shape) == (H, W, 3)
    cam = np.zeros(22, np.float32)
    assert sr.render(cam, 16, 50, seed=3) is None
    assert [c[0] for c in ctx.calls] == ["render", "deinterleave_f32"]
    assert ctx.calls[0][1:4] == (0, 1, 16) and ctx.calls[0][4] == sr.local.data_ptr()
    assert ctx.calls[1][2] == tprp and ctx.calls[1][3] == sr.image.data_ptr()
    ctx.calls.clear()
    assert sr.render(cam, 16, 50, seed=3, want_stats=True, output="u8") == {"rays": 1}
    assert [c[0] for c in ctx.calls] == ["render", "quantise", "deinterleave_u8"]
    q = ctx.calls[1]
    assert q[1] == sr.local.data_ptr() and q[2] == tprp * 64 and q[3] == 16 and q[4] == sr.local_u8.data_ptr()
    assert ctx.calls[2][3] == sr.image_u8.data_ptr() and sr.image_u8.dtype == torch.uint8 and tuple(sr.image_u8.shape) == (H, W, 3)
    ctx.calls.clear()
    sr.render(cam, 4, 50, assemble=False)                             # tiles only: nothing is gathered
    assert [c[0] for c in ctx.calls] == ["render"]
    with pytest.raises(ValueError):
        sr.render(cam, 4, 50, output="f16")
    sr.synchronize()


def test_two_contexts_alternate_and_gathers_stay_on_one_stream(fake_cuda):
    """bench.py --frames-in-flight 2: frame i renders on context i % 2 and its own stream; every gather / de-interleave
    runs on the shared stream after waiting for that frame's render event; a tile buffer is rendered into again only
    after the gather that read it."""
    W, H = 64, 40
    a, b = Recorder(), Recorder()
    sr = rd.ShardedRenderer(a, W, H, 0, 1, torch.device("cpu"), extra_contexts=[b])
    shared = sr.stream.cuda_stream
    ra, rb = (s.cuda_stream for s in sr.render_streams)
    assert len({shared, ra, rb}) == 3 and sr.locals[0].data_ptr() != sr.locals[1].data_ptr()
    cam = np.zeros(22, np.float32)
    LOG.clear()
    for i in range(4):
        sr.render(cam, 8, 50, seed=i)
    renders = [e for e in LOG if e[0] == "render"]
    assert [e[1] for e in renders] == [id(a), id(b), id(a), id(b)] and [e[2] for e in renders] == [ra, rb, ra, rb]
    assert all(e[2] == shared for e in LOG if e[0] == "deinterleave")
    assert a.calls[0][4] == sr.locals[0].data_ptr() and b.calls[0][4] == sr.locals[1].data_ptr()
    # frame 0: record(render stream) -> shared stream waits for it -> ... -> record(shared) ; frame 2 waits for that on ra
    rec = [e for e in LOG if e[0] == "record"]
    waits = [e for e in LOG if e[0] == "wait"]
    first_rendered, first_consumed = rec[0], rec[1]
    assert first_rendered[1] == ra and first_consumed[1] == shared
    assert ("wait", shared, first_rendered[2]) in waits and ("wait", ra, first_consumed[2]) in waits
    at = [i for i, e in enumerate(LOG) if e[0] == "render"]
    assert at[1] < LOG.index(("wait", ra, first_consumed[2])) < at[2]          # before context a renders again
    sr.synchronize()
