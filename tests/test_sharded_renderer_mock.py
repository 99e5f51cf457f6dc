"""ShardedRenderer's control flow without a GPU: torch's stream objects are replaced by stand-ins and the context by a
recorder, so that the order of the calls a frame queues — render tiles, (quantise,) gather, de-interleave — and the
buffers they are handed are checked on the CPU.  (The real thing runs in tests/test_gpu_*.py.)"""
import contextlib

import numpy as np
import pytest
import torch

import raytracinginoneweekend_amd as rt
from raytracinginoneweekend_amd import distributed as rd


class FakeStream:
    cuda_stream = 0x1234

    def wait_stream(self, other):
        pass

    def synchronize(self):
        pass


class Recorder:
    def __init__(self):
        self.calls = []
        self.stream = None

    def set_stream(self, handle):
        self.stream = handle

    def render_tiles_device(self, cam, params, ptr, want_stats=False):
        self.calls.append(("render", params.rank, params.world_size, params.spp, ptr))
        return {"rays": 1} if want_stats else None

    def quantise_device(self, src, n_pixels, spp, dst):
        self.calls.append(("quantise", src, n_pixels, spp, dst))

    def deinterleave_device(self, src, tprp, params, dst):
        self.calls.append(("deinterleave_f32", src, tprp, dst))

    def deinterleave_u8_device(self, src, tprp, params, dst):
        self.calls.append(("deinterleave_u8", src, tprp, dst))


@pytest.fixture()
def fake_cuda(monkeypatch):
    monkeypatch.setattr(torch.cuda, "Stream", lambda device=None: FakeStream())
    monkeypatch.setattr(torch.cuda, "current_stream", lambda device=None: FakeStream())
    monkeypatch.setattr(torch.cuda, "stream", lambda s: contextlib.nullcontext())


def test_frame_queues_render_gather_deinterleave_on_its_own_stream(fake_cuda):
    W, H = 100, 60
    ctx = Recorder()
    sr = rd.ShardedRenderer(ctx, W, H, 0, 1, torch.device("cpu"))
    assert ctx.stream == FakeStream.cuda_stream                       # the context works on the renderer's stream, not on handle 0
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
