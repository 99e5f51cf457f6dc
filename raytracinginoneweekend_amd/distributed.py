"""One process per GPU: each rank renders its interleaved tiles, the tile buffers are gathered with ONE
all_gather (RCCL over xGMI when the backend is "nccl"; gloo on CPU in the tests), and the gathered
buffer is scattered back to scanline order by rtk_deinterleave_device.

The scene is tiny and read-only, so it is replicated; there is no exchange during rendering
(SURVEY.md §8e).  torch is used here for device memory and the collective only.
"""
from __future__ import annotations

import torch
import torch.distributed as dist

from . import Context, local_tile_count, make_params

TILE_PIXELS = 64


def tiles_per_rank_padded(W: int, H: int, world: int, tile_w: int = 8, tile_h: int = 8) -> int:
    """Rank 0 owns the most tiles; every rank's buffer is padded to that count so all_gather is regular."""
    return local_tile_count(W, H, 0, world, tile_w, tile_h)


def gather_tiles(local_tiles: torch.Tensor, group=None) -> torch.Tensor:
    """[tprp, 64, 3] on every rank -> [world, tprp, 64, 3] on every rank (one collective)."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1:
        return local_tiles.unsqueeze(0)
    # concatenated along dim 0 (the layout both gloo and RCCL accept), viewed as [world, tprp, 64, 3]
    out = torch.empty((world * local_tiles.shape[0],) + tuple(local_tiles.shape[1:]), dtype=local_tiles.dtype, device=local_tiles.device)
    dist.all_gather_into_tensor(out, local_tiles.contiguous(), group=group)
    return out.view((world,) + tuple(local_tiles.shape))


class ShardedRenderer:
    """Persistent per-rank state: context(s), tile buffer(s), gathered buffer, image.

    The collective and the de-interleave of every frame are queued on ONE torch stream (`self.stream`, a non-default
    stream): the collective is issued under `torch.cuda.stream(self.stream)`, so RCCL orders it after the kernels that
    fill the tile buffer and before the kernel that reads the gathered one, and all collectives of a process are issued
    in program order on one stream.  (torch's default stream has handle 0, which rtk_context_set_stream reads as "use
    the context's own stream": work on it would NOT be ordered against torch's collectives.)

    With one context (the default) the render kernel is queued on that same stream.  With `extra_contexts`
    (experiment, bench.py --frames-in-flight 2) consecutive frames alternate between contexts, each rendering on a
    stream of its own, tied to the shared stream by events: the persistent kernel of frame i+1 starts filling the CUs
    that the tail of frame i leaves idle, while gathers still run one after another on `self.stream`.
    """

    def __init__(self, ctx: Context, W: int, H: int, rank: int, world: int, device: torch.device, tile_w: int = 8, tile_h: int = 8,
                 extra_contexts=()):
        self.ctx, self.W, self.H, self.rank, self.world = ctx, W, H, rank, world
        self.tile_w, self.tile_h = tile_w, tile_h
        self.device = device
        self.tprp = tiles_per_rank_padded(W, H, world, tile_w, tile_h)
        self.ctxs = [ctx, *extra_contexts]
        self.locals = [torch.zeros((self.tprp, TILE_PIXELS, 3), dtype=torch.float32, device=device) for _ in self.ctxs]
        self.local = self.locals[0]
        self.image = torch.zeros((H, W, 3), dtype=torch.float32, device=device)
        self.local_u8 = None     # allocated on the first render(..., output="u8")
        self.image_u8 = None
        self.stream = torch.cuda.Stream(device=device)
        self.stream.wait_stream(torch.cuda.current_stream(device))     # the zero-fills above
        assert self.stream.cuda_stream != 0
        # one context: everything on self.stream.  Several: each renders on its own stream
        self.render_streams = [self.stream] if len(self.ctxs) == 1 else [torch.cuda.Stream(device=device) for _ in self.ctxs]
        for rs in self.render_streams:
            if rs is not self.stream:
                rs.wait_stream(torch.cuda.current_stream(device))
        self._rendered = [None] * len(self.ctxs)     # events: tile buffer j filled / consumed by the gather
        self._consumed = [None] * len(self.ctxs)
        self._frame = 0
        ctx.set_stream(self.stream.cuda_stream)
        self._keep = None

    def params(self, spp: int, depth: int, seed: int = 0, run_len: int = 0):
        return make_params(self.W, self.H, spp, depth, seed, 0.001, self.rank, self.world, self.tile_w, self.tile_h, run_len)

    def render(self, cam22, spp: int, depth: int, seed: int = 0, want_stats: bool = False, assemble: bool = True, output: str = "f32"):
        """Render this rank's tiles, gather, de-interleave (asynchronous).  Returns stats (or None);
        the image is self.image (output="f32": colour sums) or self.image_u8 (output="u8": every rank applies
        write_color_binary to its own tiles first, so the gather moves a quarter of the bytes)
        once self.stream has been synchronised."""
        if output not in ("f32", "u8"):
            raise ValueError("output must be 'f32' or 'u8'")
        p = self.params(spp, depth, seed)
        j = self._frame % len(self.ctxs)
        self._frame += 1
        ctx, local, rs = self.ctxs[j], self.locals[j], self.render_streams[j]
        with torch.cuda.stream(rs):
            if rs is not self.stream and self._consumed[j] is not None:
                rs.wait_event(self._consumed[j])         # the gather of the frame that last used this tile buffer
            ctx.set_stream(rs.cuda_stream)
            st = ctx.render_tiles_device(cam22, p, local.data_ptr(), want_stats)
            if rs is not self.stream:
                self._rendered[j] = torch.cuda.Event()
                self._rendered[j].record(rs)
        if not assemble:
            return st
        with torch.cuda.stream(self.stream):
            if rs is not self.stream:
                self.stream.wait_event(self._rendered[j])
                ctx.set_stream(self.stream.cuda_stream)  # quantise / de-interleave of this frame go behind its gather
            if output == "u8" and self.local_u8 is None:
                self.local_u8 = torch.zeros((self.tprp, TILE_PIXELS, 3), dtype=torch.uint8, device=self.device)
                self.image_u8 = torch.zeros((self.H, self.W, 3), dtype=torch.uint8, device=self.device)
            if output == "f32":
                gathered = gather_tiles(local)
                ctx.deinterleave_device(gathered.data_ptr(), self.tprp, p, self.image.data_ptr())
            else:
                ctx.quantise_device(local.data_ptr(), self.tprp * TILE_PIXELS, spp, self.local_u8.data_ptr())
                gathered = gather_tiles(self.local_u8)
                ctx.deinterleave_u8_device(gathered.data_ptr(), self.tprp, p, self.image_u8.data_ptr())
            self._keep = gathered   # keep the buffer alive until the next frame has been queued behind this one
            if rs is not self.stream:
                self._consumed[j] = torch.cuda.Event()
                self._consumed[j].record(self.stream)
        return st

    def synchronize(self) -> None:
        for rs in self.render_streams:
            rs.synchronize()
        self.stream.synchronize()
