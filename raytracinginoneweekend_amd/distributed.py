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
    """Persistent per-rank state: context, tile buffer, gathered buffer, image.

    All GPU work of a frame — render kernel, finalise, all_gather, de-interleave — is queued on ONE torch stream
    (`self.stream`, a non-default stream whose handle is given to the context): the collective is issued under
    `torch.cuda.stream(self.stream)`, so RCCL orders it after the kernels that fill the tile buffer and before the
    kernel that reads the gathered one.  (torch's default stream has handle 0, which rtk_context_set_stream
    reads as "use the context's own stream": work on it would NOT be ordered against torch's collectives.)
    """

    def __init__(self, ctx: Context, W: int, H: int, rank: int, world: int, device: torch.device, tile_w: int = 8, tile_h: int = 8):
        self.ctx, self.W, self.H, self.rank, self.world = ctx, W, H, rank, world
        self.tile_w, self.tile_h = tile_w, tile_h
        self.device = device
        self.tprp = tiles_per_rank_padded(W, H, world, tile_w, tile_h)
        self.local = torch.zeros((self.tprp, TILE_PIXELS, 3), dtype=torch.float32, device=device)
        self.image = torch.zeros((H, W, 3), dtype=torch.float32, device=device)
        self.local_u8 = None     # allocated on the first render(..., output="u8")
        self.image_u8 = None
        self.stream = torch.cuda.Stream(device=device)
        self.stream.wait_stream(torch.cuda.current_stream(device))     # the zero-fills above
        assert self.stream.cuda_stream != 0
        ctx.set_stream(self.stream.cuda_stream)
        self._keep = None

    def params(self, spp: int, depth: int, seed: int = 0, run_len: int = 0):
        return make_params(self.W, self.H, spp, depth, seed, 0.001, self.rank, self.world, self.tile_w, self.tile_h, run_len)

    def render(self, cam22, spp: int, depth: int, seed: int = 0, want_stats: bool = False, assemble: bool = True, output: str = "f32"):
        """Render this rank's tiles, gather, de-interleave (asynchronous on self.stream).  Returns stats (or None);
        the image is self.image (output="f32": colour sums) or self.image_u8 (output="u8": every rank applies
        write_color_binary to its own tiles first, so the gather moves a quarter of the bytes)
        once self.stream has been synchronised."""
        if output not in ("f32", "u8"):
            raise ValueError("output must be 'f32' or 'u8'")
        p = self.params(spp, depth, seed)
        with torch.cuda.stream(self.stream):
            if output == "u8" and self.local_u8 is None:
                self.local_u8 = torch.zeros((self.tprp, TILE_PIXELS, 3), dtype=torch.uint8, device=self.device)
                self.image_u8 = torch.zeros((self.H, self.W, 3), dtype=torch.uint8, device=self.device)
            st = self.ctx.render_tiles_device(cam22, p, self.local.data_ptr(), want_stats)
            if assemble and output == "f32":
                gathered = gather_tiles(self.local)
                self.ctx.deinterleave_device(gathered.data_ptr(), self.tprp, p, self.image.data_ptr())
                self._keep = gathered   # keep the buffer alive until the next frame has been queued behind this one
            elif assemble:
                self.ctx.quantise_device(self.local.data_ptr(), self.tprp * TILE_PIXELS, spp, self.local_u8.data_ptr())
                gathered = gather_tiles(self.local_u8)
                self.ctx.deinterleave_u8_device(gathered.data_ptr(), self.tprp, p, self.image_u8.data_ptr())
                self._keep = gathered
        return st

    def synchronize(self) -> None:
        self.stream.synchronize()
