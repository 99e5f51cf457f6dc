"""World-size-2 (and 3) CPU test of the multi-rank path with the gloo backend: tile ownership,
padding, the single all_gather, and the de-interleave index math.  The tile CONTENT is synthetic (each
pixel carries its own global index), so no GPU is needed; the render that fills real tiles is covered by
tests/test_gpu_render.py::test_one_vs_n_ranks_bit_identical."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT
from np_mirror import deinterleave


def _worker(rank, world, W, H, port, out_dir):
    sys.path.insert(0, ROOT)
    import raytracinginoneweekend_amd as rt
    from raytracinginoneweekend_amd import distributed as rd
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    tw = th = 8
    tiles_x = (W + tw - 1) // tw
    tprp = rd.tiles_per_rank_padded(W, H, world)
    nlt = rt.local_tile_count(W, H, rank, world)
    assert nlt <= tprp
    local = torch.zeros((tprp, 64, 3), dtype=torch.float32)
    # what rtk_render_tiles_device would write: local tile k is global tile k*world + rank
    for k in range(nlt):
        gt = k * world + rank
        tx, ty = gt % tiles_x, gt // tiles_x
        for j in range(64):
            x, row = tx * tw + j % tw, ty * th + j // tw
            if x < W and row < H:
                p = row * W + x
                local[k, j] = torch.tensor([p, 2 * p + 1, rank], dtype=torch.float32)
    gathered = rd.gather_tiles(local)
    assert tuple(gathered.shape) == (world, tprp, 64, 3)
    img = deinterleave(gathered.numpy(), W, H, world)
    np.save(os.path.join(out_dir, f"img{rank}.npy"), img)
    # the quantised path: the same gather on uint8 tiles
    g8 = rd.gather_tiles((local % 251).to(torch.uint8))
    assert g8.dtype == torch.uint8 and tuple(g8.shape) == (world, tprp, 64, 3)
    np.save(os.path.join(out_dir, f"img8_{rank}.npy"), deinterleave(g8.numpy(), W, H, world))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,W,H", [(2, 40, 24), (2, 37, 23), (3, 50, 9)])
def test_shard_gather_deinterleave(tmp_path, world, W, H):
    port = 29000 + (os.getpid() * 7 + world * 13 + W) % 2000
    mp.spawn(_worker, args=(world, W, H, port, str(tmp_path)), nprocs=world, join=True)
    p = np.arange(W * H, dtype=np.float32).reshape(H, W)
    tiles_x = (W + 7) // 8
    rows, xs = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
    owner = ((rows // 8) * tiles_x + xs // 8) % world
    for r in range(world):
        img = np.load(tmp_path / f"img{r}.npy")
        assert np.array_equal(img[..., 0], p) and np.array_equal(img[..., 1], 2 * p + 1)
        assert np.array_equal(img[..., 2], owner)          # every pixel came from the rank that owns its tile
        img8 = np.load(tmp_path / f"img8_{r}.npy")
        assert img8.dtype == np.uint8 and np.array_equal(img8[..., 0], (p % 251).astype(np.uint8))


def _render_worker(rank, world, W, H, port, out_dir):
    """Each rank renders ITS tiles with the CPU build of the render kernel (the same device source the GPU runs),
    then the ranks exchange them exactly as bench.py does on RCCL."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import raytracinginoneweekend_amd as rt
    from raytracinginoneweekend_amd import distributed as rd
    from emul import Emul
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    e = Emul(rt.stock_scene())
    cam = rt.stock_camera(W / H)
    tprp = rd.tiles_per_rank_padded(W, H, world)
    tiles, st = e.render_tiles(cam, rt.make_params(W, H, 3, 12, 5, 0.001, rank, world), threads=2)
    local = torch.zeros((tprp, 64, 3), dtype=torch.float32)
    local[: len(tiles)] = torch.from_numpy(tiles)
    img = deinterleave(rd.gather_tiles(local).numpy(), W, H, world)
    rays = torch.tensor([st["rays"]], dtype=torch.int64)
    dist.all_reduce(rays)
    np.save(os.path.join(out_dir, f"render{rank}.npy"), img)
    np.save(os.path.join(out_dir, f"rays{rank}.npy"), rays.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_render_equals_single_rank(tmp_path):
    """World size 2 end to end on the CPU: the gathered, de-interleaved image has exactly the bits — and the ranks
    together trace exactly the rays — of a single-rank render (global-pixel RNG keys, integer accumulation)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import raytracinginoneweekend_amd as rt
    from emul import Emul
    world, W, H = 2, 37, 23
    port = 29000 + (os.getpid() * 11 + 977) % 2000
    mp.spawn(_render_worker, args=(world, W, H, port, str(tmp_path)), nprocs=world, join=True)
    ref, st = Emul(rt.stock_scene()).render(rt.stock_camera(W / H), W, H, 3, 12, seed=5, threads=2)
    for r in range(world):
        img = np.load(tmp_path / f"render{r}.npy")
        assert np.array_equal(img.view(np.uint32), ref.view(np.uint32))
        assert int(np.load(tmp_path / f"rays{r}.npy")[0]) == st["rays"]


class _EmulContext:
    """Stands in for rt.Context on the CPU: the entry points ShardedRenderer calls, taking the same raw pointers,
    backed by the CPU build of the render kernel and numpy restatements of the small kernels."""

    def __init__(self, scene):
        import ctypes
        from emul import Emul
        self._e = Emul(scene)
        self._ct = ctypes

    def _view(self, ptr, shape, dtype):
        n = int(np.prod(shape)) * np.dtype(dtype).itemsize
        return np.frombuffer((self._ct.c_char * n).from_address(ptr), dtype=dtype).reshape(shape)

    def set_stream(self, handle):
        pass

    def render_tiles_device(self, cam, params, ptr, want_stats=False):
        tiles, st = self._e.render_tiles(cam, params, threads=2)
        self._view(ptr, tiles.shape, np.float32)[...] = tiles
        return st if want_stats else None

    def quantise_device(self, src, n_pixels, spp, dst):
        from oracle import pyoracle as po
        self._view(dst, (n_pixels, 3), np.uint8)[...] = po.quantise(self._view(src, (n_pixels, 3), np.float32), spp)

    def _deint(self, src, tprp, params, dst, dtype):
        world = max(1, params.world_size)
        g = self._view(src, (world, tprp, 64, 3), dtype)
        self._view(dst, (params.height, params.width, 3), dtype)[...] = deinterleave(g, params.width, params.height, world)

    def deinterleave_device(self, src, tprp, params, dst):
        self._deint(src, tprp, params, dst, np.float32)

    def deinterleave_u8_device(self, src, tprp, params, dst):
        self._deint(src, tprp, params, dst, np.uint8)


class _NoStream:
    cuda_stream = 1

    def wait_stream(self, other): pass
    def wait_event(self, ev): pass
    def synchronize(self): pass


class _NoEvent:
    def record(self, stream): pass


def _sharded_worker(rank, world, W, H, port, out_dir):
    import contextlib
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import raytracinginoneweekend_amd as rt
    from raytracinginoneweekend_amd import distributed as rd
    torch.cuda.Stream = lambda device=None: _NoStream()
    torch.cuda.current_stream = lambda device=None: _NoStream()
    torch.cuda.stream = lambda s: contextlib.nullcontext()
    torch.cuda.Event = _NoEvent
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    scene = rt.stock_scene()
    cam = rt.stock_camera(W / H)
    sr = rd.ShardedRenderer(_EmulContext(scene), W, H, rank, world, torch.device("cpu"), extra_contexts=[_EmulContext(scene)])
    frames = []
    for i in range(3):                                   # contexts alternate: 0, 1, 0
        sr.render(cam, 2, 10, seed=10 + i)
        frames.append(sr.image.numpy().copy())
    sr.render(cam, 2, 10, seed=10, output="u8")
    np.save(os.path.join(out_dir, f"sh{rank}.npy"), np.stack(frames))
    np.save(os.path.join(out_dir, f"sh8_{rank}.npy"), sr.image_u8.numpy().copy())
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_renderer_itself_on_two_gloo_ranks(tmp_path):
    """The real ShardedRenderer (two contexts alternating, f32 and u8 output) over gloo, its context replaced by the CPU
    build of the kernel: every rank ends up with the single-rank image, frame after frame."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import raytracinginoneweekend_amd as rt
    from emul import Emul
    from oracle import pyoracle as po
    world, W, H = 2, 37, 23
    port = 29000 + (os.getpid() * 13 + 1234) % 2000
    mp.spawn(_sharded_worker, args=(world, W, H, port, str(tmp_path)), nprocs=world, join=True)
    e = Emul(rt.stock_scene())
    refs = [e.render(rt.stock_camera(W / H), W, H, 2, 10, seed=10 + i, threads=2)[0] for i in range(3)]
    for r in range(world):
        got = np.load(tmp_path / f"sh{r}.npy")
        for i in range(3):
            assert np.array_equal(got[i].view(np.uint32), refs[i].view(np.uint32)), (r, i)
        assert np.array_equal(np.load(tmp_path / f"sh8_{r}.npy"), po.quantise(refs[0], 2))
