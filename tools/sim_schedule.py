"""Work-distribution study (CPU): predicts the render kernel's lane utilisation for a frame from per-sample path
lengths recorded by the CPU emulation build, for any GPU count and run schedule.

Model = the kernel's scheduling, pass by pass: 6 workgroups x 4 waves per CU x 256 CUs persistent waves per GPU; a
wave draws packets (tile x run) from the queue in order; free lanes take the packet's units; a lane traces its
unit's samples back to back (one segment per wave pass).  Utilisation = segments / (64 x wave passes)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import raytracinginoneweekend_amd as rt
from emul import Emul

CACHE = "/tmp/seg_300x200x500.npy"


def path_lengths():
    if os.path.exists(CACHE):
        return np.load(CACHE)
    z = np.load(os.path.join(ROOT, "tests/golden/scene_stock.npz"))
    e = Emul({k: z[k] for k in z.files})
    W, H, spp = 300, 200, 500
    seg = np.zeros((H * W, spp), np.uint8)
    t0 = time.time()
    e.render_tiles(rt.stock_camera(W / H), rt.make_params(W, H, spp, 50, 1), threads=8, seg_counts=seg)
    print("recorded %d samples in %.0fs, mean %.3f" % (seg.size, time.time() - t0, seg.mean()))
    np.save(CACHE, seg)
    return seg


def simulate(seg, W, H, spp, n_gpus, runs, waves_per_gpu=6144, up=4, seed=0, dequeue=2):
    """Time-stepped (one wave pass per step, all waves in lockstep = equal speed) model of rank 0's kernel.
    seg: [h*w, S] path lengths at low resolution; the frame is W x H = up x the low resolution."""
    rng = np.random.default_rng(seed)
    h, w = H // up, W // up
    S = seg.shape[1]
    tiles_x, tiles_y = W // 8, H // 8
    local_tiles = np.arange(0, tiles_x * tiles_y, n_gpus)
    nlt = len(local_tiles)
    tx, ty = local_tiles % tiles_x, local_tiles // tiles_x
    kx, ky = np.arange(64) % 8, np.arange(64) // 8
    lowpix = (((ty[:, None] * 8 + ky[None]) // up) * w + (tx[:, None] * 8 + kx[None]) // up).reshape(-1)   # [nlt*64]
    n_runs = len(runs)
    cost = np.zeros((n_runs, nlt * 64), np.int32)             # segments of unit (run, tile, k)
    for r, L in enumerate(runs):
        cols = rng.integers(0, S, (nlt * 64, int(L)))
        cost[r] = seg[lowpix[:, None], cols].sum(1)
    cost = cost.reshape(n_runs * nlt, 64)                     # packet q = run * nlt + tile  (run-major queue)
    n_packets = n_runs * nlt
    nw = min(waves_per_gpu, n_packets)
    rem = np.zeros((nw, 64), np.int32)
    pk = np.zeros(nw, np.int64)
    pos = np.full(nw, 64, np.int64)
    pend = np.zeros(nw, np.int64)                              # packets already drawn but not started (batched dequeue)
    next_packet = 0
    passes = np.zeros(nw, np.int64)
    busy_lane_passes = 0
    lane = np.arange(64)[None, :]
    while True:
        free = rem == 0
        need = free.sum(1)
        if need.any() and (next_packet < n_packets or (pos < 64).any() or (pend > 0).any()):
            rank = np.cumsum(free, 1) - 1                         # rank of each free lane among the wave's free lanes
            avail = 64 - pos
            # part 1: from the current packet
            t1 = free & (rank < avail[:, None])
            k1 = pos[:, None] + rank
            wi, li = np.nonzero(t1)
            rem[wi, li] = cost[pk[wi], k1[wi, li]]
            took1 = np.minimum(need, avail)
            pos = pos + took1
            # part 2: waves that still need units draw a new packet, in wave order
            more = need - took1
            wants = (more > 0)
            # waves with a pending (already drawn) packet start it; the others draw `dequeue` packets, in wave order
            have = wants & (pend > 0)
            draw = wants & ~have & (next_packet < n_packets)
            ids = np.where(have, pk + 1, 0)
            nd = int(draw.sum())
            if nd:
                first = next_packet + (np.cumsum(draw) - 1) * dequeue
                ok_draw = draw & (first < n_packets)
                got = np.minimum(dequeue, n_packets - first)
                ids = np.where(ok_draw, first, ids)
                pend = np.where(ok_draw, got, pend)
                next_packet = min(n_packets, next_packet + nd * dequeue)
                have = have | ok_draw
            ok = have
            if ok.any():
                pend = np.where(ok, pend - 1, pend)
                pk = np.where(ok, ids, pk)
                pos = np.where(ok, 0, pos)
                t2 = free & ~t1 & ok[:, None]
                k2 = rank - avail[:, None]
                wi, li = np.nonzero(t2)
                rem[wi, li] = cost[pk[wi], k2[wi, li]]
                pos = pos + np.where(ok, more, 0)
        act = rem > 0
        busy = act.any(1)
        if not busy.any():
            break
        # zero-cost units (cannot happen: every sample traces >= 1 segment) would loop forever; guard
        passes += busy
        busy_lane_passes += int(act.sum())
        rem = np.where(act, rem - 1, 0)
    util = busy_lane_passes / (64.0 * passes.sum())
    return util, int(passes.max()), float(passes.mean()), busy_lane_passes


if __name__ == "__main__":
    seg = path_lengths()
    W, H, spp = 1200, 800, 500
    scheds = {"fixed runs of 8 (round-1 first kernel)": [8] * 62 + [4],
              "rtk_run_schedule default (decreasing)": [int(x) for x in rt.run_schedule(spp, W, H)[:, 1]]}
    for n in (1, 2, 4, 8):
        for name, runs in scheds.items():
            assert sum(runs) == spp, name
            t0 = time.time()
            u, mx, mean, segs = simulate(seg, W, H, spp, n, np.array(runs))
            print("GPUs %d  %-40s lane utilisation %.4f  wave passes max %d mean %.0f  kernel efficiency (segments / (waves x 64 x max passes)) %.4f  [%.0fs]"
                  % (n, name, u, mx, mean, segs / (6144 * 64.0 * mx), time.time() - t0), flush=True)
