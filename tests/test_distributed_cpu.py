"""CPU: the N > 1 path.  (1) the tile→rank ownership map and the slot→pixel map partition
the frame; (2) a world_size-2 gloo run drives the real ShardedRenderer (both exchange modes)
with an oracle-backed stand-in for the GPU and must rebuild the unsharded frame bit for bit."""
import os
import subprocess
import sys

import numpy as np
import pytest

import cases

rt = cases.rt


def dist_mod():
    from importlib import import_module
    return import_module("opencl-raytracing_amd.distributed")


@pytest.mark.parametrize("w,h,world", [(1920, 1080, 8), (173, 99, 2), (64, 48, 3), (7, 5, 4), (3840, 2160, 8)])
def test_tile_owner_map_partitions_and_balances(w, h, world):
    d = dist_mod()
    own = d.tile_owner_map(w, h, world)
    assert own.shape == (h, w) and own.min() >= 0 and own.max() < world
    counts = np.bincount(own.reshape(-1), minlength=world)
    assert counts.sum() == w * h
    if w * h >= 64 * 64 * world:
        assert counts.max() - counts.min() <= 64 * (w // 8 + 2)      # within a tile row of each other
    tiles_x = (w + 7) // 8
    assert own[h - 1, w - 1] == (((h - 1) // 8) * tiles_x + (w - 1) // 8) % world


@pytest.mark.parametrize("w,h,world", [(93, 56, 2), (1920, 1080, 8), (64, 64, 3)])
def test_slot_pixels_partition(w, h, world):
    """The host restatement of the kernel's slot→pixel map: the ranks' valid slots cover every
    pixel exactly once and agree with the tile owner map."""
    d = dist_mod()
    seen = np.zeros((h, w), np.int32)
    own = d.tile_owner_map(w, h, world)
    for r in range(world):
        valid, y, x = d.slot_pixels(w, h, r, world)
        assert len(valid) == d.shard_slots(w, h, world)
        np.add.at(seen, (y[valid], x[valid]), 1)
        assert (own[y[valid], x[valid]] == r).all()
    assert (seen == 1).all()


def test_gloo_world2_exchange_rebuilds_frame(built):
    worker = os.path.join(cases.ROOT, "tests", "dist_worker.py")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", "29731", worker]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "EXCHANGE_OK gather" in out.stdout and "EXCHANGE_OK reduce" in out.stdout
