"""N > 1 path on CPU: world_size-2 gloo processes, each rendering its spp share with the oracle, combined by
the same reduce_image() bench.py uses with RCCL."""
import importlib
import os
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
W, H, SPP, PATHS = 48, 27, 16, 1500


def _worker(rank, world, port, mode, out_path):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    pta = importlib.import_module("path-tracer_amd")
    import orc
    desc = pta.scenes.cornell_box()
    share = (pta.multigpu.weak_share if mode == "weak" else pta.multigpu.strong_share)(rank, world, W, H, SPP, PATHS)
    r = orc.OrcRender(orc.OrcScene(desc), W, H, desc["camera"], share["samples"], PATHS, slot_base=share["slotBase"],
                      ticket_begin=share["ticketBegin"], ticket_end=share["ticketEnd"])
    r.run(threads=2)
    img = torch.from_numpy(r.image().reshape(-1).copy())
    own = img.clone()
    pta.multigpu.reduce_image(img, dst=0)
    gathered = [torch.zeros_like(own) for _ in range(world)] if rank == 0 else None
    dist.gather(own, gathered, dst=0)
    if rank == 0:
        np.savez(out_path, total=img.numpy(), parts=np.stack([g.numpy() for g in gathered]))
    dist.barrier()
    dist.destroy_process_group()


def _run(mode, tmp_path, port):
    out = str(tmp_path / f"{mode}.npz")
    mp.spawn(_worker, args=(2, port, mode, out), nprocs=2, join=True)
    d = np.load(out)
    return d["total"].reshape(H, W, 4), d["parts"].reshape(2, H, W, 4)


def test_weak_split_two_ranks(tmp_path):
    total, parts = _run("weak", tmp_path, 29611)
    np.testing.assert_array_equal(total, parts[0] + parts[1])
    np.testing.assert_array_equal(parts[0][..., 3], np.full((H, W), SPP))       # each rank: its own full spp
    np.testing.assert_array_equal(total[..., 3], np.full((H, W), 2 * SPP))
    assert not np.array_equal(parts[0][..., :3], parts[1][..., :3])             # different RNG streams
    a, b = parts[0][..., 0] / SPP, parts[1][..., 0] / SPP
    assert abs(a.mean() - b.mean()) < 0.03 * a.mean()                           # same estimator


def test_strong_split_two_ranks(tmp_path):
    total, parts = _run("strong", tmp_path, 29612)
    np.testing.assert_array_equal(total[..., 3], np.full((H, W), SPP))          # every ticket rendered exactly once
    assert parts[0][..., 3].sum() == parts[1][..., 3].sum() == W * H * SPP // 2


def test_share_partitions():
    pta = importlib.import_module("path-tracer_amd")
    for world in (1, 2, 4, 8):
        for fn in (pta.multigpu.weak_share, pta.multigpu.strong_share):
            sh = [fn(r, world, 1920, 1080, 64, 1_000_000) for r in range(world)]
            assert sh[0]["ticketBegin"] == 0
            for a, b in zip(sh, sh[1:]):
                assert a["ticketEnd"] == b["ticketBegin"] and b["slotBase"] - a["slotBase"] == 1_000_000
        assert sh[-1]["ticketEnd"] == pta.multigpu.tickets_for(1920, 1080, 64)
