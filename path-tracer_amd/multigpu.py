"""Multi-GPU host logic: one process per GPU, samples-per-pixel split, one sum-reduce.

The path shards without any data-path collective: every GPU holds a full scene replica and a private
W*H*vec4f accumulation buffer, works on its own range of pixel tickets (logic.js:155-184) with its own
RNG streams (global slot index = rank * NUM_PATHS + slot, logic.js:77), and the buffers -- sums of radiance
plus sample counts, logic.js:146-149 -- are added once at the end (RCCL reduce over xGMI; gloo on CPU)."""

SAMPLES_PER_TICKET = 8   # logic.js:6


def tickets_for(width, height, spp):
    return (width * height * spp + SAMPLES_PER_TICKET - 1) // SAMPLES_PER_TICKET     # logic.js:177


def weak_share(rank, world, width, height, spp_per_gpu, num_paths, pools=1):
    """Every GPU renders spp_per_gpu; the node accumulates world * spp_per_gpu samples per pixel.
    pools: path pools per GPU (host.PooledPathTracer), num_paths slots each."""
    t = tickets_for(width, height, spp_per_gpu)
    return {"samples": spp_per_gpu * world, "numPaths": num_paths, "slotBase": rank * num_paths * pools, "pools": pools,
            "ticketBegin": rank * t, "ticketEnd": (rank + 1) * t}


def strong_share(rank, world, width, height, spp_total, num_paths, pools=1):
    """A fixed spp_total is split evenly: GPU g takes tickets [g*T/G, (g+1)*T/G)."""
    t = tickets_for(width, height, spp_total)
    return {"samples": spp_total, "numPaths": num_paths, "slotBase": rank * num_paths * pools, "pools": pools,
            "ticketBegin": rank * t // world, "ticketEnd": (rank + 1) * t // world}


class LibComm:
    """The library's own reduce (pt_comm_*, include/pt_amd.h): RCCL through libpt_amd.so, no torch.  devices: one host.Device per
    GPU of this process, rank i = devices[i] (pt_comm_create); or one Device plus (id128, rank, world) for the process-per-GPU
    form, where rank 0 makes the id with LibComm.unique_id() and the host hands it to the other ranks."""

    def __init__(self, devices, id128=None, rank=0, world=1):
        import ctypes as C
        from . import capi
        self.lib, self.h = devices[0].lib, C.c_void_p()
        if id128 is None:
            arr = (C.c_void_p * len(devices))(*[d.ctx for d in devices])
            rc = self.lib.pt_comm_create(arr, len(devices), C.byref(self.h))
        else:
            rc = self.lib.pt_comm_create_rank(devices[0].ctx, id128, rank, world, C.byref(self.h))
        capi.check(self.lib, devices[0].ctx, rc, "pt_comm_create")
        self.dev0 = devices[0]

    @staticmethod
    def unique_id():
        import ctypes as C
        from . import capi
        buf = C.create_string_buffer(128)
        capi.check(capi.load(), None, capi.load().pt_comm_unique_id(buf), "pt_comm_unique_id")
        return buf

    def reduce_image(self, root=0):
        from . import capi
        capi.check(self.lib, self.dev0.ctx, self.lib.pt_comm_reduce_image(self.h, int(root)), "pt_comm_reduce_image")

    def destroy(self):
        if self.h:
            self.lib.pt_comm_destroy(self.h)
            self.h = None


def reduce_image(image_tensor, dst=0):
    """Sum the per-rank accumulation buffers onto rank dst (backend nccl = RCCL on ROCm; gloo in CPU tests)."""
    import torch.distributed as dist
    if dist.is_initialized() and dist.get_world_size() > 1:
        if dist.get_backend() == "gloo" and image_tensor.is_cuda:      # rehearsal on a box with fewer GPUs than ranks
            host = image_tensor.cpu()
            dist.reduce(host, dst=dst, op=dist.ReduceOp.SUM)
            image_tensor.copy_(host)
        else:
            dist.reduce(image_tensor, dst=dst, op=dist.ReduceOp.SUM)
    return image_tensor
