"""Multi-GPU launch helper for the profiles/*.py scripts:
python -m torch.distributed.run --nnodes=1 --nproc-per-node G --master-addr 127.0.0.1 profiles/<script>.py ...
Every rank opens its GPU and joins the RCCL communicator of the library; rank 0 prints."""
import os


def open_dipper():
    import dipper_amd
    rank, world, local_rank = (int(os.environ.get(k, d)) for k, d in (("RANK", 0), ("WORLD_SIZE", 1), ("LOCAL_RANK", 0)))
    d = dipper_amd.Dipper(local_rank)
    dist = None
    if world > 1:
        import torch
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        uid = [d.comm_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(uid, src=0)
        d.comm_init(rank, world, uid[0])
    return d, rank, world, dist


def finish(dist):
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
