"""Process-group helpers with the reference's names (lavila/utils/distributed.py:13-102).

One process per GPU; `init_distributed_mode` keeps backend 'nccl', which on PyTorch-ROCm IS RCCL
(xGMI inside a node). Host-side glue only.
"""
import os
import shutil

import torch
import torch.distributed as dist


def get_model(model):
    if isinstance(model, (torch.nn.DataParallel, torch.nn.parallel.DistributedDataParallel)):
        return model.module
    return model


def setup_for_distributed(is_master):
    """Mutes print() on non-master ranks unless force=True is passed."""
    import builtins
    builtin_print = builtins.print

    def print(*args, **kwargs):
        force = kwargs.pop('force', False)
        if is_master or force:
            builtin_print(*args, **kwargs)

    builtins.print = print


def is_dist_avail_and_initialized():
    return dist.is_available() and dist.is_initialized()


def get_world_size():
    return dist.get_world_size() if is_dist_avail_and_initialized() else 1


def get_rank():
    return dist.get_rank() if is_dist_avail_and_initialized() else 0


def is_main_process():
    return get_rank() == 0


def save_on_master(state, is_best, output_dir, is_epoch=True):
    if not is_main_process():
        return
    ckpt_path = f'{output_dir}/checkpoint.pt'
    if is_best:
        torch.save(state, f'{output_dir}/checkpoint_best.pt')
    if is_epoch:
        ep = state['epoch']
        tag = '{:04d}'.format(ep) if isinstance(ep, int) else '{:.4f}'.format(ep)
        torch.save(state, ckpt_path)
        shutil.copy(ckpt_path, f'{output_dir}/checkpoint_{tag}.pt')


def init_distributed_mode(args):
    if 'RANK' in os.environ and 'WORLD_SIZE' in os.environ:
        args.rank = int(os.environ['RANK'])
        args.world_size = int(os.environ['WORLD_SIZE'])
        args.gpu = int(os.environ['LOCAL_RANK'])
    elif 'SLURM_PROCID' in os.environ:
        args.rank = int(os.environ['SLURM_PROCID'])
        args.gpu = args.rank % torch.cuda.device_count()
    else:
        print('Not using distributed mode')
        args.distributed = False
        return
    args.distributed = True
    torch.cuda.set_device(args.gpu)
    args.dist_backend = 'nccl'          # RCCL on ROCm
    print('| distributed init (rank {}): {}'.format(args.rank, args.dist_url), flush=True)
    dist.init_process_group(backend=args.dist_backend, init_method=args.dist_url, world_size=args.world_size,
                            rank=args.rank)
    dist.barrier()
    setup_for_distributed(args.rank == 0)
