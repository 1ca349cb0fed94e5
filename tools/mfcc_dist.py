#!/usr/bin/env python3
"""The `mfcc <config.ini>` run over all GPUs of a node, files sharded by rank:

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \\
        tools/mfcc_dist.py config.ini [scratch_dir]

Each rank runs MFCCWapperList on its share of the list on GPU LOCAL_RANK; rank 0 prints the number
of failed files.  scratch_dir (default: next to the ini) receives the per-rank lists."""
import os
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, ROOT)


def main():
    import torch
    import torch.distributed as dist
    import mfcc_amd as ma
    from mfcc_amd import sharding
    ini = sys.argv[1]
    scratch = sys.argv[2] if len(sys.argv) > 2 else os.path.join(os.path.dirname(os.path.abspath(ini)), ".mfcc_ranks")
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("mfcc_dist.py needs an MI355X: no HIP device is visible")
    torch.cuda.set_device(local_rank)
    if world > 1:
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    failed, total = sharding.wapper_list_sharded(ma, ini, scratch, dist if world > 1 else None)
    if int(os.environ.get("RANK", "0")) == 0:
        print("%d file(s) failed" % total)
    if world > 1:
        dist.destroy_process_group()
    sys.exit(1 if total else 0)


if __name__ == "__main__":
    main()
