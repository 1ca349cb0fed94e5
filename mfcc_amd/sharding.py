"""Work distribution over the GPUs of one node (one process per GPU, torch.distributed).

Two modes, matching how the reference's work divides:
  * a file list shards by file: every rank extracts its own files, nothing is exchanged;
  * one very long recording shards by frame range: every rank recomputes the few halo frames the
    regression deltas need, and the per-column z-norm statistics (sum, sum of squares, count) are
    combined with one tiny exchange before the normalisation pass.
Nothing here computes features; the arithmetic stays in libmfcc_amd.so.
"""
import os
import re

import numpy as np


def shard_files(frame_counts, world_size):
    """Longest-first greedy assignment of files to ranks.  Returns a list of index lists."""
    order = sorted(range(len(frame_counts)), key=lambda i: (-int(frame_counts[i]), i))
    loads = [0] * world_size
    parts = [[] for _ in range(world_size)]
    for i in order:
        r = min(range(world_size), key=lambda k: (loads[k], k))
        parts[r].append(i)
        loads[r] += int(frame_counts[i])
    for p in parts:
        p.sort()
    return parts


def frame_ranges(total_frames, world_size):
    """Contiguous, near-equal frame ranges [begin, end) per rank."""
    return [(total_frames * r // world_size, total_frames * (r + 1) // world_size)
            for r in range(world_size)]


def combine_sums(local, dist=None, group=None):
    """Adds the ranks' statistics vectors in rank order (all_gather, then a fixed-order sum), so
    every rank holds bit-identical totals whatever the collective's internal order is.
    `local` is a 1-D torch tensor (float64 statistics or int64 channel sums)."""
    if dist is None or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return local.clone()
    import torch
    src = local.contiguous()
    if src.is_cuda and dist.get_backend(group) == "gloo":
        src = src.cpu()  # gloo moves host memory (CPU tests, one-device rehearsals); RCCL takes device tensors
    parts = [torch.empty_like(src) for _ in range(dist.get_world_size(group))]
    dist.all_gather(parts, src, group=group)
    total = parts[0].clone()
    for p in parts[1:]:
        total += p
    return total.to(local.device)


def extract_recording_shard(ma, ctx, pcm_shard_dev, channels, bytes_per_sample, total_samples,
                            frame_begin, frame_end, dist=None, chan_sums2=None):
    """Runs one rank's share of a long recording on its GPU.

    pcm_shard_dev: uint8 tensor holding exactly the instants the shard plan asks for
    (ma.shard_geometry gives sample_begin / sample_count).  Returns (features of the rank's own
    frames, plan).  With znormFlag the statistics are exchanged through `dist` (RCCL on GPUs).
    """
    plan = ma.Plan(ctx, channels, bytes_per_sample,
                   shard=(total_samples, frame_begin, frame_end, chan_sums2))
    out = plan.extract(pcm_shard_dev)
    if ctx.cfg.znormFlag:
        sums = plan.shard_column_sums(out)
        sums = combine_sums(sums, dist)
        plan.shard_normalize(out, sums)
    own = out[plan.owned_row_begin:plan.owned_row_begin + plan.owned_rows]
    return own, plan


def recording_channel_sums(ma, ctx, pcm_dev, channels, bytes_per_sample, first_instant, count,
                           dist=None):
    """Twice the integer sums of {L, R, mid, side} over the whole recording: each rank sums the
    instants [first_instant, first_instant + count) of its own buffer, then the ranks add up."""
    local = ma.channel_sums(ctx, pcm_dev, channels, bytes_per_sample, first_instant, count)
    return combine_sums(local, dist)


def split_file_list(ini_path, rank, world_size, out_dir, file_list=None):
    """This rank's share of a run: reads the `src<TAB>dst` list the ini names (main.c:173-177: up
    to the first blank line), gives every rank the files shard_files assigns it by size on disk
    (longest first; a missing file counts as empty and is reported by the rank that gets it), and
    writes `rank<r>.list` and `rank<r>.ini` (the original ini with its fileList line replaced and
    `[IO] numDevices = 1`: a rank is one process on one GPU, the caller's current device, so it must
    not start a device thread per visible GPU as a lone process does) into out_dir.
    file_list: the list path, if the caller already parsed the ini.
    Returns (path of the per-rank ini, indices of this rank's lines)."""
    text = open(ini_path, encoding="utf-8-sig").read()
    key = re.compile(r"^(\s*fileList\s*[=:]\s*)([^;#\n]*?)(\s*(?:[;#].*)?)$", re.M)
    m = key.search(text)
    if file_list is None:
        if not m:
            raise ValueError("%s names no fileList" % ini_path)
        file_list = m.group(2).strip()
    lines = []
    with open(file_list) as f:
        for line in f:
            if line == "\n":
                break
            lines.append(line if line.endswith("\n") else line + "\n")
    sizes = []
    for line in lines:
        src = line.split("\t")[0]
        try:
            sizes.append(os.path.getsize(src))
        except OSError:
            sizes.append(0)
    mine = shard_files(sizes, world_size)[rank]
    os.makedirs(out_dir, exist_ok=True)
    lst = os.path.join(out_dir, "rank%d.list" % rank)
    with open(lst, "w") as f:
        f.writelines(lines[i] for i in mine)
    ini = os.path.join(out_dir, "rank%d.ini" % rank)
    body = (key.sub(lambda mm: mm.group(1) + lst + mm.group(3), text, count=1) if m
            else text + "\n[IO]\nfileList = %s\n" % lst)
    # drop any numDevices line of the original and pin the rank to one device thread
    body = re.sub(r"^\s*numDevices\s*[=:].*\n?", "", body, flags=re.M)
    if not body.endswith("\n"):
        body += "\n"
    body += "[IO]\nnumDevices = 1\n"
    with open(ini, "w") as f:
        f.write(body)
    return ini, mine


def wapper_list_sharded(ma, ini_path, out_dir, dist=None):
    """The whole run (MFCCWapperList = main.c:98-300) over the GPUs of a node: every rank takes its
    share of the file list (split_file_list) and runs it on its own GPU (the process's current
    device; the rank's ini says numDevices = 1); nothing is exchanged on the data path.  Returns (failed files of this rank, failed files of all ranks)."""
    rank = dist.get_rank() if dist is not None and dist.is_initialized() else 0
    world = dist.get_world_size() if dist is not None and dist.is_initialized() else 1
    ini, _ = split_file_list(ini_path, rank, world, out_dir)
    failed = ma.wapper_list(ini)
    total = failed
    if world > 1:
        import torch
        t = torch.tensor([failed], dtype=torch.int64, device="cuda" if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t)
        total = int(t.item())
    return failed, total
