"""Work distribution over the GPUs of one node (one process per GPU, torch.distributed).

Two modes, matching how the reference's work divides:
  * a file list shards by file: every rank extracts its own files, nothing is exchanged;
  * one very long recording shards by frame range: every rank recomputes the few halo frames the
    regression deltas need, and the per-column z-norm statistics (sum, sum of squares, count) are
    combined with one tiny exchange before the normalisation pass.
Nothing here computes features; the arithmetic stays in libmfcc_amd.so.
"""
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
    parts = [torch.empty_like(local) for _ in range(dist.get_world_size(group))]
    dist.all_gather(parts, local.contiguous(), group=group)
    total = parts[0].clone()
    for p in parts[1:]:
        total += p
    return total


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
