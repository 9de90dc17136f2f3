"""Multi-GPU sharding of a spectrum batch: one process per GPU, index replicated, spectra partitioned.

Each MS2 spectrum is scored independently against a read-only index (sage-cli runner.rs:311-325), so the
path shards with NO data-path collective.  The only communication is the host-side gather of the (small)
PSM records in input order — the reference's `collect()` preserves input order (runner.rs:325).
"""
import numpy as np


def plan_shards(peak_off: np.ndarray, world: int):
    """Contiguous, work-balanced shards: split the spectrum list where the cumulative peak count crosses
    k/world of the total (work per spectrum ~ number of peaks x queries).  Returns [(begin, end)] * world."""
    n = len(peak_off) - 1
    if world <= 1 or n == 0:
        return [(0, n)] + [(n, n)] * (max(world, 1) - 1)
    cum = peak_off[1:].astype(np.float64) + np.arange(1, n + 1)  # +1 per spectrum: empty spectra still cost a launch slot
    total = cum[-1]
    cuts = [0]
    for k in range(1, world):
        cuts.append(int(np.searchsorted(cum, total * k / world, side="left")))
    cuts.append(n)
    cuts = np.maximum.accumulate(np.array(cuts))
    return [(int(cuts[i]), int(cuts[i + 1])) for i in range(world)]


def shard_indices(peak_off: np.ndarray, rank: int, world: int) -> np.ndarray:
    b, e = plan_shards(peak_off, world)[rank]
    return np.arange(b, e)


def gather_features(feats: np.ndarray, counts: np.ndarray, begin: int, group=None):
    """Gather per-rank (features[n_r, report], counts[n_r]) to every rank, concatenated in input order, with
    spec_index rebased to the global batch.  Uses torch.distributed (RCCL on GPUs, gloo on CPU); host-side only."""
    import torch.distributed as dist
    f = feats.copy()
    valid = np.arange(f.shape[1])[None, :] < counts[:, None]  # slots beyond counts[i] stay zeroed
    f["spec_index"][valid] += np.uint32(begin)
    world = dist.get_world_size(group)
    parts = [None] * world
    dist.all_gather_object(parts, (begin, f, counts), group=group)
    parts.sort(key=lambda p: p[0])
    return np.concatenate([p[1] for p in parts], axis=0), np.concatenate([p[2] for p in parts], axis=0)
