"""Multi-GPU layer: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI on the
GPU node, "gloo" in the CPU tests).  The reference has no distributed code at all;
what is sharded here is what the path offers:

  * compare matrix: independent pairs.  Rows are sharded in contiguous blocks; ONE all-gather of
    the signatures gives every rank all columns; each rank computes its row block; nothing else is
    exchanged (the row blocks stay on their rank).
  * sketching: independent records.  Every rank sketches its shard of the records with no
    collective at all; merge_sketch_across_ranks() is the optional final union.

The compute is injected (`compute_block`) so that the CPU tests can drive the same sharding and
collective code with the oracle; the product default is the HIP path and raises without a GPU."""
import numpy as np


def shard_range(n_total, world, rank):
    """Contiguous, equal-size (ceil) row blocks; the last blocks may be short or empty."""
    per = (n_total + world - 1) // world
    lo = min(n_total, rank * per)
    hi = min(n_total, lo + per)
    return lo, hi, per


def _hip_compute_block(rows_t, n_rows, cols_t, n_cols, num, want):
    from . import matrix
    width = rows_t.shape[1]
    row_off = np.arange(n_rows + 1, dtype=np.uint64) * np.uint64(width)
    col_off = np.arange(n_cols + 1, dtype=np.uint64) * np.uint64(width)
    out = matrix.compare_block_dev(rows_t, row_off, cols_t, col_off, num, want=want)
    return {k: v[:n_rows, :n_cols] for k, v in out.items()}


def compare_matrix_sharded(local_sigs, n_total, num, want=("jaccard",), compute_block=None, group=None):
    """local_sigs: (per, width) int64 tensor holding this rank's rows (rows beyond its share are
    padding), every rank with the same `per` = ceil(n_total / world).  Returns this rank's row
    block: dict name -> (n_local, n_total) tensor.  One all_gather_into_tensor, no other collective."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    lo, hi, per = shard_range(n_total, world, rank)
    assert local_sigs.shape[0] == per, "every rank passes ceil(n_total/world) rows (pad the last block)"
    if world > 1:
        allsigs = torch.empty((world * per, local_sigs.shape[1]), dtype=local_sigs.dtype, device=local_sigs.device)
        dist.all_gather_into_tensor(allsigs, local_sigs.contiguous(), group=group)
    else:
        allsigs = local_sigs
    fn = compute_block or _hip_compute_block
    # the row block as a VIEW of the gathered set: the block compare then sees that its rows are a
    # slice of its columns and encodes the columns only
    rows = allsigs[lo:lo + per] if world > 1 else local_sigs
    return fn(rows, hi - lo, allsigs, n_total, num, want)


def shard_records(n_records, world, rank):
    """Records [lo, hi) sketched by `rank` (no collective involved)."""
    lo, hi, _ = shard_range(n_records, world, rank)
    return lo, hi


def merge_sketch_across_ranks(mins, abunds=None, group=None):
    """Union of the ranks' partial sketches (ascending distinct uint64 `mins`, optional counts):
    all-gather of the lengths, then of the padded arrays; returns the concatenated per-rank lists
    [(mins_r, abunds_r)] on every rank for the caller to fold with KmerMinHash.merge (exact for
    scaled sketches: abundances add; for num sketches the mins are exact)."""
    import torch
    import torch.distributed as dist
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return [(mins, abunds)]
    world = dist.get_world_size(group)
    dev = mins.device
    n = torch.tensor([mins.numel()], dtype=torch.int64, device=dev)
    sizes = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(sizes, n, group=group)
    sizes = [int(s.item()) for s in sizes]
    cap = max(max(sizes), 1)

    def gather(t):
        pad = torch.zeros(cap, dtype=torch.int64, device=dev)
        pad[: t.numel()] = t
        out = torch.empty(world * cap, dtype=torch.int64, device=dev)
        dist.all_gather_into_tensor(out, pad, group=group)
        return [out[r * cap: r * cap + sizes[r]] for r in range(world)]

    gm = gather(mins)
    ga = gather(abunds) if abunds is not None else [None] * world
    return list(zip(gm, ga))
