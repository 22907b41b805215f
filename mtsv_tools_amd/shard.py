"""Read sharding for multi-GPU runs: reads partition into contiguous blocks, one per rank; the index
is replicated; results are concatenated on the host (mode A).  No data-path collective:
torch.distributed is used only for the start barrier, the max-over-ranks timing and an optional
gather of per-rank result text."""


def shard_bounds(n_reads, rank, world):
    """[lo, hi) of the reads owned by `rank`: contiguous blocks whose sizes differ by at most one."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError("bad rank/world")
    base, rem = divmod(n_reads, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def gather_lines(local_text, dist=None):
    """Concatenate per-rank result text on rank 0 (order of lines is unspecified in the reference:
    vendor/cue/src/lib.rs:67-74).  Returns the joined text on rank 0, None elsewhere."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return local_text
    parts = [None] * dist.get_world_size() if dist.get_rank() == 0 else None
    dist.gather_object(local_text, parts, dst=0)
    return "".join(parts) if dist.get_rank() == 0 else None
