"""Work partitioning and film combination for one process per GPU (torch.distributed; the "nccl"
backend is RCCL on ROCm, "gloo" works on CPU tensors for tests).

The reference has one process and a mutex-guarded job vector (app/pathtracer.cpp:277-289,337-365).
Here every (pixel, sample) pair has its own random stream, so the image does not depend on how the
pairs are dealt out.  Two partitions are offered:

* sample slices   every rank renders all pixels for a contiguous range of sample indices; films are
                  combined by a weighted mean (one all-reduce per buffer);
* row bands       interleaved bands of image rows, band b belongs to rank b % world; a rank's film is
                  zero outside its bands, so the same weighted-mean all-reduce acts as the gather.
"""


def sample_slice(rank, world, total_samples):
    """(first_sample_index, count) of `rank`'s share of `total_samples` samples per pixel."""
    base, extra = divmod(int(total_samples), int(world))
    count = base + (1 if rank < extra else 0)
    first = rank * base + min(rank, extra)
    return first, count


def row_bands(rank, world, width, height, band_rows=8):
    """[(start_idx, number_of_pixels), ...]: the contiguous pixel runs of the row bands owned by `rank`."""
    jobs = []
    n_bands = (height + band_rows - 1) // band_rows
    for band in range(rank, n_bands, world):
        first_row = band * band_rows
        rows = min(band_rows, height - first_row)
        jobs.append((first_row * width, rows * width))
    return jobs


def band_job(rank, world, width, height, band_rows=8):
    """The row bands of `rank` (see row_bands) as one banded job for DeviceScene.render_bands / ptk_render_bands:
    (start_idx, number_of_pixels, band_pixels, band_stride)."""
    runs = row_bands(rank, world, width, height, band_rows)
    if not runs:
        return 0, 0, band_rows * width, world * band_rows * width
    return runs[0][0], sum(n for _, n in runs), band_rows * width, world * band_rows * width


def combine_films(image, position, normal, counters, group=None):
    """In-place weighted mean of per-rank films: each of `image`, `position`, `normal` is a running
    mean over `counters` samples per pixel ([H*W*3] / [H*W] tensors on any device).  Afterwards every
    rank holds the mean over all ranks' samples and the summed counters."""
    import torch
    import torch.distributed as dist
    weight = counters.to(image.dtype).repeat_interleave(3)
    for t in (image, position, normal):
        t.mul_(weight)
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    dist.all_reduce(counters, op=dist.ReduceOp.SUM, group=group)
    total = counters.to(image.dtype).repeat_interleave(3)
    safe = torch.where(total > 0, total, torch.ones_like(total))
    for t in (image, position, normal):
        t.div_(safe)
    return image, position, normal, counters
