"""Work partitioning and film combination for one process per GPU (torch.distributed; the "nccl"
backend is RCCL on ROCm, "gloo" works on CPU tensors for tests).

The reference has one process and a mutex-guarded job vector (app/pathtracer.cpp:277-289,337-365).
Here every (pixel, sample) pair has its own random stream, so the image does not depend on how the
pairs are dealt out.  Two partitions are offered:

* sample slices   every rank renders all pixels for a contiguous range of sample indices; films are
                  combined by a weighted mean (one all-reduce per buffer);
* row bands       interleaved bands of image rows, band b belongs to rank b % world; every rank sends its own
                  bands (1 / world of the frame) and an all-gather assembles the frame on every rank.

FilmCombiner holds a rank's films and does either combination without allocating; combine_films is the
plain weighted-mean all-reduce on caller-owned tensors.
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


class FilmCombiner:
    """Films of one rank in one flat tensor, and their combination over the ranks without allocating anything.

    `flat` holds image | position | normal ([3 * n_pixels] each, running means) followed by one weight per pixel;
    `counters` ([n_pixels] int64) are the samples per pixel behind the means.  The three film buffers are views of
    `flat` (hand their data_ptr() to a Film), so a combine is one collective over one buffer:

    samples  weighted mean: the means are multiplied by their sample counts, one all-reduce sums films and counts
             of all ranks, the sums are divided by the summed counts;
    tiles    the frame is made of bands of `band_pixels` pixels, band b owned by rank b % world: every rank packs its
             own bands (1 / world of the frame), one all-gather brings every rank's bands to every rank, and the bands
             are put back in frame order.  Nothing is summed, so the pixels come out bit for bit as they were rendered.

    Every temporary (weights, pack and gather buffers) is allocated here, once, outside any timed region.
    """

    def __init__(self, n_pixels, device, world, mode="samples", band_pixels=0, dtype=None):
        import torch
        self.torch = torch
        self.n_pixels, self.world, self.mode = int(n_pixels), int(world), mode
        dtype = dtype or torch.float64
        if mode == "tiles":
            if band_pixels <= 0:
                raise ValueError("tiles: band_pixels must be positive")
            self.band_pixels = int(band_pixels)
            per_round = self.band_pixels * self.world
            self.rounds = (self.n_pixels + per_round - 1) // per_round
            n = self.rounds * per_round  # the buffers cover whole rounds of one band per rank; a Film uses the first n_pixels
        elif mode == "samples":
            n = self.n_pixels
        else:
            raise ValueError("mode must be 'samples' or 'tiles'")
        self.n = n
        self.flat = torch.zeros(10 * n, dtype=dtype, device=device)
        self.image, self.position, self.normal = self.flat[0:3 * n], self.flat[3 * n:6 * n], self.flat[6 * n:9 * n]
        self.weight = self.flat[9 * n:10 * n]
        self.counters = torch.zeros(n, dtype=torch.int64, device=device)
        self.films = self.flat[0:9 * n].view(3, n, 3)
        if mode == "tiles":
            # frame order [round][rank][pixel of the band] for each of the three films and the counters
            self.film_bands = self.flat[0:9 * n].view(3, self.rounds, self.world, self.band_pixels * 3)
            self.count_bands = self.counters.view(self.rounds, self.world, self.band_pixels)
            per_rank = 9 * n // self.world
            self.send = torch.zeros(per_rank, dtype=dtype, device=device)
            self.recv = torch.zeros(per_rank * self.world, dtype=dtype, device=device)
            self.send_counts = torch.zeros(n // self.world, dtype=torch.int64, device=device)
            self.recv_counts = torch.zeros(n, dtype=torch.int64, device=device)

    def wire_bytes_per_rank(self):
        """Bytes a rank contributes to one combine: the all-reduced buffer, or its own bands."""
        if self.mode == "samples":
            return self.flat.numel() * self.flat.element_size()
        return self.send.numel() * self.send.element_size() + self.send_counts.numel() * 8

    def combine(self, rank=0, group=None):
        """Afterwards every rank holds the film of all ranks' samples and the summed counters."""
        import torch.distributed as dist
        n = self.n
        if self.mode == "samples":
            self.weight.copy_(self.counters)
            self.films.mul_(self.weight.view(1, n, 1))
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=group)  # films and weights in one collective
            self.counters.copy_(self.weight)  # (sample counts are far below 2^53: exact)
            self.weight.clamp_(min=1.0)
            self.films.div_(self.weight.view(1, n, 1))
            return
        r, w, bp = self.rounds, self.world, self.band_pixels
        self.send.view(3, r, bp * 3).copy_(self.film_bands[:, :, rank, :])
        self.send_counts.view(r, bp).copy_(self.count_bands[:, rank, :])
        dist.all_gather_into_tensor(self.recv, self.send, group=group)
        dist.all_gather_into_tensor(self.recv_counts, self.send_counts, group=group)
        # recv: [rank][film][round][band] -> frame order [film][round][rank][band]
        self.film_bands.copy_(self.recv.view(w, 3, r, bp * 3).permute(1, 2, 0, 3))
        self.count_bands.copy_(self.recv_counts.view(w, r, bp).permute(1, 0, 2))


def combine_films(image, position, normal, counters, group=None):
    """In-place weighted mean of per-rank films: each of `image`, `position`, `normal` is a running
    mean over `counters` samples per pixel ([H*W*3] / [H*W] tensors on any device).  Afterwards every
    rank holds the mean over all ranks' samples and the summed counters.  (Allocates its temporaries; the
    benchmark uses FilmCombiner, which does not.)"""
    import torch
    import torch.distributed as dist
    n = counters.numel()
    weight = counters.to(image.dtype)
    for t in (image, position, normal):
        t.view(n, 3).mul_(weight.view(n, 1))
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    dist.all_reduce(counters, op=dist.ReduceOp.SUM, group=group)
    safe = counters.to(image.dtype).clamp_(min=1.0)
    for t in (image, position, normal):
        t.view(n, 3).div_(safe.view(n, 1))
    return image, position, normal, counters
