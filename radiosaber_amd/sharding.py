"""Multi-GPU sharding of independent cells.

Cells (= the (scheduler, seed) processes the reference launches with bash `&`,
NSDI23-radiosaber-experiments/exp-customization/run_backlogged.sh:6-14) share nothing, so rank r of W
owns the contiguous block of global cell ids [r*cells_per_rank, (r+1)*cells_per_rank) and no
collective runs during the TTI loop.  The only exchange is the final sum of the per-slice cumulative
byte counters (uint64[S] held as int64), an exact integer all-reduce (RCCL over xGMI on GPUs, gloo in
the CPU tests)."""
import numpy as np

SEED_BASE = 805290992  # reference seed.h commonSeed[0]


def cell_ids_for_rank(rank: int, world: int, cells_per_rank: int) -> np.ndarray:
    if not 0 <= rank < world:
        raise ValueError(f"rank {rank} outside 0..{world - 1}")
    return np.arange(cells_per_rank, dtype=np.uint64) + np.uint64(rank * cells_per_rank)


def seeds_for_cells(cell_ids: np.ndarray) -> np.ndarray:
    """srand() argument of every cell: a fixed function of the GLOBAL cell id.  Together with the synthetic CQI grids,
    which the device generator keys on the global cell id too (BatchScheduler.synthesize_cqi(first_cell=...)), a cell's
    whole trajectory is independent of how many ranks the job runs on (tests/test_gpu_round2.py checks it on the GPU)."""
    g = np.asarray(cell_ids, np.uint64)
    return ((g * np.uint64(2654435761) + np.uint64(SEED_BASE)) % np.uint64(2**31 - 1)).astype(np.uint32)


def first_cell_for_rank(rank: int, world: int, cells_per_rank: int) -> int:
    """Global id of the rank's first cell: the `first_cell` argument of BatchScheduler.synthesize_cqi (one base seed on
    every rank)."""
    return int(cell_ids_for_rank(rank, world, cells_per_rank)[0])


def all_reduce_slice_bytes(t, dist=None):
    """Sum the per-slice byte counters over all ranks in place (torch int64 tensor)."""
    if dist is not None and dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t


def slice_throughput_mbps(slice_bytes, seconds):
    """The reference's post-hoc metric (plot_throughput.py:26-56): bytes*8/1e6/T per slice."""
    return np.asarray(slice_bytes, np.float64) * 8 / 1e6 / seconds
