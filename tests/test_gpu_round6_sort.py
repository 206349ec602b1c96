"""GPU parity for the round-6 form of the sort's workgroup levels (rs_sort_device.h): stages without branches, stop ranks in LDS,
the skipped last position slot, the straight-line single-wave finish.  Shapes are picked by where the array's chunks fall on the
waves: one, two, three and four position slots per thread, a last slot that only some waves (or only wave 0) hold, ranks in
RsMisc::hist (up to 1 024 positions) and in their own room behind the cut slots (above), built-in and shape-specialised kernels,
MaximizeCell (one std::sort) and UpperBound (S segmented sorts in one pass).  Bit-exact against the CPU oracle (real std::sort)."""
import pytest

from test_gpu_parity import _check_batch

pytestmark = pytest.mark.gpu

# (slices x users per slice, RBGs, PRBs per RBG, threads): records = slices * RBGs
SHAPES = [
    ([3] * 20, 22, 4, 512),   # 440 records: one slot, wave 7's chunk is behind the array's end
    ([3] * 25, 25, 4, 512),   # 625: two slots, the second on waves 0-1 only (ranks in RsMisc::hist)
    ([2] * 40, 25, 4, 512),   # 1 000: two slots, the second on waves 0-7 partly filled
    ([2] * 20, 64, 8, 512),   # 1 280: three slots, the third on waves 0-3 (ranks behind the cut slots)
    ([2] * 64, 25, 4, 512),   # 1 600: four slots, the fourth on wave 0 only
    ([2] * 33, 31, 4, 256),   # 1 023: four slots on four waves, the last position of the last chunk unused
    ([2] * 20, 64, 8, 1024),  # 1 280 on sixteen waves: two slots, the second on waves 0-3 (shape-specialised kernels only)
]


@pytest.mark.parametrize("jit", [False, True])
@pytest.mark.parametrize("shape", range(len(SHAPES)))
def test_maximize_cell_position_slots(rs, oracle, shape, jit):
    ues, R, G, threads = SHAPES[shape]
    if threads > 512 and not jit:
        pytest.skip("the built-in kernels take up to 512 threads per cell")
    _check_batch(rs, oracle, 9, ues, R, G, n_cells=3, n_ttis=45, threads=threads, jit=jit, seed=60 + shape)


@pytest.mark.parametrize("shape", [1, 3, 4])
def test_upper_bound_position_slots(rs, oracle, shape):
    ues, R, G, threads = SHAPES[shape]
    _check_batch(rs, oracle, 10, ues, R, G, n_cells=2, n_ttis=45, threads=threads, jit=True, seed=70 + shape)
