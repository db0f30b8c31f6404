"""Coarse timing guards (GPU): not benchmarks -- they only catch a kernel path that has become several times slower
than its alternative (a scheduling change once made the two-window matrix-core convolutions 5x slower while every
parity test stayed green).  Margins are wide on purpose."""
import numpy as np
import pytest

import rl_mm_oracle as orc

pytestmark = pytest.mark.gpu


def kernel_ms(MK, conv, size=1536):
    from lib import _native
    rng = np.random.default_rng(0)
    pad = MK // 2
    image = rng.random((size, size, 3), dtype=np.float32) * 0.8 + 0.1
    u0 = np.ascontiguousarray(np.pad(image, ((pad, pad), (pad, pad), (0, 0)), mode="edge"))
    psf = np.full((MK, MK, 3), 1.0 / (MK * MK), np.float32)
    job = _native.RLJob(size, size, MK)
    try:
        job.upload(image, u0, psf)
        win = orc.default_window(size, size, MK)
        job.run(job.params(*win, 1e9, 2, 1e-3, 10000.0, True, stop_test=0, conv=conv))          # warm-up
        st = job.run(job.params(*win, 1e9, 4, 1e-3, 10000.0, True, stop_test=0, profile=1, conv=conv))
        names = _native.KERNEL_NAMES
        return {names[k]: st.ms_kernel[k] for k in range(len(names)) if st.launches[k]}
    finally:
        job.close()


@pytest.mark.parametrize("MK", [15, 31])
def test_matrix_core_kernels_are_not_slower_than_the_vector_kernels(MK):
    vec, mat = kernel_ms(MK, 1), kernel_ms(MK, 2)
    for k in ("synth_residual", "backproject"):
        assert mat[k] < 1.5 * vec[k], (MK, k, mat[k], vec[k])
    # A11 + A13: one fused kernel on the matrix-core path for MK <= 15 (ics_synth_gradk_mfma.hip), two kernels otherwise
    gk = lambda d: d["synth_gradk"] if "synth_gradk" in d else d["synth_residual"] + d["psf_gradient"]
    assert gk(mat) < 1.5 * gk(vec), (MK, mat, vec)


def test_auto_is_the_faster_choice_at_the_crossover_sizes():
    for MK in (19, 23):
        auto, vec, mat = kernel_ms(MK, 0), kernel_ms(MK, 1), kernel_ms(MK, 2)
        # A1 + A3: two kernels, or one unit per tile pair on the tiles (round 6; its per-outer window launch of A1 is then the only "synth_residual");
        # A11 + A13: one fused kernel where it exists
        a13 = lambda d: d["synth_backproject"] if "synth_backproject" in d else d["synth_residual"] + d["backproject"]
        total = lambda d: a13(d) + (d["synth_gradk"] if "synth_gradk" in d else d["synth_residual"] + d["psf_gradient"])
        assert total(auto) < 1.25 * min(total(vec), total(mat)), (MK, auto, vec, mat)
