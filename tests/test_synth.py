"""The torch generators used by bench.py / the GPU tests produce exactly the oracle's rasters (checked on CPU)."""
import numpy as np
import pytest


@pytest.mark.parametrize("case", [(64, 48, 3, 0, "NOISY3", 1), (33, 17, 8, 2, "LANDSAT16", 3), (40, 40, 1, 5, "DEM", 4),
                                  (40, 40, 1, 7, "DEM", 4), (16, 16, 2, 6, "RANDOM", 9), (48, 32, 1, 5, "TERRACE", 4),
                                  (20, 20, 3, 0, "GRAD", 0), (24, 24, 1, 3, "DEM", 4), (16, 16, 3, 1, "NOISY3", 7),
                                  (32, 32, 1, 5, "FEW", 4), (32, 32, 1, 7, "FEW", 4), (32, 32, 1, 0, "FEW", 4), (32, 32, 1, 2, "PALETTE", 7),
                                  (32, 32, 1, 6, "PALETTE", 5), (32, 32, 3, 0, "PALETTE", 3)])
def test_synth_equals_oracle_generator(oracle, case):
    from qb3_amd import synth
    w, h, b, dt, gen, seed = case
    a = synth.generate(w, h, b, dt, gen, seed, device="cpu", rows_per_chunk=7).numpy()
    r = oracle.generate(w, h, b, dt, gen, seed)
    assert np.array_equal(a.view(np.uint8), r.view(np.uint8))
