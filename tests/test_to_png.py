"""The display transform against the reference's python_utils/to_png.py functions (imported from the
reference tree in the build container only; they are pure numpy) and against fixed known answers."""
import importlib.util
import os

import numpy as np
import pytest

from pathtracing_amd import to_png

REF = "/root/reference/python_utils/to_png.py"


def test_known_answers():
    raw = np.array([[[0.0, 0.0, 0.0], [1.0, 0.5, 0.25], [3.0, 1.0, 0.0]]])
    tm = to_png.tone_map(raw)
    np.testing.assert_allclose(tm, [[[0, 0, 0], [0.5, 0.25, 0.125], [0.75, 0.25, 0.0]]])
    g = to_png.gamma_correction(np.array([0.0, 0.0031308, 0.5, 1.0]))
    np.testing.assert_allclose(g, [0.0, 0.0031308 * 12.92, 1.055 * 0.5 ** (1 / 2.4) - 0.055, 1.0], rtol=1e-15)
    assert to_png.denoising_enabled({"pipeline": [{"mode": "skip"}, {"mode": "median"}]})
    assert not to_png.denoising_enabled({"pipeline": [{"mode": "skip"}]}) and not to_png.denoising_enabled({})


@pytest.mark.skipif(not os.path.exists(REF), reason="reference tree not present")
def test_same_pixels_as_the_reference_functions(tmp_path):
    spec = importlib.util.spec_from_file_location("ref_to_png", REF)
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)
    rng = np.random.default_rng(3)
    raw = rng.gamma(0.7, 1.5, size=(12, 17, 3))
    path = str(tmp_path / "raw_pixel.dat")
    raw.tofile(path)
    ours = to_png.load_image_data(path, 17, 12)
    theirs = ref.load_image_data(path, 17, 12)
    assert np.array_equal(ours, theirs)
    assert np.array_equal(to_png.gamma_correction(ours), ref.gamma_correction(theirs))
    assert np.array_equal(to_png.display_image(raw, True), ref.gamma_correction(theirs))
