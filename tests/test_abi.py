"""CPU tests of the C-ABI boundary: liboflk.so loads without a GPU, exports every
symbol include/oflk.h declares, and its host-only helpers behave.  No compute
entry point is called here."""
import ctypes
import re
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parents[1]


def declared_functions():
    text = (ROOT / "include" / "oflk.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(oflk_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_the_reference_surface():
    names = declared_functions()
    for must in ["oflk_compute_gradients", "oflk_from_gradients", "oflk_single_scale", "oflk_build_pyramid",
                 "oflk_warp", "oflk_upsample_flow", "oflk_pyramidal", "oflk_pyramidal_batch", "oflk_plan_create",
                 "oflk_plan_pyramidal", "oflk_last_error"]:
        assert must in names


def test_library_exports_every_declared_symbol():
    import _oflk

    L = _oflk.lib()
    for name in declared_functions():
        assert hasattr(L, name), f"{name} declared in include/oflk.h but not exported"
        assert name in _oflk.SIGNATURES, f"{name} has no ctypes signature in _oflk.py"
    assert set(_oflk.SIGNATURES) == set(declared_functions())


def test_version_and_device_count_never_fail():
    import _oflk

    assert _oflk.version().startswith("oflk ")
    assert _oflk.device_count() >= 0


@pytest.mark.parametrize("shape,levels", [((240, 320), 3), ((1080, 1920), 3), ((2160, 3840), 3), ((37, 53), 3),
                                          ((241, 321), 4), ((5, 7), 2)])
def test_level_dims_follow_int_truncation(oracle, shape, levels):
    import lucas_kanade_pyramidal as P

    got = P.pyramid_level_shapes(shape, levels)
    exp = []
    h, w = shape
    for _ in range(levels):
        exp.insert(0, (h, w))
        h, w = int(h * 0.5), int(w * 0.5)  # reference lucas_kanade_pyramidal.py:51-52
    assert got == exp == oracle.pyramid_dims(shape[0], shape[1], levels)


def test_invalid_arguments_raise_value_error():
    import lucas_kanade_pyramidal as P

    with pytest.raises(ValueError):
        P.pyramid_level_shapes((240, 320), 0)
    with pytest.raises(ValueError):
        P.pyramid_level_shapes((2, 2), 3)  # a level would be empty
    with pytest.raises(ValueError):
        P.pyramid_level_shapes((0, 5), 1)


def test_no_cpu_fallback_when_gpu_absent():
    """the product path must fail loudly, not route through any CPU code"""
    import _oflk
    import lucas_kanade_core as K

    if _oflk.device_count() > 0:
        pytest.skip("a GPU is visible here")
    a = np.zeros((16, 16), np.float32)
    with pytest.raises(_oflk.OflkError) as e:
        K.lucas_kanade_single_scale(a, a)
    assert e.value.code == _oflk.OFLK_ERR_NO_DEVICE and "no CPU path" in str(e.value)


def test_product_modules_never_touch_the_oracle():
    for f in (ROOT / "optical-flow-fpga_amd").rglob("*"):
        if f.suffix in (".py", ".hip", ".hpp", ".cpp", ".h") and f.is_file():
            text = f.read_text()
            assert "oflk_oracle" not in text and "liboflk_oracle" not in text, f


def test_product_translation_unit_carries_no_development_switches():
    """the shipped library reads no environment variable and carries no timing / ablation switches: diagnostic code
    (in-kernel stamps, sensitivity probes) only exists under -DOFLK_DIAG"""
    import re

    csrc = ROOT / "optical-flow-fpga_amd" / "csrc"
    hip = (csrc / "oflk.hip").read_text()
    hpp = (csrc / "oflk_kernels.hpp").read_text()
    assert "getenv" not in hip and "getenv" not in hpp
    for word in ("OFLK_ABLATE", "OFLK_LK16_ABL", "OFLK_LK16_TILED", "OFLK_LK16_COLS", "OFLK_TPB", "OFLK_X_"):
        assert word not in hip, word
    for word in ("OFLK_ABLATE", "OFLK_LK16_ABL", "k_lk16s", "k_lk16<"):
        assert word not in hpp, word
    # every stamp / probe site is behind the diagnostic build
    assert re.search(r"#if defined\(OFLK_STAMPS\) && !defined\(OFLK_DIAG\)\s*\n#error", hpp)
    assert "#if defined(OFLK_DIAG) && defined(OFLK_PROBE)" in hpp
    mk = (csrc / "Makefile").read_text()
    default_flags = [l for l in mk.splitlines() if l.startswith("FLAGS")][0]
    assert "OFLK_DIAG" not in default_flags
