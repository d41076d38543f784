"""GPU: rmu_probe_mfma_rate (csrc/mfma_probe.hip) -- the sustained MFMA rates bench.py reports next to the nominal peaks."""
import ctypes

import pytest

pytestmark = pytest.mark.gpu


def _rate(lib, dtype, variant, millis=150):
    v = ctypes.c_double(0.0)
    rc = lib.rmu_probe_mfma_rate(dtype, variant, millis, ctypes.byref(v))
    assert rc == 0, lib.rmu_last_error()
    return v.value


def test_sustained_rates_are_plausible_and_ordered():
    import torch  # noqa: F401  (the runtime torch loaded is the one librmu shares)
    from ragmeup_amd import _native
    lib = _native.lib()
    assert lib.rmu_init(0) == 0
    f16 = _rate(lib, 0, 0)
    skel = _rate(lib, 0, 1)
    bf16 = _rate(lib, 1, 0)
    # profiles/r06_mfma_power.txt: 1.60-1.67 / 1.44 / 1.70-1.76 PFLOP/s; bounds wide enough for any box of the pool, tight enough to catch a
    # wrong flop count or a kernel the compiler emptied (the nominal peak is 2 500)
    for v in (f16, skel, bf16):
        assert 700.0 < v < 2600.0, (f16, skel, bf16)
    assert skel < f16 * 1.02, (f16, skel)          # operand delivery never makes the part faster
    # a second call (buffers freed and re-allocated) agrees within the run-to-run spread of a power-limited rate
    again = _rate(lib, 0, 0)
    assert abs(again - f16) / f16 < 0.15, (f16, again)


def test_bench_reports_the_sustained_block(monkeypatch):
    import torch  # noqa: F401
    import bench
    out = bench.measure_sustained(millis=80)
    assert out["unit"] == "TFLOP/s" and all(out[k] and out[k] > 0 for k in ("f16_mfma_only", "f16_lds_read_per_mfma_plus_dma_fill", "bf16_mfma_only"))
    r = bench.encoder_roofline(700.0)
    assert r["frac_of_sustained"] == round(700.0 / out["bf16_mfma_only"], 4) and r["peak"] == 2500.0
