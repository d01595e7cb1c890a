"""CPU: bench.py's byte accounting equals the figures of SURVEY.md 8(d) / BASELINE.md 3, and the row sharding
the N>1 run uses covers the matrix on generator-block boundaries."""
import importlib.util
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
bench = importlib.util.module_from_spec(spec)
spec.loader.exec_module(bench)


def test_algorithmic_bytes_match_the_survey():
    assert bench.bytes_per_iter(65536, 8192, "f32") == 2_147_876_864           # cfg2
    assert bench.bytes_per_iter(2 ** 20 // 8, 16384, "f32") == 8_590_721_024   # cfg4 per GPU, 8-way
    assert bench.bytes_per_iter(2 ** 20 // 8, 16384, "bf16") == 4_295_753_728  # cfg5 per GPU
    # 70 % of the nominal 8 TB/s roofline at cfg2 is 383.6 us per iteration
    assert abs(bench.bytes_per_iter(65536, 8192, "f32") / (0.7 * 8e12) * 1e6 - 383.6) < 0.1
    assert bench.HBM_PEAK_GBPS == 8000.0


def test_workloads_are_the_baseline_configs():
    w = bench.WORKLOADS
    assert (w["cfg2"]["m"], w["cfg2"]["n"], w["cfg2"]["dtype"]) == (65536, 8192, "f32")
    assert (w["cfg4"]["m"], w["cfg4"]["n"], w["cfg4"]["dtype"]) == (2 ** 20, 16384, "f32")
    assert (w["cfg5"]["m"], w["cfg5"]["n"], w["cfg5"]["dtype"]) == (2 ** 20, 16384, "bf16")
    assert w["cfg5"]["a2"] == 10.0 and w["cfg2"]["a1_frac"] == 0.10


def test_shards_fall_on_generator_blocks():
    from fastoptsolver_amd.distributed import shard_rows
    for name in ("cfg2", "cfg4"):
        m = bench.WORKLOADS[name]["m"]
        for world in (1, 2, 4, 8):
            for r in range(world):
                lo, hi = shard_rows(m, world, r)
                assert lo % bench.BLOCK == 0 and hi % bench.BLOCK == 0


def test_upload_matrix_layouts_and_dtypes():
    """Host logic of the boundary: a host matrix reaches its device view in blocks of the SOURCE dtype / layout and is
    converted by the copy into place - for every layout the result equals the plain host cast (run here with a CPU view
    standing in for the device buffer; the GPU twin is tests/test_gpu_parity.py::test_prepare_from_host_arrays)."""
    import numpy as np
    import torch
    from fastoptsolver_amd import _core
    old = _core.UPLOAD_CHUNK_BYTES
    _core.UPLOAD_CHUNK_BYTES = 1000              # several ragged blocks
    try:
        A = np.random.default_rng(0).standard_normal((37, 29))
        F = np.asfortranarray(A)
        srcs = [torch.from_numpy(A), torch.from_numpy(F), torch.from_numpy(A.astype(np.float32)), torch.from_numpy(A)[::2, ::3],
                torch.from_numpy(F)[:, 3:20], torch.from_numpy(A.astype(np.float16)), torch.from_numpy((A * 10).astype(np.int64))]
        for src in srcs:
            for dt in (torch.float32, torch.bfloat16):
                out = torch.empty(src.shape, dtype=dt)
                assert _core.upload_matrix(src, out) is out and torch.equal(out, src.to(dt)), (src.stride(), src.dtype, dt)
                pad = torch.zeros(src.shape[0], src.shape[1] + 3, dtype=dt)
                _core.upload_matrix(src, pad[:, : src.shape[1]])
                assert torch.equal(pad[:, : src.shape[1]], src.to(dt)) and float(pad[:, src.shape[1]:].abs().sum()) == 0.0
        assert _core.upload_matrix(torch.empty(0, 5), torch.empty(0, 5)).shape == (0, 5)
    finally:
        _core.UPLOAD_CHUNK_BYTES = old
