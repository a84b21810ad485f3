"""Pin the CPU oracle (oracle/w4a16.py) against fixtures produced by the reference's own Python
(tests/golden/gen_golden.py).  CPU only."""
import os

import numpy as np
import pytest

import oracle
from conftest import golden_files, load_golden, rel_err

EXACT = golden_files("exact_")
assert EXACT, "golden fixtures missing"


def _logical(g):
    """logical (iw[K,N], s[K/G,N], z[K/G,N]) from the fixture's from_linear inputs."""
    G = int(g["G"])
    iw = oracle.quantize_intweight(g["weight"], g["scales_nk"], g["zeros_nk"], G)
    s = np.ascontiguousarray(g["scales_nk"].T).astype(np.float16)
    z = np.ascontiguousarray(g["zeros_nk"].T).astype(np.uint8)
    return iw, s, z, G


@pytest.mark.parametrize("path", EXACT, ids=[os.path.basename(p)[:-4] for p in EXACT])
def test_pack_cuda_order_matches_reference_packer(path):
    g = load_golden(path)
    iw, s, z, G = _logical(g)
    assert np.array_equal(iw, g["intweight_nk"].T.astype(np.int32))     # exact layer: rounding recovers iw
    qw, qs, qz = oracle.pack_cuda_order(iw, s, z)
    assert qw.shape == g["ref_qweight"].shape and qw.dtype == np.int32
    assert np.array_equal(qw, g["ref_qweight"])
    assert np.array_equal(qs.view(np.uint16), g["ref_qscales"].view(np.uint16))
    assert np.array_equal(qz, g["ref_qzeros"])


@pytest.mark.parametrize("path", EXACT, ids=[os.path.basename(p)[:-4] for p in EXACT])
def test_unpack_cuda_order_inverts_reference_pack(path):
    g = load_golden(path)
    iw, s, z, G = _logical(g)
    iw2, s2, z2 = oracle.unpack_cuda_order(g["ref_qweight"], g["ref_qscales"], g["ref_qzeros"])
    assert np.array_equal(iw2, iw) and np.array_equal(z2, z)
    assert np.array_equal(s2.view(np.uint16), s.view(np.uint16))


@pytest.mark.parametrize("path", EXACT, ids=[os.path.basename(p)[:-4] for p in EXACT])
def test_dequant_bit_exact_and_gemm_close(path):
    g = load_golden(path)
    iw, s, z, G = _logical(g)
    wdeq = oracle.dequantize(iw, s, z, G)
    assert np.array_equal(wdeq.view(np.uint16), g["ref_wdeq"].view(np.uint16))   # bit exact
    y = oracle.w4a16_forward(g["x"], iw, s, z, G)
    assert y.dtype == np.float16 and y.shape == g["ref_y"].shape
    assert rel_err(y, g["ref_y"]) <= 2e-3        # both fp32-accumulate; only summation order differs
    # the CUDA kernel's own numerics (fp16 split-K partials) sit inside the 1e-2 budget too
    for sk in (1, 2, 8):
        if (iw.shape[0] // 32) % sk == 0:
            assert rel_err(oracle.gemm_splitk_fp16_partials(g["x"], wdeq, sk), g["ref_y"]) <= 1e-2


@pytest.mark.parametrize("path", [p for p in EXACT if "k64" not in p], ids=lambda p: os.path.basename(p)[:-4])
def test_mi355x_order_roundtrip(path):
    g = load_golden(path)
    iw, s, z, G = _logical(g)
    qw, qs, qz = oracle.pack_mi355x(iw, s, z)
    assert qw.shape == g["ref_qweight"].shape and qs.shape == g["ref_qscales"].shape and qz.shape == g["ref_qzeros"].shape
    iw2, s2, z2 = oracle.unpack_mi355x(qw, qs, qz)
    assert np.array_equal(iw2, iw) and np.array_equal(z2, z) and np.array_equal(s2.view(np.uint16), s.view(np.uint16))
    # same multiset of nibbles as the reference pack: it is a permutation, not a re-encoding
    a = np.sort(((qw.view(np.uint32).ravel()[:, None] >> (4 * np.arange(8, dtype=np.uint32))) & 15).ravel())
    b = np.sort(((g["ref_qweight"].view(np.uint32).ravel()[:, None] >> (4 * np.arange(8, dtype=np.uint32))) & 15).ravel())
    assert np.array_equal(a, b)


def test_quantize_rounding_case():
    (path,) = golden_files("quant_")
    g = load_golden(path)
    iw, s, z, G = _logical(g)
    assert iw.min() >= 0 and iw.max() <= 15
    qw, qs, qz = oracle.pack_cuda_order(iw, s, z)
    assert np.array_equal(qw, g["ref_qweight"]) and np.array_equal(qz, g["ref_qzeros"])
    assert np.array_equal(qs.view(np.uint16), g["ref_qscales"].view(np.uint16))
    wdeq = oracle.dequantize(iw, s, z, G)
    assert np.array_equal(wdeq.view(np.uint16), g["ref_wdeq"].view(np.uint16))
    assert rel_err(oracle.w4a16_forward(g["x"], iw, s, z, G), g["ref_y"]) <= 2e-3


def test_quick_cat_matches_reference():
    (path,) = golden_files("quick_cat_")
    g = load_golden(path)
    for opt in ("qweight", "qzeros", "scales"):
        got = oracle.quick_cat_cuda_order([g[f"in{j}_{opt}"] for j in range(3)], opt)
        assert np.array_equal(got.view(np.uint8), g["cat_" + opt].view(np.uint8))
        assert np.array_equal(got.view(np.uint8), g["full_" + opt].view(np.uint8))
    with pytest.raises(ValueError):
        oracle.quick_cat_cuda_order([g["in0_qweight"], g["in1_qweight"][:, :64]], "qweight")


def test_algorithmic_counts_match_survey():
    assert oracle.algorithmic_bytes(1, 4096, 4096, 128) == 8_732_672
    assert oracle.algorithmic_bytes(8, 4096, 4096, 128) == 8_847_360
    assert oracle.algorithmic_bytes(64, 4096, 4096, 128) == 9_764_864
    assert oracle.algorithmic_bytes(512, 4096, 4096, 128) == 17_104_896
    assert oracle.algorithmic_flops(512, 4096, 4096) == 17_179_869_184


@pytest.mark.parametrize("path", EXACT, ids=[os.path.basename(p)[:-4] for p in EXACT])
def test_cpu_reference_path_restatement(path):
    """oracle/cpu_path.py (the timed CPU baseline) against the reference's own GEMM-format pack and outputs."""
    import torch
    from oracle import cpu_path
    g = load_golden(path)
    iw, s, z, G = _logical(g)
    qw, qz = cpu_path.pack_gemm_format(iw, z)
    assert np.array_equal(qw, g["ref_gemm_qweight"]) and np.array_equal(qz, g["ref_gemm_qzeros"])
    assert np.array_equal(s.view(np.uint16), g["ref_gemm_scales"].view(np.uint16))
    w = cpu_path.dequantize_gemm(torch.from_numpy(qw), torch.from_numpy(qz), torch.from_numpy(s), G)
    assert np.array_equal(w.numpy().view(np.uint16), g["ref_wdeq"].view(np.uint16))
    y = cpu_path.forward(torch.from_numpy(g["x"]), torch.from_numpy(qw), torch.from_numpy(qz), torch.from_numpy(s), G)
    assert rel_err(y.numpy(), g["ref_y"]) <= 2e-3      # same ops; host BLAS blocking may reorder the sums


# ------------------------------------------------------------------------------------------------
# BASELINE size: K = N = 4096, g = 128 (fixture = seed + SHA-256 of the reference packer's output + sampled values)
# ------------------------------------------------------------------------------------------------
def test_baseline_size_pin_packer_dequant_and_forward():
    import hashlib
    from conftest import exact_layer
    g = load_golden(golden_files("pin_")[0])
    K, N, G = int(g["K"]), int(g["N"]), int(g["G"])
    iw, s, z = exact_layer(K, N, G, int(g["seed"]))
    sha = lambda a: hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()
    qw, qs, qz = oracle.pack_cuda_order(iw, s, z)                   # the oracle's packer == the reference packer, 4096^2
    assert (sha(qw), sha(qs), sha(qz)) == (str(g["sha_qweight"]), str(g["sha_qscales"]), str(g["sha_qzeros"]))
    iw2, s2, z2 = oracle.unpack_cuda_order(qw, qs, qz)
    assert np.array_equal(iw2, iw) and np.array_equal(z2, z)
    wdeq = oracle.dequantize(iw, s, z, G)                           # sampled weights: bit exact against dequantize_gemm
    assert np.array_equal(wdeq[g["w_k"], g["w_n"]].view(np.uint16), g["w_ref"].view(np.uint16))
    y = oracle.w4a16_forward(g["x"], iw, s, z, G)                   # sampled outputs of the reference CPU path
    scale = float(np.abs(g["y_ref"].astype(np.float32)).max())
    assert float(np.abs(y[g["y_rows"], g["y_cols"]].astype(np.float32) - g["y_ref"].astype(np.float32)).max()) <= 2e-3 * scale
    cs = np.abs(y.astype(np.float32)).sum(0)
    assert float(np.abs(cs - g["col_abs_sum"]).max()) <= 2e-3 * float(g["col_abs_sum"].max())
    # the MI355X-order packer and its column sampler are inverses of each other at this size too
    cols = np.array([0, 1, 15, 16, 17, 127, 128, 2049, 4095])
    iwc, sc, zc = oracle.unpack_mi355x_columns(*oracle.pack_mi355x(iw, s, z), cols)
    assert np.array_equal(iwc, iw[:, cols]) and np.array_equal(zc, z[:, cols])
    assert np.array_equal(sc.view(np.uint16), s[:, cols].view(np.uint16))
