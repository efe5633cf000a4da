"""CPU-side checks of the C-ABI library: it loads, exports every symbol the header
declares, its host helpers agree with the oracle, and it fails loudly without a GPU."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import oracle as orc
import umi_collapse_rs_amd as umi
from umi_collapse_rs_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "umihip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(umi_[a-z_]+)\s*\(", hdr))
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    L = C.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert getattr(L, name) is not None
    assert umi.load().umi_abi_version() == 2


def test_encode_matches_oracle_and_kat(kat):
    umis = [v["umi"] for v in kat["G1_G5_encode"] if len(v["umi"]) <= 21]
    for u in umis:
        k, nm = umi.to_bitset([u])
        b = orc.to_bitset(u)
        assert int(k[0]) == orc.bits_of(b)[0]
        assert int(nm[0]) == (orc.nbits_of(b) or [0])[0]
    rng = np.random.default_rng(3)
    for L in (1, 5, 12, 20, 21):
        raw = rng.choice(np.frombuffer(b"ACGTN", np.uint8), (500, L))
        k, nm = umi.to_bitset(raw.reshape(-1), L)
        ok, onm = orc.encode_keys([bytes(r).decode() for r in raw])
        assert (k == ok).all() and (nm == onm).all()


def test_encode_wide_matches_oracle_and_kat_g5():
    """umi_encode_umis_wide (host code): to_bitset for 22..85 bases against the restatement, and
    KAT G5 -- the 22-bp UMI whose last base straddles words 0 and 1."""
    import umi_collapse_rs_amd.api as api
    rng = np.random.default_rng(3)
    for L in (22, 25, 42, 43, 64, 85):
        umis = ["".join("ACGTN"[c] for c in rng.integers(0, 5, L)) for _ in range(50)]
        k, m = api.to_bitset_wide(umis, L)
        ok, om = orc.encode_keys_wide(umis)
        assert k.shape[1] == (3 * L + 63) // 64 and (k == ok).all() and (m == om).all(), L
    k, m = api.to_bitset_wide(["ACGTACGTACGTACGTACGTAN"], 22)
    assert k[0].tolist() == [0xaf0af0af0af0af0, 0x2] and m[0].tolist() == [0x8000000000000000, 0x3]
    with pytest.raises(umi.UmiHipError) as e:
        api.to_bitset_wide(["ACGTACGTACGTACGTACGTAx"], 22)
    assert e.value.code == _lib.UMI_ERR_CHAR


def test_encode_error_behaviour():
    # reference: panic on anything outside ATCGN (utils/mod.rs:77-79)
    with pytest.raises(umi.UmiHipError) as e:
        umi.to_bitset(["ACGX"])
    assert e.value.code == _lib.UMI_ERR_CHAR
    with pytest.raises(umi.UmiHipError) as e:
        umi.to_bitset(["acgt"])
    assert e.value.code == _lib.UMI_ERR_CHAR
    with pytest.raises(umi.UmiHipError) as e:
        umi.to_bitset(["A" * 22])
    assert e.value.code == _lib.UMI_ERR_ARG


def test_no_cpu_fallback_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible")
    with pytest.raises(umi.UmiHipError) as e:
        umi.Context(0)
    assert e.value.code == _lib.UMI_ERR_NODEV
    # the product package must not reach for the oracle
    import sys
    src = "".join(open(os.path.join(ROOT, "umi_collapse_rs_amd", f)).read()
                  for f in ("__init__.py", "_lib.py", "api.py"))
    assert "oracle" not in src.replace("no CPU fallback", "")


def test_multi_device_context_without_gpu_and_argument_errors():
    import torch
    L = umi.load()
    h = C.c_void_p()
    assert L.umi_ctx_create_multi(None, 2, C.byref(h)) == _lib.UMI_ERR_ARG
    ids = (C.c_int * 2)(0, 0)
    assert L.umi_ctx_create_multi(ids, 0, C.byref(h)) == _lib.UMI_ERR_ARG
    assert L.umi_ctx_device_count(None) == 0
    if not torch.cuda.is_available():
        assert L.umi_ctx_create_multi(ids, 2, C.byref(h)) == _lib.UMI_ERR_NODEV  # no CPU fallback
        assert not h.value
    off = np.array([0, 5, 3], np.uint64)  # not monotone
    owner = np.zeros(2, np.uint32)
    assert L.umi_partition_buckets(_lib.ptr(off, C.c_uint64), 2, 2, _lib.ptr(owner, C.c_uint32)) == _lib.UMI_ERR_ARG
    assert L.umi_partition_buckets(_lib.ptr(off, C.c_uint64), 2, 0, _lib.ptr(owner, C.c_uint32)) == _lib.UMI_ERR_ARG
