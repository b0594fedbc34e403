"""CPU suite: libmeerqat_hip.so builds for gfx950, loads without a GPU, and exports every symbol that
include/meerqat_hip.h declares (no compute calls here)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "meerqat_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mq_[a-z0-9_]+)\s*\(", text)))


def test_every_declared_symbol_is_exported(hip_lib):
    names = _declared()
    assert len(names) >= 12
    for n in names:
        assert hasattr(hip_lib, n), f"{n} declared in include/meerqat_hip.h but not exported"


def test_binding_table_covers_header(hip_lib):
    from viquae_amd import _lib
    assert sorted(_lib.SIGNATURES) == _declared()


def test_host_only_entries(hip_lib):
    assert b"gfx950" in hip_lib.mq_version()
    assert hip_lib.mq_padded_rows(1) == 256 and hip_lib.mq_padded_rows(257) == 512
    assert hip_lib.mq_padded_dim(768) == 768 and hip_lib.mq_padded_dim(100) == 112
    assert hip_lib.mq_packed_bytes(1_500_000, 768) == 1_500_160 * 768 * 4
    assert hip_lib.mq_strerror(-2) == b"workspace too small"
    # argument validation happens before any HIP call
    assert hip_lib.mq_knn_search_f32(None, None, 10, 8, None, 1, 5, 0, 0, 0, None, None, None, 0, None) == -1
    assert hip_lib.mq_knn_workspace_bytes(1000, 64, 10, 4096) == 0  # k above MQ_KNN_MAX_K


def test_code_object_is_gfx950():
    from viquae_amd import build
    data = open(build.SO, "rb").read()
    assert b"gfx950" in data and b"knn_scan_kernel" in data


def test_product_does_not_import_oracle():
    """The product path must never route through the oracle (or any CPU fallback)."""
    pkg = os.path.join(ROOT, "viquae_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f"{f} imports the oracle"


def test_search_path_switches_are_set_through_the_abi(hip_lib):
    """the A/B switches are process-wide atomics behind mq_knn_set_option / mq_knn_get_option (first values from the
    environment, read once) -- not getenv() on the search path."""
    from viquae_amd import _lib
    assert hip_lib.mq_knn_get_option(_lib.KNN_OPT_SMALL_SCAN) in (0, 1)
    before = hip_lib.mq_knn_get_option(_lib.KNN_OPT_PARTITIONS)
    with _lib.knn_option(_lib.KNN_OPT_PARTITIONS, 0) as o:
        assert o.previous == before and hip_lib.mq_knn_get_option(_lib.KNN_OPT_PARTITIONS) == 0
        # the planner follows the switch at once: k = 300 over 1.5M rows is served over row ranges, or by exact rounds
        assert hip_lib.mq_knn_screen_scan_kind(1_500_000, 768, 4096, 300, 0) == 0
    assert hip_lib.mq_knn_get_option(_lib.KNN_OPT_PARTITIONS) == before
    if before:
        assert hip_lib.mq_knn_screen_scan_kind(1_500_000, 768, 4096, 300, 0) == 1
    assert hip_lib.mq_knn_set_option(99, 1) == -1 and hip_lib.mq_knn_set_option(0, -5) == -1 and hip_lib.mq_knn_get_option(-1) == -1
