"""The built gfx950 code object itself (no GPU needed): what DESIGN.md section 5 states about the production kernels'
registers and scratch is read back from the library's own metadata (llvm-readelf --notes on the unbundled .hip_fatbin,
tools/kernel_meta.py), so that a later change that brings a spill or a scratch segment back fails here and not in a profile."""
import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "tools"))
import kernel_meta  # noqa: E402


@pytest.fixture(scope="module")
def meta():
    if not os.path.exists(kernel_meta.DEFAULT_LIB):
        pytest.fail("librtamd.so is not built (python -c 'import __graft_entry__ as g; g.build()')")
    return {k.split("(")[0].replace("void ", ""): v for k, v in kernel_meta.kernel_meta().items()}


def test_production_kernels_keep_no_scratch_and_spill_no_vgpr(meta):
    # the fused shader (default: no per-pixel counters, no material table, float texture filter) and ray_trace
    # (last argument of the fused kernel: the single-frame instantiation / the one batches use, which serves both)
    for name in ("rtamd::k_persistent<0, false, false, false, false>", "rtamd::k_persistent<0, false, false, false, true>",
                 "rtamd::k_persistent<1, false, false, false, false>"):
        m = meta[name]
        assert m["private_segment_fixed_size"] == 0, (name, m)
        assert m["vgpr_spill_count"] == 0, (name, m)


def test_persistent_kernels_fit_eight_workgroups_per_cu(meta):
    """8 waves per SIMD: at most 64 VGPRs, and eight 256-thread workgroups' LDS inside a CU's 160 KiB."""
    pers = {k: v for k, v in meta.items() if k.startswith("rtamd::k_persistent<")}
    assert len(pers) == 18            # fused: counters x materials x filter x (single frame | batch); trace: counters
    for name, m in pers.items():
        assert m["vgpr_count"] <= 64, (name, m)
        assert 8 * m["group_segment_fixed_size"] <= 160 * 1024, (name, m)


def test_streaming_kernels_use_no_scratch(meta):
    for name in ("rtamd::k_generate_rays", "rtamd::k_surface_from_hits", "rtamd::k_dispatch_rays", "rtamd::k_assemble_tiles", "rtamd::k_display_rgba8"):
        assert meta[name]["private_segment_fixed_size"] == 0 and meta[name]["vgpr_spill_count"] == 0, (name, meta[name])
