"""`roofline.traffic` is only ever quoted for the kernels that produced it: the counter record carries the
identity of the kernel sources and the kernel instantiation it was measured on; a library built from edited sources, or
one that launched another instantiation, gets `traffic: null` with the reason.  No GPU."""
import os
import shutil

from conftest import ROOT


def _info(build_id, sym="map_kernel<512, 1024, 17664, 0, false, 0>"):
    return f"map={sym} LDS-staged tiles sorted by block count tile=1024 reduce=reduce_pass_kernel(m<=3)+reduce_collapse_kernel+reduce_tail_kernel(<=2048 nodes) build={build_id}"


def test_traffic_is_quoted_only_for_the_build_and_kernel_it_was_measured_on(native, tmp_path, monkeypatch):
    from vk_merkle_roots_amd import build, provenance
    here = build.source_id()
    rec = {"build": here, "map_kernel_symbol": "map_kernel<512, 1024, 17664, 0, false, 0>", "strings_per_map_launch": 1 << 26, "maxlen": 127, "slice_log2": 26,
           "map_kernel_hbm_bytes_per_launch": 7.1e9, "reduce_hbm_bytes_per_step": 2.3e9}
    t, src = provenance.traffic_from_pmc(rec, _info(here), "map_kernel_hbm_bytes_per_launch", strings_per_map_launch=1 << 26, maxlen=127)
    assert t == 7.1e9 and src["used"] and src["file"] == "profiles/pmc_latest.json"
    # another launch shape, another kernel instantiation: no figure
    assert provenance.traffic_from_pmc(rec, _info(here), "map_kernel_hbm_bytes_per_launch", strings_per_map_launch=1 << 23, maxlen=127)[0] is None
    t, src = provenance.traffic_from_pmc(rec, _info(here, "map_kernel<512, 2048, 64, 2, true, 0>"), "map_kernel_hbm_bytes_per_launch", strings_per_map_launch=1 << 26, maxlen=127)
    assert t is None and "instantiation" in src["why"]
    # a kernel header is edited: the library built from it carries another id and the old record no longer applies
    csrc = tmp_path / "csrc"
    shutil.copytree(build.CSRC, csrc)
    with open(csrc / "map_kernel.hpp", "a") as f:
        f.write("\n// an edit\n")
    monkeypatch.setattr(build, "CSRC", str(csrc))
    edited = build.source_id()
    assert edited != here
    t, src = provenance.traffic_from_pmc(rec, _info(edited), "map_kernel_hbm_bytes_per_launch", strings_per_map_launch=1 << 26, maxlen=127)
    assert t is None and src["used"] is False and "different build" in src["why"]
    t, src = provenance.traffic_from_pmc(rec, _info(edited), "reduce_hbm_bytes_per_step", strings_per_map_launch=1 << 26, maxlen=127, slice_log2=26)
    assert t is None
    # a library from before the id existed, or no record at all
    assert provenance.traffic_from_pmc(rec, "map=... reduce=...", "map_kernel_hbm_bytes_per_launch")[0] is None
    assert provenance.traffic_from_pmc(None, _info(here), "map_kernel_hbm_bytes_per_launch")[0] is None


def test_the_loaded_library_reports_the_id_of_the_sources_in_the_tree(native):
    """The id compiled into libvkmr_hip.so is the one the current sources give: the library in the tree is up to date,
    and the build parameters (issue pass settings) are part of the identity."""
    import vk_merkle_roots_amd as vk
    from vk_merkle_roots_amd import build, provenance
    info = vk.lib().vkmr_hip_kernel_info().decode()
    assert provenance.build_id_of(info) == build.source_id()
    assert build.source_id(split_every=build.SPLIT_ADD3_EVERY + 1) != build.source_id()
    assert os.path.exists(os.path.splitext(build.HIP_LIB)[0] + ".isa.json")


def test_experiment_only_sources_do_not_change_the_product_id(native, tmp_path, monkeypatch):
    """map_experiments.hpp and csrc/experiments/ are compiled into the experiments build only: editing them must not
    invalidate the records measured on the product library (and must change the experiments build's id)."""
    from vk_merkle_roots_amd import build
    csrc = tmp_path / "csrc"
    shutil.copytree(build.CSRC, csrc)
    monkeypatch.setattr(build, "CSRC", str(csrc))
    product, exp = build.source_id(), build.source_id(["-DVKMR_EXPERIMENTS"])   # of the copy (the id also covers the file order)
    assert product != exp
    with open(csrc / "map_experiments.hpp", "a") as f:
        f.write("\n// another variant\n")
    with open(csrc / "experiments" / "sha256d_lds.hpp", "a") as f:
        f.write("\n// another form\n")
    assert build.source_id() == product
    assert build.source_id(["-DVKMR_EXPERIMENTS"]) != exp
    with open(csrc / "map_kernel.hpp", "a") as f:
        f.write("\n// the product's kernel\n")
    assert build.source_id() != product
