"""tools/csrc_hash.py: the whole-tree stamp and the per-kernel-group stamps of the counter profiles."""
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import csrc_hash  # noqa: E402


def _tree(tmp_path):
    d = tmp_path / "noaa_apt_amd" / "csrc"
    d.mkdir(parents=True)
    src = os.path.join(ROOT, "noaa_apt_amd", "csrc")
    for n in ("apt_kernels.hpp", "apt_sync_corr.hpp", "apt_envelope.hpp", "Makefile", "apt_kernels_sync.hip",
              "apt_kernels_generic.hip", "apt_kernels_fused_impl.hpp", "apt_kernels_fused_48k_f32.hip",
              "apt_kernels_fused_any_impl.hpp", "apt_plan.hip"):
        shutil.copy(os.path.join(src, n), d / n)
    return str(tmp_path), d


def test_group_stamps_follow_their_own_sources_only(tmp_path):
    root, d = _tree(tmp_path)
    prof = csrc_hash.stamp({}, root)
    assert set(prof["csrc_groups_sha16"]) == {"front_end", "front_end_any", "chain"}
    assert all(csrc_hash.matches(prof, g, root) for g in prof["csrc_groups_sha16"])
    # an edit to the picker: the whole-tree stamp and the chain's change, the front ends' do not
    with open(d / "apt_kernels_sync.hip", "a") as f:
        f.write("// edit\n")
    assert csrc_hash.csrc_sha16(root) != prof["csrc_sha16"]
    assert csrc_hash.matches(prof, "front_end", root) and csrc_hash.matches(prof, "front_end_any", root)
    assert not csrc_hash.matches(prof, "chain", root)
    # an edit to a header every kernel includes, or to the build flags: nothing matches any more
    with open(d / "apt_kernels.hpp", "a") as f:
        f.write("// edit\n")
    assert not any(csrc_hash.matches(prof, g, root) for g in prof["csrc_groups_sha16"])
    # host code is in the whole-tree stamp only
    root2, d2 = _tree(tmp_path / "b")
    prof2 = csrc_hash.stamp({}, root2)
    with open(d2 / "apt_plan.hip", "a") as f:
        f.write("// edit\n")
    assert csrc_hash.csrc_sha16(root2) != prof2["csrc_sha16"]
    assert all(csrc_hash.matches(prof2, g, root2) for g in prof2["csrc_groups_sha16"])


def test_profiles_without_group_stamps_need_the_whole_tree(tmp_path):
    root, d = _tree(tmp_path)
    old = {"csrc_sha16": csrc_hash.csrc_sha16(root)}  # a round-4 profile
    assert csrc_hash.matches(old, "front_end", root)
    with open(d / "apt_kernels_sync.hip", "a") as f:
        f.write("// edit\n")
    assert not csrc_hash.matches(old, "front_end", root)
    assert not csrc_hash.matches(None, "front_end", root) and not csrc_hash.matches({}, "chain", root)
