"""CPU: bench.py's bookkeeping that needs no GPU -- the rule under which a committed PMC traffic profile may be quoted in the BENCH line's
`roofline.traffic` (a kernel change without a re-profile must not leave a stale ratio in a driver-run record)."""
import importlib.util
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _load(name, rel):
    spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, rel))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_source_fingerprint_is_stable_and_follows_the_kernel_sources(tmp_path, monkeypatch):
    build = _load("_rcx_build_t", "recnext_amd/build.py")
    a = build.source_fingerprint()
    assert a == build.source_fingerprint() and len(a) == 64
    # a copy of the sources with one byte more in one kernel file has another fingerprint
    import shutil
    pkg = tmp_path / "recnext_amd"
    shutil.copytree(os.path.join(ROOT, "recnext_amd", "csrc"), pkg / "csrc", ignore=shutil.ignore_patterns("_obj"))
    shutil.copytree(os.path.join(ROOT, "include"), tmp_path / "include")
    monkeypatch.setattr(build, "CSRC", str(pkg / "csrc"))
    monkeypatch.setattr(build, "_HERE", str(pkg))
    assert build.source_fingerprint() == a
    with open(pkg / "csrc" / "rcx_common.h", "a") as f:
        f.write("\n")
    assert build.source_fingerprint() != a


def test_traffic_is_quoted_only_for_the_sources_it_was_measured_on(tmp_path):
    bench = _load("_rcx_bench_t", "bench.py")
    kern = "rcx::cpt::k_recconv_cpt<4, 2, 0, 128, unsigned short, false, 4, 0>"
    rec = {"kernel": kern, "hbm_bytes_per_launch": 3.0e8}
    json.dump({"tag": "old", "kernels": [rec]}, open(tmp_path / "r01_traffic.json", "w"))                                   # no fingerprint: stale
    json.dump({"tag": "other", "library_sources_sha256": "0" * 64, "kernels": [rec]}, open(tmp_path / "r02_traffic.json", "w"))
    got = bench.load_traffic(kern, fingerprint="f" * 64, profiles_dir=str(tmp_path))
    assert got[0] is None and got[1] is None and "2 profile file(s)" in got[2]
    json.dump({"tag": "now", "library_sources_sha256": "f" * 64, "kernels": [dict(rec, hbm_bytes_per_launch=2.5e8)]}, open(tmp_path / "r03_traffic.json", "w"))
    got = bench.load_traffic(kern, fingerprint="f" * 64, profiles_dir=str(tmp_path))
    assert got[0] == 2.5e8 and got[1].endswith("r03_traffic.json") and got[2] is None
    assert bench.load_traffic("some::other_kernel<1>", fingerprint="f" * 64, profiles_dir=str(tmp_path))[0] is None
    # the repository's own profiles: whatever is quoted carries the current fingerprint
    build = _load("_rcx_build_t2", "recnext_amd/build.py")
    val, src, note = bench.load_traffic(kern)
    if val is not None:
        assert json.load(open(os.path.join(ROOT, src)))["library_sources_sha256"] == build.source_fingerprint()
    else:
        assert "re-run tools/collect_profiles.sh" in note
