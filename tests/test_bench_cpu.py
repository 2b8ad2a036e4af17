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


def test_plan_strings_map_to_the_device_kernel_names_rocprofv3_reports():
    """bench.py matches its per-mixer times to the committed PMC profiles by kernel name: the mapping from rcx_recconv2d_fwd_plan's strings has to
    follow the kernels' template arguments (round 5 added the tile side to k_recconv_cpt)."""
    bench = _load("_rcx_bench_t3", "bench.py")
    kn = bench.kernel_name
    assert kn("cpt(k_recconv_cpt<4, 2, 0, 128>,cb=32,nt=512,units=512,lds=137856)", 2) == "rcx::cpt::k_recconv_cpt<4, 2, 0, 128, unsigned short, false, 4, 0, 14>"
    assert kn("cpt(k_recconv_cpt<2, 1, 0, 256>,cb=64,nt=256,units=512,lds=71424)", 2) == "rcx::cpt::k_recconv_cpt<2, 1, 0, 256, unsigned short, false, 3, 0, 14>"
    assert kn("cpt(k_recconv_cpt<4, 4, 0, 0>,levels-1,cb=16,nt=256,units=8,lds=68928)", 4) == "rcx::cpt::k_recconv_cpt<4, 4, 0, 0, float, false, 3, 0, 14>"
    assert kn("cpt(k_recconv_cpt<4, 4, 0, 0, ts=16>,cb=16,nt=256,units=256,lds=88320)", 2) == "rcx::cpt::k_recconv_cpt<4, 4, 0, 0, unsigned short, false, 3, 0, 16>"
    assert kn("cpl(k_recconv_cpl14<0, 256>,cb=64,nt=64,blocks=1024,lds=0)", 2) == "rcx::cpl14::k_recconv_cpl14<0, 256, unsigned short, false, 2, false>"


def test_every_mixer_of_the_five_baseline_configs_has_a_traffic_record_in_this_rounds_profiles():
    """`roofline.traffic` must not be null for any BASELINE configuration -- every token-mixer kernel (or multi-launch unit) that the
    round's profiled bench lines name has HBM bytes per launch in the traffic file of the same run."""
    for tag in ("r05", "r05_m1", "r05_m5", "r05_a3", "r05_512"):
        line = json.loads(open(os.path.join(ROOT, "profiles", f"{tag}_bench.json")).read().strip().split("\n")[-1])
        traffic = json.load(open(os.path.join(ROOT, "profiles", f"{tag}_traffic.json")))
        have = {k["kernel"] for k in traffic["kernels"] if k.get("hbm_bytes_per_launch") is not None}
        for ent in line["token_mixers"]["per_kernel"]:
            assert ent["kernel"] in have, (tag, ent["kernel"])
