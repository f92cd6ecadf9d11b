"""The committed round profiles and the bench line speak about the same kernels: every kernel name in the committed
rocprofv3 statistics maps (tools/summarize_profiles.py, the mapping the PMC summaries are built with) onto a kernel
family that the committed bench line reports, and the dominant family's average launch time agrees between the two."""
import csv, glob, json, os, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def _latest(tag_glob):
    c = sorted(glob.glob(os.path.join(ROOT, "profiles", tag_glob)))
    assert c, tag_glob
    return c[-1]


def test_rocprof_kernel_names_map_onto_bench_families():
    import summarize_profiles as sp
    bench = json.load(open(_latest("r??_bench.json")))
    fams = {k["name"] for k in bench["roofline"]["kernels"]}
    tag = os.path.basename(_latest("r??_bench.json"))[:3]
    rows = list(csv.DictReader(open(os.path.join(ROOT, "profiles", tag + "_rocprofv3_kernel_stats.csv"))))
    ours = [r["Name"] for r in rows if r["Name"].startswith(("void k_", "k_"))]
    assert len(ours) >= 8
    for n in ours:
        f = sp.family(n, 0, {"cascade": -1})
        assert f in fams or f == sp.SMALL or f == sp.KNOT, (n, f)
    assert bench["roofline"]["kernel"] in fams


def test_dominant_kernel_duration_agrees_with_the_rocprof_trace():
    bench = json.load(open(_latest("r??_bench.json")))
    tag = os.path.basename(_latest("r??_bench.json"))[:3]
    dur = json.load(open(os.path.join(ROOT, "profiles", tag + "_kernel_family_durations.json")))
    dom = bench["roofline"]["kernel"]
    assert dom in dur
    # (the median over the profiled process's launches: its mean also holds the first, cold launches and those of the freshly built
    # plans of the end-to-end constructions - 1.96 and 2.09 ms against 1.70-1.75 in round 4's last collection)
    a, b = dur[dom].get("median_us", dur[dom]["avg_us"]) * 1e-3, bench["roofline"]["avg_launch_ms"]
    # rocprofv3's trace against bench.py's own hipEvents: the SAME process where its line was kept (the bench line printed under
    # the profiler, round 4 on), to 5 %; the committed un-profiled line is another run of the same command on the same box (clocks
    # and the tracer's presence differ by a few per cent): 8 %
    same = os.path.join(ROOT, "profiles", tag + "_bench_under_rocprofv3_kernel_trace.json")
    if os.path.exists(same):
        bs = json.load(open(same))
        assert bs["roofline"]["kernel"] == dom
        b_same = bs["roofline"]["avg_launch_ms"]
        assert abs(a - b_same) <= 0.06 * b_same, (a, b_same)      # four collections of round 4: 2.9, 4.6, 3.7 and 5.4 %, the trace always the longer
        assert abs(a - b) <= 0.08 * b, (a, b)
    else:
        assert abs(a - b) <= 0.05 * b, (a, b)
    tr = json.load(open(os.path.join(ROOT, "profiles", tag + "_pmc_traffic_by_kernel_family.json")))
    assert abs(tr[dom]["hbm_bytes_per_launch"] - bench["roofline"]["traffic"]) <= 1e-6 * bench["roofline"]["traffic"]
    raw = bench["roofline"]["traffic_raw_counters"]
    assert raw["FETCH_SIZE_KB"] == tr[dom]["FETCH_SIZE_KB_per_launch_mean"] and raw["WRITE_SIZE_KB"] == tr[dom]["WRITE_SIZE_KB_per_launch_mean"]


def test_byte_counters_of_the_committed_summary_are_consistent():
    """traffic = (2 FETCH_SIZE + WRITE_SIZE) * 1024 for every family (the gfx950 correction of MI355X_MICROARCH.md, calibrated in
    profiles/r02_pmc_calibration_known_streams.txt), the whole-pass figure is the sum over the families' launches of a pass, no
    family moves less than its algorithmic bytes, and the dominant kernel's measured traffic stays within 1.5 x of them."""
    tag = os.path.basename(_latest("r??_bench.json"))[:3]
    bench = json.load(open(_latest("r??_bench.json")))
    tr = json.load(open(os.path.join(ROOT, "profiles", tag + "_pmc_traffic_by_kernel_family.json")))
    tot = 0.0
    for fam, d in tr.items():
        if fam.startswith("_") or "hbm_bytes_per_launch" not in d:
            continue
        assert abs(d["hbm_bytes_per_launch"] - (2.0 * d["FETCH_SIZE_KB_per_launch_mean"] + d["WRITE_SIZE_KB_per_launch_mean"]) * 1024.0) < 1.0
        tot += d.get("hbm_bytes_per_pass", 0.0)
    assert abs(tot - tr["_whole_pass"]["hbm_bytes_per_pass"]) <= 1e-9 * tot
    alg = {k["name"]: k["hbm_bytes"] for k in bench["roofline"]["kernels"]}
    dom = bench["roofline"]["kernel"]
    assert tr[dom]["hbm_bytes_per_launch"] <= 1.5 * alg[dom]
    assert 0.9 * bench["whole_pass"]["hbm_bytes_algorithmic"] <= tot <= 1.5 * bench["whole_pass"]["hbm_bytes_algorithmic"]


def test_stale_traffic_is_not_quoted(tmp_path):
    """bench.py quotes the PMC bytes only from a summary stamped with the kernel sources that are in the tree now."""
    sys.path.insert(0, ROOT)
    import bench
    from csrc_hash import csrc_sha256
    fam = "k_predict_cascade (leaf update + all levels, mean/var)"
    body = {fam: {"hbm_bytes_per_launch": 4.0e9, "FETCH_SIZE_KB_per_launch_mean": 1.9e6, "WRITE_SIZE_KB_per_launch_mean": 1.0e5}}
    json.dump(dict(body, _collected_from={"csrc_sha256": csrc_sha256()}), open(tmp_path / "r99_pmc_traffic_by_kernel_family.json", "w"))
    t, raw, src, note = bench.traffic_from_profiles(fam, str(tmp_path))
    assert t == 4.0e9 and note is None and raw["FETCH_SIZE_KB"] == 1.9e6
    json.dump(dict(body, _collected_from={"csrc_sha256": "0" * 64}), open(tmp_path / "r99_pmc_traffic_by_kernel_family.json", "w"))
    t, raw, src, note = bench.traffic_from_profiles(fam, str(tmp_path))
    assert t is None and "other kernel sources" in note
    json.dump(body, open(tmp_path / "r99_pmc_traffic_by_kernel_family.json", "w"))
    t, raw, src, note = bench.traffic_from_profiles(fam, str(tmp_path))
    assert t is None and "no source stamp" in note
