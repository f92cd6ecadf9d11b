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
    a, b = dur[dom]["avg_us"] * 1e-3, bench["roofline"]["avg_launch_ms"]
    assert abs(a - b) <= 0.05 * b, (a, b)          # hipEvents inside bench.py vs rocprofv3 of the same command
    tr = json.load(open(os.path.join(ROOT, "profiles", tag + "_pmc_traffic_by_kernel_family.json")))
    assert abs(tr[dom]["hbm_bytes_per_launch"] - bench["roofline"]["traffic"]) <= 1e-6 * bench["roofline"]["traffic"]
