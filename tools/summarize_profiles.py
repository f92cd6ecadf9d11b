"""Condense the rocprofv3 outputs of tools/collect_profiles.sh into the small files kept under profiles/."""
import collections, csv, glob, json, os, shutil, sys

def family(name, grid, maxgrid):
    """rocprof kernel name (+ grid) -> bench.py kernel family name (mra_plan.hip kfam_name[0], the fused path); None for the
    runtime's own fill / copy kernels (hipMemset / hipMemcpy of the set-up: not part of a pass)"""
    if name.startswith("__amd_rocclr"): return None
    if name.startswith("void k_prior_cascade"):
        return "k_prior_cascade row pass (W of all levels)" if grid == maxgrid.get("cascade") else KNOT
    if name.startswith("void k_knot_chain"): return KNOT
    if name.startswith("void k_predict_cascade"): return "k_predict_cascade (leaf update + all levels, mean/var)"
    if name.startswith("void k_leaf_gemm<2"): return "k_leaf_gemm<COV> leaf residual V[S,o] and C"
    if name.startswith("void k_leaf_gemm<1") or name.startswith("void k_gemm_nt_lds<1") or name.startswith("void k_leaf_solve_update"):
        return "k_gemm_nt_lds<SUB> / k_leaf_solve_update leaf update (separate launch)"
    if name.startswith("void k_parent_front"): return "k_parent_front (children's Ut -> parent front -> Lt, Zt, Schur)"
    if name.startswith("void k_front"): return "k_front (assembly + partial Cholesky + Schur per level)"
    if name.startswith("void k_gemm_nt<1"): return "k_gemm_nt<SUB> front Schur complement (fronts too large for LDS)"
    if name.startswith("void k_trsm_rows2") or name.startswith("void k_chol_wave") or name.startswith("void k_chol_tiles"):
        return "k_chol_wave + k_trsm_rows2 leaf factor and solves (Lc, Ut, Tt)"
    return SMALL

KNOT = "k_knot_chain + k_prior_cascade<KNOT> knot pass (knot rows, kInv, Cholesky)"
SMALL = "small kernels (k_assemble, k_leaf_cphantom, k_sum_dnode, ...)"


def sq_summary(csv_path, out_path):
    """SQ counters of a --pmc pass, averaged per (kernel, grid): matrix-pipe busy fraction, shader clock, instruction mix."""
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(csv_path)):
        k = r["Kernel_Name"][:60] + " | grid " + r["Grid_Size"]
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        acc[k]["dur_us"].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    out = {}
    for k, c in acc.items():
        m = {n: sum(v) / len(v) for n, v in c.items()}
        if m.get("dur_us", 0) < 50:
            continue
        cyc = m.get("GRBM_GUI_ACTIVE", 0) / 8.0
        m["mfma_pipe_busy_frac"] = (m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / 1024.0) / cyc if cyc else None
        m["clock_GHz"] = cyc / (m["dur_us"] * 1e3) if m.get("dur_us") else None
        out[k] = m
    json.dump(out, open(out_path, "w"), indent=1)


def main():
    if len(sys.argv) >= 4 and sys.argv[1] == "--sq":       # tools/summarize_profiles.py --sq <counter_collection.csv> <out.json>
        sq_summary(sys.argv[2], sys.argv[3])
        return
    src, tag = sys.argv[1], sys.argv[2]
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    dst = os.path.join(root, "gpurun_out", "profiles_" + tag)
    os.makedirs(dst, exist_ok=True)

    def one(pattern):
        g = glob.glob(os.path.join(src, pattern))
        return max(g, key=os.path.getmtime) if g else None       # a merged scratch directory can hold older runs too

    ks = one("kt/*/*_kernel_stats.csv")
    if ks:
        shutil.copy(ks, os.path.join(dst, tag + "_rocprofv3_kernel_stats.csv"))

    # per-family launch durations from the kernel trace of the --stats run (the stats CSV aggregates by kernel NAME, and one
    # name can serve two families: k_prior_cascade is both the knot pass and the row cascade): these are the averages to
    # set beside bench.py's roofline.avg_launch_ms
    kt = one("kt/*/*_kernel_trace.csv")
    if kt:
        rows = list(csv.DictReader(open(kt)))
        gs = lambda r: int(r.get("Grid_Size", r.get("Grid_Size_X", 0)))
        mg = {}
        for r in rows:
            n, g = r["Kernel_Name"], gs(r)
            if n.startswith("void k_prior_cascade"): mg["cascade"] = max(mg.get("cascade", 0), g)
            pass
        fam_d = collections.defaultdict(list)
        for r in rows:
            fam = family(r["Kernel_Name"], gs(r), mg)
            if fam is not None: fam_d[fam].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
        med = lambda v: sorted(v)[len(v) // 2]
        json.dump({f: {"launches": len(v), "avg_us": sum(v) / len(v), "median_us": med(v), "min_us": min(v), "max_us": max(v), "total_us": sum(v)} for f, v in sorted(fam_d.items())},
                  open(os.path.join(dst, tag + "_kernel_family_durations.json"), "w"), indent=1)

    summary = {}
    for ctr, sub in (("FETCH_SIZE", "fetch"), ("WRITE_SIZE", "write")):
        f = one(sub + "/*/*_counter_collection.csv")
        if not f:
            continue
        rows = list(csv.DictReader(open(f)))
        mg = {}
        for r in rows:
            n, g = r["Kernel_Name"], int(r["Grid_Size"])
            if n.startswith("void k_prior_cascade"): mg["cascade"] = max(mg.get("cascade", 0), g)
            pass
        acc = collections.defaultdict(list)
        disp = collections.defaultdict(set)
        for r in rows:
            fam = family(r["Kernel_Name"], int(r["Grid_Size"]), mg)
            if fam is None: continue
            acc[fam].append(float(r["Counter_Value"]))
            disp[fam].add(r["Dispatch_Id"])
        for fam, v in acc.items():
            summary.setdefault(fam, {})[ctr + "_KB_per_launch_mean"] = sum(v) / len(v)
            summary[fam]["launches_seen_" + ctr] = len(v)
    for fam, d in summary.items():
        if "FETCH_SIZE_KB_per_launch_mean" in d and "WRITE_SIZE_KB_per_launch_mean" in d:
            # gfx950: FETCH_SIZE reports half of a wide coalesced read stream (MI355X_MICROARCH.md, HBM section)
            d["hbm_bytes_per_launch"] = (2.0 * d["FETCH_SIZE_KB_per_launch_mean"] + d["WRITE_SIZE_KB_per_launch_mean"]) * 1024.0
    # per pass: a family can hold several launches of one pass (chol + two solves; five front levels); the predictive cascade
    # runs once per pass and counts them
    PRED = "k_predict_cascade (leaf update + all levels, mean/var)"
    passes = summary.get(PRED, {}).get("launches_seen_FETCH_SIZE", 0)
    if passes:
        tot = 0.0
        for fam, d in summary.items():
            if "hbm_bytes_per_launch" in d:
                d["launches_per_pass"] = d["launches_seen_FETCH_SIZE"] / passes
                d["hbm_bytes_per_pass"] = d["hbm_bytes_per_launch"] * d["launches_per_pass"]
                tot += d["hbm_bytes_per_pass"]
        summary["_whole_pass"] = {"passes_seen": passes, "hbm_bytes_per_pass": tot}
    # which sources these counters were collected from: bench.py quotes the bytes only while the stamp matches the running tree
    from csrc_hash import csrc_sha256
    summary["_collected_from"] = {"csrc_sha256": csrc_sha256(root), "tag": tag}
    json.dump(summary, open(os.path.join(dst, tag + "_pmc_traffic_by_kernel_family.json"), "w"), indent=1)

    f = one("sq/*/*_counter_collection.csv")
    if f:
        sq_summary(f, os.path.join(dst, tag + "_pmc_sq_summary.json"))
    b = os.path.join(src, "bench.json")
    if os.path.exists(b):
        shutil.copy(b, os.path.join(dst, tag + "_bench.json"))
    print("summaries in", dst, sorted(os.listdir(dst)))


if __name__ == "__main__":
    main()
