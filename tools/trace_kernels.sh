#!/bin/bash
# per-dispatch kernel durations of a command (rocprofv3 --kernel-trace), condensed: one line per dispatch of the small
# kernels, aggregated lines for the rest.   usage: tools/trace_kernels.sh <tag> -- python3 ...
set -e
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=$1; shift; shift
OUT=$R/gpurun_out/trace_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"

rocprofv3 --kernel-trace --output-format csv -d "$OUT" -- "$@" > "$OUT/cmd.log" 2>&1
cd "$R"
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/*/*_kernel_trace.csv")[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# last complete pass: from the last k_prior_cascade with the smallest grid before the end
names = [r["Kernel_Name"] for r in rows]
ends = [i for i, n in enumerate(names) if n.startswith("k_sum_dnode")]
import os
nsum = int(os.environ.get("NSUM", "1"))            # k_sum_dnode launches per pass (2 in a split/sharded pass)
if len(ends) >= nsum + 1:
    seg = rows[ends[-nsum - 1] + 1: ends[-1] + 1]
else:
    seg = rows
t0 = int(seg[0]["Start_Timestamp"])
prev_end = t0
for r in seg:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%9.1f us  +gap %6.1f  dur %8.1f us  grid %9s wg %4s  %s" % ((s - t0) / 1e3, (s - prev_end) / 1e3, (e - s) / 1e3, r.get("Grid_Size", r.get("Grid_Size_X")), r.get("Workgroup_Size", r.get("Workgroup_Size_X")), r["Kernel_Name"][:70]))
    prev_end = e
print("pass: %.1f us from first start to last end" % ((int(seg[-1]["End_Timestamp"]) - t0) / 1e3))
PY
