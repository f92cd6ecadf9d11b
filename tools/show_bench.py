"""Print the per-kernel table of one or more bench.py JSON lines."""
import json, sys
for f in sys.argv[1:]:
    d = json.load(open(f))
    print("%s: %.3f ms/step (%.3f with kernel events), %.0f nodes/s, lik %.10e" % (f, d["ms_per_step"], d["ms_per_step_with_kernel_events"], d["value"], d["likelihood"]))
    for k in d["roofline"]["kernels"]:
        print("   %-80s %4.1f launches %8.3f ms %7.2f TFLOP/s" % (k["name"], k["launches_per_step"], k["ms_per_step"], k["tflops"]))
