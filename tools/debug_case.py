"""Run one golden case on the GPU with chosen plan options (debugging aid): python tools/debug_case.py c2 [opt=val ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import _cases as K
from pymra_amd import plan as P
name = sys.argv[1]
cs = K.load_case(name)
pl = P.HipPlan(cs["topo"], 0)
pl.set_locs(cs["locs"]); pl.set_obs(cs["y_obs"], cs["c"]["R"])
s = cs["spec"]; pl.set_kernel(s.kind, s.l, s.sig, s.scale)
for a in sys.argv[2:]:
    o, v = a.split("="); pl.set_option(int(o), int(v))
print("running", name, sys.argv[2:], flush=True)
pl.run(True, True)
print("lik", sum(pl.likelihood()), flush=True)
for k in pl.kernel_stats():
    if k["launches"]: print("  %-70s %d" % (k["name"][:70], k["launches"]))
