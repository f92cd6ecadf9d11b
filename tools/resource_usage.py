"""Parse hipcc's -Rpass-analysis=kernel-resource-usage remarks into {demangled kernel name: {vgprs, agprs, scratch, spills, occupancy, lds}}.
Used by __graft_entry__.build() (which keeps the remarks of the product build next to the library) and tests/test_resource_usage.py."""
import re
import subprocess
import sys


def parse(text):
    out = {}
    blocks = re.split(r"remark: Function Name: ", text)[1:]
    names = [b.split(" [-Rpass")[0].strip() for b in blocks]
    dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
    for b, n in zip(blocks, dem):
        def g(k):
            m = re.search(k + r": (\S+)", b)
            return int(m.group(1)) if m and m.group(1).isdigit() else None
        out[n] = dict(vgprs=g(r" VGPRs"), agprs=g("AGPRs"), scratch=g(r"ScratchSize \[bytes/lane\]"), spills=g("VGPRs Spill"),
                      sgpr_spills=g("SGPRs Spill"), occupancy=g(r"Occupancy \[waves/SIMD\]"), lds=g(r"LDS Size \[bytes/block\]"))
    return out


if __name__ == "__main__":
    res = parse(open(sys.argv[1]).read())
    only = len(sys.argv) > 2 and sys.argv[2] == "scratch"
    for n, r in sorted(res.items()):
        if only and not r["scratch"]:
            continue
        print("%-120s vgpr %3s agpr %3s scratch %4s spills %3s occ %s" % (n[:120], r["vgprs"], r["agprs"], r["scratch"], r["spills"], r["occupancy"]))
