#!/bin/bash
# A/B builds of libmra_hip.so with extra compiler flags:  tools/build_variant.sh <name> <flags...>  ->  pymra_amd/libmra_hip_<name>.so
# (run a tool or bench.py against it with PYMRA_AMD_LIB=pymra_amd/libmra_hip_<name>.so)
set -e
cd "$(dirname "$0")/.."
NAME=$1; shift
B=build/var_$NAME
mkdir -p $B
UNITS="mra_plan mra_launch_gemm mra_launch_pred mra_launch_prior_row1 mra_launch_prior_row2 mra_launch_prior_knot1 mra_launch_prior_knot2"
for u in $UNITS; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -std=c++17 -fPIC -Wno-unused-value -Wno-unused-result -Wno-pass-failed -O3 -Rpass-analysis=kernel-resource-usage "$@" -c pymra_amd/csrc/$u.hip -o $B/$u.o 2> $B/$u.remarks.txt &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o pymra_amd/libmra_hip_$NAME.so $(for u in $UNITS; do echo $B/$u.o; done) -ldl
cat $B/*.remarks.txt > pymra_amd/libmra_hip_$NAME.resource_usage.txt
ls -la pymra_amd/libmra_hip_$NAME.so
