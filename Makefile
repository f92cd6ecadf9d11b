# Convenience targets (the driver uses __graft_entry__.build(), which runs the same hipcc command as `make lib`).
HIPCC ?= /opt/rocm/bin/hipcc
SRC    = pymra_amd/csrc/mra_plan.hip
DEPS   = $(SRC) pymra_amd/csrc/mra_kernels.h pymra_amd/csrc/mra_topology.h include/mra_hip.h
FLAGS  = --offload-arch=gfx950 -std=c++17 -fPIC -shared -Wno-unused-value -Wno-unused-result

lib: pymra_amd/libmra_hip.so
pymra_amd/libmra_hip.so: $(DEPS)
	$(HIPCC) $(FLAGS) -O3 -o $@ $(SRC) -ldl

# Host-side AddressSanitizer + UBSan build of the same library (the device code is compiled as usual and never runs):
# used with MRA_HOST_DRYRUN=1 by tests/test_asan_host.py to push the native tree replay and the plan-construction
# index arithmetic through the sanitizers on a machine without a GPU.  GPU-side sanitizers are not available on the pool.
asan: pymra_amd/libmra_hip_asan.so
pymra_amd/libmra_hip_asan.so: $(DEPS)
	$(HIPCC) $(FLAGS) -O1 -g -Xarch_host -fsanitize=address -Xarch_host -fsanitize=undefined -Xarch_host -fno-omit-frame-pointer \
	    -Xarch_host -fno-sanitize-recover=undefined -o $@ $(SRC) -ldl

# Diagnostic build with in-kernel clock stamps in the prior row cascade (tools/stamps_cascade.py); never the product.
stamps: pymra_amd/libmra_hip_stamps.so
pymra_amd/libmra_hip_stamps.so: $(DEPS)
	$(HIPCC) $(FLAGS) -O3 -DMRA_STAMPS -o $@ $(SRC) -ldl

clean:
	rm -f pymra_amd/libmra_hip.so pymra_amd/libmra_hip_asan.so pymra_amd/libmra_hip_stamps.so
.PHONY: lib asan stamps clean
