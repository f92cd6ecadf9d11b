# Convenience targets (the driver uses __graft_entry__.build(), which runs the same per-file hipcc commands in parallel).
# libmra_hip.so is built from several translation units: mra_plan.hip (plan construction, launch sequence, C ABI, small kernels)
# and the dispatchers of the heavily templated kernels (mra_launch_*.hip), so that `make -j8 lib` takes ~1 min instead of 2.5.
HIPCC ?= /opt/rocm/bin/hipcc
CSRC   = pymra_amd/csrc
UNITS  = mra_plan mra_launch_gemm mra_launch_pred mra_launch_prior_row1 mra_launch_prior_row2 mra_launch_prior_knot1 mra_launch_prior_knot2
HDRS   = $(CSRC)/mra_kernels.h $(CSRC)/mra_plan_types.h $(CSRC)/mra_launch_prior.inc include/mra_hip.h
CFLAGS = --offload-arch=gfx950 -std=c++17 -fPIC -Wno-unused-value -Wno-unused-result -Wno-pass-failed
B      = build

lib: pymra_amd/libmra_hip.so
$(B)/prod/mra_plan.o $(B)/asan/mra_plan.o $(B)/stamps/mra_plan.o $(B)/whatif/mra_plan.o: $(CSRC)/mra_topology.h
$(B)/prod/%.o: $(CSRC)/%.hip $(HDRS)
	@mkdir -p $(B)/prod
	$(HIPCC) $(CFLAGS) -O3 -Rpass-analysis=kernel-resource-usage -c $< -o $@ 2> $(B)/prod/$*.remarks.txt || (cat $(B)/prod/$*.remarks.txt; exit 1)
pymra_amd/libmra_hip.so: $(UNITS:%=$(B)/prod/%.o)
	$(HIPCC) --offload-arch=gfx950 -shared -fPIC -o $@ $^ -ldl
	cat $(UNITS:%=$(B)/prod/%.remarks.txt) > pymra_amd/libmra_hip.resource_usage.txt

# Host-side AddressSanitizer + UBSan build of the same library (the device code is compiled as usual and never runs):
# used with MRA_HOST_DRYRUN=1 by tests/test_asan_host.py to push the native tree replay and the plan-construction
# index arithmetic through the sanitizers on a machine without a GPU.  GPU-side sanitizers are not available on the pool.
SAN = -Xarch_host -fsanitize=address -Xarch_host -fsanitize=undefined -Xarch_host -fno-omit-frame-pointer -Xarch_host -fno-sanitize-recover=undefined
asan: pymra_amd/libmra_hip_asan.so
$(B)/asan/%.o: $(CSRC)/%.hip $(HDRS)
	@mkdir -p $(B)/asan
	$(HIPCC) $(CFLAGS) -O1 -g $(SAN) -c $< -o $@
pymra_amd/libmra_hip_asan.so: $(UNITS:%=$(B)/asan/%.o)
	$(HIPCC) --offload-arch=gfx950 -shared -fPIC $(SAN) -o $@ $^ -ldl

# Diagnostic build with in-kernel clock stamps in the prior row cascade (tools/stamps_cascade.py); never the product.
stamps: pymra_amd/libmra_hip_stamps.so
$(B)/stamps/%.o: $(CSRC)/%.hip $(HDRS)
	@mkdir -p $(B)/stamps
	$(HIPCC) $(CFLAGS) -O3 -DMRA_STAMPS -c $< -o $@
pymra_amd/libmra_hip_stamps.so: $(UNITS:%=$(B)/stamps/%.o)
	$(HIPCC) --offload-arch=gfx950 -shared -fPIC -o $@ $^ -ldl

# Diagnostic build with the what-if switches of option 99 that CHANGE RESULTS (no Ut scatter / no W stores / constant kernel):
# timing experiments only (tools/whatif_cascade.py); the product library compiles them out and refuses the option bits.
whatif: pymra_amd/libmra_hip_whatif.so
$(B)/whatif/%.o: $(CSRC)/%.hip $(HDRS)
	@mkdir -p $(B)/whatif
	$(HIPCC) $(CFLAGS) -O3 -DMRA_WHATIF -c $< -o $@
pymra_amd/libmra_hip_whatif.so: $(UNITS:%=$(B)/whatif/%.o)
	$(HIPCC) --offload-arch=gfx950 -shared -fPIC -o $@ $^ -ldl

clean:
	rm -rf $(B) pymra_amd/libmra_hip.so pymra_amd/libmra_hip_asan.so pymra_amd/libmra_hip_stamps.so pymra_amd/libmra_hip_whatif.so pymra_amd/libmra_hip.resource_usage.txt
.PHONY: lib asan stamps whatif clean
