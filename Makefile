# Top-level build: the product library (HIP, gfx950 only) and the checkers (oracle/).
# `python -c "import __graft_entry__ as g; g.build()"` drives the same commands.
HIPCC    ?= /opt/rocm/bin/hipcc
ARCH     ?= gfx950
LIBDIR   := pathtracer_amd/lib
CSRC     := pathtracer_amd/csrc
HOST     := pathtracer_amd/host
# -ffp-contract=off: value-carrying arithmetic keeps the reference's rounding; the sphere search re-enables
# contraction locally with `#pragma clang fp contract(fast)` (pt_device.h).
# -fno-slp-vectorize: the SLP vectoriser pairs independent scalar FP32 operations into v_pk_* instructions, which issue at
# half the plain rate on gfx950 and want their operands in register pairs (moves, spills): book-1 +7.2 % without it
# (81.1 -> 75.7 ms, 19 -> 14 spilled VGPRs), C2 / C4 scenes +4.6 / +2.4 %.  Packed arithmetic stays where it is written out
# by hand and measured to pay (the few-sphere loop's filter, the brute-force search).
# -amdgpu-sched-strategy=max-ilp: +1 % on the bench kernel against the default strategy (tools/ab.py), none of the others measured better
HIPFLAGS := --offload-arch=$(ARCH) -O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize -mllvm -amdgpu-sched-strategy=max-ilp -Wall -Wno-unused-function
CXXFLAGS := -O2 -std=c++17 -fPIC -ffp-contract=off -Wall

BINDIR   := pathtracer_amd/bin

all: lib cli oracle

lib: $(LIBDIR)/libpt_hip.so

# pt_render: the reference's main() as a command-line tool over the C-ABI (links against nothing but libpt_hip.so)
cli: $(BINDIR)/pt_render
$(BINDIR)/pt_render: $(HOST)/pt_main.cpp include/pt_hip.h $(LIBDIR)/libpt_hip.so
	@mkdir -p $(BINDIR)
	g++ -O2 -std=c++17 -Wall $< -o $@ -L$(LIBDIR) -lpt_hip -pthread -Wl,-rpath,'$$ORIGIN/../lib' -Wl,-rpath-link,/opt/rocm/lib

$(LIBDIR)/pt_hip.o: $(CSRC)/pt_hip.hip $(CSRC)/pt_device.h $(CSRC)/pt_host.h $(CSRC)/pt_bvh.h $(CSRC)/pt_guard.h include/pt_hip.h
	@mkdir -p $(LIBDIR)
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

$(LIBDIR)/pt_world.o: $(HOST)/pt_world.cpp $(HOST)/pt_world.h $(CSRC)/pt_guard.h include/pt_hip.h
	@mkdir -p $(LIBDIR)
	g++ $(CXXFLAGS) -c $< -o $@

# pt_node.hip: host-only code (contexts, streams, RCCL bound with dlopen); hipcc for the HIP / RCCL headers
$(LIBDIR)/pt_node.o: $(CSRC)/pt_node.hip $(CSRC)/pt_guard.h include/pt_hip.h
	@mkdir -p $(LIBDIR)
	$(HIPCC) --offload-arch=$(ARCH) -O2 -std=c++17 -fPIC -Wall -c $< -o $@

$(LIBDIR)/libpt_hip.so: $(LIBDIR)/pt_hip.o $(LIBDIR)/pt_world.o $(LIBDIR)/pt_node.o
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC $^ -o $@ -ldl -pthread

oracle:
	$(MAKE) -C oracle all

isa: $(CSRC)/pt_hip.hip $(CSRC)/pt_device.h
	@mkdir -p $(LIBDIR)/isa
	$(HIPCC) $(HIPFLAGS) -save-temps=obj -Rpass-analysis=kernel-resource-usage -c $< -o $(LIBDIR)/isa/pt_hip.o 2> $(LIBDIR)/isa/resource_usage.txt || (cat $(LIBDIR)/isa/resource_usage.txt; false)

clean:
	rm -rf $(LIBDIR) $(BINDIR) ; $(MAKE) -C oracle clean

# Host-side hardening: the scene model, the scene-file parser, both tree builders and the
# widening under AddressSanitizer + UBSan (host code only: plain g++, no HIP), and allocation / thread-spawn fault
# injection against the built library.  tests/test_host_hardening.py builds and runs both.
SANDIR := tests/native/build
asan: $(SANDIR)/host_sanitize
$(SANDIR)/host_sanitize: tests/native/host_sanitize.cpp $(HOST)/pt_world.cpp $(HOST)/pt_world.h $(CSRC)/pt_bvh.h $(CSRC)/pt_guard.h include/pt_hip.h
	@mkdir -p $(SANDIR)
	g++ -O1 -g -std=c++17 -Wall -ffp-contract=off -fsanitize=address,undefined -fno-sanitize-recover=undefined -fno-omit-frame-pointer -static-libasan -static-libubsan tests/native/host_sanitize.cpp $(HOST)/pt_world.cpp -o $@ -pthread
oom: $(SANDIR)/host_oom
$(SANDIR)/host_oom: tests/native/host_oom.cpp include/pt_hip.h $(LIBDIR)/libpt_hip.so
	@mkdir -p $(SANDIR)
	g++ -O1 -g -std=c++17 -Wall tests/native/host_oom.cpp -o $@ -L$(LIBDIR) -lpt_hip -pthread -Wl,-rpath,'$$ORIGIN/../../../$(LIBDIR)' -Wl,-rpath-link,/opt/rocm/lib

.PHONY: all lib cli oracle isa clean asan oom

# Diagnostic build with in-kernel s_memtime stamps (never shipped, never benchmarked: read its SHARES, not its speed).
stamps: $(CSRC)/pt_hip.hip $(CSRC)/pt_device.h $(CSRC)/pt_host.h $(LIBDIR)/pt_world.o $(LIBDIR)/pt_node.o
	$(HIPCC) $(HIPFLAGS) -DPT_STAMPS -c $(CSRC)/pt_hip.hip -o $(LIBDIR)/pt_hip_stamps.o
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC $(LIBDIR)/pt_hip_stamps.o $(LIBDIR)/pt_world.o $(LIBDIR)/pt_node.o -o $(LIBDIR)/libpt_hip_stamps.so -ldl -pthread

# A/B experiment builds: make variant NAME=foo DEFS="-DPT_X -DPT_Y"  ->  pathtracer_amd/lib/ab/libpt_hip_foo.so (git-ignored)
variant: $(CSRC)/pt_hip.hip $(CSRC)/pt_device.h $(CSRC)/pt_host.h $(LIBDIR)/pt_world.o $(LIBDIR)/pt_node.o
	mkdir -p $(LIBDIR)/ab
	$(HIPCC) $(HIPFLAGS) $(DEFS) -c $(CSRC)/pt_hip.hip -o $(LIBDIR)/ab/pt_hip_$(NAME).o
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC $(LIBDIR)/ab/pt_hip_$(NAME).o $(LIBDIR)/pt_world.o $(LIBDIR)/pt_node.o -o $(LIBDIR)/ab/libpt_hip_$(NAME).so -ldl -pthread
	rm -f $(LIBDIR)/ab/pt_hip_$(NAME).o
