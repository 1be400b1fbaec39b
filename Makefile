# Builds libcirclhip.so (the product) and liborc.so (the CPU oracle, test infrastructure) without Python.
ROCM ?= /opt/rocm
HIPCC ?= $(ROCM)/bin/hipcc
ARCH ?= gfx950

UNITS = host_runtime api_mlkem api_mldsa api_prims api_x25519 api_hybrid
OBJS = $(UNITS:%=build/%.o)
HIPFLAGS = --offload-arch=$(ARCH) -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Wno-unused-variable

lib: circl_amd/libcirclhip.so
build/%.o: circl_amd/csrc/%.hip $(wildcard circl_amd/csrc/*.h) include/circl_hip.h
	@mkdir -p build
	$(HIPCC) $(HIPFLAGS) -c $< -o $@
circl_amd/libcirclhip.so: $(OBJS)
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@ $(OBJS) -lpthread

oracle:
	$(MAKE) -C oracle

example: lib
	gcc -O2 -Iinclude examples/encaps_batch.c -Lcircl_amd -lcirclhip -Wl,-rpath,$(CURDIR)/circl_amd -Wl,-rpath,$(ROCM)/lib -o build/encaps_batch

test-cpu:
	python -m pytest tests -q -m "not gpu"
test-gpu:
	python -m pytest tests -q -m gpu

.PHONY: lib oracle example test-cpu test-gpu
