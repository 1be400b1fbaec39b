# Builds libcirclhip.so (the product) and liborc.so (the CPU oracle, test infrastructure) without Python.
ROCM ?= /opt/rocm
HIPCC ?= $(ROCM)/bin/hipcc
ARCH ?= gfx950

lib: circl_amd/libcirclhip.so
circl_amd/libcirclhip.so: circl_amd/csrc/circl_hip.hip $(wildcard circl_amd/csrc/*.h) include/circl_hip.h
	$(HIPCC) --offload-arch=$(ARCH) -O3 -std=c++17 -shared -fPIC -Wall -Wno-unused-function -Wno-unused-variable -o $@ -x hip $<

oracle:
	$(MAKE) -C oracle

example: lib
	gcc -O2 -Iinclude examples/encaps_batch.c -Lcircl_amd -lcirclhip -Wl,-rpath,$(CURDIR)/circl_amd -Wl,-rpath,$(ROCM)/lib -o build/encaps_batch

test-cpu:
	python -m pytest tests -q -m "not gpu"
test-gpu:
	python -m pytest tests -q -m gpu

.PHONY: lib oracle example test-cpu test-gpu
