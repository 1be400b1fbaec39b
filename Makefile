# Builds libcirclhip.so (the product) and liborc.so (the CPU oracle, test infrastructure) without Python.
ROCM ?= /opt/rocm
HIPCC ?= $(ROCM)/bin/hipcc
ARCH ?= gfx950

UNITS = host_runtime host_coalesce api_mlkem api_mldsa api_prims api_x25519 api_hybrid
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
	gcc -O2 -Iinclude examples/resident_keys.c -Lcircl_amd -lcirclhip -Wl,-rpath,$(CURDIR)/circl_amd -Wl,-rpath,$(ROCM)/lib -o build/resident_keys
	gcc -O2 -Iinclude examples/async_epoll.c -Lcircl_amd -lcirclhip -Wl,-rpath,$(CURDIR)/circl_amd -Wl,-rpath,$(ROCM)/lib -o build/async_epoll

# ---- sanitizer builds of the HOST side (device code cannot be instrumented) --------------------------------------------
# `make tsan` / `make asan`: every host function of the library -- the runtime (slots, movers, streams, shard) and the host
# halves of the api_* units -- instrumented (-Xarch_host), linked with the race driver into tests/_san/ (git-ignored, but it
# travels to the GPU box); tests/test_gpu_sanitizers.py runs the driver and fails on any report.  The ROCm clang has no
# sanitizer runtimes of its own, so the instrumented objects link against gcc's libtsan / libasan; san_shims.c supplies the
# two entry points clang's newer instrumentation expects and gcc 11's libtsan lacks.
SAN_DIR = tests/_san
SAN_FLAGS = --offload-arch=$(ARCH) -Xarch_host -O1 -Xarch_host -g -Xarch_device -O3 -std=c++17 -fPIC -Wno-unused-function -Wno-unused-variable -Wno-unused-command-line-argument
$(SAN_DIR)/%.tsan.o: circl_amd/csrc/%.hip $(wildcard circl_amd/csrc/*.h) include/circl_hip.h
	@mkdir -p $(SAN_DIR)
	$(HIPCC) $(SAN_FLAGS) -Xarch_host -fsanitize=thread -c $< -o $@
$(SAN_DIR)/%.asan.o: circl_amd/csrc/%.hip $(wildcard circl_amd/csrc/*.h) include/circl_hip.h
	@mkdir -p $(SAN_DIR)
	$(HIPCC) $(SAN_FLAGS) -Xarch_host -fsanitize=address -Xarch_host -fno-omit-frame-pointer -c $< -o $@
$(SAN_DIR)/libcirclhip_tsan.so: $(UNITS:%=$(SAN_DIR)/%.tsan.o) tests/san_shims.c
	gcc -O1 -fPIC -c tests/san_shims.c -o $(SAN_DIR)/san_shims.o
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@ $(UNITS:%=$(SAN_DIR)/%.tsan.o) $(SAN_DIR)/san_shims.o -lpthread
$(SAN_DIR)/libcirclhip_asan.so: $(UNITS:%=$(SAN_DIR)/%.asan.o)
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@ $(UNITS:%=$(SAN_DIR)/%.asan.o) -lpthread
$(SAN_DIR)/race_driver_tsan: tests/race_driver.cpp $(SAN_DIR)/libcirclhip_tsan.so
	g++ -std=c++17 -O1 -g -fsanitize=thread -pthread -Iinclude tests/race_driver.cpp -L$(SAN_DIR) -lcirclhip_tsan \
	    -Wl,-rpath,'$$ORIGIN' -Wl,-rpath,$(ROCM)/lib -o $@
$(SAN_DIR)/race_driver_asan: tests/race_driver.cpp $(SAN_DIR)/libcirclhip_asan.so
	g++ -std=c++17 -O1 -g -fsanitize=address -fno-omit-frame-pointer -pthread -Iinclude tests/race_driver.cpp -L$(SAN_DIR) -lcirclhip_asan \
	    -Wl,-rpath,'$$ORIGIN' -Wl,-rpath,$(ROCM)/lib -o $@
tsan: $(SAN_DIR)/race_driver_tsan
asan: $(SAN_DIR)/race_driver_asan

test-cpu:
	python -m pytest tests -q -m "not gpu"
test-gpu:
	python -m pytest tests -q -m gpu

.PHONY: lib oracle example test-cpu test-gpu tsan asan
