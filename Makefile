# Build of the host library, the HIP core (C-ABI) and -- separately -- the test oracle.
# `python -c "import __graft_entry__ as g; g.build()"` drives these targets.
ROOT      := $(abspath .)
CXX       ?= g++
HIPCC     ?= /opt/rocm/bin/hipcc
CXXFLAGS  := -O2 -std=c++17 -fPIC -Wall -Wextra -ffp-contract=off -DTUNGSTEN_AMD_DEFAULT_ASSETS='"$(ROOT)/assets"'
HIPFLAGS  := -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Wall -Wno-unused-function -Wno-assume

HOST_SRC  := tungsten_amd/host/scene.cpp tungsten_amd/host/capi.cpp
CLI_SRC   := tungsten_amd/host/scene.cpp tungsten_amd/host/integrator.cpp tungsten_amd/host/tungsten_cli.cpp
CLI_BIN   := tungsten_amd/tungsten_amd_cli
HOST_LIB  := tungsten_amd/libtungsten_host.so
CORE_SRC  := $(wildcard tungsten_amd/csrc/*.hip)
CORE_OBJ  := $(patsubst tungsten_amd/csrc/%.hip,build/core/%.o,$(CORE_SRC))
CORE_DEPS := $(wildcard tungsten_amd/csrc/*.h include/*.h)
CORE_LIB  := tungsten_amd/libtungsten_amd.so
ORACLE_LIB:= oracle/liboracle.so

all: host core oracle cli
cli: $(CLI_BIN)
host: $(HOST_LIB)
core: $(CORE_LIB)
oracle: $(ORACLE_LIB)

$(HOST_LIB): $(HOST_SRC) $(wildcard tungsten_amd/host/*.hpp) include/tungsten_amd.h
	$(CXX) $(CXXFLAGS) -shared -o $@ $(HOST_SRC) -lpthread

# the reference-shaped C++ host (Integrator + StandaloneRenderer loop) linked against the C ABI
$(CLI_BIN): $(CLI_SRC) $(wildcard tungsten_amd/host/*.hpp) $(CORE_LIB)
	$(CXX) $(CXXFLAGS) -o $@ $(CLI_SRC) -Ltungsten_amd -ltungsten_amd -Wl,-rpath,'$$ORIGIN' -Wl,-rpath,/opt/rocm/lib -lpthread

# one object per translation unit (the kernel families of kernels.h are split over tu_*.hip): `make -j` builds them in parallel
build/core/%.o: tungsten_amd/csrc/%.hip $(CORE_DEPS)
	@mkdir -p build/core
	$(HIPCC) $(HIPFLAGS) -c -o $@ $<

$(CORE_LIB): $(CORE_OBJ)
	$(HIPCC) $(HIPFLAGS) -shared -o $@ $(CORE_OBJ)

# the oracle uses explicit fmaf in include/tgpu_math.h: -mfma keeps it a single instruction
$(ORACLE_LIB): oracle/oracle.cpp oracle/oracle.h include/tungsten_amd.h include/tgpu_math.h
	$(CXX) -O2 -std=c++17 -fPIC -Wall -Wextra -ffp-contract=off -mfma -shared -o $@ oracle/oracle.cpp -lpthread

# host sanitizer build: the loader and its hand-written decoders under ASan + UBSan, CPU only.
# build/asan/host_check <scene.json>... loads and flattens each scene; tests/test_host.py feeds it damaged inputs.
ASAN_FLAGS := -O1 -g -std=c++17 -fsanitize=address,undefined -fno-sanitize-recover=undefined -fno-omit-frame-pointer -ffp-contract=off \
              -DTUNGSTEN_AMD_DEFAULT_ASSETS='"$(ROOT)/assets"'
asan: build/asan/host_check
build/asan/host_check: tungsten_amd/host/host_check.cpp tungsten_amd/host/scene.cpp $(wildcard tungsten_amd/host/*.hpp) include/tungsten_amd.h
	@mkdir -p build/asan
	$(CXX) $(ASAN_FLAGS) -o $@ tungsten_amd/host/host_check.cpp tungsten_amd/host/scene.cpp -lpthread

clean:
	rm -f $(HOST_LIB) $(CORE_LIB) $(ORACLE_LIB) $(CLI_BIN)
	rm -rf build/core
.PHONY: all host core oracle cli clean asan
