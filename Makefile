# Builds the MI355X (gfx950) C-ABI library, the CPU oracle (test infrastructure) and the C++ host CLI.
HIPCC    ?= /opt/rocm/bin/hipcc
ARCH     ?= gfx950
CSRC     := leann-rs_amd/csrc
HIPFLAGS := -O3 -std=c++17 --offload-arch=$(ARCH) -fPIC -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function \
            -fhip-fp32-correctly-rounded-divide-sqrt
HIP_SRCS := $(wildcard $(CSRC)/*.hip)
HIP_OBJS := $(HIP_SRCS:.hip=.o)
HDRS     := $(wildcard $(CSRC)/*.cuh) $(wildcard $(CSRC)/*.h) include/leann_backend.h

HOST     := leann-rs_amd/host
CXXFLAGS := -O2 -std=c++17 -ffp-contract=off -Wall -Wextra -Wno-unused-parameter

all: $(CSRC)/libleann_hip.so oracle $(HOST)/leann $(HOST)/host_selftest $(HOST)/host_selftest_asan $(HOST)/serve_bench

$(CSRC)/%.o: $(CSRC)/%.hip $(HDRS)
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

$(CSRC)/libleann_hip.so: $(HIP_OBJS)
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@ $(HIP_OBJS)

# C++ host mirror of the reference's index layer + `leann search` CLI (links only the C ABI)
$(HOST)/leann: $(HOST)/leann_cli.cpp $(HOST)/leann_host.hpp $(HOST)/json.hpp include/leann_backend.h $(CSRC)/libleann_hip.so
	g++ $(CXXFLAGS) -o $@ $(HOST)/leann_cli.cpp -L$(CSRC) -lleann_hip -Wl,-rpath,'$$ORIGIN/../csrc' -Wl,-rpath,/opt/rocm/lib

$(HOST)/host_selftest: $(HOST)/host_selftest.cpp $(HOST)/leann_host.hpp $(HOST)/json.hpp include/leann_backend.h $(CSRC)/libleann_hip.so
	g++ $(CXXFLAGS) -o $@ $(HOST)/host_selftest.cpp -L$(CSRC) -lleann_hip -Wl,-rpath,'$$ORIGIN/../csrc' -Wl,-rpath,/opt/rocm/lib

# the reference server's call pattern (many threads, one query per call) without the HTTP layer
$(HOST)/serve_bench: $(HOST)/serve_bench.cpp include/leann_backend.h $(CSRC)/libleann_hip.so
	g++ $(CXXFLAGS) -pthread -o $@ $(HOST)/serve_bench.cpp -L$(CSRC) -lleann_hip -Wl,-rpath,'$$ORIGIN/../csrc' -Wl,-rpath,/opt/rocm/lib

# the host C++ (JSON parser, passage store, BM25, filters, search_with_options assembly) under AddressSanitizer + UBSan — CPU only
# (GPU sanitizers are not available on this pool); tests/test_cpu_host.py runs it over the same fixtures as the plain build
$(HOST)/host_selftest_asan: $(HOST)/host_selftest.cpp $(HOST)/leann_host.hpp $(HOST)/json.hpp include/leann_backend.h $(CSRC)/libleann_hip.so
	g++ -O1 -g -std=c++17 -ffp-contract=off -fsanitize=address,undefined -fno-sanitize-recover=undefined -fno-omit-frame-pointer -o $@ \
	    $(HOST)/host_selftest.cpp -L$(CSRC) -lleann_hip -Wl,-rpath,'$$ORIGIN/../csrc' -Wl,-rpath,/opt/rocm/lib

oracle:
	$(MAKE) -s -C oracle

clean:
	rm -f $(CSRC)/*.o $(CSRC)/*.so $(HOST)/leann $(HOST)/host_selftest $(HOST)/host_selftest_asan $(HOST)/serve_bench
	$(MAKE) -C oracle clean

.PHONY: all oracle clean
