# Builds the MI355X (gfx950) C-ABI library, the CPU oracle (test infrastructure) and the C++ host CLI.
HIPCC    ?= /opt/rocm/bin/hipcc
ARCH     ?= gfx950
CSRC     := leann-rs_amd/csrc
HIPFLAGS := -O3 -std=c++17 --offload-arch=$(ARCH) -fPIC -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function \
            -fhip-fp32-correctly-rounded-divide-sqrt
HIP_SRCS := $(wildcard $(CSRC)/*.hip)
HIP_OBJS := $(HIP_SRCS:.hip=.o)
HDRS     := $(wildcard $(CSRC)/*.cuh) $(wildcard $(CSRC)/*.h) include/leann_backend.h

all: $(CSRC)/libleann_hip.so oracle

$(CSRC)/%.o: $(CSRC)/%.hip $(HDRS)
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

$(CSRC)/libleann_hip.so: $(HIP_OBJS)
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@ $(HIP_OBJS)

oracle:
	$(MAKE) -s -C oracle

clean:
	rm -f $(CSRC)/*.o $(CSRC)/*.so
	$(MAKE) -C oracle clean

.PHONY: all oracle clean
