# Builds libldm_hip.so (hand-written HIP kernels for gfx950) in-tree.
HIPCC ?= /opt/rocm/bin/hipcc
ARCH  ?= gfx950
CSRC  := ldm_tf2_amd/csrc
OUT   := ldm_tf2_amd/lib/libldm_hip.so
SRCS  := $(wildcard $(CSRC)/*.hip)
OBJS  := $(patsubst $(CSRC)/%.hip,build/%.o,$(SRCS))
CXXFLAGS := --offload-arch=$(ARCH) -O3 -fPIC -std=c++17 -Wall -Wno-unused-function -Iinclude

all: $(OUT)

HDRS  := $(wildcard $(CSRC)/*.h) include/ldm_hip.h

build/%.o: $(CSRC)/%.hip $(HDRS)
	@mkdir -p build
	$(HIPCC) $(CXXFLAGS) -c $< -o $@

$(OUT): $(OBJS)
	@mkdir -p $(dir $(OUT))
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@ $(OBJS)

clean:
	rm -rf build $(OUT)

.PHONY: all clean
