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

# Tools build: the same sources with -DLDM_TOOLS_BUILD (timing ablations of the persistent kernel,
# A/B environment switches of the tile cost model).  Never loaded by the product path: tools/ load it
# explicitly (tools/toolslib.py).  The product library reads no result-changing environment variable.
TOUT  := ldm_tf2_amd/lib/libldm_hip_tools.so
TOBJS := $(patsubst $(CSRC)/%.hip,build_tools/%.o,$(SRCS))

build_tools/%.o: $(CSRC)/%.hip $(HDRS)
	@mkdir -p build_tools
	$(HIPCC) $(CXXFLAGS) -DLDM_TOOLS_BUILD -c $< -o $@

$(TOUT): $(TOBJS)
	@mkdir -p $(dir $(TOUT))
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@ $(TOBJS)

tools: $(TOUT)

clean:
	rm -rf build build_tools $(OUT) $(TOUT)

.PHONY: all clean tools
