# Build the gfx950 library in-tree.  `make` here == __graft_entry__.build()'s HIP step.
HIPCC ?= hipcc
ARCH ?= gfx950
CSRC := artensor_amd/csrc
LIB := artensor_amd/libartn_hip.so

all: $(LIB)

$(LIB): $(CSRC)/artn_kernels.hip $(CSRC)/artn_plan.h include/artn.h
	$(HIPCC) -O3 -std=c++17 --offload-arch=$(ARCH) -fPIC -shared -Iinclude -I$(CSRC) $< -o $@

clean:
	rm -f $(LIB)
.PHONY: all clean
