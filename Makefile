# Build the gfx950 library in-tree.  `make` here == __graft_entry__.build()'s HIP step.
HIPCC ?= hipcc
ARCH ?= gfx950
CSRC := artensor_amd/csrc
LIB := artensor_amd/libartn_hip.so

all: $(LIB)

# eight objects from ONE source (see "Translation units" in artn_kernels.hip): `make -j8` builds in about a
# minute and a half instead of four
SRCS := $(CSRC)/artn_kernels.hip $(CSRC)/artn_gemm_kernel.h $(CSRC)/artn_gemm128_kernel.h $(CSRC)/artn_pgemm_kernel.h \
        $(CSRC)/artn_bits128_kernel.h $(CSRC)/artn_plan.h include/artn.h
OBJDIR := build/obj
OBJS := $(OBJDIR)/main.o $(OBJDIR)/b128.o $(foreach k,1 2 3 4 5 6,$(OBJDIR)/bits_k$(k).o)
FLAGS := -O3 -std=c++17 --offload-arch=$(ARCH) -fPIC -Iinclude -I$(CSRC)

$(OBJDIR)/main.o: $(SRCS)
	@mkdir -p $(OBJDIR)
	$(HIPCC) $(FLAGS) -DARTN_TU_MAIN -c $< -o $@
$(OBJDIR)/b128.o: $(SRCS)
	@mkdir -p $(OBJDIR)
	$(HIPCC) $(FLAGS) -DARTN_TU_B128 -c $< -o $@
$(OBJDIR)/bits_k%.o: $(SRCS)
	@mkdir -p $(OBJDIR)
	$(HIPCC) $(FLAGS) -DARTN_TU_BITS=$* -c $< -o $@
$(LIB): $(OBJS)
	$(HIPCC) --offload-arch=$(ARCH) -fPIC -shared $(OBJS) -o $@

# the same library from one translation unit (what the diagnostic targets below do with their switches)
single: $(SRCS)
	$(HIPCC) $(FLAGS) -shared $< -o $(LIB)

# diagnostic build with in-kernel phase stamps (never loaded by the product; tools/stamps.py)
stamps: $(SRCS)
	$(HIPCC) -O3 -std=c++17 --offload-arch=$(ARCH) -fPIC -shared -DARTN_STAMPS -Iinclude -I$(CSRC) $< -o tools/libartn_hip_stamps.so

phases: $(SRCS)
	$(HIPCC) -O3 -std=c++17 --offload-arch=$(ARCH) -fPIC -shared -DARTN_PHASES -Iinclude -I$(CSRC) $< -o tools/libartn_hip_phases.so

# timing-only ablations (wrong results by construction; never loaded by the product)
ablate: $(SRCS)
	$(HIPCC) -O3 -std=c++17 --offload-arch=$(ARCH) -fPIC -shared -DARTN_ABLATE_MFMA -Iinclude -I$(CSRC) $< -o tools/libartn_hip_nomfma.so
	$(HIPCC) -O3 -std=c++17 --offload-arch=$(ARCH) -fPIC -shared -DARTN_ABLATE_MEM -Iinclude -I$(CSRC) $< -o tools/libartn_hip_nomem.so
	$(HIPCC) -O3 -std=c++17 --offload-arch=$(ARCH) -fPIC -shared -DARTN_ABLATE_MEM -DARTN_ABLATE_MFMA -Iinclude -I$(CSRC) $< -o tools/libartn_hip_nomem_nomfma.so

# stand-alone HBM copy probes (tile-structured persistent copies; diagnostics only)
probes: tools/bw_probe tools/bw_probe2
tools/bw_probe: tools/bw_probe.hip
	$(HIPCC) -O3 -std=c++17 --offload-arch=$(ARCH) $< -o $@
tools/bw_probe2: tools/bw_probe2.hip
	$(HIPCC) -O3 -std=c++17 --offload-arch=$(ARCH) $< -o $@

clean:
	rm -f $(LIB)
	rm -rf $(OBJDIR)
.PHONY: all clean probes stamps phases ablate single
