# Build the gfx950 library in-tree.  `make` here == __graft_entry__.build()'s HIP step.
HIPCC ?= hipcc
ARCH ?= gfx950
CSRC := artensor_amd/csrc
LIB := artensor_amd/libartn_hip.so

all: $(LIB)

# thirteen objects from ONE source (see "Translation units" in artn_kernels.hip): `make -j8` builds in about a
# minute and a half instead of four
SRCS := $(CSRC)/artn_kernels.hip $(CSRC)/artn_gemm_kernel.h $(CSRC)/artn_gemm128_kernel.h $(CSRC)/artn_pgemm_kernel.h $(CSRC)/artn_xgemm128_kernel.h \
        $(CSRC)/artn_bits128_kernel.h $(CSRC)/artn_bits3_kernel.h $(CSRC)/artn_wide_kernel.h $(CSRC)/artn_plan.h \
        $(CSRC)/artn_xgemm_plan.h $(CSRC)/artn_xgemm_kernel.h $(CSRC)/artn_xrow_kernel.h $(CSRC)/artn_xgemm_pc_kernel.h include/artn.h
OBJDIR := build/obj
# The product library carries what the default planner can select.  `make dev` (DEV=1) adds the development-only pieces:
# every ARTN_* planner switch of the A/B measurements in DESIGN.md (-DARTN_DEV_SWITCHES), three-step fusion (artn_k_bits3 /
# artn_contract3: -DARTN_DEV_BITS3, three more translation units) and the split-bf16 instantiations (-DARTN_DEV_SPLIT3).
DEV ?= 0
ifeq ($(DEV),1)
DEVFLAGS := -DARTN_DEV_SWITCHES -DARTN_DEV_BITS3 -DARTN_DEV_SPLIT3 -DARTN_DEV_XGPC
DEVOBJS := $(foreach k,3 4 5,$(OBJDIR)/bits3_k$(k).o)
else
DEVFLAGS :=
DEVOBJS :=
endif
# (the longest translation units first: make -j starts its jobs in this order)
OBJS := $(foreach k,6 5,$(OBJDIR)/bits_k$(k)h0.o $(OBJDIR)/bits_k$(k)h1.o) $(OBJDIR)/main.o $(OBJDIR)/bits_k4.o $(OBJDIR)/bits_k3.o \
        $(OBJDIR)/b128.o $(OBJDIR)/b128a.o $(OBJDIR)/wide.o $(OBJDIR)/bits_k2.o $(OBJDIR)/bits_k1.o $(DEVOBJS)
FLAGS := -O3 -std=c++17 --offload-arch=$(ARCH) -fPIC -Iinclude -I$(CSRC) $(DEVFLAGS)

$(OBJDIR)/main.o: $(SRCS)
	@mkdir -p $(OBJDIR)
	$(HIPCC) $(FLAGS) -DARTN_TU_MAIN -c $< -o $@
$(OBJDIR)/b128.o: $(SRCS)
	@mkdir -p $(OBJDIR)
	$(HIPCC) $(FLAGS) -DARTN_TU_B128 -c $< -o $@
$(OBJDIR)/b128a.o: $(SRCS)
	@mkdir -p $(OBJDIR)
	$(HIPCC) $(FLAGS) -DARTN_TU_B128A -c $< -o $@
$(OBJDIR)/wide.o: $(SRCS)
	@mkdir -p $(OBJDIR)
	$(HIPCC) $(FLAGS) -DARTN_TU_WIDE -c $< -o $@
$(OBJDIR)/bits3_k%.o: $(SRCS)
	@mkdir -p $(OBJDIR)
	$(HIPCC) $(FLAGS) -DARTN_TU_BITS3=$* -c $< -o $@
# (the families of 5 and 6 contracted bits in two halves each: second-stage counts 0..3 / 4..6)
$(OBJDIR)/bits_k%h0.o: $(SRCS)
	@mkdir -p $(OBJDIR)
	$(HIPCC) $(FLAGS) -DARTN_TU_BITS=$* -DARTN_TU_HALF=0 -c $< -o $@
$(OBJDIR)/bits_k%h1.o: $(SRCS)
	@mkdir -p $(OBJDIR)
	$(HIPCC) $(FLAGS) -DARTN_TU_BITS=$* -DARTN_TU_HALF=1 -c $< -o $@
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
	$(HIPCC) -O3 -std=c++17 --offload-arch=$(ARCH) -fPIC -shared -DARTN_STAMPS -DARTN_DEV_SWITCHES -Iinclude -I$(CSRC) $< -o tools/libartn_hip_stamps.so

phases: $(SRCS)
	$(HIPCC) -O3 -std=c++17 --offload-arch=$(ARCH) -fPIC -shared -DARTN_PHASES -DARTN_DEV_SWITCHES -Iinclude -I$(CSRC) $< -o tools/libartn_hip_phases.so

# timing-only ablations (wrong results by construction; never loaded by the product)
ablate: $(SRCS)
	$(HIPCC) -O3 -std=c++17 --offload-arch=$(ARCH) -fPIC -shared -DARTN_ABLATE_MFMA -Iinclude -I$(CSRC) $< -o tools/libartn_hip_nomfma.so
	$(HIPCC) -O3 -std=c++17 --offload-arch=$(ARCH) -fPIC -shared -DARTN_ABLATE_MEM -Iinclude -I$(CSRC) $< -o tools/libartn_hip_nomem.so
	$(HIPCC) -O3 -std=c++17 --offload-arch=$(ARCH) -fPIC -shared -DARTN_ABLATE_MEM -DARTN_ABLATE_MFMA -Iinclude -I$(CSRC) $< -o tools/libartn_hip_nomem_nomfma.so

# stand-alone HBM copy probes (tile-structured persistent copies; diagnostics only)
probes: tools/bw_probe tools/bw_probe2 tools/probes/plane_probe tools/probes/xrow64_probe
tools/probes/plane_probe: tools/probes/plane_probe.hip
	$(HIPCC) -O3 -std=c++17 --offload-arch=$(ARCH) $< -o $@
tools/probes/xrow64_probe: tools/probes/xrow64_probe.hip
	$(HIPCC) -O3 -std=c++17 --offload-arch=$(ARCH) $< -o $@
tools/bw_probe: tools/bw_probe.hip
	$(HIPCC) -O3 -std=c++17 --offload-arch=$(ARCH) $< -o $@
tools/bw_probe2: tools/bw_probe2.hip
	$(HIPCC) -O3 -std=c++17 --offload-arch=$(ARCH) $< -o $@

# CPU sanitizer build (SURVEY section 5 hook; AddressSanitizer + UBSan on the HOST side only -- GPU ASan is not available
# on this pool): the library's host half (planner artn_plan.h, every extern "C" entry point, the program builder) compiled
# with --offload-host-only and linked against a stand-in for the device fat binary (never launched: no GPU here), plus the
# CPU plan emulator; then the CPU test suite and the planner stress run on them.  Log: profiles/rNN_asan.log.
ASAN_RT := $(firstword $(wildcard /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so))
ASAN_DIR := build/asan
ASAN_FLAGS := -O1 -g -std=c++17 -fsanitize=address,undefined -fno-sanitize-recover=undefined -fno-omit-frame-pointer -shared-libsan -fPIC -Iinclude -I$(CSRC)
ASAN_LOG ?= profiles/r06_asan.log
asan: $(SRCS) tests/csrc/plan_emulate.cpp
	@mkdir -p $(ASAN_DIR)
	$(HIPCC) $(ASAN_FLAGS) --offload-host-only -c $(CSRC)/artn_kernels.hip -o $(ASAN_DIR)/host.o
	echo "__attribute__((aligned(4096))) const char $$(nm -u $(ASAN_DIR)/host.o | grep -o '__hip_fatbin_[0-9a-f]*' | head -1)[4096] = {0};" > $(ASAN_DIR)/fatbin_stub.c
	/opt/rocm/lib/llvm/bin/clang -fPIC -c $(ASAN_DIR)/fatbin_stub.c -o $(ASAN_DIR)/fatbin_stub.o
	$(HIPCC) -fsanitize=address,undefined -shared-libsan -fPIC -shared $(ASAN_DIR)/host.o $(ASAN_DIR)/fatbin_stub.o -o $(ASAN_DIR)/libartn_host_asan.so
	/opt/rocm/lib/llvm/bin/clang++ $(ASAN_FLAGS) -shared tests/csrc/plan_emulate.cpp -o $(ASAN_DIR)/libplan_emulate_asan.so
	( echo "# make asan: $$(date -u +%FT%TZ), sources $$(python3 -c 'import bench; print(bench.kernel_source_sha16())')"; \
	  echo "# sanitizer symbols referenced: libartn_host_asan.so $$(nm -D $(ASAN_DIR)/libartn_host_asan.so | grep -c '__asan_\|__ubsan_'), libplan_emulate_asan.so $$(nm -D $(ASAN_DIR)/libplan_emulate_asan.so | grep -c '__asan_\|__ubsan_')"; \
	  LD_PRELOAD=$(ASAN_RT) ASAN_OPTIONS=detect_leaks=0:abort_on_error=0:halt_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 \
	  ARTN_LIB=$(ASAN_DIR)/libartn_host_asan.so ARTN_EMU_LIB=$(ASAN_DIR)/libplan_emulate_asan.so \
	  python3 -m pytest tests/test_abi_cpu.py tests/test_plan_emulation.py tests/test_slice_runner.py tests/test_scheme_compilers.py \
	      tests/test_distributed.py tests/test_xgemm_emulation.py tests/test_chain_plan.py -q -m "not gpu" -p no:cacheprovider 2>&1; \
	  LD_PRELOAD=$(ASAN_RT) ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 \
	  ARTN_LIB=$(ASAN_DIR)/libartn_host_asan.so python3 tools/stress_planner.py 3000 0 2>&1 ) | tee $(ASAN_LOG)
	@! grep -E "ERROR: AddressSanitizer|runtime error:" $(ASAN_LOG)

# development build (the tools/ab_*.sh scripts and the switch sweeps need it); `make clean` before switching flavours
dev:
	$(MAKE) DEV=1 all

clean:
	rm -f $(LIB)
	rm -rf $(OBJDIR)
.PHONY: all clean probes stamps phases ablate single asan dev
