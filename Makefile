# Builds the gfx950 shared library (the C ABI of include/myraytracer_amd.h), the headless
# CLI, and the CPU oracle.  `python -c "import __graft_entry__ as g; g.build()"` runs this.
HIPCC    ?= /opt/rocm/bin/hipcc
ARCH     ?= gfx950
CSRC     := myraytracer_amd/csrc
LIBDIR   := myraytracer_amd/lib
LIB      := $(LIBDIR)/libmyraytracer_amd.so
CLI      := $(LIBDIR)/native_runner
# -ffp-contract=off: fma only where the source says fma (MRT-F32 rules).
# -fno-vectorize -fno-slp-vectorize: v_pk_* fp32 is not faster than scalar VALU on gfx950
# and SLP packing spends s_mov on SGPR pairs (profiles/r01_ubench_sphere_loop_*.txt).
# -mllvm -amdgpu-mfma-vgpr-form: the matrix-core sweep reads its MFMA results with VALU ops; in AGPRs
# every value would cost a v_accvgpr_read first.
HIPFLAGS := -O3 -std=c++17 --offload-arch=$(ARCH) -fPIC -ffp-contract=off -fno-vectorize -fno-slp-vectorize -mllvm -amdgpu-mfma-vgpr-form -Wall -Wextra -Wno-unused-parameter
SRCS     := $(CSRC)/kernels.hip $(CSRC)/tile_order.hip $(CSRC)/debug_kernels.hip $(CSRC)/api.cpp $(CSRC)/multi_gpu.cpp $(CSRC)/scenes.cpp $(CSRC)/image_io.cpp
HDRS     := $(CSRC)/mrt_internal.h $(CSRC)/mrt_ctx.h $(CSRC)/mrt_device.h $(CSRC)/width_policy.h include/myraytracer_amd.h include/myraytracer_amd_debug.h

all: $(LIB) $(CLI) oracle

# one object per source (build/ is git-ignored), so that touching the host code does not recompile the kernels
OBJDIR   := build/obj
OBJS     := $(patsubst $(CSRC)/%,$(OBJDIR)/%.o,$(SRCS))

$(OBJDIR)/%.o: $(CSRC)/% $(HDRS)
	@mkdir -p $(OBJDIR)
	$(HIPCC) $(HIPFLAGS) -x hip -c -o $@ $<

# the ISA check (scalar-load hazards of the hand-issued s_loads) gates the link
$(OBJDIR)/isa.ok: $(CSRC)/kernels.hip $(HDRS) scripts/check_isa.py
	@mkdir -p $(OBJDIR)
	python3 scripts/check_isa.py
	@touch $@

# mrt_build_id(): the sha256 over the sources (scripts/source_hash.py), regenerated whenever one of them changes
$(OBJDIR)/build_id.cpp: $(SRCS) $(HDRS) Makefile scripts/source_hash.py
	@mkdir -p $(OBJDIR)
	python3 scripts/source_hash.py --cpp > $@

$(LIB): $(OBJS) $(OBJDIR)/isa.ok $(OBJDIR)/build_id.cpp
	@mkdir -p $(LIBDIR)
	$(HIPCC) --offload-arch=$(ARCH) -fPIC -shared -o $@ $(OBJS) $(OBJDIR)/build_id.cpp -ldl

$(CLI): $(CSRC)/native_runner.cpp $(LIB)
	$(HIPCC) -O2 -std=c++17 -o $@ $(CSRC)/native_runner.cpp -L$(LIBDIR) -lmyraytracer_amd -Wl,-rpath,'$$ORIGIN'

oracle:
	$(MAKE) -s -C oracle

clean:
	rm -rf $(LIBDIR) $(OBJDIR)
	$(MAKE) -s -C oracle clean

.PHONY: all oracle clean

# diagnostic build with s_memtime phase stamps (scripts/phase_profile.py); not the product
stamps: $(SRCS) $(HDRS) $(OBJDIR)/build_id.cpp
	@mkdir -p $(LIBDIR)
	$(HIPCC) $(HIPFLAGS) -DMRT_STAMPS -shared -o $(LIBDIR)/libmyraytracer_amd_stamps.so $(SRCS) $(OBJDIR)/build_id.cpp -ldl

# experiment builds (never the product): make variant NAME=slots16 FLAGS="-DMRT_MAX_SLOTS=16" -> lib/libmyraytracer_amd_slots16.so,
# selected at run time by MRT_LIB_OVERRIDE (bench.py refuses a headline from it)
variant: $(SRCS) $(HDRS) $(OBJDIR)/build_id.cpp
	@mkdir -p $(LIBDIR)
	$(HIPCC) $(HIPFLAGS) $(FLAGS) -shared -o $(LIBDIR)/libmyraytracer_amd_$(NAME).so $(SRCS) $(OBJDIR)/build_id.cpp -ldl
