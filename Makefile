# Build of the native pieces.  Everything is compiled with -ffp-contract=off:
# the oracle (gcc) and the gfx950 kernels (hipcc) must round every expression
# at the same place (DESIGN.md "Numerical contract").
#
#   make host     -> mort_amd/lib/libmort_host.so   (C scene layer, no GPU)
#   make hip      -> mort_amd/lib/libmort_hip.so    (C-ABI + gfx950 kernels)
#   make oracle   -> oracle/libmort_oracle.so       (CPU checker, test-only)
#   make cli      -> mort_amd/bin/mort              (the `mort <scene_id>` CLI)
#   make all

ROCM ?= /opt/rocm
HIPCC ?= $(ROCM)/bin/hipcc
CC ?= gcc
ARCH ?= gfx950

# diagnostic builds (cycle counters, region marks) may spill: make hip RESCHECK= ...
RESCHECK ?= --check
LIBDIR := mort_amd/lib
BINDIR := mort_amd/bin
INC := -Iinclude

CFLAGS := -O2 -std=gnu11 -fPIC -Wall -Wextra -Wno-unused-parameter -ffp-contract=off -fno-fast-math $(INC)
# -fno-slp-vectorize: hipcc 7.2's SLP vectorizer miscompiles dev_shade.h's shade_hit when it is inlined into wf_shade_gen / mega_gen_kernel
# (found with -opt-bisect-limit: the first bad pass is SLPVectorizerPass on the kernel), and the kernels are 2-5 % faster
# without its packed-register shuffling
HIPFLAGS := -O3 -std=c++17 -fPIC --offload-arch=$(ARCH) -ffp-contract=off -fno-fast-math -fno-slp-vectorize \
            -fno-gpu-rdc -Wall -Wno-unused-parameter -Wno-unused-value -Wno-unused-result $(INC) $(HIPFLAGS_EXTRA)

HOST_SRC := mort_amd/csrc/host/mort_host.c mort_amd/csrc/host/mort_scenes.c mort_amd/csrc/host/mort_jpeg.c
HIP_SRC := $(wildcard mort_amd/csrc/hip/*.hip)
HIP_HDR := $(wildcard mort_amd/csrc/hip/*.h) $(wildcard include/*.h)

.PHONY: all host hip oracle cli clean resources
all: host oracle hip cli

host: $(LIBDIR)/libmort_host.so
$(LIBDIR)/libmort_host.so: $(HOST_SRC) $(wildcard include/*.h) mort_amd/csrc/host/mort_vec.h
	@mkdir -p $(LIBDIR)
	$(CC) $(CFLAGS) -shared -o $@ $(HOST_SRC) -lm

# one object per .hip translation unit (no relocatable device code: every kernel lives in the TU that launches it)
OBJDIR := build/hip
HIP_OBJ := $(patsubst mort_amd/csrc/hip/%.hip,$(OBJDIR)/%.o,$(HIP_SRC))

hip: $(LIBDIR)/libmort_hip.so
$(OBJDIR)/%.o: mort_amd/csrc/hip/%.hip $(HIP_HDR)
	@mkdir -p $(OBJDIR)
	$(HIPCC) $(HIPFLAGS) -c -o $@ $<
$(LIBDIR)/libmort_hip.so: $(HIP_OBJ)
	@mkdir -p $(LIBDIR)
	$(HIPCC) --offload-arch=$(ARCH) -fno-gpu-rdc -shared -o $@ $(HIP_OBJ) -lpthread -ldl
	@python3 scripts/kernel_resources.py $@ $(RESCHECK) > $(OBJDIR)/kernel_resources.txt || { cat $(OBJDIR)/kernel_resources.txt; rm -f $@; exit 1; }

# per-kernel registers / spills / private memory / LDS as the code objects state them (also checked by `make hip`)
resources: $(LIBDIR)/libmort_hip.so
	python3 scripts/kernel_resources.py $(LIBDIR)/libmort_hip.so

oracle:
	$(MAKE) -C oracle all ref

cli: $(BINDIR)/mort
$(BINDIR)/mort: mort_amd/csrc/cli/mort.c $(LIBDIR)/libmort_host.so $(LIBDIR)/libmort_hip.so
	@mkdir -p $(BINDIR)
	$(CC) $(CFLAGS) -fPIE -o $@ mort_amd/csrc/cli/mort.c -L$(LIBDIR) -lmort_host -lmort_hip \
	    -Wl,-rpath,'$$ORIGIN/../lib' -lm -lpthread

clean:
	rm -rf $(LIBDIR) $(BINDIR) $(OBJDIR)
	$(MAKE) -C oracle clean
