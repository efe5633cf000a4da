# Builds the product library (gfx950 only) and the test oracle.
HIPCC ?= /opt/rocm/bin/hipcc
ARCH ?= gfx950
PKG := umi_collapse_rs_amd
CSRC := $(PKG)/csrc
HIPFLAGS ?= -O3 -std=c++17 -fPIC --offload-arch=$(ARCH) -Wall -Wextra -Wno-unused-parameter

LIB := $(PKG)/libumihip.so
OBJDIR := build/obj
OBJS := $(OBJDIR)/umihip_kernels.o $(OBJDIR)/umihip_seg.o $(OBJDIR)/umihip_collapse.o $(OBJDIR)/umihip_stage.o $(OBJDIR)/umihip_radix.o $(OBJDIR)/umihip_wide.o $(OBJDIR)/umihip_api.o
# development build (make dev): the shipped sources plus the round-1 tile kernels, their key sort
# and their options (-DUMIHIP_DEV), as libumihip_dev.so; the legacy cross-check tests load it
DEVDIR := build/obj_dev
DEVLIB := $(PKG)/libumihip_dev.so
DEVOBJS := $(DEVDIR)/umihip_kernels.o $(DEVDIR)/umihip_seg.o $(DEVDIR)/umihip_collapse.o $(DEVDIR)/umihip_stage.o $(DEVDIR)/umihip_radix.o $(DEVDIR)/umihip_wide.o $(DEVDIR)/umihip_api.o $(DEVDIR)/umihip_legacy.o $(DEVDIR)/umihip_sort.o
HDRS := $(CSRC)/umihip_internal.h $(CSRC)/umihip_device.h $(CSRC)/umihip_plan.hpp include/umihip.h
CLI := $(PKG)/bin/umicollapse

all: $(LIB) oracle cpptest $(CLI)

# one object per translation unit (the rocPRIM sort alone takes ~20 s to compile)
$(OBJDIR)/%.o: $(CSRC)/%.hip $(HDRS)
	@mkdir -p $(OBJDIR)
	$(HIPCC) $(HIPFLAGS) -c -o $@ -x hip $<

$(OBJDIR)/%.o: $(CSRC)/%.cpp $(HDRS)
	@mkdir -p $(OBJDIR)
	$(HIPCC) $(HIPFLAGS) -c -o $@ -x hip $<

$(LIB): $(OBJS)
	$(HIPCC) $(HIPFLAGS) -shared -o $@ $(OBJS) -ldl

$(DEVDIR)/%.o: $(CSRC)/%.hip $(HDRS)
	@mkdir -p $(DEVDIR)
	$(HIPCC) $(HIPFLAGS) -DUMIHIP_DEV -c -o $@ -x hip $<

$(DEVDIR)/%.o: $(CSRC)/%.cpp $(HDRS)
	@mkdir -p $(DEVDIR)
	$(HIPCC) $(HIPFLAGS) -DUMIHIP_DEV -c -o $@ -x hip $<

$(DEVLIB): $(DEVOBJS)
	$(HIPCC) $(HIPFLAGS) -shared -o $@ $(DEVOBJS) -ldl

dev: $(DEVLIB)

$(CLI): $(LIB) $(PKG)/host/umicollapse_main.cpp $(PKG)/host/bam.hpp $(PKG)/host/bgzf.hpp include/umihip.h
	mkdir -p $(PKG)/bin
	g++ -O2 -std=c++17 -Wall -Wextra -o $@ $(PKG)/host/umicollapse_main.cpp -lz -lpthread -ldl

cli: $(CLI)

oracle:
	$(MAKE) -s -C oracle

asm: $(CSRC)/umihip_kernels.hip $(HDRS)
	mkdir -p build
	$(HIPCC) $(HIPFLAGS) -S --cuda-device-only -o build/umihip_kernels.s -x hip $(CSRC)/umihip_kernels.hip \
	    -Rpass-analysis=kernel-resource-usage 2> build/resource_usage.txt || (cat build/resource_usage.txt; false)

cpptest: $(LIB) oracle tests/cpp/test_host.cpp $(PKG)/host/umi_collapse.hpp
	mkdir -p build
	g++ -O2 -std=c++17 -Wall -Wextra -o build/test_host tests/cpp/test_host.cpp \
	    -L$(PKG) -lumihip -Loracle -lumi_oracle -Wl,-rpath,'$$ORIGIN/../$(PKG)' -Wl,-rpath,'$$ORIGIN/../oracle'

clean:
	rm -f $(LIB) $(DEVLIB) $(CLI)
	rm -rf build
	$(MAKE) -s -C oracle clean

.PHONY: all dev oracle asm clean cpptest cli
