# Build of the papr hot path for MI355X (gfx950).  No GPU is needed to build.
#   make            -> dtv-utils_amd/libpaprhip.so, bin/papr, oracle/*
#   make lib | cli | oracle
HIPCC   ?= /opt/rocm/bin/hipcc
CC      ?= gcc
ARCH    ?= gfx950
PKG     := dtv-utils_amd
CSRC    := $(PKG)/csrc
# MEASURE=1 also compiles the kernel geometries / ablations that only the measurement tools under tools/ select
HIPFLAGS := --offload-arch=$(ARCH) -O3 -std=c++17 -fPIC -ffp-contract=off -Iinclude -I$(CSRC) -Wall -Wno-unused-result $(if $(MEASURE),-DPAPR_MEASURE)
# (objects of the two flavours are kept apart: foo.o / foo.m.o)
X       := $(if $(MEASURE),.m.o,.o)
LIB     := $(if $(MEASURE),build_measure/libpaprhip.so,$(PKG)/libpaprhip.so)
CFLAGS  := -O2 -fPIC -ffp-contract=off -Wall -Wextra -Iinclude -I$(CSRC)

all: lib cli oracle tools

lib: $(LIB)

$(CSRC)/papr_host$(X): $(CSRC)/papr_host.c $(CSRC)/papr_exact_format.h include/papr_hip.h include/papr_hip_measure.h include/papr_synth.h
	$(CC) $(CFLAGS) -c $< -o $@

$(CSRC)/ts_host$(X): $(CSRC)/ts_host.c $(CSRC)/ts_walk_core.h include/ts_hip.h
	$(CC) $(CFLAGS) -c $< -o $@

$(CSRC)/ts_kernels$(X): $(CSRC)/ts_kernels.hip $(CSRC)/ts_kernels.h $(CSRC)/ts_walk_core.h include/ts_hip.h include/ts_synth.h
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

$(CSRC)/ts_runtime$(X): $(CSRC)/ts_runtime.cpp $(CSRC)/ts_kernels.h $(CSRC)/ts_line_pool.h include/ts_hip.h include/papr_hip.h
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

$(CSRC)/papr_kernels$(X): $(CSRC)/papr_kernels.hip $(CSRC)/papr_kernels.h $(CSRC)/papr_device.h $(CSRC)/papr_stream.h include/papr_synth.h
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

$(CSRC)/papr_sweep$(X): $(CSRC)/papr_sweep.hip $(CSRC)/papr_sweep_dev.h $(CSRC)/papr_skew_walk.h $(CSRC)/papr_kernels.h $(CSRC)/papr_device.h $(CSRC)/papr_stream.h
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

# the laboratory: every other kernel form of the sweep (geometries, stash forms, ablations) — `make MEASURE=1` only
$(CSRC)/measure/papr_sweep_lab$(X): $(CSRC)/measure/papr_sweep_lab.hip $(CSRC)/papr_sweep_dev.h $(CSRC)/papr_skew_walk.h $(CSRC)/papr_kernels.h $(CSRC)/papr_device.h $(CSRC)/papr_stream.h
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

$(CSRC)/papr_exact$(X): $(CSRC)/papr_exact.hip $(CSRC)/papr_exact_format.h $(CSRC)/papr_kernels.h $(CSRC)/papr_device.h $(CSRC)/papr_stream.h
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

RT_HDRS := $(CSRC)/papr_runtime_internal.h $(CSRC)/papr_kernels.h $(CSRC)/papr_exact_format.h include/papr_hip.h include/papr_exchange.h include/papr_hip_measure.h include/papr_synth.h
$(CSRC)/papr_runtime$(X): $(CSRC)/papr_runtime.cpp $(RT_HDRS)
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

$(CSRC)/papr_ingest$(X): $(CSRC)/papr_ingest.cpp $(RT_HDRS)
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

$(CSRC)/papr_sweep_rt$(X): $(CSRC)/papr_sweep_rt.cpp $(RT_HDRS)
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

$(CSRC)/papr_exact_rt$(X): $(CSRC)/papr_exact_rt.cpp $(RT_HDRS)
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

$(CSRC)/papr_analyze$(X): $(CSRC)/papr_analyze.cpp $(RT_HDRS)
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

$(CSRC)/papr_exchange$(X): $(CSRC)/papr_exchange.cpp $(RT_HDRS)
	$(HIPCC) $(HIPFLAGS) -I/opt/rocm/include -c $< -o $@

$(LIB): | build_measure
$(LIB): $(CSRC)/papr_kernels$(X) $(CSRC)/papr_sweep$(X) $(CSRC)/papr_exact$(X) $(CSRC)/papr_runtime$(X) $(CSRC)/papr_ingest$(X) \
        $(CSRC)/papr_sweep_rt$(X) $(CSRC)/papr_exact_rt$(X) $(CSRC)/papr_host$(X) $(CSRC)/ts_host$(X) \
        $(CSRC)/ts_kernels$(X) $(CSRC)/ts_runtime$(X) $(CSRC)/papr_exchange$(X) $(CSRC)/papr_analyze$(X) \
        $(if $(MEASURE),$(CSRC)/measure/papr_sweep_lab$(X))
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC $^ -o $@ -lm -lpthread -ldl

cli: bin/papr

bin/papr: $(PKG)/host/papr_main.c include/papr_hip.h include/papr_exchange.h include/papr_hip_measure.h $(LIB)
	@mkdir -p bin
	$(CC) -O2 -ffp-contract=off -Wall -Wextra -Iinclude $< -o $@ -L$(PKG) -lpaprhip -Wl,-rpath,'$$ORIGIN/../$(PKG)' -lm -lpthread

oracle:
	$(MAKE) -C oracle all

tools: bin/hbm_read_probe bin/ingest_probe bin/work_probe bin/stride_read_probe bin/h2d_contention_probe bin/zero_copy_probe bin/wg_skew_probe bin/ts_stride_probe bin/xcd_affinity_probe bin/exact_form_probe

bin/stride_read_probe: tools/stride_read_probe.hip
	@mkdir -p bin
	$(HIPCC) --offload-arch=$(ARCH) -O3 -std=c++17 -Wno-unused-result -Wno-unused-value $< -o $@

bin/exact_form_probe: tools/exact_form_probe.hip
	@mkdir -p bin
	$(HIPCC) --offload-arch=$(ARCH) -O3 -std=c++17 -ffp-contract=off -Wno-unused-result -Wno-unused-value $< -o $@

bin/xcd_affinity_probe: tools/xcd_affinity_probe.hip
	@mkdir -p bin
	$(HIPCC) --offload-arch=$(ARCH) -O3 -std=c++17 -Wno-unused-result -Wno-unused-value $< -o $@

bin/ts_stride_probe: tools/ts_stride_probe.hip
	@mkdir -p bin
	$(HIPCC) --offload-arch=$(ARCH) -O3 -std=c++17 -Wno-unused-result -Wno-unused-value $< -o $@

bin/wg_skew_probe: tools/wg_skew_probe.hip
	@mkdir -p bin
	$(HIPCC) --offload-arch=$(ARCH) -O3 -std=c++17 -Wno-unused-result -Wno-unused-value $< -o $@

bin/ingest_probe: tools/ingest_probe.cpp
	@mkdir -p bin
	$(HIPCC) --offload-arch=$(ARCH) -O2 -std=c++17 -Wno-unused-result -Wno-unused-value $< -o $@ -lpthread

bin/zero_copy_probe: tools/zero_copy_probe.hip
	@mkdir -p bin
	$(HIPCC) --offload-arch=$(ARCH) -O3 -std=c++17 -Wno-unused-result -Wno-unused-value $< -o $@

bin/h2d_contention_probe: tools/h2d_contention_probe.cpp
	@mkdir -p bin
	$(HIPCC) --offload-arch=$(ARCH) -O2 -mavx2 -std=c++17 -Wno-unused-result -Wno-unused-value $< -o $@ -lpthread

bin/hbm_read_probe: tools/hbm_read_probe.hip
	@mkdir -p bin
	$(HIPCC) --offload-arch=$(ARCH) -O3 -std=c++17 -Wno-unused-result -Wno-unused-value $< -o $@

bin/work_probe: tools/work_probe.hip
	@mkdir -p bin
	$(HIPCC) --offload-arch=$(ARCH) -O3 -std=c++17 -ffp-contract=off -Wno-unused-result -Wno-unused-value $< -o $@

clean:
	rm -f $(CSRC)/*.o $(CSRC)/measure/*.o $(PKG)/libpaprhip.so build_measure/libpaprhip.so bin/papr bin/hbm_read_probe bin/ingest_probe bin/work_probe bin/stride_read_probe bin/h2d_contention_probe bin/zero_copy_probe bin/wg_skew_probe bin/ts_stride_probe bin/xcd_affinity_probe bin/exact_form_probe
	$(MAKE) -C oracle clean

.PHONY: all lib cli oracle tools clean

build_measure:
	@mkdir -p build_measure
