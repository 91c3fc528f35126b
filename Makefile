# Build of the papr hot path for MI355X (gfx950).  No GPU is needed to build.
#   make            -> dtv-utils_amd/libpaprhip.so, bin/papr, oracle/*
#   make lib | cli | oracle
HIPCC   ?= /opt/rocm/bin/hipcc
CC      ?= gcc
ARCH    ?= gfx950
PKG     := dtv-utils_amd
CSRC    := $(PKG)/csrc
LIB     := $(PKG)/libpaprhip.so
# MEASURE=1 also compiles the kernel geometries / ablations that only the measurement tools under tools/ select
HIPFLAGS := --offload-arch=$(ARCH) -O3 -std=c++17 -fPIC -ffp-contract=off -Iinclude -I$(CSRC) -Wall -Wno-unused-result $(if $(MEASURE),-DPAPR_MEASURE)
CFLAGS  := -O2 -fPIC -ffp-contract=off -Wall -Wextra -Iinclude -I$(CSRC)

all: lib cli oracle tools

lib: $(LIB)

$(CSRC)/papr_host.o: $(CSRC)/papr_host.c $(CSRC)/papr_exact_format.h include/papr_hip.h include/papr_synth.h
	$(CC) $(CFLAGS) -c $< -o $@

$(CSRC)/ts_host.o: $(CSRC)/ts_host.c include/ts_hip.h
	$(CC) $(CFLAGS) -c $< -o $@

$(CSRC)/ts_kernels.o: $(CSRC)/ts_kernels.hip $(CSRC)/ts_kernels.h include/ts_hip.h include/ts_synth.h
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

$(CSRC)/ts_runtime.o: $(CSRC)/ts_runtime.cpp $(CSRC)/ts_kernels.h include/ts_hip.h include/papr_hip.h
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

$(CSRC)/papr_kernels.o: $(CSRC)/papr_kernels.hip $(CSRC)/papr_kernels.h $(CSRC)/papr_device.h $(CSRC)/papr_stream.h include/papr_synth.h
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

$(CSRC)/papr_sweep.o: $(CSRC)/papr_sweep.hip $(CSRC)/papr_kernels.h $(CSRC)/papr_device.h $(CSRC)/papr_stream.h
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

$(CSRC)/papr_exact.o: $(CSRC)/papr_exact.hip $(CSRC)/papr_kernels.h $(CSRC)/papr_device.h
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

RT_HDRS := $(CSRC)/papr_runtime_internal.h $(CSRC)/papr_kernels.h $(CSRC)/papr_exact_format.h include/papr_hip.h include/papr_synth.h
$(CSRC)/papr_runtime.o: $(CSRC)/papr_runtime.cpp $(RT_HDRS)
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

$(CSRC)/papr_ingest.o: $(CSRC)/papr_ingest.cpp $(RT_HDRS)
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

$(CSRC)/papr_sweep_rt.o: $(CSRC)/papr_sweep_rt.cpp $(RT_HDRS)
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

$(CSRC)/papr_exact_rt.o: $(CSRC)/papr_exact_rt.cpp $(RT_HDRS)
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

$(CSRC)/papr_analyze.o: $(CSRC)/papr_analyze.cpp $(RT_HDRS)
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

$(CSRC)/papr_exchange.o: $(CSRC)/papr_exchange.cpp $(RT_HDRS)
	$(HIPCC) $(HIPFLAGS) -I/opt/rocm/include -c $< -o $@

$(LIB): $(CSRC)/papr_kernels.o $(CSRC)/papr_sweep.o $(CSRC)/papr_exact.o $(CSRC)/papr_runtime.o $(CSRC)/papr_ingest.o \
        $(CSRC)/papr_sweep_rt.o $(CSRC)/papr_exact_rt.o $(CSRC)/papr_host.o $(CSRC)/ts_host.o \
        $(CSRC)/ts_kernels.o $(CSRC)/ts_runtime.o $(CSRC)/papr_exchange.o $(CSRC)/papr_analyze.o
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC $^ -o $@ -lm -lpthread -ldl

cli: bin/papr

bin/papr: $(PKG)/host/papr_main.c include/papr_hip.h $(LIB)
	@mkdir -p bin
	$(CC) -O2 -ffp-contract=off -Wall -Wextra -Iinclude $< -o $@ -L$(PKG) -lpaprhip -Wl,-rpath,'$$ORIGIN/../$(PKG)' -lm -lpthread

oracle:
	$(MAKE) -C oracle all

tools: bin/hbm_read_probe bin/ingest_probe bin/work_probe

bin/ingest_probe: tools/ingest_probe.cpp
	@mkdir -p bin
	$(HIPCC) --offload-arch=$(ARCH) -O2 -std=c++17 -Wno-unused-result -Wno-unused-value $< -o $@ -lpthread

bin/hbm_read_probe: tools/hbm_read_probe.hip
	@mkdir -p bin
	$(HIPCC) --offload-arch=$(ARCH) -O3 -std=c++17 -Wno-unused-result -Wno-unused-value $< -o $@

bin/work_probe: tools/work_probe.hip
	@mkdir -p bin
	$(HIPCC) --offload-arch=$(ARCH) -O3 -std=c++17 -ffp-contract=off -Wno-unused-result -Wno-unused-value $< -o $@

clean:
	rm -f $(CSRC)/*.o $(LIB) bin/papr bin/hbm_read_probe bin/ingest_probe bin/work_probe
	$(MAKE) -C oracle clean

.PHONY: all lib cli oracle tools clean
