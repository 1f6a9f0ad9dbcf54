# Builds, in-tree and for gfx950 only:
#   whisper_rs_amd/libwrs_mi355x.so        the product: every entry of include/wrs.h, nothing else; no environment variable changes what it runs
#   whisper_rs_amd/libwrs_mi355x_debug.so  the test build: the SAME objects + debug_api.cpp (include/wrs_debug.h: kernel-level hooks, forced / re-synchronised
#                                          decode, f32 parity mode) and the experiment switches honoured (csrc/common.h: exp_env); loaded by tests and smoke()
HIPCC ?= /opt/rocm/bin/hipcc
ARCH ?= gfx950
CSRC := whisper_rs_amd/csrc
OBJDIR := build/obj
SRCS := $(CSRC)/common.cpp $(CSRC)/gemm.hip $(CSRC)/gemm_pring.hip $(CSRC)/elementwise.hip $(CSRC)/attention.hip $(CSRC)/attn_self_mfma.hip $(CSRC)/logits.hip $(CSRC)/parity_f32.hip $(CSRC)/mel_fused.hip \
        $(CSRC)/mel.cpp $(CSRC)/audio_io.cpp $(CSRC)/tokenizer.cpp $(CSRC)/model.cpp $(CSRC)/weights_io.cpp $(CSRC)/decode.cpp $(CSRC)/transcribe.cpp $(CSRC)/pool.cpp $(CSRC)/capi.cpp
OBJS := $(patsubst $(CSRC)/%,$(OBJDIR)/%.o,$(SRCS))
DBG_OBJS := $(OBJS) $(OBJDIR)/debug_api.cpp.o
HDRS := $(wildcard $(CSRC)/*.h) $(wildcard $(CSRC)/*.inc) include/wrs.h include/wrs_debug.h
CXXFLAGS := -O3 -std=c++17 -fPIC --offload-arch=$(ARCH) -Wall -Wno-unused-function -Wno-unused-variable
LIB := whisper_rs_amd/libwrs_mi355x.so
LIB_DBG := whisper_rs_amd/libwrs_mi355x_debug.so

all: $(LIB) $(LIB_DBG) examples/transcribe

$(OBJDIR)/%.o: $(CSRC)/% $(HDRS)
	@mkdir -p $(OBJDIR)
	$(HIPCC) $(CXXFLAGS) -x hip -c $< -o $@

$(LIB): $(OBJS)
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@ $(OBJS)

$(LIB_DBG): $(DBG_OBJS)
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@ $(DBG_OBJS)

examples/transcribe: examples/transcribe.cpp $(LIB) include/wrs.h include/wrs_debug.h
	g++ -O2 -std=c++17 -Iinclude -o $@ examples/transcribe.cpp -Lwhisper_rs_amd -lwrs_mi355x -Wl,-rpath,'$$ORIGIN/../whisper_rs_amd'

clean:
	rm -rf build $(LIB) $(LIB_DBG) examples/transcribe

.PHONY: all clean
