# Builds libwrs_mi355x.so (gfx950 only) in-tree: whisper_rs_amd/libwrs_mi355x.so
HIPCC ?= /opt/rocm/bin/hipcc
ARCH ?= gfx950
CSRC := whisper_rs_amd/csrc
OBJDIR := build/obj
SRCS := $(CSRC)/common.cpp $(CSRC)/gemm.hip $(CSRC)/gemm_pring.hip $(CSRC)/elementwise.hip $(CSRC)/attention.hip $(CSRC)/attn_self_mfma.hip $(CSRC)/logits.hip $(CSRC)/parity_f32.hip $(CSRC)/mel_fused.hip \
        $(CSRC)/mel.cpp $(CSRC)/audio_io.cpp $(CSRC)/tokenizer.cpp $(CSRC)/model.cpp $(CSRC)/weights_io.cpp $(CSRC)/decode.cpp $(CSRC)/transcribe.cpp $(CSRC)/pool.cpp $(CSRC)/capi.cpp $(CSRC)/debug_api.cpp
OBJS := $(patsubst $(CSRC)/%,$(OBJDIR)/%.o,$(SRCS))
HDRS := $(wildcard $(CSRC)/*.h) $(wildcard $(CSRC)/*.inc) include/wrs.h include/wrs_debug.h
CXXFLAGS := -O3 -std=c++17 -fPIC --offload-arch=$(ARCH) -Wall -Wno-unused-function -Wno-unused-variable
LIB := whisper_rs_amd/libwrs_mi355x.so

all: $(LIB) examples/transcribe

$(OBJDIR)/%.o: $(CSRC)/% $(HDRS)
	@mkdir -p $(OBJDIR)
	$(HIPCC) $(CXXFLAGS) -x hip -c $< -o $@

$(LIB): $(OBJS)
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@ $(OBJS)

examples/transcribe: examples/transcribe.cpp $(LIB) include/wrs.h include/wrs_debug.h
	g++ -O2 -std=c++17 -Iinclude -o $@ examples/transcribe.cpp -Lwhisper_rs_amd -lwrs_mi355x -Wl,-rpath,'$$ORIGIN/../whisper_rs_amd'

clean:
	rm -rf build $(LIB) examples/transcribe

.PHONY: all clean
