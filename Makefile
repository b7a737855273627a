# Top-level build: libparamugsy_amd.so (HIP, gfx950) + bin/m_translate (drop-in CLI).
# `make oracle` builds the test-only CPU oracle (and, where /root/reference exists, oracle/_ref).
HIPCC    ?= hipcc
ARCH     ?= gfx950
HIPFLAGS  = --offload-arch=$(ARCH) -O3 -std=c++17 -fPIC -Wall -Wno-unused-function
CSRC      = paramugsy_amd/csrc
LIB       = paramugsy_amd/libparamugsy_amd.so
LIB_SRCS  = $(CSRC)/pm_common.hip $(CSRC)/translate_job.hip $(CSRC)/translate_host.cc $(CSRC)/side_tools.hip $(CSRC)/profiles_make.hip $(CSRC)/untranslate.hip $(wildcard $(CSRC)/dp_*.hip)
LIB_HDRS  = $(wildcard $(CSRC)/*.hpp) include/paramugsy_amd.h

.PHONY: all lib cli oracle clean

all: lib cli

lib: $(LIB)

# one object per source so that `make -j` compiles them side by side (dp_kernels.hip alone is most of the build)
OBJDIR    = build/obj
LIB_OBJS  = $(patsubst $(CSRC)/%,$(OBJDIR)/%.o,$(LIB_SRCS))

$(OBJDIR)/%.hip.o: $(CSRC)/%.hip $(LIB_HDRS)
	@mkdir -p $(OBJDIR)
	$(HIPCC) $(HIPFLAGS) -c -o $@ $<

$(OBJDIR)/%.cc.o: $(CSRC)/%.cc $(LIB_HDRS)
	@mkdir -p $(OBJDIR)
	$(HIPCC) $(HIPFLAGS) -c -o $@ $<

$(LIB): $(LIB_OBJS)
	$(HIPCC) $(HIPFLAGS) -shared -o $@ $(LIB_OBJS)

cli: bin/m_translate bin/m_sort_delta bin/maf_analyzer bin/mugsy_profiles

bin/mugsy_profiles: $(CSRC)/mugsy_profiles_main.cc $(LIB)
	mkdir -p bin
	$(HIPCC) -O2 -std=c++17 -o $@ $(CSRC)/mugsy_profiles_main.cc -Lparamugsy_amd -lparamugsy_amd -Wl,-rpath,'$$ORIGIN/../paramugsy_amd'

bin/m_sort_delta bin/maf_analyzer: $(CSRC)/side_tools_main.cc $(LIB)
	mkdir -p bin
	$(HIPCC) -O2 -std=c++17 -DPM_TOOL_SORT_DELTA -o bin/m_sort_delta $(CSRC)/side_tools_main.cc -Lparamugsy_amd -lparamugsy_amd -Wl,-rpath,'$$ORIGIN/../paramugsy_amd'
	$(HIPCC) -O2 -std=c++17 -DPM_TOOL_MAF_ANALYZER -o bin/maf_analyzer $(CSRC)/side_tools_main.cc -Lparamugsy_amd -lparamugsy_amd -Wl,-rpath,'$$ORIGIN/../paramugsy_amd'

# the drop-in links nothing of HIP: it asks a resident worker first and loads the library only when nobody listens (m_translate_main.cc)
bin/m_translate: $(CSRC)/m_translate_main.cc $(LIB)
	mkdir -p bin
	g++ -O2 -std=c++17 -Wall -o $@ $(CSRC)/m_translate_main.cc -ldl

oracle:
	$(MAKE) -C oracle oracle
	if [ -d /root/reference/lib ]; then $(MAKE) -C oracle ref; fi

clean:
	rm -rf build
	rm -f $(LIB) bin/m_translate bin/m_sort_delta bin/maf_analyzer bin/mugsy_profiles
	$(MAKE) -C oracle clean
