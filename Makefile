# Builds the engine library without Python: sage_slam_amd/libsage_ba.so (gfx950 only), the same objects, flags and
# link line as `python -m sage_slam_amd.build` (what __graft_entry__.build() runs).  For a C++ maintainer who wires the
# library into system/sources/cuda/CMakeLists.txt as INTEGRATION.md s1 shows:
#   make -j8            the engine library
#   make oracle         the CPU oracle (test infrastructure; oracle/Makefile)
#   make clean
HIPCC   ?= /opt/rocm/bin/hipcc
HOSTCXX ?= /opt/rocm/lib/llvm/bin/clang++
CSRC    := sage_slam_amd/csrc
OBJ     ?= $(CSRC)/_obj
LIB     ?= sage_slam_amd/libsage_ba.so

HIP_SRC  := photo_kernels.hip geo_kernels.hip track_kernels.hip producers.hip keypoint_kernels.hip solve_kernels.hip \
            operators.hip tracker.hip window.hip window_dist.hip window_factors.hip
HOST_SRC := host_math.cpp shard_solve.cpp
HEADERS  := $(addprefix $(CSRC)/,sage_device.h sage_internal.h host_math.h runtime_internal.h finalize_bodies.h) include/sage_ba.h
INC      := -Iinclude -I$(CSRC)
HIPFLAGS := --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function $(INC)
# pure host translation units: ROCm's clang++ without offload (function multiversioning for AVX2 / AVX-512 inside)
HOSTFLAGS := -O3 -std=c++17 -fPIC -Wall $(INC)

# make CHECK=1 OBJ=/tmp/chk_obj LIB=/tmp/libsage_ba_chk.so : libstdc++'s bounds-checked containers in all host code (the `-m gpu`
# suite runs green on such a build: DESIGN s1 "Sanitizer passes"); point the harness at it with SAGE_BA_LIB=<LIB>
ifdef CHECK
HIPFLAGS  += -D_GLIBCXX_ASSERTIONS
HOSTFLAGS += -D_GLIBCXX_ASSERTIONS
endif

OBJS := $(addprefix $(OBJ)/,$(HIP_SRC:.hip=.o) $(HOST_SRC:.cpp=.o))

.PHONY: lib oracle clean
lib: $(LIB)

$(LIB): $(OBJS)
	$(HIPCC) --offload-arch=gfx950 -shared -fPIC -o $@ $(OBJS) -lpthread

$(OBJ)/%.o: $(CSRC)/%.hip $(HEADERS) | $(OBJ)
	$(HIPCC) $(HIPFLAGS) -x hip -c $< -o $@

$(OBJ)/%.o: $(CSRC)/%.cpp $(HEADERS) | $(OBJ)
	$(HOSTCXX) $(HOSTFLAGS) -c $< -o $@

$(OBJ):
	mkdir -p $(OBJ)

oracle:
	$(MAKE) -C oracle

clean:
	rm -rf $(OBJ) $(LIB)
