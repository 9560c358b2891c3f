# Builds the gfx950 kernel library and the C-ABI in-tree (the .so travels with the tree; it is git-ignored).
HIPCC ?= /opt/rocm/bin/hipcc
ARCH  ?= gfx950
CSRC  := multimae_amd/csrc
OBJ   := build/obj
SRCS  := $(wildcard $(CSRC)/*.hip)
OBJS  := $(patsubst $(CSRC)/%.hip,$(OBJ)/%.o,$(SRCS)) $(OBJ)/error.o $(OBJ)/comm.o
LIB   := multimae_amd/libmmae_hip.so
FLAGS := --offload-arch=$(ARCH) -O3 -fPIC -std=c++17 -Wall -Wno-unused-variable -Wno-unused-but-set-variable

all: $(LIB)

$(OBJ)/%.o: $(CSRC)/%.hip $(CSRC)/common.h $(CSRC)/ln_rows.h include/mmae.h
	@mkdir -p $(OBJ)
	$(HIPCC) $(FLAGS) -c $< -o $@

$(OBJ)/error.o: $(CSRC)/error.cpp include/mmae.h
	@mkdir -p $(OBJ)
	$(HIPCC) $(FLAGS) -c $< -o $@

$(OBJ)/comm.o: $(CSRC)/comm.cpp include/mmae.h
	@mkdir -p $(OBJ)
	$(HIPCC) $(FLAGS) -c $< -o $@

$(LIB): $(OBJS)
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@ $(OBJS) -ldl

clean:
	rm -rf build $(LIB)
.PHONY: all clean
