# Top-level build.  Everything is built in-tree so the artefacts travel with the
# snapshot to the GPU box (they are git-ignored, not gpurun-ignored).
#
#   make lib      libnvk_paged_attn.so: HIP kernels + C ABI, gfx950 only
#   make cli      nano_vllm_amd/nvk_run: the reference's command line over the library
#   make oracle   the CPU checker (oracle/, test infrastructure)
#   make ref      the reference's own headers behind a shim (only where /root/reference exists)
#   make all      all of the above

HIPCC   ?= /opt/rocm/bin/hipcc
ARCH    ?= gfx950
CSRC    := nano_vllm_amd/csrc
LIB     := nano_vllm_amd/libnvk_paged_attn.so
HIPFLAGS := --offload-arch=$(ARCH) -O3 -std=c++17 -fPIC -shared -Iinclude -I$(CSRC) \
            -Wall -Wno-unused-function -fno-gpu-rdc

.PHONY: all lib cli oracle ref clean asm asan
all: lib cli oracle ref

lib: $(LIB)

$(LIB): $(CSRC)/nvk_paged_attn.hip $(CSRC)/nvk_batch.hip $(CSRC)/nvk_linear.hip $(CSRC)/nvk_model.hip $(CSRC)/nvk_shard.hip $(CSRC)/nvk_kernels.hip.h $(CSRC)/nvk_addr.hip.h $(CSRC)/nvk_block_manager.cpp $(CSRC)/nvk_tokenizer.cpp $(CSRC)/nvk_model_group.cpp $(CSRC)/nvk_internal.h $(CSRC)/nvk_device_scope.h include/nvk_paged_attn.h
	$(HIPCC) $(HIPFLAGS) -o $@ $(CSRC)/nvk_paged_attn.hip $(CSRC)/nvk_batch.hip $(CSRC)/nvk_linear.hip $(CSRC)/nvk_model.hip $(CSRC)/nvk_shard.hip $(CSRC)/nvk_block_manager.cpp $(CSRC)/nvk_tokenizer.cpp $(CSRC)/nvk_model_group.cpp -ldl -lpthread

# the reference's command line over the library (tools/nvk_run.cpp): host-only C++ over the C ABI
CLI := nano_vllm_amd/nvk_run
cli: $(CLI)
$(CLI): tools/nvk_run.cpp include/nvk_paged_attn.h $(LIB)
	g++ -std=c++17 -O2 -Wall -Iinclude -o $@ tools/nvk_run.cpp -Lnano_vllm_amd -lnvk_paged_attn -Wl,-rpath,'$$ORIGIN'

# device assembly + register/occupancy report for the kernels (not needed to run)
asm:
	mkdir -p build/asm
	$(HIPCC) --offload-arch=$(ARCH) -O3 -std=c++17 -Iinclude -I$(CSRC) -save-temps=obj -c \
	   -Rpass-analysis=kernel-resource-usage -o build/asm/nvk_paged_attn.o $(CSRC)/nvk_paged_attn.hip 2> build/resource_usage.txt

# Host code under AddressSanitizer + UndefinedBehaviorSanitizer (CPU build only; the pool runs no GPU sanitizer):
# the block allocator, the tokenizer and the CPU checker behind tests/asan_driver.cpp, and the command line
# (its JSON reader and argument handling) as an instrumented executable over the normal library.
# tests/test_sanitizers.py builds and runs them on traces, random texts, and short / damaged input files.
SAN := -fsanitize=address,undefined -fno-sanitize-recover=undefined -fno-omit-frame-pointer -g -O1 -static-libasan
asan:
	mkdir -p build/asan
	gcc $(SAN) -std=c11 -c -o build/asan/nvk_oracle.o oracle/nvk_oracle.c
	g++ $(SAN) -static-libubsan -std=c++17 -Wall -Iinclude -I$(CSRC) -o build/asan/asan_driver tests/asan_driver.cpp tests/asan_stubs.cpp \
	    $(CSRC)/nvk_block_manager.cpp $(CSRC)/nvk_tokenizer.cpp build/asan/nvk_oracle.o -lpthread -lm
	g++ $(SAN) -static-libubsan -std=c++17 -Wall -Iinclude -o build/asan/nvk_run_asan tools/nvk_run.cpp -Lnano_vllm_amd -lnvk_paged_attn \
	    -Wl,-rpath,'$$ORIGIN/../../nano_vllm_amd'

oracle:
	$(MAKE) -C oracle oracle

ref:
	$(MAKE) -C oracle ref

clean:
	rm -f $(LIB) $(CLI)
	rm -rf build
	$(MAKE) -C oracle clean
