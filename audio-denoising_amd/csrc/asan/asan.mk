# ---- host-side sanitizer build (CPU only, never shipped, never run on a GPU box): the host code of every source file
# compiled alone (no device code) with AddressSanitizer + UndefinedBehaviorSanitizer, linked against a stand-in HIP
# runtime (asan/hip_stub.cpp) and driven through the C ABI by asan/asan_driver.cpp.
ASAN_DIR  ?= ../../build/asan
ASAN_OBJS := $(patsubst %.hip,$(ASAN_DIR)/%.o,$(SRCS))
SAN       := -fsanitize=address,undefined -fno-sanitize-recover=undefined -fno-omit-frame-pointer
ASAN_FLAGS := --cuda-host-only -fno-gpu-sanitize $(SAN) -O1 -g -std=c++17 -fPIC -Wall -Wno-unused-function -Wno-unused-variable

$(ASAN_DIR)/%.o: %.hip adn_common.hpp adn_kernels.hpp adn_stft_shared.hpp adn_dwt_shared.hpp adn_dwt_tiles.hpp adn_dwt_mad1p.hpp adn_hpss.hpp adn_sortnet30.hpp ../../include/adn.h
	@mkdir -p $(ASAN_DIR)
	$(HIPCC) $(ASAN_FLAGS) -c -x hip $< -o $@

$(ASAN_DIR)/hip_stub.o: asan/hip_stub.cpp
	@mkdir -p $(ASAN_DIR)
	$(HIPCC) $(ASAN_FLAGS) -c -x hip $< -o $@

# every host object refers to the device image that a full build would embed: empty stand-ins
$(ASAN_DIR)/fatbin_stubs.c: $(ASAN_OBJS)
	nm $(ASAN_OBJS) | awk '$$1 == "U" && $$2 ~ /^__hip_fatbin_/ {print "const char " $$2 "[16] = {0};"}' | sort -u > $@

$(ASAN_DIR)/libadn_hip_asan.so: $(ASAN_OBJS) $(ASAN_DIR)/hip_stub.o $(ASAN_DIR)/fatbin_stubs.c
	/opt/rocm/lib/llvm/bin/clang++ -shared -fPIC $(SAN) $(ASAN_OBJS) $(ASAN_DIR)/hip_stub.o -x c $(ASAN_DIR)/fatbin_stubs.c -o $@

$(ASAN_DIR)/asan_driver: asan/asan_driver.cpp $(ASAN_DIR)/libadn_hip_asan.so ../../include/adn.h
	/opt/rocm/lib/llvm/bin/clang++ $(SAN) -O1 -g -std=c++17 asan/asan_driver.cpp -L$(ASAN_DIR) -ladn_hip_asan -Wl,-rpath,'$$ORIGIN' -o $@

asan: $(ASAN_DIR)/asan_driver
.PHONY: asan

