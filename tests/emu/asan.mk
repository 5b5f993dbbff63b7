# TEST INFRASTRUCTURE ONLY: sanitizer build of the wavefront emulator (CPU only).
#   make -C tests/emu -f asan.mk
#   ASAN_OPTIONS=detect_stack_use_after_return=0:detect_leaks=0 LD_PRELOAD=$(gcc -print-file-name=libasan.so) python ...
CXX ?= g++
SRC = ../../timberborn_support_solver_amd/csrc
FLAGS = -std=c++17 -fPIC -I. -I$(SRC) -Wall -Wno-unused-function -Wno-unused-variable -Wno-unknown-pragmas
libmi355sat_emu_asan.so: emu.cpp hip_shim.h $(SRC)/mi355sat.hip $(SRC)/device/kernels.hip.h
	$(CXX) $(FLAGS) -O1 -g -fsanitize=address,undefined -fno-omit-frame-pointer -shared -o $@ emu.cpp -x c++ $(SRC)/mi355sat.hip
