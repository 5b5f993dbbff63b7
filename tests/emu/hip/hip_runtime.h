// TEST INFRASTRUCTURE ONLY: stands in for <hip/hip_runtime.h> in the emulator build.
#pragma once
#include "../hip_shim.h"
