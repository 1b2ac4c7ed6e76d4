// TEST INFRASTRUCTURE: the engine + kernels compiled for the CPU lane emulator (see tbz_platform.hpp
// in this directory).  The emulation header is included first, so its include guard wins over the
// gfx950 one the product is built from.
#include "tbz_platform.hpp"
#include "../../3bz_amd/csrc/tbz_engine.hpp"
