// tbz_amd.hip — the single translation unit of lib3bz_amd.so (kernels + host engine + C ABI).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC tbz_amd.hip -o ../lib3bz_amd.so
#include "tbz_engine.hpp"
