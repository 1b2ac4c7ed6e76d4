#!/bin/bash
# Experiment build of the engine with in-kernel time stamps and trip counters (-DTBZ_WAVE_TRACE): 3bz_amd/lib3bz_trace.so.
# The product library (3bz_amd/lib3bz_amd.so, __graft_entry__.build()) carries none of it.
cd "$(dirname "$0")/../../3bz_amd/csrc" && hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -DTBZ_WAVE_TRACE tbz_amd.hip -o ../lib3bz_trace.so
